# Collects the per-round evidence on the GPU box: bench line, kernel-trace summary, the two PMC traffic passes, the MFMA
# utilisation pass and the per-shape table.  usage: bash tools/collect_profiles.sh   (then copy gpurun_out/g/* to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/g
# the profiled command is the HEADLINE workload's canonical steps and nothing else (DEPGAN_BENCH_STEP_ONLY: no
# generator-forward / generator-iteration probes, no extra engines): per-kernel averages and PMC sums then cover exactly
# the launches bench.py's `roofline` describes, and the class launch count is steps x launches-per-step
export DEPGAN_BENCH_STEP_ONLY=1
DEPGAN_PROFILE_DUMP=gpurun_out/g/launches.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g/bench_short.json 2> gpurun_out/g/bench.err
python3 tools/layer_table.py gpurun_out/g/launches.csv 2 > gpurun_out/g/layer_table.md
echo dump done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/g/kt -o r --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g/kt.log 2>&1
echo kt done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/g/pf -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g/pf.log 2>&1
echo pf done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/g/pw -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g/pw.log 2>&1
echo pw done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/g/mu -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g/mu.log 2>&1
echo mu done
python3 tools/pmc_traffic.py gpurun_out/g/pf gpurun_out/g/pw gpurun_out/g/pmc_traffic.json
python3 tools/pmc_mfma_util.py gpurun_out/g/mu gpurun_out/g/mfma_util.json
cp $(find gpurun_out/g/kt -name "*kernel_stats.csv" | head -1) gpurun_out/g/kernel_stats.csv
rm -rf gpurun_out/g/kt gpurun_out/g/pf gpurun_out/g/pw gpurun_out/g/mu
unset DEPGAN_BENCH_STEP_ONLY
python bench.py > gpurun_out/g/bench.json 2>> gpurun_out/g/bench.err
echo bench done
tail -c 600 gpurun_out/g/bench.json

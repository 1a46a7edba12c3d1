# Collects the per-round evidence on the GPU box: bench line, kernel-trace summary, the two PMC traffic passes.
# usage: bash tools/collect_profiles.sh   (then copy gpurun_out/g/{bench.json,kernel_stats.csv,pmc_traffic.json} to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/g
python bench.py > gpurun_out/g/bench.json 2> gpurun_out/g/bench.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/g/kt -o r --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g/kt.log 2>&1
echo kt done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/g/pf -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g/pf.log 2>&1
echo pf done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/g/pw -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g/pw.log 2>&1
echo pw done
find gpurun_out/g -name "*.csv" | head -20
python3 tools/pmc_traffic.py gpurun_out/g/pf gpurun_out/g/pw gpurun_out/g/pmc_traffic.json
cp $(find gpurun_out/g/kt -name "*kernel_stats.csv" | head -1) gpurun_out/g/kernel_stats.csv
tail -1 gpurun_out/g/bench.json

#!/bin/bash
# Development aid: SQ / LDS counter passes over one Winograd conv launch shape.  usage: bash tools/pmc_wino.sh "32 128 128 64 64 3"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SH="$1"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_MEM_VIOLATIONS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  CONV_PATH=8 timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/wq$i --output-format csv -- python3 tools/pmc_one_conv.py $SH > gpurun_out/wq$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/wq$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for d in ("gpurun_out/wq1", "gpurun_out/wq2", "gpurun_out/wq3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino_conv" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Counter_Name"], r["Dispatch_Id"])] += 1
launches = len({k[1] for k in n if k[0] == "SQ_WAVES"}) or 1
waves = acc["SQ_WAVES"] / launches
print("launches %d, waves per launch %.0f" % (launches, waves))
for k in sorted(acc):
    print("%-28s per launch %14.0f   per wave %10.1f" % (k, acc[k] / launches, acc[k] / launches / waves))
PY

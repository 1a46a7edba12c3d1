#!/bin/bash
# Development aid: raw SQ / SQC counters of the MFMA conv kernel for one shape (tools/pmc_one_conv.py), averaged over
# its dispatches, one rocprofv3 pass per counter group.  usage: bash tools/pmc_conv_raw.sh "32 256 256 32 32 3" [tag]
# (CONV_PATH=3 in the environment selects the bf16 kernel, 7 the wave-private one: depgan_op_conv2d's path argument)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SH="$1"; TAG="${2:-raw}"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
G4="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VMEM_RD_DATA_FIFO_FULL SQ_VMEM_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_ANY"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/${TAG}$i --output-format csv -- python3 tools/pmc_one_conv.py $SH > gpurun_out/${TAG}$i.log 2>&1 || echo "pass $i failed (see gpurun_out/${TAG}$i.log)"
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for i in (1, 2, 3, 4):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/%s%d/**/*counter_collection.csv" % (tag, i), recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm_conv_kernel" in r["Kernel_Name"] or "igemm_bf16_kernel" in r["Kernel_Name"] or "igemm_wp_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(acc):
        print("pass %d  %-32s per dispatch %16.0f   (%d dispatches)" % (i, k, acc[k] / max(len(disp[k]), 1), len(disp[k])))
PY

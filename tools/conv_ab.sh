#!/bin/bash
# Development aid: kernel-only duration of igemm_conv_kernel for several builds of the library on the same box.
#   tools/wgrad_ab.sh "<shape>" lib1.so lib2.so ...      (shape = "B H W Cin Cout K")
cd /tmp && export TMPDIR=/tmp
shape="$1"; shift
for lib in "$@"; do
  d=/tmp/cab_$$_$(basename $lib .so); mkdir -p $d
  DEPGAN_LIB=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $d -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_one_conv.py $shape > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$lib" "$shape" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "igemm_conv_kernel" in r["Name"]:
        print("%-24s %-22s calls %s avg %.1f us min %.1f us" % (sys.argv[2].split("/")[-1], sys.argv[3], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done

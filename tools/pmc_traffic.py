"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_c_pmc_traffic.json

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: counters are in KiB;
FETCH_SIZE tallies the 128-byte requests of 16-B-per-lane streaming reads at 64 B, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  adam_kernel (pure 16-B streaming over arenas of known size) is the calibration row.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# kernels of bench.py's convolution class (class 0 of the library's profile records)
CLASS0 = ("igemm_conv_kernel", "igemm_conv_head_kernel", "wino_conv_kernel", "wino_conv_head_kernel", "deconv_fwd_kernel")


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    for f in files:
        with open(f) as fh:
            per_dispatch = defaultdict(float)
            names = {}
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                key = (r.get("Process_Id", ""), r["Dispatch_Id"])
                per_dispatch[key] += float(r["Counter_Value"])      # one row per XCD / dimension: sum them
                names[key] = r["Kernel_Name"]
            for key, v in per_dispatch.items():
                a = acc[names[key]]
                a[0] += 1
                a[1] += v
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows = {}
    for name in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(name, [0, 0.0])
        nw, sw = write.get(name, [0, 0.0])
        rd = 2.0 * 1024.0 * sf / nf if nf else None        # KiB -> B, x2 gfx950 correction
        wr = 1024.0 * sw / nw if nw else None
        rows[name] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                      "hbm_bytes_per_launch": (rd or 0.0) + (wr or 0.0)}
    cls = [v for k, v in rows.items() if any(n in k for n in CLASS0)]   # bench.py's class 0
    n = sum(v["launches"] for v in cls)
    summary = {"igemm_conv_kernel_class": {
        "launches": n,
        "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in cls) / max(n, 1),
        "read_bytes_per_launch": sum((v["read_bytes_per_launch"] or 0) * v["launches"] for v in cls) / max(n, 1),
        "write_bytes_per_launch": sum((v["write_bytes_per_launch"] or 0) * v["launches"] for v in cls) / max(n, 1)},
        # set by tools/collect_profiles.sh: the passes wrapped `DEPGAN_BENCH_STEP_ONLY=1 bench.py` (canonical steps only), so
        # `launches` is steps x launches-per-step and bench.py can check that it divides
        "step_only": bool(os.environ.get("DEPGAN_BENCH_STEP_ONLY")),
        "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE KiB x1024",
        "kernels": rows}
    with open(out, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary["igemm_conv_kernel_class"]))
    for k in rows:
        if "adam" in k:
            print("calibration", k, rows[k])


if __name__ == "__main__":
    main()

"""Fold one rocprofv3 pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace) into MFMA utilisation.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/g/mu --output-format csv \
        -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma_util.py gpurun_out/g/mu profiles/r01_g_mfma_util.json

Per dispatch: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (64 cycles per v_mfma_f32_32x32x2_f32,
32 per 16x16x4); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the dispatch lasted GUI_ACTIVE / 8 shader cycles
(/opt/skills/guides/MI355X_MICROARCH.md, DVFS note) and
    mfma_util = MFMA_BUSY / (1024 * GUI_ACTIVE / 8),      clock = GUI_ACTIVE / 8 / duration.
The fp32 MFMA roofline fraction at the nominal 2.4 GHz is mfma_util * clock / 2.4 GHz.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r.get("Process_Id", ""), r["Dispatch_Id"])
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            names[key] = r["Kernel_Name"]
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                per[key]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not per:
        raise SystemExit("no counter_collection.csv under " + d)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(r.get("Process_Id", r.get("Pid", "")), r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for key, c in per.items():
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        ns = c.get("_ns") or dur.get(key) or 0.0
        if gui <= 0 or busy <= 0:
            continue
        a = acc[names[key]]
        a[0] += 1
        a[1] += busy
        a[2] += gui
        a[3] += ns
    rows = {}
    for k, (n, busy, gui, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        cyc = gui / 8.0
        rows[k] = {"launches": n, "mfma_util": busy / (1024.0 * cyc), "clock_ghz": (cyc / ns) if ns else None,
                   "avg_us": ns / n / 1e3 if ns else None}
    def fold(*subs):
        sel = [(k, acc[k]) for k in acc if any(sub in k for sub in subs)]
        busy = sum(v[1] for _, v in sel); gui = sum(v[2] for _, v in sel); ns = sum(v[3] for _, v in sel)
        return {"launches": sum(v[0] for _, v in sel), "mfma_util": busy / (1024.0 * gui / 8.0) if gui else None,
                "clock_ghz": gui / 8.0 / ns if ns else None}
    summary = {"igemm_conv_kernel_class": fold("igemm_conv_kernel", "igemm_conv_head_kernel", "wino_conv_kernel", "wino_conv_head_kernel", "deconv_fwd_kernel"),     # bench.py's class 0
               "wgrad_dma_kernel_class": fold("wgrad_dma_kernel", "deconv_wgrad_kernel"),    # ... and class 1
               "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); clock = GUI_ACTIVE / 8 / duration",
               "kernels": rows}
    with open(out, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps({k: summary[k] for k in ("igemm_conv_kernel_class", "wgrad_dma_kernel_class")}))


if __name__ == "__main__":
    main()

"""Prints the headline and the per-config summaries of a bench.py result file.  usage: python tools/show_bench.py [file]   (no file: stdin)"""
import json
import sys

d = None
for ln in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if ln.startswith("{"):
        d = json.loads(ln)
print("headline: %.1f slices/s, %.3f ms per step" % (d["value"], d["ms_per_step"]))
r = d.get("roofline")
if r:
    print("conv class %.2f TF (%.4f), traffic %s vs algorithmic %s; classes %s; g_forward %s" % (
        r["achieved"], r["frac"], r.get("traffic"), r.get("algorithmic_bytes_per_launch"), r.get("ms_per_step"),
        r.get("g_forward")))
    dk = r.get("dominant_kernel")
    if dk:
        print("dominant:", dk["kernel"], dk["avg_launch_us"], "us", dk["frac"], "| largest shape", dk["largest_shape"]["shape"],
              dk["largest_shape"]["frac"])
for k in ("config4", "config5"):
    if k in d:
        c = d[k]
        print(k, c["ms_per_step"], "ms", c["slices_per_s"], "slices/s", c["ms_per_step_by_class"],
              c.get("roofline", {}).get("frac"))
if "f32_split" in d:
    for k, v in d["f32_split"].items():
        if isinstance(v, dict):
            print("f32_split", k, v["ms_per_step"], "ms", v["slices_per_s"], v["roofline"]["frac"])
if "direct_conv" in d:
    c = d["direct_conv"]
    print("direct_conv", c["ms_per_step"], "ms", c["slices_per_s"], "slices/s conv", c["conv_class"], "g_forward", c["g_forward"])
if r and r.get("dominant_kernel"):
    print("class issued-MFMA frac", r["dominant_kernel"].get("class_mfma_issued_frac"), "by kernel", r["dominant_kernel"].get("class_by_kernel"))
if "collectives" in d:
    print("collectives", d["collectives"], d.get("scaling_efficiency"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])

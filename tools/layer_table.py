"""Per-shape roofline table from the per-launch CSV that `DEPGAN_PROFILE_DUMP=<path> python bench.py` writes
(depgan_profile_dump: class,label,ms,gflop for every launch of the two profiled steps; HIP events on the engine's
stream).  The rocprofv3 kernel names merge all shapes of a template instantiation; this table is the per-layer view.

usage: python tools/layer_table.py launches.csv [steps=2] > profiles/rNN_x_layer_table.md"""
import collections
import csv
import sys

PEAK = 157.3
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
agg = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    k = (int(r["class"]), r["label"] or "(unlabelled HBM-bound kernels)")
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["ms"])
    a[2] += float(r["gflop"])
tot = sum(a[1] for a in agg.values())
names = {0: "MFMA conv", 1: "MFMA wgrad", 2: "other"}
print("| class | shape | launches/step | ms/step | TFLOP/s | frac of %.1f | %% of step |" % PEAK)
print("|---|---|---|---|---|---|---|")
for (k, l), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = a[2] / a[1] if a[1] else 0.0
    print("| %s | %s | %.1f | %.3f | %s | %s | %.1f |" % (names[k], l, a[0] / steps, a[1] / steps,
                                                       ("%.1f" % tf) if a[2] else "-",
                                                       ("%.3f" % (tf / PEAK)) if a[2] and k != 2 else "-",
                                                       100 * a[1] / tot))
for k in (0, 1, 2):
    ms = sum(a[1] for (kk, _), a in agg.items() if kk == k)
    gf = sum(a[2] for (kk, _), a in agg.items() if kk == k)
    print("| **%s total** | | %.1f | %.3f | %s | %s | %.1f |" % (
        names[k], sum(a[0] for (kk, _), a in agg.items() if kk == k) / steps, ms / steps,
        ("%.1f" % (gf / ms)) if k != 2 else "-", ("%.3f" % (gf / ms / PEAK)) if k != 2 else "-", 100 * ms / tot))

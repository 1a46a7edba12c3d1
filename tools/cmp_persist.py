"""Development aid: per-layer step timing with and without persistent conv workgroups, same box (run by tools/cmp_persist.sh)."""
import re, sys
def load(f):
    d = {}
    for l in open(f):
        m = re.match(r"(c\d .*?)\s+n/step\s+([\d.]+)\s+([\d.]+) ms/step\s+([\d.]+) TF/s", l)
        if m: d[m.group(1).strip()] = (float(m.group(3)), float(m.group(4)))
        if l.startswith("total"): d["total"] = (float(l.split()[1]), 0)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
print("total A %.2f  B %.2f" % (a["total"][0], b["total"][0]))
rows = sorted(((b[k][0] - a[k][0], k) for k in a if k in b and k.startswith("c0")), reverse=True)
for d, k in rows[:6] + rows[-3:]:
    print("A gains %+.3f ms  %-36s B %.3f -> A %.3f  (%.1f TF/s)" % (d, k, b[k][0], a[k][0], a[k][1]))

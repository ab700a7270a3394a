"""Per-kernel time per step of two rocprofv3 --stats runs (directories with *kernel_stats.csv), sorted by the difference."""
import csv, glob, sys
def load(d, steps):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        out[r["Name"]] = (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps)
    return out
a, b = load(sys.argv[1], int(sys.argv[3])), load(sys.argv[2], int(sys.argv[3]))
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort()
print(f"total ms/step: {sum(v[1] for v in a.values()):.2f} -> {sum(v[1] for v in b.values()):.2f}; launches/step {sum(v[0] for v in a.values()):.0f} -> {sum(v[0] for v in b.values()):.0f}")
for d, k, ca, ta, cb, tb in rows[:25] + rows[-25:]:
    if abs(d) > 0.02:
        print(f"{d:+8.3f} ms  {ca:7.1f} x {ta:8.3f} -> {cb:7.1f} x {tb:8.3f}   {k[:140]}")

"""Per-kernel average durations of two rocprofv3 kernel-stats CSVs side by side: python tools/kcmp.py A.csv B.csv"""
import csv, sys
def load(p):
    return {r["Name"][:60]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b and min(a[k][0], b[k][0]) >= 20:
        print("%-60s %4d  %8.1f %8.1f  %+6.1f%%" % (k, a[k][0], a[k][1], b[k][1], (b[k][1] / a[k][1] - 1) * 100))
        ta += a[k][0] * a[k][1]; tb += b[k][0] * b[k][1]
print("total of common kernels: %.0f -> %.0f us (%+.1f%%)" % (ta, tb, (tb / ta - 1) * 100))

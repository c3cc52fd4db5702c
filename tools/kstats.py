"""Print per-kernel average durations (us) from a rocprofv3 *_kernel_stats.csv: python tools/kstats.py file.csv [substr]"""
import csv, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if pat in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Percentage"]}%')

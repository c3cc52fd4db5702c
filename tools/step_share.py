"""Per-kernel share of IN-STEP kernel time from a rocprofv3 kernel trace of `bench.py` (tools/prof_step.sh): every dispatch up
to and including the last adamw_kernel belongs to the train steps (warm-up + timed); what follows are bench.py's roofline legs
and are left out.   python tools/step_share.py kt_kernel_trace.csv out.json
out.json: {"steps": n, "sum_kernel_us_per_step": ..., "kernels": [{"kernel", "calls_per_step", "avg_us", "us_per_step", "share"}]}
bench.py reads it (profiles/r03_step_share.json) for roofline.step_share."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
step_rows = rows[:ends[-1] + 1]
nsteps = len(ends)
agg = collections.OrderedDict()
for r in step_rows:
    if "at::native" in r["Kernel_Name"] or r["Kernel_Name"].startswith("__amd_rocclr"):
        continue   # torch's RNG / fill kernels of input generation, runtime copies
    d = agg.setdefault(r["Kernel_Name"], [0, 0])
    d[0] += 1
    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
ks = [{"kernel": k, "calls_per_step": round(v[0] / nsteps, 2), "avg_us": round(v[1] / v[0] / 1e3, 2),
       "us_per_step": round(v[1] / nsteps / 1e3, 2), "share": round(v[1] / tot, 4)} for k, v in agg.items()]
ks.sort(key=lambda k: -k["share"])
out = {"steps": nsteps, "sum_kernel_us_per_step": round(tot / nsteps / 1e3, 1), "kernels": ks}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k in ks[:16]:
    print(k)
print("steps", nsteps, "sum of in-step kernel time per step (us)", out["sum_kernel_us_per_step"])

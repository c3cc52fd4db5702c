"""Timeline of one train step from a rocprofv3 kernel trace CSV: per queue, kernel by kernel, start offset / duration / gap.
    python tools/step_timeline.py kt_kernel_trace.csv [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with adamw_kernel
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
lo, hi = ends[-k - 1] + 1, ends[-k]
step = rows[lo:hi + 1]
t0 = int(step[0]["Start_Timestamp"])
print("step span %.1f us, %d launches" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, len(step)))
last_end = {}
busy = {}
for r in step:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    name = r["Kernel_Name"].replace("void ", "")[:44]
    print("q%-2s %8.1f  dur %6.1f  gap %6.1f  %s  grid %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name, r["Grid_Size_X"]))
print({q: round(v / 1e3, 1) for q, v in busy.items()})

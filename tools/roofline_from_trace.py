"""From a rocprofv3 kernel trace of `bench.py`, isolate the launches of bench.py's roofline legs (kernels timed alone after
the train steps, i.e. every launch of them after the last adamw_kernel) and print their average durations, to be compared
with `roofline*.launch_us` of the same run.   python tools/roofline_from_trace.py b_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel"))
print("kernel,launches_in_roofline_leg,timed_launches,avg_us_timed,min_us,max_us")
LEGS = (("void conv_igemm_kernel<2, 3, 8, 1, 1, 1>", "conv_igemm_kernel<2, 3, 8, 1, 1, 1> (transposed conv 64->64 3x3 s2, 16x16->32x32, B=256, gelu + act')"),
        ("void conv_wino_kernel<false>", "conv_wino_kernel<false> (64->64 3x3 s1, 16x16, B=256, gelu + act')"),
        ("void conv_igemm_kernel<0, 3, 8, 1, 2, 1>", "conv_igemm_kernel<0, 3, 8, 1, 2, 1> (64->64 3x3 s1, 16x16, B=256, gelu + act')"),
        ("void conv_igemm_kernel<1, 3, 8, 1, 2, 4>", "conv_igemm_kernel<1, 3, 8, 1, 2, 4> (64->64 3x3 s2, 32x32->16x16, B=256, gelu + act')"))
for prefix, label in LEGS:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[last + 1:] if r["Kernel_Name"].startswith(prefix)]
    timed = d[5:]  # the first 5 are the leg's warm-up launches
    if timed:
        print(f'"{label}",{len(d)},{len(timed)},{sum(timed) / len(timed) / 1e3:.2f},{min(timed) / 1e3:.2f},{max(timed) / 1e3:.2f}')
# bench.py's loss-kernel leg (vae_loss_kernel at B=65,536: the launches longer than 200 us; 2 warm-up)
loss = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("void vae_loss_kernel")]
big = [x for x in loss if x > 200000][2:]
if big:
    print(f'"vae_loss_kernel<true, false, true> (B=65536, one-launch form)",{len(big) + 2},{len(big)},'
          f"{sum(big) / len(big) / 1e3:.2f},{min(big) / 1e3:.2f},{max(big) / 1e3:.2f}")

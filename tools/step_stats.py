"""bench.py's train step timed in several windows inside one process (device clocks drift by several % over seconds:
a single window is not a measurement).   python3 tools/step_stats.py [windows] [steps_per_window]
Prints every window's ms/step, then min and median."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
device = torch.device("cuda:0")
model, optim, sched = bench.build(device)
gen = torch.Generator(device=device).manual_seed(1234)
B = int(os.environ.get("KB_BATCH", "256"))
batches = [torch.rand(B, 3, 32, 32, device=device, generator=gen) for _ in range(4)]
egen = torch.Generator(device=device).manual_seed(4321)


def step(i):
    eps = torch.randn(B, 256, device=device, generator=egen)
    model.train_step(batches[i % 4], beta=5e-6, eps=eps, optimizer=None if os.environ.get("GMK_EARLY_ADAMW", "0") == "0" else optim)
    optim.step()
    sched.step()


for i in range(50):
    step(i)
res = []
for w in range(nwin):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
srt = sorted(res)
print(" ".join(f"{r:.4f}" for r in res), "| min %.4f median %.4f" % (srt[0], srt[len(srt) // 2]))

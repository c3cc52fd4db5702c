"""Host enqueue time vs GPU time of the train step (is the step launch-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import bench
dev = torch.device("cuda:0")
model, optim, sched = bench.build(dev)
red = None
if os.environ.get("GMK_DP_REHEARSE", "0") == "1":   # one-rank RCCL group under the step (gmk/parallel.py)
    from gmk.parallel import init_from_env, GradReducer
    init_from_env()
    red = GradReducer(model, optim, force=True)
x = torch.rand(256, 3, 32, 32, device=dev)
eps = torch.randn(256, 256, device=dev)
def step():
    l = model.train_step(x, beta=5e-6, eps=eps, reducer=red, optimizer=optim if os.environ.get("GMK_EARLY_ADAMW", "0") != "0" else None)
    if red is not None: red.wait()
    optim.step(); sched.step()
for _ in range(20): step()
torch.cuda.synchronize()
for n in (3, 3, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")
print("GMK_STEP_PLAN =", os.environ.get("GMK_STEP_PLAN", "1"))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

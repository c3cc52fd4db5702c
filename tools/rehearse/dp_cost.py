"""Where the data-parallel machinery's per-rank cost goes (one GPU, one-rank RCCL group): the configs[1] step timed with
parts of the path switched off.  DP_ABL = full | noar (no all-reduce calls: segment-wise replay + stream edges only) | plain.
    for m in plain full noar; do DP_ABL=$m python tools/rehearse/dp_cost.py; done
Measured (ms per step, one box): plain 0.961, noar 0.986, full 1.005 (decoder + encoder.fc as one all-reduce; 1.014 as two).
Issuing the all-reduces from a stream of their own that waits for both weight-gradient streams: 1.73."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
mode = os.environ.get("DP_ABL", "full")
if mode != "plain":
    os.environ["GMK_DP_REHEARSE"] = "1"
import torch
import bench
from gmk import ops
from gmk.parallel import init_from_env, GradReducer
init_from_env()
dev = torch.device("cuda:0")
model, optim, sched = bench.build(dev)
red = GradReducer(model, optim, force=True) if mode != "plain" else None
if mode == "noar":
    red.ready = lambda name, merged=True: None
x = torch.rand(256, 3, 32, 32, device=dev)
eps = torch.randn(256, 256, device=dev)


def step():
    model.train_step(x, beta=5e-6, eps=eps, reducer=red)
    if red is not None:
        red.wait()
    optim.step(); sched.step()


N = int(os.environ.get("DP_ITERS", "300"))
for _ in range(30):
    step()
res = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append((1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
print(mode, "host %.3f ms  total %.3f ms per step (best of 3: %.3f)" % (res[-1][0], res[-1][1], min(r[1] for r in res)), flush=True)
import torch.distributed as dist
if dist.is_initialized():
    dist.destroy_process_group()

"""Which part of the step is on the critical path?  Times the bench.py train step with groups of libgmk launches
replaced by no-ops (outputs are left as they were: results are WRONG, only the timing means something).
    python tools/sensitivity.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gmk import ops  # noqa: E402

dev = torch.device("cuda:0")
model, optim, sched = bench.build(dev)
from gmk.parallel import GradReducer  # noqa: E402
reducer = GradReducer(model, optim)
x = torch.rand(256, 3, 32, 32, device=dev)
eps = torch.randn(256, 256, device=dev)


def step():
    model.train_step(x, beta=5e-6, eps=eps, reducer=reducer)
    reducer.wait()
    optim.step()


def timed(n=100):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


ORIG = {k: getattr(ops, k) for k in ("conv2d_wgrad", "deconv2d_wgrad", "linear_wgrad", "conv2d_dgrad", "deconv2d_dgrad",
                                     "linear_dgrad")}


def stub_wgrad(name):
    orig = ORIG[name]

    def f(*a, out=None, **k):
        if out is not None:
            return out
        return orig(*a, **k)
    return f


def stub_dgrad(name):
    orig = ORIG[name]
    cache = {}

    def f(dy, w, shape, *a, **k):
        key = (name, tuple(shape))
        if key not in cache:
            cache[key] = orig(dy, w, shape, *a, **k)
        return cache[key]
    return f


print(f"baseline                      {timed():.4f} ms")
for n in ("conv2d_wgrad", "deconv2d_wgrad"):
    setattr(ops, n, stub_wgrad(n))
print(f"no conv/deconv wgrad launches {timed():.4f} ms")
setattr(ops, "linear_wgrad", stub_wgrad("linear_wgrad"))
print(f"no wgrad launches at all      {timed():.4f} ms")
for n, f in ORIG.items():
    setattr(ops, n, f)
for n in ("conv2d_dgrad", "deconv2d_dgrad"):
    setattr(ops, n, stub_dgrad(n))
print(f"no conv/deconv dgrad launches {timed():.4f} ms")
for n, f in ORIG.items():
    setattr(ops, n, f)
print(f"baseline again                {timed():.4f} ms")

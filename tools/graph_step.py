"""A/B of the bench.py train step enqueued eagerly (Python + ctypes, ~55 launches on two streams) against the same
forward + loss + backward captured once into a HIP graph and replayed (AdamW + the cosine step stay eager behind it).
   python3 tools/graph_step.py [steps]
Prints ms/step of eager / graph / eager / graph in one process (same device, same clocks) and checks that a replay
leaves exactly the gradients an eager call leaves."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
device = torch.device("cuda:0")
model, optim, sched = bench.build(device)
gen = torch.Generator(device=device).manual_seed(1234)
B = int(os.environ.get("KB_BATCH", "256"))
batch = torch.rand(B, 3, 32, 32, device=device, generator=gen)
egen = torch.Generator(device=device).manual_seed(4321)
eps = torch.randn(B, 256, device=device, generator=egen)
beta = 5e-6


def eager():
    eps.normal_(generator=egen)
    model.train_step(batch, beta=beta, eps=eps)
    optim.step()
    sched.step()


for _ in range(10):
    eager()
torch.cuda.synchronize()

# capture: same call, same static tensors
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    g_losses = model.train_step(batch, beta=beta, eps=eps)
torch.cuda.synchronize()

# a replay leaves the gradients an eager call leaves
model.train_step(batch, beta=beta, eps=eps)
ref = optim.flat_grad.clone()
optim.flat_grad.zero_()
graph.replay()
torch.cuda.synchronize()
print("replay == eager gradients:", bool(torch.equal(ref, optim.flat_grad)), "losses", g_losses.tolist())


def graphed():
    eps.normal_(generator=egen)
    graph.replay()
    optim.step()
    sched.step()


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for name, fn in (("eager", eager), ("graph", graphed), ("eager", eager), ("graph", graphed)):
    print(f"{name}: {timeit(fn):.4f} ms/step", flush=True)

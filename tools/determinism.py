"""Run-to-run determinism of the train step: the same step from the same parameters N times, gradients compared bit for bit.
    python tools/determinism.py [reps]        (knobs: GMK_WG16, GMK_SIDE_WGRAD)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import build  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
model, optim, _ = build(dev)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(256, 3, 32, 32, device=dev, generator=g)
eps = torch.randn(256, 256, device=dev, generator=g)
ref = None
bad = {}
for r in range(reps):
    junk = torch.rand(1 << 22, device=dev)  # perturb allocator / caches between runs
    losses = model.train_step(x, beta=5e-6, eps=eps)
    torch.cuda.synchronize()
    cur = (optim.flat_grad.clone(), losses.clone())
    if ref is None:
        ref = cur
        continue
    if not torch.equal(cur[1], ref[1]):
        bad["losses"] = bad.get("losses", 0) + 1
    for name, p in model.named_parameters():
        a, b = optim.segment([p])
        if not torch.equal(cur[0][a:b], ref[0][a:b]):
            bad[name] = bad.get(name, 0) + 1
print("reps", reps, "WG16", os.environ.get("GMK_WG16", "1"), "SIDE", os.environ.get("GMK_SIDE_WGRAD", "1"),
      "-> mismatching tensors:", bad if bad else "none (bit-identical)")

# ---- whole runs: K optimizer steps from the same initial state, twice; final parameters compared bit for bit
K = int(os.environ.get("DET_STEPS", "40"))
finals = []
for run in range(3):
    if run == 2:   # the same run on ONE stream, launch by launch: no fork / join event anywhere
        os.environ["GMK_SIDE_WGRAD"] = "0"
        os.environ["GMK_STEP_PLAN"] = "0"
    torch.manual_seed(0)
    model, optim, sched = build(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    batches = [torch.rand(256, 3, 32, 32, device=dev, generator=gen) for _ in range(4)]
    egen = torch.Generator(device=dev).manual_seed(4321)
    for i in range(K):
        e = torch.randn(256, 256, device=dev, generator=egen)
        losses = model.train_step(batches[i % 4], beta=5e-6, eps=e)
        optim.step()
        sched.step()
        if run == 1 and i % 7 == 0:
            junk = torch.rand(1 << 20, device=dev)   # perturb timing / allocator in the second run only
    torch.cuda.synchronize()
    finals.append((optim.flat_param.clone(), losses.clone()))
same = torch.equal(finals[0][0], finals[1][0])
same1 = torch.equal(finals[0][0], finals[2][0])
print("three streams + launch plan against one stream, launch by launch,", K, "steps:",
      "bit-identical" if same1 else f"DIFFER (max abs {float((finals[0][0] - finals[2][0]).abs().max()):.3e})")
print("runs of", K, "steps:", "bit-identical" if same else f"DIFFER (max abs {float((finals[0][0] - finals[1][0]).abs().max()):.3e})",
      "losses", finals[0][1].tolist(), finals[1][1].tolist())

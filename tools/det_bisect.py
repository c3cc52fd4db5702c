"""Find the first step / tensor where two identical runs diverge (debugging aid for cross-stream races)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd")); sys.path.insert(0, ROOT)
import torch
from bench import build
dev = torch.device("cuda:0")
K = int(os.environ.get("DET_STEPS", "40"))
logs = []
for run in range(2):
    torch.manual_seed(0)
    model, optim, sched = build(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    batches = [torch.rand(256, 3, 32, 32, device=dev, generator=gen) for _ in range(4)]
    egen = torch.Generator(device=dev).manual_seed(4321)
    log = []
    for i in range(K):
        e = torch.randn(256, 256, device=dev, generator=egen)
        losses = model.train_step(batches[i % 4], beta=5e-6, eps=e)
        log.append((optim.flat_grad.clone(), losses.clone(), model.latent_vector.clone()))
        optim.step(); sched.step()
        if run == 1 and i % 7 == 0:
            junk = torch.rand(1 << 20, device=dev)
    torch.cuda.synchronize()
    logs.append(log)
names = [(n, optim.segment([p])) for n, p in model.named_parameters()]
for i in range(K):
    a, b = logs[0][i], logs[1][i]
    diffs = [n for n, (s, e) in names if not torch.equal(a[0][s:e], b[0][s:e])]
    if diffs or not torch.equal(a[1], b[1]) or not torch.equal(a[2], b[2]):
        print("first divergence at step", i, "losses equal", torch.equal(a[1], b[1]), "z equal", torch.equal(a[2], b[2]), "tensors:", diffs)
        for n, (s, e) in names:
            if n in diffs:
                d = (a[0][s:e] - b[0][s:e]).abs()
                nz = d.nonzero().flatten()
                print("  ", n, "count", nz.numel(), "of", e - s, "first idx", nz[:6].tolist(), "max", float(d.max()))
        break
else:
    print("no divergence in", K, "steps")

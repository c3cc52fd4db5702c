"""Debug: error pattern of conv2d_wgrad for one shape (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch, torch.nn.functional as F
from gmk import ops
torch.manual_seed(0)
for shape in [(3, 40, 64, 64, 24, 3, 2), (3, 64, 64, 64, 32, 3, 2), (1, 32, 64, 64, 32, 3, 2), (3, 32, 32, 32, 32, 3, 2)]:
    N, Cin, H, W, Cout, k, s = shape
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(N, Cout, H // s, W // s)
    y = F.conv2d(x, w, None, s, 1)
    (gw,) = torch.autograd.grad(y, (w,), dy.double())
    dw, db = ops.conv2d_wgrad(dy.cuda(), x.detach().float().cuda(), k, stride=s)
    err = (dw.cpu().double() - gw).abs()
    print(shape, "max err", float(err.max()), "scale", float(gw.abs().max()))
    bad = (err > 1e-3 * gw.abs().max()).nonzero()
    print(" bad count", bad.shape[0], "of", gw.numel())
    if bad.shape[0]:
        import collections
        print("  by cout:", sorted(collections.Counter(bad[:, 0].tolist()).items())[:40])
        print("  by cin :", sorted(collections.Counter(bad[:, 1].tolist()).items())[:40])
        print("  by ky  :", sorted(collections.Counter(bad[:, 2].tolist()).items()))
        print("  by kx  :", sorted(collections.Counter(bad[:, 3].tolist()).items()))

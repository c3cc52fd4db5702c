import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import torch.nn.functional as F
from gmk import ops
torch.manual_seed(0)
N, Cin, H, W, Cout, k = 2, 3, 32, 32, 64, 3
x = torch.rand(N, Cin, H, W).double().requires_grad_(True)
w = torch.rand(Cout, Cin, k, k).double().requires_grad_(True)
b = torch.rand(Cout).double().requires_grad_(True)
dy = torch.rand(N, Cout, H, W)
y = F.conv2d(x, w, b, 1, 1)
gw, gb = torch.autograd.grad(y, (w, b), dy.double())
dw, db = ops.conv2d_wgrad(dy.cuda(), x.detach().float().cuda(), k, 1)
dw = dw.cpu().double(); db = db.cpu().double()
print("dw ref[0]", gw[0].flatten()[:12])
print("dw got[0]", dw[0].flatten()[:12])
print("ratio", (dw / gw).flatten()[:27])
print("db ratio", (db / gb)[:8])
print("max err", (dw - gw).abs().max().item(), (db - gb).abs().max().item())
print("---- structured")
xs = torch.ones(N, Cin, H, W)
for name, dyt in [("ones", torch.ones(N, Cout, H, W)),
                  ("pix", torch.arange(H * W, dtype=torch.float32).view(1, 1, H, W).expand(N, Cout, H, W).contiguous()),
                  ("chan", torch.arange(Cout, dtype=torch.float32).view(1, Cout, 1, 1).expand(N, Cout, H, W).contiguous())]:
    dw, db = ops.conv2d_wgrad(dyt.cuda(), xs.cuda(), k, 1)
    print(name, "db", db.cpu()[:6].tolist(), "expect", dyt[:, :6].sum((0, 2, 3)).tolist())
# single-pixel probes: which pixels are counted
for pix in [0, 1, 2, 31, 32, 33, 64, 255, 256, 300, 511, 512, 1023]:
    dyt = torch.zeros(N, Cout, H, W); dyt.view(N, Cout, -1)[0, :, pix] = 1.0
    dw, db = ops.conv2d_wgrad(dyt.cuda(), xs.cuda(), k, 1)
    print("pix", pix, "db0", db.cpu()[0].item(), "db33", db.cpu()[33].item())

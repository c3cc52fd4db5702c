"""Debug aid: Winograd conv (GMK_WINO=1 path) against torch's conv2d on the GPU for a few shapes; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
os.environ.setdefault("GMK_WINO_MIN_BLOCKS", "1")
import torch.nn.functional as F
from gmk import ops
torch.manual_seed(0)
dev = torch.device("cuda:0")
for (N, C, K, H, W) in [(8, 8, 16, 8, 8), (8, 8, 64, 8, 8), (4, 16, 64, 16, 16), (256, 64, 64, 16, 16), (32, 8, 16, 4, 4), (2, 8, 16, 32, 32)]:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.1
    b = torch.randn(K, device=dev)
    y = ops.conv2d_fwd(x, w, b, act=None, want_dact=False)
    ref = F.conv2d(x, w, b, padding=1)
    err = (y - ref).abs()
    print((N, C, K, H, W), "fwd max err", float(err.max()), "scale", float(ref.abs().max()))
    if err.max() > 1e-4:
        e = err.amax(dim=(0, 1))
        print(" err by (y,x):\n", (e > 1e-4).int().cpu().numpy() if H <= 16 else (e > 1e-4).int().sum().item())
        print(" err by n:", (err.amax(dim=(1, 2, 3)) > 1e-4).int().cpu().numpy()[:16])
        print(" err by k:", (err.amax(dim=(0, 2, 3)) > 1e-4).int().cpu().numpy())
    dy = torch.randn(N, K, H, W, device=dev)
    dact = torch.rand(N, C, H, W, device=dev)
    dx = ops.conv2d_dgrad(dy, w, (N, C, H, W), 1, dact_in=dact)
    refdx = F.conv_transpose2d(dy, w, padding=1) * dact
    print("   dgrad max err", float((dx - refdx).abs().max()), "scale", float(refdx.abs().max())) if C % 16 == 0 else None

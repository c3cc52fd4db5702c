"""How conv_igemm time scales with the number of K-chunks (Cin) -> per-chunk cost and fixed cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
from gmk import ops
dev = torch.device("cuda:0")
B = 256
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for H in (16, 8):
    for Cin in (8, 16, 32, 64, 128, 256):
        x = torch.rand(B, Cin, H, H, device=dev); w = torch.rand(64, Cin, 3, 3, device=dev) * 0.1; b = torch.zeros(64, device=dev)
        d = torch.rand(B, 64, H, H, device=dev)
        f0 = t(lambda: ops.conv2d_fwd(x, w, b))                                  # no act, no dact
        f1 = t(lambda: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True))
        f2 = t(lambda: ops.conv2d_fwd(x, w, b, act="relu", want_dact=True))
        print(f"H={H} Cin={Cin:4d} chunks={Cin//8:3d}  fwd(none) {f0:7.1f}  fwd(gelu+dact) {f1:7.1f}  fwd(relu+dact) {f2:7.1f} us", flush=True)

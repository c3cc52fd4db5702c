"""K-loop rate of the implicit-GEMM kernels without their fixed costs: the same layer at Cin = 64, 128, 256 input channels
(8, 16, 32 K-chunks); the time difference per added chunk is the K loop alone.   python3 tools/kslope.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
from gmk import ops
dev = torch.device("cuda:0")
B = 256


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, mk in (
    ("conv s2 32->16 (MODE 1, MT2)", lambda C: (lambda x=torch.rand(B, C, 32, 32, device=dev), w=torch.rand(64, C, 3, 3, device=dev) * .1, b=torch.zeros(64, device=dev): ops.conv2d_fwd(x, w, b, stride=2, act="gelu", want_dact=True))),
    ("deconv 16->32 (MODE 2)", lambda C: (lambda x=torch.rand(B, C, 16, 16, device=dev), w=torch.rand(C, 64, 3, 3, device=dev) * .1, b=torch.zeros(64, device=dev): ops.deconv2d_fwd(x, w, b, act="gelu", want_dact=True))),
    ("conv s1 8x8 (MODE 0)", lambda C: (lambda x=torch.rand(B, C, 8, 8, device=dev), w=torch.rand(64, C, 3, 3, device=dev) * .1, b=torch.zeros(64, device=dev): ops.conv2d_fwd(x, w, b, stride=1, act="gelu", want_dact=True))),
    ("conv s1 16x16 (Winograd)", lambda C: (lambda x=torch.rand(B, C, 16, 16, device=dev), w=torch.rand(64, C, 3, 3, device=dev) * .1, b=torch.zeros(64, device=dev): ops.conv2d_fwd(x, w, b, stride=1, act="gelu", want_dact=True))),
):
    res = {}
    for C in (64, 128, 256):
        res[C] = t(mk(C))
    px = {"conv s2 32->16 (MODE 1, MT2)": B * 256, "deconv 16->32 (MODE 2)": B * 256, "conv s1 8x8 (MODE 0)": B * 64, "conv s1 16x16 (Winograd)": B * 256}[name]
    flop64 = 2.0 * px * 64 * 64 * 9
    slope = (res[256] - res[64]) / 3.0          # us per 64 channels of K
    print(f"{name:30s} C=64 {res[64]:6.1f}  C=128 {res[128]:6.1f}  C=256 {res[256]:6.1f} us | per 64 channels {slope:6.2f} us = "
          f"{flop64 / slope / 1e6:6.1f} TFLOP/s ({flop64 / slope / 1e6 / 157.3 * 100:4.1f} % of peak) | fixed {res[64] - slope:5.1f} us", flush=True)

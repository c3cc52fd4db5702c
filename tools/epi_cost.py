"""What the activation epilogue costs: the big forward kernels with (gelu, act'), (identity, act' stored), (identity, no act')
and (relu, act').   python3 tools/epi_cost.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
from gmk import ops
dev = torch.device("cuda:0")
B, C = 256, 64
x16, x32 = torch.rand(B, C, 16, 16, device=dev), torch.rand(B, C, 32, 32, device=dev)
img = torch.rand(B, 3, 32, 32, device=dev)
w = (torch.rand(C, C, 3, 3, device=dev) - 0.5) * 0.1
w3 = (torch.rand(C, 3, 3, 3, device=dev) - 0.5) * 0.1
b = torch.zeros(C, device=dev)
cases = {
    "deconv_16to32": lambda act, wd: ops.deconv2d_fwd(x16, w, b, act=act, want_dact=wd),
    "conv_s2_32to16": lambda act, wd: ops.conv2d_fwd(x32, w, b, stride=2, act=act, want_dact=wd),
    "conv_s1_3to64_32": lambda act, wd: ops.conv2d_fwd(img, w3, b, act=act, want_dact=wd),
    "conv_s1_16 (wino)": lambda act, wd: ops.conv2d_fwd(x16, w, b, act=act, want_dact=wd),
}
for name, fn in cases.items():
    out = []
    for act, wd in (("gelu", True), ("relu", True), (None, True), (None, False)):
        for _ in range(3):
            fn(act, wd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn(act, wd)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{act}/{'dact' if wd else 'nodact'} {e0.elapsed_time(e1) * 1e3 / 30:6.1f}")
    print(f"{name:20s}", "  ".join(out), flush=True)

"""tools/kbench.py for the configs[4] layer shapes (64x64 images, base 16, latent 512, B = 256): HIP-event time per op alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
from gmk import ops
B = int(os.environ.get("KB_BATCH", "256"))
dev = torch.device("cuda:0")
R = lambda *s: torch.rand(*s, device=dev)
W = lambda *s: (torch.rand(*s, device=dev) - 0.5) * 0.1
C = 64
x64, x32, x16 = R(B, C, 64, 64), R(B, C, 32, 32), R(B, C, 16, 16)
img, d3 = R(B, 3, 64, 64), R(B, 3, 64, 64)
w, w3i, w3o, b, b3 = W(C, C, 3, 3), W(C, 3, 3, 3), W(3, C, 3, 3), W(C), W(3)
xf, wf = R(B, 16384), W(1024, 16384)
zf, wd = R(B, 512), W(16384, 512)
dyf, dyd = R(B, 1024), R(B, 16384)
_, idx = ops.maxpool2_fwd(x64)
cases = {
    "fwd_3to64_64": (lambda: ops.conv2d_fwd(img, w3i, b, act="gelu", want_dact=True), 2 * B * 4096 * 64 * 27),
    "fwd_s1_32": (lambda: ops.conv2d_fwd(x32, w, b, act="gelu", want_dact=True), 2 * B * 1024 * 64 * 576),
    "fwd_s1_16": (lambda: ops.conv2d_fwd(x16, w, b, act="gelu", want_dact=True), 2 * B * 256 * 64 * 576),
    "fwd_64to3_64": (lambda: ops.conv2d_fwd(x64, w3o, b3), 2 * B * 4096 * 3 * 576),
    "dgrad_s1_32": (lambda: ops.conv2d_dgrad(x32, w, (B, C, 32, 32), 1), 2 * B * 1024 * 64 * 576),
    "dgrad_3to64_64": (lambda: ops.conv2d_dgrad(d3, w3o, (B, C, 64, 64), 1), 2 * B * 4096 * 3 * 576),
    "wgrad_s1_32": (lambda: ops.conv2d_wgrad(x32, x32, 3, 1), 2 * B * 1024 * 64 * 576),
    "wgrad_3in_64": (lambda: ops.conv2d_wgrad(x64, img, 3, 1), 2 * B * 4096 * 64 * 27),
    "wgrad_3out_64": (lambda: ops.conv2d_wgrad(d3, x64, 3, 1), 2 * B * 4096 * 3 * 576),
    "maxpool_fwd_64": (lambda: ops.maxpool2_fwd(x64), 0), "maxpool_bwd_64": (lambda: ops.maxpool2_bwd(x32, idx, dact_in=x64), 0),
    "up_fwd_32to64": (lambda: ops.upsample2_fwd(x32), 0), "up_bwd_64to32": (lambda: ops.upsample2_bwd(x64, dact_in=x32), 0),
    "fc_enc_fwd": (lambda: ops.linear_fwd(xf, wf, None), 2 * B * 16384 * 1024),
    "fc_enc_dgrad": (lambda: ops.linear_dgrad(dyf, wf, dact_in=xf), 2 * B * 16384 * 1024),
    "fc_enc_wgrad": (lambda: ops.linear_wgrad(dyf, xf), 2 * B * 16384 * 1024),
    "fc_dec_fwd": (lambda: ops.linear_fwd(zf, wd, None, act="gelu", want_dact=True), 2 * B * 16384 * 512),
    "fc_dec_dgrad": (lambda: ops.linear_dgrad(dyd, wd), 2 * B * 16384 * 512),
    "fc_dec_wgrad": (lambda: ops.linear_wgrad(dyd, zf), 2 * B * 16384 * 512),
}
pat = sys.argv[1:]
tot = 0.0
for name, (fn, flops) in cases.items():
    if pat and not any(p in name for p in pat):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"{name:18s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
print(f"sum {tot:.1f} us")

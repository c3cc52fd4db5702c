"""Micro-benchmark of single libgmk ops at the cfg-2 layer shapes (B=256): HIP-event timing per op.
    python tools/kbench.py [op-substring] [--iters N]
Used with rocprofv3 --pmc for counter passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
from gmk import ops  # noqa: E402

B = int(os.environ.get("KB_BATCH", "256"))
dev = torch.device("cuda:0")
R = lambda *s: torch.rand(*s, device=dev)
W = lambda *s: (torch.rand(*s, device=dev) - 0.5) * 0.1


def cases():
    C = 64
    c = {}
    x32, x16, x8 = R(B, C, 32, 32), R(B, C, 16, 16), R(B, C, 8, 8)
    img = R(B, 3, 32, 32)
    w, w3i, w3o, b, b3 = W(C, C, 3, 3), W(C, 3, 3, 3), W(3, C, 3, 3), W(C), W(3)
    d3 = R(B, 3, 32, 32)
    c["fwd_s1_3to64_32"] = (lambda: ops.conv2d_fwd(img, w3i, b, act="gelu", want_dact=True), 2 * B * 1024 * 64 * 27)
    c["fwd_s2_32to16"] = (lambda: ops.conv2d_fwd(x32, w, b, stride=2, act="gelu", want_dact=True), 2 * B * 256 * 64 * 576)
    c["fwd_s1_16"] = (lambda: ops.conv2d_fwd(x16, w, b, act="gelu", want_dact=True), 2 * B * 256 * 64 * 576)
    c["fwd_s2_16to8"] = (lambda: ops.conv2d_fwd(x16, w, b, stride=2, act="gelu", want_dact=True), 2 * B * 64 * 64 * 576)
    c["fwd_s1_8"] = (lambda: ops.conv2d_fwd(x8, w, b, act="gelu", want_dact=True), 2 * B * 64 * 64 * 576)
    c["deconv_8to16"] = (lambda: ops.deconv2d_fwd(x8, w, b, act="gelu", want_dact=True), 2 * B * 64 * 64 * 576)
    c["deconv_16to32"] = (lambda: ops.deconv2d_fwd(x16, w, b, act="gelu", want_dact=True), 2 * B * 256 * 64 * 576)
    c["fwd_s1_64to3_32"] = (lambda: ops.conv2d_fwd(x32, w3o, b3), 2 * B * 1024 * 3 * 576)
    c["dgrad_s1_16"] = (lambda: ops.conv2d_dgrad(x16, w, (B, C, 16, 16), 1, dact_in=x16), 2 * B * 256 * 64 * 576)
    c["dgrad_s1_8"] = (lambda: ops.conv2d_dgrad(x8, w, (B, C, 8, 8), 1, dact_in=x8), 2 * B * 64 * 64 * 576)
    c["dgrad_s2_16to32"] = (lambda: ops.conv2d_dgrad(x16, w, (B, C, 32, 32), 2, dact_in=x32), 2 * B * 256 * 64 * 576)
    c["dgrad_s1_3to64_32"] = (lambda: ops.conv2d_dgrad(d3, w3o, (B, C, 32, 32), 1, dact_in=x32), 2 * B * 1024 * 3 * 576)
    c["deconv_dgrad_32to16"] = (lambda: ops.deconv2d_dgrad(x32, w, (B, C, 16, 16), dact_in=x16), 2 * B * 256 * 64 * 576)
    c["wgrad_s1_16"] = (lambda: ops.conv2d_wgrad(x16, x16, 3, 1), 2 * B * 256 * 64 * 576)
    c["wgrad_s1_8"] = (lambda: ops.conv2d_wgrad(x8, x8, 3, 1), 2 * B * 64 * 64 * 576)
    c["wgrad_s2_32to16"] = (lambda: ops.conv2d_wgrad(x16, x32, 3, 2), 2 * B * 256 * 64 * 576)
    c["wgrad_deconv_16to32"] = (lambda: ops.deconv2d_wgrad(x32, x16, 3), 2 * B * 256 * 64 * 576)
    c["wgrad_s1_3in_32"] = (lambda: ops.conv2d_wgrad(x32, img, 3, 1), 2 * B * 1024 * 64 * 27)
    c["wgrad_s1_3out_32"] = (lambda: ops.conv2d_wgrad(d3, x32, 3, 1), 2 * B * 1024 * 3 * 576)
    xf, wf, bf = R(B, 4096), W(512, 4096), W(512)
    zf, wd, bd = R(B, 256), W(4096, 256), W(4096)
    dyf, dyd = R(B, 512), R(B, 4096)
    c["fc_enc_fwd"] = (lambda: ops.linear_fwd(xf, wf, bf), 2 * B * 4096 * 512)
    c["fc_enc_dgrad"] = (lambda: ops.linear_dgrad(dyf, wf, dact_in=xf), 2 * B * 4096 * 512)
    c["fc_enc_wgrad"] = (lambda: ops.linear_wgrad(dyf, xf), 2 * B * 4096 * 512)
    c["fc_dec_fwd"] = (lambda: ops.linear_fwd(zf, wd, bd, act="gelu", want_dact=True), 2 * B * 4096 * 256)
    c["fc_dec_dgrad"] = (lambda: ops.linear_dgrad(dyd, wd), 2 * B * 4096 * 256)
    c["fc_dec_wgrad"] = (lambda: ops.linear_wgrad(dyd, zf), 2 * B * 4096 * 256)
    return c


def main():
    argv = sys.argv[1:]
    iters = 20
    if "--iters" in argv:
        k = argv.index("--iters")
        iters = int(argv[k + 1])
        del argv[k:k + 2]
    pat = [a for a in argv if not a.startswith("--")]
    tot = 0.0
    for name, (fn, flops) in cases().items():
        if pat and not any(p in name for p in pat):
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        tot += us
        print(f"{name:24s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({flops / us / 1e6 / 157.3 * 100:5.1f}% of f32 MFMA peak)", flush=True)
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()

"""Experiment: does splitting a conv launch into two half-batch launches on two HIP streams hide the
prologue / epilogue (HBM write burst) of one behind the MFMA loop of the other?
    python tools/dual_stream.py
Prints wall time per full batch for (a) one B=256 launch per iteration, (b) two B=128 launches on one stream,
(c) two B=128 launches on two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
from gmk import ops  # noqa: E402

dev = torch.device("cuda:0")
B, C = 256, 64
R = lambda *s: torch.rand(*s, device=dev)
w, b = (R(C, C, 3, 3) - 0.5) * 0.1, R(C)
x32, x16, x8 = R(B, C, 32, 32), R(B, C, 16, 16), R(B, C, 8, 8)

CASES = {
    "fwd_s1_16": lambda x: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True),
    "fwd_s2_32to16": lambda x: ops.conv2d_fwd(x, w, b, stride=2, act="gelu", want_dact=True),
    "deconv_16to32": lambda x: ops.deconv2d_fwd(x, w, b, act="gelu", want_dact=True),
    "fwd_s1_8": lambda x: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True),
}
INPUT = {"fwd_s1_16": x16, "fwd_s2_32to16": x32, "deconv_16to32": x16, "fwd_s1_8": x8}


def timed(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / iters


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for name, op in CASES.items():
    x = INPUT[name]
    xa, xb = x[:B // 2].contiguous(), x[B // 2:].contiguous()

    def full():
        op(x)

    def halves_one():
        op(xa)
        op(xb)

    def halves_two():
        with torch.cuda.stream(s1):
            op(xa)
        with torch.cuda.stream(s2):
            op(xb)

    print(f"{name:16s} full {timed(full):7.1f} us   2 halves/1 stream {timed(halves_one):7.1f} us   "
          f"2 halves/2 streams {timed(halves_two):7.1f} us", flush=True)

"""Per-device choice between kernel paths whose ranking differs from one MI355X to the next.

The two vector-ALU kernels of the 3-channel-side layers (csrc/conv_thin_in.hip, conv_thin_outv.hip) are 5-6 us faster per
launch than the matrix-core kernels they replace on every device measured, yet the train step around them gets 2.8 %
faster on most devices of the pool and 1.5-3.5 % slower on some.  `tune_thin_paths` times a
caller-supplied step with the paths off and on, in interleaved windows on this device, and keeps the faster setting
(gmk_conv_set_thin_paths, include/gmk.h).  No counterpart in the reference.

The step passed in must be self-contained and disposable (a throw-away model / optimizer on synthetic input): the caller's
real parameters are never touched here.  Results of the two paths agree to fp32 rounding but not bitwise, so a tuned
run is bitwise reproducible only among runs that made the same choice; `GMK_TUNE=0` skips the tuning (both paths on,
or whatever GMK_THIN_IN / GMK_THIN_OUTV say)."""
import ctypes as C
import os
import time

import torch

from ._lib import check, lib

CANDIDATES = ((3, 1), (0, 0))   # (thin_in mask, thin_outv); the first one is the default and wins ties


def get_thin_paths():
    a, b = C.c_int(), C.c_int()
    check(lib().gmk_conv_get_thin_paths(C.byref(a), C.byref(b)))
    return a.value, b.value


def set_thin_paths(thin_in, thin_outv):
    check(lib().gmk_conv_set_thin_paths(int(thin_in), int(thin_outv)))


def tune_thin_paths(step, steps=40, rounds=3, warmup=10, margin=0.003):
    """step(): one train step of a throw-away replica on the current device.  Returns {"choice": (thin_in, thin_outv),
    "ms": {candidate: best window ms/step}}.  The non-default candidate must win by `margin` (relative)."""
    if os.environ.get("GMK_TUNE", "1") == "0" or "GMK_THIN_IN" in os.environ or "GMK_THIN_OUTV" in os.environ:
        return {"choice": get_thin_paths(), "ms": {}, "tuned": False}
    best = {c: float("inf") for c in CANDIDATES}
    for c in CANDIDATES:          # both paths compiled / first-launch costs out of the way
        set_thin_paths(*c)
        for _ in range(warmup):
            step()
    for _ in range(rounds):
        for c in CANDIDATES:
            set_thin_paths(*c)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            best[c] = min(best[c], (time.perf_counter() - t0) / steps * 1e3)
    default, other = CANDIDATES
    choice = other if best[other] < best[default] * (1.0 - margin) else default
    set_thin_paths(*choice)
    return {"choice": choice, "ms": {f"{c[0]},{c[1]}": round(v, 4) for c, v in best.items()}, "tuned": True}

"""Launches the 3x3 weight-gradient kernels of the cfg-2 step (B=256, 64->64 channels) a few times each and nothing else:
   s1_16 : conv wgrad, stride 1, 16x16           (enc.conv2 / dec.conv2)
   s2_16 : conv wgrad, stride 2, 32x32 -> 16x16  (enc.downsample1)
   dc_16 : transposed-conv wgrad, 16x16 -> 32x32 (dec.deconv2)
   s1_8  : conv wgrad, stride 1, 8x8
   python3 tools/prof_wgrad.py [s1_16 s2_16 dc_16 s1_8]
Used under `rocprofv3 --kernel-trace --pmc ...` (one counter group per run); tools/pmc_summary.py digests the CSVs.
GMK_WG16=0 selects the older 32x32x2-MFMA kernel for the same shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
from gmk import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["s1_16", "s2_16", "dc_16", "s1_8"]
REPS = 5
B, C = 256, 64
x32, x16, x8 = (torch.randn(B, C, h, h, device=dev) for h in (32, 16, 8))
for _ in range(REPS):
    if "s1_16" in which:
        ops.conv2d_wgrad(x16, x16, 3, 1)
    if "s2_16" in which:
        ops.conv2d_wgrad(x16, x32, 3, 2)
    if "dc_16" in which:
        ops.deconv2d_wgrad(x32, x16, 3)
    if "s1_8" in which:
        ops.conv2d_wgrad(x8, x8, 3, 1)
torch.cuda.synchronize()

"""Launches the kernels BASELINE.json's north_star asks rocprof evidence for, a few times each, and nothing else:
   conv  : 64->64 3x3 at B=256 (+gelu, +act'): stride 1 at 16x16 (conv_wino_kernel; conv_igemm_kernel with GMK_WINO=0)
           and stride 2 at 32x32 (conv_igemm_kernel); transposed conv 16->32 and 8->16       -> MFMA utilisation
   loss  : vae_loss_kernel at B=256 and B=65536 (fwd+bwd in one pass)                  -> HBM bytes / GB/s
   gram  : gram_partial128_kernel, X (50000 x 256) fp32 -> fp64 moments                -> MFMA (f64) utilisation, HBM bytes
   wgrad : conv_wgrad16_kernel, the three 3x3 weight-gradient shapes of the cfg-2 step  -> MFMA utilisation
   python3 tools/prof_targets.py [conv|loss|gram|wgrad ...]
Used under `rocprofv3 --kernel-trace --pmc ...` (one counter group per run); tools/pmc_summary.py digests the CSVs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch  # noqa: E402
from gmk import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["conv", "loss", "gram", "wgrad"]
REPS = 5
if "conv" in which:
    x = torch.rand(256, 64, 16, 16, device=dev)
    w = (torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.1
    b = torch.zeros(64, device=dev)
    x32 = torch.rand(256, 64, 32, 32, device=dev)
    for _ in range(REPS):
        ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True)
        ops.conv2d_fwd(x32, w, b, stride=2, act="gelu", want_dact=True)
    # the transposed conv (conv_igemm_kernel<2, ...>, the largest share of in-step kernel time) at both shapes of the step
    x8 = torch.rand(256, 64, 8, 8, device=dev)
    for _ in range(REPS):
        ops.deconv2d_fwd(x, w, b, act="gelu", want_dact=True)      # 16x16 -> 32x32, grid 262144
        ops.deconv2d_fwd(x8, w, b, act="gelu", want_dact=True)     # 8x8 -> 16x16, grid 65536
if "loss" in which:
    for B in (256, 65536):
        pre = torch.randn(B, 3, 32, 32, device=dev)
        img = torch.rand(B, 3, 32, 32, device=dev)
        stats = torch.randn(B, 256, 2, device=dev) * 0.1
        for _ in range(REPS):
            ops.vae_loss_fwd_bwd(pre, img, stats, 5e-6, sigmoid=True, mse_prior=False, fused=True)
        del pre, img, stats
if "gram" in which:
    X = torch.randn(50000, 256, device=dev)
    for _ in range(REPS):
        G = torch.zeros(256, 256, device=dev, dtype=torch.float64)
        s = torch.zeros(256, device=dev, dtype=torch.float64)
        ops.gram_accumulate(X, G, s)
if "wgrad" in which:
    x32, x16 = (torch.randn(256, 64, h, h, device=dev) for h in (32, 16))
    for _ in range(REPS):
        ops.conv2d_wgrad(x16, x16, 3, 1)
        ops.conv2d_wgrad(x16, x32, 3, 2)
        ops.deconv2d_wgrad(x32, x16, 3)
torch.cuda.synchronize()

"""Times gmk_gram_accumulate on the FID shapes (HIP events): python tools/gram_bench.py   (GMK_GRAM64=1: round-1 kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
from gmk import ops
dev = torch.device("cuda:0")
for n, D in ((50000, 256), (2048, 256), (8192, 256), (50000, 512)):
    X = torch.randn(n, D, device=dev)
    G = torch.zeros(D, D, dtype=torch.float64, device=dev); s = torch.zeros(D, dtype=torch.float64, device=dev)
    for _ in range(3):
        ops.gram_accumulate(X, G, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gram_accumulate(X, G, s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * n * D * D
    print(f"n={n:6d} D={D}: {us:8.1f} us  {n / us:8.1f} M samples/s  {fl / us / 1e6:6.1f} TFLOP/s of the full product ({fl / us / 1e6 / 78.6 * 100:.0f}% of 78.6 f64-MFMA peak)")
    Gr = (X.double().T @ X.double()) * 23   # 3 + 20 accumulations
    print("   rel err vs torch fp64:", float((G - Gr).abs().max() / Gr.abs().max()), float((s - X.double().sum(0) * 23).abs().max()))

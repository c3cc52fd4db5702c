"""bench.py's FID-stats leg alone (for rocprofv3 --kernel-trace --stats):  python3 tools/fid_leg.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import bench  # noqa: F401  (path setup)
from bench_fid import fid_stats_leg
print(fid_stats_leg(torch.device("cuda:0"), 0, 1))

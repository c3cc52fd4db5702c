"""The configs[4] train step alone (VAE latent 512, pool + nearest upsample, 3x64x64, B = 256), for rocprofv3 --kernel-trace --stats:
   python3 tools/cfg5_step.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import bench
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
print(bench.secondary_config_leg(dev, "cfg5", steps=steps, warmup=3))

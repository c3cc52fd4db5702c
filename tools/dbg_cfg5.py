import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import numpy as np, torch
from oracle import ae_oracle, detgen
from models.autoencoders import Autoencoder
from utils import get_activation_function
for hw, L, no_pool, no_up in ((64, 512, False, False), (64, 512, True, True), (32, 512, False, False)):
    base = hw // 4
    shape = dict(variational=True, latent_size=L, num_channels=64, kernel_size=3, no_pool=no_pool, no_upsample=no_up)
    tag = "cfg5/nearest"; B = 3
    params = detgen.ae_params(tag, base=base, **shape)
    x = detgen.images(tag, B, hw); eps = detgen.normal(tag + "/eps", (B, L))
    m = Autoencoder(variational=True, latent_size=L, num_channels=64, kernel_size=3, act=get_activation_function("gelu"),
                    sigmoid=True, interpolation="nearest", no_pool=no_pool, no_upsample=no_up, device="cuda:0", image_size=hw)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}); m = m.to("cuda:0")
    losses = m.train_step(torch.from_numpy(x).cuda(), beta=5e-6, eps=torch.from_numpy(eps).cuda()).cpu().double()
    cfg = dict(variational=True, latent_size=L, num_channels=64, act="gelu", sigmoid=True, no_pool=no_pool,
               no_upsample=no_up, interpolation="nearest", kernel_size=3, base=base)
    o = ae_oracle.train_step_grads(params, x, eps, 5e-6, cfg)
    o64 = ae_oracle.train_step_grads(params, x, eps, 5e-6, cfg, dtype=torch.float64)
    print("hw", hw, "pool", not no_pool, "loss", float(losses[2]), float(o["total"]))
    for k, p in m.named_parameters():
        ref = o64["grads"][k].numpy().astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        o32 = o["grads"][k].numpy().astype(np.float64)
        n = np.sqrt((ref ** 2).sum())
        print(f"  {k:28s} gpu-vs-f64 {np.sqrt(((got-ref)**2).sum())/n:.2e}   cpu32-vs-f64 {np.sqrt(((o32-ref)**2).sum())/n:.2e}")

"""Context measurement (not part of the product, not the oracle): the same VAE train step written as stock
PyTorch modules and run eagerly through PyTorch-ROCm (MIOpen convolutions, rocBLAS GEMMs, autograd, torch AdamW)
on the same GPU.  It answers "what does the reference's own code path reach on an MI355X", which is the number
libgmk has to beat; DESIGN.md section 5 quotes it.

    python tools/torch_rocm_step.py [--batch 256] [--steps 100] [--tf32]
"""
import argparse
import json
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class StockVAE(nn.Module):
    """Layer shapes of the cfg-2 model (latent 256, 64 channels, strided convs / transposed convs, GELU, sigmoid)."""

    def __init__(self, latent=256, ch=64):
        super().__init__()
        self.latent, self.ch = latent, ch
        self.enc = nn.ModuleList([nn.Conv2d(3, ch, 3, 1, 1), nn.Conv2d(ch, ch, 3, 2, 1), nn.Conv2d(ch, ch, 3, 1, 1),
                                  nn.Conv2d(ch, ch, 3, 2, 1), nn.Conv2d(ch, ch, 3, 1, 1)])
        self.enc_fc = nn.Linear(ch * 64, 2 * latent)
        self.dec_fc = nn.Linear(latent, ch * 64)
        self.dec = nn.ModuleList([nn.Conv2d(ch, ch, 3, 1, 1), nn.ConvTranspose2d(ch, ch, 3, 2, 1, 1),
                                  nn.Conv2d(ch, ch, 3, 1, 1), nn.ConvTranspose2d(ch, ch, 3, 2, 1, 1)])
        self.out = nn.Conv2d(ch, 3, 3, 1, 1)

    def forward(self, x):
        for layer in self.enc:
            x = F.gelu(layer(x))
        stats = self.enc_fc(x.flatten(1)).view(-1, self.latent, 2)
        mu, log_var = stats[:, :, 0], stats[:, :, 1]
        z = mu + torch.exp(0.5 * log_var) * torch.randn_like(mu)
        x = F.gelu(self.dec_fc(z)).view(-1, self.ch, 8, 8)
        for layer in self.dec:
            x = F.gelu(layer(x))
        return torch.sigmoid(self.out(x)), mu, log_var


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tf32", action="store_true", help="allow reduced-precision matmul/conv paths (default: strict fp32)")
    a = ap.parse_args()
    torch.backends.cudnn.allow_tf32 = a.tf32
    torch.backends.cuda.matmul.allow_tf32 = a.tf32
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    model = StockVAE().to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100000)
    x = torch.rand(a.batch, 3, 32, 32, device=dev)

    def step():
        recon, mu, log_var = model(x)
        kl = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
        loss = F.mse_loss(recon, x) + 5e-6 * kl
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(json.dumps({"what": "stock PyTorch-ROCm eager step (MIOpen/rocBLAS/autograd)", "batch": a.batch,
                      "ms_per_step": round(ms, 4), "images_per_sec": round(a.batch * 1e3 / ms, 1), "tf32": a.tf32}))


if __name__ == "__main__":
    main()

"""CPU restatement of the reference autoencoder / VAE train step (TEST INFRASTRUCTURE).

Pinned: every function here is checked against the imported reference
(/root/reference/models/autoencoders.py, main.py:27-53) by oracle/gen_golden.py in the
build container, and against the committed vectors under tests/golden/ by
tests/test_oracle_golden.py.  Floating point (fp32, optional fp64 re-run), PyTorch CPU
functional ops, no nn.Module state.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
The product path (generative-models_amd/) never does.
"""
import math

import torch
import torch.nn.functional as F

ACTS = ("relu", "leaky-relu", "elu", "selu", "gelu", "swish", "mish")


def act_forward(x, name):
    """/root/reference/utils.py:49-66 -- PyTorch default arguments for every activation."""
    if name == "relu":
        return F.relu(x)
    if name == "leaky-relu":
        return F.leaky_relu(x, 0.01)
    if name == "elu":
        return F.elu(x, 1.0)
    if name == "selu":
        return F.selu(x)
    if name == "gelu":
        return F.gelu(x)  # exact erf form
    if name == "swish":
        return F.silu(x)
    if name == "mish":
        return F.mish(x)
    raise NotImplementedError(name)


def encoder_forward(p, x, *, variational, latent_size, act, no_pool, kernel_size=3):
    """/root/reference/models/autoencoders.py:44-65."""
    pad = (kernel_size - 1) // 2
    a = lambda t: act_forward(t, act)
    x = a(F.conv2d(x, p["encoder.conv1.weight"], p["encoder.conv1.bias"], 1, pad))
    if no_pool:
        x = a(F.conv2d(x, p["encoder.downsample1.weight"], p["encoder.downsample1.bias"], 2, pad))
    else:
        x = F.max_pool2d(x, 2, 2)
    x = a(F.conv2d(x, p["encoder.conv2.weight"], p["encoder.conv2.bias"], 1, pad))
    if no_pool:
        x = a(F.conv2d(x, p["encoder.downsample2.weight"], p["encoder.downsample2.bias"], 2, pad))
    else:
        x = F.max_pool2d(x, 2, 2)
    x = a(F.conv2d(x, p["encoder.conv3.weight"], p["encoder.conv3.bias"], 1, pad))
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, p["encoder.fc.weight"], p["encoder.fc.bias"])
    if variational:
        return x.view(x.shape[0], latent_size, 2)  # mu = [...,0], log_var = [...,1] (interleaved)
    return x


def decoder_forward(p, z, *, num_channels, act, sigmoid, no_upsample, interpolation="nearest",
                    kernel_size=3, base=8):
    """/root/reference/models/autoencoders.py:93-115 (base=8 there; base=H/4 generalises cfg 5)."""
    pad = (kernel_size - 1) // 2
    a = lambda t: act_forward(t, act)

    def up(t):
        if interpolation == "nearest":
            return F.interpolate(t, scale_factor=2, mode="nearest")
        return F.interpolate(t, scale_factor=2, mode=interpolation)  # nn.Upsample: align_corners=None

    x = a(F.linear(z, p["decoder.fc.weight"], p["decoder.fc.bias"]))
    x = x.view(x.shape[0], num_channels, base, base)
    x = a(F.conv2d(x, p["decoder.conv1.weight"], p["decoder.conv1.bias"], 1, pad))
    if no_upsample:
        x = a(F.conv_transpose2d(x, p["decoder.deconv1.weight"], p["decoder.deconv1.bias"], 2, pad, 1))
    else:
        x = up(x)
    x = a(F.conv2d(x, p["decoder.conv2.weight"], p["decoder.conv2.bias"], 1, pad))
    if no_upsample:
        x = a(F.conv_transpose2d(x, p["decoder.deconv2.weight"], p["decoder.deconv2.bias"], 2, pad, 1))
    else:
        x = up(x)
    x = F.conv2d(x, p["decoder.conv3.weight"], p["decoder.conv3.bias"], 1, pad)
    return torch.sigmoid(x) if sigmoid else x


def autoencoder_forward(p, x, eps=None, *, variational, latent_size, num_channels, act, sigmoid,
                        no_pool, no_upsample, interpolation="nearest", kernel_size=3, base=8):
    """/root/reference/models/autoencoders.py:158-182.  `eps` replaces randn_like (:174)."""
    out = encoder_forward(p, x, variational=variational, latent_size=latent_size, act=act,
                          no_pool=no_pool, kernel_size=kernel_size)
    stats = None
    if variational:
        stats = out
        mu, lv = stats[:, :, 0], stats[:, :, 1]
        z = mu + torch.exp(0.5 * lv) * eps
    else:
        z = out
    recon = decoder_forward(p, z, num_channels=num_channels, act=act, sigmoid=sigmoid,
                            no_upsample=no_upsample, interpolation=interpolation,
                            kernel_size=kernel_size, base=base)
    return recon, stats, z


def train_losses(recon, image, stats, beta, *, variational, mse_prior=False):
    """/root/reference/main.py:31-48.  Returns (rec, kl_unweighted_or_None, total)."""
    rec = F.mse_loss(recon, image)
    if not variational:
        return rec, None, rec
    mu, lv = stats[:, :, 0], stats[:, :, 1]
    if mse_prior:  # main.py:39-42
        sigma = torch.exp(0.5 * lv)
        kl = (mu ** 2).sum(1).mean(0) + ((sigma - 1.0) ** 2).sum(1).mean(0)
    else:          # main.py:45
        kl = torch.mean(-0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp(), dim=1), dim=0)
    return rec, kl, rec + beta * kl


def train_step_grads(params, x, eps, beta, cfg, mse_prior=False, dtype=torch.float32):
    """Forward + loss + backward of main.py:29-51; returns losses, grads, and intermediates."""
    p = {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    x = torch.as_tensor(x).to(dtype)
    eps_t = None if eps is None else torch.as_tensor(eps).to(dtype)
    recon, stats, z = autoencoder_forward(p, x, eps_t, **cfg)
    rec, kl, total = train_losses(recon, x, stats, beta, variational=cfg["variational"], mse_prior=mse_prior)
    total.backward()
    grads = {k: v.grad.detach() for k, v in p.items()}
    return dict(rec=rec.detach(), kl=None if kl is None else kl.detach(), total=total.detach(),
                recon=recon.detach(), stats=None if stats is None else stats.detach(),
                z=z.detach(), grads=grads)


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), main.py:52,214.
    `step` is the 1-based step index of this update."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def cosine_lr(base_lr, t, t_max, eta_min=0.0):
    """Closed form of CosineAnnealingLR after t scheduler steps (main.py:53,215), t <= T_max."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / t_max)) / 2.0


def cosine_lr_recursive(lr_prev, base_lr, t, t_max, eta_min=0.0):
    """torch's chained form (what the reference actually executes, incl. t > T_max)."""
    if t == 0:
        return base_lr
    if (t - 1 - t_max) % (2 * t_max) == 0:
        return lr_prev + (base_lr - eta_min) * (1 - math.cos(math.pi / t_max)) / 2
    return (1 + math.cos(math.pi * t / t_max)) / (1 + math.cos(math.pi * (t - 1) / t_max)) * (lr_prev - eta_min) + eta_min

"""GPU parity of the full autoencoder / VAE train step against the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import case_inputs, check_against_golden, sample
from oracle import detgen

pytestmark = pytest.mark.gpu

ALL = ["ae_cfg1_b8", "vae_cfg2_b8", "vae_cfg2_b50", "vae_cfg2_b256", "vae_cfg2_msep_b8", "vae_cfg2_bigbeta_b8",
       "ae_nosigmoid_b4", "vae_pool_nearest_b4", "vae_pool_bilinear_b4", "vae_pool_bicubic_b4", "vae_pool_deconv_b4",
       "vae_stride_upsample_b4", "vae_k5_b2", "vae_act_relu_b2", "vae_act_leaky-relu_b2", "vae_act_elu_b2",
       "vae_act_selu_b2", "vae_act_gelu_b2", "vae_act_swish_b2", "vae_act_mish_b2"]


def build(case, params):
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    cfg = case["cfg"]
    m = Autoencoder(variational=cfg["variational"], latent_size=cfg["latent_size"], num_channels=cfg["num_channels"],
                    kernel_size=cfg["kernel_size"], act=get_activation_function(case["act"]), sigmoid=cfg["sigmoid"],
                    interpolation=cfg["interpolation"], no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"],
                    device="cuda:0", image_size=case["hw"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    assert list(m.state_dict().keys()) == list(sd.keys())   # same keys, same order as the reference
    m.load_state_dict(sd)
    return m.to("cuda:0")


@pytest.mark.parametrize("name", ALL)
def test_fused_train_step_matches_golden(train_golden, name):
    case = train_golden[name]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    xt = torch.from_numpy(x).cuda()
    et = None if eps is None else torch.from_numpy(eps).cuda()
    losses = m.train_step(xt, beta=case["beta"], mse_prior=case["mse_prior"], eps=et).cpu().double()
    out = dict(rec=float(losses[0]), kl=float(losses[1]), total=float(losses[2]),
               grads={k: p.grad.cpu().numpy() for k, p in m.named_parameters()})
    check_against_golden(case, out)   # 1e-4 relative: loss, per-tensor gradient norms and samples
    if "z_sample" in case:
        assert np.allclose(sample(m.latent_vector.cpu().numpy()), case["z_sample"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["ae_cfg1_b8", "vae_cfg2_b8", "vae_pool_bilinear_b4", "ae_nosigmoid_b4"])
def test_autograd_path_matches_golden(train_golden, name):
    """The reference's own loop (main.py:29-51): recon = model(image); loss from torch ops; backward."""
    case = train_golden[name]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    m.train()
    xt = torch.from_numpy(x).cuda()
    orig = torch.randn_like
    if eps is not None:
        et = torch.from_numpy(eps).cuda()
        torch.randn_like = lambda t, *a, **k: et
    try:
        recon = m(xt)
    finally:
        torch.randn_like = orig
    rec = F.mse_loss(recon, xt)
    loss = rec
    kl = None
    if case["cfg"]["variational"]:
        mu, log_var = m.stats[:, :, 0], m.stats[:, :, 1]
        kl = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
        loss = loss + case["beta"] * kl
    m.zero_grad()
    loss.backward()
    out = dict(rec=float(rec), kl=None if kl is None else float(kl), total=float(loss),
               grads={k: p.grad.cpu().numpy() for k, p in m.named_parameters()})
    check_against_golden(case, out)
    assert np.allclose(sample(recon.detach().cpu().numpy()), case["recon_sample"], atol=2e-5)


def test_eval_forward_and_sample(train_golden):
    case = train_golden["vae_cfg2_b8"]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    m.eval()
    with torch.no_grad():
        recon = m(torch.from_numpy(x).cuda())
    assert recon.shape == (8, 3, 32, 32) and m.stats.shape == (8, 256, 2) and m.latent_vector.shape == (8, 256)
    s = m.sample(num_images=5, return_samples=True)
    assert s.shape == (5, 3, 32, 32) and float(s.min()) >= 0 and float(s.max()) <= 1


def test_cpu_tensor_fails_loudly(train_golden):
    case = train_golden["vae_act_gelu_b2"]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    with pytest.raises((ValueError, RuntimeError)):
        m.cpu()(torch.from_numpy(x))


@pytest.mark.parametrize("interp,B", [("nearest", 16), ("bilinear", 16), ("bicubic", 3)])
def test_cfg5_pool_upsample_latent512_64x64_matches_oracle(interp, B):
    """BASELINE.json configs[4]: VAE with pool + upsample layers, latent_size 512, 3x64x64 inputs.  The reference cannot
    run it (8*8*num_channels is hard-coded, autoencoders.py:41,81,96: RuntimeError on 64x64), so there is no reference
    golden: PARITY UNPINNED beyond the CPU oracle, which is itself pinned against the reference at 32x32 with the same
    pool / upsample layers (vae_pool_* goldens).  Tolerance: BASELINE's 1e-4 relative on loss and per-tensor gradients."""
    from oracle import ae_oracle
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    shape = dict(variational=True, latent_size=512, num_channels=64, kernel_size=3, no_pool=False, no_upsample=False)
    tag = f"cfg5/{interp}"
    params = detgen.ae_params(tag, base=16, **shape)
    x = detgen.images(tag, B, 64)
    eps = detgen.normal(tag + "/eps", (B, 512))
    m = Autoencoder(variational=True, latent_size=512, num_channels=64, kernel_size=3, act=get_activation_function("gelu"),
                    sigmoid=True, interpolation=interp, no_pool=False, no_upsample=False, device="cuda:0", image_size=64)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m = m.to("cuda:0")
    losses = m.train_step(torch.from_numpy(x).cuda(), beta=5e-6, eps=torch.from_numpy(eps).cuda()).cpu().double()
    cfg = dict(variational=True, latent_size=512, num_channels=64, act="gelu", sigmoid=True, no_pool=False,
               no_upsample=False, interpolation=interp, kernel_size=3, base=16)
    # fp64 run of the oracle: at 64x64 an fp32 CPU run flips one near-tied max-pool arg-max (1.7e-3 on the conv1 / conv2
    # gradients against its own fp64 run), the kernels agree with fp64 to 1e-6
    o = ae_oracle.train_step_grads(params, x, eps, 5e-6, cfg, dtype=torch.float64)
    assert abs(float(losses[2]) - float(o["total"])) <= 1e-4 * abs(float(o["total"]))
    assert abs(float(losses[1]) - float(o["kl"])) <= 1e-4 * abs(float(o["kl"]))
    for k, p in m.named_parameters():
        ref = o["grads"][k].numpy().astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        err = np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30)
        assert err <= 1e-4, (k, err)

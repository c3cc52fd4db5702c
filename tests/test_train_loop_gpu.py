"""GPU parity of the epoch loop around the hot path: `train_one_epoch` with its per-epoch
beta growth (/root/reference/main.py:21-70), `compute_test_loss` (main.py:73-111), the checkpoint dict
(main.py:323-331), `Autoencoder.sample` in eval mode (models/autoencoders.py:184-189) and models restored from
full-model pickles WRITTEN BY THE REFERENCE (main.py:189-195, 338-339).

Goldens: tests/golden/train_extra.json + *.pth, produced by oracle/gen_golden.py gen_extra() from the imported
reference (train_one_epoch / compute_test_loss restated around its model, randn draws injected)."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import sample, l2
from oracle import ae_oracle, detgen

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Draws:
    """Stands in for torch.randn while the product code runs: hands out the injected draws in order."""
    def __init__(self, items):
        self.items = list(items)

    def __call__(self, *a, **k):
        return self.items.pop(0)


def _model(variational, L, C, **kw):
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    cfg = dict(variational=variational, latent_size=L, num_channels=C, kernel_size=3, act=get_activation_function("gelu"),
               sigmoid=True, interpolation="nearest", no_pool=True, no_upsample=True, device="cuda:0")
    cfg.update(kw)
    return Autoencoder(**cfg)


def _load(m, params):
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to("cuda:0")


@pytest.mark.parametrize("name", ["epoch_vae", "epoch_vae_msep", "epoch_ae"])
def test_train_one_epoch_and_compute_test_loss(extra_golden, monkeypatch, tmp_path, name):
    import main as drv
    from gmk.optim import FusedAdamW
    gold = extra_golden[name]
    variational, mse = gold["variational"], gold["mse"]
    shape = dict(variational=variational, latent_size=64, num_channels=32, kernel_size=3, no_pool=True, no_upsample=True)
    m = _load(_model(variational, 64, 32), detgen.ae_params(name, **shape))
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=gold["T_max"])
    args = types.SimpleNamespace(variational=variational, mse=mse, beta_mult=gold["beta_mult"])
    B = gold["B"]
    label = torch.zeros(B, dtype=torch.long)
    train = [(torch.from_numpy(detgen.images(f"{name}/train{i}", B)), label) for i in range(gold["nb_train"])]
    test = [(torch.from_numpy(detgen.images(f"{name}/test{i}", B)), label) for i in range(gold["nb_test"])]
    draws = [torch.from_numpy(detgen.normal(f"{name}/eps_train{i}", (B, 64))).cuda() for i in range(gold["nb_train"])] + \
            [torch.from_numpy(detgen.normal(f"{name}/eps_test{i}", (B, 64))).cuda() for i in range(gold["nb_test"])]
    if variational:
        monkeypatch.setattr(torch, "randn", _Draws(draws))
    dev = torch.device("cuda:0")
    beta, tr_rec, tr_kl = drv.train_one_epoch(args, m, epoch=0, beta=gold["beta0"], train_dataloader=train, optim=opt,
                                              lr_scheduler=sched, device=dev, writer=None, reducer=None)
    assert m.training
    lm, ls, te_rec, te_kl = drv.compute_test_loss(args, m, beta=beta, test_dataloader=test, device=dev)
    assert not m.training
    monkeypatch.undo()

    def close(got, ref, rtol=1e-4):
        assert abs(float(got) - ref) <= rtol * abs(ref) + 1e-12, (float(got), ref)

    assert abs(beta - gold["beta_after"]) < 1e-15          # beta *= beta_mult once per epoch, VAE only (main.py:65-66)
    close(tr_rec, gold["train_rec"]); close(tr_kl, gold["train_kl"])
    close(te_rec, gold["test_rec"]); close(te_kl, gold["test_kl"])
    close(lm, gold["latent_mean"]); close(ls, gold["latent_std"])
    assert abs(sched.get_last_lr()[0] - gold["lr_after"]) < 1e-12
    for k, p in m.named_parameters():
        got = p.detach().cpu().numpy()
        assert abs(l2(got) - gold["param_norm"][k]) <= 1e-5 * gold["param_norm"][k], k
        assert np.allclose(sample(got), gold["param_sample"][k], rtol=1e-5, atol=1e-5), k

    # checkpoint dict: the reference's keys and value layouts (main.py:323-331); survives torch.save / load; the optimizer
    # and scheduler states load into the reference's own classes; a resumed run continues bit for bit
    ck = drv.make_checkpoint(m, opt, sched, 0, args, beta, "losses: ...")
    assert list(ck) == ["state_dict", "optimizer", "lr_scheduler", "epoch", "args", "current_beta", "losses"]
    assert isinstance(ck["epoch"], int) and isinstance(ck["current_beta"], float) and isinstance(ck["losses"], str)
    assert list(ck["state_dict"]) == list(detgen.ae_param_shapes(**shape))
    path = tmp_path / "ck.pth"
    torch.save(ck, path)
    ck2 = torch.load(path, weights_only=False)   # a file this test wrote (it pickles `args`, like the reference's)
    stock_p = [torch.zeros_like(p, device="cpu").requires_grad_(True) for p in m.parameters()]
    stock = torch.optim.AdamW(stock_p, lr=1e-3, weight_decay=0.01)
    stock_s = torch.optim.lr_scheduler.CosineAnnealingLR(stock, T_max=gold["T_max"])
    stock.load_state_dict(ck2["optimizer"]); stock_s.load_state_dict(ck2["lr_scheduler"])
    assert stock_s.last_epoch == gold["nb_train"] and float(stock.state[stock_p[0]]["step"]) == gold["nb_train"]
    m2 = _model(variational, 64, 32)
    m2.load_state_dict(ck2["state_dict"]); m2 = m2.to(dev)
    opt2 = FusedAdamW(m2.parameters(), lr=1e-3, weight_decay=0.01)
    sched2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=gold["T_max"])
    opt2.load_state_dict(ck2["optimizer"]); sched2.load_state_dict(ck2["lr_scheduler"])
    x = train[0][0].cuda()
    e = draws[0] if variational else None
    for mm, oo, ss in ((m, opt, sched), (m2, opt2, sched2)):
        mm.train_step(x, beta=ck2["current_beta"] if variational else 0.0, mse_prior=mse, eps=e)
        oo.step(); ss.step()
    for p, q in zip(m.parameters(), m2.parameters()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("name", ["epoch_vae", "epoch_vae_msep", "epoch_ae"])
def test_eval_loss_kernel_matches_oracle(name):
    """gmk_vae_loss_eval (no-grad form of the fused loss kernel) against the oracle's restatement of main.py:84-102,
    and against the training form of the same kernel (same rec / kl, no gradient buffers touched)."""
    from gmk import ops
    variational, mse = name != "epoch_ae", name == "epoch_vae_msep"
    B, L = 24, 48
    pre = torch.from_numpy(detgen.normal(name + "/pre", (B, 3, 32, 32)) * 2.0)
    img = torch.from_numpy(detgen.images(name + "/img", B))
    stats = torch.from_numpy(detgen.normal(name + "/stats", (B, L, 2)) * 0.7) if variational else None
    beta = 0.3
    out5, recon = ops.vae_loss_eval(pre.cuda(), img.cuda(), None if stats is None else stats.cuda(), beta, sigmoid=True,
                                    mse_prior=mse, want_recon=True)
    out5 = out5.cpu().double()
    rec, kl, total = ae_oracle.train_losses(torch.sigmoid(pre.double()), img.double(), None if stats is None else stats.double(),
                                            beta, variational=variational, mse_prior=mse)
    assert abs(out5[0] - float(rec)) <= 1e-6 * float(rec)
    assert abs(out5[2] - float(total)) <= 1e-6 * float(total)
    assert torch.allclose(recon.cpu(), torch.sigmoid(pre), atol=1e-6)
    if variational:
        assert abs(out5[1] - float(kl)) <= 2e-6 * float(kl)
        assert abs(out5[3] - float(stats[:, :, 0].double().abs().mean())) <= 1e-6
        assert abs(out5[4] - float(torch.exp(0.5 * stats[:, :, 1].double()).mean())) <= 1e-6
    else:
        assert float(out5[1]) == 0.0 and float(out5[3]) == 0.0 and float(out5[4]) == 0.0
    losses, _, _, _ = ops.vae_loss_fwd_bwd(pre.cuda(), img.cuda(), None if stats is None else stats.cuda(), beta,
                                           sigmoid=True, mse_prior=mse)
    assert torch.allclose(losses.cpu().double(), out5[:3], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", ["vae_cfg2_b8", "vae_pool_bilinear_b4", "ae_nosigmoid_b4"])
def test_sample_eval_mode_matches_reference(train_golden, extra_golden, monkeypatch, name):
    """Autoencoder.sample(return_samples=True): eval() + decoder(z), z injected in place of torch.randn."""
    from helpers import case_inputs
    from test_model_gpu import build
    case, gold = train_golden[name], extra_golden[f"sample_{name}"]
    params, _, _ = case_inputs(case)
    m = build(case, params)
    m.train()
    z = torch.from_numpy(detgen.normal(f"{name}/sample_z", (5, case["cfg"]["latent_size"]))).cuda()
    monkeypatch.setattr(torch, "randn", _Draws([z]))
    s = m.sample(num_images=5, return_samples=True)
    monkeypatch.undo()
    assert not m.training and list(s.shape) == gold["shape"]
    got = s.detach().cpu().numpy()
    assert abs(l2(got) - gold["l2"]) <= 1e-5 * gold["l2"]
    assert np.allclose(sample(got), gold["sample"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["refpickle_ae_deconv", "refpickle_ae_pool_bilinear"])
def test_reference_written_full_model_pickle_runs(extra_golden, name):
    """tests/golden/refpickle_*.pth are `torch.save(model)` files of the REFERENCE's Autoencoder (tiny configs).  Unpickling
    binds models.autoencoders.{Autoencoder,Encoder,Decoder} to the classes of this package and restores the reference's
    __dict__ (no `base`, no `interpolation`).  The restored model must run: eval forward, compute_fid features, and the
    fused train step, all against the reference's outputs / the oracle."""
    import metrics
    from models.autoencoders import Autoencoder
    gold = extra_golden[name]
    m = torch.load(os.path.join(GOLDEN, name + ".pth"), weights_only=False)   # written by oracle/gen_golden.py in this repo
    assert type(m) is Autoencoder and "base" not in m.encoder.__dict__ and "interpolation" not in m.decoder.__dict__
    assert list(m.state_dict()) == gold["keys"]
    m = m.to("cuda:0")
    m.eval()
    x = torch.from_numpy(detgen.images(name, 4)).cuda()
    with torch.no_grad():
        recon = m(x)
    got = recon.cpu().numpy()
    assert abs(l2(got) - gold["recon_l2"]) <= 1e-5 * gold["recon_l2"]
    assert np.allclose(sample(got), gold["recon_sample"], atol=2e-5)
    assert np.allclose(sample(m.latent_vector.cpu().numpy()), gold["latent_sample"], rtol=1e-4, atol=1e-5)
    # what main.compute_metrics does with it (main.py:116): features through compute_fid
    fid = metrics.compute_fid(model=m, images1=x, images2=torch.from_numpy(detgen.images(name + "/b", 4)).cuda())
    assert np.isfinite(fid)
    # and a train step on the restored module, against the oracle
    pool = "pool" in name
    cfg = dict(variational=False, latent_size=16, num_channels=8, act="gelu", sigmoid=True, no_pool=not pool,
               no_upsample=not pool, interpolation="bilinear" if pool else "nearest", kernel_size=3, base=8)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    losses = m.train_step(x, beta=0.0).cpu().double()
    o = ae_oracle.train_step_grads(params, x.cpu().numpy(), None, 0.0, cfg)
    assert abs(float(losses[2]) - float(o["total"])) <= 1e-4 * float(o["total"])
    for k, p in m.named_parameters():
        ref = o["grads"][k].double()
        err = float((p.grad.cpu().double() - ref).norm() / max(float(ref.norm()), 1e-30))
        assert err <= 1e-4, (k, err)


def test_fid_eps_retry_branch(monkeypatch):
    """metrics.py:94-97 of the reference: a non-finite first pass adds eps to both covariance diagonals and retries.
    Drive the branch (first kernel result forced to NaN) and check the retry's value against the oracle."""
    import metrics
    from gmk import ops
    from oracle import metrics_oracle
    from test_metrics_gpu import _fid_features
    f1, f2 = _fid_features("fid/set1", 500, 256, 0.5, 1.0), _fid_features("fid/set2", 500, 256, 0.7, 0.8)
    m1, m2 = metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2)
    dm = [(m[0], torch.from_numpy(m[1]).cuda(), torch.from_numpy(m[2]).cuda()) for m in (m1, m2)]
    real, calls = ops.fid_from_moments, []

    def first_pass_singular(a, b, iters=50, delta=1e-12, diag_eps=0.0):
        calls.append(diag_eps)
        out = real(a, b, iters=iters, delta=delta, diag_eps=diag_eps)
        if len(calls) == 1:
            out[6] = float("nan")
        return out

    monkeypatch.setattr(ops, "fid_from_moments", first_pass_singular)
    eps = 1e-3
    fid = metrics.fid_from_moments(dm[0], dm[1], eps=eps)
    assert calls == [0.0, eps] and np.isfinite(fid)
    (mu1, s1), (mu2, s2) = metrics_oracle.stats_from_moments(*m1), metrics_oracle.stats_from_moments(*m2)
    from scipy.linalg import sqrtm
    off = np.eye(256) * eps
    d = mu1 - mu2   # metrics.py:96-97,106-108: the offset enters the matrix square root only, the traces stay un-offset
    ref = d.dot(d) + np.trace(s1) + np.trace(s2) - 2 * np.trace(sqrtm((s1 + off).dot(s2 + off)).real)
    assert abs(fid - ref) < 0.05, (fid, ref)
    assert abs(fid - float(real(dm[0], dm[1])[6])) > 1e-3               # and the offset did change the value


def test_encoder_input_gradient_and_retain_graph(train_golden):
    """The autograd path returns d loss / d image when the input requires grad, and survives a second backward."""
    from helpers import case_inputs, oracle_cfg
    from test_model_gpu import build
    case = train_golden["vae_act_gelu_b2"]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    m.train()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    et = torch.from_numpy(eps).cuda()
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: et
    try:
        recon = m(xt)
    finally:
        torch.randn_like = orig
    loss = ((recon - 0.25) ** 2).mean()
    loss.backward(retain_graph=True)
    g1 = xt.grad.clone()
    xt.grad = None
    loss.backward()
    assert torch.allclose(xt.grad, g1)
    p = {k: torch.from_numpy(v).requires_grad_(False) for k, v in params.items()}
    xc = torch.from_numpy(x).requires_grad_(True)
    rc, _, _ = ae_oracle.autoencoder_forward(p, xc, torch.from_numpy(eps), **oracle_cfg(case))
    ((rc - 0.25) ** 2).mean().backward()
    err = float((g1.cpu() - xc.grad).norm() / xc.grad.norm())
    assert err <= 1e-4, err

"""GPU parity of the optimizer step: gmk_adamw_step (FusedAdamW.step) + CosineAnnealingLR, i.e.
`optim.step(); lr_scheduler.step()` of /root/reference/main.py:52-53, 214-215.

Goldens come from the imported reference (oracle/gen_golden.py): tests/golden/train_step.json
(`param_after_*`, `lr_after`: one step) and tests/golden/train_extra.json (`steps3_*`: three
consecutive steps with one optimizer, `refckpt_vae`: a resume from a reference-written checkpoint).
The kernel itself is also pinned against torch.optim.AdamW on cloned parameters / gradients.
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import case_inputs, sample, l2
from oracle import ae_oracle, detgen
from test_model_gpu import build

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _opt(model, T_max, lr=1e-3, wd=0.01):
    from gmk.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=wd)
    return opt, torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T_max)


def _close_params(model, ref_sample, ref_norm, atol=2e-6, rtol_norm=2e-6):
    for k, p in model.named_parameters():
        got = p.detach().cpu().numpy()
        assert np.allclose(sample(got), ref_sample[k], rtol=1e-5, atol=atol), (k, sample(got)[:4], ref_sample[k][:4])
        assert abs(l2(got) - ref_norm[k]) <= rtol_norm * ref_norm[k] + 1e-9, (k, l2(got), ref_norm[k])


@pytest.mark.parametrize("name", ["ae_cfg1_b8", "vae_cfg2_b8"])
def test_one_step_matches_reference_golden(train_golden, name):
    """train_step -> FusedAdamW(lr 1e-3, wd 0.01).step() -> CosineAnnealingLR(T_max=1000).step()."""
    case = train_golden[name]
    params, x, eps = case_inputs(case)
    m = build(case, params)
    opt, sched = _opt(m, 1000)
    m.train_step(torch.from_numpy(x).cuda(), beta=case["beta"], eps=None if eps is None else torch.from_numpy(eps).cuda())
    opt.step()
    sched.step()
    _close_params(m, case["param_after_sample"], case["param_after_norm"])
    assert abs(sched.get_last_lr()[0] - case["lr_after"]) < 1e-12
    # state bookkeeping is torch's: a float32 scalar `step` per parameter, exp_avg / exp_avg_sq shaped like the parameter
    st = opt.state_dict()["state"]
    assert len(st) == len(list(m.parameters()))
    assert all(float(s["step"]) == 1.0 and s["exp_avg"].shape == p.shape for s, p in zip(st.values(), m.parameters()))


@pytest.mark.parametrize("name", ["ae_cfg1_b8", "vae_cfg2_b8"])
def test_three_steps_match_reference_golden(train_golden, extra_golden, name):
    """Three consecutive steps with ONE optimizer and a fast-moving cosine schedule (T_max=5)."""
    case, gold = train_golden[name], extra_golden[f"steps3_{name}"]
    params, _, _ = case_inputs(case)
    m = build(case, params)
    opt, sched = _opt(m, gold["T_max"])
    variational = case["cfg"]["variational"]
    for i, ref in enumerate(gold["steps"]):
        x = torch.from_numpy(detgen.images(f"{name}/step{i}", 8)).cuda()
        eps = torch.from_numpy(detgen.normal(f"{name}/eps{i}", (8, 256))).cuda() if variational else None
        losses = m.train_step(x, beta=gold["beta"] if variational else 0.0, eps=eps).cpu().double()
        opt.step()
        sched.step()
        assert abs(float(losses[0]) - ref["rec"]) <= 1e-4 * ref["rec"], (i, losses, ref)
        if variational:
            assert abs(gold["beta"] * float(losses[1]) - ref["beta_kl"]) <= 1e-4 * ref["beta_kl"]
        assert abs(sched.get_last_lr()[0] - ref["lr_after"]) < 1e-12
        for k, p in m.named_parameters():
            assert abs(l2(p.detach().cpu().numpy()) - ref["param_norm"][k]) <= 1e-5 * ref["param_norm"][k], (i, k)
    _close_params(m, gold["param_sample"], gold["steps"][-1]["param_norm"], atol=1e-5, rtol_norm=1e-5)
    names = [k for k, _ in m.named_parameters()]
    st = opt.state_dict()["state"]
    for i, k in enumerate(names):
        assert float(st[i]["step"]) == gold["state_step"] == 3.0
        for key in ("exp_avg", "exp_avg_sq"):
            got = st[i][key].cpu().numpy()
            ref_n = gold[key + "_norm"][k]
            assert abs(l2(got) - ref_n) <= 2e-4 * ref_n, (k, key, l2(got), ref_n)   # gradients carry the 1e-4 bar
            scale = ref_n / math.sqrt(got.size)
            assert np.max(np.abs(sample(got) - np.asarray(gold[key + "_sample"][k]))) <= 1e-3 * scale + 1e-12, (k, key)


def _tiny_model():
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    torch.manual_seed(3)
    return Autoencoder(variational=True, latent_size=16, num_channels=8, kernel_size=3, act=get_activation_function("gelu"),
                       sigmoid=True, no_pool=True, no_upsample=True, device="cuda:0").to("cuda:0")


def test_kernel_matches_torch_adamw_over_a_run():
    """The kernel against torch.optim.AdamW on cloned parameters and gradients (no model arithmetic involved):
    6 steps, cosine LR (changes every step), grad_scale = 1/8 (the DP mean fold), a state_dict round trip into a
    fresh FusedAdamW mid-run, and a model.cpu().cuda() re-home of the flat views.  Bar: 1e-6 relative on the
    parameters and both moment buffers, per tensor."""
    from gmk.optim import FusedAdamW
    m = _tiny_model()
    ref_p = [p.detach().cpu().clone().requires_grad_(True) for p in m.parameters()]
    ref = torch.optim.AdamW(ref_p, lr=2e-3, weight_decay=0.05, betas=(0.8, 0.95), eps=1e-7)
    ref_s = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=7)
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.05, betas=(0.8, 0.95), eps=1e-7)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=7)
    opt.grad_scale = 0.125
    g = torch.Generator().manual_seed(11)

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))

    for step in range(6):
        for p, q in zip(m.parameters(), ref_p):
            grad = torch.randn(q.shape, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g)))
            q.grad = grad * 0.125
            p.grad.copy_(grad.cuda())
        opt.step(); sched.step()
        ref.step(); ref_s.step()
        assert opt._views_ok()
        assert abs(sched.get_last_lr()[0] - ref_s.get_last_lr()[0]) < 1e-15
        for p, q in zip(m.parameters(), ref_p):
            assert rel(p, q) <= 1e-6, (step, rel(p, q))
            assert rel(opt.state[p]["exp_avg"], ref.state[q]["exp_avg"]) <= 1e-6
            assert rel(opt.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"]) <= 1e-6
            assert float(opt.state[p]["step"]) == float(ref.state[q]["step"]) == step + 1
        if step == 1:      # save / load round trip into a new optimizer + scheduler (main.py:217-220 on resume)
            sd, ssd = opt.state_dict(), sched.state_dict()
            sd = {"state": {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()},
                  "param_groups": sd["param_groups"]}
            opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.05, betas=(0.8, 0.95), eps=1e-7)
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=7)
            opt.load_state_dict(sd)
            sched.load_state_dict(ssd)
            opt.grad_scale = 0.125
            assert opt._views_ok()
        if step == 3:      # re-home: nn.Module._apply re-assigns p.data / p.grad, step() must notice and re-flatten
            m.cpu(); m.cuda()
            assert not opt._views_ok()


def test_unsupported_adamw_flags_are_refused():
    from gmk.optim import FusedAdamW
    m = _tiny_model()
    for flag in ("amsgrad", "maximize", "capturable", "differentiable"):
        with pytest.raises(NotImplementedError):
            FusedAdamW(m.parameters(), **{flag: True})
    with pytest.raises(TypeError):
        FusedAdamW(m.parameters(), nesterov=True)


def test_resume_from_reference_written_checkpoint(extra_golden):
    """tests/golden/refckpt_vae.pth was written by the REFERENCE (its Autoencoder, torch.optim.AdamW and
    CosineAnnealingLR after two steps; keys of main.py:323-331).  Resume exactly as main.py:217-222 does, take the
    third step on the GPU and land where the reference landed."""
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    gold = extra_golden["refckpt_vae"]
    ck = torch.load(os.path.join(GOLDEN, "refckpt_vae.pth"), weights_only=True)
    assert set(ck) == {"state_dict", "optimizer", "lr_scheduler", "epoch", "current_beta", "losses"}
    m = Autoencoder(variational=True, latent_size=16, num_channels=8, kernel_size=3, act=get_activation_function("gelu"),
                    sigmoid=True, interpolation="nearest", no_pool=True, no_upsample=True, device="cuda:0")
    m.load_state_dict(ck["state_dict"])
    m = m.to("cuda:0")
    opt, sched = _opt(m, 6)
    opt.load_state_dict(ck["optimizer"])
    sched.load_state_dict(ck["lr_scheduler"])
    beta = ck["current_beta"]
    assert abs(beta - gold["beta"]) < 1e-15
    x = torch.from_numpy(detgen.images("refckpt_vae/train2", 4)).cuda()
    eps = torch.from_numpy(detgen.normal("refckpt_vae/eps_train2", (4, 16))).cuda()
    losses = m.train_step(x, beta=beta, eps=eps).cpu().double()
    opt.step(); sched.step()
    assert abs(float(losses[0]) - gold["rec"]) <= 1e-4 * gold["rec"]
    assert abs(beta * float(losses[1]) - gold["beta_kl"]) <= 1e-4 * gold["beta_kl"]
    assert abs(sched.get_last_lr()[0] - gold["lr_after"]) < 1e-12
    _close_params(m, gold["param_sample"], gold["param_norm"], atol=5e-6, rtol_norm=1e-5)
    # and the other direction: what FusedAdamW saves loads into a stock torch.optim.AdamW (the reference's class)
    stock = torch.optim.AdamW([p.detach().cpu().clone().requires_grad_(True) for p in m.parameters()], lr=1e-3, weight_decay=0.01)
    stock.load_state_dict(opt.state_dict())
    for (p, q) in zip(m.parameters(), stock.param_groups[0]["params"]):
        assert torch.equal(stock.state[q]["exp_avg"], opt.state[p]["exp_avg"].cpu())
        assert float(stock.state[q]["step"]) == 3.0


def test_early_bucket_updates_equal_one_step():
    """train_step(optimizer=...) updates the decoder and encoder.fc buckets under the encoder's backward and step() the rest:
    parameters, moments and step counts after 5 steps are bit-identical to train_step() + one full step()."""
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    from gmk.optim import FusedAdamW
    dev = torch.device("cuda:0")

    def run(early):
        torch.manual_seed(0)
        m = Autoencoder(variational=True, latent_size=256, num_channels=64, kernel_size=3, act=get_activation_function("gelu"),
                        sigmoid=True, interpolation="nearest", no_pool=True, no_upsample=True, device=dev).to(dev)
        opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20)
        g = torch.Generator(device=dev).manual_seed(3)
        for _ in range(5):
            x = torch.rand(64, 3, 32, 32, device=dev, generator=g)
            eps = torch.randn(64, 256, device=dev, generator=g)
            m.train_step(x, beta=1e-3, eps=eps, optimizer=opt if early else None)
            opt.step()
            sched.step()
        torch.cuda.synchronize()
        steps = {float(opt.state[p]["step"]) for p in opt.param_groups[0]["params"]}
        return opt.flat_param.clone(), opt.flat_m.clone(), opt.flat_v.clone(), steps

    pa, ma, va, sa = run(False)
    pb, mb, vb, sb = run(True)
    assert sa == sb == {5.0}
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)

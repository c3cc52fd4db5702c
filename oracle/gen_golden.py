"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference *unmodified* from /root/reference (torchvision, which the reference
uses for plotting only and which is not installed here, is registered as an empty module),
feeds it the closed-form tensors of oracle/detgen.py, and writes small JSON fixtures under
tests/golden/.  It also asserts, while generating, that the CPU restatement in oracle/
agrees with the reference (<= 1e-6 relative on losses, 1e-5 on gradients).

    python oracle/gen_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)

# --- import the reference ------------------------------------------------------------
for name in ("torchvision", "torchvision.transforms", "torchvision.utils"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF)
import metrics as ref_metrics                      # noqa: E402  /root/reference/metrics.py
from models.autoencoders import Autoencoder as RefAutoencoder  # noqa: E402
from utils import get_activation_function as ref_get_act       # noqa: E402
from models.classifiers import Classifier as RefClassifier      # noqa: E402
sys.path.remove(REF)

from oracle import detgen, ae_oracle, metrics_oracle, classifier_oracle  # noqa: E402

torch.set_num_threads(8)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sample(t, k=24):
    f = np.asarray(t, dtype=np.float64).reshape(-1)
    stride = max(1, f.size // k)
    return [float(v) for v in f[::stride][:k]]


def l2(t):
    return float(np.sqrt((np.asarray(t, dtype=np.float64) ** 2).sum()))


def build_ref(cfg, params, act):
    m = RefAutoencoder(variational=cfg["variational"], latent_size=cfg["latent_size"],
                       num_channels=cfg["num_channels"], kernel_size=cfg["kernel_size"],
                       act=ref_get_act(act), sigmoid=cfg["sigmoid"], interpolation=cfg["interpolation"],
                       no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"], device="cpu")
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    assert list(m.state_dict().keys()) == list(sd.keys()), (list(m.state_dict().keys()), list(sd.keys()))
    m.load_state_dict(sd)
    return m


def ref_step(model, x, eps, beta, variational, mse_prior, lr=None, wd=None, t_max=None):
    """main.py:27-53 restated around the imported model; eps is injected by patching
    torch.randn_like for the duration of the forward."""
    import torch.nn.functional as F
    model.train()
    orig = torch.randn_like
    if eps is not None:
        torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        recon = model(x)
    finally:
        torch.randn_like = orig
    rec = F.mse_loss(recon, x)
    rec_val = float(rec.detach())
    loss = rec
    kl_val = None
    if variational:
        mu = model.stats[:, :, 0]
        log_var = model.stats[:, :, 1]
        sigma = torch.exp(0.5 * log_var)
        if mse_prior:
            mu_loss = F.mse_loss(mu, torch.zeros_like(mu), reduction="none")
            sigma_loss = F.mse_loss(sigma, torch.ones_like(sigma), reduction="none")
            kl = mu_loss.sum(1).mean(0) + sigma_loss.sum(1).mean(0)
        else:
            kl = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
        kl_val = float(kl.detach())
        loss = loss + beta * kl
    out = dict(rec=rec_val, kl=kl_val, total=float(loss.detach()))
    optim = sched = None
    if lr is not None:
        optim = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=t_max)
        optim.zero_grad()
    else:
        model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    out.update(recon=recon.detach(), z=model.latent_vector.detach(),
               stats=model.stats.detach() if variational else None, grads=grads)
    if optim is not None:
        optim.step()
        sched.step()
        out["params_after"] = {k: p.detach().clone() for k, p in model.named_parameters()}
        out["lr_after"] = sched.get_last_lr()[0]
    return out


def run_case(name, cfg, act, B, beta=5e-6, mse_prior=False, adamw=False, hw=32, digest_only=False):
    shape_cfg = dict(variational=cfg["variational"], latent_size=cfg["latent_size"],
                     num_channels=cfg["num_channels"], kernel_size=cfg["kernel_size"],
                     no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"])
    params = detgen.ae_params(name, **shape_cfg)
    x = detgen.images(name, B, hw)
    eps = detgen.normal(name + "/eps", (B, cfg["latent_size"])) if cfg["variational"] else None
    xt = torch.from_numpy(x)
    et = None if eps is None else torch.from_numpy(eps)

    model = build_ref(cfg, params, act)
    kw = dict(lr=1e-3, wd=0.01, t_max=1000) if adamw else {}
    r = ref_step(model, xt, et, beta, cfg["variational"], mse_prior, **kw)

    # fp64 re-run of the reference (for the goldens' own error bar)
    model64 = build_ref(cfg, params, act).double()
    r64 = ref_step(model64, xt.double(), None if et is None else et.double(), beta, cfg["variational"], mse_prior)

    # restatement check
    ocfg = dict(variational=cfg["variational"], latent_size=cfg["latent_size"], num_channels=cfg["num_channels"],
                act=act, sigmoid=cfg["sigmoid"], no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"],
                interpolation=cfg["interpolation"], kernel_size=cfg["kernel_size"], base=hw // 4)
    o = ae_oracle.train_step_grads(params, x, eps, beta, ocfg, mse_prior=mse_prior)
    assert abs(float(o["total"]) - r["total"]) <= 1e-6 * abs(r["total"]), (name, float(o["total"]), r["total"])
    for k, g in r["grads"].items():
        d = l2(o["grads"][k] - g)
        assert d <= 1e-5 * max(l2(g), 1e-12), (name, k, d, l2(g))
    assert torch.allclose(o["recon"], r["recon"], atol=1e-6)

    rec = dict(name=name, cfg=cfg, act=act, B=B, hw=hw, beta=beta, mse_prior=mse_prior,
               rec=r["rec"], kl=r["kl"], total=r["total"],
               rec64=r64["rec"], kl64=r64["kl"], total64=r64["total"],
               grad_norm={k: l2(v) for k, v in r["grads"].items()},
               grad_norm64={k: l2(v) for k, v in r64["grads"].items()})
    if not digest_only:
        rec.update(recon_sample=sample(r["recon"]), z_sample=sample(r["z"]),
                   stats_sample=None if r["stats"] is None else sample(r["stats"]),
                   grad_sample={k: sample(v) for k, v in r["grads"].items()})
    if adamw:
        rec["lr_after"] = r["lr_after"]
        rec["param_after_norm"] = {k: l2(v) for k, v in r["params_after"].items()}
        rec["param_after_sample"] = {k: sample(v) for k, v in r["params_after"].items()}
        # restated AdamW + cosine
        for k, pa in r["params_after"].items():
            p0 = torch.from_numpy(params[k])
            p1, _, _ = ae_oracle.adamw_step(p0, r["grads"][k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.01)
            assert torch.allclose(p1, pa, rtol=1e-6, atol=1e-8), (name, k)
        assert abs(ae_oracle.cosine_lr(1e-3, 1, 1000) - r["lr_after"]) < 1e-12
    print(f"{name:28s} rec {r['rec']:.8f} kl {r['kl']} total {r['total']:.8f}")
    return rec


def base_cfg(**over):
    c = dict(variational=True, latent_size=256, num_channels=64, kernel_size=3, sigmoid=True,
             interpolation="nearest", no_pool=True, no_upsample=True)
    c.update(over)
    return c


def gen_train():
    cases = []
    cases.append(run_case("ae_cfg1_b8", base_cfg(variational=False), "gelu", 8, adamw=True))
    cases.append(run_case("vae_cfg2_b8", base_cfg(), "gelu", 8, adamw=True))
    cases.append(run_case("vae_cfg2_b50", base_cfg(), "gelu", 50, digest_only=True))
    cases.append(run_case("vae_cfg2_b256", base_cfg(), "gelu", 256, digest_only=True))
    cases.append(run_case("vae_cfg2_msep_b8", base_cfg(), "gelu", 8, mse_prior=True))
    cases.append(run_case("vae_cfg2_bigbeta_b8", base_cfg(), "gelu", 8, beta=0.05))
    cases.append(run_case("ae_nosigmoid_b4", base_cfg(variational=False, sigmoid=False, num_channels=32, latent_size=64), "relu", 4))
    small = dict(num_channels=16, latent_size=32)
    cases.append(run_case("vae_pool_nearest_b4", base_cfg(no_pool=False, no_upsample=False, **small), "gelu", 4))
    cases.append(run_case("vae_pool_bilinear_b4", base_cfg(no_pool=False, no_upsample=False, interpolation="bilinear", **small), "gelu", 4))
    cases.append(run_case("vae_pool_bicubic_b4", base_cfg(no_pool=False, no_upsample=False, interpolation="bicubic", **small), "swish", 4))
    cases.append(run_case("vae_pool_deconv_b4", base_cfg(no_pool=False, no_upsample=True, **small), "gelu", 4))
    cases.append(run_case("vae_stride_upsample_b4", base_cfg(no_pool=True, no_upsample=False, **small), "gelu", 4))
    for act in ae_oracle.ACTS:
        cases.append(run_case(f"vae_act_{act}_b2", base_cfg(num_channels=8, latent_size=16), act, 2))
    cases.append(run_case("vae_k5_b2", base_cfg(num_channels=8, latent_size=16, kernel_size=5), "gelu", 2))
    with open(os.path.join(OUT, "train_step.json"), "w") as f:
        json.dump(cases, f, separators=(',', ':'))


def fid_features(tag, n, d, shift, scale):
    """Correlated synthetic features: g @ A + b with a fixed D x D mixing matrix."""
    g = detgen.normal(tag + "/g", (n, d))
    a = detgen.uniform(tag + "/A", (d, d), -1.0, 1.0) / np.sqrt(d)
    a = a + np.eye(d, dtype=np.float32) * scale
    b = detgen.uniform(tag + "/b", (d,), -1.0, 1.0) * shift
    return (g @ a + b).astype(np.float32)


def smooth_images(tag, n, hw=32):
    """Low-frequency images in [0,1): per-image random plane waves plus 10% noise."""
    ph = detgen.uniform(tag + "/phase", (n, 3, 3), 0.0, 6.2831853)
    yy, xx = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    out = np.empty((n, 3, hw, hw), np.float32)
    for c in range(3):
        w = (0.15 + 0.1 * c)
        out[:, c] = 0.45 + 0.35 * np.sin(w * xx[None] * np.cos(ph[:, c, 0, None, None]) + w * yy[None] * np.sin(ph[:, c, 1, None, None]) + ph[:, c, 2, None, None])
    out += 0.1 * detgen.uniform(tag + "/noise", (n, 3, hw, hw), 0.0, 1.0)
    return np.clip(out, 0.0, 0.999).astype(np.float32)


class _FeatModel(torch.nn.Module):
    """Feeds precomputed features through metrics.compute_fid's model interface."""
    def forward(self, x):
        self.latent_vector = x
        return x


def gen_metrics():
    out = {}
    fm = _FeatModel()
    for n in (500, 5000, 50000):
        f1 = fid_features("fid/set1", n, 256, 0.5, 1.0)
        f2 = fid_features("fid/set2", n, 256, 0.7, 0.8)
        ref = float(ref_metrics.compute_fid(model=fm, images1=torch.from_numpy(f1), images2=torch.from_numpy(f2)))
        mine = float(metrics_oracle.fid_from_features(f1, f2))
        assert abs(ref - mine) < 1e-9 * max(1, abs(ref)), (ref, mine)
        ns = metrics_oracle.fid_from_moments(metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2))
        print(f"fid n={n}: reference {ref:.6f} restated {mine:.6f} chunked+newton-schulz {ns:.6f}")
        assert abs(ns - ref) < 0.05
        out[f"fid_n{n}"] = dict(n=n, d=256, fid=ref, fid_chunked_ns=ns)
    # small-D case, and a rank-deficient one (N < D): records what the reference returns
    f1 = fid_features("fid/small1", 300, 16, 0.5, 1.0)
    f2 = fid_features("fid/small2", 300, 16, 0.1, 0.5)
    out["fid_small"] = dict(n=300, d=16, fid=float(ref_metrics.compute_fid(model=fm, images1=torch.from_numpy(f1), images2=torch.from_numpy(f2))))
    f1 = fid_features("fid/def1", 100, 256, 0.5, 1.0)
    f2 = fid_features("fid/def2", 100, 256, 0.7, 0.8)
    ref = float(ref_metrics.compute_fid(model=fm, images1=torch.from_numpy(f1), images2=torch.from_numpy(f2)))
    ns = metrics_oracle.fid_from_moments(metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2))
    print(f"fid rank-deficient n=100: reference {ref:.6f} chunked+newton-schulz {ns:.6f}")
    out["fid_rank_deficient"] = dict(n=100, d=256, fid=ref, fid_chunked_ns=ns)

    # end-to-end compute_fid through the reference autoencoder (features = latent_vector of the full model)
    cfg = base_cfg(variational=False)
    params = detgen.ae_params("fid/ae", variational=False, latent_size=256, num_channels=64, kernel_size=3, no_pool=True, no_upsample=True)
    params["encoder.fc.weight"] = params["encoder.fc.weight"] * np.float32(40.0)  # spread the features out
    model = build_ref(cfg, params, "gelu")
    im1 = torch.from_numpy(detgen.images("fid/im1", 320))
    im2 = torch.from_numpy(smooth_images("fid/im2", 320))
    ref = float(ref_metrics.compute_fid(model=model, images1=im1, images2=im2))
    out["fid_ae_e2e"] = dict(n=320, fid=ref, note="plain AE cfg1 weights detgen tag 'fid/ae' with encoder.fc.weight x40; images2 = smooth_images")
    print("fid e2e through reference AE:", ref)

    # IS known answers from the reference's own comments (metrics.py:246-247)
    class _P(torch.nn.Module):
        def forward(self, x):
            return x
    one_hot = torch.eye(3)
    unif = torch.full((3, 3), 0.33)
    kat1 = float(metrics_oracle.inception_score_from_probs(one_hot))
    kat2 = float(metrics_oracle.inception_score_from_probs(unif))
    print("IS KATs (expect 3, 1):", kat1, kat2)
    assert abs(kat1 - 3.0) < 1e-5 and abs(kat2 - 1.0) < 2e-2
    out["is_kat"] = dict(one_hot=kat1, uniform033=kat2)

    for n in (37, 500, 4096):
        logits = torch.from_numpy(detgen.normal(f"is/logits{n}", (n, 10)) * 3.0)
        clf = _P()
        is_ref = float(ref_metrics.compute_inception_score(clf, logits))
        conf_ref, div_ref = ref_metrics.compute_confidence_diversity(clf, logits)
        is_o = float(metrics_oracle.inception_score(logits))
        conf_o, div_o, hist = metrics_oracle.confidence_diversity(logits)
        assert abs(is_ref - is_o) < 1e-6 and abs(float(conf_ref) - float(conf_o)) < 1e-5 and abs(float(div_ref) - float(div_o)) < 1e-5
        out[f"is_n{n}"] = dict(n=n, inception_score=is_ref, confidence=float(conf_ref), diversity=float(div_ref),
                               hist=[int(v) for v in hist])
        print(f"IS n={n}: {is_ref:.6f} conf {float(conf_ref):.4f} div {float(div_ref):.4f}")
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_classifier():
    """Eval-mode forward of the reference Classifier (models/classifiers.py:10-65) on deterministic weights, and the
    reference metrics (metrics.py) driven by it: what main.compute_metrics does with the classifier (main.py:117-119)."""
    out = {}
    for act, L in (("gelu", 64), ("relu", 256)):
        args = types.SimpleNamespace(debug=False, latent_size=L, dropout=0.5, act_fn=ref_get_act(act))
        m = RefClassifier(args=args)
        params = detgen.classifier_params(f"clf/{act}{L}", L)
        sd = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items()}
        assert list(m.state_dict().keys()) == list(sd.keys()), (list(m.state_dict().keys()), list(sd.keys()))
        m.load_state_dict(sd)
        m.eval()
        x = torch.from_numpy(detgen.images(f"clf/{act}{L}/x", 8))
        with torch.no_grad():
            logits = m(x)
            latent = m.latent_vector
        o_logits, o_latent = classifier_oracle.classifier_forward({k: v for k, v in sd.items()}, x, act)
        assert (o_logits - logits).abs().max() <= 1e-5 * logits.abs().max(), (o_logits - logits).abs().max()
        assert (o_latent - latent).abs().max() <= 1e-5 * latent.abs().max()
        case = dict(act=act, latent_size=L, n=8, logits=[float(v) for v in logits.reshape(-1)],
                    latent_sample=sample(latent), latent_l2=l2(latent))
        # metrics through the reference functions with this classifier (N >= latent_size for a full-rank covariance
        # is not required by the reference's compute_fid itself; main.py:146-150 enforces it for its own batches)
        n = 96 if L == 64 else 320
        im1 = torch.from_numpy(detgen.images(f"clf/{act}{L}/im1", n))
        im2 = torch.from_numpy(smooth_images(f"clf/{act}{L}/im2", n))
        with torch.no_grad():
            case["fid"] = float(ref_metrics.compute_fid(model=m, images1=im1, images2=im2))
            case["inception_score"] = float(ref_metrics.compute_inception_score(m, im2))
            conf, div = ref_metrics.compute_confidence_diversity(m, im2)
            case["confidence"], case["diversity"] = float(conf), float(div)
        case["n_metrics"] = n
        print(f"classifier {act} L={L}: logits l2 {l2(logits):.5f} fid {case['fid']:.5f} IS {case['inception_score']:.5f} "
              f"conf {case['confidence']:.4f} div {case['diversity']:.4f}")
        out[f"{act}_L{L}"] = case
    with open(os.path.join(OUT, "classifier.json"), "w") as f:
        json.dump(out, f, indent=1)


# ---------------------------------------------------------------------------------------------------
# round 2: multi-step optimizer trajectory, eval-mode sampling, the epoch loop (train_one_epoch +
# compute_test_loss), and reference-written checkpoint / full-model pickle fixtures
class _Queue:
    """Replaces torch.randn_like / torch.randn while the reference runs: returns the queued tensors in order."""
    def __init__(self, items):
        self.items = list(items)

    def __call__(self, *a, **k):
        return self.items.pop(0)


def _with_patched(name, fn, body):
    orig = getattr(torch, name)
    setattr(torch, name, fn)
    try:
        return body()
    finally:
        setattr(torch, name, orig)


def ref_train_one_epoch(args, model, beta, batches, eps_list, optim, lr_scheduler):
    """/root/reference/main.py:21-70 restated around the imported model (main.py itself needs tensorboard and
    torchvision datasets to import).  `batches` replaces the DataLoader, `eps_list` the randn_like draws."""
    import torch.nn.functional as F
    model.train()
    total_train_rec_loss = 0
    total_train_kl_loss = 0
    q = _Queue(eps_list)
    for image in batches:
        reconstructed = _with_patched("randn_like", q, lambda: model(image))
        train_rec_loss = F.mse_loss(reconstructed, image)
        total_train_rec_loss += train_rec_loss
        train_loss = train_rec_loss
        if args.variational:
            mu = model.stats[:, :, 0]
            log_var = model.stats[:, :, 1]
            sigma = torch.exp(0.5 * log_var)
            if args.mse:
                mu_loss = F.mse_loss(mu, torch.zeros_like(mu), reduction='none')
                sigma_loss = F.mse_loss(sigma, torch.ones_like(sigma), reduction='none')
                train_kl_loss = mu_loss.sum(1).mean(0) + sigma_loss.sum(1).mean(0)
            else:
                train_kl_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
            total_train_kl_loss += beta * train_kl_loss
            train_loss += beta * train_kl_loss
        optim.zero_grad()
        train_loss.backward()
        optim.step()
        lr_scheduler.step()
    if args.variational:
        beta *= args.beta_mult
    return beta, total_train_rec_loss / len(batches), total_train_kl_loss / len(batches)


def ref_compute_test_loss(args, model, beta, batches, eps_list):
    """/root/reference/main.py:73-111 restated around the imported model."""
    import torch.nn.functional as F
    model.eval()
    q = _Queue(eps_list)
    with torch.no_grad():
        total_test_rec_loss = 0
        total_test_kl_loss = 0
        latent_mean = 0
        latent_std = 0
        for test_image in batches:
            test_reconstructed = _with_patched("randn_like", q, lambda: model(test_image))
            total_test_rec_loss += F.mse_loss(test_reconstructed, test_image)
            if args.variational:
                mu = model.stats[:, :, 0]
                log_var = model.stats[:, :, 1]
                sigma = torch.exp(0.5 * log_var)
                if args.mse:
                    mu_loss = F.mse_loss(mu, torch.zeros_like(mu), reduction='none')
                    sigma_loss = F.mse_loss(sigma, torch.ones_like(sigma), reduction='none')
                    test_kl_loss = mu_loss.sum(1).mean(0) + sigma_loss.sum(1).mean(0)
                else:
                    test_kl_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
                total_test_kl_loss += test_kl_loss
                latent_mean += mu.abs().mean()
                latent_std += sigma.abs().mean()
        n = len(batches)
        return latent_mean / n, latent_std / n, total_test_rec_loss / n, beta * total_test_kl_loss / n


def _epoch_inputs(tag, variational, L, nb_train, nb_test, B):
    tr = [detgen.images(f"{tag}/train{i}", B) for i in range(nb_train)]
    te = [detgen.images(f"{tag}/test{i}", B) for i in range(nb_test)]
    etr = [detgen.normal(f"{tag}/eps_train{i}", (B, L)) for i in range(nb_train)] if variational else []
    ete = [detgen.normal(f"{tag}/eps_test{i}", (B, L)) for i in range(nb_test)] if variational else []
    return tr, te, etr, ete


def gen_extra():
    out = {}
    T = torch.from_numpy

    # (1) three consecutive reference steps, ONE AdamW + CosineAnnealingLR (lr moves a lot: T_max=5)
    for name, variational in (("ae_cfg1_b8", False), ("vae_cfg2_b8", True)):
        cfg = base_cfg(variational=variational)
        shape_cfg = dict(variational=variational, latent_size=256, num_channels=64, kernel_size=3, no_pool=True, no_upsample=True)
        params = detgen.ae_params(name, **shape_cfg)
        model = build_ref(cfg, params, "gelu")
        optim = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=5)
        args = types.SimpleNamespace(variational=variational, mse=False, beta_mult=1.0)
        steps = []
        for i in range(3):
            x = T(detgen.images(f"{name}/step{i}", 8))
            eps = [T(detgen.normal(f"{name}/eps{i}", (8, 256)))] if variational else []
            _, rec, kl = ref_train_one_epoch(args, model, 5e-6, [x], eps, optim, sched)
            steps.append(dict(rec=float(rec), beta_kl=float(kl), lr_after=sched.get_last_lr()[0],
                              param_norm={k: l2(p.detach()) for k, p in model.named_parameters()}))
        st = optim.state_dict()["state"]
        names = [k for k, _ in model.named_parameters()]
        out[f"steps3_{name}"] = dict(
            steps=steps, T_max=5, lr=1e-3, wd=0.01, beta=5e-6,
            param_sample={k: sample(p.detach()) for k, p in model.named_parameters()},
            exp_avg_norm={names[i]: l2(s["exp_avg"]) for i, s in st.items()},
            exp_avg_sq_norm={names[i]: l2(s["exp_avg_sq"]) for i, s in st.items()},
            exp_avg_sample={names[i]: sample(s["exp_avg"]) for i, s in st.items()},
            exp_avg_sq_sample={names[i]: sample(s["exp_avg_sq"]) for i, s in st.items()},
            state_step=float(st[0]["step"]))
        print(f"steps3_{name}: lr {[s['lr_after'] for s in steps]}")

    # (2) eval-mode decoder: Autoencoder.sample(return_samples=True) with the torch.randn draw injected
    small = dict(num_channels=16, latent_size=32)
    for name, cfg, act in (("vae_cfg2_b8", base_cfg(), "gelu"),
                           ("vae_pool_bilinear_b4", base_cfg(no_pool=False, no_upsample=False, interpolation="bilinear", **small), "gelu"),
                           ("ae_nosigmoid_b4", base_cfg(variational=False, sigmoid=False, num_channels=32, latent_size=64), "relu")):
        shape_cfg = dict(variational=cfg["variational"], latent_size=cfg["latent_size"], num_channels=cfg["num_channels"],
                         kernel_size=3, no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"])
        model = build_ref(cfg, detgen.ae_params(name, **shape_cfg), act)
        z = T(detgen.normal(f"{name}/sample_z", (5, cfg["latent_size"])))
        with torch.no_grad():
            s = _with_patched("randn", _Queue([z]), lambda: model.sample(num_images=5, return_samples=True))
        assert not model.training
        out[f"sample_{name}"] = dict(n=5, sample=sample(s), l2=l2(s), shape=list(s.shape))
        print(f"sample_{name}: l2 {l2(s):.6f}")

    # (3) the epoch loop: train_one_epoch (3 batches) then compute_test_loss (2 batches)
    for name, variational, mse in (("epoch_vae", True, False), ("epoch_vae_msep", True, True), ("epoch_ae", False, False)):
        cfg = base_cfg(variational=variational, num_channels=32, latent_size=64)
        shape_cfg = dict(variational=variational, latent_size=64, num_channels=32, kernel_size=3, no_pool=True, no_upsample=True)
        model = build_ref(cfg, detgen.ae_params(name, **shape_cfg), "gelu")
        optim = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=6)
        args = types.SimpleNamespace(variational=variational, mse=mse, beta_mult=1.02)
        tr, te, etr, ete = _epoch_inputs(name, variational, 64, 3, 2, 8)
        beta0 = 0.01
        beta, tr_rec, tr_kl = ref_train_one_epoch(args, model, beta0, [T(a) for a in tr], [T(e) for e in etr], optim, sched)
        lm, ls, te_rec, te_kl = ref_compute_test_loss(args, model, beta, [T(a) for a in te], [T(e) for e in ete])
        out[name] = dict(variational=variational, mse=mse, B=8, nb_train=3, nb_test=2, beta0=beta0, beta_mult=1.02, T_max=6,
                         beta_after=float(beta), train_rec=float(tr_rec), train_kl=float(tr_kl), lr_after=sched.get_last_lr()[0],
                         latent_mean=float(lm), latent_std=float(ls), test_rec=float(te_rec), test_kl=float(te_kl),
                         param_norm={k: l2(p.detach()) for k, p in model.named_parameters()},
                         param_sample={k: sample(p.detach()) for k, p in model.named_parameters()})
        print(f"{name}: beta {beta:.6f} train rec {float(tr_rec):.6f} kl {float(tr_kl):.6f} test rec {float(te_rec):.6f} "
              f"kl {float(te_kl):.6f} mean {float(lm):.5f} std {float(ls):.5f}")

    # (4) files written by the REFERENCE's classes: a full-model pickle (what --ae_checkpoint loads, main.py:189-195,338-339)
    # and a resume checkpoint (main.py:217-222, 323-334).  Tiny configs so that the fixtures stay small.
    tiny = dict(num_channels=8, latent_size=16)
    for name, cfg in (("refpickle_ae_deconv", base_cfg(variational=False, **tiny)),
                      ("refpickle_ae_pool_bilinear", base_cfg(variational=False, no_pool=False, no_upsample=False,
                                                              interpolation="bilinear", **tiny))):
        shape_cfg = dict(variational=False, latent_size=16, num_channels=8, kernel_size=3, no_pool=cfg["no_pool"],
                         no_upsample=cfg["no_upsample"])
        model = build_ref(cfg, detgen.ae_params(name, **shape_cfg), "gelu")
        model.eval()
        x = T(detgen.images(name, 4))
        with torch.no_grad():
            recon = model(x)
        torch.save(model, os.path.join(OUT, name + ".pth"))   # full-model pickle of models.autoencoders.Autoencoder
        out[name] = dict(recon_sample=sample(recon), recon_l2=l2(recon), latent_sample=sample(model.latent_vector),
                         keys=list(model.state_dict().keys()))
    name = "refckpt_vae"
    cfg = base_cfg(**tiny)
    shape_cfg = dict(variational=True, latent_size=16, num_channels=8, kernel_size=3, no_pool=True, no_upsample=True)
    model = build_ref(cfg, detgen.ae_params(name, **shape_cfg), "gelu")
    optim = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=6)
    args = types.SimpleNamespace(variational=True, mse=False, beta_mult=1.02)
    tr, _, etr, _ = _epoch_inputs(name, True, 16, 3, 0, 4)
    beta, _, _ = ref_train_one_epoch(args, model, 0.01, [T(a) for a in tr[:2]], [T(e) for e in etr[:2]], optim, sched)
    ck = dict(state_dict=model.state_dict(), optimizer=optim.state_dict(), lr_scheduler=sched.state_dict(), epoch=0,
              current_beta=beta, losses="")     # keys of main.py:323-331 minus `args` (it pickles an nn.Module instance)
    torch.save(ck, os.path.join(OUT, name + ".pth"))
    args.beta_mult = 1.0
    _, rec, kl = ref_train_one_epoch(args, model, beta, [T(tr[2])], [T(etr[2])], optim, sched)   # the step after resume
    out[name] = dict(beta=float(beta), rec=float(rec), beta_kl=float(kl), lr_after=sched.get_last_lr()[0],
                     param_sample={k: sample(p.detach()) for k, p in model.named_parameters()},
                     param_norm={k: l2(p.detach()) for k, p in model.named_parameters()})
    with open(os.path.join(OUT, "train_extra.json"), "w") as f:
        json.dump(out, f, separators=(',', ':'))
    print("extra goldens written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        gen_extra()
        raise SystemExit
    if len(sys.argv) > 1 and sys.argv[1] == "classifier":
        gen_classifier()
        raise SystemExit
    gen_train()
    gen_metrics()
    gen_classifier()
    gen_extra()
    print("goldens written to", OUT)

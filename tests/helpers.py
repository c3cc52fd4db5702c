"""Shared helpers for parity tests: rebuild a golden case's inputs from oracle/detgen.py."""
import numpy as np

from oracle import detgen


def sample(t, k=24):
    f = np.asarray(t, dtype=np.float64).reshape(-1)
    stride = max(1, f.size // k)
    return f[::stride][:k]


def l2(t):
    return float(np.sqrt((np.asarray(t, dtype=np.float64) ** 2).sum()))


def case_inputs(case):
    cfg = case["cfg"]
    shape_cfg = dict(variational=cfg["variational"], latent_size=cfg["latent_size"],
                     num_channels=cfg["num_channels"], kernel_size=cfg["kernel_size"],
                     no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"])
    name = case["name"]
    params = detgen.ae_params(name, **shape_cfg)
    x = detgen.images(name, case["B"], case["hw"])
    eps = detgen.normal(name + "/eps", (case["B"], cfg["latent_size"])) if cfg["variational"] else None
    return params, x, eps


def oracle_cfg(case):
    cfg = case["cfg"]
    return dict(variational=cfg["variational"], latent_size=cfg["latent_size"], num_channels=cfg["num_channels"],
                act=case["act"], sigmoid=cfg["sigmoid"], no_pool=cfg["no_pool"], no_upsample=cfg["no_upsample"],
                interpolation=cfg["interpolation"], kernel_size=cfg["kernel_size"], base=case["hw"] // 4)


def check_against_golden(case, out, rtol_loss=1e-4, rtol_grad=1e-4):
    """`out`: dict with rec, kl, total (floats) and grads {key: array}.  Tolerances are
    BASELINE.json's: 1e-4 relative on per-step loss and (per-tensor rel-L2) gradients."""
    assert abs(out["total"] - case["total"]) <= rtol_loss * abs(case["total"]), (out["total"], case["total"])
    assert abs(out["rec"] - case["rec"]) <= rtol_loss * abs(case["rec"])
    if case["kl"] is not None:
        assert abs(out["kl"] - case["kl"]) <= rtol_loss * abs(case["kl"]), (out["kl"], case["kl"])
    for k, gn in case["grad_norm"].items():
        g = np.asarray(out["grads"][k], dtype=np.float64)
        assert abs(l2(g) - gn) <= rtol_grad * gn, (k, l2(g), gn)
        if "grad_sample" in case:
            ref = np.asarray(case["grad_sample"][k])
            got = sample(g)
            scale = gn / np.sqrt(g.size)  # rms magnitude of this tensor
            assert np.max(np.abs(got - ref)) <= 10 * rtol_grad * max(scale, np.max(np.abs(ref))), (k, got[:4], ref[:4])

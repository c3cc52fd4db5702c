"""CPU: the oracle (oracle/*.py) reproduces every golden vector generated from the reference."""
import numpy as np
import pytest
import torch

from oracle import ae_oracle, metrics_oracle, detgen
from helpers import case_inputs, oracle_cfg, check_against_golden, sample

CASES = ["ae_cfg1_b8", "vae_cfg2_b8", "vae_cfg2_b50", "vae_cfg2_msep_b8", "vae_cfg2_bigbeta_b8", "ae_nosigmoid_b4",
         "vae_pool_nearest_b4", "vae_pool_bilinear_b4", "vae_pool_bicubic_b4", "vae_pool_deconv_b4",
         "vae_stride_upsample_b4", "vae_k5_b2"] + [f"vae_act_{a}_b2" for a in ae_oracle.ACTS]


@pytest.mark.parametrize("name", CASES)
def test_train_step_matches_reference_golden(train_golden, name):
    case = train_golden[name]
    params, x, eps = case_inputs(case)
    o = ae_oracle.train_step_grads(params, x, eps, case["beta"], oracle_cfg(case), mse_prior=case["mse_prior"])
    out = dict(rec=float(o["rec"]), kl=None if o["kl"] is None else float(o["kl"]), total=float(o["total"]),
               grads={k: v.numpy() for k, v in o["grads"].items()})
    check_against_golden(case, out, rtol_loss=2e-6, rtol_grad=2e-5)
    if "recon_sample" in case:
        assert np.allclose(sample(o["recon"].numpy()), case["recon_sample"], atol=2e-6)
        assert np.allclose(sample(o["z"].numpy()), case["z_sample"], atol=1e-5, rtol=1e-5)


def test_adamw_and_cosine_match_reference_golden(train_golden):
    case = train_golden["vae_cfg2_b8"]
    params, x, eps = case_inputs(case)
    o = ae_oracle.train_step_grads(params, x, eps, case["beta"], oracle_cfg(case))
    for k, ref in case["param_after_sample"].items():
        p0 = torch.from_numpy(params[k])
        p1, m, v = ae_oracle.adamw_step(p0, o["grads"][k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.01)
        assert np.allclose(sample(p1.numpy()), ref, rtol=1e-5, atol=1e-7), k
    assert abs(ae_oracle.cosine_lr(1e-3, 1, 1000) - case["lr_after"]) < 1e-12
    lr = 1e-3
    for t in range(1, 6):
        lr = ae_oracle.cosine_lr_recursive(lr, 1e-3, t, 1000)
        assert abs(lr - ae_oracle.cosine_lr(1e-3, t, 1000)) < 1e-12


def _fid_features(tag, n, d, shift, scale):
    g = detgen.normal(tag + "/g", (n, d))
    a = detgen.uniform(tag + "/A", (d, d), -1.0, 1.0) / np.sqrt(d)
    a = a + np.eye(d, dtype=np.float32) * scale
    b = detgen.uniform(tag + "/b", (d,), -1.0, 1.0) * shift
    return (g @ a + b).astype(np.float32)


@pytest.mark.parametrize("n", [500, 5000])
def test_fid_matches_reference_golden(metrics_golden, n):
    f1 = _fid_features("fid/set1", n, 256, 0.5, 1.0)
    f2 = _fid_features("fid/set2", n, 256, 0.7, 0.8)
    gold = metrics_golden[f"fid_n{n}"]
    assert abs(metrics_oracle.fid_from_features(f1, f2) - gold["fid"]) < 1e-6 * gold["fid"]
    ns = metrics_oracle.fid_from_moments(metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2))
    assert abs(ns - gold["fid"]) < 0.05          # BASELINE.json: 0.05 absolute on FID
    assert abs(ns - gold["fid_chunked_ns"]) < 1e-6


def test_fid_rank_deficient(metrics_golden):
    f1 = _fid_features("fid/def1", 100, 256, 0.5, 1.0)
    f2 = _fid_features("fid/def2", 100, 256, 0.7, 0.8)
    ns = metrics_oracle.fid_from_moments(metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2))
    assert abs(ns - metrics_golden["fid_rank_deficient"]["fid"]) < 0.05


def test_inception_score_known_answers(metrics_golden):
    # the reference's own KATs, /root/reference/metrics.py:246-247
    assert abs(float(metrics_oracle.inception_score_from_probs(torch.eye(3))) - 3.0) < 1e-5
    assert abs(float(metrics_oracle.inception_score_from_probs(torch.full((3, 3), 0.33))) - 1.0) < 2e-2


@pytest.mark.parametrize("n", [37, 500, 4096])
def test_is_conf_div_match_reference_golden(metrics_golden, n):
    gold = metrics_golden[f"is_n{n}"]
    logits = torch.from_numpy(detgen.normal(f"is/logits{n}", (n, 10)) * 3.0)
    assert abs(float(metrics_oracle.inception_score(logits)) - gold["inception_score"]) < 1e-5
    conf, div, hist = metrics_oracle.confidence_diversity(logits)
    assert abs(float(conf) - gold["confidence"]) < 1e-4 and abs(float(div) - gold["diversity"]) < 1e-4
    assert [int(v) for v in hist] == gold["hist"]

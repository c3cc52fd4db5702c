"""Classifier eval forward: the CPU oracle against the reference's golden logits (CPU), and the
libgmk path (BatchNorm folded into conv / linear, activation in the epilogue) against the same goldens plus the
reference's metrics driven by the classifier (GPU)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import detgen, classifier_oracle

CASES = [("gelu", 64), ("relu", 256)]


def _smooth_images(tag, n, hw=32):
    """Same low-frequency images as oracle/gen_golden.py::smooth_images (pure function of the tag)."""
    ph = detgen.uniform(tag + "/phase", (n, 3, 3), 0.0, 6.2831853)
    yy, xx = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    out = np.empty((n, 3, hw, hw), np.float32)
    for c in range(3):
        w = (0.15 + 0.1 * c)
        out[:, c] = 0.45 + 0.35 * np.sin(w * xx[None] * np.cos(ph[:, c, 0, None, None]) + w * yy[None] * np.sin(ph[:, c, 1, None, None]) + ph[:, c, 2, None, None])
    out += 0.1 * detgen.uniform(tag + "/noise", (n, 3, hw, hw), 0.0, 1.0)
    return np.clip(out, 0.0, 0.999).astype(np.float32)


@pytest.mark.parametrize("act,L", CASES)
def test_oracle_matches_reference_golden(act, L):
    g = load_golden("classifier.json")[f"{act}_L{L}"]
    p = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in detgen.classifier_params(f"clf/{act}{L}", L).items()}
    x = torch.from_numpy(detgen.images(f"clf/{act}{L}/x", 8))
    logits, latent = classifier_oracle.classifier_forward(p, x, act)
    ref = np.asarray(g["logits"]).reshape(8, 10)
    assert np.abs(logits.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    assert abs(float(np.sqrt((latent.double().numpy() ** 2).sum())) - g["latent_l2"]) <= 1e-5 * g["latent_l2"]


def _build(act, L):
    from models.classifiers import Classifier
    from utils import get_activation_function
    args = types.SimpleNamespace(debug=False, latent_size=L, dropout=0.5, act_fn=get_activation_function(act))
    m = Classifier(args=args)
    sd = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in detgen.classifier_params(f"clf/{act}{L}", L).items()}
    assert list(m.state_dict().keys()) == list(sd.keys())  # the reference's state_dict layout
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize("act,L", CASES)
def test_gmk_forward_matches_reference_golden(act, L):
    g = load_golden("classifier.json")[f"{act}_L{L}"]
    m = _build(act, L)
    x = torch.from_numpy(detgen.images(f"clf/{act}{L}/x", 8)).cuda()
    logits = m(x).cpu().double().numpy()
    ref = np.asarray(g["logits"]).reshape(8, 10)
    assert np.abs(logits - ref).max() <= 1e-4 * np.abs(ref).max(), np.abs(logits - ref).max()
    lat = m.latent_vector.cpu().double().numpy()
    assert lat.shape == (8, L)
    assert abs(np.sqrt((lat ** 2).sum()) - g["latent_l2"]) <= 1e-4 * g["latent_l2"]
    # in-place parameter updates invalidate the folded-weight cache
    with torch.no_grad():
        m.fc_out.bias.add_(1.0)
        m.bn1.running_mean.mul_(1.0)
    assert np.abs(m(x).cpu().double().numpy() - (ref + 1.0)).max() <= 1e-4 * np.abs(ref).max() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("act,L", CASES)
def test_metrics_with_classifier_match_reference(act, L):
    import metrics
    g = load_golden("classifier.json")[f"{act}_L{L}"]
    m = _build(act, L)
    n = g["n_metrics"]
    im1 = torch.from_numpy(detgen.images(f"clf/{act}{L}/im1", n)).cuda()
    im2 = torch.from_numpy(_smooth_images(f"clf/{act}{L}/im2", n)).cuda()
    assert abs(metrics.compute_fid(model=m, images1=im1, images2=im2, batch_size=64) - g["fid"]) < 0.05
    assert abs(float(metrics.compute_inception_score(m, im2)) - g["inception_score"]) <= 1e-4 * g["inception_score"]
    conf, div = metrics.compute_confidence_diversity(m, im2)
    assert abs(float(conf) - g["confidence"]) < 1e-2 and abs(float(div) - g["diversity"]) < 1e-2


@pytest.mark.gpu
def test_eval_forward_has_no_cpu_path():
    from models.classifiers import Classifier
    from utils import get_activation_function
    args = types.SimpleNamespace(debug=False, latent_size=16, dropout=0.0, act_fn=get_activation_function("relu"))
    m = Classifier(args=args).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 32, 32))

"""GPU parity of the FID / IS / confidence-diversity path against the reference's golden values."""
import numpy as np
import pytest
import torch

from oracle import detgen, metrics_oracle

pytestmark = pytest.mark.gpu


def _fid_features(tag, n, d, shift, scale):
    g = detgen.normal(tag + "/g", (n, d))
    a = detgen.uniform(tag + "/A", (d, d), -1.0, 1.0) / np.sqrt(d)
    a = a + np.eye(d, dtype=np.float32) * scale
    b = detgen.uniform(tag + "/b", (d,), -1.0, 1.0) * shift
    return (g @ a + b).astype(np.float32)


class FeatModel(torch.nn.Module):
    def forward(self, x):
        self.latent_vector = x
        return x


@pytest.mark.parametrize("n,bs", [(500, 1024), (500, 128), (5000, 1024), (50000, 8192)])
def test_fid_matches_reference_golden(metrics_golden, n, bs):
    import metrics
    f1 = torch.from_numpy(_fid_features("fid/set1", n, 256, 0.5, 1.0)).cuda()
    f2 = torch.from_numpy(_fid_features("fid/set2", n, 256, 0.7, 0.8)).cuda()
    fid = metrics.compute_fid(model=FeatModel(), images1=f1, images2=f2, batch_size=bs)
    assert isinstance(fid, np.float64)
    assert abs(fid - metrics_golden[f"fid_n{n}"]["fid"]) < 0.05, (fid, metrics_golden[f"fid_n{n}"])


def test_moments_match_oracle_and_are_batching_invariant():
    from gmk import ops
    f = _fid_features("fid/set1", 3000, 256, 0.5, 1.0)
    n, s, G = metrics_oracle.moments_chunked(f)
    outs = []
    for bs in (3000, 700, 64):
        Gd = torch.zeros(256, 256, dtype=torch.float64, device="cuda")
        sd = torch.zeros(256, dtype=torch.float64, device="cuda")
        ft = torch.from_numpy(f).cuda()
        for i in range(0, 3000, bs):
            ops.gram_accumulate(ft[i:i + bs].contiguous(), Gd, sd)
        outs.append((Gd.cpu().numpy(), sd.cpu().numpy()))
        assert np.allclose(outs[-1][0], G, rtol=1e-12, atol=1e-9)
        assert np.allclose(outs[-1][1], s, rtol=1e-12, atol=1e-9)
    # sharding invariance (fp64, exact products): different batchings agree to ~1e-13 relative
    assert np.allclose(outs[0][0], outs[2][0], rtol=1e-13, atol=1e-10)


def test_gram_wide_features_on_the_unaligned_fallback_kernel():
    """D = 512 through the 64-tile fallback kernel (X not 16-byte aligned): it uses up to 32 slabs where the 128-tile plan
    has 25, so the workspace must be sized for the larger of the two plans (round-2 advisor finding: out-of-bounds slab
    writes).  Checked against the aligned path (exact fp64 sums of exact products: equal to ~1e-12 relative) and numpy."""
    from gmk import ops
    n, D = 16384, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    buf = torch.randn(n * D + 4, device="cuda", generator=g)
    xa = buf[:n * D].view(n, D)
    xu = buf[1:1 + n * D].view(n, D)          # 4-byte aligned only
    assert xu.data_ptr() % 16 != 0 and xu.is_contiguous()
    guard = torch.full((1 << 16,), 7.0, device="cuda", dtype=torch.float64)   # allocated next: a canary behind the workspace
    res = []
    for x in (xa, xu):
        G = torch.zeros(D, D, dtype=torch.float64, device="cuda")
        s = torch.zeros(D, dtype=torch.float64, device="cuda")
        ops.gram_accumulate(x, G, s)
        torch.cuda.synchronize()
        ref = x.double().T @ x.double()
        assert torch.allclose(G, ref, rtol=1e-11, atol=1e-9), float((G - ref).abs().max())
        assert torch.allclose(s, x.double().sum(0), rtol=1e-11, atol=1e-9)
        res.append(G)
    assert bool((guard == 7.0).all())


def test_fid_small_and_rank_deficient(metrics_golden):
    import metrics
    f1 = torch.from_numpy(_fid_features("fid/small1", 300, 16, 0.5, 1.0)).cuda()
    f2 = torch.from_numpy(_fid_features("fid/small2", 300, 16, 0.1, 0.5)).cuda()
    assert abs(metrics.compute_fid(FeatModel(), f1, f2) - metrics_golden["fid_small"]["fid"]) < 1e-3
    f1 = torch.from_numpy(_fid_features("fid/def1", 100, 256, 0.5, 1.0)).cuda()
    f2 = torch.from_numpy(_fid_features("fid/def2", 100, 256, 0.7, 0.8)).cuda()
    assert abs(metrics.compute_fid(FeatModel(), f1, f2) - metrics_golden["fid_rank_deficient"]["fid"]) < 0.05


def test_fid_newton_schulz_matches_oracle_algorithm():
    from gmk import ops
    f1 = _fid_features("fid/set1", 500, 256, 0.5, 1.0)
    f2 = _fid_features("fid/set2", 500, 256, 0.7, 0.8)
    m1, m2 = metrics_oracle.moments_chunked(f1), metrics_oracle.moments_chunked(f2)
    ref = metrics_oracle.fid_from_moments(m1, m2)
    dm = [(m[0], torch.from_numpy(m[1]).cuda(), torch.from_numpy(m[2]).cuda()) for m in (m1, m2)]
    out = ops.fid_from_moments(dm[0], dm[1]).cpu().numpy()
    assert abs(out[6] - ref) < 1e-6 * abs(ref), (out, ref)


def test_fid_end_to_end_through_autoencoder(metrics_golden):
    """compute_fid(model=AE, images...) against the value the imported reference produced."""
    import metrics
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    params = detgen.ae_params("fid/ae", variational=False, latent_size=256, num_channels=64, kernel_size=3,
                              no_pool=True, no_upsample=True)
    params["encoder.fc.weight"] = params["encoder.fc.weight"] * np.float32(40.0)
    m = Autoencoder(variational=False, latent_size=256, num_channels=64, kernel_size=3,
                    act=get_activation_function("gelu"), sigmoid=True, interpolation="nearest",
                    no_pool=True, no_upsample=True, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    im1 = torch.from_numpy(detgen.images("fid/im1", 320)).cuda()
    # smooth_images of oracle/gen_golden.py, restated (gen_golden imports the reference, which is absent here)
    tag, n, hw = "fid/im2", 320, 32
    ph = detgen.uniform(tag + "/phase", (n, 3, 3), 0.0, 6.2831853)
    yy, xx = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    out = np.empty((n, 3, hw, hw), np.float32)
    for c in range(3):
        w = (0.15 + 0.1 * c)
        out[:, c] = 0.45 + 0.35 * np.sin(w * xx[None] * np.cos(ph[:, c, 0, None, None]) + w * yy[None] * np.sin(ph[:, c, 1, None, None]) + ph[:, c, 2, None, None])
    out += 0.1 * detgen.uniform(tag + "/noise", (n, 3, hw, hw), 0.0, 1.0)
    im2 = torch.from_numpy(np.clip(out, 0.0, 0.999).astype(np.float32)).cuda()
    gold = metrics_golden["fid_ae_e2e"]["fid"]
    fid = metrics.compute_fid(model=m, images1=im1, images2=im2, batch_size=128)
    assert abs(fid - gold) < 0.05 and abs(fid - gold) < 2e-3 * abs(gold), (fid, gold)
    # full-model path (model(x) sets .latent_vector) gives the same number as the encoder-only fast path
    class Wrap(torch.nn.Module):
        def __init__(self, inner):
            super().__init__(); self.inner = inner
        def forward(self, x):
            y = self.inner(x); self.latent_vector = self.inner.latent_vector; return y
    fid2 = metrics.compute_fid(model=Wrap(m), images1=im1, images2=im2, batch_size=320)
    assert abs(fid2 - fid) < 1e-6 * max(1.0, abs(fid))


class Ident(torch.nn.Module):
    def forward(self, x):
        return x


@pytest.mark.parametrize("n", [37, 500, 4096])
def test_is_conf_div_match_reference_golden(metrics_golden, n):
    import metrics
    gold = metrics_golden[f"is_n{n}"]
    logits = torch.from_numpy(detgen.normal(f"is/logits{n}", (n, 10)) * 3.0).cuda()
    is_ = metrics.compute_inception_score(Ident(), logits, batch_size=1000)
    conf, div = metrics.compute_confidence_diversity(Ident(), logits, batch_size=1000)
    assert abs(float(is_) - gold["inception_score"]) < 1e-4 * gold["inception_score"]
    assert abs(float(conf) - gold["confidence"]) < 1e-3 and abs(float(div) - gold["diversity"]) < 1e-3


def test_inception_score_known_answers():
    """The reference's own KATs (metrics.py:246-247), fed as logits whose softmax is the KAT matrix."""
    import metrics
    one_hot = torch.full((3, 3), -1e4).cuda()
    one_hot.fill_diagonal_(0.0)
    assert abs(float(metrics.compute_inception_score(Ident(), one_hot)) - 3.0) < 1e-4
    unif = torch.zeros(3, 3).cuda()
    assert abs(float(metrics.compute_inception_score(Ident(), unif)) - 1.0) < 1e-4

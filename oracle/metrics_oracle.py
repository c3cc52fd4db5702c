"""CPU restatement of the reference metrics (TEST INFRASTRUCTURE; numpy / scipy / torch-CPU).

Pinned against the imported /root/reference/metrics.py by oracle/gen_golden.py and against
tests/golden/metrics.json (incl. the reference's own IS known answers, metrics.py:246-247).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
import numpy as np
import torch
from scipy.linalg import sqrtm


def fid_from_features(f1, f2, eps=1e-6):
    """/root/reference/metrics.py:75-108, starting at the feature matrices (N, D) fp32."""
    mu1, mu2 = f1.mean(0), f2.mean(0)
    s1 = np.atleast_2d(np.cov(f1, rowvar=False))
    s2 = np.atleast_2d(np.cov(f2, rowvar=False))
    return fid_from_stats(np.atleast_1d(mu1), s1, np.atleast_1d(mu2), s2, eps)


def fid_from_stats(mu1, s1, mu2, s2, eps=1e-6):
    """/root/reference/metrics.py:90-108."""
    diff = mu1 - mu2
    covmean, _ = sqrtm(s1.dot(s2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(s1.shape[0]) * eps
        covmean = sqrtm((s1 + off).dot(s2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean)


def moments_chunked(f, chunk=512):
    """The build's sharding-invariant accumulation: fp32 features, every product
    and sum in fp64 (an fp32 x fp32 product is exact in fp64), accumulated chunk by chunk.
    Returns (n, s[D] f64, G[D,D] f64).  fp64 keeps Sigma = (G - n mu mu^T)/(n-1) PSD to ~1e-16,
    which the Newton-Schulz square root below needs (an fp32 Gram leaves ~1e-7 negative
    eigenvalues on rank-deficient inputs and the iteration diverges on them)."""
    n, d = f.shape
    s = np.zeros(d, np.float64)
    g = np.zeros((d, d), np.float64)
    for i in range(0, n, chunk):
        c = np.ascontiguousarray(f[i:i + chunk], dtype=np.float32).astype(np.float64)
        g += c.T @ c
        s += c.sum(0)
    return n, s, g


def stats_from_moments(n, s, g):
    """mu = s/n (returned fp64; the reference's mu is an fp32 mean), Sigma = (G - n mu mu^T)/(n-1)."""
    mu = s / n
    sigma = (g - n * np.outer(mu, mu)) / (n - 1)
    return mu, sigma


def ns_sqrt_sym(s, iters=50, delta=1e-12):
    """sqrt(S + delta*||S||_F*I) for symmetric PSD S by the coupled Newton-Schulz iteration in
    fp64 -- the GEMM-only algorithm the GPU path runs (no Schur form).  The relative ridge
    `delta` keeps rank-deficient covariances (N < D) from diverging on rounding noise."""
    d = s.shape[0]
    nrm = np.linalg.norm(s)  # Frobenius
    if nrm == 0:
        return np.zeros_like(s)
    c = nrm * (1.0 + delta)
    y = s / c + (delta / (1.0 + delta)) * np.eye(d)
    z = np.eye(d)
    for _ in range(iters):
        t = 1.5 * np.eye(d) - 0.5 * (z @ y)
        y = y @ t
        z = t @ z
    return y * np.sqrt(c)


def sqrt_trace_sym_ns(s1, s2, iters=50, delta=1e-12):
    """tr sqrt(S1 S2) = tr sqrt(S1^1/2 S2 S1^1/2): two symmetric Newton-Schulz square roots."""
    r1 = ns_sqrt_sym(s1, iters, delta)
    m = r1 @ s2 @ r1
    m = 0.5 * (m + m.T)
    return float(np.trace(ns_sqrt_sym(m, iters, delta)))


def fid_from_moments(m1, m2, iters=50, delta=1e-12):
    mu1, s1 = stats_from_moments(*m1)
    mu2, s2 = stats_from_moments(*m2)
    d = mu1 - mu2
    return float(d.dot(d) + np.trace(s1) + np.trace(s2) - 2.0 * sqrt_trace_sym_ns(s1, s2, iters, delta))


def inception_score(logits, eps=1e-7):
    """/root/reference/metrics.py:231,251-254 from logits (N, C) torch fp32."""
    p = torch.softmax(logits, dim=1)
    h_c = -(p * torch.log(p + eps)).sum(1).mean()
    pm = p.mean(0)
    h_d = -(pm * torch.log(pm + eps)).sum()
    return torch.exp(h_d - h_c)


def inception_score_from_probs(p, eps=1e-7):
    """metrics.py:251-254 on given probabilities (for the KATs at metrics.py:246-247)."""
    h_c = -(p * torch.log(p + eps)).sum(1).mean()
    pm = p.mean(0)
    h_d = -(pm * torch.log(pm + eps)).sum()
    return torch.exp(h_d - h_c)


def confidence_diversity(logits, N=3, alpha=0.7):
    """/root/reference/metrics.py:144-201 from logits."""
    p = torch.softmax(logits, dim=1)
    nc = p.shape[1]
    top1 = (p.max(1)[0] / p.sum(1)).mean()
    if N > 1:
        sp = torch.sort(p, dim=1, descending=True)[0]
        topn = sp[:, :N].sum(1).mean()
        conf = alpha * top1 + (1 - alpha) * topn
    else:
        conf = top1
    ids = torch.argmax(p, dim=1)
    hist = torch.bincount(ids, minlength=nc).to(torch.float)
    worst = torch.tensor([0] * (nc - 1) + [p.shape[0]], dtype=torch.float)
    div = 1 - hist.std() / worst.std()
    return 100 * conf, 100 * div, hist

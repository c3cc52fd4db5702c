"""FID / confidence-diversity / inception score with the reference's function surface
(/root/reference/metrics.py:37-108, :110-201, :204-267) on libgmk kernels.

What changes underneath (results are held to the reference: 0.05 absolute on FID):
  * features are produced in batches (the reference does one un-batched forward) and never leave
    the GPU; for our Autoencoder only the encoder half runs (the decoder output is discarded by the
    reference as well, metrics.py:70-73);
  * mean / covariance come from fp64 moments (n, sum x, sum x x^T) accumulated by an MFMA kernel;
    with sharded=True under torch.distributed each rank passes ITS shard of the images and the moments
    are combined with ONE all-reduce; the default (sharded=False) is the reference's call
    shape -- every caller holds the whole set -- and runs no collective;
  * tr sqrt(S1 S2) comes from two symmetric Newton-Schulz square roots on the device instead of
    scipy's Schur sqrtm; there is no complex arithmetic, so the reference's imaginary-component
    ValueError (metrics.py:100-104) cannot occur; its non-finite -> "+eps on the diagonal" retry
    (metrics.py:94-97) is kept.
"""
import numpy as np
import torch
import torch.distributed as dist

from gmk import ops
from gmk.parallel import all_reduce_


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _features(model, images, batch_size):
    """Yields fp32 feature batches (B, D) = model.latent_vector for images[i:i+batch_size]."""
    fast = getattr(model, "encode_features", None)
    for i in range(0, images.shape[0], batch_size):
        x = images[i:i + batch_size]
        if fast is not None:
            yield fast(x)
        else:
            _ = model(x)  # the reference's contract: any module that sets .latent_vector in forward
            yield model.latent_vector.detach().float().contiguous()


def feature_moments(model, images, batch_size=1024, group=None, sharded=False):
    """(n, s (D) f64, G (D,D) f64) of model.latent_vector over `images`.  sharded=True: `images` is THIS rank's
    shard of the sample set and the moments are combined with one all-reduce over `group`; sharded=False (the
    reference's call shape, every rank holding the same images): no collective."""
    model.eval()
    s = G = None
    n = 0
    with torch.no_grad():
        for f in _features(model, images, batch_size):
            if G is None:
                D = f.shape[1]
                packed = torch.zeros(1 + D + D * D, dtype=torch.float64, device=f.device)
                s, G = packed[1:1 + D], packed[1 + D:].view(D, D)
            ops.gram_accumulate(f, G, s)
            n += f.shape[0]
    packed[0] = n
    if sharded and _world(group) > 1:
        all_reduce_(packed, group=group)  # the ONE exchange step of the FID path
    return int(round(packed[0].item())), s, G


def fid_from_moments(m1, m2, eps=1e-6, iters=50, delta=1e-12):
    out = ops.fid_from_moments(m1, m2, iters=iters, delta=delta)
    fid = float(out[6].item())
    if not np.isfinite(fid):
        print(f'fid calculation produces singular product, adding {eps} to diagonal of cov estimates')
        out = ops.fid_from_moments(m1, m2, iters=iters, delta=delta, diag_eps=eps)
        fid = float(out[6].item())
    return np.float64(fid)


def compute_fid(model=None, images1=None, images2=None, eps=1e-6, batch_size=1024, group=None, sharded=False):
    """Frechet distance between the latent features of two image sets (metrics.py:37-108).
    sharded=True: every rank passes its own shard of both sets (see feature_moments)."""
    images1 = images1.to(images2.device)
    model = model.to(images2.device)
    m1 = feature_moments(model, images1, batch_size, group, sharded)
    m2 = feature_moments(model, images2, batch_size, group, sharded)
    assert m1[1].shape == m2[1].shape, 'Training and test mean vectors have different lengths'
    return fid_from_moments(m1, m2, eps=eps)


def _logit_stats(classifier, images, N, eps, batch_size, group, sharded=False):
    device = images.device
    classifier = classifier.to(device)
    classifier.eval()
    out = None
    with torch.no_grad():
        for i in range(0, images.shape[0], batch_size):
            logits = classifier(images[i:i + batch_size]).detach().float().contiguous()
            out = ops.softmax_stats(logits, out=out, topn=N, eps=eps)
    if sharded and _world(group) > 1:
        all_reduce_(out, group=group)  # the one exchange of the IS / conf / div path
    C = (out.numel() - 4) // 2
    return out, C


def compute_confidence_diversity(classifier, images, N=3, alpha=0.7, debug=False, batch_size=4096, group=None,
                                 sharded=False):
    """metrics.py:110-201: returns (100*confidence, 100*diversity) as torch scalars on images.device."""
    st, C = _logit_stats(classifier, images, max(N, 1), 1e-7, batch_size, group, sharded)
    n = st[2 * C + 3]
    top1_ratio = (st[2 * C + 1] / n).float()
    if N > 1:
        topN_ratio = (st[2 * C + 2] / n).float()
        confidence_score = alpha * top1_ratio + (1 - alpha) * topN_ratio
    else:
        confidence_score = top1_ratio
    class_frequencies = st[C:2 * C].float()
    if debug:
        labels = ["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"]
        print(f'\nImages are classified as:')
        for l, f in zip(labels, class_frequencies.tolist()):
            print(l, int(f))
        print()
    min_diversity_batch = torch.zeros(C, dtype=torch.float, device=images.device)
    min_diversity_batch[-1] = n.float()
    diversity_score2 = 1 - class_frequencies.std() / min_diversity_batch.std()
    return 100 * confidence_score, 100 * diversity_score2


def compute_inception_score(classifier, images, eps=1e-7, batch_size=4096, group=None, sharded=False):
    """metrics.py:204-267: IS = exp(H(mean_n p) - mean_n H(p_n)), eps inside the logs."""
    st, C = _logit_stats(classifier, images, 1, eps, batch_size, group, sharded)
    n = st[2 * C + 3]
    confidence_entropy = (st[2 * C] / n).float()
    class_frequencies = (st[:C] / n).float()
    diversity_entropy = -(class_frequencies * torch.log(class_frequencies + eps)).sum()
    return torch.exp(diversity_entropy - confidence_entropy)

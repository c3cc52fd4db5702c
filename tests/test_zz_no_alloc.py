"""gmk_vae_loss_fwd_bwd_static: the allocation-free variant of gmk_vae_loss_fwd_bwd (block partials in a file-scope
__device__ array of libgmk instead of a caller workspace; no counterpart in the reference, see include/gmk.h).

Called through the C ABI (ctypes), checked against the existing entry point on the same inputs: same kernels, so
every output must be bit-identical.  Two calls in a row, the second on a smaller input (the array keeps the first
call's partials: the second result must not see them), then the size limit.  Collected last on purpose (file name)."""
import numpy as np
import pytest
import torch

from oracle import ae_oracle, detgen

pytestmark = pytest.mark.gpu


def _both(B, L, variational, flags, beta, tag):
    from gmk import lib, check, ops
    h = lib()
    pre = torch.from_numpy(detgen.normal(tag + "/pre", (B, 3, 32, 32)) * 2.0).cuda()
    img = torch.from_numpy(detgen.images(tag + "/img", B)).cuda()
    stats = torch.from_numpy(detgen.normal(tag + "/stats", (B, L, 2)) * 0.5).cuda() if variational else None
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    outs = []
    for static in (False, True):
        recon, dpre = torch.full_like(pre, -7.0), torch.full_like(pre, -7.0)
        dstats = None if stats is None else torch.full_like(stats, -7.0)
        losses = torch.full((3,), -7.0, device="cuda")
        if static:
            check(h.gmk_vae_loss_fwd_bwd_static(p(pre), p(img), p(stats), p(recon), p(dpre), p(dstats), p(losses),
                                                B, 3072, L if variational else 0, beta, flags, s))
        else:
            ws = torch.empty(h.gmk_vae_loss_workspace_bytes(), dtype=torch.uint8, device="cuda")
            check(h.gmk_vae_loss_fwd_bwd(p(pre), p(img), p(stats), p(recon), p(dpre), p(dstats), p(losses),
                                         B, 3072, L if variational else 0, beta, flags, p(ws), ws.numel(), s))
        torch.cuda.synchronize()
        outs.append((losses.cpu(), dpre.cpu(), None if dstats is None else dstats.cpu(), recon.cpu()))
    return pre.cpu(), img.cpu(), None if stats is None else stats.cpu(), outs


def test_static_scratch_entry_point_matches_workspace_entry_point():
    # first call: large enough to use all 1,024 partial slots; second call, right behind it: 3 images (9 blocks)
    for B, L, variational, flags, beta, tag in ((512, 256, True, 1, 0.05, "noalloc/a"), (3, 16, True, 1 | 2, 0.5, "noalloc/b"),
                                                (5, 0, False, 0, 0.0, "noalloc/c")):
        pre, img, stats, (ref, got) = _both(B, L, variational, flags, beta, tag)
        for a, b in zip(ref, got):
            assert (a is None and b is None) or torch.equal(a, b)
        # and both against the oracle's restatement of main.py:31-48 (fp64), as for the existing entry point
        recon = torch.sigmoid(pre.double()) if flags & 1 else pre.double()
        rec, kl, total = ae_oracle.train_losses(recon, img.double(), None if stats is None else stats.double(), beta,
                                                variational=variational, mse_prior=bool(flags & 2))
        assert abs(float(got[0][0]) - float(rec)) <= 1e-6 * float(rec)
        assert abs(float(got[0][2]) - float(total)) <= 1e-5 * float(total)
        if variational:
            assert abs(float(got[0][1]) - float(kl)) <= 1e-5 * float(kl)


def test_static_scratch_entry_point_rejects_sizes_beyond_its_maximum():
    from gmk import lib
    h = lib()
    x = torch.zeros(16, device="cuda")
    p = x.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    big = (1 << 28) // 3072 + 1     # B * chw just over GMK_LOSS_STATIC_MAX_ELEMS: rejected before anything is launched
    code = h.gmk_vae_loss_fwd_bwd_static(p, p, None, None, p, None, p, big, 3072, 0, 0.0, 1, s)
    assert code == -6 and b"maximum" in h.gmk_strerror(code)
    code = h.gmk_vae_loss_fwd_bwd_static(p, p, p, None, p, p, p, 4, 4, 1 << 27, 0.0, 1, s)   # B * L over the limit
    assert code == -6
    torch.cuda.synchronize()

"""CPU, world_size 2, gloo: the data-parallel plumbing (flat gradient buckets, overlapped all-reduce
bookkeeping, FID moment exchange) is correct by construction.  No kernels run here."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn as nn
        from models.autoencoders import Autoencoder
        from gmk.optim import FusedAdamW
        from gmk.parallel import GradReducer
        torch.manual_seed(0)
        m = Autoencoder(variational=True, latent_size=32, num_channels=8, kernel_size=3, act=nn.GELU(), sigmoid=True,
                        no_pool=True, no_upsample=True)
        opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        red = GradReducer(m, opt)
        # buckets tile the flat buffer exactly, in backward-completion order decoder -> enc.fc -> enc convs
        segs = sorted(red.buckets.values())
        assert segs[0][0] == 0 and segs[-1][1] == opt._numel
        assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        assert list(red.buckets) == ["decoder", "encoder.fc", "encoder.conv"]
        # parameters / grads are views of the flat buffers, state_dict keys are the reference's
        for p in m.parameters():
            assert p.grad is not None and p.grad.shape == p.shape
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(opt._numel, generator=g)
        opt.flat_grad.copy_(local)
        # every per-parameter .grad sees the flat data
        first = next(m.parameters())
        assert torch.equal(first.grad.reshape(-1), opt.flat_grad[:first.numel()])
        for name in ("decoder", "encoder.fc", "encoder.conv"):
            red.ready(name)
        red.wait()
        both = [torch.randn(opt._numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(both)
        assert torch.allclose(opt.flat_grad, expect, atol=1e-6)
        assert abs(opt.grad_scale - 1.0 / world) < 1e-12          # the mean is folded into AdamW
        # FID moments: one all-reduce of the packed (n, s, G)
        D = 16
        feats = torch.randn(40 + 10 * rank, D, generator=torch.Generator().manual_seed(7 + rank)).double()
        packed = torch.zeros(1 + D + D * D, dtype=torch.float64)
        packed[0] = feats.shape[0]
        packed[1:1 + D] = feats.sum(0)
        packed[1 + D:] = (feats.T @ feats).reshape(-1)
        dist.all_reduce(packed)
        allf = torch.cat([torch.randn(40 + 10 * r, D, generator=torch.Generator().manual_seed(7 + r)).double() for r in range(world)])
        assert int(packed[0]) == allf.shape[0]
        n = allf.shape[0]
        mu = packed[1:1 + D] / n
        sigma = (packed[1 + D:].view(D, D) - n * torch.outer(mu, mu)) / (n - 1)
        assert np.allclose(sigma.numpy(), np.cov(allf.numpy(), rowvar=False), atol=1e-10)
        # the kernel step must refuse to run on CPU (no fallback path)
        try:
            opt.step()
            q.put((rank, "step() ran on CPU"))
            return
        except RuntimeError:
            pass
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_library_exports_every_declared_symbol():
    """include/gmk.h <-> libgmk.so <-> ctypes table agree (no compute calls without a GPU)."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
    import gmk
    from gmk import _lib
    handle = gmk.lib()
    hdr = open(os.path.join(ROOT, "include", "gmk.h")).read()
    declared = set(re.findall(r"\b(gmk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in gmk.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert handle.gmk_strerror(-4).decode().startswith("gmk:")


def test_fused_adamw_is_a_torch_adamw_with_reference_state_layout():
    sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
    import torch.nn as nn
    from models.autoencoders import Autoencoder
    from gmk.optim import FusedAdamW
    m = Autoencoder(variational=False, latent_size=16, num_channels=8, kernel_size=3, act=nn.ReLU(), no_pool=True, no_upsample=True)
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    ref = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    assert isinstance(opt, torch.optim.AdamW)
    a, b = opt.state_dict(), ref.state_dict()
    assert a["param_groups"][0].keys() == b["param_groups"][0].keys()
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    assert "last_epoch" in sched.state_dict()

"""CPU, world_size 2, gloo: the data-parallel plumbing (flat gradient buckets, overlapped all-reduce
bookkeeping, FID / IS moment exchange through the product's metrics module) is correct by construction.  No kernels
run here; the same path with the kernels runs in tests/test_dp_equiv_gpu.py (two ranks on the GPU)."""
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
        # decoder + encoder.fc travel as ONE all-reduce issued at ready("encoder.fc"): ready("decoder") alone moves nothing
        red.ready("decoder")
        assert torch.equal(opt.flat_grad, local)
        red.ready("encoder.fc")
        a, b = red._fc_dec
        assert (a, b) == (red.buckets["encoder.fc"][0], red.buckets["decoder"][1])
        ca, cb = red.buckets["encoder.conv"]
        assert torch.equal(opt.flat_grad[ca:cb], local[ca:cb]) and not torch.equal(opt.flat_grad[a:b], local[a:b])
        red.ready("encoder.conv")
        red.wait()
        both = [torch.randn(opt._numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(both)
        assert torch.allclose(opt.flat_grad, expect, atol=1e-6)
        assert abs(opt.grad_scale - 1.0 / world) < 1e-12          # the mean is folded into AdamW
        opt.flat_grad.copy_(local)                                  # merged=False (bucket-wise early AdamW): three collectives, same sums
        for name in ("decoder", "encoder.fc", "encoder.conv"):
            red.ready(name, merged=False)
        red.wait()
        assert torch.allclose(opt.flat_grad, expect, atol=1e-6)
        # FID / IS exchange steps: the PRODUCT host path (metrics.feature_moments / _logit_stats with sharded=True), with
        # the two HIP kernels replaced by CPU stand-ins of their documented contract (include/gmk.h) -- no GPU here
        import metrics
        from gmk import ops

        def cpu_gram(x, G, s):
            xd = x.double()
            G += xd.T @ xd
            s += xd.sum(0)

        def cpu_softmax_stats(logits, out=None, topn=3, eps=1e-7):
            n, C = logits.shape
            if out is None:
                out = torch.zeros(2 * C + 4, dtype=torch.float64)
            p = torch.softmax(logits.double(), dim=1)
            out[:C] += p.sum(0)
            out[C:2 * C] += torch.bincount(p.argmax(1), minlength=C).double()
            out[2 * C] += -(p * torch.log(p + eps)).sum()
            out[2 * C + 1] += p.max(1)[0].sum()
            out[2 * C + 2] += p.sort(1, descending=True)[0][:, :topn].sum()
            out[2 * C + 3] += n
            return out

        ops.gram_accumulate, ops.softmax_stats = cpu_gram, cpu_softmax_stats

        class Feat(torch.nn.Module):
            def forward(self, x):
                self.latent_vector = x
                return x

        D = 16
        sizes = [40 + 10 * r for r in range(world)]
        feats = [torch.randn(n, D, generator=torch.Generator().manual_seed(7 + r)) for r, n in enumerate(sizes)]
        n, sv, G = metrics.feature_moments(Feat(), feats[rank], batch_size=16, group=None, sharded=True)
        allf = torch.cat(feats).double()
        assert n == allf.shape[0]
        mu = sv / n
        sigma = (G - n * torch.outer(mu, mu)) / (n - 1)
        assert np.allclose(sigma.numpy(), np.cov(allf.numpy(), rowvar=False), atol=1e-10)
        n_local, _, G_local = metrics.feature_moments(Feat(), feats[rank], batch_size=16)      # reference call shape:
        assert n_local == sizes[rank]                                                          # no collective
        from oracle import metrics_oracle
        logits = [torch.randn(30 + 5 * r, 10, generator=torch.Generator().manual_seed(70 + r)) * 3 for r in range(world)]
        ident = torch.nn.Identity()
        is_ = metrics.compute_inception_score(ident, logits[rank], sharded=True)
        conf, div = metrics.compute_confidence_diversity(ident, logits[rank], sharded=True)
        full = torch.cat(logits)
        c_ref, d_ref, _ = metrics_oracle.confidence_diversity(full)
        assert abs(float(is_) - float(metrics_oracle.inception_score(full))) < 1e-5
        assert abs(float(conf) - float(c_ref)) < 1e-4 and abs(float(div) - float(d_ref)) < 1e-4
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


def test_launch_plan_api_state_machine_without_a_gpu():
    """gmk_plan_* bookkeeping (host-only code): order errors are GMK_E_STATE, an empty plan replays as a no-op."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
    import gmk
    h = gmk.lib()
    E_STATE, E_SHAPE = -7, -1
    assert h.gmk_plan_mark() == E_STATE and h.gmk_plan_end() == E_STATE
    p = C.c_void_p()
    assert h.gmk_plan_begin(C.byref(p)) == 0 and p.value
    q = C.c_void_p()
    assert h.gmk_plan_begin(C.byref(q)) == E_STATE
    assert h.gmk_train_step_enqueue(p, 0, 0, 0, None, None, None) == E_STATE      # still recording
    assert h.gmk_plan_mark() == 0 and h.gmk_plan_mark() == 1
    assert h.gmk_plan_end() == 0
    assert h.gmk_plan_segments(p) == 2 and h.gmk_plan_calls(p) == 0
    assert h.gmk_train_step_enqueue(p, 0, 2, 0, None, None, None) == 0
    assert h.gmk_train_step_enqueue(p, 1, 3, 0, None, None, None) == E_SHAPE
    h.gmk_plan_destroy(p)
    assert h.gmk_strerror(E_STATE).decode().startswith("gmk:")


def test_side_priority_is_refused_in_multi_process_jobs(monkeypatch):
    """High-priority weight-gradient streams hung two-rank runs in round 2: the knob is refused
    when WORLD_SIZE > 1 instead of being a documented trap.  The check runs before any stream is created."""
    sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
    from gmk import ops
    monkeypatch.setenv("GMK_SIDE_PRIORITY", "-1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="GMK_SIDE_PRIORITY"):
        ops.SideQueue(torch.device("cpu"))


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


def test_dp_rank_raises_the_hip_hardware_queue_count(monkeypatch):
    """gmk.parallel.ensure_hw_queues: a data-parallel rank wants 8 HIP hardware queues (with the default 4, RCCL's queues
    put a weight-gradient stream on the main stream's queue: +17 % per step, profiles/r03_dp_rehearsal.txt); a larger
    setting of the user's is left alone."""
    from gmk import parallel
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert parallel.ensure_hw_queues() == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert parallel.ensure_hw_queues() == 16 and os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    assert parallel.ensure_hw_queues() == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8"

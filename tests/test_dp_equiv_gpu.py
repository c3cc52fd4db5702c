"""Data-parallel equivalence on the GPU (BASELINE.json configs[2], configs[3]).

MSE and KL are batch means (/root/reference/main.py:31,45), so the gradient of the global batch is the mean of the
equal-shard gradients: the product path all-reduces (SUM) the flat gradient buffer and folds 1/world into the AdamW
kernel.  These tests check exactly that arithmetic on one GPU (8 shards of 32 against the pinned oracle at B=256,
every tensor in full), then run the real multi-process path -- GradReducer buckets launched from train_step on the
side stream, FusedAdamW.grad_scale, metrics' sharded moment exchange -- with two ranks: over gloo on one GPU
(always), and over RCCL on two GPUs when the box has them.
"""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import case_inputs, oracle_cfg
from oracle import ae_oracle, detgen, metrics_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30))


def test_b256_full_tensor_parity_and_8_shard_dp_equivalence(train_golden):
    """vae_cfg2_b256 (the benchmark configuration): (1) one pass at B=256, EVERY gradient tensor in full against the
    oracle (rel-L2 <= 1e-4; the golden carries norms only); (2) the same batch as 8 shards of 32: shard gradients
    summed in the flat buffer, 1/8 applied by the AdamW kernel, against the oracle's global-batch AdamW step."""
    from gmk.optim import FusedAdamW
    from test_model_gpu import build
    case = train_golden["vae_cfg2_b256"]
    params, x, eps = case_inputs(case)
    o = ae_oracle.train_step_grads(params, x, eps, case["beta"], oracle_cfg(case))
    assert abs(float(o["total"]) - case["total"]) <= 2e-6 * case["total"]          # the oracle is the pinned one
    m = build(case, params)
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    xt, et = torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda()
    losses = m.train_step(xt, beta=case["beta"], eps=et).cpu().double()
    assert abs(float(losses[2]) - float(o["total"])) <= 1e-4 * float(o["total"])
    assert abs(float(losses[1]) - float(o["kl"])) <= 1e-4 * float(o["kl"])
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), o["grads"][k].numpy()) <= 1e-4, (k, _rel(p.grad.cpu().numpy(), o["grads"][k].numpy()))
    # --- 8 shards of 32
    world = 8
    acc = torch.zeros_like(opt.flat_grad)
    shard_losses = torch.zeros(3, dtype=torch.float64)
    for r in range(world):
        sl = slice(32 * r, 32 * (r + 1))
        shard_losses += m.train_step(xt[sl].contiguous(), beta=case["beta"], eps=et[sl].contiguous()).cpu().double()
        acc += opt.flat_grad
    assert torch.allclose(shard_losses / world, losses, rtol=1e-5)                 # batch means: mean of shard losses
    opt.flat_grad.copy_(acc)                                                       # what the SUM all-reduce leaves behind
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy() / world, o["grads"][k].numpy()) <= 1e-4, k
    opt.grad_scale = 1.0 / world
    opt.step()
    for k, p in m.named_parameters():
        p0 = torch.from_numpy(params[k])
        ref, _, _ = ae_oracle.adamw_step(p0, o["grads"][k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.01)
        assert np.allclose(p.detach().cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6), k


def test_global_batch_2048_as_8_shards_of_256_matches_the_single_process_oracle(train_golden):
    """BASELINE.json configs[2] at its stated size: the global batch of 2048 as 8 per-GPU shards of 256
    (inputs / eps = slices of ONE B = 2048 batch), flat gradients summed as the SUM all-reduce leaves them, 1/8 applied by
    the AdamW kernel, against the oracle's single-process B = 2048 step (/root/reference/main.py:31,45: batch means):
    per-tensor rel-L2 <= 1e-4 on the gradients, then one AdamW step.  The weight-gradient reduction length here is 8x
    that of any other parity case (2,097,152 pixels per weight for the 32x32 layers, split 8 ways + fixed-order slabs)."""
    from gmk.optim import FusedAdamW
    from test_model_gpu import build
    case = dict(train_golden["vae_cfg2_b256"])
    world, Bs = 8, 256
    case["B"] = world * Bs
    case["name"] = "vae_cfg2_dp2048"
    params, x, eps = case_inputs(case)
    beta = case["beta"]
    o = ae_oracle.train_step_grads(params, x, eps, beta, oracle_cfg(case))
    m = build(case, params)
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    xt, et = torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda()
    acc = torch.zeros_like(opt.flat_grad)
    shard_losses = torch.zeros(3, dtype=torch.float64)
    for r in range(world):
        sl = slice(Bs * r, Bs * (r + 1))
        shard_losses += m.train_step(xt[sl], beta=beta, eps=et[sl]).cpu().double()   # replays the launch plan from shard 1 on
        acc += opt.flat_grad
    mean_losses = shard_losses / world
    assert abs(float(mean_losses[2]) - float(o["total"])) <= 1e-4 * float(o["total"])
    assert abs(float(mean_losses[1]) - float(o["kl"])) <= 1e-4 * float(o["kl"])
    opt.flat_grad.copy_(acc)
    worst = 0.0
    for k, p in m.named_parameters():
        err = _rel(p.grad.cpu().numpy() / world, o["grads"][k].numpy())
        worst = max(worst, err)
        assert err <= 1e-4, (k, err)
    opt.grad_scale = 1.0 / world
    opt.step()
    for k, p in m.named_parameters():
        p0 = torch.from_numpy(params[k])
        ref, _, _ = ae_oracle.adamw_step(p0, o["grads"][k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.01)
        assert np.allclose(p.detach().cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6), k


def test_fid_and_is_statistics_are_shard_invariant():
    """configs[3]: partial (n, s, G) and softmax statistics of 8 uneven shards summed == one pass (<= 1e-12 relative)."""
    from gmk import ops
    from test_metrics_gpu import _fid_features
    f = torch.from_numpy(_fid_features("fid/set1", 5000, 256, 0.5, 1.0)).cuda()
    cuts = [0, 11, 700, 1212, 2500, 2501, 3333, 4096, 5000]
    G1 = torch.zeros(256, 256, dtype=torch.float64, device="cuda"); s1 = torch.zeros(256, dtype=torch.float64, device="cuda")
    ops.gram_accumulate(f, G1, s1)
    Gs = torch.zeros_like(G1); ss = torch.zeros_like(s1)
    for a, b in zip(cuts, cuts[1:]):
        G = torch.zeros_like(G1); s = torch.zeros_like(s1)
        ops.gram_accumulate(f[a:b].contiguous(), G, s)
        Gs += G; ss += s
    assert float((Gs - G1).abs().max() / G1.abs().max()) <= 1e-12 and float((ss - s1).abs().max() / s1.abs().max()) <= 1e-12
    n, s_ref, G_ref = metrics_oracle.moments_chunked(f.cpu().numpy())
    assert np.allclose(Gs.cpu().numpy(), G_ref, rtol=1e-12, atol=1e-9)
    logits = torch.from_numpy(detgen.normal("is/logits4096", (4096, 10)) * 3.0).cuda()
    one = ops.softmax_stats(logits, topn=3)
    parts = torch.zeros_like(one)
    for a, b in zip([0, 5, 1000, 1001, 3000], [5, 1000, 1001, 3000, 4096]):
        parts += ops.softmax_stats(logits[a:b].contiguous(), topn=3)
    assert float((parts - one).abs().max() / one.abs().max()) <= 1e-12


# ---------------------------------------------------------------------------------------------- two ranks
def _dp_worker(rank, world, port, backend, ngpu, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank % ngpu))
        for p in (os.path.join(ROOT, "generative-models_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        dev = torch.device("cuda", rank % ngpu)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend, rank=rank, world_size=world)
        import metrics
        from models.autoencoders import Autoencoder
        from utils import get_activation_function
        from gmk.optim import FusedAdamW
        from gmk.parallel import GradReducer
        from test_metrics_gpu import _fid_features, FeatModel
        B, L, C = 8, 64, 32                                   # per-rank batch; global batch = world * B
        shape = dict(variational=True, latent_size=L, num_channels=C, kernel_size=3, no_pool=True, no_upsample=True)
        params = detgen.ae_params("dp2", **shape)
        m = Autoencoder(variational=True, latent_size=L, num_channels=C, kernel_size=3, act=get_activation_function("gelu"),
                        sigmoid=True, interpolation="nearest", no_pool=True, no_upsample=True, device=dev)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        m = m.to(dev)
        opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        red = GradReducer(m, opt)
        assert red.world == world and abs(opt.grad_scale - 1.0 / world) < 1e-12
        x = detgen.images("dp2", world * B)
        eps = detgen.normal("dp2/eps", (world * B, L))
        sl = slice(rank * B, (rank + 1) * B)
        losses = m.train_step(torch.from_numpy(x[sl]).to(dev), beta=0.01, eps=torch.from_numpy(eps[sl]).to(dev), reducer=red)
        red.wait()
        gsum = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        # the single-process global-batch step of the reference (main.py:27-53 at B = world * B), restated by the oracle
        cfg = dict(variational=True, latent_size=L, num_channels=C, act="gelu", sigmoid=True, no_pool=True, no_upsample=True,
                   interpolation="nearest", kernel_size=3, base=8)
        o = ae_oracle.train_step_grads(params, x, eps, 0.01, cfg)
        for k, p in m.named_parameters():
            ref = o["grads"][k].numpy()
            err = _rel(gsum[k] / world, ref)
            assert err <= 1e-4, ("grad", k, err)
            p0 = torch.from_numpy(params[k])
            pr, _, _ = ae_oracle.adamw_step(p0, o["grads"][k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.01)
            assert np.allclose(p.detach().cpu().numpy(), pr.numpy(), rtol=1e-5, atol=2e-6), ("param", k)
        # replicas stay identical: every rank holds the same parameters after the step
        flat = opt.flat_param.clone()
        ref_flat = flat.clone()
        from gmk.parallel import all_reduce_, broadcast_   # gloo: host copies (its device-tensor path can hang, see parallel.py)
        broadcast_(ref_flat, src=0)
        assert torch.equal(flat, ref_flat)
        # mean of the per-rank losses == global-batch loss
        tot = losses.double().clone()
        all_reduce_(tot)
        assert abs(float(tot[2]) / world - float(o["total"])) <= 1e-4 * float(o["total"])
        # FID path: each rank passes ITS shard, one all-reduce of (n, s, G) (metrics.feature_moments, sharded=True)
        f1 = _fid_features("fid/set1", 3000, 256, 0.5, 1.0)
        f2 = _fid_features("fid/set2", 3000, 256, 0.7, 0.8)
        cut = [0, 1300, 3000] if world == 2 else np.linspace(0, 3000, world + 1).astype(int).tolist()
        a, b = cut[rank], cut[rank + 1]
        fid = metrics.compute_fid(model=FeatModel(), images1=torch.from_numpy(f1[a:b]).to(dev),
                                  images2=torch.from_numpy(f2[a:b]).to(dev), batch_size=512, sharded=True)
        ref = metrics_oracle.fid_from_features(f1, f2)
        assert abs(fid - ref) < 0.05, (fid, ref)
        n, s, G = metrics.feature_moments(FeatModel(), torch.from_numpy(f1[a:b]).to(dev), 512, None, True)
        n_ref, s_ref, G_ref = metrics_oracle.moments_chunked(f1)
        assert n == n_ref and np.allclose(G.cpu().numpy(), G_ref, rtol=1e-12, atol=1e-9)
        # IS / confidence / diversity: one all-reduce of the statistics vector
        logits = detgen.normal("is/logits4096", (4096, 10)) * 3.0
        h = 4096 // world
        mine = torch.from_numpy(logits[rank * h:(rank + 1) * h]).to(dev)

        class Ident(torch.nn.Module):
            def forward(self, t):
                return t
        is_ = metrics.compute_inception_score(Ident(), mine, sharded=True)
        conf, div = metrics.compute_confidence_diversity(Ident(), mine, sharded=True)
        lt = torch.from_numpy(logits)
        assert abs(float(is_) - float(metrics_oracle.inception_score(lt))) < 1e-4
        c_ref, d_ref, _ = metrics_oracle.confidence_diversity(lt)
        assert abs(float(conf) - float(c_ref)) < 1e-3 and abs(float(div) - float(d_ref)) < 1e-3
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


def _run_two_ranks(backend, ngpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + os.getpid() % 4000 + (7 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, backend, ngpu, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert all(r[1] == "ok" for r in res), res


def test_two_rank_dp_step_and_metrics_over_gloo_on_one_gpu():
    """Two processes on cuda:0, gloo backend (it all-reduces CUDA tensors through the host): the product DP path --
    bucketed all-reduce issued from train_step behind the side stream, 1/world in the AdamW kernel -- equals the
    single-process global-batch step; the sharded FID / IS exchanges equal the oracle on the full set."""
    _run_two_ranks("gloo", 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_rank_dp_step_and_metrics_over_rccl():
    _run_two_ranks("nccl", 2)


def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py's multi-rank flow end to end (rendezvous from the torchrun environment, replica broadcast, GradReducer
    buckets from train_step, barrier + max-over-ranks timing, sharded FID leg, rank-0 JSON line) with two ranks on ONE card:
    GMK_DIST_BACKEND=gloo + GMK_FORCE_DEVICE=0.  Only the transport differs from the driver's N-GPU run (gloo through
    host copies instead of RCCL)."""
    import json
    import subprocess
    env = dict(os.environ, GMK_DIST_BACKEND="gloo", GMK_FORCE_DEVICE="0", PYTHONUNBUFFERED="1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and np.isfinite(out["final_losses"]["total"])
    assert out["fid_stats"]["samples"] == 100000 and np.isfinite(out["fid_stats"]["fid"])


# ---------------------------------------------------------------------------------------------- one rank, real RCCL
def _rccl_one_rank_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        for p in (os.path.join(ROOT, "generative-models_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # as gmk.parallel.init_from_env does
        from models.autoencoders import Autoencoder
        from utils import get_activation_function
        from gmk.optim import FusedAdamW
        from gmk.parallel import GradReducer, all_reduce_, broadcast_
        B, L, C = 256, 256, 64                                 # configs[1], the shape bench.py runs per GPU

        def make():
            torch.manual_seed(7)
            m = Autoencoder(variational=True, latent_size=L, num_channels=C, kernel_size=3, act=get_activation_function("gelu"),
                            sigmoid=True, interpolation="nearest", no_pool=True, no_upsample=True, device=dev).to(dev)
            return m, FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        m0, o0 = make()                                        # plain single-process step
        m1, o1 = make()                                        # same step with every gradient bucket through RCCL
        assert torch.equal(o0.flat_param, o1.flat_param)
        red = GradReducer(m1, o1, force=True)
        assert red.world == 1 and red.active and red._async and o1.grad_scale == 1.0
        broadcast_(o1.flat_param, src=0)                       # bench.py's replica broadcast
        g = torch.Generator(device=dev).manual_seed(11)
        xs = [torch.rand(B, 3, 32, 32, device=dev, generator=g) for _ in range(3)]
        for i in range(6):                                     # step 0 records the launch plan, 1.. replay it in segments
            eps = torch.randn(B, L, device=dev, generator=g)
            l0 = m0.train_step(xs[i % 3], beta=5e-6, eps=eps)
            o0.step()
            l1 = m1.train_step(xs[i % 3], beta=5e-6, eps=eps, reducer=red)
            red.wait()
            o1.step()
            assert torch.equal(l0, l1), ("loss", i)
            assert torch.equal(o0.flat_grad, o1.flat_grad), ("grad", i)
            assert torch.equal(o0.flat_param, o1.flat_param), ("param", i)
        assert len(m1._step_plan[1].marks) == 3                # decoder, encoder.fc, encoder.conv: one segment each
        # the FID exchange's payload (1 + D + D*D doubles, SUM) and bench.py's max-over-ranks timing reduce
        t = torch.arange(1 + 256 + 256 * 256, device=dev, dtype=torch.float64)
        ref = t.clone()
        all_reduce_(t)
        assert torch.equal(t, ref)
        d = torch.tensor([1.25], device=dev, dtype=torch.float64)
        all_reduce_(d, op=dist.ReduceOp.MAX)
        assert float(d) == 1.25
        dist.barrier()
        torch.cuda.synchronize()
        q.put((0, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((0, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


def test_one_rank_rccl_bucketed_allreduce_under_the_replayed_step_is_bit_identical():
    """The REAL transport on the one GPU this box has: a one-rank RCCL group (a sum over one rank is the identity) with
    GradReducer(force=True), so every gradient bucket of the configs[1] step goes through an asynchronous RCCL all-reduce on
    RCCL's own stream, issued between the segments of the replayed launch plan behind the weight-gradient streams, and
    AdamW waits for it.  What it shows: the RCCL path initialises, runs under the segment-wise replay without an error or a
    hang, and leaves losses, gradients and parameters bit-identical to the plain step over six steps (an all-reduce that
    overlapped the kernel writing its bucket would write stale values back).  What it cannot show: an all-reduce that ran
    entirely too early (the identity hides it; the ordering itself is held by the two-rank gloo test above, same code
    above the transport) and xGMI traffic between two cards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(26000 + os.getpid() % 3000, q))
    p.start()
    try:
        res = q.get(timeout=240)
    finally:
        p.join(30)
        if p.is_alive():
            p.kill()
    assert res[1] == "ok", res

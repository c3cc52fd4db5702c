"""FID-stats leg of bench.py (BASELINE.json configs[3]): images -> encoder features -> fp64 moments
(n, s, G) -> one all-reduce -> FID scalar, N1 = N2 = 50,000 samples per job, sharded by rank."""
import time

import torch
import torch.distributed as dist

FID_MFLOP_PER_SAMPLE = 52.95   # encoder forward 52.82 + Gram 0.131
N_PER_SET = 50000


def fid_stats_leg(device, rank, world, n_per_set=N_PER_SET, reps=3):
    import metrics
    from gmk import ops
    from gmk.parallel import all_reduce_, broadcast_
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    torch.manual_seed(0)
    ae = Autoencoder(variational=False, latent_size=256, num_channels=64, kernel_size=3,
                     act=get_activation_function("gelu"), sigmoid=True, interpolation="nearest",
                     no_pool=True, no_upsample=True, device=device).to(device)
    with torch.no_grad():   # spread the random-init features out (as the fid_ae_e2e golden does) so that the two
        ae.encoder.fc.weight.mul_(40.0)   # sets do not collapse onto one point and the reported FID is not ~0
    if world > 1:
        for p in ae.parameters():
            broadcast_(p.data, src=0)
    local = n_per_set // world + (1 if rank < n_per_set % world else 0)
    gen = torch.Generator(device=device).manual_seed(99 + rank)
    real = torch.rand(local, 3, 32, 32, device=device, generator=gen)
    fake = torch.rand(local, 3, 32, 32, device=device, generator=gen) ** 2   # a second distribution on [0,1)

    def run():
        return metrics.compute_fid(model=ae, images1=real, images2=fake, batch_size=2048, sharded=True)

    run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        fid = run()            # ends with a host read of the scalar, like the reference
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        all_reduce_(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # moments-only (features already resident) and the fixed sqrt-trace tail, rank-local
    feats = torch.randn(local, 256, device=device)
    G = torch.zeros(256, 256, dtype=torch.float64, device=device)
    s = torch.zeros(256, dtype=torch.float64, device=device)
    ops.gram_accumulate(feats, G, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.gram_accumulate(feats, G, s)
    torch.cuda.synchronize()
    t_mom = (time.perf_counter() - t0) / 5
    m = (local, s, G)
    ops.fid_from_moments(m, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.fid_from_moments(m, m)
    torch.cuda.synchronize()
    t_sqrt = time.perf_counter() - t0
    total = 2 * n_per_set
    return {"metric": "FID-stats samples/sec", "value": round(total / dt, 1), "unit": "samples/sec",
            "samples": total, "ms": round(dt * 1e3, 2), "fid": float(fid),
            "mfma_frac": round(total / dt / world * FID_MFLOP_PER_SAMPLE * 1e6 / 157.3e12, 4),
            "moments_only_samples_per_sec": round(local / t_mom, 1),
            "sqrt_trace_ms": round(t_sqrt * 1e3, 3),
            "workload": "2 x 50,000 synthetic 3x32x32 images, plain-AE encoder features (D=256), sharded by rank"}

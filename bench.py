#!/usr/bin/env python
"""bench.py -- train images/sec of the CIFAR-10 VAE train step (BASELINE.json configs[1]/[2]) on
N MI355X GPUs of one node, plus FID-stats samples/sec, the dominant kernel's roofline fraction and
a same-run CPU baseline.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + fused sigmoid/MSE/KL loss kernel + backward + bucketed gradient all-reduce
(RCCL, overlapped with backward) + fused AdamW + cosine LR step, i.e. main.py:27-53 of the
reference, on a device-resident synthetic batch of 256 images per GPU (weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TRAIN_MFLOP_PER_IMG = 319.68      # algorithmic FLOPs of one VAE-256 train step per image
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix peak
PER_GPU_BATCH = 256


def build(device):
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    from gmk.optim import FusedAdamW
    torch.manual_seed(0)
    model = Autoencoder(variational=True, latent_size=256, num_channels=64, kernel_size=3,
                        act=get_activation_function("gelu"), sigmoid=True, interpolation="nearest",
                        no_pool=True, no_upsample=True, device=device).to(device)
    optim = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=100 * 196)
    return model, optim, sched


def _time_launches(fn, iters):
    """HIP-event time per launch of fn() on the launch stream (torch's current stream = the stream the ops launch on)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def _conv_operands(device, H):
    N, C = PER_GPU_BATCH, 64
    x = torch.rand(N, C, H, H, device=device)
    w = (torch.rand(C, C, 3, 3, device=device) - 0.5) * 0.1
    b = torch.zeros(C, device=device)
    return x, w, b


def _mfma_record(kernel, label, t, flops, pmc_prefix, grid):
    ach = flops / t / 1e12
    return {"kernel": f"{kernel} ({label})", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "launch_us": round(t * 1e6, 2),
            "flops_per_launch": flops, "traffic": pmc_traffic(pmc_prefix, grid),
            "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC, profiles/{PMC_FILE})",
            "step_share": step_share(pmc_prefix),
            "step_share_source": f"share of in-step kernel time, profiles/{SHARE_FILE} (tools/step_share.py over the committed trace)"}


def dominant_kernel_roofline(device, iters=30):
    """The kernel with the largest share of in-step kernel time: conv_igemm_kernel<2, 3, 8, 1, 1, 1> (implicit GEMM by output
    parity class: both transposed convs of the decoder and both stride-2 data gradients of the encoder, 4 launches per step),
    timed alone on its heaviest shape, the 64->64 transposed conv 16x16 -> 32x32 at B = 256 with GELU and act' stores.
    Algorithmic FLOPs: 2 * Cin * Cout * 9 per INPUT pixel = 4.83 GFLOP per launch."""
    from gmk import ops
    x, w, b = _conv_operands(device, 16)
    t = _time_launches(lambda: ops.deconv2d_fwd(x, w, b, act="gelu", want_dact=True), iters)
    flops = 2.0 * PER_GPU_BATCH * 16 * 16 * 64 * 64 * 9
    return _mfma_record("conv_igemm_kernel<2, 3, 8, 1, 1, 1>", "transposed conv 64->64 3x3 s2, 16x16 -> 32x32, B256, gelu + act'",
                        t, flops, "void conv_igemm_kernel<2, 3, 8, 1, 1, 1>", 262144)


def igemm_s2_kernel_roofline(device, iters=30):
    """Second of the implicit-GEMM class: the stride-2 conv (and the transposed conv's data gradient), 32x32 -> 16x16."""
    from gmk import ops
    x, w, b = _conv_operands(device, 32)
    t = _time_launches(lambda: ops.conv2d_fwd(x, w, b, stride=2, act="gelu", want_dact=True), iters)
    flops = 2.0 * PER_GPU_BATCH * 16 * 16 * 64 * 64 * 9
    return _mfma_record("conv_igemm_kernel<1, 3, 8, 1, 2, 4>", "conv 64->64 3x3 s2, 32x32 -> 16x16, B256, gelu + act'",
                        t, flops, "void conv_igemm_kernel<1, 3, 8, 1, 2, 4>", 131072)


def wino_kernel_roofline(device, iters=30):
    """The stride-1 16x16 layers (Winograd F(2x2,3x3), conv_wino.hip).  `achieved` / `frac` are ALGORITHMIC FLOPs (direct
    convolution) over the launch time; the kernel executes 16/36 of those multiplies on the matrix cores:
    `mfma_flops_executed` / `mfma_frac_executed` state that separately (`frac` is an effective-throughput figure whose
    ceiling is 2.25, not 1)."""
    from gmk import ops
    x, w, b = _conv_operands(device, 16)
    t = _time_launches(lambda: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True), iters)
    flops = 2.0 * PER_GPU_BATCH * 16 * 16 * 64 * 64 * 9
    winograd = os.environ.get("GMK_WINO", "3") not in ("0", "2")
    if winograd:
        out = _mfma_record("conv_wino_kernel<fwd>", "Winograd F(2x2,3x3), 64->64 3x3 s1, 16x16, B256, gelu + act'", t, flops,
                           "void conv_wino_kernel<false>", 131072)
        out["mfma_flops_executed"] = flops * 16.0 / 36.0
        out["mfma_frac_executed"] = round(out["achieved"] * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS, 4)
    else:
        out = _mfma_record("conv_igemm_kernel<0, 3, 8, 1, 2, 1>", "implicit GEMM, 64->64 3x3 s1, 16x16, B256, gelu + act'", t, flops,
                           "void conv_igemm_kernel<0, 3, 8, 1, 2, 1>", 131072)
    return out


PMC_FILE = "r03_pmc_summary.json"
SHARE_FILE = "r03_step_share.json"
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def pmc_traffic(kernel_prefix, grid=None):
    """HBM bytes per launch of a kernel from the committed PMC passes (FETCH_SIZE doubled as the gfx950 guide prescribes,
    + WRITE_SIZE; tools/pmc_summary.py), None if the file or the row is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            for row in json.load(f):
                if row["kernel"].startswith(kernel_prefix) and "hbm_read_B" in row and (grid is None or row.get("grid") == grid):
                    return round(row["hbm_read_B"] + row["hbm_write_B"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def step_share(kernel_prefix):
    """Share of in-step kernel time of a kernel from the committed trace digest (tools/step_share.py), None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", SHARE_FILE)) as f:
            for row in json.load(f)["kernels"]:
                if row["kernel"].startswith(kernel_prefix):
                    return row["share"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def loss_kernel_roofline(device, B=65536, iters=10):
    """north_star: achieved HBM GB/s of the fused loss kernel.  At B = 256 the kernel moves 10.5 MB (latency bound), so the
    bandwidth figure is taken at B = 65,536 (2.68 GB): sigmoid + MSE + beta*KL forward and backward in one pass,
    algorithmic bytes 40,960 per image (read pre + image 12,288 each, write dpre 12,288, read stats 2,048, write dstats
    2,048), HIP events on the launch stream."""
    from gmk import ops
    pre = torch.randn(B, 3, 32, 32, device=device)
    img = torch.rand(B, 3, 32, 32, device=device)
    stats = torch.randn(B, 256, 2, device=device) * 0.1
    for _ in range(2):
        ops.vae_loss_fwd_bwd(pre, img, stats, 5e-6, fused=True)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        ops.vae_loss_fwd_bwd(pre, img, stats, 5e-6, fused=True)
    e1.record(s)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    nbytes = 40960.0 * B
    # per-launch time includes the torch.empty_like of the outputs on the host side only (asynchronous): device time
    gbs = nbytes / t / 1e9
    return {"kernel": f"vae_loss_kernel<sigmoid+MSE+KL fwd+bwd, B={B}>", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "launch_us": round(t * 1e6, 1),
            "bytes_per_launch": nbytes, "traffic": pmc_traffic("void vae_loss_kernel<true, false, true>", B * 3072 // 4 if B * 3072 // 4 < 262144 else 262144),
            "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC, profiles/{PMC_FILE})"}


def secondary_config_leg(device, which, steps=10, warmup=3):
    """Secondary workloads (not `value`): the train step of BASELINE.json configs[4] (VAE with pool + nearest upsample,
    latent 512, synthetic 3x64x64, B = 256: 267.4 MFLOP forward, 788.0 MFLOP per image and step) and of
    configs[0]'s shape at the reference's default batch (plain AE, B = 50, 313.39 MFLOP per image)."""
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    from gmk.optim import FusedAdamW
    if which == "cfg5":
        kw = dict(variational=True, latent_size=512, no_pool=False, no_upsample=False, image_size=64)
        B, hw, mflop, L = 256, 64, 788.0, 512
        label = "VAE latent_size=512 --variational, pool + nearest upsample, act gelu, synthetic 3x64x64, batch 256 (BASELINE.json configs[4])"
    else:
        kw = dict(variational=False, latent_size=256, no_pool=True, no_upsample=True, image_size=32)
        B, hw, mflop, L = 50, 32, 313.39, 256
        label = "plain AE latent_size=256 --sigmoid --no_upsample --no_pool, act gelu, 3x32x32, batch 50 = the reference default (BASELINE.json configs[0] shape)"
    torch.manual_seed(0)
    model = Autoencoder(num_channels=64, kernel_size=3, act=get_activation_function("gelu"), sigmoid=True,
                        interpolation="nearest", device=device, **kw).to(device)
    optim = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=100 * 196)
    g = torch.Generator(device=device).manual_seed(99)
    xs = [torch.rand(B, 3, hw, hw, device=device, generator=g) for _ in range(2)]

    def step(i):
        eps = torch.randn(B, L, device=device, generator=g) if kw["variational"] else None
        losses = model.train_step(xs[i % 2], beta=5e-6, eps=eps)
        optim.step()
        sched.step()
        return losses

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        losses = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": label, "ms_per_step": round(dt * 1e3, 4), "images_per_sec": round(B / dt, 1), "steps": steps,
           "mflop_per_image": mflop, "step_mfma_frac": round(B / dt * mflop * 1e6 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
           "final_total_loss": float(losses[2])}
    del model, optim, xs
    torch.cuda.empty_cache()
    return out


def cpu_baseline(steps=6):
    """The oracle (CPU restatement of the reference step incl. AdamW) on this box's host cores."""
    from oracle import ae_oracle
    import torch.nn.functional as F  # noqa: F401
    # threads actually used: the box's CPU share (OMP_NUM_THREADS is set to it there; os.cpu_count() shows the whole
    # machine, of which this job may only use its share) -- both are reported
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    from models.autoencoders import Autoencoder
    from utils import get_activation_function
    torch.manual_seed(0)
    m = Autoencoder(variational=True, latent_size=256, num_channels=64, kernel_size=3,
                    act=get_activation_function("gelu"), sigmoid=True, no_pool=True, no_upsample=True)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, weight_decay=0.01)
    cfg = dict(variational=True, latent_size=256, num_channels=64, act="gelu", sigmoid=True,
               no_pool=True, no_upsample=True)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(PER_GPU_BATCH, 3, 32, 32, generator=g)

    def step():
        eps = torch.randn(PER_GPU_BATCH, 256)
        recon, stats, z = ae_oracle.autoencoder_forward(params, x, eps, **cfg)
        rec, kl, total = ae_oracle.train_losses(recon, x, stats, 5e-6, variational=True)
        opt.zero_grad()
        total.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(PER_GPU_BATCH / dt, 1), "unit": "images/sec", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{steps} train steps of batch {PER_GPU_BATCH} (oracle/ae_oracle.py, torch CPU fp32, same VAE config)",
            "ms_per_step": round(dt * 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-fid", action="store_true", help="skip the FID-stats leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[4] (64x64, pool + upsample) and B=50 legs")
    args = ap.parse_args()

    from gmk.parallel import init_from_env, GradReducer, all_reduce_, broadcast_
    rank, local, world = init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    device = torch.device("cuda", local)
    import gmk
    gmk.lib()

    # kernel-path choice for this device (gmk/tune.py), on a throw-away replica, before anything is timed
    from gmk.tune import tune_thin_paths
    t_model, t_optim, t_sched = build(device)
    t_x = torch.rand(PER_GPU_BATCH, 3, 32, 32, device=device)
    t_eps = torch.randn(PER_GPU_BATCH, 256, device=device)

    def t_step():
        t_model.train_step(t_x, beta=5e-6, eps=t_eps)
        t_optim.step()
        t_sched.step()

    tuned = tune_thin_paths(t_step)
    del t_model, t_optim, t_sched, t_x, t_eps, t_step

    model, optim, sched = build(device)
    if world > 1:  # identical replicas: broadcast rank 0's parameters
        broadcast_(optim.flat_param, src=0)
    # GMK_DP_REHEARSE=1 (tools/README.md): one rank, the whole DP machinery (RCCL all-reduce per bucket, segment-wise replay)
    # under the timed step -- what data parallelism costs a rank before any xGMI traffic.  Not a bench configuration.
    reducer = GradReducer(model, optim, force=os.environ.get("GMK_DP_REHEARSE", "0") == "1")
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    batches = [torch.rand(PER_GPU_BATCH, 3, 32, 32, device=device, generator=gen) for _ in range(4)]
    egen = torch.Generator(device=device).manual_seed(4321 + rank)
    beta = 5e-6

    def step(i):
        eps = torch.randn(PER_GPU_BATCH, 256, device=device, generator=egen)
        losses = model.train_step(batches[i % len(batches)], beta=beta, eps=eps, reducer=reducer,
                                  optimizer=None if os.environ.get("GMK_EARLY_ADAMW", "0") == "0" else optim)
        reducer.wait()
        optim.step()
        sched.step()
        return losses

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        all_reduce_(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_losses = [float(v) for v in losses.cpu()]

    out = None
    if rank == 0:
        ips = world * PER_GPU_BATCH * args.steps / dt
        out = {
            "metric": "train images/sec (CIFAR-10 VAE)", "value": round(ips, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VAE latent_size=256 --variational --beta 5e-6 --no_upsample --no_pool, act gelu, "
                                   "3x32x32, batch 256 per GPU, AdamW+cosine (BASELINE.json configs[1]; configs[2] at N=8)",
                       "global_batch": world * PER_GPU_BATCH, "parallelism": f"dp{world}"},
            "final_losses": {"rec": final_losses[0], "kl": final_losses[1], "total": final_losses[2]},
            "kernel_paths": {"thin_in,thin_outv": list(tuned["choice"]), "tuned_on_this_device": tuned["tuned"],
                             "ms_per_step_by_choice": tuned["ms"]},
            "step_mfma_frac": round(ips / world * TRAIN_MFLOP_PER_IMG * 1e6 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
        }
        if reducer.active and world == 1:
            out["dp_rehearsal"] = "one rank, every gradient bucket through an RCCL all-reduce (GMK_DP_REHEARSE=1)"
        out["roofline"] = dominant_kernel_roofline(device)
        out["roofline_igemm_s2"] = igemm_s2_kernel_roofline(device)
        out["roofline_wino"] = wino_kernel_roofline(device)
        out["roofline_loss_kernel"] = loss_kernel_roofline(device)
        if world == 1 and not args.no_secondary:
            out["other_configs"] = {"cfg5_64x64_pool_upsample_latent512_b256": secondary_config_leg(device, "cfg5"),
                                    "cfg1_shape_plain_ae_b50": secondary_config_leg(device, "cfg1", steps=50, warmup=10)}
    if not args.no_fid:
        try:
            from bench_fid import fid_stats_leg
            fid = fid_stats_leg(device, rank, world)
            if rank == 0:
                out["fid_stats"] = fid
        except ImportError:
            pass
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

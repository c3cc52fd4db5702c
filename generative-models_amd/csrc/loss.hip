// loss.hip -- fused sigmoid + MSE + beta*KL loss and gradients; reparameterisation fwd/bwd.
// HBM-bound: 40,960 algorithmic bytes per image (VAE, 3x32x32, L=256): read pre + image,
// write dpre, read stats, write dstats.  One pass, float4 per lane, wave64 shuffle reduction,
// per-block fp64 partials, fixed-order 1-block finalize (deterministic).
#include "gmk_common.h"
#include "plan.h"
#include <stdlib.h>
#include <mutex>

#define LOSS_BLOCKS 1024
#define LOSS_THREADS 256

extern "C" size_t gmk_vae_loss_workspace_bytes(void) { return sizeof(double) * 4 * LOSS_BLOCKS; }

// Fixed-order sum of the block partials -> losses.  inv_bl > 0 selects the eval form: losses[3] = mean |mu|,
// losses[4] = mean sigma (over B*L elements).  One 256-thread block.  The partials are read with agent-scope loads
// (sc1: they bypass this CU's L1), which is what the in-kernel hand-off of the one-launch form needs and costs the
// separate finalize kernel nothing.
struct LossFin { float* losses; double inv_n, inv_b, inv_bl; float beta; int variational; unsigned* counter; };
__device__ __forceinline__ double gmk_ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vae_loss_finalize_block(const double* partial, int nblocks, const LossFin f, double (*red)[4]) {
  double a = 0, b = 0, c = 0, d = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += gmk_ld_agent(partial + i); b += gmk_ld_agent(partial + LOSS_BLOCKS + i);
    if (f.inv_bl > 0) { c += gmk_ld_agent(partial + 2 * LOSS_BLOCKS + i); d += gmk_ld_agent(partial + 3 * LOSS_BLOCKS + i); }
  }
  a = gmk_wave_sum(a); b = gmk_wave_sum(b); c = gmk_wave_sum(c); d = gmk_wave_sum(d);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; red[3][threadIdx.x >> 6] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    double sb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    float rec = (float)(sa * f.inv_n);
    float kl = f.variational ? (float)(sb * f.inv_b) : 0.f;
    f.losses[0] = rec;
    f.losses[1] = kl;
    f.losses[2] = f.variational ? rec + f.beta * kl : rec;
    if (f.inv_bl > 0) {
      f.losses[3] = f.variational ? (float)((red[2][0] + red[2][1] + red[2][2] + red[2][3]) * f.inv_bl) : 0.f;
      f.losses[4] = f.variational ? (float)((red[3][0] + red[3][1] + red[3][2] + red[3][3]) * f.inv_bl) : 0.f;
    }
  }
}
__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const double* __restrict__ partial, int nblocks, const LossFin f) {
  __shared__ double red[4][4];
  vae_loss_finalize_block(partial, nblocks, f, red);
}

// GRAD = false is the no-grad form of compute_test_loss (main.py:73-111): no gradient stores, and two more sums
// (|mu| and sigma = exp(log_var / 2), main.py:101-102) ride along with the KL pass.
template <bool SIGMOID, bool WRITE_RECON, bool GRAD>
__global__ __launch_bounds__(LOSS_THREADS) void vae_loss_kernel(
    const float4* __restrict__ pre, const float4* __restrict__ image, const float2* __restrict__ stats,
    float4* __restrict__ recon_out, float4* __restrict__ dpre, float2* __restrict__ dstats,
    double* __restrict__ partial, int64_t n4, int64_t nstat, float gscale /* 2/(B*chw) */,
    float kscale /* beta/B */, int mse_prior, const LossFin fin /* fin.counter != NULL: one-launch form */) {
  const int64_t gtid = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x;
  const int64_t gstride = (int64_t)gridDim.x * LOSS_THREADS;
  float se = 0.f;  // sum of squared errors (this thread: at most a few float4)
  for (int64_t i = gtid; i < n4; i += gstride) {
    float4 p = pre[i], x = image[i], r, g;
    if (SIGMOID) {
      r.x = gmk_sigmoidf(p.x); r.y = gmk_sigmoidf(p.y); r.z = gmk_sigmoidf(p.z); r.w = gmk_sigmoidf(p.w);
    } else {
      r = p;
    }
    float dx = r.x - x.x, dy = r.y - x.y, dz = r.z - x.z, dw = r.w - x.w;
    se += dx * dx + dy * dy + dz * dz + dw * dw;
    if (GRAD) {
      g.x = gscale * dx; g.y = gscale * dy; g.z = gscale * dz; g.w = gscale * dw;
      if (SIGMOID) {
        g.x *= r.x * (1.f - r.x); g.y *= r.y * (1.f - r.y); g.z *= r.z * (1.f - r.z); g.w *= r.w * (1.f - r.w);
      }
      dpre[i] = g;
    }
    if (WRITE_RECON) recon_out[i] = r;
  }
  float kl = 0.f, amu = 0.f, sig = 0.f;
  for (int64_t i = gtid; i < nstat; i += gstride) {
    float2 s = stats[i];  // (mu, log_var)
    float2 g;
    if (mse_prior) {  // main.py:39-42
      float sg = expf(0.5f * s.y);
      kl += s.x * s.x + (sg - 1.f) * (sg - 1.f);
      g.x = kscale * 2.f * s.x;
      g.y = kscale * (sg - 1.f) * sg;
      if (!GRAD) sig += sg;
    } else {  // main.py:45
      float e = expf(s.y);
      kl += -0.5f * (1.f + s.y - s.x * s.x - e);
      g.x = kscale * s.x;
      g.y = kscale * 0.5f * (e - 1.f);
      if (!GRAD) sig += expf(0.5f * s.y);
    }
    if (GRAD) { if (dstats) dstats[i] = g; }   // dstats == NULL: the KL gradient is applied elsewhere (GMK_LOSS_NO_DSTATS)
    else amu += fabsf(s.x);
  }
  __shared__ double red[4][LOSS_THREADS / 64];
  double a = gmk_wave_sum((double)se), b = gmk_wave_sum((double)kl);
  double c = 0, d = 0;
  if (!GRAD) { c = gmk_wave_sum((double)amu); d = gmk_wave_sum((double)sig); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; red[3][wave] = d; }
  __syncthreads();
  if (fin.counter == nullptr) {
    if (threadIdx.x == 0) {
      double sa = 0, sb = 0, sc = 0, sd = 0;
      for (int w = 0; w < LOSS_THREADS / 64; ++w) { sa += red[0][w]; sb += red[1][w]; sc += red[2][w]; sd += red[3][w]; }
      partial[blockIdx.x] = sa;
      partial[LOSS_BLOCKS + blockIdx.x] = sb;
      if (!GRAD) { partial[2 * LOSS_BLOCKS + blockIdx.x] = sc; partial[3 * LOSS_BLOCKS + blockIdx.x] = sd; }
    }
    return;
  }
  // One-launch form: the block whose arrival ticket is the last one sums the partials (fixed order, same code as the
  // finalize kernel).  Hand-off per MI355X_MICROARCH.md (inter-workgroup visibility): ONE lane per block stores the
  // block's partials with agent-scope (sc1, write-through) stores, drains them (vmcnt(0)), then takes a ticket with a
  // returning agent-scope atomic add; the last block runs an agent-scope acquire (invalidates its CU's L1), waits for
  // it, and only then, behind a workgroup barrier, do its waves load the partials (sc1 loads).  The counter is reset
  // for the next launch by that block.
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0, sc = 0, sd = 0;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) { sa += red[0][w]; sb += red[1][w]; sc += red[2][w]; sd += red[3][w]; }
    __hip_atomic_store(partial + blockIdx.x, sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(partial + LOSS_BLOCKS + blockIdx.x, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!GRAD) {
      __hip_atomic_store(partial + 2 * LOSS_BLOCKS + blockIdx.x, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(partial + 3 * LOSS_BLOCKS + blockIdx.x, sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  vae_loss_finalize_block(partial, (int)gridDim.x, fin, (double(*)[4])red);
  if (threadIdx.x == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// shared by gmk_vae_loss_fwd_bwd (grad = true) and gmk_vae_loss_eval (grad = false; `losses` then has 5 floats)
static int vae_loss_launch(const float* pre, const float* image, const float* stats, float* recon_out, float* dpre,
                           float* dstats, float* losses, int64_t B, int64_t chw, int64_t L, float beta, int flags,
                           void* ws, size_t ws_bytes, gmk_stream_t stream, bool grad, unsigned* counter = nullptr) {
  GMK_REQUIRE(pre && image && losses && ws, GMK_E_NULL);
  GMK_REQUIRE(!grad || dpre, GMK_E_NULL);
  GMK_REQUIRE(B > 0 && chw > 0 && (B * chw) % 4 == 0, GMK_E_SHAPE);
  GMK_REQUIRE(ws_bytes >= gmk_vae_loss_workspace_bytes(), GMK_E_WORKSPACE);
  GMK_REQUIRE(gmk_aligned16(pre) && gmk_aligned16(image) && gmk_aligned16(dpre) && gmk_aligned16(ws), GMK_E_ALIGN);
  const bool variational = stats != nullptr;
  if (variational) GMK_REQUIRE((dstats || !grad || (flags & GMK_LOSS_NO_DSTATS)) && L > 0, GMK_E_NULL);
  if (flags & GMK_LOSS_NO_DSTATS) dstats = nullptr;
  if (recon_out) GMK_REQUIRE(gmk_aligned16(recon_out), GMK_E_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = B * chw / 4, nstat = variational ? B * L : 0;
  // float4 items per thread: small launches (B = 256: 196,608 items) are a chain of round trips -- load, block sum,
  // partial store + ticket, the last block's read of every partial -- and fewer, fatter blocks shorten the last one
  static const int per_thread = getenv("GMK_LOSS_PER_THREAD") ? atoi(getenv("GMK_LOSS_PER_THREAD")) : 4;  // 1 -> 4: -4 us of step
  int64_t want = (n4 + (int64_t)LOSS_THREADS * per_thread - 1) / ((int64_t)LOSS_THREADS * per_thread);
  int nblocks = (int)(want < LOSS_BLOCKS ? (want < 1 ? 1 : want) : LOSS_BLOCKS);
  const float gscale = (float)(2.0 / ((double)B * (double)chw));
  const float kscale = (float)((double)beta / (double)B);
  double* partial = (double*)ws;
  const bool sig = flags & GMK_LOSS_SIGMOID;
  const int msep = (flags & GMK_LOSS_MSE_PRIOR) ? 1 : 0;
  LossFin fin{losses, 1.0 / ((double)B * (double)chw), 1.0 / (double)B,
              grad ? 0.0 : 1.0 / ((double)B * (double)(L > 0 ? L : 1)), beta, variational ? 1 : 0, counter};
#define LAUNCH(S, R, G)                                                                                      \
  hipLaunchKernelGGL((vae_loss_kernel<S, R, G>), dim3(nblocks), dim3(LOSS_THREADS), 0, s, (const float4*)pre, \
                     (const float4*)image, (const float2*)stats, (float4*)recon_out, (float4*)dpre,          \
                     (float2*)dstats, partial, n4, nstat, gscale, kscale, msep, fin)
  if (grad) {
    if (sig) { if (recon_out) LAUNCH(true, true, true); else LAUNCH(true, false, true); }
    else     { if (recon_out) LAUNCH(false, true, true); else LAUNCH(false, false, true); }
  } else {
    if (sig) { if (recon_out) LAUNCH(true, true, false); else LAUNCH(true, false, false); }
    else     { if (recon_out) LAUNCH(false, true, false); else LAUNCH(false, false, false); }
  }
#undef LAUNCH
  GMK_CHECK_LAUNCH();
  if (counter) return GMK_OK;   // one-launch form: the last block has finalized
  hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, fin);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

extern "C" int gmk_vae_loss_fwd_bwd(const float* pre, const float* image, const float* stats, float* recon_out,
                                    float* dpre, float* dstats, float* losses, int64_t B, int64_t chw, int64_t L,
                                    float beta, int flags, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_vae_loss_fwd_bwd, pre, image, stats, recon_out, dpre, dstats, losses, B, chw, L, beta, flags, ws, ws_bytes, stream);
  return vae_loss_launch(pre, image, stats, recon_out, dpre, dstats, losses, B, chw, L, beta, flags, ws, ws_bytes,
                         stream, true);
}

// One launch: `sync` = 256 bytes of device memory, zero before the first call and owned by this entry point (the kernel
// leaves it zero); the last block to finish sums the partials.  Calls sharing a `sync` must be stream-ordered.
extern "C" int gmk_vae_loss_fwd_bwd_fused(const float* pre, const float* image, const float* stats, float* recon_out,
                                          float* dpre, float* dstats, float* losses, int64_t B, int64_t chw, int64_t L,
                                          float beta, int flags, void* ws, size_t ws_bytes, void* sync,
                                          gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_vae_loss_fwd_bwd_fused, pre, image, stats, recon_out, dpre, dstats, losses, B, chw, L, beta, flags, ws, ws_bytes, sync, stream);
  GMK_REQUIRE(sync, GMK_E_NULL);
  GMK_REQUIRE((((uintptr_t)sync) & 255u) == 0, GMK_E_ALIGN);
  return vae_loss_launch(pre, image, stats, recon_out, dpre, dstats, losses, B, chw, L, beta, flags, ws, ws_bytes,
                         stream, true, (unsigned*)sync);
}

extern "C" int gmk_vae_loss_eval(const float* pre, const float* image, const float* stats, float* recon_out,
                                 float* out5, int64_t B, int64_t chw, int64_t L, float beta, int flags, void* ws,
                                 size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_vae_loss_eval, pre, image, stats, recon_out, out5, B, chw, L, beta, flags, ws, ws_bytes, stream);
  return vae_loss_launch(pre, image, stats, recon_out, nullptr, nullptr, out5, B, chw, L, beta, flags, ws, ws_bytes,
                         stream, false);
}

// ---- allocation-free entry point ---------------------------------------------------------------
// gmk_vae_loss_fwd_bwd_static: the same two kernels, with the block partials in a file-scope __device__ array
// instead of a caller workspace, for callers that can neither allocate nor carry a workspace (HIP graph capture,
// serving loops).  The array's name is only a host-side stand-in: the device address is resolved once per device
// with hipGetSymbolAddress (each GPU holds its own copy at its own address) and that pointer -- never the name --
// is what the kernels receive.  Nothing in it has to start at zero: every block overwrites its own partial slots
// and the finalize kernel reads only the slots of the blocks that ran.  Calls that overlap on one GPU (two streams)
// share the array and race; calls on one stream are ordered.
#define GMK_LOSS_STATIC_MAX_DEVICES 64
static __device__ __align__(256) double scratch[4 * LOSS_BLOCKS];   // 32 KiB per GPU, held until the process ends
static double* g_scratch_ptr[GMK_LOSS_STATIC_MAX_DEVICES];          // per-device address of `scratch`
static std::mutex g_scratch_mu;

static int loss_scratch_ptr(double** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  GMK_REQUIRE(dev >= 0 && dev < GMK_LOSS_STATIC_MAX_DEVICES, GMK_E_UNSUPPORTED);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {   // first call on this device (it is the current one): resolve and keep the address
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    GMK_REQUIRE(p != nullptr, GMK_E_NULL);
    g_scratch_ptr[dev] = (double*)p;
  }
  *out = g_scratch_ptr[dev];
  return GMK_OK;
}

extern "C" int gmk_vae_loss_fwd_bwd_static(const float* pre, const float* image, const float* stats,
                                           float* recon_out, float* dpre, float* dstats, float* losses, int64_t B,
                                           int64_t chw, int64_t L, float beta, int flags, gmk_stream_t stream) {
  GMK_REQUIRE(B > 0 && chw > 0 && L >= 0, GMK_E_SHAPE);
  // stated maximum: 2^28 elements per tensor (1 GiB of fp32; 87,381 CIFAR images): within it a thread adds at most
  // 1,024 fp32 terms before its sum goes to fp64
  GMK_REQUIRE(B <= GMK_LOSS_STATIC_MAX_ELEMS && chw <= GMK_LOSS_STATIC_MAX_ELEMS && L <= GMK_LOSS_STATIC_MAX_ELEMS &&
              B * chw <= GMK_LOSS_STATIC_MAX_ELEMS && B * L <= GMK_LOSS_STATIC_MAX_ELEMS, GMK_E_TOO_LARGE);
  double* ptr = nullptr;
  int rc = loss_scratch_ptr(&ptr);
  if (rc != GMK_OK) return rc;
  return vae_loss_launch(pre, image, stats, recon_out, dpre, dstats, losses, B, chw, L, beta, flags, ptr,
                         sizeof(double) * 4 * LOSS_BLOCKS, stream, true);
}

// ---- reparameterisation ---------------------------------------------------------------
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float2* __restrict__ stats,
                                                          const float* __restrict__ eps, float* __restrict__ z,
                                                          int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float2 s = stats[i];
    z[i] = s.x + expf(0.5f * s.y) * eps[i];
  }
}

template <bool ACC>
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ dz, const float2* __restrict__ stats,
                                                          const float* __restrict__ eps, float2* __restrict__ dstats,
                                                          int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float2 s = stats[i];
    float g = dz[i];
    float2 d;
    d.x = g;
    d.y = g * eps[i] * 0.5f * expf(0.5f * s.y);
    if (ACC) { float2 o = dstats[i]; d.x += o.x; d.y += o.y; }
    dstats[i] = d;
  }
}

extern "C" int gmk_reparam_fwd(const float* stats, const float* eps, float* z, int64_t B, int64_t L,
                               gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_reparam_fwd, stats, eps, z, B, L, stream);
  GMK_REQUIRE(stats && eps && z, GMK_E_NULL);
  GMK_REQUIRE(B > 0 && L > 0, GMK_E_SHAPE);
  int64_t n = B * L;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float2*)stats, eps, z, n);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

extern "C" int gmk_reparam_bwd(const float* dz, const float* stats, const float* eps, float* dstats, int64_t B,
                               int64_t L, int accumulate, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_reparam_bwd, dz, stats, eps, dstats, B, L, accumulate, stream);
  GMK_REQUIRE(dz && stats && eps && dstats, GMK_E_NULL);
  GMK_REQUIRE(B > 0 && L > 0, GMK_E_SHAPE);
  int64_t n = B * L;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  if (accumulate)
    hipLaunchKernelGGL(reparam_bwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, (const float2*)stats, eps, (float2*)dstats, n);
  else
    hipLaunchKernelGGL(reparam_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, (const float2*)stats, eps, (float2*)dstats, n);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// pool.hip -- 2x2 max-pool and x2 upsample (nearest / bilinear / bicubic), forward and backward.
// Bandwidth-shaped elementwise kernels for the pool/upsample variants of the autoencoder
// (/root/reference/models/autoencoders.py:40,48,52,89,100,104).  PyTorch semantics:
// max-pool ties go to the first element in (row, col) scan order; interpolation uses
// align_corners=False, bicubic with A = -0.75 and border-clamped taps.
#include "gmk_common.h"
#include "plan.h"

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ idx, long long planes, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long long total = planes * OH * OW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
    const long long pl = e / ((long long)OW * OH);
    const float* p = x + (pl * H + 2 * oy) * W + 2 * ox;
    float best = p[0];
    int k = 0;
    float v = p[1];     if (v > best || v != v) { best = v; k = 1; }
    v = p[W];           if (v > best || v != v) { best = v; k = 2; }
    v = p[W + 1];       if (v > best || v != v) { best = v; k = 3; }
    y[e] = best;
    if (idx) idx[e] = (unsigned char)k;
  }
}

// dx (planes,H,W) = scatter of dy (planes,H/2,W/2) to the arg-max, times dact_in (optional)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                           const float* __restrict__ dact_in, float* __restrict__ dx,
                                                           long long planes, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long long total = planes * OH * OW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
    const long long pl = e / ((long long)OW * OH);
    const long long base = (pl * H + 2 * oy) * W + 2 * ox;
    const float g = dy[e];
    const int k = idx[e];
    float v0 = k == 0 ? g : 0.f, v1 = k == 1 ? g : 0.f, v2 = k == 2 ? g : 0.f, v3 = k == 3 ? g : 0.f;
    if (dact_in) { v0 *= dact_in[base]; v1 *= dact_in[base + 1]; v2 *= dact_in[base + W]; v3 *= dact_in[base + W + 1]; }
    dx[base] = v0; dx[base + 1] = v1; dx[base + W] = v2; dx[base + W + 1] = v3;
  }
}

// ---- x2 interpolation taps (per dimension) ----
struct Taps { int i[4]; float w[4]; int n; };

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ Taps up2_taps(int mode, int o, int in) {
  Taps t;
  if (mode == GMK_UP_NEAREST) {
    t.n = 1; t.i[0] = o >> 1; t.w[0] = 1.f;
  } else if (mode == GMK_UP_BILINEAR) {
    float src = 0.5f * (o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const float l = src - i0;
    t.n = 2; t.i[0] = i0; t.i[1] = i0 + 1 < in ? i0 + 1 : in - 1; t.w[0] = 1.f - l; t.w[1] = l;
  } else {
    const float src = 0.5f * (o + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const int i0 = (int)fl;
    const float tt = src - fl;
    const float A = -0.75f;
    t.n = 4;
    t.w[0] = cubic2(tt + 1.f, A); t.w[1] = cubic1(tt, A); t.w[2] = cubic1(1.f - tt, A); t.w[3] = cubic2(2.f - tt, A);
#pragma unroll
    for (int k = 0; k < 4; ++k) { int ii = i0 - 1 + k; t.i[k] = ii < 0 ? 0 : (ii > in - 1 ? in - 1 : ii); }
  }
  return t;
}

__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long planes, int H, int W, int mode) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = planes * OH * OW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
    const long long pl = e / ((long long)OW * OH);
    const float* p = x + pl * H * W;
    const Taps ty = up2_taps(mode, oy, H), tx = up2_taps(mode, ox, W);
    float acc = 0.f;
    for (int a = 0; a < ty.n; ++a) {
      float row = 0.f;
      for (int b = 0; b < tx.n; ++b) row += tx.w[b] * p[ty.i[a] * W + tx.i[b]];
      acc += ty.w[a] * row;
    }
    y[e] = acc;
  }
}

// weight with which output index o reads input index i (sum over clamped taps)
__device__ __forceinline__ float up2_weight(int mode, int o, int i, int in) {
  const Taps t = up2_taps(mode, o, in);
  float w = 0.f;
  for (int k = 0; k < t.n; ++k) if (t.i[k] == i) w += t.w[k];
  return w;
}

// dx (planes,H,W) = transpose-interpolation of dy (planes,2H,2W), times dact_in (optional)
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dact_in,
                                                            float* __restrict__ dx, long long planes, int H, int W, int mode) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = planes * H * W;
  const int R = mode == GMK_UP_NEAREST ? 0 : (mode == GMK_UP_BILINEAR ? 2 : 5);  // reach in output rows/cols
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H);
    const long long pl = e / ((long long)W * H);
    const float* g = dy + pl * OH * OW;
    float acc = 0.f;
    if (mode == GMK_UP_NEAREST) {
      const float* q = g + (2 * iy) * OW + 2 * ix;
      acc = (q[0] + q[1]) + (q[OW] + q[OW + 1]);
    } else {
      const int oy0 = max(0, 2 * iy - R), oy1 = min(OH - 1, 2 * iy + 1 + R);
      const int ox0 = max(0, 2 * ix - R), ox1 = min(OW - 1, 2 * ix + 1 + R);
      for (int oy = oy0; oy <= oy1; ++oy) {
        const float wy = up2_weight(mode, oy, iy, H);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = ox0; ox <= ox1; ++ox) {
          const float wx = up2_weight(mode, ox, ix, W);
          if (wx != 0.f) row += wx * g[oy * OW + ox];
        }
        acc += wy * row;
      }
    }
    if (dact_in) acc *= dact_in[e];
    dx[e] = acc;
  }
}

__global__ __launch_bounds__(256) void ew_mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) out[e] = a[e] * b[e];
}

// ---- vectorised fast paths (power-of-two widths, 16-byte aligned planes: every shape of the autoencoder path) ----------
// One thread per FOUR output pixels of the pool / four input pixels of the nearest upsample: 16-byte loads and stores,
// shifts instead of 64-bit divisions.  The generic kernels above took 1.6 TB/s (upsample forward) to 3.9 TB/s (pool forward)
// on the 64x64 configuration (configs[4], 419 / 440 MB per step through them); same arithmetic, same results.
__global__ __launch_bounds__(256) void maxpool2_fwd_v4_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                              unsigned* __restrict__ idx, long long nquads, int lgOWq, int W) {
  const int OWq = 1 << lgOWq, Wq = W >> 2;   // output quads per row; input float4 per row
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long long)gridDim.x * 256) {
    const long long orow = q >> lgOWq;       // pl * OH + oy  ->  input rows 2 orow, 2 orow + 1
    const int xq = (int)(q & (OWq - 1));
    const float4* r0 = x + (2 * orow) * Wq + 2 * xq;
    const float4 a0 = r0[0], a1 = r0[1], b0 = r0[Wq], b1 = r0[Wq + 1];
    float4 o;
    unsigned kk = 0;
    auto one = [&](float p0, float p1, float p2, float p3, float& out, int sh) {
      float best = p0; unsigned k = 0;
      if (p1 > best || p1 != p1) { best = p1; k = 1; }
      if (p2 > best || p2 != p2) { best = p2; k = 2; }
      if (p3 > best || p3 != p3) { best = p3; k = 3; }
      out = best; kk |= k << sh;
    };
    one(a0.x, a0.y, b0.x, b0.y, o.x, 0);
    one(a0.z, a0.w, b0.z, b0.w, o.y, 8);
    one(a1.x, a1.y, b1.x, b1.y, o.z, 16);
    one(a1.z, a1.w, b1.z, b1.w, o.w, 24);
    y[q] = o;
    if (idx) idx[q] = kk;
  }
}

__global__ __launch_bounds__(256) void upsample2_nearest_fwd_v4_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                                       long long nquads, int lgWq) {
  const int Wq = 1 << lgWq;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long long)gridDim.x * 256) {
    const long long row = q >> lgWq;         // pl * H + iy  ->  output rows 2 row, 2 row + 1 (2 Wq float4 each)
    const int xq = (int)(q & (Wq - 1));
    const float4 v = x[q];
    const float4 a = make_float4(v.x, v.x, v.y, v.y), b = make_float4(v.z, v.z, v.w, v.w);
    float4* o = y + (2 * row) * (2 * Wq) + 2 * xq;
    o[0] = a; o[1] = b;
    o[2 * Wq] = a; o[2 * Wq + 1] = b;
  }
}

__global__ __launch_bounds__(256) void upsample2_nearest_bwd_v4_kernel(const float4* __restrict__ dy, const float4* __restrict__ dact_in,
                                                                       float4* __restrict__ dx, long long nquads, int lgWq) {
  const int Wq = 1 << lgWq;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long long)gridDim.x * 256) {
    const long long row = q >> lgWq;
    const int xq = (int)(q & (Wq - 1));
    const float4* g = dy + (2 * row) * (2 * Wq) + 2 * xq;
    const float4 a0 = g[0], a1 = g[1], b0 = g[2 * Wq], b1 = g[2 * Wq + 1];
    // the generic kernel's summation order: (q[0] + q[1]) + (q[OW] + q[OW + 1])
    float4 r = make_float4((a0.x + a0.y) + (b0.x + b0.y), (a0.z + a0.w) + (b0.z + b0.w),
                           (a1.x + a1.y) + (b1.x + b1.y), (a1.z + a1.w) + (b1.z + b1.w));
    if (dact_in) { const float4 d = dact_in[q]; r.x *= d.x; r.y *= d.y; r.z *= d.z; r.w *= d.w; }
    dx[q] = r;
  }
}

static inline int blocks_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int gmk_maxpool2_fwd(const float* x, float* y, unsigned char* idx, int64_t planes, int H, int W, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_maxpool2_fwd, x, y, idx, planes, H, W, stream);
  GMK_REQUIRE(x && y, GMK_E_NULL);
  GMK_REQUIRE(planes > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, GMK_E_SHAPE);
  if (W % 8 == 0 && gmk_is_pow2(W) && gmk_aligned16(x) && gmk_aligned16(y) && (((uintptr_t)idx) & 3u) == 0) {
    const long long nq = planes * (H / 2) * (W / 8);
    hipLaunchKernelGGL(maxpool2_fwd_v4_kernel, dim3(blocks_for(nq)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y,
                       (unsigned*)idx, nq, gmk_ilog2(W / 8), W);
    GMK_CHECK_LAUNCH();
    return GMK_OK;
  }
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(blocks_for(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, y, idx, (long long)planes, H, W);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}
extern "C" int gmk_maxpool2_bwd(const float* dy, const unsigned char* idx, const float* dact_in, float* dx, int64_t planes, int H, int W, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_maxpool2_bwd, dy, idx, dact_in, dx, planes, H, W, stream);
  GMK_REQUIRE(dy && idx && dx, GMK_E_NULL);
  GMK_REQUIRE(planes > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, GMK_E_SHAPE);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(blocks_for(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, dy, idx, dact_in, dx, (long long)planes, H, W);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}
extern "C" int gmk_upsample2_fwd(const float* x, float* y, int64_t planes, int H, int W, int mode, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_upsample2_fwd, x, y, planes, H, W, mode, stream);
  GMK_REQUIRE(x && y, GMK_E_NULL);
  GMK_REQUIRE(planes > 0 && H >= 1 && W >= 1, GMK_E_SHAPE);
  GMK_REQUIRE(mode >= GMK_UP_NEAREST && mode <= GMK_UP_BICUBIC, GMK_E_UNSUPPORTED);
  if (mode == GMK_UP_NEAREST && W % 4 == 0 && gmk_is_pow2(W) && gmk_aligned16(x) && gmk_aligned16(y)) {
    const long long nq = planes * H * (W / 4);
    hipLaunchKernelGGL(upsample2_nearest_fwd_v4_kernel, dim3(blocks_for(nq)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (float4*)y, nq, gmk_ilog2(W / 4));
    GMK_CHECK_LAUNCH();
    return GMK_OK;
  }
  hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(blocks_for(planes * 4 * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)planes, H, W, mode);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}
extern "C" int gmk_upsample2_bwd(const float* dy, const float* dact_in, float* dx, int64_t planes, int H, int W, int mode, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_upsample2_bwd, dy, dact_in, dx, planes, H, W, mode, stream);
  GMK_REQUIRE(dy && dx, GMK_E_NULL);
  GMK_REQUIRE(planes > 0 && H >= 1 && W >= 1, GMK_E_SHAPE);
  GMK_REQUIRE(mode >= GMK_UP_NEAREST && mode <= GMK_UP_BICUBIC, GMK_E_UNSUPPORTED);
  if (mode == GMK_UP_NEAREST && W % 4 == 0 && gmk_is_pow2(W) && gmk_aligned16(dy) && gmk_aligned16(dx) && gmk_aligned16(dact_in)) {
    const long long nq = planes * H * (W / 4);
    hipLaunchKernelGGL(upsample2_nearest_bwd_v4_kernel, dim3(blocks_for(nq)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy,
                       (const float4*)dact_in, (float4*)dx, nq, gmk_ilog2(W / 4));
    GMK_CHECK_LAUNCH();
    return GMK_OK;
  }
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(blocks_for(planes * H * W)), dim3(256), 0, (hipStream_t)stream, dy, dact_in, dx, (long long)planes, H, W, mode);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}
extern "C" int gmk_ew_mul(const float* a, const float* b, float* out, int64_t n, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_ew_mul, a, b, out, n, stream);
  GMK_REQUIRE(a && b && out, GMK_E_NULL);
  GMK_REQUIRE(n > 0, GMK_E_SHAPE);
  hipLaunchKernelGGL(ew_mul_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// conv_wgrad_api.hip -- C-ABI entry points for conv / deconv weight + bias gradients.
#include "conv_wgrad.h"

#define CHSUM_SPLIT 32

// per-channel sum over (n, pixels): grid (C, split); fp64 partials, fixed-order finalize.
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ t, double* __restrict__ partial,
                                                                  int N, int C, int HW, int split) {
  const int c = blockIdx.x, sp = blockIdx.y;
  const int n_per = (N + split - 1) / split;
  const int n0 = sp * n_per, n1 = min(N, n0 + n_per);
  float acc = 0.f;
  for (int n = n0; n < n1; ++n) {
    const float* p = t + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += 256) acc += p[i];
  }
  __shared__ double red[4];
  double v = gmk_wave_sum((double)acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[c * split + sp] = red[0] + red[1] + red[2] + red[3];
}
__global__ void channel_sum_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int C, int split) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0;
  for (int i = 0; i < split; ++i) s += partial[c * split + i];
  out[c] = (float)s;
}

static int channel_sum(const float* t, float* out, int N, int C, int HW, double* partial, hipStream_t s) {
  const int split = N < CHSUM_SPLIT ? N : CHSUM_SPLIT;
  hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(C, split), dim3(256), 0, s, t, partial, N, C, HW, split);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, partial, out, C, split);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

static void wgrad_plan(WgradGeom& g, int N, int Cs, int Cb, int SH, int SW, int ksize, int sb) {
  g.N = N; g.Cs = Cs; g.Cb = Cb; g.SH = SH; g.SW = SW; g.BHt = SH * sb; g.BWt = SW * sb;
  g.lgSW = gmk_ilog2(SW); g.lgSHW = gmk_ilog2((int64_t)SH * SW);
  g.P = gmk_wgrad_chunk_px(SH, SW, sb); g.lgP = gmk_ilog2(g.P);
  g.total_chunks = ((long long)N * SH * SW) / g.P;
  g.csT = (Cs + 63) / 64; g.cbT = (Cb + 63) / 64;
  const int ntg = ksize == 3 ? 1 : ksize;
  long long want = 256 / ((long long)g.csT * g.cbT * ntg);
  if (want < 32) want = 32;
  g.G = (int)(g.total_chunks < want ? g.total_chunks : want);
}

static size_t slab_bytes(const WgradGeom& g, int ksize) {
  const int KK = ksize * ksize;
  return sizeof(float) * (size_t)g.G * g.csT * g.cbT * KK * 4096;
}

static size_t wgrad_ws_bytes(int N, int Cs, int Cb, int SH, int SW, int ksize, int sb, int Cbias) {
  WgradGeom g{};
  wgrad_plan(g, N, Cs, Cb, SH, SW, ksize, sb);
  return slab_bytes(g, ksize) + sizeof(double) * (size_t)Cbias * CHSUM_SPLIT + 256;
}

extern "C" size_t gmk_conv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize, int stride) {
  return wgrad_ws_bytes(N, Cout, Cin, H / stride, W / stride, ksize, stride, Cout);
}
extern "C" size_t gmk_deconv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize) {
  return wgrad_ws_bytes(N, Cin, Cout, H, W, ksize, 2, Cout);
}

static int wgrad_run(const float* small, const float* big, float* dw, const float* dy_for_bias, float* dbias,
                     int Cbias, int HWbias, int N, int Cs, int Cb, int SH, int SW, int ksize, int sb, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  if (ksize != 3 && ksize != 5) return GMK_E_UNSUPPORTED;
  if (!gmk_is_pow2(SH) || !gmk_is_pow2(SW) || SW * sb > 128 || SH * SW < 16) return GMK_E_SHAPE;
  if (!gmk_aligned16(ws)) return GMK_E_ALIGN;
  WgradGeom g{};
  wgrad_plan(g, N, Cs, Cb, SH, SW, ksize, sb);
  const size_t sb_bytes = slab_bytes(g, ksize);
  if (ws_bytes < sb_bytes + sizeof(double) * (size_t)Cbias * CHSUM_SPLIT) return GMK_E_WORKSPACE;
  g.s = small; g.b = big; g.slab = (float*)ws;
  int rc = ksize == 3 ? gmk_launch_wgrad_k3(g, sb, dw, s) : gmk_launch_wgrad_k5(g, sb, dw, s);
  if (rc) return rc;
  if (dbias) {
    double* partial = (double*)((char*)ws + ((sb_bytes + 15) & ~(size_t)15));
    rc = channel_sum(dy_for_bias, dbias, N, Cbias, HWbias, partial, s);
  }
  return rc;
}

extern "C" int gmk_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Cin, int H,
                                int W, int Cout, int ksize, int stride, void* ws, size_t ws_bytes,
                                gmk_stream_t stream) {
  GMK_REQUIRE(dy && x && dw && ws, GMK_E_NULL);
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2), GMK_E_SHAPE);
  return wgrad_run(dy, x, dw, dy, dbias, Cout, (H / stride) * (W / stride), N, Cout, Cin, H / stride, W / stride,
                   ksize, stride, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int gmk_deconv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Cin, int H,
                                  int W, int Cout, int ksize, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_REQUIRE(dy && x && dw && ws, GMK_E_NULL);
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0, GMK_E_SHAPE);
  return wgrad_run(x, dy, dw, dy, dbias, Cout, 4 * H * W, N, Cin, Cout, H, W, ksize, 2, ws, ws_bytes,
                   (hipStream_t)stream);
}

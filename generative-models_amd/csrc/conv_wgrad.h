// conv_wgrad.h -- weight gradient of conv / transposed conv as a split-K GEMM on fp32 MFMA.
//
//   dW[cs][cb][t] = sum_{n, sy, sx}  S[n][cs][sy][sx] * B[n][cb][sy*SB + ky - PAD][sx*SB + kx - PAD]
// "small" tensor S and "big" tensor B:
//   conv   (stride SB): S = dY (cs = out channel), B = X  (cb = in channel)  -> dW is OIHW
//   deconv (stride 2) : S = X  (cs = in channel),  B = dY (cb = out channel) -> dW is IOHW
// GEMM view: M = cs (A operand), N = cb (B operand), K = pixels (reduction), one accumulator per tap.
// Each block reduces its pixel chunks into registers and writes ONE partial slab; a second kernel
// sums the slabs in a fixed order (deterministic, no atomics).
#pragma once
#include "gmk_common.h"

struct WgradGeom {
  const float* s; const float* b; float* slab;
  int N, Cs, Cb, SH, SW, BHt, BWt;
  int lgSW, lgSHW, P, lgP;
  long long total_chunks;
  int G;               // pixel-chunk groups (gridDim.x)
  int csT, cbT;        // 64-channel tiles
  int PR, PC, PSb, npos;
};

template <int KS, int SB, int NT /* taps per block: KS*KS or KS (one ky row) */>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2, NTG = KK / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P = g.P, SS = P + 1, PSb = g.PSb, PC = g.PC;
  float* Sl = smem;            // [64][SS]
  float* Bl = smem + 64 * SS;  // [64][PSb]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = wave & 1, bq = wave >> 1;
  const int h = lane >> 5, jj = lane & 31;
  // blockIdx.y = ((csTile * cbT) + cbTile) * NTG + tapGroup
  const int tg = blockIdx.y % NTG;
  const int tile = blockIdx.y / NTG;
  const int cs0 = (tile / g.cbT) * 64, cb0 = (tile % g.cbT) * 64;
  const int SHW = 1 << g.lgSHW, BHW = g.BHt * g.BWt;
  const int t_first = tg * NT;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // this thread's patch position (npos <= 256: one slot)
  const int prow = tid / PC, pcol = tid % PC;
  const int abase = (a * 32 + jj) * SS + h;
  const int bbase = (bq * 32 + jj) * PSb;

  for (long long ch = blockIdx.x; ch < g.total_chunks; ch += g.G) {
    const long long pix = ch << g.lgP;           // flat small-pixel index of the chunk start
    const long long n = pix >> g.lgSHW;
    const int pin = (int)(pix & (SHW - 1));      // offset inside the image
    const int r0 = pin >> g.lgSW;                // first small row of the chunk
    __syncthreads();
    // stage S tile: 64 channels x P pixels (coalesced along pixels)
    for (int e = tid; e < 64 * P; e += 256) {
      const int c = e >> g.lgP, p = e & (P - 1);
      float v = 0.f;
      if (cs0 + c < g.Cs) v = g.s[((n * g.Cs + cs0 + c) << g.lgSHW) + pin + p];
      Sl[c * SS + p] = v;
    }
    // stage big patch: 64 channels x (PR x PC), zero padded
    if (tid < g.npos) {
      const int iy = r0 * SB + prow - PAD, ix = pcol - PAD;
      const bool ok = iy >= 0 && iy < g.BHt && ix >= 0 && ix < g.BWt;
      const long long goff = (n * g.Cb + cb0) * (long long)BHW + iy * g.BWt + ix;
      const int cmax = g.Cb - cb0;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) {
        float v = 0.f;
        if (ok && c < cmax) v = g.b[goff + (long long)c * BHW];
        Bl[c * PSb + tid] = v;
      }
    }
    __syncthreads();
    for (int s2 = 0; s2 < P / 2; ++s2) {
      const int p = 2 * s2 + h;
      const float av = Sl[abase + 2 * s2];
      const int poff = bbase + ((p >> g.lgSW) * SB) * PC + (p & (g.SW - 1)) * SB;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tt = t_first + t;   // NT == KK: compile-time; NT == KS: ky = tg (uniform), kx = t
        int ky, kx;
        if (NT == KK) { ky = t / KS; kx = t % KS; } else { ky = tg; kx = t; }
        (void)tt;
        const float bv = Bl[poff + ky * PC + kx];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  // write the partial slab: slab[(blockIdx.x * gridDim.y + blockIdx.y)][t][cs 64][cb 64]
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(NT * 64 * 64);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cs = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(t * 64 + cs) * 64 + bq * 32 + jj] = acc[t][r];
    }
}

// dW[cs][cb][t] = sum_g slab[g][tile,tg][t][cs][cb]   (fixed order over g)
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int G, int ny, int Cs, int Cb, int cbT) {
  constexpr int KK = KS * KS, NTG = KK / NT;
  const int y = blockIdx.y;  // (tile, tg)
  const int tg = y % NTG, tile = y / NTG;
  const int cs0 = (tile / cbT) * 64, cb0 = (tile % cbT) * 64;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < NT * 64 * 64; e += gridDim.x * 256) {
    const int t = e / 4096, cs = (e / 64) % 64, cb = e % 64;
    if (cs0 + cs >= Cs || cb0 + cb >= Cb) continue;
    float sum = 0.f;
    for (int gI = 0; gI < G; ++gI) sum += slab[((size_t)gI * ny + y) * (size_t)(NT * 4096) + e];
    dw[((size_t)(cs0 + cs) * Cb + cb0 + cb) * KK + tg * NT + t] = sum;
  }
}


int gmk_launch_wgrad_k3(WgradGeom& g, int sb, float* dw, hipStream_t s);
int gmk_launch_wgrad_k5(WgradGeom& g, int sb, float* dw, hipStream_t s);

static inline int gmk_wgrad_chunk_px(int SH, int SW, int sb) {
  int p = sb == 1 ? 64 : 32;
  if (p > SH * SW) p = SH * SW;
  if (p < SW) p = SW;
  return p;
}

template <int KS>
static int gmk_launch_wgrad_impl(WgradGeom& g, int sb, float* dw, hipStream_t s) {
  constexpr int KK = KS * KS, NT = (KS == 3) ? KK : KS, NTG = KK / NT;
  const int rows = g.P / g.SW;
  if (sb == 1) { g.PR = rows + KS - 1; g.PC = g.SW + KS - 1; }
  else { g.PR = 2 * rows + KS - 2; g.PC = 2 * g.SW + KS - 2; }
  g.npos = g.PR * g.PC;
  if (g.npos > 256) return GMK_E_SHAPE;
  g.PSb = g.npos | 1;
  const size_t lds = sizeof(float) * (64 * (size_t)(g.P + 1) + 64 * (size_t)g.PSb);
  dim3 grid(g.G, g.csT * g.cbT * NTG);
  if (sb == 1) {
    auto kern = conv_wgrad_kernel<KS, 1, NT>;
    if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
  } else {
    auto kern = conv_wgrad_kernel<KS, 2, NT>;
    if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_wgrad_reduce_kernel<KS, NT>), dim3((NT * 4096 + 255) / 256, grid.y), dim3(256), 0, s,
                     g.slab, dw, g.G, (int)grid.y, g.Cs, g.Cb, g.cbT);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

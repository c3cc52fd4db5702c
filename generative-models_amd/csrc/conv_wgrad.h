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
  int bias_mode;       // 0 none, 1 dY = small tensor, 2 dY = big tensor
};

template <int KS, int SB, int NT /* taps per block: KS*KS or KS (one ky row) */, int PX /* pixels per chunk */>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2, NTG = KK / NT;
  constexpr int SS = PX + 1;               // S tile row stride (odd: conflict-free A reads)
  constexpr int NS = (64 * PX) / 256;      // S elements per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PSb = g.PSb, PC = g.PC;
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
  const int lgP = g.lgP;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // bias-gradient partial of channel `tid` (threads 0..63)

  // this thread's patch position (npos <= 256: one slot) and S-tile elements
  const int prow = tid / PC, pcol = tid % PC;
  const bool has_pos = tid < g.npos;
  const int abase = (a * 32 + jj) * SS + h;
  const int bbase = (bq * 32 + jj) * PSb;
  const int cmax_s = g.Cs - cs0, cmax_b = g.Cb - cb0;

  float sr[NS], br[64];
  // branch-free batched loads of one pixel chunk into registers (buffer loads: OOB lanes read 0)
  auto load_chunk = [&](long long ch) {
    const long long pix = ch << lgP;
    const long long n = pix >> g.lgSHW;
    const int pin = (int)(pix & (SHW - 1));
    const int r0 = pin >> g.lgSW;
    const __amdgpu_buffer_rsrc_t rs = gmk_make_rsrc(g.s + ((n * g.Cs + cs0) << g.lgSHW), (long long)(cmax_s < 64 ? cmax_s : 64) * SHW * 4);
    const __amdgpu_buffer_rsrc_t rb = gmk_make_rsrc(g.b + (n * g.Cb + cb0) * (long long)BHW, (long long)(cmax_b < 64 ? cmax_b : 64) * BHW * 4);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      const int c = e >> lgP, p = e & (PX - 1);
      sr[u] = gmk_buffer_load(rs, ((c << g.lgSHW) + pin + p) * 4, 0);  // c >= cmax_s falls outside the descriptor
    }
    const int iy = r0 * SB + prow - PAD, ix = pcol - PAD;
    const bool okp = has_pos && iy >= 0 && iy < g.BHt && ix >= 0 && ix < g.BWt;
    const int voff = okp ? (iy * g.BWt + ix) * 4 : GMK_OOB;
#pragma unroll
    for (int c = 0; c < 64; ++c) br[c] = gmk_buffer_load(rb, voff, c * BHW * 4);  // c >= cmax_b: outside the descriptor
  };

  long long ch = blockIdx.x;
  if (ch < g.total_chunks) load_chunk(ch);
  for (; ch < g.total_chunks; ch += g.G) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      Sl[(e >> lgP) * SS + (e & (PX - 1))] = sr[u];
    }
    if (has_pos) {
#pragma unroll
      for (int c = 0; c < 64; ++c) Bl[c * PSb + tid] = br[c];
    }
    __syncthreads();
    if (ch + g.G < g.total_chunks) load_chunk(ch + g.G);  // next chunk in flight under the MFMAs
    // bias gradient: per-channel sum of dY over this chunk (bias_mode 1: dY is the small tensor,
    // 2: dY is the big tensor -> interior of the patch only, halo belongs to neighbouring chunks)
    if (g.bias_mode == 1) {
      if (tid < 64) {
        float sum = 0.f;
#pragma unroll 8
        for (int p = 0; p < PX; ++p) sum += Sl[tid * SS + p];
        bsum += sum;
      }
    } else if (g.bias_mode == 2) {
      if (tid < 64) {
        const int rows = (PX >> g.lgSW) * SB;
        float sum = 0.f;
        for (int r = 0; r < rows; ++r)
          for (int x = 0; x < g.BWt; ++x) sum += Bl[tid * PSb + (r + PAD) * PC + x + PAD];
        bsum += sum;
      }
    }
    for (int s2 = 0; s2 < PX / 2; ++s2) {
      const int p = 2 * s2 + h;
      const float av = Sl[abase + 2 * s2];
      const int poff = bbase + ((p >> g.lgSW) * SB) * PC + (p & (g.SW - 1)) * SB;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int ky, kx;
        if (NT == KK) { ky = t / KS; kx = t % KS; } else { ky = tg; kx = t; }
        const float bv = Bl[poff + ky * PC + kx];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  // write the partial slab: slab[(blockIdx.x * gridDim.y + blockIdx.y)][t][cs 64][cb 64] (+ 64 bias partials)
  constexpr int SLAB = NT * 64 * 64 + 64;
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)SLAB;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cs = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(t * 64 + cs) * 64 + bq * 32 + jj] = acc[t][r];
    }
  if (tid < 64) out[NT * 4096 + tid] = bsum;
}

// dW[cs][cb][t] = sum_g slab[g][tile,tg][t][cs][cb]   (fixed order: 4 interleaved partial sums over g,
// combined in a fixed tree) ; dbias likewise from the 64 trailing floats of each slab.
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int G, int ny, int Cs,
                                                                int Cb, int cbT, int bias_mode) {
  constexpr int KK = KS * KS, NTG = KK / NT, SLAB = NT * 4096 + 64;
  __shared__ float red[4][64];
  const int y = blockIdx.y;  // (tile, tg)
  const int tg = y % NTG, tile = y / NTG;
  const int cs0 = (tile / cbT) * 64, cb0 = (tile % cbT) * 64;
  const int el = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;  // element of the slab (incl. the bias tail)
  float sum = 0.f;
  if (e < SLAB)
    for (int gI = gq; gI < G; gI += 4) sum += slab[((size_t)gI * ny + y) * (size_t)SLAB + e];
  red[gq][el] = sum;
  __syncthreads();
  if (gq != 0 || e >= SLAB) return;
  sum = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
  if (e < NT * 4096) {
    const int t = e / 4096, cs = (e / 64) % 64, cb = e % 64;
    if (cs0 + cs < Cs && cb0 + cb < Cb) dw[((size_t)(cs0 + cs) * Cb + cb0 + cb) * KK + tg * NT + t] = sum;
  } else if (dbias && tg == 0) {
    const int c = e - NT * 4096;
    // bias_mode 1: channel of the small tensor (only the cbTile == 0 blocks carry it); 2: of the big tensor
    if (bias_mode == 1 && (tile % cbT) == 0 && cs0 + c < Cs) dbias[cs0 + c] = sum;
    if (bias_mode == 2 && (tile / cbT) == 0 && cb0 + c < Cb) dbias[cb0 + c] = sum;
  }
}

int gmk_launch_wgrad_k3(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);
int gmk_launch_wgrad_k5(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);

static inline int gmk_wgrad_chunk_px(int SH, int SW, int sb) {
  int p = sb == 1 ? 64 : 32;
  if (p > SH * SW) p = SH * SW;
  if (p < SW) p = SW;
  return p;
}

template <int KS>
static int gmk_launch_wgrad_impl(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) {
  constexpr int KK = KS * KS, NT = (KS == 3) ? KK : KS, NTG = KK / NT;
  const int rows = g.P / g.SW;
  if (sb == 1) { g.PR = rows + KS - 1; g.PC = g.SW + KS - 1; }
  else { g.PR = 2 * rows + KS - 2; g.PC = 2 * g.SW + KS - 2; }
  g.npos = g.PR * g.PC;
  if (g.npos > 256) return GMK_E_SHAPE;
  g.PSb = g.npos | 1;
  const size_t lds = sizeof(float) * (64 * (size_t)(g.P + 1) + 64 * (size_t)g.PSb);
  dim3 grid(g.G, g.csT * g.cbT * NTG);
#define GMK_WG_LAUNCH(SB_, PX_)                                                                                    \
  do {                                                                                                             \
    auto kern = conv_wgrad_kernel<KS, SB_, NT, PX_>;                                                               \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                          \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);                                                          \
  } while (0)
  if (sb == 1) {
    if (g.P == 64) GMK_WG_LAUNCH(1, 64); else if (g.P == 32) GMK_WG_LAUNCH(1, 32); else if (g.P == 16) GMK_WG_LAUNCH(1, 16);
    else return GMK_E_SHAPE;
  } else {
    if (g.P == 64) GMK_WG_LAUNCH(2, 64); else if (g.P == 32) GMK_WG_LAUNCH(2, 32); else if (g.P == 16) GMK_WG_LAUNCH(2, 16);
    else return GMK_E_SHAPE;
  }
#undef GMK_WG_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  constexpr int SLAB = NT * 4096 + 64;
  hipLaunchKernelGGL((conv_wgrad_reduce_kernel<KS, NT>), dim3((SLAB + 63) / 64, grid.y), dim3(256), 0, s, g.slab, dw,
                     dbias, g.G, (int)grid.y, g.Cs, g.Cb, g.cbT, g.bias_mode);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

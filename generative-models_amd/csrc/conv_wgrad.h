// conv_wgrad.h -- weight gradient of conv / transposed conv as a split-K GEMM on fp32 MFMA.
//
//   dW[cs][cb][t] = sum_{n, sy, sx}  S[n][cs][sy][sx] * B[n][cb][sy*SB + ky - PAD][sx*SB + kx - PAD]
// "small" tensor S and "big" tensor B:
//   conv   (stride SB): S = dY (cs = out channel), B = X  (cb = in channel)  -> dW is OIHW
//   deconv (stride 2) : S = X  (cs = in channel),  B = dY (cb = out channel) -> dW is IOHW
// GEMM view: M = cs (A operand), N = cb (B operand), K = pixels (reduction), one accumulator per tap.
//
// A block owns ONE 32 x 32 (cs x cb) output tile with all its taps and a group of pixel chunks; its four
// waves split every chunk's pixels (K) four ways and hold a full tile of accumulators each.  At the end the
// four partial tiles are summed through LDS in a fixed order and ONE slab per block is written, so the
// slab traffic is G x tiles x 9216 floats (4x less than one slab per 64x64 tile and block); a second
// kernel sums the slabs in a fixed order (deterministic, no atomics).
#pragma once
#include "gmk_common.h"

struct WgradGeom {
  const float* s; const float* b; float* slab;
  int N, Cs, Cb, SH, SW, BHt, BWt;
  int lgSW, lgSHW, lgBWt, P, lgP;
  long long total_chunks;
  int G;               // pixel-chunk groups (gridDim.x)
  int csT, cbT;        // 32-channel tiles
  int PR, PC, PSb, npos;
  int bias_mode;       // 0 none, 1 dY = small tensor, 2 dY = big tensor
};

#define GMK_WG_T 32  // channel tile (both sides)

template <int KS, int SB, int NT /* taps per block: KS*KS or KS (one ky row) */, int PX /* pixels per chunk */,
          int NSLOT /* patch positions per thread */>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2, NTG = KK / NT, T = GMK_WG_T;
  constexpr int SS = PX + 4;               // S tile row stride: 16-byte rows, columns XOR-swizzled by (row>>3)&3
  constexpr int NS = (T * PX / 4) / 256;   // float4 loads of the S tile per thread and chunk
  constexpr int WPX = PX / 4;              // pixels per wave and chunk
  static_assert(NS >= 1 && WPX >= 2, "chunk too small");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PSb = g.PSb, PC = g.PC;
  float* Sl = smem;           // [32][SS]
  float* Bl = smem + T * SS;  // [32][PSb]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, jj = lane & 31;
  // blockIdx.y = ((csTile * cbT) + cbTile) * NTG + tapGroup
  const int tg = blockIdx.y % NTG;
  const int tile = blockIdx.y / NTG;
  const int cs0 = (tile / g.cbT) * T, cb0 = (tile % g.cbT) * T;
  const int SHW = 1 << g.lgSHW, BHW = g.BHt * g.BWt;
  const int cmax_s = (g.Cs - cs0) < T ? (g.Cs - cs0) : T, cmax_b = (g.Cb - cb0) < T ? (g.Cb - cb0) : T;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // bias-gradient partial: thread (channel tid & 31, part tid >> 5) sums 1/8 of the chunk's dY pixels
  float bsum = 0.f;
  const int bch = tid & 31, bpart = tid >> 5;
  const bool bias1 = g.bias_mode == 1 && tg == 0 && (tile % g.cbT) == 0;  // only the blocks the reduce kernel reads
  const bool bias2 = g.bias_mode == 2 && tg == 0 && (tile / g.cbT) == 0;

  // S tile: float4 e = tid + u*256 -> (channel, pixel quad); B patch: NSLOT positions per thread, 32 channels each
  int prow[NSLOT], pcol[NSLOT], pos_l[NSLOT];
  bool has_pos[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int e = tid + k * 256;
    prow[k] = e / PC; pcol[k] = e % PC;
    has_pos[k] = e < g.npos;
    pos_l[k] = has_pos[k] ? e : g.npos;  // masked slots write a trash column
  }
  const int asw = (jj >> 3) & 3;
  const int bbase = jj * PSb;
  int tapoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ky = (NT == KK) ? t / KS : tg, kx = (NT == KK) ? t % KS : t;
    tapoff[t] = ky * PC + kx;
  }

  float4 sr[NS];
  float br[NSLOT][T];
  // branch-free batched loads of one pixel chunk into registers (a chunk index past the end re-reads the
  // last chunk; its contribution is skipped).  Patch positions outside the image read offset GMK_OOB -> 0.
  const long long last_chunk = g.total_chunks - 1;
  auto load_chunk = [&](long long ch) {
    ch = ch < last_chunk ? ch : last_chunk;
    const long long pix = ch * PX;
    const long long n = pix >> g.lgSHW;
    const int pin = (int)(pix & (SHW - 1));
    const int r0 = pin >> g.lgSW;
    const float* sbase = g.s + ((n * g.Cs + cs0) << g.lgSHW) + pin;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      const int c = e / (PX / 4), p4 = e % (PX / 4);
      const int cc = c < cmax_s ? c : cmax_s - 1;  // ragged tile: re-read a valid channel, zeroed on the way to LDS
      sr[u] = *(const float4*)(sbase + ((size_t)cc << g.lgSHW) + p4 * 4);
    }
    const __amdgpu_buffer_rsrc_t rb = gmk_make_rsrc(g.b + (n * g.Cb + cb0) * (long long)BHW, (long long)cmax_b * BHW * 4);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int iy = r0 * SB + prow[k] - PAD, ix = pcol[k] - PAD;
      const bool okp = has_pos[k] & (iy >= 0) & (iy < g.BHt) & (ix >= 0) & (ix < g.BWt);
      const int voff = okp ? (iy * g.BWt + ix) * 4 : GMK_OOB;
#pragma unroll
      for (int c = 0; c < T; ++c) br[k][c] = gmk_buffer_load(rb, voff, c * BHW * 4);  // c >= cmax_b: outside the descriptor -> 0
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      const int c = e / (PX / 4), p4 = e % (PX / 4);
      const int sw = (c >> 3) & 3;
      float4 v = sr[u];
      if (c >= cmax_s) v = make_float4(0.f, 0.f, 0.f, 0.f);
      // element k of the quad goes to slot k ^ sw
      const float4 w1 = (sw & 1) ? make_float4(v.y, v.x, v.w, v.z) : v;
      const float4 w2 = (sw & 2) ? make_float4(w1.z, w1.w, w1.x, w1.y) : w1;
      *(float4*)(Sl + c * SS + p4 * 4) = w2;
    }
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
#pragma unroll
      for (int c = 0; c < T; ++c) Bl[c * PSb + pos_l[k]] = br[k][c];
  };

  long long ch = blockIdx.x;
  load_chunk(ch);
  for (; ch < g.total_chunks; ch += g.G) {
    gmk_lds_barrier();  // previous chunk's LDS reads are done
    store_chunk();
    gmk_lds_barrier();
    load_chunk(ch + g.G);  // next chunk in flight under the MFMAs
    // bias gradient: per-channel sum of dY over this chunk (bias_mode 1: dY is the small tensor,
    // 2: dY is the big tensor -> interior of the patch only, the halo belongs to neighbouring chunks)
    if (bias1) {
      float v[PX / 8];
#pragma unroll
      for (int k = 0; k < PX / 8; ++k) {
        const int p = bpart * (PX / 8) + k;
        v[k] = Sl[bch * SS + (p ^ ((bch >> 3) & 3))];
      }
#pragma unroll
      for (int k = 0; k < PX / 8; ++k) bsum += v[k];
    } else if (bias2) {
      const int cnt = ((PX >> g.lgSW) * SB) << g.lgBWt;  // interior pixels of the patch, split 8 ways
      for (int q = bpart; q < cnt; q += 8) {
        const int r = q >> g.lgBWt, x = q & (g.BWt - 1);
        bsum += Bl[bch * PSb + (r + PAD) * PC + x + PAD];
      }
    }
    // MFMA over this wave's pixels, software-pipelined by hand: the operands of k-step s+1 are read from LDS
    // into a second register set before the NT MFMAs of step s issue (with one wave per SIMD nothing else
    // hides the LDS latency; left alone the compiler reuses two registers and waits before every MFMA).
    {
      float av_n, bv_n[NT];
      auto read_step = [&](int s2) {
        const int p = wave * WPX + 2 * s2 + h;
        av_n = Sl[jj * SS + (p ^ asw)];
        const int poff = bbase + ((p >> g.lgSW) * SB) * PC + (p & (g.SW - 1)) * SB;
#pragma unroll
        for (int t = 0; t < NT; ++t) bv_n[t] = Bl[poff + tapoff[t]];
      };
      read_step(0);
#pragma unroll 2
      for (int s2 = 0; s2 < WPX / 2; ++s2) {
        const float av = av_n;
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bv_n[t];
        if (s2 + 1 < WPX / 2) read_step(s2 + 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
      }
    }
  }

  // ---- combine the four waves' partial tiles in LDS (fixed order), one slab per block ----
  // slab[(blockIdx.x * gridDim.y + blockIdx.y)][t][cs 32][cb 32] followed by 32 bias partials
  constexpr int TILE = NT * T * T, SLAB = TILE + T;
  float* R = smem;  // [2][TILE]
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    __syncthreads();
    if ((wave >> 1) == ph) {
      float* Rw = R + (wave & 1) * TILE;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cs = (r & 3) + 8 * (r >> 2) + 4 * h;
          const int idx = (t * T + cs) * T + jj;
          Rw[idx] = (ph == 0) ? acc[t][r] : Rw[idx] + acc[t][r];
        }
    }
  }
  __syncthreads();
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)SLAB;
  for (int e = tid; e < TILE; e += 256) out[e] = R[e] + R[TILE + e];
  __syncthreads();
  // bias partials: 8 parts per channel, summed in a fixed order
  R[bpart * T + bch] = bsum;
  __syncthreads();
  if (tid < T) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += R[q * T + tid];
    out[TILE + tid] = s;
  }
}

// dW[cs][cb][t] = sum_g slab[g][tile,tg][t][cs][cb]   (fixed order: 8 interleaved partial sums over g,
// combined in a fixed tree) ; dbias likewise from the 32 trailing floats of each slab.
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int G, int ny, int Cs,
                                                                int Cb, int cbT, int bias_mode) {
  constexpr int KK = KS * KS, NTG = KK / NT, T = GMK_WG_T, TILE = NT * T * T, SLAB = TILE + T;
  __shared__ float red[8][32];
  const int y = blockIdx.y;  // (tile, tg)
  const int tg = y % NTG, tile = y / NTG;
  const int cs0 = (tile / cbT) * T, cb0 = (tile % cbT) * T;
  const int el = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;  // element of the slab (incl. the bias tail)
  float sum = 0.f;
  if (e < SLAB) {
#pragma unroll 4
    for (int gI = gq; gI < G; gI += 8) sum += slab[((size_t)gI * ny + y) * (size_t)SLAB + e];
  }
  red[gq][el] = sum;
  __syncthreads();
  if (gq != 0 || e >= SLAB) return;
  sum = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
  if (e < TILE) {
    const int t = e / (T * T), cs = (e / T) % T, cb = e % T;
    if (cs0 + cs < Cs && cb0 + cb < Cb) dw[((size_t)(cs0 + cs) * Cb + cb0 + cb) * KK + tg * NT + t] = sum;
  } else if (dbias && tg == 0) {
    const int c = e - TILE;
    // bias_mode 1: channel of the small tensor (only the cbTile == 0 blocks carry it); 2: of the big tensor
    if (bias_mode == 1 && (tile % cbT) == 0 && cs0 + c < Cs) dbias[cs0 + c] = sum;
    if (bias_mode == 2 && (tile / cbT) == 0 && cb0 + c < Cb) dbias[cb0 + c] = sum;
  }
}

int gmk_launch_wgrad_k3(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);
int gmk_launch_wgrad_k5(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);

// pixels per chunk: whole rows of one image, 128 (stride 1) / 64 (stride 2: the patch is ~4x the chunk)
static inline int gmk_wgrad_patch_positions(int p, int SW, int ksize, int sb) {
  const int rows = p / SW;
  return sb == 1 ? (rows + ksize - 1) * (SW + ksize - 1) : (2 * rows + ksize - 2) * (2 * SW + ksize - 2);
}
// pixels per chunk: whole rows of one image, as many as keep the patch within 3 positions per thread
// (256 for stride 1, 128 for stride 2: the stride-2 patch is ~4x the chunk)
#define GMK_WG_MAXSLOT 3
static inline int gmk_wgrad_chunk_px(int SH, int SW, int ksize, int sb) {
  int p = sb == 1 ? 256 : 128;
  if (p > SH * SW) p = SH * SW;
  while (p > SW && gmk_wgrad_patch_positions(p, SW, ksize, sb) > GMK_WG_MAXSLOT * 256 - 1) p >>= 1;
  if (p < SW) p = SW;
  return p;
}
static inline size_t gmk_wgrad_slab_floats(int ksize) {
  const int KK = ksize * ksize, NT = ksize == 3 ? KK : ksize;
  return (size_t)NT * GMK_WG_T * GMK_WG_T + GMK_WG_T;
}

template <int KS>
static int gmk_launch_wgrad_impl(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) {
  constexpr int KK = KS * KS, NT = (KS == 3) ? KK : KS, NTG = KK / NT, T = GMK_WG_T;
  const int rows = g.P / g.SW;
  if (sb == 1) { g.PR = rows + KS - 1; g.PC = g.SW + KS - 1; }
  else { g.PR = 2 * rows + KS - 2; g.PC = 2 * g.SW + KS - 2; }
  g.npos = g.PR * g.PC;
  if (g.npos > GMK_WG_MAXSLOT * 256 - 1) return GMK_E_SHAPE;  // <= 3 patch positions per thread (+ a trash column)
  const int nslot = (g.npos + 1 + 255) / 256;
  g.PSb = (g.npos + 1) | 1;
  size_t lds = sizeof(float) * (T * (size_t)(g.P + 4) + T * (size_t)g.PSb);
  const size_t red = sizeof(float) * 2 * NT * T * T;
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return GMK_E_SHAPE;
  dim3 grid(g.G, g.csT * g.cbT * NTG);
#define GMK_WG_LAUNCH(SB_, PX_, NSLOT_)                                                                            \
  do {                                                                                                             \
    auto kern = conv_wgrad_kernel<KS, SB_, NT, PX_, NSLOT_>;                                                       \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                          \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);                                                          \
  } while (0)
#define GMK_WG_SLOTS(SB_, PX_)                                                     \
  do {                                                                             \
    if (nslot == 1) GMK_WG_LAUNCH(SB_, PX_, 1);                                    \
    else if (nslot == 2) GMK_WG_LAUNCH(SB_, PX_, 2);                               \
    else GMK_WG_LAUNCH(SB_, PX_, 3);                                               \
  } while (0)
  if (sb == 1) {
    if (g.P == 256) GMK_WG_SLOTS(1, 256); else if (g.P == 128) GMK_WG_SLOTS(1, 128); else if (g.P == 64) GMK_WG_SLOTS(1, 64);
    else if (g.P == 32) GMK_WG_SLOTS(1, 32);
    else return GMK_E_SHAPE;
  } else {
    if (g.P == 128) GMK_WG_SLOTS(2, 128); else if (g.P == 64) GMK_WG_SLOTS(2, 64); else if (g.P == 32) GMK_WG_SLOTS(2, 32);
    else return GMK_E_SHAPE;
  }
#undef GMK_WG_SLOTS
#undef GMK_WG_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  constexpr int SLAB = NT * T * T + T;
  hipLaunchKernelGGL((conv_wgrad_reduce_kernel<KS, NT>), dim3((SLAB + 31) / 32, grid.y), dim3(256), 0, s, g.slab, dw,
                     dbias, g.G, (int)grid.y, g.Cs, g.Cb, g.cbT, g.bias_mode);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

// conv_wgrad_thin.h -- weight gradient of a stride-1 conv whose input OR output has very few channels
// (enc.conv1: 3 -> C, dec.conv3: C -> 3).  The generic wgrad pads the thin side to 64 channels (21x
// wasted MFMA work); here the thin side is expanded to (channel, tap) columns instead:
//
//   out[wc][tc][t] = sum_{n,y,x} wide[n][wc][y][x] * thin[n][tc][y + sgn (ky-PAD)][x + sgn (kx-PAD)]
//
//   Cin thin  (sgn = +1): wide = dY, thin = X  -> dW[wc][tc][t]          (OIHW);  dbias = sum wide ("ones" column)
//   Cout thin (sgn = -1): wide = X,  thin = dY -> dW[tc][wc][t]          (OIHW);  dbias = sum thin (interior)
//
// GEMM view: M = 64 wide channels (A operand), N = Ct*k*k (+1) columns padded to 32*NB (B operand, read
// from the zero-padded thin patch in LDS at a per-lane tap offset), K = pixels.  The kernel is HBM-bound
// (one pass over the wide tensor): 512-thread blocks each take GMK_THIN_CPB consecutive 256-pixel chunks, prefetch all of them into
// registers and pass them through one LDS tile, each of the 8 waves reduces 32 pixels of the chunk, partials are combined in LDS in a fixed
// order and one slab per block is written; a second kernel sums the slabs (deterministic, no atomics).
#pragma once
#include "gmk_common.h"

struct ThinGeom {
  const float* wide; const float* thin; float* slab;
  int N, Cw, Ct, H, W, lgW, lgHW;
  long long total_chunks;
  int G, nWT;
  int sgn;
  int PR, PC, PS, npos;  // thin patch: rows, cols (with halo), plane stride, Ct*PS positions
  int ncol;              // Ct*k*k real columns
  int ones_col;          // column `ncol` multiplies by 1 (bias gradient over the wide tensor)
  int thin_bias;         // sum the thin tensor's interior per channel (bias gradient over the thin tensor)
};

#define GMK_THIN_PX 256
#define GMK_THIN_SS 260  /* LDS row stride of the wide tile (floats): 16-byte rows */
#define GMK_THIN_THREADS 512
#define GMK_THIN_TSLOT 3
#ifndef GMK_THIN_CPB
#define GMK_THIN_CPB 2  /* chunks per block (measured: 1 -> 35.5 us, 2 -> 29.6, 4 -> 31.6, 8 -> 58 per launch incl. reduce) */
#endif

template <int KS, int NB>
__global__ __launch_bounds__(GMK_THIN_THREADS) void conv_wgrad_thin_kernel(const ThinGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2, NCB = 32 * NB;
  constexpr int PX = GMK_THIN_PX, SS = GMK_THIN_SS, NT = GMK_THIN_THREADS;
  constexpr int NS = (64 * PX / 4) / NT;  // float4 loads of the wide tile per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Wide tile [64][SS] with 16-byte rows (ds_write_b128) and the column index XOR-swizzled by (row>>3)&3 so
  // that the 32 rows an MFMA A read touches fall into 32 different banks; thin patch behind it.
  float* Wl = smem;
  float* Tl = smem + 64 * SS;
  const int PS = g.PS, PC = g.PC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, jj = lane & 31;
  const int wt = blockIdx.y, wc0 = wt * 64;
  const int HW = 1 << g.lgHW;
  const int cw = (g.Cw - wc0) < 64 ? (g.Cw - wc0) : 64;

  // per-lane B column: (thin channel, tap) -> offset into the patch; columns beyond ncol are constants
  int toff[NB];
  float bconst[NB];
  bool bvalid[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 32 + jj;
    bvalid[nb] = col < g.ncol;
    bconst[nb] = (g.ones_col && col == g.ncol) ? 1.f : 0.f;
    const int tc = col / KK, t = col % KK;
    const int dy = g.sgn * (t / KS - PAD), dx = g.sgn * (t % KS - PAD);
    toff[nb] = bvalid[nb] ? tc * PS + (PAD + dy) * PC + (PAD + dx) : 0;
  }
  // thin patch slots of this thread
  int t_loff[GMK_THIN_TSLOT], t_goff[GMK_THIN_TSLOT], t_prow[GMK_THIN_TSLOT], t_pcol[GMK_THIN_TSLOT];
#pragma unroll
  for (int k = 0; k < GMK_THIN_TSLOT; ++k) {
    const int e = tid + k * NT;
    t_loff[k] = e < g.npos ? e : g.npos;  // masked slots land in the trash word behind the patch
    const int tc = e / PS, r = e % PS;
    t_prow[k] = r / PC - PAD;
    t_pcol[k] = r % PC - PAD;
    t_goff[k] = tc * HW;
  }

  f32x16 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][nb][r] = 0.f;
  float bsum = 0.f;

  // A block owns GMK_THIN_CPB consecutive chunks and issues ALL of their loads up front into registers
  // (the whole kernel is only ~4 HBM latencies long); the MFMAs then run as the data lands.  Straight-line
  // code on purpose: inside a loop the compiler cannot count the outstanding loads and falls back to
  // s_waitcnt vmcnt(0), draining the whole prefetch queue at every chunk.
  constexpr int DEPTH = GMK_THIN_CPB;
  struct Stage { float4 w[NS]; float t[GMK_THIN_TSLOT]; };
  Stage R[DEPTH];
  // Loads and LDS stores are unconditional and branch-free (a chunk index past the end re-reads the last
  // chunk, a masked patch slot reads offset GMK_OOB -> 0 and lands in a trash word): with any of them under
  // a branch the compiler can no longer count outstanding loads and falls back to s_waitcnt vmcnt(0),
  // which drains the whole prefetch queue at every chunk.
  const long long last_chunk = g.total_chunks - 1;
  auto load_chunk = [&](Stage& st, long long ch) {
    ch = ch < last_chunk ? ch : last_chunk;
    const long long pix = ch * PX;
    const long long n = pix >> g.lgHW;
    const int pin = (int)(pix & (HW - 1));
    const int r0 = pin >> g.lgW;
    const float* wbase = g.wide + ((n * g.Cw + wc0) << g.lgHW) + pin;
    const __amdgpu_buffer_rsrc_t rt = gmk_make_rsrc(g.thin + ((n * g.Ct) << g.lgHW), (long long)g.Ct * HW * 4);
#pragma unroll
    for (int k = 0; k < GMK_THIN_TSLOT; ++k) {
      const int iy = r0 + t_prow[k], ix = t_pcol[k];
      const bool ok = (t_loff[k] < g.npos) & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
      st.t[k] = gmk_buffer_load(rt, ok ? (t_goff[k] + iy * g.W + ix) * 4 : GMK_OOB, 0);
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int c = wave + u * (NT / 64);  // one 1-KiB channel row per wave instruction
      const int cc = c < cw ? c : cw - 1;  // ragged wide tile: re-read the last valid channel, zeroed in store_chunk
      st.w[u] = *(const float4*)(wbase + ((size_t)cc << g.lgHW) + lane * 4);
    }
  };
  auto store_chunk = [&](const Stage& st) {
#pragma unroll
    for (int k = 0; k < GMK_THIN_TSLOT; ++k) Tl[t_loff[k]] = st.t[k];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int c = wave + u * (NT / 64);
      const int sw = (c >> 3) & 3;  // wave-uniform swizzle: element k of the quad goes to slot k ^ sw
      float4 v = st.w[u];
      if (c >= cw) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sw & 1) v = make_float4(v.y, v.x, v.w, v.z);
      if (sw & 2) v = make_float4(v.z, v.w, v.x, v.y);
      *(float4*)(Wl + c * SS + lane * 4) = v;
    }
  };

  const int asw = (jj >> 3) & 3;  // rows jj and 32+jj share the swizzle
  const long long ch0 = (long long)blockIdx.x * DEPTH;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_chunk(R[d], ch0 + d);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    gmk_lds_barrier();  // previous chunk's LDS reads are done (global loads stay in flight)
    store_chunk(R[d]);
    gmk_lds_barrier();
    if (ch0 + d > last_chunk) continue;  // uniform: the clamped re-read of the last chunk is not accumulated
#pragma unroll 4
    for (int s = 0; s < PX / 16; ++s) {
      const int p = wave * (PX / 8) + 2 * s + h;
      const int pa = p ^ asw;
      const float a0 = Wl[jj * SS + pa], a1 = Wl[(32 + jj) * SS + pa];
      const int pixoff = (p >> g.lgW) * PC + (p & (g.W - 1));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float b = Tl[toff[nb] + (bvalid[nb] ? pixoff : 0)];
        b = bvalid[nb] ? b : bconst[nb];
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][nb], 0, 0, 0);
      }
    }
    if (g.thin_bias && wave < g.Ct) {
#pragma unroll
      for (int k = 0; k < PX / 64; ++k) {
        const int p = lane + 64 * k;
        bsum += Tl[wave * PS + (PAD + (p >> g.lgW)) * PC + PAD + (p & (g.W - 1))];
      }
    }
  }

  // combine the 8 waves' partials in LDS in a fixed order (deterministic): waves 0-3 store, waves 4-7 add onto
  // them, the slab write sums the four regions as (R0+R1)+(R2+R3); one slab per block.
  constexpr int RS = 64 * NCB;
  float* Red = smem;
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    __syncthreads();
    if ((wave >> 2) == ph) {
      float* R = Red + (wave & 3) * RS;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int idx = row * NCB + nb * 32 + jj;
            R[idx] = (ph == 0) ? acc[i][nb][r] : R[idx] + acc[i][nb][r];
          }
    }
  }
  __syncthreads();
  constexpr int SLAB = 64 * NCB + 64;
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + wt) * (size_t)SLAB;
  for (int e = tid; e < RS; e += NT) out[e] = (Red[e] + Red[RS + e]) + (Red[2 * RS + e] + Red[3 * RS + e]);
  const float bs = gmk_wave_sumf(bsum);
  if (lane == 0) out[RS + wave] = bs;
}

// dW / dbias from the slabs: fixed-order sum over the G blocks (32 interleaved partial sums per element,
// combined by a fixed shuffle-free LDS tree).
template <int KS, int NB>
__global__ __launch_bounds__(1024) void conv_wgrad_thin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                      float* __restrict__ dbias, const ThinGeom g) {
  constexpr int KK = KS * KS, NCB = 32 * NB, SLAB = 64 * NCB + 64;
  __shared__ float red[32][33];
  const int wt = blockIdx.y;
  const int el = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float sum = 0.f;
  if (e < SLAB) {
#pragma unroll 8
    for (int gi = gq; gi < g.G; gi += 32) sum += slab[((size_t)gi * g.nWT + wt) * (size_t)SLAB + e];
  }
  red[gq][el] = sum;
  __syncthreads();
#pragma unroll
  for (int st = 16; st > 0; st >>= 1) {
    if (gq < st) red[gq][el] += red[gq + st][el];
    __syncthreads();
  }
  if (gq != 0 || e >= SLAB) return;
  sum = red[0][el];
  if (e < 64 * NCB) {
    const int wc = wt * 64 + e / NCB, col = e % NCB;
    if (wc >= g.Cw) return;
    if (col < g.ncol) {
      const int tc = col / KK, t = col % KK;
      if (g.sgn > 0) dw[((size_t)wc * g.Ct + tc) * KK + t] = sum;
      else dw[((size_t)tc * g.Cw + wc) * KK + t] = sum;
    } else if (col == g.ncol && g.ones_col && dbias) {
      dbias[wc] = sum;
    }
  } else if (g.thin_bias && dbias && wt == 0) {
    const int c = e - 64 * NCB;
    if (c < g.Ct) dbias[c] = sum;
  }
}

// eligibility + plan (shared by the workspace query and the launcher)
static inline bool gmk_thin_plan(ThinGeom& g, int N, int Cw, int Ct, int H, int W, int ksize, int sgn, bool want_bias) {
  const int KK = ksize * ksize;
  if (Ct > 8 || Ct * KK + 1 > 96 || Cw <= Ct) return false;
  if (!gmk_is_pow2(H) || !gmk_is_pow2(W) || (long long)H * W < GMK_THIN_PX || W > GMK_THIN_PX) return false;
  g.N = N; g.Cw = Cw; g.Ct = Ct; g.H = H; g.W = W; g.lgW = gmk_ilog2(W); g.lgHW = gmk_ilog2((int64_t)H * W);
  g.total_chunks = ((long long)N * H * W) / GMK_THIN_PX;
  g.nWT = (Cw + 63) / 64;
  g.G = (int)((g.total_chunks + GMK_THIN_CPB - 1) / GMK_THIN_CPB);
  g.sgn = sgn;
  g.PR = GMK_THIN_PX / W + ksize - 1; g.PC = W + ksize - 1; g.PS = g.PR * g.PC; g.npos = Ct * g.PS;
  if (g.npos > GMK_THIN_TSLOT * GMK_THIN_THREADS) return false;
  g.ncol = Ct * KK;
  g.ones_col = (sgn > 0 && want_bias) ? 1 : 0;
  g.thin_bias = (sgn < 0 && want_bias) ? 1 : 0;
  return true;
}
static inline int gmk_thin_nb(const ThinGeom& g) { return (g.ncol + 1 + 31) / 32; }
static inline size_t gmk_thin_slab_bytes(const ThinGeom& g) {
  return sizeof(float) * (size_t)g.G * g.nWT * (64 * 32 * gmk_thin_nb(g) + 64);
}

int gmk_launch_wgrad_thin(ThinGeom& g, int ksize, float* dw, float* dbias, hipStream_t s);

// gemm.hip -- fp32 MFMA (32x32x2) strided GEMM for the two fully connected layers
// (nn.Linear fwd / dgrad / wgrad, /root/reference/models/autoencoders.py:41,57,81,95).
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = A[i*sai + k*sak],  B(k,j) = B[k*sbk + j*sbj]
// 64x64 block tile, 4 waves (2x2) of one 32x32 accumulator, K-chunk 32 staged in LDS.
// Small-M shapes (M = batch) are split along K into fp32 slabs and reduced in a fixed order.
#include "gmk_common.h"
#include "plan.h"
#include <stdlib.h>

#define GT 64
#define GK 32
#define AS_STRIDE 33
#define BS_STRIDE 65

enum { GEPI_NONE = 0, GEPI_BIAS_ACT = 1, GEPI_MUL = 2, GEPI_REPARAM_FWD = 3, GEPI_REPARAM_BWD = 4 };

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* mul; float* dact;
  const float* cs_src; float* cs_out; int cs_M, cs_N;
  // VAE tail fused into the split-K reduction (GEPI_REPARAM_*): eps (M, L); FWD: C = stats (M, 2L), z (M, L) out;
  // BWD: the GEMM result is dz (M, L), stats (M, 2L) in, C = dstats (M, 2L) out, kscale = beta / B
  const float* eps; float* z; const float* stats; float kscale; int msep;   // column sums riding along: out[j] = sum_i src[i*cs_N + j] (bias gradient)
  int M, N, K;
  long long sai, sak, sbk, sbj;
  int ldc, epi, act;
  int kper;      // K range per split
  float* slab;   // != nullptr: write raw partials slab[z][M][N]
  int remap;     // lean kernel: XCD-aware tile order
};

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, int i, int j, float v) {
  const size_t idx = (size_t)i * g.ldc + j;
  if (g.epi == GEPI_BIAS_ACT) {
    if (g.bias) v += g.bias[j];
    float a, d;
    gmk_act_fwd(g.act, v, a, d);
    g.C[idx] = a;
    if (g.dact) g.dact[idx] = d;
  } else if (g.epi == GEPI_MUL) {
    if (g.mul) v *= g.mul[idx];
    g.C[idx] = v;
  } else {
    g.C[idx] = v;
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ float As[GT * AS_STRIDE];
  __shared__ float Bs[GK * BS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const int kbeg = blockIdx.z * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool a_kfast = (g.sak == 1);
  const bool b_jfast = (g.sbj == 1);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (GT * GK) / 256; ++u) {
      const int e = tid + u * 256;
      int i, k;
      if (a_kfast) { i = e / GK; k = e % GK; } else { k = e / GT; i = e % GT; }
      float v = 0.f;
      if (i0 + i < g.M && k0 + k < kend) v = g.A[(long long)(i0 + i) * g.sai + (long long)(k0 + k) * g.sak];
      As[i * AS_STRIDE + k] = v;
    }
#pragma unroll
    for (int u = 0; u < (GT * GK) / 256; ++u) {
      const int e = tid + u * 256;
      int j, k;
      if (b_jfast) { k = e / GT; j = e % GT; } else { j = e / GK; k = e % GK; }
      float v = 0.f;
      if (j0 + j < g.N && k0 + k < kend) v = g.B[(long long)(k0 + k) * g.sbk + (long long)(j0 + j) * g.sbj];
      Bs[k * BS_STRIDE + j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float a = As[(wr * 32 + l31) * AS_STRIDE + 2 * s + h];
      const float b = Bs[(2 * s + h) * BS_STRIDE + wc * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int j = j0 + wc * 32 + l31;
  if (j >= g.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i >= g.M) continue;
    if (g.slab) g.slab[((size_t)blockIdx.z * g.M + i) * g.N + j] = acc[r];
    else gemm_epilogue(g, i, j, acc[r]);
  }
}

// ---- pipelined variant: float4 global loads along each operand's contiguous axis, next K-chunk prefetched in
// registers, double-buffered K-major LDS tiles (one LDS-only barrier per chunk, MFMA operand reads conflict-free).
// Requires the contiguous extents to be multiples of 4 and 16-byte aligned bases (host checks); rows / K beyond the
// problem are clamped for the load and zeroed on the way into LDS, so the loop is branch-free.
#define GLD 68  // LDS row stride (floats) of a [GK][64] tile: 16-byte rows for ds_write_b128
#define GLK (GK + 4)  // row stride of the lean kernel's row-major [64][GK] tiles: 16-byte rows, conflict-free ds_read_b128
// a_kfast / b_kfast: the operand is contiguous along k.  (A variant that issued ALL loads of a block's 128 / 256 deep K
// range up front was measured and dropped: 128+ VGPRs halve the blocks per CU and the dgrad shapes got 30-50 % slower.)
// Bias gradient of nn.Linear (column sums of dy) as extra blocks of the weight-gradient launch: block `cb` sums 64
// columns, 4 row groups x 8 independent partial sums per thread, combined in a fixed order.
__device__ __forceinline__ void gemm_colsum_block(const GemmArgs& g, int cb, float* red /* [4][64] */) {
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = cb * 64 + c;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (j < g.cs_N) {
    int i = q;
    for (; i + 4 * 31 < g.cs_M; i += 4 * 32) {   // 32 loads in flight per thread: a chain of dependent loads is all this block is
      float v[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) v[t] = g.cs_src[(size_t)(i + 4 * t) * g.cs_N + j];
#pragma unroll
      for (int t = 0; t < 32; ++t) s[t & 7] += v[t];
    }
    for (; i < g.cs_M; i += 4) s[0] += g.cs_src[(size_t)i * g.cs_N + j];
  }
  red[q * 64 + c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (q == 0 && j < g.cs_N) g.cs_out[j] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
}

template <bool a_kfast, bool b_kfast>
__global__ __launch_bounds__(256) void gemm_f32_pipe_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][GK * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid: [column-sum blocks][tiles: x fastest, then y, then the K split].  (As extra columns of a 3-D grid the
  // column-sum blocks came with ~4,000 empty workgroups that the real tiles queued behind: 20.8 -> 12.4 us for the
  // decoder fc's weight gradient.)
  const int ncs = g.cs_src ? (g.cs_N + 63) / 64 : 0;
  int bid = blockIdx.x;
  if (bid < ncs) {   // block-uniform branch
    gemm_colsum_block(g, bid, &smem[0][0][0]);
    return;
  }
  bid -= ncs;
  const int ntx = (g.N + GT - 1) / GT, nty = (g.M + GT - 1) / GT;
  const int bx = bid % ntx, by = (bid / ntx) % nty, bz = bid / (ntx * nty);
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int i0 = by * GT, j0 = bx * GT;
  const int kbeg = bz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  // a_kfast / b_kfast (operand contiguous along k) are template parameters: as runtime flags they put every load under
  // a branch, and the compiler then waits for vmcnt(0) after each one
  // per-thread quads: k-fast operand -> 8 quads per row (row = t/8 + 32u); row-fast operand -> 16 quads per k (k = t/16 + 16u)
  const float* abase[2]; const float* bbase[2];
  int a_k[2], b_k[2];      // first k (within the chunk) of the quad
  int a_l[2], b_l[2];      // LDS offset of the quad's first element
  bool a_ok[2], b_ok[2];   // row inside the matrix
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (a_kfast) {
      const int r = tid / 8 + 32 * u, kq = tid % 8;
      const int gi = i0 + r;
      a_ok[u] = gi < g.M;
      abase[u] = g.A + (long long)(a_ok[u] ? gi : g.M - 1) * g.sai;
      a_k[u] = kq * 4; a_l[u] = kq * 4 * GLD + r;
    } else {
      const int k = tid / 16 + 16 * u, q = tid % 16;
      const int gi = i0 + q * 4;
      a_ok[u] = gi < g.M;
      abase[u] = g.A + (a_ok[u] ? gi : 0);
      a_k[u] = k; a_l[u] = k * GLD + q * 4;
    }
    if (b_kfast) {
      const int r = tid / 8 + 32 * u, kq = tid % 8;
      const int gj = j0 + r;
      b_ok[u] = gj < g.N;
      bbase[u] = g.B + (long long)(b_ok[u] ? gj : g.N - 1) * g.sbj;
      b_k[u] = kq * 4; b_l[u] = kq * 4 * GLD + r;
    } else {
      const int k = tid / 16 + 16 * u, q = tid % 16;
      const int gj = j0 + q * 4;
      b_ok[u] = gj < g.N;
      bbase[u] = g.B + (b_ok[u] ? gj : 0);
      b_k[u] = k; b_l[u] = k * GLD + q * 4;
    }
  }
  // Register prefetch ring, GPD chunks deep.  One chunk of MFMA work is ~0.5-1 us per CU, a global load under load takes
  // 1.5-2 us: with the next chunk only (round 1) every iteration of these short-K GEMMs waited out most of a load
  // (2.2 us per chunk measured); four chunks ahead the loads are back when their turn comes.  Chunks past the end of
  // the K range are clamped for the load and zeroed on their way into LDS, so the loop runs a multiple of GPD chunks
  // without a branch around any load.
  constexpr int GPD = 4;
  float4 arr[GPD][2], brr[GPD][2];
  auto zero_unless = [](bool ok, const float4& v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
  auto load_chunk_to = [&](int k0, float4* ar, float4* br) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int ka = k0 + a_k[u]; ka = ka < kend ? ka : kend - 4;  // clamped (kend % 4 == 0 for k-fast operands)
      // one unconditional load per operand (the address is selected, not the load: loads under a branch make the
      // compiler wait for vmcnt(0) at every join)
      const float* pa = abase[u] + (a_kfast ? (long long)ka : (long long)min(k0 + a_k[u], kend - 1) * g.sak);
      ar[u] = *(const float4*)pa;
      int kb = k0 + b_k[u]; kb = kb < kend ? kb : kend - 4;
      const float* pb = bbase[u] + (b_kfast ? (long long)kb : (long long)min(k0 + b_k[u], kend - 1) * g.sbk);
      br[u] = *(const float4*)pb;
    }
  };
  auto store_chunk_from = [&](int k0, int buf, const float4* ar, const float4* br) {
    float* As = smem[buf][0];
    float* Bs = smem[buf][1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // component-wise selects: written as `cond ? ar[u] : z4` the <false, false> instantiation selected between the
      // ADDRESSES of the two (z4 parked in scratch) and read the winner back with a flat load inside the K loop
      const float4 a = zero_unless(a_ok[u] && k0 + a_k[u] < kend, ar[u]);
      if (a_kfast) { float* d = As + a_l[u]; d[0] = a.x; d[GLD] = a.y; d[2 * GLD] = a.z; d[3 * GLD] = a.w; }
      else *(float4*)(As + a_l[u]) = a;
      const float4 b = zero_unless(b_ok[u] && k0 + b_k[u] < kend, br[u]);
      if (b_kfast) { float* d = Bs + b_l[u]; d[0] = b.x; d[GLD] = b.y; d[2 * GLD] = b.z; d[3 * GLD] = b.w; }
      else *(float4*)(Bs + b_l[u]) = b;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mfma_chunk = [&](int buf) {
    const float* As = smem[buf][0];
    const float* Bs = smem[buf][1];
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float a = As[(2 * s + h) * GLD + wr * 32 + l31];
      const float b = Bs[(2 * s + h) * GLD + wc * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  };
  {
    const int nch = (kend - kbeg + GK - 1) / GK;
#pragma unroll
    for (int u = 0; u < GPD; ++u) load_chunk_to(kbeg + u * GK, arr[u], brr[u]);
    store_chunk_from(kbeg, 0, arr[0], brr[0]);
    load_chunk_to(kbeg + GPD * GK, arr[0], brr[0]);
    for (int i = 0; i < nch; i += GPD) {
#pragma unroll
      for (int u = 0; u < GPD; ++u) {   // chunk c = i + u lives in LDS tile u & 1 (GPD is even), chunk c + 1 in ring slot (u + 1) % GPD
        const int k1 = kbeg + (i + u + 1) * GK;
        gmk_lds_barrier();  // tile u & 1 is complete; the other tile's readers (previous chunk) are done
        mfma_chunk(u & 1);
        store_chunk_from(k1, (u + 1) & 1, arr[(u + 1) % GPD], brr[(u + 1) % GPD]);
        load_chunk_to(k1 + GPD * GK, arr[(u + 1) % GPD], brr[(u + 1) % GPD]);
      }
    }
  }
  const int j = j0 + wc * 32 + l31;
  if (j >= g.N) return;
  if (g.slab) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (i < g.M) g.slab[((size_t)bz * g.M + i) * g.N + j] = acc[r];
    }
    return;
  }
  // direct epilogue: the per-element factors (act' of dgrad) are all fetched before the math, the bias once per lane
  float mulv[16];
  const float bv = (g.epi == GEPI_BIAS_ACT && g.bias) ? g.bias[j] : 0.f;
  if (g.epi == GEPI_MUL && g.mul) {  // one uniform branch around the whole batch of loads
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ic = i < g.M ? i : g.M - 1;
      mulv[r] = g.mul[(size_t)ic * g.ldc + j];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) mulv[r] = 1.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i >= g.M) continue;
    const size_t idx = (size_t)i * g.ldc + j;
    if (g.epi == GEPI_BIAS_ACT) {
      float a, d;
      gmk_act_fwd(g.act, acc[r] + bv, a, d);
      g.C[idx] = a;
      if (g.dact) g.dact[idx] = d;
    } else {
      g.C[idx] = acc[r] * mulv[r];
    }
  }
}

// ---- the same kernel with a K loop free of VALU work (round 3) -------------------------------------------------------
// PMC of gemm_f32_pipe_kernel at the configs[4] FC shapes: 5-7 VALU instructions per MFMA (per-chunk 64-bit address
// arithmetic, end-of-K clamps, zero selects), and fp32 MFMA shares the VALU issue port: matrix pipe 44-58 % busy whatever
// the tile shape.  Here every per-lane address is a 32-bit byte offset computed ONCE and the chunk's position is a uniform
// base pointer (scalar arithmetic + one 64-bit vector add per load); nothing is clamped or zeroed per lane:
//   - rows / columns of a ragged edge tile re-read a valid row: they only reach accumulator rows / columns that are
//     never written (an MFMA row depends on its own A row only);
//   - the K range of every block is a whole number of chunks (host: K % 32 == 0), chunks of the prefetch ring past the
//     end re-read the last chunk and are never multiplied (uniform guard around the MFMA section).
// Chunks in the same order as gemm_f32_pipe_kernel; inside a chunk the k pairs are taken in another order (mfma_chunk
// below), so results differ from it in the last bits (same fp32 FMA chain length, fixed order: deterministic).
template <bool a_kfast, bool b_kfast>
__global__ __launch_bounds__(256) void gemm_f32_lean_kernel(const GemmArgs g) {
  constexpr int TSZ = GT * GLK > GK * GLD ? GT * GLK : GK * GLD;
  __shared__ __attribute__((aligned(16))) float smem[2][2][TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncs = g.cs_src ? (g.cs_N + 63) / 64 : 0;
  int bid = blockIdx.x;
  if (bid < ncs) {   // block-uniform branch
    gemm_colsum_block(g, bid, &smem[0][0][0]);
    return;
  }
  bid -= ncs;
  const int ntx = (g.N + GT - 1) / GT, nty = (g.M + GT - 1) / GT, ntile = ntx * nty;
  // Tile order: blocks go round-robin over the 8 XCDs (one L2 each), so XCD k takes the contiguous tile range
  // [k ntile/8, (k+1) ntile/8) of its K split, and inside it the dimension with FEWER tiles runs fastest: the blocks that
  // share the big operand's tile (the weight tile of an M = 256 layer: 4 row tiles) run back to back on one L2 (PMC at
  // configs[4]'s enc.fc dgrad before: 223 MB read from HBM for 67 MB of weights).
  const int bz = bid / ntile;
  int t = bid - bz * ntile;
  if (g.remap && (ntile & 7) == 0) t = (t & 7) * (ntile >> 3) + (t >> 3);
  int bx, by;
  if (g.remap && nty <= ntx) { by = t % nty; bx = t / nty; } else { bx = t % ntx; by = t / ntx; }
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int i0 = by * GT, j0 = bx * GT;
  const int kbeg = bz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  const int nch = (kend - kbeg) / GK;   // whole chunks (host guarantees it)
  // per-lane byte offsets from the chunk's uniform base, LDS offsets of the quads
  unsigned a_bo[2], b_bo[2];
  int a_l[2], b_l[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (a_kfast) {
      const int r = tid / 8 + 32 * u, kq = tid % 8;
      const int gi = min(i0 + r, g.M - 1);
      a_bo[u] = (unsigned)(((long long)gi * g.sai + kq * 4) * 4);
      a_l[u] = r * GLK + kq * 4;
    } else {
      const int k = tid / 16 + 16 * u, q = tid % 16;
      const int gi = min(i0 + q * 4, g.M - 4);
      a_bo[u] = (unsigned)(((long long)k * g.sak + gi) * 4);
      a_l[u] = k * GLD + q * 4;
    }
    if (b_kfast) {
      const int r = tid / 8 + 32 * u, kq = tid % 8;
      const int gj = min(j0 + r, g.N - 1);
      b_bo[u] = (unsigned)(((long long)gj * g.sbj + kq * 4) * 4);
      b_l[u] = r * GLK + kq * 4;
    } else {
      const int k = tid / 16 + 16 * u, q = tid % 16;
      const int gj = min(j0 + q * 4, g.N - 4);
      b_bo[u] = (unsigned)(((long long)k * g.sbk + gj) * 4);
      b_l[u] = k * GLD + q * 4;
    }
  }
  constexpr int GPD = 4;
  float4 arr[GPD][2], brr[GPD][2];
  const int klast = kend - GK;
  auto load_chunk_to = [&](int k0, float4* ar, float4* br) {
    const int kc = k0 < klast ? k0 : klast;   // uniform
    const char* ca = (const char*)(g.A + (a_kfast ? (long long)kc : (long long)kc * g.sak));
    const char* cb = (const char*)(g.B + (b_kfast ? (long long)kc : (long long)kc * g.sbk));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ar[u] = *(const float4*)(ca + a_bo[u]);
      br[u] = *(const float4*)(cb + b_bo[u]);
    }
  };
  auto store_chunk_from = [&](int buf, const float4* ar, const float4* br) {
    float* As = smem[buf][0];
    float* Bs = smem[buf][1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = ar[u];
      *(float4*)(As + a_l[u]) = a;   // one ds_write_b128 either way: the tile's layout follows the operand's
      const float4 b = br[u];
      *(float4*)(Bs + b_l[u]) = b;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mfma_chunk = [&](int buf) {
    const float* As = smem[buf][0];
    const float* Bs = smem[buf][1];
    // The LDS tile of an operand follows its memory layout, so that every global float4 is ONE ds_write_b128:
    //   k-contiguous operand -> row-major tile [64][GK + 4]: a lane reads 4 consecutive k of its row at once (ds_read_b128,
    //     1 LDS cycle per MFMA and operand);
    //   row-contiguous operand -> k-major tile [GK][68]: one ds_read_b32 per MFMA.
    // (Both operands k-major, as in gemm_f32_pipe_kernel, cost 8 LDS cycles per MFMA -- the compiler pairs the reads into
    // ds_read2_b32 -- which with 12 waves per CU keeps the LDS busy for half of every MFMA slot; PMC: matrix pipe 55-58 %.)
    // Which k sits at which MFMA k index is free as long as A and B agree: lane half h takes k = 8m + 4h + j in step (m, j).
    const float* ap = a_kfast ? As + (wr * 32 + l31) * GLK + 4 * h : As + 4 * h * GLD + wr * 32 + l31;
    const float* bp = b_kfast ? Bs + (wc * 32 + l31) * GLK + 4 * h : Bs + 4 * h * GLD + wc * 32 + l31;
#pragma unroll
    for (int m = 0; m < GK / 8; ++m) {
      float4 a, b;
      if (a_kfast) a = *(const float4*)(ap + 8 * m);
      else a = make_float4(ap[8 * m * GLD], ap[(8 * m + 1) * GLD], ap[(8 * m + 2) * GLD], ap[(8 * m + 3) * GLD]);
      if (b_kfast) b = *(const float4*)(bp + 8 * m);
      else b = make_float4(bp[8 * m * GLD], bp[(8 * m + 1) * GLD], bp[(8 * m + 2) * GLD], bp[(8 * m + 3) * GLD]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
#pragma unroll
  for (int u = 0; u < GPD; ++u) load_chunk_to(kbeg + u * GK, arr[u], brr[u]);
  store_chunk_from(0, arr[0], brr[0]);
  load_chunk_to(kbeg + GPD * GK, arr[0], brr[0]);
  for (int i = 0; i < nch; i += GPD) {
#pragma unroll
    for (int u = 0; u < GPD; ++u) {
      gmk_lds_barrier();
      if (i + u < nch) mfma_chunk(u & 1);   // uniform
      store_chunk_from((u + 1) & 1, arr[(u + 1) % GPD], brr[(u + 1) % GPD]);
      load_chunk_to(kbeg + (i + u + 1 + GPD) * GK, arr[(u + 1) % GPD], brr[(u + 1) % GPD]);
    }
  }
  const int j = j0 + wc * 32 + l31;
  if (j >= g.N) return;
  if (g.slab) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (i < g.M) g.slab[((size_t)bz * g.M + i) * g.N + j] = acc[r];
    }
    return;
  }
  float mulv[16];
  const float bv = (g.epi == GEPI_BIAS_ACT && g.bias) ? g.bias[j] : 0.f;
  if (g.epi == GEPI_MUL && g.mul) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ic = i < g.M ? i : g.M - 1;
      mulv[r] = g.mul[(size_t)ic * g.ldc + j];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) mulv[r] = 1.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i >= g.M) continue;
    const size_t idx = (size_t)i * g.ldc + j;
    if (g.epi == GEPI_BIAS_ACT) {
      float a, d;
      gmk_act_fwd(g.act, acc[r] + bv, a, d);
      g.C[idx] = a;
      if (g.dact) g.dact[idx] = d;
    } else {
      g.C[idx] = acc[r] * mulv[r];
    }
  }
}

// Reparameterisation (/root/reference/models/autoencoders.py:169-176) in the epilogue of the encoder fc's split-K
// reduction: a thread sums the adjacent columns (2l, 2l+1) = (mu, log_var), writes them to `stats` and
// z = mu + exp(log_var / 2) * eps next to them.
__global__ __launch_bounds__(256) void gemm_splitk_reparam_fwd_kernel(const GemmArgs g, int splits) {
  const size_t total = (size_t)g.M * g.N, pairs = total / 2;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
    float2 v = make_float2(0.f, 0.f);
    for (int zz = 0; zz < splits; ++zz) {
      const float2 t = *(const float2*)(g.slab + (size_t)zz * total + 2 * p);
      v.x += t.x; v.y += t.y;
    }
    const int j = (int)((2 * p) % g.N);
    if (g.bias) { v.x += g.bias[j]; v.y += g.bias[j + 1]; }
    *(float2*)(g.C + 2 * p) = v;
    g.z[p] = v.x + expf(0.5f * v.y) * g.eps[p];
  }
}
// ... and its backward, with the KL (or --mse prior) gradient of /root/reference/main.py:39-45, in the epilogue of the
// decoder fc's dgrad reduction: dz = sum of the slabs;
//   dstats[.,0] = dz + kscale * dKL/dmu,  dstats[.,1] = dz * eps * sigma / 2 + kscale * dKL/dlog_var   (kscale = beta / B)
__device__ __forceinline__ float2 gmk_reparam_bwd_pair(float dz, float2 st, float eps, float kscale, int msep) {
  const float sg = expf(0.5f * st.y);
  float2 d;
  d.x = dz;
  d.y = dz * eps * 0.5f * sg;
  if (msep) { d.x += kscale * 2.f * st.x; d.y += kscale * (sg - 1.f) * sg; }
  else { d.x += kscale * st.x; d.y += kscale * 0.5f * (expf(st.y) - 1.f); }
  return d;
}
__global__ __launch_bounds__(256) void gemm_splitk_reparam_bwd_kernel(const GemmArgs g, int splits) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
    float dz = 0.f;
    for (int zz = 0; zz < splits; ++zz) dz += g.slab[(size_t)zz * total + p];
    const float2 st = *(const float2*)(g.stats + 2 * p);
    *(float2*)(g.C + 2 * p) = gmk_reparam_bwd_pair(dz, st, g.eps[p], g.kscale, g.msep);
  }
}
// unsplit fallbacks (shapes whose GEMM needs no split: not the VAE's): elementwise passes over the GEMM's plain output
__global__ __launch_bounds__(256) void reparam_fwd_from_stats_kernel(const GemmArgs g) {
  const size_t pairs = (size_t)g.M * g.N / 2;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
    const float2 v = *(const float2*)(g.C + 2 * p);
    g.z[p] = v.x + expf(0.5f * v.y) * g.eps[p];
  }
}
__global__ __launch_bounds__(256) void reparam_bwd_from_dz_kernel(const GemmArgs g, const float* dz) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256)
    *(float2*)(g.C + 2 * p) = gmk_reparam_bwd_pair(dz[p], *(const float2*)(g.stats + 2 * p), g.eps[p], g.kscale, g.msep);
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmArgs g, int splits) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += g.slab[(size_t)z * total + e];
    gemm_epilogue(g, (int)(e / g.N), (int)(e % g.N), v);
  }
}

// column sums: out[j] = sum_i t[i*N + j]  (bias gradient); 64 columns x 4 row groups per block, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ t, float* __restrict__ out, int M, int N) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float s0 = 0.f, s1 = 0.f;
  if (j < N) {
    int i = q;
    for (; i + 4 < M; i += 8) { s0 += t[(size_t)i * N + j]; s1 += t[(size_t)(i + 4) * N + j]; }
    for (; i < M; i += 4) s0 += t[(size_t)i * N + j];
  }
  red[q][c] = s0 + s1;
  __syncthreads();
  if (q == 0 && j < N) out[j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// Split K only while the grid is smaller than one block per CU: every split costs a slab round trip (8 MB written and
// read back at these shapes) and a reduce launch, which is more than a second wave of blocks buys (round 1 split up to
// 512 blocks: the M = 256 layers with 256 tiles then ran 12-20 us + a 5-9 us reduce each).
static int pick_splits(int M, int N, int K) {
  const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  int splits = 1;
  while (tiles * splits < 256 && K / (splits * 2) >= 128 && splits < 64) splits *= 2;
  return splits;
}

extern "C" size_t gmk_linear_workspace_bytes(int M, int K, int N) {
  // the largest slab over fwd (M x N, red K), dgrad (M x K, red N), wgrad (N x K, red M)
  size_t a = (size_t)pick_splits(M, N, K) * M * N;
  size_t b = (size_t)pick_splits(M, K, N) * M * K;
  size_t c = (size_t)pick_splits(N, K, M) * N * K;
  size_t m = a > b ? a : b;
  m = m > c ? m : c;
  return m * sizeof(float) + 256;
}

static int run_gemm(GemmArgs g, void* ws, size_t ws_bytes, hipStream_t s) {
  const int splits = pick_splits(g.M, g.N, g.K);
  const int fused = g.epi;   // GEPI_REPARAM_*: applied by the reduce kernel (or a fallback pass), never by the GEMM itself
  float* const fused_out = g.C;
  if (fused == GEPI_REPARAM_FWD) { g.epi = GEPI_BIAS_ACT; g.act = GMK_ACT_NONE; g.dact = nullptr; }
  if (fused == GEPI_REPARAM_BWD) g.epi = GEPI_NONE;
  int kper = (g.K + splits - 1) / splits;
  kper = (kper + GK - 1) / GK * GK;
  g.kper = kper;
  const int zs = (g.K + kper - 1) / kper;
  if (fused == GEPI_REPARAM_BWD && zs == 1) {   // unsplit: the GEMM writes dz to the workspace, a second pass finishes
    if (!ws || ws_bytes < sizeof(float) * (size_t)g.M * g.N) return GMK_E_WORKSPACE;
    g.C = (float*)ws;
  }
  dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, zs);
  const bool want_cs = g.cs_src != nullptr;
  if (zs > 1) {
    if (!ws || ws_bytes < sizeof(float) * (size_t)zs * g.M * g.N) return GMK_E_WORKSPACE;
    g.slab = (float*)ws;
  } else {
    g.slab = nullptr;
  }
  // pipelined kernel when every float4 load is aligned: contiguous extents % 4 == 0, 16-byte bases and row pitches
  const bool a_k = g.sak == 1, b_k = g.sbk == 1;
  const bool a_fast = a_k ? (g.K % 4 == 0 && g.sai % 4 == 0) : (g.sai == 1 && g.M % 4 == 0 && g.sak % 4 == 0);
  const bool b_fast = b_k ? (g.K % 4 == 0 && g.sbj % 4 == 0) : (g.sbj == 1 && g.N % 4 == 0 && g.sbk % 4 == 0);
  bool cs_done = false;
  // the VALU-free K loop wants whole chunks, at least 4 rows / columns, and operands whose byte offsets fit 32 bits
  static const bool lean_off = [] { const char* e = getenv("GMK_GEMM_LEAN"); return e && e[0] == '0'; }();   // A/B knob
  const long long a_span = a_k ? (long long)g.M * g.sai : (long long)GK * g.sak + g.M;
  const long long b_span = b_k ? (long long)g.N * g.sbj : (long long)GK * g.sbk + g.N;
  const bool lean = !lean_off && g.K % GK == 0 && g.M >= 4 && g.N >= 4 && a_span < (1LL << 29) && b_span < (1LL << 29);
  if (a_fast && b_fast && gmk_aligned16(g.A) && gmk_aligned16(g.B) && g.K >= 4) {
    dim3 grid1(grid.x * grid.y * grid.z);
    if (want_cs) { grid1.x += (g.cs_N + 63) / 64; cs_done = true; }   // column-sum blocks ride in the same launch, first
    if (lean) {
      static const bool remap_off = [] { const char* e = getenv("GMK_GEMM_REMAP"); return e && e[0] == '0'; }();   // A/B knob
      g.remap = remap_off ? 0 : 1;
      if (a_k && b_k) hipLaunchKernelGGL((gemm_f32_lean_kernel<true, true>), grid1, dim3(256), 0, s, g);
      else if (a_k) hipLaunchKernelGGL((gemm_f32_lean_kernel<true, false>), grid1, dim3(256), 0, s, g);
      else if (b_k) hipLaunchKernelGGL((gemm_f32_lean_kernel<false, true>), grid1, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm_f32_lean_kernel<false, false>), grid1, dim3(256), 0, s, g);
    } else
    if (a_k && b_k) hipLaunchKernelGGL((gemm_f32_pipe_kernel<true, true>), grid1, dim3(256), 0, s, g);
    else if (a_k) hipLaunchKernelGGL((gemm_f32_pipe_kernel<true, false>), grid1, dim3(256), 0, s, g);
    else if (b_k) hipLaunchKernelGGL((gemm_f32_pipe_kernel<false, true>), grid1, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_pipe_kernel<false, false>), grid1, dim3(256), 0, s, g);
  } else
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (zs > 1) {
    size_t total = (size_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    if (fused == GEPI_REPARAM_FWD) hipLaunchKernelGGL(gemm_splitk_reparam_fwd_kernel, dim3(blocks), dim3(256), 0, s, g, zs);
    else if (fused == GEPI_REPARAM_BWD) { g.C = fused_out; hipLaunchKernelGGL(gemm_splitk_reparam_bwd_kernel, dim3(blocks), dim3(256), 0, s, g, zs); }
    else hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g, zs);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  } else if (fused == GEPI_REPARAM_FWD || fused == GEPI_REPARAM_BWD) {
    size_t total = (size_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    if (fused == GEPI_REPARAM_FWD) hipLaunchKernelGGL(reparam_fwd_from_stats_kernel, dim3(blocks), dim3(256), 0, s, g);
    else { const float* dz = g.C; g.C = fused_out; hipLaunchKernelGGL(reparam_bwd_from_dz_kernel, dim3(blocks), dim3(256), 0, s, g, dz); }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (want_cs && !cs_done) {   // unaligned shapes (generic kernel): separate column-sum launch
    hipLaunchKernelGGL(colsum_kernel, dim3((g.cs_N + 63) / 64), dim3(256), 0, s, g.cs_src, g.cs_out, g.cs_M, g.cs_N);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return GMK_OK;
}

extern "C" int gmk_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* dact, int M,
                              int K, int N, int act, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_linear_fwd, x, w, bias, y, dact, M, K, N, act, ws, ws_bytes, stream);
  GMK_REQUIRE(x && w && y, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0, GMK_E_SHAPE);
  GMK_REQUIRE(act >= GMK_ACT_NONE && act <= GMK_ACT_SIGMOID, GMK_E_UNSUPPORTED);
  GemmArgs g{};
  g.A = x; g.B = w; g.C = y; g.bias = bias; g.dact = dact; g.mul = nullptr;
  g.M = M; g.N = N; g.K = K; g.sai = K; g.sak = 1; g.sbk = 1; g.sbj = K; g.ldc = N;
  g.epi = GEPI_BIAS_ACT; g.act = act;
  return run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int gmk_linear_dgrad(const float* dy, const float* w, const float* dact_in, float* dx, int M, int K,
                                int N, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_linear_dgrad, dy, w, dact_in, dx, M, K, N, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && w && dx, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0, GMK_E_SHAPE);
  GemmArgs g{};
  g.A = dy; g.B = w; g.C = dx; g.bias = nullptr; g.dact = nullptr; g.mul = dact_in;
  g.M = M; g.N = K; g.K = N; g.sai = N; g.sak = 1; g.sbk = K; g.sbj = 1; g.ldc = K;
  g.epi = GEPI_MUL; g.act = 0;
  return run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int gmk_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, int M, int K, int N,
                                void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_linear_wgrad, dy, x, dw, dbias, M, K, N, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && x && dw, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0, GMK_E_SHAPE);
  GemmArgs g{};
  g.A = dy; g.B = x; g.C = dw; g.bias = nullptr; g.dact = nullptr; g.mul = nullptr;
  g.M = N; g.N = K; g.K = M; g.sai = 1; g.sak = N; g.sbk = K; g.sbj = 1; g.ldc = K;
  g.epi = GEPI_NONE; g.act = 0;
  if (dbias) { g.cs_src = dy; g.cs_out = dbias; g.cs_M = M; g.cs_N = N; }
  return run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
}

// nn.Linear forward of the VAE encoder head fused with the reparameterisation (the bias add and z = mu + sigma * eps
// run in the epilogue of the split-K reduction): stats (M, 2L) and z (M, L) out.  N = 2L.
extern "C" int gmk_linear_fwd_reparam(const float* x, const float* w, const float* bias, const float* eps, float* stats,
                                      float* z, int M, int K, int N, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_linear_fwd_reparam, x, w, bias, eps, stats, z, M, K, N, ws, ws_bytes, stream);
  GMK_REQUIRE(x && w && eps && stats && z, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0 && N % 2 == 0, GMK_E_SHAPE);
  GemmArgs g{};
  g.A = x; g.B = w; g.C = stats; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.sai = K; g.sak = 1; g.sbk = 1; g.sbj = K; g.ldc = N;
  g.epi = GEPI_REPARAM_FWD; g.eps = eps; g.z = z;
  return run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
}

// Backward of the decoder's nn.Linear input (dz = dy @ w) fused with the backward of the reparameterisation and the
// gradient of beta * KL (or of the --mse prior): dstats (M, K, 2) out, K = latent size.  kl_scale = beta / batch.
extern "C" int gmk_linear_dgrad_reparam(const float* dy, const float* w, const float* stats, const float* eps,
                                        float* dstats, int M, int K, int N, float kl_scale, int mse_prior, void* ws,
                                        size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_linear_dgrad_reparam, dy, w, stats, eps, dstats, M, K, N, kl_scale, mse_prior, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && w && stats && eps && dstats, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0, GMK_E_SHAPE);
  GemmArgs g{};
  g.A = dy; g.B = w; g.C = dstats;
  g.M = M; g.N = K; g.K = N; g.sai = N; g.sak = 1; g.sbk = K; g.sbj = 1; g.ldc = K;
  g.epi = GEPI_REPARAM_BWD; g.eps = eps; g.stats = stats; g.kscale = kl_scale; g.msep = mse_prior ? 1 : 0;
  return run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
}

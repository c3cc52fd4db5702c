// fid.hip -- FID / inception feature statistics on the GPU (replaces the numpy / scipy tail of
// /root/reference/metrics.py:75-108 and the N x 10 reductions of :144-199, :231-254).
//
//  gmk_gram_accumulate   (n, s = sum x, G = sum x x^T) of fp32 features in fp64 on
//                        v_mfma_f64_16x16x4_f64 (an fp32 x fp32 product is exact in fp64, so the
//                        moments are independent of sharding / GPU count to ~1e-16 relative).
//  gmk_fid_from_moments  mu, Sigma from the moments; tr sqrt(S1 S2) = tr sqrt(S1^1/2 S2 S1^1/2)
//                        by two symmetric coupled Newton-Schulz square roots (fp64 MFMA GEMMs only;
//                        no Schur form), all on the device, no host sync.
//  gmk_softmax_stats     fused softmax + entropy + top-1/top-N + arg-max histogram partial sums.
#include "gmk_common.h"
#include <stdlib.h>

// =============================================================== Gram accumulation ==========
#define GR_TILE 64   // output tile (features) per block
#define GR_KC 32     // samples per LDS chunk
#define GR_STR 65    // LDS row stride (floats)

// partial[split][D][D] (upper and lower both computed), psum[split][D]
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ X, long long n, int D,
                                                           double* __restrict__ partial, double* __restrict__ psum,
                                                           int splits) {
  __shared__ float Xi[GR_KC * GR_STR];
  __shared__ float Xj[GR_KC * GR_STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = (D + GR_TILE - 1) / GR_TILE;
  const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
  const int i0 = ti * GR_TILE, j0 = tj * GR_TILE;
  const int sp = blockIdx.y;
  const long long per = ((n + splits - 1) / splits + GR_KC - 1) / GR_KC * GR_KC;
  const long long kbeg = sp * per, kend = kbeg + per < n ? kbeg + per : n;
  f64x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;  // column sum of feature j0 + tid (threads 0..63 of the ti == 0 blocks)
  const int l15 = lane & 15, l4 = lane >> 4;
  for (long long k0 = kbeg; k0 < kend; k0 += GR_KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (GR_KC * GR_TILE) / 256; ++u) {
      const int e = tid + u * 256;
      const int k = e / GR_TILE, c = e % GR_TILE;
      const bool kok = k0 + k < kend;
      Xi[k * GR_STR + c] = (kok && i0 + c < D) ? X[(k0 + k) * D + i0 + c] : 0.f;
      Xj[k * GR_STR + c] = (kok && j0 + c < D) ? X[(k0 + k) * D + j0 + c] : 0.f;
    }
    __syncthreads();
    if (ti == 0 && tid < 64) {
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < GR_KC; ++k) s += (double)Xj[k * GR_STR + tid];
      csum += s;
    }
#pragma unroll
    for (int s4 = 0; s4 < GR_KC / 4; ++s4) {
      const int k = 4 * s4 + l4;
      const double a = (double)Xi[k * GR_STR + wave * 16 + l15];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = (double)Xj[k * GR_STR + t * 16 + l15];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = partial + (size_t)sp * D * D;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 16 + l4 + 4 * r, j = j0 + t * 16 + l15;
      if (i < D && j < D) out[(size_t)i * D + j] = acc[t][r];
    }
  if (ti == 0 && tid < 64 && j0 + tid < D) psum[(size_t)sp * D + j0 + tid] = csum;
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partial, const double* __restrict__ psum,
                                                          double* __restrict__ G, double* __restrict__ s, int D, int splits) {
  const size_t total = (size_t)D * D;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total + D; e += (size_t)gridDim.x * 256) {
    double v = 0.0;
    if (e < total) {
      for (int z = 0; z < splits; ++z) v += partial[(size_t)z * total + e];
      G[e] += v;
    } else {
      const size_t j = e - total;
      for (int z = 0; z < splits; ++z) v += psum[(size_t)z * D + j];
      s[j] += v;
    }
  }
}

// ---- Gram accumulation, second version: 128 x 128 feature tiles, upper triangle only --------------------------
// The 64 x 64 kernel above computes both triangles, re-reads X once per tile pair (5x the algorithmic bytes at
// D = 256: 258 MB per 50,000 x 256 launch under rocprofv3 --pmc) and waits out every chunk's global loads
// (fp64-MFMA busy 25 %).  Here a block owns a 128 x 128 tile (ti <= tj; 3 tiles at D = 256, so X is read twice), each
// wave a 64 x 64 quadrant = 16 MFMA tiles fed by 4 + 4 operand reads per k-step; chunks of 32 samples go through
// double-buffered LDS with the next chunk prefetched in registers (one LDS-only barrier per chunk).  The lower
// triangle is mirrored by the reduce kernel.  Needs D % 4 == 0 (float4 rows); other D take the kernel above.
#define G2_T 128
#define G2_KC 32
#define G2_STR 272   // LDS row [i-tile 128 | j-tile 128 | pad 16]: 272 = 16 (mod 32), so 16 features x 2 samples hit 32 banks

__global__ __launch_bounds__(256) void gram_partial128_kernel(const float* __restrict__ X, long long n, int D,
                                                              double* __restrict__ partial, double* __restrict__ psum,
                                                              int splits, long long per) {
  __shared__ __attribute__((aligned(16))) float Xl[2][G2_KC * G2_STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // upper-triangular tile index -> (ti, tj), ti <= tj
  const int T = (D + G2_T - 1) / G2_T;
  int ti = 0, rem = blockIdx.x;
  while (rem >= T - ti) { rem -= T - ti; ++ti; }
  const int tj = ti + rem;
  const int i0 = ti * G2_T, j0 = tj * G2_T;
  const int sp = blockIdx.y;
  const long long kbeg = sp * per, kend = kbeg + per < n ? kbeg + per : n;

  // staging: 8 float4 per thread and chunk; element e = tid + u*256 -> sample k = e / 64, quad q = e % 64 of the
  // 256 staged columns (q < 32: i-tile, else j-tile)
  int s_k[8], s_col[8], s_lds[8];
  bool s_cok[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + u * 256, k = e >> 6, q = e & 63;
    const int col = (q < 32 ? i0 : j0 - 128) + 4 * q;   // global feature index of the quad
    s_k[u] = k;
    s_cok[u] = col < D;
    s_col[u] = s_cok[u] ? col : 0;
    s_lds[u] = k * G2_STR + 4 * q;
  }
  float4 pre[8];
  auto load_chunk = [&](long long k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      long long k = k0 + s_k[u];
      k = k < n ? k : n - 1;   // clamped address, zeroed on the way into LDS: no load sits under a branch
      pre[u] = *(const float4*)(X + k * D + s_col[u]);
    }
  };
  auto store_chunk = [&](long long k0, int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = s_cok[u] && (k0 + s_k[u] < kend);
      *(float4*)(&Xl[buf][s_lds[u]]) = ok ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f64x4 acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;   // column sum of feature i0 + tid (threads 0..127 of the diagonal tiles)
  const bool do_sum = (ti == tj) && tid < G2_T;

  if (kbeg < kend) {
    load_chunk(kbeg);
    store_chunk(kbeg, 0);
    load_chunk(kbeg + G2_KC);
    int buf = 0;
    for (long long k0 = kbeg; k0 < kend; k0 += G2_KC, buf ^= 1) {
      gmk_lds_barrier();                       // tile `buf` is complete; last chunk's readers of the other tile are done
      store_chunk(k0 + G2_KC, buf ^ 1);        // (past the end: zeros into a tile nobody reads)
      load_chunk(k0 + 2 * G2_KC);
      const float* Xb = Xl[buf];
      if (do_sum) {
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < G2_KC; ++k) s += (double)Xb[k * G2_STR + tid];
        csum += s;
      }
#pragma unroll
      for (int s4 = 0; s4 < G2_KC / 4; ++s4) {
        const float* row = Xb + (4 * s4 + l4) * G2_STR + l15;
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (double)row[wi * 64 + r * 16];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = (double)row[128 + wj * 64 + c * 16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[c], acc[r][c], 0, 0, 0);
      }
    }
  }
  // f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = partial + (size_t)sp * D * D;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + wi * 64 + r * 16 + l4 + 4 * q, j = j0 + wj * 64 + c * 16 + l15;
        if (i < D && j < D) out[(size_t)i * D + j] = acc[r][c][q];
      }
  if (do_sum && i0 + tid < D) psum[(size_t)sp * D + i0 + tid] = csum;
}

// G += sum_z partial[z] over the upper tiles, mirrored into the lower tiles; s += sum_z psum[z].  Fixed order.
__global__ __launch_bounds__(256) void gram_reduce128_kernel(const double* __restrict__ partial, const double* __restrict__ psum,
                                                             double* __restrict__ G, double* __restrict__ s, int D, int splits) {
  const size_t total = (size_t)D * D;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total + D; e += (size_t)gridDim.x * 256) {
    double v = 0.0;
    if (e < total) {
      const int i = (int)(e / D), j = (int)(e % D);
      if (i / G2_T > j / G2_T) continue;   // lower tiles are written by their mirror element
      for (int z = 0; z < splits; ++z) v += partial[(size_t)z * total + e];
      G[e] += v;
      if (i / G2_T < j / G2_T) G[(size_t)j * D + i] += v;
    } else {
      const size_t j = e - total;
      for (int z = 0; z < splits; ++z) v += psum[(size_t)z * D + j];
      s[j] += v;
    }
  }
}

static int gram_splits(long long n) {
  long long s = n / 256;
  if (s < 1) s = 1;
  if (s > 32) s = 32;
  return (int)s;
}
// splits and samples per split of the 128-tile kernel: about one block per CU, at least two chunks per block
static void gram128_plan(long long n, int D, int* splits, long long* per) {
  const int T = (D + G2_T - 1) / G2_T, tiles = T * (T + 1) / 2;
  long long s = 256 / tiles;
  if (s < 1) s = 1;
  if (s > n / 64) s = n / 64;
  if (s < 1) s = 1;
  long long p = ((n + s - 1) / s + G2_KC - 1) / G2_KC * G2_KC;
  *per = p;
  *splits = (int)((n + p - 1) / p);
}

extern "C" size_t gmk_gram_workspace_bytes(int64_t n, int D) {
  // the larger of the two kernels' plans: gmk_gram_accumulate falls back to the 64-tile kernel (gram_splits slabs) when X
  // is not 16-byte aligned or GMK_GRAM64 is set, and for D >= 512 that is more slabs than the 128-tile plan has
  int splits = gram_splits(n);
  if (D % 4 == 0) {
    int s128; long long per;
    gram128_plan(n, D, &s128, &per);
    if (s128 > splits) splits = s128;
  }
  return sizeof(double) * (size_t)splits * ((size_t)D * D + D) + 256;
}

extern "C" int gmk_gram_accumulate(const float* X, int64_t n, int D, double* G, double* s, void* ws, size_t ws_bytes,
                                   gmk_stream_t stream) {
  GMK_REQUIRE(X && G && s && ws, GMK_E_NULL);
  GMK_REQUIRE(n > 0 && D > 0 && D <= 4096, GMK_E_SHAPE);
  GMK_REQUIRE(ws_bytes >= gmk_gram_workspace_bytes(n, D) - 256, GMK_E_WORKSPACE);
  hipStream_t st = (hipStream_t)stream;
  size_t total = (size_t)D * D + D;
  int blocks = (int)((total + 255) / 256); if (blocks > 1024) blocks = 1024;
  static const int old = getenv("GMK_GRAM64") ? atoi(getenv("GMK_GRAM64")) : 0;   // A/B knob: the 64 x 64 kernel
  if (D % 4 == 0 && gmk_aligned16(X) && !old) {
    int splits; long long per;
    gram128_plan(n, D, &splits, &per);
    const int T = (D + G2_T - 1) / G2_T;
    double* partial = (double*)ws;
    double* psum = partial + (size_t)splits * D * D;
    hipLaunchKernelGGL(gram_partial128_kernel, dim3(T * (T + 1) / 2, splits), dim3(256), 0, st, X, (long long)n, D, partial,
                       psum, splits, per);
    GMK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gram_reduce128_kernel, dim3(blocks), dim3(256), 0, st, partial, psum, G, s, D, splits);
    GMK_CHECK_LAUNCH();
    return GMK_OK;
  }
  const int splits = gram_splits(n);
  const int tiles = (D + GR_TILE - 1) / GR_TILE;
  double* partial = (double*)ws;
  double* psum = partial + (size_t)splits * D * D;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(tiles * tiles, splits), dim3(256), 0, st, X, (long long)n, D, partial, psum, splits);
  GMK_CHECK_LAUNCH();
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, psum, G, s, D, splits);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// =============================================================== fp64 GEMM (D x D) ==========
// C = alpha * A @ B + diag * I, n x n row-major fp64, n % 16 == 0 handled by masking.
// 32x32 tile per block (4 waves, one 16x16 MFMA tile each), K chunk 32 in LDS.
#define DG_T 32
#define DG_K 32
struct DgemmBatch { const double* A[2]; const double* B[2]; double* C[2]; };

__global__ __launch_bounds__(256) void dgemm_kernel(const DgemmBatch bt, int n, double alpha, double diag) {
  __shared__ double As[DG_T * (DG_K + 1)];
  __shared__ double Bs[DG_K * (DG_T + 1)];
  const double* __restrict__ A = bt.A[blockIdx.z];
  const double* __restrict__ B = bt.B[blockIdx.z];
  double* __restrict__ C = bt.C[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = blockIdx.y * DG_T, j0 = blockIdx.x * DG_T;
  const int l15 = lane & 15, l4 = lane >> 4;
  f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
  double ra[4], rb[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * 256;
      const int i = e / DG_K, k = e % DG_K;
      ra[u] = (i0 + i < n && k0 + k < n) ? A[(size_t)(i0 + i) * n + k0 + k] : 0.0;
      const int kk = e / DG_T, j = e % DG_T;
      rb[u] = (k0 + kk < n && j0 + j < n) ? B[(size_t)(k0 + kk) * n + j0 + j] : 0.0;
    }
  };
  load(0);
  for (int k0 = 0; k0 < n; k0 += DG_K) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * 256;
      As[(e / DG_K) * (DG_K + 1) + e % DG_K] = ra[u];
      Bs[(e / DG_T) * (DG_T + 1) + e % DG_T] = rb[u];
    }
    __syncthreads();
    if (k0 + DG_K < n) load(k0 + DG_K);
#pragma unroll
    for (int s4 = 0; s4 < DG_K / 4; ++s4) {
      const int k = 4 * s4 + l4;
      const double a = As[(wr * 16 + l15) * (DG_K + 1) + k];
      const double b = Bs[k * (DG_T + 1) + wc * 16 + l15];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wr * 16 + l4 + 4 * r, j = j0 + wc * 16 + l15;
    if (i < n && j < n) C[(size_t)i * n + j] = alpha * acc[r] + (i == j ? diag : 0.0);
  }
}

static int dgemm(hipStream_t st, int n, double alpha, double diag, const double* A0, const double* B0, double* C0,
                 const double* A1 = nullptr, const double* B1 = nullptr, double* C1 = nullptr) {
  DgemmBatch bt;
  bt.A[0] = A0; bt.B[0] = B0; bt.C[0] = C0; bt.A[1] = A1; bt.B[1] = B1; bt.C[1] = C1;
  const int t = (n + DG_T - 1) / DG_T;
  hipLaunchKernelGGL(dgemm_kernel, dim3(t, t, A1 ? 2 : 1), dim3(256), 0, st, bt, n, alpha, diag);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

// ---- small device-side scalar helpers (single block) ----
// scal layout (doubles): 0 dmu2, 1 tr1, 2 tr2, 3 fro (current), 4 c (current scale), 5 trsqrt, 6 fid
__global__ __launch_bounds__(256) void fid_stats_kernel(const double* __restrict__ G1, const double* __restrict__ s1, double n1,
                                                        const double* __restrict__ G2, const double* __restrict__ s2, double n2,
                                                        int D, double diag_eps, double* __restrict__ S1, double* __restrict__ S2,
                                                        double* __restrict__ scal) {
  // Sigma = (G - s s^T / n) / (n - 1) (+ diag_eps I) ; also dmu2 and traces (block 0 only)
  const size_t total = (size_t)D * D;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e / D), j = (int)(e % D);
    const double d = (i == j) ? diag_eps : 0.0;
    S1[e] = (G1[e] - s1[i] * s1[j] / n1) / (n1 - 1.0) + d;
    S2[e] = (G2[e] - s2[i] * s2[j] / n2) / (n2 - 1.0) + d;
  }
  if (blockIdx.x == 0) {
    __shared__ double red[3][4];
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < D; i += 256) {
      const double m1 = s1[i] / n1, m2 = s2[i] / n2;
      a += (m1 - m2) * (m1 - m2);
      b += (G1[(size_t)i * D + i] - s1[i] * s1[i] / n1) / (n1 - 1.0);
      c += (G2[(size_t)i * D + i] - s2[i] * s2[i] / n2) / (n2 - 1.0);
    }
    a = gmk_wave_sum(a); b = gmk_wave_sum(b); c = gmk_wave_sum(c);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
      scal[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      scal[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      scal[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
  }
}

// Frobenius norm of S (optionally symmetrised in place first) -> scal[3]; c = fro*(1+delta) -> scal[4]
__global__ __launch_bounds__(1024) void fro_norm_kernel(double* __restrict__ S, int D, int symmetrise, double delta,
                                                        double* __restrict__ scal) {
  const size_t total = (size_t)D * D;
  if (symmetrise) {
    for (size_t e = threadIdx.x; e < total; e += 1024) {
      const int i = (int)(e / D), j = (int)(e % D);
      if (i < j) { const double v = 0.5 * (S[e] + S[(size_t)j * D + i]); S[e] = v; S[(size_t)j * D + i] = v; }
    }
    __syncthreads();
  }
  double a = 0;
  for (size_t e = threadIdx.x; e < total; e += 1024) a += S[e] * S[e];
  __shared__ double red[16];
  a = gmk_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    const double fro = sqrt(t);
    scal[3] = fro;
    scal[4] = fro * (1.0 + delta);
  }
}

// Y = S / c + delta/(1+delta) I ; Z = I     (c = scal[4]; c == 0 -> Y = 0)
__global__ __launch_bounds__(256) void ns_init_kernel(const double* __restrict__ S, int D, double delta,
                                                      const double* __restrict__ scal, double* __restrict__ Y,
                                                      double* __restrict__ Z) {
  const double c = scal[4];
  const double inv = c > 0.0 ? 1.0 / c : 0.0;
  const double rd = c > 0.0 ? delta / (1.0 + delta) : 0.0;
  const size_t total = (size_t)D * D;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const bool dg = (e / D) == (e % D);
    Y[e] = S[e] * inv + (dg ? rd : 0.0);
    Z[e] = dg ? 1.0 : 0.0;
  }
}

// Y *= sqrt(c) in place (matrix square root of S + ridge)
__global__ __launch_bounds__(256) void scale_sqrtc_kernel(double* __restrict__ Y, int D, const double* __restrict__ scal) {
  const double f = sqrt(scal[4]);
  const size_t total = (size_t)D * D;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) Y[e] *= f;
}

// trsqrt = trace(Y) * sqrt(c) ; fid = dmu2 + tr1 + tr2 - 2 trsqrt
__global__ __launch_bounds__(256) void fid_final_kernel(const double* __restrict__ Y, int D, double* __restrict__ scal) {
  __shared__ double red[4];
  double a = 0;
  for (int i = threadIdx.x; i < D; i += 256) a += Y[(size_t)i * D + i];
  a = gmk_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tr = (red[0] + red[1] + red[2] + red[3]) * sqrt(scal[4]);
    scal[5] = tr;
    scal[6] = scal[0] + scal[1] + scal[2] - 2.0 * tr;
  }
}

extern "C" size_t gmk_fid_workspace_bytes(int D) { return sizeof(double) * ((size_t)7 * D * D + 16) + 256; }

// coupled Newton-Schulz: on return Ybuf[cur] ~ sqrt(S/c + ridge); returns index of the live Y buffer
static int ns_sqrt(hipStream_t st, int D, int iters, double* Y[2], double* Z[2], double* T, int* live) {
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    int rc = dgemm(st, D, -0.5, 1.5, Z[cur], Y[cur], T);                                   // T = 1.5 I - 0.5 Z Y
    if (rc) return rc;
    rc = dgemm(st, D, 1.0, 0.0, Y[cur], T, Y[cur ^ 1], T, Z[cur], Z[cur ^ 1]);            // Y' = Y T ; Z' = T Z
    if (rc) return rc;
    cur ^= 1;
  }
  *live = cur;
  return GMK_OK;
}

extern "C" int gmk_fid_from_moments(const double* G1, const double* s1, int64_t n1, const double* G2, const double* s2,
                                    int64_t n2, int D, int iters, double delta, double diag_eps, double* out,
                                    void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_REQUIRE(G1 && s1 && G2 && s2 && out && ws, GMK_E_NULL);
  GMK_REQUIRE(D > 0 && D <= 4096 && n1 > 1 && n2 > 1 && iters > 0, GMK_E_SHAPE);
  GMK_REQUIRE(ws_bytes >= gmk_fid_workspace_bytes(D) - 256, GMK_E_WORKSPACE);
  hipStream_t st = (hipStream_t)stream;
  const size_t DD = (size_t)D * D;
  double* base = (double*)ws;
  double* S1 = base; double* S2 = base + DD;
  double* Y[2] = {base + 2 * DD, base + 3 * DD};
  double* Z[2] = {base + 4 * DD, base + 5 * DD};
  double* T = base + 6 * DD;
  double* scal = out;  // 8 doubles: dmu2, tr1, tr2, fro, c, trsqrt, fid, (unused)
  int blocks = (int)((DD + 255) / 256); if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fid_stats_kernel, dim3(blocks), dim3(256), 0, st, G1, s1, (double)n1, G2, s2, (double)n2, D, diag_eps, S1, S2, scal);
  GMK_CHECK_LAUNCH();
  // R1 = sqrt(S1)
  hipLaunchKernelGGL(fro_norm_kernel, dim3(1), dim3(1024), 0, st, S1, D, 0, delta, scal);
  hipLaunchKernelGGL(ns_init_kernel, dim3(blocks), dim3(256), 0, st, S1, D, delta, scal, Y[0], Z[0]);
  GMK_CHECK_LAUNCH();
  int live = 0;
  int rc = ns_sqrt(st, D, iters, Y, Z, T, &live);
  if (rc) return rc;
  hipLaunchKernelGGL(scale_sqrtc_kernel, dim3(blocks), dim3(256), 0, st, Y[live], D, scal);
  GMK_CHECK_LAUNCH();
  // M = R1 S2 R1 -> S1 (S1 no longer needed); T as scratch
  rc = dgemm(st, D, 1.0, 0.0, Y[live], S2, T);
  if (rc) return rc;
  rc = dgemm(st, D, 1.0, 0.0, T, Y[live], S1);
  if (rc) return rc;
  hipLaunchKernelGGL(fro_norm_kernel, dim3(1), dim3(1024), 0, st, S1, D, 1, delta, scal);
  hipLaunchKernelGGL(ns_init_kernel, dim3(blocks), dim3(256), 0, st, S1, D, delta, scal, Y[0], Z[0]);
  GMK_CHECK_LAUNCH();
  rc = ns_sqrt(st, D, iters, Y, Z, T, &live);
  if (rc) return rc;
  hipLaunchKernelGGL(fid_final_kernel, dim3(1), dim3(256), 0, st, Y[live], D, scal);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// =============================================================== softmax statistics =========
// out (doubles): [0..C) sum_n p ; [C..2C) arg-max histogram ; 2C sum_n H(p_n) (with eps inside the log) ;
//                2C+1 sum_n top-1 ; 2C+2 sum_n top-N ; 2C+3 n        (accumulated: out += ...)
#define SM_MAXC 32
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ logits, long long n, int C, int topn,
                                                            float eps, double* __restrict__ out) {
  __shared__ double red[2 * SM_MAXC + 3][4];
  double psum[SM_MAXC], hist[SM_MAXC];
#pragma unroll
  for (int c = 0; c < SM_MAXC; ++c) { psum[c] = 0; hist[c] = 0; }
  double hsum = 0, t1 = 0, tn = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float p[SM_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c) { p[c] = c < C ? logits[i * C + c] : -INFINITY; mx = fmaxf(mx, p[c]); }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c) { p[c] = c < C ? expf(p[c] - mx) : 0.f; den += p[c]; }
    float h = 0.f, best = -1.f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c) {
      if (c < C) {
        p[c] = p[c] / den;
        h -= p[c] * logf(p[c] + eps);
        if (p[c] > best) { best = p[c]; arg = c; }   // first maximum, like torch.argmax
        psum[c] += (double)p[c];
      }
    }
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c) hist[c] += (c == arg) ? 1.0 : 0.0;
    hsum += (double)h;
    t1 += (double)best;
    // top-N sum: N passes of "largest remaining" (N is small: 3)
    float acc = 0.f;
    for (int r = 0; r < topn && r < C; ++r) {
      float b = -1.f; int bi = 0;
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c) if (c < C && p[c] > b) { b = p[c]; bi = c; }
      acc += b;
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c) if (c == bi) p[c] = -2.f;
    }
    tn += (double)acc;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < SM_MAXC; ++c) {
    double a = gmk_wave_sum(psum[c]), b = gmk_wave_sum(hist[c]);
    if (lane == 0) { red[c][wave] = a; red[SM_MAXC + c][wave] = b; }
  }
  double a = gmk_wave_sum(hsum), b = gmk_wave_sum(t1), c2 = gmk_wave_sum(tn);
  if (lane == 0) { red[2 * SM_MAXC][wave] = a; red[2 * SM_MAXC + 1][wave] = b; red[2 * SM_MAXC + 2][wave] = c2; }
  __syncthreads();
  // one double atomicAdd per block and statistic (<= 64 blocks): order-dependent only at 1e-16
  if (threadIdx.x < 2 * SM_MAXC + 3) {
    const int k = threadIdx.x;
    const double v = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    int dst = -1;
    if (k < SM_MAXC) { if (k < C) dst = k; }
    else if (k < 2 * SM_MAXC) { if (k - SM_MAXC < C) dst = C + (k - SM_MAXC); }
    else dst = 2 * C + (k - 2 * SM_MAXC);
    if (dst >= 0) atomicAdd(out + dst, v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 2 * C + 3, (double)n);
}

extern "C" int gmk_softmax_stats(const float* logits, int64_t n, int C, int topn, float eps, double* out,
                                 gmk_stream_t stream) {
  GMK_REQUIRE(logits && out, GMK_E_NULL);
  GMK_REQUIRE(n > 0 && C > 0 && C <= SM_MAXC && topn >= 1, GMK_E_SHAPE);
  long long blocks = (n + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, (long long)n, C, topn, eps, out);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// gemm.hip -- fp32 MFMA (32x32x2) strided GEMM for the two fully connected layers
// (nn.Linear fwd / dgrad / wgrad, /root/reference/models/autoencoders.py:41,57,81,95).
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = A[i*sai + k*sak],  B(k,j) = B[k*sbk + j*sbj]
// 64x64 block tile, 4 waves (2x2) of one 32x32 accumulator, K-chunk 32 staged in LDS.
// Small-M shapes (M = batch) are split along K into fp32 slabs and reduced in a fixed order.
#include "gmk_common.h"

#define GT 64
#define GK 32
#define AS_STRIDE 33
#define BS_STRIDE 65

enum { GEPI_NONE = 0, GEPI_BIAS_ACT = 1, GEPI_MUL = 2 };

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* mul; float* dact;
  int M, N, K;
  long long sai, sak, sbk, sbj;
  int ldc, epi, act;
  int kper;      // K range per split
  float* slab;   // != nullptr: write raw partials slab[z][M][N]
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

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmArgs g, int splits) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += g.slab[(size_t)z * total + e];
    gemm_epilogue(g, (int)(e / g.N), (int)(e % g.N), v);
  }
}

// column sums: out[j] = sum_i t[i*N + j]  (bias gradient); one thread per column, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ t, float* __restrict__ out, int M, int N) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < M; i += 4) {
    s0 += t[(size_t)i * N + j]; s1 += t[(size_t)(i + 1) * N + j];
    s2 += t[(size_t)(i + 2) * N + j]; s3 += t[(size_t)(i + 3) * N + j];
  }
  for (; i < M; ++i) s0 += t[(size_t)i * N + j];
  out[j] = (s0 + s1) + (s2 + s3);
}

static int pick_splits(int M, int N, int K) {
  const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  int splits = 1;
  while (tiles * splits < 384 && K / (splits * 2) >= 128 && splits < 64) splits *= 2;
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
  int kper = (g.K + splits - 1) / splits;
  kper = (kper + GK - 1) / GK * GK;
  g.kper = kper;
  const int zs = (g.K + kper - 1) / kper;
  dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, zs);
  if (zs > 1) {
    if (!ws || ws_bytes < sizeof(float) * (size_t)zs * g.M * g.N) return GMK_E_WORKSPACE;
    g.slab = (float*)ws;
  } else {
    g.slab = nullptr;
  }
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (zs > 1) {
    size_t total = (size_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g, zs);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return GMK_OK;
}

extern "C" int gmk_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* dact, int M,
                              int K, int N, int act, void* ws, size_t ws_bytes, gmk_stream_t stream) {
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
  GMK_REQUIRE(dy && x && dw, GMK_E_NULL);
  GMK_REQUIRE(M > 0 && K > 0 && N > 0, GMK_E_SHAPE);
  GemmArgs g{};
  g.A = dy; g.B = x; g.C = dw; g.bias = nullptr; g.dact = nullptr; g.mul = nullptr;
  g.M = N; g.N = K; g.K = M; g.sai = 1; g.sak = N; g.sbk = K; g.sbj = 1; g.ldc = K;
  g.epi = GEPI_NONE; g.act = 0;
  int rc = run_gemm(g, ws, ws_bytes, (hipStream_t)stream);
  if (rc) return rc;
  if (dbias) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, dbias, M, N);
    GMK_CHECK_LAUNCH();
  }
  return GMK_OK;
}

// adamw.hip -- fused AdamW over one flat parameter buffer (all 24 tensors of the autoencoder in a
// single launch): torch.optim.AdamW semantics (/root/reference/main.py:52,214: decoupled weight
// decay, betas .9/.999, eps 1e-8).  HBM-bound: reads p,g,m,v and writes p,m,v = 28 B/parameter.
// `gscale` multiplies the gradient first (1/world_size after a SUM all-reduce).
#include "gmk_common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, long long n4,
                                                    float lr, float b1, float b2, float eps, float wd, float step_size,
                                                    float inv_sqrt_bc2, float gscale) {
  const float decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 P = p[i], G = g[i], M = m[i], V = v[i];
#define UPD(c)                                               \
  {                                                          \
    float gg = G.c * gscale;                                 \
    float pp = P.c * decay;                                  \
    M.c = b1 * M.c + (1.f - b1) * gg;                        \
    V.c = b2 * V.c + (1.f - b2) * gg * gg;                   \
    float denom = sqrtf(V.c) * inv_sqrt_bc2 + eps;           \
    P.c = pp - step_size * (M.c / denom);                    \
  }
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    p[i] = P; m[i] = M; v[i] = V;
  }
}

extern "C" int gmk_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, float gscale,
                              gmk_stream_t stream) {
  GMK_REQUIRE(p && g && m && v, GMK_E_NULL);
  GMK_REQUIRE(n > 0 && n % 4 == 0 && step >= 1, GMK_E_SHAPE);
  GMK_REQUIRE(gmk_aligned16(p) && gmk_aligned16(g) && gmk_aligned16(m) && gmk_aligned16(v), GMK_E_ALIGN);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                     (const float4*)g, (float4*)m, (float4*)v, n4, lr, beta1, beta2, eps, weight_decay, step_size,
                     inv_sqrt_bc2, gscale);
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}

// Dependent-chain probe for v_mfma_f32_32x32x2_f32: NACC independent accumulators per wave, 1 or 2 blocks per CU
// (= waves per SIMD).  Shows what an MFMA loop with too few independent accumulators loses to the MFMA result latency.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o tools/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int blocks_per_cu) {
  float* out;
  const int blocks = 256 * blocks_per_cu, iters = 10000;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f, 1.21f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 8 * 4096.0;
  printf("NACC %d  waves/SIMD %d: %.1f TFLOP/s\n", NACC, blocks_per_cu, flop / best / 1e9);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 2; ++w) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
  return 0;
}

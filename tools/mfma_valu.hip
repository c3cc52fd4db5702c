// Does VALU work overlap fp32 MFMA?  9 independent v_mfma_f32_32x32x2_f32 per step + NV independent v_fma_f32 per step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu.hip -o tools/bin/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  float a = 0.5f + threadIdx.x, b = 0.25f;
  for (int s = 0; s < iters; ++s) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 9; ++j) {
        const int i = (t * (NV / 9) + j) & 7;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
      }
    }
  }
  float sum = 0.f;
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int NV>
void run(float* out) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 9 * 4096.0;
  printf("9 MFMA + %3d v_fma per step: %.3f ms  %.1f TFLOP/s (MFMA flops only)\n", NV, best, flop / best / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  run<0>(out); run<0>(out); run<9>(out); run<18>(out); run<36>(out); run<72>(out); run<144>(out);
  return 0;
}

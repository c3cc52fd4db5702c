// Do LDS reads cost fp32-MFMA issue time?  Per step: 9 independent v_mfma_f32_32x32x2_f32 and ND ds_read
// instructions (immediate offsets, no address VALU) whose results feed the next step's MFMA operands.
// KIND 0: ds_read_b32, 1: ds_read2_b32, 2: ds_read_b128.   WAVES = blocks per CU (waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ds.hip -o tools/bin/mfma_ds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ND, int KIND, int STRIDE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)(lane * STRIDE * 4) + (threadIdx.x >> 6) * 4096 * 4;  // byte address, per-wave region
  float a = 0.5f + threadIdx.x, b = 0.25f;
  for (int s = 0; s < iters; ++s) {
    float na = a, nb = b;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < ND / 9; ++j) {
        constexpr int dummy = 0;
        const int off = ((t * (ND / 9) + j) * 16) & 1023;
        if (KIND == 0) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(0)); (void)off; (void)v; }
        else if (KIND == 1) { f32x2 v; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(v) : "v"(base)); (void)v; }
        else { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base)); (void)v; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a = na; b = nb;
  }
  float sum = 0.f;
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum + a;
}
template <int ND, int KIND, int STRIDE>
void run(float* out, int waves) {
  const int iters = 4000, blocks = 256 * waves;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<ND, KIND, STRIDE>), dim3(blocks), dim3(256), 16384 * 4, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 9 * 4096.0;
  const char* kn[] = {"ds_read_b32", "ds_read2_b32", "ds_read_b128"};
  printf("9 MFMA + %2d %-12s (lane stride %d floats) per step, %d wave/SIMD: %.1f TFLOP/s\n", ND, kn[KIND], STRIDE, waves, flop / best / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 1>(out, w);
    run<9, 0, 1>(out, w); run<18, 0, 1>(out, w); run<27, 0, 1>(out, w);
    run<9, 1, 1>(out, w); run<18, 1, 1>(out, w);
    run<9, 2, 4>(out, w);
    run<18, 0, 2>(out, w);   // 2-way bank conflict
    run<18, 0, 73>(out, w);  // odd stride (weights tile)
  }
  return 0;
}

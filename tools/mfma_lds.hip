// MFMA + LDS-operand issue probe: what a wgrad/igemm style inner loop (ds_read_b32 operands, fp32 32x32x2 MFMA)
// sustains with one wave per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o tools/bin/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NT, int MODE>  // MODE 0: operands from registers; 1: LDS reads pipelined by hand; 2: LDS reads, naive
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, jj = lane & 31, h = lane >> 5;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a_n = 0.f, b_n[NT];
  auto rd = [&](int s) {
    const int p = (2 * s + h) & 127;
    a_n = lds[jj * 33 + (p & 31)];
    for (int t = 0; t < NT; ++t) b_n[t] = lds[2048 + jj * stride + p + t * 3];
  };
  if (MODE == 1) rd(0);
  for (int s = 0; s < iters; ++s) {
    float a, b[NT];
    if (MODE == 0) { a = 0.5f + s; for (int t = 0; t < NT; ++t) b[t] = 0.25f * t + s; }
    else if (MODE == 1) { a = a_n; for (int t = 0; t < NT; ++t) b[t] = b_n[t]; rd(s + 1); }
    else { rd(s); a = a_n; for (int t = 0; t < NT; ++t) b[t] = b_n[t]; }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
  }
  float sum = 0.f;
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int NT, int MODE>
void run(const char* name, float* out, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, MODE>), dim3(blocks), dim3(256), 8192 * 4 + 4096, 0, out, iters, 41);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * NT * 4096.0;
  printf("%-44s blocks %4d: %.3f ms  %.1f TFLOP/s\n", name, blocks, best, flop / best / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 2048 * 256 * 4);
  run<9, 0>("9 acc, register operands", out, 256);
  run<9, 1>("9 acc, LDS operands, hand-pipelined", out, 256);
  run<9, 2>("9 acc, LDS operands, naive", out, 256);
  run<4, 1>("4 acc, LDS operands, pipelined, 1 wave/SIMD", out, 256);
  run<4, 1>("4 acc, LDS operands, pipelined, 2 waves/SIMD", out, 512);
  run<2, 1>("2 acc, LDS operands, pipelined, 2 waves/SIMD", out, 512);
  run<2, 2>("2 acc, LDS operands, naive, 2 waves/SIMD", out, 512);
  run<2, 2>("2 acc, LDS operands, naive, 4 waves/SIMD", out, 1024);
  return 0;
}

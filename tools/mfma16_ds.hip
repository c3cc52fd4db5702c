// What does the k loop of conv_wgrad16.h cost on its own?  One wave per SIMD, 9 independent accumulators.
// Per step: 9 v_mfma_f32_16x16x4_f32 (asm), and depending on MODE
//   0: nothing else (operands fixed)                          -> bare MFMA rate
//   1: + one ds_read_b32 behind every MFMA (10th behind the last), results unused, no waits
//   2: + the reads feed the NEXT step's operands, one counted s_waitcnt lgkmcnt(8) in front of every MFMA
//   3: as 2 but ONE wait per step (lgkmcnt(0)) in front of the first MFMA (reads issued a step ahead)
//   4: as 2 with v_mfma_f32_32x32x2_f32 (16 acc registers each; same flops per step: 4.5 MFMAs ~ use 9, half the steps)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_ds.hip -o tools/bin/mfma16_ds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int WPS>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x4 acc[9];
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  // conflict-free: 16 rows x stride 66 words + k
  const unsigned base = (unsigned)(((lane & 15) * 66 + (lane >> 4)) * 4) + (threadIdx.x >> 6) * 1100 * 4;
  float ra[2] = {0.5f, 0.25f}, rb[2][9];
  for (int t = 0; t < 9; ++t) { rb[0][t] = 1.f + t; rb[1][t] = 2.f + t; }
  for (int s = 0; s < iters; s += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        if (MODE == 3 && t == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(ra[h]), "v"(rb[h][t]));
        if (MODE == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(64)); (void)v; }
        if (MODE == 2 || MODE == 3) {
          if (t == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ra[h ^ 1]) : "v"(base), "n"(16));
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rb[h ^ 1][t]) : "v"(base), "n"(64));
        }
      }
      if (MODE == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(64)); (void)v; }
    }
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  float sum = 0.f;
  for (int t = 0; t < 9; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = sum + ra[0] + rb[0][3];
}

template <int WPS>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)(((lane & 31) * 33 + (lane >> 5)) * 4) + (threadIdx.x >> 6) * 1100 * 4;
  float ra[2] = {0.5f, 0.25f}, rb[2][9];
  for (int t = 0; t < 9; ++t) { rb[0][t] = 1.f + t; rb[1][t] = 2.f + t; }
  for (int s = 0; s < iters; s += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(ra[h]), "v"(rb[h][t]));
        if (t == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ra[h ^ 1]) : "v"(base), "n"(16));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rb[h ^ 1][t]) : "v"(base), "n"(64));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  float sum = 0.f;
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum + ra[0] + rb[0][3];
}

template <int MODE>
void run(float* out, int waves) {
  const int iters = 8000, blocks = 256 * waves;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    if (MODE == 4) hipLaunchKernelGGL((k32<1>), dim3(blocks), dim3(256), 8192 * 4, 0, out, iters);
    else hipLaunchKernelGGL((k16<MODE, 1>), dim3(blocks), dim3(256), 8192 * 4, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 9 * (MODE == 4 ? 4096.0 : 2048.0);
  const char* kn[] = {"16x16x4 bare", "16x16x4 + 10 unused ds_read_b32 per step", "16x16x4 + reads feed operands, lgkmcnt(8) per MFMA",
                      "16x16x4 + reads feed operands, one lgkmcnt(0) per step", "32x32x2 + reads feed operands, lgkmcnt(8) per MFMA"};
  printf("%-58s %d wave/SIMD: %6.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", kn[MODE], waves, flop / best / 1e9,
         best * 1e-3 * 2.4e9 / ((double)iters * 9));
}
int main() {
  float* out; (void)hipMalloc(&out, 2048 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0>(out, w); run<1>(out, w); run<2>(out, w); run<3>(out, w); run<4>(out, w);
  }
  return 0;
}

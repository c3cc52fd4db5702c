// gmk_common.h -- shared device/host helpers for libgmk (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "../../include/gmk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define GMK_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

#define GMK_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device, size)
// instead of on every launch (gmk_api.hip); returns hipSuccess or the runtime's error.
hipError_t gmk_ensure_dyn_lds(const void* kernel, int bytes);

static inline bool gmk_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int gmk_ilog2(int64_t v) {
  int l = 0;
  while ((1ll << l) < v) ++l;
  return l;
}
static inline bool gmk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- activations: value and derivative, PyTorch default parameters --------------------
// (reference: /root/reference/utils.py:49-66)
__device__ __forceinline__ float gmk_sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }

// exact-erf GELU and its derivative from ONE exponential and ONE reciprocal: e = exp(-z^2/2) serves both the pdf
// and erfc(|z|/sqrt2) = P(t) e, t = 1/(1 + p |z|/sqrt2)  (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7; the 0.5 of
// the cdf is folded into the coefficients).  v_exp_f32 / v_rcp_f32 are used directly (1 ulp).
__device__ __forceinline__ void gmk_gelu_fwd(float z, float& a, float& d) {
  const float az = fabsf(z);
  const float e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);          // exp(-z^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, az, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float q = p * t * e;                    // 0.5 * erfc(|z|/sqrt2)
  const float cdf = 0.5f + copysignf(0.5f - q, z);
  a = z * cdf;
  d = fmaf(z * 0.39894228040143267794f, e, cdf);
}

// The same GELU on two values at a time in packed fp32 (v_pk_mul_f32 / v_pk_fma_f32): 21 vector-ALU instructions per pair
// instead of 38.  fp32 MFMA and the vector ALU share one issue port on gfx950 and a co-resident block in its K loop re-issues a
// 64-cycle MFMA the moment the last one retires, so an overlapped epilogue advances roughly one instruction per MFMA
// (tools/block_trace.py): its length is its instruction count.  Same operations in the same order as gmk_gelu_fwd: same bits.
typedef float gmk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gmk_gelu_fwd2(gmk_f2 z, gmk_f2& a, gmk_f2& d) {
  const gmk_f2 ex = (z * z) * -0.72134752044448170368f;
  gmk_f2 e, az, t, cs;
  e.x = __builtin_amdgcn_exp2f(ex.x); e.y = __builtin_amdgcn_exp2f(ex.y);
  az.x = fabsf(z.x); az.y = fabsf(z.y);
  const gmk_f2 u = az * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  t.x = __builtin_amdgcn_rcpf(u.x); t.y = __builtin_amdgcn_rcpf(u.y);
  gmk_f2 p = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  p = p * t + (0.5f * 1.421413741f);
  p = p * t + (0.5f * -0.284496736f);
  p = p * t + (0.5f * 0.254829592f);
  const gmk_f2 q = p * t * e;
  const gmk_f2 hq = 0.5f - q;
  cs.x = copysignf(hq.x, z.x); cs.y = copysignf(hq.y, z.y);
  const gmk_f2 cdf = 0.5f + cs;
  a = z * cdf;
  d = (z * 0.39894228040143267794f) * e + cdf;
}
__device__ __forceinline__ void gmk_gelu_fwd4(const float4& z, float4& a, float4& d) {
  gmk_f2 a0, d0, a1, d1;
  gmk_gelu_fwd2(gmk_f2{z.x, z.y}, a0, d0);
  gmk_gelu_fwd2(gmk_f2{z.z, z.w}, a1, d1);
  a = make_float4(a0.x, a0.y, a1.x, a1.y);
  d = make_float4(d0.x, d0.y, d1.x, d1.y);
}

__device__ __forceinline__ void gmk_act_fwd(int act, float z, float& a, float& d) {
  switch (act) {
    case GMK_ACT_RELU:
      a = z > 0.f ? z : 0.f;
      d = z > 0.f ? 1.f : 0.f;
      break;
    case GMK_ACT_LEAKY_RELU:
      a = z > 0.f ? z : 0.01f * z;
      d = z > 0.f ? 1.f : 0.01f;
      break;
    case GMK_ACT_ELU: {
      float e = expm1f(z);
      a = z > 0.f ? z : e;
      d = z > 0.f ? 1.f : e + 1.f;
      break;
    }
    case GMK_ACT_SELU: {
      const float sc = 1.0507009873554804934193349852946f, al = 1.6732632423543772848170429916717f;
      float e = expm1f(z);
      a = sc * (z > 0.f ? z : al * e);
      d = sc * (z > 0.f ? 1.f : al * (e + 1.f));
      break;
    }
    case GMK_ACT_GELU:
      gmk_gelu_fwd(z, a, d);
      break;
    case GMK_ACT_SILU: {
      float s = 1.0f / (1.0f + expf(-z));
      a = z * s;
      d = s * (1.f + z * (1.f - s));
      break;
    }
    case GMK_ACT_MISH: {
      float sp = z > 20.f ? z : log1pf(expf(z));
      float th = tanhf(sp);
      float s = 1.0f / (1.0f + expf(-z));
      a = z * th;
      d = th + z * (1.f - th * th) * s;
      break;
    }
    case GMK_ACT_SIGMOID: {
      float s = 1.0f / (1.0f + expf(-z));
      a = s;
      d = s * (1.f - s);
      break;
    }
    default:
      a = z;
      d = 1.f;
  }
}

// Epilogue helper: run `body(actf)` with actf(float4 z, float4& a, float4& d) specialised for the (block-uniform) activation
// `act`: identity, GELU (the reference default; packed fp32) and ReLU get straight-line code, the rest share the per-element
// switch (specialising all seven triples the build time of the conv kernels for activations nobody benchmarks).
#define GMK_ACT4_SCALAR(expr_)                                                                      \
  [=](const float4& z_, float4& a_, float4& d_) {                                                   \
    auto f1_ = [=](float zz_, float& aa_, float& dd_) { expr_; };                                   \
    f1_(z_.x, a_.x, d_.x); f1_(z_.y, a_.y, d_.y); f1_(z_.z, a_.z, d_.z); f1_(z_.w, a_.w, d_.w);     \
  }
#define GMK_WITH_ACT(act, body)                                                                            \
  switch (act) {                                                                                           \
    case GMK_ACT_GELU: body([](const float4& z_, float4& a_, float4& d_) { gmk_gelu_fwd4(z_, a_, d_); }); break; \
    case GMK_ACT_RELU: body(GMK_ACT4_SCALAR(gmk_act_fwd(GMK_ACT_RELU, zz_, aa_, dd_))); break;             \
    case GMK_ACT_NONE: body([](const float4& z_, float4& a_, float4& d_) { a_ = z_; d_ = make_float4(1.f, 1.f, 1.f, 1.f); }); break; \
    default: { const int act__ = (act); body(GMK_ACT4_SCALAR(gmk_act_fwd(act__, zz_, aa_, dd_))); } break; \
  }

// ---- buffer (SRD) loads: hardware bounds check returns 0 for out-of-range offsets, so padded /
// masked tile loads need no branch and no select (an invalid lane passes GMK_OOB as its offset).
#define GMK_OOB ((int)0x80000000)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gmk_make_rsrc(const void* base, long long bytes) {
  const int nb = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0 : (int)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nb, 0x00020000);
}
__device__ __forceinline__ float gmk_buffer_load(__amdgpu_buffer_rsrc_t r, int voffset_bytes, int soffset_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset_bytes, soffset_bytes, 0));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// load (s_waitcnt vmcnt(0)), which serialises register prefetch against the barrier; this one leaves
// global loads in flight (the compiler still waits on a load's registers before their first use).
// 16-byte global load issued exactly where it is written, for software prefetch across a loop iteration.  A plain
// load whose first use is in the next iteration gets sunk behind the MFMA section (machine-sink) or delayed by the
// scheduler, which exposes the whole HBM latency once per chunk.  The compiler does not track these loads: wait with
// gmk_vm_wait_all() and pass every destination through gmk_pin() before its first use.
__device__ __forceinline__ void gmk_gload_f4_async(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmk_vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void gmk_pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// Compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N).  Used where an index has to be a constant
// expression (inline-asm immediates).
template <int I, int N, class F>
__device__ __forceinline__ void gmk_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gmk_static_for<I + 1, N>(f);
  }
}

// LDS byte address of a __shared__ object (what ds_read / ds_write take as their VGPR operand).
__device__ __forceinline__ unsigned gmk_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// ds_read_b32 with an immediate offset, issued exactly where it is written.  The compiler neither merges these into
// ds_read2_b32 (which costs fp32-MFMA issue time on gfx950, tools/mfma_ds.hip) nor tracks their completion: the caller
// waits with GMK_LDS_WAIT* before the first use.
#define GMK_DS_READ(dst, addr, byte_off) \
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off) : "memory")
// s_waitcnt lgkmcnt(n) tied to the registers it protects, so that their consumers cannot be moved above it
#define GMK_LDS_WAIT2(n, x, y) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(n))
#define GMK_LDS_WAIT3(n, x, y, z) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x), "+v"(y), "+v"(z) : "n"(n))

__device__ __forceinline__ void gmk_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wave64 sum reduction (all lanes get the result in lane 0; use after per-thread partials)
__device__ __forceinline__ double gmk_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float gmk_wave_sumf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

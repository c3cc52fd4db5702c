// conv_wgrad16.hip -- instantiations + launcher of the 16x16x4-MFMA 3x3 weight-gradient kernel.
#include "conv_wgrad16.h"

template <int SB, int LGSW>
static int launch16(WgradGeom& g, float* dw, float* dbias, hipStream_t s) {
  constexpr Wg16Geo GEO = gmk_wg16_geo(1 << LGSW, SB);
  constexpr size_t lds = 2 * sizeof(float) * (size_t)GEO.buf;
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (GEO.PX != g.P) return GMK_E_SHAPE;
  auto kern = g.bias_mode == 0 ? conv_wgrad16_kernel<SB, LGSW, 0>
            : g.bias_mode == 1 ? conv_wgrad16_kernel<SB, LGSW, 1> : conv_wgrad16_kernel<SB, LGSW, 2>;
  if (lds > 64 * 1024) {
    hipError_t e = gmk_ensure_dyn_lds((const void*)kern, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(g.G, g.csT * g.cbT);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  constexpr int SLAB = 9 * GMK_WG_T * GMK_WG_T + GMK_WG_T;
  if (g.dbg & 16) return GMK_OK;   // timing ablation (GMK_WG_DBG=16): no slab reduction, dw is not written
  if (dw == nullptr) return GMK_OK;   // deferred reduction: the slabs stay in the caller's workspace (gmk_conv_wgrad_reduce_batch)
  hipLaunchKernelGGL((conv_wgrad_reduce_nolds_kernel<3, 9>), dim3((SLAB + 127) / 128, grid.y), dim3(256), 0, s, g.slab, dw,
                     dbias, g.G, (int)grid.y, g.Cs, g.Cb, g.cbT, g.bias_mode);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

int gmk_launch_wgrad16(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) {
  if (sb == 1) {
    switch (g.lgSW) {
      case 3: return launch16<1, 3>(g, dw, dbias, s);
      case 4: return launch16<1, 4>(g, dw, dbias, s);
      case 5: return launch16<1, 5>(g, dw, dbias, s);
      case 6: return launch16<1, 6>(g, dw, dbias, s);
      default: return GMK_E_SHAPE;
    }
  }
  switch (g.lgSW) {
    case 3: return launch16<2, 3>(g, dw, dbias, s);
    case 4: return launch16<2, 4>(g, dw, dbias, s);
    case 5: return launch16<2, 5>(g, dw, dbias, s);
    default: return GMK_E_SHAPE;
  }
}

// ---- deferred reduction of several layers' slabs in ONE launch (gmk_conv_wgrad_reduce_batch) ----
// The same arithmetic as conv_wgrad_reduce_nolds_kernel<3, 9> per layer (blockIdx.z = layer): each weight gradient of the
// train step used to be followed by its own 4 us reduce launch, which in the overlapped step waits ~12 us behind the fat
// kernels of the other streams -- ten of them on the weight-gradient streams, the longer chain of backward.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_batch_kernel(const Wg16ReduceBatch b) {
  const Wg16ReduceItem& it = b.item[blockIdx.z];
  if ((int)blockIdx.y >= it.ny) return;
  constexpr int KK = 9, T = GMK_WG_T, TILE = KK * T * T, SLAB = TILE + T;
  const int y = blockIdx.y;
  const int tile = y;
  const int cs0 = (tile / it.cbT) * T, cb0 = (tile % it.cbT) * T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el = lane & 7, grp = lane >> 3;
  const int e = ((blockIdx.x * 4 + wave) * 8 + el) * 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < SLAB) {
    const float* src = it.slab + ((size_t)grp * it.ny + y) * (size_t)SLAB + e;
    const size_t step = (size_t)8 * it.ny * SLAB;
#pragma unroll 4
    for (int gI = grp; gI < it.G; gI += 8, src += step) {
      const float4 v = *(const float4*)src;
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
#pragma unroll
  for (int o = 8; o <= 32; o <<= 1) {
    sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64);
    sum.z += __shfl_xor(sum.z, o, 64); sum.w += __shfl_xor(sum.w, o, 64);
  }
  if (grp != 0 || e >= SLAB) return;
  const float out[4] = {sum.x, sum.y, sum.z, sum.w};
  if (e < TILE) {
    const int t = e / (T * T), cb = (e / T) % T, cs = e % T;
    if (cb0 + cb < it.Cb) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cs0 + cs + k < it.Cs) it.dw[((size_t)(cs0 + cs + k) * it.Cb + cb0 + cb) * KK + t] = out[k];
    }
  } else if (it.dbias) {
    const int c = e - TILE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (it.bias_mode == 1 && (tile % it.cbT) == 0 && cs0 + c + k < it.Cs) it.dbias[cs0 + c + k] = out[k];
      if (it.bias_mode == 2 && (tile / it.cbT) == 0 && cb0 + c + k < it.Cb) it.dbias[cb0 + c + k] = out[k];
    }
  }
}

int gmk_launch_wgrad16_reduce_batch(const Wg16ReduceBatch& b, int n, int max_ny, hipStream_t s) {
  constexpr int SLAB = 9 * GMK_WG_T * GMK_WG_T + GMK_WG_T;
  hipLaunchKernelGGL(conv_wgrad_reduce_batch_kernel, dim3((SLAB + 127) / 128, max_ny, n), dim3(256), 0, s, b);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

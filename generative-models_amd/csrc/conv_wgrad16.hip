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

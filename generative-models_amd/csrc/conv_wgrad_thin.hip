// conv_wgrad_thin.hip -- instantiations + launcher of the thin-channel wgrad kernel.
#include "conv_wgrad_thin.h"

template <int KS, int NB>
static int launch_thin(ThinGeom& g, float* dw, float* dbias, hipStream_t s) {
  constexpr int SLAB = 64 * 32 * NB + 64;
  size_t lds = sizeof(float) * (64 * (size_t)GMK_THIN_SS + ((g.npos + 4) & ~3));
  const size_t red = sizeof(float) * 4 * 64 * 32 * NB;  // 4 partial regions for the in-block reduction
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return GMK_E_SHAPE;
  auto kern = conv_wgrad_thin_kernel<KS, NB>;
  hipError_t e = gmk_ensure_dyn_lds((const void*)kern, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(g.G, g.nWT), dim3(GMK_THIN_THREADS), lds, s, g);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_wgrad_thin_reduce_kernel<KS, NB>), dim3((SLAB + 31) / 32, g.nWT), dim3(1024), 0, s, g.slab, dw,
                     dbias, g);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

int gmk_launch_wgrad_thin(ThinGeom& g, int ksize, float* dw, float* dbias, hipStream_t s) {
  const int nb = gmk_thin_nb(g);
  if (ksize == 3) {
    if (nb == 1) return launch_thin<3, 1>(g, dw, dbias, s);
    if (nb == 2) return launch_thin<3, 2>(g, dw, dbias, s);
    return launch_thin<3, 3>(g, dw, dbias, s);
  }
  if (nb == 1) return launch_thin<5, 1>(g, dw, dbias, s);
  if (nb == 2) return launch_thin<5, 2>(g, dw, dbias, s);
  return launch_thin<5, 3>(g, dw, dbias, s);
}

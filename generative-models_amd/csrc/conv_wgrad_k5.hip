// conv_wgrad_k5.hip -- 5x5 instantiations of the wgrad kernel (one ky row of taps per block).
#include "conv_wgrad.h"
int gmk_launch_wgrad_k5(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) { return gmk_launch_wgrad_impl<5>(g, sb, dw, dbias, s); }

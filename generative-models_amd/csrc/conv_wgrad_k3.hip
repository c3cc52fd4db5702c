// conv_wgrad_k3.hip -- 3x3 instantiations of the wgrad kernel.
#include "conv_wgrad.h"
int gmk_launch_wgrad_k3(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) { return gmk_launch_wgrad_impl<3>(g, sb, dw, dbias, s); }

// conv_igemm_k3.hip -- 3x3 instantiations of the implicit-GEMM conv kernel.
#include "conv_igemm.h"
int gmk_launch_conv_k3(ConvGeom& g, int mode, hipStream_t s) { return gmk_launch_conv_impl<3>(g, mode, s); }

// conv_igemm_k5.hip -- 5x5 instantiations of the implicit-GEMM conv kernel.
#include "conv_igemm.h"
int gmk_launch_conv_k5(ConvGeom& g, int mode, hipStream_t s) { return gmk_launch_conv_impl<5>(g, mode, s); }

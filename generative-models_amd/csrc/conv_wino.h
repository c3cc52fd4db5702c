// conv_wino.h -- stride-1 3x3 convolution (forward, and data gradient with flipped taps) by Winograd F(2x2, 3x3)
// on v_mfma_f32_16x16x4_f32: 16 multiplies per 2x2 output tile and channel pair instead of 36 (2.25x less MFMA work
// than the implicit GEMM of conv_igemm.h).  fp32 throughout; the transform constants are 0, +-1, +-1/2, so the result
// differs from the direct sum only by fp32 rounding of the same order (measured on the CPU: 2.7e-7 relative against
// 2.1e-7 of a direct fp32 convolution, fp64 reference; the parity bar is 1e-4).  Kernel: conv_wino.hip.
#pragma once
#include "conv_igemm.h"

#define WINO_TB 32          /* tiles per block */
#define WINO_MB 64          /* output channels per block */
#define WINO_CK 8           /* input channels per chunk */
#define WINO_ES 130         /* LDS row stride (floats) of the epilogue transpose: even (b64 accesses), 2c mod 32 banks */
#define WINO_RS 76          /* LDS row stride of the raw weight rows (72 floats per output channel and chunk; 16-byte rows) */

static inline bool gmk_wino_ok(const ConvGeom& g, int ksize, int mode) {
  if (ksize != 3 || mode != 0) return false;
  if (!((g.w_layout == 0 && !g.flip) || (g.w_layout == 1 && g.flip))) return false;   // conv forward / conv dgrad
  if (g.C % WINO_CK != 0 || g.M % 16 != 0) return false;
  if (g.IW < 4 || g.IW > 32 || g.IH < 4) return false;   // tile rows of at most 16 tiles (DPP row shifts between neighbours)
  const long long xbytes = 4ll * g.N * g.C * g.IH * g.IW, ybytes = 4ll * g.N * g.M * g.IH * g.IW;
  if (xbytes >= 0x7FFF0000ll || ybytes >= 0x7FFF0000ll) return false;   // 32-bit buffer offsets
  if (((long long)g.N * g.IH * g.IW / 4) % WINO_TB != 0) return false;
  return true;
}

int gmk_launch_conv_wino(ConvGeom& g, hipStream_t s);
bool gmk_wino_fwd_ready();   // forward kernel's 66 KiB LDS limit raised on the current device (once); false = use the implicit GEMM

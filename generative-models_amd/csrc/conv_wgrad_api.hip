// conv_wgrad_api.hip -- C-ABI entry points for conv / deconv weight + bias gradients.
#include "conv_wgrad.h"
#include "plan.h"
#include "conv_wgrad16.h"
#include "conv_wgrad_thin.h"

// 3x3 layers run on the 16x16x4-MFMA kernel (conv_wgrad16.h); GMK_WG16=0 selects the older 32x32x2 kernel for A/B runs.
static bool use_wg16(int N, int Cs, int Cb, int SH, int SW, int ksize, int sb) {
  static const int on = getenv("GMK_WG16") ? atoi(getenv("GMK_WG16")) : 1;
  return on && ksize == 3 && gmk_wg16_supported(N, Cs, Cb, SH, SW, sb);
}

static void wgrad_plan(WgradGeom& g, int N, int Cs, int Cb, int SH, int SW, int ksize, int sb) {
  g.N = N; g.Cs = Cs; g.Cb = Cb; g.SH = SH; g.SW = SW; g.BHt = SH * sb; g.BWt = SW * sb;
  g.lgSW = gmk_ilog2(SW); g.lgSHW = gmk_ilog2((int64_t)SH * SW); g.lgBWt = gmk_ilog2(g.BWt);
  g.P = use_wg16(N, Cs, Cb, SH, SW, ksize, sb) ? gmk_wg16_geo(SW, sb).PX : gmk_wgrad_patch(SW, ksize, sb).PX;
  g.lgP = gmk_ilog2(g.P);
  g.total_chunks = ((long long)N * SH * SW) / g.P;
  g.csT = (Cs + GMK_WG_T - 1) / GMK_WG_T; g.cbT = (Cb + GMK_WG_T - 1) / GMK_WG_T;
  const int ntg = ksize == 3 ? 1 : ksize;
  static const int blocks16 = getenv("GMK_WG16_BLOCKS") ? atoi(getenv("GMK_WG16_BLOCKS")) : 256;   // A/B knob
  const long long nblk = use_wg16(N, Cs, Cb, SH, SW, ksize, sb) ? blocks16 : 256;
  long long want = nblk / ((long long)g.csT * g.cbT * ntg);  // one block per CU (128 / 192 / 384 / 512 blocks: 6-16 % slower steps)
  if (want < 16) want = 16;
  g.G = (int)(g.total_chunks < want ? g.total_chunks : want);
}

static size_t slab_bytes(const WgradGeom& g, int ksize) {
  const int KK = ksize * ksize, NT = ksize == 3 ? KK : ksize, NTG = KK / NT;
  return sizeof(float) * (size_t)g.G * g.csT * g.cbT * NTG * gmk_wgrad_slab_floats(ksize);
}

static size_t wgrad_ws_bytes(int N, int Cs, int Cb, int SH, int SW, int ksize, int sb, int Cbias) {
  WgradGeom g{};
  { static const int dbg = getenv("GMK_WG_DBG") ? atoi(getenv("GMK_WG_DBG")) : 0; g.dbg = dbg; }
  wgrad_plan(g, N, Cs, Cb, SH, SW, ksize, sb);
  (void)Cbias;
  return slab_bytes(g, ksize) + 256;
}

// stride-1 convs with a thin input or output (<= 8 channels, e.g. the 3 image channels) take the dedicated kernel
static bool thin_route(ThinGeom& t, int N, int Cin, int H, int W, int Cout, int ksize, int stride, bool want_bias) {
  if (stride != 1 || (ksize != 3 && ksize != 5)) return false;
  if (Cin < Cout) return gmk_thin_plan(t, N, Cout, Cin, H, W, ksize, +1, want_bias);
  return gmk_thin_plan(t, N, Cin, Cout, H, W, ksize, -1, want_bias);
}

extern "C" size_t gmk_conv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize, int stride) {
  ThinGeom t{};
  if (thin_route(t, N, Cin, H, W, Cout, ksize, stride, true)) return gmk_thin_slab_bytes(t) + 256;
  return wgrad_ws_bytes(N, Cout, Cin, H / stride, W / stride, ksize, stride, Cout);
}
extern "C" size_t gmk_deconv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize) {
  return wgrad_ws_bytes(N, Cin, Cout, H, W, ksize, 2, Cout);
}

static int wgrad_run(const float* small, const float* big, float* dw, float* dbias, int bias_mode, int N, int Cs,
                     int Cb, int SH, int SW, int ksize, int sb, void* ws, size_t ws_bytes, hipStream_t s) {
  if (ksize != 3 && ksize != 5) return GMK_E_UNSUPPORTED;
  const bool wg16 = use_wg16(N, Cs, Cb, SH, SW, ksize, sb);
  if (!wg16 && !gmk_wgrad_supported(SH, SW, ksize, sb)) return GMK_E_SHAPE;  // square-ish power-of-two images, 8 <= width <= 64
  if (!gmk_aligned16(ws) || !gmk_aligned16(small) || !gmk_aligned16(big)) return GMK_E_ALIGN;
  WgradGeom g{};
  { static const int dbg = getenv("GMK_WG_DBG") ? atoi(getenv("GMK_WG_DBG")) : 0; g.dbg = dbg; }
  wgrad_plan(g, N, Cs, Cb, SH, SW, ksize, sb);
  if (ws_bytes < slab_bytes(g, ksize)) return GMK_E_WORKSPACE;
  g.s = small; g.b = big; g.slab = (float*)ws;
  g.bias_mode = dbias ? bias_mode : 0;
  if (wg16) return gmk_launch_wgrad16(g, sb, dw, dbias, s);
  return ksize == 3 ? gmk_launch_wgrad_k3(g, sb, dw, dbias, s) : gmk_launch_wgrad_k5(g, sb, dw, dbias, s);
}

extern "C" int gmk_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Cin, int H,
                                int W, int Cout, int ksize, int stride, void* ws, size_t ws_bytes,
                                gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_conv2d_wgrad, dy, x, dw, dbias, N, Cin, H, W, Cout, ksize, stride, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && x && dw && ws, GMK_E_NULL);
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2), GMK_E_SHAPE);
  ThinGeom t{};
  if (thin_route(t, N, Cin, H, W, Cout, ksize, stride, dbias != nullptr)) {
    if (!gmk_aligned16(ws) || !gmk_aligned16(dy) || !gmk_aligned16(x)) return GMK_E_ALIGN;
    if (ws_bytes < gmk_thin_slab_bytes(t)) return GMK_E_WORKSPACE;
    t.wide = t.sgn > 0 ? dy : x; t.thin = t.sgn > 0 ? x : dy; t.slab = (float*)ws;
    return gmk_launch_wgrad_thin(t, ksize, dw, dbias, (hipStream_t)stream);
  }
  return wgrad_run(dy, x, dw, dbias, 1, N, Cout, Cin, H / stride, W / stride, ksize, stride, ws, ws_bytes,
                   (hipStream_t)stream);
}

extern "C" int gmk_deconv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Cin, int H,
                                  int W, int Cout, int ksize, void* ws, size_t ws_bytes, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_deconv2d_wgrad, dy, x, dw, dbias, N, Cin, H, W, Cout, ksize, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && x && dw && ws, GMK_E_NULL);
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0, GMK_E_SHAPE);
  return wgrad_run(x, dy, dw, dbias, 2, N, Cin, Cout, H, W, ksize, 2, ws, ws_bytes, (hipStream_t)stream);
}

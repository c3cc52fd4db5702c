// conv_wgrad_api.hip -- C-ABI entry points for conv / deconv weight + bias gradients.
#include "conv_wgrad.h"
#include "plan.h"
#include "conv_wgrad16.h"
#include "conv_wgrad_thin.h"
#include <vector>

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
  if (!dw) return GMK_E_UNSUPPORTED;   // deferred reduction exists for the 3x3 kernel only (gmk_conv_wgrad_can_defer)
  return ksize == 3 ? gmk_launch_wgrad_k3(g, sb, dw, dbias, s) : gmk_launch_wgrad_k5(g, sb, dw, dbias, s);
}

// (Cs, Cb, SH, SW, sb) of a layer as the weight-gradient kernels see it: conv: small = dY; transposed conv: small = X
static void wgrad_roles(const gmk_wgrad_layer& L, int* Cs, int* Cb, int* SH, int* SW, int* sb, int* bias_mode) {
  if (L.transposed) { *Cs = L.Cin; *Cb = L.Cout; *SH = L.H; *SW = L.W; *sb = 2; *bias_mode = 2; }
  else { *Cs = L.Cout; *Cb = L.Cin; *SH = L.H / L.stride; *SW = L.W / L.stride; *sb = L.stride; *bias_mode = 1; }
}

extern "C" int gmk_conv_wgrad_can_defer(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int transposed) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2) || (transposed && stride != 2)) return 0;
  ThinGeom t{};
  if (!transposed && thin_route(t, N, Cin, H, W, Cout, ksize, stride, true)) return 0;
  gmk_wgrad_layer L{nullptr, nullptr, nullptr, N, Cin, H, W, Cout, ksize, stride, transposed};
  int Cs, Cb, SH, SW, sb, bm;
  wgrad_roles(L, &Cs, &Cb, &SH, &SW, &sb, &bm);
  return use_wg16(N, Cs, Cb, SH, SW, ksize, sb) ? 1 : 0;
}

extern "C" int gmk_conv_wgrad_reduce_batch(const gmk_wgrad_layer* layers, int n, gmk_stream_t stream) {
  GMK_REQUIRE(layers, GMK_E_NULL);
  GMK_REQUIRE(n >= 1 && n <= GMK_WG16_BATCH, GMK_E_SHAPE);
  if (gmk_plan_detail::recording()) {   // `layers` is host memory of the caller: the plan keeps its own copy
    std::vector<gmk_wgrad_layer> copy(layers, layers + n);
    gmk_plan_detail::push([copy, n, stream](const gmk_plan_detail::Subst& sub) {
      std::vector<gmk_wgrad_layer> c = copy;
      for (auto& L : c) { L.ws = sub.map(L.ws); L.dw = (float*)sub.map(L.dw); L.dbias = (float*)sub.map(L.dbias); }
      return gmk_conv_wgrad_reduce_batch(c.data(), n, (gmk_stream_t)sub.map(stream));
    });
  }
  Wg16ReduceBatch b{};
  int max_ny = 0;
  for (int i = 0; i < n; ++i) {
    const gmk_wgrad_layer& L = layers[i];
    GMK_REQUIRE(L.ws && L.dw, GMK_E_NULL);
    GMK_REQUIRE(gmk_conv_wgrad_can_defer(L.N, L.Cin, L.H, L.W, L.Cout, L.ksize, L.stride, L.transposed), GMK_E_UNSUPPORTED);
    if (!gmk_aligned16(L.ws)) return GMK_E_ALIGN;
    int Cs, Cb, SH, SW, sb, bm;
    wgrad_roles(L, &Cs, &Cb, &SH, &SW, &sb, &bm);
    WgradGeom g{};
    wgrad_plan(g, L.N, Cs, Cb, SH, SW, L.ksize, sb);
    Wg16ReduceItem& it = b.item[i];
    it.slab = (const float*)L.ws; it.dw = L.dw; it.dbias = L.dbias;
    it.G = g.G; it.ny = g.csT * g.cbT; it.Cs = Cs; it.Cb = Cb; it.cbT = g.cbT; it.bias_mode = L.dbias ? bm : 0;
    if (it.ny > max_ny) max_ny = it.ny;
  }
  return gmk_launch_wgrad16_reduce_batch(b, n, max_ny, (hipStream_t)stream);
}

extern "C" int gmk_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Cin, int H,
                                int W, int Cout, int ksize, int stride, void* ws, size_t ws_bytes,
                                gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_conv2d_wgrad, dy, x, dw, dbias, N, Cin, H, W, Cout, ksize, stride, ws, ws_bytes, stream);
  GMK_REQUIRE(dy && x && ws, GMK_E_NULL);   // dw == NULL: partial slabs only, where gmk_conv_wgrad_can_defer() says so
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2), GMK_E_SHAPE);
  ThinGeom t{};
  if (thin_route(t, N, Cin, H, W, Cout, ksize, stride, dbias != nullptr)) {
    GMK_REQUIRE(dw, GMK_E_UNSUPPORTED);
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
  GMK_REQUIRE(dy && x && ws, GMK_E_NULL);   // dw == NULL: partial slabs only (deferred reduction)
  GMK_REQUIRE(N > 0 && Cin > 0 && Cout > 0, GMK_E_SHAPE);
  return wgrad_run(x, dy, dw, dbias, 2, N, Cin, Cout, H, W, ksize, 2, ws, ws_bytes, (hipStream_t)stream);
}

// conv_api.hip -- C-ABI entry points for conv / transposed-conv forward and data gradient.
// Every variant is one launch of conv_igemm_kernel with a different weight view / mode:
//   conv fwd s1 -> MODE0, s2 -> MODE1 ; conv dgrad s1 -> MODE0 with flipped taps and [c][m] weights,
//   s2 -> MODE2 ; deconv fwd -> MODE2 ; deconv dgrad -> MODE1.
#include "conv_wino.h"
#include "plan.h"
#include <stdlib.h>
#include <mutex>

// conv_thin_out.hip: forward of stride-1 convs with <= 8 output channels (dec.conv3)
bool gmk_thin_out_ok(int N, int Cin, int H, int W, int Cout, int ksize, int stride);
int gmk_launch_conv_thin_out(const float* x, const float* w, const float* bias, float* y, float* dact, int N, int Cin,
                             int H, int W, int Cout, int ksize, int act, hipStream_t s);

// conv_thin_in.hip: stride-1 3x3 convs with <= 4 input channels (enc.conv1 forward, dec.conv3 data gradient)
bool gmk_thin_in_ok(const ConvGeom& g, int ksize, int mode);
int gmk_launch_conv_thin_in(const ConvGeom& g, hipStream_t s);

// conv_thin_outv.hip: forward of stride-1 3x3 convs with <= 4 output channels on the vector ALUs (dec.conv3)
bool gmk_thin_outv_ok(int N, int Cin, int H, int W, int Cout, int ksize, int stride);
int gmk_launch_conv_thin_outv(const float* x, const float* w, const float* bias, float* y, float* dact, int N, int Cin,
                              int H, int W, int Cout, int act, hipStream_t s);

// Kernel-path selection of the <= 4-channel-side 3x3 layers (conv_thin_in.hip, conv_thin_outv.hip).  This is the one piece
// of mutable state behind the conv entry points, and it is PER DEVICE (the faster choice differs between devices of the
// same model, gmk/tune.py): one atomic word per device, -1 = not set yet (first use reads GMK_THIN_IN / GMK_THIN_OUTV,
// default on), read once per launch with the launching thread's current device; gmk_conv_set_thin_paths sets the
// current device's word.  Atomic, so a setter racing a launch on another thread yields one of the two valid choices.
#include <atomic>
#define GMK_THIN_MAX_DEVICES 64
static std::atomic<int> g_thin_in[GMK_THIN_MAX_DEVICES], g_thin_outv[GMK_THIN_MAX_DEVICES];
static std::atomic<bool> g_thin_init{false};
static int thin_dev() {
  if (!g_thin_init.load(std::memory_order_acquire)) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!g_thin_init.load(std::memory_order_relaxed)) {
      const int in0 = getenv("GMK_THIN_IN") ? (atoi(getenv("GMK_THIN_IN")) & 3) : 3;   // bit 0: forward, bit 1: data gradient
      const int ov0 = getenv("GMK_THIN_OUTV") ? (atoi(getenv("GMK_THIN_OUTV")) != 0) : 1;
      for (int d = 0; d < GMK_THIN_MAX_DEVICES; ++d) { g_thin_in[d].store(in0); g_thin_outv[d].store(ov0); }
      g_thin_init.store(true, std::memory_order_release);
    }
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GMK_THIN_MAX_DEVICES) dev = 0;
  return dev;
}
static int thin_in_mask() { const int d = thin_dev(); return g_thin_in[d].load(std::memory_order_relaxed); }
static int thin_outv_on() { const int d = thin_dev(); return g_thin_outv[d].load(std::memory_order_relaxed); }

extern "C" int gmk_conv_set_thin_paths(int thin_in, int thin_outv) {
  GMK_REQUIRE(thin_in <= 3 && thin_outv <= 1, GMK_E_UNSUPPORTED);
  const int d = thin_dev();
  if (thin_in >= 0) g_thin_in[d].store(thin_in);
  if (thin_outv >= 0) g_thin_outv[d].store(thin_outv);
  return GMK_OK;
}

extern "C" int gmk_conv_get_thin_paths(int* thin_in, int* thin_outv) {
  GMK_REQUIRE(thin_in && thin_outv, GMK_E_NULL);
  *thin_in = thin_in_mask();
  *thin_outv = thin_outv_on();
  return GMK_OK;
}

static int conv_common_checks(int N, int Cin, int H, int W, int Cout, int ksize) {
  if (N <= 0 || Cin <= 0 || Cout <= 0) return GMK_E_SHAPE;
  if (!gmk_is_pow2(H) || !gmk_is_pow2(W) || H < 4 || W < 4 || W > 128) return GMK_E_SHAPE;
  if (ksize != 3 && ksize != 5) return GMK_E_UNSUPPORTED;
  return GMK_OK;
}

static int dispatch(ConvGeom& g, int ksize, int mode, hipStream_t s) {
  static const int dbg = getenv("GMK_CONV_DBG") ? atoi(getenv("GMK_CONV_DBG")) : 0;  // profiling ablations only
  g.dbg = dbg;
  // <= 4 input channels (the image side of the autoencoder): output-stream bound, no GEMM (conv_thin_in.hip); GMK_THIN_IN=0
  // keeps the implicit GEMM (A/B runs)
  if ((thin_in_mask() & (g.epi == 0 ? 1 : 2)) && !dbg && gmk_thin_in_ok(g, ksize, mode)) return gmk_launch_conv_thin_in(g, s);
  // stride-1 3x3 layers whose grid fills the chip (>= GMK_WINO_MIN_BLOCKS blocks of 32 tiles x 64 channels): Winograd
  // F(2x2,3x3) (conv_wino.h).  GMK_WINO: bit 0 = forward, bit 1 = data gradient; 0 keeps the implicit GEMM everywhere
  static const int wino_mask = getenv("GMK_WINO") ? atoi(getenv("GMK_WINO")) : 3;
  const int wino = wino_mask & (g.epi == 0 ? 1 : 2);
  static const int minb = getenv("GMK_WINO_MIN_BLOCKS") ? atoi(getenv("GMK_WINO_MIN_BLOCKS")) : 256;
  if (wino && !dbg && gmk_wino_ok(g, ksize, mode) && (g.flip || gmk_wino_fwd_ready()) &&
      ((long long)g.N * g.IH * g.IW / 4 / WINO_TB) * ((g.M + WINO_MB - 1) / WINO_MB) >= minb) {
    static const int wdbg = getenv("GMK_WINO_DBG") ? atoi(getenv("GMK_WINO_DBG")) : 0;   // profiling ablations only
    g.dbg = wdbg;
    return gmk_launch_conv_wino(g, s);
  }
  return ksize == 3 ? gmk_launch_conv_k3(g, mode, s) : gmk_launch_conv_k5(g, mode, s);
}

extern "C" int gmk_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* dact, int N,
                              int Cin, int H, int W, int Cout, int ksize, int stride, int act, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_conv2d_fwd, x, w, bias, y, dact, N, Cin, H, W, Cout, ksize, stride, act, stream);
  GMK_REQUIRE(x && w && y, GMK_E_NULL);
  int rc = conv_common_checks(N, Cin, H, W, Cout, ksize);
  if (rc) return rc;
  GMK_REQUIRE(stride == 1 || stride == 2, GMK_E_UNSUPPORTED);
  GMK_REQUIRE(act >= GMK_ACT_NONE && act <= GMK_ACT_SIGMOID, GMK_E_UNSUPPORTED);
  if (thin_outv_on() && gmk_thin_outv_ok(N, Cin, H, W, Cout, ksize, stride))
    return gmk_launch_conv_thin_outv(x, w, bias, y, dact, N, Cin, H, W, Cout, act, (hipStream_t)stream);
  if (gmk_thin_out_ok(N, Cin, H, W, Cout, ksize, stride))
    return gmk_launch_conv_thin_out(x, w, bias, y, dact, N, Cin, H, W, Cout, ksize, act, (hipStream_t)stream);
  const int KK = ksize * ksize;
  ConvGeom g{};
  g.x = x; g.w = w; g.bias = bias; g.dact_in = nullptr; g.y = y; g.dact_out = dact;
  g.N = N; g.C = Cin; g.M = Cout; g.IH = H; g.IW = W; g.OH = H / stride; g.OW = W / stride;
  g.BW = g.OW; g.lgBW = gmk_ilog2(g.BW); g.lgBHW = gmk_ilog2((int64_t)g.OH * g.OW);
  g.total_base = (long long)N * g.OH * g.OW;
  g.w_layout = 0; g.w_m_stride = Cin * KK; g.w_c_stride = KK; g.flip = 0; g.act = act; g.epi = 0;
  g.w_numel = Cout * Cin * KK;
  return dispatch(g, ksize, stride == 1 ? 0 : 1, (hipStream_t)stream);
}

extern "C" int gmk_conv2d_dgrad(const float* dy, const float* w, const float* dact_in, float* dx, int N, int Cin,
                                int H, int W, int Cout, int ksize, int stride, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_conv2d_dgrad, dy, w, dact_in, dx, N, Cin, H, W, Cout, ksize, stride, stream);
  GMK_REQUIRE(dy && w && dx, GMK_E_NULL);
  int rc = conv_common_checks(N, Cin, H, W, Cout, ksize);
  if (rc) return rc;
  GMK_REQUIRE(stride == 1 || stride == 2, GMK_E_UNSUPPORTED);
  const int KK = ksize * ksize;
  ConvGeom g{};
  g.x = dy; g.w = w; g.bias = nullptr; g.dact_in = dact_in; g.y = dx; g.dact_out = nullptr;
  g.N = N; g.C = Cout; g.M = Cin; g.IH = H / stride; g.IW = W / stride; g.OH = H; g.OW = W;
  g.BW = W / stride; g.lgBW = gmk_ilog2(g.BW); g.lgBHW = gmk_ilog2((int64_t)g.IH * g.IW);
  g.total_base = (long long)N * g.IH * g.IW;
  g.w_layout = 1; g.w_m_stride = KK; g.w_c_stride = Cin * KK; g.flip = stride == 1 ? 1 : 0; g.act = 0; g.epi = 1;
  g.w_numel = Cout * Cin * KK;
  return dispatch(g, ksize, stride == 1 ? 0 : 2, (hipStream_t)stream);
}

extern "C" int gmk_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* dact, int N,
                                int Cin, int H, int W, int Cout, int ksize, int act, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_deconv2d_fwd, x, w, bias, y, dact, N, Cin, H, W, Cout, ksize, act, stream);
  GMK_REQUIRE(x && w && y, GMK_E_NULL);
  int rc = conv_common_checks(N, Cin, H, W, Cout, ksize);
  if (rc) return rc;
  GMK_REQUIRE(W <= 64, GMK_E_SHAPE);
  GMK_REQUIRE(act >= GMK_ACT_NONE && act <= GMK_ACT_SIGMOID, GMK_E_UNSUPPORTED);
  const int KK = ksize * ksize;
  ConvGeom g{};
  g.x = x; g.w = w; g.bias = bias; g.dact_in = nullptr; g.y = y; g.dact_out = dact;
  g.N = N; g.C = Cin; g.M = Cout; g.IH = H; g.IW = W; g.OH = 2 * H; g.OW = 2 * W;
  g.BW = W; g.lgBW = gmk_ilog2(W); g.lgBHW = gmk_ilog2((int64_t)H * W);
  g.total_base = (long long)N * H * W;
  g.w_layout = 1; g.w_m_stride = KK; g.w_c_stride = Cout * KK; g.flip = 0; g.act = act; g.epi = 0;
  g.w_numel = Cout * Cin * KK;
  return dispatch(g, ksize, 2, (hipStream_t)stream);
}

extern "C" int gmk_deconv2d_dgrad(const float* dy, const float* w, const float* dact_in, float* dx, int N, int Cin,
                                  int H, int W, int Cout, int ksize, gmk_stream_t stream) {
  GMK_PLAN_HOOK(gmk_deconv2d_dgrad, dy, w, dact_in, dx, N, Cin, H, W, Cout, ksize, stream);
  GMK_REQUIRE(dy && w && dx, GMK_E_NULL);
  int rc = conv_common_checks(N, Cin, H, W, Cout, ksize);
  if (rc) return rc;
  GMK_REQUIRE(W <= 64, GMK_E_SHAPE);
  const int KK = ksize * ksize;
  ConvGeom g{};
  g.x = dy; g.w = w; g.bias = nullptr; g.dact_in = dact_in; g.y = dx; g.dact_out = nullptr;
  g.N = N; g.C = Cout; g.M = Cin; g.IH = 2 * H; g.IW = 2 * W; g.OH = H; g.OW = W;
  g.BW = W; g.lgBW = gmk_ilog2(W); g.lgBHW = gmk_ilog2((int64_t)H * W);
  g.total_base = (long long)N * H * W;
  g.w_layout = 0; g.w_m_stride = Cout * KK; g.w_c_stride = KK; g.flip = 0; g.act = 0; g.epi = 1;
  g.w_numel = Cout * Cin * KK;
  return dispatch(g, ksize, 1, (hipStream_t)stream);
}

#ifdef GMK_CONV_TRACE
// tools-only build (make EXTRA=-DGMK_CONV_TRACE): per-block timestamps of the LAST implicit-GEMM launch
static unsigned long long* g_trace = nullptr;
unsigned long long* gmk_conv_trace_buffer() {
  if (!g_trace) { if (hipMalloc((void**)&g_trace, 12 * 8 * 65536) != hipSuccess) g_trace = nullptr; }
  return g_trace;
}
extern "C" int gmk_debug_conv_trace_read(unsigned long long* dst, int nblocks) {
  if (!g_trace || nblocks > 65536) return -1;
  return (int)hipMemcpy(dst, g_trace, (size_t)nblocks * 32, hipMemcpyDeviceToHost);
}
extern "C" int gmk_debug_conv_trace_read_epi(unsigned long long* dst, int nblocks) {   // per block: 4 epilogue stamps (+4 unused)
  if (!g_trace || nblocks > 65536) return -1;
  return (int)hipMemcpy(dst, g_trace + 4ull * 65536, (size_t)nblocks * 64, hipMemcpyDeviceToHost);
}
#endif

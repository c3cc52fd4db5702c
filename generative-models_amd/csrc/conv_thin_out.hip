// conv_thin_out.hip -- stride-1 conv forward with very few OUTPUT channels (dec.conv3: C -> 3).
// The generic implicit GEMM pads the 3 output channels to a 32-wide MFMA tile (10x wasted work, 90 us); here the
// GEMM is turned round: N = (output channel, tap) columns (3*9 = 27 of 32), K = input channels,
//
//   part[p][(co,t)] = sum_c x[c][p] * w[co][c][t]          (MFMA: A = pixels x channels straight from global
//                                                            memory as 128-byte rows, B = weights held in registers)
//   y[co][p]        = bias[co] + sum_t part[p + off(t)][(co,t)]   (shifted gather through LDS)
//
// A block owns 256 output pixels (whole image rows) and computes `part` on them plus one halo row on each side
// (rows outside the image contribute zeros).  HBM-bound: one pass over x (+25 % halo re-reads that hit L2).
#include "gmk_common.h"

struct ThinOutGeom {
  const float* x; const float* w; const float* bias; float* y; float* dact;
  int N, C, Co, H, W, lgW, lgHW;
  int TR;        // output rows per block (TR * W = 256)
  int act;
};

template <int KS, int NB>
__global__ __launch_bounds__(256) void conv_thin_out_kernel(const ThinOutGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float P[];  // [NB*32 cols][PS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int W = g.W, HW = 1 << g.lgHW;
  const int rows_ext = g.TR + 2 * PAD;
  const int pxh = rows_ext * W;   // pixels incl. halo rows (multiple of 32: W >= 16 and TR*W = 256)
  const int PS = pxh + 4;
  const int ngroups = pxh / 32;
  const int tiles_per_img = g.H / g.TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x % tiles_per_img) * g.TR;  // first output row of the tile
  const int ncol = g.Co * KK;
  const __amdgpu_buffer_rsrc_t rx = gmk_make_rsrc(g.x + (long long)n * g.C * HW, (long long)g.C * HW * 4);

  f32x16 acc[3][NB];  // up to 3 pixel groups per wave (ngroups <= 12)
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][nb][r] = 0.f;

  for (int c0 = 0; c0 < g.C; c0 += 64) {
    // weights of this channel chunk -> registers: lane (col j, parity h) holds w[col][c0 + 2s + h], s = 0..31
    float wreg[NB][32];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = nb * 32 + j;
      const int co = col / KK, t = col % KK;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int c = c0 + 2 * s + h;
        wreg[nb][s] = (col < ncol && c < g.C) ? g.w[((size_t)co * g.C + c) * KK + t] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int grp = wave + 4 * q;
      if (grp >= ngroups) break;  // uniform
      // 32 consecutive pixels of the halo-extended tile; lane j takes pixel grp*32 + j, parity h the odd channel
      const int pe = grp * 32 + j;
      const int iy = r0 - PAD + (pe >> g.lgW), ix = pe & (W - 1);
      const bool ok = (iy >= 0) & (iy < g.H);
      const int voff = ok ? ((h << g.lgHW) + iy * W + ix) * 4 : GMK_OOB;
      float xa[32];
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const bool cok = c0 + 2 * s + 1 < g.C || (c0 + 2 * s < g.C && h == 0);  // ragged channel tail
        xa[s] = gmk_buffer_load(rx, cok ? voff : GMK_OOB, (c0 + 2 * s) * HW * 4);
      }
#pragma unroll
      for (int s = 0; s < 32; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[q][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], wreg[nb][s], acc[q][nb], 0, 0, 0);
    }
  }
  // partial sums -> LDS: lane -> column, register quad -> 4 consecutive pixels
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int grp = wave + 4 * q;
    if (grp >= ngroups) break;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        *(float4*)(P + (nb * 32 + j) * PS + grp * 32 + 8 * k4 + 4 * h) =
            make_float4(acc[q][nb][4 * k4], acc[q][nb][4 * k4 + 1], acc[q][nb][4 * k4 + 2], acc[q][nb][4 * k4 + 3]);
  }
  __syncthreads();
  // shifted gather: output pixel p = tid of the tile, every output channel
  const int p = tid;
  const int px = p & (W - 1);
  const int qc = p + PAD * W;  // index of p in the halo-extended tile
  for (int co = 0; co < g.Co; ++co) {
    float v = g.bias ? g.bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const int dy = t / KS - PAD, dx = t % KS - PAD;
      const bool okx = (px + dx >= 0) & (px + dx < W);
      const float pv = P[(co * KK + t) * PS + qc + dy * W + (okx ? dx : 0)];
      v += okx ? pv : 0.f;
    }
    float a, d;
    gmk_act_fwd(g.act, v, a, d);
    const size_t idx = ((size_t)n * g.Co + co) * HW + (size_t)r0 * W + p;
    g.y[idx] = a;
    if (g.dact) g.dact[idx] = d;
  }
}

bool gmk_thin_out_ok(int N, int Cin, int H, int W, int Cout, int ksize, int stride) {
  const int KK = ksize * ksize;
  if (stride != 1 || (ksize != 3 && ksize != 5)) return false;
  if (Cout * KK > 96 || Cout >= Cin || Cout > 8) return false;
  if (!gmk_is_pow2(H) || !gmk_is_pow2(W) || W < 16 || W > 256 || (long long)H * W < 256) return false;
  const int TR = 256 / W, PAD = (ksize - 1) / 2;
  if ((TR + 2 * PAD) * W / 32 > 12) return false;  // <= 3 pixel groups per wave
  return true;
}

template <int KS, int NB>
static int launch_thin_out(const ThinOutGeom& g, hipStream_t s) {
  constexpr int PAD = (KS - 1) / 2;
  const int pxh = (g.TR + 2 * PAD) * g.W;
  const size_t lds = sizeof(float) * (size_t)NB * 32 * (pxh + 4);
  if (lds > 160 * 1024) return GMK_E_SHAPE;
  auto kern = conv_thin_out_kernel<KS, NB>;
  if (lds > 64 * 1024) {
    hipError_t e = gmk_ensure_dyn_lds((const void*)kern, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(g.N * (g.H / g.TR)), dim3(256), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

int gmk_launch_conv_thin_out(const float* x, const float* w, const float* bias, float* y, float* dact, int N, int Cin,
                             int H, int W, int Cout, int ksize, int act, hipStream_t s) {
  ThinOutGeom g{};
  g.x = x; g.w = w; g.bias = bias; g.y = y; g.dact = dact;
  g.N = N; g.C = Cin; g.Co = Cout; g.H = H; g.W = W; g.lgW = gmk_ilog2(W); g.lgHW = gmk_ilog2((int64_t)H * W);
  g.TR = 256 / W; g.act = act;
  const int nb = (Cout * ksize * ksize + 31) / 32;
  if (ksize == 3) return nb == 1 ? launch_thin_out<3, 1>(g, s) : nb == 2 ? launch_thin_out<3, 2>(g, s) : launch_thin_out<3, 3>(g, s);
  return nb == 1 ? launch_thin_out<5, 1>(g, s) : nb == 2 ? launch_thin_out<5, 2>(g, s) : launch_thin_out<5, 3>(g, s);
}

// conv_thin_outv.hip -- stride-1 3x3 conv forward with <= 4 OUTPUT channels (dec.conv3: C -> 3) on the vector ALUs.
// The op reads a C-channel map once (67 MB at 32x32, B = 256) and writes almost nothing: 27 FMAs per input value, bound
// by the input stream.  conv_thin_out.hip turns it into an MFMA GEMM with a shifted gather through 41 KB of LDS per
// block (3 blocks per CU, 1024 blocks: 32 us = 2 TB/s); here, as in conv_thin_in.hip, there is no GEMM:
//
//   block  = 256 consecutive output pixels of one image (whole rows); wave w takes input channels [w C/4, (w+1) C/4)
//   lane   = 4 consecutive pixels: per channel three 16-byte loads (rows y-1, y, y+1; the row overlap between lanes is
//            served by the L1), the left / right neighbour values come from lanes -1 / +1 by DPP (zero at the image edge)
//   math   = packed FMAs on output-channel pairs (co0, co1), (co2, co3): weights from LDS as one 16-byte broadcast read
//            per (channel, tap), the pixel value broadcast by operand select
//   tail   = the four waves' partial sums meet in LDS; bias, activation (+ act') and one 16-byte store per lane.
// Plain fp32 FMA chains; summation order (channel quarter, channel, tap) differs from the MFMA path, same fp32 rounding.
#include "gmk_common.h"

#ifndef THIN_OUTV_TG
#define THIN_OUTV_TG 1   /* channels per register buffer: 1 / 2 / 4 measured 22.1 / 22.7 / 23.6 us */
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

struct ThinOutVGeom {
  const float* x; const float* w; const float* bias; float* y; float* dact;
  int N, C, Co, H, W, lgW, lgHW;
  int act;
};

// acc += w * x[HI] on both halves (see conv_thin_in.hip)
template <int HI>
__device__ __forceinline__ void thin_outv_fma(v2f& acc, const v2f w, const v2f x) {
  if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(x));
}

__global__ __launch_bounds__(256) void conv_thin_outv_kernel(const ThinOutVGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                  // [C][9][4]: (co0..co3) of one (channel, tap); later [wave][co][lane] float4 partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the channel-range guards below are uniform branches
  const int W = g.W, H = g.H, HW = 1 << g.lgHW;
  const int tiles = HW >> 8;
  // Blocks go round-robin over the 8 XCDs (one L2 each): hand XCD x the tiles [x nb/8, (x+1) nb/8), so that the halo rows
  // a tile shares with its neighbours (2 of 8 rows at 32x32: 25 % more HBM reads under PMC without this) hit in that L2
  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int n = b / tiles, px0 = (b % tiles) << 8;

  const int p = px0 + 4 * lane, r = p >> g.lgW, x0 = p & (W - 1);
  const bool up = r > 0, down = r < H - 1;
  const bool lok = x0 > 0, rok = x0 + 4 < W;   // a left / right neighbour inside the image (lane -1 / +1 holds it: W <= 64)
  const int cpw = g.C >> 2;   // C % (8 TG) == 0: every wave runs whole double groups
  const int c_lo = wave * cpw, c_hi = c_lo + cpw;
  const float* xp = g.x + ((long long)n * g.C + c_lo) * HW + p;

  v2f acc[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = (v2f){0.f, 0.f};

  // Channels go in groups of TG through two register buffers: the loads of group i + 1 are in flight under the FMAs of
  // group i.  Rows outside the image are loaded from the centre row's address and zeroed by a select (no branch).
  constexpr int TG = THIN_OUTV_TG;
  const int offU = up ? W : 0, offD = down ? W : 0;
  auto load_group = [&](float4 (&buf)[TG][3], int c) {
#pragma unroll
    for (int gi = 0; gi < TG; ++gi) {
      // no guard (a branch here makes the compiler wait for every load in flight at the join): the group past the
      // wave's last one is clamped to the tensor's last channels and never used
      const int cc = c + gi < g.C ? c + gi : g.C - TG + gi;
      const float* q = xp + (long long)(cc - c_lo) * HW;
      buf[gi][0] = *(const float4*)(q - offU);
      buf[gi][1] = *(const float4*)q;
      buf[gi][2] = *(const float4*)(q + offD);
    }
  };
  auto fma_group = [&](float4 (&buf)[TG][3], int c) {
#pragma unroll
    for (int gi = 0; gi < TG; ++gi) {
      const float4* wq = (const float4*)(Wl + (c + gi) * 36);
      gmk_static_for<0, 3>([&](auto ky_) {
        constexpr int ky = decltype(ky_)::value;
        float4 a = buf[gi][ky];
        if (ky == 0 && !up) a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ky == 2 && !down) a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int li = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a.w), 0x111, 0xf, 0xf, true);   // lane - 1
        const int ri = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a.x), 0x101, 0xf, 0xf, true);   // lane + 1
        const v2f lr = (v2f){lok ? __builtin_bit_cast(float, li) : 0.f, rok ? __builtin_bit_cast(float, ri) : 0.f};
        const v2f A = (v2f){a.x, a.y}, B = (v2f){a.z, a.w};
        gmk_static_for<0, 3>([&](auto kx_) {
          constexpr int kx = decltype(kx_)::value;
          const float4 w4 = wq[ky * 3 + kx];
          const v2f w01 = (v2f){w4.x, w4.y}, w23 = (v2f){w4.z, w4.w};
          gmk_static_for<0, 4>([&](auto j_) {
            constexpr int j = decltype(j_)::value, i = j + kx;   // window value i: [left, a.x, a.y, a.z, a.w, right]
            if constexpr (i == 0) { thin_outv_fma<0>(acc[0][j], w01, lr); thin_outv_fma<0>(acc[1][j], w23, lr); }
            else if constexpr (i == 5) { thin_outv_fma<1>(acc[0][j], w01, lr); thin_outv_fma<1>(acc[1][j], w23, lr); }
            else if constexpr (i <= 2) { thin_outv_fma<(i - 1) & 1>(acc[0][j], w01, A); thin_outv_fma<(i - 1) & 1>(acc[1][j], w23, A); }
            else { thin_outv_fma<(i - 3) & 1>(acc[0][j], w01, B); thin_outv_fma<(i - 3) & 1>(acc[1][j], w23, B); }
          });
        });
      });
    }
  };
  float4 bufA[TG][3], bufB[TG][3];
  load_group(bufA, c_lo);   // in flight under the weight staging
  for (int e = tid; e < g.C * 36; e += 256) {
    const int co = e & 3, ct = e >> 2;   // ct = c * 9 + t
    Wl[e] = co < g.Co ? g.w[(long long)co * g.C * 9 + ct] : 0.f;
  }
  __syncthreads();
  for (int c = c_lo; c < c_hi; c += 2 * TG) {
    load_group(bufB, c + TG);
    fma_group(bufA, c);
    load_group(bufA, c + 2 * TG);
    fma_group(bufB, c + TG);
  }

  // ---- the four channel quarters meet in LDS: Pl[wave][co][lane] (float4 = this lane's 4 pixels) ----
  __syncthreads();   // everybody is done with the weights
  float4* Pl = (float4*)smem;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      Pl[(wave * 4 + 2 * q + hh) * 64 + lane] = make_float4(acc[q][0][hh], acc[q][1][hh], acc[q][2][hh], acc[q][3][hh]);
  __syncthreads();
  const int co = wave;   // wave w finishes output channel w
  if (co < g.Co) {
    float4 s = Pl[co * 64 + lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 t = Pl[(k * 4 + co) * 64 + lane];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float bv = g.bias ? g.bias[co] : 0.f;
    float4 a, d;
    gmk_act_fwd(g.act, s.x + bv, a.x, d.x);
    gmk_act_fwd(g.act, s.y + bv, a.y, d.y);
    gmk_act_fwd(g.act, s.z + bv, a.z, d.z);
    gmk_act_fwd(g.act, s.w + bv, a.w, d.w);
    const long long idx = ((long long)n * g.Co + co) * HW + p;
    *(float4*)(g.y + idx) = a;
    if (g.dact) *(float4*)(g.dact + idx) = d;
  }
}

bool gmk_thin_outv_ok(int N, int Cin, int H, int W, int Cout, int ksize, int stride) {
  if (stride != 1 || ksize != 3) return false;
  if (Cout > 4 || Cin < 16 || Cin % (8 * THIN_OUTV_TG) != 0) return false;   // a wave takes Cin / 4 channels in double groups
  if (!gmk_is_pow2(H) || !gmk_is_pow2(W) || W < 16 || W > 64 || (long long)H * W < 256) return false;
  if ((long long)N * (H * W / 256) > 0x7FFFFFFFll) return false;
  if (sizeof(float) * (size_t)Cin * 36 > 64 * 1024) return false;   // the weight table must fit the default LDS window (Cin <= 455): wider layers take conv_thin_out / the implicit GEMM
  return true;
}

int gmk_launch_conv_thin_outv(const float* x, const float* w, const float* bias, float* y, float* dact, int N, int Cin,
                              int H, int W, int Cout, int act, hipStream_t s) {
  ThinOutVGeom g{};
  g.x = x; g.w = w; g.bias = bias; g.y = y; g.dact = dact;
  g.N = N; g.C = Cin; g.Co = Cout; g.H = H; g.W = W; g.lgW = gmk_ilog2(W); g.lgHW = gmk_ilog2((int64_t)H * W);
  g.act = act;
  size_t lds = sizeof(float) * (size_t)Cin * 36;
  if (lds < 16 * 1024) lds = 16 * 1024;   // the partial sums: 4 waves x 4 channels x 64 lanes x 16 bytes
  if (lds > 64 * 1024) return GMK_E_SHAPE;
  hipLaunchKernelGGL(conv_thin_outv_kernel, dim3((unsigned)(N * (H * W / 256))), dim3(256), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

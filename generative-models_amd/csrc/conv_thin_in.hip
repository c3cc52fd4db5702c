// conv_thin_in.hip -- stride-1 3x3 conv with very few INPUT channels (<= 4): enc.conv1 forward (3 -> C) and the data
// gradient of dec.conv3 (3 -> C, flipped taps, times act').  Both write a C-channel 32x32 map and read almost nothing:
// 9 * 3 = 27 multiplies per output against 8 bytes (forward: y and act') or 12 bytes (dgrad: act' in, dx out) of HBM
// traffic, so the op is bound by the output stream.  The implicit GEMM pads K = 27 to 36, runs its K loop and its
// epilogue one after the other and holds one block per CU (38 us for 137 MB at B = 256); here there is no GEMM at all:
//
//   block  = 1024 consecutive output pixels of one image (whole rows) x THIN_IN_MB output channels
//   thread = 4 consecutive pixels; its 3 x 6 input window per channel sits in registers for the whole block
//   loop   = output channel PAIRS: weights (and the bias) come from LDS as (w[m], w[m+1]) pairs read at a wave-uniform
//            address (broadcast), the 27 x 4 FMAs of the pair are packed (v_pk_fma_f32: two channels per lane and
//            instruction), then bias / activation (+ act') or the act' multiply, and one 16-byte store per lane, i.e.
//            1 KiB contiguous per wave and instruction.
// Plain fp32 FMA chains in (channel, tap) order -- the same arithmetic as the MFMA path up to summation order.
#include "conv_igemm.h"

#ifndef THIN_IN_MB
#define THIN_IN_MB 16   /* output channels per block */
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

// acc += w * x[HI] on both halves: the packed FMA with the pixel value broadcast by operand select (the compiler would
// materialise the (x, x) pair in two more registers per value: 108 instead of 54 for the window)
template <int HI>
__device__ __forceinline__ void thin_in_fma(v2f& acc, const v2f w, const v2f x) {
  if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(x));
}

template <int C, bool DGRAD>
__global__ __launch_bounds__(256) void conv_thin_in_kernel(const ConvGeom g) {
  constexpr int KC = C * 9;
  constexpr int KP = (KC + 1 + 1) & ~1;   // per channel pair: KC weights + the bias, as (m, m+1) float pairs; even for b128 reads
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int W = g.IW, H = g.IH, lgW = g.lgBW;
  const int R = 1024 >> lgW;              // rows per block
  const int WS = W + 4;                   // padded row: column x at index x + 1, zero columns at 0 and W + 1
  const int PS = (R + 2) * WS;
  float* Wl = smem;                       // [MB / 2][KP][2]
  float* Xl = smem + THIN_IN_MB * KP;     // [C][R + 2][WS]
  const int tid = threadIdx.x;
  const int tiles = H / R;
  const int n = blockIdx.x / tiles, r0 = (blockIdx.x % tiles) * R;
  const int m0 = blockIdx.y * THIN_IN_MB;
  const long long HW = (long long)H * W;

  // ---- weights (tap order of the gather: flipped for the data gradient) and bias -> LDS ----
  for (int e = tid; e < THIN_IN_MB * (KC + 1); e += 256) {
    const int mi = e / (KC + 1), k = e % (KC + 1);
    const int m = m0 + mi;
    float v = 0.f;
    if (m < g.M) {
      if (k < KC) {
        const int c = k / 9, t = k % 9;
        v = g.w[(long long)m * g.w_m_stride + (long long)c * g.w_c_stride + (DGRAD ? 8 - t : t)];
      } else if (!DGRAD && g.bias) {
        v = g.bias[m];
      }
    }
    Wl[((mi >> 1) * KP + k) * 2 + (mi & 1)] = v;
  }
  // ---- input rows r0 - 1 .. r0 + R (zero outside the image) -> LDS ----
  const int lgQ = lgW - 2, nq = (R + 2) << lgQ;   // float4 quads per row, quads per channel
  const float* xn = g.x + (long long)n * C * HW;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    for (int e = tid; e < nq; e += 256) {
      const int pr = e >> lgQ, xq = e & ((1 << lgQ) - 1);
      const int iy = r0 - 1 + pr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < H) v = *(const float4*)(xn + c * HW + (long long)iy * W + 4 * xq);
      float* d = Xl + c * PS + pr * WS + 1 + 4 * xq;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (tid < R + 2) {
      Xl[c * PS + tid * WS] = 0.f;
      Xl[c * PS + tid * WS + W + 1] = 0.f;
    }
  }
  __syncthreads();

  // ---- this thread's window: rows row .. row + 2, padded columns x0 .. x0 + 5 ----
  const int p = 4 * tid, row = p >> lgW, x0 = p & (W - 1);
  v2f xin[C][3][3];   // value i of a window row: half i & 1 of pair i >> 1
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* s = Xl + c * PS + (row + ky) * WS + x0;
      const float4 a = *(const float4*)s;
      const float2 b = *(const float2*)(s + 4);
      xin[c][ky][0] = (v2f){a.x, a.y};
      xin[c][ky][1] = (v2f){a.z, a.w};
      xin[c][ky][2] = (v2f){b.x, b.y};
    }

  const long long obase = ((long long)n * g.M + m0) * HW + (long long)r0 * W + p;   // this thread's quad in channel m0
  const int mcount = g.M - m0 < THIN_IN_MB ? g.M - m0 : THIN_IN_MB;
  const bool mul = DGRAD && g.dact_in != nullptr;

  auto run = [&](auto actf) {
    float4 dd[2], ddn[2];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      ddn[mm] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (mul && mm < mcount) ddn[mm] = *(const float4*)(g.dact_in + obase + mm * HW);
    }
#pragma unroll 1
    for (int mp = 0; mp < THIN_IN_MB / 2; ++mp) {
      if (2 * mp >= mcount) break;   // uniform
      dd[0] = ddn[0]; dd[1] = ddn[1];
      if (mul) {   // act' factors of the next pair: in flight under this pair's FMAs
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
          if (2 * mp + 2 + mm < mcount) ddn[mm] = *(const float4*)(g.dact_in + obase + (2 * mp + 2 + mm) * HW);
      }
      const v2f* wp = (const v2f*)(Wl + mp * KP * 2);
      v2f acc[4];
      const v2f bv = wp[KC];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = bv;
      gmk_static_for<0, KC>([&](auto k_) {
        constexpr int k = decltype(k_)::value, c = k / 9, ky = (k % 9) / 3, kx = k % 3;
        const v2f wv = wp[k];
        gmk_static_for<0, 4>([&](auto j_) {
          constexpr int i = decltype(j_)::value + kx;
          thin_in_fma<i & 1>(acc[decltype(j_)::value], wv, xin[c][ky][i >> 1]);
        });
      });
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        if (2 * mp + mm >= mcount) break;   // uniform
        const long long idx = obase + (2 * mp + mm) * HW;
        const float z0 = acc[0][mm], z1 = acc[1][mm], z2 = acc[2][mm], z3 = acc[3][mm];
        if (DGRAD) {
          const float4 d4 = dd[mm];
          *(float4*)(g.y + idx) = make_float4(z0 * d4.x, z1 * d4.y, z2 * d4.z, z3 * d4.w);
        } else {
          float4 a, d;
          actf(make_float4(z0, z1, z2, z3), a, d);
          *(float4*)(g.y + idx) = a;
          if (g.dact_out) *(float4*)(g.dact_out + idx) = d;
        }
      }
    }
  };
  if (DGRAD) run([](const float4& z, float4& a, float4& d) { a = z; d = z; });   // (unused on the dgrad path)
  else GMK_WITH_ACT(g.act, run)
}

bool gmk_thin_in_ok(const ConvGeom& g, int ksize, int mode) {
  if (ksize != 3 || mode != 0) return false;
  if (g.C < 1 || g.C > 4) return false;
  if (!((g.w_layout == 0 && !g.flip && g.epi == 0) || (g.w_layout == 1 && g.flip && g.epi == 1))) return false;
  if (!gmk_is_pow2(g.IW) || !gmk_is_pow2(g.IH) || g.IW < 8 || g.IW > 128) return false;
  if ((long long)g.IH * g.IW < 1024) return false;   // a block is 1024 pixels of ONE image
  if ((1 << g.lgBW) != g.IW) return false;
  return true;
}

template <int C>
static int launch_thin_in(const ConvGeom& g, hipStream_t s) {
  constexpr int KC = C * 9, KP = (KC + 2) & ~1;
  const int R = 1024 / g.IW;
  const size_t lds = sizeof(float) * ((size_t)THIN_IN_MB * KP + (size_t)C * (R + 2) * (g.IW + 4));
  dim3 grid((unsigned)((long long)g.N * (g.IH / R)), (unsigned)((g.M + THIN_IN_MB - 1) / THIN_IN_MB));
  if (g.epi == 0) hipLaunchKernelGGL((conv_thin_in_kernel<C, false>), grid, dim3(256), lds, s, g);
  else hipLaunchKernelGGL((conv_thin_in_kernel<C, true>), grid, dim3(256), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

int gmk_launch_conv_thin_in(const ConvGeom& g, hipStream_t s) {
  if ((long long)g.N * (g.IH / (1024 / g.IW)) > 0x7FFFFFFFll) return GMK_E_TOO_LARGE;
  switch (g.C) {
    case 1: return launch_thin_in<1>(g, s);
    case 2: return launch_thin_in<2>(g, s);
    case 3: return launch_thin_in<3>(g, s);
    case 4: return launch_thin_in<4>(g, s);
  }
  return GMK_E_SHAPE;
}

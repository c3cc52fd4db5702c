// conv_wino.hip -- Winograd F(2x2,3x3) stride-1 convolution kernel + launcher (see conv_wino.h).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      per 2x2 output tile, d = 4x4 input patch, g = 3x3 taps
//
// GEMM view, one per position xi = 4a+b of the 4x4 transform domain:  M_xi[tile][cout] = sum_c V_xi[tile][c] U_xi[c][cout]
//   A operand (MFMA rows)    = tiles  : V, the transformed input patches
//   B operand (MFMA columns) = couts  : U, the transformed weights
// A block owns 32 consecutive tiles (whole tile rows, possibly several images: 128 output pixels) x 64 output channels;
// each of its 4 waves owns 16 output channels x all 32 tiles x ALL 16 positions (2 x 16 accumulators of 4 registers), so
// the output transform A^T M A stays inside a lane and no wave shares an output with another.
// Per K-chunk of 8 input channels every thread transforms ONE (tile, channel) patch (16 dword buffer loads, zero padding
// by the descriptor's range check, 32 adds) and TWO (cout, channel) weight slices (~56 flops) in registers and writes
// them to LDS ([xi][c][tile] and [xi][c][cout]; 16-lane halves XOR-swapped per channel so that neither the k-lanes of an
// MFMA operand read nor the channel groups of a weight write share a bank).  The loads of the next chunk are issued
// before the MFMA section and land under it.  Weight slices: the data gradient reads them with 18 dword loads per thread
// ((Cin, Cout, 3, 3)-major addressing: 16 lanes = 576 contiguous bytes); the forward's (Cout, Cin, 3, 3) rows would put
// every lane on its own cache line (measured: 43 us of address processing alone), so there each wave copies its 16
// output channels' 288-byte row pieces with float4 loads into an LDS strip of its own and reads the slices back from it.
// The MFMA section is a hand-ordered stream of {ds_read_b32 x3 four steps ahead, counted lgkmcnt, two asm MFMAs}, as in
// conv_wgrad16.h.  Epilogue: output transform in registers, per-wave transpose through LDS, then bias + activation
// (+ act') or the act' multiply of dgrad on whole 512-byte channel rows.
#include "conv_wino.h"

#define WINO_LDS_FLOATS(DG) (16 * WINO_CK * 32 + 16 * WINO_CK * 64 + ((DG) ? 0 : 64 * WINO_RS + 4))

template <bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvGeom g) {
  constexpr int CK = WINO_CK, ES = WINO_ES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vl = smem;                    // [16 xi][8 c][32 tiles]
  float* Ul = smem + 16 * CK * 32;     // [16 xi][8 c][64 couts]
  float* Rl = Ul + 16 * CK * 64;       // forward only: raw weight rows [64 couts][WINO_RS], one 16-row strip per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int W = g.IW, H = g.IH, HW = H * W;
  const int lgW = g.lgBW, lgHW = g.lgBHW, lgTW = lgW - 1, lgTHW = lgHW - 2, TW = W >> 1;
  // blocks go round-robin over the 8 XCDs: XCD x takes the tile groups [x nb/8, (x+1) nb/8), so the halo rows shared by
  // neighbouring groups of one image are read from HBM once (one L2 per XCD)
  int bx = blockIdx.x;
  if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
  const long long T0 = (long long)bx * WINO_TB;
  const int m0 = blockIdx.y * WINO_MB;
  const __amdgpu_buffer_rsrc_t rx = gmk_make_rsrc(g.x, (long long)g.N * g.C * HW * 4);

  // ---- input slot of this thread: tile ti, channel ci of the chunk ----
  const int ti = tid & 31, ci = tid >> 5;
  int vrow[4];   // byte offsets of the patch rows at column 2tx (out of range for rows outside the image)
  bool colL, colR;
  {
    const long long T = T0 + ti;
    const int n = (int)(T >> lgTHW), rem = (int)(T & ((1 << lgTHW) - 1));
    const int ty = rem >> lgTW, tx = rem & (TW - 1);
    colL = tx > 0; colR = tx < TW - 1;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const int iy = 2 * ty - 1 + dy;
      const bool ok = n < g.N && iy >= 0 && iy < H;
      vrow[dy] = ok ? (int)((((long long)n * g.C + ci) * HW + iy * W + 2 * tx) * 4) : GMK_OOB;
    }
  }
  // ---- weight slots: every wave transforms the slices of its own 16 output channels; a lane takes the output channel
  // PAIR mk, mk+1 (mk = 16 wave + 2 (lane & 7)) of channel wc = lane >> 3 of the chunk, so that each transformed value
  // goes to LDS as one 8-byte store ----
  const int nch = g.C / CK;
  const int mk = 16 * wave + 2 * (lane & 7), wc = lane >> 3;
  // DGRAD ((Cin, Cout, 3, 3)-major addressing): the pair's 18 floats are contiguous; 9 float2 loads
  const bool wok = m0 + mk < g.M;   // (M % 16 == 0: both channels of the pair or neither)
  const unsigned wvoff = wok ? (unsigned)(((m0 + mk) * g.w_m_stride + wc * g.w_c_stride) * 4) : 0u;
  // forward: the wave's 16 rows x 72 floats = 288 float4 pieces, piece e = lane + 64 u: row e / 18, quad e % 18
  int pvoff[5], ploff[5];
  unsigned pok = 0;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int e = lane + 64 * u, row = e / 18, q = e % 18;
    const bool ok = e < 288 && m0 + 16 * wave + row < g.M;
    pvoff[u] = ok ? ((m0 + 16 * wave + row) * g.w_m_stride + q * 4) * 4 : 0;   // (plain loads: masked pieces read element 0)
    pok |= ok ? (1u << u) : 0u;
    ploff[u] = e < 288 ? (16 * wave + row) * WINO_RS + q * 4 : 64 * WINO_RS;   // surplus lanes: 4 trash words behind the rows
  }
  const int rraw = mk * WINO_RS + wc * 9;   // this lane's two 9-float slices in the strip: rows mk and mk + 1
  // ---- LDS offsets (floats) ----
  // bank swizzle of both tiles: column ^ 16 on channels with (c ^ (c >> 1)) & 1 -- channels c and c+1 (the k-lanes of
  // one 32-lane read pass, the two channels of one 16-lane weight write pass) land in different halves
  const int vw = ci * 32 + (ti ^ (16 * ((ci ^ (ci >> 1)) & 1)));
  const int uw = wc * 64 + (mk ^ (16 * ((wc ^ (wc >> 1)) & 1)));
  unsigned addrA[2], addrB;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
    addrA[tt] = gmk_lds_addr(Vl) + (unsigned)(kq * 32 + ((16 * tt + l15) ^ (16 * ((kq ^ (kq >> 1)) & 1)))) * 4u;
  addrB = gmk_lds_addr(Ul) + (unsigned)(kq * 64 + ((16 * wave + l15) ^ (16 * ((kq ^ (kq >> 1)) & 1)))) * 4u;

  f32x4 acc[16][2];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc[x][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Prefetch distances: input patches (HBM) TWO chunks ahead in two register sets, weights (L2) one chunk ahead -- with
  // everything one chunk ahead the loads had a single MFMA section (~1 us) to land and every chunk waited out the rest
  // of an HBM round trip.
  // A thread loads only the two middle columns of its 4x4 patch (one 8-byte load per row); the outer columns are the
  // neighbouring tiles' middle columns and come from lanes -1 / +1 by DPP row shifts (tile rows are at most 16 tiles
  // long and 16-lane aligned; a shift across a tile row's end fetches a value that the column masks zero anyway).
  float xin2[2][8];
  float wraw[2][9];   // the pair's two slices (DGRAD: loaded directly, flipped taps)
  f32x4 wq[5];        // forward: this lane's pieces of the wave's raw rows
  auto issue_x = [&](float* xin, int c0) {
    const int soff = c0 * HW * 4;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      // (two dword loads: the 64-bit buffer-load builtin of this compiler emits a single buffer_load_dword)
      xin[2 * dy] = gmk_buffer_load(rx, vrow[dy], soff);
      xin[2 * dy + 1] = gmk_buffer_load(rx, vrow[dy] + 4, soff);
    }
  };
  auto issue_w = [&](int c0) {
    const int wsoff = c0 * g.w_c_stride * 4;
    if constexpr (DGRAD) {
      const float2* wp = (const float2*)((const char*)g.w + (wvoff + (unsigned)wsoff));
#pragma unroll
      for (int q = 0; q < 9; ++q) {   // floats 2q, 2q+1 of the 18: slice (2q) / 9, tap 8 - (2q) % 9 (flipped)
        const float2 v = wp[q];
        wraw[(2 * q) / 9][8 - (2 * q) % 9] = wok ? v.x : 0.f;
        wraw[(2 * q + 1) / 9][8 - (2 * q + 1) % 9] = wok ? v.y : 0.f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 5; ++u) wq[u] = *(const f32x4*)((const char*)g.w + (unsigned)(pvoff[u] + wsoff));
    }
  };
  auto write_phase = [&](const float* xin) {
    // V = B^T d B
    float tr[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d[4];
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        if (j == 1) d[dy] = xin[2 * dy];
        else if (j == 2) d[dy] = xin[2 * dy + 1];
        else if (j == 0) {   // column 2tx-1 = lane-1's second column (row_shr:1)
          const int nb = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xin[2 * dy + 1]), 0x111, 0xf, 0xf, true);
          d[dy] = colL ? __builtin_bit_cast(float, nb) : 0.f;
        } else {             // column 2tx+2 = lane+1's first column (row_shl:1)
          const int nb = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xin[2 * dy]), 0x101, 0xf, 0xf, true);
          d[dy] = colR ? __builtin_bit_cast(float, nb) : 0.f;
        }
      }
      tr[j] = d[0] - d[2]; tr[4 + j] = d[1] + d[2]; tr[8 + j] = d[2] - d[1]; tr[12 + j] = d[1] - d[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t0 = tr[4 * i], t1 = tr[4 * i + 1], t2 = tr[4 * i + 2], t3 = tr[4 * i + 3];
      Vl[vw + (4 * i + 0) * (CK * 32)] = t0 - t2;
      Vl[vw + (4 * i + 1) * (CK * 32)] = t1 + t2;
      Vl[vw + (4 * i + 2) * (CK * 32)] = t2 - t1;
      Vl[vw + (4 * i + 3) * (CK * 32)] = t1 - t3;
    }
    // U = G g G^T
    if constexpr (!DGRAD) {   // through the wave's own strip (a wave's LDS accesses execute in order: no barrier)
#pragma unroll
      for (int u = 0; u < 5; ++u)
        *(float4*)(Rl + ploff[u]) = ((pok >> u) & 1u) ? make_float4(wq[u][0], wq[u][1], wq[u][2], wq[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wraw[j][t] = Rl[rraw + j * WINO_RS + t];
    }
    float uo[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float u[4][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float g0 = wraw[j][c], g1 = wraw[j][3 + c], g2 = wraw[j][6 + c];
        const float sh = 0.5f * (g0 + g2);
        u[0][c] = g0; u[1][c] = fmaf(0.5f, g1, sh); u[2][c] = fmaf(-0.5f, g1, sh); u[3][c] = g2;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sh = 0.5f * (u[i][0] + u[i][2]);
        uo[j][4 * i + 0] = u[i][0];
        uo[j][4 * i + 1] = fmaf(0.5f, u[i][1], sh);
        uo[j][4 * i + 2] = fmaf(-0.5f, u[i][1], sh);
        uo[j][4 * i + 3] = u[i][2];
      }
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) *(float2*)(Ul + uw + x * (CK * 64)) = make_float2(uo[0][x], uo[1][x]);
  };
  // MFMA section of one chunk: step q = (k-step s, position xi), s outer so that consecutive MFMAs never share an
  // accumulator; three reads per step (tiles 0-15, tiles 16-31, couts) issued PF steps ahead
  constexpr int NQ = 2 * 16, PF = 4;
  auto mfma_phase = [&]() {
    float ra[PF + 1][2], rb[PF + 1];
    auto issue = [&](auto q_) {
      (void)ra; (void)rb; (void)addrA; (void)addrB;
      constexpr int q = decltype(q_)::value, s = q >> 4, xi = q & 15, slot = q % (PF + 1);
      GMK_DS_READ(ra[slot][0], addrA[0], (xi * CK + 4 * s) * 32 * 4);
      GMK_DS_READ(ra[slot][1], addrA[1], (xi * CK + 4 * s) * 32 * 4);
      GMK_DS_READ(rb[slot], addrB, (xi * CK + 4 * s) * 64 * 4);
    };
    gmk_static_for<0, PF>(issue);
    gmk_static_for<0, NQ>([&](auto q_) {
      (void)ra; (void)rb; (void)issue; (void)acc;
      constexpr int q = decltype(q_)::value, xi = q & 15, slot = q % (PF + 1);
      if constexpr (q + PF < NQ) issue(std::integral_constant<int, q + PF>{});
      constexpr int ahead = (NQ - 1 - q < PF ? NQ - 1 - q : PF) * 3;
      asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(ahead) : "memory");
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi][0]) : "v"(ra[slot][0]), "v"(rb[slot]));
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi][1]) : "v"(ra[slot][1]), "v"(rb[slot]));
    });
  };

  issue_w(0);
  issue_x(xin2[0], 0);
  if (nch > 1) issue_x(xin2[1], CK);
  // g.dbg: ablations for profiling builds / runs (GMK_WINO_DBG; results are wrong when set): 1 = no epilogue,
  // 4 = transforms + LDS writes only for the first chunk, 8 = no MFMA section
  auto chunk = [&](int i, float* xin) {     // xin: the register set holding chunk i, reloaded with chunk i + 2
    if (!(g.dbg & 4) || i == 0) write_phase(xin);
    gmk_lds_barrier();                      // the chunk's tiles are complete
    if (i + 1 < nch) issue_w((i + 1) * CK);
    if (i + 2 < nch) issue_x(xin, (i + 2) * CK);
    if (!(g.dbg & 8)) mfma_phase();
    gmk_lds_barrier();                      // everybody is done reading them
  };
  for (int i = 0; i < nch; i += 2) {
    chunk(i, xin2[0]);
    if (i + 1 < nch) chunk(i + 1, xin2[1]);
  }
  if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) return;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results are read by the VALU below

  // ---- epilogue: Y = A^T M A per (tile, cout) in registers, transposed through this wave's LDS strip ----
  const long long total_px = (long long)g.N * HW;
  float* Ew = smem + wave * (16 * ES);
  // dgrad: the act' factors of this lane's eight row quads, fetched before the math
  const int q4 = lane & 31, ch = lane >> 5;
  const long long G0 = T0 * 4 + q4 * 4;
  const int nq = (int)(G0 >> lgHW), pixq = (int)(G0 & (HW - 1));
  float4 dds[8];
  float bvs[8];
  long long idxs[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int m = m0 + 16 * wave + ch + 2 * it;
    const int mc = m < g.M ? m : g.M - 1;
    idxs[it] = ((long long)nq * g.M + mc) * HW + pixq;
    dds[it] = make_float4(1.f, 1.f, 1.f, 1.f);
    bvs[it] = (g.epi == 0 && g.bias) ? g.bias[mc] : 0.f;
    if (g.epi != 0 && g.dact_in && G0 < total_px) dds[it] = *(const float4*)(g.dact_in + idxs[it]);
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p0[4], p1[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float M0 = acc[b][tt][r], M1 = acc[4 + b][tt][r], M2 = acc[8 + b][tt][r], M3 = acc[12 + b][tt][r];
        p0[b] = (M0 + M1) + M2;
        p1[b] = (M1 - M2) - M3;
      }
      const int t = 16 * tt + 4 * kq + r;
      const int P = (2 * (t >> lgTW)) * W + 2 * (t & (TW - 1));
      float* d = Ew + l15 * ES + P;
      *(float2*)d = make_float2((p0[0] + p0[1]) + p0[2], (p0[1] - p0[2]) - p0[3]);
      *(float2*)(d + W) = make_float2((p1[0] + p1[1]) + p1[2], (p1[1] - p1[2]) - p1[3]);
    }
  }
  // the strip is private to the wave: LDS accesses of one wave complete in order, no barrier needed
  auto store_rows = [&](auto actf) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int cl = ch + 2 * it;
      const int m = m0 + 16 * wave + cl;
      if (m >= g.M || G0 >= total_px) continue;
      const float2 lo = *(const float2*)(Ew + cl * ES + 4 * q4), hi = *(const float2*)(Ew + cl * ES + 4 * q4 + 2);
      float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
      if (g.epi == 0) {
        const float bv = bvs[it];
        float4 a, dd;
        actf(make_float4(v.x + bv, v.y + bv, v.z + bv, v.w + bv), a, dd);
        *(float4*)(g.y + idxs[it]) = a;
        if (g.dact_out) *(float4*)(g.dact_out + idxs[it]) = dd;
      } else {
        const float4 dd = dds[it];
        v.x *= dd.x; v.y *= dd.y; v.z *= dd.z; v.w *= dd.w;
        *(float4*)(g.y + idxs[it]) = v;
      }
    }
  };
  if (g.epi != 0) store_rows([](const float4& z, float4& a, float4& d) { a = z; d = z; });   // (unused on the dgrad path)
  else GMK_WITH_ACT(g.act, store_rows)
}


// The forward kernel needs 66 KiB of dynamic LDS: the limit is raised once per device (not on every launch); if the
// runtime refuses, the dispatcher keeps the layer on the implicit GEMM instead of failing the call.
#include <mutex>
bool gmk_wino_fwd_ready() {
  static std::mutex mu;
  static signed char state[64];   // per device: 0 = not tried, 1 = ok, -1 = refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       WINO_LDS_FLOATS(0) * 4);
    if (e != hipSuccess) (void)hipGetLastError();   // clear the sticky error: the caller falls back
    state[dev] = e == hipSuccess ? 1 : -1;
  }
  return state[dev] > 0;
}

int gmk_launch_conv_wino(ConvGeom& g, hipStream_t s) {
  const long long tiles = (long long)g.N * g.IH * g.IW / 4;
  dim3 grid((unsigned)(tiles / WINO_TB), (unsigned)((g.M + WINO_MB - 1) / WINO_MB));
  if (g.flip) {
    hipLaunchKernelGGL(conv_wino_kernel<true>, grid, dim3(256), WINO_LDS_FLOATS(1) * 4, s, g);
  } else {
    constexpr int lds = WINO_LDS_FLOATS(0) * 4;   // 66 KiB: above the 64 KiB default (gmk_wino_fwd_ready raised the limit)
    hipLaunchKernelGGL(conv_wino_kernel<false>, grid, dim3(256), lds, s, g);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

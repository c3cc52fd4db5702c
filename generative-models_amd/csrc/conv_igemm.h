// conv_igemm.h -- implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
//
// One kernel template covers every "gather" shaped conv of the path:
//   MODE 0  stride-1 conv            (conv fwd; conv dgrad with flipped taps)
//   MODE 1  stride-2 conv            (downsample fwd; transposed-conv dgrad)
//   MODE 2  stride-2 transposed conv (deconv fwd; stride-2 conv dgrad), computed per output
//           parity class so that only the taps that really contribute are multiplied.
//
// GEMM view: D[m][p] = sum_{c,t} Wsel[m][c][t] * X[c][p shifted by tap t]
//   A operand (MFMA rows, 32 per tile)  = weights  (m = output channel of this op)
//   B operand (MFMA cols, 32 per tile)  = pixels   -> the accumulator has the pixel on the lane,
//                                                    so NCHW stores are 128-B coalesced rows.
// "Base grid": output pixels for MODE 0/1, input pixels for MODE 2 (each base position yields the
// four outputs (2y+py, 2x+px)).  A block owns PT consecutive base positions (whole rows, possibly
// several images) x MB output channels; the input patch (with halo, zero padded) of CK input
// channels and the matching weight slice are staged in LDS per K-chunk.
#pragma once
#include "gmk_common.h"
#include <stdlib.h>

struct ConvGeom {
  const float* x; const float* w; const float* bias; const float* dact_in;
  float* y; float* dact_out;
  int N, C, M;
  int IH, IW, OH, OW;
  int BW, lgBW, lgBHW;
  long long total_base;
  int w_layout;  // 0: weights addressed [m][c][t] (c,t contiguous per m); 1: [c][m][t]
  int w_m_stride, w_c_stride, flip, w_numel;
  int act, epi;  // epi 0: +bias, act (and act' to dact_out); 1: * dact_in (dgrad)
  int seg_px, lg_seg_px, nseg;
  int PR, PC, PS, npos;
#ifdef GMK_CONV_TRACE
  unsigned long long* trace;   // tools-only build: per block {hw id, t_start, t_k_done, t_end} (wall_clock64 ticks)
#endif
  int dbg;  // ablation bits for profiling builds (0 in production): 1 = no epilogue stores, 2 = no K loop, 4 = every chunk reloads chunk 0 (cache-resident operands, wrong results: the K loop without HBM latency)
};

#define CONV_MAXSLOT 4
// slot-count instantiations per mode: stride-2 patches are ~4x larger than the tile
#define GMK_NS_LO(MODE_) ((MODE_) == 1 ? 2 : 1)
#define GMK_NS_HI(MODE_) ((MODE_) == 1 ? 4 : 2)

// NSLOT = ceil(npos / 256): patch positions handled per thread (compile time so that the
// prefetch registers are allocated exactly).
template <int MODE, int KS, int CK, int WM, int MT, int NSLOT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvGeom g) {
  constexpr int WP = 4 / WM, PT = WP * 32, MB = WM * MT * 32, KK = KS * KS, KC = CK * KK, WS = KC | 1;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NCLS = (MODE == 2) ? 4 : 1;
  constexpr int SB = (MODE == 1) ? 2 : 1;
  constexpr int TLO = (MODE == 2) ? (KS - 1 - PAD) / 2 : 0;
  constexpr int ORG = (MODE == 2) ? TLO : PAD;  // patch origin offset (rows/cols before the base position)
  constexpr int NW = (MB * KC + 255) / 256;     // weight elements staged per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // LDS: [W tile 0][W tile 1][X patch 0][X patch 1]; chunk i lives in buffer i & 1 and is written while chunk i-1 is multiplied
  float* Wl = smem;
  float* Xl = smem + 2 * MB * WS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef GMK_CONV_TRACE
  const unsigned long long tr_t0 = wall_clock64();
#endif
  const int wm = wave / WP, wp = wave % WP;
  const int h = lane >> 5, j = lane & 31;
  // Blocks go round-robin over the 8 XCDs (one L2 each; linear block id = x + y gridDim.x): XCD k takes the position
  // tiles [k nb/8, (k+1) nb/8), so the halo rows that neighbouring tiles of one image share are fetched from HBM once
  int bxr = blockIdx.x;
  if ((gridDim.x & 7) == 0) bxr = (bxr & 7) * (gridDim.x >> 3) + (bxr >> 3);
  const long long p0 = (long long)bxr * PT;
  const int m0 = blockIdx.y * MB;
  const int BHW = 1 << g.lgBHW;
  const int PR = g.PR, PC = g.PC, PS = g.PS;
  const int IHW = g.IH * g.IW;

  // ---- per-thread patch slots (position decode done once; channels advance by IHW / PS) ----
  // Loads go through buffer descriptors: an out-of-image / padded lane passes GMK_OOB and reads 0.
  const long long n0 = p0 >> g.lgBHW;  // first image touched by this block (uniform)
  long long nimg = g.N - n0;
  if (nimg > g.nseg + 1) nimg = g.nseg + 1;
  const __amdgpu_buffer_rsrc_t rx = gmk_make_rsrc(g.x + n0 * g.C * (long long)IHW, nimg * g.C * (long long)IHW * 4);
  const __amdgpu_buffer_rsrc_t rw = gmk_make_rsrc(g.w, (long long)g.w_numel * 4);
  const __amdgpu_buffer_rsrc_t rw0 = gmk_make_rsrc(g.w, 0);  // zero records: every load returns 0
  int s_voff[NSLOT], s_loff[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    int e = tid + k * 256;
    if (e >= g.npos) e = g.npos - 1;  // surplus threads duplicate the last position (same value to the same address)
    s_voff[k] = GMK_OOB;
    {
      int seg = e / (PR * PC), r = e % (PR * PC);
      int prow = r / PC, pcol = r % PC;
      long long P = p0 + (long long)seg * g.seg_px;
      long long n = P >> g.lgBHW;
      int r0 = ((int)(P & (BHW - 1))) >> g.lgBW;
      int iy = r0 * SB + prow - ORG, ix = pcol - ORG;
      s_loff[k] = seg * CK * PS + prow * PC + pcol;
      if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW)
        s_voff[k] = (int)(((n - n0) * g.C * (long long)IHW + iy * g.IW + ix) * 4);
    }
  }
  // ---- per-thread weight slots: element e = tid + u*256 of the [MB][KC] chunk ----
  int w_voff[NW], w_loff[NW];
#pragma unroll
  for (int u = 0; u < NW; ++u) {
    int e = tid + u * 256;
    if (e >= MB * KC) e = MB * KC - 1;  // surplus threads duplicate the last element
    int m, c, t;
    if (g.w_layout == 0) { m = e / KC; int r = e % KC; c = r / KK; t = r % KK; }
    else { c = e / (MB * KK); int r = e % (MB * KK); m = r / KK; t = r % KK; }
    const int gm = m0 + m;
    w_voff[u] = gm < g.M ? (gm * g.w_m_stride + c * g.w_c_stride + t) * 4 : GMK_OOB;
    const int tt = g.flip ? (KK - 1 - t) : t;
    w_loff[u] = m * WS + c * KK + tt;
  }

  // ---- lane bases for the MFMA operand reads ----
  const int q = wp * 32 + j;  // base position within the block
  const int seg = q >> g.lg_seg_px, qs = q & (g.seg_px - 1);
  const int br = qs >> g.lgBW, bx = qs & (g.BW - 1);
  const int bbase = seg * CK * PS + h * PS + (br * SB) * PC + bx * SB;
  const int abase = (wm * MT * 32 + j) * WS + h * KK;

  // patch (row, column) and parity class of tap t
  auto tap_row = [](int t) constexpr { const int ky = t / KS; return MODE == 2 ? (((ky + PAD) & 1) + PAD - ky) / 2 + TLO : ky; };
  auto tap_col = [](int t) constexpr { const int kx = t % KS; return MODE == 2 ? (((kx + PAD) & 1) + PAD - kx) / 2 + TLO : kx; };
  auto tap_cls = [](int t) constexpr { return MODE == 2 ? (((t / KS + PAD) & 1) * 2 + ((t % KS + PAD) & 1)) : 0; };
  constexpr int NSTEP = (CK / 2) * KK, RPS = 1 + MT, PF = 12 / RPS;
  static_assert(MT == 1 || MT == 2, "operand ring is written for 1 or 2 channel tiles per wave");
  const int XSZ = g.nseg * CK * PS;  // floats per X patch buffer
  unsigned addrA[2], addrB[2][CK / 2][KS];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    addrA[p] = gmk_lds_addr(Wl) + (unsigned)(p * MB * WS + abase) * 4u;
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp)
#pragma unroll
      for (int r = 0; r < KS; ++r)
        addrB[p][cp][r] = gmk_lds_addr(Xl) + (unsigned)(p * XSZ + bbase + cp * 2 * PS + r * PC) * 4u;
  }

  f32x16 acc[NCLS][MT];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

  float xr[NSLOT][CK], wr[NW];
  // Batched loads of one K-chunk into the staging registers.  Full chunks need no masks; the ragged tail chunk
  // (C % CK channels) masks the channels beyond C on both operands.
  auto load_full = [&](int c0) {
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const int soff = (c0 + c) * IHW * 4;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) xr[k][c] = gmk_buffer_load(rx, s_voff[k], soff);
    }
    const int wsoff = c0 * g.w_c_stride * 4;
#pragma unroll
    for (int u = 0; u < NW; ++u) wr[u] = gmk_buffer_load(rw, w_voff[u], wsoff);
  };
  auto load_tail = [&](int c0) {
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const bool cok = c0 + c < g.C;  // uniform
      const int soff = cok ? (c0 + c) * IHW * 4 : 0;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) xr[k][c] = gmk_buffer_load(rx, cok ? s_voff[k] : GMK_OOB, soff);
    }
    const int wsoff = c0 * g.w_c_stride * 4;
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int c = (w_loff[u] % WS) / KK;  // channel of this element, recovered from its LDS offset
      wr[u] = gmk_buffer_load(rw, (c0 + c < g.C) ? w_voff[u] : GMK_OOB, wsoff);
    }
  };
  auto write_buf0 = [&]() {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
#pragma unroll
      for (int c = 0; c < CK; ++c) Xl[s_loff[k] + c * PS] = xr[k][c];
#pragma unroll
    for (int u = 0; u < NW; ++u) Wl[w_loff[u]] = wr[u];
  };

  // Pipeline over the full chunks: chunk 0 is staged up front and chunk 1 is loaded into the staging registers.
  // Iteration i multiplies buffer i & 1 and, spread over its k-steps, writes the registers (chunk i+1) into the other
  // buffer and reloads each register with chunk i+2 right behind its write -- one register set, a full iteration of
  // load latency, one barrier per chunk.  All of it is straight-line code (exact vmcnt counting); chunks past the end
  // are clamped to the last one and land in a buffer nobody reads.
  const int nch = g.C / CK;            // full chunks
  const int tail_c0 = nch * CK;        // first channel of the ragged tail chunk (== C if there is none)
  if (nch > 0) {
    load_full(0);
    write_buf0();
    if (nch > 1) load_full(CK);
    else {  // single chunk: the padding chunk of the pair gets zero weights
#pragma unroll
      for (int u = 0; u < NW; ++u) wr[u] = 0.f;
    }
  }

  constexpr int NITEM = NSLOT * CK + NW, IPS = (NITEM + NSTEP - 1) / NSTEP;  // staging items, items per k-step
  auto chunk_body = [&](auto P_, int i) {
    constexpr int P = decltype(P_)::value;
    gmk_lds_barrier();  // buffer P is complete; everybody is done reading buffer P ^ 1 (reloads stay in flight)
    float* Xw = Xl + (P ^ 1) * XSZ;
    float* Ww = Wl + (P ^ 1) * MB * WS;
    const int cn = (g.dbg & 4) ? 0 : (i + 2 < nch ? i + 2 : nch - 1) * CK;  // chunk to reload (uniform); dbg 4: always chunk 0
    int soffc[CK];
#pragma unroll
    for (int c = 0; c < CK; ++c) soffc[c] = (cn + c) * IHW * 4;
    const int wsoff = cn * g.w_c_stride * 4;
    const __amdgpu_buffer_rsrc_t rwn = (i + 2 < nch) ? rw : rw0;  // uniform select; chunks past the end read zeros
    auto stage_item = [&](auto q_) {
      constexpr int q = decltype(q_)::value;
      if constexpr (q < NSLOT * CK) {
        constexpr int k = q / CK, c = q % CK;
        Xw[s_loff[k] + c * PS] = xr[k][c];
        xr[k][c] = gmk_buffer_load(rx, s_voff[k], soffc[c]);
      } else if constexpr (q < NITEM) {
        constexpr int u = q - NSLOT * CK;
        Ww[w_loff[u]] = wr[u];
        wr[u] = gmk_buffer_load(rwn, w_voff[u], wsoff);
      }
    };
    // MFMA over the chunk: k-step = (channel pair cp, tap t); lane half h takes channel 2cp+h.  Operands come from LDS
    // through hand-placed ds_read_b32 (immediate offsets from loop-invariant address registers, no VALU in the loop),
    // PF k-steps ahead of their MFMAs; the ring of PF+1 register sets is indexed at compile time.
    float rb[PF + 1], ra[PF + 1][MT];
    auto issue = [&](auto s_) {
      constexpr int s = decltype(s_)::value, cp = s / KK, t = s % KK, slot = s % (PF + 1);
      GMK_DS_READ(rb[slot], addrB[P][cp][tap_row(t)], tap_col(t) * 4);
      GMK_DS_READ(ra[slot][0], addrA[P], (cp * 2 * KK + t) * 4);
      if constexpr (MT == 2) GMK_DS_READ(ra[slot][1], addrA[P], (32 * WS + cp * 2 * KK + t) * 4);
    };
    gmk_static_for<0, PF>(issue);
    gmk_static_for<0, NSTEP>([&](auto s_) {
      constexpr int s = decltype(s_)::value, t = s % KK, slot = s % (PF + 1);
      if constexpr (s + PF < NSTEP) issue(std::integral_constant<int, s + PF>{});
      // LDS operations complete in order, so the staging writes issued in between only make this wait stricter
      constexpr int ahead = (NSTEP - 1 - s < PF ? NSTEP - 1 - s : PF) * RPS;  // reads issued after this step's
      if constexpr (MT == 2) GMK_LDS_WAIT3(ahead, rb[slot], ra[slot][0], ra[slot][1]);
      else GMK_LDS_WAIT2(ahead, rb[slot], ra[slot][0]);
      constexpr int cls = tap_cls(t);
#pragma unroll
      for (int i2 = 0; i2 < MT; ++i2)
        acc[cls][i2] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[slot], ra[slot][i2], acc[cls][i2], 0, 0, 0);  // rows = pixels, cols = channels
      gmk_static_for<s * IPS, (s + 1) * IPS>(stage_item);
    });
  };
  // Chunks go in pairs (buffer 0, buffer 1) so that the buffer index is a compile-time constant; an odd chunk count is
  // padded with one chunk whose weights are all zero (loaded through the empty descriptor rw0).
  const int nloop = (g.dbg & 2) ? 0 : nch;
  for (int i = 0; i < nloop; i += 2) {
    chunk_body(std::integral_constant<int, 0>{}, i);
    chunk_body(std::integral_constant<int, 1>{}, i + 1);
  }
  // Ragged tail (C % CK channels, e.g. the 3 image channels): one masked chunk through buffer 0, compiler-scheduled
  if (tail_c0 < g.C && !(g.dbg & 2)) {
    __syncthreads();  // the pipeline's LDS reads and staging writes are done
    load_tail(tail_c0);
    write_buf0();
    __syncthreads();
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const float b = Xl[bbase + cp * 2 * PS + tap_row(t) * PC + tap_col(t)];
#pragma unroll
        for (int i2 = 0; i2 < MT; ++i2) {
          const float a = Wl[abase + i2 * 32 * WS + cp * 2 * KK + t];
          acc[tap_cls(t)][i2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[tap_cls(t)][i2], 0, 0, 0);
        }
      }
    }
  }

#ifdef GMK_CONV_TRACE
  const unsigned long long tr_t1 = wall_clock64();
#endif
  // ---- epilogue ----
  // Accumulator layout (rows = pixels, cols = channels): lane -> channel (lane & 31); register r -> pixel
  // (r & 3) + 8 (r >> 2) + 4 h of the wave's 32-pixel tile.  Stored like that every lane would write 16-byte
  // pieces of 32 different channel planes per instruction (partial cache lines; MODE 2: 80 of 140 us).  So each
  // wave transposes its tile through LDS ([channel][NPX output floats] rows) and then stores whole 128-byte
  // rows: NPX/4 lanes cover one channel row with 16 bytes each, 64/(NPX/4) channels per instruction.
  // Bias, activation (+ act') and the act' multiply of dgrad are applied on the way out.
  constexpr int NPX = (MODE == 2) ? 64 : 32;   // output floats per channel, wave and pass (MODE 2: one row parity)
  constexpr int ES = NPX + 4;                  // LDS row stride (16-byte rows)
  constexpr int QPR = NPX / 4;                 // float4 quads per row
  constexpr int CPI = 64 / QPR;                // channels per store instruction
  float* Ew = smem + wave * 32 * ES;
  if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) return;
  const int eq = lane % QPR, ec = lane / QPR;
  // position of this lane's quad: MODE 0/1: 4 output pixels = base positions 4eq..4eq+3 ; MODE 2: 4 output floats
  // = base positions 2eq, 2eq+1 (x parity interleaved)
  const long long Pq = p0 + wp * 32 + ((MODE == 2) ? 2 * eq : 4 * eq);
  const bool pok = Pq < g.total_base;
  const long long nq = Pq >> g.lgBHW;
  const int remq = (int)(Pq & (BHW - 1));
  const int byq = remq >> g.lgBW, bxq = remq & (g.BW - 1);
  // (channel tile i, row parity py) passes, unrolled at compile time: with runtime loop indices the accumulator array
  // of the MODE 2 / MT 2 variant was indexed dynamically and went to scratch memory
  constexpr int NPY = (MODE == 2) ? 2 : 1;
  gmk_static_for<0, MT * NPY>([&](auto ip_) {
    constexpr int i = decltype(ip_)::value / NPY, py = decltype(ip_)::value % NPY;
    {
      // bias of this lane's channels and, for dgrad, the act' factors of its rows: ALL fetched before the math, so the
      // HBM latency of the act' read is paid once per pass and not once per store instruction
      float bvs[32 / CPI];
      long long idxs[32 / CPI];
      float4 dds[32 / CPI];
#pragma unroll
      for (int it = 0; it < 32 / CPI; ++it) {
        const int m = m0 + (wm * MT + i) * 32 + ec + it * CPI;
        const int mc = m < g.M ? m : g.M - 1;  // clamped for the prefetch; rows beyond M are not stored
        bvs[it] = (g.epi == 0 && g.bias) ? g.bias[mc] : 0.f;
        if (MODE != 2) idxs[it] = (nq * g.M + mc) * (long long)BHW + remq;
        else idxs[it] = (nq * g.M + mc) * (long long)(4 * BHW) + (long long)(2 * byq + py) * (2 * g.BW) + 2 * bxq;
        dds[it] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (g.epi != 0 && g.dact_in && pok) dds[it] = *(const float4*)(g.dact_in + idxs[it]);
      }
      gmk_lds_barrier();  // main-loop LDS reads / the previous pass's reads are done (the prefetches above stay in flight)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        if (MODE != 2) {
          *(float4*)(Ew + j * ES + 8 * k4 + 4 * h) =
              make_float4(acc[0][i][4 * k4], acc[0][i][4 * k4 + 1], acc[0][i][4 * k4 + 2], acc[0][i][4 * k4 + 3]);
        } else {
          float* d = Ew + j * ES + 16 * k4 + 8 * h;
          *(float4*)d = make_float4(acc[py * 2][i][4 * k4], acc[py * 2 + 1][i][4 * k4], acc[py * 2][i][4 * k4 + 1],
                                    acc[py * 2 + 1][i][4 * k4 + 1]);
          *(float4*)(d + 4) = make_float4(acc[py * 2][i][4 * k4 + 2], acc[py * 2 + 1][i][4 * k4 + 2],
                                          acc[py * 2][i][4 * k4 + 3], acc[py * 2 + 1][i][4 * k4 + 3]);
        }
      }
      gmk_lds_barrier();
#ifdef GMK_CONV_TRACE
      if (g.trace && tid == 0 && py < 2 && i == 0) g.trace[4ull * 65536 + 8ull * (blockIdx.x + blockIdx.y * gridDim.x) + 2 * py] = wall_clock64();
#endif
      // the activation switch is hoisted out of the store loop: GELU (the reference default) and identity
      // get straight-line code, the other activations share the generic switch
      auto store_rows = [&](auto actf) {
#pragma unroll
        for (int it = 0; it < 32 / CPI; ++it) {
          const int c = ec + it * CPI;
          const int m = m0 + (wm * MT + i) * 32 + c;
          if (m >= g.M || !pok) continue;  // (inside store_rows' own loop)
          float4 v = *(const float4*)(Ew + c * ES + 4 * eq);
          const long long idx = idxs[it];
          if (g.epi == 0) {
            const float bv = bvs[it];
            float4 a, d;
            actf(make_float4(v.x + bv, v.y + bv, v.z + bv, v.w + bv), a, d);
            *(float4*)(g.y + idx) = a;
            if (g.dact_out) *(float4*)(g.dact_out + idx) = d;
          } else {
            const float4 dd = dds[it];
            v.x *= dd.x; v.y *= dd.y; v.z *= dd.z; v.w *= dd.w;
            *(float4*)(g.y + idx) = v;
          }
        }
      };
      if (g.epi != 0) store_rows([](const float4& z, float4& a, float4& d) { a = z; d = z; });   // (unused on the dgrad path)
      else GMK_WITH_ACT(g.act, store_rows)
#ifdef GMK_CONV_TRACE
      if (g.trace && tid == 0 && py < 2 && i == 0) g.trace[4ull * 65536 + 8ull * (blockIdx.x + blockIdx.y * gridDim.x) + 2 * py + 1] = wall_clock64();
#endif
    }
  });
#ifdef GMK_CONV_TRACE
  if (g.trace && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's stores are acknowledged
    const unsigned long long tr_t2 = wall_clock64();
    const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID[3:0]
    unsigned long long* t = g.trace + 4ull * lin;
    t[0] = ((unsigned long long)xcc << 32) | hw; t[1] = tr_t0; t[2] = tr_t1; t[3] = tr_t2;
  }
#endif
}

// Host-side launcher (one instantiation set per kernel size, defined in conv_igemm_k{3,5}.hip)
int gmk_launch_conv_k3(ConvGeom& g, int mode, hipStream_t s);
int gmk_launch_conv_k5(ConvGeom& g, int mode, hipStream_t s);

template <int KS>
static int gmk_launch_conv_impl(ConvGeom& g, int mode, hipStream_t s) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
  const int BH = (1 << g.lgBHW) / g.BW;
  // tile choice (measured per mode at B = 256, tools/kbench.py under rocprofv3):
  //   MODE 0/1, >= 512 tiles of 128 positions: 128 positions x 64 channels (WM 1, MT 2): two MFMAs per patch read
  //   MODE 1 small, everything with <= 32 output channels ... : 64 x 64 (WM 2, MT 1) resp. 128 x 32 (WM 1, MT 1)
  //   MODE 0 small: 128 x 32 (1 us faster than 64 x 64 on the 8x8 layers)
  //   MODE 2: 128 positions x 32 channels (WM 1, MT 1): its four parity accumulators rule out MT 2 (spills), and
  //   against 64 x 64 the weight slice staged per chunk halves: 89 -> 65 us for the 16->32 transposed conv.
  int WM, MT;
  if (g.M <= 32) { WM = 1; MT = 1; }
  else if (mode == 2) { WM = 1; MT = 1; }   // MT 2 (339 registers, one wave per SIMD) measured in round 3: 73 against 68 us
  else if (g.total_base / 128 >= 512) { WM = 1; MT = 2; }
  else if (mode == 0) { WM = 1; MT = 1; }
  else { WM = 2; MT = 1; }
  const int WP = 4 / WM, PT = WP * 32, MB = WM * MT * 32;
  if (g.BW > PT) return GMK_E_SHAPE;
  const int CK = (mode == 0 && g.C < 8) ? 4 : 8;
  const int bhw = 1 << g.lgBHW;
  g.seg_px = PT < bhw ? PT : bhw;
  g.lg_seg_px = gmk_ilog2(g.seg_px);
  g.nseg = PT / g.seg_px;
  const int seg_rows = g.seg_px / g.BW;
  if (mode == 0) { g.PR = seg_rows + KS - 1; g.PC = g.BW + KS - 1; }
  else if (mode == 1) { g.PR = 2 * seg_rows + KS - 2; g.PC = 2 * g.BW + KS - 2; }
  else { const int tlo = (KS - 1 - PAD) / 2, thi = (PAD + 1) / 2; g.PR = seg_rows + tlo + thi; g.PC = g.BW + tlo + thi; }
  g.PS = g.PR * g.PC;
  g.npos = g.nseg * g.PR * g.PC;
  if (g.npos > 256 * (mode == 1 ? 4 : 2)) return GMK_E_SHAPE;
  (void)BH;
  size_t lds = 2 * sizeof(float) * ((size_t)MB * ((CK * KK) | 1) + (size_t)g.nseg * CK * g.PS);  // double-buffered
  const size_t lds_epi = sizeof(float) * 4 * 32 * ((mode == 2 ? 64 : 32) + 4);  // per-wave transpose tiles of the epilogue
  if (lds < lds_epi) lds = lds_epi;
  if (lds > 160 * 1024) return GMK_E_SHAPE;
  dim3 grid((unsigned)((g.total_base + PT - 1) / PT), (unsigned)((g.M + MB - 1) / MB));
  const int nslot = (g.npos + 255) / 256;
#ifdef GMK_CONV_TRACE
  unsigned long long* gmk_conv_trace_buffer();
  g.trace = gmk_conv_trace_buffer();
#endif
#define GMK_CONV_LAUNCH(MODE_, CK_, WM_, MT_)                                                              \
  do {                                                                                                     \
    auto kern = (nslot <= GMK_NS_LO(MODE_)) ? conv_igemm_kernel<MODE_, KS, CK_, WM_, MT_, GMK_NS_LO(MODE_)> \
                                            : conv_igemm_kernel<MODE_, KS, CK_, WM_, MT_, GMK_NS_HI(MODE_)>; \
    if (lds > 64 * 1024) {                                                                                 \
      hipError_t e = gmk_ensure_dyn_lds((const void*)kern, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                  \
    }                                                                                                      \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);                                                  \
  } while (0)
#define GMK_CONV_TILE(MODE_, CK_)                                 \
  do {                                                            \
    if (WM == 1 && MT == 2) GMK_CONV_LAUNCH(MODE_, CK_, 1, 2);    \
    else if (WM == 2) GMK_CONV_LAUNCH(MODE_, CK_, 2, 1);          \
    else GMK_CONV_LAUNCH(MODE_, CK_, 1, 1);                       \
  } while (0)
  if (mode == 0) { if (CK == 4) GMK_CONV_TILE(0, 4); else GMK_CONV_TILE(0, 8); }
  else if (mode == 1) GMK_CONV_TILE(1, 8);
  else GMK_CONV_TILE(2, 8);
#undef GMK_CONV_TILE
#undef GMK_CONV_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}
#ifdef GMK_CONV_TRACE
unsigned long long* gmk_conv_trace_buffer();   // conv_api.hip (tools-only build)
#endif

// conv_wgrad16.h -- 3x3 weight gradient of conv / transposed conv on v_mfma_f32_16x16x4_f32, no in-block reduction.
//
// Same GEMM view, operands, chunking over pixels, slab format and reduce kernel as conv_wgrad.h:
//   dW[cs][cb][t] = sum_{n, sy, sx}  S[n][cs][sy][sx] * B[n][cb][sy*SB + ky - 1][sx*SB + kx - 1]
// What changes is how a block's four waves share the work.  conv_wgrad.h gives every wave a full 32 x 32 x 9 tile
// (144 accumulator registers, one wave per SIMD) over a quarter of each chunk's pixels, sums the four partial tiles
// through LDS at the end (7 us of a 50 us launch) and needs two barriers per chunk around an un-overlapped staging
// phase.  Here each wave owns ONE 16 x 16 quadrant of the block's 32 x 32 tile with all 9 taps (36 accumulator
// registers) and reduces ALL pixels of every chunk, so
//   * there is nothing to combine inside the block: the accumulators go straight to the slab;
//   * ~110 registers per lane: the kernel co-resides with two dgrad waves per SIMD on the other stream (the 320-register
//     kernel left room for one, and none for its own 24-register reduce kernel, which waited 15 us on average);
//   * LDS is double buffered: chunk i+1 is written (and chunk i+2 fetched into the same registers, one iteration ahead,
//     exact vmcnt counting) between the MFMA k-steps of chunk i; ONE LDS-only barrier per chunk.
// fp32 MFMA issue is shared with the vector ALU on gfx950 (tools/mfma_valu.hip), so the k loop holds only ds_read_b32,
// v_mfma and the staging instructions: every operand address is <per-lane base> + immediate.
//
// LDS images (per buffer).  MFMA lane l supplies row i = l & 15 and reduction index k = l >> 4 (pixel 4s + k of k-step s):
//   S tile  [32][PX + 2]   : rows 8-byte aligned, row stride = 2 (mod 4) words -> the 32 lanes of a half-wave (16 rows x
//                            2 consecutive pixels) hit 32 distinct banks without any swizzle;
//   B patch [32][PSb]      : PSb = 2 (mod 4) likewise.  SB = 1: patch rows [4 zero | BWt image columns | 4 zero].
//                            SB = 2: the image columns are de-interleaved, [3 unused, O[-1] = 0 | odd columns O | even columns E]
//                            so that the stride-2 operand walk is unit stride in LDS: tap kx of S column x reads
//                            O[x-1], E[x], O[x] for kx = 0, 1, 2.
// Staging is float4 buffer loads (out-of-image rows and ragged channel tiles pass an out-of-range offset and read 0:
// no select on the data path) and one ds_write2_b64 per float4.
#pragma once
#include "conv_wgrad.h"

// timing ablations for A/B builds (make EXTRA=-DGMK_WG16_ABL=n; results are WRONG when set): 1 = no vmcnt waits,
// 2 = no per-chunk barrier, 4 = no staging (LDS writes + refill loads) inside the k loop
#ifndef GMK_WG16_ABL
#define GMK_WG16_ABL 0
#endif

typedef int gmk_i32x4 __attribute__((ext_vector_type(4)));
typedef float gmk_f32x2 __attribute__((ext_vector_type(2)));

struct Wg16Geo { int PX, rows, PR, PC, PSb, SS, npos, nslot, ns, ksteps, buf; };
constexpr Wg16Geo gmk_wg16_geo(int SW, int sb) {
  Wg16Geo w{};
  int px = sb == 1 ? 64 : 32;   // pixels per chunk (whole S rows)
  if (px < SW) px = SW;
  w.PX = px;
  w.rows = px / SW;
  w.PR = sb == 1 ? w.rows + 2 : 2 * w.rows + 1;
  w.PC = sb == 1 ? SW + 8 : 2 * SW + 4;
  int ps = w.PR * w.PC;
  while (ps % 4 != 2) ++ps;
  w.PSb = ps;
  w.SS = px + 2;
  w.npos = w.PR * (SW * sb / 4);
  w.nslot = (GMK_WG_T * w.npos + 255) / 256;
  w.ns = (GMK_WG_T * px / 4) / 256;
  w.ksteps = px / 4;
  // floats per buffer: S tile, patch, and a trash area behind the patch for dead staging slots (a slot writes two 8-byte
  // pieces, SW words apart when the columns are de-interleaved)
  w.buf = GMK_WG_T * w.SS + GMK_WG_T * w.PSb + (sb == 1 ? 4 : SW + 4);
  w.buf = (w.buf + 3) / 4 * 4;
  return w;
}

static inline bool gmk_wg16_supported(int N, int Cs, int Cb, int SH, int SW, int sb) {
  if (!gmk_is_pow2(SH) || !gmk_is_pow2(SW) || SW < 8 || SW > (sb == 1 ? 64 : 32)) return false;
  const Wg16Geo w = gmk_wg16_geo(SW, sb);
  if ((long long)SH * SW < w.PX) return false;
  // 32-bit buffer offsets: both tensors below 2 GiB
  const long long sbytes = 4ll * N * Cs * SH * SW, bbytes = 4ll * N * Cb * SH * SW * sb * sb;
  return sbytes < 0x7FFF0000ll && bbytes < 0x7FFF0000ll;
}

__device__ __forceinline__ gmk_i32x4 gmk_raw_srd(const void* base, long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  gmk_i32x4 r;
  r.x = (int)(unsigned)a;
  r.y = (int)((unsigned)(a >> 32) & 0xFFFFu);
  r.z = (int)(bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : bytes);
  r.w = 0x00020000;
  return r;
}

template <int SB, int LGSW, int BM /* bias mode of the launch: 0 none, 1 dY = small tensor, 2 dY = big tensor */>
__global__ __launch_bounds__(256) void conv_wgrad16_kernel(const WgradGeom g) {
  constexpr int NT = 9, PAD = 1, T = GMK_WG_T;
  constexpr int SW = 1 << LGSW, BWt = SW * SB, QPR = BWt / 4;
  constexpr Wg16Geo GEO = gmk_wg16_geo(SW, SB);
  constexpr int PX = GEO.PX, PR = GEO.PR, PC = GEO.PC, PSb = GEO.PSb, SS = GEO.SS, NPOS = GEO.npos;
  constexpr int NSLOT = GEO.nslot, NS = GEO.ns, KSTEPS = GEO.ksteps, NSTG = NS + NSLOT;
  constexpr int BUFB = GEO.buf * 4;        // bytes per LDS buffer
  constexpr int PATCH0 = T * SS * 4;       // byte offset of the patch inside a buffer
  static_assert(NS >= 1 && NSTG <= KSTEPS, "one staging slot per k-step");
  static_assert(SW >= 4 && (KSTEPS * 4) % SW == 0, "a k-step's 4 pixels lie in one S row; chunks are whole rows");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4, qs = wave >> 1, qb = wave & 1;
  const int tile = blockIdx.y;
  const int cs0 = (tile / g.cbT) * T, cb0 = (tile % g.cbT) * T;
  const int SHW = 1 << g.lgSHW, BHW = g.BHt * BWt;
  const int cmax_s = (g.Cs - cs0) < T ? (g.Cs - cs0) : T, cmax_b = (g.Cb - cb0) < T ? (g.Cb - cb0) : T;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias-gradient partials ride on the staging registers (a slot always carries one fixed channel); only the blocks
  // whose slab tail the reduce kernel reads keep them (the others multiply by 0)
  const bool bias1 = BM == 1 && (tile % g.cbT) == 0;  // dY is the small tensor
  const bool bias2 = BM == 2 && (tile / g.cbT) == 0;  // dY is the big tensor

  // ---- staging slots: constant per thread ------------------------------------------------------------
  // slot u < NS: S quad; slot NS + k: patch quad.  voff = byte offset for the buffer load (GMK_OOB: reads 0),
  // loff = LDS byte offset inside a buffer, prel = patch row - PAD (valid rows satisfy 0 <= r0*SB + prel < BHt).
  int voff[NSTG], loff[NSTG], prel[NSLOT];
  float bmul[NSTG];   // bias partials: 1 for quads that belong to this chunk's own dY pixels
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int e = tid + u * 256;
    const int c = e / (PX / 4), p4 = e % (PX / 4);
    voff[u] = c < cmax_s ? ((c << g.lgSHW) + 4 * p4) * 4 : GMK_OOB;
    loff[u] = (c * SS + 4 * p4) * 4;
    bmul[u] = bias1 ? 1.f : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int e = tid + k * 256;
    const int c = e / NPOS, r = e % NPOS;
    const int prow = r / QPR, q = r % QPR;
    const bool live = c < cmax_b;   // also false for e >= 32 * NPOS
    voff[NS + k] = live ? (c * BHW + prow * BWt + 4 * q) * 4 : GMK_OOB;   // relative to (b - PAD rows), see the SRD below
    prel[k] = prow - PAD;
    const int col = SB == 1 ? 4 + 4 * q : 4 + 2 * q;
    loff[NS + k] = PATCH0 + (live ? c * PSb + prow * PC + col : T * PSb) * 4;   // dead slots: the trash area behind the patch
    bmul[NS + k] = (bias2 && live && prow >= PAD && prow < PAD + (PX >> LGSW) * SB) ? 1.f : 0.f;
  }
  // ---- MFMA operand bases (bytes, buffer 0) ------------------------------------------------------------
  const unsigned lds0 = gmk_lds_addr(smem);
  const unsigned addrA0 = lds0 + ((qs * 16 + i16) * SS + kq) * 4;
  const unsigned addrB0 = lds0 + PATCH0 + ((qb * 16 + i16) * PSb + kq) * 4;

  const gmk_i32x4 srd_s = gmk_raw_srd(g.s, 4ll * g.N * g.Cs * SHW);
  const gmk_i32x4 srd_b = gmk_raw_srd(g.b - PAD * BWt, 4ll * g.N * g.Cb * BHW + 4ll * PAD * BWt);

  f32x4 stg[NSTG];
  float bsum[NSTG];
#pragma unroll
  for (int u = 0; u < NSTG; ++u) { bsum[u] = 0.f; stg[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const long long last_chunk = g.total_chunks - 1;
  // uniform per-chunk offsets: soff_s, soff_b (bytes), r0s = first image row of the patch's row PAD.  A chunk index past
  // the end (the look-ahead of a block's last iterations) gets descriptors with zero records: every lane reads 0, so
  // nothing is fetched, and what lands in LDS / the bias sums is harmless.
  auto chunk_soff = [&](long long ch, int& so_s, int& so_b, int& r0s, gmk_i32x4& rs, gmk_i32x4& rb_) {
    rs = srd_s; rb_ = srd_b;
    if (ch > last_chunk) { rs.z = 0; rb_.z = 0; }
    ch = ch < last_chunk ? ch : last_chunk;
    const long long pix = ch * PX;
    const int n = (int)(pix >> g.lgSHW);
    const int pin = (int)(pix & (SHW - 1));
    const int r0 = pin >> LGSW;
    so_s = (((n * g.Cs + cs0) << g.lgSHW) + pin) * 4;
    so_b = ((n * g.Cb + cb0) * BHW + r0 * SB * BWt) * 4;
    r0s = r0 * SB;
  };
  // The loads are asynchronous and invisible to the compiler: "+v" keeps every staging register in ONE physical register
  // for its whole life (with "=v" the allocator may give the new value another register and copy it back at the loop
  // edge -- before the data has arrived).
  auto issue_load = [&](auto u_, const gmk_i32x4 rs, const gmk_i32x4 rb_, int so_s, int so_b, int r0s) {
    constexpr int u = decltype(u_)::value;
    (void)prel; (void)voff; (void)stg; (void)rs; (void)rb_; (void)so_s; (void)so_b; (void)r0s;  // capture (uses sit in if-constexpr branches)
    if constexpr (u < NS) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(stg[u]) : "v"(voff[u]), "s"(rs), "s"(so_s) : "memory");
    } else {
      const int iy = r0s + prel[u - NS];
      const int vo = ((unsigned)iy < (unsigned)g.BHt) ? voff[u] : GMK_OOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(stg[u]) : "v"(vo), "s"(rb_), "s"(so_b) : "memory");
    }
  };
  // write staging register u into the buffer at byte offset wbase (the registers already hold zeros where needed)
  auto write_slot = [&](auto u_, unsigned wbase) {
    constexpr int u = decltype(u_)::value;
    (void)bsum; (void)bmul; (void)loff; (void)lds0;
    gmk_pin(stg[u]);
    if constexpr ((BM == 1 && u < NS) || (BM == 2 && u >= NS)) {
      // before the register is handed to the next load: the old value must be dead by then
      float s4 = ((stg[u].x + stg[u].y) + stg[u].z) + stg[u].w;   // a chain: pairs would be SLP-packed into v_pk_add + moves
      bsum[u] = fmaf(s4, bmul[u], bsum[u]);
      asm volatile("" : "+v"(bsum[u]));
    }
    const unsigned wa = lds0 + wbase + loff[u];
    if constexpr (u < NS || SB == 1) {
      const gmk_f32x2 lo = {stg[u].x, stg[u].y}, hi = {stg[u].z, stg[u].w};
      asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(wa), "v"(lo), "v"(hi) : "memory");
    } else {
      const gmk_f32x2 od = {stg[u].y, stg[u].w}, ev = {stg[u].x, stg[u].z};
      asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:%3" : : "v"(wa), "v"(od), "v"(ev), "n"(SW / 2) : "memory");
    }
  };

  // ---- prologue: chunk 0 -> buffer 0, chunk 1 in flight ------------------------------------------------
  long long ch = blockIdx.x;
  const long long G = g.G;
  {
    int so_s, so_b, r0s;
    gmk_i32x4 rs, rb_;
    chunk_soff(ch, so_s, so_b, r0s, rs, rb_);
    gmk_static_for<0, NSTG>([&](auto u_) { issue_load(u_, rs, rb_, so_s, so_b, r0s); });
    // under the first loads' latency: zero the halo words of every patch row in both buffers (image data never lands there)
    for (int e = tid; e < 2 * T * PR * (SB == 1 ? 2 : 1); e += 256) {
      const int bufi = e & 1, rr = e >> 1;
      const int side = SB == 1 ? rr & 1 : 0, r2 = SB == 1 ? rr >> 1 : rr;
      const int c = r2 / PR, prow = r2 % PR;
      float2* hz = (float2*)((char*)smem + bufi * BUFB + PATCH0 + (c * PSb + prow * PC + (side ? 4 + BWt : 0)) * 4);   // rows are 8-byte aligned
      hz[0] = make_float2(0.f, 0.f);
      hz[1] = make_float2(0.f, 0.f);
    }
    gmk_vm_wait_all();
    gmk_static_for<0, NSTG>([&](auto u_) { write_slot(u_, 0u); });
    chunk_soff(ch + G, so_s, so_b, r0s, rs, rb_);
    gmk_static_for<0, NSTG>([&](auto u_) { issue_load(u_, rs, rb_, so_s, so_b, r0s); });
  }
  gmk_lds_barrier();

  // ---- main loop ---------------------------------------------------------------------------------------
  unsigned cur = 0;   // byte offset of the buffer being multiplied
  for (; ch < g.total_chunks; ch += G) {
    int so_s, so_b, r0s;
    gmk_i32x4 rs, rb_;
    chunk_soff(ch + 2 * G, so_s, so_b, r0s, rs, rb_);
    const unsigned aA = addrA0 + cur, aB = addrB0 + cur, wbase = cur ^ (unsigned)BUFB;
    float ra[2], rb[2][NT];
    auto read_one = [&](auto idx_) {   // read r of k-step s: r = 0 the S operand, r = 1 + t the patch operand of tap t
      constexpr int idx = decltype(idx_)::value, s = idx / (1 + NT), r = idx % (1 + NT), slot = s & 1;
      (void)ra; (void)rb; (void)aA; (void)aB;   // capture them (their only uses sit in if-constexpr branches)
      if constexpr (r == 0) {
        GMK_DS_READ(ra[slot], aA, (4 * s) * 4);
      } else {
        constexpr int lr = (4 * s) >> LGSW, lcx = (4 * s) & (SW - 1);
        constexpr int t = r - 1, ky = t / 3, kx = t % 3;
        constexpr int K = SB == 1 ? (lr + ky) * PC + 4 + lcx + kx - PAD
                                  : (2 * lr + ky) * PC + (kx == 1 ? 4 + SW + lcx : (kx == 0 ? 3 + lcx : 4 + lcx));
        GMK_DS_READ(rb[slot][t], aB, K * 4);
      }
    };
    auto reads = [&](auto s_) {
      constexpr int s = decltype(s_)::value;
      gmk_static_for<s * (1 + NT), (s + 1) * (1 + NT)>(read_one);
    };
    // The k loop as one hand-ordered stream (every statement is asm volatile, so the order below IS the issue order):
    // a 16x16x4 MFMA holds the pipe for 32 cycles, which hides one ds_read and little else, so the ten operand reads of
    // step s+1 are spread over the nine MFMAs of step s (issued in a burst between two steps they stalled the pipe for
    // ~50 cycles per step: 58 % MFMA-busy under rocprofv3 --pmc).  Each MFMA is preceded by a counted wait for exactly
    // its own operand: read r+1 of this step, behind which 8-t reads of this step and t of the next are in flight.
    reads(std::integral_constant<int, 0>{});
    gmk_static_for<0, KSTEPS * NT>([&](auto i_) {
      constexpr int i = decltype(i_)::value, s = i / NT, t = i % NT, slot = s & 1;
      constexpr bool more = s + 1 < KSTEPS;
      (void)wbase; (void)rs; (void)rb_; (void)so_s; (void)so_b; (void)r0s; (void)ra; (void)rb; (void)acc;
      asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"((NT - 1 - t) + (more ? t : 0)) : "memory");
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(ra[slot]), "v"(rb[slot][t]));
      if constexpr (more) {
        read_one(std::integral_constant<int, (s + 1) * (1 + NT) + t>{});
        if constexpr (t == NT - 1) read_one(std::integral_constant<int, (s + 1) * (1 + NT) + NT>{});
      }
      // staging slot s of this chunk iteration: LDS write behind MFMA 2, refill load behind MFMA 5
      if constexpr (s < NSTG && t == 2 && !(GMK_WG16_ABL & 4)) {
        // register s holds chunk ch+G (requested one iteration ago): everything but the NSTG-1 younger requests is back
        if constexpr (!(GMK_WG16_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NSTG - 1) : "memory");
        write_slot(std::integral_constant<int, s>{}, wbase);
      }
      if constexpr (s < NSTG && t == 5 && !(GMK_WG16_ABL & 4))
        issue_load(std::integral_constant<int, s>{}, rs, rb_, so_s, so_b, r0s);   // chunk ch+2G
    });
    if constexpr (!(GMK_WG16_ABL & 2)) gmk_lds_barrier();   // the other buffer is complete, this one is free
    cur ^= (unsigned)BUFB;
  }
  gmk_vm_wait_all();   // drain the look-ahead loads before the registers die
  // the MFMAs above are inline asm: the compiler's hazard recognizer does not see that the accumulators were written by
  // the matrix pipe, so the wait states between the last MFMA and the first read of its result are spelled out
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");

  // ---- accumulators -> slab[(blockIdx.x * gridDim.y + blockIdx.y)][t][cb 32][cs 32] (+ 32 bias partials) --------
  // lane (i16, kq), register r holds cs = qs*16 + 4*kq + r, cb = qb*16 + i16: a float4 of four consecutive cs.
  constexpr int TILE = NT * T * T, SLAB = TILE + T;
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)SLAB;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    *(float4*)(out + t * T * T + (qb * 16 + i16) * T + qs * 16 + 4 * kq) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  // bias partials: per-slot sums -> LDS (indexed by staging element), one thread per channel adds its channel's
  // contiguous run of elements in a fixed order (block-uniform branch)
  if (bias1 || bias2) {
    float* R = smem;
    __syncthreads();
    if (bias1) {
#pragma unroll
      for (int u = 0; u < NS; ++u) R[tid + u * 256] = bsum[u];
    } else {
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) R[tid + k * 256] = bsum[NS + k];
    }
    __syncthreads();
    if (tid < T) {
      float sum = 0.f;
      const int run = bias1 ? PX / 4 : NPOS;
#pragma unroll 8
      for (int q = 0; q < run; ++q) sum += R[tid * run + q];
      out[TILE + tid] = sum;
    }
  }
}

// dW[cs][cb][t] = sum_g slab[g][tile][t][cb][cs], dbias from the 32 trailing floats: a fixed-order sum like
// conv_wgrad_reduce_kernel, without LDS (a launch that uses shared memory cannot start on a CU whose 160 KB are taken
// by the other stream's dgrad blocks) and with the G slabs of an element spread over 8 lane groups: next to the
// dgrad kernels every load takes 1-2 us, so the serial chain per thread has to be short (8 slabs = 2 batches of 4
// loads instead of 32 slabs: 14 -> 5 us in the overlapped step).  Lane = group * 8 + element quad: 8 lanes read 128
// contiguous bytes of one slab; three cross-lane adds combine the groups in the fixed order ((0+1)+(2+3))+((4+5)+(6+7)).
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_nolds_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                      float* __restrict__ dbias, int G, int ny, int Cs,
                                                                      int Cb, int cbT, int bias_mode) {
  constexpr int KK = KS * KS, NTG = KK / NT, T = GMK_WG_T, TILE = NT * T * T, SLAB = TILE + T;
  const int y = blockIdx.y;
  const int tg = y % NTG, tile = y / NTG;
  const int cs0 = (tile / cbT) * T, cb0 = (tile % cbT) * T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el = lane & 7, grp = lane >> 3;
  const int e = ((blockIdx.x * 4 + wave) * 8 + el) * 4;   // first of this lane's 4 slab elements (bias tail included)
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < SLAB) {
    const float* src = slab + ((size_t)grp * ny + y) * (size_t)SLAB + e;
    const size_t step = (size_t)8 * ny * SLAB;
#pragma unroll 4
    for (int gI = grp; gI < G; gI += 8, src += step) {
      const float4 v = *(const float4*)src;
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
#pragma unroll
  for (int o = 8; o <= 32; o <<= 1) {
    sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64);
    sum.z += __shfl_xor(sum.z, o, 64); sum.w += __shfl_xor(sum.w, o, 64);
  }
  if (grp != 0 || e >= SLAB) return;
  const float out[4] = {sum.x, sum.y, sum.z, sum.w};
  if (e < TILE) {
    const int t = e / (T * T), cb = (e / T) % T, cs = e % T;
    if (cb0 + cb < Cb) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cs0 + cs + k < Cs) dw[((size_t)(cs0 + cs + k) * Cb + cb0 + cb) * KK + tg * NT + t] = out[k];
    }
  } else if (dbias && tg == 0) {
    const int c = e - TILE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (bias_mode == 1 && (tile % cbT) == 0 && cs0 + c + k < Cs) dbias[cs0 + c + k] = out[k];
      if (bias_mode == 2 && (tile / cbT) == 0 && cb0 + c + k < Cb) dbias[cb0 + c + k] = out[k];
    }
  }
}

int gmk_launch_wgrad16(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);   // dw == nullptr: slabs only (deferred reduction)

// deferred reduction of up to GMK_WG16_BATCH layers' slabs in one launch
#define GMK_WG16_BATCH 8
struct Wg16ReduceItem { const float* slab; float* dw; float* dbias; int G, ny, Cs, Cb, cbT, bias_mode; };
struct Wg16ReduceBatch { Wg16ReduceItem item[GMK_WG16_BATCH]; };
int gmk_launch_wgrad16_reduce_batch(const Wg16ReduceBatch& b, int n, int max_ny, hipStream_t s);

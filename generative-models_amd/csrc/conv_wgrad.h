// conv_wgrad.h -- weight gradient of conv / transposed conv as a split-K GEMM on fp32 MFMA.
//
//   dW[cs][cb][t] = sum_{n, sy, sx}  S[n][cs][sy][sx] * B[n][cb][sy*SB + ky - PAD][sx*SB + kx - PAD]
// "small" tensor S and "big" tensor B:
//   conv   (stride SB): S = dY (cs = out channel), B = X  (cb = in channel)  -> dW is OIHW
//   deconv (stride 2) : S = X  (cs = in channel),  B = dY (cb = out channel) -> dW is IOHW
// GEMM view: M = cs (A operand), N = cb (B operand), K = pixels (reduction), one accumulator per tap.
//
// A block owns ONE 32 x 32 (cs x cb) output tile with all its taps and a group of pixel chunks; its four
// waves split every chunk's pixels (K) four ways and hold a full tile of accumulators each.  At the end the
// four partial tiles are summed through LDS in a fixed order and ONE slab per block is written, so the
// slab traffic is G x tiles x 9216 floats (4x less than one slab per 64x64 tile and block); a second
// kernel sums the slabs in a fixed order (deterministic, no atomics).
//
// LDS images: S tile [32][PX+4] and B patch [32][PR rows][BWt+8] (4 zero floats of halo on either side of every row,
// so that image rows start 16-byte aligned): both are filled with float4 global loads / ds_write_b128 and have
// their column index XOR-swizzled by (channel>>3)&3, which makes the per-lane (lane = channel) MFMA operand reads
// conflict-free although the row strides are multiples of 4 floats.
//
// The S width is a template parameter: on gfx950 the fp32 MFMA shares the vector ALU's issue (measured,
// tools/mfma_valu.hip: every VALU instruction between MFMAs costs ~4.5 cycles of MFMA time, nothing overlaps), so
// the MFMA loop must not compute addresses.  With the geometry known at compile time every operand read is
// <one of 6 per-lane base registers> + immediate offset, the loop is fully unrolled (no register rotation moves),
// and its only instructions are ds_read_b32 and v_mfma.
#pragma once
#include "gmk_common.h"

struct WgradGeom {
  const float* s; const float* b; float* slab;
  int N, Cs, Cb, SH, SW, BHt, BWt;
  int lgSW, lgSHW, lgBWt, P, lgP;
  long long total_chunks;
  int G;               // pixel-chunk groups (gridDim.x)
  int csT, cbT;        // 32-channel tiles
  int bias_mode;       // 0 none, 1 dY = small tensor, 2 dY = big tensor
  int dbg;             // profiling ablations (GMK_WG_DBG, 0 in production): 4 = no final reduction, 8 = no chunk loop
};

#define GMK_WG_T 32  // channel tile (both sides)

// ---- compile-time geometry of a chunk (shared by the kernel, the launcher and the workspace planner) ----
struct WgradPatch { int PX, PR, PC, PSb, npos, nslot; size_t lds; };
constexpr WgradPatch gmk_wgrad_patch_for(int p, int SW, int ksize, int sb) {
  WgradPatch w{};
  const int rows = p / SW, BWt = SW * sb;
  w.PX = p;
  w.PR = sb == 1 ? rows + ksize - 1 : 2 * rows + ksize - 2;
  w.PC = BWt + 8;
  w.PSb = ((w.PR * w.PC + 31) / 32) * 32 + 4;  // == 4 (mod 32), room for a trash quad behind the patch
  w.npos = w.PR * (BWt / 4);
  w.nslot = (GMK_WG_T * w.npos + 255) / 256;
  w.lds = sizeof(float) * ((size_t)GMK_WG_T * (p + 4) + (size_t)GMK_WG_T * w.PSb);
  return w;
}
// pixels per chunk: whole rows, 128 for stride 1 and 64 for stride 2 (the stride-2 patch is ~4x the chunk).  Larger
// chunks (256 / 128) and a one-barrier reduction over 4 parked tiles make this kernel ~8 % faster ALONE, but at ~150 KB
// of LDS it then owns its CU; at <= 74 KB the dgrad kernels of the main stream co-reside with it and the whole
// backward pass is 2 % faster (measured A/B on one device: 1.347 -> 1.323 ms/step).  Images are taken to be at least
// one chunk large (SH*SW >= PX, checked by the host).
#define GMK_WG_MAXSLOT 24
#ifndef GMK_WG_MAXPX_S1
#define GMK_WG_MAXPX_S1 128
#endif
#ifndef GMK_WG_MAXPX_S2
#define GMK_WG_MAXPX_S2 64
#endif
#define GMK_WG_RED_STRIDE 36  /* row stride (floats) of the transposed partial tiles parked in LDS by the in-block reduction */
constexpr WgradPatch gmk_wgrad_patch(int SW, int ksize, int sb) {
  int p = sb == 1 ? GMK_WG_MAXPX_S1 : GMK_WG_MAXPX_S2;
  if (p > SW * SW) p = SW * SW;
  while (p > SW) {
    const WgradPatch w = gmk_wgrad_patch_for(p, SW, ksize, sb);
    if (w.nslot <= GMK_WG_MAXSLOT && w.lds <= 150 * 1024) break;
    p >>= 1;
  }
  if (p < SW) p = SW;
  return gmk_wgrad_patch_for(p, SW, ksize, sb);
}

template <int KS, int SB, int NT /* taps per block: KS*KS or KS (one ky row) */, int LGSW /* log2 of the S width */>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradGeom g) {
  constexpr int KK = KS * KS, PAD = (KS - 1) / 2, NTG = KK / NT, T = GMK_WG_T;
  constexpr int SW = 1 << LGSW, BWt = SW * SB, QPR = BWt / 4;
  constexpr WgradPatch GEO = gmk_wgrad_patch(SW, KS, SB);
  constexpr int PX = GEO.PX, PR = GEO.PR, PC = GEO.PC, PSb = GEO.PSb, NPOS = GEO.npos, NSLOT = GEO.nslot;
  constexpr int SS = PX + 4;               // S tile row stride: 16-byte rows, columns XOR-swizzled by (row>>3)&3
  constexpr int NS = (T * PX / 4) / 256;   // float4 loads of the S tile per thread and chunk
  constexpr int WPX = PX / 4;              // pixels per wave and chunk
  static_assert(NS >= 1 && WPX >= 4 && (WPX % 4) == 0, "chunk too small");
  static_assert(WPX % SW == 0 || SW % WPX == 0, "a wave's pixels are whole rows or a part of one row");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Sl = smem;           // [32][SS]
  float* Bl = smem + T * SS;  // [32][PSb]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, jj = lane & 31;
  // blockIdx.y = ((csTile * cbT) + cbTile) * NTG + tapGroup
  const int tg = blockIdx.y % NTG;
  const int tile = blockIdx.y / NTG;
  const int cs0 = (tile / g.cbT) * T, cb0 = (tile % g.cbT) * T;
  const int SHW = 1 << g.lgSHW, BHW = g.BHt * BWt;
  const int cmax_s = (g.Cs - cs0) < T ? (g.Cs - cs0) : T, cmax_b = (g.Cb - cb0) < T ? (g.Cb - cb0) : T;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // bias-gradient partials, taken from the staging registers on their way to LDS: every staging slot of a thread
  // belongs to one fixed channel, so each slot keeps a running sum (bias_mode 1: dY is the small tensor -> S quads;
  // 2: dY is the big tensor -> the interior rows of the B patch, the halo rows belong to neighbouring chunks); the
  // per-slot sums are combined per channel in a fixed order after the main loop.
  const bool bias1 = g.bias_mode == 1 && tg == 0 && (tile % g.cbT) == 0;  // only the blocks the reduce kernel reads
  const bool bias2 = g.bias_mode == 2 && tg == 0 && (tile / g.cbT) == 0;

  // B patch: quad e = tid + k*256 -> (channel, patch row, quad of the image row); per slot: global offset, LDS
  // offset, and (row | swizzle<<8), 0x3FF for a dead slot
  int q_goff[NSLOT], q_loff[NSLOT], q_meta[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int e = tid + k * 256;
    const int c = e / NPOS, r = e % NPOS;
    const int prow = r / QPR, q = r % QPR;
    const bool live = c < cmax_b;  // also false for e >= 32 * NPOS
    q_goff[k] = live ? c * BHW + (prow - PAD) * BWt + 4 * q : 0;
    q_loff[k] = live ? c * PSb + prow * PC + 4 + 4 * q : PR * PC;  // dead slots write the trash quad of channel 0
    q_meta[k] = live ? (prow | (((c >> 3) & 3) << 8)) : 0x3FF;
  }
  // zero the halo quads of every patch row once (image rows are written at columns 4 .. 4+BWt-1 only)
  for (int e = tid; e < 2 * T * PR; e += 256) {
    const int side = e & 1, rr = e >> 1;
    const int c = rr / PR, prow = rr % PR;
    *(float4*)(Bl + c * PSb + prow * PC + (side ? 4 + BWt : 0)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- per-lane operand bases of the MFMA loop (everything else is an immediate) ----
  // This wave reduces pixels wp0 .. wp0+WPX-1 of the chunk; k-step s takes pixels wp0 + 2s + h.
  const int asw = (jj >> 3) & 3;
  const int wp0 = wave * WPX;
  // A: logical column wp0 + 2s + h of S row jj, stored at column ^ asw.  (2s + h) & 3 = (2s & 2) + h:
  const float* abase[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) abase[m] = Sl + jj * SS + wp0 + (((2 * m) + h) ^ asw);  // + (2s & ~3)
  // B: logical index lc + K(s,t), K compile time, lc = wave / lane part; stored at index ^ asw:
  //    (lc + 4a + m) ^ asw = ((lc + m) ^ asw) + 4a
  const int lc = (((wp0 >> LGSW) * SB) * PC) + ((wp0 & (SW - 1)) + h) * SB + ((NT == KK) ? 0 : tg * PC);  // tap-row blocks (5x5): ky = tg
  const float* bbase[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) bbase[m] = Bl + jj * PSb + ((lc + m) ^ asw);  // + (K & ~3) for K & 3 == m

  static_assert(NT == 9 || NT == 5, "operand wait lists are written for 9 (3x3) or 5 (one 5x5 tap row) taps per block");
  unsigned addrA[2], addrB[4];
#pragma unroll
  for (int m = 0; m < 2; ++m) addrA[m] = gmk_lds_addr(abase[m]);
#pragma unroll
  for (int m = 0; m < 4; ++m) addrB[m] = gmk_lds_addr(bbase[m]);
  f32x4 sr[NS];  // prefetch registers (gmk_gload_f4_async)
  f32x4 br[NSLOT];
  float bs_s[NS], bs_b[NSLOT];
#pragma unroll
  for (int u = 0; u < NS; ++u) bs_s[u] = 0.f;
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) bs_b[k] = 0.f;
  // branch-free batched loads of one pixel chunk into registers (a chunk index past the end re-reads the
  // last chunk; its contribution is skipped).  Patch rows outside the image re-read a valid quad, zeroed on the way
  // to LDS.
  const long long last_chunk = g.total_chunks - 1;
  int cur_r0 = 0;  // first S row of the chunk currently held in the prefetch registers
  auto load_chunk = [&](long long ch) {
    ch = ch < last_chunk ? ch : last_chunk;
    const long long pix = ch * PX;
    const long long n = pix >> g.lgSHW;
    const int pin = (int)(pix & (SHW - 1));
    const int r0 = pin >> LGSW;
    cur_r0 = r0;
    const float* sbase = g.s + ((n * g.Cs + cs0) << g.lgSHW) + pin;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      const int c = e / (PX / 4), p4 = e % (PX / 4);
      const int cc = c < cmax_s ? c : cmax_s - 1;  // ragged tile: re-read a valid channel, zeroed on the way to LDS
      gmk_gload_f4_async(sr[u], sbase + ((size_t)cc << g.lgSHW) + p4 * 4);
    }
    const float* bbase_g = g.b + (n * g.Cb + cb0) * (long long)BHW + (long long)(r0 * SB) * BWt;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int iy = r0 * SB + (q_meta[k] & 0xFF) - PAD;
      const bool ok = (q_meta[k] != 0x3FF) & (iy >= 0) & (iy < g.BHt);
      gmk_gload_f4_async(br[k], bbase_g + (ok ? q_goff[k] : 0));
    }
  };
  auto store_chunk = [&]() {
    gmk_vm_wait_all();  // issued a whole MFMA section ago
#pragma unroll
    for (int u = 0; u < NS; ++u) gmk_pin(sr[u]);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) gmk_pin(br[k]);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int e = tid + u * 256;
      const int c = e / (PX / 4), p4 = e % (PX / 4);
      const int sw = (c >> 3) & 3;
      float4 v = make_float4(sr[u].x, sr[u].y, sr[u].z, sr[u].w);
      if (c >= cmax_s) v = make_float4(0.f, 0.f, 0.f, 0.f);
      // element k of the quad goes to slot k ^ sw
      const float4 w1 = (sw & 1) ? make_float4(v.y, v.x, v.w, v.z) : v;
      const float4 w2 = (sw & 2) ? make_float4(w1.z, w1.w, w1.x, w1.y) : w1;
      *(float4*)(Sl + c * SS + p4 * 4) = w2;
    }
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int iy = cur_r0 * SB + (q_meta[k] & 0xFF) - PAD;
      const bool ok = (q_meta[k] != 0x3FF) & (iy >= 0) & (iy < g.BHt);
      const int sw = (q_meta[k] >> 8) & 3;
      float4 v = ok ? make_float4(br[k].x, br[k].y, br[k].z, br[k].w) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 w1 = (sw & 1) ? make_float4(v.y, v.x, v.w, v.z) : v;
      const float4 w2 = (sw & 2) ? make_float4(w1.z, w1.w, w1.x, w1.y) : w1;
      *(float4*)(Bl + q_loff[k]) = w2;
    }
    if (bias1) {
#pragma unroll
      for (int u = 0; u < NS; ++u) bs_s[u] += (sr[u].x + sr[u].y) + (sr[u].z + sr[u].w);
    }
    if (bias2) {
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) {
        const int prow = q_meta[k] & 0xFF;  // 0xFF for a dead slot
        const bool interior = prow >= PAD && prow < PAD + (PX >> LGSW) * SB;
        bs_b[k] += interior ? (br[k].x + br[k].y) + (br[k].z + br[k].w) : 0.f;
      }
    }
  };

  long long ch = blockIdx.x;
  if (g.dbg & 8) ch = g.total_chunks;
  load_chunk(ch);
  for (; ch < g.total_chunks; ch += g.G) {
    gmk_lds_barrier();  // previous chunk's LDS reads are done
    store_chunk();
    gmk_lds_barrier();
    load_chunk(ch + g.G);  // next chunk in flight under the MFMAs
    // MFMA over this wave's pixels: fully unrolled, every operand address = per-lane base + immediate.  The reads are
    // hand-placed ds_read_b32 one k-step (1 + NT reads) ahead of their MFMAs: left to the compiler they are merged into
    // ds_read2_b32 (which costs fp32-MFMA issue time on gfx950) and sunk next to their use.
    float ra[2], rbv[2][NT];
    auto read_one = [&](auto idx_) {  // read r of k-step s2: r = 0 the S operand, r = 1 + t the patch operand of tap t
      constexpr int idx = decltype(idx_)::value, s2 = idx / (1 + NT), r = idx % (1 + NT), slot = s2 & 1;
      (void)ra; (void)rbv; (void)addrA; (void)addrB;  // capture them (their only uses sit in if-constexpr branches)
      if constexpr (r == 0) {
        GMK_DS_READ(ra[slot], addrA[s2 & 1], ((2 * s2) & ~3) * 4);
      } else {
        // local pixel 2*s2 of the wave -> (row, column) relative to the wave's first pixel
        constexpr int lr = (WPX >= SW) ? (2 * s2) >> LGSW : 0;
        constexpr int lcx = (WPX >= SW) ? (2 * s2) & (SW - 1) : 2 * s2;
        constexpr int t = r - 1, ky = (NT == KK) ? t / KS : 0, kx = (NT == KK) ? t % KS : t;
        constexpr int K = (lr * SB + ky) * PC + lcx * SB + kx + 4 - PAD;
        GMK_DS_READ(rbv[slot][t], addrB[K & 3], (K & ~3) * 4);
      }
    };
    auto issue = [&](auto s_) {
      constexpr int s2 = decltype(s_)::value;
      gmk_static_for<s2 * (1 + NT), (s2 + 1) * (1 + NT)>(read_one);
    };
    issue(std::integral_constant<int, 0>{});
    gmk_static_for<0, WPX / 2>([&](auto s_) {
      constexpr int s2 = decltype(s_)::value, slot = s2 & 1;
      if constexpr (s2 + 1 < WPX / 2) issue(std::integral_constant<int, s2 + 1>{});
      constexpr int ahead = (s2 + 1 < WPX / 2) ? 1 + NT : 0;
      if constexpr (NT == 9)
        asm volatile("s_waitcnt lgkmcnt(%10)"
                     : "+v"(ra[slot]), "+v"(rbv[slot][0]), "+v"(rbv[slot][1]), "+v"(rbv[slot][2]), "+v"(rbv[slot][3]),
                       "+v"(rbv[slot][4]), "+v"(rbv[slot][5]), "+v"(rbv[slot][6]), "+v"(rbv[slot][7]), "+v"(rbv[slot][8])
                     : "n"(ahead));
      else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(ra[slot]), "+v"(rbv[slot][0]), "+v"(rbv[slot][1]), "+v"(rbv[slot][2]), "+v"(rbv[slot][3]),
                       "+v"(rbv[slot][4])
                     : "n"(ahead));
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot], rbv[slot][t], acc[t], 0, 0, 0);
    });
  }

  // The last iteration's look-ahead loads (a clamped re-read, never stored) are invisible to the compiler: without this
  // drain their data lands in registers that the reduction below already reuses.  Alone on the chip they return long
  // before, but under the two-stream backward pass they did not always (run-to-run differences of 2e-2 in the parameters
  // after 40 steps, tools/determinism.py); a kernel that ends with untracked loads in flight is wrong in any case.
  gmk_vm_wait_all();
  if (g.dbg & 4) return;
  // ---- combine the four waves' partial tiles in LDS (fixed order), one slab per block ----
  // slab[(blockIdx.x * gridDim.y + blockIdx.y)][t][cb 32][cs 32] followed by 32 bias partials.
  // A lane's accumulator quad (registers 4q .. 4q+3) is four consecutive cs of one cb, so with the tile kept
  // TRANSPOSED ([t][cb][cs], rows padded to 36 floats: 16-byte accesses of 8 consecutive lanes hit distinct banks)
  // every exchange is a ds_*_b128: 36 per lane and round instead of 144 dword accesses (this reduction was 11 us of
  // every launch).  Fixed order (w0 + w2) + (w1 + w3) through two regions: waves 0,1 park their tiles, waves 2,3 add theirs.
  constexpr int TILE = NT * T * T, SLAB = TILE + T, RS = GMK_WG_RED_STRIDE, RG = NT * T * RS;
  float* R = smem;  // [2][NT][32][RS]
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    __syncthreads();
    if ((wave >> 1) == ph) {
      float* Rw = R + (wave & 1) * RG + jj * RS + 4 * h;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4* p = (float4*)(Rw + t * T * RS + 8 * q);
          float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
          if (ph == 1) {
            const float4 o = *p;
            v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
          }
          *p = v;
        }
    }
  }
  __syncthreads();
  float* out = g.slab + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)SLAB;
  for (int e = tid * 4; e < TILE; e += 1024) {
    const int row = e >> 5, col = e & 31;  // (t * 32 + cb, cs)
    const float4 r0 = *(const float4*)(R + row * RS + col), r1 = *(const float4*)(R + RG + row * RS + col);
    *(float4*)(out + e) = make_float4(r0.x + r1.x, r0.y + r1.y, r0.z + r1.z, r0.w + r1.w);
  }
  __syncthreads();
  // bias partials: per-slot sums -> LDS (indexed by staging element), then one thread per channel adds its
  // channel's contiguous run of elements in a fixed order
  if (bias1 || bias2) {
    if (bias1) {
#pragma unroll
      for (int u = 0; u < NS; ++u) R[tid + u * 256] = bs_s[u];
    } else {
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) R[tid + k * 256] = bs_b[k];
    }
  }
  __syncthreads();
  if (tid < T) {
    float sum = 0.f;
    if (bias1) {
#pragma unroll 8
      for (int q = 0; q < PX / 4; ++q) sum += R[tid * (PX / 4) + q];
    } else if (bias2) {
#pragma unroll 8
      for (int q = 0; q < NPOS; ++q) sum += R[tid * NPOS + q];
    }
    out[TILE + tid] = sum;
  }
}

// dW[cs][cb][t] = sum_g slab[g][tile,tg][t][cb][cs]   (fixed order: 8 interleaved partial sums over g,
// combined in a fixed tree) ; dbias likewise from the 32 trailing floats of each slab.
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int G, int ny, int Cs,
                                                                int Cb, int cbT, int bias_mode) {
  // One thread owns 4 consecutive slab elements (16-byte loads), 8 threads split the G slabs of those elements:
  // few, fat blocks, because this kernel usually has to squeeze into what a large dgrad kernel of the main stream
  // leaves free on the CUs (many thin blocks took 35-40 us there instead of 4).
  constexpr int KK = KS * KS, NTG = KK / NT, T = GMK_WG_T, TILE = NT * T * T, SLAB = TILE + T;
  static_assert(TILE % 4 == 0 && SLAB % 4 == 0, "slabs are read as float4");
  __shared__ float4 red[8][32];
  const int y = blockIdx.y;  // (tile, tg)
  const int tg = y % NTG, tile = y / NTG;
  const int cs0 = (tile / cbT) * T, cb0 = (tile % cbT) * T;
  const int el = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const int e = (blockIdx.x * 32 + el) * 4;  // first of this thread's 4 elements (the bias tail included)
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < SLAB) {
    const float* src = slab + ((size_t)gq * ny + y) * (size_t)SLAB + e;
    const size_t step = (size_t)8 * ny * SLAB;
#pragma unroll 2
    for (int gI = gq; gI < G; gI += 8, src += step) {
      const float4 v = *(const float4*)src;
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  red[gq][el] = sum;
  __syncthreads();
  if (gq != 0 || e >= SLAB) return;
  auto add4 = [](const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  const float4 lo = add4(add4(red[0][el], red[1][el]), add4(red[2][el], red[3][el]));
  const float4 hi = add4(add4(red[4][el], red[5][el]), add4(red[6][el], red[7][el]));
  const float4 tot = add4(lo, hi);
  const float out[4] = {tot.x, tot.y, tot.z, tot.w};
  if (e < TILE) {
    const int t = e / (T * T), cb = (e / T) % T, cs = e % T;  // slabs are [t][cb][cs]: the 4 elements share t and cb
    if (cb0 + cb < Cb) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cs0 + cs + k < Cs) dw[((size_t)(cs0 + cs + k) * Cb + cb0 + cb) * KK + tg * NT + t] = out[k];
    }
  } else if (dbias && tg == 0) {
    const int c = e - TILE;
    // bias_mode 1: channel of the small tensor (only the cbTile == 0 blocks carry it); 2: of the big tensor
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (bias_mode == 1 && (tile % cbT) == 0 && cs0 + c + k < Cs) dbias[cs0 + c + k] = out[k];
      if (bias_mode == 2 && (tile / cbT) == 0 && cb0 + c + k < Cb) dbias[cb0 + c + k] = out[k];
    }
  }
}

int gmk_launch_wgrad_k3(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);
int gmk_launch_wgrad_k5(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s);

static inline bool gmk_wgrad_supported(int SH, int SW, int ksize, int sb) {
  if (!gmk_is_pow2(SH) || !gmk_is_pow2(SW) || SW < 8 || SW * sb > 128 || (SW * sb) % 4 != 0) return false;
  const WgradPatch w = gmk_wgrad_patch(SW, ksize, sb);
  return (long long)SH * SW >= w.PX && w.PX >= 32 && w.nslot <= GMK_WG_MAXSLOT && w.lds <= 160 * 1024;
}
static inline size_t gmk_wgrad_slab_floats(int ksize) {
  const int KK = ksize * ksize, NT = ksize == 3 ? KK : ksize;
  return (size_t)NT * GMK_WG_T * GMK_WG_T + GMK_WG_T;
}

template <int KS>
static int gmk_launch_wgrad_impl(WgradGeom& g, int sb, float* dw, float* dbias, hipStream_t s) {
  constexpr int KK = KS * KS, NT = (KS == 3) ? KK : KS, NTG = KK / NT, T = GMK_WG_T;
  if (!gmk_wgrad_supported(g.SH, g.SW, KS, sb)) return GMK_E_SHAPE;
  const WgradPatch w = gmk_wgrad_patch(g.SW, KS, sb);
  if (w.PX != g.P) return GMK_E_SHAPE;
  size_t lds = w.lds;
  const size_t red = sizeof(float) * 2 * NT * T * GMK_WG_RED_STRIDE;  // two parked partial tiles
  if (lds < red) lds = red;
  dim3 grid(g.G, g.csT * g.cbT * NTG);
#define GMK_WG_LAUNCH(SB_, LGSW_)                                                                                  \
  do {                                                                                                             \
    auto kern = conv_wgrad_kernel<KS, SB_, NT, LGSW_>;                                                             \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = gmk_ensure_dyn_lds((const void*)kern, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                          \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);                                                          \
  } while (0)
  if (sb == 1) {
    switch (g.lgSW) {
      case 3: GMK_WG_LAUNCH(1, 3); break;
      case 4: GMK_WG_LAUNCH(1, 4); break;
      case 5: GMK_WG_LAUNCH(1, 5); break;
      case 6: GMK_WG_LAUNCH(1, 6); break;
      default: return GMK_E_SHAPE;
    }
  } else {
    switch (g.lgSW) {
      case 3: GMK_WG_LAUNCH(2, 3); break;
      case 4: GMK_WG_LAUNCH(2, 4); break;
      case 5: GMK_WG_LAUNCH(2, 5); break;
      default: return GMK_E_SHAPE;
    }
  }
#undef GMK_WG_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  constexpr int SLAB = NT * T * T + T;
  hipLaunchKernelGGL((conv_wgrad_reduce_kernel<KS, NT>), dim3((SLAB + 127) / 128, grid.y), dim3(256), 0, s, g.slab, dw,
                     dbias, g.G, (int)grid.y, g.Cs, g.Cb, g.cbT, g.bias_mode);
  e = hipGetLastError();
  return e == hipSuccess ? GMK_OK : (int)e;
}

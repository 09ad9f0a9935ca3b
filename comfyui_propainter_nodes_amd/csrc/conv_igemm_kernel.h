// conv_igemm_kernel.h -- the flat-tile implicit-GEMM kernel template and its launcher family, shared by the four translation
// units that instantiate it (conv_igemm_hh.hip: f16 -> f16, _hf: f16 -> f32, _ff: f32 -> f32, _fh: f32 -> f16).
// One translation unit used to hold every instantiation: 4 of the library's 5 build minutes.
#pragma once
// (formerly the head of conv_igemm.hip) implicit-GEMM convolution for gfx950 MFMA (see include/propainter_mi355.h).
//
// Formulation:  D[cout][pixel] = sum_k Wp[cout][k] * X[pixel][k],   k = (tap, segment, channel)
// The weight tile is the MFMA A operand (rows = output channels), the gathered
// channels-last pixel tile is the B operand (cols = output pixels).  With the 16x16 C/D
// map (lane -> col = l&15, rows 4*(l>>4)+r) every lane ends up with 4 *consecutive output
// channels of one pixel*, i.e. one 8-byte (f16) / 16-byte (f32) channels-last store.
//
// Work-group = 256 threads = 4 waves arranged WC x WP; each wave owns (TC*16) x (TP*16).
// K is walked in chunks of 32 channels in (segment, channel chunk, tap) order -- the tap innermost, so that the
// taps of one chunk re-read the same lines of neighbouring pixels back to back.  One barrier per chunk.
// Staging: tiles whose rows fill whole 64-lane passes are copied by global_load_lds into a lane-linear LDS image
// (XOR swizzle applied to the SOURCE address), f16 / f32-half-chunk tiles in a 3-stage ring with counted
// s_waitcnt vmcnt(N) and a bare s_barrier (two chunks in flight); the other tiles go global -> VGPR -> LDS,
// double-buffered.  LDS rows are unpadded; 16-byte piece p of row r sits at p ^ swz(r) (conflict-free for the
// 16-lane service groups of ds_read_b128 on gfx950).
//
//   f16 : v_mfma_f32_16x16x32_f16, one MFMA per (tc,tp) per chunk.
//   f32 : v_mfma_f32_32x32x2_f32 (wave tile a multiple of 32x32) or v_mfma_f32_16x16x4_f32: exact f32 products.
//         A lane reads a float4 = 4 consecutive k and feeds element j to MFMA step j for BOTH operands, which
//         only permutes the summation order.
//   f32 on the f16 pipe (PP_F32X2): conv_split.hip.
#include "conv_common.h"

namespace pp {

// Staging geometry shared by the kernel and its launcher.
template <typename T, int BC, int BP, int NT = 256>
struct TileGeom {
  static constexpr int EPP = 16 / (int)sizeof(T);  // elements per 16-byte piece
  // K chunk staged per pipeline step: 32 channels of one (tap, segment); f32 tiles whose rows fill whole
  // 64-row DMA passes use 16-channel half chunks so that a 3-stage ring still fits 3 work-groups per CU
  // (the weight tile may be padded by up to a third to whole passes: 96 -> 128 rows)
  // (f32 only: measured on MI355X, the padded-DMA 96-wide f16 tile is slower than its register-staged form)
  static constexpr bool pad_ok(int rows, int rpp) {
    return sizeof(T) == 4 ? ((rows + rpp - 1) / rpp * rpp - rows) * 3 <= rows : rows % rpp == 0;
  }
  static constexpr int BK = (sizeof(T) == 4 && BP % 64 == 0 && pad_ok(BC, 64)) ? 16 : 32;
  static constexpr int PPR = BK / EPP;   // 16-byte pieces per tile row
  static constexpr int RPP = NT / PPR;   // tile rows filled per pass of the NT threads of the work-group
  // DMA: tiles staged with global_load_lds (no VGPR round trip, no ds_write) when every wave-instruction of the
  // staging pass covers whole tile rows of both tiles; otherwise global -> VGPR -> LDS.
  // A pixel tile shorter than one pass (32-pixel tiles) is copied by the first BP*PPR/64 waves' worth of lanes; the
  // other waves repeat the same copies (identical data, same LDS slots) so that every wave counts the same vmcnt.
  static constexpr bool XPARTIAL = (BP < RPP) && (RPP % BP == 0) && (BP * PPR >= 64);
  static constexpr bool DMA = ((BP % RPP == 0) || XPARTIAL) && pad_ok(BC, RPP);
  static constexpr int BCP = DMA ? (BC + RPP - 1) / RPP * RPP : BC;  // weight-tile rows allocated in LDS
  // Ring of NST stages, NST-1 chunks in flight across the single barrier per chunk: 3 when one stage is <= 16 KiB;
  // 6 for the f16 32-pixel tiles (launched when the grid is about one work-group per CU: nothing else hides the
  // global->LDS latency of the long serial K loop of the recurrences' convolutions)
  static constexpr int NST = !DMA ? 2 : (sizeof(T) == 2 && BP == 32) ? 6 : (sizeof(T) == 2 || BK == 16) ? 3 : 2;
};

template <typename T, typename OT, int WC, int WP, int TC, int TP, bool M32, int KC>
__global__ void __launch_bounds__(WC * WP * 64) conv_igemm_kernel(const ConvK p) {
  constexpr int NT = WC * WP * 64;  // threads per work-group (4 waves; 8 for the 256-channel tiles)
  typedef TileGeom<T, WC * TC * 16, WP * TP * 16, NT> G;
  constexpr int BK = G::BK;
  constexpr int CM = 32 / BK;    // pipeline steps per 32-channel chunk of the packed weights
  constexpr int EPP = G::EPP;
  constexpr int PPR = G::PPR;    // pieces per tile row (4 f16, 8 / 4 f32)
  constexpr int LDK = BK;                   // LDS row pitch in elements (no padding: XOR-swizzled pieces)
  // 16-byte piece p of tile row r is stored at piece p ^ swz(r): conflict-free for the ds_read_b128
  // lane groups AND the ds_write_b128 groups of gfx950 (f16: 4 pieces/row, f32: 8 pieces/row)
  constexpr int SWZ_MASK = PPR - 1;
  // swz(r) = (r >> SWZ_SHIFT) & (PPR - 1).  The 16-lane service groups of ds_read_b128 must hit 16 distinct
  // 16-byte bank slots: shift 1 does that for 64-byte rows read 16 rows x 4 k-groups (f16, 16x16 f32) and for
  // 128-byte rows read 32 rows x 2 k-halves (M32, BK 32); 64-byte rows read 32 x 2 (M32, BK 16) need shift 2.
  constexpr int SWZ_SHIFT = (M32 && PPR == 4) ? 2 : 1;
  constexpr int BC = WC * TC * 16;
  constexpr int BP = WP * TP * 16;
  constexpr int RPP = G::RPP;  // rows filled per pass
  constexpr int XPASS = (BP + RPP - 1) / RPP;
  constexpr int WPASS = (BC + RPP - 1) / RPP;
  constexpr int BCP = G::BCP;
  constexpr int STAGE = KC * (BP + BCP) * LDK;  // elements per pipeline stage (KC chunks per barrier)
  constexpr bool DMA = G::DMA;
  // NST == 3: two chunks in flight across the (single) barrier per chunk; the MFMA phase of one chunk is too
  // short to hide a global->LDS round trip with only one chunk ahead.
  constexpr int NST = G::NST;
  // global_load_lds instructions per thread per chunk (ablation builds count only what they issue)
  constexpr int NLOADS = XPASS + WPASS;
  typedef typename Frag<T>::piece piece_t;

  T* smem = reinterpret_cast<T*>(PP_DYN_SMEM);  // [2 stages][KC][ X: BP rows | W: BC rows ][LDK]

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wc = wave / WP;
  const int wp = wave % WP;
  const int z = (int)blockIdx.z;
  int tile_p, tile_c;
  flat_tile_of(p, BC, tile_p, tile_c);
  const int64_t p_base = (int64_t)tile_p * BP;
  const int c_base = tile_c * BC;

  const int pc = tid % PPR;   // LDS piece slot inside a tile row (lane-linear: slot index == tid within a pass)
  const int row0 = tid / PPR;
  // LDS slot (r, pc) holds global piece pc ^ swz(r); RPP is a multiple of 16, so swz(r) = swz(row0) for every pass
  const int pcs = pc ^ ((row0 >> SWZ_SHIFT) & SWZ_MASK);
  // pixel-tile row of this thread's piece (partial pass: the waves beyond the tile wrap around; BP is a multiple of
  // 16, so the swizzle of the wrapped row is the same)
  constexpr int XWAVES = G::XPARTIAL ? BP * PPR / 64 : NT / 64;
  const int wave_x = G::XPARTIAL ? wave % XWAVES : wave;
  const int row0x = G::XPARTIAL ? row0 % BP : row0;

  // ---- per-thread pixel rows of the X tile --------------------------------------------
  int py0[XPASS], px0[XPASS];
  int64_t pn[XPASS], prow[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int r = row0x + i * RPP;
    const int64_t m = p_base + r;
    const int64_t mm = (r < BP && m < p.M) ? m : p.M - 1;  // rows past M: clamped, results never stored
    const int wo = (int)(mm % p.Wo);
    const int64_t t = mm / p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    py0[i] = ho * p.sh - p.ph;
    px0[i] = wo * p.sw - p.pw;
    pn[i] = (int64_t)n * p.H * p.W;
    prow[i] = pn[i] + (int64_t)py0[i] * p.W + px0[i];  // pixel index of tap (0,0); may be "negative" for padded taps
  }
  // ---- per-thread weight rows (rows past Cout: clamped, results never stored) ------------
  const T* wbase = reinterpret_cast<const T*>(p.weight) + (int64_t)z * p.w_zoff;
  const T* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + row0 + i * RPP;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * EPP;
  }

  piece_t xreg[KC][XPASS];
  piece_t wreg[KC][WPASS];

  // K iterator (wave-uniform): (segment ; channel chunk inside the segment ; tap ky,kx) with the TAP INNERMOST:
  // the taps of one channel chunk re-read the same 128-byte lines of neighbouring pixels back to back (L1/L2 hits).
  // With the tap outermost a work-group streams all channels between two visits of a line, and the ~100 resident
  // work-groups of an XCD push each other's lines out of the 4 MiB L2 (every tap then refetches over the fabric).
  // The packed weights keep their [tap][segment][channel] order: the chunk's row offset is computed, not streamed.
  // Advanced incrementally (no integer divisions); segment parameters are picked with constant
  // indices only, so the kernel-argument arrays stay in SGPRs instead of being copied to scratch.
  int it_q = 0, it_ky = 0, it_kx = 0, it_seg = 0, it_rem = 0, it_sbase = 0;
  auto it_woff = [&]() PP_INLINE_LAMBDA { return (it_ky * p.kw + it_kx) * p.chunks_per_tap * 32 + it_sbase + it_rem * BK; };
  const T* it_base = reinterpret_cast<const T*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int it_C = p.in_C[0], it_ldc = p.in_ldc[0], it_chunks = p.seg_chunks[0] * CM;
  auto select_segment = [&](int seg) PP_INLINE_LAMBDA {
#pragma unroll
    for (int s = 0; s < PP_MAX_SEG; ++s) {
      if (seg == s) {
        it_base = reinterpret_cast<const T*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
        it_C = p.in_C[s];
        it_ldc = p.in_ldc[s];
        it_chunks = p.seg_chunks[s] * CM;
      }
    }
  };
  auto advance = [&]() PP_INLINE_LAMBDA {
    ++it_q;
    if (++it_kx == p.kw) {
      it_kx = 0;
      if (++it_ky == p.kh) {
        it_ky = 0;
        if (++it_rem == it_chunks) {
          it_rem = 0;
          it_sbase += it_chunks * BK;
          if (p.nseg > 1) select_segment(++it_seg);
        }
      }
    }
  };

  // global -> registers for the next chunk of the K iterator.  Every load is UNCONDITIONAL (a predicated
  // load costs an exec-mask branch per piece): out-of-image taps / padded channels read a safe address and
  // are zeroed by a select; tile rows past M / Cout read a clamped row, their results are never stored.
  auto load_chunk = [&](auto kci) PP_INLINE_LAMBDA {
    constexpr int kc = decltype(kci)::value;
    const bool live = (KC == 1) || (it_q < p.nchunks * CM);
    const int c0 = it_rem * BK + pcs * EPP;
    const bool cvalid = live && (c0 < it_C);
    const T* sbase = it_base;
    const int ldc = it_ldc;
    const int dy = it_ky * p.dh, dx = it_kx * p.dw;
    const int64_t tapoff = (int64_t)dy * p.W + dx;  // wave-uniform pixel offset of this tap
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int y = py0[i] + dy, x = px0[i] + dx;
      bool ok = cvalid;
      int64_t pix;
      if (p.pad_mode == PP_PAD_REPLICATE) {
        const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y);
        const int xc = x < 0 ? 0 : (x >= p.W ? p.W - 1 : x);
        pix = pn[i] + (int64_t)yc * p.W + xc;
      } else {
        ok = ok && (y >= 0) && (y < p.H) && (x >= 0) && (x < p.W);
        pix = prow[i] + tapoff;
      }
      const T* src = ok ? sbase + pix * ldc + c0 : sbase;
      piece_t v = *reinterpret_cast<const piece_t*>(src);
      if (!ok) {
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] = (T)0;
      }
      xreg[kc][i] = v;
    }
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      piece_t v = *reinterpret_cast<const piece_t*>(wrow[i] + (live ? it_woff() : 0));
      if (KC > 1 && !live) {
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] = (T)0;
      }
      wreg[kc][i] = v;
    }
    if (live) advance();
  };
  auto load_stage = [&]() PP_INLINE_LAMBDA { static_for<KC>([&](auto kci) { load_chunk(kci); }); };

  // DMA variant of load_chunk + store: the same source addresses, but every wave-instruction copies 64 x 16 bytes
  // straight into the lane-linear LDS image (slot index = i*256 + tid) of stage `buf`; zeros come from pp_zero16.
  auto dma_chunk = [&](int buf, auto kci) PP_INLINE_LAMBDA {
    constexpr int kc = decltype(kci)::value;
    T* xt = smem + buf * STAGE + kc * (BP + BCP) * LDK;
    T* wt = xt + BP * LDK;
    const int c0 = it_rem * BK + pcs * EPP;
    const bool cvalid = c0 < it_C;
    const T* sbase = it_base;
    const int ldc = it_ldc;
    const int dy = it_ky * p.dh, dx = it_kx * p.dw;
    const int64_t tapoff = (int64_t)dy * p.W + dx;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int y = py0[i] + dy, x = px0[i] + dx;
      bool ok = cvalid;
      int64_t pix;
      if (p.pad_mode == PP_PAD_REPLICATE) {
        const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y);
        const int xc = x < 0 ? 0 : (x >= p.W ? p.W - 1 : x);
        pix = pn[i] + (int64_t)yc * p.W + xc;
      } else {
        ok = ok && (y >= 0) && (y < p.H) && (x >= 0) && (x < p.W);
        pix = prow[i] + tapoff;
      }
      const void* src = ok ? static_cast<const void*>(sbase + pix * ldc + c0) : static_cast<const void*>(pp_zero16);
      glds16(src, xt + (i * NT + wave_x * 64) * EPP);
    }
    const int woff = it_woff();
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      glds16(wrow[i] + woff, wt + (i * NT + wave * 64) * EPP);
    advance();
  };
  auto dma_stage = [&](int buf) PP_INLINE_LAMBDA { static_for<KC>([&](auto kci) { dma_chunk(buf, kci); }); };

  auto store_stage = [&](int buf) PP_INLINE_LAMBDA {
    static_for<KC>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      T* xs = smem + buf * STAGE + kc * (BP + BCP) * LDK;
      T* ws = xs + BP * LDK;
#pragma unroll
      for (int i = 0; i < XPASS; ++i) {
        const int r = row0 + i * RPP;
        if (r < BP) *reinterpret_cast<piece_t*>(xs + r * LDK + pc * EPP) = xreg[kc][i];
      }
#pragma unroll
      for (int i = 0; i < WPASS; ++i) {
        const int r = row0 + i * RPP;
        if (r < BC) *reinterpret_cast<piece_t*>(ws + r * LDK + pc * EPP) = wreg[kc][i];
      }
    });
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  // M32 (f32 only): 32x32x2 MFMA tiles, (TC/2) x (TP/2) accumulators of 16 registers
  constexpr int TC2 = (TC + 1) / 2, TP2 = (TP + 1) / 2;
  f16v acc32[TC2][TP2];
#pragma unroll
  for (int a = 0; a < TC2; ++a)
#pragma unroll
    for (int b = 0; b < TP2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[a][b][r] = 0.f;
  const int r32 = lane & 31, kh32 = lane >> 5;

  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int fswz = (frow >> SWZ_SHIFT) & SWZ_MASK;  // tile rows are multiples of 16 apart: swizzle depends on frow only

  const int nstages = (p.nchunks * CM + KC - 1) / KC;
  if constexpr (DMA) {
    static_assert(KC == 1, "the DMA path stages exactly one live chunk per call");
    dma_stage(0);
  } else {
    load_stage();
    store_stage(0);
  }
  __syncthreads();

  auto compute = [&](int buf) PP_INLINE_LAMBDA {
    static_for<KC>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      const T* xt = smem + buf * STAGE + kc * (BP + BCP) * LDK;
      const T* wt = xt + BP * LDK;
      if constexpr (sizeof(T) == 2) {
        const T* xs = xt + (wp * TP * 16 + frow) * LDK;
        const T* ws = wt + (wc * TC * 16 + frow) * LDK;
        h8 af[TC], bf[TP];
#pragma unroll
        for (int a = 0; a < TC; ++a) af[a] = lds_frag(ws + a * 16 * LDK + (fgrp ^ fswz) * 8);
#pragma unroll
        for (int b = 0; b < TP; ++b) bf[b] = lds_frag(xs + b * 16 * LDK + (fgrp ^ fswz) * 8);
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
          for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af[a], bf[b], acc[a][b]);
      } else if constexpr (M32) {
        const T* xs32 = xt + (wp * TP * 16 + r32) * LDK;
        const T* ws32 = wt + (wc * TC * 16 + r32) * LDK;
#pragma unroll
        for (int sub = 0; sub < BK / 8; ++sub) {  // 8 k per sub-chunk: lane half kh32 supplies k = 4*kh32 + j at step j
          f4 af[TC2], bf[TP2];
#pragma unroll
          for (int a = 0; a < TC2; ++a)
            af[a] = *reinterpret_cast<const f4*>(ws32 + a * 32 * LDK + ((sub * 2 + kh32) ^ ((r32 >> SWZ_SHIFT) & SWZ_MASK)) * 4);
#pragma unroll
          for (int b = 0; b < TP2; ++b)
            bf[b] = *reinterpret_cast<const f4*>(xs32 + b * 32 * LDK + ((sub * 2 + kh32) ^ ((r32 >> SWZ_SHIFT) & SWZ_MASK)) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < TC2; ++a)
#pragma unroll
              for (int b = 0; b < TP2; ++b) acc32[a][b] = mfma_32x32x2_f32(af[a][j], bf[b][j], acc32[a][b]);
        }
      } else {
        const T* xs = xt + (wp * TP * 16 + frow) * LDK;
        const T* ws = wt + (wc * TC * 16 + frow) * LDK;
#pragma unroll
        for (int sub = 0; sub < BK / 16; ++sub) {
          f4 af[TC], bf[TP];
#pragma unroll
          for (int a = 0; a < TC; ++a)
            af[a] = *reinterpret_cast<const f4*>(ws + a * 16 * LDK + ((sub * 4 + fgrp) ^ fswz) * 4);
#pragma unroll
          for (int b = 0; b < TP; ++b)
            bf[b] = *reinterpret_cast<const f4*>(xs + b * 16 * LDK + ((sub * 4 + fgrp) ^ fswz) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
              for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x4_f32(af[a][j], bf[b][j], acc[a][b]);
        }
      }
    });

  };

  if constexpr (NST >= 3) {
    static_assert((NST - 2) * NLOADS <= 63, "vmcnt is a 6-bit counter");
    // prologue issued chunk 0 (and waited for it); put chunks 1 .. NST-2 in flight as well
#pragma unroll
    for (int j = 1; j < NST - 1; ++j)
      if (j < nstages) dma_stage(j);
    // this wave's copies of chunk qs have landed when at most min(NST-2, chunks after qs) later chunks are pending
    auto wait_landed = [&](int after) PP_INLINE_LAMBDA {
      static_for<NST - 1>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        if (after == c || (c == NST - 2 && after > c)) pp_wait_vmcnt<c * NLOADS>();
      });
    };
    int st = 0;  // qs % NST
    for (int qs = 0; qs < nstages; ++qs) {
      if (qs > 0) {
        wait_landed(nstages - 1 - qs);
        pp_barrier();  // every wave's part of chunk qs is visible; everyone is done reading stage (qs-1) % NST
      }
      if (qs + NST - 1 < nstages) dma_stage(st == 0 ? NST - 1 : st - 1);
      compute(st);
      st = st + 1 == NST ? 0 : st + 1;
    }
  } else {
    for (int qs = 0; qs < nstages; ++qs) {
      const int buf = qs & 1;
      if (qs + 1 < nstages) {
        if constexpr (DMA) dma_stage(buf ^ 1); else load_stage();
      }
      compute(buf);
      if constexpr (!DMA) {
        if (qs + 1 < nstages) store_stage(buf ^ 1);
      }
      __syncthreads();  // (DMA: the barrier's release also waits for the outstanding global_load_lds, vmcnt(0))
    }
  }

  // ---- epilogue -------------------------------------------------------------------------
  // (accumulator tiles are passed BY VALUE with compile-time indices: any runtime indexing of
  //  acc[][] would push the whole accumulator array to scratch memory)
  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  if constexpr (M32) {
    epilogue_quads<OT, TC2 * 4, TP2>(
        p, e,
        [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
          m = p_base + wp * TP * 16 + decltype(bi)::value * 32 + r32;
          ok = m < p.M;
        },
        [&](auto ai) PP_INLINE_LAMBDA {
          constexpr int a = decltype(ai)::value / 4, q = decltype(ai)::value % 4;
          return c_base + wc * TC * 16 + a * 32 + 8 * q + 4 * kh32;
        },
        [&](auto ai, auto bi) PP_INLINE_LAMBDA {
          constexpr int a = decltype(ai)::value / 4, q = decltype(ai)::value % 4, b = decltype(bi)::value;
          return f4{acc32[a][b][4 * q], acc32[a][b][4 * q + 1], acc32[a][b][4 * q + 2], acc32[a][b][4 * q + 3]};
        });
  } else {
    constexpr bool EPI_FITS = (size_t)WC * WP * epi_lds_wave_bytes<TC>() <= (size_t)NST * STAGE * sizeof(T);
    epilogue_any<OT, TC, TP, EPI_FITS>(
        p, e, reinterpret_cast<unsigned char*>(smem), wave, lane, c_base + wc * TC * 16,
        [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
          m = p_base + wp * TP * 16 + decltype(bi)::value * 16 + frow;
          ok = m < p.M;
        },
        [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
        [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; },
        [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
          m0 = p_base + wp * TP * 16 + decltype(bi)::value * 16;
          nvalid = (int)(p.M - m0 < 16 ? p.M - m0 : 16);
        });
  }
}

template <typename T, typename OT, int WC, int WP, int TC, int TP>
static int launch_cfg(void* stream, const ConvK& k, int Z) {
  // f32: 32x32x2 MFMA tiles whenever the per-wave tile is a multiple of 32x32
  constexpr bool M32 = (sizeof(T) == 4) && (TC % 2 == 0) && (TP % 2 == 0);
  // KC = 32-channel chunks staged per barrier.  Measured on MI355X (r01): KC = 2 for f16 halves the barriers
  // but doubles LDS per work-group (2 instead of 3-4 resident work-groups per CU) and is a net loss.
  constexpr int KC = 1;
  constexpr int BC = WC * TC * 16;
  constexpr int BP = WP * TP * 16;
  constexpr int NT = WC * WP * 64;
  typedef TileGeom<T, BC, BP, NT> G;
  const size_t smem = (size_t)G::NST * KC * (G::BCP + BP) * G::BK * sizeof(T);
  dim3 grid((unsigned)(((k.M + BP - 1) / BP) * ((k.Cout + BC - 1) / BC)), 1u, (unsigned)Z);
  PP_ALLOW_BIG_LDS((&conv_igemm_kernel<T, OT, WC, WP, TC, TP, M32, KC>), smem);
  PP_LAUNCH((conv_igemm_kernel<T, OT, WC, WP, TC, TP, M32, KC>), grid, dim3(NT), smem, stream, k);
  return pp_check_launch("pp_conv2d");
}

// Launchers of one (T, OT) kernel family for a wave arrangement WC x WP with TC x TP MFMA tiles per wave.
template <typename T, typename OT>
struct IgemmFamily {
  template <int WC, int WP, int TC, int TP>
  static int run(void* stream, const ConvK& k, int Z) { return launch_cfg<T, OT, WC, WP, TC, TP>(stream, k, Z); }
  static constexpr bool m32_wide96 = sizeof(T) == 4;
  static constexpr int xl_min_blocks = sizeof(T) == 2 ? 512 : 1024;
};
}  // namespace pp

// conv_halo_f16_kernel.h -- the f16 form of the halo-tile convolution (see conv_halo.hip for the idea and conv_halo_common.h for
// the shared geometry): stride-1 multi-tap convolutions of flow completion and the generator.
#pragma once
#include "conv_halo_common.h"

namespace pp {

// ---------------------------------------------------------------------------------------------------------
// f16 form: 64-byte LDS rows (32 channels), both operands copied by global_load_lds.  The halo tile of the NEXT channel
// chunk is copied into the second pixel stage while the current chunk's taps are multiplied (no register staging, no
// extra barrier); weights run two taps ahead in a 3-stage ring.  conv_igemm_kernel issues one 1-KiB copy instruction
// per 16 pixel rows of 64 bytes (half a cache line each) for EVERY tap -- the r01 ablation attributes 44 % of the f16
// 3x3 convolution to that gather; here a chunk's pixels are copied once for all its taps.
// swz(r) = (r >> 1) & 3 is conflict-free for 16 consecutive rows starting at any row (brute-forced against the
// ds_read_b128 service groups).
template <typename OT, int WC, int WP, int TC, int TP, int XPASS>
__global__ void __launch_bounds__(WC * WP * 64) conv_halo_f16_kernel(const ConvK p, const HaloGeom g) {
  typedef half_t T;
  constexpr int NT = WC * WP * 64;
  constexpr int TH = WP * TP;
  constexpr int ROWB = 64;                       // bytes per LDS row (32 channels)
  constexpr int RPP = NT / 4;                    // rows per copy pass (4 pieces of 16 bytes per row)
  constexpr int BC = WC * TC * 16;
  constexpr int WPASS = (BC + RPP - 1) / RPP;
  constexpr int BCP = WPASS * RPP;
  constexpr int XSTAGE = XPASS * RPP * ROWB, WSTAGE = BCP * ROWB;
  static_assert(TH == 8, "tile rows");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);  // [2 pixel stages][3 weight stages]
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wc = wave / WP;
  const int wp = wave % WP;
  const int z = (int)blockIdx.z;
  int L;
  {
    const int nwg = (int)gridDim.x, id = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ct = L % g.nct;
  const int pt = L / g.nct;
  const int txi = pt % g.tiles_x;
  const int tyi = (pt / g.tiles_x) % g.tiles_y;
  const int n = pt / (g.tiles_x * g.tiles_y);
  const int ty0 = tyi * TH, tx0 = txi * kHaloTW;
  const int c_base = ct * BC;

  const int pc = tid & 3;
  const int row0 = tid >> 2;
  const int pcs = pc ^ ((row0 >> 1) & 3);        // LDS slot (r, pc) holds source piece pc ^ swz(r); RPP % 16 == 0
  const T* wbase = reinterpret_cast<const T*>(p.weight) + (int64_t)z * p.w_zoff;
  const T* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + row0 + i * RPP;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 8;
  }
  int xpix[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int hr = row0 + i * RPP;
    const int hy = hr / g.hw, hx = hr - hy * g.hw;
    const int iy = ty0 - p.ph + hy, ix = tx0 - p.pw + hx;
    const bool ok = hr < g.hrows && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xpix[i] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }

  const int ntaps = p.kh * p.kw;
  int w_tap = 0, w_rem = 0, w_seg = 0, w_sbase = 0, w_chunks = p.seg_chunks[0];
  auto fetch_w = [&](int wbuf) PP_INLINE_LAMBDA {
    unsigned char* wt = smem + 2 * XSTAGE + wbuf * WSTAGE;
    const int woff = w_tap * p.chunks_per_tap * 32 + w_sbase + w_rem * 32;
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      glds16(wrow[i] + woff, wt + (i * NT + wave * 64) * 16);
    if (++w_tap == ntaps) {
      w_tap = 0;
      if (++w_rem == w_chunks) {
        w_rem = 0;
        w_sbase += w_chunks * 32;
        ++w_seg;
#pragma unroll
        for (int s = 1; s < PP_MAX_SEG; ++s)
          if (w_seg == s) w_chunks = p.seg_chunks[s];
      }
    }
  };
  int x_rem = 0, x_seg = 0;
  const T* x_base = reinterpret_cast<const T*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int x_C = p.in_C[0], x_ldc = p.in_ldc[0], x_chunks = p.seg_chunks[0];
  auto fetch_x = [&](int xbuf) PP_INLINE_LAMBDA {
    unsigned char* xt = smem + xbuf * XSTAGE;
    const int c0 = x_rem * 32 + pcs * 8;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const bool ok = xpix[i] >= 0 && c0 < x_C;
      const void* src = ok ? static_cast<const void*>(x_base + (int64_t)xpix[i] * x_ldc + c0) : static_cast<const void*>(pp_zero16);
      glds16(src, xt + (i * NT + wave * 64) * 16);
    }
    if (++x_rem == x_chunks) {
      x_rem = 0;
      ++x_seg;
#pragma unroll
      for (int s = 1; s < PP_MAX_SEG; ++s) {
        if (x_seg == s && s < p.nseg) {
          x_base = reinterpret_cast<const T*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
          x_C = p.in_C[s];
          x_ldc = p.in_ldc[s];
          x_chunks = p.seg_chunks[s];
        }
      }
    }
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int wroff = (fgrp ^ ((frow >> 1) & 3)) << 4;
  const int xrow_base = wp * TP * g.hw + frow;

  auto compute = [&](int tapoff, int xbuf, int wbuf) PP_INLINE_LAMBDA {
    const unsigned char* ws = smem + 2 * XSTAGE + wbuf * WSTAGE + (wc * TC * 16 + frow) * ROWB;
    const unsigned char* xs = smem + xbuf * XSTAGE;
    h8 af[TC], bf[TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) af[a] = lds_frag(ws + a * 16 * ROWB + wroff);
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int r = xrow_base + b * g.hw + tapoff;
      bf[b] = lds_frag(xs + r * ROWB + ((fgrp ^ ((r >> 1) & 3)) << 4));
    }
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af[a], bf[b], acc[a][b]);
  };

  const int total = p.nchunks, nck = p.chunks_per_tap;
  fetch_x(0);
  fetch_w(0);
  fetch_w(1);
  pp_wait_vmcnt<WPASS>();   // pixels of chunk 0 and weights of step 0 landed (step 1 may still be in flight)
  pp_barrier();
  int w0 = 0, tap = 0, ky = 0, kx = 0, chunk = 0;
  for (int q = 0; q < total; ++q) {
    const int w1 = w0 == 2 ? 0 : w0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;
    const bool more_w = q + 2 < total;
    const bool next_chunk = chunk + 1 < nck;
    const bool pre_last = tap == ntaps - 2;
    if (more_w) fetch_w(w2);
    if (pre_last && next_chunk) fetch_x((chunk + 1) & 1);  // the other pixel stage: last read one chunk ago
    compute(ky * p.dh * g.hw + kx * p.dw, chunk & 1, w0);
    if (pre_last && next_chunk) {
      // queue: [weights q+1] [weights q+2] [pixels]: weights q+1 must have landed
      if (more_w) pp_wait_vmcnt<WPASS + XPASS>(); else pp_wait_vmcnt<XPASS>();
    } else {
      // (last tap: queue [weights q+1] [pixels] [weights q+2] -- retiring up to the pixels)
      if (more_w) pp_wait_vmcnt<WPASS>(); else pp_wait_vmcnt<0>();
    }
    pp_barrier();
    w0 = w1;
    if (++kx == p.kw) {
      kx = 0;
      ++ky;
    }
    if (++tap == ntaps) {
      tap = 0;
      ky = 0;
      ++chunk;
    }
  }

  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  const int ox = tx0 + frow;
  constexpr bool EPI_FITS = WC * WP * epi_lds_wave_bytes<TC>() <= 2 * XSTAGE + 3 * WSTAGE;
  epilogue_any<OT, TC, TP, EPI_FITS>(
      p, e, smem, wave, lane, c_base + wc * TC * 16,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        const int oy = ty0 + wp * TP + decltype(bi)::value;
        m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
        ok = oy < p.Ho && ox < p.Wo;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; },
      [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
        const int oy = ty0 + wp * TP + decltype(bi)::value;
        m0 = ((int64_t)n * p.Ho + oy) * p.Wo + tx0;
        nvalid = oy < p.Ho ? p.Wo - tx0 : 0;
      });
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 / dilation 1 form with compile-time taps (the encoder / decoder / propagation convolutions).  In the kernel above
// a step is ONE tap: 16 MFMAs per wave between two barriers, with the swizzled fragment address of every pixel fragment
// recomputed per tap (~5 vector-ALU operations each) -- for a single-product f16 tile that bookkeeping weighs three
// times what it does in the PP_F32X2 kernels.  Here a step is one ROW of taps (ky, kx = 0..2):
//   - the weights of the three taps are copied together into one of TWO 3-tap stages (one step ahead), so a chunk has
//     3 barriers instead of 9 and 48 MFMAs between them; inside a step hipcc overlaps the fragment reads of tap t+1
//     with the MFMAs of tap t (nothing synchronises between them);
//   - the 9 x TP swizzled pixel-fragment offsets are computed once per kernel and kept in registers (an f16 tile has
//     64 accumulator registers, there is room), the weight-fragment offsets are immediates: no address arithmetic in
//     the loop;
//   - pixel tiles are double-buffered per chunk as above and copied a whole chunk ahead (at the chunk's first step).
// LDS: 2 x 12 KB pixels + 2 x 24 KB weights = 72 KB: two work-groups per CU.
template <typename OT, int WC, int WP, int TC, int TP>
__global__ void __launch_bounds__(WC * WP * 64, 2) conv_halo_f16_ct_kernel(const ConvK p, const HaloGeom g) {
  typedef half_t T;
  constexpr int NT = WC * WP * 64;
  constexpr int TH = WP * TP;
  constexpr int ROWB = 64;
  constexpr int RPP = NT / 4;
  constexpr int BC = WC * TC * 16;
  constexpr int WPASS = (BC + RPP - 1) / RPP;
  constexpr int BCP = WPASS * RPP;
  constexpr int HW = kHaloTW + 2, HROWS = (TH + 2) * HW;
  constexpr int XPASS = (HROWS + RPP - 1) / RPP;
  constexpr int XSTAGE = XPASS * RPP * ROWB, WTAP = BCP * ROWB, WSTAGE = 3 * WTAP;
  static_assert(TH == 8 && NT == 256, "tile");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);  // [2 pixel stages][2 weight stages of 3 taps]
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
#ifdef PP_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wc = wave / WP;
  const int wp = wave % WP;
  const int z = (int)blockIdx.z;
  int L;
  {
    const int nwg = (int)gridDim.x, id = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ct = L % g.nct;
  const int pt = L / g.nct;
  const int txi = pt % g.tiles_x;
  const int tyi = (pt / g.tiles_x) % g.tiles_y;
  const int n = pt / (g.tiles_x * g.tiles_y);
  const int ty0 = tyi * TH, tx0 = txi * kHaloTW;
  const int c_base = ct * BC;

  const int pc = tid & 3;
  const int row0 = tid >> 2;
  const int pcs = pc ^ ((row0 >> 1) & 3);        // LDS slot (r, pc) holds source piece pc ^ swz(r); RPP % 16 == 0
  const T* wbase = reinterpret_cast<const T*>(p.weight) + (int64_t)z * p.w_zoff;
  const T* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + row0 + i * RPP;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 8;
  }
  int xpix[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int hr = row0 + i * RPP;
    const int hy = hr / HW, hx = hr - hy * HW;
    const int iy = ty0 - p.ph + hy, ix = tx0 - p.pw + hx;
    const bool ok = hr < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xpix[i] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }

  // K iterators (weights: chunk of the step being copied; pixels: chunk being copied)
  int w_rem = 0, w_seg = 0, w_sbase = 0, w_chunks = p.seg_chunks[0];
  const int tapstride = p.chunks_per_tap * 32;
  auto fetch_w = [&](int stage, int ky) PP_INLINE_LAMBDA {  // the three taps of kernel row ky of the iterator's chunk
    unsigned char* wt = smem + 2 * XSTAGE + stage * WSTAGE;
    const int woff = w_sbase + w_rem * 32;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < WPASS; ++i)
        glds16(wrow[i] + ((ky * 3 + t) * tapstride + woff), wt + t * WTAP + (i * NT + wave * 64) * 16);
  };
  auto w_next_chunk = [&]() PP_INLINE_LAMBDA {
    if (++w_rem == w_chunks) {
      w_rem = 0;
      w_sbase += w_chunks * 32;
      ++w_seg;
#pragma unroll
      for (int s = 1; s < PP_MAX_SEG; ++s)
        if (w_seg == s) w_chunks = p.seg_chunks[s];
    }
  };
  int x_rem = 0, x_seg = 0;
  const T* x_base = reinterpret_cast<const T*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int x_C = p.in_C[0], x_ldc = p.in_ldc[0], x_chunks = p.seg_chunks[0];
  auto fetch_x = [&](int xbuf) PP_INLINE_LAMBDA {
    unsigned char* xt = smem + xbuf * XSTAGE;
    const int c0 = x_rem * 32 + pcs * 8;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const bool ok = xpix[i] >= 0 && c0 < x_C;
      const void* src = ok ? static_cast<const void*>(x_base + (int64_t)xpix[i] * x_ldc + c0) : static_cast<const void*>(pp_zero16);
      glds16(src, xt + (i * NT + wave * 64) * 16);
    }
    if (++x_rem == x_chunks) {
      x_rem = 0;
      ++x_seg;
#pragma unroll
      for (int s = 1; s < PP_MAX_SEG; ++s) {
        if (x_seg == s && s < p.nseg) {
          x_base = reinterpret_cast<const T*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
          x_C = p.in_C[s];
          x_ldc = p.in_ldc[s];
          x_chunks = p.seg_chunks[s];
        }
      }
    }
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  // fragment offsets: weights = one per-lane base + immediates; pixels = 9 x TP swizzled offsets, computed once
  const unsigned char* wfrag = smem + 2 * XSTAGE + (wc * TC * 16 + frow) * ROWB + ((fgrp ^ ((frow >> 1) & 3)) << 4);
  int xoff[9][TP];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int r = (wp * TP + b + tap / 3) * HW + frow + tap % 3;
      xoff[tap][b] = r * ROWB + ((fgrp ^ ((r >> 1) & 3)) << 4);
    }

  auto compute_row = [&](auto kyc, int xbuf, int wstage) PP_INLINE_LAMBDA {
    constexpr int ky = decltype(kyc)::value;
    const unsigned char* ws = wfrag + wstage * WSTAGE;
    const unsigned char* xs = smem + xbuf * XSTAGE;
    static_for<3>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      h8 af[TC], bf[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) af[a] = lds_frag(ws + t * WTAP + a * 16 * ROWB);
#pragma unroll
      for (int b = 0; b < TP; ++b) bf[b] = lds_frag(xs + xoff[ky * 3 + t][b]);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af[a], bf[b], acc[a][b]);
    });
  };

  // ---- pipeline: step s = chunk * 3 + ky; weights of step s in stage s & 1 (copied one step ahead), pixels of chunk c
  // in stage c & 1 (copied at the first step of chunk c - 1)
  const int nck = p.chunks_per_tap;
  fetch_x(0);
  fetch_w(0, 0);
  pp_wait_vmcnt<0>();
  pp_barrier();
  int ws = 0;
  for (int chunk = 0; chunk < nck; ++chunk) {
    const bool next_chunk = chunk + 1 < nck;
    static_for<3>([&](auto kyc) {
      constexpr int ky = decltype(kyc)::value;
      // (the barrier that ended the previous step: every wave is done with the stages written below)
      if constexpr (ky < 2) {
        fetch_w(ws ^ 1, ky + 1);
      } else {
        if (next_chunk) {
          w_next_chunk();
          fetch_w(ws ^ 1, 0);
        }
      }
      if constexpr (ky == 0) {
        if (next_chunk) fetch_x((chunk + 1) & 1);
      }
      compute_row(kyc, chunk & 1, ws);
      pp_wait_vmcnt<0>();                        // this wave's copies for the next step (and chunk) have landed
      pp_barrier();
      ws ^= 1;
    });
  }

  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  const int ox = tx0 + frow;
  constexpr bool EPI_FITS = WC * WP * epi_lds_wave_bytes<TC>() <= 2 * XSTAGE + 2 * WSTAGE;
  epilogue_any<OT, TC, TP, EPI_FITS>(
      p, e, smem, wave, lane, c_base + wc * TC * 16,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        const int oy = ty0 + wp * TP + decltype(bi)::value;
        m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
        ok = oy < p.Ho && ox < p.Wo;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; },
      [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
        const int oy = ty0 + wp * TP + decltype(bi)::value;
        m0 = ((int64_t)n * p.Ho + oy) * p.Wo + tx0;
        nvalid = oy < p.Ho ? p.Wo - tx0 : 0;
      });
}

template <typename OT, int WC, int WP, int TC, int TP>
static int launch_halo_f16_ct_cfg(void* stream, const ConvK& k, int Z, HaloGeom g) {
  constexpr int BC = WC * TC * 16;
  constexpr int NT = WC * WP * 64;
  constexpr int RPP = NT / 4;
  constexpr int BCP = (BC + RPP - 1) / RPP * RPP;
  constexpr int XPASS = (10 * 18 + RPP - 1) / RPP;
  const size_t smem = (size_t)(2 * XPASS * RPP + 2 * 3 * BCP) * 64;
  g.nct = (k.Cout + BC - 1) / BC;
  dim3 grid((unsigned)(g.ntiles * g.nct), 1u, (unsigned)Z);
  PP_ALLOW_BIG_LDS((&conv_halo_f16_ct_kernel<OT, WC, WP, TC, TP>), smem);
  PP_LAUNCH((conv_halo_f16_ct_kernel<OT, WC, WP, TC, TP>), grid, dim3(NT), smem, stream, k, g);
  return pp_check_launch("pp_conv2d");
}

template <typename OT, int WC, int WP, int TC, int TP, int XPASS>
static int launch_halo_f16_cfg(void* stream, const ConvK& k, int Z, HaloGeom g) {
  constexpr int BC = WC * TC * 16;
  constexpr int NT = WC * WP * 64;
  constexpr int RPP = NT / 4;
  constexpr int BCP = (BC + RPP - 1) / RPP * RPP;
  const size_t smem = (size_t)(2 * XPASS * RPP + 3 * BCP) * 64;
  g.nct = (k.Cout + BC - 1) / BC;
  dim3 grid((unsigned)(g.ntiles * g.nct), 1u, (unsigned)Z);
  PP_ALLOW_BIG_LDS((&conv_halo_f16_kernel<OT, WC, WP, TC, TP, XPASS>), smem);
  PP_LAUNCH((conv_halo_f16_kernel<OT, WC, WP, TC, TP, XPASS>), grid, dim3(NT), smem, stream, k, g);
  return pp_check_launch("pp_conv2d");
}

template <typename OT>
static int launch_halo_f16_t(void* stream, const ConvK& k, int Z) {
  HaloGeom g;
  if (!halo_geometry(k, Z, 320, &g, 8, options().halo_min_cout)) return 1;
  const bool big = g.hrows > 192;  // dilated 3x3 (12 x 20, 14 x 22): five 64-row copy passes instead of three
  if (options().halo_ct && k.kh == 3 && k.kw == 3 && k.dh == 1 && k.dw == 1) {
    if (k.Cout > 64 && !options().halo_c64) {
      const int waste128 = (k.Cout + 127) / 128 * 128 - k.Cout;
      const int waste96 = (k.Cout + 95) / 96 * 96 - k.Cout;
      if (waste96 + 32 <= waste128) return launch_halo_f16_ct_cfg<OT, 2, 2, 3, 4>(stream, k, Z, g);
      return launch_halo_f16_ct_cfg<OT, 2, 2, 4, 4>(stream, k, Z, g);
    }
    return launch_halo_f16_ct_cfg<OT, 1, 4, 4, 2>(stream, k, Z, g);
  }
  if (k.Cout > 64) {
    const int waste128 = (k.Cout + 127) / 128 * 128 - k.Cout;
    const int waste96 = (k.Cout + 95) / 96 * 96 - k.Cout;
    if (waste96 + 32 <= waste128)
      return big ? launch_halo_f16_cfg<OT, 2, 2, 3, 4, 5>(stream, k, Z, g) : launch_halo_f16_cfg<OT, 2, 2, 3, 4, 3>(stream, k, Z, g);
    return big ? launch_halo_f16_cfg<OT, 2, 2, 4, 4, 5>(stream, k, Z, g) : launch_halo_f16_cfg<OT, 2, 2, 4, 4, 3>(stream, k, Z, g);
  }
  return big ? launch_halo_f16_cfg<OT, 1, 4, 4, 2, 5>(stream, k, Z, g) : launch_halo_f16_cfg<OT, 1, 4, 4, 2, 3>(stream, k, Z, g);
}

// (r05: the two storage types are compiled in their own translation units -- conv_halo_f16_h.hip / conv_halo_f16_f.hip -- because
//  this header was the build's long pole: 7 minutes in one unit with the epilogue variants; conv_halo_f16.hip keeps the dispatch)
template <typename OT>
static int launch_halo_f16_small_t(void* stream, const ConvK& k, int Z, const HaloGeom& g) {
  return launch_halo_f16_ct_cfg<OT, 1, 4, 1, 2>(stream, k, Z, g);
}

}  // namespace pp

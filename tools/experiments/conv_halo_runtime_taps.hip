// ARCHIVED (round 3): the runtime-tap form of the PP_F32X2 halo-tile kernel.  No shipped layer selected it once the
// compile-time-tap kernels (3x3, 1x5, 5x1, dilation 1) existed; geometries outside those three now take conv_split_kernel
// (conv_split.hip).  Kept for the record: to build it again paste the kernel and launch_halo_cfg back into conv_halo.hip before
// the compile-time-tap kernel and return launch_halo_cfg<WC, WP, TC, TP>(stream, k, Z, g) at the end of launch_halo_any.
// Measurements: profiles/r02_conv_layers.log (runtime taps 277-320 TF/s where the compile-time-tap form reaches 293-365).
#include "conv_halo_common.h"

namespace pp {

template <int WC, int WP, int TC, int TP>
__global__ void __launch_bounds__(WC * WP * 64, 2) conv_halo_split_kernel(const ConvK p, const HaloGeom g) {
  typedef float OT;
  constexpr int NT = WC * WP * 64;
  constexpr int TH = WP * TP;                    // one B fragment (16 pixels) per tile row
  constexpr int XROWS = NT / 4, WROWS = NT / 8;  // LDS rows covered by one pass of the work-group
  constexpr int BC = WC * TC * 16;
  constexpr int BCP = (BC + WROWS - 1) / WROWS * WROWS;
  constexpr int ROWB = 128;
  constexpr int XPASS = (kHaloMaxRows + XROWS - 1) / XROWS;
  constexpr int WPASS = BCP / WROWS;
  constexpr int XBYTES = XPASS * XROWS * ROWB, WSTAGE = BCP * ROWB;
  constexpr int NX = 2 * XPASS;                  // hidden pixel loads per thread and chunk
  constexpr float LINV = 1.f / 2048.f;
  static_assert(TH == 8, "tile rows");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wc = wave / WP;
  const int wp = wave % WP;
  const int z = (int)blockIdx.z;

  // ---- work-group -> (pixel tile, channel tile).  Consecutive work-group ids are dealt round-robin to the 8 XCDs
  // (each with its own L2): remap so that every XCD owns a contiguous range of the linear tile space, with the
  // channel tiles of one pixel tile adjacent (the second channel tile finds the pixels in L2).
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

  // Slot swizzle of LDS row r (a function of r mod 8): swz(r) = ((b0 ^ b2) << 2) | (b1 << 1), b_i = bit i of r.
  // Brute-forced over all GF(2)-linear maps of r mod 16 against the gfx950 service groups (MI355X_MICROARCH.md, LDS):
  //  - ds_read_b128 fragment reads (16 CONSECUTIVE rows starting at ANY row x 4 k-groups per plane, four
  //    non-contiguous 16-lane groups {0-3,12-15,20-27}, ...): every group hits 16 distinct 16-byte slots of the
  //    256-byte bank row for every start row (conv_split_kernel's swizzle is 2-way here unless the start row is a
  //    multiple of 4: measured 20-26 % of the LDS cycles as SQ_LDS_BANK_CONFLICT);
  //  - ds_write_b128 pixel stores (contiguous 8-lane groups = 2 rows x 4 octets of one plane, 128-byte bank row):
  //    8 distinct slots.
  auto swz = [](int r) PP_INLINE_LAMBDA { return (((r ^ (r >> 2)) & 1) << 2) | (r & 2); };

  // weights: lane-linear DMA image (as conv_split_kernel)
  const int pc = tid & 7;
  const int wrow0 = tid >> 3;
  const int pcs = pc ^ swz(wrow0);
  const float* wbase = reinterpret_cast<const float*>(p.weight) + (int64_t)z * p.w_zoff;
  const float* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + wrow0 + i * WROWS;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 4;
  }

  // pixels: thread = (halo row xrow0 + XROWS*i, channel octet xj); the row's input pixel is fixed for the whole kernel
  const int xj = tid & 3;
  const int xrow0 = tid >> 2;
  const int xoff_h = (xj ^ swz(xrow0)) << 4;
  const int xoff_l = ((xj + 4) ^ swz(xrow0)) << 4;
  int xpix[XPASS];  // input pixel index (n, y, x) of the halo row, -1 = outside the image / past the halo tile
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int hr = xrow0 + i * XROWS;
    const int hy = hr / g.hw, hx = hr - hy * g.hw;
    const int iy = ty0 - p.ph + hy, ix = tx0 - p.pw + hx;
    const bool ok = hr < g.hrows && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xpix[i] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }

  f4 xreg[XPASS][2];  // hidden loads: valid only after the counted wait in store_x
  int xok = 0;

  // ---- K iterators.  Weights run two (chunk, tap) steps ahead of the multiply, pixels one chunk ahead.
  const int ntaps = p.kh * p.kw;
  int w_tap = 0, w_rem = 0, w_seg = 0, w_sbase = 0, w_chunks = p.seg_chunks[0];
  auto w_advance = [&]() PP_INLINE_LAMBDA {
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
  auto fetch_w = [&](int wbuf) PP_INLINE_LAMBDA {
    unsigned char* wt = smem + XBYTES + wbuf * WSTAGE;
    const int woff = w_tap * p.chunks_per_tap * 32 + w_sbase + w_rem * 32;
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      glds16(wrow[i] + woff, wt + (i * NT + wave * 64) * 16);
    w_advance();
  };

  int x_rem = 0, x_seg = 0;
  const float* x_base = reinterpret_cast<const float*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int x_C = p.in_C[0], x_ldc = p.in_ldc[0], x_chunks = p.seg_chunks[0];
  // pixels of the iterator's chunk -> registers (unconditional loads; rows outside the image and padded channels
  // read a safe address and are zeroed by a select in store_x)
  auto fetch_x = [&]() PP_INLINE_LAMBDA {
    const int c0 = x_rem * 32 + xj * 8;
    int okbits = 0;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const bool ok0 = xpix[i] >= 0 && c0 < x_C, ok1 = xpix[i] >= 0 && c0 + 4 < x_C;
      const float* src = ok0 ? x_base + (int64_t)xpix[i] * x_ldc + c0 : x_base;
      gload16_hidden(xreg[i][0], src);
      gload16_hidden(xreg[i][1], src + (ok1 ? 4 : 0));
      okbits |= ((ok0 ? 1 : 0) | (ok1 ? 2 : 0)) << (2 * i);
    }
    xok = okbits;
    if (++x_rem == x_chunks) {
      x_rem = 0;
      ++x_seg;
#pragma unroll
      for (int s = 1; s < PP_MAX_SEG; ++s) {
        if (x_seg == s && s < p.nseg) {
          x_base = reinterpret_cast<const float*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
          x_C = p.in_C[s];
          x_ldc = p.in_ldc[s];
          x_chunks = p.seg_chunks[s];
        }
      }
    }
  };
  // split (h: round toward zero, saturating; l: exact remainder * 2048) and store the halo tile; the caller has
  // waited for the hidden loads
  auto store_x = [&]() PP_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      f4 v[2] = {xreg[i][0], xreg[i][1]};
      if (!((xok >> (2 * i)) & 1)) v[0] = f4{0.f, 0.f, 0.f, 0.f};
      if (!((xok >> (2 * i)) & 2)) v[1] = f4{0.f, 0.f, 0.f, 0.f};
      h8 h, l;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const float c0 = v[e >> 2][e & 3], c1 = v[e >> 2][(e & 3) + 1];
        h2 hh, ll;
        split_pair(c0, c1, hh, ll);
        h[e] = hh[0];
        h[e + 1] = hh[1];
        l[e] = ll[0];
        l[e + 1] = ll[1];
      }
      unsigned char* rowp = smem + (xrow0 + i * XROWS) * ROWB;
      *reinterpret_cast<h8*>(rowp + xoff_h) = h;
      *reinterpret_cast<h8*>(rowp + xoff_l) = l;
    }
  };

  f4 acc[TC][TP], accx[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
      accx[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    }

  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int wroff_h = (fgrp ^ swz(frow)) << 4;
  const int wroff_l = ((fgrp + 4) ^ swz(frow)) << 4;
  const int xrow_base = wp * TP * g.hw + frow;  // LDS row of this lane's pixel in fragment b = 0 at tap (0,0)

  auto compute = [&](int tapoff, int wbuf) PP_INLINE_LAMBDA {
    const unsigned char* ws = smem + XBYTES + wbuf * WSTAGE + (wc * TC * 16 + frow) * ROWB;
    h8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      ah[a] = lds_frag(ws + a * 16 * ROWB + wroff_h);
      al[a] = lds_frag(ws + a * 16 * ROWB + wroff_l);
    }
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int r = xrow_base + b * g.hw + tapoff;
      const int s = swz(r);
      const unsigned char* xr = smem + r * ROWB;
      bh[b] = lds_frag(xr + ((fgrp ^ s) << 4));
      bl[b] = lds_frag(xr + (((fgrp + 4) ^ s) << 4));
    }
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(ah[a], bh[b], acc[a][b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) accx[a][b] = mfma_16x16x32_f16(ah[a], bl[b], accx[a][b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) accx[a][b] = mfma_16x16x32_f16(al[a], bh[b], accx[a][b]);
  };

  // ---- pipeline.  step q = (chunk, tap); weights of step q live in ring stage q % 3.
  const int total = p.nchunks;               // chunks_per_tap * ntaps, >= 2
  const int nck = p.chunks_per_tap;          // channel chunks
  fetch_x();                                 // chunk 0
  fetch_w(0);
  fetch_w(1);
  wait_vmcnt_hidden<2 * WPASS>();            // the pixel loads were issued first: retired when only the copies remain
  store_x();
  wait_vmcnt_hidden<WPASS>();                // weights of step 0 landed (step 1 may still be in flight)
  pp_wait_lgkm0();
  pp_barrier();

  int w0 = 0;                                // q % 3
  int tap = 0, ky = 0, kx = 0, chunk = 0;
  for (int q = 0; q < total; ++q) {
    const int w1 = w0 == 2 ? 0 : w0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;
    const bool more_w = q + 2 < total;
    const bool next_chunk = chunk + 1 < nck;
    const bool pre_last = tap == ntaps - 2, last = tap == ntaps - 1;
    if (more_w) fetch_w(w2);
    if (pre_last && next_chunk) fetch_x();   // issued AFTER this step's weight copies
    compute(ky * p.dh * g.hw + kx * p.dw, w0);
    if (last && next_chunk) {
      pp_wait_lgkm0();
      pp_barrier();                          // every wave has read its last fragments of the current halo tile
      // queue: [weights q+1] [pixels] [weights q+2 (this step)]: retire up to the pixels
      if (more_w) wait_vmcnt_hidden<WPASS>(); else wait_vmcnt_hidden<0>();
      store_x();
    } else if (pre_last && next_chunk) {
      // queue: [weights q+1] [weights q+2] [pixels]: weights q+1 must have landed, the rest stays in flight
      if (more_w) wait_vmcnt_hidden<WPASS + NX>(); else wait_vmcnt_hidden<NX>();
    } else {
      if (more_w) wait_vmcnt_hidden<WPASS>(); else wait_vmcnt_hidden<0>();
    }
    pp_wait_lgkm0();
    pp_barrier();                            // bare barrier: the copies of step q+2 stay in flight across it
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
  epilogue_quads<OT, TC, TP>(
      p, e,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        const int oy = ty0 + wp * TP + decltype(bi)::value;
        m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
        ok = oy < p.Ho && ox < p.Wo;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA {
        return acc[decltype(ai)::value][decltype(bi)::value] + accx[decltype(ai)::value][decltype(bi)::value] * LINV;
      });
}


template <int WC, int WP, int TC, int TP>
static int launch_halo_cfg(void* stream, const ConvK& k, int Z, HaloGeom g) {
  constexpr int BC = WC * TC * 16;
  constexpr int NT = WC * WP * 64;
  constexpr int BCP = (BC + NT / 8 - 1) / (NT / 8) * (NT / 8);
  constexpr int XPASS = (kHaloMaxRows + NT / 4 - 1) / (NT / 4);
  const size_t smem = (size_t)(XPASS * (NT / 4) + 3 * BCP) * 128;
  g.nct = (k.Cout + BC - 1) / BC;
  dim3 grid((unsigned)(g.ntiles * g.nct), 1u, (unsigned)Z);
  static const bool lds_ok = (pp_allow_big_lds(reinterpret_cast<const void*>(&conv_halo_split_kernel<WC, WP, TC, TP>), smem), true);
  (void)lds_ok;
  PP_LAUNCH((conv_halo_split_kernel<WC, WP, TC, TP>), grid, dim3(NT), smem, stream, k, g);
  return pp_check_launch("pp_conv2d");
}


}  // namespace pp

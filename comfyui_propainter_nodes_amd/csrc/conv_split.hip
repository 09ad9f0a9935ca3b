// conv_split.hip -- pp_conv2d in PP_F32X2 mode: f32 convolution on the f16 matrix pipe (see conv_igemm.hip for the
// implicit-GEMM formulation and include/propainter_mi355.h for the weight layout).
#include "conv_common.h"

namespace pp {

// ---------------------------------------------------------------------------------------------------------
// f32 convolution on the f16 matrix pipe ("split" mode, dtype PP_F32X2).
//
// Every f32 operand value v is represented by two f16 terms  v ~= h + l,  h = f16_rtz(v), l = f16_rtz(v - h)
// (>= 20 significand bits worst case, 22 typical; r05: the low term is UNSCALED -- the matrix pipe honours f16 subnormals,
// tools/probes/mfma_denorm.hip; r01-r04 carried l * 2048 and a second accumulator set for the cross terms).  Then
//     sum_k w x  ~=  sum_k wh xh  +  sum_k wh xl  +  sum_k wl xh        (the wl xl term is < 2^-20 of the sum)
// i.e. three v_mfma_f32_16x16x32_f16 (16x the f32 MFMA rate each) into ONE fp32 accumulator set instead of sixteen
// v_mfma_f32_32x32x2_f32 steps.  Constant weights carry a power-of-two scale per layer (ops.split_pack_weight) so that
// their low terms are normal f16 numbers; the epilogue multiplies the accumulators by its inverse (ConvK::acc_scale).
//
// LDS tile row = one 32-channel chunk = 128 bytes = 8 16-byte slots: slots 0-3 the h terms (k 0-31), 4-7 the
// l terms, slot s stored at s ^ swz(row).  Weights are split on the host (same byte size and chunk
// order as the f32 packing) and copied by global_load_lds; pixels are loaded as f32 (8 channels per thread), split
// in registers and written with one ds_write_b128 per plane.  One barrier per chunk, two chunks in flight.
template <typename OT, int WC, int WP, int TC, int TP>
__global__ void __launch_bounds__(WC * WP * 64, 2) conv_split_kernel(const ConvK p) {
  constexpr int NT = WC * WP * 64;          // threads per work-group (4 waves; 8 for the 256-channel tile)
  constexpr int XROWS = NT / 4, WROWS = NT / 8;  // tile rows covered by one pass of the work-group
  constexpr int BC = WC * TC * 16;
  constexpr int BP = WP * TP * 16;
  constexpr int BCP = (BC + WROWS - 1) / WROWS * WROWS;  // weight rows staged (rows past BC are never read)
  constexpr int ROWB = 128;                 // bytes per tile row
  constexpr int XPASS = (BP + XROWS - 1) / XROWS;  // pixel passes: 4 threads per row (8 channels each)
  constexpr int WPASS = BCP / WROWS;        // weight passes: 8 threads per row (16 bytes each)
  // LDS: 2 pixel stages + 3 weight stages.  Pixels of chunk q+2 are in flight to registers and weights of chunk
  // q+2 in flight to LDS while chunk q is multiplied: two chunks of latency cover per work-group.
  constexpr int XSTAGE = BP * ROWB, WSTAGE = BCP * ROWB;
  // vector-memory instructions per thread per chunk
  constexpr int NLOADS = WPASS + 2 * XPASS;

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);

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

  // slot swizzle of tile row r (depends on r mod 16 only): swz(r) = ((r >> 1) & 7) ^ ((r & 1) << 2).
  //  - fragment reads (16 rows x 4 k-groups per plane): every 16-lane service group of ds_read_b128 hits 16
  //    distinct 16-byte slots of the 256-byte bank row;
  //  - pixel writes (ds_write_b128, 8-lane groups = 2 rows x 4 slots of one plane): the odd row's plane lives in
  //    the other half of the 128-byte row, so the 8 lanes cover 8 distinct slots.
  auto swz = [](int r) PP_INLINE_LAMBDA { return ((r >> 1) & 7) ^ ((r & 1) << 2); };

  // weights: lane-linear DMA image, LDS slot pc of row wrow0 holds source piece pc ^ swz
  const int pc = tid & 7;
  const int wrow0 = tid >> 3;
  const int pcs = pc ^ swz(wrow0);
  // pixels: thread = (row xrow0 + 64 i, channel octet xj): h octet -> slot xj ^ swz, l octet -> slot (xj + 4) ^ swz
  const int xj = tid & 3;
  const int xrow0 = tid >> 2;
  const int xoff_h = (xj ^ swz(xrow0)) << 4;
  const int xoff_l = ((xj + 4) ^ swz(xrow0)) << 4;

  // per pass: pixel index of tap (0,0) and its (y, x).  (They used to share one register as two 16-bit halves: a
  // temporal convolution of flow completion sees a clip as a [T] x [2 h w] image whose width passes 32767 from
  // 360x640 frames on, and x wrapped negative -- every tap of those pixels was treated as padding.)
  int64_t prow[XPASS];
  int py0[XPASS], px0[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int r = xrow0 + i * XROWS;
    const int64_t m = p_base + r;
    const int64_t mm = (r < BP && m < p.M) ? m : p.M - 1;  // rows past M: clamped, results never stored
    const int wo = (int)(mm % p.Wo);
    const int64_t t = mm / p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const int y0 = ho * p.sh - p.ph, x0 = wo * p.sw - p.pw;
    prow[i] = (int64_t)n * p.H * p.W + (int64_t)y0 * p.W + x0;
    py0[i] = y0;
    px0[i] = x0;
  }
  const float* wbase = reinterpret_cast<const float*>(p.weight) + (int64_t)z * p.w_zoff;
  const float* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + wrow0 + i * WROWS;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 4;
  }

  f4 xreg[2][XPASS][2];  // two chunks in flight (hidden loads: valid only after the counted wait in store_x)
  int xok[2] = {0, 0};   // validity bits of the half octets (2 per pass) of each register set

  // K iterator: tap innermost (see conv_igemm_kernel)
  int it_ky = 0, it_kx = 0, it_seg = 0, it_rem = 0, it_sbase = 0;
  auto it_woff = [&]() PP_INLINE_LAMBDA { return (it_ky * p.kw + it_kx) * p.chunks_per_tap * 32 + it_sbase + it_rem * 32; };
  const float* it_base = reinterpret_cast<const float*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int it_C = p.in_C[0], it_ldc = p.in_ldc[0], it_chunks = p.seg_chunks[0];
  auto select_segment = [&](int seg) PP_INLINE_LAMBDA {
#pragma unroll
    for (int s = 0; s < PP_MAX_SEG; ++s) {
      if (seg == s) {
        it_base = reinterpret_cast<const float*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
        it_C = p.in_C[s];
        it_ldc = p.in_ldc[s];
        it_chunks = p.seg_chunks[s];
      }
    }
  };
  auto advance = [&]() PP_INLINE_LAMBDA {
    if (++it_kx == p.kw) {
      it_kx = 0;
      if (++it_ky == p.kh) {
        it_ky = 0;
        if (++it_rem == it_chunks) {
          it_rem = 0;
          it_sbase += it_chunks * 32;
          if (p.nseg > 1) select_segment(++it_seg);
        }
      }
    }
  };

  // next chunk of the K iterator: weights -> LDS stage `buf` (DMA), pixels -> registers (unconditional loads;
  // out-of-image taps and padded channels read a safe address and are zeroed by a select)
  auto fetch = [&](int wbuf, auto par) PP_INLINE_LAMBDA {
    constexpr int P = decltype(par)::value;
    unsigned char* wt = smem + 2 * XSTAGE + wbuf * WSTAGE;
    const int woff = it_woff();
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      glds16(wrow[i] + woff, wt + (i * NT + wave * 64) * 16);
    const int c0 = it_rem * 32 + xj * 8;
    const int dy = it_ky * p.dh, dx = it_kx * p.dw;
    const int64_t tapoff = (int64_t)dy * p.W + dx;
    const float* cbase = it_base + c0;
    int okbits = 0;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int y = py0[i] + dy, x = px0[i] + dx;
      bool ok = true;
      int64_t pix;
      if (p.pad_mode == PP_PAD_REPLICATE) {
        const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y);
        const int xc = x < 0 ? 0 : (x >= p.W ? p.W - 1 : x);
        pix = prow[i] + (int64_t)(yc - py0[i]) * p.W + (xc - px0[i]);
      } else {
        ok = ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
        pix = prow[i] + tapoff;
      }
      // segment channel counts are multiples of 4: each half octet is either fully valid or padding
      const bool ok0 = ok && (c0 < it_C), ok1 = ok && (c0 + 4 < it_C);
      const float* src = ok0 ? cbase + pix * it_ldc : it_base;
      gload16_hidden(xreg[P][i][0], src);
      gload16_hidden(xreg[P][i][1], src + (ok1 ? 4 : 0));
      okbits |= (ok0 ? 1 : 0) << (2 * i) | (ok1 ? 2 : 0) << (2 * i);
    }
    xok[P] = okbits;
    advance();
  };
  // split the fetched pixels (split_pair: h round toward zero, saturating; l: the exact remainder, round toward zero)
  // and write one 16-byte octet per plane
  // `later` = vector-memory instructions this thread issued after the loads of register set P (0 or NLOADS): waiting
  // until only those are outstanding retires, in order, this chunk's weight copies and pixel loads.
  auto store_x = [&](auto par, auto later) PP_INLINE_LAMBDA {
    constexpr int P = decltype(par)::value;
    unsigned char* xs = smem + P * XSTAGE;
    wait_vmcnt_hidden<decltype(later)::value>();
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      if (BP % XROWS != 0 && xrow0 + i * XROWS >= BP) continue;
      f4 v[2] = {xreg[P][i][0], xreg[P][i][1]};
      if (!((xok[P] >> (2 * i)) & 1)) v[0] = f4{0.f, 0.f, 0.f, 0.f};
      if (!((xok[P] >> (2 * i)) & 2)) v[1] = f4{0.f, 0.f, 0.f, 0.f};
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
      unsigned char* rowp = xs + (xrow0 + i * XROWS) * ROWB;
      *reinterpret_cast<h8*>(rowp + xoff_h) = h;
      *reinterpret_cast<h8*>(rowp + xoff_l) = l;
    }
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    }

  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int roff_h = (fgrp ^ swz(frow)) << 4;
  const int roff_l = ((fgrp + 4) ^ swz(frow)) << 4;

  auto compute = [&](int xbuf, int wbuf) PP_INLINE_LAMBDA {
    const unsigned char* xs = smem + xbuf * XSTAGE + (wp * TP * 16 + frow) * ROWB;
    const unsigned char* ws = smem + 2 * XSTAGE + wbuf * WSTAGE + (wc * TC * 16 + frow) * ROWB;
    h8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      ah[a] = lds_frag(ws + a * 16 * ROWB + roff_h);
      al[a] = lds_frag(ws + a * 16 * ROWB + roff_l);
    }
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      bh[b] = lds_frag(xs + b * 16 * ROWB + roff_h);
      bl[b] = lds_frag(xs + b * 16 * ROWB + roff_l);
    }
    // three sweeps over the tile grid: two MFMAs on one accumulator are always TC*TP instructions apart.  Product order
    // (high x low, low x high, high x high) = the halo kernel's (conv_halo.hip frees the operands of a step in that order): the two
    // kernels sum every accumulator in the same order, i.e. bit-identical results.
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(ah[a], bl[b], acc[a][b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(al[a], bh[b], acc[a][b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(ah[a], bh[b], acc[a][b]);
  };

  const int nstages = p.nchunks;
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  fetch(0, P0{});
  if (nstages > 1) fetch(1, P1{});
  typedef std::integral_constant<int, 0> L0;
  typedef std::integral_constant<int, NLOADS> LN;
  if (nstages > 1) store_x(P0{}, LN{}); else store_x(P0{}, L0{});
  pp_wait_lgkm0();
  pp_barrier();
  // iteration qs (parity P = qs & 1): pixels of chunk qs are in LDS stage P, weights in stage qs % 3; chunk qs+1 is in
  // registers set 1-P / in flight to weight stage (qs+1) % 3.
  int w0 = 0;  // qs % 3
  auto iteration = [&](int qs, auto par) PP_INLINE_LAMBDA {
    constexpr int P = decltype(par)::value;
    typedef std::integral_constant<int, 1 - P> Q;
    const int w1 = w0 == 2 ? 0 : w0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;
    if (qs + 2 < nstages) fetch(w2, par);  // register set P was stored to LDS one iteration ago
    compute(P, w0);
    if (qs + 1 < nstages) {
      if (qs + 2 < nstages) store_x(Q{}, LN{}); else store_x(Q{}, L0{});
    }
    pp_wait_lgkm0();
    pp_barrier();  // bare barrier: the copies of chunk qs+2 stay in flight across it
    w0 = w1;
  };
  for (int qs = 0; qs < nstages; qs += 2) {
    iteration(qs, P0{});
    if (qs + 1 < nstages) iteration(qs + 1, P1{});
  }

  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  constexpr bool EPI_FITS = WC * WP * epi_lds_wave_bytes<TC>() <= 2 * XSTAGE + 3 * WSTAGE;
  epilogue_any<OT, TC, TP, EPI_FITS, true, true>(
      p, e, smem, wave, lane, c_base + wc * TC * 16,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        m = p_base + wp * TP * 16 + decltype(bi)::value * 16 + frow;
        ok = m < p.M;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA {
        return acc[decltype(ai)::value][decltype(bi)::value];
      },
      [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
        m0 = p_base + wp * TP * 16 + decltype(bi)::value * 16;
        nvalid = (int)(p.M - m0 < 16 ? p.M - m0 : 16);
      });
}

template <typename OT, int WC, int WP, int TC, int TP>
static int launch_split_cfg(void* stream, const ConvK& k, int Z) {
  constexpr int BC = WC * TC * 16;
  constexpr int BP = WP * TP * 16;
  constexpr int NT = WC * WP * 64;
  constexpr int BCP = (BC + NT / 8 - 1) / (NT / 8) * (NT / 8);
  const size_t smem = (size_t)(2 * BP + 3 * BCP) * 128;
  dim3 grid((unsigned)(((k.M + BP - 1) / BP) * ((k.Cout + BC - 1) / BC)), 1u, (unsigned)Z);
  PP_ALLOW_BIG_LDS((&conv_split_kernel<OT, WC, WP, TC, TP>), smem);
  PP_LAUNCH((conv_split_kernel<OT, WC, WP, TC, TP>), grid, dim3(NT), smem, stream, k);
  return pp_check_launch("pp_conv2d");
}

template <typename OT>
struct SplitFamily {
  template <int WC, int WP, int TC, int TP>
  static int run(void* stream, const ConvK& k, int Z) { return launch_split_cfg<OT, WC, WP, TC, TP>(stream, k, Z); }
  static constexpr bool m32_wide96 = false;
  static constexpr int xl_min_blocks = 1024;
};

int launch_split(void* stream, const ConvK& k, int Z) {
  const int rc = launch_halo_split(stream, k, Z);  // stride-1 multi-tap convolutions: pixel tile + halo staged once per chunk
  if (rc != 1) return rc;
  return launch_by_cout<SplitFamily<float>>(stream, k, Z);
}

}  // namespace pp

// conv_halo.hip -- pp_conv2d in PP_F32X2 mode for stride-1 convolutions with more than one tap: the pixel operand is
// staged ONCE per 32-channel chunk as a spatial tile with its halo, and every tap of the chunk reads its MFMA
// fragments from that tile at a shifted row offset.
//
// conv_split_kernel (conv_split.hip) walks K = (segment, channel chunk, tap) with the tap innermost and re-gathers a
// 128-pixel tile for every tap: 9 (3x3) or 5 (1x5 / 5x1) times the same pixels are loaded from L2/HBM, split into
// their two f16 terms in registers and written to LDS, and the gather addresses of every tap are recomputed
// (r01 SQ counters: ~240 non-MFMA instructions per wave and chunk around 48 MFMAs, matrix pipe 32 % busy).  Here the
// output tile is TH x 16 pixels of one image (TH = 8): the input tile (TH + (kh-1) dh) x (16 + (kw-1) dw) pixels is
// loaded, split and stored once per chunk -- 180 rows for 3x3, 160 for 1x5, 192 for 5x1 instead of 9 x 128 / 5 x 128 --
// and the B fragment of tile row y for tap (ky, kx) is the 16 consecutive LDS rows starting at
// (y + ky) * halo_width + kx.  Per tap the work-group only copies the next weight tile (global_load_lds into a 3-stage
// ring), reads fragments and issues MFMAs.
//
// Only the compile-time-tap form ships (RAFT's 3x3, 1x5 and 5x1 convolutions, dilation 1: every PP_F32X2 layer of the
// pipeline with more than one tap).  Its runtime-tap predecessor (any tap shape / dilation, XOR-swizzled pixel tile) is
// archived in tools/experiments/conv_halo_runtime_taps.hip; other geometries take conv_split_kernel.
//
// Arithmetic, weight packing, K order and epilogue are those of conv_split_kernel (same results up to fp32 summation
// order -- identical order in fact: chunk by chunk, tap by tap).  The f16 form lives in conv_halo_f16.hip.
#include "conv_halo_common.h"
#ifdef PP_HALO_TRACE
#include <vector>
#endif

namespace pp {

// ---------------------------------------------------------------------------------------------------------
// Compile-time-tap form (KH x KW known, dilation 1: RAFT's 3x3, 1x5 and 5x1 convolutions).
//
// Measured model behind it (profiles/r02_conv_counters.md, r02_halo_ablation.md): a wave's MFMAs (16 cycles each) and its
// other vector instructions (4 cycles each) add up on the SIMD -- cutting the flat kernel's vector instructions by 38 %
// bought exactly the predicted 0.33 ms on the 3x3 256->192 convolution -- and the runtime-tap halo kernel still issues
// ~120 vector + ~90 scalar instructions around the 48 MFMAs of a tap: fragment addresses (row, swizzle, shift per fragment
// and tap), 64-bit weight addresses per copy, v_readfirstlane for every LDS-DMA destination, the tap / segment iterators.
// Here
//  * the taps are unrolled, so the pixel-fragment address of (tile row b, tap) is ONE per-lane base register plus an
//    immediate offset: the pixel tile uses a padded 160-byte row pitch instead of an XOR swizzle (brute-forced: pitch
//    160 is conflict-free for 16 consecutive rows at any start row against the ds_read_b128 service groups; 144 / 176 /
//    192 are 2-way), and the weight fragments sit on aligned rows (XOR swizzle folded into the per-lane base);
//  * the weight copies use a scalar running pointer plus per-lane 32-bit offsets (no 64-bit vector adds), the wave index is
//    read into a scalar register once, so the LDS-DMA destinations (M0) are scalar arithmetic;
//  * the pipeline decisions (which tap fetches the next pixel tile, which one stores it, the counted waits) are resolved at
//    compile time.
//
// (r05: the 64-channel-tile / three-work-groups-per-CU build of these kernels -- -DPP_HALO_TRIM64, prepared in r04 -- was measured:
//  GRU 1x5 329.8 -> 317.3 TF/s, 5x1 128-channel 290 -> 255, 3x3 unchanged; and the last fragment reads of a step into a second
//  register set -- -DPP_HALO_TAILBUF --: identical times.  Both hooks are archived in tools/experiments/r05_halo_hooks.patch.)
// Phase trace (tools/trace_halo.sh, -DPP_HALO_TRACE; not defined in the product build): every wave accumulates, in scalar
// registers, the s_memtime ticks it spends per tap in (issue = weight / pixel copies issued) (compute = fragment reads + MFMAs)
// (close = counted waits + barrier [+ split / store of the next pixel tile at the last tap of a chunk]) and writes the sums behind
// the epilogue -- no store and no extra wait inside the loop, so the counted vmcnt waits are untouched.
#ifdef PP_HALO_TRACE
#define PP_TR_NOW(v)                        \
  __builtin_amdgcn_sched_barrier(0);        \
  const uint32_t v = (uint32_t)__builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_sched_barrier(0)
#else
#define PP_TR_NOW(v)
#endif

template <int WC, int WP, int TC, int TP, int KH, int KW>
__global__ void __launch_bounds__(WC * WP * 64, 2) conv_halo_split_ct_kernel(const ConvK p, const HaloGeom g) {
  typedef float OT;
  constexpr int NT = WC * WP * 64;
  constexpr int TH = WP * TP;
  constexpr int XROWS = NT / 4, WROWS = NT / 8;
  constexpr int BC = WC * TC * 16;
  constexpr int BCP = (BC + WROWS - 1) / WROWS * WROWS;
  constexpr int ROWB = 128;                      // weight rows
  constexpr int XP = 160;                        // padded pixel-row pitch (bytes): conflict-free at any start row
  constexpr int HW = kHaloTW + KW - 1;           // halo tile width
  constexpr int HROWS = (TH + KH - 1) * HW;
  constexpr int XPASS = (HROWS + XROWS - 1) / XROWS;
  constexpr int WPASS = BCP / WROWS;
  constexpr int XBYTES = XPASS * XROWS * XP, WSTAGE = BCP * ROWB;
  constexpr int NX = 2 * XPASS;
  constexpr int NTAPS = KH * KW;
  static_assert(TH == 8 && NTAPS >= 2 && HROWS <= kHaloMaxRows, "geometry");
  static_assert(WC * WP * epi_lds_wave_bytes<TC>() <= XBYTES + 3 * WSTAGE, "the epilogue's staging rows live in the tile memory");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  PP_TR_NOW(tr_start);
#ifdef PP_HALO_TRACE
  uint32_t tr_issue = 0, tr_comp = 0, tr_close = 0, tr_issue_l = 0, tr_comp_l = 0, tr_close_l = 0;
#endif
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
#ifdef PP_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations become scalar arithmetic
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

  auto swz = [](int r) PP_INLINE_LAMBDA { return ((r >> 1) & 7) ^ ((r & 1) << 2); };  // weight rows (aligned fragments)

  // weights: scalar running pointer + per-lane element offsets
  const int pc = tid & 7;
  const int wrow0 = tid >> 3;
  const int pcs = pc ^ swz(wrow0);
  const float* wptr = reinterpret_cast<const float*>(p.weight) + (int64_t)z * p.w_zoff;
  uint32_t wlane[WPASS];                        // unsigned BYTE offsets: scalar base + 32-bit lane offset addressing
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + wrow0 + i * WROWS;
    wlane[i] = (uint32_t)((co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 4) * 4u;
  }
  // pixels
  const int xj = tid & 3;
  const int xrow0 = tid >> 2;
  int xpix[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int hr = xrow0 + i * XROWS;
    const int hy = hr / HW, hx = hr - hy * HW;
    const int iy = ty0 - p.ph + hy, ix = tx0 - p.pw + hx;
    const bool ok = hr < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xpix[i] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }
  f4 xreg[XPASS][2];
  int xok = 0;

  // K iterators
  int w_rem = 0, w_seg = 0, w_sbase = 0, w_chunks = p.seg_chunks[0];
  const int tapstride = p.chunks_per_tap * 32;
  const uint32_t wring_off = lds_offset_of(smem) + (uint32_t)XBYTES + (uint32_t)(wave * 64 * 16);   // this wave's slot of ring stage 0
  auto fetch_w = [&](int wbuf, int tap) PP_INLINE_LAMBDA {  // weights of (the iterator's chunk, tap) -> ring stage wbuf
    const uint32_t wt = wring_off + (uint32_t)(wbuf * WSTAGE);
    const char* src = reinterpret_cast<const char*>(wptr + (tap * tapstride + w_sbase + w_rem * 32));
#pragma unroll
    for (int i = 0; i < WPASS; ++i) glds16_s(src, wlane[i], wt + (uint32_t)(i * NT * 16));
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
  const float* x_base = reinterpret_cast<const float*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int x_C = p.in_C[0], x_ldc = p.in_ldc[0], x_chunks = p.seg_chunks[0];
  auto fetch_x = [&]() PP_INLINE_LAMBDA {
    const int c0 = x_rem * 32 + xj * 8;
    int okbits = 0;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const bool ok0 = xpix[i] >= 0 && c0 < x_C, ok1 = xpix[i] >= 0 && c0 + 4 < x_C;
      // unsigned 32-bit byte offsets from a scalar base (the launcher checks N*H*W*ldc < 2^30 elements): no 64-bit
      // per-lane address arithmetic and no 64-bit loop invariants to keep in registers
      const uint32_t off = ok0 ? (uint32_t)(xpix[i] * x_ldc + c0) * 4u : 0u;
      gload16_hidden_s(xreg[i][0], x_base, off);
      gload16_hidden_s(xreg[i][1], x_base, off + (ok1 ? 16u : 0u));
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
  // Next to a matrix-bound partner wave every vector-ALU instruction costs about one MFMA slot (r05 trace), so the split does no
  // more than it must: a wave's 16 rows of a pass that lie wholly behind the halo tile are skipped, and when every piece the wave
  // fetched exists (interior tiles, full channel chunks: one ballot per chunk) the zero-fill selects are skipped too.
  auto split_store = [&](int i, auto masked) PP_INLINE_LAMBDA {
    f4 v[2] = {xreg[i][0], xreg[i][1]};
    if constexpr (decltype(masked)::value) {
      if (!((xok >> (2 * i)) & 1)) v[0] = f4{0.f, 0.f, 0.f, 0.f};
      if (!((xok >> (2 * i)) & 2)) v[1] = f4{0.f, 0.f, 0.f, 0.f};
    }
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
    unsigned char* rowp = smem + (xrow0 + i * XROWS) * XP + xj * 16;
    *reinterpret_cast<h8*>(rowp) = h;
    *reinterpret_cast<h8*>(rowp + 64) = l;
  };
  auto store_x = [&]() PP_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      if (wave * 16 + i * XROWS >= HROWS) continue;             // (scalar) this wave's rows of the pass are never read
#ifdef PP_EMU
      const bool all_ok = false;
#else
      const bool all_ok = __builtin_amdgcn_ballot_w64(((xok >> (2 * i)) & 3) != 3) == 0ull;   // (wave-uniform)
#endif
      if (all_ok) split_store(i, std::false_type{}); else split_store(i, std::true_type{});
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
  // per-lane fragment bases; everything else of a fragment address is an immediate
  const unsigned char* xfrag = smem + (wp * TP * HW + frow) * XP + fgrp * 16;
  const unsigned char* wfrag_h = smem + XBYTES + (wc * TC * 16 + frow) * ROWB + ((fgrp ^ swz(frow)) << 4);
  const unsigned char* wfrag_l = smem + XBYTES + (wc * TC * 16 + frow) * ROWB + (((fgrp + 4) ^ swz(frow)) << 4);

  // ---- software-pipelined steps (r05) -----------------------------------------------------------------------------------
  // Step q = chunk * NTAPS + tap; its weights live in ring stage q % 3.  r02-r04 ran every step as copy issue -> fragment reads ->
  // wait -> 48 MFMAs -> waits -> barrier; the s_memtime phase trace (profiles/r05_f32x2_phase_trace.md) shows what that costs a wave
  // that has the SIMD to itself: 342 ticks of copy issue + 1038 of reads and MFMAs (768 of them matrix work) + 92 of closing waits
  // = 1472 per 768, and 1786 when two such waves share a SIMD (1536 would be the matrix pipe's own time).  Now the fragments of
  // step q are IN REGISTERS when the step begins (loaded during step q - 1), the step issues its three products group by group,
  // and behind each group the fragments of step q + 1 replace the operands that just died:
  //     G1  acc += ah x bl   | the copies of step q + 3 (into the stage step q just vacated), the next chunk's pixel fetch,
  //                          | and at a chunk's last tap the split + store of the next pixel tile (+ one extra barrier)
  //     G2  acc += al x bh   | <- bl of step q + 1 (pixel tile)
  //     G3  acc += ah x bh   | <- al of step q + 1 (weights, stage (q + 1) % 3: landed and published by the barrier of step q - 1)
  //     tail                 | <- ah, bh of step q + 1; counted wait for the weights of step q + 2; barrier
  // so no MFMA of a step waits for LDS or for a copy, the live fragment registers never exceed one step's 16, and with ONE
  // accumulator set (64 registers) the wave tile still fits the 256 registers of two work-groups per CU.  A ring stage is
  // restaged one barrier after its last fragment read retired (pp_barrier: lgkmcnt(0) first) -- the r04 rule.
  h8 ah[TC], al[TC], bh[TP], bl[TP];
  auto read_ah = [&](int wbuf) PP_INLINE_LAMBDA {
#pragma unroll
    for (int a = 0; a < TC; ++a) ah[a] = lds_frag(wfrag_h + wbuf * WSTAGE + a * 16 * ROWB);
  };
  auto read_al = [&](int wbuf) PP_INLINE_LAMBDA {
#pragma unroll
    for (int a = 0; a < TC; ++a) al[a] = lds_frag(wfrag_l + wbuf * WSTAGE + a * 16 * ROWB);
  };
  auto read_bh = [&](auto tapc) PP_INLINE_LAMBDA {
    constexpr int tapoff = (decltype(tapc)::value / KW) * HW + (decltype(tapc)::value % KW);
#pragma unroll
    for (int b = 0; b < TP; ++b) bh[b] = lds_frag(xfrag + (b * HW + tapoff) * XP);
  };
  auto read_bl = [&](auto tapc) PP_INLINE_LAMBDA {
    constexpr int tapoff = (decltype(tapc)::value / KW) * HW + (decltype(tapc)::value % KW);
#pragma unroll
    for (int b = 0; b < TP; ++b) bl[b] = lds_frag(xfrag + (b * HW + tapoff) * XP + 64);
  };
  typedef std::integral_constant<int, 0> Tap0;

  static_assert(NTAPS >= 3, "three steps of weights are in flight");
  const int nck = p.chunks_per_tap;
  fetch_x();
  fetch_w(0, 0);
  fetch_w(1, 1);
  fetch_w(2, 2);                               // NTAPS >= 3: steps 0, 1, 2 are taps of chunk 0
  wait_vmcnt_hidden<3 * WPASS>();              // the pixels
  store_x();
  wait_vmcnt_hidden<2 * WPASS>();              // the weights of step 0
  pp_wait_lgkm0();
  pp_barrier();
  read_ah(0);
  read_al(0);
  read_bh(Tap0{});
  read_bl(Tap0{});
  wait_vmcnt_hidden<WPASS>();                  // the weights of step 1
  pp_wait_lgkm0();
  pp_barrier();                                // ... published; every wave holds step 0's fragments: stage 0 is free
  int w0 = 0;                                  // ring stage of the current step
  PP_TR_NOW(tr_loop0);
#ifdef PP_HALO_TRACE
  uint32_t tr_prev = tr_loop0;
#endif
  constexpr int XF = NTAPS - 3;                // the tap that fetches the next chunk's pixels (to registers)
  for (int chunk = 0; chunk < nck; ++chunk) {
    const bool next_chunk = chunk + 1 < nck;
    static_for<NTAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      constexpr bool last = tap == NTAPS - 1;
      typedef std::integral_constant<int, (tap + 1) % NTAPS> NextTap;
      const int w1 = w0 == 2 ? 0 : w0 + 1;
      const bool has3 = (tap + 3 < NTAPS) || next_chunk;   // step q + 3 exists
      const bool has1 = !last || next_chunk;               // step q + 1 exists
      // ---- G1
      if constexpr (tap == XF) {
        if (next_chunk) fetch_x();             // (queue: in front of this step's copies)
      }
      if (has3) {
        if constexpr (tap + 3 == NTAPS) w_next_chunk();    // the iterator moves on when the look-ahead crosses the chunk end
        fetch_w(w0, (tap + 3) % NTAPS);
      }
      if constexpr (last) {
        // the next chunk's pixel tile replaces this one: every wave's last reads of it (step q's own bh / bl, issued in step
        // q - 1) retired before the barrier that closed step q - 1; the pixels were fetched three taps ago and the counted wait
        // of tap XF + 1 retired them
        if (next_chunk) store_x();
      }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(ah[a], bl[b], acc[a][b]);
#ifndef PP_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (last) {
        if (next_chunk) {
          pp_wait_lgkm0();
          pp_barrier();                        // the new pixel tile is complete
        }
      }
      PP_TR_NOW(tr_a);
      // ---- G2
      if (has1) read_bl(NextTap{});
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(al[a], bh[b], acc[a][b]);
#ifndef PP_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      // ---- G3
      if (has1) read_al(w1);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(ah[a], bh[b], acc[a][b]);
#ifndef PP_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      PP_TR_NOW(tr_b);
      // ---- tail
      if (has1) {
        read_ah(w1);
        read_bh(NextTap{});
      }
      // the weights of step q + 2 must have landed; this step's copies (step q + 3) and, at tap XF, the pixel fetch issued in
      // front of them may stay in flight (the queue retires in order: one tap later the pixels are through as well)
      if (has3) {
        if constexpr (tap == XF) {
          if (next_chunk) wait_vmcnt_hidden<WPASS + NX>(); else wait_vmcnt_hidden<WPASS>();
        } else {
          wait_vmcnt_hidden<WPASS>();
        }
      } else {
        wait_vmcnt_hidden<0>();
      }
      pp_wait_lgkm0();
      pp_barrier();
#ifndef PP_EMU
      __builtin_amdgcn_sched_barrier(0);       // unrolled taps: keep the next tap's address arithmetic / reads out of this one
#endif
      PP_TR_NOW(tr_c);
#ifdef PP_HALO_TRACE
      if constexpr (last) {
        tr_issue_l += tr_a - tr_prev; tr_comp_l += tr_b - tr_a; tr_close_l += tr_c - tr_b;
      } else {
        tr_issue += tr_a - tr_prev; tr_comp += tr_b - tr_a; tr_close += tr_c - tr_b;
      }
      tr_prev = tr_c;
#endif
      w0 = w1;
    });
  }
  PP_TR_NOW(tr_loop1);

  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  const int ox = tx0 + frow;
  auto rowfn = [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
    const int oy = ty0 + wp * TP + decltype(bi)::value;
    m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
    ok = oy < p.Ho && ox < p.Wo;
  };
  auto chanfn = [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; };
  auto valfn = [&](auto ai, auto bi) PP_INLINE_LAMBDA {
    return acc[decltype(ai)::value][decltype(bi)::value];
  };
  epilogue_any<OT, TC, TP, true, true, true>(p, e, smem, wave, lane, c_base + wc * TC * 16, rowfn, chanfn, valfn,
                                 [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
                                   const int oy = ty0 + wp * TP + decltype(bi)::value;
                                   m0 = ((int64_t)n * p.Ho + oy) * p.Wo + tx0;
                                   nvalid = oy < p.Ho ? p.Wo - tx0 : 0;
                                 });
#ifdef PP_HALO_TRACE
  PP_TR_NOW(tr_end);
  if (g.trace && lane == 0) {
    uint32_t* t = g.trace + ((size_t)blockIdx.x * (WC * WP) + wave) * 20;
    for (int i = 0; i < 7; ++i) t[12 + i] = e.st[i] - tr_loop1;
    t[0] = tr_loop0 - tr_start; t[1] = tr_loop1 - tr_loop0; t[2] = tr_end - tr_loop1;
    t[3] = tr_issue; t[4] = tr_comp; t[5] = tr_close; t[6] = tr_issue_l; t[7] = tr_comp_l; t[8] = tr_close_l;
    t[9] = (uint32_t)nck; t[10] = tr_start; t[11] = tr_end;
  }
#endif
}

template <int WC, int WP, int TC, int TP, int KH, int KW>
static int launch_halo_ct_cfg(void* stream, const ConvK& k, int Z, HaloGeom g) {
  constexpr int BC = WC * TC * 16;
  constexpr int NT = WC * WP * 64;
  constexpr int BCP = (BC + NT / 8 - 1) / (NT / 8) * (NT / 8);
  constexpr int HROWS = (8 + KH - 1) * (kHaloTW + KW - 1);
  constexpr int XPASS = (HROWS + NT / 4 - 1) / (NT / 4);
  const size_t smem = (size_t)XPASS * (NT / 4) * 160 + (size_t)3 * BCP * 128;
  g.nct = (k.Cout + BC - 1) / BC;
  dim3 grid((unsigned)(g.ntiles * g.nct), 1u, (unsigned)Z);
#ifdef PP_HALO_TRACE
  // PP_HALO_TRACE_SOLO=1: pad the dynamic LDS beyond half a CU's, so ONE work-group is resident per CU (what a wave's tap costs
  // when it has the SIMD to itself)
  static const bool solo = getenv("PP_HALO_TRACE_SOLO") && getenv("PP_HALO_TRACE_SOLO")[0] == '1';
  const size_t smem_l = solo ? (size_t)100 * 1024 : smem;
  const size_t nrec = (size_t)grid.x * (WC * WP) * 20;
  static uint32_t* trace = nullptr;
  static size_t trace_cap = 0;
  if (trace_cap < nrec) {
    if (trace) hipFree(trace);
    hipMalloc(&trace, nrec * 4);
    trace_cap = nrec;
  }
  hipMemsetAsync(trace, 0, nrec * 4, (hipStream_t)stream);
  g.trace = trace;
  PP_ALLOW_BIG_LDS((&conv_halo_split_ct_kernel<WC, WP, TC, TP, KH, KW>), smem_l);
  PP_LAUNCH((conv_halo_split_ct_kernel<WC, WP, TC, TP, KH, KW>), grid, dim3(NT), smem_l, stream, k, g);
  {
    static int launches = 0;
    if (++launches == 3) {   // one warmed-up launch per process and kernel instantiation
      std::vector<uint32_t> h(nrec);
      hipStreamSynchronize((hipStream_t)stream);
      hipMemcpy(h.data(), trace, nrec * 4, hipMemcpyDeviceToHost);
      const size_t nw = nrec / 20;
      double s[9] = {0}, es[7] = {0};
      for (size_t w = 0; w < nw; ++w) {
        for (int i = 0; i < 9; ++i) s[i] += h[w * 20 + i];
        for (int i = 0; i < 7; ++i) es[i] += h[w * 20 + 12 + i];
      }
      fprintf(stderr, "{\"epilogue_stamps_after_loop\": {\"barrier\": %.0f, \"bias\": %.0f, \"row0_loads\": %.0f, \"row0\": %.0f, \"row1\": %.0f, \"row2\": %.0f, \"row3\": %.0f}}\n",
              es[0] / nw, es[1] / nw, es[2] / nw, es[3] / nw, es[4] / nw, es[5] / nw, es[6] / nw);
      // (start / end ticks are per-XCD clocks: the launch's span is read off wave 0's XCD only)
      const int nck = (int)h[9], NTAPS = KH * KW;
      const double taps = (double)nck * (NTAPS - 1), lasts = nck;
      fprintf(stderr,
              "{\"halo_trace\": \"<%d,%d,%d,%d,%d,%d>\", \"solo\": %d, \"Cout\": %d, \"chunks\": %d, \"waves\": %zu, \"prologue\": %.0f, \"loop\": %.0f, "
              "\"epilogue\": %.0f, \"per_tap\": {\"issue\": %.1f, \"compute\": %.1f, \"close\": %.1f}, "
              "\"per_last_tap\": {\"issue\": %.1f, \"compute\": %.1f, \"close\": %.1f}, \"mfma_ticks_per_tap\": %d}\n",
              WC, WP, TC, TP, KH, KW, (int)solo, k.Cout, nck, nw, s[0] / nw, s[1] / nw, s[2] / nw, s[3] / nw / taps, s[4] / nw / taps,
              s[5] / nw / taps, s[6] / nw / lasts, s[7] / nw / lasts, s[8] / nw / lasts, TC * TP * 3 * 16);
    }
  }
  return pp_check_launch("pp_conv2d");
#else
  PP_ALLOW_BIG_LDS((&conv_halo_split_ct_kernel<WC, WP, TC, TP, KH, KW>), smem);
  PP_LAUNCH((conv_halo_split_ct_kernel<WC, WP, TC, TP, KH, KW>), grid, dim3(NT), smem, stream, k, g);
  return pp_check_launch("pp_conv2d");
#endif
}

template <int WC, int WP, int TC, int TP>
static int launch_halo_any(void* stream, const ConvK& k, int Z, const HaloGeom& g);

// returns 1 when the convolution is not eligible (the caller falls back to conv_split_kernel)
int launch_halo_split(void* stream, const ConvK& k, int Z) {
  HaloGeom g;
  if (!halo_geometry(k, Z, kHaloMaxRows, &g)) return 1;
  // (A two-group form -- one 8-wave work-group per CU, two pixel tiles sharing the weight ring, group 1 held half a step
  // behind group 0 by an extra barrier so that one group's MFMA burst always met the other's loads -- was built, checked
  // on the MI355X and measured SLOWER: RAFT GRU 1x5 228 vs 277 TF/s, 3x3 256->192 254 vs 320; removed)
  if (k.Cout > 64) {
    const int waste128 = (k.Cout + 127) / 128 * 128 - k.Cout;
    const int waste96 = (k.Cout + 95) / 96 * 96 - k.Cout;
    if (waste96 + 32 <= waste128) return launch_halo_any<2, 2, 3, 4>(stream, k, Z, g);  //  96 x (8 x 16)
    return launch_halo_any<2, 2, 4, 4>(stream, k, Z, g);                                // 128 x (8 x 16)
  }
  return launch_halo_any<1, 4, 4, 2>(stream, k, Z, g);                                  //  64 x (8 x 16)
}

// the three tap shapes of RAFT (dilation 1, 32-bit element offsets); anything else -- and everything under
// PP_CONV_HALO_CT=0 (A/B runs) -- returns 1: conv_split_kernel
template <int WC, int WP, int TC, int TP>
static int launch_halo_any(void* stream, const ConvK& k, int Z, const HaloGeom& g) {
  int64_t max_ldc = 0;
  for (int sgm = 0; sgm < k.nseg; ++sgm) max_ldc = k.in_ldc[sgm] > max_ldc ? k.in_ldc[sgm] : max_ldc;
  const bool off32 = (int64_t)k.N * k.H * k.W * max_ldc < ((int64_t)1 << 30) && (int64_t)k.Cout * k.Kp < ((int64_t)1 << 30);
  if (options().halo_ct && k.dh == 1 && k.dw == 1 && off32) {
    if (k.kh == 3 && k.kw == 3) return launch_halo_ct_cfg<WC, WP, TC, TP, 3, 3>(stream, k, Z, g);
    if (k.kh == 1 && k.kw == 5) return launch_halo_ct_cfg<WC, WP, TC, TP, 1, 5>(stream, k, Z, g);
    if (k.kh == 5 && k.kw == 1) return launch_halo_ct_cfg<WC, WP, TC, TP, 5, 1>(stream, k, Z, g);
  }
  return 1;
}

}  // namespace pp

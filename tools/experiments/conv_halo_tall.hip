// conv_halo_tall.hip -- PP_F32X2 halo-tile convolution, one wave per SIMD: a 256-thread work-group owns a
// (TC*32 channels) x (16 x 16 pixels) output tile and each of its 4 waves a (TC*16) x (8 rows x 16 pixels) quarter
// with TC x 8 x 2 accumulator quads (256 registers for TC = 4: the kernel is built for ONE work-group per CU, where
// a wave has the 512-register budget, accumulators in AGPRs).
//
// Why: in conv_halo_split_ct_kernel (two 4-wave work-groups per CU) the matrix pipe is ~40 % busy -- a wave issues
// its weight copies, reads 16 fragments, waits, then bursts 48 MFMAs, and the two co-resident waves of a SIMD mostly
// serialise (profiles/r02_halo_ablation.md: the savings of removing MFMAs / loads / fragment reads are additive).
// Here the overlap is built INSIDE the wave instead of hoped for between waves:
//   - a step (one tap of one 32-channel chunk) is 96 MFMAs in four phases of 24 (two tile rows of the wave each);
//   - while a phase multiplies, the wave reads the pixel fragments of the next phase; while the last phase multiplies,
//     it reads the weight fragments and first pixel fragments of the NEXT step (double-buffered fragment registers)
//     and issues the weight copies of the step after that -- LDS reads and LDS-DMA copies ride in the issue slots
//     between MFMAs (16 pipe cycles each; the order is pinned with sched_group_barrier), no wait sits between a
//     read and the MFMAs that follow it;
//   - per MFMA the work-group copies half the weight bytes of the 8-row tile (16 rows share a weight stage) and a
//     smaller halo (324 staged pixels per 256 outputs instead of 180 per 128 for 3x3);
//   - the pixel tile is double-buffered, so the split / store of the next chunk's pixels also runs beside MFMAs.
// One barrier per step, placed before the last phase: it publishes the weights of step q+1 (copied two steps ahead into a
// 3-stage ring) before the last phase of step q reads them, and retires the readers of the stage the next copy overwrites.
//
// Arithmetic, weight packing, K order (chunk by chunk, tap by tap) and epilogue are those of conv_split_kernel /
// conv_halo_split_kernel: same fp32 summation order per output.
#include "conv_halo_common.h"

namespace pp {

template <int TC, int KH, int KW>
__global__ void __launch_bounds__(256, 1) conv_halo_split_tall_kernel(const ConvK p, const HaloGeom g) {
  typedef float OT;
  constexpr int WP = 2, TP = 8;                  // wave grid 2 (channels) x 2 (rows); 8 tile rows per wave ...
  constexpr int HP = 2, NPH = TP / HP;           // ... multiplied in 4 phases of 2 rows
  constexpr int NT = 256;
  constexpr int TH = WP * TP;                    // 16 tile rows
  constexpr int XROWS = NT / 4, WROWS = NT / 8;
  constexpr int BC = 2 * TC * 16;
  static_assert(BC % WROWS == 0, "channel tile");
  constexpr int ROWB = 128;                      // weight rows
  constexpr int XP = 160;                        // padded pixel-row pitch (bytes): conflict-free at any start row
  constexpr int HW = kHaloTW + KW - 1;
  constexpr int HROWS = (TH + KH - 1) * HW;
  constexpr int XPASS = (HROWS + XROWS - 1) / XROWS;
  constexpr int WPASS = BC / WROWS;
  constexpr int XBUF = HROWS * XP, WSTAGE = BC * ROWB;
  constexpr int NX = 2 * XPASS;
  constexpr int NTAPS = KH * KW;
  constexpr float LINV = 1.f / 2048.f;
  static_assert(NTAPS >= 3 && XBUF % 16 == 0, "geometry");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);  // [pixels 0][pixels 1][weights 0][weights 1][weights 2]
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

  auto swz = [](int r) PP_INLINE_LAMBDA { return ((r >> 1) & 7) ^ ((r & 1) << 2); };  // weight rows (aligned fragments)

  // weights: scalar running pointer + per-lane byte offsets
  const int pc = tid & 7;
  const int wrow0 = tid >> 3;
  const int pcs = pc ^ swz(wrow0);
  const float* wptr = reinterpret_cast<const float*>(p.weight) + (int64_t)z * p.w_zoff;
  uint32_t wlane[WPASS];
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
  auto fetch_w = [&](int stage, int tap) PP_INLINE_LAMBDA {  // weights of (the iterator's chunk, tap) -> ring stage
    unsigned char* wt = smem + 2 * XBUF + stage * WSTAGE;
    const char* src = reinterpret_cast<const char*>(wptr + (tap * tapstride + w_sbase + w_rem * 32));
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      if constexpr (!(PP_ABLATE & 4)) glds16(src + (size_t)wlane[i], wt + (i * NT + wave * 64) * 16);
  };
  // (iterator updates are written as selects: a branch would split the scheduling region of the unrolled steps)
  auto w_next_chunk = [&]() PP_INLINE_LAMBDA {
    const bool wrap = w_rem + 1 == w_chunks;
    w_sbase += wrap ? w_chunks * 32 : 0;
    w_seg += wrap ? 1 : 0;
    w_rem = wrap ? 0 : w_rem + 1;
    int nc = w_chunks;
#pragma unroll
    for (int s = 1; s < PP_MAX_SEG; ++s) nc = (wrap && w_seg == s) ? p.seg_chunks[s] : nc;
    w_chunks = nc;
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
      const uint32_t off = ok0 ? (uint32_t)(xpix[i] * x_ldc + c0) * 4u : 0u;
      if constexpr (!(PP_ABLATE & 2)) {
        gload16_hidden_s(xreg[i][0], x_base, off);
        gload16_hidden_s(xreg[i][1], x_base, off + (ok1 ? 16u : 0u));
      } else {
        asm volatile("" : "=v"(xreg[i][0]), "=v"(xreg[i][1]) : "v"(off), "v"(ok1));
      }
      okbits |= ((ok0 ? 1 : 0) | (ok1 ? 2 : 0)) << (2 * i);
    }
    xok = okbits;
    const bool wrap = x_rem + 1 == x_chunks;
    x_seg += wrap ? 1 : 0;
    x_rem = wrap ? 0 : x_rem + 1;
#pragma unroll
    for (int s = 1; s < PP_MAX_SEG; ++s) {
      const bool sw = wrap && x_seg == s && s < p.nseg;
      x_base = sw ? reinterpret_cast<const float*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s] : x_base;
      x_C = sw ? p.in_C[s] : x_C;
      x_ldc = sw ? p.in_ldc[s] : x_ldc;
      x_chunks = sw ? p.seg_chunks[s] : x_chunks;
    }
  };
  auto store_x = [&](int xbuf) PP_INLINE_LAMBDA {
    if constexpr (PP_ABLATE & 64) return;
    unsigned char* xt = smem + xbuf * XBUF;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      if (xrow0 + i * XROWS >= HROWS) continue;  // (only the last pass can run past the halo tile)
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
      unsigned char* rowp = xt + (xrow0 + i * XROWS) * XP + xj * 16;
      *reinterpret_cast<h8*>(rowp) = h;
      *reinterpret_cast<h8*>(rowp + 64) = l;
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
  const unsigned char* xfrag = smem + (wp * TP * HW + frow) * XP + fgrp * 16;
  const unsigned char* wfrag_h = smem + 2 * XBUF + (wc * TC * 16 + frow) * ROWB + ((fgrp ^ swz(frow)) << 4);
  const unsigned char* wfrag_l = smem + 2 * XBUF + (wc * TC * 16 + frow) * ROWB + (((fgrp + 4) ^ swz(frow)) << 4);

  struct FragA {
    h8 h[TC], l[TC];
  };
  struct FragB {
    h8 h[HP], l[HP];
  };
  auto load_a = [&](FragA& f, int stage) PP_INLINE_LAMBDA {
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      f.h[a] = lds_frag(wfrag_h + stage * WSTAGE + a * 16 * ROWB);
      f.l[a] = lds_frag(wfrag_l + stage * WSTAGE + a * 16 * ROWB);
    }
  };
  auto load_b = [&](FragB& f, auto phc, auto tapc, int xbuf) PP_INLINE_LAMBDA {
    constexpr int ph = decltype(phc)::value, tap = decltype(tapc)::value;
    constexpr int tapoff = (tap / KW) * HW + (tap % KW);
    const unsigned char* xb = xfrag + xbuf * XBUF;
#pragma unroll
    for (int b = 0; b < HP; ++b) {
      f.h[b] = lds_frag(xb + ((ph * HP + b) * HW + tapoff) * XP);
      f.l[b] = lds_frag(xb + ((ph * HP + b) * HW + tapoff) * XP + 64);
    }
  };
  auto mfma_phase = [&](const FragA& fa, const FragB& fb, auto phc) PP_INLINE_LAMBDA {
    constexpr int ph = decltype(phc)::value;
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < HP; ++b) acc[a][ph * HP + b] = mfma_16x16x32_f16(fa.h[a], fb.h[b], acc[a][ph * HP + b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < HP; ++b) accx[a][ph * HP + b] = mfma_16x16x32_f16(fa.h[a], fb.l[b], accx[a][ph * HP + b]);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < HP; ++b) accx[a][ph * HP + b] = mfma_16x16x32_f16(fa.l[a], fb.h[b], accx[a][ph * HP + b]);
  };
  using I0 = std::integral_constant<int, 0>;
  // instruction order of a phase for hipcc's scheduler: `n` x (one instruction of class `mask`, then one MFMA); the
  // rest of the phase's MFMAs follow.  0x020 vector-memory reads (the LDS-DMA copies), 0x100 LDS reads, 0x008 MFMA.
  auto interleave = [](auto maskc, auto nc) PP_INLINE_LAMBDA {
#ifndef PP_EMU
    constexpr int mask = decltype(maskc)::value, cnt = decltype(nc)::value;
    static_for<cnt>([&](auto) {
      __builtin_amdgcn_sched_group_barrier(mask, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    });
#endif
  };
  using MaskVmem = std::integral_constant<int, 0x020>;
  using MaskLds = std::integral_constant<int, 0x100>;
  auto phase_fence = []() PP_INLINE_LAMBDA {
#ifndef PP_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
  };

  // ---- pipeline: step q = chunk * NTAPS + tap; weights of step q in ring stage q % 3 (copied two steps ahead: the
  // copies of a step issued one step ahead were measured to arrive too late), pixels of chunk c in buffer c & 1
  const int nck = p.chunks_per_tap;
  fetch_x();
  fetch_w(0, 0);
  wait_vmcnt_hidden<0>();
  store_x(0);
  pp_wait_lgkm0();
  pp_barrier();
  fetch_w(1, 1);                               // NTAPS >= 3: steps 1 and 2 are taps 1 and 2 of chunk 0
  fetch_w(2, 2);
  FragA fa;
  FragB fb;                                    // pixel fragments of the phase about to be multiplied
  load_a(fa, 0);
  load_b(fb, I0{}, I0{}, 0);
  int ws = 0;                                  // ring stage of the current step
  int xb = 0;                                  // pixel buffer of the current chunk
  phase_fence();
  // one chunk = NTAPS steps; `nextc` (compile time): another chunk follows (the last chunk is a second instantiation,
  // so the steady-state steps contain no branch)
  auto run_chunk = [&](auto nextc) PP_INLINE_LAMBDA {
    constexpr bool next_chunk = decltype(nextc)::value;
    static_for<NTAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      constexpr bool x_issue = tap == NTAPS - 3, x_fly = tap == NTAPS - 2, last = tap == NTAPS - 1;
      using NextTap = std::integral_constant<int, (tap + 1) % NTAPS>;
      constexpr bool more1 = (tap + 1 < NTAPS) || next_chunk;  // a step q+1 exists
      constexpr bool more2 = (tap + 2 < NTAPS) || next_chunk;
      constexpr bool more3 = (tap + 3 < NTAPS) || next_chunk;
      constexpr int W2 = more2 ? WPASS : 0;                    // copies of step q+2 in flight at the step's barrier
      const int ws1 = ws == 2 ? 0 : ws + 1;
      FragA na;
      static_for<NPH>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        FragB nb;
        if constexpr (ph == 0 && last && next_chunk) {
          // queue: [weights q+1] [pixels of the next chunk] [weights q+2]: retire the pixels, split them beside phase 0
          wait_vmcnt_hidden<W2>();
          store_x(xb ^ 1);
        }
        if constexpr (ph + 1 < NPH) {
          // ---- multiply rows 2ph, 2ph+1; read the pixel fragments of the next phase beside the MFMAs
          load_b(nb, std::integral_constant<int, ph + 1>{}, tapc, xb);
          mfma_phase(fa, fb, phc);
          interleave(MaskLds{}, std::integral_constant<int, 2 * HP>{});
          fb = nb;
        } else {
          // ---- the step's barrier: weights q+1 landed (every wave waits for its own copies) and visible; every
          // wave has read the weight fragments of step q (stage ws is free) and, in the last tap, stored the next
          // pixel tile.  Then the last phase: copies of step q+3, weight + first pixel fragments of step q+1.
          if constexpr (x_fly && next_chunk) wait_vmcnt_hidden<NX + W2>();  // queue: [weights q+1] [weights q+2] [pixels]
          else wait_vmcnt_hidden<W2>();                                     // queue: [weights q+1] [weights q+2]
          pp_wait_lgkm0();
          pp_barrier();
          if constexpr (more3) {
            if constexpr (tap + 3 == NTAPS) w_next_chunk();
            fetch_w(ws, (tap + 3) % NTAPS);
          }
          if constexpr (x_issue && next_chunk) fetch_x();
          if constexpr (more1) {
            load_a(na, ws1);
            load_b(nb, I0{}, NextTap{}, last ? (xb ^ 1) : xb);
          }
          mfma_phase(fa, fb, phc);
          if constexpr (more3) interleave(MaskVmem{}, std::integral_constant<int, WPASS>{});
          if constexpr (more1) interleave(MaskLds{}, std::integral_constant<int, 2 * TC + 2 * HP>{});
          if constexpr (more1) {
            fa = na;
            fb = nb;
          }
        }
        phase_fence();
      });
      ws = ws1;
      if constexpr (last) xb ^= 1;
    });
  };
  for (int chunk = 0; chunk + 1 < nck; ++chunk) run_chunk(std::true_type{});
  run_chunk(std::false_type{});

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

template <int TC, int KH, int KW>
static int launch_tall_cfg(void* stream, const ConvK& k, int Z, HaloGeom g) {
  constexpr int BC = 2 * TC * 16;
  constexpr int HROWS = (16 + KH - 1) * (kHaloTW + KW - 1);
  const size_t smem = (size_t)2 * HROWS * 160 + (size_t)3 * BC * 128;
  g.nct = (k.Cout + BC - 1) / BC;
  dim3 grid((unsigned)(g.ntiles * g.nct), 1u, (unsigned)Z);
  static const bool lds_ok = (pp_allow_big_lds(reinterpret_cast<const void*>(&conv_halo_split_tall_kernel<TC, KH, KW>), smem), true);
  (void)lds_ok;
  PP_LAUNCH((conv_halo_split_tall_kernel<TC, KH, KW>), grid, dim3(256), smem, stream, k, g);
  return pp_check_launch("pp_conv2d");
}

// EXPERIMENT, off by default: measured on the MI355X (tools/gpu_r2_k.sh, gpu_r2_m.sh) this kernel reaches 223-309 TF/s on the
// RAFT shapes where the 8-row compile-time-tap kernels reach 293-365 -- with one wave per SIMD every exposed cycle (the
// 32-quad epilogue, the prologue's load latency, barrier skew) is dead matrix-pipe time, and two co-resident 4-wave
// work-groups hide more of it than the in-wave pipeline recovers (profiles/r02_tall_experiment.md).
// PP_CONV_HALO_TALL=1 enables it for eligible problems (3x3 / 1x5 / 5x1 taps at dilation 1, more than 64 output
// channels, 32-bit addressable operands, >= 1024 work-groups, no more wasted tile rows than the 8-row tiles), "force"
// lifts the size rules (tests).  Returns 1 when the 8-row halo kernels should run.
int launch_halo_tall(void* stream, const ConvK& k, int Z) {
  const char* e = getenv("PP_CONV_HALO_TALL");
  if (!e || e[0] == '0') return 1;
  const bool force = e[0] == 'f';
  if (k.dh != 1 || k.dw != 1 || k.Cout <= 64) return 1;
  const bool k33 = k.kh == 3 && k.kw == 3, k15 = k.kh == 1 && k.kw == 5, k51 = k.kh == 5 && k.kw == 1;
  if (!(k33 || k15 || k51)) return 1;
  int64_t max_ldc = 0;
  for (int sgm = 0; sgm < k.nseg; ++sgm) max_ldc = k.in_ldc[sgm] > max_ldc ? k.in_ldc[sgm] : max_ldc;
  if ((int64_t)k.N * k.H * k.W * max_ldc >= ((int64_t)1 << 30) || (int64_t)k.Cout * k.Kp >= ((int64_t)1 << 30)) return 1;
  const int waste128 = (k.Cout + 127) / 128 * 128 - k.Cout;
  const int waste96 = (k.Cout + 95) / 96 * 96 - k.Cout;
  const bool c96 = waste96 + 32 <= waste128;
  if (!force) {
    const int64_t ntiles = (int64_t)k.N * ((k.Wo + kHaloTW - 1) / kHaloTW) * ((k.Ho + 15) / 16);
    const int64_t blocks = ntiles * ((k.Cout + (c96 ? 95 : 127)) / (c96 ? 96 : 128)) * Z;
    if (blocks < 1024) return 1;                                   // a few rounds of work-groups per CU at least
    if ((k.Ho + 15) / 16 * 16 > (k.Ho + 7) / 8 * 8) return 1;      // the 8-row tiles waste fewer rows
  }
  HaloGeom g;
  if (!halo_geometry(k, Z, 1 << 30, &g, 16)) return 1;
  if (c96) {
    if (k33) return launch_tall_cfg<3, 3, 3>(stream, k, Z, g);
    if (k15) return launch_tall_cfg<3, 1, 5>(stream, k, Z, g);
    return launch_tall_cfg<3, 5, 1>(stream, k, Z, g);
  }
  if (k33) return launch_tall_cfg<4, 3, 3>(stream, k, Z, g);
  if (k15) return launch_tall_cfg<4, 1, 5>(stream, k, Z, g);
  return launch_tall_cfg<4, 5, 1>(stream, k, Z, g);
}

}  // namespace pp

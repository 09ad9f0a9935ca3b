// conv_ksplit.hip -- pp_conv2d (f16) for problems with FEW output pixels and a LONG reduction: the per-step
// convolutions of the flow-completion recurrence (M = 2 x 45 x 80 = 7200 pixels, Cout = 128, K = 1152 ... 3456).
//
// With 128 x 32 tiles such a problem is 225 work-groups on 256 CUs: one 4-wave work-group per CU walks 36-108 K chunks
// one after the other, each chunk a global -> LDS copy, a barrier and 4 MFMAs per wave -- pure latency (r01: 39 us per
// launch, 3 700 launches per clip), and smaller tiles do not help because the chain length stays the same.  Here the
// work-group has KS = 4 groups of 4 waves; group g reduces the g-th quarter of the K chunks of the SAME 128 x 32 tile with
// its own LDS ring (same copies, swizzle and fragment reads as conv_igemm_kernel's 128 x 32 f16 tile), so four chunk
// chains are in flight per CU and the serial chain is a quarter as long; the partial accumulators meet in LDS and group 0
// runs the fused epilogue.  No extra launch, no HBM round trip for partial sums.
#include "conv_common.h"

namespace pp {

constexpr int kKS = 4;  // K groups per work-group

// NST = ring stages per K group: NST - 1 chunk copies of a group in flight (3 ships).  r04 tested the hypothesis that the 13.8 us
// of the 128 -> 128 step convolution (295 KB of weights per work-group) are the copy stream of a CU with 4 x 2 chunks of 10 KB in
// flight: NST 4 (160 KB of LDS, 4 x 3 in flight; PP_CONV_KSPLIT_NST=4) measured 8.1 vs 7.9 ms over the 470 launches of a clip --
// no gain, the chain is bound by its barrier-per-chunk step, not by the copies.
template <typename OT, int NST>
__global__ void __launch_bounds__(kKS * 256) conv_ksplit_kernel(const ConvK p) {
  typedef half_t T;
  constexpr int NT = 256;                  // threads per K group
  constexpr int WC = 4, TC = 2, TP = 2;    // group = 4 waves stacked over the 128 output channels, 32 x 32 per wave
  constexpr int BC = WC * TC * 16, BP = TP * 16;
  constexpr int BK = 32, EPP = 8, PPR = BK / EPP, LDK = BK;
  constexpr int RPP = NT / PPR;            // 64 tile rows per pass
  constexpr int WPASS = BC / RPP;          // 2
  constexpr int XWAVES = BP * PPR / 64;    // the first 2 waves' worth of lanes cover the pixel tile; the others repeat it
  constexpr int STAGE = (BP + BC) * LDK;   // elements per ring stage
  constexpr int NLOADS = 1 + WPASS;

  T* smem_all = reinterpret_cast<T*>(PP_DYN_SMEM);
  const int tid_all = (int)threadIdx.x;
  const int grp = tid_all >> 8;            // K group
  const int tid = tid_all & 255;
  const int lane = tid & 63;
  const int wave = tid >> 6;               // wave inside the group = its channel block
  T* smem = smem_all + grp * NST * STAGE;
  const int z = (int)blockIdx.z;
  // 32-pixel tiles in XCD-contiguous order (r03; Cout <= 128 here, gridDim.y == 1): dealt round robin, each of the 8 XCDs
  // pulled the whole 2-6 MB input of such a layer through its own L2 -- r03_pmc_fetch_size.json: 18.8 MB read per launch
  // where inputs + weights average ~6.5 MB
  const int ptile = (p.tile_order && gridDim.y == 1) ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int64_t p_base = (int64_t)ptile * BP;
  const int c_base = (int)blockIdx.y * BC;

  const int pc = tid % PPR;
  const int row0 = tid / PPR;
  const int pcs = pc ^ ((row0 >> 1) & (PPR - 1));
  const int wave_x = wave % XWAVES;
  const int row0x = row0 % BP;

  // ---- this thread's pixel row of the X tile and weight rows -----------------------------------------
  int py0, px0;
  int64_t pn, prow;
  {
    const int64_t m = p_base + row0x;
    const int64_t mm = m < p.M ? m : p.M - 1;
    const int wo = (int)(mm % p.Wo);
    const int64_t t = mm / p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    py0 = ho * p.sh - p.ph;
    px0 = wo * p.sw - p.pw;
    pn = (int64_t)n * p.H * p.W;
    prow = pn + (int64_t)py0 * p.W + px0;
  }
  const T* wbase = reinterpret_cast<const T*>(p.weight) + (int64_t)z * p.w_zoff;
  const T* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + row0 + i * RPP;
    wrow[i] = wbase + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * EPP;
  }

  // ---- this group's share of the K chunks (K order: segment, channel chunk, tap -- tap innermost) --------
  const int ntaps = p.kh * p.kw;
  const int per = (p.nchunks + kKS - 1) / kKS;
  const int q0 = grp * per;
  const int nst = q0 >= p.nchunks ? 0 : (p.nchunks - q0 < per ? p.nchunks - q0 : per);
  int it_ky = 0, it_kx = 0, it_seg = 0, it_rem = 0, it_sbase = 0;
  {
    int r = q0;
#pragma unroll
    for (int s = 0; s < PP_MAX_SEG; ++s) {
      const int n_s = s < p.nseg ? p.seg_chunks[s] * ntaps : 0;
      if (it_seg == s && r >= n_s && s + 1 < p.nseg) {
        r -= n_s;
        it_sbase += p.seg_chunks[s] * 32;
        it_seg = s + 1;
      }
    }
    it_rem = r / ntaps;
    const int tap = r - it_rem * ntaps;
    it_ky = tap / p.kw;
    it_kx = tap - it_ky * p.kw;
  }
  const T* it_base = reinterpret_cast<const T*>(p.in_ptr[0]) + (int64_t)z * p.in_zoff[0];
  int it_C = p.in_C[0], it_ldc = p.in_ldc[0], it_chunks = p.seg_chunks[0];
  auto select_segment = [&](int seg) PP_INLINE_LAMBDA {
#pragma unroll
    for (int s = 0; s < PP_MAX_SEG; ++s) {
      if (seg == s) {
        it_base = reinterpret_cast<const T*>(p.in_ptr[s]) + (int64_t)z * p.in_zoff[s];
        it_C = p.in_C[s];
        it_ldc = p.in_ldc[s];
        it_chunks = p.seg_chunks[s];
      }
    }
  };
  select_segment(it_seg);
  auto advance = [&]() PP_INLINE_LAMBDA {
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

  auto dma_stage = [&](int buf) PP_INLINE_LAMBDA {
    T* xt = smem + buf * STAGE;
    T* wt = xt + BP * LDK;
    const int c0 = it_rem * BK + pcs * EPP;
    const int dy = it_ky * p.dh, dx = it_kx * p.dw;
    const int y = py0 + dy, x = px0 + dx;
    bool ok = c0 < it_C;
    int64_t pix;
    if (p.pad_mode == PP_PAD_REPLICATE) {
      const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y);
      const int xc = x < 0 ? 0 : (x >= p.W ? p.W - 1 : x);
      pix = pn + (int64_t)yc * p.W + xc;
    } else {
      ok = ok && (y >= 0) && (y < p.H) && (x >= 0) && (x < p.W);
      pix = prow + (int64_t)dy * p.W + dx;
    }
    const void* src = ok ? static_cast<const void*>(it_base + pix * it_ldc + c0) : static_cast<const void*>(pp_zero16);
    glds16(src, xt + (wave_x * 64) * EPP);
    const int woff = (it_ky * p.kw + it_kx) * p.chunks_per_tap * 32 + it_sbase + it_rem * BK;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) glds16(wrow[i] + woff, wt + (i * NT + wave * 64) * EPP);
    advance();
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int fswz = (frow >> 1) & (PPR - 1);

  auto compute = [&](int buf) PP_INLINE_LAMBDA {
    const T* xs = smem + buf * STAGE + frow * LDK;
    const T* ws = smem + buf * STAGE + BP * LDK + (wave * TC * 16 + frow) * LDK;
    h8 af[TC], bf[TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) af[a] = lds_frag(ws + a * 16 * LDK + (fgrp ^ fswz) * 8);
#pragma unroll
    for (int b = 0; b < TP; ++b) bf[b] = lds_frag(xs + b * 16 * LDK + (fgrp ^ fswz) * 8);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af[a], bf[b], acc[a][b]);
  };

  // this wave's copies of chunk qs have landed when at most min(NST-2, chunks after qs) later chunks are pending
  auto wait_landed = [&](int after) PP_INLINE_LAMBDA {
    static_for<NST - 1>([&](auto ci) {
      constexpr int c = decltype(ci)::value;
      if (after == c || (c == 0 && after < 0) || (c == NST - 2 && after > c)) pp_wait_vmcnt<c * NLOADS>();
    });
  };

  // ---- pipeline: every group runs `per` steps (work-group-wide barriers), the last group may have fewer live ones ----
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (j < nst) dma_stage(j);
  wait_landed(nst - 1);
  pp_barrier();
  int st = 0;
  for (int qs = 0; qs < per; ++qs) {
    if (qs > 0) {
      wait_landed(nst - 1 - qs);
      pp_barrier();  // chunk qs visible to the group; everyone is done reading stage (qs-1) % NST
    }
    if (qs + NST - 1 < nst) dma_stage(st == 0 ? NST - 1 : st - 1);
    if (qs < nst) compute(st);
    st = st + 1 == NST ? 0 : st + 1;
  }

  // ---- reduce the KS partial tiles through LDS (the rings are free after the barrier) ----------------------
  pp_wait_lgkm0();
  pp_barrier();
  f4* red = reinterpret_cast<f4*>(PP_DYN_SMEM);
  if (grp > 0) {
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) red[(((grp - 1) * 4 + wave) * (TC * TP) + a * TP + b) * 64 + lane] = acc[a][b];
  }
  pp_wait_lgkm0();
  pp_barrier();
  if (grp > 0) return;
#pragma unroll
  for (int g = 0; g < kKS - 1; ++g)
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] += red[((g * 4 + wave) * (TC * TP) + a * TP + b) * 64 + lane];

  EpiCtx<OT> e;
  e.bias = p.bias ? p.bias + (int64_t)z * p.bias_zoff : nullptr;
  e.out = reinterpret_cast<OT*>(p.out) + (int64_t)z * p.out_zoff;
  e.aux1 = p.aux1 ? reinterpret_cast<const OT*>(p.aux1) + (int64_t)z * p.aux1_zoff : nullptr;
  e.aux2 = p.aux2 ? reinterpret_cast<const OT*>(p.aux2) + (int64_t)z * p.aux2_zoff : nullptr;
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  // staging rows of the transposed epilogue: behind the partial tiles (the other groups have left; no barrier may follow)
  constexpr size_t RED_BYTES = (size_t)(kKS - 1) * 4 * TC * TP * 64 * sizeof(f4);
  constexpr bool EPI_FITS = RED_BYTES + 4 * epi_lds_wave_bytes<TC>() <= (size_t)kKS * NST * STAGE * sizeof(T);
  epilogue_any<OT, TC, TP, EPI_FITS, false>(
      p, e, reinterpret_cast<unsigned char*>(PP_DYN_SMEM) + RED_BYTES, wave, lane, c_base + wave * TC * 16,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        m = p_base + decltype(bi)::value * 16 + frow;
        ok = m < p.M;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wave * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; },
      [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
        m0 = p_base + decltype(bi)::value * 16;
        nvalid = (int)(p.M - m0 < 16 ? p.M - m0 : 16);
      });
}

template <typename OT, int NST>
static int launch_ksplit_n(void* stream, const ConvK& k, int Z) {
  constexpr size_t smem = (size_t)kKS * NST * (32 + 128) * 32 * sizeof(half_t);  // 120 KiB (NST 3) / 160 KiB (NST 4)
  dim3 grid((unsigned)((k.M + 31) / 32), (unsigned)((k.Cout + 127) / 128), (unsigned)Z);
  PP_ALLOW_BIG_LDS((&conv_ksplit_kernel<OT, NST>), smem);
  PP_LAUNCH((conv_ksplit_kernel<OT, NST>), grid, dim3(kKS * 256), smem, stream, k);
  return pp_check_launch("pp_conv2d");
}

template <typename OT>
static int launch_ksplit_t(void* stream, const ConvK& k, int Z) {
  // (same chunks in the same order per group whatever the ring depth: bit-identical results)
  return options().ksplit_nst == 3 ? launch_ksplit_n<OT, 3>(stream, k, Z) : launch_ksplit_n<OT, 4>(stream, k, Z);
}

// returns 1 when the problem is not a small-image / long-K one (the caller uses the regular tiles).  The decision is a
// function of the LAYER (pixels of one image, Cout, K) and never of the batch: split-K sums the chunks in a different
// order than the flat tiles, and a rank of a sharded run that batches fewer windows / frames than the single-GPU run must
// still pick the same kernel for the same layer so that both runs agree bit for bit (r02 decided on N*Ho*Wo).
// PP_CONV_KSPLIT=0 disables, "force" selects it for every f16 problem with at least 4 chunks (tests) -- pp_options.h.
int launch_ksplit_f16(void* stream, const ConvK& k, int Z, bool out_f16) {
  const int mode = options().ksplit;
  if (mode == 0 || k.nchunks < kKS) return 1;
  if (mode != 2) {
    if (k.many_images) return 1;   // (ABI v12: a layer that always sees a whole sub-video's images -- the halo / flat tiles fill the chip)
    const int64_t img_blocks32 = (((int64_t)k.Ho * k.Wo + 31) / 32) * ((k.Cout + 127) / 128);
    // an image of at most ~160 32-pixel tiles (45x80 of flow completion: 113) and a reduction worth splitting
    // (>= 8 chunks per group)
    if (img_blocks32 > 160 || k.nchunks < 8 * kKS || k.Cout <= 64) return 1;
  }
  return out_f16 ? launch_ksplit_t<half_t>(stream, k, Z) : launch_ksplit_t<float>(stream, k, Z);
}

}  // namespace pp

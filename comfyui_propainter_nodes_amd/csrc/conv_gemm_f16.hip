// conv_gemm_f16.hip -- pp_conv2d (f16) for 1x1 / stride-1 / unpadded single-segment layers: plain GEMMs
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]
// i.e. every Linear of the sparse transformer (sparse_transformer.py:79-123 q|k|v, proj; :413-433 fc1 / fc2 of the fusion
// feed-forward: 48 ms of an 80-frame clip), the soft-split / soft-composite embeddings (:21-77) and the 1x1 GEMM behind the
// deformable convolution's sampled columns (propainter.py:73-82).
//
// The flat implicit-GEMM kernel (conv_igemm_kernel.h) serves these layers with its general machinery: a tap / segment iterator,
// per-piece bounds tests and 64-bit gather addresses every 32-channel chunk, ONE barrier per 16 MFMAs of a wave (r03: 330-540
// TF/s on the transformer's GEMMs while the compile-time 3x3 kernel, 48 MFMAs per barrier, runs 830-950).  A GEMM needs none of
// it: both operands are dense row-major matrices, a thread's copy sources advance by 64 bytes per chunk.  This kernel
//   * stages KC chunks (KC x 32 k) of both operands per barrier by global_load_lds into an NST-stage ring -- same lane-linear
//     LDS image, source-side XOR swizzle and ds_read_b128 fragment reads as the flat kernel's DMA tiles;
//   * keeps the copy sources as (row pointer, running k offset): one 64-bit add per copy, no bounds logic except the
//     channel tail of the LAST chunk (C % 32 != 0: 1960-channel fc2) and the dead chunks of a partial last stage, both
//     served from pp_zero16;
//   * counts vmcnt by hand (NST - 1 stages in flight across a bare s_barrier);
//   * walks its tiles in the flat kernels' XCD-contiguous, channel-adjacent order (flat_tile_of) and ends in the shared
//     fused epilogue.
// It issues exactly the flat kernel's MFMAs in the flat kernel's order (chunk by chunk into the same accumulators), so its
// results are BIT-IDENTICAL to conv_igemm_kernel's: choosing it by problem size cannot make a sharded run differ from a
// single-GPU run (tests/test_conv.py::test_gemm_kernel_equals_flat_kernel).
#include "conv_common.h"

#include <stdio.h>

namespace pp {

// Work-group timeline trace (tools/trace_gemm.sh builds a second copy of this translation unit with -DPP_GEMM_TRACE; not defined in
// the product build): wave 0 of every work-group stamps s_memtime at its start, after its prologue copies are issued, when the
// first stage has landed, at the end of the loop and at its end, plus its XCC, into a device array read back by pp_debug_gemm_trace.
#ifdef PP_GEMM_TRACE
__device__ unsigned long long pp_gemm_trace_buf[8192 * 6];
#define PP_GT_NOW(v)                         \
  __builtin_amdgcn_sched_barrier(0);         \
  const unsigned long long v = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_sched_barrier(0)
#else
#define PP_GT_NOW(v)
#endif
// De-phasing experiment (tools/bench_gemm_dephase.py, -DPP_GEMM_DEPHASE; not defined in the product build): work-groups
// [lo, hi) of a launch sleep `ticks` shader-clock ticks before they start, so that the two work-groups of a CU stop computing
// and storing in lockstep (profiles/r06_gemm_timeline.md).
#ifdef PP_GEMM_DEPHASE
__device__ int pp_gemm_dephase_cfg[4];
#endif

// PATCH (pp_conv2d_params.flat_taps): the pixel operand is F.unfold(x) of a kh x kw / stride / padding patch grid, gathered on
// the fly -- row m = output position (n, i, j), column k = (ky, kx, c); a 16-byte piece is 8 channels of ONE tap (C % 8 == 0),
// so its source is either 8 consecutive channels of one input pixel or, outside the image, zeros.  Same chunks in the same
// order as the GEMM over the materialised patch matrix: bit-identical to unfold + GEMM.
template <typename OT, int WC, int WP, int TC, int TP, int KC, int NST, bool PATCH = false>
__global__ void __launch_bounds__(WC * WP * 64) conv_gemm_f16_kernel(const ConvK p) {
  typedef half_t T;
  constexpr int NT = WC * WP * 64;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int RPP = NT / 4;  // tile rows per copy pass (a 64-byte row = 4 pieces of 16 bytes)
  static_assert(BP % RPP == 0 && BC % RPP == 0, "whole copy passes");
  constexpr int XPASS = BP / RPP, WPASS = BC / RPP;
  constexpr int CH = (BP + BC) * 32;  // elements of one chunk image: [X: BP rows | W: BC rows][32]
  constexpr int STAGE = KC * CH;
  constexpr int NL = KC * (XPASS + WPASS);  // copies per thread and stage
  static_assert((NST - 2) * NL <= 63, "vmcnt is a 6-bit counter");

  PP_GT_NOW(gt_start);
#ifdef PP_GEMM_DEPHASE
  {
    const int b = (int)blockIdx.x, mod = pp_gemm_dephase_cfg[3];
    const bool late = mod > 0 ? (b / mod) % 2 == 1 && b < pp_gemm_dephase_cfg[1] : (b >= pp_gemm_dephase_cfg[0] && b < pp_gemm_dephase_cfg[1]);
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while ((long long)(__builtin_amdgcn_s_memtime() - t0) < pp_gemm_dephase_cfg[2]) __builtin_amdgcn_s_sleep(8);
    }
  }
#endif
  T* smem = reinterpret_cast<T*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WP, wp = wave % WP;
  int tile_p, tile_c;
  flat_tile_of(p, BC, tile_p, tile_c);
  const int64_t p_base = (int64_t)tile_p * BP;
  const int c_base = tile_c * BC;

  // this thread's slot of every copy pass: LDS piece pc of tile row row0 (+ i * RPP) holds global piece pcs = pc ^ swz(row)
  const int pc = tid & 3, row0 = tid >> 2;
  const int pcs = pc ^ ((row0 >> 1) & 3);  // (RPP is a multiple of 16: the swizzle is the same in every pass)
  const T* xsrc[XPASS];
  const T* wsrc[WPASS];
  int py0[XPASS], px0[XPASS];   // PATCH: input coordinates of tap (0, 0) of this thread's rows
  {
    const T* xb = reinterpret_cast<const T*>(p.in_ptr[0]);
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int64_t m = p_base + row0 + i * RPP;
      const int64_t mm = m < p.M ? m : p.M - 1;  // rows past M: clamped, results never stored
      if constexpr (PATCH) {
        const int wo = (int)(mm % p.Wo);
        const int64_t t = mm / p.Wo;
        const int ho = (int)(t % p.Ho);
        const int64_t n = t / p.Ho;
        py0[i] = ho * p.sh - p.ph;
        px0[i] = wo * p.sw - p.pw;
        xsrc[i] = xb + n * p.H * p.W * p.in_ldc[0];   // image base; pixel and channel offsets per piece
      } else {
        py0[i] = px0[i] = 0;
        xsrc[i] = xb + mm * p.in_ldc[0] + pcs * 8;
      }
    }
    const T* wb = reinterpret_cast<const T*>(p.weight);
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      const int co = c_base + row0 + i * RPP;
      wsrc[i] = wb + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 8;
    }
  }
  const int C = PATCH ? p.in_C[0] * p.kh * p.kw : p.in_C[0];   // valid k
  const int Cin = p.in_C[0], ldc = p.in_ldc[0];
  const float inv_c = 1.f / (float)Cin, inv_kw = 1.f / (float)p.kw;   // (k + 0.5) * inv is exact for these small integers
  const int nchunks = p.nchunks;
  int q = 0;  // next chunk to copy

  auto dma_stage = [&](int buf) PP_INLINE_LAMBDA {
    static_for<KC>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      T* xt = smem + buf * STAGE + kc * CH;
      T* wt = xt + BP * 32;
      const int k0 = q * 32;
      const bool live = q < nchunks;                  // (a partial last stage copies zeros for its dead chunks)
      const bool xok = live && (k0 + pcs * 8 < C);    // channel tail of the last chunk (its weights are zero-padded)
      if constexpr (PATCH) {
        const int kk = k0 + pcs * 8;
        const int tap = (int)(((float)kk + 0.5f) * inv_c);
        const int c = kk - tap * Cin;
        const int ky = (int)(((float)tap + 0.5f) * inv_kw);
        const int kx = tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
          const int y = py0[i] + ky * p.dh, x = px0[i] + kx * p.dw;
          const bool in = xok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
          const T* src = xsrc[i] + ((int64_t)y * p.W + x) * ldc + c;
          glds16(in ? static_cast<const void*>(src) : static_cast<const void*>(pp_zero16), xt + (i * NT + wave * 64) * 8);
        }
      } else {
#pragma unroll
        for (int i = 0; i < XPASS; ++i)
          glds16(xok ? static_cast<const void*>(xsrc[i] + k0) : static_cast<const void*>(pp_zero16), xt + (i * NT + wave * 64) * 8);
      }
#pragma unroll
      for (int i = 0; i < WPASS; ++i)
        glds16(live ? static_cast<const void*>(wsrc[i] + k0) : static_cast<const void*>(pp_zero16), wt + (i * NT + wave * 64) * 8);
      ++q;
    });
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fgrp = lane >> 4;
  const int fpiece = (fgrp ^ ((frow >> 1) & 3)) * 8;

  auto compute = [&](int buf) PP_INLINE_LAMBDA {
    static_for<KC>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      const T* xs = smem + buf * STAGE + kc * CH + (wp * TP * 16 + frow) * 32 + fpiece;
      const T* ws = smem + buf * STAGE + kc * CH + BP * 32 + (wc * TC * 16 + frow) * 32 + fpiece;
      h8 af[TC], bf[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) af[a] = lds_frag(ws + a * 16 * 32);
#pragma unroll
      for (int b = 0; b < TP; ++b) bf[b] = lds_frag(xs + b * 16 * 32);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af[a], bf[b], acc[a][b]);
    });
  };

  const int nstages = (nchunks + KC - 1) / KC;
  // prologue: stages 0 .. NST-2 in flight
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (j < nstages) dma_stage(j);
  // this wave's copies of stage qs have landed when at most min(NST-2, stages after qs) later stages are pending
  auto wait_landed = [&](int after) PP_INLINE_LAMBDA {
    static_for<NST - 1>([&](auto ci) {
      constexpr int c = decltype(ci)::value;
      if (after == c || (c == NST - 2 && after > c)) pp_wait_vmcnt<c * NL>();
    });
  };
  PP_GT_NOW(gt_issued);
#ifdef PP_GEMM_TRACE
  unsigned long long gt_first = 0;
#endif
  int st = 0;
  for (int qs = 0; qs < nstages; ++qs) {
    wait_landed(nstages - 1 - qs);
    pp_barrier();  // every wave's part of stage qs is visible; everyone is done reading stage (qs-1) % NST
#ifdef PP_GEMM_TRACE
    if (qs == 0) {
      PP_GT_NOW(gt_f);
      gt_first = gt_f;
    }
#endif
    if (qs + NST - 1 < nstages) dma_stage(st == 0 ? NST - 1 : st - 1);
    compute(st);
    st = st + 1 == NST ? 0 : st + 1;
  }
  PP_GT_NOW(gt_loop);

  EpiCtx<OT> e;
  e.bias = p.bias;
  e.out = reinterpret_cast<OT*>(p.out);
  e.aux1 = reinterpret_cast<const OT*>(p.aux1);
  e.aux2 = reinterpret_cast<const OT*>(p.aux2);
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  // (r05: the LDS-transposed epilogue LOSES on these short-reduction GEMMs -- fc1 631 -> 460, qkv 588 -> 459 TF/s: four exposed LDS
  //  round trips per wave against a loop of ~8 000 cycles -- so they keep the direct quads: EPI_FITS = false)
  constexpr bool EPI_FITS = false;
  // r06: f16 outputs leave the wave as 16-byte stores of 8 consecutive channels (conv_common.h: epilogue_octs_fast) whenever every
  // tensor of the launch allows it -- all of the transformer's Linears do
  bool stored = false;
  if constexpr (sizeof(OT) == 2 && TC % 2 == 0) {
    if (p.epi_oct && epi_oct_ok<OT>(p, e)) {
      stored = true;
      epilogue_octs_fast<TC, TP>(
          p, e, c_base + wc * TC * 16, fgrp,
          [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
            m = p_base + wp * TP * 16 + decltype(bi)::value * 16 + frow;
            ok = m < p.M;
          },
          [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; });
    }
  }
  if (!stored) epilogue_any<OT, TC, TP, EPI_FITS>(
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
#ifdef PP_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_GT_NOW(gt_end);
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = pp_gemm_trace_buf + (size_t)blockIdx.x * 6;
    t[0] = gt_start; t[1] = gt_issued; t[2] = gt_first; t[3] = gt_loop; t[4] = gt_end; t[5] = xcc & 15;
  }
#endif
}

template <typename OT, int WC, int WP, int TC, int TP, int KC, int NST, bool PATCH = false>
static int launch_gemm_cfg(void* stream, const ConvK& k) {
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr size_t smem = (size_t)NST * KC * (BC + BP) * 32 * sizeof(half_t);
  static_assert(smem <= 160 * 1024, "LDS of one CU");
  dim3 grid((unsigned)(((k.M + BP - 1) / BP) * ((k.Cout + BC - 1) / BC)), 1u, 1u);
  PP_ALLOW_BIG_LDS((&conv_gemm_f16_kernel<OT, WC, WP, TC, TP, KC, NST, PATCH>), smem);
  PP_LAUNCH((conv_gemm_f16_kernel<OT, WC, WP, TC, TP, KC, NST, PATCH>), grid, dim3(WC * WP * 64), smem, stream, k);
  return pp_check_launch("pp_conv2d");
}

template <typename OT>
static int launch_gemm_t(void* stream, const ConvK& k, int cfg) {
  // cfg (PP_CONV_GEMM_CFG, tuning / tests): 0 = by the layer's Cout.  Measured on the MI355X at the transformer's shapes
  // (tools/bench_gemm.py, profiles/r04_bench_gemm.md; flat kernel 463 / 474 / 556 / 652 TF/s on qkv / proj / fc1 / fc2):
  //   1: 256 x 128, 8 waves, 2 chunks per barrier, 3 stages (144 KB: one work-group per CU)   374 / 375 / 438 / 566
  //   2: 128 x 128, 4 waves, 2 chunks per barrier, 2 stages ( 64 KB: two per CU)              386 / 384 / 450 / 568
  //   3: 128 x 256, 8 waves, 2 chunks per barrier, 3 stages (144 KB)                          378 / 374 / 442 / 576
  //   4: 128 x 128, 4 waves, 1 chunk  per barrier, 4 stages ( 64 KB)                          388 / 376 / 444 / 569
  //   5: 256 x 128, 8 waves, 1 chunk  per barrier, 3 stages ( 72 KB: two per CU)              568 / 513 / 651 / 711
  //   6: 128 x 256, 8 waves, 1 chunk  per barrier, 3 stages ( 72 KB)
  //   7: 128 x 128, 4 waves, 1 chunk  per barrier, 3 stages ( 48 KB: three per CU)   r06, paired-quad stores: 437-488 / 365-374 / 465-491
  //      against 613-682 / 475-499 / 626-689 for 5 on the same box (tools/bench_gemm_cfg.py): three independent phases per CU do not
  //      make up for twice the operand copies per MFMA
  // Fewer barriers per MFMA (1-3) do not pay for halving the resident waves: 4 waves per SIMD is what hides the copies.
  if (cfg == 0) cfg = (k.Cout + 255) / 256 * 256 - k.Cout <= k.Cout / 8 ? 5 : 6;
  if (options().trace) fprintf(stderr, "pp_conv2d: GEMM kernel cfg %d, M %lld, K %d, Cout %d\n", cfg, (long long)k.M, k.in_C[0], k.Cout);
  switch (cfg) {
    case 1: return launch_gemm_cfg<OT, 4, 2, 4, 4, 2, 3>(stream, k);
    case 2: return launch_gemm_cfg<OT, 2, 2, 4, 4, 2, 2>(stream, k);
    case 3: return launch_gemm_cfg<OT, 2, 4, 4, 4, 2, 3>(stream, k);
    case 4: return launch_gemm_cfg<OT, 2, 2, 4, 4, 1, 4>(stream, k);
    case 6: return launch_gemm_cfg<OT, 2, 4, 4, 4, 1, 3>(stream, k);
    case 7: return launch_gemm_cfg<OT, 2, 2, 4, 4, 1, 3>(stream, k);   // r06: 128 x 128, 4 waves, 3 stages (48 KB: three per CU)
    default: return launch_gemm_cfg<OT, 4, 2, 4, 4, 1, 3>(stream, k);
  }
}

// returns 1 when the layer is not a plain GEMM (the caller uses the flat tiles).  PP_CONV_GEMM=0 disables, "force" takes every
// eligible layer whatever its size (tests).
int launch_gemm_f16(void* stream, const ConvK& k, int Z, bool out_f16) {
  const Options& o = options();
  if (o.gemm == 0) return 1;
  if (Z != 1 || k.nseg != 1 || k.kh != 1 || k.kw != 1 || k.sh != 1 || k.sw != 1 || k.ph != 0 || k.pw != 0) return 1;
  if (k.Ho != k.H || k.Wo != k.W || k.pad_mode == PP_PAD_REPLICATE) return 1;
  if (o.gemm != 2) {
    // large problems only: below ~2 work-groups per CU the 32- / 16-pixel flat tiles and the split-K kernel fill the chip better
    const int64_t blocks128 = ((k.M + 127) / 128) * ((k.Cout + 127) / 128);
    if (blocks128 < 512 || k.Cout < 96) return 1;
  }
  return out_f16 ? launch_gemm_t<half_t>(stream, k, o.gemm_cfg) : launch_gemm_t<float>(stream, k, o.gemm_cfg);
}


// pp_conv2d_params.flat_taps: conv(x) == linear(unfold(x)) with the patches gathered inside the GEMM kernel
int launch_gemm_f16_patch(void* stream, const ConvK& k, int Z, bool out_f16) {
  if (Z != 1 || k.nseg != 1 || (k.in_C[0] & 7) || k.pad_mode == PP_PAD_REPLICATE)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: flat_taps needs one f16 segment with C % 8 == 0, zero padding, Z 1");
  if (options().trace) fprintf(stderr, "pp_conv2d: GEMM kernel, patch gather %dx%d s%d, C %d, M %lld, Cout %d\n", k.kh, k.kw, k.sh, k.in_C[0], (long long)k.M, k.Cout);
  const bool wide = (k.Cout + 255) / 256 * 256 - k.Cout <= k.Cout / 8;
  if (wide) return out_f16 ? launch_gemm_cfg<half_t, 4, 2, 4, 4, 1, 3, true>(stream, k) : launch_gemm_cfg<float, 4, 2, 4, 4, 1, 3, true>(stream, k);
  return out_f16 ? launch_gemm_cfg<half_t, 2, 4, 4, 4, 1, 3, true>(stream, k) : launch_gemm_cfg<float, 2, 4, 4, 4, 1, 3, true>(stream, k);
}

}  // namespace pp

#ifdef PP_GEMM_DEPHASE
extern "C" int32_t pp_debug_gemm_dephase(int32_t lo, int32_t hi, int32_t ticks, int32_t mod) {
  const int v[4] = {lo, hi, ticks, mod};
  return hipMemcpyToSymbol(HIP_SYMBOL(pp::pp_gemm_dephase_cfg), v, sizeof(v)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef PP_GEMM_TRACE
// (trace build only) copies the work-group timeline of the LAST conv_gemm_f16_kernel launch to the host: 6 x uint64 per work-group
extern "C" int32_t pp_debug_gemm_trace(void* host_out, int64_t nwg) {
  if (nwg > 8192) nwg = 8192;
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pp::pp_gemm_trace_buf), (size_t)nwg * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

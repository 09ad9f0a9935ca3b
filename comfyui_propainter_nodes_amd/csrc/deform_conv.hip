// deform_conv.hip -- pp_deform_conv: the modulated deformable 3x3 convolution of the two recurrences
// (torchvision.ops.deform_conv2d at recurrent_flow_completion.py:44-53 and propainter.py:73-82) as ONE kernel: the
// bilinear sampling of pp_deform_cols feeds the MFMA B operand from registers, the 9*Cin-channel column tensor is never
// written (r02/r03: pp_deform_cols wrote it -- 265 MB per feature-propagation step -- and a 1x1 pp_conv2d read it back).
//
// Formulation (that of the 1x1 convolution over the columns):  D[cout][pixel] = sum_q W[cout][q*32 .. +32] . col[pixel][q*32 .. +32],
// chunk q = (tap, 32-channel chunk of the input).  v_mfma_f32_16x16x32_f16 takes its B operand as: lane l holds the 8
// consecutive k of k-group l>>4 for column l&15 -- for a chunk of the columns that is "the 8 channels [kg*8, kg*8+8) of
// pixel l&15 sampled at tap t", i.e. exactly one item of the sampling kernel (one 16-byte load per bilinear corner, 8
// channels never straddle a deformable group).  So every lane samples its own fragment: 3 offset / mask loads, 4 corner
// loads, the same fp32 blend in the same order as deform_cols_kernel, one rounding to f16 -- the values the column tensor
// would have held, bit for bit -- and nothing of the pixel operand goes through LDS.
//
// Work-group = KS K-groups x NW waves.  A wave owns ALL 128 output channels (8 A fragments per chunk, read from an LDS
// weight tile the NW waves of its group share: global_load_lds, 2 stages, one barrier per chunk) of its own TP x 16
// pixels, so no pixel is sampled twice.  The K loop is latency-bound (offsets -> corner addresses -> corners -> blend ->
// MFMA); the loads of chunk q+1 are issued before the MFMAs of chunk q and the offsets of chunk q+2 with them, the rest
// is covered by occupancy.  KS > 1 (flow completion: 2 x 45 x 80 pixels, 72 chunks) splits the chunk range over KS groups
// working on the same pixel tile -- KS chains in flight per work-group, each 1/KS as long -- and the partial accumulators
// meet in LDS, as in conv_ksplit.hip (same chunk ranges, same summation order).  Bias / activation / fused epilogue: conv_common.h.
// The shipped form is KS = 4 groups of 2 waves x 16 pixels, used by the host for small images with a long reduction --
// conv_ksplit.hip's selection rule, so that either form of the deformable convolution sums in the same order and the
// results are bit-identical.  Work-groups walk the pixels in XCD-contiguous order (pp_device.h): dealt round robin, every
// XCD's L2 saw every feature line and the kernel ran 1.2-1.8x slower.
#include "conv_common.h"

namespace pp {

struct DeformSrc {
  const void* x0;
  int x0_C, x0_ldc;
  const void* x1;
  int x1_C, x1_ldc;
  const float* om;
  int om_ldc;
  const float* flow;
  int flow_ldc;
  int H, W, dg, cg;
  int cchunks;  // 32-channel chunks of the input (Cin / 32)
  int xcd;      // work-groups in XCD-contiguous order (pp_device.h)
};

// what a lane keeps between issuing the corner loads of a chunk and blending them
struct DeformRaw {
  h8 q00, q01, q10, q11;
  float w00, w01, w10, w11;
  int ok;  // bit 0..3: corner 00, 01, 10, 11 contributes
};

template <typename OT, int NW, int KS, int TP>
__global__ void __launch_bounds__(KS * NW * 64) deform_conv_kernel(const DeformSrc d, const ConvK p) {
  typedef half_t T;
  constexpr int TC = 8, BC = TC * 16;      // 128 output channels per work-group, all of them in every wave
  constexpr int BP = NW * TP * 16;         // pixels per work-group
  constexpr int NT = NW * 64;              // threads per K group
  constexpr int LDK = 32, EPP = 8, PPR = 4;
  constexpr int RPP = NT / PPR;            // weight-tile rows copied per pass of the group
  constexpr int WPASS = BC / RPP;
  constexpr int STAGE = BC * LDK;          // elements per weight stage (8 KiB)
  constexpr int NST = 2;
  static_assert(BC % RPP == 0 && RPP % 8 == 0, "weight tile passes");

  T* smem_all = reinterpret_cast<T*>(PP_DYN_SMEM);
  const int tid_all = (int)threadIdx.x;
  const int grp = tid_all / NT;
  const int tid = tid_all - grp * NT;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  T* smem = smem_all + grp * NST * STAGE;
  const int pblk = d.xcd ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int64_t p_base = (int64_t)pblk * BP;
  const int c_base = (int)blockIdx.y * BC;

  // ---- weight tile: lane-linear DMA image, LDS piece slot pc of row r holds source piece pc ^ swz(r) ------------
  const int pc = tid % PPR;
  const int row0 = tid / PPR;
  const int pcs = pc ^ ((row0 >> 1) & (PPR - 1));
  const T* wrow[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int co = c_base + row0 + i * RPP;
    wrow[i] = reinterpret_cast<const T*>(p.weight) + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * EPP;
  }

  // ---- this lane's pixels (one per 16-pixel sub-tile) and channel piece -----------------------------------------
  const int frow = lane & 15;
  const int fgrp = lane >> 4;
  const int fswz = (frow >> 1) & (PPR - 1);
  const int hw = d.H * d.W;
  int py[TP], px[TP];
  int64_t pimg[TP];        // pixel index of the image's first pixel
  const float* omrow[TP];
  float fdx[TP], fdy[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    const int64_t m = p_base + (wave * TP + b) * 16 + frow;
    const int64_t mm = m < p.M ? m : p.M - 1;  // pixels past M: clamped, results never stored
    const int q = (int)(mm % hw);
    py[b] = q / d.W;
    px[b] = q - py[b] * d.W;
    pimg[b] = mm - q;
    omrow[b] = d.om + mm * d.om_ldc;
    fdx[b] = 0.f;
    fdy[b] = 0.f;
    if (d.flow) {
      const float* f = d.flow + mm * d.flow_ldc;
      fdx[b] = f[0];
      fdy[b] = f[1];
    }
  }

  // ---- this group's share of the chunks ------------------------------------------------------------------------
  const int per = (p.nchunks + KS - 1) / KS;
  const int q0 = grp * per;
  const int nst = q0 >= p.nchunks ? 0 : (p.nchunks - q0 < per ? p.nchunks - q0 : per);

  auto dma_weights = [&](int buf, int q) PP_INLINE_LAMBDA {
    T* wt = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) glds16(wrow[i] + (int64_t)q * 32, wt + (i * NT + wave * 64) * EPP);
  };

  // offsets and modulation mask of (pixel, tap, deformable group of this lane's channel piece) for chunk q
  float o_dy[TP], o_dx[TP], o_m[TP];
  auto load_offsets = [&](int q) PP_INLINE_LAMBDA {
    const int tap = q / d.cchunks;
    const int cc = q - tap * d.cchunks;
    const int g = (cc * 32 + fgrp * 8) / d.cg;
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      o_dy[b] = omrow[b][g * 18 + 2 * tap];
      o_dx[b] = omrow[b][g * 18 + 2 * tap + 1];
      o_m[b] = omrow[b][d.dg * 18 + g * 9 + tap];
    }
  };

  // corner addresses / weights from the offsets in o_* and the four 16-byte corner loads (unconditional: a corner
  // that does not contribute reads the first pixel of the image; deform_cols_kernel's arithmetic, same order).
  // The two inputs' per-lane bases live in registers: picking d.x0 / d.x1 inside the loop made the compiler index the
  // kernel-argument block in memory (a dependent load and a full vmcnt drain per chunk).
  // Per sub-tile: the image base of this lane's channel piece in either input (64-bit, once); inside the loop the corner
  // offsets are 32-bit element offsets within one image.
  const int ldc_a = d.x0_ldc, ldc_b = d.x1 ? d.x1_ldc : d.x0_ldc, split_c = d.x0_C;
  const T* img_a[TP];
  const T* img_b[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    img_a[b] = reinterpret_cast<const T*>(d.x0) + fgrp * 8 + pimg[b] * ldc_a;
    img_b[b] = d.x1 ? reinterpret_cast<const T*>(d.x1) + fgrp * 8 - d.x0_C + pimg[b] * ldc_b : img_a[b];
  }
  DeformRaw raw[TP];
  auto issue_corners = [&](int q) PP_INLINE_LAMBDA {
    const int tap = q / d.cchunks;
    const int cc = q - tap * d.cchunks;
    const bool first = cc * 32 < split_c;  // wave-uniform: x0_C is a multiple of 32
    const int ldc = first ? ldc_a : ldc_b;
    const int rowstep = d.W * ldc;
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const float dy = o_dy[b] + fdy[b], dx = o_dx[b] + fdx[b];
      const float m = o_m[b];
      float fpy = (float)(py[b] - 1 + ky) + dy;
      float fpx = (float)(px[b] - 1 + kx) + dx;
      if (!(fabsf(fpy) < 1.0e8f)) fpy = -1.0e8f;  // NaN / Inf offsets: out of range, float->int conversions stay defined
      if (!(fabsf(fpx) < 1.0e8f)) fpx = -1.0e8f;
      const bool inside = (fpy > -1.f) && (fpy < (float)d.H) && (fpx > -1.f) && (fpx < (float)d.W);
      const float fy = floorf(fpy), fx = floorf(fpx);
      const int y0 = (int)fy, x0 = (int)fx;
      const float ly = fpy - fy, lx = fpx - fx;
      DeformRaw& r = raw[b];
      r.w00 = (1.f - ly) * (1.f - lx) * m;
      r.w01 = (1.f - ly) * lx * m;
      r.w10 = ly * (1.f - lx) * m;
      r.w11 = ly * lx * m;
      const bool y0ok = inside && y0 >= 0, y1ok = inside && (y0 + 1 <= d.H - 1);
      const bool x0ok = x0 >= 0, x1ok = (x0 + 1 <= d.W - 1);
      const bool k00 = y0ok && x0ok, k01 = y0ok && x1ok, k10 = y1ok && x0ok, k11 = y1ok && x1ok;
      r.ok = (k00 ? 1 : 0) | (k01 ? 2 : 0) | (k10 ? 4 : 0) | (k11 ? 8 : 0);
      const T* img = (first ? img_a[b] : img_b[b]) + cc * 32;
      const int o00 = (y0 * d.W + x0) * ldc;   // (only used when the corner lies inside the image: no overflow)
      r.q00 = *reinterpret_cast<const h8*>(img + (k00 ? o00 : 0));
      r.q01 = *reinterpret_cast<const h8*>(img + (k01 ? o00 + ldc : 0));
      r.q10 = *reinterpret_cast<const h8*>(img + (k10 ? o00 + rowstep : 0));
      r.q11 = *reinterpret_cast<const h8*>(img + (k11 ? o00 + rowstep + ldc : 0));
    }
  };

  h8 bf[TP];
  // deform_cols_kernel's blend (v = 0; v += w * q per contributing corner, in corner order; one rounding to f16) on channel
  // PAIRS: the same IEEE operations per element, issued as packed fp32 multiplies / adds
  auto blend = [&]() PP_INLINE_LAMBDA {
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const DeformRaw& r = raw[b];
      const bool k00 = r.ok & 1, k01 = r.ok & 2, k10 = r.ok & 4, k11 = r.ok & 8;
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f2 v = {0.f, 0.f};
        const f2 t00 = v + r.w00 * f2{(float)r.q00[e], (float)r.q00[e + 1]};
        v = k00 ? t00 : v;
        const f2 t01 = v + r.w01 * f2{(float)r.q01[e], (float)r.q01[e + 1]};
        v = k01 ? t01 : v;
        const f2 t10 = v + r.w10 * f2{(float)r.q10[e], (float)r.q10[e + 1]};
        v = k10 ? t10 : v;
        const f2 t11 = v + r.w11 * f2{(float)r.q11[e], (float)r.q11[e + 1]};
        v = k11 ? t11 : v;
        o[e] = sat_half(v[0]);
        o[e + 1] = sat_half(v[1]);
      }
      bf[b] = o;
    }
  };

  f4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) PP_INLINE_LAMBDA {
    const T* ws = smem + buf * STAGE + frow * LDK + (fgrp ^ fswz) * 8;
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const h8 af = lds_frag(ws + a * 16 * LDK);
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = mfma_16x16x32_f16(af, bf[b], acc[a][b]);
    }
  };

  // ---- pipeline.  Step qs: copy the weights of chunk qs+1, issue its corner loads and the offset loads of chunk qs+2,
  // multiply chunk qs, blend chunk qs+1, one barrier.  The loop body is unconditional (a conditional body made the compiler
  // merge the loop-carried offsets with register copies, i.e. wait for their loads before the MFMAs); the offsets past the
  // group's last chunk re-read that chunk's.  Every group executes per + 1 barriers (they are work-group wide).
  if (nst > 0) {
    dma_weights(0, q0);
    load_offsets(q0);
    issue_corners(q0);
    load_offsets(q0 + (nst > 1 ? 1 : 0));
    blend();
  }
  pp_wait_vmcnt<0>();
  pp_barrier();  // weights of the first chunk visible to the group
  for (int qs = 0; qs + 1 < nst; ++qs) {
    const int buf = qs & 1;
    // (the corner loads go first: hipcc drains the vector-memory queue, vmcnt(0), before it reads the loop-carried
    // offsets, and a weight copy issued ahead of that point would be waited for on the spot)
    issue_corners(q0 + qs + 1);          // o_* hold the offsets of chunk qs+1
    dma_weights(buf ^ 1, q0 + qs + 1);   // (stage of chunk qs-1: its readers passed the barrier of step qs-1)
    load_offsets(q0 + (qs + 2 < nst ? qs + 2 : nst - 1));
    pp_sched_fence();                    // every load of the step is in flight before the MFMAs start
    compute(buf);
    pp_sched_fence();
    blend();                             // B fragments of chunk qs+1
    pp_wait_vmcnt<0>();                  // this wave's part of the next weight tile has landed
    pp_barrier();
  }
  if (nst > 0) compute((nst - 1) & 1);
  if constexpr (KS > 1) {
    for (int i = nst > 0 ? nst : 1; i <= per; ++i) pp_barrier();
  }

  // ---- the KS partial tiles meet in LDS (the rings are free after the last barrier); group 0 adds them in group order,
  // conv_ksplit.hip's order: with the same chunk ranges per group the result equals pp_deform_cols + that kernel bit for bit
  if constexpr (KS > 1) {
    f4* red = reinterpret_cast<f4*>(PP_DYN_SMEM);
    pp_wait_lgkm0();
    if (grp > 0) {
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) red[(((grp - 1) * NW + wave) * (TC * TP) + a * TP + b) * 64 + lane] = acc[a][b];
    }
    pp_wait_lgkm0();
    pp_barrier();
    if (grp > 0) return;
#pragma unroll
    for (int g = 0; g < KS - 1; ++g)
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] += red[((g * NW + wave) * (TC * TP) + a * TP + b) * 64 + lane];
  }

  EpiCtx<OT> e;
  e.bias = p.bias;
  e.out = reinterpret_cast<OT*>(p.out);
  e.aux1 = reinterpret_cast<const OT*>(p.aux1);
  e.aux2 = reinterpret_cast<const OT*>(p.aux2);
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  // staging rows of the transposed epilogue: behind the partial tiles (the launcher sizes the LDS for them; the other groups
  // have left, so no barrier may follow)
  constexpr size_t RED_BYTES = (size_t)(KS - 1) * NW * TC * TP * 64 * sizeof(f4);
  epilogue_any<OT, TC, TP, true, KS == 1>(
      p, e, reinterpret_cast<unsigned char*>(PP_DYN_SMEM) + RED_BYTES, wave, lane, c_base,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        m = p_base + (wave * TP + decltype(bi)::value) * 16 + frow;
        ok = m < p.M;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA { return acc[decltype(ai)::value][decltype(bi)::value]; },
      [&](auto bi, int64_t& m0, int& nvalid) PP_INLINE_LAMBDA {
        m0 = p_base + (wave * TP + decltype(bi)::value) * 16;
        nvalid = (int)(p.M - m0 < 16 ? p.M - m0 : 16);
      });
}

template <typename OT, int NW, int KS, int TP>
static int launch_deform_cfg(void* stream, const DeformSrc& d, const ConvK& k) {
  constexpr int BP = NW * TP * 16;
  constexpr size_t ring = (size_t)KS * 2 * 128 * 32 * sizeof(half_t);
  constexpr size_t red = (size_t)(KS - 1) * NW * 8 * TP * 64 * sizeof(f4) + (size_t)NW * epi_lds_wave_bytes<8>();  // + the epilogue's staging rows
  constexpr size_t smem = ring > red ? ring : red;
  dim3 grid((unsigned)((k.M + BP - 1) / BP), (unsigned)((k.Cout + 127) / 128), 1u);
  PP_ALLOW_BIG_LDS((&deform_conv_kernel<OT, NW, KS, TP>), smem);
  PP_LAUNCH((deform_conv_kernel<OT, NW, KS, TP>), grid, dim3(KS * NW * 64), smem, stream, d, k);
  return pp_check_launch("pp_deform_conv");
}

template <typename OT>
static int launch_deform(void* stream, const DeformSrc& d, const ConvK& k) {
  // One form ships: 4 K groups of 2 waves on 32 pixels (8 waves, 225 work-groups for flow completion's 7200 pixels, 256
  // registers per lane; 4-wave groups would be 113 work-groups capped at 128 registers, which spills).  Measured on the
  // MI355X (profiles/r03_deform_fusion.md): 46.9 us against 54.3 us for pp_deform_cols + pp_conv2d at flow completion's
  // shape; at feature propagation's 8 x 90 x 160 pixels it is 378 us against 358 us (the flat forms -- 4 waves x 16 / 32
  // pixels, no K split: 410 / 489 us -- are not instantiated), so the host keeps the two launches there (ops.deform_fused).
  return launch_deform_cfg<OT, 2, 4, 1>(stream, d, k);
}

}  // namespace pp

extern "C" int32_t pp_deform_conv(void* stream, const pp_deform_cols_params* s, const pp_conv2d_params* g) {
  using namespace pp;
  if (!s || !s->x0 || !s->om) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_conv: null argument");
  if (s->dtype != PP_F16) return pp_fail(PP_ERR_UNSUPPORTED, "pp_deform_conv: f16 tensors only (fp32 storage: pp_deform_cols + pp_conv2d)");
  ConvK k;
  const int bad = convk_from_params(g, &k, "pp_deform_conv", true);
  if (bad != PP_OK) return bad;
  const int64_t Cin = s->x0_C + (s->x1 ? s->x1_C : 0);
  if (g->dtype != PP_F16 || g->Z != 1 || g->nseg != 1 || g->kh != 1 || g->kw != 1 || g->sh != 1 || g->sw != 1 || g->ph != 0 ||
      g->pw != 0 || g->in_C[0] != 9 * Cin || g->N != s->N || g->H != s->H || g->W != s->W || g->Ho != s->H || g->Wo != s->W)
    return pp_fail(PP_ERR_BAD_ARG, "pp_deform_conv: the convolution block must describe the 1x1 f16 convolution over the 9*Cin columns");
  DeformSrc d;
  d.x0 = s->x0; d.x0_C = (int)s->x0_C; d.x0_ldc = (int)s->x0_ldc;
  d.x1 = s->x1; d.x1_C = s->x1 ? (int)s->x1_C : 0; d.x1_ldc = (int)s->x1_ldc;
  d.om = reinterpret_cast<const float*>(s->om); d.om_ldc = (int)s->om_ldc;
  d.flow = reinterpret_cast<const float*>(s->flow); d.flow_ldc = (int)s->flow_ldc;
  d.H = (int)s->H; d.W = (int)s->W; d.dg = s->dg;
  if (d.dg < 1 || Cin % d.dg != 0) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_conv: channels not divisible by dg");
  d.cg = (int)(Cin / d.dg);
  if ((d.cg & 7) != 0 || (Cin & 31) != 0 || (d.x0_C & 31) != 0)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_deform_conv: needs 8-channel pieces inside one deformable group and whole 32-channel chunks per input");
  if ((d.x0_ldc & 7) != 0 || (d.x1 && (d.x1_ldc & 7) != 0) || (reinterpret_cast<uintptr_t>(d.x0) & 15) != 0 ||
      (d.x1 && (reinterpret_cast<uintptr_t>(d.x1) & 15) != 0))
    return pp_fail(PP_ERR_BAD_ARG, "pp_deform_conv: inputs must be 16-byte aligned with 16-byte pitches");
  d.cchunks = (int)(Cin / 32);
  d.xcd = options().deform_xcd && k.Cout <= 128;  // (grid.y == 1: blockIdx.x alone decides the XCD)
  return g->out_dtype == PP_F16 ? launch_deform<half_t>(stream, d, k) : launch_deform<float>(stream, d, k);
}

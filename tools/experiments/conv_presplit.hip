// conv_presplit.hip -- pp_conv2d in PP_F32X2 mode for 1x1 / stride-1 layers whose INPUT already carries the two-term
// operand split (pp_conv2d_params.in_presplit; r03).
//
// A 1x1 convolution has no taps over which to amortise the operand split of PP_F32X2: conv_split_kernel loads f32 pixels to
// registers and spends ~4 vector-ALU operations per value on h = f16_rtz(v), l = f16_rtz((v - h) * 2048) -- for RAFT's 324 ->
// 256 correlation projection (every GRU iteration) that arithmetic, not the three MFMA products, bounds the kernel (174 TF/s
// algorithmic where the 3x3 layers of the same family reach 380-410).  The producer of that tensor is ours (pp_corr_lookup;
// pp_im2col for the 7x7 flow stem), so it emits the split form directly: every 32-channel chunk of a pixel is 128 bytes
// [32 x h | 32 x l] -- the very layout of a split-packed weight chunk and of an LDS tile row of conv_split_kernel.  Here BOTH
// operands therefore travel global -> LDS by global_load_lds (no registers, no arithmetic on the way), the MFMA part is
// conv_split_kernel's (same three sweeps per chunk, same accumulation order: results are bit-identical to the in-kernel
// split of the same values).
//
// Work-group: WC x WP waves, tile (WC*TC*16) channels x (WP*TP*16) pixels, NS LDS stages of one 32-channel chunk each
// (rows of 128 bytes, 8 threads per row, slot s of row r holds source piece s ^ swz(r)); one barrier per chunk, counted
// vmcnt so that the copies of the next NS-1 chunks stay in flight across it.
#include "conv_common.h"

namespace pp {

template <typename OT, int WC, int WP, int TC, int TP, int NS>
__global__ void __launch_bounds__(WC * WP * 64, (WC * WP > 4 ? 1 : 2)) conv_presplit_kernel(const ConvK p) {
  constexpr int NT = WC * WP * 64;
  constexpr int RPP = NT / 8;               // tile rows covered by one pass of the work-group
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int XPASS = (BP + RPP - 1) / RPP, WPASS = (BC + RPP - 1) / RPP;
  constexpr int BPP = XPASS * RPP, BCP = WPASS * RPP;
  constexpr int ROWB = 128;
  constexpr int STAGE = (BPP + BCP) * ROWB;  // [pixels | weights]
  constexpr int NL = XPASS + WPASS;          // copies per thread and chunk
  constexpr float LINV = 1.f / 2048.f;
  static_assert(RPP % 16 == 0, "the slot swizzle depends on row mod 16");

  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
#ifdef PP_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wc = wave / WP, wp = wave % WP;
  const int64_t p_base = (int64_t)blockIdx.x * BP;
  const int c_base = (int)blockIdx.y * BC;

  auto swz = [](int r) PP_INLINE_LAMBDA { return ((r >> 1) & 7) ^ ((r & 1) << 2); };  // conv_split_kernel's
  const int pc = tid & 7, row0 = tid >> 3;
  const int pcs = pc ^ swz(row0);

  // source rows: a pixel's chunk q is 128 bytes at (m * ldc + 32 q) floats; weights likewise at (co * Kp + 32 q)
  const float* xsrc[XPASS];
  const float* wsrc[WPASS];
  {
    const float* xb = reinterpret_cast<const float*>(p.in_ptr[0]);
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int64_t m = p_base + row0 + i * RPP;
      xsrc[i] = xb + (m < p.M ? m : p.M - 1) * p.in_ldc[0] + pcs * 4;  // rows past M: clamped, results never stored
    }
    const float* wb = reinterpret_cast<const float*>(p.weight);
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      const int co = c_base + row0 + i * RPP;
      wsrc[i] = wb + (int64_t)(co < p.Cout ? co : p.Cout - 1) * p.Kp + pcs * 4;
    }
  }
  auto fetch = [&](int q, int stage) PP_INLINE_LAMBDA {
    unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) glds16(xsrc[i] + q * 32, st + (i * NT + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < WPASS; ++i) glds16(wsrc[i] + q * 32, st + BPP * ROWB + (i * NT + wave * 64) * 16);
  };

  f4 acc[TC][TP], accx[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
      accx[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    }
  const int frow = lane & 15, fgrp = lane >> 4;
  const int roff_h = (fgrp ^ swz(frow)) << 4;
  const int roff_l = ((fgrp + 4) ^ swz(frow)) << 4;
  auto compute = [&](int stage) PP_INLINE_LAMBDA {
    const unsigned char* xs = smem + stage * STAGE + (wp * TP * 16 + frow) * ROWB;
    const unsigned char* ws = smem + stage * STAGE + BPP * ROWB + (wc * TC * 16 + frow) * ROWB;
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
    // the three sweeps of conv_split_kernel, in its order
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

  const int n = p.nchunks;
  // chunks q+1 .. q+NS-1 are in flight while chunk q is multiplied; the stage of chunk q+NS-1 was read in step q-1
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < n) fetch(s, s);
  int stage = 0;
  for (int q = 0; q < n; ++q) {
    // copies still allowed in flight: those of the chunks after q that have been issued (at most NS-2 of them)
    const int later = (n - 1 - q) < (NS - 2) ? (n - 1 - q) : (NS - 2);
    if (NS >= 3 && later >= 1) pp_wait_vmcnt<(NS >= 3 ? NL : 0)>(); else pp_wait_vmcnt<0>();
    pp_barrier();
    int nxt = stage + NS - 1;
    if (nxt >= NS) nxt -= NS;
    if (q + NS - 1 < n) fetch(q + NS - 1, nxt);
    compute(stage);
    stage = stage + 1 == NS ? 0 : stage + 1;
  }

  EpiCtx<OT> e;
  e.bias = p.bias;
  e.out = reinterpret_cast<OT*>(p.out);
  e.aux1 = reinterpret_cast<const OT*>(p.aux1);
  e.aux2 = reinterpret_cast<const OT*>(p.aux2);
  e.pre = reinterpret_cast<const OT*>(p.pre_add);
  epilogue_quads<OT, TC, TP>(
      p, e,
      [&](auto bi, int64_t& m, bool& ok) PP_INLINE_LAMBDA {
        m = p_base + wp * TP * 16 + decltype(bi)::value * 16 + frow;
        ok = m < p.M;
      },
      [&](auto ai) PP_INLINE_LAMBDA { return c_base + wc * TC * 16 + decltype(ai)::value * 16 + fgrp * 4; },
      [&](auto ai, auto bi) PP_INLINE_LAMBDA {
        return acc[decltype(ai)::value][decltype(bi)::value] + accx[decltype(ai)::value][decltype(bi)::value] * LINV;
      });
}

template <typename OT, int WC, int WP, int TC, int TP, int NS>
static int launch_presplit_cfg(void* stream, const ConvK& k) {
  constexpr int NT = WC * WP * 64, RPP = NT / 8;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int BPP = (BP + RPP - 1) / RPP * RPP, BCP = (BC + RPP - 1) / RPP * RPP;
  const size_t smem = (size_t)NS * (BPP + BCP) * 128;
  dim3 grid((unsigned)((k.M + BP - 1) / BP), (unsigned)((k.Cout + BC - 1) / BC), 1);
  static const bool lds_ok = (pp_allow_big_lds(reinterpret_cast<const void*>(&conv_presplit_kernel<OT, WC, WP, TC, TP, NS>), smem), true);
  (void)lds_ok;
  PP_LAUNCH((conv_presplit_kernel<OT, WC, WP, TC, TP, NS>), grid, dim3(NT), smem, stream, k);
  return pp_check_launch("pp_conv2d");
}

// 1x1, stride 1, no padding, one input segment whose pitch and channel count are whole 32-channel chunks, one z slice
int launch_presplit(void* stream, const ConvK& k, int Z) {
  if (Z != 1 || k.nseg != 1 || k.kh != 1 || k.kw != 1 || k.sh != 1 || k.sw != 1 || k.ph != 0 || k.pw != 0)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: in_presplit needs a 1x1 stride-1 convolution over one segment");
  if ((k.in_C[0] & 31) != 0 || (k.in_ldc[0] & 31) != 0 || k.Ho != k.H || k.Wo != k.W)
    return pp_fail(PP_ERR_BAD_ARG, "pp_conv2d: in_presplit needs whole 32-channel chunks (128-byte [h | l] groups)");
  // 256 x 128 tiles, 8 waves, 3 stages (144 KB: one work-group per CU) when Cout fills them; 128 x 128, 2 stages otherwise
  if (k.Cout > 128 && (k.Cout + 255) / 256 * 256 - k.Cout <= 64) return launch_presplit_cfg<float, 4, 2, 4, 4, 3>(stream, k);
  return launch_presplit_cfg<float, 2, 2, 4, 4, 2>(stream, k);
}

}  // namespace pp

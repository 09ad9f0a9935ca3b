// conv_patch.hip -- pp_conv2d (PP_F32X2, flat_taps) for convolutions whose INPUT has at most 4 channels: RAFT's motion-encoder
// convolution on the 2-channel flow (7x7, 2 -> 128, every GRU iteration: update.py:100-106) and the 7x7 / stride-2 stems of
// its two encoders on the 3-channel frames (extractor.py:130-136).
//
// Until r05 these ran as pp_im2col + a 1x1 PP_F32X2 convolution over the patch matrix: at 158 x 45 x 80 pixels the patch matrix
// of the flow is 291 MB written and read back per iteration (224 us + 120 us, 6.9 ms per clip) for a layer whose real input is
// 4.5 MB.  Here the patch matrix only ever exists as MFMA operand fragments in LDS:
//   - a 256-thread work-group owns an 8 x 16 (128 channels) or 4 x 16 (64 channels) output tile of one image and ALL output channels;
//   - the input tile with its halo ((TH-1) s + kh rows x (16-1) s + kw columns x C floats, <= 6 KB) is read once;
//   - the threads build the B operand of v_mfma_f32_16x16x32_f16 -- lane l of a fragment holds the 8 consecutive
//     k = (ky, kx, c) of k-group l >> 4 for pixel l & 15 -- by gathering from that tile through a k -> tile-offset table,
//     split every value into its PP_F32X2 terms (split_pair: h = f16_rtz(v), l = f16_rtz(v - h)) and store the two fragments
//     (16 bytes per lane, fragment-major: every later read is one conflict-free ds_read_b128);
//   - 2 x 2 waves (channel half x pixel half) then run the three products per multiply-add of PP_F32X2 (conv_split.hip) into one
//     fp32 accumulator set; the weight fragments (PP_F32X2 packing, <= 80 KB per layer, L2 / L1 resident) come straight from
//     global memory, 16 bytes per lane;
//   - epilogue = store_quad / store_quad_fast of conv_common.h (bias, activation, any fused op), after ConvK::acc_scale.
// Weights are the im2col form the 1x1 convolution used: one (ky, kx, c)-ordered row per output channel, zero padded to a
// multiple of 32 at the END only (ops.make_conv_spec on the [Cout, kh*kw*C, 1, 1] view) -- the `flat_taps` contract of
// conv_gemm_f16.hip's patch-gathering form, here for f32 tensors with C <= 4.
#include "conv_common.h"

namespace pp {

constexpr int kPatchTW = 16;
#ifdef PP_EMU
#define PP_TWO_WAVES_PER_SIMD
#else
#define PP_TWO_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(2)))   // at most 256 registers per lane: two work-groups per CU
#endif
constexpr int kPatchMaxHaloPerThread = 6;   // (4 - 1) * 2 + 7 = 13 rows x (16 - 1) * 2 + 7 = 37 columns x 3 floats = 1443 <= 6 * 256

// 4 waves = WCN channel groups x 4 / WCN pixel-row groups; CB: 16-channel blocks per wave (Cout = WCN * CB * 16); TH: tile rows
// (TH x 16 output pixels), PB = TH * WCN / 4 rows per wave; NCK: 32-element chunks of the (ky, kx, c) patch vector.
//
// PERSISTENT work-groups (r06, second form): the first form launched one work-group per tile and measured 269 us for the 7x7 on the
// flow (im2col + 1x1: 329): each tile paid the L2 latency of its weight fragments chunk by chunk (64 KB of weights do not stay in a
// 32 KB L1) and of its input tile with nothing to overlap them with.  Here a work-group keeps ALL its weight fragments in registers
// for its whole life (NCK x CB x 2 fragments per lane), walks a contiguous range of tiles, and requests the next tile's input while
// the matrix pipe works on the current one.
template <int WCN, int CB, int TH, int NCK>
__global__ void __launch_bounds__(256) PP_TWO_WAVES_PER_SIMD conv_patch_split_kernel(const ConvK p, const int tiles_x, const int tiles_y, const int ntiles) {
  constexpr int PB = TH * WCN / 4;
  constexpr int NQ = kPatchMaxHaloPerThread;
  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.in_C[0], ldc = p.in_ldc[0];
  const int HH = (TH - 1) * p.sh + p.kh, HWd = (kPatchTW - 1) * p.sw + p.kw;
  const int nhalo = HH * HWd * C;
  const int Kv = p.kh * p.kw * C;
  float* halo = reinterpret_cast<float*>(smem);
  int* koff = reinterpret_cast<int*>(smem + (size_t)((nhalo * 4 + 15) & ~15));
  unsigned char* frag = reinterpret_cast<unsigned char*>(koff) + (size_t)NCK * 32 * 4;

  // this work-group's tiles: a contiguous range, the ranges of one XCD's work-groups adjacent (neighbouring tiles share halo
  // columns / rows: they come out of that XCD's L2)
  const int L = p.tile_order ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = L * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  if (t_begin >= t_end) return;

  // channel group wc, pixel rows PB wp .. PB wp + PB - 1
  const int wc = wave % WCN, wp = wave / WCN;
  const int frow = lane & 15, fgrp = lane >> 4;

  // ---- every weight fragment this wave will ever need: chunk kc of 16-channel block a = 32 h | 32 l f16 at byte kc * 128
  h8 ah[NCK][CB], al[NCK][CB];
  {
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.weight);
#pragma unroll
    for (int a = 0; a < CB; ++a) {
      int row = (wc * CB + a) * 16 + frow;
      row = row < p.Cout ? row : p.Cout - 1;
      const unsigned char* w = wbase + (int64_t)row * p.Kp * 4 + fgrp * 16;
#pragma unroll
      for (int kc = 0; kc < NCK; ++kc) {
        ah[kc][a] = *reinterpret_cast<const h8*>(w + kc * 128);
        al[kc][a] = *reinterpret_cast<const h8*>(w + kc * 128 + 64);
      }
    }
  }

  // ---- the k -> tile-offset table, and this thread's elements of an input tile (the same positions for every tile)
  for (int k = tid; k < NCK * 32; k += 256) {
    const int tap = k / C, c = k - tap * C;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    koff[k] = k < Kv ? (ky * HWd + kx) * C + c : -1;
  }
  int hpos[NQ];      // hy << 16 | hx << 4 | c   (-1: past the tile)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + q * 256;
    const int c = i % C, r = i / C;
    const int hy = r / HWd, hx = r - hy * HWd;
    hpos[q] = i < nhalo ? (hy << 16 | hx << 4 | c) : -1;
  }
  const float* in = reinterpret_cast<const float*>(p.in_ptr[0]);
  float hv[NQ];
  auto load_halo = [&](int t) PP_INLINE_LAMBDA {
    const int txi = t % tiles_x, tyi = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
    const int iy0 = tyi * TH * p.sh - p.ph, ix0 = txi * kPatchTW * p.sw - p.pw;
    const float* img = in + (int64_t)n * p.H * p.W * ldc;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int hp = hpos[q];
      const int iy = iy0 + (hp >> 16), ix = ix0 + ((hp >> 4) & 0xfff);
      float v = 0.f;
      if (hp >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = img[((int64_t)iy * p.W + ix) * ldc + (hp & 15)];
      hv[q] = v;
    }
  };

  // the bias quads of this lane's channels, read ONCE: stores count in vmcnt, so a bias load behind the previous quad's store
  // would wait for that store's round trip -- 16 serialised round trips per tile (r05 found the same in the halo kernels' epilogue)
  f4 bq[CB];
#pragma unroll
  for (int a = 0; a < CB; ++a) {
    const int c = (wc * CB + a) * 16 + fgrp * 4;
    bq[a] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (p.bias && c + r < p.Cout) bq[a][r] = p.bias[c + r];
  }
  EpiCtx<float> e;
  e.bias = nullptr;          // (added below, before store_quad's own steps: the same order of operations)
  e.out = reinterpret_cast<float*>(p.out);
  e.aux1 = reinterpret_cast<const float*>(p.aux1);
  e.aux2 = reinterpret_cast<const float*>(p.aux2);
  e.pre = reinterpret_cast<const float*>(p.pre_add);
  const bool fast = epi_fast_ok(p, e);
  const int gbase = (frow * p.sw) * C, growstep = p.sh * HWd * C;

  load_halo(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    // ---- input tile -> LDS (the previous tile's fragment readers are past their loop: the barrier below orders them)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (hpos[q] >= 0) halo[tid + q * 256] = hv[q];
    __syncthreads();

    // ---- B-operand fragments of the tile: item j = (tile row pb, chunk kc), one fragment lane per thread lane
    for (int j = wave; j < TH * NCK; j += 4) {
      const int kc = j % NCK, pb = j / NCK;
      const int base = gbase + pb * growstep;
      const int* ko = koff + kc * 32 + 8 * fgrp;
      const u4 o0 = *reinterpret_cast<const u4*>(ko), o1 = *reinterpret_cast<const u4*>(ko + 4);
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o = (int)(i < 4 ? o0[i & 3] : o1[i & 3]);
        const float x = halo[base + (o >= 0 ? o : 0)];
        v[i] = o >= 0 ? x : 0.f;
      }
      h8 hh, ll;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h2 a, b;
        split_pair(v[2 * i], v[2 * i + 1], a, b);
        hh[2 * i] = a[0];
        hh[2 * i + 1] = a[1];
        ll[2 * i] = b[0];
        ll[2 * i + 1] = b[1];
      }
      unsigned char* dst = frag + ((size_t)(pb * NCK + kc) * 2) * 1024 + lane * 16;
      *reinterpret_cast<h8*>(dst) = hh;
      *reinterpret_cast<h8*>(dst + 1024) = ll;
    }
    __syncthreads();
    if (t + 1 < t_end) load_halo(t + 1);   // in flight under this tile's matrix work and stores

    f4 acc[CB][PB];
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
      for (int b = 0; b < PB; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    // (fragments of ONE pixel block at a time: PB x 2 fragment registers at once cost the 8-row tile its second wave per SIMD)
    static_for<NCK>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      static_for<PB>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        const unsigned char* f = frag + ((size_t)((wp * PB + b) * NCK + kc) * 2) * 1024 + lane * 16;
        const h8 bh = *reinterpret_cast<const h8*>(f);
        const h8 bl = *reinterpret_cast<const h8*>(f + 1024);
#pragma unroll
        for (int a = 0; a < CB; ++a) acc[a][b] = mfma_16x16x32_f16(ah[kc][a], bl, acc[a][b]);
#pragma unroll
        for (int a = 0; a < CB; ++a) acc[a][b] = mfma_16x16x32_f16(al[kc][a], bh, acc[a][b]);
#pragma unroll
        for (int a = 0; a < CB; ++a) acc[a][b] = mfma_16x16x32_f16(ah[kc][a], bh, acc[a][b]);
      });
    });

    // ---- epilogue: lane holds 4 consecutive channels (4 fgrp .. + 3 of its block) of pixel (row PB wp + b, column frow)
    const int txi = t % tiles_x, tyi = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
    const int ox = txi * kPatchTW + frow;
    static_for<PB>([&](auto bi) {
      constexpr int b = decltype(bi)::value;
      const int oy = tyi * TH + wp * PB + b;
      if (oy < p.Ho && ox < p.Wo) {
        const int64_t m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
        static_for<CB>([&](auto ai) {
          constexpr int a = decltype(ai)::value;
          const int c = (wc * CB + a) * 16 + fgrp * 4;
          if (c < p.Cout) {
            const f4 v = acc[a][b] * p.acc_scale + bq[a];
            if (fast) store_quad_fast(p, e, v, m, c); else store_quad(p, e, v, m, c);
          }
        });
      }
    });
  }
}

// PP_F32X2 + flat_taps: f32 input of at most 4 channels, one segment, zero padding, dilation 1, Cout 64 or 128, at most 5 chunks
// of (ky, kx, c) (160 patch elements: 7 x 7 x 3 = 147).  Anything else is refused (the caller keeps pp_im2col + 1x1 for it).
template <int WCN, int CB, int TH, int NCK>
static int launch_patch_cfg(void* stream, const ConvK& k) {
  const int tiles_x = (k.Wo + kPatchTW - 1) / kPatchTW, tiles_y = (k.Ho + TH - 1) / TH;
  const int64_t ntiles = (int64_t)k.N * tiles_x * tiles_y;
  if (ntiles >= ((int64_t)1 << 31)) return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: too many tiles");
  const int HH = (TH - 1) * k.sh + k.kh, HWd = (kPatchTW - 1) * k.sw + k.kw;
  const int nhalo = HH * HWd * k.in_C[0];
  if (nhalo > kPatchMaxHaloPerThread * 256 || HWd > 0xfff)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: PP_F32X2 flat_taps input tile too large (taps x stride)");
  const size_t smem = (size_t)((nhalo * 4 + 15) & ~15) + (size_t)NCK * 32 * 4 + (size_t)TH * NCK * 2 * 1024;
  // persistent work-groups: as many as the chip holds (256 CUs x the work-groups per CU the LDS / register budget allows),
  // each with a contiguous range of tiles
  const int per_cu = smem > 75 * 1024 ? 1 : 2;
  int64_t nwg = 256 * per_cu;
  if (nwg > ntiles) nwg = ntiles;
  dim3 grid((unsigned)nwg), block(256);
  PP_ALLOW_BIG_LDS((&conv_patch_split_kernel<WCN, CB, TH, NCK>), 150 * 1024);
  PP_LAUNCH((conv_patch_split_kernel<WCN, CB, TH, NCK>), grid, block, smem, stream, k, tiles_x, tiles_y, (int)ntiles);
  return pp_check_launch("pp_conv2d");
}

int launch_patch_split(void* stream, const ConvK& k, int Z) {
  if (Z != 1 || k.nseg != 1 || k.in_C[0] < 1 || k.in_C[0] > 4 || k.dh != 1 || k.dw != 1 || k.pad_mode != PP_PAD_ZEROS)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: PP_F32X2 flat_taps needs one zero-padded input of 1..4 channels, dilation 1");
  if ((k.Cout != 64 && k.Cout != 128) || k.nchunks > 5)
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: PP_F32X2 flat_taps serves Cout 64 / 128 and kh*kw*C <= 160");
  // 128 channels: 8 x 16 pixel tiles, the four waves each own 32 channels of ALL its pixels (64 accumulator + 16 NCK weight
  // registers per lane: two work-groups per CU; channel half x pixel half would need 32 NCK weight registers = one work-group);
  // 64 channels: 4 x 16 tiles, channel half x pixel half (with 5 chunks an 8-row tile's fragments are 80 KB = one work-group per CU)
#define PP_PATCH_NCK(WCN_, CB_, TH_)                                                            \
  switch (k.nchunks) {                                                                          \
    case 1: return launch_patch_cfg<WCN_, CB_, TH_, 1>(stream, k);                              \
    case 2: return launch_patch_cfg<WCN_, CB_, TH_, 2>(stream, k);                              \
    case 3: return launch_patch_cfg<WCN_, CB_, TH_, 3>(stream, k);                              \
    case 4: return launch_patch_cfg<WCN_, CB_, TH_, 4>(stream, k);                              \
    default: return launch_patch_cfg<WCN_, CB_, TH_, 5>(stream, k);                             \
  }
  if (k.Cout == 128) {
    PP_PATCH_NCK(4, 2, 8)
  }
  PP_PATCH_NCK(2, 2, 4)
#undef PP_PATCH_NCK
}

}  // namespace pp

// corr_lookup_conv.hip -- pp_corr_lookup_conv: RAFT's correlation lookup FUSED with the motion encoder's first convolution
//   cor = relu(convc1(corr_lookup(pyramid, coords)))        CorrBlock.__call__ (model/modules/RAFT/corr.py:29-50) feeding
//                                                           BasicMotionEncoder.convc1 (update.py:94-112), every GRU iteration.
//
// r01-r05 ran two launches: pp_corr_lookup wrote the 324 sampled correlations of every pixel (737 MB per iteration at cfg 2,
// 553-580 us) and the 1x1 PP_F32X2 convolution 324 -> 256 read them back (397-415 us, the slowest PP_F32X2 layer: 224 TF/s).
// Here the sampled vector never leaves the CU: it is exactly the B operand of the projection's MFMAs.
//   * a 256-thread work-group owns a tile of 32 consecutive pixels (of the flattened pair x h x w space) and all 256 output
//     channels; work-groups are persistent and walk a contiguous range of tiles;
//   * phase A (the lookup, vector-ALU bound): each wave evaluates 8 pixels, one after the other, with pp_corr_lookup's own
//     steps -- the 12 x 16 window of every pyramid level staged into a wave-private LDS region (the loads of pixel i + 1 are in
//     flight while pixel i is evaluated), the 36-entry coordinate table, the bilinear blends operation by operation -- and
//     writes every value, split into its PP_F32X2 terms, into the tile's B-operand fragments in LDS
//     ([16-pixel block][chunk of 32 channels][h | l][64 lanes x 16 bytes], channels 324..351 stay zero);
//   * phase B (the projection, matrix-pipe bound): wave w owns output channels 64 w .. 64 w + 63 of all 32 pixels; weight
//     fragments (PP_F32X2 packing: 32 h | 32 l f16 per chunk) stream from L2 one chunk ahead of their use, B fragments are
//     conflict-free ds_read_b128; three products per multiply-add into one fp32 accumulator set (conv_split.hip's arithmetic);
//     epilogue = acc_scale, bias, activation, 16-byte channels-last stores (store_quad_fast / store_quad).
// Two work-groups share a CU (60 KB of LDS each), so one's lookup overlaps the other's matrix work.
#include "conv_common.h"
#include "corr_lookup.h"

namespace pp {

constexpr int kLcNCK = 11;                    // 324 lookup channels -> 352 = 11 chunks of 32
constexpr int kLcPB = 2;                      // 16-pixel blocks per tile
constexpr int kLcPT = kLcPB * 16;             // pixels per tile
constexpr int kLcPix = kLcPT / 4;             // pixels per wave and tile
constexpr int kLcFragBytes = kLcPB * kLcNCK * 2 * 1024;
constexpr int kLcWinBytes = 4 * kCorrLvl * 4; // one wave's windows
constexpr int kLcTabBytes = 2 * 36 * 8;       // one wave's coordinate table

#ifdef PP_EMU
#define PP_LC_OCCUPANCY
#else
#define PP_LC_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2)))   // at most 256 registers per lane: two work-groups per CU
#endif

__global__ void __launch_bounds__(256) PP_LC_OCCUPANCY corr_lookup_conv_kernel(const LookupK k, const ConvK p, const int ntiles) {
  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* frag = smem;
  float* mywin = reinterpret_cast<float*>(smem + kLcFragBytes + wave * kLcWinBytes);
  CoordEntry* mytab = reinterpret_cast<CoordEntry*>(smem + kLcFragBytes + 4 * kLcWinBytes + wave * kLcTabBytes);   // [2][36]

  const int L = p.tile_order ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = L * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  if (t_begin >= t_end) return;
  const int64_t npix = p.M;
  const int hw = k.h * k.w;
  const int frow = lane & 15, fgrp = lane >> 4;

  // ---- once per work-group: the fragment buffer starts as zeros (the lookup writes channels 0..323 of every pixel of every tile;
  //      channels 324..351 of the last chunk are never written and must multiply as zeros)
  for (int i = tid; i < kLcFragBytes / 16; i += 256) reinterpret_cast<u4*>(frag)[i] = u4{0u, 0u, 0u, 0u};

  // this wave's weight rows (output channels 64 wave + 16 a + frow): chunk kc = 32 h | 32 l f16 at byte kc * 128
  const unsigned char* wrow[4];
  f4 bq[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int row = (wave * 4 + a) * 16 + frow;
    row = row < p.Cout ? row : p.Cout - 1;
    wrow[a] = reinterpret_cast<const unsigned char*>(p.weight) + (int64_t)row * p.Kp * 4 + fgrp * 16;
    const int c = (wave * 4 + a) * 16 + fgrp * 4;
    bq[a] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (p.bias && c + r < p.Cout) bq[a][r] = p.bias[c + r];   // read once: a bias load behind a store would wait for the store
  }
  EpiCtx<float> e;
  e.bias = nullptr;   // (added by hand below, before store_quad's own steps: the same order of operations)
  e.out = reinterpret_cast<float*>(p.out);
  e.aux1 = reinterpret_cast<const float*>(p.aux1);
  e.aux2 = reinterpret_cast<const float*>(p.aux2);
  e.pre = reinterpret_cast<const float*>(p.pre_add);
  const bool fast = epi_fast_ok(p, e);

  // ---- per-pixel lookup state: the window loads of ONE pixel in flight in registers
  f4 wv[4];
  float ws[4][3];
  int ox[4], oy[4], spx = 0, spy = 0;
  float sfx = 0.f, sfy = 0.f;
  bool sact = false;

  // the flow of this wave's 8 pixels of a tile: lane 2 i + c holds component c of pixel i
  auto load_flows = [&](int t) PP_INLINE_LAMBDA -> float {
    const int64_t pix = (int64_t)t * kLcPT + wave * kLcPix + (lane >> 1);
    return (lane < 2 * kLcPix && pix < npix) ? k.flow[pix * k.flow_ldc + (lane & 1)] : 0.f;
  };
  // request the windows of pixel `pix` (corr_lookup_kernel's staging loads, into registers)
  auto stage = [&](int64_t pix, float fx, float fy) PP_INLINE_LAMBDA {
    sact = pix < npix;
    const int64_t pc = sact ? pix : 0;
    const int pp_ = (int)(pc % hw);
    spy = pp_ / k.w;
    spx = pp_ - spy * k.w;
    sfx = fx;
    sfy = fy;
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {
      const float scale = 1.f / (float)(1 << lvl);
      float bx = ((float)spx + fx) * scale, by = ((float)spy + fy) * scale;
      if (!(fabsf(bx) < 1.0e6f)) bx = -1.0e6f;  // NaN / Inf / absurd flow: a window far outside the plane (all zeros)
      if (!(fabsf(by) < 1.0e6f)) by = -1.0e6f;
      ox[lvl] = ((int)floorf(bx) - 5) & ~3;
      oy[lvl] = (int)floorf(by) - 5;
      const int H = k.ph[lvl], W = k.pw[lvl];
      const int tl = k.tiled[lvl];
      const int64_t pitch = plane_pitch(H, W, tl);
      const float* plane = k.pyr[lvl] + pc * pitch;
      wv[lvl] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 3; ++it) ws[lvl][it] = 0.f;
      if (tl || ((W & 3) == 0 && (pitch & 3) == 0)) {
        const int Hv = tl ? ((H + 3) & ~3) : H, Wv = tl ? ((W + 7) & ~7) : W;
        if (lane < kCorrRows * 4) {
          const int r = lane >> 2, q = lane & 3;
          const int y = oy[lvl] + r, x0 = ox[lvl] + 4 * q;
          if (sact && (unsigned)y < (unsigned)Hv && (unsigned)x0 < (unsigned)Wv)
            wv[lvl] = *reinterpret_cast<const f4*>(plane + plane_off(y, x0, W, tl));
        }
      } else {
#pragma unroll
        for (int it = 0; it < kCorrLvl / 64; ++it) {
          const int idx = lane + it * 64;
          const int r = idx / kCorrCols, c = idx - r * kCorrCols;
          const int y = oy[lvl] + r, x = ox[lvl] + c;
          if (sact && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ws[lvl][it] = plane[(int64_t)y * W + x];
        }
      }
    }
  };
  // windows -> the wave's LDS region, then the coordinate table of that pixel (corr_lookup_kernel, operation by operation)
  auto commit = [&]() PP_INLINE_LAMBDA {
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {
      const int H = k.ph[lvl], W = k.pw[lvl];
      const int tl = k.tiled[lvl];
      float* lw = mywin + lvl * kCorrLvl;
      if (tl || ((W & 3) == 0 && (plane_pitch(H, W, tl) & 3) == 0)) {
        if (lane < kCorrRows * 4) *reinterpret_cast<f4*>(lw + (lane >> 2) * kCorrCols + 4 * (lane & 3)) = wv[lvl];
      } else {
#pragma unroll
        for (int it = 0; it < kCorrLvl / 64; ++it) lw[lane + it * 64] = ws[lvl][it];
      }
    }
    if (lane < 36) {
      const int lvl = lane / 9, o = lane - lvl * 9;
      const float scale = 1.f / (float)(1 << lvl);
      int olx = 0, oly = 0, H = 1, W = 1;
#pragma unroll
      for (int l = 0; l < 4; ++l)
        if (lvl == l) {
          olx = ox[l];
          oly = oy[l];
          H = k.ph[l];
          W = k.pw[l];
        }
      float cx = ((float)spx + sfx) * scale + (float)(o - 4);
      float cy = ((float)spy + sfy) * scale + (float)(o - 4);
      if (!(fabsf(cx) < 1.0e8f)) cx = -1.0e8f;
      if (!(fabsf(cy) < 1.0e8f)) cy = -1.0e8f;
      const float xn = 2.f * cx / (float)(W - 1) - 1.f;
      const float yn = 2.f * cy / (float)(H - 1) - 1.f;
      const float ix = ((xn + 1.f) / 2.f) * (float)(W - 1);
      const float iy = ((yn + 1.f) / 2.f) * (float)(H - 1);
      const float flx = floorf(ix), fly = floorf(iy);
      mytab[lane] = CoordEntry{(int)flx - olx, ix - flx};
      mytab[36 + lane] = CoordEntry{(int)fly - oly, iy - fly};
    }
    pp_wave_lds_fence();
  };
  // the 324 samples of the committed pixel -> its column of the tile's B fragments (pt = pixel index inside the tile)
  auto evaluate = [&](int pt) PP_INLINE_LAMBDA {
    unsigned char* col = frag + (size_t)(pt >> 4) * (kLcNCK * 2 * 1024) + (pt & 15) * 16;
#pragma unroll
    for (int it = 0; it < (324 + 63) / 64; ++it) {
      const int ch = lane + it * 64;
      if (ch < 324) {
        const int lvl = ch / 81;
        const int r = ch - lvl * 81;
        const int i = r / 9, j = r - i * 9;
        const CoordEntry ex = mytab[lvl * 9 + i], ey = mytab[36 + lvl * 9 + j];
        const int lx = ex.corner, ly = ey.corner;
        const float ax = ex.frac, ay = ey.frac;
        float v = 0.f;
        if ((unsigned)lx < (unsigned)(kCorrCols - 1) && (unsigned)ly < (unsigned)(kCorrRows - 1)) {
          const float* w0 = mywin + lvl * kCorrLvl + ly * kCorrCols + lx;
          v += w0[0] * (1.f - ax) * (1.f - ay);
          v += w0[1] * ax * (1.f - ay);
          v += w0[kCorrCols] * (1.f - ax) * ay;
          v += w0[kCorrCols + 1] * ax * ay;
        }
        h2 hh, ll;
        split_pair(v, 0.f, hh, ll);
        // channel ch = chunk ch >> 5, k-group (ch >> 3) & 3, element ch & 7 of the fragment lane (k-group, pixel)
        unsigned char* dst = col + (size_t)(ch >> 5) * 2048 + ((ch >> 3) & 3) * 256 + (ch & 7) * 2;
        *reinterpret_cast<half_t*>(dst) = hh[0];
        *reinterpret_cast<half_t*>(dst + 1024) = ll[0];
      }
    }
    pp_wave_lds_fence();   // the next pixel's commit() overwrites the windows / the table these lanes have just read
  };

  float fl = load_flows(t_begin);
  __syncthreads();   // the zeroed fragment buffer is visible to every wave
  for (int t = t_begin; t < t_end; ++t) {
    // ---------------------------------------------------------------- phase A: this wave's 8 pixels of the tile
    const int64_t pix0 = (int64_t)t * kLcPT + wave * kLcPix;
    stage(pix0, shfl_idx(fl, 0), shfl_idx(fl, 1));
#pragma unroll
    for (int i = 0; i < kLcPix; ++i) {
      commit();
      if (i + 1 < kLcPix) stage(pix0 + i + 1, shfl_idx(fl, 2 * i + 2), shfl_idx(fl, 2 * i + 3));   // in flight under evaluate(i)
      evaluate(wave * kLcPix + i);
    }
    if (t + 1 < t_end) fl = load_flows(t + 1);   // in flight under phase B
    // the first weight chunk is requested before the barrier: it travels while the slower waves finish their pixels
    h8 ah[2][4], al[2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      ah[0][a] = *reinterpret_cast<const h8*>(wrow[a]);
      al[0][a] = *reinterpret_cast<const h8*>(wrow[a] + 64);
    }
    pp_barrier();

    // ---------------------------------------------------------------- phase B: 64 channels x 32 pixels per wave
    f4 acc[4][kLcPB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < kLcPB; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<kLcNCK>([&](auto kci) {
      constexpr int kc = decltype(kci)::value;
      constexpr int S = kc & 1;
      if constexpr (kc + 1 < kLcNCK) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          ah[1 - S][a] = *reinterpret_cast<const h8*>(wrow[a] + (kc + 1) * 128);
          al[1 - S][a] = *reinterpret_cast<const h8*>(wrow[a] + (kc + 1) * 128 + 64);
        }
      }
      static_for<kLcPB>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        const unsigned char* f = frag + ((size_t)(b * kLcNCK + kc) * 2) * 1024 + lane * 16;
        const h8 bh = *reinterpret_cast<const h8*>(f);
        const h8 bl = *reinterpret_cast<const h8*>(f + 1024);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma_16x16x32_f16(ah[S][a], bl, acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma_16x16x32_f16(al[S][a], bh, acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma_16x16x32_f16(ah[S][a], bh, acc[a][b]);
      });
    });
    pp_barrier();   // every wave has read the tile's fragments: the next tile's lookup may overwrite them

    // epilogue (under the other waves' / the other work-group's next phase): lane = 4 channels of pixel (block b, column frow)
    static_for<kLcPB>([&](auto bi) {
      constexpr int b = decltype(bi)::value;
      const int64_t m = (int64_t)t * kLcPT + b * 16 + frow;
      if (m < npix) {
        static_for<4>([&](auto ai) {
          constexpr int a = decltype(ai)::value;
          const int c = (wave * 4 + a) * 16 + fgrp * 4;
          if (c < p.Cout) {
            const f4 v = acc[a][b] * p.acc_scale + bq[a];
            if (fast) store_quad_fast(p, e, v, m, c); else store_quad(p, e, v, m, c);
          }
        });
      }
    });
  }
}

}  // namespace pp

extern "C" int32_t pp_corr_lookup_conv(void* stream, const pp_corr_lookup_params* s, const pp_conv2d_params* g) {
  using namespace pp;
  if (!s || !s->flow) return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup_conv: null argument");
  ConvK c;
  const int bad = convk_from_params(g, &c, "pp_corr_lookup_conv", true);
  if (bad != PP_OK) return bad;
  if (g->dtype != PP_F32X2 || g->out_dtype != PP_F32 || g->Z != 1 || g->nseg != 1 || g->in_C[0] != 324 || g->kh != 1 || g->kw != 1 ||
      g->sh != 1 || g->sw != 1 || g->ph != 0 || g->pw != 0 || g->Cout != 256 || g->N != s->N || g->H != s->h || g->W != s->w ||
      g->Ho != s->h || g->Wo != s->w || g->flat_taps)
    return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup_conv: the convolution block must describe the PP_F32X2 1x1 convolution 324 -> 256 over the lookup's pixels");
  LookupK k;
  for (int l = 0; l < 4; ++l) {
    if (!s->pyr[l] || s->ph[l] < 2 || s->pw[l] < 2)
      return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup_conv: every pyramid level needs >= 2 rows and columns (H,W >= 128)");
    k.pyr[l] = (const float*)s->pyr[l];
    k.ph[l] = (int)s->ph[l];
    k.pw[l] = (int)s->pw[l];
    k.tiled[l] = (int)s->tiled[l];
  }
  k.flow = (const float*)s->flow;
  k.flow_ldc = (int)s->flow_ldc;
  k.out = nullptr;
  k.out_ldc = 0;
  k.h = (int)s->h;
  k.w = (int)s->w;
  k.total = s->N * s->h * s->w * 324;
  if (c.M <= 0 || c.M >= ((int64_t)1 << 36)) return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup_conv: empty or oversized problem");
  const int64_t ntiles = (c.M + kLcPT - 1) / kLcPT;
  const size_t smem = (size_t)kLcFragBytes + 4 * (size_t)kLcWinBytes + 4 * (size_t)kLcTabBytes;
  int64_t nwg = 512;   // persistent: 256 CUs x 2 work-groups (60 KB of LDS, <= 256 registers per lane)
  if (nwg > ntiles) nwg = ntiles;
  PP_ALLOW_BIG_LDS((&corr_lookup_conv_kernel), smem);
  PP_LAUNCH(corr_lookup_conv_kernel, dim3((unsigned)nwg), dim3(256), smem, stream, k, c, (int)ntiles);
  return pp_check_launch("pp_corr_lookup_conv");
}

// conv_direct.hip -- pp_conv2d for convolutions with at most 4 output channels (RAFT flow head 256 -> 2 at every GRU
// iteration, the generator's output convolution 64 -> 3, the flow-completion output convolution 32 -> 2).
//
// On the implicit-GEMM kernels such a layer occupies a 32- or 64-channel MFMA tile for 2-3 real channels: the 256 -> 2
// flow head ran at 9 algorithmic TF/s, 0.56 ms per launch, 1.0 TB/s of input -- an HBM-bound layer (582 MB of input,
// 4.5 MB of output) held at a fifth of the HBM rate by wasted matrix work.  Here it is what it is, a streaming
// reduction on the vector ALU (fp32 FMAs: PP_F32X2's three-product f16 arithmetic is only an fp32 stand-in):
//   - a 256-thread work-group owns a 16 x 16 output tile, one output pixel per thread, COUT fp32 accumulators;
//   - per 32-channel chunk the (16 + kh - 1) x (16 + kw - 1) input tile is read once with coalesced 16-byte loads (into
//     registers, one chunk ahead of the FMAs) and stored into LDS (pixel pitch = chunk bytes + 16, an odd number of
//     16-byte slots: a wave's ds_read_b128 of 64 different pixels is conflict free up to the tile-row wrap), every tap
//     then reads its channels from there;
//   - the weights are decoded ONCE per work-group into LDS as fp32 (from the packed f16 / PP_F32X2 / f32 layouts of
//     ops.pack_conv_weight) and read as wave-uniform (broadcast) ds_read_b128.
// Epilogue = store_quad's arithmetic per scalar (bias, pre_add, activation split, scale, fused op), any output view.
#include "conv_common.h"

namespace pp {

constexpr int kDirT = 16;  // output tile edge

template <typename T, typename OT, int COUT, bool WG>
__global__ void __launch_bounds__(256) conv_small_cout_kernel(const ConvK p, const int tiles_x, const int tiles_y, const int wmode) {
  constexpr int EPP = 16 / (int)sizeof(T);       // elements per 16-byte piece
  constexpr int PP_ = 32 / EPP;                  // pieces per pixel and 32-channel chunk (8 fp32, 4 f16)
  constexpr int PITCH = 32 * (int)sizeof(T) + 16;
  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);
  const int tid = (int)threadIdx.x;
  const int ntaps = p.kh * p.kw;
  const int nck = p.chunks_per_tap;              // one segment: chunks of the input
  const int hw = kDirT + p.kw - 1, hrows = (kDirT + p.kh - 1) * hw;
  // weights as fp32 [tap][chunk][COUT][32]: WG = the caller's table in global memory (wave-uniform addresses: hipcc reads
  // them with scalar loads, they never touch LDS), else decoded here into LDS
  float* wl = reinterpret_cast<float*>(smem);
  unsigned char* xt = smem + (WG ? (size_t)0 : (size_t)ntaps * nck * COUT * 32 * sizeof(float));
  const float* __restrict__ wg = p.weight_f32;

  // (16 x 16 tiles in row-major order; XCD-contiguous like the flat tiles: a tile's 2-pixel halo ring is then read by
  // neighbours on the same XCD instead of being fetched into up to four L2s)
  const int bid = p.tile_order ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int txi = bid % tiles_x, tyi = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int ty0 = tyi * kDirT, tx0 = txi * kDirT;

  // ---- weights -> fp32 in LDS (wmode: the packing of p.weight, PP_F32 / PP_F16 / PP_F32X2)
  for (int i = tid; !WG && i < ntaps * nck * COUT * 32; i += 256) {
    const int j = i & 31, co = (i >> 5) % COUT, tk = (i >> 5) / COUT;  // tk = tap * nck + chunk
    float w;
    if (co >= p.Cout) {
      w = 0.f;                                   // (COUT is 2 or 4: a 3-channel layer has no fourth weight row)
    } else if (wmode == PP_F32X2) {
      const half_t* row = reinterpret_cast<const half_t*>(reinterpret_cast<const float*>(p.weight) + (int64_t)co * p.Kp + tk * 32);
      // ABI v9 packing: h = f16(S w), l = f16(S w - h) UNSCALED, S = 1 / acc_scale a power of two per layer (this
      // decoder still read the r04 form, l x 2^-11 and no acc_scale, so a table-less PP_F32X2 launch returned S x the result)
      w = ((float)row[j] + (float)row[32 + j]) * (p.acc_scale != 0.f ? p.acc_scale : 1.f);
    } else if (wmode == PP_F16) {
      w = (float)reinterpret_cast<const half_t*>(p.weight)[(int64_t)co * p.Kp + tk * 32 + j];
    } else {
      w = reinterpret_cast<const float*>(p.weight)[(int64_t)co * p.Kp + tk * 32 + j];
    }
    wl[i] = w;
  }

  const int ty = tid >> 4, tx = tid & 15;
  const T* in = reinterpret_cast<const T*>(p.in_ptr[0]);
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;

  // ---- the thread's pieces of the input tile: global offset (-1 = outside the image: zero) and LDS offset (-1 = past the
  // tile), fixed for the whole kernel; the pieces of chunk k+1 are loaded into registers while chunk k is multiplied
  constexpr int MAXP = (18 * 18 * PP_ + 255) / 256;
  int64_t goff[MAXP];
  int loff[MAXP], cpos[MAXP];
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int i = tid + q * 256;
    const int hr = i / PP_, j = i - hr * PP_;
    const int hy = hr / hw, hx = hr - hy * hw;
    const int iy = ty0 - p.ph + hy, ix = tx0 - p.pw + hx;
    const bool in_tile = i < hrows * PP_;
    const bool in_img = in_tile && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[q] = in_img ? ((int64_t)(n * p.H + iy) * p.W + ix) * p.in_ldc[0] + j * EPP : -1;
    loff[q] = in_tile ? hr * PITCH + j * 16 : -1;
    cpos[q] = j * EPP;
  }
  f4 xr[MAXP];
  auto load_chunk = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      xr[q] = f4{0.f, 0.f, 0.f, 0.f};
      if (goff[q] >= 0 && k * 32 + cpos[q] < p.in_C[0]) xr[q] = *reinterpret_cast<const f4*>(in + goff[q] + k * 32);
    }
  };
  load_chunk(0);
  for (int k = 0; k < nck; ++k) {
    __syncthreads();                             // the previous chunk's readers are done (and, first, the weights are stored)
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
      if (loff[q] >= 0) *reinterpret_cast<f4*>(xt + loff[q]) = xr[q];
    __syncthreads();
    if (k + 1 < nck) load_chunk(k + 1);          // in flight under this chunk's FMAs
    for (int tap = 0; tap < ntaps; ++tap) {
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const unsigned char* xp = xt + ((ty + ky) * hw + tx + kx) * PITCH;
      const float* wp = (WG ? wg : wl) + (tap * nck + k) * COUT * 32;
#pragma unroll
      for (int j = 0; j < PP_; ++j) {
        if constexpr (sizeof(T) == 2) {
          const h8 xh = *reinterpret_cast<const h8*>(xp + j * 16);
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const f4 w0 = *reinterpret_cast<const f4*>(wp + co * 32 + j * 8);
            const f4 w1 = *reinterpret_cast<const f4*>(wp + co * 32 + j * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co] = __builtin_fmaf((float)xh[e], w0[e], acc[co]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co] = __builtin_fmaf((float)xh[4 + e], w1[e], acc[co]);
          }
        } else {
          const f4 raw = *reinterpret_cast<const f4*>(xp + j * 16);
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const f4 w0 = *reinterpret_cast<const f4*>(wp + co * 32 + j * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co] = __builtin_fmaf(raw[e], w0[e], acc[co]);
          }
        }
      }
    }
  }

  // ---- epilogue (store_quad's operations, per scalar)
  const int oy = ty0 + ty, ox = tx0 + tx;
  if (oy >= p.Ho || ox >= p.Wo) return;
  const int64_t m = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
  OT* out = reinterpret_cast<OT*>(p.out);
  const OT* aux1 = reinterpret_cast<const OT*>(p.aux1);
  const OT* aux2 = reinterpret_cast<const OT*>(p.aux2);
  const OT* pre = reinterpret_cast<const OT*>(p.pre_add);
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    if (co >= p.Cout) break;
    float v = acc[co];
    if (p.bias) v += p.bias[co];
    if (pre) v += to_f32(pre[m * p.pre_add_ldc + co]);
    if (p.act_split > 0 && co >= p.act_split) {
      v = apply_act(v, p.act2, p.act_param);
    } else {
      v = apply_act(v, p.act, p.act_param);
      if (p.out_scale != 0.f) v *= p.out_scale;
    }
    if (p.epi != PP_EPI_NONE) {
      const float a1 = to_f32(aux1[m * p.aux1_ldc + co]);
      if (p.epi == PP_EPI_MUL_AUX1) {
        v *= a1;
      } else if (p.epi == PP_EPI_ADD_AUX1) {
        v += a1;
      } else if (p.epi == PP_EPI_ADD_AUX1_RELU) {
        const float s = v + a1;
        v = s > 0.f ? s : 0.f;
      } else if (p.epi == PP_EPI_GRU) {
        const float h = to_f32(aux2[m * p.aux2_ldc + co]);
        v = (1.f - a1) * h + a1 * v;
      }
    }
    out[m * p.out_ldc + co] = from_f32<OT>(v);
  }
}

template <typename T, typename OT>
static int launch_direct_t(void* stream, const ConvK& k, int wmode) {
  const int tiles_x = (k.Wo + kDirT - 1) / kDirT, tiles_y = (k.Ho + kDirT - 1) / kDirT;
  const int64_t blocks = (int64_t)k.N * tiles_x * tiles_y;
  const int hrows = (kDirT + k.kh - 1) * (kDirT + k.kw - 1);
  const int cmax = k.Cout <= 2 ? 2 : 4;
  const bool wg = k.weight_f32 != nullptr;
  const size_t smem = (wg ? 0 : (size_t)k.nchunks * cmax * 32 * sizeof(float)) + (size_t)hrows * (32 * sizeof(T) + 16);
  if (smem > 150 * 1024 || blocks >= ((int64_t)1 << 31)) return 1;
  dim3 grid((unsigned)blocks), block(256);
#define PP_DIRECT_LAUNCH(C, W)                                                                                              \
  do {                                                                                                                      \
    PP_ALLOW_BIG_LDS((&conv_small_cout_kernel<T, OT, C, W>), 150 * 1024);                                                     \
    PP_LAUNCH((conv_small_cout_kernel<T, OT, C, W>), grid, block, smem, stream, k, tiles_x, tiles_y, wmode);                \
  } while (0)
  if (cmax == 2) {
    if (wg) PP_DIRECT_LAUNCH(2, true); else PP_DIRECT_LAUNCH(2, false);
  } else {
    if (wg) PP_DIRECT_LAUNCH(4, true); else PP_DIRECT_LAUNCH(4, false);
  }
#undef PP_DIRECT_LAUNCH
  return pp_check_launch("pp_conv2d");
}

// returns 1 when the convolution is not eligible (the caller goes on to the implicit-GEMM kernels): at most 4 output
// channels, one input segment, stride 1, dilation 1, zero padding, at most 3 x 3 taps, one z slice, and enough
// pixels PER IMAGE that the layer is a streaming problem (>= 2048: RAFT's flow head at 45 x 80 and everything larger; the
// rule looks at one image, never at the batch, so that a rank of a sharded run picks the same kernel as the single-GPU run).
// PP_CONV_DIRECT=0 disables, "force" lifts the size rule (tests) -- pp_options.h.
int launch_direct_small_cout(void* stream, const ConvK& k, int Z, int dtype, bool out_f16) {
  const int mode = options().direct;
  if (mode == 0) return 1;
  const bool force = mode == 2;
  if (k.epi_from != 0) return 1;
  if (k.Cout > 4 || k.nseg != 1 || Z != 1 || k.sh != 1 || k.sw != 1 || k.dh != 1 || k.dw != 1 || k.pad_mode != PP_PAD_ZEROS) return 1;
  if (k.kh > 3 || k.kw > 3) return 1;
  if (k.Ho != k.H + 2 * k.ph - (k.kh - 1) || k.Wo != k.W + 2 * k.pw - (k.kw - 1)) return 1;
  if (!force && (int64_t)k.Ho * k.Wo < 2048) return 1;
  if (dtype == PP_F16) return out_f16 ? launch_direct_t<half_t, half_t>(stream, k, dtype) : launch_direct_t<half_t, float>(stream, k, dtype);
  if (out_f16) return 1;
  return launch_direct_t<float, float>(stream, k, dtype);
}

}  // namespace pp

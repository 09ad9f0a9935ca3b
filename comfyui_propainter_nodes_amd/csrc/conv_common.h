// conv_common.h -- pieces shared by the convolution translation units (conv_igemm.hip, conv_split.hip):
// kernel-argument block, fused epilogue, compile-time loops, tile-family dispatch.
#pragma once
#include "pp_device.h"
#include "pp_host.h"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "pp_options.h"
#include <utility>

namespace pp {


// helper lambdas of the kernels must never become real calls (a call makes the kernel-argument struct and the
// register arrays addressable: both would move to scratch memory)
#define PP_INLINE_LAMBDA __attribute__((always_inline))

struct ConvK {
  const void* in_ptr[PP_MAX_SEG];
  int in_C[PP_MAX_SEG];
  int in_ldc[PP_MAX_SEG];
  int64_t in_zoff[PP_MAX_SEG];
  int seg_chunks[PP_MAX_SEG];
  int nseg;
  int N, H, W, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  int pad_mode;
  const void* weight;
  int64_t w_zoff;
  int Kp;
  const float* bias;
  int64_t bias_zoff;
  int Cout;
  int64_t M;
  void* out;
  int out_ldc;
  int64_t out_zoff;
  int act, act2, act_split;
  float act_param, out_scale;
  int epi;
  int epi_from;  // the epilogue op applies to channels >= epi_from and reads aux1 / aux2 at channel c - epi_from
  const void* aux1;
  int aux1_ldc;
  int64_t aux1_zoff;
  const void* aux2;
  int aux2_ldc;
  int64_t aux2_zoff;
  int chunks_per_tap;
  int nchunks;
  const void* pre_add;
  int pre_add_ldc;
  int tile_order;  // flat-tile kernels: 1 = XCD-contiguous, channel tiles of a pixel tile adjacent (flat_tile_of); 0 = launch order
  const float* weight_f32;  // optional fp32 [tap][chunk][Cout padded to 2 / 4][32] table (conv_direct.hip)
  int flat_taps;            // weights = one (ky, kx, c)-ordered row per output channel; conv_gemm_f16.hip gathers the patches
  float acc_scale;          // PP_F32X2: 1 / (power-of-two scale of the packed weights); 1 otherwise
  int epi_lds;              // 1 (default): LDS-transposed epilogue (epilogue_quads_lds); 0 (PP_CONV_EPI=direct): quads stored as the MFMA leaves them
  int many_images;          // pp_conv2d_params.many_images (ABI v12): never the small-image split-K kernel
  int epi_oct;              // 1 (default): the GEMM kernel's f16 outputs as 16-byte stores of paired quads (epilogue_octs_fast); 0 (PP_CONV_EPI_OCT=0): 8-byte quads
};

__device__ __forceinline__ float apply_act(float v, int act, float param) {
  switch (act) {
    case PP_ACT_RELU: return v > 0.f ? v : 0.f;
    case PP_ACT_LEAKY: return v > 0.f ? v : v * param;
    case PP_ACT_SIGMOID: return sigmoidf_(v);
    case PP_ACT_TANH: return tanhf_(v);
    case PP_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    default: return v;
  }
}

template <typename T>
struct Frag;
template <>
struct Frag<half_t> {
  typedef h8 piece;  // 16 bytes
};
template <>
struct Frag<float> {
  typedef f4 piece;
};

// compile-time loop: the body receives std::integral_constant<int, I>, so every index derived from it
// is a constant expression (register arrays indexed with it can never fall back to scratch memory)
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <typename OT>
struct EpiCtx {
  const float* bias;
  OT* out;
  const OT* aux1;
  const OT* aux2;
  const OT* pre;
#ifdef PP_HALO_TRACE
  mutable uint32_t st[8];   // tools/trace_halo.sh build only: s_memtime stamps inside the epilogue
#endif
};
#ifdef PP_HALO_TRACE
#define PP_EPI_STAMP(e, i)                 \
  __builtin_amdgcn_sched_barrier(0);       \
  (e).st[i] = (uint32_t)__builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_sched_barrier(0)
#else
#define PP_EPI_STAMP(e, i)
#endif

// 4 consecutive channels of one pixel row as floats: one 8/16-byte load when `vec`, else guarded scalars
template <typename ET>
__device__ __forceinline__ f4 load_quad(const ET* src, bool vec, int nvalid) {
  f4 r = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    if constexpr (sizeof(ET) == 2) {
      const h4 t = *reinterpret_cast<const h4*>(src);
      r = f4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    } else {
      r = *reinterpret_cast<const f4*>(src);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nvalid) r[i] = to_f32(src[i]);
  }
  return r;
}

template <typename ET>
__device__ __forceinline__ bool quad_aligned(const ET* ptr, int64_t ldc) {
  return ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(ptr) & (4 * sizeof(ET) - 1)) == 0);
}

// Work-group -> (pixel tile, channel tile) of the flat-tile kernels (1-D grid of npt * nct work-groups per z).
// r01-r03 launched a 2-D grid, pixel tiles along x: the nct work-groups that read the SAME pixel tile were npt dispatches
// apart and on different XCDs, so a GEMM with several channel tiles (the transformer's 512 -> 1536 / 1960 layers: 12 / 16)
// fetched its activation matrix from the fabric once per channel tile (rocprofv3 FETCH_SIZE of the f16 family: ~2.4x the
// algorithmic bytes).  Order 1: each XCD owns a contiguous range of the linear tile space with the channel tiles of one pixel
// tile adjacent (the halo-tile kernels' mapping): they run back to back on one XCD and the pixel tile comes out of its L2.
__device__ __forceinline__ void flat_tile_of(const ConvK& p, int BC, int& pt, int& ct) {
  const int nct = (p.Cout + BC - 1) / BC;
  const int nwg = (int)gridDim.x, id = (int)blockIdx.x;
  if (p.tile_order) {
    // channel tiles in groups of at most 16 (2 MB of f16 weights at K = 512): a layer with more (the soft-composite's
    // 512 -> 6272: 49) would cycle through more weights than an L2 holds between two uses of a tile; group by group it
    // re-reads the pixel tiles once per group instead
    constexpr int G = 16;
    const int L = xcd_contiguous_block(id, nwg);
    const int npt = nwg / nct;
    const int full = (nct / G) * (npt * G);   // work-groups in whole groups
    if (L < full) {
      const int r = L % (npt * G);
      ct = (L / (npt * G)) * G + r % G;
      pt = r / G;
    } else {
      const int rem = nct % G, r = L - full;
      ct = (nct / G) * G + r % rem;
      pt = r / rem;
    }
  } else {
    const int npt = nwg / nct;
    pt = id % npt;
    ct = id / npt;
  }
}

// bias + activation(s) + scale + fused epilogue op + channels-last store of 4 consecutive channels
template <typename OT>
__device__ __forceinline__ void store_quad(const ConvK& p, const EpiCtx<OT>& e, f4 accv, int64_t m, int c) {
  const int nvalid = p.Cout - c;  // >= 1 (caller checks c < Cout)
  const bool full = nvalid >= 4;
  f4 v = accv;
  if (e.bias) {
    const f4 b = load_quad(e.bias + c, full && ((c & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.bias) & 15) == 0), nvalid);
    v += b;
  }
  if (e.pre) {
    const OT* src = e.pre + m * p.pre_add_ldc + c;
    v += load_quad(src, full && quad_aligned(src, p.pre_add_ldc), nvalid);
  }
  if (p.act_split > 0 && c + 3 >= p.act_split && c < p.act_split) {
    // quad straddles the act/act2 boundary (never happens for the shipped nets: split % 4 == 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (c + r >= p.act_split) {
        v[r] = apply_act(v[r], p.act2, p.act_param);
      } else {
        v[r] = apply_act(v[r], p.act, p.act_param);
        if (p.out_scale != 0.f) v[r] *= p.out_scale;
      }
    }
  } else if (p.act_split > 0 && c >= p.act_split) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act2, p.act_param);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act, p.act_param);
    if (p.out_scale != 0.f) v *= p.out_scale;
  }
  if (p.epi != PP_EPI_NONE && c >= p.epi_from) {  // (epi_from % 4 == 0: no quad straddles it)
    const OT* s1 = e.aux1 + m * p.aux1_ldc + (c - p.epi_from);
    const f4 a1 = load_quad(s1, full && quad_aligned(s1, p.aux1_ldc), nvalid);
    if (p.epi == PP_EPI_MUL_AUX1) {
      v *= a1;
    } else if (p.epi == PP_EPI_ADD_AUX1) {
      v += a1;
    } else if (p.epi == PP_EPI_ADD_AUX1_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = v[r] + a1[r];
        v[r] = s > 0.f ? s : 0.f;
      }
    } else if (p.epi == PP_EPI_GRU) {
      const OT* s2 = e.aux2 + m * p.aux2_ldc + (c - p.epi_from);
      const f4 h = load_quad(s2, full && quad_aligned(s2, p.aux2_ldc), nvalid);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (1.f - a1[r]) * h[r] + a1[r] * v[r];
    }
  }
  OT* dst = e.out + m * p.out_ldc + c;
  if (full && quad_aligned(dst, p.out_ldc)) {
    if constexpr (sizeof(OT) == 2) {
      h4 o = {sat_half(v[0]), sat_half(v[1]), sat_half(v[2]), sat_half(v[3])};
      *reinterpret_cast<h4*>(dst) = o;
    } else {
      *reinterpret_cast<f4*>(dst) = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < nvalid) dst[r] = from_f32<OT>(v[r]);
  }
}

// ---- fast epilogue.  store_quad() above decides per lane and per quad whether a tensor can be touched with a vector
// access (alignment, partial channel quads, an activation boundary inside a quad): correct for any view, but it costs
// ~200 executed instructions per quad with both forms of every access in the instruction stream -- for a 16-quad wave
// tile more issue time than the MFMAs of a short reduction.  Whether EVERY quad of a launch is a full aligned vector on
// every tensor is a property of the launch: epi_fast_ok() tests it once (scalar work), store_quad_fast() then has no
// per-lane decisions left but the activation split.
template <typename OT>
__device__ __forceinline__ bool epi_fast_ok(const ConvK& p, const EpiCtx<OT>& e) {
  constexpr uintptr_t AM = 4 * sizeof(OT) - 1;
  bool ok = (p.Cout & 3) == 0 && (p.act_split & 3) == 0;
  ok = ok && (p.out_ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(e.out) & AM) == 0;
  ok = ok && (!e.bias || (reinterpret_cast<uintptr_t>(e.bias) & 15) == 0);
  ok = ok && (!e.pre || ((p.pre_add_ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(e.pre) & AM) == 0));
  if (p.epi != PP_EPI_NONE) {
    ok = ok && (p.aux1_ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(e.aux1) & AM) == 0;
    if (p.epi == PP_EPI_GRU) ok = ok && (p.aux2_ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(e.aux2) & AM) == 0;
  }
  return ok;
}

template <typename ET>
__device__ __forceinline__ f4 load_quad_vec(const ET* src) {
  if constexpr (sizeof(ET) == 2) {
    const h4 t = *reinterpret_cast<const h4*>(src);
    return f4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
  } else {
    return *reinterpret_cast<const f4*>(src);
  }
}

__device__ __forceinline__ f4 apply_act4(f4 v, int act, float param) {  // one (uniform) switch per quad
  switch (act) {
    case PP_ACT_RELU:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      break;
    case PP_ACT_LEAKY:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * param;
      break;
    case PP_ACT_SIGMOID:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
      break;
    case PP_ACT_TANH:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
      break;
    case PP_ACT_GELU:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));
      break;
    default: break;
  }
  return v;
}

// the arithmetic of store_quad() (same operations in the same order) for launches that passed epi_fast_ok()
template <typename OT>
__device__ __forceinline__ void store_quad_fast(const ConvK& p, const EpiCtx<OT>& e, f4 v, int64_t m, int c) {
  if (e.bias) v += *reinterpret_cast<const f4*>(e.bias + c);
  if (e.pre) v += load_quad_vec(e.pre + m * p.pre_add_ldc + c);
  if (p.act_split > 0 && c >= p.act_split) {  // (act_split % 4 == 0: no quad straddles the boundary)
    v = apply_act4(v, p.act2, p.act_param);
  } else {
    v = apply_act4(v, p.act, p.act_param);
    if (p.out_scale != 0.f) v *= p.out_scale;
  }
  if (p.epi != PP_EPI_NONE && c >= p.epi_from) {
    const f4 a1 = load_quad_vec(e.aux1 + m * p.aux1_ldc + (c - p.epi_from));
    if (p.epi == PP_EPI_MUL_AUX1) {
      v *= a1;
    } else if (p.epi == PP_EPI_ADD_AUX1) {
      v += a1;
    } else if (p.epi == PP_EPI_ADD_AUX1_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = v[r] + a1[r];
        v[r] = s > 0.f ? s : 0.f;
      }
    } else if (p.epi == PP_EPI_GRU) {
      const f4 h = load_quad_vec(e.aux2 + m * p.aux2_ldc + (c - p.epi_from));
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (1.f - a1[r]) * h[r] + a1[r] * v[r];
    }
  }
  OT* dst = e.out + m * p.out_ldc + c;
  if constexpr (sizeof(OT) == 2) {
    h4 o = {sat_half(v[0]), sat_half(v[1]), sat_half(v[2]), sat_half(v[3])};
    *reinterpret_cast<h4*>(dst) = o;
  } else {
    *reinterpret_cast<f4*>(dst) = v;
  }
}

// The epilogue of a wave tile of NA x NB quads: row(b, m, ok) gives the output pixel of quad column b (and whether it
// exists), chan(a) the first channel of quad row a for this lane, val(a, b) the finished accumulator quad (compile-time
// indices: the callers' accumulators are register arrays).  One uniform branch picks the fast or the general form.
template <typename OT, int NA, int NB, typename RowFn, typename ChanFn, typename ValFn>
__device__ __forceinline__ void epilogue_quads_fast(const ConvK& p, const EpiCtx<OT>& e, RowFn row, ChanFn chan, ValFn val) {
  static_for<NB>([&](auto bi) {
    int64_t m;
    bool ok;
    row(bi, m, ok);
    static_for<NA>([&](auto ai) {
      const int c = chan(ai);
      if (ok && c < p.Cout) store_quad_fast<OT>(p, e, val(ai, bi), m, c);
    });
  });
}

// The general (per-lane decisions) form alone, for launches that fail epi_fast_ok()
template <typename OT, int NA, int NB, typename RowFn, typename ChanFn, typename ValFn>
__device__ __forceinline__ void epilogue_quads_general(const ConvK& p, const EpiCtx<OT>& e, RowFn row, ChanFn chan, ValFn val) {
  static_for<NB>([&](auto bi) {
    int64_t m;
    bool ok;
    row(bi, m, ok);
    static_for<NA>([&](auto ai) {
      const int c = chan(ai);
      if (ok && c < p.Cout) store_quad<OT>(p, e, val(ai, bi), m, c);
    });
  });
}

template <typename OT, int NA, int NB, typename RowFn, typename ChanFn, typename ValFn>
__device__ __forceinline__ void epilogue_quads(const ConvK& p, const EpiCtx<OT>& e, RowFn row, ChanFn chan, ValFn val) {
  if (epi_fast_ok<OT>(p, e)) {
    epilogue_quads_fast<OT, NA, NB>(p, e, row, chan, val);
  } else {
    epilogue_quads_general<OT, NA, NB>(p, e, row, chan, val);
  }
}

// ---- paired quads (r06, f16 outputs).  Two channel-adjacent 16 x 16 tiles of a wave leave lane (g, j) -- g = lane >> 4 -- the
// channels 4g..4g+3 of BOTH tiles for pixel j: 8-byte stores, 32 bytes per pixel and instruction.  Four v_permlane16_swap (rows 1 <-> 0
// and 3 <-> 2 between the two quads) give every lane 8 CONSECUTIVE channels of one tile instead -- rows 0 / 2: channels 0-7 / 8-15 of
// the first tile, rows 1 / 3: of the second -- so a wave stores 16 bytes per lane, 64 bytes per pixel and instruction, in half the
// store (and residual / pre-add load) instructions.  Pure data movement in front of store_quad_fast()'s arithmetic: the same bits.
template <typename OT>
__device__ __forceinline__ bool epi_oct_ok(const ConvK& p, const EpiCtx<OT>& e) {
  if constexpr (sizeof(OT) != 2) {
    return false;
  } else {
    bool ok = epi_fast_ok<OT>(p, e);
    ok = ok && (p.Cout & 7) == 0 && (p.act_split & 7) == 0 && (p.out_ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(e.out) & 15) == 0;
    ok = ok && (!e.pre || ((p.pre_add_ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(e.pre) & 15) == 0));
    if (p.epi != PP_EPI_NONE) {
      ok = ok && (p.epi_from & 7) == 0 && (p.aux1_ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(e.aux1) & 15) == 0;
      if (p.epi == PP_EPI_GRU) ok = ok && (p.aux2_ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(e.aux2) & 15) == 0;
    }
    return ok;
  }
}

__device__ __forceinline__ void load_oct_vec(const half_t* src, f4& lo, f4& hi) {
  const h8 t = *reinterpret_cast<const h8*>(src);
  lo = f4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
  hi = f4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
}

// store_quad_fast() on the two quads of 8 consecutive channels c..c+7 of pixel m (same operations in the same order per value)
__device__ __forceinline__ void store_oct_fast(const ConvK& p, const EpiCtx<half_t>& e, f4 v0, f4 v1, int64_t m, int c) {
  if (e.bias) {
    v0 += *reinterpret_cast<const f4*>(e.bias + c);
    v1 += *reinterpret_cast<const f4*>(e.bias + c + 4);
  }
  if (e.pre) {
    f4 a, b;
    load_oct_vec(e.pre + m * p.pre_add_ldc + c, a, b);
    v0 += a;
    v1 += b;
  }
  if (p.act_split > 0 && c >= p.act_split) {
    v0 = apply_act4(v0, p.act2, p.act_param);
    v1 = apply_act4(v1, p.act2, p.act_param);
  } else {
    v0 = apply_act4(v0, p.act, p.act_param);
    v1 = apply_act4(v1, p.act, p.act_param);
    if (p.out_scale != 0.f) {
      v0 *= p.out_scale;
      v1 *= p.out_scale;
    }
  }
  if (p.epi != PP_EPI_NONE && c >= p.epi_from) {
    f4 a0, a1;
    load_oct_vec(e.aux1 + m * p.aux1_ldc + (c - p.epi_from), a0, a1);
    if (p.epi == PP_EPI_MUL_AUX1) {
      v0 *= a0;
      v1 *= a1;
    } else if (p.epi == PP_EPI_ADD_AUX1) {
      v0 += a0;
      v1 += a1;
    } else if (p.epi == PP_EPI_ADD_AUX1_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s0 = v0[r] + a0[r], s1 = v1[r] + a1[r];
        v0[r] = s0 > 0.f ? s0 : 0.f;
        v1[r] = s1 > 0.f ? s1 : 0.f;
      }
    } else if (p.epi == PP_EPI_GRU) {
      f4 h0, h1;
      load_oct_vec(e.aux2 + m * p.aux2_ldc + (c - p.epi_from), h0, h1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] = (1.f - a0[r]) * h0[r] + a0[r] * v0[r];
        v1[r] = (1.f - a1[r]) * h1[r] + a1[r] * v1[r];
      }
    }
  }
  const h8 o = {sat_half(v0[0]), sat_half(v0[1]), sat_half(v0[2]), sat_half(v0[3]),
                sat_half(v1[0]), sat_half(v1[1]), sat_half(v1[2]), sat_half(v1[3])};
  *reinterpret_cast<h8*>(e.out + m * p.out_ldc + c) = o;
}

// wave tile of NA x NB quads, NA even; chan0 = the wave tile's first channel; g = lane >> 4
template <int NA, int NB, typename RowFn, typename ValFn>
__device__ __forceinline__ void epilogue_octs_fast(const ConvK& p, const EpiCtx<half_t>& e, int chan0, int g, RowFn row, ValFn val) {
  static_assert(NA % 2 == 0, "quad rows are paired");
  static_for<NB>([&](auto bi) {
    int64_t m;
    bool ok;
    row(bi, m, ok);
    static_for<NA / 2>([&](auto a2) {
      constexpr int A = 2 * decltype(a2)::value;
      f4 v0 = val(std::integral_constant<int, A>{}, bi), v1 = val(std::integral_constant<int, A + 1>{}, bi);
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // every lane takes part, whatever `ok`
        float a = v0[r], b = v1[r];
        swap_rows16(a, b);
        v0[r] = a;
        v1[r] = b;
      }
      const int c = chan0 + (A + (g & 1)) * 16 + (g >> 1) * 8;
      if (ok && c < p.Cout) store_oct_fast(p, e, v0, v1, m, c);
    });
  });
}

// ---- LDS-transposed epilogue (r05).  The MFMA leaves a lane 4 consecutive channels of ONE pixel per 16 x 16 tile, neighbouring
// lanes neighbouring PIXELS: a wave's store instruction above is 16 separate 64-byte (f32) / 32-byte (f16) pieces a channel pitch
// apart, and the aux / pre-add loads of the fused epilogues gather the same way.  The s_memtime phase trace of the PP_F32X2 halo
// kernels (profiles/r05_f32x2_phase_trace.md) put the epilogue at 14 000 of a work-group's 101 000 cycles -- 16 such stores per
// wave -- with the matrix pipe idle meanwhile.  Here a wave stages one quad ROW (16 pixels x NA*16 channels of raw accumulators,
// fp32) in a wave-private LDS region and reads it back with neighbouring lanes on neighbouring CHANNELS of one pixel: every
// store / aux load of 16 lanes is NA*64 contiguous bytes (f32; half for f16), i.e. whole cache lines.  The arithmetic per element
// is store_quad_fast()'s, unchanged: results are bit-identical to the direct form.
// Layout of the staging region: [16 pixels][NA*16 floats + 4 pad floats] (the pad spreads the 8-lane groups of ds_write_b128 over
// all banks).  row0(b, m0, nvalid): output pixel of column 0 of quad row b and the number of valid columns (<= 16, may be <= 0).
template <int NA>
constexpr int epi_lds_pitch() { return NA * 64 + 16; }
template <int NA>
constexpr int epi_lds_wave_bytes() { return 16 * epi_lds_pitch<NA>(); }

// Compile-time forms of apply_act4() / the fused epilogue op (same operations in the same order as store_quad_fast(): the
// results are bit-identical); a negative template value = decided at run time (the generic variant).
template <int ACT>
__device__ __forceinline__ f4 act4_ct(f4 v, int act_rt, float param) {
  if constexpr (ACT < 0) {
    return apply_act4(v, act_rt, param);
  } else if constexpr (ACT == PP_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    return v;
  } else if constexpr (ACT == PP_ACT_LEAKY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * param;
    return v;
  } else if constexpr (ACT == PP_ACT_SIGMOID) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
    return v;
  } else if constexpr (ACT == PP_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
    return v;
  } else if constexpr (ACT == PP_ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));
    return v;
  } else {
    return v;
  }
}

// One variant of the LDS-transposed epilogue.  ACT / ACT2 / EPI / PRE / OSC: compile-time activation of the channels below /
// from p.act_split, the fused op, whether a pre-activation addend exists and whether out_scale applies (ACT2 = -2: the launch has
// no activation split; ACT = -1: everything decided at run time -- the catch-all variant).
//
// What the r05 measurements say an epilogue costs (profiles/r05_f32x2_phase_trace.md): 10-15 000 of a PP_F32X2 halo work-group's
// ~100 000 cycles, for 16 stores per wave.  Not the stores (tools/probes/store_rate: a wave issues 16 KB in ~700 cycles next to a
// loaded chip; coalescing them changed nothing), not instruction fetch (variants: same time), only partly vmcnt (stores count in
// vmcnt on gfx9, so a load's wait behind a store waits for its write acknowledgement: -3 000 cycles once no load follows a store).
// The stamps inside the epilogue show ~2 500 cycles per quad ROW, 1 500 when the wave has the SIMD to itself: the ~200 vector-ALU
// instructions of a row each wait for a gap between the MFMAs of the work-group that shares the SIMD (one every 16 cycles) -- next to
// a matrix-bound partner a plain VALU instruction costs about as much as an MFMA.  So this form is written for INSTRUCTION COUNT:
//   * addresses: a scalar row base (64-bit SALU) + per-lane 32-bit byte offsets computed ONCE (pixel and channel quad of a lane
//     never change): a store / aux load is one instruction with no address arithmetic (r02-r04: ~6 VALU of 64-bit math per access);
//   * accumulator scale and bias are ONE fma per element (exact: the scale is a power of two); a launch without bias adds -0.0,
//     which changes no bit; run-time switches (bias, pre, out_scale) are compile-time or scalar, never per-lane selects;
//   * whole rows (all 16 pixels inside the image, all channels below Cout) take a path without any per-lane predicate.
// No wait for a load ever has an older store in front of it: bias quads are loaded once before the first store, the loads of quad
// row b + 1 are issued BEFORE the stores of row b, and a variant without aux / pre tensors has no load in its loop.
template <typename OT, int NA, int NB, bool SCALED, int ACT, int ACT2, int EPI, int PRE, int OSC, typename Row0Fn, typename ValFn>
__device__ __forceinline__ void epilogue_lds_variant(const ConvK& p, const EpiCtx<OT>& e, unsigned char* wlds, int lane, int c_wave,
                                                     Row0Fn row0, ValFn val) {
  constexpr int PITCH = epi_lds_pitch<NA>();
  constexpr int QPR = NA * 4;              // quads per staged pixel
  constexpr bool RT = ACT < 0;             // the run-time variant
  const int frow = lane & 15, fgrp = lane >> 4;
  const bool has_pre = RT ? e.pre != nullptr : PRE != 0;
  const bool has_osc = RT ? p.out_scale != 0.f : OSC != 0;
  const int epi = RT ? p.epi : EPI;
  const int cmax = p.Cout - 4;             // (fast form: Cout % 4 == 0) loads of a lane past Cout are clamped, its stores masked
  const float sc = SCALED ? p.acc_scale : 1.f;   // (PP_F32X2 kernels hand over RAW accumulators: ConvK::acc_scale)
  // per-lane constants of the transposed layout: quad i of a row is pixel px[i], channel quad cc[i]
  int cc[NA], px[NA];
  bool cok[NA], second[NA], from[NA];
  uint32_t o_out[NA], o_a1[NA], o_a2[NA], o_pre[NA], lrd[NA];
  f4 bq[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = i * 64 + lane;
    px[i] = q / QPR;
    const int c = c_wave + (q - px[i] * QPR) * 4;
    cok[i] = c < p.Cout;
    cc[i] = c < cmax ? c : cmax;
    second[i] = ACT2 != -2 && p.act_split > 0 && cc[i] >= p.act_split;
    from[i] = cc[i] >= p.epi_from;
    const int ce = from[i] ? cc[i] - p.epi_from : 0;
    o_out[i] = (uint32_t)(px[i] * p.out_ldc + cc[i]) * (uint32_t)sizeof(OT);
    o_a1[i] = (uint32_t)(px[i] * p.aux1_ldc + ce) * (uint32_t)sizeof(OT);
    o_a2[i] = (uint32_t)(px[i] * p.aux2_ldc + ce) * (uint32_t)sizeof(OT);
    o_pre[i] = (uint32_t)(px[i] * p.pre_add_ldc + cc[i]) * (uint32_t)sizeof(OT);
    lrd[i] = (uint32_t)(px[i] * PITCH + (q - px[i] * QPR) * 16);
    bq[i] = f4{-0.f, -0.f, -0.f, -0.f};    // (x * sc + -0.0 == x * sc bit for bit)
    if (e.bias) bq[i] = *reinterpret_cast<const f4*>(e.bias + cc[i]);
  }
  const bool full_c = c_wave + NA * 16 <= p.Cout;   // (uniform) every channel quad of the wave tile exists
  typedef typename std::conditional<sizeof(OT) == 2, h4, f4>::type rawq;   // a loaded quad as it travels: converted at its use
  struct RowLd {
    rawq a1[NA], a2[NA], pr[NA];
    int64_t m0;
    int nvalid;
  };
  auto cvt = [](rawq r) PP_INLINE_LAMBDA {
    if constexpr (sizeof(OT) == 2) return f4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    else return r;
  };
  // (the loads of row b + 1 are issued before the stores of row b: two buffers)
  // (the fp32 variants with three loaded tensors -- GRU blend + pre-add, the catch-all -- keep ONE buffer: two are 96 registers beside
  //  the 64 accumulators, which made the 256-register kernels spill 56-116 registers: tests/test_isa_audit.py, tools/shipped_isa.py)
  constexpr bool AHEAD = !(sizeof(OT) == 4 && (RT || EPI == PP_EPI_GRU));
  RowLd rl[AHEAD ? 2 : 1];
  auto ldq = [&](const OT* base, uint32_t off) PP_INLINE_LAMBDA {
    return *reinterpret_cast<const rawq*>(reinterpret_cast<const char*>(base) + off);
  };
  auto issue_row = [&](auto bi, RowLd& r) PP_INLINE_LAMBDA {
    row0(bi, r.m0, r.nvalid);
    if (!has_pre && epi == PP_EPI_NONE) return;
    const OT* bp = e.pre + r.m0 * p.pre_add_ldc;
    const OT* b1 = e.aux1 + r.m0 * p.aux1_ldc;
    const OT* b2 = e.aux2 + r.m0 * p.aux2_ldc;
    if (r.nvalid >= 16) {                  // (uniform) every pixel of the row exists: no predicate
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (has_pre) r.pr[i] = ldq(bp, o_pre[i]);
        if (epi != PP_EPI_NONE) r.a1[i] = ldq(b1, o_a1[i]);
        if (epi == PP_EPI_GRU) r.a2[i] = ldq(b2, o_a2[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        r.pr[i] = r.a1[i] = r.a2[i] = rawq{0, 0, 0, 0};
        if (px[i] < r.nvalid) {
          if (has_pre) r.pr[i] = ldq(bp, o_pre[i]);
          if (epi != PP_EPI_NONE) r.a1[i] = ldq(b1, o_a1[i]);
          if (epi == PP_EPI_GRU) r.a2[i] = ldq(b2, o_a2[i]);
        }
      }
    }
  };
  // store_quad_fast()'s arithmetic on one quad, operation by operation (acc * sc + bias as one fma: sc is a power of two)
  auto finish = [&](f4 x, int i, const RowLd& r) PP_INLINE_LAMBDA {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __builtin_fmaf(x[k], sc, bq[i][k]);
    if (has_pre) x += cvt(r.pr[i]);
    if (second[i]) {
      x = act4_ct<ACT2>(x, p.act2, p.act_param);
    } else {
      x = act4_ct<ACT>(x, p.act, p.act_param);
      if (has_osc) x *= p.out_scale;
    }
    if (epi != PP_EPI_NONE && from[i]) {
      const f4 a1 = cvt(r.a1[i]);
      if (epi == PP_EPI_MUL_AUX1) {
        x *= a1;
      } else if (epi == PP_EPI_ADD_AUX1) {
        x += a1;
      } else if (epi == PP_EPI_ADD_AUX1_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = x[k] + a1[k];
          x[k] = t > 0.f ? t : 0.f;
        }
      } else if (epi == PP_EPI_GRU) {
        const f4 a2 = cvt(r.a2[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (1.f - a1[k]) * a2[k] + a1[k] * x[k];
      }
    }
    return x;
  };
  auto put = [&](OT* rowp, uint32_t off, f4 x) PP_INLINE_LAMBDA {
    OT* dst = reinterpret_cast<OT*>(reinterpret_cast<char*>(rowp) + off);
    if constexpr (sizeof(OT) == 2) {
      h4 o = {sat_half(x[0]), sat_half(x[1]), sat_half(x[2]), sat_half(x[3])};
      *reinterpret_cast<h4*>(dst) = o;
    } else {
      *reinterpret_cast<f4*>(dst) = x;
    }
  };
  PP_EPI_STAMP(e, 1);
  if constexpr (AHEAD) issue_row(std::integral_constant<int, 0>{}, rl[0]);
  PP_EPI_STAMP(e, 2);
  static_for<NB>([&](auto bi) {
    constexpr int b = decltype(bi)::value;
    RowLd& r = rl[AHEAD ? (b & 1) : 0];
    if constexpr (!AHEAD) issue_row(bi, r);
    f4 v[NA];
    static_for<NA>([&](auto ai) {
      *reinterpret_cast<f4*>(wlds + frow * PITCH + (decltype(ai)::value * 16 + fgrp * 4) * 4) = val(ai, bi);
    });
    pp_wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NA; ++i) v[i] = *reinterpret_cast<const f4*>(wlds + lrd[i]);
    if constexpr (AHEAD && b + 1 < NB) issue_row(std::integral_constant<int, b + 1>{}, rl[(b + 1) & 1]);  // before this row's stores
    OT* rowp = e.out + r.m0 * p.out_ldc;
    if (r.nvalid >= 16 && full_c) {        // (uniform) the whole row is stored: no predicate
#pragma unroll
      for (int i = 0; i < NA; ++i) put(rowp, o_out[i], finish(v[i], i, r));
    } else if (r.nvalid > 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const f4 x = finish(v[i], i, r);
        if (cok[i] && px[i] < r.nvalid) put(rowp, o_out[i], x);
      }
    }
    pp_wave_lds_fence();  // the next row's staging writes come after this row's reads
    if constexpr (b < 4) { PP_EPI_STAMP(e, 3 + b); }
  });
}

// Dispatch to the variant of the launch's (act, act2, epi, pre_add, out_scale): the combinations the pipeline's layers use are
// compiled in, anything else takes the run-time variant (same arithmetic).
template <typename OT, int NA, int NB, bool SCALED, typename Row0Fn, typename ValFn>
__device__ __forceinline__ void epilogue_quads_lds(const ConvK& p, const EpiCtx<OT>& e, unsigned char* wlds, int lane, int c_wave,
                                                   Row0Fn row0, ValFn val) {
  const int a2 = p.act_split > 0 ? p.act2 : -2;
  const int pre = e.pre != nullptr ? 1 : 0;
  const int osc = p.out_scale != 0.f ? 1 : 0;
#define PP_EPI_VARIANT(A, A2, E, P, O)                                                                       \
  if (p.act == (A) && a2 == (A2) && p.epi == (E) && pre == (P) && osc == (O)) {                             \
    epilogue_lds_variant<OT, NA, NB, SCALED, (A), (A2), (E), (P), (O)>(p, e, wlds, lane, c_wave, row0, val);        \
    return;                                                                                                  \
  }
  PP_EPI_VARIANT(PP_ACT_NONE, -2, PP_EPI_NONE, 0, 0)
  PP_EPI_VARIANT(PP_ACT_RELU, -2, PP_EPI_NONE, 0, 0)
  PP_EPI_VARIANT(PP_ACT_LEAKY, -2, PP_EPI_NONE, 0, 0)
  PP_EPI_VARIANT(PP_ACT_NONE, -2, PP_EPI_ADD_AUX1, 0, 0)
  PP_EPI_VARIANT(PP_ACT_RELU, -2, PP_EPI_ADD_AUX1_RELU, 0, 0)
  PP_EPI_VARIANT(PP_ACT_LEAKY, -2, PP_EPI_ADD_AUX1, 0, 0)
  PP_EPI_VARIANT(PP_ACT_SIGMOID, -2, PP_EPI_MUL_AUX1, 1, 0)
  PP_EPI_VARIANT(PP_ACT_TANH, -2, PP_EPI_GRU, 1, 0)
  PP_EPI_VARIANT(PP_ACT_TANH, PP_ACT_RELU, PP_EPI_NONE, 0, 0)
  PP_EPI_VARIANT(PP_ACT_TANH, PP_ACT_SIGMOID, PP_EPI_NONE, 0, 1)
  PP_EPI_VARIANT(PP_ACT_NONE, -2, PP_EPI_NONE, 0, 1)
#undef PP_EPI_VARIANT
  epilogue_lds_variant<OT, NA, NB, SCALED, -1, -1, -1, -1, -1>(p, e, wlds, lane, c_wave, row0, val);
}

// One epilogue entry for every convolution kernel: the general form when the launch's views are not vector-aligned, the
// LDS-transposed form when the work-group's LDS holds the waves' staging rows (FITS, a compile-time fact of the launcher's LDS
// size) and PP_CONV_EPI is not "direct", else the direct fast form.  `smem` = the work-group's dynamic LDS: every wave must be done
// with it (the barrier below) and no LDS-DMA copy may be in flight (vmcnt).
template <typename OT, int NA, int NB, bool FITS, bool BARRIER = true, bool SCALED = false, typename RowFn, typename ChanFn,
          typename ValFn, typename Row0Fn>
__device__ __forceinline__ void epilogue_any(const ConvK& p, const EpiCtx<OT>& e, unsigned char* smem, int wave, int lane, int c_wave,
                                             RowFn row, ChanFn chan, ValFn val, Row0Fn row0) {
  // SCALED (the PP_F32X2 kernels): val() returns RAW accumulators, to be multiplied by ConvK::acc_scale before the bias
  auto val_s = [&](auto ai, auto bi) PP_INLINE_LAMBDA {
    if constexpr (SCALED) return val(ai, bi) * p.acc_scale;
    else return val(ai, bi);
  };
  if constexpr (FITS) {
    if (!epi_fast_ok<OT>(p, e) || !p.epi_lds) {   // odd views, or PP_CONV_EPI=direct (A/B: same bits)
      epilogue_quads_general<OT, NA, NB>(p, e, row, chan, val_s);
      return;
    }
    if constexpr (BARRIER) {   // (false: the caller hands over LDS no other wave touches any more)
      pp_wait_vmcnt<0>();
      pp_barrier();
    }
    PP_EPI_STAMP(e, 0);
    epilogue_quads_lds<OT, NA, NB, SCALED>(p, e, smem + wave * epi_lds_wave_bytes<NA>(), lane, c_wave, row0, val);
  } else {
    // The GEMM / flat implicit-GEMM tiles live on <= 128 registers per wave (two 8-wave work-groups per CU) and keep the r04 fast
    // form.  r05 measured the lean variants on them three times (LDS-transposed; on the direct quads with the catch-all variant;
    // on the direct quads with four variants and one row buffer at 128 registers, 8 spills): fc1 611 -> 460 / 460 / 642, qkv 588 ->
    // 464 / 464 / 566 TF/s -- their epilogue is not what they wait for.  (tools/experiments/r05_halo_hooks.patch holds the
    // direct-quad form.)
    epilogue_quads<OT, NA, NB>(p, e, row, chan, val_s);
  }
}

// (the PP_F32X2 operand split lives in pp_device.h: split_pair)
// one 16-byte f16 MFMA fragment from LDS
__device__ __forceinline__ h8 lds_frag(const void* ptr) { return *reinterpret_cast<const h8*>(ptr); }

template <typename F>
static int launch_by_cout(void* stream, const ConvK& k, int Z) {
  // Small problems (the per-step convolutions of the two recurrences: M = 2*45*80 or 90*160 pixels) would
  // fill only a fraction of the 256 CUs with 128-pixel tiles: switch to 32-pixel tiles (4x the work-groups).
  const int64_t blocks128 = ((k.M + 127) / 128) * ((k.Cout + 127) / 128) * Z;
  // Defaults (each rule parity-checked and timed on the MI355X, profiles/r02_*): 8-wave 256-channel x 128-pixel tiles
  // for large problems whose Cout pads to 256 as cheaply as to 128 (half the pixel-tile gather per flop: clip 642 ->
  // 633 ms), 32-pixel tiles for the per-step convolutions of the recurrences and 16-pixel tiles when even those leave
  // CUs idle (-> 627 ms).  PP_CONV_TILE pins one family for tests: large | small | xlforce | tiny | classic
  // (classic = the r01 rules: no 8-wave and no 16-pixel tiles).
  const int forced = options().tile;
  const bool small = forced == 2 || ((forced == 0 || forced == 5 || forced == 6) && blocks128 < 224);
  // (... for every Cout whose padding to 256-channel tiles wastes no more than 128-channel tiles would, and at most 1/8)
  const int waste256 = (k.Cout + 255) / 256 * 256 - k.Cout;
  const bool fits256 = (k.Cout + 255) / 256 * 256 == (k.Cout + 127) / 128 * 128 && waste256 * 8 <= k.Cout;
  // (f16: from 512 128-wide work-groups on -- the transformer's 1960->512 / 512->512 projections, 864 of them, run 647 / 496
  // TF/s on the 8-wave tiles against 430 / 341 on 128 x 128 ones; the PP_F32X2 family keeps the validated 1024)
  if (fits256 && (forced == 4 || (forced == 0 && blocks128 >= F::xl_min_blocks)))
    return F::template run<4, 2, 4, 4>(stream, k, Z);                                // 256 x 128, 8 waves
  if (k.Cout > 64) {
    // 16-pixel tiles when 32-pixel tiles still give at most ~1 work-group per CU
    if ((forced == 5 || forced == 0) && blocks128 * 4 < 320) return F::template run<4, 1, 2, 1>(stream, k, Z);  // 128 x 16
    if (small) return F::template run<4, 1, 2, 2>(stream, k, Z);                     // 128 x  32
    // 96-wide tiles when they waste clearly fewer output channels than 128-wide ones (Cout 192, 576, ...)
    const int waste128 = (k.Cout + 127) / 128 * 128 - k.Cout;
    const int waste96 = (k.Cout + 95) / 96 * 96 - k.Cout;
    if (waste96 + 32 <= waste128) {                                                  //  96 x 128
      // f32 MFMA: one wave column of 96 x 32 so that the wave tile is made of 32x32 MFMA blocks
      if constexpr (F::m32_wide96) return F::template run<1, 4, 6, 2>(stream, k, Z);
      else return F::template run<2, 2, 3, 4>(stream, k, Z);
    }
    return F::template run<2, 2, 4, 4>(stream, k, Z);                                // 128 x 128
  }
  if (k.Cout > 32) {
    if (small) return F::template run<2, 2, 2, 1>(stream, k, Z);                     //  64 x  32
    return F::template run<1, 4, 4, 2>(stream, k, Z);                                //  64 x 128
  }
  if (k.Cout > 16) return F::template run<1, 4, 2, 2>(stream, k, Z);                 //  32 x 128
  return F::template run<1, 4, 1, 4>(stream, k, Z);                                  //  16 x 256
}

// flat-tile implicit-GEMM families (conv_igemm_kernel.h), one translation unit each
int launch_igemm_hh(void* stream, const ConvK& k, int Z);                 // f16 tensors, f16 output
int launch_igemm_hf(void* stream, const ConvK& k, int Z);                 // f16 tensors, f32 output
int launch_igemm_ff(void* stream, const ConvK& k, int Z);                 // f32 tensors (exact f32 MFMA products), f32 output
int launch_igemm_fh(void* stream, const ConvK& k, int Z);                 // f32 tensors, f16 output
// f32 convolution on the f16 matrix pipe (PP_F32X2): conv_split.hip
int launch_split(void* stream, const ConvK& k, int Z);
// conv_igemm.hip: parameter block -> kernel arguments with the shared argument checks
int convk_from_params(const pp_conv2d_params* p, ConvK* k, const char* who, bool virtual_input);
// f16 convolutions with few output pixels and a long reduction (conv_ksplit.hip); returns 1 when not eligible
int launch_ksplit_f16(void* stream, const ConvK& k, int Z, bool out_f16);
// ... and its halo-tile form for stride-1 multi-tap convolutions (conv_halo.hip); returns 1 when not eligible
int launch_halo_split(void* stream, const ConvK& k, int Z);
int launch_halo_f16(void* stream, const ConvK& k, int Z, bool out_f16);
int launch_halo_f16_small_cout(void* stream, const ConvK& k, int Z, bool out_f16);   // <= 16 output channels, 3x3 (PP_CONV_SMALL_HALO)
// conv_gemm_f16.hip: 1x1 / stride-1 / unpadded single-segment f16 layers (plain GEMMs); returns 1 when not eligible
int launch_gemm_f16(void* stream, const ConvK& k, int Z, bool out_f16);
// ... its patch-gather form (pp_conv2d_params.flat_taps); PP_ERR_UNSUPPORTED when the layer is not eligible
int launch_gemm_f16_patch(void* stream, const ConvK& k, int Z, bool out_f16);
// conv_direct.hip: at most 4 output channels, streaming fp32-FMA kernel; returns 1 when not eligible
int launch_direct_small_cout(void* stream, const ConvK& k, int Z, int dtype, bool out_f16);
// conv_patch.hip (r06): PP_F32X2 + flat_taps -- f32 input of at most 4 channels (RAFT's 7x7 convolutions on the flow / the frames),
// the patch matrix built as MFMA operand fragments in LDS instead of pp_im2col's tensor
int launch_patch_split(void* stream, const ConvK& k, int Z);

}  // namespace pp

// transformer_kernels.hip -- sparse spatiotemporal transformer kernels other than the GEMMs
// (which are pp_conv2d): LayerNorm, token pooling, fold / unfold of the fusion feed-forward and soft composition, and the final
// uint8 compose.  Contracts and reference call sites: include/propainter_mi355.h.
#include "pp_device.h"
#include "pp_host.h"

namespace pp {

static inline unsigned nblk3(int64_t total) { return pp_blocks_1d(total); }  // (records a >= 2^32-thread launch: pp_host.h)

// ----------------------------------------------------------------------------------------
// LayerNorm: one wave per token, 8 channels per lane (C = 512)
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) layernorm512_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int fh, int fw, int Hp,
                                                           int Wp, int64_t ntok, float eps) {
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = lane_id();
  const int64_t tk = tok < ntok ? tok : ntok - 1;  // keep the wave convergent for the shuffles
  float f[8];
  ld8(x + tk * 512 + lane * 8, f);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += shfl_xor(s, m);
  const float mean = s * (1.f / 512.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f[i] -= mean;
    q += f[i] * f[i];
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) q += shfl_xor(q, m);
  const float rstd = 1.f / sqrtf(q * (1.f / 512.f) + eps);
  if (tok >= ntok) return;
  const int64_t per = (int64_t)fh * fw;
  const int64_t t = tok / per;
  const int r = (int)(tok % per);
  const int y = r / fw, xx = r - y * fw;
  T* dst = out + ((t * Hp + y) * (int64_t)Wp + xx) * 512 + lane * 8;
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f[i] * rstd * gamma[lane * 8 + i] + beta[lane * 8 + i];
  st8(dst, o);
}

// ----------------------------------------------------------------------------------------
// depth-wise 4x4 stride-4 token pooling
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pool_tokens_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          int Hp, int Wp, int C, int ph, int pw, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over T*ph*pw*(C/8)
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  const int64_t tok = idx / pieces;
  const int j = (int)(tok % pw);
  const int64_t r = tok / pw;
  const int i = (int)(r % ph);
  const int64_t t = r / ph;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = b[pc * 8 + c];
  for (int ky = 0; ky < 4; ++ky)
    for (int kx = 0; kx < 4; ++kx) {
      float v[8];
      ld8(x + ((t * Hp + 4 * i + ky) * (int64_t)Wp + 4 * j + kx) * C + pc * 8, v);
      const float* ww = w + (ky * 4 + kx) * C + pc * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += v[c] * ww[c];
    }
  st8(out + tok * C + pc * 8, acc);
}

// ----------------------------------------------------------------------------------------
// fold (overlap-add [+average]) and unfold+GELU, kernel 7 / stride 3 / padding 3, tap-major vectors
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fold_kernel(const T* __restrict__ in, T* __restrict__ out, int H,
                                                   int W, int C, int fh, int fw, int normalize, int gelu, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over T*H*W*(C/8)
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  const int64_t pix = idx / pieces;
  const int x = (int)(pix % W);
  const int64_t r = pix / W;
  const int y = (int)(r % H);
  const int64_t t = r / H;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  // token i covers rows 3i-3 .. 3i+3 (kernel 7, stride 3, padding 3)
  const int i0 = (y - 3 < 0) ? 0 : (y - 3 + 2) / 3;
  const int i1 = ((y + 3) / 3 < fh - 1) ? (y + 3) / 3 : fh - 1;
  const int j0 = (x - 3 < 0) ? 0 : (x - 3 + 2) / 3;
  const int j1 = ((x + 3) / 3 < fw - 1) ? (x + 3) / 3 : fw - 1;
  for (int i = i0; i <= i1; ++i) {
    const int ky = y - (3 * i - 3);
    for (int j = j0; j <= j1; ++j) {
      const int kx = x - (3 * j - 3);
      float v[8];
      ld8(in + ((t * fh + i) * (int64_t)fw + j) * (49 * C) + (ky * 7 + kx) * C + pc * 8, v);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += v[c];
    }
  }
  const int cnt = (i1 - i0 + 1) * (j1 - j0 + 1);
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = (normalize && cnt > 0) ? acc[c] / (float)cnt : acc[c];
  if (gelu) {
    // r04: the exact (erf) GELU of fc2's input (sparse_transformer.py:83) applied HERE, once per folded value, instead of in
    // pp_unfold_gelu on each of its up to 9 unfolded copies (5x fewer erf evaluations; the unfold becomes a copy).  The value
    // is rounded to the storage type first -- it is the number pp_unfold_gelu used to read back -- so the result is bit-identical.
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float v = to_f32(from_f32<T>(o[c]));
      o[c] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    }
  }
  st8(out + pix * C + pc * 8, o);
}

template <typename T>
__global__ void __launch_bounds__(256) unfold_gelu_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          int H, int W, int C, int fh, int fw, int pre_activated, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over T*fh*fw*49*(C/8)
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  int64_t r = idx / pieces;
  const int tap = (int)(r % 49);
  r /= 49;
  const int j = (int)(r % fw);
  r /= fw;
  const int i = (int)(r % fh);
  const int64_t t = r / fh;
  const int y = 3 * i - 3 + tap / 7, x = 3 * j - 3 + tap % 7;
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = 0.f;
  if (y >= 0 && y < H && x >= 0 && x < W) {
    float v[8];
    ld8(in + ((t * H + y) * (int64_t)W + x) * C + pc * 8, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = pre_activated ? v[c] : 0.5f * v[c] * (1.f + erff(v[c] * 0.70710678118654752f));
  }
  st8(out + ((t * fh + i) * (int64_t)fw + j) * (49 * C) + tap * C + pc * 8, o);
}

// ----------------------------------------------------------------------------------------
// uint8 compose
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) compose_u8_kernel(const T* __restrict__ pred, int pred_ldc,
                                                         const int* __restrict__ frame_ids,
                                                         const int* __restrict__ first,
                                                         const unsigned char* __restrict__ masks,
                                                         const unsigned char* __restrict__ orig,
                                                         unsigned char* __restrict__ comp, int64_t HW, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over L*HW
  if (idx >= total) return;
  const int l = (int)(idx / HW);
  const int64_t p = idx - (int64_t)l * HW;
  const int64_t gp = (int64_t)frame_ids[l] * HW + p;
  const bool m = masks[gp] != 0;
  const bool fst = first[l] != 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    unsigned char img;
    if (m) {
      const float v = ((float)pred[idx * pred_ldc + c] + 1.f) / 2.f * 255.f;  // (pred+1)/2 * 255, then truncation
      img = (unsigned char)(int)v;
    } else {
      img = orig[gp * 3 + c];
    }
    if (fst) {
      comp[gp * 3 + c] = img;
    } else {
      const float b = (float)comp[gp * 3 + c] * 0.5f + (float)img * 0.5f;
      comp[gp * 3 + c] = (unsigned char)(int)b;
    }
  }
}

}  // namespace pp

extern "C" int32_t pp_layernorm(void* stream, const pp_layernorm_params* p) {
  using namespace pp;
  if (!p || !p->x || !p->out || !p->gamma || !p->beta) return pp_fail(PP_ERR_BAD_ARG, "pp_layernorm: null argument");
  if (p->C != 512) return pp_fail(PP_ERR_UNSUPPORTED, "pp_layernorm: C must be 512");
  const int64_t ntok = p->T * p->fh * p->fw;
  if (ntok <= 0 || p->Hp < p->fh || p->Wp < p->fw) return pp_fail(PP_ERR_BAD_ARG, "pp_layernorm: bad geometry");
#define PP_BY_DTYPE(DT, CALL)                                               \
  if ((DT) == PP_F16) {                                                     \
    typedef half_t T;                                                       \
    CALL;                                                                   \
  } else if ((DT) == PP_F32) {                                              \
    typedef float T;                                                        \
    CALL;                                                                   \
  } else {                                                                  \
    return pp_fail(PP_ERR_UNSUPPORTED, "storage dtype must be PP_F16 or PP_F32"); \
  }
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((layernorm512_kernel<T>), dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, stream,
                                  (const T*)p->x, (T*)p->out, (const float*)p->gamma, (const float*)p->beta, (int)p->fh,
                                  (int)p->fw, (int)p->Hp, (int)p->Wp, ntok, p->eps))
  return pp_check_launch("pp_layernorm");
}

extern "C" int32_t pp_pool_tokens(void* stream, const pp_pool_tokens_params* p) {
  using namespace pp;
  if (!p || !p->x || !p->out || !p->weight || !p->bias) return pp_fail(PP_ERR_BAD_ARG, "pp_pool_tokens: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_pool_tokens: C must be a multiple of 8");
  const int ph = (int)(p->Hp / 4), pw = (int)(p->Wp / 4);
  const int64_t total = p->T * ph * pw * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_pool_tokens: empty problem");
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((pool_tokens_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->x,
                                  (T*)p->out, (const float*)p->weight, (const float*)p->bias, (int)p->Hp, (int)p->Wp,
                                  (int)p->C, ph, pw, total))
  return pp_check_launch("pp_pool_tokens");
}

extern "C" int32_t pp_fold(void* stream, const pp_fold_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: C must be a multiple of 8");
  const int64_t total = p->T * p->H * p->W * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: empty problem");
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((fold_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->in, (T*)p->out,
                                  (int)p->H, (int)p->W, (int)p->C, (int)p->fh, (int)p->fw, (int)p->normalize, (int)p->gelu, total))
  return pp_check_launch("pp_fold");
}

extern "C" int32_t pp_unfold_gelu(void* stream, const pp_unfold_gelu_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: C must be a multiple of 8");
  const int64_t total = p->T * p->fh * p->fw * 49 * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: empty problem");
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((unfold_gelu_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->in,
                                  (T*)p->out, (int)p->H, (int)p->W, (int)p->C, (int)p->fh, (int)p->fw, (int)p->pre_activated, total))
  return pp_check_launch("pp_unfold_gelu");
}

extern "C" int32_t pp_compose_u8(void* stream, const pp_compose_u8_params* p) {
  using namespace pp;
  if (!p || !p->pred || !p->frame_ids || !p->first || !p->masks || !p->orig || !p->comp)
    return pp_fail(PP_ERR_BAD_ARG, "pp_compose_u8: null argument");
  const int64_t HW = p->H * p->W;
  const int64_t total = p->L * HW;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_compose_u8: empty problem");
  PP_BY_DTYPE(p->pred_dtype, PP_LAUNCH((compose_u8_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->pred,
                                       (int)p->pred_ldc, (const int*)p->frame_ids, (const int*)p->first,
                                       (const unsigned char*)p->masks, (const unsigned char*)p->orig,
                                       (unsigned char*)p->comp, HW, total))
  return pp_check_launch("pp_compose_u8");
}

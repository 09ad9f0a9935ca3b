// prop_kernels.hip -- flow-guided propagation kernels (all gather / elementwise, HBM bound).
//
// Coordinate arithmetic follows the reference's CPU path operation by operation, in fp32
// with FMA contraction disabled (build flag -ffp-contract=off), because the nearest-neighbour
// warp of image propagation rounds the sampling coordinate (round-half-even) and a one-ulp
// difference can flip a pixel:
//   flow_warp (flow_loss_utils.py:38-41):  n = 2*(x + f)/max(W-1,1) - 1
//   grid_sample, align_corners=True (ATen CPU kernel): i = (n + 1) * ((W-1)/2)
#include "pp_device.h"
#include "pp_host.h"

namespace pp {

static inline unsigned nblk2(int64_t total) { return pp_blocks_1d(total); }  // (records a >= 2^32-thread launch: pp_host.h)

__device__ __forceinline__ float warp_coord(int pos, float f, int size) {
  const float g = (float)pos + f;
  const float d = (float)(size - 1 > 1 ? size - 1 : 1);
  const float n = 2.0f * g / d - 1.0f;
  const float r = (n + 1.0f) * ((float)(size - 1) / 2.0f);
  // non-finite / absurd coordinates (NaN or Inf flows) are mapped far outside the image: every later
  // float->int conversion stays defined and the sample is simply "out of range" (zeros)
  return (fabsf(r) < 1.0e8f) ? r : -1.0e8f;
}

// bilinear sample of a C-channel fp32 pixel array (pitch ldc) with zero padding
template <int C>
__device__ __forceinline__ void bilinear_f32(const float* __restrict__ img, int ldc, int H, int W, float ix, float iy,
                                             float* out) {
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy;
  const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
#pragma unroll
  for (int c = 0; c < C; ++c) out[c] = 0.f;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  if (yin0 && xin0) {
    const float* s = img + ((int64_t)y0 * W + x0) * ldc;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] += s[c] * (wx0 * wy0);
  }
  if (yin0 && xin1) {
    const float* s = img + ((int64_t)y0 * W + x0 + 1) * ldc;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] += s[c] * (wx1 * wy0);
  }
  if (yin1 && xin0) {
    const float* s = img + ((int64_t)(y0 + 1) * W + x0) * ldc;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] += s[c] * (wx0 * wy1);
  }
  if (yin1 && xin1) {
    const float* s = img + ((int64_t)(y0 + 1) * W + x0 + 1) * ldc;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] += s[c] * (wx1 * wy1);
  }
}

// fbConsistencyCheck (propainter.py:27-36) for one pixel
__device__ __forceinline__ bool fb_valid(const float* __restrict__ flow_check, int H, int W, float fx, float fy, float ix,
                                         float iy) {
  float bw[2];
  bilinear_f32<2>(flow_check, 2, H, W, ix, iy, bw);
  const float dx = fx + bw[0], dy = fy + bw[1];
  const float mag = (fx * fx + fy * fy) + (bw[0] * bw[0] + bw[1] * bw[1]);
  return (dx * dx + dy * dy) < (0.01f * mag + 0.5f);
}

// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) img_prop_step_kernel(const float* __restrict__ f_prev,
                                                            const unsigned char* __restrict__ m_prev,
                                                            const float* __restrict__ x_cur,
                                                            const unsigned char* __restrict__ m_cur,
                                                            const float* __restrict__ flow_prop,
                                                            const float* __restrict__ flow_check,
                                                            float* __restrict__ f_new, unsigned char* __restrict__ m_new,
                                                            int H, int W, int first, int mask_input) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
  float c0 = x_cur[idx * 3], c1 = x_cur[idx * 3 + 1], c2 = x_cur[idx * 3 + 2];
  const unsigned char mc = m_cur[idx];
  if (mask_input && mc) c0 = c1 = c2 = 0.f;
  if (first) {
    f_new[idx * 3] = c0;
    f_new[idx * 3 + 1] = c1;
    f_new[idx * 3 + 2] = c2;
    m_new[idx] = mc;
    return;
  }
  const float fx = flow_prop[idx * 2], fy = flow_prop[idx * 2 + 1];
  const float ix = warp_coord(x, fx, W), iy = warp_coord(y, fy, H);
  const bool valid = fb_valid(flow_check, H, W, fx, fy, ix, iy);
  // nearest warp of the propagated frame (round half to even), zeros outside
  const float rx = nearbyintf(ix), ry = nearbyintf(iy);
  float w0 = 0.f, w1 = 0.f, w2 = 0.f;
  if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) {
    const int64_t s = ((int64_t)ry * W + (int64_t)rx) * 3;
    w0 = f_prev[s];
    w1 = f_prev[s + 1];
    w2 = f_prev[s + 2];
  }
  // bilinear warp of the propagated mask, binarised at 0.1 (propainter.py:180-183)
  float mv = 0.f;
  {
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    if (yin0 && xin0) mv += (m_prev[(int64_t)y0 * W + x0] ? 1.f : 0.f) * (wx0 * wy0);
    if (yin0 && xin1) mv += (m_prev[(int64_t)y0 * W + x0 + 1] ? 1.f : 0.f) * (wx1 * wy0);
    if (yin1 && xin0) mv += (m_prev[(int64_t)(y0 + 1) * W + x0] ? 1.f : 0.f) * (wx0 * wy1);
    if (yin1 && xin1) mv += (m_prev[(int64_t)(y0 + 1) * W + x0 + 1] ? 1.f : 0.f) * (wx1 * wy1);
  }
  const bool mvalid = mv > 0.1f;
  const bool fill = valid && !mvalid;  // flow_vaild_mask * (1 - mask_prop_valid)
  const bool u = mc && fill;
  f_new[idx * 3] = u ? w0 : c0;
  f_new[idx * 3 + 1] = u ? w1 : c1;
  f_new[idx * 3 + 2] = u ? w2 : c2;
  m_new[idx] = (mc && !fill) ? 1 : 0;
}

template <typename T>
__global__ void __launch_bounds__(256) pack_encoder_input_kernel(const float* __restrict__ frames,
                                                                 const float* __restrict__ prop,
                                                                 const unsigned char* __restrict__ m_in,
                                                                 const unsigned char* __restrict__ m_upd,
                                                                 T* __restrict__ out, float* __restrict__ updated,
                                                                 int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float m = m_in[idx] ? 1.f : 0.f;
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    v[c] = frames[idx * 3 + c] * (1.f - m) + prop[idx * 3 + c] * m;
    if (updated) updated[idx * 3 + c] = v[c];
  }
  const float o[8] = {v[0], v[1], v[2], m, m_upd[idx] ? 1.f : 0.f, 0.f, 0.f, 0.f};
  st8(out + idx * 8, o);
}

__global__ void __launch_bounds__(256) flow_down4_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                         int W, int h, int w, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over N*h*w
  if (idx >= total) return;
  const int j = (int)(idx % w);
  const int64_t t = idx / w;
  const int i = (int)(t % h);
  const int64_t n = t / h;
  const float* base = in + ((n * H + (4 * i + 1)) * (int64_t)W + (4 * j + 1)) * 2;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float v00 = base[c], v01 = base[2 + c];
    const float v10 = base[(int64_t)W * 2 + c], v11 = base[(int64_t)W * 2 + 2 + c];
    // ATen upsample_bilinear2d: h0l*(w0l*v00 + w1l*v01) + h1l*(w0l*v10 + w1l*v11), all lambdas 0.5
    const float v = 0.5f * (0.5f * v00 + 0.5f * v01) + 0.5f * (0.5f * v10 + 0.5f * v11);
    out[idx * 2 + c] = v / 4.0f;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) featprop_aux_kernel(const float* __restrict__ flow_prop,
                                                           const float* __restrict__ flow_check,
                                                           const T* __restrict__ maskpair, T* __restrict__ out,
                                                           int H, int W, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over N*H*W
  if (idx >= total) return;
  const int hw = H * W;
  const int p = (int)(idx % hw);
  const int64_t n = idx / hw;
  const int y = p / W, x = p - y * W;
  const float fx = flow_prop[idx * 2], fy = flow_prop[idx * 2 + 1];
  const float ix = warp_coord(x, fx, W), iy = warp_coord(y, fy, H);
  const bool valid = fb_valid(flow_check + n * (int64_t)hw * 2, H, W, fx, fy, ix, iy);
  const T* mp = maskpair + idx * 8;
  const float o[8] = {fx, fy, valid ? 1.f : 0.f, (float)mp[0], (float)mp[1], 0.f, 0.f, 0.f};
  st8(out + idx * 8, o);
}

template <typename T>
__global__ void __launch_bounds__(256) flow_warp_kernel(const T* __restrict__ x, int x_ldc,
                                                        const float* __restrict__ flow, T* __restrict__ out,
                                                        int out_ldc, int H, int W, int C, int64_t total) {
  // one thread per (pixel, 8-channel piece)
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  const int64_t pix = idx / pieces;
  const int hw = H * W;
  const int p = (int)(pix % hw);
  const int64_t n = pix / hw;
  const int y = p / W, xx = p - y * W;
  const float ix = warp_coord(xx, flow[pix * 2], W), iy = warp_coord(y, flow[pix * 2 + 1], H);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  const T* base = x + n * (int64_t)hw * x_ldc + pc * 8;
  auto add = [&](int yy, int xc, float wgt) {
    const T* s = base + ((int64_t)yy * W + xc) * x_ldc;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += to_f32(s[c]) * wgt;
  };
  if (yin0 && xin0) add(y0, x0, wx0 * wy0);
  if (yin0 && xin1) add(y0, x0 + 1, wx1 * wy0);
  if (yin1 && xin0) add(y0 + 1, x0, wx0 * wy1);
  if (yin1 && xin1) add(y0 + 1, x0 + 1, wx1 * wy1);
  T* dst = out + pix * out_ldc + pc * 8;
#pragma unroll
  for (int c = 0; c < 8; ++c) dst[c] = from_f32<T>(acc[c]);
}

}  // namespace pp

extern "C" int32_t pp_img_prop_step(void* stream, const pp_img_prop_step_params* p) {
  using namespace pp;
  if (!p || !p->x_cur || !p->m_cur || !p->f_new || !p->m_new) return pp_fail(PP_ERR_BAD_ARG, "pp_img_prop_step: null argument");
  if (!p->first && (!p->f_prev || !p->m_prev || !p->flow_prop || !p->flow_check))
    return pp_fail(PP_ERR_BAD_ARG, "pp_img_prop_step: null state/flow");
  const int64_t total = p->H * p->W;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_img_prop_step: empty problem");
  PP_LAUNCH(img_prop_step_kernel, dim3(nblk2(total)), dim3(256), 0, stream, (const float*)p->f_prev,
            (const unsigned char*)p->m_prev, (const float*)p->x_cur, (const unsigned char*)p->m_cur,
            (const float*)p->flow_prop, (const float*)p->flow_check, (float*)p->f_new, (unsigned char*)p->m_new,
            (int)p->H, (int)p->W, (int)p->first, (int)p->mask_input);
  return pp_check_launch("pp_img_prop_step");
}

extern "C" int32_t pp_pack_encoder_input(void* stream, const pp_pack_encoder_input_params* p) {
  using namespace pp;
  if (!p || !p->frames || !p->prop || !p->m_in || !p->m_upd || !p->out)
    return pp_fail(PP_ERR_BAD_ARG, "pp_pack_encoder_input: null argument");
  if (p->total_pixels <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_pack_encoder_input: empty problem");
  if (p->out_dtype == PP_F16) {
    PP_LAUNCH((pack_encoder_input_kernel<half_t>), dim3(nblk2(p->total_pixels)), dim3(256), 0, stream, (const float*)p->frames,
              (const float*)p->prop, (const unsigned char*)p->m_in, (const unsigned char*)p->m_upd, (half_t*)p->out,
              (float*)p->updated, p->total_pixels);
  } else if (p->out_dtype == PP_F32) {
    PP_LAUNCH((pack_encoder_input_kernel<float>), dim3(nblk2(p->total_pixels)), dim3(256), 0, stream, (const float*)p->frames,
              (const float*)p->prop, (const unsigned char*)p->m_in, (const unsigned char*)p->m_upd, (float*)p->out,
              (float*)p->updated, p->total_pixels);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_pack_encoder_input: out_dtype");
  }
  return pp_check_launch("pp_pack_encoder_input");
}

extern "C" int32_t pp_flow_down4(void* stream, const pp_flow_down4_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_down4: null argument");
  if (p->H % 4 || p->W % 4) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_down4: H and W must be multiples of 4");
  const int h = (int)(p->H / 4), w = (int)(p->W / 4);
  const int64_t total = p->N * h * w;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_down4: empty problem");
  PP_LAUNCH(flow_down4_kernel, dim3(nblk2(total)), dim3(256), 0, stream, (const float*)p->in, (float*)p->out, (int)p->H,
            (int)p->W, h, w, total);
  return pp_check_launch("pp_flow_down4");
}

extern "C" int32_t pp_featprop_aux(void* stream, const pp_featprop_aux_params* p) {
  using namespace pp;
  if (!p || !p->flow_prop || !p->flow_check || !p->maskpair || !p->out)
    return pp_fail(PP_ERR_BAD_ARG, "pp_featprop_aux: null argument");
  const int64_t total = p->N * p->H * p->W;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_featprop_aux: empty problem");
  if (p->dtype == PP_F16) {
    PP_LAUNCH((featprop_aux_kernel<half_t>), dim3(nblk2(total)), dim3(256), 0, stream, (const float*)p->flow_prop,
              (const float*)p->flow_check, (const half_t*)p->maskpair, (half_t*)p->out, (int)p->H, (int)p->W, total);
  } else if (p->dtype == PP_F32) {
    PP_LAUNCH((featprop_aux_kernel<float>), dim3(nblk2(total)), dim3(256), 0, stream, (const float*)p->flow_prop,
              (const float*)p->flow_check, (const float*)p->maskpair, (float*)p->out, (int)p->H, (int)p->W, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_featprop_aux: dtype");
  }
  return pp_check_launch("pp_featprop_aux");
}

extern "C" int32_t pp_flow_warp(void* stream, const pp_flow_warp_params* p) {
  using namespace pp;
  if (!p || !p->x || !p->flow || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_warp: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_warp: C must be a multiple of 8");
  const int64_t total = p->N * p->H * p->W * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_warp: empty problem");
  if (p->dtype == PP_F16) {
    PP_LAUNCH((flow_warp_kernel<half_t>), dim3(nblk2(total)), dim3(256), 0, stream, (const half_t*)p->x, (int)p->x_ldc,
              (const float*)p->flow, (half_t*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, total);
  } else if (p->dtype == PP_F32) {
    PP_LAUNCH((flow_warp_kernel<float>), dim3(nblk2(total)), dim3(256), 0, stream, (const float*)p->x, (int)p->x_ldc,
              (const float*)p->flow, (float*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_flow_warp: dtype");
  }
  return pp_check_launch("pp_flow_warp");
}

// io_kernels.hip -- the byte / integer plumbing either side of the networks, on the device:
// ComfyUI IMAGE -> uint8 frames (+ the outpaint canvas), MASK -> dilated binary masks, composed uint8 -> IMAGE,
// and the binary mask planes the generator derives from the dilated masks.  All of it is bit-exact integer
// work that the reference does on the host with numpy / PIL / scipy (utils/image_utils.py); HBM-bound streaming
// kernels, one element (or one 16-byte group) per lane, coalesced.
#include "pp_device.h"
#include "pp_host.h"

namespace pp {

static inline unsigned nblk(int64_t total) { return pp_blocks_1d(total); }  // (records a >= 2^32-thread launch: pp_host.h)

// convert_image_to_frames (image_utils.py:106-116): u8 = trunc(clip(v*255, 0, 255)); to_tensors + "*2-1"
// (:178-191): f = (u8/255)*2 - 1 as separate fp32 operations (the build disables FMA contraction).  The frame is
// placed at (oy, ox) inside a zero canvas [Ho][Wo] (extrapolation, :200-252; oy = ox = 0 and Ho,Wo = H,W otherwise).
__global__ void frames_from_image_kernel(const float* __restrict__ img, unsigned char* __restrict__ u8,
                                         float* __restrict__ f32, int H, int W, int Ho, int Wo, int oy, int ox,
                                         int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % 3);
  const int64_t pix = i / 3;
  const int x = (int)(pix % Wo);
  const int y = (int)((pix / Wo) % Ho);
  const int64_t t = pix / ((int64_t)Wo * Ho);
  const int sy = y - oy, sx = x - ox;
  unsigned char u = 0;
  if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
    const float v = img[((t * H + sy) * W + sx) * 3 + c] * 255.0f;
    u = (unsigned char)(int)fminf(fmaxf(v, 0.0f), 255.0f);
  }
  u8[i] = u;
  if (f32) f32[i] = ((float)u / 255.0f) * 2.0f - 1.0f;
}

// uint8 frames (already resized on the host, or any uint8 source) -> fp32 frames in [-1,1]
__global__ void frames_from_u8_kernel(const unsigned char* __restrict__ u8, float* __restrict__ f32, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  f32[i] = ((float)u8[i] / 255.0f) * 2.0f - 1.0f;
}

// handle_output (image_utils.py:276-290): float32(k) / 255.0
__global__ void image_from_u8_kernel(const unsigned char* __restrict__ u8, float* __restrict__ out, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const unsigned int v = ((const unsigned int*)u8)[i];
  f4 o;
  o[0] = (float)(v & 255u) / 255.0f;
  o[1] = (float)((v >> 8) & 255u) / 255.0f;
  o[2] = (float)((v >> 16) & 255u) / 255.0f;
  o[3] = (float)(v >> 24) / 255.0f;
  ((f4*)out)[i] = o;
}

// read_masks (image_utils.py:142-175): scipy.ndimage.binary_dilation(arr, iterations=k) with the default cross
// structuring element and border value 0 = "some non-zero pixel within L1 distance <= k" (the image is a rectangle,
// so every shortest L1 path stays inside it).  Two exact passes:
//   rowdist: d[y][x] = min |dx| <= k with in[y][x+dx] != 0, else 255
//   coldil : out[y][x] = any |dy| <= k with d[y+dy][x] <= k - |dy|
// k = 0 reduces to binary_mask(arr > 0.1) = (arr != 0) for uint8 data.  A float MASK is first taken through
// convert_mask_to_frames (:126-139): u8 = trunc(clamp(m*255, 0, 255)).
template <typename TIn>
__device__ __forceinline__ bool mask_nonzero(TIn v);
template <>
__device__ __forceinline__ bool mask_nonzero<unsigned char>(unsigned char v) {
  return v != 0;
}
template <>
__device__ __forceinline__ bool mask_nonzero<float>(float v) {
  return (int)fminf(fmaxf(v * 255.0f, 0.0f), 255.0f) != 0;
}

template <typename TIn>
__global__ void mask_rowdist_kernel(const TIn* __restrict__ in, unsigned char* __restrict__ d, int W, int k, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const TIn* row = in + (i - x);
  int best = 255;
  if (mask_nonzero<TIn>(row[x])) {
    best = 0;
  } else {
    for (int dx = 1; dx <= k; ++dx) {
      if ((x - dx >= 0 && mask_nonzero<TIn>(row[x - dx])) || (x + dx < W && mask_nonzero<TIn>(row[x + dx]))) {
        best = dx;
        break;
      }
    }
  }
  d[i] = (unsigned char)best;
}

__global__ void mask_coldil_kernel(const unsigned char* __restrict__ d, unsigned char* __restrict__ out, int H, int W, int k,
                                   int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const int y = (int)((i / W) % H);
  const unsigned char* col = d + (i - (int64_t)y * W - x) + x;  // plane base + x
  int hit = 0;
  for (int dy = -k; dy <= k && !hit; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    const int r = k - (dy < 0 ? -dy : dy);
    hit = (int)col[(int64_t)yy * W] <= r;
  }
  out[i] = (unsigned char)hit;
}

// Binary planes the generator derives from the dilated / updated masks (propainter.py:409-428):
//   maskpair[t][i][j] = (m_in[t][4i][4j], m_upd[t][4i][4j], 0...) as 8 f16 channels (nearest x1/4),
//   tokmask[t][a][b]  = MaxPool2d(7, 3, 3) of the 1/4-res m_in plane (> 0).
template <typename T>
__global__ void clip_masks_kernel(const unsigned char* __restrict__ m_in, const unsigned char* __restrict__ m_upd,
                                  T* __restrict__ maskpair, int H, int W, int h, int w, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % w);
  const int y = (int)((i / w) % h);
  const int64_t t = i / ((int64_t)w * h);
  const int64_t src = (t * H + 4 * y) * W + 4 * x;
  const float o[8] = {(float)m_in[src], (float)m_upd[src], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  st8(maskpair + i * 8, o);
}

__global__ void token_mask_kernel(const unsigned char* __restrict__ m_in, unsigned char* __restrict__ tok, int H, int W,
                                  int h, int w, int fh, int fw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i % fw);
  const int a = (int)((i / fw) % fh);
  const int64_t t = i / ((int64_t)fw * fh);
  int hit = 0;
  for (int ky = 0; ky < 7 && !hit; ++ky) {
    const int y = a * 3 - 3 + ky;
    if (y < 0 || y >= h) continue;
    for (int kx = 0; kx < 7; ++kx) {
      const int x = b * 3 - 3 + kx;
      if (x < 0 || x >= w) continue;
      if (m_in[(t * H + 4 * y) * W + 4 * x]) {
        hit = 1;
        break;
      }
    }
  }
  tok[i] = (unsigned char)hit;
}

// sparse_transformer.py:321-326: a (wh x ww) token window is "masked" iff any token mask of the window's local
// frames [g0, g0+lt) inside it is set (pad tokens count as 0).
// One 64-thread work-group per window: the lt * wh * ww mask bytes are read in parallel and OR-reduced through LDS (r02 used
// one thread per window walking them one dependent byte load at a time: 80 us per launch for 36 windows).
__global__ void __launch_bounds__(64) window_flags_kernel(const unsigned char* __restrict__ tok, int* __restrict__ flags, int g0,
                                                          int lt, int fh, int fw, int wh, int ww, int nww, int nwin) {
  __shared__ int hit;
  const int win = (int)blockIdx.x;
  const int wy = win / nww, wx = win % nww;
  if (threadIdx.x == 0) hit = 0;
  __syncthreads();
  const int per = wh * ww, total = lt * per;
  int mine = 0;
  for (int i = (int)threadIdx.x; i < total; i += 64) {
    const int f = g0 + i / per, r = i % per;
    const int y = wy * wh + r / ww, x = wx * ww + r % ww;
    if (y < fh && x < fw && tok[((int64_t)f * fh + y) * fw + x]) mine = 1;
  }
  if (mine) hit = 1;  // (benign race: every writer stores 1)
  __syncthreads();
  if (threadIdx.x == 0) flags[win] = hit;
}

}  // namespace pp

extern "C" int32_t pp_frames_from_image(void* stream, const pp_frames_from_image_params* p) {
  using namespace pp;
  if (!p || !p->out_u8 || (!p->image && !p->in_u8)) return pp_fail(PP_ERR_BAD_ARG, "pp_frames_from_image: null argument");
  if (p->image) {
    if (p->oy < 0 || p->ox < 0 || p->oy + p->H > p->Ho || p->ox + p->W > p->Wo)
      return pp_fail(PP_ERR_BAD_ARG, "pp_frames_from_image: the frame does not fit the canvas");
    const int64_t total = p->T * p->Ho * p->Wo * 3;
    if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_frames_from_image: empty problem");
    PP_LAUNCH(frames_from_image_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->image,
              (unsigned char*)p->out_u8, (float*)p->out_f32, (int)p->H, (int)p->W, (int)p->Ho, (int)p->Wo, (int)p->oy,
              (int)p->ox, total);
  } else {
    if (!p->out_f32) return pp_fail(PP_ERR_BAD_ARG, "pp_frames_from_image: uint8 input needs out_f32");
    const int64_t total = p->T * p->Ho * p->Wo * 3;
    if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_frames_from_image: empty problem");
    PP_LAUNCH(frames_from_u8_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const unsigned char*)p->in_u8,
              (float*)p->out_f32, total);
  }
  return pp_check_launch("pp_frames_from_image");
}

extern "C" int32_t pp_image_from_u8(void* stream, const pp_image_from_u8_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_image_from_u8: null argument");
  if (p->total <= 0 || p->total % 4) return pp_fail(PP_ERR_BAD_ARG, "pp_image_from_u8: element count must be a positive multiple of 4");
  PP_LAUNCH(image_from_u8_kernel, dim3(nblk(p->total / 4)), dim3(256), 0, stream, (const unsigned char*)p->in, (float*)p->out,
            p->total / 4);
  return pp_check_launch("pp_image_from_u8");
}

extern "C" int32_t pp_mask_dilate(void* stream, const pp_mask_dilate_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out || !p->scratch) return pp_fail(PP_ERR_BAD_ARG, "pp_mask_dilate: null argument");
  if (p->iterations < 0 || p->iterations > 254) return pp_fail(PP_ERR_BAD_ARG, "pp_mask_dilate: iterations out of range");
  const int64_t total = p->N * p->H * p->W;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_mask_dilate: empty problem");
  if (p->dtype == PP_F32) {
    PP_LAUNCH((mask_rowdist_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->in,
              (unsigned char*)p->scratch, (int)p->W, (int)p->iterations, total);
  } else if (p->dtype == PP_U8) {
    PP_LAUNCH((mask_rowdist_kernel<unsigned char>), dim3(nblk(total)), dim3(256), 0, stream, (const unsigned char*)p->in,
              (unsigned char*)p->scratch, (int)p->W, (int)p->iterations, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_mask_dilate: dtype");
  }
  PP_LAUNCH(mask_coldil_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const unsigned char*)p->scratch,
            (unsigned char*)p->out, (int)p->H, (int)p->W, (int)p->iterations, total);
  return pp_check_launch("pp_mask_dilate");
}

extern "C" int32_t pp_clip_masks(void* stream, const pp_clip_masks_params* p) {
  using namespace pp;
  if (!p || !p->m_in || !p->m_upd || !p->maskpair || !p->tokmask) return pp_fail(PP_ERR_BAD_ARG, "pp_clip_masks: null argument");
  if (p->H % 4 || p->W % 4) return pp_fail(PP_ERR_BAD_ARG, "pp_clip_masks: H and W must be multiples of 4");
  const int h = (int)(p->H / 4), w = (int)(p->W / 4);
  const int fh = (h + 6 - 7) / 3 + 1, fw = (w + 6 - 7) / 3 + 1;
  if (p->fh != fh || p->fw != fw) return pp_fail(PP_ERR_BAD_ARG, "pp_clip_masks: token grid does not match H, W");
  const int64_t total = p->T * h * w, ttok = p->T * fh * fw;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_clip_masks: empty problem");
  if (p->dtype == PP_F16) {
    PP_LAUNCH((clip_masks_kernel<half_t>), dim3(nblk(total)), dim3(256), 0, stream, (const unsigned char*)p->m_in,
              (const unsigned char*)p->m_upd, (half_t*)p->maskpair, (int)p->H, (int)p->W, h, w, total);
  } else if (p->dtype == PP_F32) {
    PP_LAUNCH((clip_masks_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, (const unsigned char*)p->m_in,
              (const unsigned char*)p->m_upd, (float*)p->maskpair, (int)p->H, (int)p->W, h, w, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_clip_masks: dtype");
  }
  PP_LAUNCH(token_mask_kernel, dim3(nblk(ttok)), dim3(256), 0, stream, (const unsigned char*)p->m_in,
            (unsigned char*)p->tokmask, (int)p->H, (int)p->W, h, w, fh, fw, ttok);
  return pp_check_launch("pp_clip_masks");
}

extern "C" int32_t pp_window_flags(void* stream, const pp_window_flags_params* p) {
  using namespace pp;
  if (!p || !p->tokmask || !p->flags) return pp_fail(PP_ERR_BAD_ARG, "pp_window_flags: null argument");
  if (p->wh <= 0 || p->ww <= 0 || p->lt <= 0 || p->g0 < 0 || p->g0 + p->lt > p->T)
    return pp_fail(PP_ERR_BAD_ARG, "pp_window_flags: bad window / frame range");
  const int nwh = (int)((p->fh + p->wh - 1) / p->wh), nww = (int)((p->fw + p->ww - 1) / p->ww);
  const int nwin = nwh * nww;
  PP_LAUNCH(window_flags_kernel, dim3(nwin), dim3(64), 0, stream, (const unsigned char*)p->tokmask,
            (int*)p->flags, (int)p->g0, (int)p->lt, (int)p->fh, (int)p->fw, (int)p->wh, (int)p->ww, nww, nwin);
  return pp_check_launch("pp_window_flags");
}

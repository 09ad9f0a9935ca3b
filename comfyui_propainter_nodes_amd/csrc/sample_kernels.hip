// sample_kernels.hip -- bandwidth-bound sampling kernels shared by the flow-completion and
// inpainting stages: deformable-conv column sampling, bilinear 2x upsampling, and the
// flow-completion input/output elementwise passes.  Contracts: include/propainter_mi355.h.
#include "pp_device.h"
#include "pp_host.h"
#include "pp_options.h"

namespace pp {

static inline unsigned nblk(int64_t total) { return pp_blocks_1d(total); }  // (records a >= 2^32-thread launch: pp_host.h)

// ----------------------------------------------------------------------------------------
// deformable-conv column sampling: one thread per (pixel, tap k, deformable group g)
// ----------------------------------------------------------------------------------------
struct DeformK {
  const void* x0;
  int x0_C, x0_ldc;
  const void* x1;
  int x1_C, x1_ldc;
  const float* om;
  int om_ldc;
  const float* flow;
  int flow_ldc;
  void* cols;
  int H, W, dg, cg, Cin;
  int64_t total;
  int xcd;  // work-groups in XCD-contiguous order (pp_device.h)
};

template <typename T>
__global__ void __launch_bounds__(256) deform_cols_kernel(const DeformK k) {
  const int blk = k.xcd ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int64_t idx = (int64_t)blk * 256 + threadIdx.x;
  if (idx >= k.total) return;
  const int g = (int)(idx % k.dg);
  const int64_t t = idx / k.dg;
  const int tap = (int)(t % 9);
  const int64_t pix = t / 9;
  const int hw = k.H * k.W;
  const int p = (int)(pix % hw);
  const int64_t n = pix / hw;
  const int y = p / k.W, x = p - y * k.W;
  const float* om = k.om + pix * k.om_ldc;
  float dy = om[g * 18 + 2 * tap];
  float dx = om[g * 18 + 2 * tap + 1];
  const float m = om[k.dg * 18 + g * 9 + tap];
  if (k.flow) {
    const float* f = k.flow + pix * k.flow_ldc;
    dx += f[0];
    dy += f[1];
  }
  float py = (float)(y - 1 + tap / 3) + dy;
  float px = (float)(x - 1 + tap % 3) + dx;
  if (!(fabsf(py) < 1.0e8f)) py = -1.0e8f;  // NaN / Inf offsets: out of range, float->int conversions stay defined
  if (!(fabsf(px) < 1.0e8f)) px = -1.0e8f;
  T* dst = reinterpret_cast<T*>(k.cols) + pix * (int64_t)(9 * k.Cin) + tap * k.Cin + g * k.cg;
  // source slab of this group's channels
  const int c0 = g * k.cg;
  const T* src;
  int ldc;
  if (c0 < k.x0_C) {
    src = reinterpret_cast<const T*>(k.x0) + c0;
    ldc = k.x0_ldc;
  } else {
    src = reinterpret_cast<const T*>(k.x1) + (c0 - k.x0_C);
    ldc = k.x1_ldc;
  }
  src += n * (int64_t)hw * ldc;
  const bool inside = (py > -1.f) && (py < (float)k.H) && (px > -1.f) && (px < (float)k.W);
  const float fy = floorf(py), fx = floorf(px);
  const int y0 = (int)fy, x0 = (int)fx;
  const float ly = py - fy, lx = px - fx;
  const float w00 = (1.f - ly) * (1.f - lx) * m, w01 = (1.f - ly) * lx * m;
  const float w10 = ly * (1.f - lx) * m, w11 = ly * lx * m;
  const bool y0ok = inside && y0 >= 0, y1ok = inside && (y0 + 1 <= k.H - 1);
  const bool x0ok = x0 >= 0, x1ok = (x0 + 1 <= k.W - 1);
  const T* r00 = src + ((int64_t)y0 * k.W + x0) * ldc;
  const T* r01 = r00 + ldc;
  const T* r10 = r00 + (int64_t)k.W * ldc;
  const T* r11 = r10 + ldc;
  if constexpr (sizeof(T) == 2) {
    if ((k.cg & 7) == 0) {  // 16-byte pieces (channels-last rows are 16-byte aligned: C, ldc multiples of 8)
      for (int c = 0; c < k.cg; c += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        auto acc = [&](const T* r, float wgt) {
          const h8 q = *reinterpret_cast<const h8*>(r + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += wgt * (float)q[e];
        };
        if (y0ok && x0ok) acc(r00, w00);
        if (y0ok && x1ok) acc(r01, w01);
        if (y1ok && x0ok) acc(r10, w10);
        if (y1ok && x1ok) acc(r11, w11);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sat_half(v[e]);
        *reinterpret_cast<h8*>(dst + c) = o;
      }
      return;
    }
  }
  for (int c = 0; c < k.cg; ++c) {
    float v = 0.f;
    if (y0ok && x0ok) v += w00 * to_f32(r00[c]);
    if (y0ok && x1ok) v += w01 * to_f32(r01[c]);
    if (y1ok && x0ok) v += w10 * to_f32(r10[c]);
    if (y1ok && x1ok) v += w11 * to_f32(r11[c]);
    dst[c] = from_f32<T>(v);
  }
}

// ----------------------------------------------------------------------------------------
// bilinear 2x upsampling, align_corners=True (one thread per output pixel x 8-channel piece)
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_kernel(const T* __restrict__ in, int in_ldc, T* __restrict__ out,
                                                         int out_ldc, int H, int W, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int64_t opix = idx / C;
  const int Wo = 2 * W, Ho = 2 * H;
  const int xo = (int)(opix % Wo);
  const int64_t t = opix / Wo;
  const int yo = (int)(t % Ho);
  const int64_t n = t / Ho;
  const float sy = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sy * (float)yo, fx = sx * (float)xo;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const T* base = in + n * (int64_t)H * W * in_ldc + c;
  const float v00 = to_f32(base[((int64_t)y0 * W + x0) * in_ldc]);
  const float v01 = to_f32(base[((int64_t)y0 * W + x1) * in_ldc]);
  const float v10 = to_f32(base[((int64_t)y1 * W + x0) * in_ldc]);
  const float v11 = to_f32(base[((int64_t)y1 * W + x1) * in_ldc]);
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  out[opix * out_ldc + c] = from_f32<T>(v);
}

// 8 channels per thread (C % 8 == 0): 16-byte loads / stores.  (r04: also the fp32-storage tensors -- the per-element form needs
// one thread per element, 4.7e9 for flow completion's last upsample at 160 images of 720x1280x32, and HIP truncates a launch of
// 2^32 threads or more: pp_host.h, pp_blocks_1d.)
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_h8_kernel(const T* __restrict__ in, int in_ldc,
                                                            T* __restrict__ out, int out_ldc, int H, int W,
                                                            int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over N*2H*2W*(C/8)
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  const int64_t opix = idx / pieces;
  const int Wo = 2 * W, Ho = 2 * H;
  const int xo = (int)(opix % Wo);
  const int64_t t = opix / Wo;
  const int yo = (int)(t % Ho);
  const int64_t n = t / Ho;
  const float sy = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sy * (float)yo, fx = sx * (float)xo;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const T* base = in + n * (int64_t)H * W * in_ldc + pc * 8;
  float v00[8], v01[8], v10[8], v11[8], o[8];
  ld8(base + ((int64_t)y0 * W + x0) * in_ldc, v00);
  ld8(base + ((int64_t)y0 * W + x1) * in_ldc, v01);
  ld8(base + ((int64_t)y1 * W + x0) * in_ldc, v10);
  ld8(base + ((int64_t)y1 * W + x1) * in_ldc, v11);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    o[e] = (1.f - ly) * ((1.f - lx) * v00[e] + lx * v01[e]) + ly * ((1.f - lx) * v10[e] + lx * v11[e]);
  st8(out + opix * out_ldc + pc * 8, o);
}

// r06: one thread per INPUT pixel x 8-channel piece computes the 2 x 2 output block (2i, 2i+1) x (2j, 2j+1).  With align_corners the
// even output row 2i lies between input rows i-1 and i and the odd one between i and i+1 (columns alike), so the block reads a 3 x 3
// neighbourhood: 9 loads per 4 stores instead of 16 (2.2-2.6 TB/s -> see tools/bench_upsample.py).  Every output keeps the per-pixel
// arithmetic of the kernel above -- its own y0 / y1 / ly from the same float expressions -- and only where a row / column index it
// asks for is one already loaded for its neighbour is the load shared: the same bits.
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_b4_kernel(const T* __restrict__ in, int in_ldc,
                                                            T* __restrict__ out, int out_ldc, int H, int W,
                                                            int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over N*H*W*(C/8)
  if (idx >= total) return;
  const int pieces = C / 8;
  const int pc = (int)(idx % pieces);
  const int64_t ipix = idx / pieces;
  const int j = (int)(ipix % W);
  const int64_t t = ipix / W;
  const int i = (int)(t % H);
  const int64_t n = t / H;
  const int Wo = 2 * W, Ho = 2 * H;
  const float sy = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  int y0[2], y1[2], x0[2], x1[2];
  float ly[2], lx[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float fy = sy * (float)(2 * i + a), fx = sx * (float)(2 * j + a);
    y0[a] = (int)fy;
    x0[a] = (int)fx;
    y1[a] = y0[a] + (y0[a] < H - 1 ? 1 : 0);
    x1[a] = x0[a] + (x0[a] < W - 1 ? 1 : 0);
    ly[a] = fy - (float)y0[a];
    lx[a] = fx - (float)x0[a];
  }
  // rows r = {y0[0], y1[0] (== y0[1] in exact arithmetic), y1[1]}; the rare lanes where float rounding says otherwise reload
  const int ry[3] = {y0[0], y1[0], y1[1]}, rx[3] = {x0[0], x1[0], x1[1]};
  const T* base = in + n * (int64_t)H * W * in_ldc + pc * 8;
  float v[3][3][8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) ld8(base + ((int64_t)ry[a] * W + rx[b]) * in_ldc, v[a][b]);
  const bool shared_y = y0[1] == y1[0], shared_x = x0[1] == x1[0];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float o[8];
      if (shared_y && shared_x) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = (1.f - ly[a]) * ((1.f - lx[b]) * v[a][b][e] + lx[b] * v[a][b + 1][e]) +
                 ly[a] * ((1.f - lx[b]) * v[a + 1][b][e] + lx[b] * v[a + 1][b + 1][e]);
      } else {   // (never taken for the sizes in use: kept so that the result cannot depend on a rounding of sy * yo)
        float v00[8], v01[8], v10[8], v11[8];
        ld8(base + ((int64_t)y0[a] * W + x0[b]) * in_ldc, v00);
        ld8(base + ((int64_t)y0[a] * W + x1[b]) * in_ldc, v01);
        ld8(base + ((int64_t)y1[a] * W + x0[b]) * in_ldc, v10);
        ld8(base + ((int64_t)y1[a] * W + x1[b]) * in_ldc, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = (1.f - ly[a]) * ((1.f - lx[b]) * v00[e] + lx[b] * v01[e]) + ly[a] * ((1.f - lx[b]) * v10[e] + lx[b] * v11[e]);
      }
      const int64_t opix = (n * Ho + (2 * i + a)) * Wo + (2 * j + b);
      st8(out + opix * out_ldc + pc * 8, o);
    }
  }
}

// ----------------------------------------------------------------------------------------
// flow-completion input / output passes
// ----------------------------------------------------------------------------------------
template <typename OT>
__global__ void __launch_bounds__(256) rfc_prep_kernel(const float* __restrict__ flows,
                                                       const unsigned char* __restrict__ masks,
                                                       OT* __restrict__ out, int T, int64_t HW, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [2][T][HW]
  if (idx >= total) return;
  const int64_t p = idx % HW;
  const int64_t r = idx / HW;
  const int t = (int)(r % T);
  const int d = (int)(r / T);
  const float m = masks[(int64_t)(t + d) * HW + p] ? 1.f : 0.f;
  const float* f = flows + idx * 2;
  const int to = d ? (T - 1 - t) : t;
  OT* o = out + (((int64_t)to * 2 + d) * HW + p) * 4;
  if constexpr (sizeof(OT) == 2) {
    h4 v = {sat_half(f[0] * (1.f - m)), sat_half(f[1] * (1.f - m)), (half_t)m, (half_t)0.f};
    *reinterpret_cast<h4*>(o) = v;
  } else {
    *reinterpret_cast<f4*>(o) = f4{f[0] * (1.f - m), f[1] * (1.f - m), m, 0.f};
  }
}

template <typename PT>
__global__ void __launch_bounds__(256) flow_combine_kernel(const PT* __restrict__ pred, int pred_ldc,
                                                           const float* __restrict__ flows,
                                                           const unsigned char* __restrict__ masks,
                                                           float* __restrict__ out, int T, int64_t HW, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [2][T][HW]
  if (idx >= total) return;
  const int64_t p = idx % HW;
  const int64_t r = idx / HW;
  const int t = (int)(r % T);
  const int d = (int)(r / T);
  const float m = masks[(int64_t)(t + d) * HW + p] ? 1.f : 0.f;
  const int tp = d ? (T - 1 - t) : t;
  const PT* pr = pred + (((int64_t)tp * 2 + d) * HW + p) * pred_ldc;
  const float* f = flows + idx * 2;
  out[idx * 2 + 0] = (float)pr[0] * m + f[0] * (1.f - m);
  out[idx * 2 + 1] = (float)pr[1] * m + f[1] * (1.f - m);
}

}  // namespace pp

extern "C" int32_t pp_deform_cols(void* stream, const pp_deform_cols_params* p) {
  using namespace pp;
  if (!p || !p->x0 || !p->om || !p->cols) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_cols: null argument");
  if (p->dtype != PP_F16 && p->dtype != PP_F32) return pp_fail(PP_ERR_UNSUPPORTED, "pp_deform_cols: dtype");
  DeformK k;
  k.x0 = p->x0; k.x0_C = (int)p->x0_C; k.x0_ldc = (int)p->x0_ldc;
  k.x1 = p->x1; k.x1_C = p->x1 ? (int)p->x1_C : 0; k.x1_ldc = (int)p->x1_ldc;
  k.om = (const float*)p->om; k.om_ldc = (int)p->om_ldc;
  k.flow = (const float*)p->flow; k.flow_ldc = (int)p->flow_ldc;
  k.cols = p->cols;
  k.H = (int)p->H; k.W = (int)p->W; k.dg = p->dg;
  k.Cin = k.x0_C + k.x1_C;
  if (k.dg < 1 || k.Cin % k.dg != 0) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_cols: channels not divisible by dg");
  k.cg = k.Cin / k.dg;
  if (k.x0_C % k.cg != 0) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_cols: a deformable group straddles the two inputs");
  k.total = p->N * p->H * p->W * 9 * k.dg;
  if (k.total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_deform_cols: empty problem");
  k.xcd = options().deform_xcd;
  if (p->dtype == PP_F16) {
    PP_LAUNCH((deform_cols_kernel<half_t>), dim3(nblk(k.total)), dim3(256), 0, stream, k);
  } else {
    PP_LAUNCH((deform_cols_kernel<float>), dim3(nblk(k.total)), dim3(256), 0, stream, k);
  }
  return pp_check_launch("pp_deform_cols");
}

extern "C" int32_t pp_upsample2x(void* stream, const pp_upsample2x_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_upsample2x: null argument");
  const int64_t total = p->N * 4 * p->H * p->W * p->C;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_upsample2x: empty problem");
  const bool vec8 = (p->C % 8) == 0 && (p->in_ldc % 8) == 0 && (p->out_ldc % 8) == 0 &&
                    ((reinterpret_cast<uintptr_t>(p->in) | reinterpret_cast<uintptr_t>(p->out)) & 15) == 0;
  const bool blocks4 = options().upsample_b4 != 0;   // (PP_UPSAMPLE_B4=0: one thread per output pixel, r01-r05)
  if (p->dtype == PP_F16 && vec8 && blocks4) {
    const int64_t tv = total / 32;
    PP_LAUNCH((upsample2x_b4_kernel<half_t>), dim3(nblk(tv)), dim3(256), 0, stream, (const half_t*)p->in, (int)p->in_ldc,
              (half_t*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, tv);
  } else if (p->dtype == PP_F32 && vec8 && blocks4) {
    const int64_t tv = total / 32;
    PP_LAUNCH((upsample2x_b4_kernel<float>), dim3(nblk(tv)), dim3(256), 0, stream, (const float*)p->in, (int)p->in_ldc,
              (float*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, tv);
  } else if (p->dtype == PP_F16 && vec8) {
    const int64_t tv = total / 8;
    PP_LAUNCH((upsample2x_h8_kernel<half_t>), dim3(nblk(tv)), dim3(256), 0, stream, (const half_t*)p->in, (int)p->in_ldc,
              (half_t*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, tv);
  } else if (p->dtype == PP_F32 && vec8) {
    const int64_t tv = total / 8;
    PP_LAUNCH((upsample2x_h8_kernel<float>), dim3(nblk(tv)), dim3(256), 0, stream, (const float*)p->in, (int)p->in_ldc,
              (float*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, tv);
  } else if (p->dtype == PP_F16) {
    PP_LAUNCH((upsample2x_kernel<half_t>), dim3(nblk(total)), dim3(256), 0, stream, (const half_t*)p->in,
              (int)p->in_ldc, (half_t*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, total);
  } else if (p->dtype == PP_F32) {
    PP_LAUNCH((upsample2x_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->in,
              (int)p->in_ldc, (float*)p->out, (int)p->out_ldc, (int)p->H, (int)p->W, (int)p->C, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_upsample2x: dtype");
  }
  return pp_check_launch("pp_upsample2x");
}

extern "C" int32_t pp_rfc_prep(void* stream, const pp_rfc_prep_params* p) {
  using namespace pp;
  if (!p || !p->flows || !p->masks || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_rfc_prep: null argument");
  const int64_t HW = p->H * p->W;
  const int64_t total = 2 * p->T * HW;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_rfc_prep: empty problem");
  if (p->out_dtype == PP_F16) {
    PP_LAUNCH((rfc_prep_kernel<half_t>), dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->flows,
              (const unsigned char*)p->masks, (half_t*)p->out, (int)p->T, HW, total);
  } else if (p->out_dtype == PP_F32) {
    PP_LAUNCH((rfc_prep_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->flows,
              (const unsigned char*)p->masks, (float*)p->out, (int)p->T, HW, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_rfc_prep: out_dtype");
  }
  return pp_check_launch("pp_rfc_prep");
}

extern "C" int32_t pp_flow_combine(void* stream, const pp_flow_combine_params* p) {
  using namespace pp;
  if (!p || !p->pred || !p->flows || !p->masks || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_combine: null argument");
  const int64_t HW = p->H * p->W;
  const int64_t total = 2 * p->T * HW;
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_flow_combine: empty problem");
  if (p->pred_dtype == PP_F16) {
    PP_LAUNCH((flow_combine_kernel<half_t>), dim3(nblk(total)), dim3(256), 0, stream, (const half_t*)p->pred,
              (int)p->pred_ldc, (const float*)p->flows, (const unsigned char*)p->masks, (float*)p->out, (int)p->T, HW, total);
  } else if (p->pred_dtype == PP_F32) {
    PP_LAUNCH((flow_combine_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, (const float*)p->pred,
              (int)p->pred_ldc, (const float*)p->flows, (const unsigned char*)p->masks, (float*)p->out, (int)p->T, HW, total);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_flow_combine: pred_dtype");
  }
  return pp_check_launch("pp_flow_combine");
}

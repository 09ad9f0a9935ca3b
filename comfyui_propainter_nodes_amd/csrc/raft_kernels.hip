// raft_kernels.hip -- the non-GEMM kernels of the RAFT stage (all HBM / gather bound):
// im2col for tiny-Cin convolutions, instance norm, correlation pyramid pooling, correlation
// window lookup and convex 8x upsampling.  See include/propainter_mi355.h for the contracts
// and the reference call sites.
#include "pp_device.h"
#include "pp_host.h"
#include "corr_lookup.h"

#include <type_traits>

namespace pp {

// ----------------------------------------------------------------------------------------
// im2col
// ----------------------------------------------------------------------------------------
// One thread writes 16 bytes (4 f32 / 8 f16 consecutive k of one output pixel): the pixel decode (three integer
// divisions) is shared by the group and the patch matrix leaves in whole 16-byte stores.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) im2col_kernel(const TI* __restrict__ in, int in_ldc, int N, int H, int W, int C,
                                                      int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                                                      int pad_mode, TO* __restrict__ out, int Kpad, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(TO);
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over M * Kpad / VEC
  if (idx >= total) return;
  const int groups = Kpad / VEC;
  const int k0 = (int)(idx % groups) * VEC;
  const int64_t m = idx / groups;
  const int wo = (int)(m % Wo);
  const int64_t t = m / Wo;
  const int ho = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const int y0 = ho * sh - ph, x0 = wo * sw - pw;
  const TI* img = in + (int64_t)n * H * W * in_ldc;
  const int kvalid = kh * kw * C;
  int c = k0 % C, tap = k0 / C;
  int ky = tap / kw, kx = tap - ky * kw;
  TO vals[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float v = 0.f;
    if (k0 + e < kvalid) {
      int y = y0 + ky, x = x0 + kx;
      bool ok = true;
      if (pad_mode == PP_PAD_REPLICATE) {
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
      } else {
        ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
      }
      if (ok) v = to_f32(img[((int64_t)y * W + x) * in_ldc + c]);
    }
    vals[e] = from_f32<TO>(v);
    if (++c == C) {
      c = 0;
      if (++kx == kw) {
        kx = 0;
        ++ky;
      }
    }
  }
  typedef typename std::conditional<sizeof(TO) == 4, f4, h8>::type vec_t;
  vec_t o;
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = vals[e];
  *reinterpret_cast<vec_t*>(out + m * Kpad + k0) = o;
}

// ----------------------------------------------------------------------------------------
// PP_F32X2 operand packing of an ACTIVATION matrix that plays the weight role (the all-pairs volume: both operands are
// feature maps): every 32-float chunk of a row becomes 32 f16 h = f16_rtz(v) followed by 32 f16 l = f16_rtz(v - h) (split_pair),
// the layout pp_conv2d(PP_F32X2) expects for its A operand (host: ops.split_pack_weight).  One thread = 8 values.
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split_pack_kernel(const float* __restrict__ in, half_t* __restrict__ out, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over rows*K/8
  if (i >= total8) return;
  const f4 a = *reinterpret_cast<const f4*>(in + i * 8), b = *reinterpret_cast<const f4*>(in + i * 8 + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  h8 h, l;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    h2 hh, ll;
    split_pair(v[e], v[e + 1], hh, ll);
    h[e] = hh[0];
    h[e + 1] = hh[1];
    l[e] = ll[0];
    l[e + 1] = ll[1];
  }
  const int64_t chunk = i >> 2;            // 32-value chunk = 4 octets
  const int oct = (int)(i & 3);
  half_t* base = out + chunk * 64;         // 64 halves (128 bytes) per chunk
  *reinterpret_cast<h8*>(base + oct * 8) = h;
  *reinterpret_cast<h8*>(base + 32 + oct * 8) = l;
}

// ----------------------------------------------------------------------------------------
// instance norm (channels-last fp32): partial sums in double, then fused apply
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) instnorm_partial_kernel(const float* __restrict__ x, int x_ldc, int64_t HW,
                                                               int C, int nchunks, double* __restrict__ partials) {
  // grid = (nchunks, N); thread t handles channel t % C with pixel lane t / C
  const int n = (int)blockIdx.y;
  const int chunk = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  const int lanes = 256 / C;  // >= 1 (C <= 256)
  const int c = tid % C;
  const int pl = tid / C;
  const int64_t per = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * per;
  const int64_t p1 = p0 + per < HW ? p0 + per : HW;
  double s = 0.0, ss = 0.0;
  if (pl < lanes) {
    const float* base = x + (int64_t)n * HW * x_ldc + c;
    for (int64_t p = p0 + pl; p < p1; p += lanes) {
      const double v = (double)base[p * x_ldc];
      s += v;
      ss += v * v;
    }
  }
  __shared__ double sh_s[256];
  __shared__ double sh_ss[256];
  sh_s[tid] = s;
  sh_ss[tid] = ss;
  __syncthreads();
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int l = 0; l < lanes; ++l) {
      a += sh_s[l * C + tid];
      b += sh_ss[l * C + tid];
    }
    double* dst = partials + (((int64_t)n * nchunks + chunk) * C + tid) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

__global__ void __launch_bounds__(256) instnorm_apply_kernel(const float* __restrict__ x, int x_ldc,
                                                             float* __restrict__ y, int y_ldc,
                                                             const float* __restrict__ skip, int skip_ldc, int64_t HW,
                                                             int C, int nchunks, const double* __restrict__ partials,
                                                             float eps, int relu_pre, int relu_post) {
  const int n = (int)blockIdx.y;
  const int chunk = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  __shared__ float sh_mean[256];
  __shared__ float sh_rstd[256];
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      const double* src = partials + (((int64_t)n * nchunks + k) * C + tid) * 2;
      a += src[0];
      b += src[1];
    }
    const double mean = a / (double)HW;
    double var = b / (double)HW - mean * mean;
    if (var < 0.0) var = 0.0;
    sh_mean[tid] = (float)mean;
    sh_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int lanes = 256 / C;
  const int c = tid % C;
  const int pl = tid / C;
  if (pl >= lanes) return;
  const int64_t per = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * per;
  const int64_t p1 = p0 + per < HW ? p0 + per : HW;
  const float mean = sh_mean[c], rstd = sh_rstd[c];
  for (int64_t p = p0 + pl; p < p1; p += lanes) {
    const int64_t pix = (int64_t)n * HW + p;
    float v = (x[pix * x_ldc + c] - mean) * rstd;
    if (relu_pre) v = v > 0.f ? v : 0.f;
    if (skip) v += skip[pix * skip_ldc + c];
    if (relu_post) v = v > 0.f ? v : 0.f;
    y[pix * y_ldc + c] = v;
  }
}

// 16-byte forms (C % 4 == 0, 16-byte aligned rows): a lane owns 4 consecutive channels, so a wave instruction moves
// 1 KiB instead of 256 bytes -- the scalar forms above ran at 1.4-1.9 TB/s on RAFT's feature encoder (8.8 GB read by the
// statistics pass, 17.7 GB moved by the apply pass per 80-frame clip).  Statistics still accumulate in double.
__global__ void __launch_bounds__(256) instnorm_partial4_kernel(const float* __restrict__ x, int x_ldc, int64_t HW,
                                                                int C, int nchunks, double* __restrict__ partials) {
  const int n = (int)blockIdx.y;
  const int chunk = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  const int C4 = C >> 2;
  const int lanes = 256 / C4;  // >= 4 (C <= 256)
  const int c4 = tid % C4;
  const int pl = tid / C4;
  const int64_t per = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * per;
  const int64_t p1 = p0 + per < HW ? p0 + per : HW;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
  if (pl < lanes) {
    const float* base = x + (int64_t)n * HW * x_ldc + c4 * 4;
    for (int64_t p = p0 + pl; p < p1; p += lanes) {
      const f4 v = *reinterpret_cast<const f4*>(base + p * x_ldc);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        s[e] += d;
        ss[e] += d * d;
      }
    }
  }
  // reduce the pixel lanes of every channel quad through LDS: [pl][c4][8]
  __shared__ double sh[256 * 8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sh[tid * 8 + e] = s[e];
    sh[tid * 8 + 4 + e] = ss[e];
  }
  __syncthreads();
  if (tid < C) {
    const int q = tid >> 2, e = tid & 3;
    double a = 0.0, b = 0.0;
    for (int l = 0; l < lanes; ++l) {
      a += sh[(l * C4 + q) * 8 + e];
      b += sh[(l * C4 + q) * 8 + 4 + e];
    }
    double* dst = partials + (((int64_t)n * nchunks + chunk) * C + tid) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

__global__ void __launch_bounds__(256) instnorm_apply4_kernel(const float* __restrict__ x, int x_ldc,
                                                              float* __restrict__ y, int y_ldc,
                                                              const float* __restrict__ skip, int skip_ldc, int64_t HW,
                                                              int C, int nchunks, const double* __restrict__ partials,
                                                              float eps, int relu_pre, int relu_post) {
  const int n = (int)blockIdx.y;
  const int chunk = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  __shared__ float sh_mean[256];
  __shared__ float sh_rstd[256];
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      const double* src = partials + (((int64_t)n * nchunks + k) * C + tid) * 2;
      a += src[0];
      b += src[1];
    }
    const double mean = a / (double)HW;
    double var = b / (double)HW - mean * mean;
    if (var < 0.0) var = 0.0;
    sh_mean[tid] = (float)mean;
    sh_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int lanes = 256 / C4;
  const int c = (tid % C4) * 4;
  const int pl = tid / C4;
  if (pl >= lanes) return;
  const int64_t per = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * per;
  const int64_t p1 = p0 + per < HW ? p0 + per : HW;
  const f4 mean = {sh_mean[c], sh_mean[c + 1], sh_mean[c + 2], sh_mean[c + 3]};
  const f4 rstd = {sh_rstd[c], sh_rstd[c + 1], sh_rstd[c + 2], sh_rstd[c + 3]};
  for (int64_t p = p0 + pl; p < p1; p += lanes) {
    const int64_t pix = (int64_t)n * HW + p;
    f4 v = (*reinterpret_cast<const f4*>(x + pix * x_ldc + c) - mean) * rstd;
    if (relu_pre) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (skip) v += *reinterpret_cast<const f4*>(skip + pix * skip_ldc + c);
    if (relu_post) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f4*>(y + pix * y_ldc + c) = v;
  }
}

// ----------------------------------------------------------------------------------------
// 2x2 average pooling of the correlation planes
// ----------------------------------------------------------------------------------------
// (plane_pitch / plane_off: corr_lookup.h)

__global__ void __launch_bounds__(256) avgpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int H, int W, int Ho, int Wo, int in_tiled, int out_tiled,
                                                         int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B * (output plane pitch): consecutive threads
  if (idx >= total) return;                                      // write consecutive floats in either layout
  const int64_t opitch = plane_pitch(Ho, Wo, out_tiled);
  const int64_t b = idx / opitch;
  const int r = (int)(idx - b * opitch);
  int yo, xo;
  if (out_tiled) {
    const int tw = (Wo + 7) >> 3, tile = r >> 5;
    yo = ((tile / tw) << 2) + ((r >> 3) & 3);
    xo = ((tile % tw) << 3) + (r & 7);
  } else {
    yo = r / Wo;
    xo = r - yo * Wo;
  }
  float v = 0.f;  // the padding of a tiled plane is zero
  if (yo < Ho && xo < Wo) {
    const float* src = in + b * plane_pitch(H, W, in_tiled);
    v = (src[plane_off(2 * yo, 2 * xo, W, in_tiled)] + src[plane_off(2 * yo, 2 * xo + 1, W, in_tiled)] +
         src[plane_off(2 * yo + 1, 2 * xo, W, in_tiled)] + src[plane_off(2 * yo + 1, 2 * xo + 1, W, in_tiled)]) * 0.25f;
  }
  out[idx] = v;
}

// ----------------------------------------------------------------------------------------
// correlation window lookup
// ----------------------------------------------------------------------------------------
// (LookupK, CoordEntry, the window geometry: corr_lookup.h -- shared with the fused lookup + projection kernel, corr_lookup_conv.hip)
__global__ void __launch_bounds__(256) corr_lookup_kernel(const LookupK k) {
  __shared__ __attribute__((aligned(16))) float win[4][4 * kCorrLvl];
  __shared__ __attribute__((aligned(8))) CoordEntry tab[4][2][36];
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int64_t npix = k.total / 324;
  const int64_t pix = (int64_t)blockIdx.x * 4 + wave;  // n*h*w + y*w + x
  const bool active = pix < npix;
  const int64_t pc = active ? pix : 0;
  const int hw = k.h * k.w;
  const int p = (int)(pc % hw);
  const int py = p / k.w, px = p - py * k.w;
  const float fx = k.flow[pc * k.flow_ldc + 0];
  const float fy = k.flow[pc * k.flow_ldc + 1];
  float* mywin = win[wave];
  int ox[4], oy[4];  // window origin: row of the first window row, 4-aligned column at or below the first window column
#pragma unroll
  for (int lvl = 0; lvl < 4; ++lvl) {
    const float scale = 1.f / (float)(1 << lvl);
    float bx = ((float)px + fx) * scale, by = ((float)py + fy) * scale;
    if (!(fabsf(bx) < 1.0e6f)) bx = -1.0e6f;  // NaN / Inf / absurd flow: a window far outside the plane (all zeros)
    if (!(fabsf(by) < 1.0e6f)) by = -1.0e6f;
    ox[lvl] = ((int)floorf(bx) - 5) & ~3;
    oy[lvl] = (int)floorf(by) - 5;
    const int H = k.ph[lvl], W = k.pw[lvl];
    const int tl = k.tiled[lvl];
    const int64_t pitch = plane_pitch(H, W, tl);
    const float* plane = k.pyr[lvl] + pc * pitch;
    float* lw = mywin + lvl * kCorrLvl;
    if (tl || ((W & 3) == 0 && (pitch & 3) == 0)) {
      // 16-byte pieces; a tiled plane is valid (zero) up to its padded size
      const int Hv = tl ? ((H + 3) & ~3) : H, Wv = tl ? ((W + 7) & ~7) : W;
      if (lane < kCorrRows * 4) {
        const int r = lane >> 2, q = lane & 3;
        const int y = oy[lvl] + r, x0 = ox[lvl] + 4 * q;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (active && (unsigned)y < (unsigned)Hv && (unsigned)x0 < (unsigned)Wv)
          v = *reinterpret_cast<const f4*>(plane + plane_off(y, x0, W, tl));
        *reinterpret_cast<f4*>(lw + r * kCorrCols + 4 * q) = v;
      }
    } else {
#pragma unroll
      for (int it = 0; it < kCorrLvl / 64; ++it) {
        const int idx = lane + it * 64;
        const int r = idx / kCorrCols, c = idx - r * kCorrCols;
        const int y = oy[lvl] + r, x = ox[lvl] + c;
        const bool in = active && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        lw[idx] = in ? plane[(int64_t)y * W + x] : 0.f;
      }
    }
  }
  // r04: the sample coordinates of a level depend on i (x) or j (y) alone -- 9 + 9 values per level instead of 81 pairs: the
  // first 36 lanes evaluate the reference's normalise -> unnormalise round trip (bilinear_sampler, RAFT/utils/utils.py:69-74,
  // operation by operation) once per (level, offset) into a wave-private table {window-local corner, fraction}; the 324
  // outputs then read two table entries each.  Same values, same blend arithmetic: bit-identical to evaluating the round trip
  // per output (r03: ~50 vector instructions per output, the kernel was issue-bound at 0.3 of the HBM rate).
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
    float cx = ((float)px + fx) * scale + (float)(o - 4);
    float cy = ((float)py + fy) * scale + (float)(o - 4);
    if (!(fabsf(cx) < 1.0e8f)) cx = -1.0e8f;
    if (!(fabsf(cy) < 1.0e8f)) cy = -1.0e8f;
    const float xn = 2.f * cx / (float)(W - 1) - 1.f;
    const float yn = 2.f * cy / (float)(H - 1) - 1.f;
    const float ix = ((xn + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((yn + 1.f) / 2.f) * (float)(H - 1);
    const float flx = floorf(ix), fly = floorf(iy);
    tab[wave][0][lane] = CoordEntry{(int)flx - olx, ix - flx};
    tab[wave][1][lane] = CoordEntry{(int)fly - oly, iy - fly};
  }
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int it = 0; it < (324 + 63) / 64; ++it) {
    const int ch = lane + it * 64;
    if (ch >= 324) break;
    const int lvl = ch / 81;
    const int r = ch - lvl * 81;
    const int i = r / 9, j = r - i * 9;
    const CoordEntry ex = tab[wave][0][lvl * 9 + i], ey = tab[wave][1][lvl * 9 + j];
    const int lx = ex.corner, ly = ey.corner;
    const float ax = ex.frac, ay = ey.frac;
    float v = 0.f;
    // window-local corner (every in-range coordinate falls inside the staged 12 x 16 window; anything else lies outside
    // the plane)
    if ((unsigned)lx < (unsigned)(kCorrCols - 1) && (unsigned)ly < (unsigned)(kCorrRows - 1)) {
      const float* w0 = mywin + lvl * kCorrLvl + ly * kCorrCols + lx;
      // out-of-plane corners are staged as 0: the sum below equals the reference's corner-by-corner accumulation
      v += w0[0] * (1.f - ax) * (1.f - ay);
      v += w0[1] * ax * (1.f - ay);
      v += w0[kCorrCols] * (1.f - ax) * ay;
      v += w0[kCorrCols + 1] * ax * ay;
    }
    k.out[pix * k.out_ldc + ch] = v;
  }
}

// ----------------------------------------------------------------------------------------
// convex upsampling: one 64-thread group per coarse pixel, thread = (a, b) of the 8x8 patch
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) convex_upsample_kernel(const float* __restrict__ mask, int mask_ldc,
                                                              const float* __restrict__ flow, int flow_ldc,
                                                              float* __restrict__ out, int h, int w, int64_t ncoarse) {
  const int64_t cp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cp >= ncoarse) return;
  const int ab = (int)(threadIdx.x & 63);
  const int a = ab >> 3, b = ab & 7;
  const int hw = h * w;
  const int64_t n = cp / hw;
  const int p = (int)(cp % hw);
  const int i = p / w, j = p - i * w;
  const float* m = mask + cp * mask_ldc + ab;
  float logit[9];
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    logit[t] = m[t * 64];
    mx = fmaxf(mx, logit[t]);
  }
  float den = 0.f, ux = 0.f, uy = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float e = __expf(logit[t] - mx);
    den += e;
    const int yy = i + t / 3 - 1, xx = j + t % 3 - 1;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
      const float* f = flow + (n * hw + (int64_t)yy * w + xx) * flow_ldc;
      ux += e * 8.f * f[0];
      uy += e * 8.f * f[1];
    }
  }
  const int64_t W8 = 8 * (int64_t)w;
  const int64_t o = ((n * 8 * h + (8 * i + a)) * W8 + (8 * j + b)) * 2;
  out[o] = ux / den;
  out[o + 1] = uy / den;
}

static inline unsigned blocks_for(int64_t total) { return pp_blocks_1d(total); }  // (records a >= 2^32-thread launch: pp_host.h)

}  // namespace pp

extern "C" int32_t pp_im2col(void* stream, const pp_im2col_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_im2col: null argument");
  if (p->Kpad < (int64_t)p->kh * p->kw * p->C) return pp_fail(PP_ERR_BAD_ARG, "pp_im2col: Kpad too small");
  if (p->Kpad % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_im2col: Kpad must be a multiple of 8");
  const int64_t total = p->N * p->Ho * p->Wo * (p->Kpad / (p->out_dtype == PP_F16 ? 8 : 4));  // 16-byte groups
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_im2col: empty problem");
#define PP_IM2COL(TI, TO)                                                                                         \
  PP_LAUNCH((im2col_kernel<TI, TO>), dim3(blocks_for(total)), dim3(256), 0, stream, (const TI*)p->in,            \
            (int)p->in_ldc, (int)p->N, (int)p->H, (int)p->W, (int)p->C, (int)p->Ho, (int)p->Wo, p->kh, p->kw,     \
            p->sh, p->sw, p->ph, p->pw, p->pad_mode, (TO*)p->out, (int)p->Kpad, total)
  if (p->dtype == PP_F32 && p->out_dtype == PP_F32) {
    PP_IM2COL(float, float);
  } else if (p->dtype == PP_F32 && p->out_dtype == PP_F16) {
    PP_IM2COL(float, half_t);
  } else if (p->dtype == PP_F16 && p->out_dtype == PP_F16) {
    PP_IM2COL(half_t, half_t);
  } else {
    return pp_fail(PP_ERR_UNSUPPORTED, "pp_im2col: dtype combination");
  }
#undef PP_IM2COL
  return pp_check_launch("pp_im2col");
}

extern "C" int32_t pp_split_pack(void* stream, const pp_split_pack_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_split_pack: null argument");
  if (p->rows <= 0 || p->K <= 0 || p->K % 32) return pp_fail(PP_ERR_BAD_ARG, "pp_split_pack: K must be a positive multiple of 32");
  const int64_t total8 = p->rows * p->K / 8;
  PP_LAUNCH(split_pack_kernel, dim3(blocks_for(total8)), dim3(256), 0, stream, (const float*)p->in, (half_t*)p->out, total8);
  return pp_check_launch("pp_split_pack");
}

extern "C" int32_t pp_instnorm(void* stream, const pp_instnorm_params* p) {
  using namespace pp;
  if (!p || !p->x || !p->y || !p->partials) return pp_fail(PP_ERR_BAD_ARG, "pp_instnorm: null argument");
  if (p->C < 1 || p->C > 256) return pp_fail(PP_ERR_UNSUPPORTED, "pp_instnorm: C must be in [1,256]");
  if (p->nchunks < 1 || p->nchunks > 65535 || p->N < 1 || p->N > 65535)
    return pp_fail(PP_ERR_BAD_ARG, "pp_instnorm: bad nchunks/N");
  dim3 grid((unsigned)p->nchunks, (unsigned)p->N);
  auto al16 = [](const void* q, int64_t ldc) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ldc & 3) == 0; };
  const bool vec = (p->C & 3) == 0 && al16(p->x, p->x_ldc) && al16(p->y, p->y_ldc) && (!p->skip || al16(p->skip, p->skip_ldc));
  if (vec) {
    PP_LAUNCH(instnorm_partial4_kernel, grid, dim3(256), 0, stream, (const float*)p->x, (int)p->x_ldc, p->HW, (int)p->C,
              (int)p->nchunks, (double*)p->partials);
  } else {
    PP_LAUNCH(instnorm_partial_kernel, grid, dim3(256), 0, stream, (const float*)p->x, (int)p->x_ldc, p->HW, (int)p->C,
              (int)p->nchunks, (double*)p->partials);
  }
  int rc = pp_check_launch("pp_instnorm(partial)");
  if (rc) return rc;
  if (vec) {
    PP_LAUNCH(instnorm_apply4_kernel, grid, dim3(256), 0, stream, (const float*)p->x, (int)p->x_ldc, (float*)p->y,
              (int)p->y_ldc, (const float*)p->skip, (int)p->skip_ldc, p->HW, (int)p->C, (int)p->nchunks,
              (const double*)p->partials, p->eps, p->relu_pre, p->relu_post);
  } else {
    PP_LAUNCH(instnorm_apply_kernel, grid, dim3(256), 0, stream, (const float*)p->x, (int)p->x_ldc, (float*)p->y,
              (int)p->y_ldc, (const float*)p->skip, (int)p->skip_ldc, p->HW, (int)p->C, (int)p->nchunks,
              (const double*)p->partials, p->eps, p->relu_pre, p->relu_post);
  }
  return pp_check_launch("pp_instnorm(apply)");
}

extern "C" int32_t pp_avgpool2x2(void* stream, const pp_avgpool2x2_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_avgpool2x2: null argument");
  const int Ho = (int)(p->H / 2), Wo = (int)(p->W / 2);
  if (Ho <= 0 || Wo <= 0 || p->B <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_avgpool2x2: empty problem");
  const int64_t opitch = p->out_tiled ? (int64_t)((Ho + 3) / 4) * ((Wo + 7) / 8) * 32 : (int64_t)Ho * Wo;
  const int64_t total = p->B * opitch;
  PP_LAUNCH(avgpool2x2_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, (const float*)p->in, (float*)p->out,
            (int)p->H, (int)p->W, Ho, Wo, (int)p->in_tiled, (int)p->out_tiled, total);
  return pp_check_launch("pp_avgpool2x2");
}

extern "C" int32_t pp_corr_lookup(void* stream, const pp_corr_lookup_params* p) {
  using namespace pp;
  if (!p || !p->flow || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup: null argument");
  LookupK k;
  for (int l = 0; l < 4; ++l) {
    if (!p->pyr[l] || p->ph[l] < 2 || p->pw[l] < 2)
      return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup: every pyramid level needs >= 2 rows and columns (H,W >= 128)");
    k.pyr[l] = (const float*)p->pyr[l];
    k.ph[l] = (int)p->ph[l];
    k.pw[l] = (int)p->pw[l];
    k.tiled[l] = (int)p->tiled[l];
  }
  k.flow = (const float*)p->flow;
  k.flow_ldc = (int)p->flow_ldc;
  k.out = (float*)p->out;
  k.out_ldc = (int)p->out_ldc;
  k.h = (int)p->h;
  k.w = (int)p->w;
  k.total = p->N * p->h * p->w * 324;
  if (k.total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_corr_lookup: empty problem");
  PP_LAUNCH(corr_lookup_kernel, dim3((unsigned)((k.total / 324 + 3) / 4)), dim3(256), 0, stream, k);
  return pp_check_launch("pp_corr_lookup");
}

extern "C" int32_t pp_convex_upsample(void* stream, const pp_convex_upsample_params* p) {
  using namespace pp;
  if (!p || !p->mask || !p->flow || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_convex_upsample: null argument");
  const int64_t ncoarse = p->N * p->h * p->w;
  if (ncoarse <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_convex_upsample: empty problem");
  PP_LAUNCH(convex_upsample_kernel, dim3((unsigned)((ncoarse + 3) / 4)), dim3(256), 0, stream, (const float*)p->mask,
            (int)p->mask_ldc, (const float*)p->flow, (int)p->flow_ldc, (float*)p->out, (int)p->h, (int)p->w, ncoarse);
  return pp_check_launch("pp_convex_upsample");
}

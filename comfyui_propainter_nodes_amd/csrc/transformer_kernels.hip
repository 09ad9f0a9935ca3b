// transformer_kernels.hip -- sparse spatiotemporal transformer kernels other than the GEMMs
// (which are pp_conv2d): LayerNorm, token pooling, the fused sparse window attention (MFMA,
// flash-style), fold / unfold of the fusion feed-forward and soft composition, and the final
// uint8 compose.  Contracts and reference call sites: include/propainter_mi355.h.
#include "pp_device.h"
#include "pp_host.h"

namespace pp {

static inline unsigned nblk3(int64_t total) { return (unsigned)((total + 255) / 256); }

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
// sparse window attention
//
// block = (128-query tile | frame, head, window), 4 waves x 32 queries (two groups of 16).  Per 32-key tile:
//   S^T = K . Q^T   (A = K tile from LDS, B = Q fragments in registers)  -> lane holds the scores of
//                    ONE query per group (col = lane&15) against keys {4g+r, 16+4g+r}, g = lane>>4
//   online softmax   lane-local + two xor-shuffles (16, 32) across the 4 lane groups
//   O^T += V^T . P^T (A = V^T tile from LDS in the same permuted key order, B = P in registers)
// so the probabilities never leave registers and the per-query statistics stay lane-local.  Every K / V^T fragment read
// from LDS feeds the MFMAs of BOTH query groups (r01: one group per wave, 64-query blocks: twice the fragment reads and
// twice the key gathers per MFMA).
// V^T tile: element (d, key) lives at row d, 8-byte key block ((key >> 2) ^ ((d >> 4) & 7)).  The transposing stores of a
// staged V row (one key, 16 consecutive d per thread, 8 threads per key) then go to 8 different bank groups; un-swizzled,
// the 8 threads of a key sit 16 rows = 1280 bytes apart, i.e. on ONE bank: an 8-way conflict on each of the 16 scalar
// stores per thread and tile (r01).  The fragment reads use one row block (dt) per instruction, so the XOR is uniform
// across a read's lanes and they stay conflict-free.
// ----------------------------------------------------------------------------------------
template <typename T>
struct AttnK {
  const T* qkv;
  const T* pkv;
  const int* win_masked;
  const int* t_ind;
  T* out;
  int t, nt, Hp, Wp, fh, fw, npool, nww;
  float scale;
  signed char nb[148 * 2];
};

constexpr int kWinH = 5, kWinW = 9, kWinTok = 45, kHeads = 4, kHeadDim = 128, kDim = 512;
constexpr int kKP = kHeadDim + 8;  // K tile row pitch (halves)
constexpr int kVP = 32 + 8;        // V^T tile row pitch (halves)
constexpr int kQG = 2;             // 16-query groups per wave
constexpr int kQBlock = 4 * kQG * 16;

template <typename T>
__global__ void __launch_bounds__(256) window_attention_kernel(const AttnK<T> k) {
  __shared__ __attribute__((aligned(16))) half_t Ks[32 * kKP];
  __shared__ __attribute__((aligned(16))) half_t Vt[kHeadDim * kVP];

  const int win = (int)blockIdx.z;
  const int head = (int)blockIdx.y;
  const int wi = win / k.nww, wj = win - wi * k.nww;
  const int r0 = wi * kWinH, c0 = wj * kWinW;
  const bool masked = k.win_masked[win] != 0;
  const int per_frame = kWinTok + 148 + k.npool;
  int nq, nk, qbase, frame = 0;
  if (masked) {
    nq = k.t * kWinTok;
    qbase = (int)blockIdx.x * kQBlock;
    if (qbase >= nq) return;
    nk = k.nt * per_frame;
  } else {
    frame = (int)blockIdx.x;
    nq = kWinTok;
    qbase = 0;
    nk = kWinTok;
  }
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int qcol = lane & 15, g = lane >> 4;

  // ---- this lane's queries (one per group) -------------------------------------------------
  bool qvalid[kQG];
  int qt[kQG], qy[kQG], qx[kQG];
  h8 qf[kQG][4];
#pragma unroll
  for (int qi = 0; qi < kQG; ++qi) {
    const int ql = qbase + (wave * kQG + qi) * 16 + qcol;
    qvalid[qi] = ql < nq;
    const int qc = qvalid[qi] ? ql : 0;
    qt[qi] = masked ? qc / kWinTok : frame;
    const int qpos = masked ? qc - qt[qi] * kWinTok : qc;
    qy[qi] = r0 + qpos / kWinW;
    qx[qi] = c0 + qpos % kWinW;
    const T* qptr = k.qkv + ((int64_t)(qt[qi] * k.Hp + qy[qi]) * k.Wp + qx[qi]) * (3 * kDim) + head * kHeadDim;
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) qf[qi][dc] = ld8h(qptr + dc * 32 + g * 8);  // (fp32 storage: f16 MFMA operands)
  }

  f4 o[kQG][8];
  float m_run[kQG], l_run[kQG];
#pragma unroll
  for (int qi = 0; qi < kQG; ++qi) {
    m_run[qi] = -1.0e30f;
    l_run[qi] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[qi][i] = f4{0.f, 0.f, 0.f, 0.f};
  }
  // a wave whose 32 queries are all past the end still takes part in the staging and the barriers
  const bool wave_live = qbase + wave * kQG * 16 < nq;

  const int kr = tid >> 3;           // key row staged by this thread
  const int dbase = (tid & 7) * 16;  // 16 head-dim values
  const int vblk = ((kr >> 2) ^ (tid & 7)) * 4 + (kr & 3);  // swizzled key slot of this thread's V^T stores ((d>>4)&7 == tid&7)

  // K / V of a 32-key tile travel global -> registers -> LDS; the loads of tile kt+32 are issued right after the barrier
  // that publishes tile kt, so their latency hides behind the 32 MFMAs and the softmax of tile kt (r02: the loop used to
  // load, store and only then compute; 267 -> 256 us.  Also measured and NOT kept: skipping the O rescale when no running
  // maximum moved (wave-uniform branch) together with a per-window LDS table of the key offsets: 272 us).
  h8 kv0, kv1, vv0, vv1;
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const int kid = kt + kr;
#pragma unroll
    for (int e = 0; e < 8; ++e) kv0[e] = kv1[e] = vv0[e] = vv1[e] = (half_t)0.f;
    if (kid < nk) {
      const T *kp, *vp;
      int fr, r;
      if (masked) {
        const int fi = kid / per_frame;
        r = kid - fi * per_frame;
        fr = k.t_ind[fi];
      } else {
        fr = frame;
        r = kid;
      }
      if (r < kWinTok + 148) {
        int y, x;
        if (r < kWinTok) {
          y = r0 + r / kWinW;
          x = c0 + r % kWinW;
        } else {
          const int ni = r - kWinTok;
          y = (r0 + (int)k.nb[2 * ni] + k.Hp) % k.Hp;
          x = (c0 + (int)k.nb[2 * ni + 1] + k.Wp) % k.Wp;
        }
        const T* tokp = k.qkv + ((int64_t)(fr * k.Hp + y) * k.Wp + x) * (3 * kDim) + head * kHeadDim;
        kp = tokp + kDim;
        vp = tokp + 2 * kDim;
      } else {
        const T* tokp = k.pkv + ((int64_t)fr * k.npool + (r - kWinTok - 148)) * (2 * kDim) + head * kHeadDim;
        kp = tokp;
        vp = tokp + kDim;
      }
      kv0 = ld8h(kp + dbase);
      kv1 = ld8h(kp + dbase + 8);
      vv0 = ld8h(vp + dbase);
      vv1 = ld8h(vp + dbase + 8);
    }
  };
  load_tile(0);
  for (int kt = 0; kt < nk; kt += 32) {
    // ---- stage K [32][128] and V^T [128][32] (loaded during the previous tile) ---------------
    *reinterpret_cast<h8*>(Ks + kr * kKP + dbase) = kv0;
    *reinterpret_cast<h8*>(Ks + kr * kKP + dbase + 8) = kv1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Vt[(dbase + e) * kVP + vblk] = vv0[e];
      Vt[(dbase + 8 + e) * kVP + vblk] = vv1[e];
    }
    __syncthreads();
    if (kt + 32 < nk) load_tile(kt + 32);

    if (wave_live) {
      // ---- S^T = K . Q^T ---------------------------------------------------------------------
      f4 s0[kQG], s1[kQG];
#pragma unroll
      for (int qi = 0; qi < kQG; ++qi) s0[qi] = s1[qi] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dc = 0; dc < 4; ++dc) {
        const h8 a0 = *reinterpret_cast<const h8*>(Ks + qcol * kKP + dc * 32 + g * 8);
        const h8 a1 = *reinterpret_cast<const h8*>(Ks + (16 + qcol) * kKP + dc * 32 + g * 8);
#pragma unroll
        for (int qi = 0; qi < kQG; ++qi) {
          s0[qi] = mfma_16x16x32_f16(a0, qf[qi][dc], s0[qi]);
          s1[qi] = mfma_16x16x32_f16(a1, qf[qi][dc], s1[qi]);
        }
      }
      h8 pf[kQG];
      float alpha[kQG];
#pragma unroll
      for (int qi = 0; qi < kQG; ++qi) {
        float sc[8];
        float mt = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[r] = (kt + 4 * g + r < nk) ? s0[qi][r] * k.scale : -1.0e30f;
          sc[4 + r] = (kt + 16 + 4 * g + r < nk) ? s1[qi][r] * k.scale : -1.0e30f;
          mt = fmaxf(mt, fmaxf(sc[r], sc[4 + r]));
        }
        mt = fmaxf(mt, shfl_xor(mt, 16));
        mt = fmaxf(mt, shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run[qi], mt);
        alpha[qi] = __expf(m_run[qi] - m_new);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float pv = __expf(sc[i] - m_new);
          ps += pv;
          pf[qi][i] = (half_t)pv;
        }
        ps += shfl_xor(ps, 16);
        ps += shfl_xor(ps, 32);
        l_run[qi] = l_run[qi] * alpha[qi] + ps;
        m_run[qi] = m_new;
      }
      // ---- O^T = alpha * O^T + V^T . P^T -----------------------------------------------------
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const half_t* vrow = Vt + (dt * 16 + qcol) * kVP;  // rows d = dt*16 + qcol: (d >> 4) & 7 == dt
        const h4 lo = *reinterpret_cast<const h4*>(vrow + 4 * (g ^ dt));
        const h4 hi = *reinterpret_cast<const h4*>(vrow + 4 * ((4 + g) ^ dt));
        const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int qi = 0; qi < kQG; ++qi) {
          o[qi][dt] = o[qi][dt] * alpha[qi];
          o[qi][dt] = mfma_16x16x32_f16(a, pf[qi], o[qi][dt]);
        }
      }
    }
    __syncthreads();
  }

  // ---- normalise and scatter back to the unpadded token grid -------------------------------
#pragma unroll
  for (int qi = 0; qi < kQG; ++qi) {
    if (!qvalid[qi] || qy[qi] >= k.fh || qx[qi] >= k.fw) continue;
    const float inv = 1.f / l_run[qi];
    T* dst = k.out + ((int64_t)(qt[qi] * k.fh + qy[qi]) * k.fw + qx[qi]) * kDim + head * kHeadDim;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      if constexpr (sizeof(T) == 2) {
        h4 v = {(half_t)(o[qi][dt][0] * inv), (half_t)(o[qi][dt][1] * inv), (half_t)(o[qi][dt][2] * inv),
                (half_t)(o[qi][dt][3] * inv)};
        *reinterpret_cast<h4*>(dst + dt * 16 + 4 * g) = v;
      } else {
        *reinterpret_cast<f4*>(dst + dt * 16 + 4 * g) = o[qi][dt] * inv;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// fold (overlap-add [+average]) and unfold+GELU, kernel 7 / stride 3 / padding 3, tap-major vectors
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fold_kernel(const T* __restrict__ in, T* __restrict__ out, int H,
                                                   int W, int C, int fh, int fw, int normalize, int64_t total) {
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
  st8(out + pix * C + pc * 8, o);
}

template <typename T>
__global__ void __launch_bounds__(256) unfold_gelu_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          int H, int W, int C, int fh, int fw, int64_t total) {
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
    for (int c = 0; c < 8; ++c) o[c] = 0.5f * v[c] * (1.f + erff(v[c] * 0.70710678118654752f));
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

template <typename T>
static int launch_window_attention(void* stream, const pp_window_attention_params* p) {
  using namespace pp;
  AttnK<T> k;
  k.qkv = (const T*)p->qkv;
  k.pkv = (const T*)p->pkv;
  k.win_masked = (const int*)p->win_masked;
  k.t_ind = (const int*)p->t_ind;
  k.out = (T*)p->out;
  k.t = (int)p->t; k.nt = (int)p->nt; k.Hp = (int)p->Hp; k.Wp = (int)p->Wp; k.fh = (int)p->fh; k.fw = (int)p->fw;
  k.npool = (int)p->npool;
  k.nww = k.Wp / kWinW;
  k.scale = p->scale;
  // rolled-neighbour offsets relative to the window origin (sparse_transformer.py:184-197, 229-283):
  // rows {-3..1} u {3..7}, cols {-5..3} u {5..13}, minus the window's own 5x9 footprint -> 148 tokens
  int n = 0;
  const int eh = (kWinH + 1) / 2, ew = (kWinW + 1) / 2;
  for (int ri = 0; ri < 2 * kWinH; ++ri) {
    const int dr = ri < kWinH ? ri - eh : ri - kWinH + eh;
    for (int ci = 0; ci < 2 * kWinW; ++ci) {
      const int dc = ci < kWinW ? ci - ew : ci - kWinW + ew;
      if (dr >= 0 && dr < kWinH && dc >= 0 && dc < kWinW) continue;
      k.nb[2 * n] = (signed char)dr;
      k.nb[2 * n + 1] = (signed char)dc;
      ++n;
    }
  }
  if (n != 148) return pp_fail(PP_ERR_LAUNCH, "pp_window_attention: internal neighbour table error");
  const int nwin = (k.Hp / kWinH) * k.nww;
  dim3 grid((unsigned)k.t, kHeads, (unsigned)nwin);
  PP_LAUNCH((window_attention_kernel<T>), grid, dim3(256), 0, stream, k);
  return pp_check_launch("pp_window_attention");
}

extern "C" int32_t pp_window_attention(void* stream, const pp_window_attention_params* p) {
  using namespace pp;
  if (!p || !p->qkv || !p->pkv || !p->win_masked || !p->t_ind || !p->out)
    return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: null argument");
  if (p->Hp % kWinH || p->Wp % kWinW) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: grid not padded to 5x9 windows");
  if (p->t < 1 || p->nt < 1 || p->t > 65535) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: bad t / nt");
  if (p->dtype == PP_F16) return launch_window_attention<half_t>(stream, p);
  if (p->dtype == PP_F32) return launch_window_attention<float>(stream, p);
  return pp_fail(PP_ERR_UNSUPPORTED, "pp_window_attention: dtype");
}

extern "C" int32_t pp_fold(void* stream, const pp_fold_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: C must be a multiple of 8");
  const int64_t total = p->T * p->H * p->W * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_fold: empty problem");
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((fold_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->in, (T*)p->out,
                                  (int)p->H, (int)p->W, (int)p->C, (int)p->fh, (int)p->fw, (int)p->normalize, total))
  return pp_check_launch("pp_fold");
}

extern "C" int32_t pp_unfold_gelu(void* stream, const pp_unfold_gelu_params* p) {
  using namespace pp;
  if (!p || !p->in || !p->out) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: null argument");
  if (p->C % 8) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: C must be a multiple of 8");
  const int64_t total = p->T * p->fh * p->fw * 49 * (p->C / 8);
  if (total <= 0) return pp_fail(PP_ERR_BAD_ARG, "pp_unfold_gelu: empty problem");
  PP_BY_DTYPE(p->dtype, PP_LAUNCH((unfold_gelu_kernel<T>), dim3(nblk3(total)), dim3(256), 0, stream, (const T*)p->in,
                                  (T*)p->out, (int)p->H, (int)p->W, (int)p->C, (int)p->fh, (int)p->fw, total))
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

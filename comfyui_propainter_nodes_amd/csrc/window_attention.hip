// window_attention.hip -- the core of SparseWindowAttention.forward (sparse_transformer.py:218-385), flash-style on MFMA.
// Contract: include/propainter_mi355.h (pp_window_attention).  Two kernels:
//   window_attention_f16_kernel      f16 storage (the node's fp16 "enable"): K / V tiles by LDS-DMA, double-buffered,
//                                    32x32x16 MFMAs, V consumed through the transposing LDS read;
//   window_attention_generic_kernel  fp32 storage (fp16 "disable"): register-staged tiles converted to f16 MFMA operands,
//                                    16x16x32 MFMAs (the round-1/2 kernel); also f16 storage beyond the f16 kernel's
//                                    limits (> 1024 key frames, frames of >= 2^30 elements).
#include "attn_device.h"
#include "pp_host.h"

#include <stdio.h>
#include <type_traits>

namespace pp {

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
__global__ void __launch_bounds__(256) window_attention_generic_kernel(const AttnK<T> k) {
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
// fp32 kernel (r06, ABI v11 `exact`): the reference's fp32 attention (sparse_transformer.py:366-393 under fp16 "disable") with
// NOTHING rounded to f16 -- q, k, v and the probabilities stay fp32, both products run on v_mfma_f32_16x16x4_f32, the
// exponentials are libm's expf.  Same work decomposition and the same key order as the generic kernel above (block =
// 128-query tile | frame, head, window; 32-key tiles; online softmax per lane with two xor-shuffles), so the two differ only
// in operand precision.  An opt-in for comparisons at fp32 level: ~5x the time of the f16-operand kernel.
//   S^T = K . Q^T : A = K[key = lane & 15][d = 4 s + g], B = q[query = lane & 15][d = 4 s + g], s = 0..31  (g = lane >> 4)
//                   -> lane holds the scores of query (lane & 15) against keys 4 g + r, r = 0..3, of a 16-key block
//   O^T += V^T P^T: step r of a 16-key block contracts keys {4 g + r}: A = V[key 4 g + r][d = 16 dt + (lane & 15)],
//                   B = the lane's own probability r -- the probabilities never leave registers
// LDS rows have a pitch of 132 floats (132 mod 64 = 4): both fragment reads touch 64 different banks.
// ----------------------------------------------------------------------------------------
constexpr int kXP = kHeadDim + 4;

__global__ void __launch_bounds__(256) window_attention_f32_exact_kernel(const AttnK<float> k) {
  __shared__ __attribute__((aligned(16))) float Ks[32 * kXP];
  __shared__ __attribute__((aligned(16))) float Vs[32 * kXP];

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

  bool qvalid[kQG];
  int qt[kQG], qy[kQG], qx[kQG];
  float qf[kQG][32];   // q[d = 4 s + g]
#pragma unroll
  for (int qi = 0; qi < kQG; ++qi) {
    const int ql = qbase + (wave * kQG + qi) * 16 + qcol;
    qvalid[qi] = ql < nq;
    const int qc = qvalid[qi] ? ql : 0;
    qt[qi] = masked ? qc / kWinTok : frame;
    const int qpos = masked ? qc - qt[qi] * kWinTok : qc;
    qy[qi] = r0 + qpos / kWinW;
    qx[qi] = c0 + qpos % kWinW;
    const float* qptr = k.qkv + ((int64_t)(qt[qi] * k.Hp + qy[qi]) * k.Wp + qx[qi]) * (3 * kDim) + head * kHeadDim;
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[qi][s] = qptr[4 * s + g];
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
  const bool wave_live = qbase + wave * kQG * 16 < nq;

  const int kr = tid >> 3;           // key row staged by this thread
  const int dbase = (tid & 7) * 16;  // 16 head-dim values
  f4 kst[4], vst[4];
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const int kid = kt + kr;
#pragma unroll
    for (int e = 0; e < 4; ++e) kst[e] = vst[e] = f4{0.f, 0.f, 0.f, 0.f};
    if (kid < nk) {
      const float *kp, *vp;
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
        const float* tokp = k.qkv + ((int64_t)(fr * k.Hp + y) * k.Wp + x) * (3 * kDim) + head * kHeadDim;
        kp = tokp + kDim;
        vp = tokp + 2 * kDim;
      } else {
        const float* tokp = k.pkv + ((int64_t)fr * k.npool + (r - kWinTok - 148)) * (2 * kDim) + head * kHeadDim;
        kp = tokp;
        vp = tokp + kDim;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kst[e] = *reinterpret_cast<const f4*>(kp + dbase + 4 * e);
        vst[e] = *reinterpret_cast<const f4*>(vp + dbase + 4 * e);
      }
    }
  };
  load_tile(0);
  for (int kt = 0; kt < nk; kt += 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<f4*>(Ks + kr * kXP + dbase + 4 * e) = kst[e];
      *reinterpret_cast<f4*>(Vs + kr * kXP + dbase + 4 * e) = vst[e];
    }
    __syncthreads();
    if (kt + 32 < nk) load_tile(kt + 32);

    if (wave_live) {
      f4 sacc[kQG][2];
#pragma unroll
      for (int qi = 0; qi < kQG; ++qi) sacc[qi][0] = sacc[qi][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float a0 = Ks[qcol * kXP + 4 * s + g];
        const float a1 = Ks[(16 + qcol) * kXP + 4 * s + g];
#pragma unroll
        for (int qi = 0; qi < kQG; ++qi) {
          sacc[qi][0] = mfma_16x16x4_f32(a0, qf[qi][s], sacc[qi][0]);
          sacc[qi][1] = mfma_16x16x4_f32(a1, qf[qi][s], sacc[qi][1]);
        }
      }
      float pr[kQG][8], alpha[kQG];
#pragma unroll
      for (int qi = 0; qi < kQG; ++qi) {
        float sc[8];
        float mt = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[r] = (kt + 4 * g + r < nk) ? sacc[qi][0][r] * k.scale : -1.0e30f;
          sc[4 + r] = (kt + 16 + 4 * g + r < nk) ? sacc[qi][1][r] * k.scale : -1.0e30f;
          mt = fmaxf(mt, fmaxf(sc[r], sc[4 + r]));
        }
        mt = fmaxf(mt, shfl_xor(mt, 16));
        mt = fmaxf(mt, shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run[qi], mt);
        alpha[qi] = expf(m_run[qi] - m_new);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pr[qi][i] = expf(sc[i] - m_new);
          ps += pr[qi][i];
        }
        ps += shfl_xor(ps, 16);
        ps += shfl_xor(ps, 32);
        l_run[qi] = l_run[qi] * alpha[qi] + ps;
        m_run[qi] = m_new;
      }
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
        for (int qi = 0; qi < kQG; ++qi) o[qi][dt] = o[qi][dt] * alpha[qi];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = Vs[(kb * 16 + 4 * g + r) * kXP + dt * 16 + qcol];
#pragma unroll
            for (int qi = 0; qi < kQG; ++qi) o[qi][dt] = mfma_16x16x4_f32(a, pr[qi][kb * 4 + r], o[qi][dt]);
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int qi = 0; qi < kQG; ++qi) {
    if (!qvalid[qi] || qy[qi] >= k.fh || qx[qi] >= k.fw) continue;
    const float inv = 1.f / l_run[qi];
    float* dst = k.out + ((int64_t)(qt[qi] * k.fh + qy[qi]) * k.fw + qx[qi]) * kDim + head * kHeadDim;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f4*>(dst + dt * 16 + 4 * g) = o[qi][dt] * inv;
  }
}


// ----------------------------------------------------------------------------------------
// f16 kernel (r03)
//
// Work-group = 4 waves; a wave owns 32 queries of one (window, head).  Masked window: the work-group is a 128-query
// block of the window's 45*t queries and walks all nt * (45 + 148 + npool) keys in 32-key tiles; unmasked window: the
// work-group handles two frames (waves 0,1 / 2,3), two tiles (45 valid keys) each.
//   tiles     K and V rows (256 B each) are copied global -> LDS by global_load_lds (16 B per lane, 4 rows per wave
//             instruction) into two 4-slot rings (8 KB per slot): K three-to-four tiles ahead of its use, V three --
//             one tile of compute hides ~1/3 of the copy latency, measured r03 with a 2-stage ring: 20 % of the kernel
//             time was the wait for the copy issued one tile earlier.  ONE barrier per tile.  The LDS image of a tile is
//             [32 rows][16 slots of 16 B]; the slot permutation is applied to the SOURCE address of the copy:
//               K: slot = chunk ^ (row & 15)        -> the ds_read_b128 fragment reads (32 rows, one chunk) are conflict-free
//               V: slot = chunk ^ ((row & 3) << 2)  -> the transposing reads (4 rows x 64 B per half-wave) are conflict-free
//             (SQ_LDS_BANK_CONFLICT = 0.04 % of SQ_LDS_IDX_ACTIVE, profiles/r03_attention_counters.md)
//   S^T       = K . Q^T on v_mfma_f32_32x32x16_f16 (A = K rows from LDS, B = Q fragments in registers): lane (h, q)
//             holds, for query q = lane & 31, the scores of keys (r&3) + 8*(r>>2) + 4*h, r = 0..15, of the tile;
//   softmax   online, lane-local over its 16 scores + one v_permlane32_swap for the row maximum; the row sum stays
//             split over the two half-waves until the end; O / l are rescaled only when some row maximum of the wave
//             grew by more than 2^kAtDefer = 2^8 since the last rescale (wave-uniform branch; until then the tile's
//             probabilities are taken against the OLD reference maximum, at most 2^8 too large: exact in f32/f16 range);
//   O^T      += V^T . P^T: B = the lane's own probabilities (registers 8m..8m+7 of S as f16: keys 16m+4h+{0..3} and
//             16m+8+4h+{0..3}), A = two ds_read_b64_tr_b16 of the row-major V tile in the same key order -- the
//             probabilities never leave registers and V is never transposed in memory.
//   pipeline  S^T of tile i+1 is issued BEFORE the softmax of tile i (its 8 MFMAs run under the exponentials of tile i),
//             then O^T += of tile i: inside one wave the matrix pipe and the vector ALU overlap; the second wave of the
//             SIMD (another work-group: 2 per CU) fills the rest.
// Key addressing: per work-group tables in LDS (spatial offset of the 193 own + rolled-neighbour tokens, the frames of
// t_ind); lanes 0..7 of a wave resolve the wave's 8 rows of a tile five tiles ahead into a wave-private row table (one
// (key frame, index in the frame) pair per lane, advancing by 32 per tile: at most one frame wrap since 45 + 148 > 32).
// Work-group -> (query block, head, window) mapping is XCD-aware, see the kernel.
// ----------------------------------------------------------------------------------------
struct AttnF16K {
  const half_t* qkv;
  const half_t* pkv;
  const int* win_masked;
  const int* t_ind;
  half_t* out;
  int t, nt, Hp, Wp, fh, fw, npool, nww;
  int nx;        // work-groups per (head, window): max(128-query blocks of a masked window, frame pairs)
  int npairs;    // (window, head) pairs
#ifdef PP_ATTN_TRACE
  long long* trace;  // tools/trace_attention.sh: [0] claim flag, [1] steps, then 8 time stamps per step of two waves
#endif
  float scale_log2e;
  signed char nb[148 * 2];
};

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

constexpr int kAtTile = 32;                     // keys per tile
constexpr int kAtRow = kHeadDim * 2;            // bytes per key row
constexpr int kAtSlot = kAtTile * kAtRow;       // bytes per ring slot (8 KB)
constexpr int kAtRing = 4;                      // slots per ring
constexpr int kAtVBase = kAtRing * kAtSlot;     // V ring behind the K ring
constexpr int kAtSpatial = kWinTok + 148;       // own + rolled-neighbour tokens per key frame
constexpr int kAtMaxNt = 1024;
constexpr int kAtTabs = 2 * kAtRing * kAtSlot;  // [rowtab: 4 waves x 16 tiles x 8 rows x i64][koff][tind]
constexpr int kAtRowTab = 4 * 16 * 8 * 8;
constexpr float kAtDefer = 8.f;  // rescale O / l only when a row maximum grew by more than 2^8 (see softmax_head)
constexpr int kAtSmem = kAtTabs + kAtRowTab + (kAtSpatial + 3 + kAtMaxNt) * 4;
static_assert(kHeads == 4, "pair decoding");

__global__ void __launch_bounds__(256, 2) window_attention_f16_kernel(const AttnF16K k) {
  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);  // [K ring][V ring][rowtab][koff][tind]
  int* koff = reinterpret_cast<int*>(smem + kAtTabs + kAtRowTab);
  int* tind = koff + kAtSpatial + 3;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int h = lane >> 5, q32 = lane & 31;
  int64_t* rowtab = reinterpret_cast<int64_t*>(smem + kAtTabs) + wave * 128;  // this wave's [16 tiles][8 rows]

  // ---- work-group -> (x, head, window), XCD-aware: the hardware deals work-groups to the 8 XCDs round robin
  // (id & 7); the (window, head) pairs are dealt the same way, so neighbouring (mostly equally masked) windows spread
  // over all XCDs, and the work-groups of ONE pair follow each other on one XCD: they run together and share its L2.
  const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
  const int pair = xcd + 8 * (jx / k.nx);
  const int bx = jx % k.nx;
  if (pair >= k.npairs) return;
  const int head = pair & (kHeads - 1);
  const int win = pair >> 2;
  const int wi = win / k.nww, wj = win - wi * k.nww;
  const int r0 = wi * kWinH, c0 = wj * kWinW;
  const bool masked = k.win_masked[win] != 0;
  const int nq = masked ? k.t * kWinTok : kWinTok;
  if (masked ? bx * 128 >= nq : bx * 2 >= k.t) return;

  // ---- tables -------------------------------------------------------------------------------------
  if (tid < kAtSpatial) {
    int y, x;
    if (tid < kWinTok) {
      y = r0 + tid / kWinW;
      x = c0 + tid % kWinW;
    } else {
      const int ni = tid - kWinTok;
      y = (r0 + (int)k.nb[2 * ni] + k.Hp) % k.Hp;
      x = (c0 + (int)k.nb[2 * ni + 1] + k.Wp) % k.Wp;
    }
    koff[tid] = (y * k.Wp + x) * (3 * kDim) + kDim + head * kHeadDim;  // K row of the token inside its frame; V = +kDim
  }
  for (int i = tid; i < k.nt; i += 256) tind[i] = k.t_ind[i];
  __syncthreads();

  // ---- this lane's query ------------------------------------------------------------------------
  int qframe, qpos;
  bool qvalid;
  if (masked) {
    const int ql = bx * 128 + wave * 32 + q32;
    qvalid = ql < nq;
    const int qc = qvalid ? ql : 0;
    qframe = qc / kWinTok;
    qpos = qc - qframe * kWinTok;
  } else {
    qframe = 2 * bx + (wave >> 1);
    qpos = (wave & 1) * 32 + q32;
    qvalid = qframe < k.t && qpos < kWinTok;
    if (!qvalid) qframe = 0, qpos = 0;
  }
  const int qy = r0 + qpos / kWinW, qx = c0 + qpos % kWinW;
  h8 qf[8];
  {
    const half_t* qptr = k.qkv + ((int64_t)(qframe * k.Hp + qy) * k.Wp + qx) * (3 * kDim) + head * kHeadDim + 8 * h;
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const h8*>(qptr + 16 * c);
  }

  // ---- copy roles: this wave copies rows 8*wave + 4*jj + (lane >> 4), jj = 0,1, of a tile ----------------
  // keys past the end alias a valid row (their scores are masked, and 0 x finite = 0)
  const int g16 = lane >> 4, slot = lane & 15;
  const int64_t frame_bytes = (int64_t)k.Hp * k.Wp * (3 * kDim) * 2;
  const int per_frame = kAtSpatial + k.npool;
  const unsigned char* qkv_b = reinterpret_cast<const unsigned char*>(k.qkv);
  const int64_t pkv_rel = reinterpret_cast<const unsigned char*>(k.pkv) - qkv_b + (int64_t)head * kAtRow;
  // Row addresses: every 8 steps a wave resolves its 8 rows of the next 8 tiles with all 64 lanes (lane = (tile, row):
  // one (key frame, index in the frame) pair per lane, advancing by 256 keys per call) into a wave-private table of 16
  // tiles; the two row offsets a lane copies are fetched from it one step before the copy is issued (a wave issues in
  // order: an LDS read followed at once by its use stalls the MFMAs queued behind it for the LDS latency); V of a tile is
  // copied one step after its K and reuses the registers.
  int kr = (lane >> 3) * kAtTile + 8 * wave + (lane & 7), kfi = 0;
  auto resolve_group = [&](int group) __attribute__((always_inline)) {  // tiles 8*group .. 8*group+7
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)  // 256 keys per call, a key frame has at least 193
      if (kr >= per_frame) {
        kr -= per_frame;
        ++kfi;
      }
    const int fr = tind[kfi < k.nt ? kfi : k.nt - 1];  // past the end: alias the last frame (masked keys)
    const int64_t sp = (int64_t)fr * frame_bytes + (int64_t)koff[kr < kAtSpatial ? kr : 0] * 2;
    const int64_t pl = pkv_rel + ((int64_t)fr * k.npool + (kr - kAtSpatial)) * (2 * kDim * 2);
    rowtab[((8 * group + (lane >> 3)) & 15) * 8 + (lane & 7)] = kr < kAtSpatial ? sp : pl;
    kr += 8 * kAtTile;
  };
  // wave-uniform parts of the swizzles: row & 15 = (8*wave + 4*jj + g16) & 15, row & 3 = g16
  const int rq0 = (8 * wave + g16) & 15, rq1 = (8 * wave + 4 + g16) & 15;
  const int kch0 = (slot ^ rq0) << 4, kch1 = (slot ^ rq1) << 4, vch = kDim * 2 + ((slot ^ (g16 << 2)) << 4);
  struct Rows {
    int64_t a, b;
  };
  auto fetch_rows = [&](int tile) __attribute__((always_inline)) -> Rows {
    const int64_t* rows = rowtab + (tile & 15) * 8;
    return Rows{rows[g16], rows[4 + g16]};
  };
  auto issue_k = [&](auto ring_slot, const Rows& r) __attribute__((always_inline)) {
    unsigned char* st = smem + decltype(ring_slot)::value * kAtSlot + 8 * wave * kAtRow;
    glds16(qkv_b + r.a + kch0, st);
    glds16(qkv_b + r.b + kch1, st + 4 * kAtRow);
  };
  auto issue_v = [&](auto ring_slot, const Rows& r) __attribute__((always_inline)) {
    unsigned char* st = smem + kAtVBase + decltype(ring_slot)::value * kAtSlot + 8 * wave * kAtRow;
    glds16(qkv_b + r.a + vch, st);
    glds16(qkv_b + r.b + vch, st + 4 * kAtRow);
  };

  // ---- fragment offsets -----------------------------------------------------------------------------
  int kfo[8];  // K fragment: row q32, chunk 2c + h
#pragma unroll
  for (int c = 0; c < 8; ++c) kfo[c] = q32 * kAtRow + (((2 * c + h) ^ (q32 & 15)) << 4);
  int vfo[4];  // V fragment of d block `db`: rows 4h + (i >> 2), columns 32 db + 16 (g16 & 1) + 4 (i & 3)
  {
    const int i = lane & 15, ir = i >> 2;
#pragma unroll
    for (int db = 0; db < 4; ++db)
      vfo[db] = kAtVBase + (4 * h + ir) * kAtRow + ((4 * (db ^ ir) + 2 * (g16 & 1) + ((i & 3) >> 1)) << 4) + (i & 1) * 8;
  }

  f16v o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const float sc = k.scale_log2e;
#ifdef PP_ATTN_TRACE
  __shared__ int traced;
  if (tid == 0) traced = masked && bx == 1 && atomicCAS(reinterpret_cast<unsigned long long*>(k.trace), 0ull, 1ull) == 0ull;
  __syncthreads();
  const bool tr_on = traced && (wave == 0 || wave == 3) && lane == 0;
#define PP_TR(step, slot)                                                                                  \
  if (tr_on && (step) < 100) k.trace[2 + (((wave ? 1 : 0) * 100 + (step)) * 8) + (slot)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define PP_TR(step, slot)
#endif

  // S^T of the 32-key tile in K ring slot `ring_slot`: the 8 fragment reads are issued early (kfrags, right after the
  // barrier that publishes the tile) so that their LDS latency passes under the copy issue and the row maxima
  struct KFrags {
    h8 f[8];
  };
  auto kfrags = [&](auto ring_slot) __attribute__((always_inline)) -> KFrags {
    const unsigned char* st = smem + decltype(ring_slot)::value * kAtSlot;
    KFrags r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r.f[c] = *reinterpret_cast<const h8*>(st + kfo[c]);
    return r;
  };
  auto qk = [&](const KFrags& kf) __attribute__((always_inline)) -> f16v {
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s = mfma_32x32x16_f16(kf.f[c], qf[c], s);
    return s;
  };
  // online softmax, part 1: row maxima of one tile's scores (keys >= nvalid masked out when TAIL)
  auto softmax_head = [&](f16v& s, auto tail, int nvalid) __attribute__((always_inline)) {
    if constexpr (decltype(tail)::value) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) + 4 * h >= nvalid) s[r] = -1.0e30f;
    }
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = pair32_max(mt);
    // Deferred rescale: O and l are rescaled (and the reference maximum moved) only when some row maximum of the wave grew
    // by more than 2^kAtDefer in the exponent; otherwise the probabilities of this tile are taken against the OLD
    // reference and may reach 2^8 -- exact in f16 / fp32 arithmetic up to rounding (f16 keeps 11 significant bits at any
    // magnitude, 2^8 x 32 keys is far from its range limit), and O / l are normalised by the same reference at the end.
    // The decision covers the whole tile before any of its probabilities exists, so nothing is scaled twice or not at all.
    const float m_new = fmaxf(m_run, mt);
    if (wave_any((m_new - m_run) * sc > kAtDefer)) {
      const float alpha = fast_exp2((m_run - m_new) * sc);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m_run = m_new;
    }
  };
  // part 2: probabilities, then O^T += V^T . P^T from V ring slot `ring_slot` (branch-free: one basic block together
  // with the S^T MFMAs of the next tile issued in front of it, so the exponentials run under those MFMAs)
  auto exp_pv = [&](const f16v& s, auto ring_slot) __attribute__((always_inline)) {
    const float nm = -m_run * sc;
    float ps = 0.f;
    h8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(__builtin_fmaf(s[r], sc, nm));
      ps += p;
      pf[r >> 3][r & 7] = (half_t)p;
    }
    l_run += ps;
    // two groups of 16 keys: the transposing reads are instructions hipcc does not see (lds_tr16_issue)
    h4 fa[8], fb[8];
    constexpr int kKeyOff = 8 * kAtRow;
    constexpr int base = decltype(ring_slot)::value * kAtSlot;
    auto vissue = [&](auto off, h4* f) __attribute__((always_inline)) {
      constexpr int B = base + decltype(off)::value;
      lds_tr16_issue<B>(f[0], smem + vfo[0]);
      lds_tr16_issue<B + kKeyOff>(f[1], smem + vfo[0]);
      lds_tr16_issue<B>(f[2], smem + vfo[1]);
      lds_tr16_issue<B + kKeyOff>(f[3], smem + vfo[1]);
      lds_tr16_issue<B>(f[4], smem + vfo[2]);
      lds_tr16_issue<B + kKeyOff>(f[5], smem + vfo[2]);
      lds_tr16_issue<B>(f[6], smem + vfo[3]);
      lds_tr16_issue<B + kKeyOff>(f[7], smem + vfo[3]);
    };
    auto vmfma = [&](int m, h4* f) __attribute__((always_inline)) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const h4 lo = f[2 * db], hi = f[2 * db + 1];
        const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[db] = mfma_32x32x16_f16(a, pf[m], o[db]);
      }
    };
    vissue(std::integral_constant<int, 0>{}, fa);
    vissue(std::integral_constant<int, 16 * kAtRow>{}, fb);
    lds_tr16_wait<8>(fa[0], fa[1], fa[2], fa[3], fa[4], fa[5], fa[6], fa[7]);
    vmfma(0, fa);
    lds_tr16_wait<0>(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7]);
    vmfma(1, fb);
  };
  typedef std::false_type NoTail;
  typedef std::true_type Tail;
  typedef std::integral_constant<int, 0> R0;
  typedef std::integral_constant<int, 1> R1;
  typedef std::integral_constant<int, 2> R2;
  typedef std::integral_constant<int, 3> R3;

  if (masked) {
    const int nk = k.nt * per_frame;
    const int ntiles = (nk + kAtTile - 1) / kAtTile;
    const int last = ntiles - 1;
    // copies are issued in a fixed pattern -- per step: V of tile s+3, K of tile s+4 (tiles past the end resolve to
    // aliases of valid rows: uniform instruction counts keep the vmcnt arithmetic simple; the slots they land in are
    // never read)
    resolve_group(0);
    wave_lds_fence();
    Rows rv, rk;
    rk = fetch_rows(0);
    issue_k(R0{}, rk);                          // step -4
    rv = rk; rk = fetch_rows(1);
    issue_v(R0{}, rv); issue_k(R1{}, rk);       // step -3
    rv = rk; rk = fetch_rows(2);
    issue_v(R1{}, rv); issue_k(R2{}, rk);       // step -2
    rv = rk; rk = fetch_rows(3);
    issue_v(R2{}, rv); issue_k(R3{}, rk);       // step -1
    rv = rk; rk = fetch_rows(4);                // for step 0
    pp_wait_vmcnt<12>();                        // K of tile 0 has landed (this wave's rows)
    pp_barrier();
    f16v s_cur = qk(kfrags(R0{}));
    // step i (ring position P = i & 3): V(i) and K(i+1) have landed; issue V(i+3) -> V slot (P+3)&3 and K(i+4) -> K slot P
    // (both read for the last time in step i-1); S^T(i+1) from K slot (P+1)&3, softmax(i), O^T += from V slot P
    auto step = [&](int i, auto P, auto P1, auto P3) __attribute__((always_inline)) {
      PP_TR(i, 0);
      pp_wait_vmcnt<8>();
      PP_TR(i, 1);
      pp_barrier();
      PP_TR(i, 2);
      const KFrags kf = kfrags(P1);
      issue_v(P3, rv);
      issue_k(P, rk);
      rv = rk;
      rk = fetch_rows(i + 5);
      if (decltype(P)::value == 0 && (i & 7) == 0) resolve_group((i >> 3) + 1);  // first read in step i + 3
      PP_TR(i, 3);
      // S^T of the next tile is issued in the block of the row maxima (its 8 MFMAs run under the max chain), the
      // exponentials then overlap the O^T MFMAs
      const f16v s_next = qk(kf);
      softmax_head(s_cur, NoTail{}, kAtTile);
      PP_TR(i, 4);
      exp_pv(s_cur, P);
      s_cur = s_next;
      PP_TR(i, 5);
    };
    auto final_step = [&](auto P) __attribute__((always_inline)) {  // the last tile: masked tail, nothing left to copy
      pp_wait_vmcnt<0>();
      pp_barrier();
      softmax_head(s_cur, Tail{}, nk - last * kAtTile);
      exp_pv(s_cur, P);
    };
    int i = 0;
    for (; i + 4 <= last; i += 4) {
      step(i, R0{}, R1{}, R3{});
      step(i + 1, R1{}, R2{}, R0{});
      step(i + 2, R2{}, R3{}, R1{});
      step(i + 3, R3{}, R0{}, R2{});
    }
    const int rem = last - i;  // 0..3 full tiles, then the last one at ring position rem
    if (rem > 0) step(i, R0{}, R1{}, R3{});
    if (rem > 1) step(i + 1, R1{}, R2{}, R0{});
    if (rem > 2) step(i + 2, R2{}, R3{}, R1{});
    if (rem == 0) final_step(R0{});
    else if (rem == 1) final_step(R1{});
    else if (rem == 2) final_step(R2{});
    else final_step(R3{});
  } else {
    // two frames: waves 0,1 -> frame 2bx (ring slots 0,1), waves 2,3 -> frame 2bx+1 (slots 2,3); keys = the 45 own
    // tokens of the frame as two tiles (rows past 44 alias token 44, a missing second frame aliases the first: masked /
    // never read).  Every wave copies its 8 rows of all four slots.
    const int f1 = 2 * bx + 1 < k.t ? 2 * bx + 1 : 2 * bx;
    if (lane < 8) {
      const int ra = 8 * wave + lane, rb = 32 + 8 * wave + lane;
      rowtab[0 * 8 + lane] = (int64_t)(2 * bx) * frame_bytes + (int64_t)koff[ra] * 2;
      rowtab[1 * 8 + lane] = (int64_t)(2 * bx) * frame_bytes + (int64_t)koff[rb < kWinTok ? rb : kWinTok - 1] * 2;
      rowtab[2 * 8 + lane] = (int64_t)f1 * frame_bytes + (int64_t)koff[ra] * 2;
      rowtab[3 * 8 + lane] = (int64_t)f1 * frame_bytes + (int64_t)koff[rb < kWinTok ? rb : kWinTok - 1] * 2;
    }
    wave_lds_fence();
    const Rows u0 = fetch_rows(0), u1 = fetch_rows(1), u2 = fetch_rows(2), u3 = fetch_rows(3);
    issue_k(R0{}, u0); issue_v(R0{}, u0);
    issue_k(R1{}, u1); issue_v(R1{}, u1);
    issue_k(R2{}, u2); issue_v(R2{}, u2);
    issue_k(R3{}, u3); issue_v(R3{}, u3);
    pp_wait_vmcnt<0>();
    pp_barrier();
    auto frame = [&](auto A, auto B) __attribute__((always_inline)) {
      f16v s = qk(kfrags(A));
      softmax_head(s, NoTail{}, kAtTile);
      exp_pv(s, A);
      s = qk(kfrags(B));
      softmax_head(s, Tail{}, kWinTok - kAtTile);
      exp_pv(s, B);
    };
    if (wave >> 1) frame(R2{}, R3{}); else frame(R0{}, R1{});
  }

  // ---- normalise and scatter back to the unpadded token grid -------------------------------------
  const float l_tot = pair32_sum(l_run);
  if (!qvalid || qy >= k.fh || qx >= k.fw) return;
  const float inv = 1.f / l_tot;
  half_t* dst = k.out + ((int64_t)(qframe * k.fh + qy) * k.fw + qx) * kDim + head * kHeadDim + 4 * h;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const h4 v = {(half_t)(o[db][4 * rg] * inv), (half_t)(o[db][4 * rg + 1] * inv), (half_t)(o[db][4 * rg + 2] * inv),
                    (half_t)(o[db][4 * rg + 3] * inv)};
      *reinterpret_cast<h4*>(dst + 32 * db + 8 * rg) = v;
    }
}

}  // namespace pp

// rolled-neighbour offsets relative to the window origin (sparse_transformer.py:184-197, 229-283):
// rows {-3..1} u {3..7}, cols {-5..3} u {5..13}, minus the window's own 5x9 footprint -> 148 tokens
static int fill_neighbour_table(signed char* nb) {
  using namespace pp;
  int n = 0;
  const int eh = (kWinH + 1) / 2, ew = (kWinW + 1) / 2;
  for (int ri = 0; ri < 2 * kWinH; ++ri) {
    const int dr = ri < kWinH ? ri - eh : ri - kWinH + eh;
    for (int ci = 0; ci < 2 * kWinW; ++ci) {
      const int dc = ci < kWinW ? ci - ew : ci - kWinW + ew;
      if (dr >= 0 && dr < kWinH && dc >= 0 && dc < kWinW) continue;
      if (n < 148) {
        nb[2 * n] = (signed char)dr;
        nb[2 * n + 1] = (signed char)dc;
      }
      ++n;
    }
  }
  return n;
}

template <typename T>
static int launch_window_attention_generic(void* stream, const pp_window_attention_params* p) {
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
  if (fill_neighbour_table(k.nb) != 148) return pp_fail(PP_ERR_LAUNCH, "pp_window_attention: internal neighbour table error");
  const int nwin = (k.Hp / kWinH) * k.nww;
  dim3 grid((unsigned)k.t, kHeads, (unsigned)nwin);
  if constexpr (std::is_same<T, float>::value) {
    if (p->exact) {
      PP_LAUNCH(window_attention_f32_exact_kernel, grid, dim3(256), 0, stream, k);
      return pp_check_launch("pp_window_attention");
    }
  }
  PP_LAUNCH((window_attention_generic_kernel<T>), grid, dim3(256), 0, stream, k);
  return pp_check_launch("pp_window_attention");
}

static int launch_window_attention_f16(void* stream, const pp_window_attention_params* p) {
  using namespace pp;
  AttnF16K k;
  k.qkv = (const half_t*)p->qkv;
  k.pkv = (const half_t*)p->pkv;
  k.win_masked = (const int*)p->win_masked;
  k.t_ind = (const int*)p->t_ind;
  k.out = (half_t*)p->out;
  k.t = (int)p->t; k.nt = (int)p->nt; k.Hp = (int)p->Hp; k.Wp = (int)p->Wp; k.fh = (int)p->fh; k.fw = (int)p->fw;
  k.npool = (int)p->npool;
  k.nww = k.Wp / kWinW;
  k.scale_log2e = p->scale * 1.4426950408889634f;
  if (fill_neighbour_table(k.nb) != 148) return pp_fail(PP_ERR_LAUNCH, "pp_window_attention: internal neighbour table error");
  const int nwin = (k.Hp / kWinH) * k.nww;
  const int nqb = (k.t * kWinTok + 127) / 128, npair = (k.t + 1) / 2;
  k.nx = nqb > npair ? nqb : npair;
  k.npairs = kHeads * nwin;
  const int nwg = 8 * ((k.npairs + 7) / 8) * k.nx;
  PP_ALLOW_BIG_LDS((&window_attention_f16_kernel), kAtSmem);
#ifdef PP_ATTN_TRACE
  static long long* trace = nullptr;
  const size_t tbytes = (2 + 2 * 100 * 8) * sizeof(long long);
  if (!trace) hipMalloc(&trace, tbytes);
  hipMemsetAsync(trace, 0, tbytes, (hipStream_t)stream);
  k.trace = trace;
#endif
  PP_LAUNCH(window_attention_f16_kernel, dim3((unsigned)nwg), dim3(256), kAtSmem, stream, k);
#ifdef PP_ATTN_TRACE
  {
    static long long host[2 + 2 * 100 * 8];
    hipMemcpyAsync(host, trace, tbytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    hipStreamSynchronize((hipStream_t)stream);
    static int dumped = 0;
    if (dumped++ == 3) {  // one warmed-up launch
      const char* names[6] = {"top", "vmcnt", "barrier", "issue+resolve", "head", "qk+exp+pv"};
      for (int w = 0; w < 2; ++w) {
        double sum[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int st = 4; st < 76; ++st) {
          const long long* t = host + 2 + (w * 100 + st) * 8;
          const long long* tn = host + 2 + (w * 100 + st + 1) * 8;
          if (!t[0] || !tn[0]) continue;
          for (int j = 0; j < 5; ++j) sum[j + 1] += (double)(t[j + 1] - t[j]);
          sum[0] += (double)(tn[0] - t[0]);
          ++n;
        }
        fprintf(stderr, "attention trace wave %d (%d steps): step %.0f clk =", w ? 3 : 0, n, n ? sum[0] / n : 0.0);
        for (int j = 1; j < 6; ++j) fprintf(stderr, " %s %.0f |", names[j], n ? sum[j] / n : 0.0);
        fprintf(stderr, "\n");
        for (int st = 20; st < 26; ++st) {
          const long long* t = host + 2 + (w * 100 + st) * 8;
          fprintf(stderr, "   step %d:", st);
          for (int j = 0; j < 5; ++j) fprintf(stderr, " %lld", t[j + 1] - t[j]);
          fprintf(stderr, "  -> next %lld\n", host[2 + (w * 100 + st + 1) * 8] - t[0]);
        }
      }
    }
  }
#endif
  return pp_check_launch("pp_window_attention");
}

extern "C" int32_t pp_window_attention(void* stream, const pp_window_attention_params* p) {
  using namespace pp;
  if (!p || !p->qkv || !p->pkv || !p->win_masked || !p->t_ind || !p->out)
    return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: null argument");
  if (p->Hp % kWinH || p->Wp % kWinW) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: grid not padded to 5x9 windows");
  if (p->t < 1 || p->nt < 1 || p->t > 65535) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: bad t / nt");
  if (p->exact != 0 && (p->exact != 1 || p->dtype != PP_F32))
    return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: exact is 0 or 1, and 1 only with PP_F32 storage");
  if (p->dtype == PP_F16) {
    // the f16 kernel keeps 32-bit element offsets inside a frame and the t_ind table in LDS; beyond those limits the
    // generic kernel (64-bit addressing, no table) serves f16 storage as it did until r02
    if (p->nt > kAtMaxNt || (int64_t)p->Hp * p->Wp * (3 * kDim) >= (int64_t)1 << 30)
      return launch_window_attention_generic<half_t>(stream, p);
    return launch_window_attention_f16(stream, p);
  }
  if (p->dtype == PP_F32) return launch_window_attention_generic<float>(stream, p);
  return pp_fail(PP_ERR_UNSUPPORTED, "pp_window_attention: dtype");
}

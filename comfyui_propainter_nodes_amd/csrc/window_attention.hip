// window_attention.hip -- the core of SparseWindowAttention.forward (sparse_transformer.py:218-385), flash-style on MFMA.
// Contract: include/propainter_mi355.h (pp_window_attention).  Two kernels:
//   window_attention_f16_kernel      f16 storage (the node's fp16 "enable"): K / V tiles by LDS-DMA, double-buffered,
//                                    32x32x16 MFMAs, V consumed through the transposing LDS read;
//   window_attention_generic_kernel  fp32 storage (fp16 "disable"): register-staged tiles converted to f16 MFMA operands,
//                                    16x16x32 MFMAs (the round-1/2 kernel).
#include "attn_device.h"
#include "pp_host.h"

#include <stdlib.h>
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
// f16 kernel (r03)
//
// Work-group = 4 waves; a wave owns 32 queries of one (window, head).  Masked window: the work-group is a 128-query
// block of the window's 45*t queries and walks all nt * (45 + 148 + npool) keys in 64-key tiles; unmasked window: the
// work-group handles two frames (waves 0,1 / 2,3), one 64-key tile (45 valid keys) each.
//   tiles     K and V rows (256 B each) are copied global -> LDS by global_load_lds (16 B per lane, 4 rows per wave
//             instruction), tile i+1 in flight while tile i is consumed; ONE barrier per tile.  The LDS image of a
//             tile is [64 rows][16 slots of 16 B]; the slot permutation is applied to the SOURCE address of the copy:
//               K: slot = chunk ^ (row & 15)        -> the ds_read_b128 fragment reads (32 rows, one chunk) are conflict-free
//               V: slot = chunk ^ ((row & 3) << 2)  -> the transposing reads (4 rows x 64 B per half-wave) are conflict-free
//   S^T       = K . Q^T on v_mfma_f32_32x32x16_f16 (A = K rows from LDS, B = Q fragments in registers): lane (h, q)
//             holds, for query q = lane & 31, the scores of keys 32*kb + (r&3) + 8*(r>>2) + 4*h, r = 0..15, kb = 0,1;
//   softmax   online, lane-local over its 32 scores + one v_permlane32_swap for the row maximum; the row sum stays
//             split over the two half-waves until the end;
//   O^T      += V^T . P^T: B = the lane's own probabilities (registers 8m..8m+7 of S[kb] as f16: keys 16m+4h+{0..3} and
//             16m+8+4h+{0..3}), A = two ds_read_b64_tr_b16 of the row-major V tile in the same key order -- the
//             probabilities never leave registers and V is never transposed in memory.
// Key addressing: per work-group tables in LDS (spatial offset of the 193 own + rolled-neighbour tokens, the frames of
// t_ind); a lane's four key rows advance by 64 per tile with at most one frame wrap (45 + 148 > 64).
// Work-group -> (query block, head, window) mapping is XCD-aware: the blocks that share a key set are neighbours in
// the linear order of ONE XCD, so the key set is fetched into one L2 (r02: every 128-query block re-fetched it through
// whichever of the 8 L2s it landed on: 2.3x the algorithmic traffic).
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
  int dbg;
  float scale_log2e;
  signed char nb[148 * 2];
};

constexpr int kAtTile = 64;                     // keys per tile
constexpr int kAtRow = kHeadDim * 2;            // bytes per key row
constexpr int kAtHalf = kAtTile * kAtRow;       // K (or V) bytes per stage
constexpr int kAtStage = 2 * kAtHalf;           // K | V
constexpr int kAtSpatial = kWinTok + 148;       // own + rolled-neighbour tokens per key frame
constexpr int kAtMaxNt = 1024;
constexpr int kAtTabs = 2 * kAtStage;                                  // [rowtab 2 x 64 x i64][koff][tind]
constexpr int kAtSmem = kAtTabs + 2 * kAtTile * 8 + (kAtSpatial + 3 + kAtMaxNt) * 4;
static_assert(kHeads == 4, "pair decoding");

__global__ void __launch_bounds__(256, 2) window_attention_f16_kernel(const AttnF16K k) {
  unsigned char* smem = reinterpret_cast<unsigned char*>(PP_DYN_SMEM);  // [stage 0: K | V][stage 1: K | V][rowtab][koff][tind]
  int64_t* rowtab = reinterpret_cast<int64_t*>(smem + kAtTabs);         // [2][64]: byte offset of a tile's K rows from qkv
  int* koff = reinterpret_cast<int*>(smem + kAtTabs + 2 * kAtTile * 8);
  int* tind = koff + kAtSpatial + 3;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int h = lane >> 5, q32 = lane & 31;

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

  // ---- copy roles: this wave copies rows 16*wave + 4*jj + (lane >> 4), jj = 0..3, of a tile --------
  // Row addresses come from a per-tile table in LDS written by wave 0 two tiles ahead (lane = key slot of the tile:
  // one (key frame, index inside the frame) pair per lane, advancing by 64 per tile with at most one frame wrap since
  // 45 + 148 > 64); keys past the end alias a valid row (their scores are masked, and 0 x finite = 0).
  const int g16 = lane >> 4, slot = lane & 15;
  const int64_t frame_bytes = (int64_t)k.Hp * k.Wp * (3 * kDim) * 2;
  const int per_frame = kAtSpatial + k.npool;
  const unsigned char* qkv_b = reinterpret_cast<const unsigned char*>(k.qkv);
  const int64_t pkv_rel = reinterpret_cast<const unsigned char*>(k.pkv) - qkv_b + (int64_t)head * kAtRow;
  int kr = lane, kfi = 0;  // wave 0: the key this lane resolves next
  auto produce = [&](int par) __attribute__((always_inline)) {
    const int fi = kfi < k.nt ? kfi : k.nt - 1;
    const int fr = tind[fi];
    const int64_t sp = (int64_t)fr * frame_bytes + (int64_t)koff[kr < kAtSpatial ? kr : 0] * 2;
    const int64_t pp = pkv_rel + ((int64_t)fr * k.npool + (kr - kAtSpatial)) * (2 * kDim * 2);
    rowtab[par * kAtTile + lane] = kr < kAtSpatial ? sp : pp;
    kr += kAtTile;
    if (kr >= per_frame) {
      kr -= per_frame;
      ++kfi;
    }
  };
  auto issue = [&](auto stage, const int64_t* rows) __attribute__((always_inline)) {  // copy the 64 rows `rows[]` into `stage`
    unsigned char* st = smem + decltype(stage)::value * kAtStage;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int rq = 4 * jj + g16;  // row & 15
      const unsigned char* rp = qkv_b + rows[16 * wave + rq];
      glds16(rp + ((slot ^ rq) << 4), st + (16 * wave + 4 * jj) * kAtRow);
      glds16(rp + kDim * 2 + ((slot ^ (g16 << 2)) << 4), st + kAtHalf + (16 * wave + 4 * jj) * kAtRow);
    }
  };

  // ---- fragment offsets -----------------------------------------------------------------------------
  int kfo[8];  // K fragment: row q32 (+32 for the second key block), chunk 2c + h
#pragma unroll
  for (int c = 0; c < 8; ++c) kfo[c] = q32 * kAtRow + (((2 * c + h) ^ (q32 & 15)) << 4);
  int vfo[4];  // V fragment of d block `db`: rows 4h + (i >> 2), columns 32 db + 16 (g16 & 1) + 4 (i & 3)
  {
    const int i = lane & 15, ir = i >> 2;
#pragma unroll
    for (int db = 0; db < 4; ++db)
      vfo[db] = kAtHalf + (4 * h + ir) * kAtRow + ((4 * (db ^ ir) + 2 * (g16 & 1) + ((i & 3) >> 1)) << 4) + (i & 1) * 8;
  }

  f16v o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const float sc = k.scale_log2e;

  // one 64-key tile in `stage`; keys >= nvalid (counted from the tile start) are masked out
  auto tile = [&](auto stage, int nvalid) __attribute__((always_inline)) {
    const unsigned char* st = smem + decltype(stage)::value * kAtStage;
    f16v s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const h8 a = *reinterpret_cast<const h8*>(st + kfo[c] + kb * 32 * kAtRow);
        s[kb] = mfma_32x32x16_f16(a, qf[c], s[kb]);
      }
    }
    if (nvalid < kAtTile) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= nvalid) s[kb][r] = -1.0e30f;
    }
    float mt = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
    mt = pair32_max(mt);
    const float m_new = fmaxf(m_run, mt);
    // the running maximum of most rows stops moving after a few tiles: rescale O and l only when some row of the wave
    // moved (wave-uniform branch; exact -- nothing is deferred)
    if (wave_any(m_new > m_run)) {
      const float alpha = fast_exp2((m_run - m_new) * sc);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m_run = m_new;
    }
    const float nm = -m_run * sc;
    float ps = 0.f;
    h8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[kb][r], sc, nm));
        ps += p;
        pf[kb][r >> 3][r & 7] = (half_t)p;
      }
    l_run += ps;
    // four groups (kb, m) of 16 keys: the transposing reads of group g+1 are in flight under the MFMAs of group g
    // (two register sets; the reads are instructions hipcc does not see, see lds_tr16_issue)
    h4 fa[8], fb[8];
    constexpr int kKeyOff = 8 * kAtRow;
    auto vissue = [&](auto grp, h4* f) __attribute__((always_inline)) {
      constexpr int G = decltype(grp)::value;
      constexpr int base = decltype(stage)::value * kAtStage + ((G >> 1) * 32 + 16 * (G & 1)) * kAtRow;
      lds_tr16_issue<base>(f[0], smem + vfo[0]);
      lds_tr16_issue<base + kKeyOff>(f[1], smem + vfo[0]);
      lds_tr16_issue<base>(f[2], smem + vfo[1]);
      lds_tr16_issue<base + kKeyOff>(f[3], smem + vfo[1]);
      lds_tr16_issue<base>(f[4], smem + vfo[2]);
      lds_tr16_issue<base + kKeyOff>(f[5], smem + vfo[2]);
      lds_tr16_issue<base>(f[6], smem + vfo[3]);
      lds_tr16_issue<base + kKeyOff>(f[7], smem + vfo[3]);
    };
    auto vmfma = [&](int kb, int m, h4* f) __attribute__((always_inline)) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const h4 lo = f[2 * db], hi = f[2 * db + 1];
        const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[db] = mfma_32x32x16_f16(a, pf[kb][m], o[db]);
      }
    };
    vissue(std::integral_constant<int, 0>{}, fa);
    vissue(std::integral_constant<int, 1>{}, fb);
    lds_tr16_wait<8>(fa[0], fa[1], fa[2], fa[3], fa[4], fa[5], fa[6], fa[7]);
    vmfma(0, 0, fa);
    vissue(std::integral_constant<int, 2>{}, fa);
    lds_tr16_wait<8>(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7]);
    vmfma(0, 1, fb);
    vissue(std::integral_constant<int, 3>{}, fb);
    lds_tr16_wait<8>(fa[0], fa[1], fa[2], fa[3], fa[4], fa[5], fa[6], fa[7]);
    vmfma(1, 0, fa);
    lds_tr16_wait<0>(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7]);
    vmfma(1, 1, fb);
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;

  if (masked) {
    const int nk = k.nt * per_frame;
    const int ntiles = (nk + kAtTile - 1) / kAtTile;
    if (wave == 0) {
      produce(0);
      produce(1);
    }
    __syncthreads();
    issue(S0{}, rowtab);
    for (int i = 0; i < ntiles; i += 2) {
      pp_wait_vmcnt<0>();
      pp_wait_lgkm0();
      pp_barrier();
      if (i + 1 < ntiles && !((k.dbg & 1) && i > 1)) issue(S1{}, rowtab + kAtTile);
      if (wave == 0) produce(0);  // rows of tile i+2
      tile(S0{}, nk - i * kAtTile);
      if (i + 1 < ntiles) {
        pp_wait_vmcnt<0>();
        pp_wait_lgkm0();
        pp_barrier();
        if (i + 2 < ntiles && !((k.dbg & 1) && i > 1)) issue(S0{}, rowtab);
        if (wave == 0) produce(1);  // rows of tile i+3
        tile(S1{}, nk - (i + 1) * kAtTile);
      }
    }
  } else {
    // two frames: frame 2bx -> stage 0, frame 2bx+1 -> stage 1; the 45 own tokens of the frame are the keys
    // (rows 45..63 alias token 44, a missing second frame aliases the first: masked / never read)
    if (wave < 2) {
      const int fr = 2 * bx + wave < k.t ? 2 * bx + wave : 2 * bx;
      rowtab[wave * kAtTile + lane] = (int64_t)fr * frame_bytes + (int64_t)koff[lane < kWinTok ? lane : kWinTok - 1] * 2;
    }
    __syncthreads();
    issue(S0{}, rowtab);
    issue(S1{}, rowtab + kAtTile);
    pp_wait_vmcnt<0>();
    pp_barrier();
    if (wave >> 1) tile(S1{}, kWinTok); else tile(S0{}, kWinTok);
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
  k.dbg = getenv("PP_ATTN_DBG") ? atoi(getenv("PP_ATTN_DBG")) : 0;
  const int nwg = 8 * ((k.npairs + 7) / 8) * k.nx;
  static const bool lds_ok = (pp_allow_big_lds(reinterpret_cast<const void*>(&window_attention_f16_kernel), kAtSmem), true);
  (void)lds_ok;
  PP_LAUNCH(window_attention_f16_kernel, dim3((unsigned)nwg), dim3(256), kAtSmem, stream, k);
  return pp_check_launch("pp_window_attention");
}

extern "C" int32_t pp_window_attention(void* stream, const pp_window_attention_params* p) {
  using namespace pp;
  if (!p || !p->qkv || !p->pkv || !p->win_masked || !p->t_ind || !p->out)
    return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: null argument");
  if (p->Hp % kWinH || p->Wp % kWinW) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: grid not padded to 5x9 windows");
  if (p->t < 1 || p->nt < 1 || p->t > 65535) return pp_fail(PP_ERR_BAD_ARG, "pp_window_attention: bad t / nt");
  if (p->dtype == PP_F16) {
    // 32-bit element offsets inside a frame, the t_ind table in LDS
    if (p->nt > kAtMaxNt || p->Hp * p->Wp * (3 * kDim) >= (int64_t)1 << 30)
      return pp_fail(PP_ERR_UNSUPPORTED, "pp_window_attention: more than 1024 key frames or a frame beyond 2^30 elements");
    return launch_window_attention_f16(stream, p);
  }
  if (p->dtype == PP_F32) return launch_window_attention_generic<float>(stream, p);
  return pp_fail(PP_ERR_UNSUPPORTED, "pp_window_attention: dtype");
}

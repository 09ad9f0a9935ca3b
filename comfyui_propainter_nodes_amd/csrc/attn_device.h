// attn_device.h -- device primitives used by the window-attention kernel only (kept out of pp_device.h so that
// tuning them rebuilds one translation unit).  gfx950; the PP_EMU branch holds the emulator's equivalents (tests only).
#pragma once
#include "pp_device.h"

namespace pp {

#ifndef PP_EMU
// 32x32x16 f16: lane l holds A[row = l&31][k = 8*(l>>5) .. +7], B[k = 8*(l>>5) .. +7][col = l&31]; C/D as 32x32x2 f32.
__device__ __forceinline__ f16v mfma_32x32x16_f16(h8 a, h8 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// ds_read_b64_tr_b16 (checked on the MI355X by tools/probes/probe_gfx950.hip): every lane passes the LDS address of 4
// consecutive 16-bit values (8-byte aligned); within each group of 16 lanes (i = l & 15), lane i receives element (i & 3)
// of the lanes 4*j + (i >> 2), j = 0..3.  With lane i pointing at row (i >> 2), columns 4*(i & 3).. of a [4][16] block,
// lane i therefore receives column i of the block, rows 0..3: a transposing fragment read.
typedef short pp_sv4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ h4 lds_read_tr16(const half_t* p) {
  const pp_sv4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pp_sv4*)p);
  return __builtin_bit_cast(h4, r);
}
// The same read as an instruction the compiler does not see: next to global_load_lds copies hipcc waits vmcnt(0) before
// every ds_read_tr16 builtin (SIInsertWaitcnts cannot tell that the copy in flight targets the OTHER stage) and so
// serialises copy and compute.  `dst` is NOT valid until lds_tr16_wait<N>() names it (N = LDS reads allowed to stay in flight).
template <int OFF>
__device__ __forceinline__ void lds_tr16_issue(h4& dst, const void* lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
               : "=v"(dst)
               : "v"((unsigned)(uintptr_t)lds_addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_tr16_wait(h4& a, h4& b, h4& c, h4& d, h4& e, h4& f, h4& g, h4& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
// max / sum of a value over the lane pair (l, l ^ 32): v_permlane32_swap(x, x) returns {x of the low half, x of the high half}
__device__ __forceinline__ float pair32_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float pair32_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// LDS writes of this wave visible to its other lanes: the hardware executes a wave's LDS operations in order, so only the
// counter wait is needed; the emulator runs lanes as independent fibers and needs a rendezvous here
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
inline f16v mfma_32x32x16_f16(h8 a, h8 b, f16v c) {
  struct Slot {
    h8 a, b;
    unsigned char pad[32];
  };
  Slot* s = reinterpret_cast<Slot*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l].a = a;
  s[l].b = b;
  pp_emu::wave_sync();
  const int col = l & 31;
  f16v d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = 0.f;
    for (int g = 0; g < 2; ++g) {
      const h8 av = s[row + 32 * g].a;
      const h8 bv = s[col + 32 * g].b;
      for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
    }
    d[r] += acc;
  }
  pp_emu::wave_sync();
  return d;
}
inline h4 lds_read_tr16(const half_t* p) {
  struct Slot {
    const half_t* p;
    unsigned char pad[56];
  };
  Slot* s = reinterpret_cast<Slot*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l].p = p;
  pp_emu::wave_sync();
  const int G = l >> 4, i = l & 15;
  h4 r;
  for (int j = 0; j < 4; ++j) r[j] = s[16 * G + 4 * j + (i >> 2)].p[i & 3];
  pp_emu::wave_sync();
  return r;
}
template <int OFF>
inline void lds_tr16_issue(h4& dst, const void* lds_addr) {
  dst = lds_read_tr16(reinterpret_cast<const half_t*>(static_cast<const unsigned char*>(lds_addr) + OFF));
}
template <int N>
inline void lds_tr16_wait(h4&, h4&, h4&, h4&, h4&, h4&, h4&, h4&) {}
inline float pair32_max(float v) { return fmaxf(v, shfl_xor(v, 32)); }
inline float pair32_sum(float v) {
  const float o = shfl_xor(v, 32);
  return (pp_emu::cur->lane & 32) ? o + v : v + o;  // low half first, as on the device
}
inline float fast_exp2(float x) { return exp2f(x); }
inline int wave_uniform(int v) { return v; }
inline void wave_lds_fence() { pp_emu::wave_sync(); }
inline bool wave_any(bool p) {
  float f = p ? 1.f : 0.f;
  for (int m = 1; m < 64; m <<= 1) f = fmaxf(f, shfl_xor(f, m));
  return f != 0.f;
}
#endif

}  // namespace pp

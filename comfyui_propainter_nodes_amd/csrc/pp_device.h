// pp_device.h -- device-side vocabulary shared by every kernel of libpropainter_mi355.
//
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, MFMA fragments as documented in
// /opt/skills/guides/cdna_hip_programming.md section 3.  When PP_EMU is defined (tests only,
// tests/emu/) the same sources are compiled for x86 and the wave-level primitives below
// are replaced by rendezvous emulations with identical lane->element maps.
#pragma once

#ifdef PP_EMU
#include "pp_emu.h"
#else
#include <hip/hip_runtime.h>
#define PP_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
extern __shared__ __attribute__((aligned(16))) unsigned char pp_dyn_smem_[];
#define PP_DYN_SMEM (pp_dyn_smem_)
#endif

#include <stdint.h>

namespace pp {

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// XCD-aware work-group order: the hardware deals the work-groups of a 1-D grid to the 8 XCDs round robin (id & 7), each
// XCD with its own 4 MiB L2.  Logical block L makes XCD x process a CONTIGUOUS eighth of the blocks, so neighbouring
// blocks -- which read neighbouring pixels -- share an L2 instead of spreading every line over all eight.
#ifdef PP_EMU
inline
#else
__device__ __forceinline__
#endif
int xcd_contiguous_block(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------
// MFMA wrappers.
//   16x16x32 f16 : lane l holds A[row = l&15][k = 8*(l>>4) .. +7] and
//                  B[k = 8*(l>>4) .. +7][col = l&15];
//   16x16x4  f32 : lane l holds A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
//   C/D (both)   : lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3.
// ---------------------------------------------------------------------------------------
// (The PP_ABLATE profiling hooks of rounds 1-2 -- compile one ingredient of the convolution kernels out and read its cost off
// the timing difference -- are no longer part of the product sources: tools/ablate_hooks.patch re-inserts them into a scratch
// copy of csrc/ for tools/ablate_*.sh.)
#ifndef PP_EMU
__device__ __forceinline__ f4 mfma_16x16x32_f16(h8 a, h8 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma_16x16x4_f32(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 32x32x2 f32: lane l holds A[row = l&31][k = l>>5], B[k = l>>5][col = l&31];
// C/D: lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r = 0..15.
__device__ __forceinline__ f16v mfma_32x32x2_f32(float a, float b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4): lane l of the wave writes
// lds_wave_base + 16*l; the base must be wave-uniform.  Completion is tracked by vmcnt (a following
// __syncthreads() waits for it).  The source address is per lane.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same copy with a wave-uniform 64-bit base in SGPRs and an unsigned 32-bit per-lane BYTE offset (the instruction's saddr
// form): no 64-bit per-lane address -- hipcc builds one with a v_lshl_add_u64 per copy from the builtin above, and next to a
// matrix-bound partner wave a vector-ALU instruction costs about one MFMA slot (r05).  lds_wave_base must be wave-uniform.
// Not visible to the compiler's vmcnt accounting: for kernels that count their waits by hand (conv_halo.hip).
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_wave_off) {
  // lds_wave_off: the wave's destination as an LDS BYTE OFFSET (lds_offset_of() once per kernel + integer arithmetic): a pointer
  // here would cost a generic -> LDS cast with its null check (two scalar instructions per copy)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase),
               "s"(__builtin_amdgcn_readfirstlane(lds_wave_off))
               : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_offset_of(const void* lds_ptr) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_ptr;
}
__device__ __attribute__((aligned(16))) const unsigned int pp_zero16[4] = {0u, 0u, 0u, 0u};
// counted wait on the vector-memory queue (global_load_lds copies are tracked by vmcnt) and a bare barrier that
// does NOT drain that queue (unlike __syncthreads()), so copies can stay in flight across it
template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// A 16-byte global load the compiler does not see (inline asm): next to global_load_lds copies hipcc would wait
// vmcnt(0) at the first use of an ordinary load's result and drain the whole pipeline.  The destination is NOT valid
// until wait_vmcnt_hidden<N>() has been executed (the caller counts the vector-memory queue by hand).
__device__ __forceinline__ void gload16_hidden(f4& dst, const void* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
// the same with a wave-uniform base in SGPRs and an unsigned 32-bit per-lane BYTE offset: no 64-bit address arithmetic
__device__ __forceinline__ void gload16_hidden_s(f4& dst, const void* sbase, uint32_t byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(sbase) : "memory");
}
// counted wait for hidden loads + a scheduling barrier: no instruction (in particular no consumer of a hidden
// load's destination) is moved across it.  The destinations are defined once and only read afterwards, so the
// register allocator has no reason to copy them while the data is in flight (audited in the -save-temps output).
template <int N>
__device__ __forceinline__ void wait_vmcnt_hidden() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// LDS writes / reads of this wave retired (what __syncthreads() would wait for besides vmcnt)
__device__ __forceinline__ void pp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the instruction scheduler moves nothing across this point (emits no code)
__device__ __forceinline__ void pp_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Work-group barrier that does NOT drain the vector-memory queue (LDS-DMA copies stay in flight across it) but DOES retire this
// wave's LDS reads first.  r04: without the lgkmcnt(0) hipcc is free to sink the tail of a step -- the last fragment reads'
// `s_waitcnt lgkmcnt` and the MFMAs behind it -- BELOW the s_barrier (it did, in conv_halo_f16_ct_kernel: two ds_read_b128 of
// the weight stage in flight across the barrier).  The next step's global_load_lds then restages that buffer one barrier after
// reads that have not returned: a write-after-read race that never showed on a quiet chip (an LDS read returns long before a copy
// lands) and corrupted ~0.2 % of the launches once another stream kept the CUs' LDS queues busy (tools/diag_kernels_under_load.py:
// whole weight-fragment rows stale for one wave).  The rule (cdna_hip_programming.md, LDS-DMA staging): restage a buffer one
// phase after its last read only when an lgkmcnt before the barrier retired those reads.
__device__ __forceinline__ void pp_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// Lanes of ONE wave hand data to each other through a wave-private LDS region: the wave executes its LDS instructions in
// program order, so no s_barrier is needed -- but the COMPILER must not move a lane's reads above the other lanes' writes: per
// thread the two addresses differ, so nothing in the C++ memory model orders them.  r05 first used wavefront-scope fences +
// __builtin_amdgcn_wave_barrier() here; LLVM lowers those to nothing and keeps its freedom: one build of the flat PP_F32X2 kernel
// read its staging rows before writing them (stale weight bytes as floats: 4e32 in tests/test_conv.py on the MI355X; every earlier
// build had happened to keep the order).  An asm statement with a memory clobber is a barrier the optimiser cannot see through;
// the lgkmcnt(0) in it retires the writes before the reads are issued.
__device__ __forceinline__ void pp_wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl_idx(float v, int src) { return __shfl(v, src, 64); }
// v_permlane16_swap: rows are 16 lanes; row 1 of `a` <-> row 0 of `b`, row 3 of `a` <-> row 2 of `b` (one VALU instruction, gfx950)
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  a = __builtin_bit_cast(float, (unsigned)r[0]);
  b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#else
inline void pp_wave_lds_fence() { pp_emu::wave_sync(); }
inline void glds16(const void* g, void* lds_wave_base) {
  memcpy(static_cast<unsigned char*>(lds_wave_base) + 16 * pp_emu::cur->lane, g, 16);
}
static const unsigned int pp_zero16[4] = {0u, 0u, 0u, 0u};
inline uint32_t lds_offset_of(const void* lds_ptr) {   // (emulator: offsets are relative to the block's dynamic LDS)
  return (uint32_t)(static_cast<const unsigned char*>(lds_ptr) - pp_emu::dyn_smem);
}
inline void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_wave_off) {
  memcpy(pp_emu::dyn_smem + lds_wave_off + 16 * pp_emu::cur->lane, static_cast<const char*>(sbase) + voff, 16);
}
template <int N>
inline void pp_wait_vmcnt() {}
inline void pp_wait_lgkm0() {}
inline void pp_sched_fence() {}
inline void gload16_hidden(f4& dst, const void* src) { memcpy(&dst, src, 16); }
inline void gload16_hidden_s(f4& dst, const void* sbase, uint32_t byte_off) { memcpy(&dst, static_cast<const char*>(sbase) + byte_off, 16); }
template <int N>
inline void wait_vmcnt_hidden() {}
inline void pp_barrier() { pp_emu::barrier(); }
inline f4 mfma_16x16x32_f16(h8 a, h8 b, f4 c) {
  struct Slot {
    h8 a, b;
    unsigned char pad[32];
  };
  Slot* s = reinterpret_cast<Slot*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l].a = a;
  s[l].b = b;
  pp_emu::wave_sync();
  const int col = l & 15;
  f4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = 0.f;
    for (int g = 0; g < 4; ++g) {
      const h8 av = s[row + 16 * g].a;
      const h8 bv = s[col + 16 * g].b;
      for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
    }
    d[r] += acc;
  }
  pp_emu::wave_sync();
  return d;
}
inline f4 mfma_16x16x4_f32(float a, float b, f4 c) {
  struct Slot {
    float a, b;
    unsigned char pad[56];
  };
  Slot* s = reinterpret_cast<Slot*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l].a = a;
  s[l].b = b;
  pp_emu::wave_sync();
  const int col = l & 15;
  f4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int g = 0; g < 4; ++g) acc = fmaf(s[row + 16 * g].a, s[col + 16 * g].b, acc);
    d[r] = acc;
  }
  pp_emu::wave_sync();
  return d;
}
inline f16v mfma_32x32x2_f32(float a, float b, f16v c) {
  struct Slot {
    float a, b;
    unsigned char pad[56];
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
    float acc = d[r];
    for (int g = 0; g < 2; ++g) acc = fmaf(s[row + 32 * g].a, s[col + 32 * g].b, acc);
    d[r] = acc;
  }
  pp_emu::wave_sync();
  return d;
}
inline float shfl_xor(float v, int mask) {
  float* s = reinterpret_cast<float*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l * 16] = v;
  pp_emu::wave_sync();
  const float r = s[((l ^ mask) & 63) * 16];
  pp_emu::wave_sync();
  return r;
}
inline float shfl_idx(float v, int src) {
  float* s = reinterpret_cast<float*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l * 16] = v;
  pp_emu::wave_sync();
  const float r = s[(src & 63) * 16];
  pp_emu::wave_sync();
  return r;
}
inline void swap_rows16(float& a, float& b) {
  struct Slot {
    float a, b;
    unsigned char pad[56];
  };
  Slot* s = reinterpret_cast<Slot*>(pp_emu::wave_scratch());
  const int l = pp_emu::cur->lane;
  s[l].a = a;
  s[l].b = b;
  pp_emu::wave_sync();
  const bool odd = (l >> 4) & 1;
  const float na = odd ? s[l - 16].b : a, nb = odd ? b : s[l + 16].a;
  pp_emu::wave_sync();
  a = na;
  b = nb;
}
inline int lane_id() { return pp_emu::cur->lane; }
#endif

// ---------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(half_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) {
  return v;
}
// f32 -> f16 with SATURATION: a value beyond the f16 range is stored as +-65504 instead of +-Inf (one v_med3_f32 in front of the
// round-to-nearest convert; identity inside the range).  The two learned recurrences run 80-160 dependent steps on f16 tensors:
// with weights that are not contractive their activations grow geometrically, and one Inf turns into NaN at the next
// Inf - Inf or 0 x Inf (bilinear blends, residual adds) and then into every pixel downstream.  Saturated values stay finite
// (r04; tests/test_rfc.py::test_undamped_recurrences_saturate).  A NaN input stays NaN (r05).
__device__ __forceinline__ half_t sat_half(float v) {
  // (v_med3_f32 / fminf(fmaxf()) return a bound for a NaN input, which would turn an upstream NaN bug into a
  //  plausible -65504.  v_max_f32 / v_min_f32 with IEEE mode on return the OTHER operand only for a quiet NaN in one of them and
  //  so cannot be used either: select explicitly -- a NaN stays a NaN.)
#ifdef PP_EMU
  return v != v ? (half_t)v : (half_t)fminf(fmaxf(v, -65504.f), 65504.f);
#else
  const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  return (half_t)(v != v ? v : c);
#endif
}
template <>
__device__ __forceinline__ half_t from_f32<half_t>(float v) {
  return sat_half(v);
}

// two floats -> two f16, round toward zero (saturates at +-65504 instead of overflowing to infinity)
__device__ __forceinline__ h2 cvt_pkrtz_f16(float a, float b) {
#ifdef PP_EMU
  auto rtz = [](float x) -> half_t {
    if (x != x) return (half_t)x;
    if (x > 65504.f) return (half_t)65504.f;
    if (x < -65504.f) return (half_t)-65504.f;
    half_t h = (half_t)x;  // round to nearest even
    const float hf = (float)h;
    if ((hf > x && x > 0.f) || (hf < x && x < 0.f)) {  // rounded away from zero: step one ulp back
      unsigned short bits;
      memcpy(&bits, &h, 2);
      bits -= 1;
      memcpy(&h, &bits, 2);
    }
    return h;
  };
  return h2{rtz(a), rtz(b)};
#else
  return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
#endif
}

// The PP_F32X2 operand split of two values (f32 convolutions on the f16 matrix pipe).
__device__ __forceinline__ void split_pair(float c0, float c1, h2& hh, h2& ll) {
  // r05: v ~ h + l with h = f16_rtz(v), l = f16_rtz(v - h) UNSCALED: the matrix pipe honours f16 subnormal inputs
  // (tools/probes/mfma_denorm.hip, measured on the MI355X), so the low term needs no 2^11 scale to survive, the three products of a
  // multiply-add share one fp32 accumulator (64 registers instead of 128 per wave tile) and the split is 2 vector operations per
  // value instead of 4 (packed convert, exact fma, packed convert).  Precision: |v - h - l| <= 2^-20 |v| (both conversions
  // truncate), and never more than 2^-24 absolute below |v| = 2^-4 (the f16 subnormal step); beyond the f16 range h saturates at
  // 65504 and l at 65504: |v| up to 131008 keeps an absolute error <= 32, larger values saturate -- never Inf / NaN.
  hh = cvt_pkrtz_f16(c0, c1);
  const float t0 = __builtin_fmaf((float)hh[0], -1.f, c0), t1 = __builtin_fmaf((float)hh[1], -1.f, c1);
  ll = cvt_pkrtz_f16(t0, t1);
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence
__device__ __forceinline__ float fast_rcp(float x) {
#ifdef PP_EMU
  return 1.f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.f + __expf(-x)); }
// tanh(x) = sign(x) (1 - e^{-2|x|}) / (1 + e^{-2|x|}); absolute error < 1e-7
__device__ __forceinline__ float tanhf_(float x) {
  const float t = __expf(-2.f * fabsf(x));
  const float r = (1.f - t) * fast_rcp(1.f + t);
  return copysignf(r, x);
}

__device__ __forceinline__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 8 consecutive elements of an f16 / f32 activation tensor as floats (one 16-byte / two 16-byte accesses): the storage
// type of the two f16-capable networks is f16 for the node's fp16 "enable" and f32 for "disable".
__device__ __forceinline__ void ld8(const half_t* p, float* f) {
  const h8 v = *reinterpret_cast<const h8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ void ld8(const float* p, float* f) {
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = a[i];
    f[4 + i] = b[i];
  }
}
__device__ __forceinline__ void st8(half_t* p, const float* f) {
  h8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = sat_half(f[i]);
  *reinterpret_cast<h8*>(p) = v;
}
__device__ __forceinline__ void st8(float* p, const float* f) {
  *reinterpret_cast<f4*>(p) = f4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<f4*>(p + 4) = f4{f[4], f[5], f[6], f[7]};
}
__device__ __forceinline__ h8 ld8h(const half_t* p) { return *reinterpret_cast<const h8*>(p); }
__device__ __forceinline__ h8 ld8h(const float* p) {  // fp32 storage, f16 MFMA operand
  float f[8];
  ld8(p, f);
  h8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = sat_half(f[i]);
  return v;
}

}  // namespace pp

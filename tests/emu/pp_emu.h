// pp_emu.h -- TEST INFRASTRUCTURE ONLY (never loaded by the product path).
//
// A tiny SPMD emulator that lets the *same* kernel sources under
// comfyui_propainter_nodes_amd/csrc/ be compiled for x86 with the ROCm host
// clang (-DPP_EMU) and executed in the GPU-less build container, so that the
// index arithmetic of every kernel (gathers, LDS tiling, MFMA fragment use,
// barriers) can be checked against oracle/ before GPU minutes are spent.
//
// Model: blocks of a launch are distributed over a few OS worker threads; the
// threads of one block are cooperative fibers (ucontext) on one OS thread.
// __syncthreads() and the wave-level exchanges (shuffles, MFMA) are rendezvous
// points between fibers.  `__shared__` becomes `static thread_local`, which is
// exactly "one instance per resident block" because a worker runs one block at
// a time.  The MFMA wrappers implement the gfx950 fragment layouts documented
// in /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct pp_emu_uint3 {
  unsigned x, y, z;
};
typedef void* hipStream_t;

namespace pp_emu {
struct ThreadCtx {
  pp_emu_uint3 tid;
  pp_emu_uint3 bid;
  dim3 bdim;
  dim3 gdim;
  int linear_tid;
  int lane;
  int wave;
};
extern thread_local ThreadCtx* cur;
extern thread_local unsigned char* dyn_smem;

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void barrier();
void wave_sync();
// 64 lanes x 64 bytes of exchange space per wave
unsigned char* wave_scratch();
int active_lanes_in_wave();
}  // namespace pp_emu

#define threadIdx (pp_emu::cur->tid)
#define blockIdx (pp_emu::cur->bid)
#define blockDim (pp_emu::cur->bdim)
#define gridDim (pp_emu::cur->gdim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __syncthreads() pp_emu::barrier()

#define __expf(x) expf(x)

#define PP_LAUNCH(kernel, grid, block, smem, stream, ...) \
  pp_emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define PP_DYN_SMEM (pp_emu::dyn_smem)

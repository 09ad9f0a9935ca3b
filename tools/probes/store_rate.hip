// store_rate.hip -- what does a work-group pay to write its 64 KB output tile?  (r05: the phase trace of the PP_F32X2 halo kernels
// shows 10-14 k cycles for the 16 dwordx4 stores per wave of the epilogue, whatever their coalescing.)
// Every wave issues 16 global_store_dwordx4 (1 KB each); s_memtime before the first, after the last ISSUE, and after vmcnt(0).
//   PATTERN 0: each store instruction writes 1 KB contiguous (16 lanes = 256 B of one pixel row, 4 rows)      [transposed epilogue]
//   PATTERN 1: each store instruction writes 16 pieces of 64 B, 1 KB apart                                      [direct MFMA layout]
//   POLICY 0 default, 1 nt, 2 sc1, 3 sc0 sc1
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o tools/probes/store_rate tools/probes/store_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int POLICY>
__global__ void __launch_bounds__(256) probe(float* out, unsigned* tr, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // some MFMA-free busy work first so that work-groups do not all store at t = 0 (spin differs per block)
  float acc = (float)lane;
  for (int i = 0; i < spin * (1 + (int)(blockIdx.x % 7)); ++i) acc = acc * 1.0001f + 0.5f;
  f4 v = {acc, acc + 1.f, acc + 2.f, acc + 3.f};
  // tile of this work-group: 64 pixels x 256 channels f32 = 64 KB; wave w owns channels [64 w, 64 w + 64)
  char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * 65536;
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    char* dst;
    if (PATTERN == 0) dst = base + ((s * 4 + (lane >> 4)) * 1024) + wave * 256 + (lane & 15) * 16;          // 4 pixels x 256 B
    else dst = base + (((s >> 2) * 16 + (lane & 15)) * 1024) + wave * 256 + (s & 3) * 64 + (lane >> 4) * 16;  // 16 pixels x 64 B
    if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
    if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
    if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned t2 = (unsigned)__builtin_amdgcn_s_memtime();
  if (lane == 0) {
    tr[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
    tr[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0;
  }
}

template <int PATTERN, int POLICY>
static void run(int grid, int spin, float* out, unsigned* tr) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<PATTERN, POLICY><<<grid, 256>>>(out, tr, spin); hipDeviceSynchronize();
  hipEventRecord(e0); probe<PATTERN, POLICY><<<grid, 256>>>(out, tr, spin); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h((size_t)grid * 8);
  hipMemcpy(h.data(), tr, h.size() * 4, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < grid * 4; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  printf("{\"pattern\": %d, \"policy\": %d, \"grid\": %d, \"spin\": %d, \"issue_ticks\": %.0f, \"drained_ticks\": %.0f, \"ms\": %.4f, \"GB_s\": %.0f}\n",
         PATTERN, POLICY, grid, spin, a / (grid * 4), b / (grid * 4), ms, grid * 65536.0 / ms / 1e6);
}

int main() {
  const int maxgrid = 9480;
  float* out; unsigned* tr;
  hipMalloc(&out, (size_t)maxgrid * 65536); hipMalloc(&tr, (size_t)maxgrid * 8 * 4);
  for (int spin : {0, 4000})
    for (int grid : {1, 256, 512, 9480}) {
      run<0, 0>(grid, spin, out, tr); run<1, 0>(grid, spin, out, tr);
      run<0, 1>(grid, spin, out, tr); run<0, 2>(grid, spin, out, tr); run<0, 3>(grid, spin, out, tr);
    }
  return 0;
}

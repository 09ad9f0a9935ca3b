// dispatch_rate.hip -- how fast does the MI355X start work-groups of a given shape?  (r06: the SQ counters of conv_gemm_f16_kernel show
// 0.74 resident waves per SIMD over an 85-us launch whose 1 285 work-groups of 8 waves each live ~6 us: is the launch paced by the
// DISPATCH of 512-thread / 72-KB work-groups rather than by their execution?)
// Every work-group spins for `spin` cycles (s_memtime) and records its start time, its XCC and CU; the host prints the launch duration
// (HIP events), and from the stamps the spread of start times per XCC: with free slots everywhere, work-group i of an XCC should start
// as soon as a slot frees up.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o tools/probes/dispatch_rate tools/probes/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

extern __shared__ unsigned char smem[];

__global__ void probe(unsigned long long* stamps, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) smem[0] = 1;
  unsigned long long t = t0;
  while ((long long)(t - t0) < spin) t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[blockIdx.x * 3] = t0;
    stamps[blockIdx.x * 3 + 1] = t;
    stamps[blockIdx.x * 3 + 2] = xcc & 0xf;
  }
}

int main(int argc, char** argv) {
  struct Cfg { int grid, block, lds, spin; const char* what; };
  // s_memtime counts shader-clock ticks on gfx950 (r05 phase traces): spin is in ticks (12 600 ticks ~ 6 us = a GEMM work-group's life)
  const Cfg cfgs[] = {
      {1446, 512, 72 * 1024, 12600, "GEMM qkv: 1446 x 512 threads x 72 KB, 6 us each"},
      {1446, 512, 72 * 1024, 0, "same shape, empty work-groups"},
      {1446, 256, 36 * 1024, 12600, "1446 x 256 threads x 36 KB, 6 us"},
      {2892, 256, 36 * 1024, 6300, "2892 x 256 threads x 36 KB, 3 us"},
      {1446, 512, 1024, 12600, "1446 x 512 threads x 1 KB, 6 us"},
      {225, 1024, 96 * 1024, 3150, "split-K step: 225 x 1024 threads x 96 KB, 1.5 us"},
      {225, 1024, 96 * 1024, 0, "same shape, empty"},
      {512, 512, 72 * 1024, 12600, "one round: 512 x 512 threads x 72 KB, 6 us"},
  };
  unsigned long long* d;
  hipMalloc(&d, 8192 * 3 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (const Cfg& c : cfgs) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe, dim3(c.grid), dim3(c.block), c.lds, 0, d, c.spin);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe, dim3(c.grid), dim3(c.block), c.lds, 0, d, c.spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(c.grid * 3);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0, emax = 0;
    int per_xcc[16] = {0};
    for (int i = 0; i < c.grid; ++i) {
      tmin = std::min(tmin, h[i * 3]);
      tmax = std::max(tmax, h[i * 3]);
      emax = std::max(emax, h[i * 3 + 1]);
      per_xcc[h[i * 3 + 2] & 15]++;
    }
    const double rounds = (double)c.grid / (256.0 * std::max(1, std::min(160 * 1024 / std::max(c.lds, 1), 2048 / c.block)));
    printf("%-58s: %7.1f us per launch | last start %8llu ticks, last end %8llu ticks after the first start | %.2f rounds x %d ticks = %.0f ticks if slots refill at once | WGs on XCC0..7:",
           c.what, ms * 1000.0 / reps, tmax - tmin, emax - tmin, rounds, c.spin, ceil(rounds) * c.spin);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
  }
  return 0;
}

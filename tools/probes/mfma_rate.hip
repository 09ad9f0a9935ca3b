// mfma_rate.hip -- what the gfx950 matrix pipe SUSTAINS, per instruction shape and operand content (r04).
// Every SIMD issues independent MFMAs back to back from registers (no memory traffic), 2 waves per SIMD, ~2-4 ms per case.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_rate tools/probes/mfma_rate.hip && tools/probes/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16, 3: 16x16x32 bf16;  DATA 0: small smooth values, 1: random bits
template <int MODE, int DATA>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    if (DATA == 0) { a[j] = (half_t)(0.001f * (threadIdx.x + j)); b[j] = (half_t)(0.002f * ((int)threadIdx.x - j)); }
    else { a[j] = (half_t)(((int)(lcg(s) & 0xffff) - 32768) * (1.f / 16384.f)); b[j] = (half_t)(((int)(lcg(s) & 0xffff) - 32768) * (1.f / 16384.f)); }
  }
  b8 ab = __builtin_bit_cast(b8, a), bb = __builtin_bit_cast(b8, b);
  float acc = 0.f;
  // inline asm: hipcc rotates builtin accumulators of such a loop through overlapping AGPR ranges (~40 copies per 8 MFMAs), so
  // the builtin form measured the copies, not the pipe
  if constexpr (MODE == 0 || MODE == 2) {
    f16v c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
#define PP_M32(c) if constexpr (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
                  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(ab), "v"(bb));
    for (int it = 0; it < iters; ++it) { PP_M32(c0) PP_M32(c1) PP_M32(c2) PP_M32(c3) }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i] + c2[i] + c3[i];
  } else {
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
#define PP_M16(c) if constexpr (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
                  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(ab), "v"(bb));
    for (int it = 0; it < iters; ++it) { PP_M16(c0) PP_M16(c1) PP_M16(c2) PP_M16(c3) PP_M16(c4) PP_M16(c5) PP_M16(c6) PP_M16(c7) }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    f4 t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    acc += t[0] + t[1] + t[2] + t[3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int DATA>
static void run(const char* name, float* dout) {
  const int nblk = 512, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE, DATA><<<nblk, 256>>>(dout, 100); hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); rate<MODE, DATA><<<nblk, 256>>>(dout, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 4 * 32768.0;   // (4 x 32x32x16 = 8 x 16x16x32 per iteration)
    printf("%-34s %.3f ms  %.0f TFLOP/s\n", name, ms, flop / ms / 1e9);
  }
}

int main() {
  float* dout; hipMalloc(&dout, 512 * 256 * 4);
  run<0, 0>("32x32x16 f16, smooth operands", dout);
  run<1, 0>("16x16x32 f16, smooth operands", dout);
  run<0, 1>("32x32x16 f16, random operands", dout);
  run<1, 1>("16x16x32 f16, random operands", dout);
  run<2, 1>("32x32x16 bf16, random operands", dout);
  run<3, 1>("16x16x32 bf16, random operands", dout);
  return 0;
}

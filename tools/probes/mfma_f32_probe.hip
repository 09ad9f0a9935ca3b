// Standalone probe (not part of the library): where does an f32 MFMA tile loop lose time on gfx950?
//   mode 0: 64 x v_mfma_f32_32x32x2_f32 per chunk, operands in registers
//   mode 1: + 16 ds_read_b128 per chunk (operands from LDS, swizzled like conv_igemm)
//   mode 2: + 8 ds_write_b128 per thread per chunk + __syncthreads (double-buffered LDS)
//   mode 3: + 8 global_load_dwordx4 per thread per chunk (streaming, prefetch one chunk ahead)
// hipcc --offload-arch=gfx950 -O3 mfma_f32_probe.hip -o mfma_f32_probe && ./mfma_f32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ g, float* __restrict__ out, int nch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][256 rows][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kh = lane >> 5;
  f16v acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f4 regs[8];
  for (int i = 0; i < 8; ++i) regs[i] = f4{(float)tid, 1.f, 2.f, 3.f};
  const float* gp = g + ((size_t)blockIdx.x * 256 + tid) * 4;
  const size_t gstride = (size_t)gridDim.x * 256 * 4;
  if (MODE >= 1) {
    for (int i = 0; i < 16; ++i) *(f4*)&lds[(i * 256 + tid) * 4] = regs[i & 7];
    __syncthreads();
  }
  for (int q = 0; q < nch; ++q) {
    const int buf = q & 1;
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) regs[i] = *(const f4*)(gp + (size_t)((q * 8 + i) & 63) * gstride);
    }
    const float* xs = lds + buf * 8192 + ((wave & 1) * 64 + r32) * 32;
    const float* ws = lds + buf * 8192 + 4096 + ((wave >> 1) * 64 + r32) * 32;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f4 af[2], bf[2];
      if (MODE >= 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = *(const f4*)(ws + a * 32 * 32 + ((sub * 2 + kh) ^ (r32 & 7)) * 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = *(const f4*)(xs + b * 32 * 32 + ((sub * 2 + kh) ^ (r32 & 7)) * 4);
      } else {
        af[0] = regs[0]; af[1] = regs[1]; bf[0] = regs[2]; bf[1] = regs[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
    if (MODE >= 2) {
      float* dst = lds + (buf ^ 1) * 8192;
      const int row = tid >> 3, pc = tid & 7;
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f4*)&dst[(row + i * 32) * 32 + ((pc ^ (row & 7)) * 4)] = regs[i];
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const float* g, float* out, int blocks, int nch) {
  hipFuncSetAttribute((const void*)&probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256, 65536>>>(g, out, nch);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) probe<MODE><<<blocks, 256, 65536>>>(g, out, nch);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = 2.0 * 128 * 128 * 32 * (double)nch * blocks;
  printf("mode %d blocks %d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, ms, flops / ms / 1e9);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)2048 * 256 * 4 * 64 * 4);
  hipMalloc(&out, (size_t)8192 * 256 * 4);
  hipMemset(g, 0, (size_t)2048 * 256 * 4 * 64 * 4);
  for (int blocks : {512, 2048}) {
    run<0>(g, out, blocks, 60); run<1>(g, out, blocks, 60); run<2>(g, out, blocks, 60); run<3>(g, out, blocks, 60);
  }
  return 0;
}

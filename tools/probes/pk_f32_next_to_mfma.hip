// pk_f32_next_to_mfma.hip -- r06: deform_cols_kernel (csrc/sample_kernels.hip) computes WRONG values (the solo run agrees with a host
// recomputation) whenever one of the library's MFMA convolution kernels runs next to it on a second stream; torch's elementwise
// kernels next to it do no harm, and flow_warp next to the same convolutions is fine.  deform_cols's ISA differs from flow_warp's in
// its packed-fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32).  Minimal question: does a kernel of packed fp32 VALU math give different
// results when a pure-MFMA kernel shares the CUs?
//   victim PK : y[i] = sum_j (a_j * x[i] + b_j)   on float2 pairs -> v_pk_mul_f32 / v_pk_add_f32   (contract off)
//   victim SC : the same arithmetic, one float at a time (an asm barrier per element keeps hipcc from pairing them)
//   neighbour : every wave issues v_mfma_f32_16x16x32_f16 back to back from registers (no memory traffic) for ~200 us
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result -o tools/probes/pk_f32_next_to_mfma tools/probes/pk_f32_next_to_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// victim SWZ: a packed add whose second operand is SWAPPED (hipcc: v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]) -- the instruction
// deform_cols uses to add the (x, y) flow to its (dy, dx) offsets
__global__ void __launch_bounds__(256) victim_swz(const float* __restrict__ x, float* __restrict__ y, int n, int rounds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i * 8 >= n) return;
  f2 v[4], acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = *reinterpret_cast<const f2*>(x + i * 8 + 2 * e);
    acc[e] = f2{0.f, 0.f};
  }
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f2 t = v[e];
      asm volatile("" : "+v"(t));          // (keeps the swizzle a register-level operand select of the add)
      acc[e] = acc[e] + f2{t[1], t[0]};
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) *reinterpret_cast<f2*>(y + i * 8 + 2 * e) = acc[e];
}

template <bool PACKED>
__global__ void __launch_bounds__(256) victim(const float* __restrict__ x, float* __restrict__ y, int n, int rounds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i * 8 >= n) return;
  f2 v[4], acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = *reinterpret_cast<const f2*>(x + i * 8 + 2 * e);
    acc[e] = f2{0.f, 0.f};
  }
  for (int r = 0; r < rounds; ++r) {
    const float a = 1.0f + 0.001f * (float)(r & 7), b = 0.5f - 0.01f * (float)(r & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (PACKED) {
        const f2 t = v[e] * f2{a, a};     // v_pk_mul_f32
        acc[e] = acc[e] + t;              // v_pk_add_f32
        acc[e] = acc[e] + f2{b, b};
      } else {
        float t0 = v[e][0] * a, t1 = v[e][1] * a;
        asm volatile("" : "+v"(t0));
        asm volatile("" : "+v"(t1));
        float s0 = acc[e][0] + t0, s1 = acc[e][1] + t1;
        asm volatile("" : "+v"(s0));
        asm volatile("" : "+v"(s1));
        acc[e][0] = s0 + b;
        acc[e][1] = s1 + b;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) *reinterpret_cast<f2*>(y + i * 8 + 2 * e) = acc[e];
}

__global__ void __launch_bounds__(256) mfma_neighbour(float* sink, int iters) {
  h8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (float)(threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (float)(threadIdx.x * 3 + e));
  }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[threadIdx.x] = c0[0];
}

int main() {
  const int n = 1 << 24, rounds = 64;
  float *x, *y, *sink;
  hipMalloc(&x, n * 4);
  hipMalloc(&y, n * 4);
  hipMalloc(&sink, 4096);
  std::vector<float> hx(n), ref(n), got(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  for (int packed = 2; packed >= 0; --packed) {
    auto run = [&](hipStream_t s) {
      if (packed == 2) hipLaunchKernelGGL(victim_swz, dim3(n / 8 / 256), dim3(256), 0, s, x, y, n, rounds);
      else if (packed) hipLaunchKernelGGL(victim<true>, dim3(n / 8 / 256), dim3(256), 0, s, x, y, n, rounds);
      else hipLaunchKernelGGL(victim<false>, dim3(n / 8 / 256), dim3(256), 0, s, x, y, n, rounds);
    };
    run(s1);
    hipDeviceSynchronize();
    hipMemcpy(ref.data(), y, n * 4, hipMemcpyDeviceToHost);
    int bad_runs = 0;
    long bad_vals = 0;
    for (int it = 0; it < 10; ++it) {
      hipMemsetAsync(y, 0, n * 4, s1);
      hipDeviceSynchronize();
      for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(mfma_neighbour, dim3(2048), dim3(256), 0, s2, sink, 20000);
      run(s1);
      hipDeviceSynchronize();
      hipMemcpy(got.data(), y, n * 4, hipMemcpyDeviceToHost);
      long d = 0;
      for (int i = 0; i < n; ++i) d += memcmp(&got[i], &ref[i], 4) != 0;
      bad_runs += d != 0;
      bad_vals += d;
    }
    printf("victim %s next to the MFMA kernel: %d of 10 runs differ from the solo run (%ld values)\n", packed == 2 ? "SWIZZLED packed add (op_sel)" : packed ? "PACKED (v_pk_mul_f32 / v_pk_add_f32)" : "SCALAR", bad_runs, bad_vals);
  }
  return 0;
}

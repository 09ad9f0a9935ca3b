// pk_op_sel_forms.hip -- r06 follow-up of pk_f32_next_to_mfma.hip: WHICH packed VOP3P forms return wrong values next to another
// kernel's MFMA waves?  Every victim repeats ONE instruction (inline asm, so the form is exactly the one named) on register operands;
// the neighbour issues v_mfma_f32_16x16x32_f16 back to back on a second stream.  A victim's result next to the neighbour is compared
// with its own solo run, bit for bit, 10 times.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/pk_op_sel_forms tools/probes/pk_op_sel_forms.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define FORMS(X)                                                                                              \
  X(0, "v_pk_add_f32 (no op_sel)", "v_pk_add_f32 %0, %0, %1")                                                 \
  X(1, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]  (src1 halves swapped)", "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]") \
  X(2, "v_pk_add_f32 op_sel_hi:[1,0]  (src1 low half broadcast)", "v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]") \
  X(3, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]  (src1 high half broadcast)", "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]") \
  X(4, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]  (src0 halves swapped)", "v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]") \
  X(5, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")   \
  X(6, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_fma_f32 %0, %0, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]") \
  X(7, "v_pk_fma_f32 op_sel_hi:[1,0,1]  (src1 low half broadcast)", "v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]") \
  X(8, "v_pk_mov_b32 op_sel:[1,0]  (halves swapped)", "v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]\n\tv_pk_add_f32 %0, %0, %1")

#define H_FORMS(X)                                                                                            \
  X(20, "v_pk_add_f16 (no op_sel)", "v_pk_add_f16 %0, %0, %1")                                                \
  X(21, "v_pk_add_f16 op_sel:[0,1] op_sel_hi:[1,0]  (src1 halves swapped)", "v_pk_add_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]") \
  X(22, "v_pk_fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_fma_f16 %0, %0, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]") \
  X(23, "v_pk_mul_f16 op_sel_hi:[1,0]  (src1 low half broadcast)", "v_pk_mul_f16 %0, %0, %1 op_sel_hi:[1,0]")

template <int FORM>
__global__ void __launch_bounds__(256) victim(const float* __restrict__ x, float* __restrict__ y, int n, int rounds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i * 8 >= n) return;
  if constexpr (FORM < 20) {
    f2 v[4], acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = *reinterpret_cast<const f2*>(x + i * 8 + 2 * e) * 0.01f + 1.0f;   // ~1: products and sums stay finite for `rounds` steps
      acc[e] = v[e];
    }
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#define X(id, name, text) if constexpr (FORM == id) asm volatile(text : "+v"(acc[e]) : "v"(v[e]));
        FORMS(X)
#undef X
        if constexpr (FORM == 5 || FORM == 6 || FORM == 7) acc[e] = acc[e] * 0.5f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<f2*>(y + i * 8 + 2 * e) = acc[e];
  } else {
    unsigned v[8], acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = x[i * 8 + e] * 0.01f + 1.0f, b = x[i * 8 + (e ^ 1)] * 0.01f + 0.75f;
      const _Float16 ha = (_Float16)a, hb = (_Float16)b;
      unsigned short ua, ub;
      memcpy(&ua, &ha, 2);
      memcpy(&ub, &hb, 2);
      v[e] = (unsigned)ua | ((unsigned)ub << 16);
      acc[e] = v[e];
    }
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#define X(id, name, text) if constexpr (FORM == id) asm volatile(text : "+v"(acc[e]) : "v"(v[e]));
        H_FORMS(X)
#undef X
        if constexpr (FORM == 22 || FORM == 23) asm volatile("v_pk_mul_f16 %0, %0, 0.5 op_sel_hi:[1,0]" : "+v"(acc[e]));
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) reinterpret_cast<unsigned*>(y)[i * 8 + e] = acc[e];
  }
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

// a neighbour without matrix instructions: the same loop on plain fp32 FMAs (is it the MFMA, or any busy neighbour?)
__global__ void __launch_bounds__(256) valu_neighbour(float* sink, int iters) {
  float c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3;
  for (int i = 0; i < iters * 8; ++i) {
    c0 = fmaf(c0, 0.999f, 0.5f);
    c1 = fmaf(c1, 0.998f, 0.25f);
    c2 = fmaf(c2, 0.997f, 0.125f);
    c3 = fmaf(c3, 0.996f, 0.0625f);
  }
  if (c0 + c1 + c2 + c3 == 12345.678f) sink[threadIdx.x] = c0;
}

template <int FORM>
static void test(const char* name, const float* x, float* y, float* sink, int n, hipStream_t s1, hipStream_t s2) {
  const int rounds = 64;
  std::vector<float> ref(n), got(n);
  hipLaunchKernelGGL(victim<FORM>, dim3(n / 8 / 256), dim3(256), 0, s1, x, y, n, rounds);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), y, n * 4, hipMemcpyDeviceToHost);
  for (int nb = 0; nb < 2; ++nb) {
    int bad_runs = 0;
    long bad_vals = 0, bad_lanes_48 = 0, bad_low = 0;
    for (int it = 0; it < 10; ++it) {
      hipMemsetAsync(y, 0, n * 4, s1);
      hipDeviceSynchronize();
      for (int k = 0; k < 4; ++k) {
        if (nb == 0) hipLaunchKernelGGL(mfma_neighbour, dim3(2048), dim3(256), 0, s2, sink, 20000);
        else hipLaunchKernelGGL(valu_neighbour, dim3(2048), dim3(256), 0, s2, sink, 20000);
      }
      hipLaunchKernelGGL(victim<FORM>, dim3(n / 8 / 256), dim3(256), 0, s1, x, y, n, rounds);
      hipDeviceSynchronize();
      hipMemcpy(got.data(), y, n * 4, hipMemcpyDeviceToHost);
      long d = 0;
      for (int i = 0; i < n; ++i)
        if (memcmp(&got[i], &ref[i], 4) != 0) {
          ++d;
          const int lane = (i / 8) & 63;
          bad_lanes_48 += lane >= 48;
          bad_low += (i & 1) == 0;
        }
      bad_runs += d != 0;
      bad_vals += d;
    }
    printf("| `%s` | %s | %d of 10 | %ld | %s | %s |\n", name, nb == 0 ? "MFMA" : "fp32 FMA", bad_runs, bad_vals,
           bad_vals ? (bad_lanes_48 == bad_vals ? "all in lanes 48..63" : "other lanes too") : "-",
           bad_vals ? (bad_low == bad_vals ? "low halves only" : bad_low == 0 ? "high halves only" : "both halves") : "-");
  }
}

int main() {
  const int n = 1 << 22;
  float *x, *y, *sink;
  hipMalloc(&x, n * 4);
  hipMalloc(&y, n * 4);
  hipMalloc(&sink, 4096);
  std::vector<float> hx(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  printf("| victim instruction (repeated 64 x on 4-8 register operands per lane) | neighbour on the second stream | runs that differ from the solo run | values | lanes | halves |\n|---|---|---|---|---|---|\n");
#define X(id, name, text) test<id>(name, x, y, sink, n, s1, s2);
  FORMS(X)
  H_FORMS(X)
#undef X
  return 0;
}

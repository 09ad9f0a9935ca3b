// mfma_denorm.hip -- does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL inputs, or flush them?  (r05: decides whether the
// PP_F32X2 low terms can be kept UNSCALED -- l = f16(v - h) -- and summed into the same accumulator as the high products.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_denorm tools/probes/mfma_denorm.hip && tools/probes/mfma_denorm
// Case k: A = 2^-e (all elements), B = 1 (all elements): D = 32 * 2^-e when the input is honoured, 0 when flushed.
// Also: f32 results below 2^-126 (subnormal OUTPUT) and the product of two subnormals' exactness are printed.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* av, const float* bv, float* out, int n) {
  for (int c = 0; c < n; ++c) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av[c]; b[j] = (_Float16)bv[c]; }
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) out[c] = d[0];
  }
}

int main() {
  const int n = 10;
  float ha[n], hb[n], ho[n];
  for (int c = 0; c < n; ++c) { ha[c] = ldexpf(1.f, -14 - c); hb[c] = 1.f; }   // 2^-14 (smallest normal) .. 2^-23 (subnormal)
  ha[9] = ldexpf(1.f, -24);                                                     // the smallest subnormal
  ha[8] = ldexpf(3.f, -24); hb[8] = ldexpf(5.f, -24);                           // subnormal x subnormal: 15 * 2^-48 * 32
  float *da, *db, *dout;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dout, n);
  hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  int honoured = 1;
  for (int c = 0; c < n; ++c) {
    const double want = 32.0 * (double)(_Float16)ha[c] * (double)(_Float16)hb[c];
    printf("a = %.3e  b = %.3e  D = %.6e  expected %.6e  %s\n", ha[c], hb[c], ho[c], want, ho[c] == (float)want ? "ok" : "DIFFERS");
    if (ho[c] != (float)want) honoured = 0;
  }
  printf("{\"mfma_f16_subnormal_inputs_honoured\": %s}\n", honoured ? "true" : "false");
  return 0;
}

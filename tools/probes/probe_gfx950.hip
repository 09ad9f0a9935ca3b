// probe_gfx950.hip -- hardware semantics the attention kernel relies on, checked on a real MI355X:
//   (1) ds_read_b64_tr_b16 lane/element map   (2) v_permlane32_swap   (3) v_mfma_f32_32x32x16_f16 fragment maps
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_gfx950.hip -o gpurun_out/probe_gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__global__ void probe_tr(const int* __restrict__ addr_b16, unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)(lds) + 2u * (unsigned)addr_b16[threadIdx.x];
  u2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (unsigned short)(r[0] & 0xffff);
  out[threadIdx.x * 4 + 1] = (unsigned short)(r[0] >> 16);
  out[threadIdx.x * 4 + 2] = (unsigned short)(r[1] & 0xffff);
  out[threadIdx.x * 4 + 3] = (unsigned short)(r[1] >> 16);
}

__global__ void probe_swap(unsigned* __restrict__ out) {
  const unsigned x = 100u + threadIdx.x, y = 1000u + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}

// D[32][32] = A[32][16] . B[16][32]; lane l: A[row = l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col = l&31]
__global__ void probe_mfma(const half_t* __restrict__ A, const half_t* __restrict__ B, float* __restrict__ D) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// (4) sustained matrix-pipe rate: every SIMD of the chip issues independent 32x32x16 f16 MFMAs back to back for ~1 ms;
// reports TFLOP/s against the wall clock, and s_memtime ticks per MFMA (32 cycles each when the pipe is saturated -> the
// tick rate of s_memtime and the clock the chip sustains under matrix load)
__global__ void __launch_bounds__(256) probe_rate(float* out, long long* ticks, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (half_t)(0.001f * (threadIdx.x + j)); b[j] = (half_t)(0.002f * (threadIdx.x - j)); }
  f16v c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  int bad = 0;
  // (1) canonical image: 16-lane group G reads rows 4G..4G+3 of a [*][16] b16 matrix; lane i: row i>>2, cols 4(i&3)..
  {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) { int G = l >> 4, i = l & 15; addr[l] = (4 * G + (i >> 2)) * 16 + 4 * (i & 3); }
    int* da; unsigned short* dout; CK(hipMalloc(&da, 256)); CK(hipMalloc(&dout, 512));
    CK(hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice));
    probe_tr<<<1, 64>>>(da, dout); CK(hipDeviceSynchronize());
    std::vector<unsigned short> out(256); CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
    // model: result[lane][j] = value at addr[16G + 4j + (i>>2)] + (i&3)
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int G = l >> 4, i = l & 15; int want = addr[16 * G + 4 * j + (i >> 2)] + (i & 3);
      if (out[l * 4 + j] != want) { if (bad < 10) printf("tr canonical: lane %d elem %d got %d want %d\n", l, j, out[l*4+j], want); ++bad; }
    }
    printf("tr canonical lane0: %d %d %d %d ; lane1: %d %d %d %d ; lane5: %d %d %d %d ; lane 17: %d %d %d %d\n", out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[20], out[21], out[22], out[23], out[68], out[69], out[70], out[71]);
    // scattered addresses (arbitrary 8-byte aligned, per lane): model must still hold
    for (int l = 0; l < 64; ++l) addr[l] = ((l * 37 + 11) % 1000) * 4;
    CK(hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice));
    probe_tr<<<1, 64>>>(da, dout); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int G = l >> 4, i = l & 15; int want = addr[16 * G + 4 * j + (i >> 2)] + (i & 3);
      if (out[l * 4 + j] != want) { if (bad2 < 10) printf("tr scattered: lane %d elem %d got %d want %d\n", l, j, out[l*4+j], want); ++bad2; }
    }
    printf("ds_read_b64_tr_b16 model: canonical %s, scattered %s\n", bad ? "MISMATCH" : "ok", bad2 ? "MISMATCH" : "ok");
    bad += bad2;
  }
  // (2) permlane32_swap(x, y): r0 = {x.lo, y.lo}, r1 = {x.hi, y.hi} ?
  {
    unsigned* d; CK(hipMalloc(&d, 512)); probe_swap<<<1, 64>>>(d); CK(hipDeviceSynchronize());
    std::vector<unsigned> o(128); CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    printf("permlane32_swap(x=100+l, y=1000+l): lane0 r0=%u r1=%u ; lane5 r0=%u r1=%u ; lane32 r0=%u r1=%u ; lane40 r0=%u r1=%u\n", o[0], o[1], o[10], o[11], o[64], o[65], o[80], o[81]);
    int b = 0;
    for (int l = 0; l < 64; ++l) {
      unsigned w0 = l < 32 ? 100u + l : 1000u + (l - 32);   // r0: lo half keeps x, hi half receives y.lo
      unsigned w1 = l < 32 ? 100u + l + 32 : 1000u + l;     // r1: lo half receives x.hi, hi half keeps y
      if (o[2 * l] != w0 || o[2 * l + 1] != w1) ++b;
    }
    printf("permlane32_swap model: %s\n", b ? "MISMATCH" : "ok"); bad += b;
  }
  // (3) mfma 32x32x16 f16 with asymmetric integer matrices
  {
    std::vector<half_t> A(32 * 16), B(16 * 32); std::vector<float> D(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (half_t)(float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (half_t)(float)((k * 2 + j * 7) % 5 - 2);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += (float)A[i * 16 + k] * (float)B[k * 32 + j];
    half_t *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    probe_mfma<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int b = 0; for (int i = 0; i < 1024; ++i) if (D[i] != R[i]) ++b;
    printf("mfma_f32_32x32x16_f16 fragment maps: %s (%d mismatches)\n", b ? "MISMATCH" : "ok", b); bad += b;
  }
  // (4)
  {
    const int nblk = 256 * 2, iters = 20000;   // 2 work-groups of 4 waves per CU: 2 waves per SIMD
    float* dout; long long* dt; CK(hipMalloc(&dout, nblk * 256 * 4)); CK(hipMalloc(&dt, nblk * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe_rate<<<nblk, 256>>>(dout, dt, 100); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); probe_rate<<<nblk, 256>>>(dout, dt, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> t(nblk); CK(hipMemcpy(t.data(), dt, nblk * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : t) avg += (double)v; avg /= nblk;
      const double flop = (double)nblk * 4 * iters * 4 * 32768.0;
      // per SIMD: 2 waves x 4 MFMAs x iters, 32 cycles each when saturated
      printf("mfma rate: %.3f ms, %.0f TFLOP/s; s_memtime ticks per work-group %.0f = %.1f ticks per MFMA slot (32 cycles if saturated) -> tick rate %.0f MHz, implied core clock %.0f MHz\n",
             ms, flop / ms / 1e9, avg, avg / (2.0 * 4 * iters), avg / ms / 1e3, (2.0 * 4 * iters * 32) / ms / 1e3);
    }
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad ? 1 : 0;
}

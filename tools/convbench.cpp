// convbench -- times pp_conv2d through the C ABI only (no Python, no torch): a second client of
// include/propainter_mi355.h and a fast-starting micro-benchmark for kernel work on the MI355X.
//
//   hipcc -O2 -std=c++17 -I include tools/convbench.cpp -L comfyui_propainter_nodes_amd -lpropainter_mi355 \
//         -Wl,-rpath,'$ORIGIN/../comfyui_propainter_nodes_amd' -o tools/convbench
//   tools/convbench [name ...]        (no names: every shape below)
//
// Buffers hold small pseudo-random values; weights are written directly in the packed layout
// [Cout][tap][segment][c padded to 32] (PP_F32X2: the split layout), so timing needs no host packing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "propainter_mi355.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Shape {
  const char* name;
  int dtype;  // PP_F32, PP_F16, PP_F32X2
  int N, H, W;
  std::vector<int> segC;
  int Cout, kh, kw, ph, pw;
};

static const std::vector<Shape> SHAPES = {
    {"raft_gru_1x5_f32x2", PP_F32X2, 158, 45, 80, {128, 128}, 256, 1, 5, 0, 2},
    {"raft_gru128_1x5_f32x2", PP_F32X2, 158, 45, 80, {128, 128}, 128, 1, 5, 0, 2},
    {"raft_gru128_5x1_f32x2", PP_F32X2, 158, 45, 80, {128, 128}, 128, 5, 1, 2, 0},
    {"raft_gru_1x5_f32", PP_F32, 158, 45, 80, {128, 128}, 256, 1, 5, 0, 2},
    {"raft_convc2_f32x2", PP_F32X2, 158, 45, 80, {256}, 192, 3, 3, 1, 1},
    {"raft_convc1_f32x2", PP_F32X2, 158, 45, 80, {324}, 256, 1, 1, 0, 0},     // the correlation features' 1x1 projection (flat tiles)
    {"raft_convf1_f32x2", PP_F32X2, 158, 45, 80, {98}, 128, 1, 1, 0, 0},      // the 7x7 flow stem behind pp_im2col (flat tiles)
    {"raft_fh1_f32x2", PP_F32X2, 158, 45, 80, {128}, 256, 3, 3, 1, 1},
    {"raft_fh2_f32x2", PP_F32X2, 158, 45, 80, {256}, 2, 3, 3, 1, 1},
    {"dec6_f16", PP_F16, 14, 360, 640, {64}, 3, 3, 3, 1, 1},
    {"rfc_up2_f16", PP_F16, 158, 360, 640, {32}, 2, 3, 3, 1, 1},
    {"enc_3x3_256_384_f16", PP_F16, 16, 90, 160, {256}, 384, 3, 3, 1, 1},
    {"f16_3x3_256_512", PP_F16, 16, 90, 160, {256}, 512, 3, 3, 1, 1},
    {"dcn_offset_f16", PP_F16, 16, 90, 160, {128, 128, 8}, 128, 3, 3, 1, 1},
    {"fc1_f16", PP_F16, 1, 1, 29160, {512}, 1960, 1, 1, 0, 0},
    {"qkv_f16", PP_F16, 1, 1, 27540, {512}, 1536, 1, 1, 0, 0},
    {"fc2_f16", PP_F16, 1, 1, 27540, {1960}, 512, 1, 1, 0, 0},
    {"proj_f16", PP_F16, 1, 1, 27540, {512}, 512, 1, 1, 0, 0},
    {"rfc_step_f16", PP_F16, 2, 45, 80, {128, 128}, 128, 3, 3, 1, 1},
    {"rfc_off0_f16", PP_F16, 2, 45, 80, {128, 128, 128}, 128, 3, 3, 1, 1},
    {"rfc_bb2_f16", PP_F16, 2, 45, 80, {128}, 128, 3, 3, 1, 1},
    {"rfc_dcn_f16", PP_F16, 2, 45, 80, {2304}, 128, 1, 1, 0, 0},
    {"featprop_bb2_f16", PP_F16, 14, 90, 160, {128}, 128, 3, 3, 1, 1},
    {"dec_3x3_128_128_f16", PP_F16, 11, 180, 320, {128}, 128, 3, 3, 1, 1},
};

static uint32_t rng_state = 12345u;
static float frand() {  // uniform in [-0.5, 0.5)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)(rng_state >> 8) / 16777216.f - 0.5f;
}

static uint16_t f2h(float f) {  // float -> f16 bits (round to nearest even, no denormal / inf handling needed here)
  _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}

static void* device_random(size_t n, int esz, float scale) {
  std::vector<unsigned char> host(n * esz);
  for (size_t i = 0; i < n; ++i) {
    const float v = frand() * scale;
    if (esz == 4) memcpy(&host[i * 4], &v, 4);
    else {
      const uint16_t h = f2h(v);
      memcpy(&host[i * 2], &h, 2);
    }
  }
  void* d = nullptr;
  CK(hipMalloc(&d, n * esz));
  CK(hipMemcpy(d, host.data(), n * esz, hipMemcpyHostToDevice));
  return d;
}

static void run(const Shape& s, int iters) {
  const int esz = s.dtype == PP_F16 ? 2 : 4;
  pp_conv2d_params p;
  memset(&p, 0, sizeof(p));
  p.dtype = s.dtype;
  p.out_dtype = s.dtype == PP_F16 ? PP_F16 : PP_F32;
  p.nseg = (int)s.segC.size();
  int64_t kp = 0;
  std::vector<void*> bufs;
  for (int i = 0; i < p.nseg; ++i) {
    const int c = s.segC[i];
    void* x = device_random((size_t)s.N * s.H * s.W * c, esz, 2.f);
    bufs.push_back(x);
    p.in_ptr[i] = x;
    p.in_C[i] = c;
    p.in_ldc[i] = c;
    kp += (c + 31) / 32 * 32;
  }
  kp *= (int64_t)s.kh * s.kw;
  p.N = s.N; p.H = s.H; p.W = s.W;
  p.Ho = s.H + 2 * s.ph - (s.kh - 1);
  p.Wo = s.W + 2 * s.pw - (s.kw - 1);
  p.kh = s.kh; p.kw = s.kw; p.sh = p.sw = p.dh = p.dw = 1; p.ph = s.ph; p.pw = s.pw;
  // PP_F32X2 weights are f16 pairs in an f32-sized container: random f16 bit patterns of the right magnitude
  void* w = device_random((size_t)s.Cout * kp * (s.dtype == PP_F32X2 ? 2 : 1), s.dtype == PP_F32 ? 4 : 2, 0.1f);
  bufs.push_back(w);
  p.weight = w;
  if (s.Cout <= 4 && p.nseg == 1 && !(getenv("PP_CONV_DIRECT_TABLE") && getenv("PP_CONV_DIRECT_TABLE")[0] == '0')) {
    // the optional fp32 [tap * chunk][Cout padded to 2 / 4][32] table of the <= 4-channel layers (timing only: random values)
    void* t = device_random((size_t)(kp / 32) * (s.Cout <= 2 ? 2 : 4) * 32, 4, 0.1f);
    bufs.push_back(t);
    p.weight_f32 = t;
  }
  p.Cout = s.Cout;
  p.Z = 1;
  const size_t on = (size_t)s.N * p.Ho * p.Wo * s.Cout;
  void* out = nullptr;
  CK(hipMalloc(&out, on * esz));
  bufs.push_back(out);
  p.out = out;
  p.out_ldc = s.Cout;
  p.act = PP_ACT_RELU;
  for (int i = 0; i < 3; ++i)
    if (pp_conv2d(nullptr, &p) != 0) {
      fprintf(stderr, "%s: %s\n", s.name, pp_last_error());
      exit(3);
    }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) pp_conv2d(nullptr, &p);
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  int cin = 0;
  for (int c : s.segC) cin += c;
  const double flops = 2.0 * s.N * p.Ho * p.Wo * s.Cout * cin * s.kh * s.kw;
  // PP_CONVBENCH_SUM=1: a checksum of the output bits (A/B of kernel variants that must agree bit for bit)
  unsigned long long sum = 0;
  if (getenv("PP_CONVBENCH_SUM")) {
    std::vector<unsigned char> host(on * esz);
    CK(hipMemcpy(host.data(), out, on * esz, hipMemcpyDeviceToHost));
    for (size_t i = 0; i + 8 <= host.size(); i += 8) {
      unsigned long long v;
      memcpy(&v, &host[i], 8);
      sum = sum * 1099511628211ull + v;
    }
  }
  printf("{\"name\": \"%s\", \"ms\": %.4f, \"TFLOPs\": %.1f, \"out_hash\": \"%016llx\"}\n", s.name, ms, flops / ms / 1e9, sum);
  fflush(stdout);
  for (void* b : bufs) CK(hipFree(b));
}

int main(int argc, char** argv) {
  if (pp_version() != PP_ABI_VERSION) {
    fprintf(stderr, "ABI version mismatch\n");
    return 1;
  }
  for (const Shape& s : SHAPES) {
    bool want = argc <= 1;
    for (int i = 1; i < argc; ++i) want = want || (strcmp(argv[i], s.name) == 0);
    if (want) run(s, 20);
  }
  return 0;
}

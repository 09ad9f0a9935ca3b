// corr_lookup.h -- the correlation-pyramid lookup's vocabulary (CorrBlock.__call__, model/modules/RAFT/corr.py:29-50), shared by
// pp_corr_lookup (raft_kernels.hip) and the fused lookup + projection kernel pp_corr_lookup_conv (corr_lookup_conv.hip).
#pragma once
#include "pp_device.h"

namespace pp {

// planes are row-major [H][W] or tiled [ceil(H/4)][ceil(W/8)][4][8] (128-byte tiles, see pp_corr_lookup)
__device__ __forceinline__ int64_t plane_pitch(int H, int W, int tiled) {
  return tiled ? (int64_t)((H + 3) >> 2) * ((W + 7) >> 3) * 32 : (int64_t)H * W;
}
__device__ __forceinline__ int plane_off(int y, int x, int W, int tiled) {
  return tiled ? (((y >> 2) * ((W + 7) >> 3) + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7) : y * W + x;
}

struct LookupK {
  const float* pyr[4];
  int ph[4];
  int pw[4];
  int tiled[4];
  const float* flow;
  int flow_ldc;
  float* out;
  int out_ldc;
  int h, w;
  int64_t total;  // N*h*w*324
};

// One wave per (pair, pixel).  The 9x9 bilinear samples of a level touch a 10x10 window of that pixel's correlation
// plane; the wave first copies a 12-row window per level (one more row / column each side: the reference's normalise ->
// unnormalise round trip can move a coordinate across an integer by a few ulps) into LDS, then every lane evaluates 5-6 of
// the 324 outputs from LDS with the reference's arithmetic and the wave writes 324 contiguous floats.
// r03: the window is copied as 16-byte pieces -- its columns start at the 4-aligned x below the window's first column and
// span 16 (12 rows x 4 pieces = 48 lanes, ONE load instruction per level) -- on the levels whose rows allow it (4 x 8
// tiled planes, whose zero padding doubles as the out-of-plane value; row-major planes with W % 4 == 0); the 5 x 10 level
// keeps scalar loads.  r02 issued 144 4-byte loads per level (3 instructions of scattered dwords).
constexpr int kCorrRows = 12, kCorrCols = 16;
constexpr int kCorrLvl = kCorrRows * kCorrCols;

struct CoordEntry {
  int corner;   // floor of the sample coordinate, relative to the staged window's origin
  float frac;   // its fractional part
};

}  // namespace pp

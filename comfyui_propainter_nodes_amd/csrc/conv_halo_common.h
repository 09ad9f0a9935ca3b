// conv_halo_common.h -- geometry / eligibility shared by the two halo-tile translation units (conv_halo.hip: PP_F32X2,
// conv_halo_f16.hip: f16).
#pragma once
#include "pp_options.h"
#include "conv_common.h"

#include <stdio.h>

namespace pp {

struct HaloGeom {
  int tiles_x, tiles_y;  // output tiles per image
  int hw;                // halo tile width  = 16 + (kw-1)*dw
  int hrows;             // halo tile pixels = (TH + (kh-1)*dh) * hw   (<= kHaloMaxRows)
  int nct;               // output-channel tiles
  int ntiles;            // N * tiles_y * tiles_x
#ifdef PP_HALO_TRACE
  unsigned* trace;       // tools/trace_halo.sh build of conv_halo.hip only: 12 words per wave
#endif
};

constexpr int kHaloTW = 16;
constexpr int kHaloMaxRows = 192;

// PP_CONV_HALO=0 keeps every PP_F32X2 convolution on conv_split_kernel; "force" uses the halo kernel for every
// eligible geometry regardless of the problem size (tests) -- pp_options.h.
static inline int halo_mode() { return options().halo; }

// Geometry + eligibility shared by both forms; returns false when the flat-tile kernels should run instead.
static inline bool halo_geometry(const ConvK& k, int Z, int max_rows, HaloGeom* g, int TH = 8, int min_cout = 33) {
  const int mode = halo_mode();
  if (mode == 0) return false;
  const int ntaps = k.kh * k.kw;
  if (ntaps < 2 || k.sh != 1 || k.sw != 1 || k.pad_mode != PP_PAD_ZEROS || k.Cout < min_cout) return false;
  g->hw = kHaloTW + (k.kw - 1) * k.dw;
  g->hrows = (TH + (k.kh - 1) * k.dh) * g->hw;
  if (g->hrows > max_rows) return false;
  // stride 1: the output grid is the input grid shifted by the padding
  if (k.Ho != k.H + 2 * k.ph - k.dh * (k.kh - 1) || k.Wo != k.W + 2 * k.pw - k.dw * (k.kw - 1)) return false;
  if ((int64_t)k.N * k.H * k.W >= (int64_t)1 << 31) return false;
  g->tiles_x = (k.Wo + kHaloTW - 1) / kHaloTW;
  g->tiles_y = (k.Ho + TH - 1) / TH;
  const int64_t ntiles = (int64_t)k.N * g->tiles_x * g->tiles_y;
  const int64_t blocks = ntiles * ((k.Cout + 127) / 128) * Z;   // (a lower bound of the work-groups for narrower tiles)
  if (blocks >= ((int64_t)1 << 30)) return false;
  if (mode != 2) {
    if (blocks < 224) return false;  // small problems: the 32-pixel flat tiles fill the chip better
    // partial tiles compute pixels that are never stored: stay with the flat 128-pixel tiles when that wastes > 1/5
    if ((int64_t)ntiles * TH * kHaloTW * 4 > k.M * 5) return false;
  }
  g->ntiles = (int)ntiles;
  g->nct = 0;
  if (options().trace) fprintf(stderr, "pp_conv2d: halo-tile kernel (%d-row tiles), %dx%d taps, Cout %d, %d tiles\n", TH, k.kh, k.kw, k.Cout, g->ntiles);
  return true;
}

}  // namespace pp

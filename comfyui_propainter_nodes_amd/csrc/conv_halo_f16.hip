// conv_halo_f16.hip -- dispatch of the f16 halo-tile kernels (conv_halo_f16_kernel.h; compiled per output type in
// conv_halo_f16_h.hip / conv_halo_f16_f.hip)
#include "conv_halo_common.h"

namespace pp {

int launch_halo_f16_h(void* stream, const ConvK& k, int Z);
int launch_halo_f16_f(void* stream, const ConvK& k, int Z);
int launch_halo_f16_small_h(void* stream, const ConvK& k, int Z, const HaloGeom& g);
int launch_halo_f16_small_f(void* stream, const ConvK& k, int Z, const HaloGeom& g);

// 3x3 f16 layers with at most 16 output channels (the generator's 64 -> 3 output layer at full resolution) on 16-channel MFMA
// tiles of the compile-time-tap kernel: 4 waves over the 8 tile rows, one 16 x 32 block per wave.  Twelve of the 16 output rows are
// padding, but the layer is a stream of its input through the matrix pipe either way; r04 A/B against the vector-ALU kernel of
// conv_direct.hip (tools/ab_small_cout.sh): the generator's 64 -> 3 layer at 2.5 M pixels 300 -> 186 us (4.3 -> 2.6 ms per clip),
// whole clip within the noise; PP_CONV_SMALL_HALO=0 restores the vector-ALU kernel.  Returns 1 when not eligible.
int launch_halo_f16_small_cout(void* stream, const ConvK& k, int Z, bool out_f16) {
  if (!options().small_halo || k.Cout > 16 || k.kh != 3 || k.kw != 3 || k.dh != 1 || k.dw != 1) return 1;
  // (at least two 32-channel chunks: with one, the work-group's prologue and epilogue are all there is -- flow completion's
  //  32 -> 2 output layer measured 2.07 ms here against 1.27 ms on the vector-ALU kernel)
  if (k.nchunks < 2 * 9) return 1;
  HaloGeom g;
  if (!halo_geometry(k, Z, 320, &g, 8, 1)) return 1;
  return out_f16 ? launch_halo_f16_small_h(stream, k, Z, g) : launch_halo_f16_small_f(stream, k, Z, g);
}

// f16 convolutions; returns 1 when not eligible (the caller falls back to conv_igemm_kernel)
int launch_halo_f16(void* stream, const ConvK& k, int Z, bool out_f16) {
  return out_f16 ? launch_halo_f16_h(stream, k, Z) : launch_halo_f16_f(stream, k, Z);
}

}  // namespace pp

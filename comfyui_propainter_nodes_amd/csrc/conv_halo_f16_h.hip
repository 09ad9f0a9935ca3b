// conv_halo_f16_h.hip -- the f16 halo-tile kernels with f16 output (conv_halo_f16_kernel.h), one translation unit per storage type
#include "conv_halo_f16_kernel.h"

namespace pp {
int launch_halo_f16_h(void* stream, const ConvK& k, int Z) { return launch_halo_f16_t<half_t>(stream, k, Z); }
int launch_halo_f16_small_h(void* stream, const ConvK& k, int Z, const HaloGeom& g) { return launch_halo_f16_small_t<half_t>(stream, k, Z, g); }
}  // namespace pp

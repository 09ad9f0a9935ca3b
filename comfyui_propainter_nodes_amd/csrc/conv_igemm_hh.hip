// conv_igemm_hh.hip -- one dtype family of the flat-tile implicit-GEMM kernels (conv_igemm_kernel.h), in its own translation
// unit so that the families compile in parallel.
#include "conv_igemm_kernel.h"

namespace pp {

int launch_igemm_hh(void* stream, const ConvK& k, int Z) { return launch_by_cout<IgemmFamily<half_t, half_t>>(stream, k, Z); }

}  // namespace pp

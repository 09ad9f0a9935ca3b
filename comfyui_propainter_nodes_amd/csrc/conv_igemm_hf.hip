// conv_igemm_hf.hip -- one dtype family of the flat-tile implicit-GEMM kernels (conv_igemm_kernel.h), in its own translation
// unit so that the families compile in parallel.
#include "conv_igemm_kernel.h"

namespace pp {

int launch_igemm_hf(void* stream, const ConvK& k, int Z) { return launch_by_cout<IgemmFamily<half_t, float>>(stream, k, Z); }

}  // namespace pp

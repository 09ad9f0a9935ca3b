// conv_igemm_ff.hip -- one dtype family of the flat-tile implicit-GEMM kernels (conv_igemm_kernel.h), in its own translation
// unit so that the families compile in parallel.
#include "conv_igemm_kernel.h"

namespace pp {

int launch_igemm_ff(void* stream, const ConvK& k, int Z) { return launch_by_cout<IgemmFamily<float, float>>(stream, k, Z); }

}  // namespace pp

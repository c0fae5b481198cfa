// one instantiation unit of the implicit-GEMM kernels (see igemm_impl.h)
#include "igemm_impl.h"

namespace rpe {
template int launch_nt_mode<float, MODE_DENSE>(NTArgs<float>&, hipStream_t);
template int launch_nt_mode<float, MODE_STEM>(NTArgs<float>&, hipStream_t);
}  // namespace rpe

// one instantiation unit of the implicit-GEMM kernels (see igemm_impl.h)
#include "igemm_impl.h"

namespace rpe {
template int launch_nt_mode<bf16, MODE_HALO>(NTArgs<bf16>&, hipStream_t);
}  // namespace rpe

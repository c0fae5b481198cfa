// one instantiation unit of the implicit-GEMM kernels (see igemm_impl.h)
#include "igemm_impl.h"

namespace rpe {
template int launch_tn<bf16>(TNArgs<bf16>&, int, hipStream_t, long*);
}  // namespace rpe

// one instantiation unit of the implicit-GEMM kernels (see igemm_impl.h)
#include "igemm_impl.h"

namespace rpe {
template int launch_tn<f16>(TNArgs<f16>&, int, hipStream_t, long*);
}  // namespace rpe

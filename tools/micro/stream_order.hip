// Microbenchmark: does the ORDER in which workgroups walk a tensor change a streaming kernel's HBM rate?
// out = f(y, res): reads two 411-MB tensors, writes one (the shape of bn_apply with a residual); also a 1-read 1-write form.
//   mode 0: grid-stride (block b, thread t -> chunk b*256 + t, then + grid*256): consecutive 4-KB pieces go to consecutive blocks,
//           i.e. round-robin over the 8 XCDs (what norm.hip does today)
//   mode 1: the same loop, but blockIdx remapped so that blocks on one XCD walk adjacent pieces (xcd_remap)
//   mode 2: every block owns ONE contiguous span of the tensor (n / grid chunks), blocks in launch order
//   mode 3: contiguous spans + XCD remap: each XCD streams one contiguous eighth of the tensor
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ u32x4 f(u32x4 a, u32x4 b) { u32x4 r = {a.x + b.x, a.y ^ b.y, a.z + b.z, a.w ^ b.w}; return r; }

template <int MODE, int NREAD, int UNR>
__global__ __launch_bounds__(256) void stream_kernel(u32x4* __restrict__ out, const u32x4* __restrict__ y, const u32x4* __restrict__ res, long n, long span) {
    const int nb = gridDim.x;
    const int b = (MODE == 1 || MODE == 3) ? xcd_remap(blockIdx.x, nb) : blockIdx.x;
    long i0, i1, stride;
    if (MODE <= 1) { i0 = (long)b * 256 + threadIdx.x; i1 = n; stride = (long)nb * 256; }
    else { i0 = (long)b * span + threadIdx.x; i1 = (long)(b + 1) * span; if (i1 > n) i1 = n; stride = 256; }
    for (long i = i0; i < i1; i += stride * UNR) {
        u32x4 vy[UNR], vr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * stride;
            if (j < i1) { vy[u] = __builtin_nontemporal_load(y + j); if (NREAD == 2) vr[u] = __builtin_nontemporal_load(res + j); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * stride;
            if (j < i1) out[j] = NREAD == 2 ? f(vy[u], vr[u]) : f(vy[u], vy[u]);
        }
    }
}

template <int MODE, int NREAD>
static int run(const char* name, u32x4* out, const u32x4* y, const u32x4* res, long n, int grid) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const long span = ((n + grid - 1) / grid + 255) / 256 * 256;
    hipLaunchKernelGGL((stream_kernel<MODE, NREAD, 4>), dim3(grid), dim3(256), 0, 0, out, y, res, n, span);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream_kernel<MODE, NREAD, 4>), dim3(grid), dim3(256), 0, 0, out, y, res, n, span);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("  %-34s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, ms, (NREAD + 1) * (double)n * 16 / ms / 1e9);
    return 0;
}

int main() {
    const long bytes = 802816L * 256 * 2, n = bytes / 16;
    u32x4 *out, *y, *res;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&res, bytes));
    CK(hipMemset(y, 1, bytes)); CK(hipMemset(res, 2, bytes));
    for (int grid : {2048, 4096, 8192, 25088}) {
        printf("2 reads + 1 write (3 x 411 MB):\n");
        if (run<0, 2>("grid-stride", out, y, res, n, grid)) return 1;
        if (run<1, 2>("grid-stride + xcd remap", out, y, res, n, grid)) return 1;
        if (run<2, 2>("contiguous span per block", out, y, res, n, grid)) return 1;
        if (run<3, 2>("contiguous span + xcd remap", out, y, res, n, grid)) return 1;
        printf("1 read + 1 write (2 x 411 MB):\n");
        if (run<0, 1>("grid-stride", out, y, res, n, grid)) return 1;
        if (run<1, 1>("grid-stride + xcd remap", out, y, res, n, grid)) return 1;
        if (run<2, 1>("contiguous span per block", out, y, res, n, grid)) return 1;
        if (run<3, 1>("contiguous span + xcd remap", out, y, res, n, grid)) return 1;
    }
    return 0;
}

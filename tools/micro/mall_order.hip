// Microbenchmark: does a consumer kernel that walks a tensor in the REVERSE order of its producer read (part of) it from the 256-MiB
// Infinity Cache?  A chain of streaming kernels ping-pongs between two buffers of S bytes (one read + one write per element, the shape
// of the BatchNorm apply passes); every block owns one contiguous span, spans dealt to the XCDs as contiguous eighths of the tensor
// (the row-tile order of the conv kernels).  mode F: every kernel walks its XCD's eighth upwards; mode A: directions alternate kernel
// by kernel (each reads what the previous one wrote LAST first).  Loads / stores: default policy or non-temporal.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcd_remap_dir(int bid, int nwg, int rev) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    const int cnt = q + (x < r ? 1 : 0);
    if (rev) i = cnt - 1 - i;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void stream_kernel(u32x4* __restrict__ out, const u32x4* __restrict__ y, long n, long span, int rev) {
    const int b = xcd_remap_dir(blockIdx.x, gridDim.x, rev);
    const long i0 = (long)b * span + threadIdx.x;
    long i1 = (long)(b + 1) * span; if (i1 > n) i1 = n;
    for (long i = i0; i < i1; i += 256 * UNR) {
        u32x4 vy[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const long j = i + u * 256; if (j < i1) vy[u] = NTL ? __builtin_nontemporal_load(y + j) : y[j]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * 256;
            if (j < i1) { u32x4 r = {vy[u].x + 1u, vy[u].y, vy[u].z ^ 3u, vy[u].w}; if (NTS) __builtin_nontemporal_store(r, out + j); else out[j] = r; }
        }
    }
}

template <bool NTL, bool NTS>
static int run(const char* name, u32x4* a, u32x4* b, long n, int alternate) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long span = 256L * 4 * 4;   // 16 KB per block and operand
    const int grid = (int)((n + span - 1) / span);
    const int chain = 20;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) CK(hipEventRecord(e0, 0));
        for (int k = 0; k < chain; ++k) {
            u32x4 *src = (k & 1) ? b : a, *dst = (k & 1) ? a : b;
            hipLaunchKernelGGL((stream_kernel<NTL, NTS, 4>), dim3(grid), dim3(256), 0, 0, dst, src, n, span, alternate ? (k & 1) : 0);
        }
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= chain;
    printf("  %-44s %s: %.3f ms per kernel  %.2f TB/s\n", name, alternate ? "alternating" : "all upwards ", ms, 2.0 * (double)n * 16 / ms / 1e9);
    return 0;
}

int main() {
    for (long mb : {51L, 103L, 154L, 205L, 308L, 411L, 822L}) {
        const long bytes = mb * 1000000L / 4096 * 4096, n = bytes / 16;
        u32x4 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        printf("tensor %ld MB (one read + one write per kernel)\n", mb);
        for (int alt = 0; alt < 2; ++alt) {
            if (run<false, false>("default loads, default stores", a, b, n, alt)) return 1;
            if (run<true, false>("non-temporal loads, default stores", a, b, n, alt)) return 1;
            if (run<true, true>("non-temporal loads, non-temporal stores", a, b, n, alt)) return 1;
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}

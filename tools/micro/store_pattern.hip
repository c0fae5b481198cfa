// Microbenchmark: does the shape of a GEMM epilogue's stores (partial rows vs whole rows) change the HBM write rate?
//   hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern && ./store_pattern
// Output tensor [M][N] bf16 (N = 256 -> 512-byte rows, M = 802816 -> 411 MB, beyond the Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: tile = 128 rows x 128 columns (256-B pieces of the rows); a wave store = 8 rows x 128 B (the nt_kernel epilogue today)
// mode 1: tile = 128 rows x 128 columns, a wave store = 4 rows x 256 B
// mode 2: tile = 64 rows x 256 columns (whole rows), a wave store = 2 rows x 512 B = 1 KB contiguous
// mode 3: tile = 128 rows x 256 columns (whole rows), wave store = 8 rows x 128 B (the "wide" tile with the old lane map)
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(unsigned short* __restrict__ out, const unsigned short* __restrict__ in, long M, int N, int tiles_n, int reads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD remap as in the library: consecutive logical tiles share an XCD
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const int tile_n = lb % tiles_n, tile_m = lb / tiles_n;
    u32x4 v = {(unsigned)lb, (unsigned)lane, 3u, 4u};
    if (reads) {   // an A operand of K = 64: 128 B per row, read once per tile row block
        const int BMr = (MODE == 2) ? 64 : 128;
        const u32x4* src = (const u32x4*)(in + ((long)tile_m * BMr) * 64);
        for (int k = threadIdx.x; k < BMr * 8; k += 256) { u32x4 t = src[k]; v.x ^= t.x; v.y ^= t.y; v.z ^= t.z; v.w ^= t.w; }
    }
    if (MODE == 0 || MODE == 3) {
        const int BN = MODE == 0 ? 128 : 256;
        const int NWN = BN / 64;                 // waves along N: 2 or 4 -> 8 waves for mode 3 (emulated by 2 passes)
        for (int pass = 0; pass < (MODE == 3 ? 2 : 1); ++pass) {
            const int w = wave + 4 * pass;
            const int wave_m = w / NWN, wave_n = w % NWN;
            const int erow = lane / 8, echk = lane % 8;
            for (int f = 0; f < 4; ++f)
                for (int ps = 0; ps < 2; ++ps) {
                    const long m = (long)tile_m * 128 + wave_m * 64 + f * 16 + ps * 8 + erow;
                    const int n = tile_n * BN + wave_n * 64 + echk * 8;
                    if (m < M) *(u32x4*)(out + m * N + n) = v;
                }
        }
    } else if (MODE == 1) {
        // 4 waves x 32 rows x 128 columns; wave store = 4 rows x 256 B
        const int erow = lane / 16, echk = lane % 16;
        for (int st = 0; st < 8; ++st) {
            const long m = (long)tile_m * 128 + wave * 32 + st * 4 + erow;
            const int n = tile_n * 128 + echk * 8;
            if (m < M) *(u32x4*)(out + m * N + n) = v;
        }
    } else {
        // 64 rows x 256 columns; wave w: rows 16w..16w+15, store = 2 rows x 512 B
        const int erow = lane / 32, echk = lane % 32;
        for (int st = 0; st < 8; ++st) {
            const long m = (long)tile_m * 64 + wave * 16 + st * 2 + erow;
            const int n = echk * 8;
            if (m < M) *(u32x4*)(out + m * N + n) = v;
        }
    }
}

template <int MODE> static float run(unsigned short* out, const unsigned short* in, long M, int N, int reads, int iters) {
    const int BM = MODE == 2 ? 64 : 128, BN = (MODE == 2 || MODE == 3) ? 256 : 128;
    const int tiles_n = N / BN;
    const long tiles_m = (M + BM - 1) / BM;
    const int grid = (int)(tiles_m * tiles_n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((store_kernel<MODE>), dim3(grid), dim3(256), 0, 0, out, in, M, N, tiles_n, reads);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((store_kernel<MODE>), dim3(grid), dim3(256), 0, 0, out, in, M, N, tiles_n, reads);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

__global__ void copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void fill_kernel(u32x4* __restrict__ dst, long n) {
    u32x4 v = {1u, 2u, 3u, 4u};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void read_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long n) {
    u32x4 v = {0u, 0u, 0u, 0u};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { u32x4 t = src[i]; v.x ^= t.x; v.y ^= t.y; v.z ^= t.z; v.w ^= t.w; }
    if (v.x == 0x12345678u) dst[0] = v;
}

int main() {
    const long M = 802816; const int N = 256;
    unsigned short *out, *in, *big;
    hipMalloc(&out, M * N * 2); hipMalloc(&in, M * 64 * 2); hipMalloc(&big, M * N * 2);
    hipMemset(in, 1, M * 64 * 2); hipMemset(big, 1, M * N * 2);
    const double wb = (double)M * N * 2, rb = (double)M * 64 * 2;
    for (int reads = 0; reads < 2; ++reads) {
        float t0 = run<0>(out, in, M, N, reads, 10), t1 = run<1>(out, in, M, N, reads, 10), t2 = run<2>(out, in, M, N, reads, 10), t3 = run<3>(out, in, M, N, reads, 10);
        const double by = wb + (reads ? rb : 0);
        printf("reads=%d  tile128x128/8x128B %.3f ms %.2f TB/s | tile128x128/4x256B %.3f ms %.2f TB/s | rows64x256/2x512B %.3f ms %.2f TB/s | tile128x256/8x128B %.3f ms %.2f TB/s\n",
               reads, t0, by / t0 / 1e9, t1, by / t1 / 1e9, t2, by / t2 / 1e9, t3, by / t3 / 1e9);
    }
    // reference streams over the same 411 MB
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long n16 = M * N * 2 / 16;
    for (int k = 0; k < 3; ++k) {
        float ms = 0;
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < 10; ++i) {
            if (k == 0) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (u32x4*)out, n16);
            if (k == 1) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (u32x4*)out, (const u32x4*)big, n16);
            if (k == 2) hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (u32x4*)out, (const u32x4*)big, n16);
        }
        hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("%s 411 MB: %.3f ms  %.2f TB/s\n", k == 0 ? "fill (write only)" : k == 1 ? "copy (read + write)" : "read only", ms, (k == 1 ? 2 : 1) * wb / ms / 1e9);
    }
    return 0;
}

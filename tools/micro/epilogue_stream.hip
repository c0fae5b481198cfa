// Microbenchmark: the memory behaviour of the fused conv1 data gradient ("B5": out[M][N] = f(acc, addend[M][N], y[M][N]) with a small
// GEMM operand A[M][K]) WITHOUT the matrix math: which part of the 3.4 TB/s such launches reach is the access pattern, which the
// per-workgroup load -> compute -> epilogue life cycle?   N = 256, K = 64, M = 802816 (layer1 at 256 images): 103 + 3 x 411 MB.
//   mode 0: one 128 x 128 tile per workgroup, all loads of the tile issued before any store (no phases): the pattern alone
//   mode 1: the same tile, but in the kernel's phases: A tile first (wait), then 8 epilogue steps with the operands 4 steps ahead
//   mode 2: 64-row x 256-column row panels (contiguous 32 KB per operand), operands 4 steps ahead
//   mode 3: mode 2, persistent: each workgroup walks panels p, p + grid, ...; the next panel's A rows are requested before this panel's stores
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ u32x4 mix(u32x4 a, u32x4 b, u32x4 c) { u32x4 r = {a.x + b.x ^ c.x, a.y ^ b.y + c.y, a.z + b.z ^ c.z, a.w ^ b.w + c.w}; return r; }

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned short* __restrict__ out, const unsigned short* __restrict__ add, const unsigned short* __restrict__ y,
                                         const unsigned short* __restrict__ A, long M, int npanels) {
    constexpr int N = 256, K = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u32x4 sA[128 * 8];
    extern __shared__ char pad_lds[];   // only its SIZE matters: it sets how many workgroups fit a CU
    if (M < 0) pad_lds[threadIdx.x] = 1;
    if (MODE <= 1) {
        const int lb = xcd_remap(blockIdx.x, gridDim.x);
        const int tile_n = lb & 1, tile_m = lb >> 1;
        // A tile: 128 rows x 128 B
        u32x4 acc = {0, 0, 0, 0};
        for (int i = threadIdx.x; i < 128 * 8; i += 256) { u32x4 t = *(const u32x4*)(A + ((long)tile_m * 128) * K + i * 8); sA[i] = t; }
        if (MODE == 1) { __syncthreads(); acc = sA[(threadIdx.x * 7) & 1023]; }
        const int wave_m = wave >> 1, wave_n = wave & 1, erow = lane >> 3, echk = lane & 7;
        u32x4 qa[8], qy[8];
        auto row = [&](int t) { return (long)tile_m * 128 + wave_m * 64 + (t >> 1) * 16 + (t & 1) * 8 + erow; };
        const int n = tile_n * 128 + wave_n * 64 + echk * 8;
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) { qa[t] = *(const u32x4*)(add + row(t) * N + n); qy[t] = *(const u32x4*)(y + row(t) * N + n); }
            __syncthreads(); acc = sA[(threadIdx.x * 7) & 1023];
#pragma unroll
            for (int t = 0; t < 8; ++t) *(u32x4*)(out + row(t) * N + n) = mix(qa[t], qy[t], acc);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) { qa[t] = *(const u32x4*)(add + row(t) * N + n); qy[t] = *(const u32x4*)(y + row(t) * N + n); }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                *(u32x4*)(out + row(t) * N + n) = mix(qa[t & 3], qy[t & 3], acc);
                if (t + 4 < 8) { qa[t & 3] = *(const u32x4*)(add + row(t + 4) * N + n); qy[t & 3] = *(const u32x4*)(y + row(t + 4) * N + n); }
            }
        }
    } else {
        // row panel: 64 rows x 256 columns; wave w: rows 16w .. 16w+15; a step = 2 rows x 512 B per wave
        const int erow = lane >> 5, echk = lane & 31;
        int p = MODE == 3 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
        const int pstep = MODE == 3 ? (int)gridDim.x : npanels;
        u32x4 nextA[2];
        for (int i = 0; i < 2; ++i) nextA[i] = *(const u32x4*)(A + ((long)p * 64) * K + (threadIdx.x + i * 256) * 8);
        for (; p < npanels; p += pstep) {
            for (int i = 0; i < 2; ++i) sA[threadIdx.x + i * 256] = nextA[i];
            __syncthreads();
            u32x4 acc = sA[(threadIdx.x * 7) & 511];
            __syncthreads();
            if (MODE == 3 && p + pstep < npanels)
                for (int i = 0; i < 2; ++i) nextA[i] = *(const u32x4*)(A + ((long)(p + pstep) * 64) * K + (threadIdx.x + i * 256) * 8);
            auto row = [&](int t) { return (long)p * 64 + wave * 16 + t * 2 + erow; };
            const int n = echk * 8;
            u32x4 qa[4], qy[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { qa[t] = *(const u32x4*)(add + row(t) * N + n); qy[t] = *(const u32x4*)(y + row(t) * N + n); }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                *(u32x4*)(out + row(t) * N + n) = mix(qa[t & 3], qy[t & 3], acc);
                if (t + 4 < 8) { qa[t & 3] = *(const u32x4*)(add + row(t + 4) * N + n); qy[t & 3] = *(const u32x4*)(y + row(t + 4) * N + n); }
            }
        }
    }
}

template <int MODE> static int run(const char* name, unsigned short* out, const unsigned short* add, const unsigned short* y, const unsigned short* A, long M, int grid3, int pad = 0) {
    const int npanels = (int)(M / 64);
    const int grid = MODE <= 1 ? (int)(M / 128) * 2 : (MODE == 2 ? npanels : grid3);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), pad, 0, out, add, y, A, M, npanels);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), pad, 0, out, add, y, A, M, npanels);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    const double bytes = (double)M * 64 * 2 + 3.0 * M * 256 * 2;
    printf("%-64s grid %6d lds %3d KB: %.3f ms  %.2f TB/s\n", name, grid, 16 + pad / 1024, ms, bytes / ms / 1e9);
    return 0;
}

int main() {
    const long M = 802816;
    unsigned short *out, *add, *y, *A;
    CK(hipMalloc(&out, M * 256 * 2)); CK(hipMalloc(&add, M * 256 * 2)); CK(hipMalloc(&y, M * 256 * 2)); CK(hipMalloc(&A, M * 64 * 2));
    CK(hipMemset(add, 1, M * 256 * 2)); CK(hipMemset(y, 2, M * 256 * 2)); CK(hipMemset(A, 3, M * 64 * 2));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>("128x128 tiles, all loads up front", out, add, y, A, M, 0)) return 1;
        if (run<1>("128x128 tiles, A tile -> wait -> epilogue steps 4 ahead", out, add, y, A, M, 0)) return 1;
        if (run<2>("64x256 row panels, operands 4 steps ahead", out, add, y, A, M, 0)) return 1;
        for (int pad : {0, 16 << 10, 32 << 10, 48 << 10, 64 << 10})
            if (run<1>("128x128 tiles, phases, occupancy limited by LDS", out, add, y, A, M, 0, pad)) return 1;
        if (run<3>("64x256 row panels, persistent (768 wgs)", out, add, y, A, M, 768)) return 1;
        if (run<3>("64x256 row panels, persistent (1536 wgs)", out, add, y, A, M, 1536)) return 1;
        if (run<3>("64x256 row panels, persistent (2048 wgs)", out, add, y, A, M, 2048)) return 1;
    }
    return 0;
}

// BatchNorm apply + ReLU of a bottleneck's bn2 FUSED with the Gram matrix of its output (16-bit element types, 64 or 128 channels):
//   a = relu(y * scale + shift)  written once,  S = a^T a  and  s1 = colsum(a)  accumulated on the matrix cores from the tile that is
// on its way out anyway.  S / s1 are what a y3-free block's BN3 statistics (rpe_bn_stats_from_gram) and weight-gradient combine
// (rpe_conv1x1_wgrad_combine) read; the separate rpe_gram launch re-read all of `a` (p bytes per block on the forward's critical path)
// and the apply pass was a launch of its own.  One pass: 2p bytes (read y, write a).
//
// Work split: G workgroups, each a contiguous span of rows; per macro step a workgroup moves 8 x 16 B per thread (256 rows of 64
// channels or 128 rows of 128), applies the affine map + ReLU in registers, stores the 16-bit result to memory AND into an LDS image
// [32-row step][row][chunk ^ swizzle], then multiplies every 32-row step tile with itself: the 4 waves own a quarter of S's columns
// each and read their MFMA operands with ds_read_b64_tr_b16 (the transposing read the weight-gradient kernel uses).  The next macro
// step's loads are in flight while the matrix cores run.  Every workgroup stores its fp32 partial (row-major S, then s1) to its own
// slab; gram_reduce_kernel adds the slabs in workgroup order: deterministic, no atomics.
// replaces: bn2 -> ReLU of a torchvision Bottleneck in training mode (util/model_utils.py:136; called at models/naive.py:316) plus the
// statistics pass of bn3 (DESIGN.md, y3-free blocks).
#include <stdlib.h>
#include <string.h>

#include "igemm.h"

using namespace rpe;

namespace {

// 16-byte chunk permutation inside a row of the LDS image (as tn_swz of igemm_impl.h for 16-bit types: the 8 rows a 32-lane half of a
// transposing read touches cover all 64 banks)
template <int CPR> __device__ __forceinline__ int gram_swz(int row) {
    if (CPR >= 16) return ((row & 3) | ((row >> 1) & 4)) << 1;
    return (((row >> 1) & 1) | ((row >> 2) & 2)) << 1;
}

template <typename T, int P>
__global__ __launch_bounds__(256) void bn_apply_gram_kernel(const T* __restrict__ y, T* __restrict__ out, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long M, long rows_per_wg, float* __restrict__ slab) {
    constexpr int CPR = P / 8;            // 16-byte chunks per row
    constexpr int RP = 256 / CPR;         // rows covered by one pass of the 256 threads
    constexpr int NP = 32 / RP;           // passes per 32-row step
    constexpr int MS = 8 / NP;            // 32-row steps per macro step (8 chunks per thread)
    constexpr int MROWS = MS * 32;
    constexpr int IF = P / 16;            // i fragments (all of S's rows)
    constexpr int JF = P / 64;            // j fragments per wave (a quarter of S's columns)
    static_assert(P == 64 || P == 128, "64 or 128 channels");
    __shared__ u32x4 lds[MS * 32 * CPR];  // 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = tid % CPR, prow = tid / CPR;
    const long r_begin = (long)blockIdx.x * rows_per_wg;
    long r_end = r_begin + rows_per_wg;
    if (r_end > M) r_end = M;
    float sc[8], sh[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[chunk * 8 + e]; sh[e] = shift[chunk * 8 + e]; cs[e] = 0.f; }
    f32x4 acc[IF][JF];
#pragma unroll
    for (int a = 0; a < IF; ++a)
#pragma unroll
        for (int b = 0; b < JF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 v[8];
    auto row_of = [&](long r0, int u) -> long { return r0 + (u / NP) * 32 + (u % NP) * RP + prow; };
    auto issue = [&](long r0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = row_of(r0, u);
            v[u] = r < r_end ? __builtin_nontemporal_load((const u32x4*)(y + r * P + chunk * 8)) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    const int fg = lane >> 4, fl = lane & 15, fq = fl >> 2, fp = fl & 3;
    if (r_begin < r_end) issue(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += MROWS) {
        // affine + ReLU, store, LDS image (rows past the span contribute zeros)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = row_of(r0, u);
            u32x4 o = {0u, 0u, 0u, 0u};
            if (r < r_end) {
                float f[8];
                chunk_to_f<T>(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);   // same expression as bn_apply_kernel
                o = f_to_chunk<T>(f);
                *(u32x4*)(out + r * P + chunk * 8) = o;
                chunk_to_f<T>(o, f);                                                       // the values AS STORED: what S is made of
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += f[e];
            }
            const int lr = (u % NP) * RP + prow;   // row inside the 32-row step
            lds[((u / NP) * 32 + lr) * CPR + (chunk ^ gram_swz<CPR>(lr))] = o;
        }
        if (r0 + MROWS < r_end) issue(r0 + MROWS);   // in flight while the matrix cores run
        __syncthreads();
#pragma unroll
        for (int st = 0; st < MS; ++st) {
            if (r0 + st * 32 >= r_end) break;        // (uniform) nothing but zeros from here on
            const char* tile = (const char*)(lds + st * 32 * CPR);
            auto frag = [&](int col0) -> u32x4 {
                const int col = col0 + 4 * fp;
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 8 * fg + 4 * h + fq;
                    const int ch = (col >> 3) ^ gram_swz<CPR>(row);
                    const char* ad = tile + (row * CPR + ch) * 16 + (col & 7) * 2;
                    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                    u32x2 tt = __builtin_bit_cast(u32x2, t);
                    w[2 * h] = tt.x; w[2 * h + 1] = tt.y;
                }
                return u32x4{w[0], w[1], w[2], w[3]};
            };
            u32x4 qf[JF];
#pragma unroll
            for (int b = 0; b < JF; ++b) qf[b] = frag(wave * (P / 4) + b * 16);
#pragma unroll
            for (int a = 0; a < IF; ++a) {
                const u32x4 pf = frag(a * 16);
#pragma unroll
                for (int b = 0; b < JF; ++b) Mma<T>::run(pf, qf[b], acc[a][b]);
            }
        }
        __syncthreads();   // every wave is done with the image before the next macro step overwrites it
    }
    // this workgroup's partial: S row-major [P][P], then s1 [P]
    float* mine = slab + (long)blockIdx.x * (P * P + P);
#pragma unroll
    for (int a = 0; a < IF; ++a)
#pragma unroll
        for (int b = 0; b < JF; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[(a * 16 + 4 * fg + e) * P + wave * (P / 4) + b * 16 + fl] = acc[a][b][e];
    float* red = (float*)lds;   // [RP][P]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[prow * P + chunk * 8 + e] = cs[e];
    __syncthreads();
    if (tid < P) {
        float t = 0.f;
        for (int k = 0; k < RP; ++k) t += red[k * P + tid];
        mine[P * P + tid] = t;
    }
}

// out [ones_row + 1][P]: rows [0, P) = sum of the slabs' S, row ones_row = sum of their s1 (the layout rpe_gram writes).  One block per
// 64 float4 groups; its 8 waves take every 8th slab each, the 8 partial sums are added in wave order (as tn_reduce_kernel).
__global__ __launch_bounds__(512) void gram_reduce_kernel(const float* __restrict__ slab, int G, int P, float* __restrict__ out, int ones_row) {
    __shared__ f32x4 sh[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n4 = (P * P + P) / 4;
    const int idx = blockIdx.x * 64 + lane;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (idx < n4) {
        const f32x4* src = (const f32x4*)slab + idx;
        int g = w;
        for (; g + 56 < G; g += 64) {   // 8 loads in flight per lane (the launch sits on the forward's critical path)
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(g + 8 * u) * n4];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
        }
        for (; g + 8 < G; g += 16) {
            const f32x4 a = src[(long)g * n4], b = src[(long)(g + 8) * n4];
            s0 += a; s1 += b;
        }
        if (g < G) s0 += src[(long)g * n4];
    }
    sh[w][lane] = s0 + s1;
    __syncthreads();
    if (w != 0 || idx >= n4) return;
    f32x4 sum = sh[0][lane];
#pragma unroll
    for (int i = 1; i < 8; ++i) sum += sh[i][lane];
    const int e = idx * 4;
    float* dst = e < P * P ? out + e : out + (long)ones_row * P + (e - P * P);
    *(f32x4*)dst = sum;
}

struct GramPlan { int G; long rows_per_wg; long slab_bytes; };
static GramPlan gram_plan(long rows, int C) {
    const int mrows = C == 64 ? 256 : 128;
    long G = (rows + mrows - 1) / mrows;
    if (G > 512) G = 512;                      // two workgroups per CU keep enough loads in flight; more only grows the slabs
    long rpw = (rows + G - 1) / G;
    rpw = (rpw + mrows - 1) / mrows * mrows;
    G = (rows + rpw - 1) / rpw;
    return {(int)G, rpw, G * ((long)C * C + C) * 4};
}

template <typename T>
static int apply_gram_t(const void* y, void* out, const float* scale, const float* shift, long rows, int C, float* gram_out, void* ws, hipStream_t s) {
    const GramPlan pl = gram_plan(rows, C);
    if (C == 64) hipLaunchKernelGGL((bn_apply_gram_kernel<T, 64>), dim3(pl.G), dim3(256), 0, s, (const T*)y, (T*)out, scale, shift, rows, pl.rows_per_wg, (float*)ws);
    else hipLaunchKernelGGL((bn_apply_gram_kernel<T, 128>), dim3(pl.G), dim3(256), 0, s, (const T*)y, (T*)out, scale, shift, rows, pl.rows_per_wg, (float*)ws);
    RPE_CHECK_LAUNCH();
    prof_split(s, "gram_reduce_kernel");
    const int n4 = (C * C + C) / 4;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((n4 + 63) / 64), dim3(512), 0, s, (const float*)ws, pl.G, C, gram_out, (int)rpe_gram_ones_row(C));
    RPE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" long rpe_bn_apply_gram_workspace_bytes(int dtype, long rows, int C) {
    if ((dtype != RPE_BF16 && dtype != RPE_F16) || rows <= 0 || (C != 64 && C != 128)) return -1;
    return gram_plan(rows, C).slab_bytes;
}

extern "C" int rpe_bn_apply_gram(int dtype, const void* y, void* out, const float* scale, const float* shift, long rows, int C, float* gram_out,
                                 void* workspace, long workspace_bytes, void* stream) {
    note_kernel("bn_apply_gram_kernel");
    if (!y || !out || !scale || !shift || !gram_out || rows <= 0) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_gram: null argument or no rows");
    if (C != 64 && C != 128) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_gram: 64 or 128 channels (the planes of layers 1-2)");
    if ((((uintptr_t)y) | ((uintptr_t)out) | ((uintptr_t)workspace) | ((uintptr_t)gram_out)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "bn_apply_gram: 16-byte aligned tensors");
    if (!workspace || workspace_bytes < gram_plan(rows, C).slab_bytes) return rpe_set_error(RPE_ERR_WORKSPACE, "bn_apply_gram: workspace smaller than rpe_bn_apply_gram_workspace_bytes()");
    if (dtype == RPE_BF16) return apply_gram_t<bf16>(y, out, scale, shift, rows, C, gram_out, workspace, (hipStream_t)stream);
    if (dtype == RPE_F16) return apply_gram_t<f16>(y, out, scale, shift, rows, C, gram_out, workspace, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_gram: 16-bit element types only");
}

// Row-streaming form of the y3-free bottleneck's two widest GEMMs (layers 1-2, 16-bit element types): a 1x1 conv whose reduction is
// only 64 / 128 (256 with the K-concatenated operand) deep but whose output is 4x as wide, with everything that follows it fused.
//
//   forward  (rpe_conv1x1_fwd_bn):    out = relu((x W^T) * scale + shift + identity [* res_scale + res_shift]), packed ReLU mask
//
// On the tiled kernel (nt_kernel role 5, 128 x 128 tiles) such a launch is a 2..4-step K loop in front of an epilogue that moves 96 KB
// per workgroup: every workgroup pays descriptor set-up, a cold operand ring and the LDS round trip of its accumulators for ~1 us of
// matrix work, and the launch streams at 3.7-4.5 TB/s where the BatchNorm passes reach 6.  Here the WEIGHTS are the resident operand:
// a wave owns 64 output channels and keeps their W rows as MFMA fragments in registers for its whole life (32 / 64 VGPRs), a workgroup
// (4 waves = 256 channels) walks a contiguous span of rows 16 at a time, and the activation rows arrive as 16-byte loads that ARE the
// second MFMA operand (lane = (row, 8 k)): no LDS, no barrier, nothing per tile.  The fragment -> channel map is permuted so that
// accumulator register e of fragment c in lane (row = l & 15, g = l >> 4) is channel 16 g + 4 c + e of the wave's 64: a lane ends up
// with 16 CONSECUTIVE channels of one row -- two 16-byte chunks of the identity in, two of the output out, two mask bytes.
// The next step's operands are requested before the current one is multiplied.  K steps are accumulated in ascending order by the
// same instruction as nt_kernel (operands in the same roles), so the results are bitwise those of the tiled form.
// replaces: conv3 -> bn3 -> (+ identity) -> ReLU of a torchvision Bottleneck in training mode (util/model_utils.py:136; called at
// models/naive.py:316), statistics from the Gram matrix (DESIGN.md, y3-free blocks).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm.h"

using namespace rpe;

namespace rpe {

template <bool NT> __device__ __forceinline__ u32x4 sld16(const void* p) {
    if (NT) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}

// RF: depth of the fragment ring (16 rows each)
template <typename T, int K, int RF, bool RESBN>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ res,
                                                                  T* __restrict__ out, unsigned char* __restrict__ mask,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res_scale, const float* __restrict__ res_shift,
                                                                  long M, int N, long rows_per_wg, int rev) {
    constexpr int KS = K / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m16 = lane & 15, g = lane >> 4;
    const int ncol = N >> 8;                                  // 256-channel column blocks
    const int lb = xcd_remap_dir(blockIdx.x, gridDim.x, rev);
    const int cblk = lb % ncol;                               // (the column blocks of one span are neighbours: they share the x rows in L2)
    const long span = lb / ncol;
    const int colbase = cblk * 256 + wave * 64;               // this wave's 64 channels
    const int n0 = colbase + 16 * g;                          // this lane's 16 channels
    const long r_begin = span * rows_per_wg;
    long r_end = r_begin + rows_per_wg;
    if (r_end > M) r_end = M;
    if (r_begin >= r_end) return;

    // weight fragments: row i = l & 15 of fragment c is channel colbase + 16 (i >> 2) + 4 c + (i & 3); 8 k from 8 g of K step ks
    u32x4 wf[4][KS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[c][ks] = *(const u32x4*)(w + (long)(colbase + 16 * (m16 >> 2) + 4 * c + (m16 & 3)) * K + ks * 32 + g * 8);
    float fsc[16], fsh[16], frs[RESBN ? 16 : 1];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        fsc[j] = scale[n0 + j];
        fsh[j] = shift[n0 + j];
        if (RESBN) { fsh[j] += res_shift[n0 + j]; frs[j] = res_scale[n0 + j]; }
    }
    // A ring of RF fragment buffers (registers): fragment i lives in buffer i % RF and its loads are requested as soon as fragment i - RF
    // has been consumed, so RF - 1 fragments (16 rows each: 2 KB of identity + the x rows per wave) are in flight while one is multiplied
    // -- with separate "current" and "next" steps the same registers kept only half of them in flight.
    const long nfrag = (r_end - r_begin + 15) / 16;
    auto frag_row0 = [&](long i) -> long { return r_begin + (rev ? nfrag - 1 - i : i) * 16; };
    u32x4 af[RF][KS], idv[RF][2];
    auto request = [&](long i, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        long row = frag_row0(i) + m16;
        if (row >= r_end) row = r_end - 1;                   // (tail rows re-read the span's last row; their results are not stored)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[buf][ks] = *(const u32x4*)(x + row * K + ks * 32 + g * 8);
        if (res) {
            idv[buf][0] = sld16<true>(res + row * N + n0);
            idv[buf][1] = sld16<true>(res + row * N + n0 + 8);
        }
    };
    auto work = [&](long i, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c = 0; c < 4; ++c) Mma<T>::run(wf[c][ks], af[buf][ks], acc[c]);
        const long row = frag_row0(i) + m16;
        float ad[16];
        if (res) { chunk_to_f<T>(idv[buf][0], ad); chunk_to_f<T>(idv[buf][1], ad + 8); }
        float v[16];
        unsigned bits = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * c + e;
                float t = fmaf(acc[c][e], fsc[j], fsh[j]);            // same expressions, same order as nt_kernel's role-5 epilogue / bn_apply_kernel
                if (res) t = RESBN ? fmaf(ad[j], frs[j], t) : fmaf(ad[j], 1.f, t);
                bits |= (t > 0.f ? 1u : 0u) << j;
                v[j] = fmaxf(t, 0.f);
            }
        if (row < r_end) {
            *(u32x4*)(out + row * N + n0) = f_to_chunk<T>(v);
            *(u32x4*)(out + row * N + n0 + 8) = f_to_chunk<T>(v + 8);
            if (mask) *(unsigned short*)(mask + ((row * N + n0) >> 3)) = (unsigned short)bits;
        }
    };
    auto each_buf = [&](auto fn) {   // fn(integral_constant<int, b>) for b = 0 .. RF - 1
        fn(std::integral_constant<int, 0>{});
        if constexpr (RF > 1) fn(std::integral_constant<int, 1>{});
        if constexpr (RF > 2) fn(std::integral_constant<int, 2>{});
        if constexpr (RF > 3) fn(std::integral_constant<int, 3>{});
    };
    static_assert(RF >= 1 && RF <= 4, "ring of 1..4 fragment buffers");
    each_buf([&](auto bc) { if (decltype(bc)::value < nfrag) request(decltype(bc)::value, bc); });
    for (long i = 0; i < nfrag; i += RF)
        each_buf([&](auto bc) {
            const long j = i + decltype(bc)::value;
            if (j < nfrag) {
                work(j, bc);
                if (j + RF < nfrag) request(j + RF, bc);
            }
        });
}

// whether the streaming form takes a role-5 launch (conv_api.hip asks before building the tiled launch)
bool conv1x1_stream_fwd_ok(int dtype, long M, int N, int K, const void* y_out) {
    static const bool off = getenv("RPE_NO_STREAM1X1") != nullptr;
    return !off && dtype != RPE_F32 && !y_out && (K == 64 || K == 128) && N >= 256 && (N % 256) == 0 && M >= 512;
}

template <typename T>
int conv1x1_stream_fwd(const T* x, const T* w, const T* res, T* out, unsigned char* mask, const float* scale, const float* shift, const float* res_scale,
                       const float* res_shift, long M, int N, int K, hipStream_t s) {
    if (res_scale && (!res || !res_shift)) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_stream_fwd: the residual and its shift are required with res_scale");
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)res) | ((uintptr_t)out)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "conv1x1_stream_fwd: operands must be 16-byte aligned");
    const int ncol = N / 256;
    // two workgroups per CU (<= 256 VGPRs), all resident at once: each walks ONE contiguous span of rows
    long spans = 512 / ncol;
    if (spans < 1) spans = 1;
    long rows = (M + spans - 1) / spans;
    rows = (rows + 15) / 16 * 16;
    spans = (M + rows - 1) / rows;
    const int rev = walk_take();
    const dim3 grid((unsigned)(spans * ncol));
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv1x1_stream_fwd_kernel<%s,%d>", Elem<T>::kName, K);
#define RPE_S1(KK, RF, RB) hipLaunchKernelGGL((conv1x1_stream_fwd_kernel<T, KK, RF, RB>), grid, dim3(256), 0, s, x, w, res, out, mask, scale, shift, res_scale, res_shift, M, N, rows, rev)
    if (K == 64) { if (res_scale) RPE_S1(64, 4, true); else RPE_S1(64, 4, false); }
    else { if (res_scale) RPE_S1(128, 3, true); else RPE_S1(128, 3, false); }
#undef RPE_S1
    RPE_CHECK_LAUNCH();
    return 0;
}
template int conv1x1_stream_fwd<bf16>(const bf16*, const bf16*, const bf16*, bf16*, unsigned char*, const float*, const float*, const float*, const float*, long, int, int, hipStream_t);
template int conv1x1_stream_fwd<f16>(const f16*, const f16*, const f16*, f16*, unsigned char*, const float*, const float*, const float*, const float*, long, int, int, hipStream_t);

}  // namespace rpe

// Row-streaming form of the y3-free bottleneck's conv3 forward (layers 1-2, 16-bit element types): a 1x1 conv whose reduction is only 64 /
// 128 deep but whose output is 4x as wide, with everything that follows it fused (rpe_conv1x1_fwd_bn):
//
//     out = relu((x W^T) * scale + shift + identity [* res_scale + res_shift]),  packed ReLU mask
//
// On the tiled kernel (nt_kernel role 5, 128 x 128 tiles) such a launch is a 2..4-step K loop in front of an epilogue that moves 96 KB
// per workgroup: every workgroup pays descriptor set-up, a cold operand ring and the LDS round trip of its accumulators for ~1 us of
// matrix work (213 / 130 us per launch in layers 1 / 2).  Here the WEIGHTS are the resident operand: a wave owns 64 output channels and
// keeps their W rows as MFMA fragments in registers for its whole life (32 / 64 VGPRs); a workgroup (4 waves = 256 channels) walks ONE
// contiguous span of rows 16 at a time; the x rows and the wave's 128-byte slice of the identity rows arrive by LDS-DMA in a ring that
// belongs to the WAVE (no workgroup barrier anywhere) and are read back as the second MFMA operand / the epilogue's addend.  The
// fragment -> channel map is permuted so that a lane ends up with two 16-byte chunks of ONE row (channels 8 g .. 8 g + 7 and 32 + 8 g ..):
// two chunks of identity in, two of output out, and the four lanes of a row merge their mask bytes into one 8-byte store.
// K steps are accumulated in ascending order by the same instruction with the operands in the same roles as nt_kernel, so the result is
// bitwise the tiled form's (tests/test_gpu_ops.py::test_conv1x1_forward_with_bn_from_gram, both walk directions).  190 / 120 us per launch.
// What still bounds a step are the wave's own stores: see the vmcnt note at the kernel.
// replaces: conv3 -> bn3 -> (+ identity) -> ReLU of a torchvision Bottleneck in training mode (util/model_utils.py:136; called at
// models/naive.py:316), statistics from the Gram matrix (DESIGN.md, y3-free blocks).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm.h"

using namespace rpe;

namespace rpe {

// one LDS-DMA instruction from inline assembly (see igemm_impl.h, dma16_asm: the compiler's waitcnt pass must not see these loads -- with
// stores pending beside them it can only answer "vmcnt(0)", which would drain the whole ring in front of every fragment)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void s_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff) {
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(rs) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N> __device__ __forceinline__ void s_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// R: ring slots per WAVE (one 16-row fragment each: the x rows [16][K] and the wave's 128-byte slice of 16 identity rows).  A wave fills
// and drains its own ring -- no workgroup barrier anywhere; R - 1 fragments are in flight while one is multiplied.
// vmcnt: the DMA instructions of a wave return in issue order, so "at most (R - 1) * NI outstanding" means fragment i has landed if
// nothing but the NI-instruction requests of fragments i + 1 .. i + R - 1 was issued behind it.  The epilogue's stores count on the same
// counter and may return out of order with the loads: they can only make this wait longer (it then also covers the newest stores),
// never shorter -- had fragment i's loads not returned, they and the (R - 1) NI younger loads alone would exceed the count.
template <typename T, int K, int R, bool RESBN>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ res,
                                                                  T* __restrict__ out, unsigned char* __restrict__ mask,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res_scale, const float* __restrict__ res_shift,
                                                                  long M, int N, long rows_per_wg, int rev) {
    constexpr int KS = K / 32;
    constexpr int A_BYTES = 16 * K * 2, SLOT = A_BYTES + 2048;   // x rows, then 16 x 128 B of identity
    constexpr int NA = A_BYTES / 1024, NI = NA + 2;               // DMA instructions per fragment
    constexpr int CPR = K / 8;                                    // 16-byte chunks per x row
    __shared__ u32x4 lds[4 * R * SLOT / 16];
    typedef __attribute__((address_space(3))) char lds_char;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m16 = lane & 15, g = lane >> 4;
    const int ncol = N >> 8;                                  // 256-channel column blocks
    const int lb = xcd_remap_dir(blockIdx.x, gridDim.x, rev);
    const int cblk = lb % ncol;                               // (the column blocks of one span are neighbours: they share the x rows in L2)
    const long span = lb / ncol;
    const int colbase = cblk * 256 + wave * 64;               // this wave's 64 channels
    const long r_begin = span * rows_per_wg;
    long r_end = r_begin + rows_per_wg;
    if (r_end > M) r_end = M;
    if (r_begin >= r_end) return;
    const int span_rows = (int)(r_end - r_begin);

    // Channel map: accumulator register e of fragment c in lane (row = l & 15, g = l >> 4) is channel (c >> 1) * 32 + 8 g + 4 (c & 1) + e of
    // the wave's 64 -- a lane ends up with two 16-byte chunks of one row: channels 8 g .. 8 g + 7 (fragments 0, 1) and 32 + 8 g .. (2, 3).
    // weight fragments: row i = l & 15 of fragment c is that channel for (g, e) = (i >> 2, i & 3); 8 k from 8 g of K step ks
    u32x4 wf[4][KS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[c][ks] = *(const u32x4*)(w + (long)(colbase + (c >> 1) * 32 + 8 * (m16 >> 2) + 4 * (c & 1) + (m16 & 3)) * K + ks * 32 + g * 8);
    const int nA = colbase + 8 * g, nB = nA + 32;             // first channel of the lane's two chunks
    float fsc[16], fsh[16], frs[RESBN ? 16 : 1];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = (j < 8 ? nA : nB) + (j & 7);
        fsc[j] = scale[n];
        fsh[j] = shift[n];
        if (RESBN) { fsh[j] += res_shift[n]; frs[j] = res_scale[n]; }
    }
    // descriptors over this span (32-bit offsets; rows past the span's end lie beyond num_records and read as zeros)
    // (every input is made provably uniform: the asm operand must live in SGPRs, and the span arithmetic above went through the vector ALU)
    auto uptr = [](const void* ptr) -> void* {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(uptr(x + r_begin * K), 0, __builtin_amdgcn_readfirstlane(span_rows * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(uptr(res + r_begin * N + colbase), 0,
                                                                          __builtin_amdgcn_readfirstlane((span_rows * N - colbase) * 2), 0x00020000);
    const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)((lds_char*)lds + wave * (R * SLOT)));
    // DMA geometry (lane-linear 1-KB pieces; the chunk permutation is applied on the source side):
    //   x, 128-byte rows (K = 64): piece j = rows 8 j .. 8 j + 7, lane -> (row l >> 3, slot l & 7), slot holds chunk slot ^ ((row >> 1) & 7)
    //   x, 256-byte rows (K = 128): piece j = rows 4 j .. 4 j + 3, lane -> (row l >> 4, slot l & 15), slot holds chunk slot ^ (row & 15)
    //   identity (the wave's 128-byte row slice): as x with 128-byte rows
    // so that the 16 lanes a ds_read_b128 services together (MI355X_MICROARCH.md, LDS) hit 16 different 16-byte bank groups.
    unsigned xa_off[NA], id_off[2];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = CPR == 8 ? 8 * j + (lane >> 3) : 4 * j + (lane >> 4);
        const int slot = CPR == 8 ? (lane & 7) : (lane & 15);
        const int chunk = CPR == 8 ? slot ^ ((row >> 1) & 7) : slot ^ (row & 15);
        xa_off[j] = (unsigned)((row * K + chunk * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * j + (lane >> 3), slot = lane & 7, piece = slot ^ ((row >> 1) & 7);
        id_off[j] = (unsigned)((row * N + piece * 8) * 2);
    }
    const long nfrag = (span_rows + 15) / 16;
    auto frag_local = [&](long i) -> int { return (int)((rev ? nfrag - 1 - i : i) * 16); };   // first row of fragment i, relative to the span
    auto request = [&](long i) {
        const unsigned dst = ring + (unsigned)((int)(i % R) * SLOT);
        // (fragments past the end: offsets beyond the descriptors, the slot is filled with zeros that nobody reads)
        const unsigned rowoff_x = i < nfrag ? (unsigned)(frag_local(i) * K * 2) : 0x40000000u;
        const unsigned rowoff_r = i < nfrag ? (unsigned)(frag_local(i) * N * 2) : 0x40000000u;
#pragma unroll
        for (int j = 0; j < NA; ++j) s_dma16(rs_x, dst + j * 1024, xa_off[j] + rowoff_x);
#pragma unroll
        for (int j = 0; j < 2; ++j) s_dma16(rs_r, dst + A_BYTES + j * 1024, id_off[j] + rowoff_r);
    };
    const int fx = CPR == 8 ? ((m16 >> 1) & 7) : (m16 & 15), fi = (m16 >> 1) & 7;
    auto work = [&](long i) {
        const lds_char* sl = (const lds_char*)lds + wave * (R * SLOT) + (int)(i % R) * SLOT;
        u32x4 af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *(const __attribute__((address_space(3))) u32x4*)(sl + m16 * (K * 2) + (((ks * 4 + g) ^ fx) * 16));
        const u32x4 idA = *(const __attribute__((address_space(3))) u32x4*)(sl + A_BYTES + m16 * 128 + ((g ^ fi) * 16));
        const u32x4 idB = *(const __attribute__((address_space(3))) u32x4*)(sl + A_BYTES + m16 * 128 + (((4 + g) ^ fi) * 16));
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c = 0; c < 4; ++c) Mma<T>::run(wf[c][ks], af[ks], acc[c]);
        const long row = r_begin + frag_local(i) + m16;
        float ad[16];
        chunk_to_f<T>(idA, ad);
        chunk_to_f<T>(idB, ad + 8);
        float v[16];
        unsigned bits = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * c + e;                              // (j < 8: chunk A, else chunk B)
                float t = fmaf(acc[c][e], fsc[j], fsh[j]);            // same expressions, same order as nt_kernel's role-5 epilogue / bn_apply_kernel
                t = RESBN ? fmaf(ad[j], frs[j], t) : fmaf(ad[j], 1.f, t);
                bits |= (t > 0.f ? 1u : 0u) << j;
                v[j] = fmaxf(t, 0.f);
            }
        // the wave's 64 channels of a row are 8 mask bytes: the four lanes of a row (g = 0..3, 16 lanes apart) merge theirs -- byte g of each
        // half -- and lane g = 0 stores them at once (the byte-per-lane form cost 14 us of the 226 at layer 1: 16 scattered 4-byte pieces per store)
        {
            unsigned lo = (bits & 0xffu) << (8 * (g & 1)), hi = (bits >> 8) << (8 * (g & 1));
            lo |= (unsigned)__shfl_xor((int)lo, 16); hi |= (unsigned)__shfl_xor((int)hi, 16);
            lo <<= 16 * (g >> 1); hi <<= 16 * (g >> 1);
            lo |= (unsigned)__shfl_xor((int)lo, 32); hi |= (unsigned)__shfl_xor((int)hi, 32);
            if (row < r_end && g == 0) *(u32x2*)(mask + ((row * N + colbase) >> 3)) = u32x2{lo, hi};
        }
        if (row < r_end) {
            *(u32x4*)(out + row * N + nA) = f_to_chunk<T>(v);
            *(u32x4*)(out + row * N + nB) = f_to_chunk<T>(v + 8);
        }
    };
#pragma unroll
    for (int i = 0; i < R - 1; ++i) request(i);
    for (long i = 0; i < nfrag; ++i) {
        request(i + R - 1);               // into the slot fragment i - 1 was read from (its reads have been consumed: program order)
        s_wait_vm<(R - 1) * NI>();        // fragment i has landed
        work(i);
    }
    s_wait_vm<0>();                       // no LDS-DMA may still be on its way when the workgroup's LDS is handed on
}

// whether the streaming form takes a role-5 launch (conv_api.hip asks before building the tiled launch)
bool conv1x1_stream_fwd_ok(int dtype, long M, int N, int K, const void* y_out) {
    static const bool off = getenv("RPE_NO_STREAM1X1") != nullptr;
    return !off && dtype != RPE_F32 && !y_out && (K == 64 || K == 128) && N >= 256 && (N % 256) == 0 && M >= 512;
}
// (the launcher further requires the identity and the mask: conv_api.hip falls back to the tiled form without them)

template <typename T>
int conv1x1_stream_fwd(const T* x, const T* w, const T* res, T* out, unsigned char* mask, const float* scale, const float* shift, const float* res_scale,
                       const float* res_shift, long M, int N, int K, hipStream_t s) {
    if (!res || !mask) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_stream_fwd: the identity and the mask buffer are required");
    if (res_scale && !res_shift) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_stream_fwd: res_shift is required with res_scale");
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)res) | ((uintptr_t)out)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "conv1x1_stream_fwd: operands must be 16-byte aligned");
    const int ncol = N / 256;
    // two workgroups per CU (80 / 72 KB of LDS), all resident at once: each walks ONE contiguous span of rows
    long spans = 512 / ncol;
    if (spans < 1) spans = 1;
    long rows = (M + spans - 1) / spans;
    rows = (rows + 15) / 16 * 16;
    spans = (M + rows - 1) / rows;
    const int rev = walk_take();
    const dim3 grid((unsigned)(spans * ncol));
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv1x1_stream_fwd_kernel<%s,%d>", Elem<T>::kName, K);
#define RPE_S1(KK, RR, RB) hipLaunchKernelGGL((conv1x1_stream_fwd_kernel<T, KK, RR, RB>), grid, dim3(256), 0, s, x, w, res, out, mask, scale, shift, res_scale, res_shift, M, N, rows, rev)
    // ring depth: 5 x 4 KB (K = 64) / 3 x 6 KB (K = 128) per wave = 80 / 72 KB per workgroup, two workgroups per CU.  Shallower rings with
    // three or four workgroups per CU measured level (the wave's own stores, not the loads, bound a step: the counted wait in front of
    // fragment i also covers the stores of fragment i - 1, see the kernel).
    if (K == 64) { if (res_scale) RPE_S1(64, 5, true); else RPE_S1(64, 5, false); }
    else { if (res_scale) RPE_S1(128, 3, true); else RPE_S1(128, 3, false); }
#undef RPE_S1
    RPE_CHECK_LAUNCH();
    return 0;
}
template int conv1x1_stream_fwd<bf16>(const bf16*, const bf16*, const bf16*, bf16*, unsigned char*, const float*, const float*, const float*, const float*, long, int, int, hipStream_t);
template int conv1x1_stream_fwd<f16>(const f16*, const f16*, const f16*, f16*, unsigned char*, const float*, const float*, const float*, const float*, long, int, int, hipStream_t);

}  // namespace rpe

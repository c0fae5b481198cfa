// Implicit-GEMM argument blocks shared by the conv / linear entry points.
#pragma once
#include "common.h"

namespace rpe {

enum { MODE_DENSE = 0, MODE_CONV = 1, MODE_STEM = 2, MODE_HALO = 3 };

// MODE_HALO (nt_kernel; 3x3 / stride 1 / pad 1 convs and their data gradients, 16-bit types): the activation operand is staged ONCE per
// 64-channel chunk as a halo patch -- the input rows a tile of whole output rows touches, zero padding included -- and the nine taps
// read their MFMA fragments from it at shifted addresses; only the weight tile moves per K step (half the LDS fill of the gathered
// form, no per-tap address refresh).  kHaloPatchPx: pixels (128 B each) of the patch region.
constexpr int kHaloPatchPx = 256;

// one 16 x 16 MFMA step over a 16-byte chunk per lane and operand (32 K elements of a 16-bit type, 16 of fp32): first operand =
// the N side, second = the M side; acc element e of lane l is (row l & 15 of M, column 4 * (l >> 4) + e of N)
template <typename T> struct Mma;
template <> struct Mma<bf16> {
    __device__ static __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    __device__ static __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane (row = l&15, g = l>>4) holds k = 4g..4g+3 of its row; step kk multiplies component kk
    // of both operands, i.e. the K order inside a 16-wide step is permuted identically on both sides.
    __device__ static __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
    }
};


// How row m and column k of the implicit "im2col" matrix map onto an NHWC tensor.
//   m = (b*Ho + oh)*Wo + ow ;  k = (r*S + s)*C + c
//   numerator  nh = oh*sn + base_h + tap_sign*r   (same for w)
//   source row ih = nh >> sd_shift, valid iff nh >= 0, nh divisible by 1<<sd_shift, ih < H
// forward conv : sn = stride, sd_shift = 0, base = -pad, tap_sign = +1 (tensor = x)
// data gradient: sn = 1, sd_shift = log2(stride), base = +pad, tap_sign = -1 (tensor = dy)
struct Gather {
    int H, W, C;
    int Ho, Wo;
    int R, S;
    int sn, sd_shift;
    int base_h, base_w;
    int tap_sign;
    FastDiv div_hw, div_w;
    long img_stride;
    // stride-2 data gradient only: rows are enumerated per output-pixel parity class (h&1, w&1) so that a tile's K loop
    // walks just the taps that can hit a non-zero of the zero-interleaved dy (1, 2, 2 or 4 of a 3x3; 1 or 0 of a 1x1)
    // instead of multiplying 3/4 zeros.  parity = 1: Ho/Wo above are the HALF dims, rows_q = B*(H/2)*(W/2) rows per class.
    int parity;
    int rows_q;
    // MODE_HALO: a tile = halo_rt whole output rows (global row index b*H + h) = halo_px = halo_rt * W <= 128 pixels; the patch holds
    // the padded input rows P0 .. P1 of the padded row space P(b, h) = b*(H+2) + h + 1 (two zero rows between images), W + 2 pixels each
    int halo_rt, halo_px, halo_rows;   // halo_rows = B * H
    FastDiv div_h, div_pw, div_hp;     // H, W + 2, H + 2
};

// rows per tile of the halo form for an H x W image, or 0 when the patch of no tile height fits (kHaloPatchPx pixels)
static inline int halo_rows_per_tile(int H, int W) {
    if (W + 2 > kHaloPatchPx / 3 || W > 128) return 0;
    for (int rt = 128 / W; rt >= 1; --rt) {
        const int cross = (H % rt == 0) ? 0 : (rt + H - 2) / H;    // image boundaries a tile can straddle
        if ((rt + 2 + 2 * cross) * (W + 2) <= kHaloPatchPx) return rt;
    }
    return 0;
}

template <typename T> struct NTArgs {
    const T* A;     // activations (dense [M][lda] or NHWC tensor described by g)
    // K-concatenated dense A operand (role 1 only): columns k >= K1 of the im2col matrix come from a second row-major tensor
    // A2[M][lda2] (K1 a multiple of the K step).  Used by the data gradient of a 1x1 conv with the BN backward of its output
    // folded in: dx = [dz | a_in] * [A o W ; G]^T (rpe_bn_bwd_fold_conv1x1).
    const T* A2;
    int lda2, K1;
    const T* Bw;    // weights [N][ldb], K contiguous
    T* C;           // output [M][ldc]
    int M, N, K;
    int lda, ldb, ldc;
    Gather g;
    const float* bias;   // [N] or null
    const T* addend;     // [M][ld_add] or null : C = acc (+bias) + addend
    int ld_add;
    int relu;
    float* stats_part;   // [tiles_m][2][N] per-tile column partial sums, or null:
                         //   bn_mode == 0: (sum acc, sum acc^2)  -> forward batch statistics
                         //   bn_mode != 0: (sum dz, sum dz*xhat) -> fused BatchNorm-backward reduction
    // fused BN backward on the data-gradient output: C = dz = (acc + addend) * [relu mask]
    int bn_mode;         // 0 off; 1 mask = bn_a > 0; 2 mask = bn_y*bn_scale + bn_shift > 0; 3 no mask; 4 mask = bits of bn_mask;
                         // 5 mask = bits of bn_mask and ONLY sum dz is emitted (bn_y is not read: the layer's raw output does not exist,
                         //   sum dz*xhat follows from the weight gradient's first product, rpe_bn_backward_coeffs_t)
    const unsigned char* bn_mask;   // [M][ldc/8] bytes (mode 4)
    const T* bn_y;       // raw conv output the BN normalised, [M][ldc]
    const T* bn_a;       // BN(+residual)+ReLU output, [M][ldc] (mode 1)
    const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
    int tiles_m, tiles_n;
    int rev;             // walk every XCD's share of the tiles downwards (common.h, xcd_remap_dir): filled by the launcher (walk_take)
    int role;            // 0 conv forward (training), 1 conv data gradient, 2 Linear, 3 conv forward (inference: bias/addend/ReLU),
                         // 5 1x1 conv forward (training) with BatchNorm + residual + ReLU + mask in the epilogue:
                         // selects the epilogue variant compiled into the kernel and tags its symbol in profiles
    long a_elems;        // elements of the tensor behind A (conv modes: set by the caller; dense: M * lda, filled by the launcher)
    unsigned b_bytes;    // buffer-descriptor extent of Bw (filled by the launcher, < 2 GiB)
    // role 5: training forward of a 1x1 conv whose BatchNorm statistics are known BEFORE the launch (from the Gram matrix of its
    // input: rpe_gram + rpe_bn_stats_from_gram): C = relu(acc * fwd_scale + fwd_shift + addend [* res_scale + res_shift]) with the
    // packed ReLU mask (mask_out, 16-bit element types) -- the raw conv output is written only when y_out is set
    const float *fwd_scale, *fwd_shift, *res_scale, *res_shift;
    unsigned char* mask_out;
    T* y_out;
    // bn_mode 6 / 7 (dense data-gradient role, 16-bit types): mode 5 plus the SIDE PRODUCT  Tm[n][k] = sum_m C[m][n] * t_a[m][k]  of the
    // tile on its way out (C = dz as stored) with a second row-major tensor t_a [M][64 (mode 6) | 128 (mode 7)] -- the weight gradient's
    // first product dz^T a of the PRODUCING block (rpe_conv1x1_dgrad_bn_t).  The launch is persistent: workgroup g walks row tiles
    // g / tiles_n + it * (grid / tiles_n) of column tile g % tiles_n (tiles_per_wg iterations), keeps its 128 x 64|128 partial in
    // registers and stores it once to t_slab [grid][128][64|128] fp32; the launcher's slab sum adds them in workgroup order.
    const T* t_a;
    float* t_slab;
    int tiles_per_wg;
    // Split-K form of the inference forward (few output tiles, long K: one rollout frame).  role 3 with `slab` set and
    // splits > 1 runs as role 4: grid.y = splits, workgroup (tile, z) walks K steps [z * split_steps, (z+1) * split_steps)
    // and stores its raw fp32 accumulators (fragment order) into slab[z][tile]; nt_split_epilogue_kernel then adds the
    // splits in z order and applies bias / addend / ReLU.  Planned by nt_split_plan().
    float* slab;
    long slab_bytes;
    int splits, split_steps;
};

// split plan of an inference-forward launch with 64 x 64 tiles and K steps of `bk` elements: number of splits (1 = not
// split) and K steps per split.  A split launch costs a second (epilogue) launch, so it must buy at least 4x the workgroups.
// linear: the few-row fp32 Linear layers of the heads (256 x 3655 -> 1024: 64 tiles walking 229 K steps alone; the LSTM input
// projections 256 x 3648 -> 2048: 128 tiles) aim at 512 workgroups.
static inline int nt_split_plan(long M, int N, int K, int bk, int* steps_per_split, bool linear = false) {
    const long tiles = ((M + 63) / 64) * ((N + 63) / 64);
    const int nk = (K + bk - 1) / bk;
    if (steps_per_split) *steps_per_split = nk;
    if (M > 1024 || tiles >= (linear ? 256 : 64) || nk < 32) return 1;
    long S = ((linear ? 512 : 128) + tiles - 1) / tiles;
    if (S > nk / 8) S = nk / 8;
    if (S > 32) S = 32;
    if (S < (linear ? 2 : 4)) return 1;
    const int steps = (nk + (int)S - 1) / (int)S;
    if (steps_per_split) *steps_per_split = steps;
    return (nk + steps - 1) / steps;
}
// grid of the persistent data-gradient launch with the T side product (NTArgs::t_a): two workgroups per CU (68 KB of LDS, ~200 VGPRs), a
// multiple of the column tiles, never more than the tiles there are
static inline long nt_tfuse_grid(long M, int N) {
    const long tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    long per = 512 / tiles_n;
    if (per > tiles_m) per = tiles_m;
    if (per < 1) per = 1;
    return per * tiles_n;
}
static inline int nt_tfuse_tiles_per_wg(long M, int N) {
    const long tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128, per = nt_tfuse_grid(M, N) / tiles_n;
    return (int)((tiles_m + per - 1) / per);
}
static inline long nt_split_slab_bytes(long M, int N, int splits) { return ((M + 63) / 64) * ((N + 63) / 64) * (long)splits * 64 * 64 * 4; }

template <typename T> struct TNArgs {
    const T* P;     // [M][ldp]   (dy / upstream gradient), columns i (Cout)
    const T* Q;     // im2col source (dense [M][ldq] or NHWC tensor described by g), columns j (K)
    float* D;       // [I][ldd] fp32: D[i][j] += sum_m P[m][i] * Q[m][j] with atomics, or (slab mode) D[i][j] = that sum, overwritten
    // Row-concatenated P (dense DMA path): output rows i in [I1, I1 + I2) take their P columns from a second tensor P2[M][ldp2]
    // (I1 a multiple of the I tile), and the tile starting at row ones_i0 (>= 0, a multiple of the I tile) multiplies Q by an
    // all-ones P: its rows are the column sums of Q.  One launch then yields dz^T x, x^T x and colsum(x) (folded weight gradient).
    const T* P2;
    int ldp2, I1, I2, ones_i0;
    int p_cols;     // the columns P really has (rows i >= p_cols of D stay zero up to I1 / the all-ones tile); 0 = I1 with P2, else I
    float* slab;    // optional workspace of >= slab_bytes (16-byte aligned): per-workgroup fp32 tiles, summed by tn_reduce_kernel in a
    long slab_bytes;  // fixed order (deterministic, no float atomics); null -> atomic accumulation into D
    int accumulate;   // slab mode: D += sum instead of D = sum
    int M, I, J;
    int ldp, ldq, ldd;
    Gather g;
    int tiles_i, tiles_j, splits;
    int rev;             // walk every XCD's share of the (split, tile) pairs downwards (common.h): filled by the launcher (walk_take)
    int rows_per_split;  // multiple of the m-step
    long q_elems;        // elements of the tensor behind Q (conv modes: set by the caller; dense: M * ldq, filled by the launcher)
};

// halo form of the 3x3 / stride-1 / pad-1 weight gradient (wgrad_halo.hip): all nine taps of a (co tile, ci chunk) in one workgroup, x streamed
// once through a ring of pixels in LDS.  wgrad_halo_ok: whether the deterministic (slab) weight gradient of this conv takes that form.
bool wgrad_halo_ok(const rpe_conv_desc* d, int dtype);
int wgrad_halo_set_min_w(int w);
template <typename T>
int conv_wgrad_halo(const rpe_conv_desc* d, const void* x, const void* dy, float* dw, void* slab, long slab_bytes, long* slab_query, hipStream_t s);

// row-streaming 1x1 conv + BatchNorm + identity + ReLU + mask (stream1x1.hip): the y3-free bottleneck's conv3 in layers 1-2
bool conv1x1_stream_fwd_ok(int dtype, long M, int N, int K, const void* y_out);
template <typename T>
int conv1x1_stream_fwd(const T* x, const T* w, const T* res, T* out, unsigned char* mask, const float* scale, const float* shift, const float* res_scale,
                       const float* res_shift, long M, int N, int K, hipStream_t s);

}  // namespace rpe

// C-ABI entry points for the MFMA implicit-GEMM kernels (conv forward / data-gradient /
// weight-gradient, stem conv, Linear) and the library's error channel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "igemm.h"

namespace rpe {
template <typename T> int launch_nt(NTArgs<T>& a, int mode, hipStream_t s);
template <typename T> int launch_tn(TNArgs<T>& a, int mode, hipStream_t s, long* slab_query = nullptr);
}  // namespace rpe
using namespace rpe;

static thread_local char g_err[512] = "";

extern "C" int rpe_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}
int rpe_set_error_hip(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return RPE_ERR_HIP;
}
extern "C" const char* rpe_last_error(void) { return g_err; }
extern "C" int rpe_abi_version(void) { return RPE_ABI_VERSION; }

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static int check_desc(const rpe_conv_desc* d) {
    if (!d) return rpe_set_error(RPE_ERR_SHAPE, "conv: null descriptor");
    if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->in_c <= 0 || d->out_c <= 0 || d->kh <= 0 || d->kw <= 0)
        return rpe_set_error(RPE_ERR_SHAPE, "conv: non-positive dimension");
    if (d->stride != 1 && d->stride != 2) return rpe_set_error(RPE_ERR_SHAPE, "conv: stride must be 1 or 2");
    if (d->pad < 0 || d->pad >= d->kh || d->pad >= d->kw) return rpe_set_error(RPE_ERR_SHAPE, "conv: pad must be smaller than the kernel");
    return 0;
}
static inline int out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }
// stride-2 data gradients with even input dims run in parity-class order (see igemm.h Gather::parity)
static inline bool dgrad_parity(const rpe_conv_desc* d) { return d->stride == 2 && !(d->in_h & 1) && !(d->in_w & 1); }
static inline bool is_dense(const rpe_conv_desc* d) { return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0; }
// The halo form of nt_kernel (igemm.h MODE_HALO): 3x3 / stride 1 / pad 1 with 16-bit elements and `chan` gathered channels (in_c forward,
// out_c for the data gradient) a multiple of 64.  RPE_NO_HALO=1 keeps the per-tap gathered form (A/B switch).
static inline int halo_rt(const rpe_conv_desc* d, int dtype, int chan) {
    static const bool off = getenv("RPE_NO_HALO") && atoi(getenv("RPE_NO_HALO"));
    if (off || dtype == RPE_F32 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || (chan % 64)) return 0;
    return halo_rows_per_tile(d->in_h, d->in_w);
}
template <typename T> static inline int dtype_of() { return sizeof(T) == 4 ? RPE_F32 : std::is_same<T, bf16>::value ? RPE_BF16 : RPE_F16; }
static inline void halo_gather(Gather& g, const rpe_conv_desc* d, int rt) {
    g.halo_rt = rt; g.halo_px = rt * d->in_w; g.halo_rows = d->batch * d->in_h;
    g.div_h = make_fastdiv(d->in_h); g.div_pw = make_fastdiv(d->in_w + 2); g.div_hp = make_fastdiv(d->in_h + 2);
}

template <typename T>
static int conv_fwd_t(const rpe_conv_desc* d, const void* x, const void* w, void* y, float* stats, const float* bias, const void* addend, int relu,
                      hipStream_t s, void* ws = nullptr, long ws_bytes = 0, long* ws_query = nullptr) {
    const int Ho = out_dim(d->in_h, d->kh, d->stride, d->pad), Wo = out_dim(d->in_w, d->kw, d->stride, d->pad);
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)x; a.Bw = (const T*)w; a.C = (T*)y;
    a.M = d->batch * Ho * Wo; a.N = d->out_c; a.K = d->kh * d->kw * d->in_c;
    a.lda = d->in_c; a.ldb = a.K; a.ldc = d->out_c;
    a.stats_part = stats;
    a.bias = bias; a.addend = (const T*)addend; a.ld_add = d->out_c; a.relu = relu;
    if (bias || addend || relu) { a.role = 3; a.stats_part = nullptr; }   // inference epilogue (no batch statistics)
    if (ws_query || ws) {
        // few output tiles and a long reduction (a rollout frame): split K over grid.y, partial tiles through the workspace
        const int S = nt_split_plan(a.M, a.N, a.K, 4 * Elem<T>::kChunk, nullptr);
        const long need = S > 1 ? nt_split_slab_bytes(a.M, a.N, S) : 0;
        if (ws_query) { *ws_query = need; return 0; }
        if (S > 1 && a.role == 3) {
            if (ws_bytes < need || (((uintptr_t)ws) & 15)) return rpe_set_error(RPE_ERR_WORKSPACE, "conv2d_fwd_affine_ws: workspace smaller than rpe_conv2d_fwd_affine_workspace_bytes() or unaligned");
            a.slab = (float*)ws; a.slab_bytes = ws_bytes; a.splits = S;
        }
    }
    if (is_dense(d)) return launch_nt<T>(a, MODE_DENSE, s);
    Gather& g = a.g;
    g.H = d->in_h; g.W = d->in_w; g.C = d->in_c; g.Ho = Ho; g.Wo = Wo; g.R = d->kh; g.S = d->kw;
    g.sn = d->stride; g.sd_shift = 0; g.base_h = -d->pad; g.base_w = -d->pad; g.tap_sign = 1;
    g.div_hw = make_fastdiv(Ho * Wo); g.div_w = make_fastdiv(Wo);
    g.img_stride = (long)d->in_h * d->in_w * d->in_c;
    a.a_elems = (long)d->batch * g.img_stride;
    if (const int rt = (a.slab && a.splits > 1) ? 0 : halo_rt(d, dtype_of<T>(), d->in_c)) { halo_gather(g, d, rt); return launch_nt<T>(a, MODE_HALO, s); }
    return launch_nt<T>(a, MODE_CONV, s);
}

template <typename T>
static int conv_dgrad_t(const rpe_conv_desc* d, const void* dy, const void* w_crsk, void* dx, const void* addend, const rpe_bn_bwd_epilogue* bn,
                        hipStream_t s) {
    const int Ho = out_dim(d->in_h, d->kh, d->stride, d->pad), Wo = out_dim(d->in_w, d->kw, d->stride, d->pad);
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)dy; a.Bw = (const T*)w_crsk; a.C = (T*)dx;
    a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c; a.K = d->kh * d->kw * d->out_c;
    a.lda = d->out_c; a.ldb = a.K; a.ldc = d->in_c;
    a.addend = (const T*)addend; a.ld_add = d->in_c;
    a.role = 1;
    if (bn) {
        // y == null with a_mask: the layer's raw output does not exist (y3-free bottleneck) -- mask from the bits, only sum dz is emitted
        const bool no_y = !bn->y && bn->a_mask;
        if (!bn->stats_part || (!no_y && (!bn->y || !bn->mean || !bn->invstd))) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_dgrad_bn: y, mean, invstd, stats_part are required (y may be null with a_mask: sum dz only)");
        a.bn_mode = no_y ? 5 : bn->a_mask ? 4 : bn->a_out ? 1 : (bn->scale && bn->shift ? 2 : 3);
        if (no_y && (!is_dense(d) || sizeof(T) != 2)) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_dgrad_bn: the sum-dz-only form is for 1x1 / stride-1 convs of a 16-bit element type");
        if (bn->a_mask && (d->in_c % 8)) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_dgrad_bn: a_mask needs in_c % 8 == 0");
        a.bn_y = (const T*)bn->y; a.bn_a = (const T*)bn->a_out; a.bn_mask = bn->a_mask;
        a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
        a.stats_part = bn->stats_part;
    }
    if (is_dense(d)) return launch_nt<T>(a, MODE_DENSE, s);
    Gather& g = a.g;
    g.H = Ho; g.W = Wo; g.C = d->out_c; g.Ho = d->in_h; g.Wo = d->in_w; g.R = d->kh; g.S = d->kw;
    g.sn = 1; g.sd_shift = ilog2(d->stride); g.base_h = d->pad; g.base_w = d->pad; g.tap_sign = -1;
    g.div_hw = make_fastdiv(d->in_h * d->in_w); g.div_w = make_fastdiv(d->in_w);
    g.img_stride = (long)Ho * Wo * d->out_c;
    a.a_elems = (long)d->batch * g.img_stride;
    if (dgrad_parity(d)) {
        // enumerate output pixels per parity class; Ho/Wo of the row space become the half dims
        g.parity = 1;
        g.Ho = d->in_h / 2; g.Wo = d->in_w / 2;
        g.rows_q = d->batch * g.Ho * g.Wo;
        g.div_hw = make_fastdiv(g.Ho * g.Wo); g.div_w = make_fastdiv(g.Wo);
    }
    if (const int rt = halo_rt(d, dtype_of<T>(), d->out_c)) { halo_gather(g, d, rt); return launch_nt<T>(a, MODE_HALO, s); }
    return launch_nt<T>(a, MODE_CONV, s);
}

// slab / slab_bytes: workspace of the deterministic form (null: atomic accumulation); slab_query: only report the bytes needed
template <typename T>
static int conv_wgrad_t(const rpe_conv_desc* d, const void* x, const void* dy, float* dw, void* slab, long slab_bytes, long* slab_query, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {   // 3x3 / stride 1 / pad 1 on wide maps, deterministic form: the halo kernel (wgrad_halo.hip)
        if ((slab || slab_query) && wgrad_halo_ok(d, dtype_of<T>())) return conv_wgrad_halo<T>(d, x, dy, dw, slab, slab_bytes, slab_query, s);
    }
    const int Ho = out_dim(d->in_h, d->kh, d->stride, d->pad), Wo = out_dim(d->in_w, d->kw, d->stride, d->pad);
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.P = (const T*)dy; a.Q = (const T*)x; a.D = dw;
    a.slab = (float*)slab; a.slab_bytes = slab_bytes;
    a.M = d->batch * Ho * Wo; a.I = d->out_c; a.J = d->kh * d->kw * d->in_c;
    a.ldp = d->out_c; a.ldq = d->in_c; a.ldd = a.J;
    if (is_dense(d)) return launch_tn<T>(a, MODE_DENSE, s, slab_query);
    Gather& g = a.g;
    g.H = d->in_h; g.W = d->in_w; g.C = d->in_c; g.Ho = Ho; g.Wo = Wo; g.R = d->kh; g.S = d->kw;
    g.sn = d->stride; g.sd_shift = 0; g.base_h = -d->pad; g.base_w = -d->pad; g.tap_sign = 1;
    g.div_hw = make_fastdiv(Ho * Wo); g.div_w = make_fastdiv(Wo);
    g.img_stride = (long)d->in_h * d->in_w * d->in_c;
    a.q_elems = (long)d->batch * g.img_stride;
    return launch_tn<T>(a, MODE_CONV, s, slab_query);
}

// ---------------------------------------------------------------------------------------------
// BN backward folded into the data gradient of the 1x1 conv that produced the BN's input.
//   y = a_in W^T (1x1, stride 1), z = BN(y);   dy = A o dz + B' + C' o y   per channel k of y, with
//   A = gamma r, C' = -gamma r^2 c2, B' = -A c1 - C' mean   (r = invstd, c1 = mean(dz), c2 = mean(dz xhat))
//   dx = dy W = dz (A o W) + a_in G + 1 b^T,   G = W^T diag(C') W  (Ci x Ci),   b = W^T B'
// so the data gradient reads dz and the conv's (4x smaller) INPUT instead of a materialised dy: the streaming dz, y -> dy pass
// leaves the critical path (it still feeds the weight gradient, on the side stream).
// replaces: the BatchNorm2d backward + conv3 data gradient of a torchvision Bottleneck as torch autograd runs them
// (util/model_utils.py:136 constructs the network; models/naive.py:316 is the call whose backward this is).
// ---------------------------------------------------------------------------------------------
// one wave per row.  rows [0, Ci): row n of the K-concatenated weight, wk[n][k] = A_k wd[n][k] (k < Co), and b[n];
// rows [Ci, Ci + Co): row k of the G-GEMM operand pm[k][:] = C'_k wf[k][:]
template <typename T>
__global__ __launch_bounds__(64) void bn_fold_scale_kernel(const T* __restrict__ wd, const T* __restrict__ wf, const float* __restrict__ gamma,
                                                          const float* __restrict__ invstd, const float* __restrict__ mean, const float* __restrict__ c1,
                                                          const float* __restrict__ c2, T* __restrict__ wk, T* __restrict__ pm, float* __restrict__ bias,
                                                          int Co, int Ci) {
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row < Ci) {
        float acc = 0.f;
        for (int k = lane; k < Co; k += 64) {
            const float a = gamma[k] * invstd[k];
            const float cp = -a * invstd[k] * c2[k];
            const float bp = -a * c1[k] - cp * mean[k];
            const float w = Elem<T>::to_f(wd[(long)row * Co + k]);
            wk[(long)row * (Co + Ci) + k] = Elem<T>::from_f(a * w);
            acc = fmaf(bp, w, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) bias[row] = acc;
    } else {
        const int k = row - Ci;
        const float cp = -gamma[k] * invstd[k] * invstd[k] * c2[k];
        for (int n = lane; n < Ci; n += 64) pm[(long)k * Ci + n] = Elem<T>::from_f(cp * Elem<T>::to_f(wf[(long)k * Ci + n]));
    }
}
// wk[n][Co + n'] = G[n'][n]  (g is the TN GEMM result g[i][j] = sum_k pm[k][i] wf[k][j])
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_g_kernel(const float* __restrict__ g, T* __restrict__ wk, int Co, int Ci) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Ci * Ci) return;
    const int n = (int)(idx / Ci), np = (int)(idx - (long)n * Ci);
    wk[(long)n * (Co + Ci) + Co + np] = Elem<T>::from_f(g[(long)np * Ci + n]);
}

// The whole fold preparation in ONE launch (it sits on the data-gradient chain, once per bottleneck block: the four-launch form --
// scale, G GEMM, slab sum, transpose-convert -- cost ~40 us there).  Blocks [0, (Ci/64)^2): a 64 x 64 tile of
// G[n][n'] = sum_k wd[n][k] C'_k wd[n'][k] on the matrix cores straight from global memory (the weights are L2-resident; the 4 waves
// split K, the C' scaling applied to the n' operand in registers), written as wk[n][Co + n'].  The remaining blocks: four rows n
// of wk[n][k] = A_k wd[n][k] and bias[n] = sum_k B'_k wd[n][k] each.  wd = [Ci][Co], the data-gradient copy of the weight.
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_fused_kernel(const T* __restrict__ wd, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                           const float* __restrict__ mean, const float* __restrict__ c1, const float* __restrict__ c2,
                                                           T* __restrict__ wk, float* __restrict__ bias, int Co, int Ci) {
    constexpr int CE = Elem<T>::kChunk, KS = 4 * CE;
    extern __shared__ float cp_s[];   // C'_k = -gamma_k invstd_k^2 c2_k
    const int gt = Ci / 64, nG = gt * gt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ldk = (long)Co + Ci;
    if ((int)blockIdx.x < nG) {
        // one 64 x 64 tile of G per block; the four waves split K = Co (each walks a quarter: all its loads in flight at once) and
        // their accumulators meet through LDS in wave order
        for (int k = threadIdx.x; k < Co; k += 256) cp_s[k] = -gamma[k] * invstd[k] * invstd[k] * c2[k];
        __syncthreads();
        f32x4* red = (f32x4*)(cp_s + Co);                        // [16 fragments][64 lanes]
        const int bi = blockIdx.x / gt, bj = blockIdx.x - bi * gt;
        const int i0 = bi * 64, j0 = bj * 64;                    // rows n (plain operand) / columns n' (scaled operand)
        const int fr = lane & 15, fc = lane >> 4;
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kq = Co / 4, kbeg = wave * kq;
#pragma unroll 4
        for (int k0 = kbeg; k0 < kbeg + kq; k0 += KS) {
            const int k = k0 + fc * CE;
            u32x4 am[4], bn[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) am[mi] = *(const u32x4*)(wd + (long)(i0 + mi * 16 + fr) * Co + k);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) bn[nj] = *(const u32x4*)(wd + (long)(j0 + nj * 16 + fr) * Co + k);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                float v[CE];
                chunk_to_f<T>(bn[nj], v);
#pragma unroll
                for (int e = 0; e < CE; ++e) v[e] *= cp_s[k + e];
                bn[nj] = f_to_chunk<T>(v);
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mma<T>::run(bn[nj], am[mi], acc[nj][mi]);
        }
        for (int r = 1; r < 4; ++r) {                            // waves 1, 2, 3 hand their tiles to wave 0, in that order
            if (wave == r) {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) red[(nj * 4 + mi) * 64 + lane] = acc[nj][mi];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[nj][mi] += red[(nj * 4 + mi) * 64 + lane];
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    T* dst = wk + (long)(i0 + mi * 16 + fr) * ldk + Co + j0 + nj * 16 + fc * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[e] = Elem<T>::from_f(acc[nj][mi][e]);
                }
        }
        return;
    }
    // rows: the per-channel coefficients once per block (A_k, B'_k in LDS), then one wave per row n, 16-byte chunks per lane
    float* a_s = cp_s;
    float* bp_s = cp_s + Co;
    for (int k = threadIdx.x; k < Co; k += 256) {
        const float a = gamma[k] * invstd[k];
        const float cp = -a * invstd[k] * c2[k];
        a_s[k] = a;
        bp_s[k] = -a * c1[k] - cp * mean[k];
    }
    __syncthreads();
    const int row = ((int)blockIdx.x - nG) * 4 + wave;
    if (row >= Ci) return;
    float acc = 0.f;
    for (int k = lane * CE; k < Co; k += 64 * CE) {
        float w[CE];
        chunk_to_f<T>(*(const u32x4*)(wd + (long)row * Co + k), w);
        float o[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { o[e] = a_s[k + e] * w[e]; acc = fmaf(bp_s[k + e], w[e], acc); }
        T* dst = wk + (long)row * ldk + k;
        if ((((long)row * ldk + k) * (long)sizeof(T)) % 16 == 0) *(u32x4*)dst = f_to_chunk<T>(o);
        else {
#pragma unroll
            for (int e = 0; e < CE; ++e) dst[e] = Elem<T>::from_f(o[e]);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) bias[row] = acc;
}

template <typename T>
static int bn_fold_t(int Co, int Ci, const void* wf, const void* wd, const float* gamma, const float* invstd, const float* mean, const float* c1c2,
                     void* w_kcat, float* bias, void* scratch, long scratch_bytes, hipStream_t s) {
    if ((Ci % 64) == 0 && (Co % (16 * Elem<T>::kChunk)) == 0 && Co <= 8192) {
        const int gt = Ci / 64;
        hipLaunchKernelGGL((bn_fold_fused_kernel<T>), dim3(gt * gt + (Ci + 3) / 4), dim3(256), (size_t)Co * 8 + 16 * 64 * 16, s, (const T*)wd, gamma, invstd, mean, c1c2,
                           c1c2 + Co, (T*)w_kcat, bias, Co, Ci);
        RPE_CHECK_LAUNCH();
        note_kernel("bn_fold_fused_kernel");
        return 0;
    }
    // scratch: pm [Co][Ci] T | g [Ci][Ci] fp32 | slab of the G GEMM
    const long pm_bytes = ((long)Co * Ci * (long)sizeof(T) + 255) / 256 * 256, g_bytes = ((long)Ci * Ci * 4 + 255) / 256 * 256;
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.M = Co; a.I = Ci; a.J = Ci; a.ldp = Ci; a.ldq = Ci; a.ldd = Ci;
    long slab = 0;
    if (int e = launch_tn<T>(a, MODE_DENSE, s, &slab)) return e;
    if (!scratch) return rpe_set_error(RPE_ERR_WORKSPACE, "bn_bwd_fold_conv1x1: null scratch");
    if (scratch_bytes < pm_bytes + g_bytes + slab) return rpe_set_error(RPE_ERR_WORKSPACE, "bn_bwd_fold_conv1x1: scratch smaller than rpe_bn_bwd_fold_scratch_bytes()");
    T* pm = (T*)scratch;
    float* g = (float*)((char*)scratch + pm_bytes);
    hipLaunchKernelGGL((bn_fold_scale_kernel<T>), dim3(Ci + Co), dim3(64), 0, s, (const T*)wd, (const T*)wf, gamma, invstd, mean, c1c2, c1c2 + Co, (T*)w_kcat, pm,
                       bias, Co, Ci);
    RPE_CHECK_LAUNCH();
    a.P = pm; a.Q = (const T*)wf; a.D = g;
    a.slab = (float*)((char*)scratch + pm_bytes + g_bytes); a.slab_bytes = scratch_bytes - pm_bytes - g_bytes;
    if (int e = launch_tn<T>(a, MODE_DENSE, s)) return e;
    hipLaunchKernelGGL((bn_fold_g_kernel<T>), dim3((unsigned)(((long)Ci * Ci + 255) / 256)), dim3(256), 0, s, g, (T*)w_kcat, Co, Ci);
    RPE_CHECK_LAUNCH();
    note_kernel("bn_fold_scale_kernel + tn_kernel(G) + bn_fold_g_kernel");
    return 0;
}
template <typename T> static int bn_fold_scratch_t(int Co, int Ci, long* bytes) {
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.M = Co; a.I = Ci; a.J = Ci; a.ldp = Ci; a.ldq = Ci; a.ldd = Ci;
    long slab = 0;
    if (int e = launch_tn<T>(a, MODE_DENSE, nullptr, &slab)) return e;
    *bytes = ((long)Co * Ci * (long)sizeof(T) + 255) / 256 * 256 + ((long)Ci * Ci * 4 + 255) / 256 * 256 + slab;
    return 0;
}

// ---- the weight-gradient side of the same fold ------------------------------------------------
//   dW[c][n] = sum_m dy[m][c] a_in[m][n] = A_c (dz^T a_in)[c][n] + B'_c colsum(a_in)[n] + C'_c (W (a_in^T a_in))[c][n]
// (y = a_in W^T makes sum_m y[m][c] a_in[m][n] = (W S)[c][n] with S = a_in^T a_in, Ci x Ci): the weight gradient reads dz and
// a_in only -- the streaming dz, y -> dy pass and the dy tensor disappear.
// dw[c][n] = A_c (D1[c][n] - c1_c s1[n]) + C'_c ((W S)[c][n] - mean_c s1[n]);  dp = [D1 (Co rows) ; S (Ci rows) ; ... ; s1 (row `ones_row`)]
// One block = 16 rows c x 64 columns n: W rows and the S column chunk go through LDS, W S is formed in fp32 on the vector units
// (Co Ci Ci MACs in all: 1 M .. 67 M for layers 1-3).
// (d1 = dz^T a_in [Co][Ci], S = a_in^T a_in [Ci][Ci], s1 = colsum(a_in) [Ci]: three pointers -- one launch's output buffer in the folded
// weight gradient, the backward's T and the forward's Gram buffer in a y3-free block)
__global__ __launch_bounds__(256) void wgrad_fold_combine_kernel(float* __restrict__ dw, const float* __restrict__ d1, const float* __restrict__ S,
                                                                const float* __restrict__ s1v, const float* __restrict__ w,
                                                                const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                                const float* __restrict__ mean, const float* __restrict__ c1, const float* __restrict__ c2,
                                                                int Co, int Ci) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // Ws [16][Ci] | Ss [KC][64], KC = min(Ci, 128) rows of S at a time
    float* Ws = sm;
    float* Ss = sm + 16 * Ci;
    const int KC = Ci < 128 ? Ci : 128;
    const int c0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 16 * Ci; i += 256) Ws[i] = w[(long)c0 * Ci + i];
    const int cl = threadIdx.x >> 4, nq = (threadIdx.x & 15) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Ci; k0 += KC) {
        __syncthreads();   // (Ws written / the previous chunk of S consumed)
        for (int i = threadIdx.x; i < KC * 16; i += 256) {   // 16 float4 per row of the 64-column chunk
            const int k = i >> 4, q4 = i & 15;
            *(f32x4*)(Ss + k * 64 + q4 * 4) = (n0 + q4 * 4 < Ci) ? *(const f32x4*)(S + (long)(k0 + k) * Ci + n0 + q4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        for (int k = 0; k < KC; ++k) {
            const float wv = Ws[cl * Ci + k0 + k];
            const f32x4 sv = *(const f32x4*)(Ss + k * 64 + nq);
            acc[0] = fmaf(wv, sv[0], acc[0]); acc[1] = fmaf(wv, sv[1], acc[1]); acc[2] = fmaf(wv, sv[2], acc[2]); acc[3] = fmaf(wv, sv[3], acc[3]);
        }
    }
    const int c = c0 + cl;
    const float a = gamma[c] * invstd[c];
    const float cp = -a * invstd[c] * c2[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + nq + j;
        if (n < Ci) {
            const float s1 = s1v[n];
            dw[(long)c * Ci + n] = a * (d1[(long)c * Ci + n] - c1[c] * s1) + cp * (acc[j] - mean[c] * s1);
        }
    }
}

struct WFoldPlan { long dp_off, slab_off, total; int ones_row, rows; };
template <typename T> static void wfold_args(TNArgs<T>& a, long M, int Co, int Ci, int ones_row) {
    memset(&a, 0, sizeof(a));
    a.M = (int)M; a.I = ones_row + 1; a.J = Ci; a.ldp = Co; a.ldq = Ci; a.ldd = Ci;
    a.ldp2 = Ci; a.I1 = Co; a.I2 = Ci; a.ones_i0 = ones_row;
}
template <typename T> static int wfold_plan(long M, int Co, int Ci, WFoldPlan& pl) {
    auto al = [](long b) { return (b + 255) / 256 * 256; };
    pl.ones_row = Co + (Ci + 127) / 128 * 128;   // [0, Co): dz^T x | [Co, Co + Ci): x^T x | row ones_row: colsum(x)   (128-row tiles)
    pl.rows = pl.ones_row + 1;
    TNArgs<T> a;
    wfold_args<T>(a, M, Co, Ci, pl.ones_row);
    a.P2 = (const T*)16;   // (placeholder so that the query plans the concatenated problem; nothing is dereferenced)
    long slab = 0;
    if (int e = launch_tn<T>(a, MODE_DENSE, nullptr, &slab)) return e;
    pl.dp_off = 0;
    pl.slab_off = al((long)pl.rows * Ci * 4);
    pl.total = pl.slab_off + slab;
    return 0;
}

template <typename T>
static int conv1x1_wgrad_folded_t(const rpe_conv_desc* d, const void* dz, const void* a_in, const float* w_master, const float* gamma, const float* invstd,
                                  const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, long* query, hipStream_t s) {
    const long M = (long)d->batch * d->in_h * d->in_w;
    const int Co = d->out_c, Ci = d->in_c;
    if ((Co % 128) || (Ci % 64) || Ci > 512 || (Ci > 128 && (Ci % 128))) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded: out_c % 128 == 0, in_c in {64, 128, 256, 384, 512}");
    WFoldPlan pl;
    if (int e = wfold_plan<T>(M, Co, Ci, pl)) return e;
    if (query) { *query = pl.total; return 0; }
    if (!scratch || scratch_bytes < pl.total) return rpe_set_error(RPE_ERR_WORKSPACE, "conv1x1_wgrad_folded: scratch smaller than rpe_conv1x1_wgrad_folded_scratch_bytes()");
    char* sc = (char*)scratch;
    float* dp = (float*)(sc + pl.dp_off);
    TNArgs<T> a;
    wfold_args<T>(a, M, Co, Ci, pl.ones_row);   // one launch: dz^T x, x^T x and colsum(x)
    a.P = (const T*)dz; a.P2 = (const T*)a_in; a.Q = (const T*)a_in; a.D = dp;
    a.slab = (float*)(sc + pl.slab_off); a.slab_bytes = scratch_bytes - pl.slab_off;
    if (int e = launch_tn<T>(a, MODE_DENSE, s)) return e;
    prof_split(s, "wgrad_fold_combine_kernel");
    const size_t lds = (size_t)(16 * Ci + (Ci < 128 ? Ci : 128) * 64) * 4;
    hipLaunchKernelGGL(wgrad_fold_combine_kernel, dim3(Co / 16, (Ci + 63) / 64), dim3(256), lds, s, dw, dp, dp + (long)Co * Ci, dp + (long)pl.ones_row * Ci, w_master,
                       gamma, invstd, mean, c1c2, c1c2 + Co, Co, Ci);
    RPE_CHECK_LAUNCH();
    return 0;
}

// ---- fused conv1 data gradient of the NEXT block + T = dz^T a of the producing (y3-free) block --------------------------------
// sum of the persistent launch's per-workgroup partials [grid][128][P] in workgroup order: column tile t owns slabs t, t + tiles_n, ...;
// out rows [128 t, 128 t + 128).  One block per 64 float4 groups, 8 waves over the slabs (as tn_reduce_kernel), fixed order.
__global__ __launch_bounds__(512) void t_slab_sum_kernel(const float* __restrict__ slab, int nslabs, long stride4, int n4, float* __restrict__ out) {
    __shared__ f32x4 sh[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane, tile = blockIdx.y;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (idx < n4) {
        const f32x4* src = (const f32x4*)slab + (long)tile * n4 + idx;
        int g = w;
        for (; g + 8 < nslabs; g += 16) {
            const f32x4 a = src[(long)g * stride4], b = src[(long)(g + 8) * stride4];
            s0 += a; s1 += b;
        }
        if (g < nslabs) s0 += src[(long)g * stride4];
    }
    sh[w][lane] = s0 + s1;
    __syncthreads();
    if (w != 0 || idx >= n4) return;
    f32x4 sum = sh[0][lane];
#pragma unroll
    for (int i = 1; i < 8; ++i) sum += sh[i][lane];
    ((f32x4*)out)[(long)tile * n4 + idx] = sum;
}

template <typename T>
static int conv1x1_dgrad_bn_t_t(const rpe_conv_desc* d, const void* dy, const void* w_crsk, void* dz, const void* addend, const rpe_bn_bwd_epilogue* bn,
                                const void* a_prev, int P, float* t_out, void* ws, long ws_bytes, hipStream_t s) {
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)dy; a.Bw = (const T*)w_crsk; a.C = (T*)dz;
    a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c; a.K = d->out_c;
    a.lda = d->out_c; a.ldb = a.K; a.ldc = d->in_c;
    a.addend = (const T*)addend; a.ld_add = d->in_c;
    a.role = 1;
    a.bn_mode = P == 64 ? 6 : 7;
    a.bn_mask = bn->a_mask; a.stats_part = bn->stats_part;
    a.t_a = (const T*)a_prev; a.t_slab = (float*)ws;
    a.tiles_per_wg = nt_tfuse_tiles_per_wg(a.M, a.N);
    const long grid = nt_tfuse_grid(a.M, a.N);
    if (ws_bytes < grid * 128L * P * 4) return rpe_set_error(RPE_ERR_WORKSPACE, "conv1x1_dgrad_bn_t: workspace smaller than rpe_conv1x1_dgrad_bn_t_workspace_bytes()");
    if (int e = launch_nt<T>(a, MODE_DENSE, s)) return e;
    prof_split(s, "t_slab_sum_kernel");
    const int tiles_n = a.N / 128, n4 = 128 * P / 4;
    hipLaunchKernelGGL(t_slab_sum_kernel, dim3((n4 + 63) / 64, tiles_n), dim3(512), 0, s, (const float*)ws, (int)(grid / tiles_n), (long)tiles_n * n4, n4, t_out);
    RPE_CHECK_LAUNCH();
    return 0;
}

// Gram matrix and column sums of x [M][C] in one TN launch (P = Q = x plus the all-ones tile): out [ones_row + 1][C] fp32 with
// x^T x in rows [0, C) and colsum(x) in row ones_row = roundup(C, 128).  ws: the launch's slab (deterministic fixed-order sum).
static inline int gram_ones_row(int C) { return (C + 127) / 128 * 128; }
template <typename T>
static int gram_t(const void* x, long M, int C, float* out, void* ws, long ws_bytes, long* query, hipStream_t s) {
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.M = (int)M; a.I = gram_ones_row(C) + 1; a.J = C; a.ldp = C; a.ldq = C; a.ldd = C;
    a.ones_i0 = gram_ones_row(C); a.p_cols = C;
    if (query) return launch_tn<T>(a, MODE_DENSE, nullptr, query);
    a.P = (const T*)x; a.Q = (const T*)x; a.D = out;
    a.slab = (float*)ws; a.slab_bytes = ws_bytes;
    long need = 0;
    {
        TNArgs<T> q = a;
        if (int e = launch_tn<T>(q, MODE_DENSE, nullptr, &need)) return e;
    }
    if (!ws || ws_bytes < need) return rpe_set_error(RPE_ERR_WORKSPACE, "gram: workspace smaller than rpe_gram_workspace_bytes()");
    return launch_tn<T>(a, MODE_DENSE, s);
}

// training forward of a 1x1 / stride-1 conv with its BatchNorm (+ residual [under its own BN]) + ReLU + packed mask fused into the
// epilogue; scale / shift come from rpe_bn_stats_from_gram
template <typename T>
static int conv1x1_fwd_bn_t(const rpe_conv_desc* d, const void* x, const void* w, void* out, void* y_out, const float* scale, const float* shift,
                            const void* residual, const float* res_scale, const float* res_shift, unsigned char* mask, hipStream_t s) {
    const long M_ = (long)d->batch * d->in_h * d->in_w;
    if (residual && mask && conv1x1_stream_fwd_ok(Elem<T>::kDtype, M_, d->out_c, d->in_c, y_out))   // layers 1-2: the row-streaming form (stream1x1.hip), bitwise the tiled one
        return conv1x1_stream_fwd<T>((const T*)x, (const T*)w, (const T*)residual, (T*)out, mask, scale, shift, res_scale, res_shift, M_, d->out_c, d->in_c, s);
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)x; a.Bw = (const T*)w; a.C = (T*)out; a.y_out = (T*)y_out;
    a.M = d->batch * d->in_h * d->in_w; a.N = d->out_c; a.K = d->in_c;
    a.lda = d->in_c; a.ldb = d->in_c; a.ldc = d->out_c;
    a.addend = (const T*)residual; a.ld_add = d->out_c;
    a.fwd_scale = scale; a.fwd_shift = shift; a.res_scale = res_scale; a.res_shift = res_shift; a.mask_out = mask;
    a.role = 5;
    return launch_nt<T>(a, MODE_DENSE, s);
}

// data gradient of a 1x1 / stride-1 conv from A = [dz (M x Co) | a_in (M x Ci)] and the folded weight w_kcat [Ci][Co + Ci]
template <typename T>
static int conv1x1_dgrad_kcat_t(const rpe_conv_desc* d, const void* dz, const void* a_in, const void* w_kcat, const float* bias, void* dx,
                                const rpe_bn_bwd_epilogue* bn, hipStream_t s) {
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)dz; a.A2 = (const T*)a_in; a.Bw = (const T*)w_kcat; a.C = (T*)dx;
    a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c; a.K = d->out_c + d->in_c; a.K1 = d->out_c;
    a.lda = d->out_c; a.lda2 = d->in_c; a.ldb = a.K; a.ldc = d->in_c;
    a.bias = bias;
    a.role = 1;
    if (bn) {
        if (!bn->y || !bn->mean || !bn->invstd || !bn->stats_part) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_kcat: y, mean, invstd, stats_part are required");
        a.bn_mode = bn->a_mask ? 4 : bn->a_out ? 1 : (bn->scale && bn->shift ? 2 : 3);
        a.bn_y = (const T*)bn->y; a.bn_a = (const T*)bn->a_out; a.bn_mask = bn->a_mask;
        a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
        a.stats_part = bn->stats_part;
    }
    return launch_nt<T>(a, MODE_DENSE, s);
}

// ---- the same fold for the conv in FRONT of a narrow BatchNorm (a bottleneck's conv1: x [M][Ci = 4p] -> y [M][Co = p]) ------------
//   dy = A o dz + B' + C' o y  (per channel k of y)   =>   dx = dy W = [dz | y] [A o W ; C' o W] + 1 b^T,  b = W^T B'
// y itself is the narrow tensor here, so it is the second K-concatenated operand directly (K: p -> 2p, no Gram matrix): the streaming
// dz, y -> dy pass (3p per block on the data-gradient chain) disappears.  wd = [Ci][Co], the data-gradient copy of the weight.
// one wave per row n of wd: wk[n][k] = A_k wd[n][k], wk[n][Co + k] = C'_k wd[n][k], bias[n] = sum_k B'_k wd[n][k]
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_y_kernel(const T* __restrict__ wd, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                       const float* __restrict__ mean, const float* __restrict__ c1, const float* __restrict__ c2,
                                                       T* __restrict__ wk, float* __restrict__ bias, int Co, int Ci) {
    constexpr int CE = Elem<T>::kChunk;
    extern __shared__ float coef[];   // A | C' | B'  (Co each)
    for (int k = threadIdx.x; k < Co; k += 256) {
        const float a = gamma[k] * invstd[k];
        const float cp = -a * invstd[k] * c2[k];
        coef[k] = a; coef[Co + k] = cp; coef[2 * Co + k] = -a * c1[k] - cp * mean[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Ci) return;
    float acc = 0.f;
    for (int k = lane * CE; k < Co; k += 64 * CE) {
        float w[CE], o1[CE], o2[CE];
        chunk_to_f<T>(*(const u32x4*)(wd + (long)row * Co + k), w);
#pragma unroll
        for (int e = 0; e < CE; ++e) { o1[e] = coef[k + e] * w[e]; o2[e] = coef[Co + k + e] * w[e]; acc = fmaf(coef[2 * Co + k + e], w[e], acc); }
        *(u32x4*)(wk + (long)row * (2 * Co) + k) = f_to_chunk<T>(o1);
        *(u32x4*)(wk + (long)row * (2 * Co) + Co + k) = f_to_chunk<T>(o2);
    }
    acc = wave_sum(acc);
    if (lane == 0) bias[row] = acc;
}
template <typename T>
static int bn_fold_y_t(int Co, int Ci, const void* wd, const float* gamma, const float* invstd, const float* mean, const float* c1c2, void* w_kcat,
                       float* bias, hipStream_t s) {
    hipLaunchKernelGGL((bn_fold_y_kernel<T>), dim3((Ci + 3) / 4), dim3(256), (size_t)Co * 12, s, (const T*)wd, gamma, invstd, mean, c1c2, c1c2 + Co, (T*)w_kcat, bias,
                       Co, Ci);
    RPE_CHECK_LAUNCH();
    note_kernel("bn_fold_y_kernel");
    return 0;
}

// data gradient from A = [dz (M x Co) | y (M x Co)] and w_kcat [Ci][2 Co] (bn_fold_y_t), + bias, + the shortcut gradient, with the
// fused BN-backward epilogue of the layer behind (the previous block's bn3: ReLU mask bits, partial sums)
template <typename T>
static int conv1x1_dgrad_kcat_y_t(const rpe_conv_desc* d, const void* dz, const void* y, const void* w_kcat, const float* bias, void* dx, const void* addend,
                                  const rpe_bn_bwd_epilogue* bn, hipStream_t s) {
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)dz; a.A2 = (const T*)y; a.Bw = (const T*)w_kcat; a.C = (T*)dx;
    a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c; a.K = 2 * d->out_c; a.K1 = d->out_c;
    a.lda = d->out_c; a.lda2 = d->out_c; a.ldb = a.K; a.ldc = d->in_c;
    a.bias = bias;
    a.addend = (const T*)addend; a.ld_add = d->in_c;
    a.role = 1;
    if (bn) {
        if (!bn->y || !bn->mean || !bn->invstd || !bn->stats_part) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_kcat_y: y, mean, invstd, stats_part are required");
        a.bn_mode = bn->a_mask ? 4 : bn->a_out ? 1 : (bn->scale && bn->shift ? 2 : 3);
        a.bn_y = (const T*)bn->y; a.bn_a = (const T*)bn->a_out; a.bn_mask = bn->a_mask;
        a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
        a.stats_part = bn->stats_part;
    }
    return launch_nt<T>(a, MODE_DENSE, s);
}

// weight gradient of the same conv from dz, y and x (no dy):  dW[k][n] = A_k (Dz[k][n] - c1_k s1[n]) + C'_k (Dy[k][n] - mean_k s1[n])
// with Dz = dz^T x, Dy = y^T x, s1 = colsum(x): ONE row-concatenated TN launch (P = dz | P2 = y | all-ones tile) + an element-wise combine.
// dp rows: [0, Co) Dz | [I1, I1 + Co) Dy | row `ones_row` s1, I1 = roundup(Co, 128).
__global__ __launch_bounds__(256) void wgrad_fold_y_combine_kernel(float* __restrict__ dw, const float* __restrict__ dp, const float* __restrict__ gamma,
                                                                  const float* __restrict__ invstd, const float* __restrict__ mean,
                                                                  const float* __restrict__ c1, const float* __restrict__ c2, int Co, int Ci, int I1, int ones_row) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one float4 of dw [Co][Ci]
    const long total = (long)Co * Ci / 4;
    if (idx >= total) return;
    const int k = (int)(idx / (Ci / 4)), n = (int)(idx - (long)k * (Ci / 4)) * 4;
    const float a = gamma[k] * invstd[k];
    const float cp = -a * invstd[k] * c2[k];
    const f32x4 dzv = *(const f32x4*)(dp + (long)k * Ci + n), dyv = *(const f32x4*)(dp + (long)(I1 + k) * Ci + n), s1 = *(const f32x4*)(dp + (long)ones_row * Ci + n);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a * (dzv[j] - c1[k] * s1[j]) + cp * (dyv[j] - mean[k] * s1[j]);
    *(f32x4*)(dw + (long)k * Ci + n) = o;
}
struct WFoldYPlan { long slab_off, total; int I1, ones_row; };
template <typename T> static void wfold_y_args(TNArgs<T>& a, long M, int Co, int Ci, const WFoldYPlan& pl) {
    memset(&a, 0, sizeof(a));
    a.M = (int)M; a.I = pl.ones_row + 1; a.J = Ci; a.ldp = Co; a.ldq = Ci; a.ldd = Ci;
    a.ldp2 = Co; a.I1 = pl.I1; a.I2 = Co; a.ones_i0 = pl.ones_row; a.p_cols = Co;
}
template <typename T> static int wfold_y_plan(long M, int Co, int Ci, WFoldYPlan& pl) {
    pl.I1 = (Co + 127) / 128 * 128;
    pl.ones_row = 2 * pl.I1;
    TNArgs<T> a;
    wfold_y_args<T>(a, M, Co, Ci, pl);
    a.P2 = (const T*)16;   // (placeholder: the query plans the concatenated problem, nothing is dereferenced)
    long slab = 0;
    if (int e = launch_tn<T>(a, MODE_DENSE, nullptr, &slab)) return e;
    pl.slab_off = ((long)(pl.ones_row + 1) * Ci * 4 + 255) / 256 * 256;
    pl.total = pl.slab_off + slab;
    return 0;
}
template <typename T>
static int conv1x1_wgrad_folded_y_t(const rpe_conv_desc* d, const void* dz, const void* y, const void* x, const float* gamma, const float* invstd,
                                    const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, long* query, hipStream_t s) {
    const long M = (long)d->batch * d->in_h * d->in_w;
    const int Co = d->out_c, Ci = d->in_c;
    if ((Co % 64) || (Ci % 64)) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded_y: out_c and in_c must be multiples of 64");
    WFoldYPlan pl;
    if (int e = wfold_y_plan<T>(M, Co, Ci, pl)) return e;
    if (query) { *query = pl.total; return 0; }
    if (!scratch || scratch_bytes < pl.total) return rpe_set_error(RPE_ERR_WORKSPACE, "conv1x1_wgrad_folded_y: scratch smaller than rpe_conv1x1_wgrad_folded_y_scratch_bytes()");
    float* dp = (float*)scratch;
    TNArgs<T> a;
    wfold_y_args<T>(a, M, Co, Ci, pl);
    a.P = (const T*)dz; a.P2 = (const T*)y; a.Q = (const T*)x; a.D = dp;
    a.slab = (float*)((char*)scratch + pl.slab_off); a.slab_bytes = scratch_bytes - pl.slab_off;
    if (int e = launch_tn<T>(a, MODE_DENSE, s)) return e;
    prof_split(s, "wgrad_fold_y_combine_kernel");
    hipLaunchKernelGGL(wgrad_fold_y_combine_kernel, dim3((unsigned)(((long)Co * Ci / 4 + 255) / 256)), dim3(256), 0, s, dw, dp, gamma, invstd, mean, c1c2, c1c2 + Co, Co, Ci,
                       pl.I1, pl.ones_row);
    RPE_CHECK_LAUNCH();
    return 0;
}

// The stem's image operand is ZERO-BORDERED: x4 [B][H + 6][W + 6][4] with the image at rows / columns [3, 3 + H) x [3, 3 + W) and zeros
// around it (RPE_STEM_PAD = 3 = conv1's padding).  Over it the 7x7 / stride 2 / pad 3 conv is an 8 x 8 / stride 2 / pad 0 conv whose 8th
// tap row / column meets zero weights (rpe_pack_stem_weight): every 16-byte chunk of an im2col row is in bounds and aligned.
static void stem_gather(Gather& g, int H, int W) {
    const int Ho = out_dim(H, 7, 2, 3), Wo = out_dim(W, 7, 2, 3);
    g.H = H + 2 * RPE_STEM_PAD; g.W = W + 2 * RPE_STEM_PAD; g.C = 4; g.Ho = Ho; g.Wo = Wo; g.R = 8; g.S = 8;
    g.sn = 2; g.sd_shift = 0; g.base_h = 0; g.base_w = 0; g.tap_sign = 1;
    g.div_hw = make_fastdiv(Ho * Wo); g.div_w = make_fastdiv(Wo);
    g.img_stride = (long)g.H * g.W * 4;
}

template <typename T>
static int stem_fwd_t(const void* x4, const void* w, void* y, float* stats, const float* bias, int relu, int B, int H, int W, hipStream_t s) {
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    stem_gather(a.g, H, W);
    a.A = (const T*)x4; a.Bw = (const T*)w; a.C = (T*)y;
    a.M = B * a.g.Ho * a.g.Wo; a.N = 64; a.K = 224;   // 7 real kernel rows x (8 taps x 4 channels); the 8th, all-zero row of the packed weight is skipped
    a.lda = 4; a.ldb = 256; a.ldc = 64;
    a.a_elems = (long)B * a.g.img_stride;
    a.stats_part = stats;
    a.bias = bias; a.relu = relu;
    if (bias || relu) { a.role = 3; a.stats_part = nullptr; }
    if (H % 2 || W % 2) return rpe_set_error(RPE_ERR_SHAPE, "stem conv: even image dims");
    return launch_nt<T>(a, MODE_STEM, s);
}

template <typename T>
static int stem_wgrad_t(const void* x4, const void* dy, float* dw_packed, int B, int H, int W, void* slab, long slab_bytes, long* slab_query, hipStream_t s) {
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    stem_gather(a.g, H, W);
    a.P = (const T*)dy; a.Q = (const T*)x4; a.D = dw_packed;
    a.slab = (float*)slab; a.slab_bytes = slab_bytes;
    a.M = B * a.g.Ho * a.g.Wo; a.I = 64; a.J = 256;   // (all 8 x 8 packed taps: rows / columns 7 of dw_packed are not gradients -- rpe_unpack_stem_grad skips them)
    a.ldp = 64; a.ldq = 4; a.ldd = 256;
    a.q_elems = (long)B * a.g.img_stride;
    if (H % 2 || W % 2) return rpe_set_error(RPE_ERR_SHAPE, "stem conv: even image dims");
    return launch_tn<T>(a, MODE_STEM, s, slab_query);
}

// Few-row fp32 Linear (a rollout frame is ONE row; the heads' layers are fp32): y[m][n] = x[m][:] . w[n][:] (+bias) (+addend) (relu).
// The MFMA tile kernel put 8..16 workgroups on the chip for these (135 us for 1 x 3655 -> 1024: 15 MB of weights at 110 GB/s).
// Here one workgroup per output column: 256 lanes stride over K (coalesced dword loads, any alignment), fixed-order reduction
// (lane butterfly, then the 4 waves in order) -- the weights stream once at chip rate.
template <int MT>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ y, int ldy, int M, int N, int K,
                                                          int relu, const float* __restrict__ addend, int ld_add) {
    __shared__ float red[4][MT];
    const int n = blockIdx.x, m0 = blockIdx.y * MT;
    const float* wr = w + (long)n * ldw;
    const float* xr[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xr[i] = x + (long)(m0 + i < M ? m0 + i : M - 1) * ldx;   // (rows past M repeat the last one; never stored)
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    int k = threadIdx.x;
    for (; k + 768 < K; k += 1024) {   // four independent weight loads in flight per lane
        const float w0 = wr[k], w1 = wr[k + 256], w2 = wr[k + 512], w3 = wr[k + 768];
#pragma unroll
        for (int i = 0; i < MT; ++i)
            acc[i] = fmaf(xr[i][k + 768], w3, fmaf(xr[i][k + 512], w2, fmaf(xr[i][k + 256], w1, fmaf(xr[i][k], w0, acc[i]))));
    }
    for (; k < K; k += 256) {
        const float w0 = wr[k];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = fmaf(xr[i][k], w0, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) red[wave][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < MT && m0 + (int)threadIdx.x < M) {
        const int i = threadIdx.x, m = m0 + i;
        float v = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        if (bias) v += bias[n];
        if (addend) v += addend[(long)m * ld_add + n];
        if (relu) v = fmaxf(v, 0.f);
        y[(long)m * ldy + n] = v;
    }
}

static const int kLinearRowsMax = 8;   // rows up to which the per-column kernel is used (the weights are re-read per group of 8 rows)

static int linear_rows(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M, int N, int K, int relu,
                       const float* addend, int ld_add, hipStream_t s) {
    note_kernel("linear_rows_kernel");
    const dim3 block(256);
    if (M == 1) hipLaunchKernelGGL((linear_rows_kernel<1>), dim3(N, 1), block, 0, s, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add);
    else if (M == 2) hipLaunchKernelGGL((linear_rows_kernel<2>), dim3(N, 1), block, 0, s, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add);
    else if (M <= 4) hipLaunchKernelGGL((linear_rows_kernel<4>), dim3(N, 1), block, 0, s, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add);
    else hipLaunchKernelGGL((linear_rows_kernel<8>), dim3(N, (M + 7) / 8), block, 0, s, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int linear_fwd_t(const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int M, int N, int K, int relu,
                        const void* addend, int ld_add, hipStream_t s, void* ws = nullptr, long ws_bytes = 0, long* ws_query = nullptr) {
    if (ws_query) {
        const int S = (M > kLinearRowsMax || sizeof(T) != 4) ? nt_split_plan(M, N, K, 4 * Elem<T>::kChunk, nullptr, true) : 1;
        *ws_query = S > 1 ? nt_split_slab_bytes(M, N, S) : 0;
        return 0;
    }
    if constexpr (sizeof(T) == 4) {
        if (x && w && y && M >= 1 && M <= kLinearRowsMax && N >= 1 && K >= 1 && ldx >= K && ldw >= K && ldy >= N)
            return linear_rows((const float*)x, ldx, (const float*)w, ldw, bias, (float*)y, ldy, M, N, K, relu, (const float*)addend, ld_add, s);
    }
    NTArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.A = (const T*)x; a.Bw = (const T*)w; a.C = (T*)y;
    a.M = M; a.N = N; a.K = K; a.lda = ldx; a.ldb = ldw; a.ldc = ldy;
    a.bias = bias; a.addend = (const T*)addend; a.ld_add = ld_add; a.relu = relu;
    a.role = 2;
    if (ws) {   // few output tiles, long K (the heads' layers at a few hundred rows): split K over grid.y through the workspace
        const int S = nt_split_plan(M, N, K, 4 * Elem<T>::kChunk, nullptr, true);
        if (S > 1 && ws_bytes >= nt_split_slab_bytes(M, N, S) && !(((uintptr_t)ws) & 15)) { a.slab = (float*)ws; a.slab_bytes = ws_bytes; a.splits = S; }
    }
    return launch_nt<T>(a, MODE_DENSE, s);
}

template <typename T>
static int linear_wgrad_t(const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int M, int N, int K, void* slab, long slab_bytes,
                          int accumulate, long* slab_query, hipStream_t s) {
    TNArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.P = (const T*)dy; a.Q = (const T*)x; a.D = dw;
    a.slab = (float*)slab; a.slab_bytes = slab_bytes; a.accumulate = accumulate;
    a.M = M; a.I = N; a.J = K; a.ldp = lddy; a.ldq = ldx; a.ldd = lddw;
    return launch_tn<T>(a, MODE_DENSE, s, slab_query);
}

#define DISPATCH(dtype, fn, ...)                                          \
    do {                                                                  \
        if ((dtype) == RPE_F32) return fn<float>(__VA_ARGS__);            \
        if ((dtype) == RPE_BF16) return fn<bf16>(__VA_ARGS__);            \
        if ((dtype) == RPE_F16) return fn<f16>(__VA_ARGS__);            \
        return rpe_set_error(RPE_ERR_DTYPE, #fn ": unsupported dtype");   \
    } while (0)

extern "C" {

int rpe_conv_out_hw(const rpe_conv_desc* d, int* ho, int* wo) {
    if (int e = check_desc(d)) return e;
    *ho = out_dim(d->in_h, d->kh, d->stride, d->pad);
    *wo = out_dim(d->in_w, d->kw, d->stride, d->pad);
    return 0;
}

long rpe_conv_stats_tiles(long rows) { return (rows + 127) / 128; }

long rpe_conv2d_fwd_stats_tiles(const rpe_conv_desc* d, int dtype) {
    if (!d || d->stride <= 0) return 0;
    if (const int rt = halo_rt(d, dtype, d->in_c)) return ((long)d->batch * d->in_h + rt - 1) / rt;
    return ((long)d->batch * out_dim(d->in_h, d->kh, d->stride, d->pad) * out_dim(d->in_w, d->kw, d->stride, d->pad) + 127) / 128;
}

long rpe_conv2d_dgrad_stats_tiles(const rpe_conv_desc* d, int dtype) {
    if (!d) return 0;
    if (const int rt = halo_rt(d, dtype, d->out_c)) return ((long)d->batch * d->in_h + rt - 1) / rt;
    const long rows = (long)d->batch * d->in_h * d->in_w;
    if (!is_dense(d) && dgrad_parity(d)) return 4 * ((rows / 4 + 127) / 128);
    return (rows + 127) / 128;
}

int rpe_conv2d_fwd(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* y, float* stats_part, void* stream) {
    if (int e = check_desc(d)) return e;
    DISPATCH(dtype, conv_fwd_t, d, x, w_krsc, y, stats_part, nullptr, nullptr, 0, (hipStream_t)stream);
}

int rpe_conv2d_fwd_affine(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* out, const float* bias, const void* addend,
                          int relu, void* stream) {
    if (int e = check_desc(d)) return e;
    DISPATCH(dtype, conv_fwd_t, d, x, w_krsc, out, nullptr, bias, addend, relu, (hipStream_t)stream);
}

static int fwd_ws_query(const rpe_conv_desc* d, int dtype, long* bytes) {
    DISPATCH(dtype, conv_fwd_t, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, 0L, bytes);
}

long rpe_conv2d_fwd_affine_workspace_bytes(const rpe_conv_desc* d, int dtype) {
    long bytes = 0;
    if (check_desc(d) || fwd_ws_query(d, dtype, &bytes)) return -1;
    return bytes;
}

int rpe_conv2d_fwd_affine_ws(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* out, const float* bias, const void* addend,
                             int relu, void* workspace, long workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (!bias && !addend && !relu) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_fwd_affine_ws: the inference epilogue needs a bias, an addend or ReLU");
    DISPATCH(dtype, conv_fwd_t, d, x, w_krsc, out, nullptr, bias, addend, relu, (hipStream_t)stream, workspace, workspace_bytes, nullptr);
}

int rpe_conv2d_dgrad(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, void* stream) {
    if (int e = check_desc(d)) return e;
    DISPATCH(dtype, conv_dgrad_t, d, dy, w_crsk, dx, addend, nullptr, (hipStream_t)stream);
}

int rpe_conv2d_dgrad_bn(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dz, const void* addend,
                        const rpe_bn_bwd_epilogue* bn, void* stream) {
    if (int e = check_desc(d)) return e;
    if (!bn) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_dgrad_bn: null epilogue descriptor");
    DISPATCH(dtype, conv_dgrad_t, d, dy, w_crsk, dz, addend, bn, (hipStream_t)stream);
}

long rpe_bn_bwd_fold_scratch_bytes(int dtype, int out_c, int in_c) {
    long bytes = 0;
    int rc;
    if (out_c <= 0 || in_c <= 0) return -1;
    if (dtype == RPE_F32) rc = bn_fold_scratch_t<float>(out_c, in_c, &bytes);
    else if (dtype == RPE_BF16) rc = bn_fold_scratch_t<bf16>(out_c, in_c, &bytes);
    else if (dtype == RPE_F16) rc = bn_fold_scratch_t<f16>(out_c, in_c, &bytes);
    else return -1;
    return rc ? -1 : bytes;
}

int rpe_bn_bwd_fold_conv1x1(int dtype, int out_c, int in_c, const void* w_fwd, const void* w_dgrad, const float* gamma, const float* invstd,
                            const float* mean, const float* c1c2, void* w_kcat, float* bias, void* scratch, long scratch_bytes, void* stream) {
    if (out_c <= 0 || in_c <= 0 || (out_c % 64) || (in_c % 8) || !w_fwd || !w_dgrad || !gamma || !invstd || !mean || !c1c2 || !w_kcat || !bias)
        return rpe_set_error(RPE_ERR_SHAPE, "bn_bwd_fold_conv1x1: bad arguments (out_c % 64 == 0, in_c % 8 == 0)");
    DISPATCH(dtype, bn_fold_t, out_c, in_c, w_fwd, w_dgrad, gamma, invstd, mean, c1c2, w_kcat, bias, scratch, scratch_bytes, (hipStream_t)stream);
}

static int wfold_dispatch(const rpe_conv_desc* d, int dtype, const void* dz, const void* a_in, const float* w_master, const float* gamma, const float* invstd,
                          const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, long* query, hipStream_t s) {
    DISPATCH(dtype, conv1x1_wgrad_folded_t, d, dz, a_in, w_master, gamma, invstd, mean, c1c2, dw, scratch, scratch_bytes, query, s);
}

long rpe_conv1x1_wgrad_folded_scratch_bytes(const rpe_conv_desc* d, int dtype) {
    long bytes = 0;
    if (check_desc(d) || wfold_dispatch(d, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &bytes, nullptr)) return -1;
    return bytes;
}

int rpe_conv1x1_wgrad_folded(const rpe_conv_desc* d, int dtype, const void* dz, const void* a_in, const float* w_master, const float* gamma,
                             const float* invstd, const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded: 1x1 / stride 1 / no padding only");
    if (!dz || !a_in || !w_master || !gamma || !invstd || !mean || !c1c2 || !dw) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded: null argument");
    return wfold_dispatch(d, dtype, dz, a_in, w_master, gamma, invstd, mean, c1c2, dw, scratch, scratch_bytes, nullptr, (hipStream_t)stream);
}


int rpe_bn_bwd_fold_y_conv1x1(int dtype, int out_c, int in_c, const void* w_dgrad, const float* gamma, const float* invstd, const float* mean,
                              const float* c1c2, void* w_kcat, float* bias, void* stream) {
    if (out_c <= 0 || in_c <= 0 || (out_c % 64) || out_c > 4096 || !w_dgrad || !gamma || !invstd || !mean || !c1c2 || !w_kcat || !bias)
        return rpe_set_error(RPE_ERR_SHAPE, "bn_bwd_fold_y_conv1x1: bad arguments (out_c % 64 == 0)");
    DISPATCH(dtype, bn_fold_y_t, out_c, in_c, w_dgrad, gamma, invstd, mean, c1c2, w_kcat, bias, (hipStream_t)stream);
}

int rpe_conv1x1_dgrad_kcat_y(const rpe_conv_desc* d, int dtype, const void* dz, const void* y, const void* w_kcat, const float* bias, void* dx,
                             const void* addend, const rpe_bn_bwd_epilogue* bn, void* stream) {
    if (int e = check_desc(d)) return e;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_kcat_y: 1x1 / stride 1 / no padding only");
    if (!dz || !y || !w_kcat || !bias || !dx || (d->out_c % 64)) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_kcat_y: bad arguments (out_c % 64 == 0)");
    DISPATCH(dtype, conv1x1_dgrad_kcat_y_t, d, dz, y, w_kcat, bias, dx, addend, bn, (hipStream_t)stream);
}

static int wfold_y_dispatch(const rpe_conv_desc* d, int dtype, const void* dz, const void* y, const void* x, const float* gamma, const float* invstd,
                            const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, long* query, hipStream_t s) {
    DISPATCH(dtype, conv1x1_wgrad_folded_y_t, d, dz, y, x, gamma, invstd, mean, c1c2, dw, scratch, scratch_bytes, query, s);
}

long rpe_conv1x1_wgrad_folded_y_scratch_bytes(const rpe_conv_desc* d, int dtype) {
    long bytes = 0;
    if (check_desc(d) || wfold_y_dispatch(d, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &bytes, nullptr)) return -1;
    return bytes;
}

int rpe_conv1x1_wgrad_folded_y(const rpe_conv_desc* d, int dtype, const void* dz, const void* y, const void* x, const float* gamma, const float* invstd,
                               const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded_y: 1x1 / stride 1 / no padding only");
    if (!dz || !y || !x || !gamma || !invstd || !mean || !c1c2 || !dw) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_folded_y: null argument");
    return wfold_y_dispatch(d, dtype, dz, y, x, gamma, invstd, mean, c1c2, dw, scratch, scratch_bytes, nullptr, (hipStream_t)stream);
}

int rpe_conv1x1_wgrad_combine(const rpe_conv_desc* d, const float* dzt_a, const float* gram, const float* w_master, const float* gamma, const float* invstd,
                              const float* mean, const float* c1c2, float* dw, void* stream) {
    if (int e = check_desc(d)) return e;
    const int Co = d->out_c, Ci = d->in_c;
    if (!is_dense(d) || (Co % 16) || (Ci % 4) || Ci > 512) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_combine: 1x1 / stride 1, out_c % 16 == 0, in_c % 4 == 0, in_c <= 512");
    if (!dzt_a || !gram || !w_master || !gamma || !invstd || !mean || !c1c2 || !dw) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_wgrad_combine: null argument");
    note_kernel("wgrad_fold_combine_kernel");
    const size_t lds = (size_t)(16 * Ci + (Ci < 128 ? Ci : 128) * 64) * 4;
    hipLaunchKernelGGL(wgrad_fold_combine_kernel, dim3(Co / 16, (Ci + 63) / 64), dim3(256), lds, (hipStream_t)stream, dw, dzt_a, gram, gram + (long)gram_ones_row(Ci) * Ci,
                       w_master, gamma, invstd, mean, c1c2, c1c2 + Co, Co, Ci);
    RPE_CHECK_LAUNCH();
    return 0;
}

long rpe_conv1x1_dgrad_bn_t_workspace_bytes(const rpe_conv_desc* d, int P) {
    if (check_desc(d) || !is_dense(d) || (d->in_c % 128) || (P != 64 && P != 128)) return -1;
    return nt_tfuse_grid((long)d->batch * d->in_h * d->in_w, d->in_c) * 128L * P * 4;
}

int rpe_conv1x1_dgrad_bn_t(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dz, const void* addend, const rpe_bn_bwd_epilogue* bn,
                           const void* a_prev, int P, float* t_out, void* workspace, long workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (!is_dense(d) || (d->in_c % 128) || (d->out_c % 8)) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_bn_t: 1x1 / stride 1, in_c % 128 == 0");
    if (P != 64 && P != 128) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_bn_t: the second operand has 64 or 128 channels");
    if (!dy || !w_crsk || !dz || !bn || !bn->a_mask || !bn->stats_part || bn->y || !a_prev || !t_out || !workspace)
        return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_bn_t: dy, w, dz, bn (a_mask + stats_part, y = NULL), a_prev, t_out and the workspace are required");
    if ((((uintptr_t)a_prev) | ((uintptr_t)t_out) | ((uintptr_t)workspace)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "conv1x1_dgrad_bn_t: 16-byte aligned operands");
    if (dtype == RPE_BF16) return conv1x1_dgrad_bn_t_t<bf16>(d, dy, w_crsk, dz, addend, bn, a_prev, P, t_out, workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == RPE_F16) return conv1x1_dgrad_bn_t_t<f16>(d, dy, w_crsk, dz, addend, bn, a_prev, P, t_out, workspace, workspace_bytes, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "conv1x1_dgrad_bn_t: 16-bit element types only");
}

long rpe_gram_ones_row(int C) { return gram_ones_row(C); }

long rpe_gram_workspace_bytes(int dtype, long M, int C) {
    long bytes = 0;
    int rc;
    if (M <= 0 || C <= 0 || (C % 64)) return -1;
    if (dtype == RPE_F32) rc = gram_t<float>(nullptr, M, C, nullptr, nullptr, 0, &bytes, nullptr);
    else if (dtype == RPE_BF16) rc = gram_t<bf16>(nullptr, M, C, nullptr, nullptr, 0, &bytes, nullptr);
    else if (dtype == RPE_F16) rc = gram_t<f16>(nullptr, M, C, nullptr, nullptr, 0, &bytes, nullptr);
    else return -1;
    return rc ? -1 : bytes;
}

int rpe_gram(int dtype, const void* x, long M, int C, float* out, void* workspace, long workspace_bytes, void* stream) {
    if (!x || !out || M <= 0 || C <= 0 || (C % 64)) return rpe_set_error(RPE_ERR_SHAPE, "gram: x [M][C] with C % 64 == 0");
    DISPATCH(dtype, gram_t, x, M, C, out, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

int rpe_conv1x1_fwd_bn(const rpe_conv_desc* d, int dtype, const void* x, const void* w, void* out, void* y_out, const float* scale, const float* shift,
                       const void* residual, const float* res_scale, const float* res_shift, unsigned char* relu_mask, void* stream) {
    if (int e = check_desc(d)) return e;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_fwd_bn: 1x1 / stride 1 / no padding only");
    if (dtype == RPE_F32) return rpe_set_error(RPE_ERR_DTYPE, "conv1x1_fwd_bn: 16-bit element types only (the fp32 path keeps the two-pass form)");
    if (!x || !w || !out || !scale || !shift || (res_scale && (!res_shift || !residual)) || (d->out_c % 8))
        return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_fwd_bn: bad arguments");
    if (dtype == RPE_BF16) return conv1x1_fwd_bn_t<bf16>(d, x, w, out, y_out, scale, shift, residual, res_scale, res_shift, relu_mask, (hipStream_t)stream);
    if (dtype == RPE_F16) return conv1x1_fwd_bn_t<f16>(d, x, w, out, y_out, scale, shift, residual, res_scale, res_shift, relu_mask, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "conv1x1_fwd_bn: unsupported dtype");
}

int rpe_conv1x1_dgrad_kcat(const rpe_conv_desc* d, int dtype, const void* dz, const void* a_in, const void* w_kcat, const float* bias, void* dx,
                           const rpe_bn_bwd_epilogue* bn, void* stream) {
    if (int e = check_desc(d)) return e;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return rpe_set_error(RPE_ERR_SHAPE, "conv1x1_dgrad_kcat: 1x1 / stride 1 / no padding only");
    DISPATCH(dtype, conv1x1_dgrad_kcat_t, d, dz, a_in, w_kcat, bias, dx, bn, (hipStream_t)stream);
}

int rpe_conv2d_wgrad(const rpe_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_krsc, void* stream) {
    if (int e = check_desc(d)) return e;
    DISPATCH(dtype, conv_wgrad_t, d, x, dy, dw_krsc, nullptr, 0L, nullptr, (hipStream_t)stream);
}

static int wgrad_ws_query(const rpe_conv_desc* d, int dtype, long* bytes) {
    DISPATCH(dtype, conv_wgrad_t, d, nullptr, nullptr, nullptr, nullptr, 0L, bytes, nullptr);
}

long rpe_conv2d_wgrad_workspace_bytes(const rpe_conv_desc* d, int dtype) {
    if (check_desc(d)) return -1;
    long bytes = 0;
    if (wgrad_ws_query(d, dtype, &bytes)) return -1;
    return bytes;
}

int rpe_conv2d_wgrad_det(const rpe_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_krsc, void* workspace, long workspace_bytes,
                         void* stream) {
    if (int e = check_desc(d)) return e;
    if (!workspace) return rpe_set_error(RPE_ERR_WORKSPACE, "conv2d_wgrad_det: null workspace");
    long need = 0;
    if (int e = wgrad_ws_query(d, dtype, &need)) return e;
    if (workspace_bytes < need) return rpe_set_error(RPE_ERR_WORKSPACE, "conv2d_wgrad_det: workspace smaller than rpe_conv2d_wgrad_workspace_bytes()");
    DISPATCH(dtype, conv_wgrad_t, d, x, dy, dw_krsc, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

int rpe_conv2d_wgrad_halo_min_width(int width) { return wgrad_halo_set_min_w(width); }

long rpe_x4_bytes(int dtype, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return (long)B * (H + 2 * RPE_STEM_PAD) * (W + 2 * RPE_STEM_PAD) * 4 * (dtype == RPE_F32 ? 4 : 2);
}

int rpe_stem_conv_fwd(int dtype, const void* x4, const void* w_packed, void* y, float* stats_part, int B, int H, int W, void* stream) {
    if (B <= 0 || H < 7 || W < 7) return rpe_set_error(RPE_ERR_SHAPE, "stem_conv: bad shape");
    DISPATCH(dtype, stem_fwd_t, x4, w_packed, y, stats_part, nullptr, 0, B, H, W, (hipStream_t)stream);
}

int rpe_stem_conv_fwd_affine(int dtype, const void* x4, const void* w_packed, void* out, const float* bias, int relu, int B, int H, int W,
                             void* stream) {
    if (B <= 0 || H < 7 || W < 7) return rpe_set_error(RPE_ERR_SHAPE, "stem_conv: bad shape");
    DISPATCH(dtype, stem_fwd_t, x4, w_packed, out, nullptr, bias, relu, B, H, W, (hipStream_t)stream);
}

int rpe_stem_conv_wgrad(int dtype, const void* x4, const void* dy, float* dw_packed, int B, int H, int W, void* stream) {
    if (B <= 0 || H < 7 || W < 7) return rpe_set_error(RPE_ERR_SHAPE, "stem_conv: bad shape");
    DISPATCH(dtype, stem_wgrad_t, x4, dy, dw_packed, B, H, W, nullptr, 0L, nullptr, (hipStream_t)stream);
}

static int stem_ws_query(int dtype, int B, int H, int W, long* bytes) { DISPATCH(dtype, stem_wgrad_t, nullptr, nullptr, nullptr, B, H, W, nullptr, 0L, bytes, nullptr); }

long rpe_stem_conv_wgrad_workspace_bytes(int dtype, int B, int H, int W) {
    long bytes = 0;
    if (B <= 0 || H < 7 || W < 7 || stem_ws_query(dtype, B, H, W, &bytes)) return -1;
    return bytes;
}

int rpe_stem_conv_wgrad_det(int dtype, const void* x4, const void* dy, float* dw_packed, int B, int H, int W, void* workspace, long workspace_bytes,
                            void* stream) {
    if (B <= 0 || H < 7 || W < 7) return rpe_set_error(RPE_ERR_SHAPE, "stem_conv: bad shape");
    long need = 0;
    if (int e = stem_ws_query(dtype, B, H, W, &need)) return e;
    if (!workspace || workspace_bytes < need) return rpe_set_error(RPE_ERR_WORKSPACE, "stem_conv_wgrad_det: workspace smaller than rpe_stem_conv_wgrad_workspace_bytes()");
    DISPATCH(dtype, stem_wgrad_t, x4, dy, dw_packed, B, H, W, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

int rpe_linear_fwd(int dtype, const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int M, int N, int K,
                   int relu, const void* addend, int ld_add, void* stream) {
    DISPATCH(dtype, linear_fwd_t, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add, (hipStream_t)stream);
}

long rpe_linear_fwd_workspace_bytes(int dtype, int M, int N, int K) {
    long bytes = 0;
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    int rc;
    if (dtype == RPE_F32) rc = linear_fwd_t<float>(nullptr, K, nullptr, K, nullptr, nullptr, N, M, N, K, 0, nullptr, 0, nullptr, nullptr, 0, &bytes);
    else if (dtype == RPE_BF16) rc = linear_fwd_t<bf16>(nullptr, K, nullptr, K, nullptr, nullptr, N, M, N, K, 0, nullptr, 0, nullptr, nullptr, 0, &bytes);
    else if (dtype == RPE_F16) rc = linear_fwd_t<f16>(nullptr, K, nullptr, K, nullptr, nullptr, N, M, N, K, 0, nullptr, 0, nullptr, nullptr, 0, &bytes);
    else return -1;
    return rc ? -1 : bytes;
}

int rpe_linear_fwd_ws(int dtype, const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int M, int N, int K,
                      int relu, const void* addend, int ld_add, void* workspace, long workspace_bytes, void* stream) {
    DISPATCH(dtype, linear_fwd_t, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, addend, ld_add, (hipStream_t)stream, workspace, workspace_bytes, nullptr);
}

int rpe_linear_wgrad(int dtype, const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int M, int N, int K, void* stream) {
    DISPATCH(dtype, linear_wgrad_t, dy, lddy, x, ldx, dw, lddw, M, N, K, nullptr, 0L, 1, nullptr, (hipStream_t)stream);
}

static int linear_ws_query(int dtype, int M, int N, int K, long* bytes) {
    DISPATCH(dtype, linear_wgrad_t, nullptr, N, nullptr, K, nullptr, K, M, N, K, nullptr, 0L, 0, bytes, nullptr);
}

long rpe_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K) {
    long bytes = 0;
    if (M <= 0 || N <= 0 || K <= 0 || linear_ws_query(dtype, M, N, K, &bytes)) return -1;
    return bytes;
}

int rpe_linear_wgrad_det(int dtype, const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int M, int N, int K, int accumulate,
                         void* workspace, long workspace_bytes, void* stream) {
    long need = 0;
    if (int e = linear_ws_query(dtype, M, N, K, &need)) return e;
    if (!workspace || workspace_bytes < need) return rpe_set_error(RPE_ERR_WORKSPACE, "linear_wgrad_det: workspace smaller than rpe_linear_wgrad_workspace_bytes()");
    DISPATCH(dtype, linear_wgrad_t, dy, lddy, x, ldx, dw, lddw, M, N, K, workspace, workspace_bytes, accumulate, nullptr, (hipStream_t)stream);
}

}  // extern "C"

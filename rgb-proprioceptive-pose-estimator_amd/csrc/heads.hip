// Head-side kernels of the pose path: early-feature auxiliary head, depth head, LSTM cell,
// PoseDistanceLoss (+ on-device validation metrics), Adam, weight packing.  All fp32 except
// where a tensor of the conv trunk (compute type T) is read or written.
#include "common.h"

namespace rpe {

// ---------------------------------------------------------------------------------------------
// aux head: Conv2d(64 -> 1, 1x1, bias) -> MaxPool2d(2) -> Flatten  [* depth feature]
// reference: models/naive.py:223-231,318-330.  a1 = relu(bn1(conv1 x)) is [B][H][W][64] (T).
// A group of LPP = 64/CE lanes owns one output pixel (2x2 input window).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void aux_fwd_kernel(const T* __restrict__ a1, const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ depth_feat, float* __restrict__ out, long ld_out,
                                                     float* __restrict__ raw, unsigned char* __restrict__ idx, int B, int H, int W) {
    constexpr int CE = Elem<T>::kChunk, LPP = 64 / CE;
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo;
    const int sub = threadIdx.x % LPP;
    float wv[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) wv[e] = w[sub * CE + e];
    const float bv = bias[0];
    const int groups = blockDim.x / LPP;
    for (long o = (long)blockIdx.x * groups + threadIdx.x / LPP; o < total; o += (long)gridDim.x * groups) {
        long t = o;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ih = oh * 2 + (k >> 1), iw = ow * 2 + (k & 1);
            float v[CE];
            chunk_to_f<T>(*(const u32x4*)(a1 + (((long)b * H + ih) * W + iw) * 64 + sub * CE), v);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < CE; ++e) d += v[e] * wv[e];
#pragma unroll
            for (int s = 1; s < LPP; s <<= 1) d += __shfl_xor(d, s);
            d += bv;
            if (d > best || d != d) { best = d; bi = k; }
        }
        if (sub == 0) {
            const long pos = (long)oh * Wo + ow;
            const float df = depth_feat ? depth_feat[(long)b * Ho * Wo + pos] : 1.f;
            out[(long)b * ld_out + pos] = best * df;
            raw[(long)b * Ho * Wo + pos] = best;
            idx[(long)b * Ho * Wo + pos] = (unsigned char)bi;
        }
    }
}

// backward: d_a1 (full tensor, zero except the winning pixel; skipped when null), dw (64), dbias, d_depth_feat
template <typename T>
__global__ __launch_bounds__(256) void aux_bwd_kernel(const float* __restrict__ dout, long ld_dout, const T* __restrict__ a1,
                                                     const float* __restrict__ w, const float* __restrict__ depth_feat,
                                                     const float* __restrict__ raw, const unsigned char* __restrict__ idx,
                                                     T* __restrict__ d_a1, float* __restrict__ dw, float* __restrict__ dbias,
                                                     float* __restrict__ d_depth_feat, int B, int H, int W, float* __restrict__ part,
                                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_shift) {
    // bn_scale / bn_shift (round 4): `a1` is the RAW stem conv output y and the activated value is recomputed at the winner pixel,
    // a1 = round_T(relu(y * scale + shift)) -- the tensor itself is not written when the aux head rides on the stem's apply + pool pass
    constexpr int CE = Elem<T>::kChunk, LPP = 64 / CE;
    __shared__ float sh_dw[64];
    __shared__ float sh_db;
    if (threadIdx.x < 64) sh_dw[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) sh_db = 0.f;
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo;
    const int sub = threadIdx.x % LPP;
    float wv[CE], gw[CE], bsc[CE], bsh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { wv[e] = w[sub * CE + e]; gw[e] = 0.f; bsc[e] = bn_scale ? bn_scale[sub * CE + e] : 1.f; bsh[e] = bn_scale ? bn_shift[sub * CE + e] : 0.f; }
    float gb = 0.f;
    const int groups = blockDim.x / LPP;
    for (long o = (long)blockIdx.x * groups + threadIdx.x / LPP; o < total; o += (long)gridDim.x * groups) {
        long t = o;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const long pos = (long)oh * Wo + ow;
        const long flat = (long)b * Ho * Wo + pos;
        float d = dout[(long)b * ld_dout + pos];
        if (depth_feat) {
            if (sub == 0 && d_depth_feat) d_depth_feat[flat] = d * raw[flat];
            d *= depth_feat[flat];
        }
        const int bi = idx[flat];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ih = oh * 2 + (k >> 1), iw = ow * 2 + (k & 1);
            const long off = (((long)b * H + ih) * W + iw) * 64 + sub * CE;
            float g[CE];
            if (k == bi) {
                float v[CE];
                chunk_to_f<T>(*(const u32x4*)(a1 + off), v);
                if (bn_scale) {
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[e] = fmaxf(fmaf(v[e], bsc[e], bsh[e]), 0.f);
                    chunk_to_f<T>(f_to_chunk<T>(v), v);   // as the forward rounded it
                }
#pragma unroll
                for (int e = 0; e < CE; ++e) { g[e] = d * wv[e]; gw[e] += d * v[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < CE; ++e) g[e] = 0.f;
            }
            if (d_a1) *(u32x4*)(d_a1 + off) = f_to_chunk<T>(g);
            else if (k == bi) break;   // compact form (rpe_stem_bwd gathers the a1 gradient itself): only the winner is read
        }
        if (sub == 0) gb += d;
    }
    // lanes l, l + LPP, l + 2 LPP, .. of a wave own the same channels: fold them with shuffles first, so that LPP lanes per wave
    // (not all 64) touch the shared accumulators -- the 2048 same-address LDS atomics per block were most of this kernel's time
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        for (int o = LPP; o < 64; o <<= 1) gw[e] += __shfl_xor(gw[e], o);
    }
    for (int o = LPP; o < 64; o <<= 1) gb += __shfl_xor(gb, o);
    __shared__ float sh_w[4][65];   // per-wave sums, added in wave order below (no shared atomics: the block's sums are order-free)
    if ((threadIdx.x & 63) < LPP) {
#pragma unroll
        for (int e = 0; e < CE; ++e) sh_w[threadIdx.x >> 6][sub * CE + e] = gw[e];
        if (sub == 0) sh_w[threadIdx.x >> 6][64] = gb;
    }
    __syncthreads();
    if (threadIdx.x < 64) sh_dw[threadIdx.x] = ((sh_w[0][threadIdx.x] + sh_w[1][threadIdx.x]) + sh_w[2][threadIdx.x]) + sh_w[3][threadIdx.x];
    if (threadIdx.x == 64) sh_db = ((sh_w[0][64] + sh_w[1][64]) + sh_w[2][64]) + sh_w[3][64];
    __syncthreads();
    if (part) {   // deterministic form: this block's 65 sums as plain stores, added in block order by aux_bwd_reduce_kernel
        if (threadIdx.x < 64) part[(long)threadIdx.x * gridDim.x + blockIdx.x] = sh_dw[threadIdx.x];   // [65][blocks]: a column is contiguous
        if (threadIdx.x == 64) part[64L * gridDim.x + blockIdx.x] = sh_db;
        return;
    }
    if (threadIdx.x < 64) atomicAdd(&dw[threadIdx.x], sh_dw[threadIdx.x]);
    if (threadIdx.x == 0) atomicAdd(dbias, sh_db);
}

// dw[0..63], dbias = sums over the blocks of part [65][blocks]: one workgroup per column, every lane a strided share of the (contiguous)
// column, the lanes' sums folded in a fixed order
__global__ __launch_bounds__(256) void aux_bwd_reduce_kernel(const float* __restrict__ part, int blocks, float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float sh[256];
    const int c = blockIdx.x;
    const float* col = part + (long)c * blocks;
    float s = 0.f;
    for (int r = threadIdx.x; r < blocks; r += 256) s += col[r];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (c < 64) dw[c] = sh[0]; else *dbias = sh[0]; }
}

// ---------------------------------------------------------------------------------------------
// depth head: AvgPool2d(2) x2 -> InstanceNorm2d(1, affine, eps 1e-5) -> Flatten
// reference: models/naive.py:233-240.  depth is [B][1][H][W] fp32; one block per image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ feat, float* __restrict__ xhat, int H, int W) {
    extern __shared__ float pooled[];  // Ho*Wo
    __shared__ double red[2][4];
    const int Ho = H / 4, Wo = W / 4, n = Ho * Wo;
    const float* img = depth + (long)blockIdx.x * H * W;
    double s = 0.0, q = 0.0;
    for (int o = threadIdx.x; o < n; o += blockDim.x) {
        const int oh = o / Wo, ow = o - oh * Wo;
        // avg of the four 2x2 averages (same association as two AvgPool2d(2) passes)
        float acc = 0.f;
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 2; ++c) {
                const float* p = img + (long)(oh * 4 + a * 2) * W + ow * 4 + c * 2;
                acc += (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
            }
        acc *= 0.25f;
        pooled[o] = acc;
        s += acc;
        q += (double)acc * acc;
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float wv = w[0], bv = b[0];
    for (int o = threadIdx.x; o < n; o += blockDim.x) {
        const float xh = (pooled[o] - (float)mean) * invstd;
        xhat[(long)blockIdx.x * n + o] = xh;
        feat[(long)blockIdx.x * n + o] = xh * wv + bv;
    }
}

// ---------------------------------------------------------------------------------------------
// The same two heads on ANY hooked feature map (models/naive.py:196-240: feature_layer_nums other than the scripts' (9,)):
// conv1's raw output, bn1, layer1..layer3 outputs -- C = 64..1024 channels, H x W = 112^2..14^2.  Not a hot path (no reference
// script hooks anything but bn1): one lane group per output pixel walking the channel chunks, a DENSE feature gradient.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void auxc_fwd_kernel(const T* __restrict__ x, int C, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ depth_feat, float* __restrict__ out, long ld_out,
                                                      float* __restrict__ raw, unsigned char* __restrict__ idx, int B, int H, int W) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpp = C / CE, LPP = cpp < 64 ? cpp : 64;     // chunks per pixel; lanes per output pixel (a power of two)
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo;
    const int sub = threadIdx.x % LPP, groups = blockDim.x / LPP;
    const float bv = bias[0];
    for (long o = (long)blockIdx.x * groups + threadIdx.x / LPP; o < total; o += (long)gridDim.x * groups) {
        long t = o;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < 4; ++k) {
            const int ih = oh * 2 + (k >> 1), iw = ow * 2 + (k & 1);
            const T* px = x + (((long)b * H + ih) * W + iw) * C;
            float d = 0.f;
            for (int ch = sub; ch < cpp; ch += LPP) {
                float v[CE];
                chunk_to_f<T>(*(const u32x4*)(px + ch * CE), v);
#pragma unroll
                for (int e = 0; e < CE; ++e) d += v[e] * w[ch * CE + e];
            }
            for (int s = 1; s < LPP; s <<= 1) d += __shfl_xor(d, s);
            d += bv;
            if (d > best || d != d) { best = d; bi = k; }
        }
        if (sub == 0) {
            const long pos = (long)oh * Wo + ow;
            const float df = depth_feat ? depth_feat[(long)b * Ho * Wo + pos] : 1.f;
            out[(long)b * ld_out + pos] = best * df;
            raw[(long)b * Ho * Wo + pos] = best;
            idx[(long)b * Ho * Wo + pos] = (unsigned char)bi;
        }
    }
}

// d_x must be zero on entry (pixels outside the 2x2 windows and the three losers of every window keep a zero gradient)
template <typename T>
__global__ __launch_bounds__(256) void auxc_bwd_kernel(const float* __restrict__ dout, long ld_dout, const T* __restrict__ x, int C,
                                                      const float* __restrict__ w, const float* __restrict__ depth_feat,
                                                      const float* __restrict__ raw, const unsigned char* __restrict__ idx,
                                                      T* __restrict__ d_x, float* __restrict__ dw, float* __restrict__ dbias,
                                                      float* __restrict__ d_depth_feat, int B, int H, int W) {
    constexpr int CE = Elem<T>::kChunk;
    extern __shared__ float sh_dwc[];   // [C] + 1
    for (int i = threadIdx.x; i <= C; i += blockDim.x) sh_dwc[i] = 0.f;
    __syncthreads();
    const int cpp = C / CE, LPP = cpp < 64 ? cpp : 64;
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo;
    const int sub = threadIdx.x % LPP, groups = blockDim.x / LPP;
    float gb = 0.f;
    for (long o = (long)blockIdx.x * groups + threadIdx.x / LPP; o < total; o += (long)gridDim.x * groups) {
        long t = o;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const long pos = (long)oh * Wo + ow;
        const long flat = (long)b * Ho * Wo + pos;
        float d = dout[(long)b * ld_dout + pos];
        if (depth_feat) {
            if (sub == 0 && d_depth_feat) d_depth_feat[flat] = d * raw[flat];
            d *= depth_feat[flat];
        }
        const int bi = idx[flat];
        const long pix = (((long)b * H + oh * 2 + (bi >> 1)) * W + ow * 2 + (bi & 1)) * C;
        for (int ch = sub; ch < cpp; ch += LPP) {
            float v[CE], g[CE];
            chunk_to_f<T>(*(const u32x4*)(x + pix + ch * CE), v);
#pragma unroll
            for (int e = 0; e < CE; ++e) { g[e] = d * w[ch * CE + e]; atomicAdd(&sh_dwc[ch * CE + e], d * v[e]); }
            *(u32x4*)(d_x + pix + ch * CE) = f_to_chunk<T>(g);
        }
        if (sub == 0) gb += d;
    }
    if (sub == 0) atomicAdd(&sh_dwc[C], gb);
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(&dw[i], sh_dwc[i]);
    if (threadIdx.x == 0) atomicAdd(dbias, sh_dwc[C]);
}

// AvgPool2d(2) x `pools` (each flooring odd sizes, as torch does) -> InstanceNorm2d(1, affine) -> Flatten: an output pixel is the mean
// of the 2^pools x 2^pools window at (oh, ow) * 2^pools.  One block per image, one wave per output pixel at a time.
__global__ __launch_bounds__(256) void depth_pools_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ w, const float* __restrict__ b,
                                                             float* __restrict__ feat, float* __restrict__ xhat, int H, int W, int pools) {
    extern __shared__ float pooled[];  // Ho*Wo
    __shared__ double red[2];
    const int win = 1 << pools, Ho = H >> pools, Wo = W >> pools, n = Ho * Wo;
    const float* img = depth + (long)blockIdx.x * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.f / (float)(win * win);
    for (int o = wave; o < n; o += 4) {
        const int oh = o / Wo, ow = o - oh * Wo;
        float acc = 0.f;
        for (int i = lane; i < win * win; i += 64) {
            const int r = i >> pools, c = i & (win - 1);
            acc += img[(long)(oh * win + r) * W + ow * win + c];
        }
        acc = wave_sum(acc);
        if (lane == 0) pooled[o] = acc * inv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // n <= 3136 values: one thread sums them in double, in order
        double s = 0.0, q = 0.0;
        for (int o = 0; o < n; ++o) { s += pooled[o]; q += (double)pooled[o] * pooled[o]; }
        red[0] = s; red[1] = q;
    }
    __syncthreads();
    const double mean = red[0] / n;
    double var = red[1] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float wv = w[0], bv = b[0];
    for (int o = threadIdx.x; o < n; o += blockDim.x) {
        const float xh = (pooled[o] - (float)mean) * invstd;
        xhat[(long)blockIdx.x * n + o] = xh;
        feat[(long)blockIdx.x * n + o] = xh * wv + bv;
    }
}

// dst += src over 16-byte chunks (a hooked feature's gradient joins the trunk's own gradient of that tensor)
template <typename T>
__global__ __launch_bounds__(256) void tensor_add_kernel(T* __restrict__ dst, const T* __restrict__ src, long chunks) {
    constexpr int CE = Elem<T>::kChunk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long)gridDim.x * blockDim.x) {
        float a[CE], b[CE];
        chunk_to_f<T>(*(const u32x4*)(dst + i * CE), a);
        chunk_to_f<T>(*(const u32x4*)(src + i * CE), b);
#pragma unroll
        for (int e = 0; e < CE; ++e) a[e] += b[e];
        *(u32x4*)(dst + i * CE) = f_to_chunk<T>(a);
    }
}

// dw += sum d*xhat ; db += sum d   (the depth image itself needs no gradient)
__global__ __launch_bounds__(256) void depth_bwd_kernel(const float* __restrict__ d_feat, const float* __restrict__ xhat, long n, float* dw, float* db) {
    __shared__ float red[2][4];
    float s = 0.f, q = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = d_feat[i];
        s += d * xhat[i];
        q += d;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dw, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(db, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// ---------------------------------------------------------------------------------------------
// LSTM cell (torch gate order i, f, g, o).  gates: [N][4H] pre-activations WITHOUT biases
// (x W_ih^T + h W_hh^T); overwritten with the activated gates, which the backward re-reads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                           const float* __restrict__ c_prev, float* __restrict__ c_out, float* __restrict__ h_out,
                                                           int N, int Hd) {
    const long total = (long)N * Hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Hd), j = (int)(i - (long)n * Hd);
        float* g = gates + (long)n * 4 * Hd;
        const float gi = sigmoidf_(g[j] + b_ih[j] + b_hh[j]);
        const float gf = sigmoidf_(g[Hd + j] + b_ih[Hd + j] + b_hh[Hd + j]);
        const float gg = tanhf(g[2 * Hd + j] + b_ih[2 * Hd + j] + b_hh[2 * Hd + j]);
        const float go = sigmoidf_(g[3 * Hd + j] + b_ih[3 * Hd + j] + b_hh[3 * Hd + j]);
        const float c = gf * (c_prev ? c_prev[i] : 0.f) + gi * gg;
        g[j] = gi; g[Hd + j] = gf; g[2 * Hd + j] = gg; g[3 * Hd + j] = go;
        c_out[i] = c;
        h_out[i] = go * tanhf(c);
    }
}

// dh: total gradient wrt h_t (output grad + recurrent); dc_io: in = dL/dc_t from step t+1, out = dL/dc_{t-1}
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gates_act, const float* __restrict__ c_prev,
                                                           const float* __restrict__ c_cur, const float* __restrict__ dh,
                                                           float* __restrict__ dc_io, float* __restrict__ dgates, int N, int Hd) {
    const long total = (long)N * Hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Hd), j = (int)(i - (long)n * Hd);
        const float* g = gates_act + (long)n * 4 * Hd;
        const float gi = g[j], gf = g[Hd + j], gg = g[2 * Hd + j], go = g[3 * Hd + j];
        const float tc = tanhf(c_cur[i]);
        const float dhv = dh[i];
        const float dc = dc_io[i] + dhv * go * (1.f - tc * tc);
        const float cp = c_prev ? c_prev[i] : 0.f;
        float* dg = dgates + (long)n * 4 * Hd;
        dg[j] = dc * gg * gi * (1.f - gi);
        dg[Hd + j] = dc * cp * gf * (1.f - gf);
        dg[2 * Hd + j] = dc * gi * (1.f - gg * gg);
        dg[3 * Hd + j] = dhv * tc * go * (1.f - go);
        dc_io[i] = dc * gf;
    }
}

// ---------------------------------------------------------------------------------------------
// PoseDistanceLoss forward + gradient + validation metrics in one launch (single block).
// reference: models/losses.py:47-128.  metric: 0 l2, 1 l1, 2 linf, 3 combined; mode: 0 position, 1 pose.
// out[0] = loss, out[1] = sum_i sqrt(|dp_i|^2 + eps) (val position error), out[2] = sum_i |angle_i| (val orientation error)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_loss_kernel(const float* __restrict__ pred, const float* __restrict__ truth, long n, int metric,
                                                       int mode, float scale, float alpha, float eps, float* __restrict__ out,
                                                       float* __restrict__ grad) {
    __shared__ double red[3][4];
    double l_acc = 0.0, p_acc = 0.0, a_acc = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float* p = pred + i * 7;
        const float* t = truth + i * 7;
        float d[3], ad[3], g[7];
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[k] = p[k] - t[k]; ad[k] = fabsf(d[k]); sq += d[k] * d[k]; g[k] = 0.f; }
        const float l2 = sqrtf(sq + eps);
        float pos = 0.f;
        if (metric == 0 || metric == 3) {
            pos += l2;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += d[k] / l2;
        }
        if (metric == 1 || metric == 3) {
            pos += ad[0] + ad[1] + ad[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += (d[k] > 0.f) ? 1.f : ((d[k] < 0.f) ? -1.f : 0.f);
        }
        if (metric == 2 || metric == 3) {
            int am = 0;  // torch.max(dim) gradient goes to the first maximal index
            if (ad[1] > ad[am]) am = 1;
            if (ad[2] > ad[am]) am = 2;
            pos += ad[am];
            g[am] += (d[am] > 0.f) ? 1.f : ((d[am] < 0.f) ? -1.f : 0.f);
        }
        // quaternion part: qhat = q / |q| (no eps: an all-zero quaternion yields NaN, as in the reference)
        const float q0 = p[3], q1 = p[4], q2 = p[5], q3 = p[6];
        const float mag = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        const float h0 = q0 / mag, h1 = q1 / mag, h2 = q2 / mag, h3 = q3 / mag;
        const float ip = h0 * t[3] + h1 * t[4] + h2 * t[5] + h3 * t[6];
        float ori = 0.f;
        float gh[4] = {0.f, 0.f, 0.f, 0.f};  // dL/d qhat
        if (mode == 1) {
            ori = (1.f - ip * ip) + fmaxf(-h3, 0.f);
            const float c = -2.f * ip;
            gh[0] = c * t[3]; gh[1] = c * t[4]; gh[2] = c * t[5]; gh[3] = c * t[6];
            if (-h3 >= 0.f) gh[3] -= 1.f;  // torch.clamp(min=0) passes the gradient at the boundary
        }
        // d qhat / d q = (I - qhat qhat^T) / |q|
        // (mode 0, "position": the normalised quaternion is not part of the reference's graph -- models/losses.py:124-126 --
        // so its gradient is exactly zero, also for an all-zero predicted quaternion where the Jacobian below is 0/0)
        g[3] = g[4] = g[5] = g[6] = 0.f;
        if (mode == 1) {
            const float hd = gh[0] * h0 + gh[1] * h1 + gh[2] * h2 + gh[3] * h3;
            g[3] = (gh[0] - h0 * hd) / mag; g[4] = (gh[1] - h1 * hd) / mag; g[5] = (gh[2] - h2 * hd) / mag; g[6] = (gh[3] - h3 * hd) / mag;
        }
        if (grad) {
#pragma unroll
            for (int k = 0; k < 3; ++k) grad[i * 7 + k] = scale * g[k];
#pragma unroll
            for (int k = 3; k < 7; ++k) grad[i * 7 + k] = scale * alpha * g[k];
        }
        l_acc += (double)pos + (double)alpha * (double)ori;
        p_acc += (double)l2;
        // validation angle: w of qhat * truth^-1 (xyzw), clipped; 2 acos(w) wrapped to [-pi, pi]; |.|
        const float tt = t[3] * t[3] + t[4] * t[4] + t[5] * t[5] + t[6] * t[6];
        float w = ip / tt;
        w = fminf(fmaxf(w, -1.f), 1.f);
        double ang = 0.0;
        if (sqrt(1.0 - (double)w * (double)w) != 0.0) ang = 2.0 * acos((double)w);
        if (ang > 3.14159265358979323846) ang -= 2.0 * 3.14159265358979323846;
        a_acc += fabs(ang);
    }
    l_acc = wave_sum_d(l_acc); p_acc = wave_sum_d(p_acc); a_acc = wave_sum_d(a_acc);
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; red[0][wv] = l_acc; red[1][wv] = p_acc; red[2][wv] = a_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = scale * (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        out[1] = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        out[2] = (float)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Adam over flat fp32 buffers (torch.optim.Adam defaults: no amsgrad, no weight decay)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float omb1, float b2, float omb2, float eps, float bc1, float bc2_sqrt) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = mm[k] + (gg[k] - mm[k]) * omb1;
            vv[k] = vv[k] * b2 + gg[k] * gg[k] * omb2;
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] -= (lr / bc1) * (mm[k] / denom);
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = v[i] * b2 + gi * gi * omb2;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

// ---------------------------------------------------------------------------------------------
// Dynamic loss scaling for the fp16 compute path (config C5).  All state lives in one device array so that a train step
// never synchronises the host:  st[0] scale, st[1] 1/scale, st[2] found_inf (set by the unscale pass), st[3] skip (this
// step's decision, read by the Adam kernel), st[4] consecutive finite steps, st[5] optimizer steps taken.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amp_unscale_kernel(float* __restrict__ g, long n, float* __restrict__ st) {
    const float inv = st[1];
    bool bad = false;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((f32x4*)g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] *= inv; bad |= !(fabsf(v[k]) <= 3.4028234e38f); }   // inf or NaN
        ((f32x4*)g)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float v = g[i] * inv;
        g[i] = v;
        bad |= !(fabsf(v) <= 3.4028234e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) st[2] = 1.f;   // same value from every writer: a plain store is enough
}

__global__ void amp_update_kernel(float* __restrict__ st, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st[2] != 0.f) {
        st[0] = fmaxf(st[0] * backoff, 1.f); st[1] = 1.f / st[0];
        st[3] = 1.f; st[4] = 0.f;
    } else {
        st[3] = 0.f; st[5] += 1.f;
        const float t = st[4] + 1.f;
        if (t >= (float)interval) { st[0] = fminf(st[0] * growth, 16777216.f); st[1] = 1.f / st[0]; st[4] = 0.f; }
        else st[4] = t;
    }
    st[2] = 0.f;
}

// Adam whose step count and skip decision live on the device (same update rule as adam_kernel)
__global__ __launch_bounds__(256) void adam_amp_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                      long n, float lr, double b1, double b2, float eps, const float* __restrict__ st) {
    if (st[3] != 0.f) return;   // non-finite gradients this step: parameters and moments stay as they are
    const double step = (double)st[5];
    const float bc1 = (float)(1.0 - pow(b1, step)), bc2_sqrt = (float)sqrt(1.0 - pow(b2, step));
    const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), fb2 = (float)b2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = v[i] * fb2 + gi * gi * omb2;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing / small utilities
// ---------------------------------------------------------------------------------------------
// w: [Co][R][S][Ci] fp32 (channels_last storage of an OIHW parameter)
//   -> fwd  [Co][R*S*Ci]   (T, only when T != float: the fp32 master already has this layout)
//   -> dgrad [Ci][R][S][Co] (T)
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Co, int RS, int Ci) {
    const long total = (long)Co * RS * Ci;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const long t = i / Ci;
        const int rs = (int)(t % RS), co = (int)(t / RS);
        const float v = w[i];
        if (wf) wf[i] = Elem<T>::from_f(v);
        if (wd) wd[((long)ci * RS + rs) * Co + co] = Elem<T>::from_f(v);
    }
}

template <typename T> __device__ inline void store4(T* o, float a, float b, float c, float d);
template <> __device__ inline void store4<float>(float* o, float a, float b, float c, float d) { *(f32x4*)o = f32x4{a, b, c, d}; }
template <> __device__ inline void store4<bf16>(bf16* o, float a, float b, float c, float d) {
    u32x2 v; v.x = pack_bf16x2(a, b); v.y = pack_bf16x2(c, d);
    *(u32x2*)o = v;
}
template <> __device__ inline void store4<f16>(f16* o, float a, float b, float c, float d) {
    u32x2 v; v.x = pack_f16x2(a, b); v.y = pack_f16x2(c, d);
    *(u32x2*)o = v;
}

// all conv layers of a trunk in ONE launch: a table of per-layer descriptors, flat element index space.
// Layers whose Co and Ci are multiples of 64 (every trunk conv) go 64x64 tile by tile through LDS, so both the source read
// and the two destination writes (forward layout = source order; data-gradient layout = [Ci][RS][Co], a transpose) move
// whole 128/256-byte rows; other layers fall back to element-wise scatter.
constexpr int kPackMaxLayers = 255;
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_multi_kernel(const rpe_pack_desc* __restrict__ tab, int nlayers, long total) {
    __shared__ long starts[kPackMaxLayers + 1];   // (ResNet-152 has 154 packed convs)
    __shared__ float tile[64][65];
    for (int i = threadIdx.x; i <= nlayers && i <= kPackMaxLayers; i += blockDim.x) starts[i] = i < nlayers ? tab[i].start : total;
    __syncthreads();
    const long chunk = 4096;   // one 64x64 tile; layer starts are multiples of it whenever Co, Ci are multiples of 64
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
    for (long base = (long)blockIdx.x * chunk; base < total; base += (long)gridDim.x * chunk) {
        int l = 0;
        while (l + 1 < nlayers && starts[l + 1] <= base) ++l;
        const rpe_pack_desc d = tab[l];
        const bool tiled = (d.Co % 64 == 0) && (d.Ci % 64 == 0) && (d.start % chunk == 0) && starts[l + 1] >= base + chunk;
        if (tiled) {
            const long t = (base - d.start) / chunk;
            const int cit = d.Ci / 64;
            const int ci0 = (int)(t % cit) * 64;
            const int rs = (int)((t / cit) % d.RS);
            const int co0 = (int)(t / ((long)cit * d.RS)) * 64;
            __syncthreads();   // previous tile fully consumed
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int co = tr + 16 * it;
                const long j = ((long)(co0 + co) * d.RS + rs) * d.Ci + ci0 + tc;
                f32x4 v = *(const f32x4*)(d.src + j);
                tile[co][tc] = v.x; tile[co][tc + 1] = v.y; tile[co][tc + 2] = v.z; tile[co][tc + 3] = v.w;
                if (d.wf) {
                    if (d.scale) { const float sc = d.scale[co0 + co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }   // tile[] keeps the unscaled value for wd
                    store4<T>((T*)d.wf + j, v.x, v.y, v.z, v.w);
                }
            }
            __syncthreads();
            if (d.wd) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ci = tr + 16 * it;
                    store4<T>((T*)d.wd + ((long)(ci0 + ci) * d.RS + rs) * d.Co + co0 + tc, tile[tc][ci], tile[tc + 1][ci], tile[tc + 2][ci],
                              tile[tc + 3][ci]);
                }
            }
            continue;
        }
        for (long i = base + threadIdx.x; i < base + chunk && i < total; i += blockDim.x) {
            int ll = l;
            while (ll + 1 < nlayers && starts[ll + 1] <= i) ++ll;
            const rpe_pack_desc e = tab[ll];
            const long j = i - e.start;
            const int ci = (int)(j % e.Ci);
            const long t = j / e.Ci;
            const int rs = (int)(t % e.RS), co = (int)(t / e.RS);
            const float v = e.src[j];
            if (e.wf) ((T*)e.wf)[j] = Elem<T>::from_f(e.scale ? v * e.scale[co] : v);
            if (e.wd) ((T*)e.wd)[((long)ci * e.RS + rs) * e.Co + co] = Elem<T>::from_f(v);
        }
    }
}

// stem: OIHW [64][3][7][7] fp32 -> [64][8][8][4] T (taps and channel zero padded)
template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 256) return;
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) & 7, co = i >> 8;
    float v = 0.f;
    if (c < 3 && s < 7 && r < 7) v = w[((co * 3 + c) * 7 + r) * 7 + s];
    if (scale) v *= scale[co];
    out[i] = Elem<T>::from_f(v);
}
__global__ void unpack_stem_grad_kernel(const float* __restrict__ d, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 3 * 49) return;
    const int s = i % 7, r = (i / 7) % 7, c = (i / 49) % 3, co = i / 147;
    dw[i] = d[((co * 8 + r) * 8 + s) * 4 + c];
}

// out[c][r] = in[r][c]  (fp32, out leading dimension ldo >= rows, zero padded)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ldi, int ldo) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = by + k, c = bx + tx;
        tile[k][tx] = (r < rows && c < cols) ? in[(long)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = bx + k, r = by + tx;
        if (c < cols && r < ldo) out[(long)c * ldo + r] = (r < rows) ? tile[tx][k] : 0.f;
    }
}

// dy *= (out > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dx[i] = out[i] > 0.f ? dy[i] : 0.f;
}

// db[c] (+)= sum_r x[r][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long rows, int cols, int ld, float* __restrict__ out, int accumulate) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < cols)
        for (long r = rl; r < rows; r += 8) s += x[r * ld + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        for (int i = 1; i < 8; ++i) s += sh[i][cl];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// dst[r][off + c] = src[r][c]  (fp32 strided copy used to assemble the fused feature rows)
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}

static inline int ew_grid(long n, int per_block = 256) {
    long g = (n + per_block - 1) / per_block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace rpe

using namespace rpe;

extern "C" {

int rpe_aux_head_fwd(int dtype, const void* a1, const float* w, const float* bias, const float* depth_feat, float* out, long ld_out,
                     float* raw, unsigned char* idx, int B, int H, int W, void* stream) {
    if ((H | W) & 1) return rpe_set_error(RPE_ERR_SHAPE, "aux_head: H and W must be even");
    const long total = (long)B * (H / 2) * (W / 2);
    if (dtype == RPE_F32) hipLaunchKernelGGL((aux_fwd_kernel<float>), dim3(ew_grid(total, 16)), dim3(256), 0, (hipStream_t)stream, (const float*)a1, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((aux_fwd_kernel<bf16>), dim3(ew_grid(total, 32)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a1, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((aux_fwd_kernel<f16>), dim3(ew_grid(total, 32)), dim3(256), 0, (hipStream_t)stream, (const f16*)a1, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else return rpe_set_error(RPE_ERR_DTYPE, "aux_head: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

static int aux_bwd_launch(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                          const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, float* part,
                          long part_floats, void* stream, const float* bn_scale = nullptr, const float* bn_shift = nullptr);

long rpe_aux_head_bwd_workspace_floats(int dtype, int B, int H, int W) {
    const long total = (long)B * (H / 2) * (W / 2);
    return (long)ew_grid(total, (dtype == RPE_F32 ? 16 : 32) * 4) * 65;
}

/* deterministic form: per-block partial sums through the workspace, added in a fixed order; dw / dbias are OVERWRITTEN */
int rpe_aux_head_bwd_det(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                         const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, float* workspace,
                         long workspace_floats, void* stream) {
    if (!workspace || workspace_floats < rpe_aux_head_bwd_workspace_floats(dtype, B, H, W))
        return rpe_set_error(RPE_ERR_WORKSPACE, "aux_head_bwd_det: workspace smaller than rpe_aux_head_bwd_workspace_floats()");
    return aux_bwd_launch(dtype, dout, ld_dout, a1, w, depth_feat, raw, idx, d_a1, dw, dbias, d_depth_feat, B, H, W, workspace, workspace_floats, stream);
}

int rpe_aux_head_bwd(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                     const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, void* stream) {
    return aux_bwd_launch(dtype, dout, ld_dout, a1, w, depth_feat, raw, idx, d_a1, dw, dbias, d_depth_feat, B, H, W, nullptr, 0, stream);
}

/* the same from the RAW stem conv output y and bn1's scale / shift: the activated tensor a1 was never written (rpe_bn_apply_maxpool3x3s2_aux
 * with a = NULL); compact form only (the stem backward gathers the a1 gradient itself: rpe_stem_bwd) */
int rpe_aux_head_bwd_det_y(int dtype, const float* dout, long ld_dout, const void* y, const float* bn_scale, const float* bn_shift, const float* w,
                           const float* depth_feat, const float* raw, const unsigned char* idx, float* dw, float* dbias, float* d_depth_feat, int B, int H,
                           int W, float* workspace, long workspace_floats, void* stream) {
    if (!y || !bn_scale || !bn_shift) return rpe_set_error(RPE_ERR_SHAPE, "aux_head_bwd_det_y: y, scale and shift are required");
    if (!workspace || workspace_floats < rpe_aux_head_bwd_workspace_floats(dtype, B, H, W))
        return rpe_set_error(RPE_ERR_WORKSPACE, "aux_head_bwd_det_y: workspace smaller than rpe_aux_head_bwd_workspace_floats()");
    return aux_bwd_launch(dtype, dout, ld_dout, y, w, depth_feat, raw, idx, nullptr, dw, dbias, d_depth_feat, B, H, W, workspace, workspace_floats, stream, bn_scale, bn_shift);
}

static int aux_bwd_launch(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                          const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, float* part,
                          long part_floats, void* stream, const float* bn_scale, const float* bn_shift) {
    const long total = (long)B * (H / 2) * (W / 2);
    // the gather is latency-bound (winner index -> winner pixel): many short blocks when their sums leave as plain stores (part);
    // with global atomics every block ends in 65 adds on the same 65 addresses, which is what then bounds the launch: fewer blocks
    int g = ew_grid(total, (dtype == RPE_F32 ? 16 : 32) * 4);
    if (!part && g > 1024) g = 1024;
    if (dtype == RPE_F32) hipLaunchKernelGGL((aux_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, (const float*)a1, w, depth_feat, raw, idx, (float*)d_a1, dw, dbias, d_depth_feat, B, H, W, part, bn_scale, bn_shift);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((aux_bwd_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, (const bf16*)a1, w, depth_feat, raw, idx, (bf16*)d_a1, dw, dbias, d_depth_feat, B, H, W, part, bn_scale, bn_shift);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((aux_bwd_kernel<f16>), dim3(g), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, (const f16*)a1, w, depth_feat, raw, idx, (f16*)d_a1, dw, dbias, d_depth_feat, B, H, W, part, bn_scale, bn_shift);
    else return rpe_set_error(RPE_ERR_DTYPE, "aux_head: unsupported dtype");
    RPE_CHECK_LAUNCH();
    if (part) {
        hipLaunchKernelGGL(aux_bwd_reduce_kernel, dim3(65), dim3(256), 0, (hipStream_t)stream, (const float*)part, g, dw, dbias);
        RPE_CHECK_LAUNCH();
    }
    return 0;
}

int rpe_depth_head_fwd(const float* depth, const float* w, const float* b, float* feat, float* xhat, int B, int H, int W, void* stream) {
    if ((H | W) & 3) return rpe_set_error(RPE_ERR_SHAPE, "depth_head: H and W must be multiples of 4");
    const size_t sh = (size_t)(H / 4) * (W / 4) * sizeof(float);
    hipLaunchKernelGGL(depth_fwd_kernel, dim3(B), dim3(256), sh, (hipStream_t)stream, depth, w, b, feat, xhat, H, W);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_tensor_add(int dtype, void* dst, const void* src, long n, void* stream) {
    const int ce = dtype == RPE_F32 ? 4 : 8;
    if (n <= 0 || (n % ce) || !dst || !src) return rpe_set_error(RPE_ERR_SHAPE, "tensor_add: n must be a positive multiple of the 16-byte chunk");
    const int g = ew_grid(n / ce);
    if (dtype == RPE_F32) hipLaunchKernelGGL((tensor_add_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (float*)dst, (const float*)src, n / ce);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((tensor_add_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (bf16*)dst, (const bf16*)src, n / ce);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((tensor_add_kernel<f16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (f16*)dst, (const f16*)src, n / ce);
    else return rpe_set_error(RPE_ERR_DTYPE, "tensor_add: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

static int hook_shape_ok(int dtype, int C, int H, int W) {
    const int ce = dtype == RPE_F32 ? 4 : 8;
    if (C < ce || (C % ce) || C > 1024 || H < 2 || W < 2) return 0;
    const int cpp = C / ce;
    return (cpp & (cpp - 1)) == 0;   // lanes per pixel = min(64, chunks per pixel) must be a power of two
}

int rpe_aux_head_fwd_c(int dtype, const void* x, int C, const float* w, const float* bias, const float* depth_feat, float* out, long ld_out,
                       float* raw, unsigned char* idx, int B, int H, int W, void* stream) {
    if (!hook_shape_ok(dtype, C, H, W)) return rpe_set_error(RPE_ERR_SHAPE, "aux_head_fwd_c: C must be 16-byte chunks, a power of two of them, <= 1024");
    const long total = (long)B * (H / 2) * (W / 2);
    const int g = ew_grid(total, 4);
    if (dtype == RPE_F32) hipLaunchKernelGGL((auxc_fwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, C, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((auxc_fwd_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, C, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((auxc_fwd_kernel<f16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const f16*)x, C, w, bias, depth_feat, out, ld_out, raw, idx, B, H, W);
    else return rpe_set_error(RPE_ERR_DTYPE, "aux_head_fwd_c: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_aux_head_bwd_c(int dtype, const float* dout, long ld_dout, const void* x, int C, const float* w, const float* depth_feat, const float* raw,
                       const unsigned char* idx, void* d_x, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, void* stream) {
    if (!hook_shape_ok(dtype, C, H, W) || !d_x) return rpe_set_error(RPE_ERR_SHAPE, "aux_head_bwd_c: bad shape or null gradient tensor");
    const long total = (long)B * (H / 2) * (W / 2);
    const size_t bytes = (size_t)B * H * W * C * (dtype == RPE_F32 ? 4 : 2);
    if (hipError_t he = hipMemsetAsync(d_x, 0, bytes, (hipStream_t)stream)) return rpe_set_error_hip(he, __FILE__, __LINE__);
    const int g = ew_grid(total, 4 * 8);
    const size_t sh = (size_t)(C + 1) * sizeof(float);
    if (dtype == RPE_F32) hipLaunchKernelGGL((auxc_bwd_kernel<float>), dim3(g), dim3(256), sh, (hipStream_t)stream, dout, ld_dout, (const float*)x, C, w, depth_feat, raw, idx, (float*)d_x, dw, dbias, d_depth_feat, B, H, W);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((auxc_bwd_kernel<bf16>), dim3(g), dim3(256), sh, (hipStream_t)stream, dout, ld_dout, (const bf16*)x, C, w, depth_feat, raw, idx, (bf16*)d_x, dw, dbias, d_depth_feat, B, H, W);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((auxc_bwd_kernel<f16>), dim3(g), dim3(256), sh, (hipStream_t)stream, dout, ld_dout, (const f16*)x, C, w, depth_feat, raw, idx, (f16*)d_x, dw, dbias, d_depth_feat, B, H, W);
    else return rpe_set_error(RPE_ERR_DTYPE, "aux_head_bwd_c: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_depth_head_fwd_pools(const float* depth, const float* w, const float* b, float* feat, float* xhat, int B, int H, int W, int pools, void* stream) {
    if (pools < 1 || pools > 6 || (H >> pools) < 1 || (W >> pools) < 1) return rpe_set_error(RPE_ERR_SHAPE, "depth_head_fwd_pools: 1..6 pooling steps, non-empty output");
    const size_t sh = (size_t)(H >> pools) * (W >> pools) * sizeof(float);
    hipLaunchKernelGGL(depth_pools_fwd_kernel, dim3(B), dim3(256), sh, (hipStream_t)stream, depth, w, b, feat, xhat, H, W, pools);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_depth_head_bwd(const float* d_feat, const float* xhat, long n, float* dw, float* db, void* stream) {
    hipLaunchKernelGGL(depth_bwd_kernel, dim3(ew_grid(n, 256 * 16)), dim3(256), 0, (hipStream_t)stream, d_feat, xhat, n, dw, db);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_lstm_cell_fwd(float* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* c_out, float* h_out, int N, int Hd,
                      void* stream) {
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_grid((long)N * Hd)), dim3(256), 0, (hipStream_t)stream, gates, b_ih, b_hh, c_prev, c_out, h_out, N, Hd);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c_cur, const float* dh, float* dc_io, float* dgates, int N,
                      int Hd, void* stream) {
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_grid((long)N * Hd)), dim3(256), 0, (hipStream_t)stream, gates_act, c_prev, c_cur, dh, dc_io, dgates, N, Hd);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_pose_loss(const float* pred, const float* truth, long n, int metric, int mode, float scale, float alpha, float eps, float* out3,
                  float* grad, void* stream) {
    if (metric < 0 || metric > 3 || mode < 0 || mode > 1) return rpe_set_error(RPE_ERR_SHAPE, "pose_loss: invalid metric/mode");
    if (n <= 0) return rpe_set_error(RPE_ERR_SHAPE, "pose_loss: empty batch");
    hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, truth, n, metric, mode, scale, alpha, eps, out3, grad);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps, int step, void* stream) {
    if (n <= 0) return 0;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "adam: buffers must be 16-byte aligned");
    // scalars in double, as torch.optim.Adam computes them on the host (1 - 0.999f != 0.001f)
    const double b1d = beta1, b2d = beta2;
    const float bc1 = (float)(1.0 - pow(b1d, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow(b2d, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)(1.0 - b1d),
                       (float)beta2, (float)(1.0 - b2d), (float)eps, bc1, bc2s);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_amp_unscale(float* grads, long n, float* state, void* stream) {
    if (n <= 0) return 0;
    if (!grads || !state || (((uintptr_t)grads) & 15)) return rpe_set_error(RPE_ERR_ALIGN, "amp_unscale: gradient buffer must be 16-byte aligned");
    hipLaunchKernelGGL(amp_unscale_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, grads, n, state);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_amp_update(float* state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    if (!state || growth_interval <= 0) return rpe_set_error(RPE_ERR_SHAPE, "amp_update: bad arguments");
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, growth_factor, backoff_factor, growth_interval);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_adam_step_amp(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps, const float* state,
                      void* stream) {
    if (n <= 0) return 0;
    if (!state) return rpe_set_error(RPE_ERR_SHAPE, "adam_step_amp: null state");
    hipLaunchKernelGGL(adam_amp_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, beta1, beta2, (float)eps, state);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_pack_conv_weight(int dtype, const float* w_krsc, void* w_fwd, void* w_dgrad, int Co, int R, int S, int Ci, void* stream) {
    const long n = (long)Co * R * S * Ci;
    if (dtype == RPE_F32) hipLaunchKernelGGL((pack_conv_weight_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w_krsc, (float*)w_fwd, (float*)w_dgrad, Co, R * S, Ci);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((pack_conv_weight_kernel<bf16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w_krsc, (bf16*)w_fwd, (bf16*)w_dgrad, Co, R * S, Ci);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((pack_conv_weight_kernel<f16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w_krsc, (f16*)w_fwd, (f16*)w_dgrad, Co, R * S, Ci);
    else return rpe_set_error(RPE_ERR_DTYPE, "pack_conv_weight: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_pack_conv_weights_multi(int dtype, const rpe_pack_desc* table_dev, int nlayers, long total, void* stream) {
    note_kernel("pack_conv_weights_multi_kernel");
    if (nlayers <= 0 || nlayers > kPackMaxLayers || total <= 0) return rpe_set_error(RPE_ERR_SHAPE, "pack_conv_weights_multi: bad table (1..255 layers)");
    const int grid = (int)((total + 4095) / 4096 < 8192 ? (total + 4095) / 4096 : 8192);
    if (dtype == RPE_F32) hipLaunchKernelGGL((pack_conv_weights_multi_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table_dev, nlayers, total);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((pack_conv_weights_multi_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table_dev, nlayers, total);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((pack_conv_weights_multi_kernel<f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table_dev, nlayers, total);
    else return rpe_set_error(RPE_ERR_DTYPE, "pack_conv_weights_multi: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_pack_stem_weight(int dtype, const float* w_oihw, const float* scale, void* out, void* stream) {
    note_kernel("pack_stem_weight_kernel");
    if (dtype == RPE_F32) hipLaunchKernelGGL((pack_stem_weight_kernel<float>), dim3(64), dim3(256), 0, (hipStream_t)stream, w_oihw, scale, (float*)out);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((pack_stem_weight_kernel<bf16>), dim3(64), dim3(256), 0, (hipStream_t)stream, w_oihw, scale, (bf16*)out);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((pack_stem_weight_kernel<f16>), dim3(64), dim3(256), 0, (hipStream_t)stream, w_oihw, scale, (f16*)out);
    else return rpe_set_error(RPE_ERR_DTYPE, "pack_stem_weight: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_unpack_stem_grad(const float* d_packed, float* dw_oihw, void* stream) {
    hipLaunchKernelGGL(unpack_stem_grad_kernel, dim3((64 * 147 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_packed, dw_oihw);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_transpose_f32(const float* in, float* out, int rows, int cols, int ldi, int ldo, void* stream) {
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (ldo + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols, ldi, ldo);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_relu_bwd(const float* out, const float* dy, float* dx, long n, void* stream) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, out, dy, dx, n);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_colsum(const float* x, long rows, int cols, int ld, float* out, int accumulate, void* stream) {
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, out, accumulate);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_copy2d(const float* src, int ld_src, float* dst, int ld_dst, long rows, int cols, void* stream) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, cols);
    RPE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

// BatchNorm (train / eval), ReLU, residual add, pooling, layout staging: HBM-bound NHWC kernels.
// All tensors are [rows][C] with C contiguous; every thread moves 16-byte chunks.
#include "common.h"
#include <cstdlib>

namespace rpe {

// ---------------------------------------------------------------------------------------------
// BN forward: finalize batch statistics from the conv epilogue's per-tile partial sums
// ---------------------------------------------------------------------------------------------
// part: [tiles][2][C] fp32 partial sums.  One launch, grid = (C/32, NS): slice s sums tiles s, s+NS, ... in double into
// dpart [NS][2][C]; the LAST slice block of a 32-channel column group to arrive (device-scope counter behind the dpart
// slices, left at zero again) sums the NS slices in a fixed order and finishes the per-channel work.  Deterministic, and
// >= 256 blocks stay busy even for C = 64 (one block per 32 channels alone took 1.1 ms on the stem's 25088 tiles); the
// separate finalize launch this replaces cost ~5 us x 106 launches per train step.
struct BnFwdFin {
    double count;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long* num_batches;
    float momentum, eps;
    float *scale, *shift, *save_mean, *save_invstd;
    __device__ void operator()(int c, double s, double q) const {
        // every input is read BEFORE the first output is stored (gamma, beta, the running statistics; without running statistics the two
        // reads go to gamma and are ignored): with load, store, load, store ... (rounds 1-3) each load waited for the store in front of it
        // as well -- loads and stores share one counter on gfx950 and the compiler can only answer vmcnt(0) -- i.e. four dependent round
        // trips at the end of every one of the step's ~50 forward finalize launches (round 4, from the ISA)
        const float g = gamma[c], bt = beta[c];
        const float* rmp = running_mean ? running_mean : gamma;
        const float* rvp = running_var ? running_var : gamma;
        const float rm = rmp[c], rv = rvp[c];
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = bt - (float)mean * sc;
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * rv + momentum * (float)unb;
        }
        if (num_batches && c == 0) *num_batches += 1;
    }
};
// BN backward pass 2: dgamma, dbeta and the two per-channel coefficients of pass 3
struct BnBwdFin {
    double count;
    float *dgamma, *dbeta, *c1, *c2;
    __device__ void operator()(int c, double s, double q) const {
        if (dbeta) dbeta[c] = (float)s;
        if (dgamma) dgamma[c] = (float)q;
        c1[c] = (float)(s / count);
        c2[c] = (float)(q / count);
    }
};

// BN backward coefficients of a bottleneck's bn3 when the raw conv3 output y = a W^T was never written (y3-free block): only sum dz
// comes from the partial rows (their second half is 0); sum dz*y = rowdot(T_c, W_c) with T = dz^T a -- the weight gradient's first
// product, [C][Ci] fp32 -- and W the compute-dtype weight the forward multiplied with, so sum dz*xhat = invstd (sum dz*y - mean sum dz).
// The row dot products are formed by the 8 row lanes of the finishing block (dot / kDot below), summed in lane order.
template <typename T> struct BnBwdFinT {
    static constexpr bool kDot = true;
    double count;
    float *dgamma, *dbeta, *c1, *c2;
    const float* Tm;
    const T* w;
    int Ci;
    const float *mean, *invstd;
    __device__ double dot(int c, int part, int nparts) const {
        // (all of a lane's loads requested before the first is used: the one-at-a-time form was a chain of 8..16 dependent global
        // round trips on the data-gradient chain -- 22 us per launch against 12 for the plain coefficients)
        double r = 0.0;
        for (int k0 = part; k0 < Ci; k0 += 16 * nparts) {
            float tv[16], wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u * nparts;
                tv[u] = k < Ci ? Tm[(long)c * Ci + k] : 0.f;
                wv[u] = k < Ci ? Elem<T>::to_f(w[(long)c * Ci + k]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) r += (double)tv[u] * (double)wv[u];
        }
        return r;
    }
    __device__ void operator()(int c, double s, double d) const {
        const float is_ = invstd[c], mu_ = mean[c];   // (inputs before the first store: see BnFwdFin)
        const double q = (double)is_ * (d - (double)mu_ * s);
        if (dbeta) dbeta[c] = (float)s;
        if (dgamma) dgamma[c] = (float)q;
        c1[c] = (float)(s / count);
        c2[c] = (float)(q / count);
    }
};
template <typename Fin, typename = void> struct FinHasDot { static constexpr bool value = false; };
template <typename Fin> struct FinHasDot<Fin, decltype((void)Fin::kDot)> { static constexpr bool value = true; };

template <typename Fin>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ part, int tiles, int C, int NS, double* dpart,
                                                             unsigned* counters, const Fin fin) {
    __shared__ double sh[2][8][32];
    __shared__ int last;
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        // 4 independent row loads in flight per thread (the loop is latency-, not bandwidth-bound)
        const int st = 8 * NS;
        int t = blockIdx.y + rl * NS;
        // (a functor with the row-dot form ignores the second sum: its half of every partial row -- zeros -- is not even read)
        constexpr bool NOQ = FinHasDot<Fin>::value;
        for (; t + 3 * st < tiles; t += 4 * st) {
            const float a0 = part[((long)t * 2 + 0) * C + c], b0 = NOQ ? 0.f : part[((long)t * 2 + 1) * C + c];
            const float a1 = part[((long)(t + st) * 2 + 0) * C + c], b1 = NOQ ? 0.f : part[((long)(t + st) * 2 + 1) * C + c];
            const float a2 = part[((long)(t + 2 * st) * 2 + 0) * C + c], b2 = NOQ ? 0.f : part[((long)(t + 2 * st) * 2 + 1) * C + c];
            const float a3 = part[((long)(t + 3 * st) * 2 + 0) * C + c], b3 = NOQ ? 0.f : part[((long)(t + 3 * st) * 2 + 1) * C + c];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            q += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; t < tiles; t += st) {
            s += (double)part[((long)t * 2 + 0) * C + c];
            if (!NOQ) q += (double)part[((long)t * 2 + 1) * C + c];
        }
    }
    sh[0][rl][cl] = s;
    sh[1][rl][cl] = q;
    __syncthreads();
    if (rl == 0) {
        for (int i = 1; i < 8; ++i) { s += sh[0][i][cl]; q += sh[1][i][cl]; }
    }
    if (NS > 1) {
        // Hand-off without an agent-scope release/acquire FENCE: on this chip a release fence writes back the whole XCD L2
        // (full of the conv output just written) and cost ~20 us per launch.  What is relied on instead (gfx950 behaviour
        // as measured in MI355X_MICROARCH.md, "Valid forms", table row 1 -- not an architectural guarantee of the HIP
        // memory model, hence tests/test_gpu_ops.py::test_bn_reduce_handoff_under_load):
        //   (1) every byte handed off is written by a relaxed agent-scope atomic store = global_store ... sc1 (write-through to
        //       memory, nothing left dirty in this XCD's L2);
        //   (2) every storing wave drains its stores (s_waitcnt vmcnt(0)) and then reaches the workgroup barrier, BEFORE
        //   (3) ONE lane signals with an agent-scope atomic add on the column group's counter; the workgroup whose add returns
        //       NS-1 is the last arriver, told through LDS behind a barrier;
        //   (4) the last arriver reads every slice with relaxed agent-scope atomic loads = global_load ... sc1, which bypass
        //       the CU's L1 (the only cache that could hold a stale copy: it never caches these lines otherwise) and are served
        //       from L2 / memory, where (1) put the data.
        // No plain load ever touches dpart.
        if (rl == 0 && c < C) {
            __hip_atomic_store(&dpart[((long)blockIdx.y * 2 + 0) * C + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&dpart[((long)blockIdx.y * 2 + 1) * C + c], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(&counters[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = prev == (unsigned)NS - 1;
            if (last) __hip_atomic_store(&counters[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all arrivals are in
        }
        __syncthreads();
        if (!last) return;
        s = 0.0; q = 0.0;
        if (c < C) {
            // NS <= 64 (reduce_slices): at most 8 slices per thread, all requested before the first is added
            double vs[8], vq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // (branch-free requests: a slice index past NS re-reads slice 0 and is zeroed below)
                const int i = rl + 8 * u, ic = i < NS ? i : 0;
                vs[u] = __hip_atomic_load(&dpart[((long)ic * 2 + 0) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vq[u] = __hip_atomic_load(&dpart[((long)ic * 2 + 1) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (rl + 8 * u >= NS) { vs[u] = 0.0; vq[u] = 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += vs[u]; q += vq[u]; }
        }
        sh[0][rl][cl] = s;
        sh[1][rl][cl] = q;
        __syncthreads();
        if (rl == 0) {
            for (int i = 1; i < 8; ++i) { s += sh[0][i][cl]; q += sh[1][i][cl]; }
        }
    }
    if constexpr (FinHasDot<Fin>::value) {
        // row dot products: 8 CONSECUTIVE lanes share a channel and walk its row together (coalesced 32-byte runs; with the channel on the
        // fast lane index every load touched 32 lines), their partial sums meet by xor shuffles in a fixed order
        const int dc = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
        double d = dc < C ? fin.dot(dc, part, 8) : 0.0;
        d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
        __syncthreads();   // (sh was read by the row-0 lanes above)
        if (part == 0) sh[0][0][threadIdx.x >> 3] = d;
        __syncthreads();
        if (rl == 0 && c < C) fin(c, s, sh[0][0][cl]);
        return;
    }
    if (rl == 0 && c < C) fin(c, s, q);
}

// Batch statistics of y = x W^T (a 1x1 conv) WITHOUT y: mean_c = w_c . colsum(x) / M,  E[y_c^2] = w_c^T (x^T x) w_c / M.  gram: fp32
// [rows >= Ci + 1][Ci] with x^T x in rows [0, Ci) and colsum(x) in row `ones_row` (rpe_gram); W is the compute-dtype copy the conv
// multiplies with.  One block per output channel; the quadratic form is centred (S - s1 s1^T / M) and summed in double, the
// threads' partial sums meet in a fixed order.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void bn_gram_stats_kernel(const T* __restrict__ w, int Ci, const float* __restrict__ gram, int ones_row, int vec, const BnFwdFin fin) {
    __shared__ __attribute__((aligned(16))) float ws[1024];
    __shared__ __attribute__((aligned(16))) float s1s[1024];
    __shared__ double red[2][NT / 64];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* s1 = gram + (long)ones_row * Ci;
    for (int i = threadIdx.x; i < Ci; i += NT) { ws[i] = Elem<T>::to_f(w[(long)c * Ci + i]); s1s[i] = s1[i]; }
    __syncthreads();
    const double inv_count = 1.0 / fin.count;
    double m = 0.0;
    for (int i = threadIdx.x; i < Ci; i += NT) m += (double)ws[i] * (double)s1s[i];
    // The block sits on the forward's critical path once per y3-free block and is pure load latency (S comes from another XCD's L2 or
    // from memory): the one-row-per-wave walk of round 2 took 23 us for 128 x 128, 256 threads with 8 scalar loads in flight 9 us
    // (64 x 64) / 24 us (128 x 128: eight dependent rounds).  Now 1024 threads with four 16-byte loads each: 128 x 128 is ONE round.
    const int n = Ci * Ci;
    const int ci_shift = (Ci & (Ci - 1)) == 0 ? __builtin_ctz(Ci) : -1;
    double q0 = 0.0, q1 = 0.0;
    if (vec) {   // Ci % 4 == 0, gram 16-byte aligned: element idx = i * Ci + j, four consecutive j per load
        const int n4 = n >> 2;
        for (int base = threadIdx.x; base < n4; base += NT * 4) {
            f32x4 sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int f = base + u * NT; sv[u] = *(const f32x4*)(gram + (long)(f < n4 ? f : 0) * 4); }   // (branch-free: all four requested at once; a chunk past the end is not used below)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = base + u * NT;
                if (f < n4) {
                    const int idx = f * 4;
                    const int i = ci_shift >= 0 ? idx >> ci_shift : idx / Ci, j = idx - i * Ci;
                    const double si = (double)s1s[i] * inv_count;
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t += (double)ws[j + k] * ((double)sv[u][k] - si * (double)s1s[j + k]);
                    t *= (double)ws[i];
                    if (u & 1) q1 += t; else q0 += t;
                }
            }
        }
    } else {
        for (int base = threadIdx.x; base < n; base += NT * 8) {
            float sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int idx = base + u * NT; sv[u] = idx < n ? gram[idx] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NT;
                if (idx < n) {
                    const int i = ci_shift >= 0 ? idx >> ci_shift : idx / Ci, j = idx - i * Ci;
                    const double t = (double)ws[i] * (double)ws[j] * ((double)sv[u] - (double)s1s[i] * (double)s1s[j] * inv_count);
                    if (u & 1) q1 += t; else q0 += t;
                }
            }
        }
    }
    double q = q0 + q1;
    m = wave_sum_d(m);
    q = wave_sum_d(q);
    if (lane == 0) { red[0][wave] = m; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sm = 0.0, sq = 0.0;
        for (int i = 0; i < NT / 64; ++i) { sm += red[0][i]; sq += red[1][i]; }   // wave order
        // fin expects (sum y, sum y^2): sum y^2 = centred form + (sum y)^2 / M
        fin(c, sm, sq + sm * sm * inv_count);
    }
}

__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---------------------------------------------------------------------------------------------
// BN apply (+ residual) (+ ReLU):  a = relu(y*scale + shift + res)
// ---------------------------------------------------------------------------------------------
// streaming-kernel load/store helpers: NT = non-temporal (the operand is not read again before it would be evicted anyway)
template <bool NT> __device__ inline u32x4 ld16(const void* p) {
    if (NT) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}
template <bool NT> __device__ inline void st16(void* p, const u32x4& v) {
    if (NT) __builtin_nontemporal_store(v, (u32x4*)p);
    else *(u32x4*)p = v;
}

// Work distribution of the streaming kernels (measured: tools/micro/stream_order.hip, profiles/r03_stream_order.txt).  A block owns ONE
// contiguous span of 256 * SUB * UNR chunks and every thread issues its UNR independent 16-byte loads per operand once: 5.4-6.0 TB/s
// over three 411-MB tensors, against 4.2-4.8 TB/s for the round-2 form (a capped grid whose blocks stride over the whole tensor, i.e.
// consecutive 4-KB pieces dealt round-robin to the 8 XCDs, several passes per thread).  Thread t handles chunks
// base + s*256 + t + u*256*SUB (s < SUB = max(1, cpr/256), u < UNR): the channel chunk (s*256 + t) % cpr does not depend on u, so
// the per-channel coefficients live in registers; the launcher checks that cpr = C/CE is a power of two.
template <typename T, int UNR, bool NT, bool RESBN = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ out,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      long nchunks, int C, int relu, unsigned char* __restrict__ mask,
                                                      const float* __restrict__ res_scale, const float* __restrict__ res_shift, int rev) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const int sub = cpr > 256 ? cpr / 256 : 1;
    const long ustride = 256L * sub, span = ustride * UNR;
    // every XCD walks one contiguous eighth of the spans, upwards or (rev) downwards: the order of the conv kernels' row tiles (common.h)
    const int bx = xcd_remap_dir(blockIdx.x, gridDim.x, rev);
    for (long base = (long)bx * span; base < nchunks; base += (long)gridDim.x * span)
    for (int s = 0; s < sub; ++s) {
    const long i = base + s * 256 + threadIdx.x;
    const int c0 = (int)(i % cpr) * CE;
    // res_scale / res_shift: the residual is itself a raw conv output whose BatchNorm (the projection shortcut's, no ReLU) is
    // applied here on the fly -- its own apply pass and the normalised copy it wrote are gone; the shift folds into sh
    float sc[CE], sh[CE], rs[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e];
        rs[e] = 1.f;
        if (RESBN) { rs[e] = res_scale[c0 + e]; sh[e] += res_shift[c0 + e]; }
    }
    {
        u32x4 vy[UNR], vr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * ustride;
            if (j < nchunks) {
                vy[u] = ld16<NT>(y + j * CE);
                if (res) vr[u] = ld16<NT>(res + j * CE);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * ustride;
            if (j >= nchunks) break;
            float v[CE], r[CE];
            chunk_to_f<T>(vy[u], v);
            if (res) chunk_to_f<T>(vr[u], r);
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float t = fmaf(v[e], sc[e], sh[e]);  // same expression as the fused dgrad epilogue's mask
                if (RESBN) t = fmaf(r[e], rs[e], t);     // (a compile-time variant: the plain residual add keeps its code)
                else if (res) t += r[e];
                v[e] = relu ? fmaxf(t, 0.f) : t;
                bits |= (t > 0.f ? 1u : 0u) << e;
            }
            *(u32x4*)(out + j * CE) = f_to_chunk<T>(v);
            if (CE == 8 && mask) mask[j] = (unsigned char)bits;   // one byte per 8-channel chunk (16-bit element types only)
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// BN backward, pass 1: per-block partial sums of dz and dz*xhat, dz = dA * (a_out > 0)
// ---------------------------------------------------------------------------------------------
// grid = (row blocks, column slabs); slab width SW = min(C, 256*CE); TPR = SW/CE threads per row.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dA, const T* __restrict__ a_out, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           long M, int C, int SW, int rows_per_block, float* __restrict__ part) {
    constexpr int CE = Elem<T>::kChunk;
    __shared__ float sh[2 * 256 * CE];
    const int TPR = SW / CE, RPI = 256 / TPR;
    const int cc = threadIdx.x % TPR, rr = threadIdx.x / TPR;
    const int col = blockIdx.y * SW + cc * CE;
    float mu[CE], is[CE], s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { mu[e] = mean[col + e]; is[e] = invstd[col + e]; s1[e] = 0.f; s2[e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    for (long r = r0 + rr; r < r1; r += RPI) {
        float d[CE], yy[CE], a[CE];
        chunk_to_f<T>(*(const u32x4*)(dA + r * C + col), d);
        chunk_to_f<T>(*(const u32x4*)(y + r * C + col), yy);
        if (a_out) chunk_to_f<T>(*(const u32x4*)(a_out + r * C + col), a);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float dz = (a_out && !(a[e] > 0.f)) ? 0.f : d[e];
            s1[e] += dz;
            s2[e] += dz * (yy[e] - mu[e]) * is[e];
        }
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        sh[(rr * SW + cc * CE + e) * 2 + 0] = s1[e];
        sh[(rr * SW + cc * CE + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SW * 2; i += 256) {
        float t = 0.f;
        for (int k = 0; k < RPI; ++k) t += sh[k * SW * 2 + i];
        const int c = i >> 1, which = i & 1;
        part[((long)blockIdx.x * 2 + which) * C + blockIdx.y * SW + c] = t;
    }
}

// pass 3: dy = gamma*invstd*(dz - c1 - xhat*c2); optionally also emits dz (gradient of the residual branch)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dA, const T* __restrict__ a_out, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ c1,
                                                          const float* __restrict__ c2, T* __restrict__ dy, T* __restrict__ dz_out,
                                                          long nchunks, int C) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cpr) * CE;
        float d[CE], yy[CE], a[CE], o[CE];
        chunk_to_f<T>(*(const u32x4*)(dA + i * CE), d);
        chunk_to_f<T>(*(const u32x4*)(y + i * CE), yy);
        if (a_out) chunk_to_f<T>(*(const u32x4*)(a_out + i * CE), a);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int c = c0 + e;
            const float dz = (a_out && !(a[e] > 0.f)) ? 0.f : d[e];
            const float xh = (yy[e] - mean[c]) * invstd[c];
            o[e] = gamma[c] * invstd[c] * (dz - c1[c] - xh * c2[c]);
            d[e] = dz;
        }
        *(u32x4*)(dy + i * CE) = f_to_chunk<T>(o);
        if (dz_out) *(u32x4*)(dz_out + i * CE) = f_to_chunk<T>(d);
    }
}

// pass 3 when dz is already materialised (fused data-gradient epilogue): dy = gamma*invstd*(dz - c1 - xhat*c2)
template <typename T, int UNR, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_dz_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ c1, const float* __restrict__ c2, T* __restrict__ dy,
                                                             long nchunks, int C, int rev) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    // contiguous span per block, one pass per thread (see bn_apply_kernel)
    const int sub = cpr > 256 ? cpr / 256 : 1;
    const long ustride = 256L * sub, span = ustride * UNR;
    const int bx = xcd_remap_dir(blockIdx.x, gridDim.x, rev);
    for (long base = (long)bx * span; base < nchunks; base += (long)gridDim.x * span)
    for (int s = 0; s < sub; ++s) {
    const long i = base + s * 256 + threadIdx.x;
    const int c0 = (int)(i % cpr) * CE;
    float k0[CE], k1[CE], k2[CE];  // dy = k0*dz + k1 + k2*y
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const float gi = gamma[c0 + e] * invstd[c0 + e];
        k0[e] = gi;
        k2[e] = -gi * invstd[c0 + e] * c2[c0 + e];
        k1[e] = -gi * c1[c0 + e] - k2[e] * mean[c0 + e];
    }
    {
        u32x4 vd[UNR], vy[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * ustride;
            if (j < nchunks) { vd[u] = ld16<NT>(dz + j * CE); vy[u] = ld16<NT>(y + j * CE); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * ustride;
            if (j >= nchunks) break;
            float d[CE], yy[CE];
            chunk_to_f<T>(vd[u], d);
            chunk_to_f<T>(vy[u], yy);
#pragma unroll
            for (int e = 0; e < CE; ++e) d[e] = fmaf(k0[e], d[e], fmaf(k2[e], yy[e], k1[e]));
            *(u32x4*)(dy + j * CE) = f_to_chunk<T>(d);
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// MaxPool 3x3 / stride 2 / pad 1 (ResNet stem).  Forward keeps the winning tap (0..8) per output
// element so the backward is a gather.  Ties: first tap in (kh, kw) scan order, as torch does.
// ---------------------------------------------------------------------------------------------
// the CE winner taps of one chunk in one 4- / 8-byte store (CE single-byte stores were what bounded the pool pass: 3.4 TB/s)
template <int CE> __device__ __forceinline__ void store_winners(unsigned char* dst, const unsigned char* bi) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) lo |= (unsigned)bi[e] << (8 * e);
    if (CE == 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) hi |= (unsigned)bi[4 + e] << (8 * e);
        *(u32x2*)dst = u32x2{lo, hi};
    } else {
        *(unsigned*)dst = lo;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, unsigned char* __restrict__ idx,
                                                         int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long total = (long)B * Ho * Wo * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr);
        long pix = i / cpr;
        const int ow = (int)(pix % Wo); pix /= Wo;
        const int oh = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float best[CE];
        unsigned char bi[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                float v[CE];
                chunk_to_f<T>(*(const u32x4*)(x + (((long)b * H + ih) * W + iw) * C + cc * CE), v);
#pragma unroll
                for (int e = 0; e < CE; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = (unsigned char)(kh * 3 + kw); }
            }
        }
        *(u32x4*)(out + i * CE) = f_to_chunk<T>(best);
        store_winners<CE>(idx + i * CE, bi);
    }
}

// The stem's BatchNorm apply + ReLU and the max pool behind it in ONE pass over the conv output: a = relu(y * scale + shift) is written
// (the aux head and the hooks read it) and pooled on the fly -- the separate pool pass re-read all of a (411 MB at 256 images).
// One thread per pooled pixel and 16-byte channel chunk: it reads the 3 x 3 window of y (5 of the 9 chunks are its neighbours': cache
// hits), writes the four a pixels of its own 2 x 2 block (taps (1..2, 1..2): every input pixel has exactly one owner; H and W even)
// and the pooled chunk with its winner taps.  Values are compared AFTER rounding to T, ties / NaN exactly as maxpool_fwd_kernel on the
// stored a: the results are bitwise those of bn_apply_kernel followed by maxpool_fwd_kernel.
// AUX (round 4): the bn1 aux head -- Conv2d(64 -> 1, 1x1) + MaxPool2d(2) (x the depth feature), models/naive.py:223-231,318-330 -- rides
// along: a thread's own 2 x 2 block of a IS the aux head's pooling window, and the 8 (4 in fp32) threads of a pooled pixel hold all 64
// channels of it, so the 1x1 conv is 8 multiply-adds per pixel and thread plus a shuffle sum over the group; the first thread writes
// the feature, the raw maximum and the winner tap exactly as aux_fwd_kernel does (same products of the ROUNDED a, same tie rule).
// With `a` null the activated tensor is not written at all: nobody else reads it (the stem backward works from y), which saves its
// 411-MB write here and the 411-MB read of the separate aux launch.
struct StemAuxFwd {
    const float *w, *bias, *depth_feat;   // [64], [1], [B][Ho*Wo] or null
    float* out; long ld_out;              // feature columns of the fused feature rows
    float* raw; unsigned char* idx;       // [B][Ho*Wo]: the window maximum before the depth product, its winner tap
};
template <typename T, bool AUX>
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                              T* __restrict__ a, T* __restrict__ out, unsigned char* __restrict__ idx,
                                                              int B, int H, int W, int C, int Ho, int Wo, const StemAuxFwd aux, int rev) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long total = (long)B * Ho * Wo * cpr;
    const long span = (long)gridDim.x * blockDim.x;
    const int bx = xcd_remap_dir(blockIdx.x, gridDim.x, rev);   // (the conv that wrote y dealt its row tiles to the XCDs the same way: common.h)
    // (AUX: every thread of a pooled pixel's group takes part in the shuffles, so the loop bound is rounded up to whole groups -- total is a
    // multiple of cpr, and cpr divides the wave: a group is either all in or all out)
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < total; i += span) {
        const int cc = (int)(i % cpr);
        long pix = i / cpr;
        const int ow = (int)(pix % Wo); pix /= Wo;
        const int oh = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float sc[CE], sh[CE], aw[CE], ad[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < CE; ++e) { sc[e] = scale[cc * CE + e]; sh[e] = shift[cc * CE + e]; aw[e] = AUX ? aux.w[cc * CE + e] : 0.f; }
        // all nine taps are requested before the first is used: a tap outside the image reads the clamped pixel instead (its value is
        // ignored).  With the load under `if (inside)` (rounds 3-4) every tap was a basic block with its own wait -- nine dependent round
        // trips per thread, 2.7 TB/s (from the ISA, round 4)
        u32x4 raw[9];
        bool ok[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ih = oh * 2 - 1 + k / 3, iw = ow * 2 - 1 + k % 3;
            ok[k] = ih >= 0 && ih < H && iw >= 0 && iw < W;
            const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
            raw[k] = *(const u32x4*)(y + (((long)b * H + ihc) * W + iwc) * C + cc * CE);
        }
        float best[CE];
        unsigned char bi[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (!ok[k]) continue;
            float v[CE];
            chunk_to_f<T>(raw[k], v);
#pragma unroll
            for (int e = 0; e < CE; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);   // bn_apply_kernel's expression
            const u32x4 packed = f_to_chunk<T>(v);
            if (k / 3 >= 1 && k % 3 >= 1 && a) {   // this thread's own 2 x 2 block
                const int ih = oh * 2 - 1 + k / 3, iw = ow * 2 - 1 + k % 3;
                *(u32x4*)(a + (((long)b * H + ih) * W + iw) * C + cc * CE) = packed;
            }
            chunk_to_f<T>(packed, v);               // compare what the pool pass would have read back
            if (AUX && k / 3 >= 1 && k % 3 >= 1) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < CE; ++e) d += v[e] * aw[e];
                ad[(k / 3 - 1) * 2 + (k % 3 - 1)] = d;
            }
#pragma unroll
            for (int e = 0; e < CE; ++e)
                if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = (unsigned char)k; }
        }
        *(u32x4*)(out + i * CE) = f_to_chunk<T>(best);
        store_winners<CE>(idx + i * CE, bi);
        if (AUX) {
            const float bv = aux.bias[0];
            float bestd = -INFINITY;
            int bk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d = ad[k];
                for (int o = 1; o < cpr; o <<= 1) d += __shfl_xor(d, o);   // the group's lanes are consecutive and aligned (cpr divides 64)
                d += bv;
                if (d > bestd || d != d) { bestd = d; bk = k; }
            }
            if (cc == 0) {
                const long pos = (long)oh * Wo + ow, flat = (long)b * Ho * Wo + pos;
                const float df = aux.depth_feat ? aux.depth_feat[flat] : 1.f;
                aux.out[(long)b * aux.ld_out + pos] = bestd * df;
                aux.raw[flat] = bestd;
                aux.idx[flat] = (unsigned char)bk;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                         const T* __restrict__ addend, T* __restrict__ dx,
                                                         int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long total = (long)B * H * W * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr);
        long pix = i / cpr;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[CE];
        if (addend) chunk_to_f<T>(*(const u32x4*)(addend + i * CE), acc);
        else {
#pragma unroll
            for (int e = 0; e < CE; ++e) acc[e] = 0.f;
        }
        // windows (oh, ow) with ih = oh*2 - 1 + kh == h
        for (int kh = 0; kh < 3; ++kh) {
            const int t = h + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int u = w + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= Wo) continue;
                const long o = (((long)b * Ho + oh) * Wo + ow) * C + cc * CE;
                float d[CE];
                chunk_to_f<T>(*(const u32x4*)(dout + o), d);
                const unsigned char* ip = idx + o;
#pragma unroll
                for (int e = 0; e < CE; ++e)
                    if (ip[e] == (unsigned char)(kh * 3 + kw)) acc[e] += d[e];
            }
        }
        *(u32x4*)(dx + i * CE) = f_to_chunk<T>(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// Stem backward, fused: gradient of a1 = relu(bn1(y)) gathered on the fly from its two consumers -- the 3x3/2 max pool
// (winner taps) and, optionally, the aux head (1x1 conv 64->1 + 2x2 max pool: winner pixel of each window gets g*w[c]) --
// then the ReLU mask (recomputed from y) and BatchNorm backward.  Two passes over y (reduce, apply) instead of
// maxpool_bwd + aux scatter + bn_bwd_reduce + bn_bwd_apply over three 411 MB tensors (dA, d_a1, a1) that no longer exist.
// ---------------------------------------------------------------------------------------------
struct StemAux {
    const float* dout;            // [B][ld] gradient of the aux feature columns, or null (no aux head)
    long ld;
    const float* depth_feat;      // [B][Ho2*Wo2] or null: aux feature was multiplied by it
    const unsigned char* idx;     // [B][Ho2*Wo2] winner (0..3) of each 2x2 window
    const float* w;               // [64] aux conv weight
};

// The 2 x 2 block of a1 pixels (2 oh + r, 2 ow + c), r, c in {0, 1}, CE channels from c0 -- the pixels pooled window (oh, ow) owns (its taps
// (1..2, 1..2); every a1 pixel has exactly one owner: H, W even).  Their pool gradient comes from four windows: the own one, and for
// the block's right column / bottom row the neighbours (oh, ow + 1) [tap column 0], (oh + 1, ow) [tap row 0], (oh + 1, ow + 1) [tap 0]:
//   (0,0): tap 4 of W00                      (0,1): tap 5 of W00 + tap 3 of W01
//   (1,0): tap 7 of W00 + tap 1 of W10       (1,1): tap 8 of W00 + tap 6 of W01 + tap 2 of W10 + tap 0 of W11
// so a thread loads four (gradient chunk, winner word) pairs and four y chunks for four output chunks -- the per-PIXEL gather of
// rounds 1-2 (1..4 windows each: 34 loads per four chunks, 1.7 TB/s) was bound by its loads, not by HBM.  The aux head's 2 x 2 max pool
// has the same blocks: one winner per thread.  dz[r * 2 + c][e]; yy = the four y chunks.
template <typename T>
__device__ __forceinline__ void stem_dz_block(const T* __restrict__ dpool, const unsigned char* __restrict__ pidx, const StemAux& ax, const float* axw,
                                              const float* sc, const float* sh, const float (*yy)[Elem<T>::kChunk], int b, int oh, int ow, int c0,
                                              int Ho, int Wo, float (*dz)[Elem<T>::kChunk]) {
    constexpr int CE = Elem<T>::kChunk;
    const long o00 = (((long)b * Ho + oh) * Wo + ow) * 64 + c0;
    const bool right = ow + 1 < Wo, below = oh + 1 < Ho;
    float d[4][CE];
    unsigned long long taps[4];
    // the four windows and the aux head's three scalars are requested together, a window past the edge reads the own window instead (and
    // is then zeroed): with the loads under `if (inside)` / `if (aux)` each was a basic block with a wait of its own (round 4, from the ISA)
    const long wo[4] = {o00, right ? o00 + 64 : o00, below ? o00 + (long)Wo * 64 : o00, (right && below) ? o00 + (long)Wo * 64 + 64 : o00};
    const bool wok[4] = {true, right, below, right && below};
    u32x4 wraw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wraw[k] = *(const u32x4*)(dpool + wo[k]);
        taps[k] = CE == 8 ? *(const unsigned long long*)(pidx + wo[k]) : (unsigned long long)*(const unsigned*)(pidx + wo[k]);
    }
    const long apos = (long)oh * Wo + ow, aflat = (long)b * Ho * Wo + apos;   // (the aux head pools the same 2 x 2 blocks: Ho = H / 2)
    // (no aux head / no depth feature: the reads go to the pooled gradient and its winner bytes, always there, and are ignored)
    const unsigned char* awp = ax.dout ? ax.idx + aflat : pidx;
    const float* agp = ax.dout ? ax.dout + (long)b * ax.ld + apos : (const float*)dpool;
    const float* adp = (ax.dout && ax.depth_feat) ? ax.depth_feat + aflat : (const float*)dpool;
    const int awin = (int)*awp;
    const float ag = *agp, adf = *adp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        chunk_to_f<T>(wraw[k], d[k]);
        if (!wok[k]) {
#pragma unroll
            for (int e = 0; e < CE; ++e) d[k][e] = 0.f;
            taps[k] = ~0ull;   // (no tap is 255)
        }
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const unsigned t0 = (unsigned)(taps[0] >> (8 * e)) & 0xffu, t1 = (unsigned)(taps[1] >> (8 * e)) & 0xffu;
        const unsigned t2 = (unsigned)(taps[2] >> (8 * e)) & 0xffu, t3 = (unsigned)(taps[3] >> (8 * e)) & 0xffu;
        dz[0][e] = t0 == 4u ? d[0][e] : 0.f;
        dz[1][e] = (t0 == 5u ? d[0][e] : 0.f) + (t1 == 3u ? d[1][e] : 0.f);
        dz[2][e] = (t0 == 7u ? d[0][e] : 0.f) + (t2 == 1u ? d[2][e] : 0.f);
        dz[3][e] = (t0 == 8u ? d[0][e] : 0.f) + (t1 == 6u ? d[1][e] : 0.f) + (t2 == 2u ? d[2][e] : 0.f) + (t3 == 0u ? d[3][e] : 0.f);
    }
    if (ax.dout) {
        const int win = awin;
        float g = ag;
        if (ax.depth_feat) g *= adf;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < CE; ++e) dz[k][e] += win == k ? g * axw[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < CE; ++e)
            if (!(fmaf(yy[k][e], sc[e], sh[e]) > 0.f)) dz[k][e] = 0.f;   // the forward's expression (bn_apply_kernel)
}

// pass 1: block partial sums of dz and dz*xhat -> part [gridDim.x][2][64].  A block owns `rows_per_block` consecutive POOLED rows
// (b, oh) = two image rows each (a contiguous span of y); thread = (pooled pixel lane, 16-byte channel chunk).
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(const T* __restrict__ dpool, const unsigned char* __restrict__ pidx, const StemAux ax,
                                                             const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, int B, int H, int W,
                                                             int Ho, int Wo, int rows_per_block, float* __restrict__ part) {
    constexpr int CE = Elem<T>::kChunk, CPR = 64 / CE, PPB = 256 / CPR;   // chunks per pixel, pooled pixels per block pass
    __shared__ float sh[PPB][64][2];
    const int cc = threadIdx.x % CPR, pl = threadIdx.x / CPR, c0 = cc * CE;
    float sc[CE], sf[CE], mu[CE], is[CE], s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = scale[c0 + e]; sf[e] = shift[c0 + e]; mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; s1[e] = 0.f; s2[e] = 0.f; }
    float axw[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) axw[e] = ax.dout ? ax.w[c0 + e] : 0.f;
    const int row_end = min(B * Ho, ((int)blockIdx.x + 1) * rows_per_block);
    for (int row = blockIdx.x * rows_per_block; row < row_end; ++row) {
        const int b = row / Ho, oh = row - b * Ho;
        for (int ow = pl; ow < Wo; ow += PPB) {
            const long p00 = (((long)b * H + 2 * oh) * W + 2 * ow) * 64 + c0;
            float yy[4][CE], dz[4][CE];
            chunk_to_f<T>(*(const u32x4*)(y + p00), yy[0]);
            chunk_to_f<T>(*(const u32x4*)(y + p00 + 64), yy[1]);
            chunk_to_f<T>(*(const u32x4*)(y + p00 + (long)W * 64), yy[2]);
            chunk_to_f<T>(*(const u32x4*)(y + p00 + (long)W * 64 + 64), yy[3]);
            stem_dz_block<T>(dpool, pidx, ax, axw, sc, sf, yy, b, oh, ow, c0, Ho, Wo, dz);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < CE; ++e) { s1[e] += dz[k][e]; s2[e] += dz[k][e] * (yy[k][e] - mu[e]) * is[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) { sh[pl][c0 + e][0] = s1[e]; sh[pl][c0 + e][1] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x >> 1, which = threadIdx.x & 1;
        float t = 0.f;
        for (int k = 0; k < PPB; ++k) t += sh[k][c][which];
        part[((long)blockIdx.x * 2 + which) * 64 + c] = t;
    }
}

// pass 2: dy = gamma*invstd*(dz - c1 - xhat*c2)
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_apply_kernel(const T* __restrict__ dpool, const unsigned char* __restrict__ pidx, const StemAux ax,
                                                            const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ c1, const float* __restrict__ c2,
                                                            T* __restrict__ dy, int B, int H, int W, int Ho, int Wo, int rows_per_block) {
    constexpr int CE = Elem<T>::kChunk, CPR = 64 / CE, PPB = 256 / CPR;
    const int cc = threadIdx.x % CPR, pl = threadIdx.x / CPR, c0 = cc * CE;
    float sc[CE], sf[CE], k0[CE], k1[CE], k2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        sc[e] = scale[c0 + e]; sf[e] = shift[c0 + e];
        const float gi = gamma[c0 + e] * invstd[c0 + e];
        k0[e] = gi;
        k2[e] = -gi * invstd[c0 + e] * c2[c0 + e];
        k1[e] = -gi * c1[c0 + e] - k2[e] * mean[c0 + e];
    }
    float axw[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) axw[e] = ax.dout ? ax.w[c0 + e] : 0.f;
    const int row_end = min(B * Ho, ((int)blockIdx.x + 1) * rows_per_block);
    for (int row = blockIdx.x * rows_per_block; row < row_end; ++row) {
        const int b = row / Ho, oh = row - b * Ho;
        for (int ow = pl; ow < Wo; ow += PPB) {
            const long p00 = (((long)b * H + 2 * oh) * W + 2 * ow) * 64 + c0;
            const long off[4] = {p00, p00 + 64, p00 + (long)W * 64, p00 + (long)W * 64 + 64};
            float yy[4][CE], dz[4][CE];
#pragma unroll
            for (int k = 0; k < 4; ++k) chunk_to_f<T>(*(const u32x4*)(y + off[k]), yy[k]);
            stem_dz_block<T>(dpool, pidx, ax, axw, sc, sf, yy, b, oh, ow, c0, Ho, Wo, dz);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int e = 0; e < CE; ++e) dz[k][e] = fmaf(k0[e], dz[k][e], fmaf(k2[e], yy[k][e], k1[e]));
                *(u32x4*)(dy + off[k]) = f_to_chunk<T>(dz[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Global average pool [B][HW][C] (T) -> [B][C] (f32), and its backward
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int HW, int C) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * cpr) return;
    const int cc = (int)(i % cpr), b = (int)(i / cpr);
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
        float v[CE];
        chunk_to_f<T>(*(const u32x4*)(x + ((long)b * HW + p) * C + cc * CE), v);
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] += v[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < CE; ++e) out[(long)b * C + cc * CE + e] = acc[e] * inv;
}

// few images (rollout frames): 8 lanes share one 16-byte channel chunk and walk the pixels 8 apart (one image alone gave ONE
// workgroup walking 49 pixels serially: 15 us); the lanes' sums meet in a fixed butterfly order
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_few_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int HW, int C) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool ok = i < (long)B * cpr;
    const int cc = ok ? (int)(i % cpr) : 0, b = ok ? (int)(i / cpr) : 0;
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    for (int p = sub; p < HW; p += 8) {
        float v[CE];
        chunk_to_f<T>(*(const u32x4*)(x + ((long)b * HW + p) * C + cc * CE), v);
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        acc[e] += __shfl_xor(acc[e], 1, 64);
        acc[e] += __shfl_xor(acc[e], 2, 64);
        acc[e] += __shfl_xor(acc[e], 4, 64);
    }
    if (ok && sub == 0) {
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < CE; ++e) out[(long)b * C + cc * CE + e] = acc[e] * inv;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, int B, int HW, int C) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long total = (long)B * HW * cpr;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr);
        const int b = (int)(i / ((long)cpr * HW));
        float v[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = dout[(long)b * C + cc * CE + e] * inv;
        *(u32x4*)(dx + i * CE) = f_to_chunk<T>(v);
    }
}

// ---------------------------------------------------------------------------------------------
// Device staging: NCHW fp32 image batch -> NHWC4 (channel 3 = 0) in the compute type
// ---------------------------------------------------------------------------------------------
// The staged image is ZERO-BORDERED (include/rpe_hip.h, RPE_STEM_PAD): [B][H + 6][W + 6][4], the image in the middle.  The staging
// kernels write the interior only; the border is zeroed once by whoever owns the buffer (the engine at bind time).
__device__ __forceinline__ long x4_index(long b, int h, int w, int H, int W) {
    return ((b * (H + 2 * RPE_STEM_PAD) + h + RPE_STEM_PAD) * (long)(W + 2 * RPE_STEM_PAD) + w + RPE_STEM_PAD) * 4;
}
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W) {
    const int HW = H * W;
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
        const float* src = img + b * 3 * HW + p;
        const float r = src[0], g = src[HW], bl = src[2 * (long)HW];
        const int h = (int)(p / W), w = (int)(p - (long)h * W);
        T* dst = out + x4_index(b, h, w, H, W);
        if (sizeof(T) == 4) {
            *(f32x4*)dst = f32x4{r, g, bl, 0.f};
        } else {
            u32x2 t; t.x = pack2<T>(r, g); t.y = pack2<T>(bl, 0.f);
            *(u32x2*)dst = t;
        }
    }
}

// Simulator frames: uint8 [B][Hs][Ws][3] -> centre crop (H, W) -> (x/255 - mean)/std -> NHWC4 compute type.
// replaces ToPILImage -> Resize(256) -> CenterCrop(224) -> ToTensor -> Normalize (util/data_utils.py:48-54) for frames whose
// shorter side already is the resize target (robosuite's 256x256 default), and the .cuda() copy of the fp32 result.
template <typename T>
__global__ __launch_bounds__(256) void frames_u8_to_nhwc4_kernel(const unsigned char* __restrict__ fr, T* __restrict__ out, int B, int Hs, int Ws,
                                                                int H, int W, float m0, float m1, float m2, float i0, float i1, float i2) {
    const int top = (Hs - H) / 2, left = (Ws - W) / 2;
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long t = i / W;
        const int h = (int)(t % H);
        const long b = t / H;
        const unsigned char* p = fr + ((b * Hs + (top + h)) * Ws + (left + w)) * 3;
        const float r = ((float)p[0] * (1.f / 255.f) - m0) * i0, g = ((float)p[1] * (1.f / 255.f) - m1) * i1, bl = ((float)p[2] * (1.f / 255.f) - m2) * i2;
        T* dst = out + x4_index(b, h, w, H, W);
        if (sizeof(T) == 4) {
            *(f32x4*)dst = f32x4{r, g, bl, 0.f};
        } else {
            u32x2 v; v.x = pack2<T>(r, g); v.y = pack2<T>(bl, 0.f);
            *(u32x2*)dst = v;
        }
    }
}

// Frames whose shorter side is not the resize target: Pillow's antialiased 8-bit bilinear resample (what Resize(256) on a PIL image
// runs, util/data_utils.py:48-54) in its own arithmetic -- per output pixel a window of taps with 22-bit fixed-point weights
// (tables built on the host in double precision, util/data_utils.py pil_bilinear_tables), accumulated from 1 << 21 and shifted
// back, horizontally into an 8-bit intermediate, then vertically -- so the result is bit-identical to the CPU transform.
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const unsigned char* __restrict__ fr, unsigned char* __restrict__ tmp, int B, int Hs, int Ws, int Wr,
                                                         const int* __restrict__ xb, const int* __restrict__ xk, int ksx) {
    const long total = (long)B * Hs * Wr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wr);
        const long row = i / Wr;   // b * Hs + y
        const int x0 = xb[2 * xo], n = xb[2 * xo + 1];
        const unsigned char* p = fr + (row * Ws + x0) * 3;
        const int* k = xk + (long)xo * ksx;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int t = 0; t < n; ++t) { const int kv = k[t]; s0 += p[3 * t] * kv; s1 += p[3 * t + 1] * kv; s2 += p[3 * t + 2] * kv; }
        unsigned char* o = tmp + i * 3;
        o[0] = (unsigned char)min(max(s0 >> 22, 0), 255); o[1] = (unsigned char)min(max(s1 >> 22, 0), 255); o[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
    }
}
// vertical pass + centre crop (top, left inside the resized Hr x Wr image) + (x/255 - mean)/std -> NHWC4 compute type
template <typename T>
__global__ __launch_bounds__(256) void resize_v_crop_norm_kernel(const unsigned char* __restrict__ tmp, T* __restrict__ out, int B, int Hs, int Wr, int top, int left,
                                                                int H, int W, const int* __restrict__ yb, const int* __restrict__ yk, int ksy, int vertical,
                                                                float m0, float m1, float m2, float i0, float i1, float i2) {
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long t2 = i / W;
        const int h = (int)(t2 % H);
        const long b = t2 / H;
        const int yo = top + h, xo = left + w;
        int c0, c1, c2;
        if (vertical) {
            const int y0 = yb[2 * yo], n = yb[2 * yo + 1];
            const int* k = yk + (long)yo * ksy;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int t = 0; t < n; ++t) {
                const unsigned char* p = tmp + ((b * Hs + y0 + t) * Wr + xo) * 3;
                const int kv = k[t];
                s0 += p[0] * kv; s1 += p[1] * kv; s2 += p[2] * kv;
            }
            c0 = min(max(s0 >> 22, 0), 255); c1 = min(max(s1 >> 22, 0), 255); c2 = min(max(s2 >> 22, 0), 255);
        } else {
            const unsigned char* p = tmp + ((b * Hs + yo) * Wr + xo) * 3;
            c0 = p[0]; c1 = p[1]; c2 = p[2];
        }
        const float r = ((float)c0 * (1.f / 255.f) - m0) * i0, g = ((float)c1 * (1.f / 255.f) - m1) * i1, bl = ((float)c2 * (1.f / 255.f) - m2) * i2;
        T* dst = out + x4_index(b, h, w, H, W);
        if (sizeof(T) == 4) {
            *(f32x4*)dst = f32x4{r, g, bl, 0.f};
        } else {
            u32x2 v; v.x = pack2<T>(r, g); v.y = pack2<T>(bl, 0.f);
            *(u32x2*)dst = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline int ew_grid(long n, int per_block = 256) {
    long g = (n + per_block - 1) / per_block;
    // (round 2 capped this at 8192 blocks that stride over the tensor: 4.7 TB/s where one-pass blocks stream at 5.4-6.0,
    // tools/micro/stream_order.hip)
    if (g > (1L << 20)) g = 1L << 20;
    if (g < 1) g = 1;
    return (int)g;
}

// streaming BN kernels: unroll / grid cap / non-temporal loads (RPE_EW_UNR, RPE_EW_GRID, RPE_EW_NT for experiments)
struct EwCfg { int unr; long cap; bool nt; };
static inline EwCfg ew_cfg() {
    static const EwCfg c = [] {
        // non-temporal operand loads: the streamed tensors are read once here (their next reader is a different kernel, >= their own
        // size of traffic later) and stop evicting the weights / partial sums the neighbouring GEMM launches keep in L2.  Measured
        // on one box, three alternations: 20.84 vs 21.17 ms/step (round 1, before the folds: level).  RPE_EW_NT=0 restores cached loads.
        EwCfg v{4, 1L << 20, true};
        if (const char* e = getenv("RPE_EW_UNR")) v.unr = atoi(e);
        if (const char* e = getenv("RPE_EW_GRID")) v.cap = atol(e);
        if (const char* e = getenv("RPE_EW_NT")) v.nt = atoi(e) != 0;
        // (measured and rejected in round 3, profiles/r03_ab_stream_ew.txt: capping the blocks per CU with unused LDS, which gains 8 % in
        // tools/micro/epilogue_stream.hip, and 8 chunks per thread -- level or slower inside the step)
        return v;
    }();
    return c;
}
// grid for `n` 16-byte chunks of rows with `cpr` chunks each (a power of two): one block per contiguous span of
// 256 * max(1, cpr/256) * unr chunks (bn_apply_kernel), capped at cfg.cap blocks (the kernels loop over further spans)
static inline long ew_grid_rows(long n, int cpr, const EwCfg& cfg) {
    const long span = 256L * (cpr > 256 ? cpr / 256 : 1) * cfg.unr;
    long g = (n + span - 1) / span;
    if (g > cfg.cap) g = cfg.cap;
    if (g < 1) g = 1;
    return g;
}
static inline bool ew_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// slices for the staged partial reduction: enough blocks to cover the chip, >= 8 tiles per slice
static inline int reduce_slices(int tiles, int C) {
    const int colblocks = (C + 31) / 32;
    int ns = (512 + colblocks - 1) / colblocks;
    if (ns > 64) ns = 64;
    if (ns > tiles / 8) ns = tiles / 8;
    if (ns > RPE_BN_MAX_SLICES) ns = RPE_BN_MAX_SLICES;
    if (ns < 1) ns = 1;
    return ns;
}
template <typename Fin>
static int reduce_finalize(const float* part, int tiles, int C, double* dpart, const Fin& fin, hipStream_t s) {
    if (C > 32 * 128) return rpe_set_error(RPE_ERR_SHAPE, "bn: more than 4096 channels");
    // (a single-stage form for the small reductions -- one 1024-thread block owning 8..32 columns outright, no hand-off -- measured
    // slower in the step: 21.42 vs 21.34 ms)
    const int ns = reduce_slices(tiles, C);
    unsigned* counters = (unsigned*)dpart;   // RPE_BN_DPART_DOUBLES(C): 64 doubles of counters (128 column groups), then the slices
    hipLaunchKernelGGL((reduce_finalize_kernel<Fin>), dim3((C + 31) / 32, ns), dim3(256), 0, s, part, tiles, C, ns, dpart + 64, counters, fin);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int bn_apply_launch(const void* y, const void* res, void* out, const float* scale, const float* shift, long M, int C, int relu, unsigned char* mask,
                    hipStream_t s, const float* res_scale = nullptr, const float* res_shift = nullptr) {
    constexpr int CE = Elem<T>::kChunk;
    if (C % CE) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply: C must be a multiple of the 16-byte chunk");
    const long n = M * C / CE;
    const int cpr = C / CE;
    EwCfg cfg = ew_cfg();
    // the unrolled forms keep the per-channel coefficients in registers across the unroll, which needs the unroll stride (256 chunks or
    // a multiple) to be a multiple of the chunks per row: a power of two.  Any other channel count (e.g. C = 192) takes the form without
    // unroll, whose coefficients are re-read per 256-chunk piece.
    if (!ew_pow2(cpr)) cfg.unr = 1;
    const long g = ew_grid_rows(n, cpr, cfg);
    if (mask && sizeof(T) != 2) return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_mask: the packed ReLU mask is written for 16-bit element types only");
#define RPE_BN_APPLY(U, N, R) hipLaunchKernelGGL((bn_apply_kernel<T, U, N, R>), dim3((unsigned)g), dim3(256), 0, s, (const T*)y, (const T*)res, (T*)out, scale, shift, n, C, relu, mask, res_scale, res_shift, walk_take())
    if (res_scale) {   // the residual under its own BatchNorm (rpe_bn_apply_res_bn): one configuration
        if (!res || !res_shift) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply: the residual and its shift are required with res_scale");
        if (cfg.unr == 1) { if (cfg.nt) RPE_BN_APPLY(1, true, true); else RPE_BN_APPLY(1, false, true); }
        else if (cfg.nt) RPE_BN_APPLY(4, true, true); else RPE_BN_APPLY(4, false, true);
    }
    else if (cfg.nt) { if (cfg.unr == 1) RPE_BN_APPLY(1, true, false); else if (cfg.unr == 2) RPE_BN_APPLY(2, true, false); else RPE_BN_APPLY(4, true, false); }
    else { if (cfg.unr == 1) RPE_BN_APPLY(1, false, false); else if (cfg.unr == 2) RPE_BN_APPLY(2, false, false); else RPE_BN_APPLY(4, false, false); }
#undef RPE_BN_APPLY
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int bn_apply_dz_launch(const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma, const float* c1,
                              const float* c2, void* dy, long M, int C, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    const int cpr = C / CE;
    const long n = M * C / CE;
    EwCfg cfg = ew_cfg();
    if (!ew_pow2(cpr)) cfg.unr = 1;   // (see bn_apply_launch)
    const long g = ew_grid_rows(n, cpr, cfg);
#define RPE_BN_DZ(U, N) hipLaunchKernelGGL((bn_bwd_apply_dz_kernel<T, U, N>), dim3((unsigned)g), dim3(256), 0, s, (const T*)dz, (const T*)y, mean, invstd, gamma, c1, c2, (T*)dy, n, C, walk_take())
    if (cfg.nt) { if (cfg.unr == 1) RPE_BN_DZ(1, true); else if (cfg.unr == 2) RPE_BN_DZ(2, true); else RPE_BN_DZ(4, true); }
    else { if (cfg.unr == 1) RPE_BN_DZ(1, false); else if (cfg.unr == 2) RPE_BN_DZ(2, false); else RPE_BN_DZ(4, false); }
#undef RPE_BN_DZ
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int bn_bwd_launch(const void* dA, const void* a_out, const void* y, const float* mean, const float* invstd, const float* gamma,
                  float* dgamma, float* dbeta, void* dy, void* dz_out, long M, int C, float* part, long part_floats, float* c1c2,
                  double* dpart, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    if (C % CE) return rpe_set_error(RPE_ERR_SHAPE, "bn_bwd: C must be a multiple of the 16-byte chunk");
    int SW = C < 256 * CE ? C : 256 * CE;
    if (256 % (SW / CE)) return rpe_set_error(RPE_ERR_SHAPE, "bn_bwd: C/chunk must divide 256");
    const int slabs = C / SW;
    const int RPI = 256 / (SW / CE);
    long nb = (M + (long)RPI * 8 - 1) / ((long)RPI * 8);  // >= 8 iterations per block
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    long rpb = (M + nb - 1) / nb;
    nb = (M + rpb - 1) / rpb;
    if (nb * 2 * C > part_floats) return rpe_set_error(RPE_ERR_WORKSPACE, "bn_bwd: partial-sum workspace too small");
    note_kernel("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3((unsigned)nb, slabs), dim3(256), 0, s, (const T*)dA, (const T*)a_out, (const T*)y, mean,
                       invstd, M, C, SW, (int)rpb, part);
    RPE_CHECK_LAUNCH();
    float* c1 = c1c2;
    float* c2 = c1c2 + C;
    prof_split(s, "reduce_finalize_kernel<BnBwdFin>");
    if (int e = reduce_finalize(part, (int)nb, C, dpart, BnBwdFin{(double)M, dgamma, dbeta, c1, c2}, s)) return e;
    if (!dy) return 0;   // statistics only (rpe_bn_backward_reduce): the apply pass is folded into the consumers
    // no ReLU in front and no dz wanted (projection-shortcut BNs): dz == dA, the streaming dz -> dy kernel does the third pass
    if (!a_out && !dz_out && ew_pow2(C / CE)) { prof_split(s, "bn_bwd_apply_dz_kernel"); return bn_apply_dz_launch<T>(dA, y, mean, invstd, gamma, c1, c2, dy, M, C, s); }
    prof_split(s, "bn_bwd_apply_kernel");
    const long n = M * C / CE;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_grid(n)), dim3(256), 0, s, (const T*)dA, (const T*)a_out, (const T*)y, mean, invstd, gamma,
                       (const float*)c1, (const float*)c2, (T*)dy, (T*)dz_out, n, C);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int bn_bwd_from_dz_launch(const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma, const float* stats_part,
                          int tiles, float* dgamma, float* dbeta, void* dy, long M, int C, float* c1c2, double* dpart, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    if (C % CE) return rpe_set_error(RPE_ERR_SHAPE, "bn_bwd_from_dz: C must be a multiple of the 16-byte chunk");
    const int cpr = C / CE;
    float* c1 = c1c2;
    float* c2 = c1c2 + C;
    note_kernel("reduce_finalize_kernel<BnBwdFin>");
    if (int e = reduce_finalize(stats_part, tiles, C, dpart, BnBwdFin{(double)M, dgamma, dbeta, c1, c2}, s)) return e;
    prof_split(s, "bn_bwd_apply_dz_kernel");
    return bn_apply_dz_launch<T>(dz, y, mean, invstd, gamma, c1, c2, dy, M, C, s);
}

template <typename T>
int stem_bwd_launch(const void* dpool, const unsigned char* pidx, const StemAux& ax, const void* y, const float* scale, const float* shift,
                    const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dy, int B, int H, int W,
                    float* part, long part_floats, float* c1c2, double* dpart, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long nbl = part_floats / (2 * 64);
    if (nbl > 1024) nbl = 1024;   // (512 .. 8192 blocks measured within 5 %: the block form is no longer latency-bound)
    if (nbl > (long)B * Ho) nbl = (long)B * Ho;
    if (nbl < 1) return rpe_set_error(RPE_ERR_WORKSPACE, "stem_bwd: partial-sum workspace too small");
    const int rpb = (int)(((long)B * Ho + nbl - 1) / nbl);          // consecutive POOLED rows (two image rows each) per block
    const int nb = (int)(((long)B * Ho + rpb - 1) / rpb);
    note_kernel("stem_bwd_reduce_kernel");
    hipLaunchKernelGGL((stem_bwd_reduce_kernel<T>), dim3(nb), dim3(256), 0, s, (const T*)dpool, pidx, ax, (const T*)y, scale, shift, mean, invstd, B, H, W,
                       Ho, Wo, rpb, part);
    RPE_CHECK_LAUNCH();
    float* c1 = c1c2;
    float* c2 = c1c2 + 64;
    prof_split(s, "reduce_finalize_kernel<BnBwdFin>");
    if (int e = reduce_finalize(part, nb, 64, dpart, BnBwdFin{(double)B * H * W, dgamma, dbeta, c1, c2}, s)) return e;
    prof_split(s, "stem_bwd_apply_kernel");
    const int rpa = 2;   // two pooled rows (four image rows: 57 KB of y at 112 pixels) per block
    hipLaunchKernelGGL((stem_bwd_apply_kernel<T>), dim3((B * Ho + rpa - 1) / rpa), dim3(256), 0, s, (const T*)dpool, pidx, ax, (const T*)y, scale, shift, mean, invstd,
                       gamma, (const float*)c1, (const float*)c2, (T*)dy, B, H, W, Ho, Wo, rpa);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T> int bn_apply_dz_checked(const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma,
                                                    const float* c1c2, void* dy, long rows, int C, hipStream_t s) {
    if (C % Elem<T>::kChunk) return rpe_set_error(RPE_ERR_SHAPE, "bn_backward_apply_dz: C must be a multiple of the 16-byte chunk");
    return bn_apply_dz_launch<T>(dz, y, mean, invstd, gamma, c1c2, c1c2 + C, dy, rows, C, s);
}

}  // namespace rpe

using namespace rpe;

extern "C" {

int rpe_bn_finalize(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches, float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, double* dpart, void* stream) {
    note_kernel("reduce_finalize_kernel<BnFwdFin>");
    if (tiles <= 0 || C <= 0 || count <= 0) return rpe_set_error(RPE_ERR_SHAPE, "bn_finalize: empty problem");
    return reduce_finalize(part, tiles, C, dpart, BnFwdFin{(double)count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, scale,
                                                           shift, save_mean, save_invstd}, (hipStream_t)stream);
}

int rpe_bn_stats_from_gram(int dtype, const void* w, int Co, int Ci, const float* gram, int ones_row, long count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* num_batches, float momentum, float eps, float* scale, float* shift,
                           float* save_mean, float* save_invstd, void* stream) {
    note_kernel("bn_gram_stats_kernel");
    if (!w || !gram || Co <= 0 || Ci <= 0 || Ci > 1024 || count <= 0 || ones_row < Ci) return rpe_set_error(RPE_ERR_SHAPE, "bn_stats_from_gram: bad arguments (in_c <= 1024)");
    const BnFwdFin fin{(double)count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, scale, shift, save_mean, save_invstd};
    hipStream_t s = (hipStream_t)stream;
    const int vec = (Ci & 3) == 0 && ((uintptr_t)gram & 15) == 0;
    if (dtype == RPE_F32) hipLaunchKernelGGL((bn_gram_stats_kernel<float, 1024>), dim3(Co), dim3(1024), 0, s, (const float*)w, Ci, gram, ones_row, vec, fin);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((bn_gram_stats_kernel<bf16, 1024>), dim3(Co), dim3(1024), 0, s, (const bf16*)w, Ci, gram, ones_row, vec, fin);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((bn_gram_stats_kernel<f16, 1024>), dim3(Co), dim3(1024), 0, s, (const f16*)w, Ci, gram, ones_row, vec, fin);
    else return rpe_set_error(RPE_ERR_DTYPE, "bn_stats_from_gram: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* scale, float* shift, void* stream) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_bn_apply(int dtype, const void* y, const void* residual, void* out, const float* scale, const float* shift, long rows, int C,
                 int relu, void* stream) {
    note_kernel("bn_apply_kernel");
    if (dtype == RPE_F32) return bn_apply_launch<float>(y, residual, out, scale, shift, rows, C, relu, nullptr, (hipStream_t)stream);
    if (dtype == RPE_BF16) return bn_apply_launch<bf16>(y, residual, out, scale, shift, rows, C, relu, nullptr, (hipStream_t)stream);
    if (dtype == RPE_F16) return bn_apply_launch<f16>(y, residual, out, scale, shift, rows, C, relu, nullptr, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_apply: unsupported dtype");
}

int rpe_bn_apply_mask(int dtype, const void* y, const void* residual, void* out, const float* scale, const float* shift, long rows, int C,
                      unsigned char* relu_mask, void* stream) {
    note_kernel("bn_apply_kernel<mask>");
    if (!relu_mask || (C % 8)) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_mask: mask buffer and C % 8 == 0 required");
    if (dtype == RPE_BF16) return bn_apply_launch<bf16>(y, residual, out, scale, shift, rows, C, 1, relu_mask, (hipStream_t)stream);
    if (dtype == RPE_F16) return bn_apply_launch<f16>(y, residual, out, scale, shift, rows, C, 1, relu_mask, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_mask: 16-bit element types only");
}

int rpe_bn_apply_res_bn(int dtype, const void* y, const void* res_y, const float* res_scale, const float* res_shift, void* out, const float* scale,
                        const float* shift, long rows, int C, int relu, unsigned char* relu_mask, void* stream) {
    note_kernel(relu_mask ? "bn_apply_kernel<mask,res_bn>" : "bn_apply_kernel<res_bn>");
    if (!res_y || !res_scale || !res_shift) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_res_bn: the residual and its scale / shift are required");
    if (relu_mask && (dtype == RPE_F32 || (C % 8) || !relu)) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_res_bn: the packed mask needs a 16-bit element type, C % 8 == 0 and relu");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RPE_F32) return bn_apply_launch<float>(y, res_y, out, scale, shift, rows, C, relu, nullptr, s, res_scale, res_shift);
    if (dtype == RPE_BF16) return bn_apply_launch<bf16>(y, res_y, out, scale, shift, rows, C, relu, relu_mask, s, res_scale, res_shift);
    if (dtype == RPE_F16) return bn_apply_launch<f16>(y, res_y, out, scale, shift, rows, C, relu, relu_mask, s, res_scale, res_shift);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_res_bn: unsupported dtype");
}

int rpe_bn_backward(int dtype, const void* dA, const void* a_out, const void* y, const float* mean, const float* invstd,
                    const float* gamma, float* dgamma, float* dbeta, void* dy, void* dz_out, long rows, int C, float* part,
                    long part_floats, float* c1c2, double* dpart, void* stream) {
    if (dtype == RPE_F32)
        return bn_bwd_launch<float>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, dy, dz_out, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_BF16)
        return bn_bwd_launch<bf16>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, dy, dz_out, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_F16)
        return bn_bwd_launch<f16>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, dy, dz_out, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_backward: unsupported dtype");
}

/* the reduction half alone: dgamma, dbeta and c1c2 = (mean(dz), mean(dz * xhat)) -- for a BatchNorm whose apply pass is folded into
 * its consumers (rpe_bn_bwd_fold_conv1x1 / rpe_conv1x1_wgrad_folded) and whose sums no fused epilogue has produced */
int rpe_bn_backward_reduce(int dtype, const void* dA, const void* a_out, const void* y, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma, float* dbeta, long rows, int C, float* part, long part_floats, float* c1c2, double* dpart, void* stream) {
    if (dtype == RPE_F32) return bn_bwd_launch<float>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, nullptr, nullptr, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_BF16) return bn_bwd_launch<bf16>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, nullptr, nullptr, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_F16) return bn_bwd_launch<f16>(dA, a_out, y, mean, invstd, gamma, dgamma, dbeta, nullptr, nullptr, rows, C, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_backward_reduce: unsupported dtype");
}

int rpe_bn_backward_from_dz(int dtype, const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma,
                            const float* stats_part, int tiles, float* dgamma, float* dbeta, void* dy, long rows, int C, float* c1c2,
                            double* dpart, void* stream) {
    if (dtype == RPE_F32)
        return bn_bwd_from_dz_launch<float>(dz, y, mean, invstd, gamma, stats_part, tiles, dgamma, dbeta, dy, rows, C, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_BF16)
        return bn_bwd_from_dz_launch<bf16>(dz, y, mean, invstd, gamma, stats_part, tiles, dgamma, dbeta, dy, rows, C, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_F16)
        return bn_bwd_from_dz_launch<f16>(dz, y, mean, invstd, gamma, stats_part, tiles, dgamma, dbeta, dy, rows, C, c1c2, dpart, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_backward_from_dz: unsupported dtype");
}

// The two halves of rpe_bn_backward_from_dz as separate calls: the per-channel coefficients (one tiny launch) are what the
// K-concatenated data gradient needs (rpe_bn_bwd_fold_conv1x1); the streaming dz -> dy pass then only feeds the weight gradient
// and can run on another stream.
int rpe_bn_backward_coeffs(const float* stats_part, int tiles, int C, long rows, float* dgamma, float* dbeta, float* c1c2, double* dpart, void* stream) {
    note_kernel("reduce_finalize_kernel<BnBwdFin>");
    if (!stats_part || !c1c2 || !dpart || tiles <= 0 || C <= 0 || rows <= 0) return rpe_set_error(RPE_ERR_SHAPE, "bn_backward_coeffs: bad arguments");
    return reduce_finalize(stats_part, tiles, C, dpart, BnBwdFin{(double)rows, dgamma, dbeta, c1c2, c1c2 + C}, (hipStream_t)stream);
}

int rpe_bn_backward_coeffs_t(int dtype, const float* stats_part, int tiles, int C, long rows, const float* dzt_a, const void* w, int Ci, const float* mean,
                             const float* invstd, float* dgamma, float* dbeta, float* c1c2, double* dpart, void* stream) {
    note_kernel("reduce_finalize_kernel<BnBwdFinT>");
    if (!stats_part || !c1c2 || !dpart || !dzt_a || !w || !mean || !invstd || tiles <= 0 || C <= 0 || Ci <= 0 || rows <= 0)
        return rpe_set_error(RPE_ERR_SHAPE, "bn_backward_coeffs_t: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RPE_BF16) return reduce_finalize(stats_part, tiles, C, dpart, BnBwdFinT<bf16>{(double)rows, dgamma, dbeta, c1c2, c1c2 + C, dzt_a, (const bf16*)w, Ci, mean, invstd}, s);
    if (dtype == RPE_F16) return reduce_finalize(stats_part, tiles, C, dpart, BnBwdFinT<f16>{(double)rows, dgamma, dbeta, c1c2, c1c2 + C, dzt_a, (const f16*)w, Ci, mean, invstd}, s);
    if (dtype == RPE_F32) return reduce_finalize(stats_part, tiles, C, dpart, BnBwdFinT<float>{(double)rows, dgamma, dbeta, c1c2, c1c2 + C, dzt_a, (const float*)w, Ci, mean, invstd}, s);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_backward_coeffs_t: unsupported dtype");
}

int rpe_bn_backward_apply_dz(int dtype, const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma, const float* c1c2,
                             void* dy, long rows, int C, void* stream) {
    note_kernel("bn_bwd_apply_dz_kernel");
    if (dtype == RPE_F32) return bn_apply_dz_checked<float>(dz, y, mean, invstd, gamma, c1c2, dy, rows, C, (hipStream_t)stream);
    if (dtype == RPE_BF16) return bn_apply_dz_checked<bf16>(dz, y, mean, invstd, gamma, c1c2, dy, rows, C, (hipStream_t)stream);
    if (dtype == RPE_F16) return bn_apply_dz_checked<f16>(dz, y, mean, invstd, gamma, c1c2, dy, rows, C, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "bn_backward_apply_dz: unsupported dtype");
}

int rpe_maxpool3x3s2_fwd(int dtype, const void* x, void* out, unsigned char* idx, int B, int H, int W, int C, void* stream) {
    note_kernel("maxpool_fwd_kernel");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long n = (long)B * Ho * Wo * C;
    if (dtype == RPE_F32) hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, idx, B, H, W, C, Ho, Wo);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, idx, B, H, W, C, Ho, Wo);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((maxpool_fwd_kernel<f16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out, idx, B, H, W, C, Ho, Wo);
    else return rpe_set_error(RPE_ERR_DTYPE, "maxpool: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_bn_apply_maxpool3x3s2(int dtype, const void* y, const float* scale, const float* shift, void* a, void* out, unsigned char* idx, int B, int H,
                              int W, int C, void* stream) {
    note_kernel("bn_apply_maxpool_kernel");
    if (B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || C <= 0 || (C % 8)) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_maxpool: H and W even, C % 8 == 0");
    if (!a) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_maxpool: the activated output is required (rpe_bn_apply_maxpool3x3s2_aux may omit it)");
    const int Ho = H / 2, Wo = W / 2;
    const long n = (long)B * Ho * Wo * C;
    hipStream_t s = (hipStream_t)stream;
    const StemAuxFwd none{};
    if (dtype == RPE_F32) hipLaunchKernelGGL((bn_apply_maxpool_kernel<float, false>), dim3(ew_grid(n / 4)), dim3(256), 0, s, (const float*)y, scale, shift, (float*)a, (float*)out, idx, B, H, W, C, Ho, Wo, none, walk_take());
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((bn_apply_maxpool_kernel<bf16, false>), dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16*)y, scale, shift, (bf16*)a, (bf16*)out, idx, B, H, W, C, Ho, Wo, none, walk_take());
    else if (dtype == RPE_F16) hipLaunchKernelGGL((bn_apply_maxpool_kernel<f16, false>), dim3(ew_grid(n / 8)), dim3(256), 0, s, (const f16*)y, scale, shift, (f16*)a, (f16*)out, idx, B, H, W, C, Ho, Wo, none, walk_take());
    else return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_maxpool: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_bn_apply_maxpool3x3s2_aux(int dtype, const void* y, const float* scale, const float* shift, void* a, void* out, unsigned char* idx, int B, int H,
                                  int W, const float* aux_w, const float* aux_bias, const float* depth_feat, float* aux_out, long ld_aux_out, float* aux_raw,
                                  unsigned char* aux_idx, void* stream) {
    note_kernel("bn_apply_maxpool_kernel<aux>");
    if (B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_maxpool_aux: H and W even (64 channels)");
    if (!aux_w || !aux_bias || !aux_out || !aux_raw || !aux_idx || ld_aux_out < (long)(H / 2) * (W / 2)) return rpe_set_error(RPE_ERR_SHAPE, "bn_apply_maxpool_aux: aux head weight, bias and outputs are required");
    const int Ho = H / 2, Wo = W / 2, C = 64;
    const long n = (long)B * Ho * Wo * C;
    hipStream_t s = (hipStream_t)stream;
    const StemAuxFwd aux{aux_w, aux_bias, depth_feat, aux_out, ld_aux_out, aux_raw, aux_idx};
    // whole blocks of whole groups: the grid covers every chunk exactly once (no grid-stride tail splits a group)
    if (dtype == RPE_F32) hipLaunchKernelGGL((bn_apply_maxpool_kernel<float, true>), dim3(ew_grid(n / 4)), dim3(256), 0, s, (const float*)y, scale, shift, (float*)a, (float*)out, idx, B, H, W, C, Ho, Wo, aux, walk_take());
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((bn_apply_maxpool_kernel<bf16, true>), dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16*)y, scale, shift, (bf16*)a, (bf16*)out, idx, B, H, W, C, Ho, Wo, aux, walk_take());
    else if (dtype == RPE_F16) hipLaunchKernelGGL((bn_apply_maxpool_kernel<f16, true>), dim3(ew_grid(n / 8)), dim3(256), 0, s, (const f16*)y, scale, shift, (f16*)a, (f16*)out, idx, B, H, W, C, Ho, Wo, aux, walk_take());
    else return rpe_set_error(RPE_ERR_DTYPE, "bn_apply_maxpool_aux: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_maxpool3x3s2_bwd(int dtype, const void* dout, const unsigned char* idx, const void* addend, void* dx, int B, int H, int W, int C,
                         void* stream) {
    note_kernel("maxpool_bwd_kernel");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long n = (long)B * H * W * C;
    if (dtype == RPE_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)dout, idx, (const float*)addend, (float*)dx, B, H, W, C, Ho, Wo);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, idx, (const bf16*)addend, (bf16*)dx, B, H, W, C, Ho, Wo);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((maxpool_bwd_kernel<f16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)dout, idx, (const f16*)addend, (f16*)dx, B, H, W, C, Ho, Wo);
    else return rpe_set_error(RPE_ERR_DTYPE, "maxpool: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_avgpool_fwd(int dtype, const void* x, float* out, int B, int HW, int C, void* stream) {
    if ((long)B * C <= 65536 && HW >= 16) {   // fewer chunks than lanes on the chip: 8 lanes per chunk
        if (dtype == RPE_F32) hipLaunchKernelGGL((avgpool_fwd_few_kernel<float>), dim3(ceil_div((long)B * C / 4 * 8, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, B, HW, C);
        else if (dtype == RPE_BF16) hipLaunchKernelGGL((avgpool_fwd_few_kernel<bf16>), dim3(ceil_div((long)B * C / 8 * 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, out, B, HW, C);
        else if (dtype == RPE_F16) hipLaunchKernelGGL((avgpool_fwd_few_kernel<f16>), dim3(ceil_div((long)B * C / 8 * 8, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, out, B, HW, C);
        else return rpe_set_error(RPE_ERR_DTYPE, "avgpool: unsupported dtype");
        RPE_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == RPE_F32) hipLaunchKernelGGL((avgpool_fwd_kernel<float>), dim3(ceil_div((long)B * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, B, HW, C);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((avgpool_fwd_kernel<bf16>), dim3(ceil_div((long)B * C / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, out, B, HW, C);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((avgpool_fwd_kernel<f16>), dim3(ceil_div((long)B * C / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, out, B, HW, C);
    else return rpe_set_error(RPE_ERR_DTYPE, "avgpool: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_avgpool_bwd(int dtype, const float* dout, void* dx, int B, int HW, int C, void* stream) {
    const long n = (long)B * HW * C;
    if (dtype == RPE_F32) hipLaunchKernelGGL((avgpool_bwd_kernel<float>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dout, (float*)dx, B, HW, C);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, dout, (bf16*)dx, B, HW, C);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((avgpool_bwd_kernel<f16>), dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, dout, (f16*)dx, B, HW, C);
    else return rpe_set_error(RPE_ERR_DTYPE, "avgpool: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_stage_image_nhwc4(int dtype, const float* img_nchw, void* out, int B, int H, int W, void* stream) {
    note_kernel("nchw_to_nhwc4_kernel");
    const long n = (long)B * H * W;
    if (dtype == RPE_F32) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, img_nchw, (float*)out, B, H, W);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<bf16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, img_nchw, (bf16*)out, B, H, W);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<f16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, img_nchw, (f16*)out, B, H, W);
    else return rpe_set_error(RPE_ERR_DTYPE, "stage_image: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_stage_frames_u8(int dtype, const unsigned char* frames, void* out, int B, int Hs, int Ws, int H, int W, const float* mean3_host,
                        const float* std3_host, void* stream) {
    note_kernel("frames_u8_to_nhwc4_kernel");
    if (B <= 0 || H <= 0 || W <= 0 || Hs < H || Ws < W || !mean3_host || !std3_host) return rpe_set_error(RPE_ERR_SHAPE, "stage_frames_u8: bad shape");
    const long n = (long)B * H * W;
    const float i0 = 1.f / std3_host[0], i1 = 1.f / std3_host[1], i2 = 1.f / std3_host[2];
    if (dtype == RPE_F32) hipLaunchKernelGGL((frames_u8_to_nhwc4_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, frames, (float*)out, B, Hs, Ws, H, W, mean3_host[0], mean3_host[1], mean3_host[2], i0, i1, i2);
    else if (dtype == RPE_BF16) hipLaunchKernelGGL((frames_u8_to_nhwc4_kernel<bf16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, frames, (bf16*)out, B, Hs, Ws, H, W, mean3_host[0], mean3_host[1], mean3_host[2], i0, i1, i2);
    else if (dtype == RPE_F16) hipLaunchKernelGGL((frames_u8_to_nhwc4_kernel<f16>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, frames, (f16*)out, B, Hs, Ws, H, W, mean3_host[0], mean3_host[1], mean3_host[2], i0, i1, i2);
    else return rpe_set_error(RPE_ERR_DTYPE, "stage_frames_u8: unsupported dtype");
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_stage_frames_u8_resized(int dtype, const unsigned char* frames, void* out, int B, int Hs, int Ws, int Hr, int Wr, int top, int left, int H, int W,
                                const int* xb, const int* xk, int ksx, const int* yb, const int* yk, int ksy, unsigned char* tmp, const float* mean3_host,
                                const float* std3_host, void* stream) {
    note_kernel("resize_h_u8_kernel + resize_v_crop_norm_kernel");
    if (B <= 0 || H <= 0 || W <= 0 || Hr < top + H || Wr < left + W || top < 0 || left < 0 || !mean3_host || !std3_host || !frames || !out)
        return rpe_set_error(RPE_ERR_SHAPE, "stage_frames_u8_resized: bad shape");
    const bool horiz = Wr != Ws, vert = Hr != Hs;
    if ((horiz && (!xb || !xk || ksx <= 0 || !tmp)) || (vert && (!yb || !yk || ksy <= 0))) return rpe_set_error(RPE_ERR_SHAPE, "stage_frames_u8_resized: missing tap tables / intermediate");
    const unsigned char* mid = frames;
    if (horiz) {
        hipLaunchKernelGGL(resize_h_u8_kernel, dim3(ew_grid((long)B * Hs * Wr)), dim3(256), 0, (hipStream_t)stream, frames, tmp, B, Hs, Ws, Wr, xb, xk, ksx);
        RPE_CHECK_LAUNCH();
        mid = tmp;
    }
    const long n = (long)B * H * W;
    const float i0 = 1.f / std3_host[0], i1 = 1.f / std3_host[1], i2 = 1.f / std3_host[2];
#define RPE_RV(T) hipLaunchKernelGGL((resize_v_crop_norm_kernel<T>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, mid, (T*)out, B, Hs, Wr, top, left, H, W, yb, yk, ksy, vert ? 1 : 0, mean3_host[0], mean3_host[1], mean3_host[2], i0, i1, i2)
    if (dtype == RPE_F32) RPE_RV(float);
    else if (dtype == RPE_BF16) RPE_RV(bf16);
    else if (dtype == RPE_F16) RPE_RV(f16);
    else return rpe_set_error(RPE_ERR_DTYPE, "stage_frames_u8_resized: unsupported dtype");
#undef RPE_RV
    RPE_CHECK_LAUNCH();
    return 0;
}

int rpe_stem_bwd(int dtype, const void* dpool, const unsigned char* pool_idx, const void* y, const float* scale, const float* shift, const float* mean,
                 const float* invstd, const float* gamma, const float* aux_dout, long aux_ld, const float* aux_depth_feat,
                 const unsigned char* aux_idx, const float* aux_w, float* dgamma, float* dbeta, void* dy, int B, int H, int W, float* part,
                 long part_floats, float* c1c2, double* dpart, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return rpe_set_error(RPE_ERR_SHAPE, "stem_bwd: H and W must be even");
    if (aux_dout && (!aux_idx || !aux_w)) return rpe_set_error(RPE_ERR_SHAPE, "stem_bwd: aux gradient needs its winner indices and weight");
    const StemAux ax{aux_dout, aux_ld, aux_depth_feat, aux_idx, aux_w};
    if (dtype == RPE_F32) return stem_bwd_launch<float>(dpool, pool_idx, ax, y, scale, shift, mean, invstd, gamma, dgamma, dbeta, dy, B, H, W, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_BF16) return stem_bwd_launch<bf16>(dpool, pool_idx, ax, y, scale, shift, mean, invstd, gamma, dgamma, dbeta, dy, B, H, W, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    if (dtype == RPE_F16) return stem_bwd_launch<f16>(dpool, pool_idx, ax, y, scale, shift, mean, invstd, gamma, dgamma, dbeta, dy, B, H, W, part, part_floats, c1c2, dpart, (hipStream_t)stream);
    return rpe_set_error(RPE_ERR_DTYPE, "stem_bwd: unsupported dtype");
}

}  // extern "C"

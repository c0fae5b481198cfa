// MFMA implicit-GEMM kernels for gfx950 (CDNA4, wave64).
//
//   nt_kernel : C[M][N] = gather(A)[M][K] * W[N][K]^T  (+bias, +addend, relu, BN partial stats)
//               conv forward, conv data-gradient, Linear forward / data-gradient.
//   tn_kernel : D[I][J] += sum_m P[m][I] * gather(Q)[m][J]   (fp32 atomics)
//               conv weight-gradient, Linear weight-gradient.
//
// Layout: activations NHWC (channels contiguous), weights [N][K] with K = (r, s, c) contiguous.
// Both operands are staged global -> registers -> LDS in 16-byte chunks (4 chunks = one
// MFMA K-step per row: 32 bf16 or 16 f32), double buffered, one barrier per K-step.
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so the
// accumulator registers of a lane run along N (channels): the epilogue stores 4 consecutive
// channels per lane straight to NHWC memory (8 B bf16 / 16 B f32) without an LDS transpose.
#include "igemm.h"

namespace rpe {

// LDS slot permutation for the [row][4 x 16 B] staging tiles: chunk ^= f(row>>2 & 3),
// f = {0,2,3,1}.  ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS); with lane -> (row = l&15, chunk = l>>4)
// this f puts the four row-quads of every group on four different 16-byte slots.
__device__ __forceinline__ int nt_swz(int row, int chunk) { return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

// Bijective XCD remap: consecutive logical tiles share one XCD's L2 (blocks b, b+8 share an XCD).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    __device__ static __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
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

__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }
__device__ __forceinline__ u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

// One 16-byte chunk of the stem (conv1) im2col row: x4 is [B][H][W][4], k = (r*8 + s)*4 + c with the
// 7x7 taps padded to 8x8 (tap 7 reads as zero).  A chunk covers 2 pixels (bf16) or 1 pixel (f32).
template <typename T>
__device__ __forceinline__ u32x4 stem_chunk(const T* x, long img_base, int hb, int wb, int H, int W, int k, bool ok) {
    constexpr int CE = Elem<T>::kChunk;
    const int r = k >> 5, s0 = (k & 31) >> 2;
    const int ih = hb + r;
    u32x4 v = zero16();
    if (!ok || r >= 7 || ih < 0 || ih >= H) return v;
    if (CE == 4) {
        const int iw = wb + s0;
        if (s0 < 7 && iw >= 0 && iw < W) v = ld16(x + img_base + ((long)ih * W + iw) * 4);
    } else {
        const int iw0 = wb + s0, iw1 = iw0 + 1;
        if (s0 < 7 && iw0 >= 0 && iw0 < W) { u32x2 t = *(const u32x2*)(x + img_base + ((long)ih * W + iw0) * 4); v.x = t.x; v.y = t.y; }
        if (s0 + 1 < 7 && iw1 >= 0 && iw1 < W) { u32x2 t = *(const u32x2*)(x + img_base + ((long)ih * W + iw1) * 4); v.z = t.x; v.w = t.y; }
    }
    return v;
}

// -----------------------------------------------------------------------------------------------
// NT kernel
// -----------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void nt_kernel(const NTArgs<T> p) {
    constexpr int CE = Elem<T>::kChunk, BK = 4 * CE;
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int AR = BM / 64, BR = BN / 64;
    constexpr int STAGE = (BM + BN) * 4;  // 16-byte units
    __shared__ u32x4 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int lb = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lrow = tid >> 2, lchunk = tid & 3;
    const Gather& g = p.g;

    long a_base[AR];
    int a_hb[AR], a_wb[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + lrow + 64 * i;
        a_ok[i] = m < p.M;
        a_hb[i] = a_wb[i] = 0;
        if (MODE == MODE_DENSE) {
            a_base[i] = (long)m * p.lda + lchunk * CE;
        } else {
            const unsigned mm = a_ok[i] ? (unsigned)m : 0u;
            const unsigned b = fd_div(mm, g.div_hw);
            const unsigned rem = mm - b * g.div_hw.d;
            const unsigned oh = fd_div(rem, g.div_w);
            const unsigned ow = rem - oh * g.div_w.d;
            a_base[i] = (long)b * g.img_stride;
            a_hb[i] = (int)oh * g.sn + g.base_h;
            a_wb[i] = (int)ow * g.sn + g.base_w;
        }
    }
    long b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + lrow + 64 * i;
        b_ok[i] = n < p.N;
        b_off[i] = (long)n * p.ldb + lchunk * CE;
    }

    u32x4 ra[AR], rb[BR];
    // uniform K-walk state for MODE_CONV: k = (r*S + s)*C + c0
    int kbase = 0, c0 = 0, tr = 0, ts = 0;

    auto load_tile = [&]() {
        const bool kok = (kbase + lchunk * CE) < p.K;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            u32x4 v = zero16();
            if (MODE == MODE_DENSE) {
                if (a_ok[i] && kok) v = ld16(p.A + a_base[i] + kbase);
            } else if (MODE == MODE_CONV) {
                const int nh = a_hb[i] + g.tap_sign * tr, nw = a_wb[i] + g.tap_sign * ts;
                const int msk = (1 << g.sd_shift) - 1;
                const int ih = nh >> g.sd_shift, iw = nw >> g.sd_shift;
                const bool ok = a_ok[i] && nh >= 0 && nw >= 0 && ((nh | nw) & msk) == 0 && ih < g.H && iw < g.W;
                if (ok) v = ld16(p.A + a_base[i] + ((long)ih * g.W + iw) * g.C + c0 + lchunk * CE);
            } else {
                v = stem_chunk<T>(p.A, a_base[i], a_hb[i], a_wb[i], g.H, g.W, kbase + lchunk * CE, a_ok[i]);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = (b_ok[i] && kok) ? ld16(p.Bw + b_off[i] + kbase) : zero16();
    };
    auto advance_k = [&]() {
        kbase += BK;
        if (MODE == MODE_CONV) {
            c0 += BK;
            if (c0 >= g.C) { c0 = 0; if (++ts >= g.S) { ts = 0; ++tr; } }
        }
    };
    auto store_tile = [&](int st) {
        u32x4* base = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < AR; ++i) { const int row = lrow + 64 * i; base[row * 4 + nt_swz(row, lchunk)] = ra[i]; }
#pragma unroll
        for (int i = 0; i < BR; ++i) { const int row = lrow + 64 * i; base[BM * 4 + row * 4 + nt_swz(row, lchunk)] = rb[i]; }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tile();
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fc = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { advance_k(); load_tile(); }
        const u32x4* base = lds + cur * STAGE;
        u32x4 af[FM], wf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) { const int row = wave_m * WM + i * 16 + fr; af[i] = base[row * 4 + nt_swz(row, fc)]; }
#pragma unroll
        for (int i = 0; i < FN; ++i) { const int row = wave_n * WN + i * 16 + fr; wf[i] = base[BM * 4 + row * 4 + nt_swz(row, fc)]; }
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) Mma<T>::run(wf[a], af[b], acc[a][b]);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = ..+fr][n = ..+4*fc + 0..3] per 16x16 fragment -----------
    if (p.stats_part) {
        float* red = (float*)lds;  // [2 wave_m][BN][2]
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            float s[4], q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = 0.f; q[j] = 0.f; }
#pragma unroll
            for (int b = 0; b < FM; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float v = acc[a][b][j]; s[j] += v; q[j] += v * v; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s[j] += __shfl_xor(s[j], o); q[j] += __shfl_xor(q[j], o); }
            }
            if (fr == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = wave_n * WN + a * 16 + fc * 4 + j;
                    red[(wave_m * BN + nl) * 2 + 0] = s[j];
                    red[(wave_m * BN + nl) * 2 + 1] = q[j];
                }
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            const float s = red[tid * 2] + red[(BN + tid) * 2];
            const float q = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            p.stats_part[((long)tile_m * 2 + 0) * p.N + n0 + tid] = s;
            p.stats_part[((long)tile_m * 2 + 1) * p.N + n0 + tid] = q;
        }
    }
    const bool vec_c = (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0;
    const bool vec_add = p.addend && (p.ld_add & 3) == 0 && (((uintptr_t)p.addend) & 15) == 0;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wave_m * WM + b * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wave_n * WN + a * 16 + fc * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j];
            const bool full = n + 3 < p.N;
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += p.bias[n + j];
            }
            if (p.addend) {
                const T* ad = p.addend + (long)m * p.ld_add + n;
                if (full && vec_add) {
                    if (CE == 4) { f32x4 t = *(const f32x4*)ad; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
                    else { u32x2 t = *(const u32x2*)ad; v[0] += __uint_as_float(t.x << 16); v[1] += __uint_as_float(t.x & 0xffff0000u);
                           v[2] += __uint_as_float(t.y << 16); v[3] += __uint_as_float(t.y & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += Elem<T>::to_f(ad[j]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            T* cp = p.C + (long)m * p.ldc + n;
            if (full && vec_c) {
                if (CE == 4) { *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]}; }
                else { u32x2 t; t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); *(u32x2*)cp = t; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < p.N) cp[j] = Elem<T>::from_f(v[j]);
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// TN kernel (weight gradients)
// -----------------------------------------------------------------------------------------------
// LDS tiles are [m][cols] row-major (cols contiguous, as in memory).  The bf16 operands are
// fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose), the f32 ones with ds_read_b32.
// 16-byte chunk permutation inside a row so the 8 rows a 32-lane half touches cover all 64 banks.
template <typename T, int CPR> __device__ __forceinline__ int tn_swz(int row) {
    if (sizeof(T) == 4) return (row & 1) << 2;                                  // f32: +16 dwords for odd rows
    if (CPR >= 16) return ((row & 3) | ((row >> 1) & 4)) << 1;                  // 256-B rows
    return (((row >> 1) & 1) | ((row >> 2) & 2)) << 1;                          // 128-B rows (parity picks the half)
}

template <typename T, int BI, int BJ, int MODE>
__global__ __launch_bounds__(256) void tn_kernel(const TNArgs<T> p) {
    constexpr int CE = Elem<T>::kChunk, BMK = 4 * CE;
    constexpr int WI = BI / 2, WJ = BJ / 2, FI = WI / 16, FJ = WJ / 16;
    constexpr int CPI = BI / CE, CPJ = BJ / CE;        // chunks per row
    constexpr int RPI = 256 / CPI, RPJ = 256 / CPJ;    // rows per pass
    constexpr int NPI = (BMK + RPI - 1) / RPI, NPJ = (BMK + RPJ - 1) / RPJ;
    constexpr int PT = BMK * CPI, QT = BMK * CPJ;      // tile sizes in 16-byte units
    constexpr int STAGE = PT + QT;
    __shared__ u32x4 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_i = wave >> 1, wave_j = wave & 1;
    const int nt = p.tiles_i * p.tiles_j;
    const int lb = xcd_remap(blockIdx.x, nt * p.splits);
    const int split = lb / nt, t2 = lb - split * nt;
    const int tile_j = t2 % p.tiles_j, tile_i = t2 / p.tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    if (m_begin >= m_end) return;
    const Gather& g = p.g;

    const int p_cc = tid % CPI, p_r = tid / CPI;
    const int q_cc = tid % CPJ, q_r = tid / CPJ;
    const bool p_col_ok = (i0 + p_cc * CE) < p.I;
    const bool q_col_ok = (j0 + q_cc * CE) < p.J;
    // conv: the BJ-wide column tile lies inside one tap (BJ divides C)
    int tap_r = 0, tap_s = 0, tap_c = 0;
    if (MODE == MODE_CONV) { const int rs = j0 / g.C; tap_c = j0 - rs * g.C; tap_r = rs / g.S; tap_s = rs - tap_r * g.S; }

    u32x4 rp[NPI], rq[NPJ];
    auto load_tile = [&](int mb) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int row = p_r + i * RPI;
            const int m = mb + row;
            rp[i] = (row < BMK && m < m_end && p_col_ok) ? ld16(p.P + (long)m * p.ldp + i0 + p_cc * CE) : zero16();
        }
#pragma unroll
        for (int i = 0; i < NPJ; ++i) {
            const int row = q_r + i * RPJ;
            const int m = mb + row;
            u32x4 v = zero16();
            const bool ok = row < BMK && m < m_end && q_col_ok;
            if (MODE == MODE_DENSE) {
                if (ok) v = ld16(p.Q + (long)m * p.ldq + j0 + q_cc * CE);
            } else {
                const unsigned mm = ok ? (unsigned)m : 0u;
                const unsigned b = fd_div(mm, g.div_hw);
                const unsigned rem = mm - b * g.div_hw.d;
                const unsigned oh = fd_div(rem, g.div_w);
                const unsigned ow = rem - oh * g.div_w.d;
                const int hb = (int)oh * g.sn + g.base_h, wb = (int)ow * g.sn + g.base_w;
                if (MODE == MODE_CONV) {
                    const int ih = hb + tap_r, iw = wb + tap_s;
                    if (ok && ih >= 0 && iw >= 0 && ih < g.H && iw < g.W)
                        v = ld16(p.Q + (long)b * g.img_stride + ((long)ih * g.W + iw) * g.C + tap_c + q_cc * CE);
                } else {
                    v = stem_chunk<T>(p.Q, (long)b * g.img_stride, hb, wb, g.H, g.W, j0 + q_cc * CE, ok);
                }
            }
            rq[i] = v;
        }
    };
    auto store_tile = [&](int st) {
        u32x4* base = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < NPI; ++i) { const int row = p_r + i * RPI; if (row < BMK) base[row * CPI + (p_cc ^ tn_swz<T, CPI>(row))] = rp[i]; }
#pragma unroll
        for (int i = 0; i < NPJ; ++i) { const int row = q_r + i * RPJ; if (row < BMK) base[PT + row * CPJ + (q_cc ^ tn_swz<T, CPJ>(row))] = rq[i]; }
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (m_end - m_begin + BMK - 1) / BMK;
    load_tile(m_begin);
    store_tile(0);
    __syncthreads();
    const int fg = lane >> 4, fl = lane & 15;
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) load_tile(m_begin + (st + 1) * BMK);
        const char* pb = (const char*)(lds + cur * STAGE);
        const char* qb = (const char*)(lds + cur * STAGE + PT);
        if (sizeof(T) == 2) {
            // lane (g = l>>4, q = (l&15)>>2, pp = l&3) addresses row m = 8g + 4h + q, columns base + 4pp..4pp+3;
            // it receives column base + (l&15) for rows 8g + 4h + 0..3  -> MFMA k = 8g + (4h + e)
            const int q = fl >> 2, pp = fl & 3;
            u32x4 pf[FI], qf[FJ];
#pragma unroll
            for (int a = 0; a < FI; ++a) {
                const int col = wave_i * WI + a * 16 + 4 * pp;  // element column inside the tile
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 8 * fg + 4 * h + q;
                    const int chunk = (col >> 3) ^ tn_swz<T, CPI>(row);
                    const char* ad = pb + (row * CPI + chunk) * 16 + (col & 7) * 2;
                    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                    u32x2 tt = __builtin_bit_cast(u32x2, t);
                    w[2 * h] = tt.x; w[2 * h + 1] = tt.y;
                }
                pf[a] = u32x4{w[0], w[1], w[2], w[3]};
            }
#pragma unroll
            for (int b = 0; b < FJ; ++b) {
                const int col = wave_j * WJ + b * 16 + 4 * pp;
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 8 * fg + 4 * h + q;
                    const int chunk = (col >> 3) ^ tn_swz<T, CPJ>(row);
                    const char* ad = qb + (row * CPJ + chunk) * 16 + (col & 7) * 2;
                    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                    u32x2 tt = __builtin_bit_cast(u32x2, t);
                    w[2 * h] = tt.x; w[2 * h + 1] = tt.y;
                }
                qf[b] = u32x4{w[0], w[1], w[2], w[3]};
            }
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pf[a]), __builtin_bit_cast(bf16x8, qf[b]), acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = kk * 4 + fg;
                float pf[FI], qf[FJ];
#pragma unroll
                for (int a = 0; a < FI; ++a) {
                    const int col = wave_i * WI + a * 16 + fl;
                    pf[a] = *(const float*)(pb + (row * CPI + ((col >> 2) ^ tn_swz<T, CPI>(row))) * 16 + (col & 3) * 4);
                }
#pragma unroll
                for (int b = 0; b < FJ; ++b) {
                    const int col = wave_j * WJ + b * 16 + fl;
                    qf[b] = *(const float*)(qb + (row * CPJ + ((col >> 2) ^ tn_swz<T, CPJ>(row))) * 16 + (col & 3) * 4);
                }
#pragma unroll
                for (int a = 0; a < FI; ++a)
#pragma unroll
                    for (int b = 0; b < FJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a], qf[b], acc[a][b], 0, 0, 0);
            }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }
    // D[i = ..+4*fg+reg][j = ..+fl]: for a fixed register 16 lanes add 64 contiguous bytes of one row
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) {
            const int j = j0 + wave_j * WJ + b * 16 + fl;
            if (j >= p.J) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wave_i * WI + a * 16 + 4 * fg + r;
                if (i < p.I) atomicAdd(p.D + (long)i * p.ldd + j, acc[a][b][r]);
            }
        }
}

// -----------------------------------------------------------------------------------------------
// host launchers
// -----------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int MODE> static int launch_nt_cfg(NTArgs<T>& a, hipStream_t s) {
    a.tiles_m = ceil_div(a.M, BM);
    a.tiles_n = ceil_div(a.N, BN);
    const long nwg = (long)a.tiles_m * a.tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffL) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: bad grid");
    hipLaunchKernelGGL((nt_kernel<T, BM, BN, MODE>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T> int launch_nt(NTArgs<T>& a, int mode, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: empty problem");
    if ((a.ldb % CE) || (((uintptr_t)a.Bw) & 15) || (((uintptr_t)a.A) & 15))
        return rpe_set_error(RPE_ERR_ALIGN, "igemm_nt: operands must be 16-byte aligned with ld a multiple of the 16-byte chunk");
    if (mode == MODE_DENSE && ((a.lda % CE) || (a.K % CE))) return rpe_set_error(RPE_ERR_ALIGN, "igemm_nt: dense lda/K must be chunk multiples");
    if (mode == MODE_CONV && (a.g.C % (4 * CE))) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: conv channels must be a multiple of the K-step");
    const bool wide = a.N > 64;
    // stats partials are indexed by the m-tile: callers size them with rpe_conv_stats_tiles() (BM = 128)
    if (mode == MODE_DENSE) return wide ? launch_nt_cfg<T, 128, 128, MODE_DENSE>(a, s) : launch_nt_cfg<T, 128, 64, MODE_DENSE>(a, s);
    if (mode == MODE_CONV) return wide ? launch_nt_cfg<T, 128, 128, MODE_CONV>(a, s) : launch_nt_cfg<T, 128, 64, MODE_CONV>(a, s);
    return launch_nt_cfg<T, 128, 64, MODE_STEM>(a, s);
}
template int launch_nt<float>(NTArgs<float>&, int, hipStream_t);
template int launch_nt<bf16>(NTArgs<bf16>&, int, hipStream_t);

template <typename T, int BI, int BJ, int MODE> static int launch_tn_cfg(TNArgs<T>& a, hipStream_t s) {
    constexpr int BMK = 4 * Elem<T>::kChunk;
    a.tiles_i = ceil_div(a.I, BI);
    a.tiles_j = ceil_div(a.J, BJ);
    const long tiles = (long)a.tiles_i * a.tiles_j;
    // enough workgroups to fill 256 CUs a few times over, but keep >= 16 m-steps per split
    long want = (1536 + tiles - 1) / tiles;
    long max_splits = a.M / (16 * BMK);
    if (max_splits < 1) max_splits = 1;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    long rps = (a.M + want - 1) / want;
    rps = (rps + BMK - 1) / BMK * BMK;
    a.rows_per_split = (int)rps;
    a.splits = (int)((a.M + rps - 1) / rps);
    const long nwg = tiles * a.splits;
    hipLaunchKernelGGL((tn_kernel<T, BI, BJ, MODE>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T> int launch_tn(TNArgs<T>& a, int mode, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    if (a.M <= 0 || a.I <= 0 || a.J <= 0) return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: empty problem");
    if ((a.ldp % CE) || (((uintptr_t)a.P) & 15) || (((uintptr_t)a.Q) & 15))
        return rpe_set_error(RPE_ERR_ALIGN, "igemm_tn: operands must be 16-byte aligned with ld a multiple of the 16-byte chunk");
    if (mode == MODE_DENSE && (a.ldq % CE)) return rpe_set_error(RPE_ERR_ALIGN, "igemm_tn: dense ldq must be a chunk multiple");
    if (mode == MODE_STEM) return launch_tn_cfg<T, 64, 64, MODE_STEM>(a, s);
    const bool wide_i = a.I > 64;
    if (mode == MODE_DENSE) {
        const bool wide_j = a.J > 64;
        if (wide_i && wide_j) return launch_tn_cfg<T, 128, 128, MODE_DENSE>(a, s);
        if (wide_j) return launch_tn_cfg<T, 64, 128, MODE_DENSE>(a, s);
        if (wide_i) return launch_tn_cfg<T, 128, 64, MODE_DENSE>(a, s);
        return launch_tn_cfg<T, 64, 64, MODE_DENSE>(a, s);
    }
    if (a.g.C % 64) return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: conv channels must be a multiple of 64");
    const bool wide_j = (a.g.C % 128) == 0;
    if (wide_i && wide_j) return launch_tn_cfg<T, 128, 128, MODE_CONV>(a, s);
    if (wide_j) return launch_tn_cfg<T, 64, 128, MODE_CONV>(a, s);
    if (wide_i) return launch_tn_cfg<T, 128, 64, MODE_CONV>(a, s);
    return launch_tn_cfg<T, 64, 64, MODE_CONV>(a, s);
}
template int launch_tn<float>(TNArgs<float>&, int, hipStream_t);
template int launch_tn<bf16>(TNArgs<bf16>&, int, hipStream_t);

}  // namespace rpe

// MFMA implicit-GEMM kernels for gfx950 (CDNA4, wave64): templates + configuration choice.  Instantiated per element type and
// staging mode in igemm_nt_*.hip / igemm_tn_*.hip (parallel translation units); argument checks and dispatch in igemm.hip.
//
//   nt_kernel : C[M][N] = gather(A)[M][K] * W[N][K]^T  (+bias, +addend, relu, BN partial stats, fused BN backward)
//               conv forward, conv data-gradient, Linear forward / data-gradient.
//   tn_kernel : D[I][J] = sum_m P[m][I] * gather(Q)[m][J]   (per-workgroup fp32 slabs summed in a fixed order by tn_reduce_kernel;
//               fp32 atomics only when the caller provides no slab workspace)
//               conv weight-gradient, Linear weight-gradient.
//
// Layout: activations NHWC (channels contiguous), weights [N][K] with K = (r, s, c) contiguous.
// Operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds through raw buffer descriptors: 32-bit lane offsets, the
// hardware range check supplies the zeros of padding taps and tails) into a 2..3-slot ring, one raw s_barrier per K step --
// since round 4 the 7x7 stem too, over a zero-bordered image (rounds 1-3 staged its two-pixel chunks through registers; what is
// left of that form in tn_kernel -- USE_DMA = false, stem_chunk -- is no longer instantiated).
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so the accumulator registers of a lane run
// along N (channels); the epilogue passes them once through LDS so that every lane owns 8 consecutive channels of one row.
#include <stdio.h>
#include <stdlib.h>

#pragma once
#include "igemm.h"

constexpr int kEpiDepthLean = 8;   // ... of the epilogues without the y operand (role 5, BN-backward mode 5)
constexpr int kEpiDepth = 4;   // epilogue operand prefetch distance of the fused data gradients, in steps (see nt_kernel, PIPE); 8 measured level

namespace rpe {

// LDS slot permutation of the [row][KCH x 16 B] staging tiles (slot = chunk ^ f(row)).
// KCH = 4 (64-B rows): f = {0,2,3,1}[(row>>2)&3].  ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS); with lane -> (row = l&15, chunk = l>>4) this f puts the four
// row-quads of every group on four different 16-byte slots.  KCH = 8 (128-B rows): f = row & 7 (T2 of the guide).
template <int KCH> __device__ __forceinline__ int nt_swz(int row, int chunk) {
    if (KCH == 4) return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
    return chunk ^ (row & 7);
}

// Bijective XCD remap: consecutive logical tiles share one XCD's L2 (blocks b, b+8 share an XCD).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

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

struct TagFirst { static constexpr bool value = true; };   // first K step of a tile: accumulators start from zero
struct TagNext { static constexpr bool value = false; };

// The wait in front of a ring barrier.  vmcnt(N): this wave's LDS-DMA pieces of the tile that is read next have landed (RAW).
// lgkmcnt(0): this wave's ds_reads of the tile just multiplied have RETURNED, not merely been issued -- the slot they read is
// re-filled by LDS-DMA right after the barrier (WAR).  hipcc leaves the last fragment reads of a K step in flight across
// the barrier (their lgkmcnt sits in front of the MFMAs that follow it); with several workgroups per CU queueing on the
// LDS while a second stream's kernel shares the CU such a read was occasionally served after the next tile's DMA write:
// a few per cent error in one 8-channel slab of a weight gradient, a few times per hundred launches (tools/repro_check.py).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// One LDS-DMA instruction (64 lanes x 16 B -> 1 KB of LDS at the uniform address `lds_addr`) issued from inline assembly.  tn_kernel uses
// this form, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler's waitcnt pass tracks the builtin as a write to LDS and, unable to
// see that ds_read_b64_tr_b16 (an intrinsic: no alias information) reads ANOTHER ring slot, put `s_waitcnt vmcnt(0)` in front of the first
// transposing read of every K step -- i.e. right behind the request for the next tile: rounds 1-3's ring kept no tile in flight while a
// wave multiplied, whatever its depth (found in round 4 in the ISA; nt_kernel's plain ds_read_b128 never drew that wait).  The asm form is
// invisible to that pass; the counted waits (wait_vmcnt) and the barriers of the ring are the synchronisation, as designed.
// M0 carries the LDS address (one wait state between its write and the DMA: s_nop).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_asm(const __amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
#pragma clang diagnostic pop

// -----------------------------------------------------------------------------------------------
// NT kernel.  Tile BM x BN = (64*WAVES_M) x BN, 2*WAVES_M waves of 64 x (BN/2), K-step = KCH 16-byte chunks per row.
//   <2, 128|64, 4, ., 3>: 128-row tile, 64-B K rows, 3-slot ring  (K < 1024, stem)
//   <2, 128|64, 8, ., 2>: 128-row tile, 128-B K rows, 2-slot ring (K >= 1024: half the barriers per FLOP)
//   <1, 64, 4, ., 3>    : 64x64 tile, 2 waves                     (few-row Linear layers)
// -----------------------------------------------------------------------------------------------
// ROLE selects the epilogue compiled into the kernel (see the epilogue): 0 conv forward in training, 1 conv data gradient,
// 2 Linear, 3 conv forward in inference, 4 split-K partial tile.
// BNM: the data gradient's fused BN-backward mode (NTArgs::bn_mode), a template parameter so that each launch carries one
// epilogue variant only (role 1; 0 elsewhere).
// (second launch-bound = minimum waves per SIMD: with the mode a constant hipcc hoisted the epilogue loads of mode 2 into 252
// VGPRs -- one workgroup per CU, 25 % slower)
// Measured and rejected in rounds 1-2, code removed in round 3: 256-row / 8-wave tiles, 8 waves on the 128 x 128
// tile, 128 x 256 / 8-wave tiles, non-temporal epilogue stores, 4- and 5-slot rings.
template <typename T, int WAVES_M, int BN, int KCH, int MODE, int NST, int ROLE, int BNM = 0>
__global__ __launch_bounds__(128 * WAVES_M, 2) void nt_kernel(const NTArgs<T> p) {
    constexpr int WAVES_N = 2;   // waves along N: 64 x BN/2 per wave
    constexpr int CE = Elem<T>::kChunk, BK = KCH * CE;
    constexpr int NW = WAVES_N * WAVES_M, NTHR = 64 * NW;
    constexpr int BM = 64 * WAVES_M, WM = 64, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
    static_assert(WN % 16 == 0 && WN >= 16, "a wave needs at least one 16-column fragment");
    constexpr int RPI = 64 / KCH;                       // rows covered by one 64-lane x 16-B DMA instruction
    constexpr int AR = BM / RPI / NW, BR = BN / RPI / NW;  // DMA instructions (= 16-B chunks per thread) per K-step
    static_assert((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0, "tile rows must split evenly over the waves");
    constexpr bool HALO = MODE == MODE_HALO;            // 3x3 / stride 1 / pad 1: activation operand as a resident halo patch (igemm.h)
    static_assert(!HALO || (KCH == 8 && CE == 8 && WAVES_M == 2), "halo form: 16-bit types, 64-channel chunks, 128-row tiles");
    constexpr int STAGE = HALO ? BN * KCH : (BM + BN) * KCH;   // 16-byte units (halo: the ring carries the weight tile only)
    constexpr int PATCH16 = HALO ? kHaloPatchPx * 8 : 0;       // halo patch region behind the ring
    constexpr int EPI16 = NW * 16 * (WN + 4) / 4;      // epilogue staging (NW waves x 16 rows x (WN+4) floats)
    // (MODE_STEM staged its operand through registers until round 4: two 8-byte pixels per chunk with separate bounds.  Over the
    // zero-bordered image every chunk is an in-bounds, aligned 16-byte read and the stem rides the LDS-DMA ring like the others.)
    constexpr bool DMA = true;
    constexpr int NSTAGE = DMA ? NST : 2;              // DMA path: ring of NST slots, NST-1 tiles in flight
    static_assert(NST >= 2 && NST <= (MODE == MODE_HALO ? 4 : 3), "ring depth 2..3 (halo form: ..4)");
    // T side product (NTArgs::t_a; BNM 6 / 7): behind the epilogue's staging rows the LDS holds the dz tile as stored (16-bit, [row][16
    // chunks], swizzled for the transposing reads) and 128 x 64 / 64 x 128 rows of the second operand; both alias the (dead) ring.
    constexpr bool TFUSE = ROLE == 1 && MODE == MODE_DENSE && (BNM == 6 || BNM == 7);
    constexpr int TP = BNM == 7 ? 128 : 64;                    // channels of t_a
    constexpr int T_DZ0 = 1280, T_A0 = T_DZ0 + 128 * 128 / 8, T_END = T_A0 + 1024;
    static_assert(!TFUSE || (BM == 128 && BN == 128 && CE == 8 && EPI16 <= T_DZ0), "T side product: 128 x 128 tiles of a 16-bit type");
    static_assert(!TFUSE || NSTAGE * STAGE <= T_A0, "T side product: the second operand's LDS image must lie behind the operand ring (it is filled while the ring runs)");
    constexpr int LDS16 = (NSTAGE * STAGE + PATCH16 > EPI16) ? NSTAGE * STAGE + PATCH16 : EPI16;
    __shared__ u32x4 lds[(TFUSE && T_END > LDS16) ? T_END : LDS16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    f32x4 tacc[TFUSE ? 2 : 1][TFUSE ? TP / 16 : 1];   // T partial of this workgroup: rows (2 wave + a) * 16.., all TP columns
    if constexpr (TFUSE) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TP / 16; ++b) tacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int n_iter = TFUSE ? p.tiles_per_wg : 1;   // (persistent only with the side product: its partial lives in registers across tiles)
    for (int it = 0; it < n_iter; ++it) {
    int lb_, tn_, tm_;
    if constexpr (TFUSE) {
        const int per = (int)gridDim.x / p.tiles_n;
        tn_ = (int)blockIdx.x % p.tiles_n;
        tm_ = (int)blockIdx.x / p.tiles_n + it * per;
        if (tm_ >= p.tiles_m) break;
        lb_ = tm_ * p.tiles_n + tn_;
    } else {
        lb_ = xcd_remap_dir(blockIdx.x, p.tiles_m * p.tiles_n, p.rev);
        tn_ = lb_ % p.tiles_n; tm_ = lb_ / p.tiles_n;
    }
    const int lb = lb_, tile_n = tn_, tile_m = tm_;
    const int n0 = tile_n * BN;
    const Gather& g = p.g;
    // Row space.  Normal: row m = m0 + local.  Parity mode (stride-2 dgrad): tile_m = tq * 4 + cls (classes interleaved:
    // they cost 1 : 2 : 2 : 4 taps, and the XCD remap hands each XCD a contiguous tile range -- class-major order left two
    // XCDs with all the work); local rows index the class-local pixel list (b, h', w') -> output pixel (b, 2h'+ph, 2w'+pw).
    int cls = 0, ph = 0, pw = 0, m0 = tile_m * BM, row_lim = p.M;
    if (MODE == MODE_CONV && g.parity) {
        cls = tile_m & 3;
        m0 = (tile_m >> 2) * BM;
        ph = cls >> 1; pw = cls & 1;
        row_lim = g.rows_q;
    }
    if (HALO) {   // a tile = halo_rt whole output rows; its tail rows (halo_px .. 127) are masked like rows past M
        m0 = tile_m * g.halo_px;
        row_lim = m0 + g.halo_px < p.M ? m0 + g.halo_px : p.M;
    }
    // output row (for C / addend / BN operands) of tile-local row index `lr`, or -1 when outside the problem
    auto out_row = [&](int lr) -> long {
        const int m = m0 + lr;
        if (m >= row_lim) return -1;
        if (!(MODE == MODE_CONV && g.parity)) return m;
        const unsigned b = fd_div((unsigned)m, g.div_hw);
        const unsigned rem = (unsigned)m - b * g.div_hw.d;
        const unsigned hh = fd_div(rem, g.div_w);
        const unsigned ww = rem - hh * g.div_w.d;
        return ((long)b * (2 * g.Ho) + (2 * hh + ph)) * (2 * g.Wo) + (2 * ww + pw);
    };
    // Staging.  DMA path (dense / conv): one buffer_load_dwordx4 ... lds writes 64 lanes x 16 B = RPI rows x KCH slots straight
    // into LDS (no VGPR round trip, no ds_write).  The LDS image is lane-linear, so the slot swizzle is applied on the
    // SOURCE side: the lane that fills slot s of row r fetches logical chunk s ^ f(r).  Padding taps / tails use an offset
    // beyond the descriptor's range.  Stem path (two 8-byte pixels per chunk with separate bounds) keeps register staging.
    int a_row[AR], a_chunk[AR], b_row[BR], b_chunk[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        a_row[i] = (wave * AR + i) * RPI + lane / KCH; a_chunk[i] = nt_swz<KCH>(a_row[i], lane % KCH);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        b_row[i] = (wave * BR + i) * RPI + lane / KCH; b_chunk[i] = nt_swz<KCH>(b_row[i], lane % KCH);
    }

    // DMA path addressing: buffer_load ... lds through two raw buffer descriptors (A, weights).  Per lane and chunk a 32-bit
    // BYTE offset (voffset), per K-step a uniform byte offset in an SGPR (soffset): the K loop itself then carries no
    // per-lane address arithmetic (dense) or only a per-TAP refresh (conv).  Rows / taps / K tails that must read as zero
    // get voffset = 0x80000000: beyond num_records (< 2 GiB, checked by the launcher), the hardware range check returns 0.
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ES = (int)sizeof(T);
    // The A descriptors start at THIS tile's first row (dense) / first image (conv), so the 32-bit offsets span one tile and a
    // tensor may be of any size (round 1 capped a tensor at 2 GiB = ~1300 images in bf16); num_records = the bytes from there to
    // the end of the tensor, clamped below 2^31 (a tile never reaches that far, and rows past M are masked by a_ok anyway).
    const unsigned tile_b0 = (MODE == MODE_CONV || MODE == MODE_STEM) ? (unsigned)__builtin_amdgcn_readfirstlane((int)fd_div((unsigned)(m0 < row_lim ? m0 : 0), g.div_hw))
                             : HALO ? (unsigned)__builtin_amdgcn_readfirstlane((int)fd_div((unsigned)(tile_m * g.halo_rt), g.div_h)) : 0u;
    const long tile_a = (MODE == MODE_DENSE) ? (long)m0 * p.lda : (long)tile_b0 * g.img_stride;     // elements
    unsigned a_voff[AR], b_voff[BR], a_vbase[AR];
    long a_base[AR];
    int a_hb[AR], a_wb[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + a_row[i];
        a_ok[i] = m < row_lim;
        a_hb[i] = a_wb[i] = 0;
        a_voff[i] = OOB; a_vbase[i] = 0;
        if (MODE == MODE_DENSE) {
            a_base[i] = (long)m * p.lda + a_chunk[i] * CE;
            if (a_ok[i]) a_voff[i] = (unsigned)(((long)a_row[i] * p.lda + a_chunk[i] * CE) * ES);   // relative to the tile's first row (tile_a)
        } else {
            const unsigned mm = a_ok[i] ? (unsigned)m : (unsigned)m0;
            const unsigned b = fd_div(mm, g.div_hw);
            const unsigned rem = mm - b * g.div_hw.d;
            unsigned oh = fd_div(rem, g.div_w);
            unsigned ow = rem - oh * g.div_w.d;
            if (MODE == MODE_CONV && g.parity) { oh = 2 * oh + ph; ow = 2 * ow + pw; }
            a_base[i] = (long)b * g.img_stride;
            a_vbase[i] = (unsigned)(((long)(b - tile_b0) * g.img_stride + a_chunk[i] * CE) * ES);   // relative to the tile's first image
            a_hb[i] = (int)oh * g.sn + g.base_h;
            a_wb[i] = (int)ow * g.sn + g.base_w;
            // stem over the zero-bordered NHWC4 image (g.H, g.W = the padded dims): row m's K row r is the 8 pixels (2 oh + r, 2 ow ..+7) --
            // 32 contiguous elements; the tap row advances through the SGPR offset (dma_tile)
            if (MODE == MODE_STEM && a_ok[i]) a_voff[i] = a_vbase[i] + (unsigned)((((long)oh * 2 * g.W + (long)ow * 2) * 4) * ES);
        }
    }
    constexpr bool KCAT = MODE == MODE_DENSE && ROLE == 1;   // a second A tensor for k >= K1 (NTArgs::A2)
    unsigned a_voff2[AR];
    if (KCAT) {
#pragma unroll
        for (int i = 0; i < AR; ++i) a_voff2[i] = (a_ok[i] && p.A2) ? (unsigned)(((long)a_row[i] * p.lda2 + a_chunk[i] * CE) * ES) : OOB;
    }
    long b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + b_row[i];
        b_ok[i] = n < p.N;
        b_off[i] = (long)n * p.ldb + b_chunk[i] * CE;
        b_voff[i] = b_ok[i] ? (unsigned)(b_off[i] * ES) : OOB;
    }

    // uniform K-walk state for MODE_CONV: k = (r*S + s)*C + c0.  Parity mode visits taps r0, r0+2, .. x s0, s0+2, ..
    int kbase = 0, c0 = 0, tr = 0, ts = 0, tstep = 1, s_first = 0;
    int nk = (p.K + BK - 1) / BK;
    if (MODE == MODE_CONV && g.parity) {
        const int r0 = (ph + g.base_h) & 1, s0 = (pw + g.base_w) & 1;
        const int nr = r0 < g.R ? (g.R - r0 + 1) / 2 : 0, ns = s0 < g.S ? (g.S - s0 + 1) / 2 : 0;
        tr = r0; ts = s0; s_first = s0; tstep = 2;
        kbase = (tr * g.S + ts) * g.C;
        nk = nr * ns * (g.C / BK);
    }
    if (ROLE == 4) {   // split-K partial: this workgroup's slice of the K steps (never with parity classes: forward only)
        const int k0 = (int)blockIdx.y * p.split_steps;
        nk = nk - k0 < p.split_steps ? nk - k0 : p.split_steps;
        kbase = k0 * BK;
        if (MODE == MODE_CONV) {
            const int tap = kbase / g.C;
            c0 = kbase - tap * g.C;
            tr = tap / g.S;
            ts = tap - tr * g.S;
        }
    }

    // conv: voffsets of the A chunks for the current tap (tr, ts); refreshed only when the tap changes
    auto tap_offsets = [&]() {
        const int msk = (1 << g.sd_shift) - 1;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int nh = a_hb[i] + g.tap_sign * tr, nw = a_wb[i] + g.tap_sign * ts;
            const int ih = nh >> g.sd_shift, iw = nw >> g.sd_shift;
            const bool ok = a_ok[i] && nh >= 0 && nw >= 0 && ((nh | nw) & msk) == 0 && ih < g.H && iw < g.W;
            a_voff[i] = ok ? a_vbase[i] + (unsigned)((ih * g.W + iw) * g.C) * ES : OOB;
        }
    };
    bool tap_dirty = true;
    auto advance_k = [&]() {
        kbase += BK;
        if (MODE == MODE_CONV) {
            c0 += BK;
            if (c0 >= g.C) {
                c0 = 0;
                ts += tstep;
                if (ts >= g.S) { ts = s_first; tr += tstep; }
                if (tstep != 1) kbase = (tr * g.S + ts) * g.C;
                tap_dirty = true;
            }
        }
    };
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void lds_void;
    auto clamp31 = [](long bytes) -> int { return (int)(bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes)); };
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + tile_a), 0, clamp31((p.a_elems - tile_a) * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bw, 0, (int)p.b_bytes, 0x00020000);
    const long tile_a2 = (long)m0 * p.lda2;
    const __amdgpu_buffer_rsrc_t rs_a2 = (KCAT && p.A2) ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + tile_a2), 0, clamp31(((long)p.M * p.lda2 - tile_a2) * ES), 0x00020000) : rs_a;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably uniform: the LDS base of a DMA goes through M0
    const bool ktail = (p.K % BK) != 0;                        // dense only (conv: C % BK == 0, checked by the launcher)
    // ---- T side product: operand staging --------------------------------------------------------
    constexpr int T_CPR = TP / 8, T_ROWS = TP == 64 ? 128 : 64;   // 16-byte chunks per row of t_a; rows per staged pass (16 KB)
    auto t_swz16 = [](int row) -> int { return ((row & 3) | ((row >> 1) & 4)) << 1; };
    auto t_swz = [&](int row) -> int { return T_CPR >= 16 ? t_swz16(row) : ((((row >> 1) & 1) | ((row >> 2) & 2)) << 1); };
    const __amdgpu_buffer_rsrc_t rs_t = TFUSE ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.t_a + (long)m0 * TP), 0, clamp31(((long)p.M - m0) * TP * ES), 0x00020000) : rs_a;
    auto t_stage = [&](int row_base) {   // rows row_base .. row_base + T_ROWS of this tile's t_a rows -> LDS [row][T_CPR chunks], 4 LDS-DMA instructions per wave
        if constexpr (TFUSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sl = (wave_u * 4 + i) * 64 + lane, row = sl / T_CPR, slot = sl % T_CPR;
                const unsigned vo = (unsigned)(((row_base + row) * TP + (slot ^ t_swz(row)) * 8) * ES);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_void*)((lds_char*)lds + T_A0 * 16 + (wave_u * 4 + i) * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
    };
    // requested BEFORE the operand ring's first tiles (its region lies behind the ring): vmcnt completes in issue order, so the ring's
    // counted waits cover it and it has landed when the K loop ends -- no wait on the epilogue's stores later (rows past M lie outside
    // the descriptor: zeros)
    if constexpr (TFUSE) t_stage(0);
    auto dma_tile = [&](int st) {
        lds_char* base = (lds_char*)lds + st * (STAGE * 16);
        if (MODE == MODE_CONV && tap_dirty) { tap_offsets(); tap_dirty = false; }
        const int so_a = (MODE == MODE_CONV ? c0 : MODE == MODE_STEM ? (kbase >> 5) * g.W * 4 + (kbase & 31) : kbase) * ES, so_b = kbase * ES;
        if (KCAT && p.A2 && kbase >= p.K1) {   // (uniform: the K step lies in the second tensor)
            const int so_a2 = (kbase - p.K1) * ES;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                unsigned vo = a_voff2[i];
                if (ktail && kbase + a_chunk[i] * CE >= p.K) vo = OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_void*)(base + (wave_u * AR + i) * 1024), 16, (int)vo, so_a2, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                unsigned vo = a_voff[i];
                if (MODE == MODE_DENSE && ktail && kbase + a_chunk[i] * CE >= p.K) vo = OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(base + (wave_u * AR + i) * 1024), 16, (int)vo, so_a, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            unsigned vo = b_voff[i];
            if (ktail && kbase + b_chunk[i] * CE >= p.K) vo = OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(base + BM * KCH * 16 + (wave_u * BR + i) * 1024), 16, (int)vo, so_b, 0, 0);
        }
    };
    // ---- epilogue lane geometry (needed early: the data-gradient form prefetches its epilogue operands) ----
    constexpr int LDW = WN + 4;     // staged row pitch in floats (+4: conflict-free float4 writes)
    constexpr int CPW = WN / 8;     // 8-channel chunks per staged row
    constexpr int RPP = 64 / CPW;   // rows per pass
    constexpr int NPASS = 16 / RPP;
    constexpr int NSTEP = FM * NPASS;   // epilogue steps of a wave: (fragment, pass) pairs, 8 channels of one row per lane each
    const int erow = lane / CPW, echk = lane % CPW;
    const int nl = wave_n * WN + echk * 8;
    const int n = n0 + nl;
    const bool ncol_ok = n < p.N;
    // Epilogue features by ROLE, resolved at compile time (the generic epilogue is ~6000 instructions of mostly untaken
    // branches, and the short-K launches are instruction-issue bound in it): conv launches (roles 0, 1) always move whole
    // 16-byte chunks (N % 8 == 0, aligned rows: checked by the launcher), only the data gradient has the BN modes, only the
    // forward / Linear roles have bias and ReLU.
    // Roles: 0 conv forward in training (batch-statistics partial sums only), 1 conv data gradient (addend, BN modes),
    // 2 Linear (bias / addend / ReLU, any alignment), 3 conv forward in inference (bias = BN shift, addend = identity, ReLU).
    // 5 1x1 conv forward in training with the BatchNorm apply (+ residual [under its own BN]) + ReLU + packed mask fused (statistics known
    // before the launch: Gram matrix of the input).
    constexpr bool VEC_ONLY = ROLE != 2, HAS_BN = ROLE == 1, HAS_AFFINE = ROLE >= 2 && ROLE != 5, HAS_ADDEND = ROLE != 0, HAS_FWD_STATS = ROLE == 0;
    constexpr bool FWD_BN = ROLE == 5;
    const bool nfull = VEC_ONLY ? ncol_ok : n + 7 < p.N;
    const bool vec_c = VEC_ONLY ? ncol_ok : nfull && (p.ldc % CE == 0) && (((uintptr_t)p.C) & 15) == 0;
    const T* const e_addend = HAS_ADDEND ? p.addend : nullptr;
    const bool vec_add = VEC_ONLY ? (ncol_ok && e_addend != nullptr) : nfull && e_addend && (p.ld_add % CE == 0) && (((uintptr_t)e_addend) & 15) == 0;
    constexpr int bn_mode = HAS_BN ? (TFUSE ? 5 : BNM) : 0;   // (the side-product modes run mode 5's epilogue)
    const float* const e_bias = (HAS_AFFINE || ROLE == 1) ? p.bias : nullptr;   // (role 1: the constant term of a folded BN backward)
    const int e_relu = HAS_AFFINE ? p.relu : 0;
    // Data-gradient launches with short K are bound by the epilogue's operand stream (residual gradient, y, a_out: up to
    // three reads and one write per output element against K/N-th of that for the GEMM operands).  One step at a time
    // keeps ~1-3 KB per wave in flight; here the operands of the next DEPTH steps are requested ahead (the first DEPTH
    // before the K loop even starts), 16 B per lane and operand, into registers that are recycled step by step.
    constexpr bool PIPE = (ROLE == 1 || ROLE == 5) && CE == 8;
    // the y3-free epilogues -- role 5 and BN-backward mode 5 -- read one operand stream less and have the registers for a deeper prefetch
    // (8 steps: 154 / 141 VGPRs, still three workgroups per CU; 18.96-19.09 vs 19.00-19.14 ms/step on one box, profiles/r04_ab_epi_depth.txt)
    constexpr int kDepthHere = (ROLE == 5 || (ROLE == 1 && BNM == 5)) ? kEpiDepthLean : kEpiDepth;
    constexpr int DEPTH = PIPE ? (NSTEP < kDepthHere ? NSTEP : kDepthHere) : 1;
    u32x4 qd[DEPTH], qy[DEPTH], qa[DEPTH];
    auto step_row = [&](int t) -> long { return out_row(wave_m * WM + (t / NPASS) * 16 + (t % NPASS) * RPP + erow); };
    auto issue = [&](int t) {
        if (!PIPE) return;
        const long m = step_row(t);
        const int sl = t % DEPTH;
        if (m < 0 || !ncol_ok) return;
        if (vec_add) qd[sl] = ld16(e_addend + m * p.ld_add + n);
        if (bn_mode && vec_c) {
            if (bn_mode != 5) qy[sl] = ld16(p.bn_y + m * p.ldc + n);
            if (bn_mode == 1) qa[sl] = ld16(p.bn_a + m * p.ldc + n);
            if (bn_mode == 4 || bn_mode == 5) qa[sl].x = p.bn_mask[(m * p.ldc + n) >> 3];   // one byte: the ReLU bits of this lane's 8 channels
        }
    };
    if (PIPE) {
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) issue(t);
    }

    f32x4 acc[FN][FM];   // started by the first K step's MFMAs from an inline-constant zero (no 64-register clear per tile)

    const int fr = lane & 15, fc = lane >> 4;
    auto compute = [&](int st, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const u32x4* base = lds + st * STAGE;
#pragma unroll
        for (int ks = 0; ks < KCH / 4; ++ks) {
            u32x4 af[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) { const int row = wave_m * WM + i * 16 + fr; af[i] = base[row * KCH + nt_swz<KCH>(row, ks * 4 + fc)]; }
#pragma unroll
            for (int i = 0; i < FN; ++i) { const int row = wave_n * WN + i * 16 + fr; wf[i] = base[BM * KCH + row * KCH + nt_swz<KCH>(row, ks * 4 + fc)]; }
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    if (FIRST && ks == 0) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                    Mma<T>::run(wf[a], af[b], acc[a][b]);
                }
        }
    };
    if (nk <= 0) {   // (parity classes of a strided 1x1 data gradient have no tap at all: the tile is the epilogue's addend only)
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (HALO) {
        // ---- halo form -----------------------------------------------------------------------------
        // Patch = padded input rows P0 .. P1 (P(b, h) = b*(H+2) + h + 1: one zero row above and below every image) x W + 2 pixels
        // (one zero column left and right) x 64 channels = 128 B per pixel, LDS image [pixel q][8 slots of 16 B], slot = chunk ^ (q & 7).
        // Output pixel (b, h, w) sits at patch pixel qc = (P(b, h) - P0) * (W+2) + w + 1; tap (r, s) reads pixel qc + dq with the
        // UNIFORM dq = (base_h + sign*r) * (W+2) + base_w + sign*s  (forward: base -1, sign +1; data gradient: base +1, sign -1).
        // K order: chunk-major, tap-minor -- the weight column of step (chunk c, tap t) is t*C + 64 c.
        constexpr int NI = BR;                     // LDS-DMA instructions per wave and weight tile
        constexpr int PF = NSTAGE - 1;
        constexpr int PNI = kHaloPatchPx / 8 / NW;  // patch instructions per wave (8 pixels each)
        const int H = g.H, W = g.W, PW = W + 2;
        const int gr0 = tile_m * g.halo_rt;         // first output row (global row index b*H + h) of the tile
        const int h0 = gr0 - (int)tile_b0 * H;
        const int P0 = (int)tile_b0 * (H + 2) + h0;
        int gr1 = gr0 + g.halo_rt; if (gr1 > g.halo_rows) gr1 = g.halo_rows; gr1 -= 1;
        const int b1 = (int)fd_div((unsigned)gr1, g.div_h);
        const int npx = (b1 * (H + 2) + (gr1 - b1 * H) + 2 - P0 + 1) * PW;
        unsigned pv[PNI];
#pragma unroll
        for (int j = 0; j < PNI; ++j) {
            const int q = (j * NW + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (q & 7);
            const int pr = (int)fd_div((unsigned)q, g.div_pw), pc = q - pr * PW;
            const int PR = P0 + pr;
            const int b = (int)fd_div((unsigned)PR, g.div_hp), hp = PR - b * (H + 2);
            const bool ok = q < npx && hp >= 1 && hp <= H && pc >= 1 && pc <= W;
            pv[j] = ok ? (unsigned)(((long)(b - (int)tile_b0) * g.img_stride + (long)((hp - 1) * W + (pc - 1)) * g.C + chunk * CE) * ES) : OOB;
        }
        int qc[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int lr = wave_m * WM + i * 16 + (lane & 15);
            const int lrc = (m0 + lr < row_lim) ? lr : 0;
            const int rr = (int)fd_div((unsigned)lrc, g.div_w), w = lrc - rr * W;
            const int b = (int)fd_div((unsigned)(gr0 + rr), g.div_h);
            qc[i] = (rr + 2 * (b - (int)tile_b0) + 1) * PW + w + 1;
        }
        nk = 9 * (g.C / BK);
        lds_char* const patch = (lds_char*)lds + NSTAGE * (STAGE * 16);
        int i_tap = 0, i_c0 = 0;   // issue side of the weight ring
        auto dma_w = [&](int st) {
            lds_char* base = (lds_char*)lds + st * (STAGE * 16);
            const int so_b = (i_tap * g.C + i_c0) * ES;
#pragma unroll
            for (int i = 0; i < BR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(base + (wave_u * BR + i) * 1024), 16, (int)b_voff[i], so_b, 0, 0);
            if (++i_tap == 9) { i_tap = 0; i_c0 += BK; }
        };
        auto stage_patch = [&](int c0h) {
#pragma unroll
            for (int j = 0; j < PNI; ++j) {
                const int t = j * NW + wave_u;
                if (t * 8 < npx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(patch + t * 1024), 16, (int)pv[j], c0h * ES, 0, 0);
            }
        };
        const u32x4* const pbase = lds + NSTAGE * STAGE;
        const int hfr = lane & 15, hfc = lane >> 4;
        auto compute_halo = [&](int st, int dq, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const u32x4* wbase = lds + st * STAGE;
#pragma unroll
            for (int ks = 0; ks < KCH / 4; ++ks) {
                u32x4 af[FM], wf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) { const int q = qc[i] + dq; af[i] = pbase[q * 8 + ((ks * 4 + hfc) ^ (q & 7))]; }
#pragma unroll
                for (int i = 0; i < FN; ++i) { const int row = wave_n * WN + i * 16 + hfr; wf[i] = wbase[row * KCH + nt_swz<KCH>(row, ks * 4 + hfc)]; }
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) {
                        if (FIRST && ks == 0) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                        Mma<T>::run(wf[a], af[b], acc[a][b]);
                    }
            }
        };
        stage_patch(0);
#pragma unroll
        for (int t = 0; t < PF; ++t) dma_w(t);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int st = 0, c_tap = 0, c_c0 = 0, c_tr = 0, c_ts = 0;
        auto k_step = [&](int kt, auto first_tag) {
            if (kt + PF < nk) { int s2 = st + PF; if (s2 >= NSTAGE) s2 -= NSTAGE; dma_w(s2); }
            const int dq = (g.base_h + g.tap_sign * c_tr) * PW + g.base_w + g.tap_sign * c_ts;
            compute_halo(st, dq, first_tag);
            if (++c_ts == 3) { c_ts = 0; ++c_tr; }
            if (++c_tap == 9 && kt + 1 < nk) {
                // next 64-channel chunk: every wave is done with the patch (WAR), then it is re-filled; the weight tiles in flight land with it
                c_tap = 0; c_tr = 0; c_c0 += BK;
                wait_vmcnt<3 * NI>();
                __builtin_amdgcn_s_barrier();
                stage_patch(c_c0);
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            } else {
                int newer = nk - 2 - kt;
                if (newer > PF - 1) newer = PF - 1;
                if (newer >= 3) wait_vmcnt<3 * NI>(); else if (newer == 2) wait_vmcnt<2 * NI>(); else if (newer == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (++st == NSTAGE) st = 0;
        };
        k_step(0, TagFirst{});
        for (int kt = 1; kt < nk; ++kt) k_step(kt, TagNext{});
    } else {
        // 3-slot ring, tiles kt+1 and kt+2 in flight while tile kt is multiplied.  A tile is NI LDS-DMA instructions per
        // wave; vmcnt counts them in issue order, so "all but the newest NI landed" == tile kt+1 is complete.  The raw
        // s_barrier (not __syncthreads, which would drain vmcnt to 0) then publishes it to the other waves; slot (kt+2)%3
        // was last read in iteration kt-1, i.e. before the barrier every wave has already passed.
        constexpr int NI = AR + BR;
        constexpr int PF = NSTAGE - 1;  // tiles kept in flight ahead of the one being multiplied
        // a reduction of <= NSTAGE steps never re-uses a slot: all of it is requested up front (the 2-slot / 32-KB form of the K = 64
        // launches, four workgroups per CU instead of three, lives on this)
        const bool allin = nk <= NSTAGE;
        const int pf = allin ? nk : PF;
        // prologue: tiles 0 .. pf-1
#pragma unroll
        for (int t = 0; t < NSTAGE; ++t) {
            if (t < pf) { if (t > 0) advance_k(); dma_tile(t); }
        }
        // tile 0 must have landed: allow the (pf - 1) newer tiles to stay in flight
        {
            const int newer = (nk < pf ? nk : pf) - 1;
            if (newer >= 3) wait_vmcnt<3 * NI>(); else if (newer == 2) wait_vmcnt<2 * NI>(); else if (newer == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        int st = 0;
        auto k_step = [&](int kt, auto first_tag) {
            const bool pre = !allin && kt + PF < nk;
            if (pre) { advance_k(); int s2 = st + PF; if (s2 >= NSTAGE) s2 -= NSTAGE; dma_tile(s2); }
            compute(st, first_tag);
            // tile kt+1 must be complete; tiles kt+2 .. may still be in flight
            int newer = nk - 2 - kt;               // tiles issued after kt+1
            if (newer > pf - 1) newer = pf - 1;
            if (newer >= 3) wait_vmcnt<3 * NI>(); else if (newer == 2) wait_vmcnt<2 * NI>(); else if (newer == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (++st == NSTAGE) st = 0;
        };
        if (nk > 0) k_step(0, TagFirst{});
        for (int kt = 1; kt < nk; ++kt) k_step(kt, TagNext{});
    }

    if constexpr (ROLE == 4) {
        // split-K partial: raw accumulators in fragment order, 1 KB per wave store; nt_split_epilogue_kernel finishes the tile
        float* dst = p.slab + (((size_t)blockIdx.y * ((size_t)p.tiles_m * p.tiles_n) + lb) * NW + wave) * (size_t)(FN * FM * 256);
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) *(f32x4*)(dst + ((a * FM + b) * 64 + lane) * 4) = acc[a][b];
        return;
    }
    // ---- epilogue ---------------------------------------------------------------------------------
    // A lane's accumulators hold 4 channels of 16 scattered rows.  They go through LDS once so that every lane
    // ends up with 8 consecutive channels of ONE row: epilogue operands (addend, and for the fused BN-backward
    // form y / a_out) are then read, and the result written, as 16 bytes per lane = whole 128-B lines per 8 lanes.
    float* stg = (float*)lds + wave * (16 * LDW);  // 16 staged rows (one 16-row fragment) per wave at a time
    float cs[8], cq[8], cmean[8], cinv[8], csc[8], csh[8], cbias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; cmean[j] = 0.f; cinv[j] = 0.f; csc[j] = 0.f; csh[j] = 0.f; cbias[j] = 0.f; }
    // Per-column coefficients.  Conv roles (whole 8-channel chunks: ncol_ok covers all eight) read them under ONE uniform test per
    // array, so that all loads are requested before the first is used: with the test per element (rounds 1-3) every element was a
    // basic block of its own and the compiler put `s_waitcnt vmcnt(0)` behind each -- eight dependent round trips (~5 us) between the
    // K loop and the epilogue of every tile, found in the ISA in round 4.
    if (VEC_ONLY ? ncol_ok : true) {
        float tm[8];
        if (e_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (VEC_ONLY || n + j < p.N) cbias[j] = e_bias[n + j];
        }
        if (bn_mode && bn_mode != 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { tm[j] = 0.f; if (VEC_ONLY || n + j < p.N) { cinv[j] = p.bn_invstd[n + j]; tm[j] = p.bn_mean[n + j]; } }
#pragma unroll
            for (int j = 0; j < 8; ++j) cmean[j] = -tm[j] * cinv[j];   // xhat = y*inv + (-mean*inv): one fma
        }
        if (bn_mode == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (VEC_ONLY || n + j < p.N) { csc[j] = p.bn_scale[n + j]; csh[j] = p.bn_shift[n + j]; }
        }
    }
    float fsc[8], fsh[8], frs[8];   // role 5: BN scale / shift of this layer (the shift carries the residual's shift) and the residual's scale
    if constexpr (FWD_BN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fsc[j] = 0.f; fsh[j] = 0.f; frs[j] = 1.f; }
        if (ncol_ok) {   // (role 5 moves whole chunks)
            float t0[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { fsc[j] = p.fwd_scale[n + j]; fsh[j] = p.fwd_shift[n + j]; t0[j] = 0.f; }
            if (p.res_shift) {
#pragma unroll
                for (int j = 0; j < 8; ++j) t0[j] = p.res_shift[n + j];
            }
            if (p.res_scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) frs[j] = p.res_scale[n + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) fsh[j] += t0[j];
        }
    }
    auto load8 = [&](const T* base, long off, bool vec, float* out) {
        if (VEC_ONLY || vec) {
            if (CE == 8) { chunk_to_f<T>(ld16(base + off), out); }
            else { chunk_to_f<T>(ld16(base + off), out); chunk_to_f<T>(ld16(base + off + 4), out + 4); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = (n + j < p.N) ? Elem<T>::to_f(base[off + j]) : 0.f;
        }
    };
    // the staging rows are private to a wave and LDS executes a wave's accesses in order: a wave-level fence between its
    // writes and its cross-lane reads is enough; the workgroup barrier is only needed once, after the K loop
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    __syncthreads();   // every wave is done with the K loop's LDS tiles
#pragma unroll
    for (int qf = 0; qf < FM; ++qf) {
        wave_sync();   // this wave's reads of the previous fragment have been issued (LDS is in order per wave)
#pragma unroll
        for (int a = 0; a < FN; ++a) *(f32x4*)(stg + fr * LDW + a * 16 + fc * 4) = acc[a][qf];
        wave_sync();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + erow;
            const int t = qf * NPASS + ps, sl = t % DEPTH;
            const long m = out_row(wave_m * WM + qf * 16 + r);
            float v[8];
            {
                const f32x4 t0 = *(const f32x4*)(stg + r * LDW + echk * 8);
                const f32x4 t1 = *(const f32x4*)(stg + r * LDW + echk * 8 + 4);
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            }
            if (m >= 0 && ncol_ok) {
                if (HAS_FWD_STATS && p.stats_part) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { cs[j] += v[j]; cq[j] += v[j] * v[j]; }
                }
                if (e_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += cbias[j];
                }
                if constexpr (FWD_BN) {
                    if (p.y_out) *(u32x4*)(p.y_out + m * p.ldc + n) = f_to_chunk<T>(v);   // (16-bit element types only: CE == 8)
                    float ad[8];
                    if (e_addend) {
                        if constexpr (PIPE) chunk_to_f<T>(qd[sl], ad);   // (PIPE roles move whole chunks: inside this guard vec_add holds)
                        else load8(e_addend, m * p.ld_add + n, vec_add, ad);
                    }
                    unsigned bits = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float t = fmaf(v[j], fsc[j], fsh[j]);       // same expression order as bn_apply_kernel
                        if (e_addend) t = fmaf(ad[j], frs[j], t);
                        bits |= (t > 0.f ? 1u : 0u) << j;
                        v[j] = fmaxf(t, 0.f);
                    }
                    if (p.mask_out) p.mask_out[(m * p.ldc + n) >> 3] = (unsigned char)bits;
                } else
                if (e_addend) {
                    float ad[8];
                    // (PIPE roles move whole chunks: inside the m / ncol_ok guard vec_add and vec_c hold, so the prefetched registers are THE
                    // operands.  With the run-time fallback `else load8(..)` compiled beside them the wait in front of the first use had to cover a
                    // load that might just have been issued: vmcnt(0) in every step, which drained the whole prefetch -- round 4, from the ISA.)
                    if constexpr (PIPE) chunk_to_f<T>(qd[sl], ad);
                    else load8(e_addend, m * p.ld_add + n, vec_add, ad);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ad[j];
                }
                if (bn_mode) {
                    // v = dA (gradient wrt the BN output after ReLU).  dz = dA * [a_out > 0]; partial sums of dz and dz*xhat.
                    float yy[8], aa[8];
                    unsigned mbits = 0;
                    if constexpr (PIPE) {
                        if (bn_mode != 5) chunk_to_f<T>(qy[sl], yy);
                        if (bn_mode == 1) chunk_to_f<T>(qa[sl], aa);
                        if (bn_mode == 4 || bn_mode == 5) mbits = qa[sl].x;
                    } else {
                        if (bn_mode != 5) load8(p.bn_y, m * p.ldc + n, vec_c, yy);
                        if (bn_mode == 1) load8(p.bn_a, m * p.ldc + n, vec_c, aa);
                        if (bn_mode == 4 || bn_mode == 5) mbits = p.bn_mask[(m * p.ldc + n) >> 3];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool on = bn_mode == 1 ? (aa[j] > 0.f)
                                        : bn_mode == 2 ? (fmaf(yy[j], csc[j], csh[j]) > 0.f)
                                        : (bn_mode == 4 || bn_mode == 5) ? ((mbits >> j) & 1u) != 0 : true;
                        const float dz = on ? v[j] : 0.f;
                        if (bn_mode != 5) { cs[j] += dz; cq[j] = fmaf(dz, fmaf(yy[j], cinv[j], cmean[j]), cq[j]); }   // (mode 5: the second sum stays 0)
                        v[j] = dz;
                    }
                    if (bn_mode == 5) {
                        // sum of the dz values AS STORED (rounded to T): the other half of this BatchNorm's reduction, sum dz*y, is formed
                        // from the stored tensor (T = dz^T a, rpe_bn_backward_coeffs_t), and sum dz*xhat = invstd (sum dz*y - mean sum dz)
                        // cancels: with the unrounded sum here the two halves disagreed by the rounding noise times mean / std
                        float vr[8];
                        chunk_to_f<T>(f_to_chunk<T>(v), vr);
#pragma unroll
                        for (int j = 0; j < 8; ++j) cs[j] += vr[j];
                    }
                }
                if (e_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                T* cp = p.C + m * p.ldc + n;
                if (VEC_ONLY || vec_c) {
                    if (CE == 8) { *(u32x4*)cp = f_to_chunk<T>(v); }
                    else { *(u32x4*)cp = f_to_chunk<T>(v); *(u32x4*)(cp + 4) = f_to_chunk<T>(v + 4); }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (n + j < p.N) cp[j] = Elem<T>::from_f(v[j]);
                }
                if constexpr (TFUSE) {   // the same 16 bytes into the LDS image of the tile
                    const int lrow = wave_m * WM + qf * 16 + r;
                    lds[T_DZ0 + lrow * 16 + ((wave_n * (WN / 8) + echk) ^ t_swz16(lrow))] = f_to_chunk<T>(v);
                }
            } else if constexpr (TFUSE) {    // rows past M contribute zeros
                const int lrow = wave_m * WM + qf * 16 + r;
                lds[T_DZ0 + lrow * 16 + ((wave_n * (WN / 8) + echk) ^ t_swz16(lrow))] = zero16();
            }
            if (PIPE && t + DEPTH < NSTEP) issue(t + DEPTH);   // recycle this step's operand registers
        }
    }
    if constexpr (TFUSE) {
        // ---- T side product: tacc[c][k] += sum over the tile's rows of dz[row][c] * t_a[row][k] ----------------------------
        // operands by ds_read_b64_tr_b16 as in tn_kernel: lane (g = l>>4, q = (l&15)>>2, pp = l&3) addresses row 8g + 4h + q, columns
        // base + 4pp.., and receives column base + (l&15) for rows 8g + 4h + 0..3.  Wave w owns c = 32 w .. 32 w + 31 (two fragments).
        const int tg = lane >> 4, tq = (lane & 15) >> 2, tpp = lane & 3;
        auto t_frag = [&](const char* tile, int cpr, int row0, int col0) -> u32x4 {
            const int col = col0 + 4 * tpp;
            unsigned w[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = row0 + 8 * tg + 4 * h + tq;
                const int ch = (col >> 3) ^ (cpr >= 16 ? t_swz16(row) : ((((row >> 1) & 1) | ((row >> 2) & 2)) << 1));
                const char* ad = tile + (row * cpr + ch) * 16 + (col & 7) * 2;
                s16x4 tt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                u32x2 t2 = __builtin_bit_cast(u32x2, tt);
                w[2 * h] = t2.x; w[2 * h + 1] = t2.y;
            }
            return u32x4{w[0], w[1], w[2], w[3]};
        };
        const char* dz_tile = (const char*)(lds + T_DZ0);
        const char* a_tile = (const char*)(lds + T_A0);
#pragma unroll
        for (int pass = 0; pass < 128 / T_ROWS; ++pass) {
            if (pass > 0) {
                __syncthreads();                 // every wave is done with the first half of t_a
                t_stage(pass * T_ROWS);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (128-channel operand only: this one also waits for the epilogue's stores)
            }
            __syncthreads();                     // dz tile complete (pass 0), t_a rows landed
#pragma unroll
            for (int ks = 0; ks < T_ROWS / 32; ++ks) {
                u32x4 pf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) pf[a] = t_frag(dz_tile, 16, pass * T_ROWS + ks * 32, (2 * wave + a) * 16);
#pragma unroll
                for (int b = 0; b < TP / 16; ++b) {
                    const u32x4 qf2 = t_frag(a_tile, T_CPR, ks * 32, b * 16);
#pragma unroll
                    for (int a = 0; a < 2; ++a) Mma<T>::run(pf[a], qf2, tacc[a][b]);
                }
            }
        }
    }
    if ((HAS_FWD_STATS || HAS_BN) && p.stats_part) {
        // column partials: every lane parks its 8 (sum, sum-of-products) pairs in LDS, then one thread per (128-row half,
        // column, statistic) adds the 2 x RPP row groups in a fixed order (a shuffle butterfly cost 48 ds_bpermute + 48 adds
        // per lane; these launches are instruction bound).  The partial-sum buffer is always indexed by 128-row tiles
        // (rpe_conv_stats_tiles).
        __syncthreads();  // every wave is done reading its staging rows
        float* red = (float*)lds;  // [WAVES_M][RPP][BN][2]
        {
            float* dst = red + ((wave_m * RPP + erow) * BN + nl) * 2;
#pragma unroll
            for (int j = 0; j < 8; j += 2) *(f32x4*)(dst + 2 * j) = f32x4{cs[j], cq[j], cs[j + 1], cq[j + 1]};
        }
        __syncthreads();
        for (int i = tid; i < (WAVES_M / 2) * BN * 2; i += NTHR) {
            const int h = i / (BN * 2), cw = i - h * (BN * 2), c = cw >> 1, which = cw & 1;
            long t128 = (long)tile_m * (WAVES_M / 2) + h;
            if (MODE == MODE_CONV && g.parity) t128 = (long)cls * ((g.rows_q + 127) / 128) + m0 / 128 + h;
            if (n0 + c < p.N && m0 + h * 128 < row_lim) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * RPP; ++k) t += red[(2 * h * RPP + k) * BN * 2 + cw];
                p.stats_part[(t128 * 2 + which) * p.N + n0 + c] = t;
            }
        }
    }
    if constexpr (TFUSE) __syncthreads();   // the next tile's operand ring overwrites the LDS images and the reduction scratch
    }   // tile loop
    if constexpr (TFUSE) {
        // this workgroup's partial of T, row-major [128 c][TP k]: element e of lane l of fragment (a, b) is (c = 16 (2 wave + a) + 4 (l>>4) + e, k = 16 b + (l&15))
        float* mine = p.t_slab + (long)blockIdx.x * (128 * TP);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TP / 16; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) mine[((2 * wave + a) * 16 + 4 * (lane >> 4) + e) * TP + b * 16 + (lane & 15)] = tacc[a][b][e];
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

// KSUB_: MFMA K sub-steps (32 bf16 / 16 f32 rows each) per staged tile; NSLOT: ring depth (NSLOT-1 tiles in flight behind a counted
// vmcnt, as in nt_kernel -- the round-1 version kept ONE tile in flight and drained vmcnt to 0 every step: 1.2 TB/s on the
// HBM-bound 1x1 layers); SLAB: every workgroup stores its fp32 tile to its own slab (plain 16-byte stores in fragment order) and
// tn_reduce_kernel sums the slabs in a fixed order -- deterministic, and no float atomics (1.3 TB/s chip-wide, 23x write
// amplification in round 1).
template <typename T, int BI, int BJ, int MODE, bool USE_DMA = true, int KSUB_ = 2, int NSLOT = 2, bool SLAB = false>
__global__ __launch_bounds__(256) void tn_kernel(const TNArgs<T> p) {
    constexpr bool DMA = USE_DMA && MODE != MODE_STEM;  // LDS-DMA staging through buffer descriptors; stem: registers
    constexpr int CE = Elem<T>::kChunk, ES = (int)sizeof(T);
    constexpr int KSUB = DMA ? KSUB_ : 1;              // MFMA K sub-steps (4 chunks = 32 bf16 / 16 f32 rows each) per staged tile
    constexpr int BMK = 4 * CE * KSUB;
    constexpr int WI = BI / 2, WJ = BJ / 2, FI = WI / 16, FJ = WJ / 16;
    constexpr int CPI = BI / CE, CPJ = BJ / CE;        // chunks per row
    constexpr int RPI = 256 / CPI, RPJ = 256 / CPJ;    // rows per pass
    constexpr int NPI = (BMK + RPI - 1) / RPI, NPJ = (BMK + RPJ - 1) / RPJ;
    constexpr int PT = BMK * CPI, QT = BMK * CPJ;      // tile sizes in 16-byte units
    constexpr int STAGE = PT + QT;
    constexpr int NSTAGE = DMA ? NSLOT : 2;            // DMA: ring of NSLOT slots, NSLOT-1 tiles in flight while one is multiplied
    static_assert(NSLOT >= 2 && NSLOT <= 4, "ring depth 2..4");
    static_assert(BMK % RPI == 0 && BMK % RPJ == 0, "tile rows must split evenly over the passes");
    __shared__ u32x4 lds[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_i = wave >> 1, wave_j = wave & 1;
    const int nt = p.tiles_i * p.tiles_j;
    const int lb = xcd_remap_dir(blockIdx.x, nt * p.splits, p.rev);
    const int split = lb / nt, t2 = lb - split * nt;
    const int tile_j = t2 % p.tiles_j, tile_i = t2 / p.tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    if (!SLAB && m_begin >= m_end) return;             // (slab mode: an empty split still owes its slab of zeros)
    const Gather& g = p.g;

    // thread -> (row, LDS slot) of pass i: row = tid / CP + i * RP, slot = tid % CP (64 consecutive chunks per wave: the
    // lane-linear image one DMA instruction writes).  The slot holds logical chunk slot ^ swz(row).
    const int p_slot = tid % CPI, p_r = tid / CPI;
    const int q_slot = tid % CPJ, q_r = tid / CPJ;
    // conv: a column tile may span several taps (C = 64 with 128-column tiles: two taps, so dy is walked 5 instead of 9 times for a
    // 3x3 conv); every 16-byte chunk lies inside one tap (C % 8 == 0), its tap is fixed per thread and pass (tap_of below)

    auto p_src = [&](int mb, int i) -> const T* {
        const int row = p_r + i * RPI;
        const int cc = p_slot ^ tn_swz<T, CPI>(row);
        const int m = mb + row;
        return (m < m_end && (i0 + cc * CE) < p.I) ? p.P + (long)m * p.ldp + i0 + cc * CE : nullptr;
    };
    auto q_src = [&](int mb, int i) -> const T* {   // dense / conv only
        const int row = q_r + i * RPJ;
        const int cc = q_slot ^ tn_swz<T, CPJ>(row);
        const int m = mb + row;
        const bool ok = m < m_end && (j0 + cc * CE) < p.J;
        if (MODE == MODE_DENSE) return ok ? p.Q + (long)m * p.ldq + j0 + cc * CE : nullptr;
        const unsigned mm = ok ? (unsigned)m : 0u;
        const unsigned b = fd_div(mm, g.div_hw);
        const unsigned rem = mm - b * g.div_hw.d;
        const unsigned oh = fd_div(rem, g.div_w);
        const unsigned ow = rem - oh * g.div_w.d;
        const int cj = j0 + cc * CE, rs = cj / g.C, tc = cj - rs * g.C, tr = rs / g.S, ts = rs - tr * g.S;   // this chunk's tap
        const int ih = (int)oh * g.sn + g.base_h + tr, iw = (int)ow * g.sn + g.base_w + ts;
        return (ok && ih >= 0 && iw >= 0 && ih < g.H && iw < g.W)
                   ? p.Q + (long)b * g.img_stride + ((long)ih * g.W + iw) * g.C + tc : nullptr;
    };
    // DMA path addressing (as in nt_kernel): raw buffer descriptors, a 32-bit byte offset per lane and chunk.  Rows advance
    // by BMK per step: dense operands add a uniform byte stride to their offsets; the conv operand re-derives (image, oh, ow)
    // of its rows with two multiply-shift divisions.  Rows past M fall outside num_records and read as zero (the hardware
    // range check covers the voffset, which is why the row advance lives there and not in the SGPR offset); columns past I / J
    // and padding taps are sent there explicitly (0x80000000; extents < 2 GiB are checked by the launcher).
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr unsigned OOB = 0x80000000u;
    // which tensor this tile's P columns come from (uniform per workgroup): P, P2 (rows >= I1), or none (the all-ones tile)
    const bool ones_tile = DMA && MODE == MODE_DENSE && p.ones_i0 > 0 && i0 == p.ones_i0;
    const bool second_p = DMA && MODE == MODE_DENSE && p.P2 != nullptr && i0 >= p.I1 && !ones_tile;
    const int p_col0 = second_p ? i0 - p.I1 : i0, p_cols = ones_tile ? 0 : (second_p ? p.I2 : (p.p_cols > 0 ? p.p_cols : (p.P2 ? p.I1 : p.I))), p_ld = second_p ? p.ldp2 : p.ldp;
    // the descriptors start at this split's first row (dense) / first image (conv operand): 32-bit offsets span one split, tensors may
    // be of any size; num_records = the bytes from there to the end of the tensor, clamped below 2^31 (see nt_kernel)
    auto clamp31 = [](long bytes) -> int { return (int)(bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes)); };
    const long p_tile = (long)m_begin * p_ld;
    const unsigned q_b0 = (MODE == MODE_CONV) ? (unsigned)__builtin_amdgcn_readfirstlane((int)fd_div((unsigned)m_begin, g.div_hw)) : 0u;
    const long q_tile = (MODE == MODE_CONV) ? (long)q_b0 * g.img_stride : (long)m_begin * p.ldq;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)((second_p ? p.P2 : p.P) + p_tile), 0, clamp31(((long)p.M * p_ld - p_tile) * (long)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Q + q_tile), 0, clamp31((p.q_elems - q_tile) * (long)sizeof(T)), 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned p_voff[NPI], q_voff[NPJ];
    int q_cc[NPJ];
    bool q_okc[NPJ];
    // conv operand: (image, oh, ow) of each of the thread's rows, advanced by BMK rows per step with adds and carries only
    // (BMK = nb images + qh output rows + rw pixels); offsets are built with 24-bit multiplies (full rate; the launcher
    // checks that the image stride and pixel counts fit)
    unsigned q_b[NPJ], q_oh[NPJ], q_ow[NPJ];
    int q_tr[NPJ], q_ts[NPJ], q_tc[NPJ];   // tap (r, s) and channel offset inside the tap of this thread's chunk in pass i
    unsigned adv_b = 0, adv_h = 0, adv_w = 0;
    if (DMA && MODE == MODE_CONV) {
        const unsigned hw = g.div_hw.d, wo = g.div_w.d;
        adv_b = BMK / hw;
        const unsigned rem = BMK - adv_b * hw;
        adv_h = rem / wo;
        adv_w = rem - adv_h * wo;
    }
    if (DMA) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int row = p_r + i * RPI;
            const int cc = p_slot ^ tn_swz<T, CPI>(row);
            p_voff[i] = (p_col0 + cc * CE) < p_cols ? (unsigned)(((long)row * p_ld + p_col0 + cc * CE) * ES) : OOB;
        }
#pragma unroll
        for (int i = 0; i < NPJ; ++i) {
            const int row = q_r + i * RPJ;
            q_cc[i] = q_slot ^ tn_swz<T, CPJ>(row);
            q_okc[i] = (j0 + q_cc[i] * CE) < p.J;
            q_tr[i] = q_ts[i] = q_tc[i] = 0;
            if (MODE == MODE_CONV) {
                const int cj = j0 + q_cc[i] * CE, rs = cj / g.C;
                q_tc[i] = cj - rs * g.C; q_tr[i] = rs / g.S; q_ts[i] = rs - q_tr[i] * g.S;
            }
            q_voff[i] = OOB;
            q_b[i] = q_oh[i] = q_ow[i] = 0;
            if (MODE == MODE_DENSE && q_okc[i]) q_voff[i] = (unsigned)(((long)row * p.ldq + j0 + q_cc[i] * CE) * ES);
            if (MODE == MODE_CONV) {
                const unsigned m = (unsigned)(m_begin + row);
                const unsigned bb = fd_div(m, g.div_hw);
                q_b[i] = bb - q_b0;                                  // image index relative to the split's first image
                const unsigned rem = m - bb * g.div_hw.d;
                q_oh[i] = fd_div(rem, g.div_w);
                q_ow[i] = rem - q_oh[i] * g.div_w.d;
            }
        }
    }
    const unsigned p_step = (unsigned)(BMK * p_ld * ES), q_step = (unsigned)(BMK * p.ldq * ES);
    auto dma_tile = [&](int st) {
        const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)((lds_char*)lds + st * (STAGE * 16)));
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            dma16_asm(rs_p, lds_u + (unsigned)((i * 4 + wave_u) * 1024), p_voff[i]);
            if (p_voff[i] < OOB) p_voff[i] += p_step;   // (the compare keeps the out-of-range marker where it is)
        }
#pragma unroll
        for (int i = 0; i < NPJ; ++i) {
            unsigned vo = q_voff[i];
            if (MODE == MODE_DENSE) {
                if (q_voff[i] < OOB) q_voff[i] += q_step;
            } else {
                // rows past M have q_b >= batch: their offset lies beyond num_records, no separate test
                const int ih = (int)__umul24(q_oh[i], (unsigned)g.sn) + g.base_h + q_tr[i], iw = (int)__umul24(q_ow[i], (unsigned)g.sn) + g.base_w + q_ts[i];
                const bool ok = q_okc[i] && ih >= 0 && iw >= 0 && ih < g.H && iw < g.W;
                const unsigned pix = __umul24((unsigned)ih, (unsigned)g.W) + (unsigned)iw;
                vo = ok ? (__umul24(q_b[i], (unsigned)g.img_stride) + __umul24(pix, (unsigned)g.C) + (unsigned)q_tc[i]) * ES : OOB;
                q_ow[i] += adv_w;
                const bool cw = q_ow[i] >= g.div_w.d;
                q_ow[i] -= cw ? g.div_w.d : 0u;
                q_oh[i] += adv_h + (cw ? 1u : 0u);
                const bool ch = q_oh[i] >= (unsigned)g.Ho;
                q_oh[i] -= ch ? (unsigned)g.Ho : 0u;
                q_b[i] += adv_b + (ch ? 1u : 0u);
            }
            dma16_asm(rs_q, lds_u + (unsigned)(PT * 16 + (i * 4 + wave_u) * 1024), vo);
        }
    };
    u32x4 rp[NPI], rq[NPJ];
    auto load_tile = [&](int mb) {  // register staging (stem)
#pragma unroll
        for (int i = 0; i < NPI; ++i) { const T* src = p_src(mb, i); rp[i] = src ? ld16(src) : zero16(); }
#pragma unroll
        for (int i = 0; i < NPJ; ++i) {
            const int row = q_r + i * RPJ;
            const int cc = q_slot ^ tn_swz<T, CPJ>(row);
            const int m = mb + row;
            const bool ok = m < m_end && (j0 + cc * CE) < p.J;
            const unsigned mm = ok ? (unsigned)m : 0u;
            const unsigned b = fd_div(mm, g.div_hw);
            const unsigned rem = mm - b * g.div_hw.d;
            const unsigned oh = fd_div(rem, g.div_w);
            const unsigned ow = rem - oh * g.div_w.d;
            if (MODE == MODE_STEM) rq[i] = stem_chunk<T>(p.Q, (long)b * g.img_stride, (int)oh * g.sn + g.base_h, (int)ow * g.sn + g.base_w, g.H, g.W, j0 + cc * CE, ok);
            else { const T* src = q_src(mb, i); rq[i] = src ? ld16(src) : zero16(); }
        }
    };
    auto store_tile = [&](int st) {
        u32x4* base = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < NPI; ++i) base[(p_r + i * RPI) * CPI + p_slot] = rp[i];
#pragma unroll
        for (int i = 0; i < NPJ; ++i) base[PT + (q_r + i * RPJ) * CPJ + q_slot] = rq[i];
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = m_end > m_begin ? (m_end - m_begin + BMK - 1) / BMK : 0;
    const int fg = lane >> 4, fl = lane & 15;
    auto compute = [&](int cur) {
        const char* pb = (const char*)(lds + cur * STAGE);
        const char* qb = (const char*)(lds + cur * STAGE + PT);
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
        const int rb = ks * 4 * CE;   // first row of this K sub-step inside the staged tile
        if (sizeof(T) == 2) {
            // lane (g = l>>4, q = (l&15)>>2, pp = l&3) addresses row m = 8g + 4h + q, columns base + 4pp..4pp+3;
            // it receives column base + (l&15) for rows 8g + 4h + 0..3  -> MFMA k = 8g + (4h + e)
            const int q = fl >> 2, pp = fl & 3;
            u32x4 pf[FI], qf[FJ];
            if (ones_tile) {   // P = all ones: the tile's rows become the column sums of Q
                const unsigned one2 = Elem<T>::kDtype == RPE_BF16 ? 0x3F803F80u : 0x3C003C00u;
#pragma unroll
                for (int a = 0; a < FI; ++a) pf[a] = u32x4{one2, one2, one2, one2};
            } else
#pragma unroll
            for (int a = 0; a < FI; ++a) {
                const int col = wave_i * WI + a * 16 + 4 * pp;  // element column inside the tile
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = rb + 8 * fg + 4 * h + q;
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
                    const int row = rb + 8 * fg + 4 * h + q;
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
                    Mma<T>::run(pf[a], qf[b], acc[a][b]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = rb + kk * 4 + fg;
                float pf[FI], qf[FJ];
#pragma unroll
                for (int a = 0; a < FI; ++a) {
                    const int col = wave_i * WI + a * 16 + fl;
                    pf[a] = ones_tile ? 1.f : *(const float*)(pb + (row * CPI + ((col >> 2) ^ tn_swz<T, CPI>(row))) * 16 + (col & 3) * 4);
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
        }
    };
    if (DMA) {
        // Ring of NSTAGE slots: tiles st+1 .. st+PF are in flight while tile st is multiplied.  A tile is NI LDS-DMA
        // instructions per wave; vmcnt counts them in issue order, so "all but the newest k*NI landed" == every tile up to
        // st+PF-k is complete.  The raw s_barrier (not __syncthreads, which would drain vmcnt to 0) publishes tile st+1 to
        // the other waves; the slot refilled in iteration st was last read in iteration st-1, i.e. before a barrier every wave
        // has passed, with its ds_reads retired by the lgkmcnt(0) of wait_vmcnt (WAR, see there).
        constexpr int NI = NPI + NPJ, PF = NSTAGE - 1;
        auto wait_newer = [&](int newer) {
            if (newer >= 3) wait_vmcnt<3 * NI>(); else if (newer == 2) wait_vmcnt<2 * NI>(); else if (newer == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
        };
#pragma unroll
        for (int t = 0; t < PF; ++t)
            if (t < nsteps) dma_tile(t);
        wait_newer((nsteps < PF ? nsteps : PF) - 1);
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int st = 0; st < nsteps; ++st) {
            if (st + PF < nsteps) { int s2 = slot + PF; if (s2 >= NSTAGE) s2 -= NSTAGE; dma_tile(s2); }   // (tiles are requested in row order)
            compute(slot);
            int newer = nsteps - 2 - st;   // tiles issued after st+1
            if (newer > PF - 1) newer = PF - 1;
            wait_newer(newer);
            __builtin_amdgcn_s_barrier();
            if (++slot == NSTAGE) slot = 0;
        }
    } else {
        load_tile(m_begin);
        store_tile(0);
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            const bool more = st + 1 < nsteps;
            if (more) load_tile(m_begin + (st + 1) * BMK);
            compute(cur);
            if (more) store_tile(cur ^ 1);
            __syncthreads();
        }
    }
    if (SLAB) {
        // fragment order: [split][tile][wave][a][b][lane] x 4 floats -- one 1-KB store per wave and fragment; tn_reduce_kernel
        // knows the order
        float* slab = p.slab + ((long)split * nt + t2) * (BI * BJ);
#pragma unroll
        for (int a = 0; a < FI; ++a)
#pragma unroll
            for (int b = 0; b < FJ; ++b) *(f32x4*)(slab + ((((wave * FI + a) * FJ + b) * 64 + lane) << 2)) = acc[a][b];
        return;
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

// Sums the per-split slabs of tn_kernel<.., SLAB = true> in a fixed order and writes D (overwritten, or added to when `accumulate`).
// One block per 64 consecutive 16-byte fragment groups (1 KB per split: one coalesced wave load); its 8 waves take every 8th
// split each (several loads in flight per wave -- with few tiles and hundreds of splits a one-thread-per-element loop was
// latency bound: 170 us for the 64x64 layer1 gradient), then the 8 partial sums are added in wave order.
template <int BI, int BJ>
__global__ __launch_bounds__(512) void tn_reduce_kernel(const float* __restrict__ slab, float* __restrict__ D, int I, int J, int ldd, int tiles_j,
                                                       int nt, int splits, int accumulate) {
    constexpr int WI = BI / 2, WJ = BJ / 2, FI = WI / 16, FJ = WJ / 16, PER_TILE = BI * BJ / 4;
    static_assert(PER_TILE % 64 == 0, "whole waves per tile");
    __shared__ f32x4 sh[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + lane;
    const long stride = (long)nt * PER_TILE;
    const f32x4* src = (const f32x4*)slab + idx;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int sp = w;
    for (; sp + 56 < splits; sp += 64) {   // 8 loads in flight per lane (same order of additions as the two-at-a-time loop below)
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(sp + 8 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
    }
    for (; sp + 8 < splits; sp += 16) {
        const f32x4 a = src[(long)sp * stride], b = src[(long)(sp + 8) * stride];
        s0 += a; s1 += b;
    }
    if (sp < splits) s0 += src[(long)sp * stride];
    sh[w][lane] = s0 + s1;
    __syncthreads();
    if (w != 0) return;
    f32x4 sum = sh[0][lane];
#pragma unroll
    for (int i = 1; i < 8; ++i) sum += sh[i][lane];
    const int tile = (int)(idx / PER_TILE), r = (int)(idx - (long)tile * PER_TILE);
    const int frag = r >> 6;
    const int b = frag % FJ, a = (frag / FJ) % FI, wave = frag / (FI * FJ);
    const int tile_j = tile % tiles_j, tile_i = tile / tiles_j;
    const int j = tile_j * BJ + (wave & 1) * WJ + b * 16 + (lane & 15);
    if (j >= J) return;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = tile_i * BI + (wave >> 1) * WI + a * 16 + 4 * (lane >> 4) + rr;
        if (i < I) { float* d = D + (long)i * ldd + j; *d = accumulate ? *d + sum[rr] : sum[rr]; }
    }
}

// Second half of a split-K inference forward (NTArgs::slab): out = relu(sum_z partial[z] + bias + addend).  The partials are
// the accumulators of nt_kernel<T, 1, 64, 4, ., 3, 4> in fragment order: tile (64 x 64) -> 2 waves (32 columns each) -> fragment
// (a, b) = 16 columns x 16 rows -> lane (row = lane & 15, 4 columns from 4 * (lane >> 4)).  One thread per 16-byte piece; the
// splits are added in z order (fixed: the result does not depend on how the workgroups were scheduled).
template <typename T>
__global__ __launch_bounds__(256) void nt_split_epilogue_kernel(const float* __restrict__ slab, int splits, int tiles_m, int tiles_n, int M, int N,
                                                               const float* __restrict__ bias, const T* __restrict__ addend, int ld_add, int relu,
                                                               T* __restrict__ C, int ldc) {
    const long piece = (long)blockIdx.x * 256 + threadIdx.x;      // 16-byte piece of one split's slab
    const long tiles = (long)tiles_m * tiles_n;
    if (piece >= tiles * 1024) return;
    const int lane = (int)(piece & 63), frag = (int)((piece >> 6) & 7), wave = (int)((piece >> 9) & 1);
    const long tile = piece >> 10;
    const int a = frag >> 2, b = frag & 3;
    const int tile_n = (int)(tile % tiles_n), tile_m = (int)(tile / tiles_n);
    const long m = (long)tile_m * 64 + b * 16 + (lane & 15);
    const int n = tile_n * 64 + wave * 32 + a * 16 + (lane >> 4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const f32x4* src = (const f32x4*)slab + piece;
    const long zstride = tiles * 1024;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
        const f32x4 p0 = src[(long)z * zstride], p1 = src[(long)(z + 1) * zstride], p2 = src[(long)(z + 2) * zstride], p3 = src[(long)(z + 3) * zstride];
        v += p0; v += p1; v += p2; v += p3;
    }
    for (; z < splits; ++z) v += src[(long)z * zstride];
    if (m >= M || n >= N) return;
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (n + j >= N) break;                                     // (Linear layers: any N)
        if (bias) o[j] += bias[n + j];
        if (addend) o[j] += Elem<T>::to_f(addend[m * ld_add + n + j]);
        if (relu) o[j] = fmaxf(o[j], 0.f);
        C[m * ldc + n + j] = Elem<T>::from_f(o[j]);
    }
}

template <typename T, int MODE> static int launch_nt_split(NTArgs<T>& a, hipStream_t s) {
    constexpr int BK = 4 * Elem<T>::kChunk;
    int steps = 0;
    const int S = nt_split_plan(a.M, a.N, a.K, BK, &steps, a.role == 2);
    a.tiles_m = ceil_div(a.M, 64);
    a.tiles_n = ceil_div(a.N, 64);
    const long tiles = (long)a.tiles_m * a.tiles_n;
    if (S < 2 || !a.slab || a.slab_bytes < nt_split_slab_bytes(a.M, a.N, S) || (MODE == MODE_CONV && a.g.parity))
        return rpe_set_error(RPE_ERR_WORKSPACE, "igemm_nt: split-K launch without a plan or with a slab smaller than planned");
    a.splits = S; a.split_steps = steps;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "nt_kernel<%s,1,64,4,%d,3,4,0>", Elem<T>::kName, MODE);
    hipLaunchKernelGGL((nt_kernel<T, 1, 64, 4, MODE, 3, 4, 0>), dim3((unsigned)tiles, (unsigned)S), dim3(128), 0, s, a);
    RPE_CHECK_LAUNCH();
    prof_split(s, "nt_split_epilogue_kernel");
    hipLaunchKernelGGL((nt_split_epilogue_kernel<T>), dim3((unsigned)((tiles * 1024 + 255) / 256)), dim3(256), 0, s, (const float*)a.slab, S, a.tiles_m,
                       a.tiles_n, a.M, a.N, a.bias, a.addend, a.ld_add, a.relu, a.C, a.ldc);
    RPE_CHECK_LAUNCH();
    return 0;
}

// -----------------------------------------------------------------------------------------------
// host launchers (templates; instantiated per element type and staging mode in the igemm_*.hip units, which compile in
// parallel -- one unit with every configuration took 5 minutes)
// -----------------------------------------------------------------------------------------------

template <typename T, int WAVES_M, int BN, int KCH, int MODE, int NST, int ROLE> static int launch_nt_role(NTArgs<T>& a, hipStream_t s, long nwg) {
    snprintf(g_last_kernel, sizeof(g_last_kernel), "nt_kernel<%s,%d,%d,%d,%d,%d,%d,%d>", Elem<T>::kName, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? a.bn_mode : 0);
    const dim3 grid((unsigned)nwg), block(128 * WAVES_M);
    if (ROLE == 1) {
        switch (a.bn_mode) {
            case 1: hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 1 : 0>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 2 : 0>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 3 : 0>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 4 : 0>), grid, block, 0, s, a); break;
            case 5:   // (the fused conv1 data gradients of the y3-free bottleneck blocks: dense launches of a 16-bit type only)
                if constexpr (MODE == MODE_DENSE && sizeof(T) == 2 && WAVES_M == 2) { hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 5 : 0>), grid, block, 0, s, a); break; }
                else return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: BN-backward mode 5 is a dense 128-row launch of a 16-bit element type");
            case 6: case 7:   // mode 5 + the T side product (persistent: grid = the workgroups chosen by the caller, launch_nt_mode)
                if constexpr (MODE == MODE_DENSE && sizeof(T) == 2 && WAVES_M == 2 && BN == 128 && KCH == 4) {
                    if (a.bn_mode == 6) hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 6 : 0>), grid, block, 0, s, a);
                    else hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, ROLE == 1 ? 7 : 0>), grid, block, 0, s, a);
                    break;
                } else return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: the T side product is a dense 128 x 128-tile launch of a 16-bit element type");
            default: hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, 0>), grid, block, 0, s, a); break;
        }
    } else {
        hipLaunchKernelGGL((nt_kernel<T, WAVES_M, BN, KCH, MODE, NST, ROLE, 0>), grid, block, 0, s, a);
    }
    RPE_CHECK_LAUNCH();
    return 0;
}

template <typename T, int WAVES_M, int BN, int KCH, int MODE, int NST = 3> static int launch_nt_cfg(NTArgs<T>& a, hipStream_t s) {
    constexpr int BM = 64 * WAVES_M;
    a.tiles_m = (MODE == MODE_CONV && a.g.parity) ? 4 * ceil_div(a.g.rows_q, BM) : MODE == MODE_HALO ? ceil_div(a.g.halo_rows, a.g.halo_rt) : ceil_div(a.M, BM);
    a.tiles_n = ceil_div(a.N, BN);
    const long nwg = (long)a.tiles_m * a.tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffL) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: bad grid");
    if (a.role == 1) return launch_nt_role<T, WAVES_M, BN, KCH, MODE, NST, 1>(a, s, nwg);
    if (a.role == 2) return launch_nt_role<T, WAVES_M, BN, KCH, MODE, NST, 2>(a, s, nwg);
    if (a.role == 3) return launch_nt_role<T, WAVES_M, BN, KCH, MODE, NST, 3>(a, s, nwg);
    return launch_nt_role<T, WAVES_M, BN, KCH, MODE, NST, 0>(a, s, nwg);
}

// configuration choice for one staging mode (argument checks are in launch_nt, igemm.hip)
template <typename T, int MODE> int launch_nt_mode(NTArgs<T>& a, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    const bool wide = a.N > 64;
    a.rev = walk_take();
    // BN partial sums are always indexed by 128-row tiles (rpe_conv_stats_tiles), whatever the M tile
    if constexpr (MODE == MODE_STEM) {
        return launch_nt_cfg<T, 2, 64, 4, MODE_STEM>(a, s);
    } else if constexpr (MODE == MODE_HALO) {
        // 64 KB of LDS (2 x 16 KB weight ring + 32 KB patch; 48 KB with 64 output channels): two workgroups per CU
        // (ring depth, isolated at 256 images: 3 slots = 80 KB level with 2 on layers 2-4 and 20 % slower on layer1; 4 slots = one workgroup
        // per CU, 35 % slower everywhere: the second workgroup on the CU, not the prefetch distance, is what keeps the matrix cores fed)
        return wide ? launch_nt_cfg<T, 2, 128, 8, MODE_HALO, 2>(a, s) : launch_nt_cfg<T, 2, 64, 8, MODE_HALO, 2>(a, s);
    } else {
        if ((a.role == 3 || (a.role == 2 && MODE == MODE_DENSE)) && a.slab && a.splits > 1) return launch_nt_split<T, MODE>(a, s);
        if (a.role == 1 && (a.bn_mode == 6 || a.bn_mode == 7)) {   // fused conv1 data gradient + T side product (rpe_conv1x1_dgrad_bn_t): persistent
            if constexpr (MODE == MODE_DENSE && sizeof(T) == 2) {
                a.tiles_m = ceil_div(a.M, 128); a.tiles_n = ceil_div(a.N, 128);
                if ((a.N % 128) || !a.t_a || !a.t_slab || a.tiles_per_wg <= 0) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: T side product needs N % 128 == 0, t_a, t_slab and a plan");
                const long nwg = nt_tfuse_grid(a.M, a.N);
                if ((long)a.tiles_per_wg * (nwg / a.tiles_n) < a.tiles_m) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: T side product: tiles_per_wg does not cover the row tiles");
                if (a.K <= 8 * CE && a.M >= 4096) return launch_nt_role<T, 2, 128, 4, MODE_DENSE, 2, 1>(a, s, nwg);
                return launch_nt_role<T, 2, 128, 4, MODE_DENSE, 3, 1>(a, s, nwg);
            } else {
                return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: the T side product is a dense launch of a 16-bit element type");
            }
        }
        if (a.role == 5) {   // 1x1 training forward with the BatchNorm apply fused (rpe_conv1x1_fwd_bn): 16-bit element types, dense
            if constexpr (MODE == MODE_DENSE && sizeof(T) == 2) {
                a.tiles_m = ceil_div(a.M, 128); a.tiles_n = ceil_div(a.N, 128);
                const long nwg = (long)a.tiles_m * a.tiles_n;
                if (a.K <= 8 * CE && a.M >= 4096) return launch_nt_role<T, 2, 128, 4, MODE_DENSE, 2, 5>(a, s, nwg);   // (K = 64: both steps up front, 32 KB of LDS)
                return launch_nt_role<T, 2, 128, 4, MODE_DENSE, 3, 5>(a, s, nwg);
            } else {
                return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: the fused-BN forward is a dense launch of a 16-bit element type");
            }
        }
        // (256-row / 8-wave tiles, 1 workgroup per CU: measured on the ResNet shapes at bs256 they gain 3..10 % in isolation for K >= 1024,
        // lose 10..25 % for short K, and LOSE inside the train step -- fused epilogues, 2 waves/SIMD in lockstep: 34.4 vs 32.9 ms/step.)
        // Long reductions (K >= 1024: the 3x3 convs from layer2 on and the deep 1x1s): 128-B K rows (BK 64) with a 2-slot
        // ring (64 KB LDS, 2 workgroups per CU) -- half the barriers per FLOP; measured +10..15 % there, -5..15 % on short K.
        // (4- and 5-slot rings of 64-B rows, same or 1.25x the LDS and 96 / 128 instead of 64 K elements in flight, measured
        // 8 % slower on the forward shapes and level on the data gradients: the per-step cost, not the prefetch depth, is
        // what the longer rows buy back.  Only the stride-2 parity-class data gradients, 1-4 taps deep, gained 15 %.)
        if (a.M >= 1024 && a.K >= 1024) return wide ? launch_nt_cfg<T, 2, 128, 8, MODE, 2>(a, s) : launch_nt_cfg<T, 2, 64, 8, MODE, 2>(a, s);
        if constexpr (MODE == MODE_DENSE) {
            // few-row Linear layers (the 256-row fusion MLP and ResNet fc: 4..16 tiles of 128x128 on 256 CUs, each walking K
            // alone at the fp32 MFMA rate -- 277 us for 256x1024x3655): 64x64 tiles / 2 waves put 4..8x as many workgroups
            // on the chip
            if (!a.stats_part && (long)ceil_div(a.M, 128) * ceil_div(a.N, wide ? 128 : 64) < 96) return launch_nt_cfg<T, 1, 64, 4, MODE_DENSE>(a, s);
        }
        if constexpr (MODE == MODE_DENSE) {
            // K <= two 64-byte steps (layer1's 64-channel 1x1 convs): a 2-slot ring with both steps requested up front = 32 KB of LDS,
            // four workgroups per CU instead of three -- more operand bytes in flight for launches that live on them
            if (a.role <= 1 && a.K <= 8 * CE && a.M >= 4096 && wide) return launch_nt_cfg<T, 2, 128, 4, MODE, 2>(a, s);
        }
        return wide ? launch_nt_cfg<T, 2, 128, 4, MODE>(a, s) : launch_nt_cfg<T, 2, 64, 4, MODE>(a, s);
    }
}

// Split of M over workgroups for one tile configuration.  Atomic mode: every workgroup adds its whole BIxBJ fp32 tile with
// atomics (64 KB at 128x128) at ~1.3 TB/s chip-wide -> ~512 workgroups of 128x128.  Slab mode: the same bytes are written once
// with plain stores and read once by tn_reduce_kernel.
template <typename T, int BI, int BJ, int MODE> static void plan_tn_cfg(TNArgs<T>& a, int BMK, int wg_div) {
    a.tiles_i = ceil_div(a.I, BI);
    a.tiles_j = ceil_div(a.J, BJ);
    const long tiles = (long)a.tiles_i * a.tiles_j;
    // (256..768 measured within +-2 % ALONE; RPE_TN_WGS: experiments with fewer, longer workgroups beside the data-gradient chain)
    static const long target_all = getenv("RPE_TN_WGS") ? atol(getenv("RPE_TN_WGS")) : 512;
    // RPE_TN_WGS_BIGM="wgs,Mmin": another target for the long reductions only (layers 1-2, where the second stream has slack)
    static long big_wgs = 0, big_m = 0;
    static const bool big_set = getenv("RPE_TN_WGS_BIGM") && sscanf(getenv("RPE_TN_WGS_BIGM"), "%ld,%ld", &big_wgs, &big_m) == 2 && big_wgs > 0;
    const long target_wgs = (big_set && a.M >= big_m) ? big_wgs : target_all;
    // (scaled so the atomic / slab bytes, not the workgroup count, stay constant across tile sizes)
    const long wgs = target_wgs * (128 * 128) / (BI * BJ) / wg_div;   // (wg_div 2: one 96-KB workgroup per CU)
    // round DOWN when that still fills >= 70 % of the target: the target is what is resident at once (64 KB of LDS per
    // 128x128 workgroup = 2 per CU), and e.g. 144 tiles x 4 splits = 576 workgroups ran as a full round plus a 12 % round
    long want = wgs / tiles;
    if (want < 1 || tiles * want * 10 < wgs * 7) want = (wgs + tiles - 1) / tiles;
    long max_splits = a.M / (16 * BMK);
    if (max_splits < 1) max_splits = 1;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    long rps = (a.M + want - 1) / want;
    rps = (rps + BMK - 1) / BMK * BMK;
    a.rows_per_split = (int)rps;
    a.splits = (int)((a.M + rps - 1) / rps);
}

// Ring configuration of the DMA path.  Rounds 2-3 measured the 2-slot ring fastest wherever operands are re-read through L2 and blamed the
// LDS fill rate; round 4 found the cause in the ISA -- the compiler drained every request in front of the first transposing read
// (dma16_asm above), so no ring depth had kept a tile in flight.  With the DMA in assembly (bench.py on one box, profiles/r04_ab_tn_asm_dma.txt):
// three 64-row slots (96 KB at 128 x 128: one workgroup per CU, two tiles in flight) win wherever the reduction is long -- M >= 40 000 rows,
// layers 1-3 at 256 images: 18.52 -> 18.31 ms/step -- and two slots (two workgroups per CU) where it is short (layer 4, the heads); 32-row
// steps in 3 or 4 slots lose 0.2-0.3 ms.  RPE_TN_RING="ksub,nslot[,Mmin]" overrides (experiments).
static inline void tn_ring(long M, int& ksub, int& nslot) {
    ksub = 2;
    nslot = M >= 40000 ? 3 : 2;
    static const char* ov = getenv("RPE_TN_RING");   // experiment: "ksub,nslot[,Mmin]" for every launch with M >= Mmin (default 0)
    if (ov) {
        int k = 0, n = 0; long mm = 0;
        if (sscanf(ov, "%d,%d,%ld", &k, &n, &mm) >= 2 && (k == 1 || k == 2) && (n == 2 || n == 3 || (n == 4 && k == 1)) && M >= mm) { ksub = k; nslot = n; }
    }
}

template <typename T, int BI, int BJ, int MODE, int KS, int NS> static int launch_tn_ring(TNArgs<T>& a, hipStream_t s, long nwg) {
    if (a.slab) {
        hipLaunchKernelGGL((tn_kernel<T, BI, BJ, MODE, true, KS, NS, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        RPE_CHECK_LAUNCH();
        prof_split(s, "tn_reduce_kernel");
        const long nt = (long)a.tiles_i * a.tiles_j, groups = nt * (BI * BJ / 4);
        hipLaunchKernelGGL((tn_reduce_kernel<BI, BJ>), dim3((unsigned)(groups / 64)), dim3(512), 0, s, a.slab, a.D, a.I, a.J, a.ldd, a.tiles_j,
                           (int)nt, a.splits, a.accumulate);
    } else {
        hipLaunchKernelGGL((tn_kernel<T, BI, BJ, MODE, true, KS, NS, false>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    }
    RPE_CHECK_LAUNCH();
    return 0;
}

// slab_query: only report the slab bytes this problem needs (rpe_*_wgrad_workspace_bytes), launch nothing
template <typename T, int BI, int BJ, int MODE> static int launch_tn_cfg(TNArgs<T>& a, hipStream_t s, long* slab_query) {
    // register staging (the stem) walks 4 chunks of rows per step, the LDS-DMA ring 8
    const bool dma = MODE != MODE_STEM;
    int ksub = 1, nslot = 2;
    if (dma) tn_ring(a.M, ksub, nslot);
    // the stem's 64 x 256 tile is 40 KB per slot: two slots = two workgroups per CU (0.191 ms), three = one (0.306 ms; the register-staged
    // form of rounds 1-3: 0.286 ms) -- profiles/r04_ab_stem_dma.txt
    if (BJ == 256) nslot = 2;
    const int BMK = (dma ? 4 * ksub : 4) * Elem<T>::kChunk;
    // (the register-staged stem kernel is bound by the latency of its load -> LDS -> barrier steps, not by bytes: 2x / 4x the
    // workgroups measured level, as did reading dy once instead of four times)
    plan_tn_cfg<T, BI, BJ, MODE>(a, BMK, (dma && nslot == 3 && ksub == 2) ? 2 : 1);
    const long nwg = (long)a.tiles_i * a.tiles_j * a.splits;
    const long slab_bytes = nwg * (long)(BI * BJ) * 4;
    if (slab_query) { *slab_query = slab_bytes; return 0; }
    if (a.slab && a.slab_bytes < slab_bytes) a.slab = nullptr;   // too small a workspace: atomic accumulation
    // buffer-descriptor extents (DMA path): rows past M must fall outside them, padding uses offset 2^31
    // (tensors of any size: the kernel's descriptors start at each split's first row / image; a split itself must stay below 2 GiB)
    if (MODE == MODE_DENSE) a.q_elems = (long)a.M * a.ldq;
    {
        const long span = (long)a.rows_per_split * (a.ldp > a.ldq ? a.ldp : a.ldq) * (long)sizeof(T);
        const long span_q = MODE == MODE_DENSE ? 0 : ((long)a.rows_per_split / (a.g.Ho * a.g.Wo) + 2) * a.g.img_stride * (long)sizeof(T);
        if (dma && (span >= (1L << 31) || span_q >= (1L << 31))) return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: one split of the reduction spans 2 GiB or more");
    }
    if (!a.P2 && a.ones_i0 > 0) {   // P (p_cols columns) + the all-ones tile only: x^T x and colsum(x) in one launch
        if (!dma || MODE != MODE_DENSE || a.p_cols <= 0 || a.p_cols > a.ones_i0 || (a.ones_i0 % BI))
            return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: bad all-ones tile (dense DMA path, p_cols <= ones_i0, ones_i0 a multiple of the I tile)");
    } else
    if (a.P2 || a.ones_i0 > 0) {
        const long p2b = (long)a.M * a.ldp2 * (long)sizeof(T);
        if (!dma || MODE != MODE_DENSE || !a.P2 || (a.I1 % BI) || (a.ones_i0 > 0 && (a.ones_i0 % BI)) || a.I1 <= 0 || a.I2 <= 0 || (a.ldp2 % Elem<T>::kChunk) ||
            (((uintptr_t)a.P2) & 15) || p2b <= 0)
            return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: bad row-concatenated P operand (dense DMA path, I1 / ones_i0 multiples of the I tile)");
    }
    snprintf(g_last_kernel, sizeof(g_last_kernel), "tn_kernel<%s,%d,%d,%d,%d,%d,%d,%d>", Elem<T>::kName, BI, BJ, MODE, dma ? 1 : 0, dma ? ksub : 1,
             dma ? nslot : 2, a.slab ? 1 : 0);
    if constexpr (MODE == MODE_STEM) {
        if (a.slab) {
            hipLaunchKernelGGL((tn_kernel<T, BI, BJ, MODE, false, 1, 2, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
            RPE_CHECK_LAUNCH();
            prof_split(s, "tn_reduce_kernel");
            const long nt = (long)a.tiles_i * a.tiles_j, groups = nt * (BI * BJ / 4);
            hipLaunchKernelGGL((tn_reduce_kernel<BI, BJ>), dim3((unsigned)(groups / 64)), dim3(512), 0, s, a.slab, a.D, a.I, a.J, a.ldd,
                               a.tiles_j, (int)nt, a.splits, a.accumulate);
        } else {
            hipLaunchKernelGGL((tn_kernel<T, BI, BJ, MODE, false, 1, 2, false>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        }
        RPE_CHECK_LAUNCH();
        return 0;
    } else {
        if (ksub == 1) {
            if (nslot == 4) return launch_tn_ring<T, BI, BJ, MODE, 1, 4>(a, s, nwg);
            if (nslot == 3) return launch_tn_ring<T, BI, BJ, MODE, 1, 3>(a, s, nwg);
            return launch_tn_ring<T, BI, BJ, MODE, 1, 2>(a, s, nwg);
        }
        if (nslot == 3) return launch_tn_ring<T, BI, BJ, MODE, 2, 3>(a, s, nwg);
        return launch_tn_ring<T, BI, BJ, MODE, 2, 2>(a, s, nwg);
    }
    return 0;
}

template <typename T> int launch_tn(TNArgs<T>& a, int mode, hipStream_t s, long* slab_query) {
    constexpr int CE = Elem<T>::kChunk;
    a.rev = slab_query ? 0 : walk_take();
    if (a.M <= 0 || a.I <= 0 || a.J <= 0) return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: empty problem");
    if (!slab_query) {
        if ((a.ldp % CE) || (((uintptr_t)a.P) & 15) || (((uintptr_t)a.Q) & 15))
            return rpe_set_error(RPE_ERR_ALIGN, "igemm_tn: operands must be 16-byte aligned with ld a multiple of the 16-byte chunk");
        if (mode == MODE_DENSE && (a.ldq % CE)) return rpe_set_error(RPE_ERR_ALIGN, "igemm_tn: dense ldq must be a chunk multiple");
        if (a.slab && (((uintptr_t)a.slab) & 15)) return rpe_set_error(RPE_ERR_ALIGN, "igemm_tn: the slab workspace must be 16-byte aligned");
    }
    // stem (7x7, J = 224 of 256 packed columns): ONE 64 x 256 tile, so the 411-MB dy operand is read once (round 2 walked it once
    // per 64-column tile: four times, 0.35 ms at the end of the step with nothing beside it)
    // round 4: the operand is the zero-bordered image, over which the stem is an 8 x 8 / stride 2 / pad 0 conv of 4 channels (taps 7 meet
    // zero weights): every 16-byte chunk of its im2col row (two pixels of one tap row) is in bounds and aligned, so it takes the
    // LDS-DMA path of the conv mode (the register-staged form it replaces ran at 1.8 TB/s)
    if (mode == MODE_STEM) {
        if (a.g.C != 4 || a.g.R != 8 || a.g.S != 8 || a.g.sn != 2 || a.g.base_h || a.g.base_w || a.J != 256 || a.g.img_stride >= (1L << 24))
            return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: the stem operand is the zero-bordered NHWC4 image (8 x 8 taps, stride 2, no padding, 256 packed columns)");
        return launch_tn_cfg<T, 64, 256, MODE_CONV>(a, s, slab_query);
    }
    const bool wide_i = a.I > 64;
    if (mode == MODE_DENSE) {
        const bool wide_j = a.J > 64;
        if (wide_i && wide_j) return launch_tn_cfg<T, 128, 128, MODE_DENSE>(a, s, slab_query);
        if (wide_j) return launch_tn_cfg<T, 64, 128, MODE_DENSE>(a, s, slab_query);
        if (wide_i) return launch_tn_cfg<T, 128, 64, MODE_DENSE>(a, s, slab_query);
        return launch_tn_cfg<T, 64, 64, MODE_DENSE>(a, s, slab_query);
    }
    if (a.g.C % 64) return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: conv channels must be a multiple of 64");
    if (a.g.img_stride >= (1L << 24) || (long)a.g.H * a.g.W >= (1L << 24))
        return rpe_set_error(RPE_ERR_SHAPE, "igemm_tn: an image of 2^24 elements or more (offsets are built with 24-bit multiplies)");
    const bool wide_j = (a.g.C % 128) == 0 || ((a.g.C % 64) == 0 && a.J >= 128);   // (64-channel layers: a 128-column tile spans two taps)
    if (wide_i && wide_j) return launch_tn_cfg<T, 128, 128, MODE_CONV>(a, s, slab_query);
    if (wide_j) return launch_tn_cfg<T, 64, 128, MODE_CONV>(a, s, slab_query);
    if (wide_i) return launch_tn_cfg<T, 128, 64, MODE_CONV>(a, s, slab_query);
    return launch_tn_cfg<T, 64, 64, MODE_CONV>(a, s, slab_query);
}

}  // namespace rpe

// Halo form of the 3x3 / stride-1 / pad-1 weight gradient (16-bit element types), gfx950.
//
//   dW[co][r][s][ci] = sum over output pixels (b, oh, ow) of  dy[b][oh][ow][co] * x[b][oh + r - 1][ow + s - 1][ci]
//
// replaces: the weight half of the autograd backward of the 3x3 nn.Conv2d layers of the torchvision ResNet the reference builds
// (util/model_utils.py:136; conv2 of every Bottleneck, both convs of a BasicBlock), as run by models/naive.py:316's backward.
//
// tn_kernel (igemm_impl.h) treats this as D[co][(tap, ci)] = dy^T im2col(x): a 128-column tile of D is ONE tap (two for 64 channels), so
// nine (five) workgroups per row span each stage the same dy rows and the same -- shifted -- x pixels again: 32 KB of LDS fill and 32
// transposing LDS reads per 32 MFMAs and wave, which is what bounds it (25 % matrix-core busy, 265 TFLOP/s beside the data-gradient
// chain: the step's largest symbol).  Here ONE workgroup owns all nine taps of a (co tile, ci chunk):
//
//   * the reduction runs over the positions g of the PADDED pixel grid [B][H + 2][W + 2] (dy and x both placed at (h + 1, w + 1)): tap
//     (r, s) of position g then reads x at position g + (r - 1)(W + 2) + (s - 1), one uniform shift -- image borders, the two pad rows
//     between images and the ends of the tensor are zeros supplied by the buffer descriptor's range check (dy is zero at every pad
//     position, so whatever finite x lies beside it does not count).  Cost: (H + 2)(W + 2) / (H W) of the useful MFMAs (1.07 at 56 x 56,
//     1.15 at 28 x 28, 1.31 at 14 x 14), which is why narrow maps keep the gathered form (rpe_conv2d_wgrad_halo_min_width);
//   * x is read ONCE per workgroup: its positions stream through a ring of RINGP pixels in LDS (CC channels each), 64 new positions per
//     step, and every tap reads its MFMA fragments from the ring at pixel (g + shift) & (RINGP - 1) -- per-lane row addresses of
//     ds_read_b64_tr_b16, no gather, no per-tap staging;
//   * dy tiles (64 positions x BI channels) go through a (PF + 1)-slot ring as in tn_kernel;
//   * per 32-position MFMA step a wave reads 4 dy fragments and 9 x fragments for 36 MFMAs (tn_kernel: 8 for 16), its accumulators are
//     BI/16/WI x 9 fragments = 144 VGPRs: wave (wi, wj) owns co [64 wi, +64) x ci [16 wj, +16) x 9 taps;
//   * LDS-DMA from inline assembly (dma16_asm), counted vmcnt + one raw s_barrier per step (tn_kernel's ring discipline).
//
// Configurations: WI = 2: BI = 128 output channels x CC = 32 input channels; WI = 1: BI = 64 x CC = 64 (64-channel layers).
// Every workgroup stores its 128 x 288 (64 x 576) fp32 tile to its own slab; wgrad_halo_reduce_kernel adds the splits in a fixed order
// (deterministic, no float atomics).
#include <string.h>

#include "igemm_impl.h"

namespace rpe {

template <typename T> struct WHArgs {
    const T* dy;      // [B][H][W][Co]
    const T* x;       // [B][H][W][Ci]
    float* slab;      // [splits][tiles][4 waves][36 fragments][64 lanes][4]
    int B, H, W, Co, Ci;
    int tiles_c;      // Ci / CC (tiles = Co / BI * tiles_c)
    int nt, splits;
    int rows_per_split;   // padded-grid positions per split, a multiple of 64
    int G;                // B (H + 2)(W + 2)
    int E;                // x blocks a step reads beyond its own: (63 + 2 (W + 3)) / 64
    int rev;
    unsigned dy_bytes, x_bytes;
};

template <typename T, int WI, int PF, int RINGP>
__global__ __launch_bounds__(256, 2) void wgrad_halo_kernel(const WHArgs<T> p) {
    constexpr int ES = 2;
    constexpr int BI = 64 * WI, WJ = 4 / WI, CC = 16 * WJ;
    constexpr int CPI = BI / 8, CPQ = CC / 8;          // 16-byte chunks per dy row / per x pixel
    constexpr int RPI = 256 / CPI, NPI = 64 / RPI;     // dy rows per pass, passes per 64-row tile (= DMA instructions per wave)
    constexpr int PPI = 64 / CPQ;                      // x pixels per DMA instruction
    constexpr int NPQ = 64 / PPI / 4;                  // x DMA instructions per wave and 64-position block
    constexpr int NSLOT = PF + 1, NBLK = RINGP / 64;
    constexpr int PT = 64 * CPI;                       // dy tile in 16-byte units
    constexpr int NI = NPI + NPQ;                      // DMA instructions per wave and step
    static_assert(sizeof(T) == 2, "16-bit element types");
    static_assert((RINGP & (RINGP - 1)) == 0 && NBLK >= PF + 2, "ring: a power of two, at least PF + E + 1 blocks");
    __shared__ u32x4 lds[NSLOT * PT + RINGP * CPQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_i = WI == 2 ? wave >> 1 : 0, wave_j = WI == 2 ? wave & 1 : wave;
    const int lb = xcd_remap_dir(blockIdx.x, p.nt * p.splits, p.rev);
    const int split = lb / p.nt, t2 = lb - split * p.nt;
    const int tile_c = t2 % p.tiles_c, tile_i = t2 / p.tiles_c;
    const int i0 = tile_i * BI, c0 = tile_c * CC;
    const int g0 = split * p.rows_per_split;
    const int g1 = min(p.G, g0 + p.rows_per_split);
    const int nsteps = g1 > g0 ? (g1 - g0 + 63) / 64 : 0;
    const int PW = p.W + 2, PH = p.H + 2, per_img = PH * PW, HL = PW + 1;

    typedef __attribute__((address_space(3))) char lds_char;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // 64 positions further on the padded grid = adv_b images + adv_h rows + adv_w pixels
    const int adv_b = 64 / per_img, adv_r = 64 - adv_b * per_img, adv_h = adv_r / PW, adv_w = adv_r - adv_h * PW;
    auto advance = [&](int& img, int& hp, int& wp) {
        wp += adv_w;
        const bool cw = wp >= PW;
        wp -= cw ? PW : 0;
        hp += adv_h + (cw ? 1 : 0);
        const bool ch = hp >= PH;
        hp -= ch ? PH : 0;
        img += adv_b + (ch ? 1 : 0);
    };
    auto pixel_off = [&](int img, int hp, int wp, int ld) -> unsigned {   // element offset of padded position (img, hp, wp); meaningful for valid ones only
        return ((unsigned)(img * p.H + hp - 1) * (unsigned)p.W + (unsigned)(wp - 1)) * (unsigned)ld;
    };

    // ---- dy tile: thread -> (row = tid / CPI + i RPI, LDS slot tid % CPI), the lane-linear image one DMA instruction writes ----
    const int p_slot = tid % CPI, p_r = tid / CPI;
    int d_img[NPI], d_hp[NPI], d_wp[NPI], d_left[NPI], d_col[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int row = p_r + i * RPI;
        d_col[i] = i0 + (p_slot ^ tn_swz<T, CPI>(row)) * 8;
        const int g = g0 + row;
        d_img[i] = g / per_img;
        const int rem = g - d_img[i] * per_img;
        d_hp[i] = rem / PW;
        d_wp[i] = rem - d_hp[i] * PW;
        d_left[i] = g1 - g;            // > 0: the position belongs to this split
    }
    auto dma_dy = [&](int st) {
        const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)((lds_char*)lds + st * (PT * 16)));
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const bool ok = d_left[i] > 0 && d_img[i] < p.B && d_hp[i] >= 1 && d_hp[i] <= p.H && d_wp[i] >= 1 && d_wp[i] <= p.W;
            const unsigned vo = ok ? (pixel_off(d_img[i], d_hp[i], d_wp[i], p.Co) + (unsigned)d_col[i]) * ES : OOB;
            dma16_asm(rs_p, lds_u + (unsigned)((i * 4 + wave_u) * 1024), vo);
            advance(d_img[i], d_hp[i], d_wp[i]);
            d_left[i] -= 64;
        }
    };
    // ---- x ring: block u = stream positions [64 u, 64 u + 64) = grid positions g0 - HL + 64 u ..; wave w fills pixels (w NPQ + j) PPI .. ----
    // (CC = 64: 128-byte pixels, chunk slot ^ 2 [(pixel >> 1) & 1] so that the four consecutive pixels a 16-lane group of
    // ds_read_b64_tr_b16 touches fall on four different 32-byte bank groups; 64-byte pixels need no permutation)
    auto xswz = [](int pix) -> int { return CPQ == 8 ? ((pix >> 1) & 1) << 1 : 0; };
    int x_img[NPQ], x_hp[NPQ], x_wp[NPQ], x_col[NPQ];
#pragma unroll
    for (int j = 0; j < NPQ; ++j) {
        const int px = (wave * NPQ + j) * PPI + lane / CPQ;
        x_col[j] = c0 + ((lane % CPQ) ^ xswz(px)) * 8;
        int g = g0 - HL + px;
        x_img[j] = 0;
        if (g < 0) { g += per_img; x_img[j] = -1; }        // (HL <= per_img: one image back at most)
        const int im = g / per_img;
        x_img[j] += im;
        const int rem = g - im * per_img;
        x_hp[j] = rem / PW;
        x_wp[j] = rem - x_hp[j] * PW;
    }
    const unsigned xring_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)((lds_char*)lds + NSLOT * PT * 16));
    auto dma_x = [&](int u) {
        const unsigned blk = xring_u + (unsigned)((u & (NBLK - 1)) * 64 * CC * ES);
#pragma unroll
        for (int j = 0; j < NPQ; ++j) {
            const bool ok = x_img[j] >= 0 && x_img[j] < p.B && x_hp[j] >= 1 && x_hp[j] <= p.H && x_wp[j] >= 1 && x_wp[j] <= p.W;
            const unsigned vo = ok ? (pixel_off(x_img[j], x_hp[j], x_wp[j], p.Ci) + (unsigned)x_col[j]) * ES : OOB;
            dma16_asm(rs_q, blk + (unsigned)((wave_u * NPQ + j) * 1024), vo);
            advance(x_img[j], x_hp[j], x_wp[j]);
        }
    };

    f32x4 acc[4][9];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing (tn_kernel's): lane (fg = l >> 4, q = (l & 15) >> 2, pp = l & 3) addresses row 8 fg + 4 h + q, columns base + 4 pp ..
    // and receives column base + (l & 15) for rows 8 fg + 4 h + 0..3 -> MFMA k = 8 fg + 4 h + e
    const int fg = lane >> 4, fl = lane & 15, q = fl >> 2, pp = fl & 3;
    const int xcol = wave_j * 16 + 4 * pp;                         // channel inside the CC-wide pixel
    const int xchunk = xcol >> 3, xbyte = (xcol & 7) * ES;
    const char* const xb = (const char*)(lds + NSLOT * PT);
    auto rd_tr = [](const char* ad) -> u32x2 {
        s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
        return __builtin_bit_cast(u32x2, t);
    };
    auto compute = [&](int st, int step) {
        const char* pb = (const char*)(lds + st * PT);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 pf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int col = wave_i * 64 + a * 16 + 4 * pp;
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = ks * 32 + 8 * fg + 4 * h + q;
                    const int chunk = (col >> 3) ^ tn_swz<T, CPI>(row);
                    const u32x2 tt = rd_tr(pb + (row * CPI + chunk) * 16 + (col & 7) * ES);
                    w[2 * h] = tt.x; w[2 * h + 1] = tt.y;
                }
                pf[a] = u32x4{w[0], w[1], w[2], w[3]};
            }
            // stream position of this lane's rows at tap (0, 0): 64 step + row (the stream starts HL = W + 3 positions before g0, which is
            // exactly tap (0, 0)'s shift); tap (r, s) adds r (W + 2) + s
            const int pr0 = step * 64 + ks * 32 + 8 * fg + q;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sh = (t / 3) * PW + (t % 3);
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pr = pr0 + 4 * h + sh;
                    const u32x2 tt = rd_tr(xb + ((pr & (RINGP - 1)) * CPQ + (xchunk ^ xswz(pr))) * 16 + xbyte);
                    w[2 * h] = tt.x; w[2 * h + 1] = tt.y;
                }
                const u32x4 qf = u32x4{w[0], w[1], w[2], w[3]};
#pragma unroll
                for (int a = 0; a < 4; ++a) Mma<T>::run(pf[a], qf, acc[a][t]);
            }
        }
    };

    // Ring (tn_kernel's discipline): iteration t requests group t + PF = {dy tile t + PF, x block t + PF + E}, multiplies tile t -- which
    // reads x blocks t .. t + E -- and then waits until group t + 1 has landed: "all but the newest (PF - 1) NI instructions".  The E x
    // blocks of the prologue are older than every group.  A slot / block is refilled one barrier after its last reader (NBLK >= PF + E + 1).
    auto wait_newer = [&](int newer) {
        if (newer >= 2) wait_vmcnt<2 * NI>(); else if (newer == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
    };
    if (nsteps > 0) {
        for (int u = 0; u < p.E; ++u) dma_x(u);
#pragma unroll
        for (int t = 0; t < PF; ++t)
            if (t < nsteps) { dma_dy(t); dma_x(t + p.E); }
        wait_newer((nsteps < PF ? nsteps : PF) - 1);
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int st = 0; st < nsteps; ++st) {
            if (st + PF < nsteps) { int s2 = slot + PF; if (s2 >= NSLOT) s2 -= NSLOT; dma_dy(s2); dma_x(st + PF + p.E); }
            compute(slot, st);
            int newer = nsteps - 2 - st;
            if (newer > PF - 1) newer = PF - 1;
            wait_newer(newer);
            __builtin_amdgcn_s_barrier();
            if (++slot == NSLOT) slot = 0;
        }
    }
    // fragment order [split][tile][wave][a][tap][lane] x 4 floats: one 1-KB store per wave and fragment (wgrad_halo_reduce_kernel knows it)
    float* slab = p.slab + (((long)split * p.nt + t2) * 4 + wave) * (36L * 256);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t) *(f32x4*)(slab + ((a * 9 + t) * 64 + lane) * 4) = acc[a][t];
}

// Adds the splits' slabs in a fixed order and writes dW [Co][3][3][Ci] (overwritten).  One block per fragment (64 lanes x 16 B = 1 KB per
// split); its 8 waves take every 8th split each, several loads in flight, then the 8 partial sums are added in wave order (tn_reduce_kernel).
template <int WI>
__global__ __launch_bounds__(512) void wgrad_halo_reduce_kernel(const float* __restrict__ slab, float* __restrict__ D, int Ci, int tiles_c, int nt, int splits) {
    constexpr int BI = 64 * WI, WJ = 4 / WI, CC = 16 * WJ;
    __shared__ f32x4 sh[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long stride = (long)nt * (4 * 36 * 64);
    const f32x4* src = (const f32x4*)slab + (long)blockIdx.x * 64 + lane;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int sp = w;
    for (; sp + 24 < splits; sp += 32) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[(long)(sp + 8 * u) * stride];
        s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3];
    }
    for (; sp < splits; sp += 8) s0 += src[(long)sp * stride];
    sh[w][lane] = s0 + s1;
    __syncthreads();
    if (w != 0) return;
    f32x4 sum = sh[0][lane];
#pragma unroll
    for (int i = 1; i < 8; ++i) sum += sh[i][lane];
    const int group = blockIdx.x, tile = group / 144, r = group - tile * 144;
    const int wave = r / 36, a = (r % 36) / 9, tap = r % 9;
    const int tile_c = tile % tiles_c, tile_i = tile / tiles_c;
    const int wave_i = WI == 2 ? wave >> 1 : 0, wave_j = WI == 2 ? wave & 1 : wave;
    const int ci = tile_c * CC + wave_j * 16 + (lane & 15);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int co = tile_i * BI + wave_i * 64 + a * 16 + 4 * (lane >> 4) + rr;
        D[((long)co * 9 + tap) * Ci + ci] = sum[rr];
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
static int g_wgrad_halo_min_w = -1;   // -1: not set yet (RPE_WGRAD_HALO_MINW or the default)
static int wgrad_halo_min_w() {
    if (g_wgrad_halo_min_w < 0) {
        const char* e = getenv("RPE_WGRAD_HALO_MINW");
        g_wgrad_halo_min_w = e ? atoi(e) : 28;
        if (g_wgrad_halo_min_w < 0) g_wgrad_halo_min_w = 0;
    }
    return g_wgrad_halo_min_w;
}
int wgrad_halo_set_min_w(int w) {
    const int prev = wgrad_halo_min_w();
    g_wgrad_halo_min_w = w < 0 ? 0 : w;
    return prev;
}

bool wgrad_halo_ok(const rpe_conv_desc* d, int dtype) {
    static const bool off = getenv("RPE_NO_WGRAD_HALO") != nullptr;
    if (off || dtype == RPE_F32 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return false;
    if (d->in_w < wgrad_halo_min_w() || d->in_w > 60 || d->in_h < 2) return false;         // (E = (63 + 2 (W + 3)) / 64 <= 2)
    if (d->out_c % 64) return false;
    const int cc = (d->out_c % 128 == 0) ? 32 : 64;
    if (d->in_c % cc) return false;
    const long px = (long)d->batch * d->in_h * d->in_w;
    if (px * d->out_c * 2 >= (1L << 31) || px * d->in_c * 2 >= (1L << 31)) return false;   // 32-bit byte offsets from the tensor's base
    if ((long)d->batch * (d->in_h + 2) * (d->in_w + 2) >= (1L << 30)) return false;
    return true;
}

template <typename T>
int conv_wgrad_halo(const rpe_conv_desc* d, const void* x, const void* dy, float* dw, void* slab, long slab_bytes, long* slab_query, hipStream_t s) {
    const int WI = (d->out_c % 128 == 0) ? 2 : 1;
    const int BI = 64 * WI, CC = WI == 2 ? 32 : 64;
    WHArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.dy = (const T*)dy; a.x = (const T*)x; a.slab = (float*)slab;
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Co = d->out_c; a.Ci = d->in_c;
    a.tiles_c = d->in_c / CC;
    a.nt = (d->out_c / BI) * a.tiles_c;
    a.G = d->batch * (d->in_h + 2) * (d->in_w + 2);
    a.E = (63 + 2 * (d->in_w + 3)) / 64;
    // Workgroups: 96, NOT one or two per CU.  These launches run on the engine's second stream beside the HBM-bound data-gradient chain of
    // layers 1-2, which has the slack: what counts is how little they disturb that chain, not how soon they finish.  Measured in the step
    // (profiles/r04_ab_wgrad_halo.txt, ms/step): gathered form 17.80; halo form with 1024 workgroups 17.87, 512: 17.82, 384: 17.77, 256: 17.73,
    // 192: 17.69, 128: 17.68, 96: 17.63, 80: 17.68, 64: 17.77 -- although ALONE the 512-workgroup launch is the fastest (0.088 vs 0.132 ms at
    // layer 1, profiles/r04_micro_wgrad_halo.txt).  Fewer workgroups also mean a smaller slab (147 KB each).  RPE_WGRAD_HALO_WGS overrides.
    static const long target = getenv("RPE_WGRAD_HALO_WGS") ? atol(getenv("RPE_WGRAD_HALO_WGS")) : 96;
    // (maps narrower than 28 pixels -- layers 3-4, where the second stream is level with the data-gradient chain -- want the launch done soon)
    static const long target_narrow = getenv("RPE_WGRAD_HALO_WGS_NARROW") ? atol(getenv("RPE_WGRAD_HALO_WGS_NARROW")) : 512;
    long splits = (d->in_w >= 28 ? target : target_narrow) / a.nt;
    if (splits < 1) splits = 1;
    long rps = (a.G + splits - 1) / splits;
    rps = (rps + 63) / 64 * 64;
    if (rps < 256) rps = 256;                           // (at least four steps per workgroup)
    splits = (a.G + rps - 1) / rps;
    a.rows_per_split = (int)rps; a.splits = (int)splits;
    const long need = (long)a.nt * splits * (4L * 36 * 256) * 4;
    if (slab_query) { *slab_query = need; return 0; }
    if (!x || !dy || !dw || !slab) return rpe_set_error(RPE_ERR_SHAPE, "conv2d_wgrad (halo form): null operand / workspace");
    if (slab_bytes < need) return rpe_set_error(RPE_ERR_WORKSPACE, "conv2d_wgrad (halo form): workspace smaller than rpe_conv2d_wgrad_workspace_bytes()");
    if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)slab) | ((uintptr_t)dw)) & 15) return rpe_set_error(RPE_ERR_ALIGN, "conv2d_wgrad (halo form): operands must be 16-byte aligned");
    const long px = (long)d->batch * d->in_h * d->in_w;
    a.dy_bytes = (unsigned)(px * d->out_c * 2); a.x_bytes = (unsigned)(px * d->in_c * 2);
    a.rev = walk_take();
    const long nwg = (long)a.nt * splits;
    const dim3 grid((unsigned)nwg), block(256);
    // ring depth: dy tiles two ahead (one for the 64-channel form on wide maps: 128-byte pixels, 48 KB instead of 88); the x ring holds
    // PF + E + 1 blocks of 64 positions, rounded up to a power of two
    if (WI == 2) {
        snprintf(g_last_kernel, sizeof(g_last_kernel), "wgrad_halo_kernel<%s,2,2,%d>", Elem<T>::kName, a.E == 2 ? 512 : 256);
        if (a.E == 2) hipLaunchKernelGGL((wgrad_halo_kernel<T, 2, 2, 512>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_halo_kernel<T, 2, 2, 256>), grid, block, 0, s, a);
    } else {
        snprintf(g_last_kernel, sizeof(g_last_kernel), "wgrad_halo_kernel<%s,1,%d,256>", Elem<T>::kName, a.E == 2 ? 1 : 2);
        if (a.E == 2) hipLaunchKernelGGL((wgrad_halo_kernel<T, 1, 1, 256>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_halo_kernel<T, 1, 2, 256>), grid, block, 0, s, a);
    }
    RPE_CHECK_LAUNCH();
    prof_split(s, "wgrad_halo_reduce_kernel");
    const unsigned groups = (unsigned)(a.nt * 144);
    if (WI == 2) hipLaunchKernelGGL((wgrad_halo_reduce_kernel<2>), dim3(groups), dim3(512), 0, s, (const float*)slab, dw, d->in_c, a.tiles_c, a.nt, a.splits);
    else hipLaunchKernelGGL((wgrad_halo_reduce_kernel<1>), dim3(groups), dim3(512), 0, s, (const float*)slab, dw, d->in_c, a.tiles_c, a.nt, a.splits);
    RPE_CHECK_LAUNCH();
    return 0;
}
template int conv_wgrad_halo<bf16>(const rpe_conv_desc*, const void*, const void*, float*, void*, long, long*, hipStream_t);
template int conv_wgrad_halo<f16>(const rpe_conv_desc*, const void*, const void*, float*, void*, long, long*, hipStream_t);

}  // namespace rpe

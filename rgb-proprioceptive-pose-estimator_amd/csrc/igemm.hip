// Implicit-GEMM entry points: argument checks and dispatch to the per-type / per-mode instantiation units
// (igemm_nt_*.hip, igemm_tn_*.hip; kernels and configuration choice live in igemm_impl.h).
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "igemm.h"

namespace rpe {

thread_local char g_last_kernel[96] = "";
thread_local ProfHook g_prof_hook = {nullptr, nullptr};
thread_local int g_walk_mode = 0, g_walk_next = 0;

template <typename T, int MODE> int launch_nt_mode(NTArgs<T>& a, hipStream_t s);

template <typename T> int launch_nt(NTArgs<T>& a, int mode, hipStream_t s) {
    constexpr int CE = Elem<T>::kChunk;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: empty problem");
    if ((a.ldb % CE) || (((uintptr_t)a.Bw) & 15) || (((uintptr_t)a.A) & 15))
        return rpe_set_error(RPE_ERR_ALIGN, "igemm_nt: operands must be 16-byte aligned with ld a multiple of the 16-byte chunk");
    if (mode == MODE_DENSE && ((a.lda % CE) || (a.K % CE))) return rpe_set_error(RPE_ERR_ALIGN, "igemm_nt: dense lda/K must be chunk multiples");
    if ((mode == MODE_CONV || mode == MODE_HALO) && (a.g.C % (8 * CE))) return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: conv channels must be a multiple of 8 chunks");
    if (mode == MODE_HALO) {
        const Gather& g = a.g;
        if (sizeof(T) != 2 || g.R != 3 || g.S != 3 || g.sn != 1 || g.sd_shift != 0 || g.parity || g.H != g.Ho || g.W != g.Wo || g.halo_rt <= 0 ||
            g.halo_rt != halo_rows_per_tile(g.H, g.W) || g.halo_px != g.halo_rt * g.W || g.halo_px > 128 || g.halo_rows * (long)g.W != a.M ||
            (g.base_h != -1 && g.base_h != 1) || g.base_h != g.base_w || g.tap_sign != -g.base_h || a.K != 9 * g.C || (a.slab && a.splits > 1))
            return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: the halo form is for 3x3 / stride 1 / pad 1 convs of 16-bit types");
    }
    {
        // buffer-descriptor extents of the two DMA operands; rows that must read as zero use offset 2^31, so both stay below it
        // the activation operand may be of any size: the kernel's descriptors start at each tile's first row (dense) / first image
        // (conv), so only ONE tile's span has to stay below 2 GiB; the weights use one descriptor
        if (mode == MODE_DENSE) a.a_elems = (long)a.M * a.lda;
        const long bb = (long)a.N * a.ldb * (long)sizeof(T);
        const long tile_span = (mode == MODE_DENSE ? 256L * (a.lda > a.lda2 ? a.lda : a.lda2) : (256L / (a.g.Ho * a.g.Wo > 0 ? a.g.Ho * a.g.Wo : 1) + 2) * a.g.img_stride) * (long)sizeof(T);
        if (a.a_elems <= 0 || bb <= 0 || bb >= (1L << 31) || tile_span >= (1L << 31))
            return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: the weight operand or one tile of the activation operand spans 2 GiB or more");
        a.b_bytes = (unsigned)bb;
        if (a.A2) {
            if (mode != MODE_DENSE || a.role != 1 || a.K1 <= 0 || a.K1 >= a.K || (a.K1 % (8 * CE)) || (a.lda2 % CE) || (((uintptr_t)a.A2) & 15))
                return rpe_set_error(RPE_ERR_SHAPE, "igemm_nt: bad K-concatenated operand (dense data-gradient role, K1 a multiple of 8 chunks)");
        }
    }
    if (a.role != 2) {
        // conv roles run the vector-only epilogue: whole 16-byte chunks of 8 channels, aligned rows
        if ((a.N % 8) || (a.ldc % CE) || (((uintptr_t)a.C) & 15) || (a.addend && ((a.ld_add % CE) || (((uintptr_t)a.addend) & 15))) ||
            (a.bn_mode && ((((uintptr_t)a.bn_y) & 15) || (a.bn_a && (((uintptr_t)a.bn_a) & 15)))))
            return rpe_set_error(RPE_ERR_ALIGN, "igemm_nt: conv outputs need N % 8 == 0 and 16-byte aligned rows");
    }
    if (mode == MODE_STEM) return launch_nt_mode<T, MODE_STEM>(a, s);
    if (mode == MODE_DENSE) return launch_nt_mode<T, MODE_DENSE>(a, s);
    if (mode == MODE_HALO) {
        if constexpr (sizeof(T) == 2) return launch_nt_mode<T, MODE_HALO>(a, s);
        else return rpe_set_error(RPE_ERR_DTYPE, "igemm_nt: the halo form needs a 16-bit type");
    }
    return launch_nt_mode<T, MODE_CONV>(a, s);
}
template int launch_nt<float>(NTArgs<float>&, int, hipStream_t);
template int launch_nt<bf16>(NTArgs<bf16>&, int, hipStream_t);
template int launch_nt<f16>(NTArgs<f16>&, int, hipStream_t);

}  // namespace rpe

extern "C" const char* rpe_last_kernel_name(void) { return rpe::g_last_kernel; }
extern "C" void rpe_set_walk_direction(int mode) { rpe::g_walk_mode = (mode == 1 || mode == 2) ? mode : 0; rpe::g_walk_next = 0; }

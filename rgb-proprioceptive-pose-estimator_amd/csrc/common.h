// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the pose
// train-step path.  Wave size is 64 everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rpe_hip.h"

namespace rpe {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

// ---- element traits ---------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kChunk = 4;  // elements per 16-byte chunk
    static constexpr int kDtype = RPE_F32;
    static constexpr const char* kName = "f32";
    __device__ static inline float to_f(float v) { return v; }
    __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elem<bf16> {
    static constexpr int kChunk = 8;
    static constexpr int kDtype = RPE_BF16;
    static constexpr const char* kName = "bf16";
    __device__ static inline float to_f(bf16 v) { return (float)v; }
    __device__ static inline bf16 from_f(float v) { return (bf16)v; }
};
template <> struct Elem<f16> {   // IEEE half: the reduced-precision type of BASELINE config C5 (needs loss scaling: amp.py)
    static constexpr int kChunk = 8;
    static constexpr int kDtype = RPE_F16;
    static constexpr const char* kName = "f16";
    __device__ static inline float to_f(f16 v) { return (float)v; }
    __device__ static inline f16 from_f(float v) { return (f16)v; }
};

// 16 bytes of T viewed as raw dwords (used for staging through registers / LDS)
struct alignas(16) Chunk16 {
    u32x4 v;
};

__device__ inline float bf16_bits_to_f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// unpack a 16-byte chunk of T into floats (N = Elem<T>::kChunk)
template <typename T> __device__ inline void chunk_to_f(const u32x4& c, float* f);
template <> __device__ inline void chunk_to_f<float>(const u32x4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ inline void chunk_to_f<bf16>(const u32x4& c, float* f) {
    f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
    f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
    f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
    f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <> __device__ inline void chunk_to_f<f16>(const u32x4& c, float* f) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    const unsigned w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_ v = __builtin_convertvector(__builtin_bit_cast(f16x2, w[i]), f32x2_);
        f[2 * i] = v.x; f[2 * i + 1] = v.y;
    }
}
__device__ inline unsigned pack_f16x2(float lo, float hi) {   // round-to-nearest-even (not the rtz pack instruction)
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ inline unsigned pack_bf16x2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving); the element-wise cast + shift + or form compiled to four
    // instructions per pair, a visible share of the instruction-bound epilogues and streaming kernels
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
// two floats -> one dword of two 16-bit elements of T
template <typename T> __device__ inline unsigned pack2(float lo, float hi);
template <> __device__ inline unsigned pack2<bf16>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ inline unsigned pack2<f16>(float lo, float hi) { return pack_f16x2(lo, hi); }
template <> __device__ inline unsigned pack2<float>(float lo, float) { return __float_as_uint(lo); }   // (unused; keeps templates uniform)
template <typename T> __device__ inline u32x4 f_to_chunk(const float* f);
template <> __device__ inline u32x4 f_to_chunk<float>(const float* f) {
    u32x4 c; c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]); c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]);
    return c;
}
template <> __device__ inline u32x4 f_to_chunk<bf16>(const float* f) {
    u32x4 c; c.x = pack_bf16x2(f[0], f[1]); c.y = pack_bf16x2(f[2], f[3]); c.z = pack_bf16x2(f[4], f[5]); c.w = pack_bf16x2(f[6], f[7]);
    return c;
}

template <> __device__ inline u32x4 f_to_chunk<f16>(const float* f) {
    u32x4 c; c.x = pack_f16x2(f[0], f[1]); c.y = pack_f16x2(f[2], f[3]); c.z = pack_f16x2(f[4], f[5]); c.w = pack_f16x2(f[6], f[7]);
    return c;
}

// ---- fast unsigned division by a runtime constant (valid for n < 2^31) -----
struct FastDiv {
    unsigned d, mul, shr;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;  // ceil(log2 d)
    unsigned p = 31 + lg;
    f.mul = (unsigned)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
    return f;
}
__device__ inline unsigned fd_div(unsigned n, const FastDiv& f) { return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr); }

// ---- wave reductions -------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- walk direction ("boustrophedon") -----------------------------------------------------------------------------
// Workgroups are dispatched in blockIdx order and the big kernels hand every XCD one contiguous eighth of the row tiles / spans
// (blocks b, b + 8, .. share an XCD), so a kernel writes the END of every eighth last.  A consumer that walks its eighths DOWNWARDS
// reads the bytes the producer wrote last first -- the part of a tensor larger than the 256-MiB Infinity Cache that is still
// resident there (tools/micro/mall_order.hip: a chain of 411-MB read + write passes 5.7 -> 6.7 TB/s when directions alternate; one
// that walks upwards like its producer meets nothing but evicted lines).  rev = 0: upwards.  The host side keeps the walk MODE of the
// calling thread (rpe_set_walk_direction: 0 upwards, 1 downwards, 2 alternate): every launcher of a direction-aware kernel (nt_kernel,
// tn_kernel, the BatchNorm apply passes) takes the direction of its launch with walk_take(), which in mode 2 flips it for the next one.
__device__ __forceinline__ int xcd_remap_dir(int bid, int nwg, int rev) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int i = bid >> 3;
    if (rev) i = q + (x < r ? 1 : 0) - 1 - i;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
extern thread_local int g_walk_mode, g_walk_next;
inline int walk_take() {
    if (g_walk_mode != 2) return g_walk_mode;
    const int r = g_walk_next;
    g_walk_next ^= 1;
    return r;
}

// name of the kernel (symbol, short form) the last C-ABI call of this thread launched: what rpe_last_kernel_name() returns and
// the engine's per-symbol profile keys on (defined in igemm.hip)
extern thread_local char g_last_kernel[96];
inline void note_kernel(const char* name) { snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", name); }
// Per-KERNEL spans inside one entry point (the engine's per-symbol profile: rpe_resnet50_profile / bench.py's `roofline`): an entry
// point that launches several kernels calls prof_split(stream, name of the NEXT kernel) between them.  With a hook installed (thread
// local; the engine's PROF macro does it while profiling) the span of the kernels launched so far is closed under the name noted so
// far and a new one starts; without a hook it only notes the name.  Spans then carry the symbols rocprofv3 lists, one each.
struct ProfHook { void (*split)(void* ctx, hipStream_t s); void* ctx; };
extern thread_local ProfHook g_prof_hook;
inline void prof_split(hipStream_t s, const char* next_kernel) {
    if (g_prof_hook.split) g_prof_hook.split(g_prof_hook.ctx, s);
    note_kernel(next_kernel);
}

}  // namespace rpe

#define RPE_CHECK_LAUNCH()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return rpe_set_error_hip(e__, __FILE__, __LINE__); \
    } while (0)

extern "C" int rpe_set_error(int code, const char* msg);
int rpe_set_error_hip(hipError_t e, const char* file, int line);

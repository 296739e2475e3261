// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of dreg_nerf_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DREG_OK 0
#define DREG_EINVAL (-1)
#define DREG_ELAUNCH (-2)

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR) instead of five integer VALU
// instructions per value — the elementwise kernels (BatchNorm apply, softmax probabilities, conv epilogues) are VALU-bound
// on exactly this.  Same results as the integer formula for every finite value and infinity; NaN stays (quiet) NaN.
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 hw_bf16x2_t;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float hw_f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    const hw_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// n / d for n*d < 2^32 with magic = ceil(2^32 / d) (d >= 2); d == 1 passes magic = 0.
__device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t magic) {
    return magic ? __umulhi(n, magic) : n;
}
static inline uint32_t host_magic(uint32_t d) {
    return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Kernel-variant knobs.  The product library (libdreg_nerf_hip.so) has NO process-global mutable state: every knob is a compile-time
// constant there and its setter does not exist.  The same sources built with -DDREG_PROBE (libdreg_nerf_hip_probe.so, loaded explicitly by
// tools/ and the variant tests: include/dreg_nerf_probe.h) make them mutable and export the setters.
#ifdef DREG_PROBE
#define DREG_KNOB(type, name, value) static type name = value
#else
#define DREG_KNOB(type, name, value) static constexpr type name = value
#endif

#define DREG_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

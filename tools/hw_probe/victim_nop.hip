// Root-cause probe for the packed-fp32 co-execution fault: the LayerNorm-backward row body of victim.hip,
// compiled WITH the SLP vectoriser (v_pk_*_f32 present), in variants that put wait states / an unpacked VALU read between the
// s_waitcnt vmcnt(0) that releases the loaded registers and the first packed instruction that reads them.
//   VARIANT 0: as compiled (v_pk_add_f32 is the first reader, right behind s_waitcnt vmcnt(0))
//   VARIANT 1: s_nop 7 (8 wait states) on every loaded register          VARIANT 2: s_nop 0 (1 wait state)
//   VARIANT 3: 4 x s_nop 7                                               VARIANT 4: s_nop 7 on the statistics (mean, rstd) only
//   VARIANT 5: s_nop 7 on x only                                         VARIANT 6: an unpacked v_mov_b32 of every loaded register first
//   VARIANT 7: statistics loaded through the scalar path (s_load via readfirstlane'd address) instead of a uniform vector load
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int VARIANT>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, const unsigned short* __restrict__ dy, const float* __restrict__ g,
                                                 const float* __restrict__ stats, float* __restrict__ dx, int N)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
    const int r0 = blockIdx.x * 64;
    for (int row = r0 + wave; row < min(r0 + 64, N); row += 4) {
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __uint_as_float((uint32_t)dy[(size_t)row * 256 + lane * 4 + i] << 16);
        float mean, rstd;
        if (VARIANT == 7) {
            const float* sp = stats + 2 * (size_t)__builtin_amdgcn_readfirstlane(row);
            mean = sp[0]; rstd = sp[1];
        } else { mean = stats[2 * row]; rstd = stats[2 * row + 1]; }
        if (VARIANT == 1) asm volatile("s_nop 7" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(mean), "+v"(rstd));
        if (VARIANT == 2) asm volatile("s_nop 0" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(mean), "+v"(rstd));
        if (VARIANT == 3) asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(mean), "+v"(rstd));
        if (VARIANT == 4) asm volatile("s_nop 7" : "+v"(mean), "+v"(rstd));
        if (VARIANT == 5) asm volatile("s_nop 7" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        if (VARIANT == 6) asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3\n v_mov_b32 %4, %4\n v_mov_b32 %5, %5"
                                       : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(mean), "+v"(rstd));
        const float xh[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
        const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1 += d[i] * gv[i]; s2 += d[i] * gv[i] * xh[i]; }
        s1 = wsum(s1) * (1.f / 256.f); s2 = wsum(s2) * (1.f / 256.f);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rstd * (d[i] * gv[i] - s1 - xh[i] * s2);
        *reinterpret_cast<float4*>(dx + (size_t)row * 256 + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
extern "C" int probe_ln_variant(int variant, const void* x, const void* dy, const void* g, const void* stats, void* dx, int N, void* st)
{
#define GO(V) case V: hipLaunchKernelGGL(ln_kernel<V>, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)st, (const float*)x, (const unsigned short*)dy, (const float*)g, (const float*)stats, (float*)dx, N); break;
    switch (variant) { GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) default: return -1; }
    return (int)hipGetLastError();
}

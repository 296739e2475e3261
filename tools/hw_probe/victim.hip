// Self-checking victim kernels for the co-execution probe (tools/hw_probe/run.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

__global__ void fill_kernel(uint32_t* x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = hash32((uint32_t)i);
}
// V1: three loads in flight per iteration (like the LayerNorm backward: dwordx2, dwordx4, uniform dwordx2); compare with the expected bits.
__global__ __launch_bounds__(256) void load_check_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ y, const uint32_t* __restrict__ s, int rows, int* __restrict__ err, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it)
        for (int row = blockIdx.x * 64 + wave; row < min(blockIdx.x * 64 + 64, rows); row += 4) {
            const uint2 a = *reinterpret_cast<const uint2*>(y + (size_t)row * 128 + lane * 2);
            const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)row * 256 + lane * 4);
            const uint2 st = *reinterpret_cast<const uint2*>(s + (size_t)row * 2);
            const uint32_t b = (uint32_t)row * 256 + lane * 4, c = (uint32_t)row * 128 + lane * 2, d = (uint32_t)row * 2;
            int bad = 0;
            bad |= (v.x != hash32(b)) << 0; bad |= (v.y != hash32(b + 1)) << 1; bad |= (v.z != hash32(b + 2)) << 2; bad |= (v.w != hash32(b + 3)) << 3;
            bad |= (a.x != hash32(c)) << 4; bad |= (a.y != hash32(c + 1)) << 5; bad |= (st.x != hash32(d)) << 6; bad |= (st.y != hash32(d + 1)) << 7;
            if (bad) { const int k = atomicAdd(err, 1); if (k < 60) { err[4 + 4 * k] = row; err[5 + 4 * k] = lane; err[6 + 4 * k] = bad; err[7 + 4 * k] = (int)v.x; } }
        }
}
// V2: arithmetic only (packed fp32 + ds_bpermute reductions), no loads inside the loop
__global__ __launch_bounds__(256) void valu_check_kernel(int* __restrict__ err, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    float2 a = make_float2(seed + lane, seed - lane), b = make_float2(0.5f, 0.25f);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 c; c.x = a.x - b.x; c.y = a.y - b.x; c.x *= b.y; c.y *= b.y;
        float s = c.x + c.y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        acc += s; a.x += 1.f; a.y += 1.f;
    }
    // closed form: sum over lanes of ((seed+lane+it-0.5)+(seed-lane+it-0.5))*0.25 = 64*(2*seed+2*it-1)*0.25
    float ref = 0.f;
    for (int it = 0; it < iters; ++it) ref += 16.f * (2.f * seed + 2.f * it - 1.f);
    if (acc != ref) { const int k = atomicAdd(err + 1, 1); if (k < 8) { err[260 + 2 * k] = lane; err[261 + 2 * k] = __float_as_int(acc - ref); } }
}
// V3: the LayerNorm-backward row body (same load pattern and arithmetic) that also stores what it loaded and its xhat
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void ln_debug_kernel(const float* __restrict__ x, const unsigned short* __restrict__ dy, const float* __restrict__ g,
                                                       const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ xseen,
                                                       float* __restrict__ xhat, float* __restrict__ stseen, int N, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
    const int r0 = blockIdx.x * 64;
    for (int row = r0 + wave; row < min(r0 + 64, N); row += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __uint_as_float((uint32_t)dy[(size_t)row * 256 + lane * 4 + i] << 16);
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
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
        if (mode & 1) *reinterpret_cast<float4*>(xseen + (size_t)row * 256 + lane * 4) = v;
        if (mode & 2) *reinterpret_cast<float4*>(xhat + (size_t)row * 256 + lane * 4) = make_float4(xh[0], xh[1], xh[2], xh[3]);
        if (mode & 4) *reinterpret_cast<float2*>(stseen + (size_t)row * 128 + lane * 2) = make_float2(mean, rstd);
    }
}
// Aggressors without any memory traffic: a spin of MFMA instructions (two shapes) or of plain fp32 FMAs
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* __restrict__ out, int iters, int shape)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0; f32x16 d0 = {0}, d1 = {0};
    if (shape == 0)
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
    else
        for (int it = 0; it < iters; ++it) { d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0); }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
}
extern "C" {
int probe_mfma_spin(void* out, int iters, int shape, int blocks, void* st)
{ hipLaunchKernelGGL(mfma_spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, (float*)out, iters, shape); return (int)hipGetLastError(); }
int probe_ln_debug(const void* x, const void* dy, const void* g, const void* stats, void* dx, void* xseen, void* xhat, void* stseen, int N, int mode, void* st)
{ hipLaunchKernelGGL(ln_debug_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)st, (const float*)x, (const unsigned short*)dy, (const float*)g, (const float*)stats,
                     (float*)dx, (float*)xseen, (float*)xhat, (float*)stseen, N, mode); return (int)hipGetLastError(); }
int probe_fill(void* x, size_t n, void* st) { hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, (hipStream_t)st, (uint32_t*)x, n); return (int)hipGetLastError(); }
int probe_load_check(const void* x, const void* y, const void* s, int rows, void* err, int iters, void* st)
{ hipLaunchKernelGGL(load_check_kernel, dim3((rows + 63) / 64), dim3(256), 0, (hipStream_t)st, (const uint32_t*)x, (const uint32_t*)y, (const uint32_t*)s, rows, (int*)err, iters); return (int)hipGetLastError(); }
int probe_valu_check(void* err, int iters, int blocks, void* st)
{ hipLaunchKernelGGL(valu_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, (int*)err, iters, 3.0f); return (int)hipGetLastError(); }
}

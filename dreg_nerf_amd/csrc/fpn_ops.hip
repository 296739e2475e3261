// Memory-bound companions of the 3-D convolutions on the FPN3D path (gfx950): per-grid BatchNorm
// (statistics / apply / backward), 3x3x3 stride-2 max-pool, the top-down 2x nearest "downsample-sum"
// (backward of the upsample-add fused in the conv epilogue), channel column sums (bias gradients) and the
// fused trilinear gather that replaces F.interpolate + advanced indexing.
//
// Reference call sites: conerf/model/resnet3d.py:95-113,121-123,157-161 (BatchNorm3d/ReLU/MaxPool3d),
// conerf/model/feature_pyramid_net.py:58-61 (_upsample), conerf/register/nerf_regtr.py:138-147 (gather).
// All tensors are NDHWC, 16-byte channel granules per thread, fp32 math.
#include "common.h"
#include "../../include/dreg_nerf.h"   // dreg_bn_extra (and a signature check of every entry point defined here)
extern "C" int dreg_fill_zero(void* p, size_t bytes, void* stream);   // fpn_ops.hip (include/dreg_nerf.h)

template <typename T> struct Gran;
template <> struct Gran<float> {
    static constexpr int G = 4;
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) { float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    // the 16-byte granule as it lies in memory (kept in registers between two phases of a kernel)
    static __device__ __forceinline__ void unpack(const uint4& q, float (&v)[4]) { v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w); }
    static __device__ __forceinline__ uint4 pack(const float (&v)[4]) { return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])); }
};
template <> struct Gran<bf16_t> {
    static constexpr int G = 8;
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
        uint4 q = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2bf2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ void unpack(const uint4& q, float (&v)[8]) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2bf2(v[2 * i], v[2 * i + 1]);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// flat element index -> (channel granule, x, y, z, batch).  The grids here have < 2^32 granules: 32-bit divisions (~25 instructions each)
// instead of the 64-bit ones the size_t loop index would drag in (~120 each, four per element: the pooling kernels were instruction-bound)
__device__ __forceinline__ void decode_gxyzb(size_t i, int CG, int W, int H, int D, int& cg, int& x, int& y, int& z, int& b)
{
    if ((i >> 32) == 0) {
        uint32_t r = (uint32_t)i;
        uint32_t q = r / (uint32_t)CG; cg = (int)(r - q * (uint32_t)CG); r = q;
        q = r / (uint32_t)W; x = (int)(r - q * (uint32_t)W); r = q;
        q = r / (uint32_t)H; y = (int)(r - q * (uint32_t)H); r = q;
        q = r / (uint32_t)D; z = (int)(r - q * (uint32_t)D); b = (int)q;
    } else {
        size_t r = i;
        cg = (int)(r % CG); r /= CG;
        x = (int)(r % W); r /= W;
        y = (int)(r % H); r /= H;
        z = (int)(r % D); b = (int)(r / D);
    }
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------ BN statistics
// partial[b][chunk][c][2] = (sum x, sum x^2) over the chunk's voxels.  grid (chunks, B, slabs), 256 threads.
// MODE 0: plain sums of x.  MODE 1 (backward): sums of g and g*xhat with g = dy * (y > 0 if relu).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ mean_rstd,
    float* __restrict__ partial, int V, int C, int rows_per_chunk, int relu, const float* __restrict__ scale_shift = nullptr,
    T* __restrict__ gout = nullptr)   // MODE 1: the masked gradient g is also stored (it IS the residual branch's gradient; the apply pass then reads it instead of dy and y)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const int cgs = CG < 256 ? CG : 256;          // granule columns handled by this block
    const int rpi = 256 / cgs;                    // rows per iteration
    const int t = threadIdx.x;
    const int cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int v0 = chunk * rows_per_chunk, v1 = min(v0 + rows_per_chunk, V);
    float s1[G], s2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s1[i] = s2[i] = 0.f;
    float mu[G], rs[G];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < G; ++i) { mu[i] = mean_rstd[((size_t)b * C + cg * G + i) * 2]; rs[i] = mean_rstd[((size_t)b * C + cg * G + i) * 2 + 1]; }
    }
    // ReLU mask without reading y (y == nullptr; BatchNorms without a residual): y > 0  <=>  x*scale + shift > 0
    float sc[G], sh[G];
    const bool remask = MODE == 1 && relu && y == nullptr;
    if (remask) {
#pragma unroll
        for (int i = 0; i < G; ++i) { sc[i] = scale_shift[((size_t)b * C + cg * G + i) * 2]; sh[i] = scale_shift[((size_t)b * C + cg * G + i) * 2 + 1]; }
    }
    if (r0 < rpi) {
        constexpr int UN = 4;                            // rows in flight per thread (sums stay in row order: bit-identical to UN = 1)
        for (int v = v0 + r0; v < v1; v += rpi * UN) {
            float xv[UN][G], gv[UN][G], yv[UN][G];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < v1) {
                    const size_t off = ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G;
                    Gran<T>::ld(x + off, xv[u]);
                    if (MODE == 1) {
                        Gran<T>::ld(dy + off, gv[u]);
                        if (relu && !remask) Gran<T>::ld(y + off, yv[u]);
                    }
                }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < v1) {
                    if (MODE == 0) {
#pragma unroll
                        for (int i = 0; i < G; ++i) { s1[i] += xv[u][i]; s2[i] += xv[u][i] * xv[u][i]; }
                    } else {
                        if (remask) {
#pragma unroll
                            for (int i = 0; i < G; ++i) gv[u][i] = (xv[u][i] * sc[i] + sh[i]) > 0.f ? gv[u][i] : 0.f;
                        } else if (relu) {
#pragma unroll
                            for (int i = 0; i < G; ++i) gv[u][i] = yv[u][i] > 0.f ? gv[u][i] : 0.f;
                        }
                        if (gout) Gran<T>::st(gout + ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G, gv[u]);
#pragma unroll
                        for (int i = 0; i < G; ++i) { s1[i] += gv[u][i]; s2[i] += gv[u][i] * (xv[u][i] - mu[i]) * rs[i]; }
                    }
                }
        }
    }
    __shared__ float red[256][2 * 8 + 1];
#pragma unroll
    for (int i = 0; i < G; ++i) { red[t][i] = s1[i]; red[t][G + i] = s2[i]; }
    __syncthreads();
    if (t < cgs) {
        float a1[G], a2[G];
#pragma unroll
        for (int i = 0; i < G; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
        for (int r = 0; r < rpi; ++r)
#pragma unroll
            for (int i = 0; i < G; ++i) { a1[i] += red[r * cgs + t][i]; a2[i] += red[r * cgs + t][G + i]; }
        float* dst = partial + (((size_t)b * nchunks + chunk) * C + (size_t)cg * G) * 2;
#pragma unroll
        for (int i = 0; i < G; ++i) { dst[2 * i] = a1[i]; dst[2 * i + 1] = a2[i]; }
    }
}

// one block per channel, one wave per grid: finalise mean / biased var (lanes stride over the chunk partials, all loads of a
// wave in flight at once), emit scale/shift and (mean, rstd); thread 0 then updates the running stats sequentially over the grids
// (reference: one grid per BatchNorm call, src then tgt — nerf_regtr.py:135).
constexpr int BN_MAX_GRIDS = 256;
__global__ __launch_bounds__(512) void bn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ scale_shift, float* __restrict__ mean_rstd,
                                   int B, int nchunks, int C, int V, float eps, float momentum, int train)
{
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const float ga = gamma[c], be = beta[c];
    if (!train) {
        const float rstd = 1.0f / sqrtf(running_var[c] + eps);
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            scale_shift[((size_t)b * C + c) * 2] = ga * rstd;
            scale_shift[((size_t)b * C + c) * 2 + 1] = be - running_mean[c] * ga * rstd;
            mean_rstd[((size_t)b * C + c) * 2] = running_mean[c];
            mean_rstd[((size_t)b * C + c) * 2 + 1] = rstd;
        }
        return;
    }
    __shared__ double stat[BN_MAX_GRIDS][2];
    for (int b = wave; b < B; b += nw) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
        for (int k = lane; k < nchunks; k += 64) {
            const float2 p = *reinterpret_cast<const float2*>(partial + (((size_t)b * nchunks + k) * C + c) * 2);
            s1 += p.x; s2 += p.y;
        }
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        const double mean = s1 / V;
        double var = s2 / V - mean * mean;
        if (var < 0) var = 0;
        if (lane == 0) {
            const float rstd = 1.0f / sqrtf((float)var + eps);
            scale_shift[((size_t)b * C + c) * 2] = ga * rstd;
            scale_shift[((size_t)b * C + c) * 2 + 1] = be - (float)mean * ga * rstd;
            mean_rstd[((size_t)b * C + c) * 2] = (float)mean;
            mean_rstd[((size_t)b * C + c) * 2 + 1] = rstd;
            stat[b][0] = mean; stat[b][1] = var;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float rm = running_mean[c], rv = running_var[c];
        for (int b = 0; b < B; ++b) {
            const double mean = stat[b][0], var = stat[b][1];
            const float unbiased = V > 1 ? (float)(var * V / (V - 1)) : (float)var;
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
}

// y = [relu]( x * scale[b,c] + shift[b,c] [+ res] )
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, const T* __restrict__ res,
                                T* __restrict__ y, size_t total_gran, int V, int C, int relu)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total_gran; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const int b = (int)((i / CG) / V);
        float xv[G], rv[G];
        Gran<T>::ld(x + i * G, xv);
        if (res) Gran<T>::ld(res + i * G, rv);
        const float* ss = scale_shift + ((size_t)b * C + (size_t)cg * G) * 2;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            float v = xv[k] * ss[2 * k] + ss[2 * k + 1];
            if (res) v += rv[k];
            xv[k] = relu ? fmaxf(v, 0.f) : v;
        }
        Gran<T>::st(y + i * G, xv);
    }
}

// backward finalize (one block per channel, one wave per grid): per (b,c) coefficients c1 = sum(g)/V, c2 = sum(g*xhat)/V;
// dgamma/dbeta summed over the grids in order by thread 0.
__global__ __launch_bounds__(512) void bn_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ coef, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int B, int nchunks, int C, int V, int accumulate)
{
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __shared__ double stat[BN_MAX_GRIDS][2];
    for (int b = wave; b < B; b += nw) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
        for (int k = lane; k < nchunks; k += 64) {
            const float2 p = *reinterpret_cast<const float2*>(partial + (((size_t)b * nchunks + k) * C + c) * 2);
            s1 += p.x; s2 += p.y;
        }
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        if (lane == 0) {
            coef[((size_t)b * C + c) * 2] = (float)(s1 / V);
            coef[((size_t)b * C + c) * 2 + 1] = (float)(s2 / V);
            stat[b][0] = s1; stat[b][1] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double dg = 0.0, db = 0.0;
        for (int b = 0; b < B; ++b) { db += stat[b][0]; dg += stat[b][1]; }
        dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
        dbeta[c] = accumulate ? dbeta[c] + (float)db : (float)db;
    }
}

// dx = gamma*rstd * (g - c1 - xhat*c2),  g = dy * (y > 0 if relu);  dres = g (optional)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                    const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift,
                                    const float* __restrict__ coef, T* __restrict__ dx, T* __restrict__ dres,
                                    size_t total_gran, int V, int C, int relu)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total_gran; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const int b = (int)((i / CG) / V);
        float xv[G], gv[G], yv[G];
        Gran<T>::ld(x + i * G, xv);
        Gran<T>::ld(dy + i * G, gv);
        const size_t pc = ((size_t)b * C + (size_t)cg * G) * 2;
        if (relu && y == nullptr) {
#pragma unroll
            for (int k = 0; k < G; ++k) gv[k] = (xv[k] * scale_shift[pc + 2 * k] + scale_shift[pc + 2 * k + 1]) > 0.f ? gv[k] : 0.f;
        } else if (relu) { Gran<T>::ld(y + i * G, yv);
#pragma unroll
            for (int k = 0; k < G; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f; }
        if (dres) Gran<T>::st(dres + i * G, gv);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const float xh = (xv[k] - mean_rstd[pc + 2 * k]) * mean_rstd[pc + 2 * k + 1];
            xv[k] = scale_shift[pc + 2 * k] * (gv[k] - coef[pc + 2 * k] - xh * coef[pc + 2 * k + 1]);
        }
        Gran<T>::st(dx + i * G, xv);
    }
}

// Column-fixed forms of the two apply kernels (grid (chunks, B, slabs) like bn_partial_kernel): a thread keeps ONE channel granule,
// holds that granule's per-channel parameters in registers and walks down the rows of its chunk, UN rows in flight.  (The flat
// forms above re-load 2-6 parameter floats per element on every iteration, which made them instruction- rather than HBM-bound.)
// res_ss (optional): the residual is itself the output of a ReLU-free BatchNorm that was not applied (the downsample branch of a
// bottleneck's first block, resnet3d.py:104-110): res holds that layer's INPUT and res_ss its (scale, shift); its output is formed here,
// rounded to the activation dtype exactly as the separate apply pass stored it (bit-identical, one 2 x tensor pass less).
template <typename T, int UN>
__global__ __launch_bounds__(256) void bn_apply_cols_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, const T* __restrict__ res,
                                                            T* __restrict__ y, int V, int C, int rows_per_chunk, int relu,
                                                            const float* __restrict__ res_ss = nullptr)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    if (r0 >= rpi || cg >= CG) return;
    const int v0 = blockIdx.x * rows_per_chunk, v1 = min(v0 + rows_per_chunk, V);
    float sc[G], sh[G], rsc[G], rsh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        sc[i] = scale_shift[((size_t)b * C + cg * G + i) * 2]; sh[i] = scale_shift[((size_t)b * C + cg * G + i) * 2 + 1];
        if (res_ss) { rsc[i] = res_ss[((size_t)b * C + cg * G + i) * 2]; rsh[i] = res_ss[((size_t)b * C + cg * G + i) * 2 + 1]; }
    }
    for (int v = v0 + r0; v < v1; v += rpi * UN) {
        float xv[UN][G], rv[UN][G];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < v1) {
                const size_t off = ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G;
                Gran<T>::ld(x + off, xv[u]);
                if (res) Gran<T>::ld(res + off, rv[u]);
            }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < v1) {
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    float o = xv[u][k] * sc[k] + sh[k];
                    if (res) {
                        float r = rv[u][k];
                        if (res_ss) { r = r * rsc[k] + rsh[k]; if constexpr (sizeof(T) == 2) r = bf2f(f2bf(r)); }
                        o += r;
                    }
                    xv[u][k] = relu ? fmaxf(o, 0.f) : o;
                }
                Gran<T>::st(y + ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G, xv[u]);
            }
    }
}
template <typename T, int UN>
__global__ __launch_bounds__(256) void bn_bwd_apply_cols_kernel(const T* __restrict__ x, const T* dy, const T* __restrict__ y,
                                                                const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift,
                                                                const float* __restrict__ coef, T* __restrict__ dx, T* dres,
                                                                int V, int C, int rows_per_chunk, int relu, int g_stored = 0)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    if (r0 >= rpi || cg >= CG) return;
    if (g_stored) { dy = dres; dres = nullptr; relu = 0; }     // the statistics pass stored the masked gradient in dres: one tensor read less
    const int v0 = blockIdx.x * rows_per_chunk, v1 = min(v0 + rows_per_chunk, V);
    float mu[G], rs[G], sc[G], sh[G], c1[G], c2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t pc = ((size_t)b * C + cg * G + i) * 2;
        mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1]; c1[i] = coef[pc]; c2[i] = coef[pc + 1];
    }
    const bool remask = relu && y == nullptr;
    for (int v = v0 + r0; v < v1; v += rpi * UN) {
        float xv[UN][G], gv[UN][G], yv[UN][G];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < v1) {
                const size_t off = ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G;
                Gran<T>::ld(x + off, xv[u]);
                Gran<T>::ld(dy + off, gv[u]);
                if (relu && !remask) Gran<T>::ld(y + off, yv[u]);
            }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < v1) {
                const size_t off = ((size_t)b * V + v + u * rpi) * C + (size_t)cg * G;
                if (remask) {
#pragma unroll
                    for (int k = 0; k < G; ++k) gv[u][k] = (xv[u][k] * sc[k] + sh[k]) > 0.f ? gv[u][k] : 0.f;
                } else if (relu) {
#pragma unroll
                    for (int k = 0; k < G; ++k) gv[u][k] = yv[u][k] > 0.f ? gv[u][k] : 0.f;
                }
                if (dres) Gran<T>::st(dres + off, gv[u]);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const float xh = (xv[u][k] - mu[k]) * rs[k];
                    xv[u][k] = sc[k] * (gv[u][k] - c1[k] - xh * c2[k]);
                }
                Gran<T>::st(dx + off, xv[u]);
            }
    }
}

// ------------------------------------------------------------------------------------------------ small volumes: fused passes
// The 16^3 / 8^3 / 4^3 levels of the ResNet (layer2-4: 42 of the 53 BatchNorms, V = 4096 / 512 / 64 voxels per grid) are launch
// latency in the three-kernel form (statistics, finalize, apply: ~25 us for a few us of traffic).  Statistics are per GRID and per
// channel, so a workgroup that owns (grid b, 4 channel granules = 64 bytes of every row) needs nobody else for the normalisation:
// statistics of its rows (fp32 per thread, xor-shuffle tree over the 16 row lanes of a wave, the four waves summed in fp64 in a
// fixed order), scale / shift, then the apply pass over the same rows (L2-warm) — ONE launch instead of two, no cross-workgroup
// coupling.  What does couple the grids — the running statistics (sequential over the grids, as the reference's one-grid-per-call
// updates: nerf_regtr.py:135) and dgamma / dbeta (sum over the grids) — is a second, tiny launch over the per-grid results.
// (A "last block arrives" single launch was measured 3x slower than three launches: the agent-scope fence writes the L2 back; a
// single launch that walks the grids in sequence starves the chip: 8-64 workgroups with 8 dependent passes each.)
constexpr int BNS_COLS = 4, BNS_ROWS = 64;      // 256 threads = 4 granule columns x 64 row lanes
template <int G> __device__ __forceinline__ void bns_reduce(float (&s1)[G], float (&s2)[G], float (*red)[BNS_COLS][16], int t)
{
    // lanes of a wave: col = lane & 3, row lane = lane >> 2 (16 per wave): xor over lane bits 2..5
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
        for (int i = 0; i < G; ++i) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
    const int lane = t & 63, wave = t >> 6;
    if (lane < BNS_COLS) {
#pragma unroll
        for (int i = 0; i < G; ++i) { red[wave][lane][i] = s1[i]; red[wave][lane][8 + i] = s2[i]; }
    }
}
// NR > 0 (V == NR * 64 rows per grid: 8^3 -> 8, 4^3 -> 1): a thread's rows are loaded ONCE, all loads in flight together, and stay in
// registers between the statistics and the apply phase (the general form, NR = 0, walks the rows twice, four loads in flight): the
// kernels are short enough that the second walk and the exposed latencies were most of their time.  Same sums in the same order.
// part (NR > 0 only): x has not been formed yet — it is the sum of `nsplit` fp32 slices part[s * slice + element] left by a split-K convolution
// (dreg_conv3d_igemm_defer's sk_nsplit / sk_slice out arguments, handed on as dreg_bn_extra.splitk_part); the slices are added in ascending order, rounded to T and stored to x exactly as the convolution's own
// reduce pass would have (the backward pass reads x), and the statistics use the rounded values: bit-identical, one launch less.
template <typename T>
__device__ __forceinline__ uint4 bns_sum_slices(const float* __restrict__ part, int nsplit, size_t slice, size_t off)
{
    constexpr int G = Gran<T>::G;
    float v[G];
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* p = part + (size_t)sp * slice + off;
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        if constexpr (G == 8) { const float4 b = *reinterpret_cast<const float4*>(p + 4); v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w; }
    }
    return Gran<T>::pack(v);
}
template <typename T, int NR = 0>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ scale_shift, float* __restrict__ mean_rstd, float* __restrict__ var_out,
                                                           int V, int C, float eps, int relu,
                                                           const float* __restrict__ part = nullptr, int nsplit = 0, size_t slice = 0)
{
    constexpr int G = Gran<T>::G;
    const int t = threadIdx.x, col = t & (BNS_COLS - 1), rl = t >> 2, b = blockIdx.y;
    const int c0 = (blockIdx.x * BNS_COLS + col) * G;               // first channel of this thread's granule
    __shared__ float red[4][BNS_COLS][16];
    __shared__ float s_sc[BNS_COLS * 8], s_sh[BNS_COLS * 8];
    float s1[G], s2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s1[i] = s2[i] = 0.f;
    constexpr int NRR = NR > 0 ? NR : 1;
    uint4 xr[NRR], rr[NRR];
    if constexpr (NR > 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t off = ((size_t)b * V + rl + j * BNS_ROWS) * C + c0;
            if (part) { xr[j] = bns_sum_slices<T>(part, nsplit, slice, off); *reinterpret_cast<uint4*>(const_cast<T*>(x) + off) = xr[j]; }
            else xr[j] = *reinterpret_cast<const uint4*>(x + off);
            if (res) rr[j] = *reinterpret_cast<const uint4*>(res + off);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float xv[G];
            Gran<T>::unpack(xr[j], xv);
#pragma unroll
            for (int i = 0; i < G; ++i) { s1[i] += xv[i]; s2[i] += xv[i] * xv[i]; }
        }
    } else {
#pragma unroll 4
    for (int v = rl; v < V; v += BNS_ROWS) {
        float xv[G];
        Gran<T>::ld(x + ((size_t)b * V + v) * C + c0, xv);
#pragma unroll
        for (int i = 0; i < G; ++i) { s1[i] += xv[i]; s2[i] += xv[i] * xv[i]; }
    }
    }
    bns_reduce<G>(s1, s2, red, t);
    __syncthreads();
    if (t < BNS_COLS * G) {
        const int cc = t / G, ci = t % G, myc = blockIdx.x * BNS_COLS * G + t;
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a1 += red[w][cc][ci]; a2 += red[w][cc][8 + ci]; }
        const double mean = a1 / V;
        double var = a2 / V - mean * mean;
        if (var < 0) var = 0;
        const float rstd = 1.0f / sqrtf((float)var + eps);
        const float ga = gamma[myc], be = beta[myc];
        const float sc = ga * rstd, sh = be - (float)mean * ga * rstd;
        scale_shift[((size_t)b * C + myc) * 2] = sc;
        scale_shift[((size_t)b * C + myc) * 2 + 1] = sh;
        mean_rstd[((size_t)b * C + myc) * 2] = (float)mean;
        mean_rstd[((size_t)b * C + myc) * 2 + 1] = rstd;
        var_out[(size_t)b * C + myc] = (float)var;
        s_sc[t] = sc; s_sh[t] = sh;
    }
    __syncthreads();
    float sc[G], sh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { sc[i] = s_sc[col * G + i]; sh[i] = s_sh[col * G + i]; }
    if constexpr (NR > 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t off = ((size_t)b * V + rl + j * BNS_ROWS) * C + c0;
            float xv[G], rvv[G];
            Gran<T>::unpack(xr[j], xv);
            if (res) Gran<T>::unpack(rr[j], rvv);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                float o = xv[i] * sc[i] + sh[i];
                if (res) o += rvv[i];
                xv[i] = relu ? fmaxf(o, 0.f) : o;
            }
            Gran<T>::st(y + off, xv);
        }
        return;
    }
#pragma unroll 4
    for (int v = rl; v < V; v += BNS_ROWS) {
        const size_t off = ((size_t)b * V + v) * C + c0;
        float xv[G], rvv[G];
        Gran<T>::ld(x + off, xv);
        if (res) Gran<T>::ld(res + off, rvv);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            float o = xv[i] * sc[i] + sh[i];
            if (res) o += rvv[i];
            xv[i] = relu ? fmaxf(o, 0.f) : o;
        }
        Gran<T>::st(y + off, xv);
    }
}
// running statistics from the per-grid (mean, biased variance), sequentially over the grids: one thread per channel
__global__ void bn_running_update_kernel(const float* __restrict__ mean_rstd, const float* __restrict__ var, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, int B, int V, int C, float momentum)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    for (int b = 0; b < B; ++b) {
        const float mean = mean_rstd[((size_t)b * C + c) * 2], vb = var[(size_t)b * C + c];
        const float unbiased = V > 1 ? (float)((double)vb * V / (V - 1)) : vb;
        rm = (1.f - momentum) * rm + momentum * mean;
        rv = (1.f - momentum) * rv + momentum * unbiased;
    }
    running_mean[c] = rm; running_var[c] = rv;
}

// backward of the same: per grid c1 = sum(g)/V, c2 = sum(g xhat)/V, dx = gamma rstd (g - c1 - xhat c2), dres = g; the per-grid sums go to
// sums[b][c][2] and a second tiny launch adds them over the grids in order into dgamma / dbeta
template <typename T, int NR = 0>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                           const float* __restrict__ scale_shift, const float* __restrict__ mean_rstd,
                                                           T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ sums,
                                                           int V, int C, int relu,
                                                           const float* __restrict__ part = nullptr, int nsplit = 0, size_t slice = 0)   // NR > 0: dy = the rounded sum of these split-K slices (never stored: nobody else reads it)
{
    constexpr int G = Gran<T>::G;
    const int t = threadIdx.x, col = t & (BNS_COLS - 1), rl = t >> 2, b = blockIdx.y;
    const int c0 = (blockIdx.x * BNS_COLS + col) * G;
    __shared__ float red[4][BNS_COLS][16];
    __shared__ float s_c1[BNS_COLS * 8], s_c2[BNS_COLS * 8];
    const bool remask = relu && y == nullptr;
    float mu[G], rs[G], sc[G], sh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t pc = ((size_t)b * C + c0 + i) * 2;
        mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1];
    }
    float s1[G], s2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s1[i] = s2[i] = 0.f;
    constexpr int NRR = NR > 0 ? NR : 1;
    uint4 xr[NRR], gr[NRR];                       // NR > 0: this thread's rows of x and of the MASKED gradient, as they lie in memory
    if constexpr (NR > 0) {
        uint4 yr[NRR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t off = ((size_t)b * V + rl + j * BNS_ROWS) * C + c0;
            xr[j] = *reinterpret_cast<const uint4*>(x + off);
            gr[j] = part ? bns_sum_slices<T>(part, nsplit, slice, off) : *reinterpret_cast<const uint4*>(dy + off);
            if (relu && !remask) yr[j] = *reinterpret_cast<const uint4*>(y + off);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float xv[G], gv[G], yv[G];
            Gran<T>::unpack(xr[j], xv);
            Gran<T>::unpack(gr[j], gv);
            if (relu && !remask) Gran<T>::unpack(yr[j], yv);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if (remask) gv[i] = (xv[i] * sc[i] + sh[i]) > 0.f ? gv[i] : 0.f;
                else if (relu) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
                s1[i] += gv[i]; s2[i] += gv[i] * (xv[i] - mu[i]) * rs[i];
            }
            if (relu) gr[j] = Gran<T>::pack(gv);      // exact: every element is the stored value or zero
        }
    } else {
#pragma unroll 4
    for (int v = rl; v < V; v += BNS_ROWS) {
        const size_t off = ((size_t)b * V + v) * C + c0;
        float xv[G], gv[G], yv[G];
        Gran<T>::ld(x + off, xv);
        Gran<T>::ld(dy + off, gv);
        if (relu && !remask) Gran<T>::ld(y + off, yv);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (remask) gv[i] = (xv[i] * sc[i] + sh[i]) > 0.f ? gv[i] : 0.f;
            else if (relu) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
            s1[i] += gv[i]; s2[i] += gv[i] * (xv[i] - mu[i]) * rs[i];
        }
    }
    }
    bns_reduce<G>(s1, s2, red, t);
    __syncthreads();
    if (t < BNS_COLS * G) {
        const int cc = t / G, ci = t % G, myc = blockIdx.x * BNS_COLS * G + t;
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a1 += red[w][cc][ci]; a2 += red[w][cc][8 + ci]; }
        s_c1[t] = (float)(a1 / V); s_c2[t] = (float)(a2 / V);
        sums[((size_t)b * C + myc) * 2] = (float)a1; sums[((size_t)b * C + myc) * 2 + 1] = (float)a2;
    }
    __syncthreads();
    float c1[G], c2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { c1[i] = s_c1[col * G + i]; c2[i] = s_c2[col * G + i]; }
    if constexpr (NR > 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t off = ((size_t)b * V + rl + j * BNS_ROWS) * C + c0;
            float xv[G], gv[G];
            Gran<T>::unpack(xr[j], xv);
            Gran<T>::unpack(gr[j], gv);
            if (dres) *reinterpret_cast<uint4*>(dres + off) = gr[j];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const float xh = (xv[i] - mu[i]) * rs[i];
                xv[i] = sc[i] * (gv[i] - c1[i] - xh * c2[i]);
            }
            Gran<T>::st(dx + off, xv);
        }
        return;
    }
#pragma unroll 4
    for (int v = rl; v < V; v += BNS_ROWS) {
        const size_t off = ((size_t)b * V + v) * C + c0;
        float xv[G], gv[G], yv[G];
        Gran<T>::ld(x + off, xv);
        Gran<T>::ld(dy + off, gv);
        if (relu && !remask) Gran<T>::ld(y + off, yv);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (remask) gv[i] = (xv[i] * sc[i] + sh[i]) > 0.f ? gv[i] : 0.f;
            else if (relu) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
        }
        if (dres) Gran<T>::st(dres + off, gv);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const float xh = (xv[i] - mu[i]) * rs[i];
            xv[i] = sc[i] * (gv[i] - c1[i] - xh * c2[i]);
        }
        Gran<T>::st(dx + off, xv);
    }
}
__global__ void bn_param_grad_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double db = 0.0, dg = 0.0;
    for (int b = 0; b < B; ++b) { db += sums[((size_t)b * C + c) * 2]; dg += sums[((size_t)b * C + c) * 2 + 1]; }
    dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
    dbeta[c] = accumulate ? dbeta[c] + (float)db : (float)db;
}
// The two tiny cross-grid launches of every small BatchNorm (42 of the 53 layers: running statistics in the forward pass, dgamma / dbeta
// in the backward pass) batched over the layers of a pass: one launch each instead of 28.  Same arithmetic per channel.
struct BnTailDesc {              // 48 bytes
    const float* a;              // running update: mean_rstd [B][C][2];   parameter gradients: sums [B][C][2]
    const float* b;              // running update: var [B][C];            parameter gradients: unused
    float* o0; float* o1;        // running update: running_mean / running_var;   parameter gradients: dgamma / dbeta
    int B, V, C, block0;         // block0: first workgroup (256 channels each) of this layer in the batched launch
};
static_assert(sizeof(BnTailDesc) == 48, "descriptor layout is part of the ABI");
template <int WHAT>
__global__ void bn_tail_batched_kernel(const BnTailDesc* __restrict__ descs, int n, int block_base, float momentum, int accumulate)
{
    const int blk = (int)blockIdx.x + block_base;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= blk) lo = mid; else hi = mid - 1;
    }
    const BnTailDesc d = descs[lo];
    const int c = (blk - d.block0) * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    if (WHAT == 0) {
        float rm = d.o0[c], rv = d.o1[c];
        for (int b = 0; b < d.B; ++b) {
            const float mean = d.a[((size_t)b * d.C + c) * 2], vb = d.b[(size_t)b * d.C + c];
            const float unbiased = d.V > 1 ? (float)((double)vb * d.V / (d.V - 1)) : vb;
            rm = (1.f - momentum) * rm + momentum * mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        d.o0[c] = rm; d.o1[c] = rv;
    } else {
        double db = 0.0, dg = 0.0;
        for (int b = 0; b < d.B; ++b) { db += d.a[((size_t)b * d.C + c) * 2]; dg += d.a[((size_t)b * d.C + c) * 2 + 1]; }
        d.o0[c] = accumulate ? d.o0[c] + (float)dg : (float)dg;
        d.o1[c] = accumulate ? d.o1[c] + (float)db : (float)db;
    }
}
DREG_KNOB(int, g_bn_small_regs, 1);  // tuning (include/dreg_nerf_probe.h): the small BatchNorms keep their rows in registers between the two phases (8^3 / 4^3 volumes)
DREG_KNOB(int, g_bn_store_g, 1);     // tuning (include/dreg_nerf_probe.h): residual BatchNorm backward stores the masked gradient in the statistics pass
DREG_KNOB(int, g_bn_debug_skip, 0);      // measurement only (tools/ab_step.py): bit 0 / bit 1 leave out the forward / backward statistics pass (stale statistics)
DREG_KNOB(int, g_bn_small_maxv, 512);    // tuning (include/dreg_nerf_probe.h): largest per-grid volume served by the fused kernels, 0 = never
static inline bool bn_small_ok(int B, int V, int C, int G) { return V >= 2 && V <= g_bn_small_maxv && C % (BNS_COLS * G) == 0 && B >= 1; }

// ------------------------------------------------------------------------------------------------ max-pool 3^3 s2 p1
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ arg,
                                   int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)B * Do * Ho * Wo * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int cg, ox, oy, oz, b;
        decode_gxyzb(i, CG, Wo, Ho, Do, cg, ox, oy, oz, b);
        float best[G];
        int bi[G];
#pragma unroll
        for (int k = 0; k < G; ++k) { best[k] = -INFINITY; bi[k] = 0; }
        for (int dz = 0; dz < 3; ++dz) {
            const int z = oz * 2 - 1 + dz; if ((unsigned)z >= (unsigned)Di) continue;
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = oy * 2 - 1 + dy; if ((unsigned)yy >= (unsigned)Hi) continue;
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = ox * 2 - 1 + dx; if ((unsigned)xx >= (unsigned)Wi) continue;
                    float v[G];
                    Gran<T>::ld(x + ((((size_t)b * Di + z) * Hi + yy) * Wi + xx) * C + (size_t)cg * G, v);
                    const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
                    for (int k = 0; k < G; ++k) if (v[k] > best[k]) { best[k] = v[k]; bi[k] = tap; }
                }
            }
        }
        Gran<T>::st(y + i * G, best);
#pragma unroll
        for (int k = 0; k < G; ++k) arg[i * G + k] = (uint8_t)bi[k];
    }
}

// gather-form backward: every input voxel sums the dy of the windows whose arg-max tap points at it
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ arg, T* __restrict__ dx,
                                   int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int accumulate)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)B * Di * Hi * Wi * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int cg, ix, iy, iz, b;
        decode_gxyzb(i, CG, Wi, Hi, Di, cg, ix, iy, iz, b);
        float acc[G];
#pragma unroll
        for (int k = 0; k < G; ++k) acc[k] = 0.f;
        // The windows that contain input voxel i: per axis the one starting at floor(i/2) and, for odd i, the next one — at most eight.
        // They are enumerated statically (same z, y, x ascending order as a loop nest), so all their loads — gradient granule and the
        // G arg-max bytes as ONE word — are in flight together instead of one dependent pair per loop iteration.
        const int oz0 = iz >> 1, oy0 = iy >> 1, ox0 = ix >> 1;
        uint4 gq[8];
        uint2 aq[8];
        bool ok[8];
        int tp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int uz = u >> 2, uy = (u >> 1) & 1, ux = u & 1;
            const int oz = oz0 + uz, oy = oy0 + uy, ox = ox0 + ux;
            ok[u] = (!uz || (iz & 1)) && (!uy || (iy & 1)) && (!ux || (ix & 1)) && oz < Do && oy < Ho && ox < Wo;
            tp[u] = ((iz - 2 * oz + 1) * 3 + (iy - 2 * oy + 1)) * 3 + (ix - 2 * ox + 1);
            if (ok[u]) {
                const size_t o = ((((size_t)b * Do + oz) * Ho + oy) * Wo + ox) * C + (size_t)cg * G;
                gq[u] = *reinterpret_cast<const uint4*>(dy + o);
                if constexpr (G == 8) aq[u] = *reinterpret_cast<const uint2*>(arg + o);
                else aq[u] = make_uint2(*reinterpret_cast<const uint32_t*>(arg + o), 0u);
            }
        }
        uint4 pq = make_uint4(0, 0, 0, 0);
        if (accumulate) pq = *reinterpret_cast<const uint4*>(dx + i * G);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ok[u]) {
                float g[G];
                Gran<T>::unpack(gq[u], g);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const uint32_t a = ((k < 4 ? aq[u].x : aq[u].y) >> (8 * (k & 3))) & 0xffu;
                    if ((int)a == tp[u]) acc[k] += g[k];
                }
            }
        if (accumulate) {       // a second contribution to an existing gradient (the FPN lateral's): one read-modify-write, no temporary
            float prev[G];
            Gran<T>::unpack(pq, prev);
#pragma unroll
            for (int k = 0; k < G; ++k) acc[k] += prev[k];
        }
        Gran<T>::st(dx + i * G, acc);
    }
}

// ------------------------------------------------------------------------------------------------ stem: BatchNorm + ReLU + max-pool in one pass
// resnet3d.py:118-123 runs conv1 -> bn1 -> relu -> maxpool; the full-resolution activation between them (8 x 64^3 x 64 bf16 =
// 268 MB) is only ever read by the pool.  Forward: the pool kernel normalises on the fly (y rounded to the activation dtype exactly as
// the unfused pair stores it, same tap order, same strict comparison: pooled values and arg-max taps are bit-identical) and the
// 268 MB write + read disappear.  Backward: the pooled gradient is un-pooled on the fly (gather form: every input voxel sums the
// windows whose arg-max points at it) inside the statistics pass and again inside the apply pass — the dense dy is never written.
template <typename T>
__global__ void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, T* __restrict__ y, uint8_t* __restrict__ arg,
                                           int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)B * Do * Ho * Wo * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int cg, ox, oy, oz, b;
        decode_gxyzb(i, CG, Wo, Ho, Do, cg, ox, oy, oz, b);
        float sc[G], sh[G], best[G];
        int bi[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            sc[k] = scale_shift[((size_t)b * C + cg * G + k) * 2]; sh[k] = scale_shift[((size_t)b * C + cg * G + k) * 2 + 1];
            best[k] = -INFINITY; bi[k] = 0;
        }
        for (int dz = 0; dz < 3; ++dz) {
            const int z = oz * 2 - 1 + dz; if ((unsigned)z >= (unsigned)Di) continue;
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = oy * 2 - 1 + dy; if ((unsigned)yy >= (unsigned)Hi) continue;
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = ox * 2 - 1 + dx; if ((unsigned)xx >= (unsigned)Wi) continue;
                    float v[G];
                    Gran<T>::ld(x + ((((size_t)b * Di + z) * Hi + yy) * Wi + xx) * C + (size_t)cg * G, v);
                    const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        float o = v[k] * sc[k] + sh[k];
                        o = relu ? fmaxf(o, 0.f) : o;
                        if constexpr (sizeof(T) == 2) o = bf2f(f2bf(o));         // what the unfused BatchNorm stores
                        if (o > best[k]) { best[k] = o; bi[k] = tap; }
                    }
                }
            }
        }
        Gran<T>::st(y + i * G, best);
#pragma unroll
        for (int k = 0; k < G; ++k) arg[i * G + k] = (uint8_t)bi[k];
    }
}
// g[k] = (sum of the pooled gradients whose arg-max is this input voxel) masked by the ReLU (x*scale + shift > 0)
template <typename T>
__device__ __forceinline__ void unpool_gather(const T* __restrict__ dp, const uint8_t* __restrict__ arg, int b, int iz, int iy, int ix,
                                              int Do, int Ho, int Wo, int C, int c0, float (&acc)[Gran<T>::G])
{
    constexpr int G = Gran<T>::G;
#pragma unroll
    for (int k = 0; k < G; ++k) acc[k] = 0.f;
    for (int oz = max(0, iz / 2); oz <= min(Do - 1, (iz + 1) / 2); ++oz) {
        const int dz = iz - 2 * oz + 1; if (dz < 0 || dz > 2) continue;
        for (int oy = max(0, iy / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy) {
            const int dyy = iy - 2 * oy + 1; if (dyy < 0 || dyy > 2) continue;
            for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
                const int dxx = ix - 2 * ox + 1; if (dxx < 0 || dxx > 2) continue;
                const int tap = (dz * 3 + dyy) * 3 + dxx;
                const size_t o = ((((size_t)b * Do + oz) * Ho + oy) * Wo + ox) * C + (size_t)c0;
                float g[G];
                Gran<T>::ld(dp + o, g);
                const uint2 a8 = *reinterpret_cast<const uint2*>(arg + o);      // G <= 8 arg-max bytes
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const uint32_t a = ((k < 4 ? a8.x : a8.y) >> (8 * (k & 3))) & 0xffu;
                    if ((int)a == tap) acc[k] += g[k];
                }
            }
        }
    }
}
// MODE 0: partial[b][chunk][c][2] = (sum g, sum g*xhat) over the chunk's input voxels (grid as bn_partial_kernel);
// MODE 1: dx = scale * (g - c1 - xhat*c2)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dp, const uint8_t* __restrict__ arg,
                                                          const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift,
                                                          const float* __restrict__ coef, float* __restrict__ partial, T* __restrict__ dx,
                                                          int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int rows_per_chunk, int relu)
{
    constexpr int G = Gran<T>::G;
    static_assert(G == 8, "arg-max bytes are read eight at a time");
    const int V = Di * Hi * Wi;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int v0 = chunk * rows_per_chunk, v1 = min(v0 + rows_per_chunk, V);
    float mu[G], rs[G], sc[G], sh[G], c1[G], c2[G], s1[G], s2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t pc = ((size_t)b * C + cg * G + i) * 2;
        mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1];
        if (MODE == 1) { c1[i] = coef[pc]; c2[i] = coef[pc + 1]; }
        s1[i] = s2[i] = 0.f;
    }
    if (r0 < rpi && cg < CG)
        for (int v = v0 + r0; v < v1; v += rpi) {
            const int ix = v % Wi, iy = (v / Wi) % Hi, iz = v / (Wi * Hi);
            float xv[G], g[G];
            const size_t off = ((size_t)b * V + v) * C + (size_t)cg * G;
            Gran<T>::ld(x + off, xv);
            unpool_gather<T>(dp, arg, b, iz, iy, ix, Do, Ho, Wo, C, cg * G, g);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (relu && !((xv[k] * sc[k] + sh[k]) > 0.f)) g[k] = 0.f;
                const float xh = (xv[k] - mu[k]) * rs[k];
                if (MODE == 0) { s1[k] += g[k]; s2[k] += g[k] * xh; }
                else xv[k] = sc[k] * (g[k] - c1[k] - xh * c2[k]);
            }
            if (MODE == 1) Gran<T>::st(dx + off, xv);
        }
    if (MODE == 0) {
        __shared__ float red[256][2 * 8 + 1];
#pragma unroll
        for (int i = 0; i < G; ++i) { red[t][i] = s1[i]; red[t][G + i] = s2[i]; }
        __syncthreads();
        if (t < cgs) {
            float a1[G], a2[G];
#pragma unroll
            for (int i = 0; i < G; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
            for (int r = 0; r < rpi; ++r)
#pragma unroll
                for (int i = 0; i < G; ++i) { a1[i] += red[r * cgs + t][i]; a2[i] += red[r * cgs + t][G + i]; }
            float* dst = partial + (((size_t)b * nchunks + chunk) * C + (size_t)cg * G) * 2;
#pragma unroll
            for (int i = 0; i < G; ++i) { dst[2 * i] = a1[i]; dst[2 * i + 1] = a2[i]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 2x downsample-sum
// out[b,z,y,x,:] = sum over the (cropped) 2x2x2 children of g  (backward of nearest x2 upsample + crop)
// the same over a list of coarse voxels (ascending flat indices into [B,Dc,Hc,Wc]): the other rows of `out` are not touched
template <typename T>
__global__ void downsample_sum_rows_kernel(const T* __restrict__ g, T* __restrict__ out, const int* __restrict__ rows, int nrows,
                                           int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int C)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)nrows * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t ri = (uint32_t)(i / (uint32_t)CG);       // total < 2^32 granules
        const int cg = (int)((uint32_t)i - ri * (uint32_t)CG);
        const size_t orow = (size_t)rows[ri];
        int cg0, x, y, z, b;
        decode_gxyzb(orow, 1, Wc, Hc, Dc, cg0, x, y, z, b);
        float acc[G];
#pragma unroll
        for (int k = 0; k < G; ++k) acc[k] = 0.f;
        for (int a = 0; a < 2; ++a) { const int zz = 2 * z + a; if (zz >= Df) continue;
            for (int c = 0; c < 2; ++c) { const int yy = 2 * y + c; if (yy >= Hf) continue;
                for (int d = 0; d < 2; ++d) { const int xx = 2 * x + d; if (xx >= Wf) continue;
                    float v[G];
                    Gran<T>::ld(g + ((((size_t)b * Df + zz) * Hf + yy) * Wf + xx) * C + (size_t)cg * G, v);
#pragma unroll
                    for (int k = 0; k < G; ++k) acc[k] += v[k];
                }
            }
        }
        Gran<T>::st(out + orow * C + (size_t)cg * G, acc);
    }
}
template <typename T>
__global__ void downsample_sum_kernel(const T* __restrict__ g, T* __restrict__ out, int B, int Df, int Hf, int Wf,
                                      int Dc, int Hc, int Wc, int C)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)B * Dc * Hc * Wc * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int cg, x, y, z, b;
        decode_gxyzb(i, CG, Wc, Hc, Dc, cg, x, y, z, b);
        float acc[G];
#pragma unroll
        for (int k = 0; k < G; ++k) acc[k] = 0.f;
        for (int a = 0; a < 2; ++a) { const int zz = 2 * z + a; if (zz >= Df) continue;
            for (int c = 0; c < 2; ++c) { const int yy = 2 * y + c; if (yy >= Hf) continue;
                for (int d = 0; d < 2; ++d) { const int xx = 2 * x + d; if (xx >= Wf) continue;
                    float v[G];
                    Gran<T>::ld(g + ((((size_t)b * Df + zz) * Hf + yy) * Wf + xx) * C + (size_t)cg * G, v);
#pragma unroll
                    for (int k = 0; k < G; ++k) acc[k] += v[k];
                } } }
        Gran<T>::st(out + i * G, acc);
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] (+)= sum_m g[m][c]   two-stage: partial[chunk][c] then a final pass
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ g, float* __restrict__ partial, size_t M, int C, int rows_per_chunk)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const int cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.y * cgs + (t % cgs), r0 = t / cgs;
    const size_t v0 = (size_t)blockIdx.x * rows_per_chunk, v1 = min(v0 + (size_t)rows_per_chunk, M);
    float s[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s[i] = 0.f;
    if (r0 < rpi) {
        constexpr int UN = 4;                      // rows in flight per thread; added in row order (bit-identical to one at a time)
        for (size_t v = v0 + r0; v < v1; v += (size_t)rpi * UN) {
            uint4 q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + (size_t)u * rpi < v1) q[u] = *reinterpret_cast<const uint4*>(g + (v + (size_t)u * rpi) * C + (size_t)cg * G);
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + (size_t)u * rpi < v1) {
                    float x[G];
                    Gran<T>::unpack(q[u], x);
#pragma unroll
                    for (int i = 0; i < G; ++i) s[i] += x[i];
                }
        }
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int i = 0; i < G; ++i) red[t][i] = s[i];
    __syncthreads();
    if (t < cgs) {
#pragma unroll
        for (int i = 0; i < G; ++i) { float a = 0.f; for (int r = 0; r < rpi; ++r) a += red[r * cgs + t][i]; partial[(size_t)blockIdx.x * C + (size_t)cg * G + i] = a; }
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nchunks, int C, int accumulate)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
#pragma unroll 4
    for (int k = lane; k < nchunks; k += 64) s += partial[(size_t)k * C + c];
    s = wave_sum_d(s);
    if (lane == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// The bias gradients of ALL linear layers of a backward pass in two launches (the point-set executor keeps every layer's output
// gradient until its pass ends; per layer they were a partial + a final launch, ~90 launches per step).  bf16 gradients with C <= 2048
// (one slab); a record's chunks are dreg_colsum's (colsum_rows_per_chunk), every partial and final sum is formed in the same order.
struct ColsumDesc { const bf16_t* g; float* out; float* partial; int M, C, rpc, nch, pblock0, fblock0, accumulate, pad; };
static_assert(sizeof(ColsumDesc) == 56, "column-sum record: 56 bytes");
__device__ __forceinline__ int colsum_find(const ColsumDesc* __restrict__ descs, int n, int blk, bool final_pass)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((final_pass ? descs[mid].fblock0 : descs[mid].pblock0) <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void colsum_partial_batched_kernel(const ColsumDesc* __restrict__ descs, int n)
{
    constexpr int G = 8;
    const ColsumDesc d = descs[colsum_find(descs, n, blockIdx.x, false)];
    const int chunk = blockIdx.x - d.pblock0;
    const int C = d.C, CG = C / G;
    const int cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = t % cgs, r0 = t / cgs;
    const size_t M = (size_t)d.M;
    const size_t v0 = (size_t)chunk * d.rpc, v1 = min(v0 + (size_t)d.rpc, M);
    float s[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s[i] = 0.f;
    if (r0 < rpi) {
        constexpr int UN = 4;
        for (size_t v = v0 + r0; v < v1; v += (size_t)rpi * UN) {
            uint4 q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + (size_t)u * rpi < v1) q[u] = *reinterpret_cast<const uint4*>(d.g + (v + (size_t)u * rpi) * C + (size_t)cg * G);
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + (size_t)u * rpi < v1) {
                    float x[G];
                    Gran<bf16_t>::unpack(q[u], x);
#pragma unroll
                    for (int i = 0; i < G; ++i) s[i] += x[i];
                }
        }
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int i = 0; i < G; ++i) red[t][i] = s[i];
    __syncthreads();
    if (t < cgs) {
#pragma unroll
        for (int i = 0; i < G; ++i) { float a = 0.f; for (int r = 0; r < rpi; ++r) a += red[r * cgs + t][i]; d.partial[(size_t)chunk * C + (size_t)cg * G + i] = a; }
    }
}
__global__ __launch_bounds__(256) void colsum_final_batched_kernel(const ColsumDesc* __restrict__ descs, int n)
{
    const ColsumDesc d = descs[colsum_find(descs, n, blockIdx.x, true)];
    const int c = (blockIdx.x - d.fblock0) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= d.C) return;
    double s = 0.0;
#pragma unroll 4
    for (int k = lane; k < d.nch; k += 64) s += d.partial[(size_t)k * d.C + c];
    s = wave_sum_d(s);
    if (lane == 0) d.out[c] = d.accumulate ? d.out[c] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------------ trilinear gather
// feats[n][:] = trilinear sample (align_corners=True) of p1[b] at fine voxel index idx[n] of a (Zr,Xr,Yr) grid whose
// flat index is (x*Yr + y)*Zr + z  (nerf_regtr.py:144-147).  p1 dims (d,h,w) correspond to (z,x,y).
struct TriAxis { int i0, i1; float t; };
__device__ __forceinline__ TriAxis tri_axis(int i, int n_out, int n_in) {
    const float s = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    const float f = (float)i * s;
    TriAxis a;
    a.i0 = min((int)f, n_in - 1);
    a.i1 = min(a.i0 + 1, n_in - 1);
    a.t = f - (float)a.i0;
    return a;
}
template <typename T, typename TO>
__global__ void trilinear_gather_fwd_kernel(const T* __restrict__ p1, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                            TO* __restrict__ out, int N, int d, int h, int w, int C, int Zr, int Xr, int Yr)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const size_t total = (size_t)N * CG;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG), n = (int)(i / CG);
        const int64_t f = idx[n];
        const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
        const int b = pt_batch[n];
        const TriAxis az = tri_axis(z, Zr, d), ax = tri_axis(x, Xr, h), ay = tri_axis(y, Yr, w);
        float acc[G];
#pragma unroll
        for (int k = 0; k < G; ++k) acc[k] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int zi = (c & 4) ? az.i1 : az.i0, xi = (c & 2) ? ax.i1 : ax.i0, yi = (c & 1) ? ay.i1 : ay.i0;
            const float wgt = ((c & 4) ? az.t : 1.f - az.t) * ((c & 2) ? ax.t : 1.f - ax.t) * ((c & 1) ? ay.t : 1.f - ay.t);
            float v[G];
            Gran<T>::ld(p1 + ((((size_t)b * d + zi) * h + xi) * w + yi) * C + (size_t)cg * G, v);
#pragma unroll
            for (int k = 0; k < G; ++k) acc[k] += v[k] * wgt;
        }
        if constexpr (sizeof(TO) == 4) {
#pragma unroll
            for (int k = 0; k < G; ++k) out[(size_t)n * C + (size_t)cg * G + k] = acc[k];
        } else {
            Gran<T>::st((T*)out + (size_t)n * C + (size_t)cg * G, acc);
        }
    }
}
// The gather fused with the first voxel-average round that consumes it (grid_downsample.py:6-94 on the gathered features): one wave per
// output row sums its members' trilinear samples — every sample formed exactly as trilinear_gather_fwd_kernel forms it (corner order,
// fp32 products and sums), the members added in ascending order and scaled as segment_mean_kernel does: bit-identical to the two
// launches, without the [N, C] fp32 feature tensor between them (150 MB written and read back per step at 4 pairs).
// order / starts / nseg: the round's plan (member indices are local to the pair: + point_start).
template <typename T>
__global__ __launch_bounds__(256) void gather_segment_mean_kernel(const T* __restrict__ p1, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                                                  int point_start, const uint32_t* __restrict__ order, const uint32_t* __restrict__ starts,
                                                                  const int* __restrict__ nseg, float* __restrict__ out,
                                                                  int d, int h, int w, int C, int Zr, int Xr, int Yr)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= *nseg) return;
    const uint32_t s0 = starts[seg], s1 = starts[seg + 1];
    const float inv = 1.f / (float)(s1 - s0);
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t j = s0; j < s1; ++j) {
            const int n = point_start + (int)order[j];
            const int64_t f = idx[n];
            const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
            const int b = pt_batch[n];
            const TriAxis az = tri_axis(z, Zr, d), ax = tri_axis(x, Xr, h), ay = tri_axis(y, Yr, w);
            float v[8][4], wg[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int zi = (c & 4) ? az.i1 : az.i0, xi = (c & 2) ? ax.i1 : ax.i0, yi = (c & 1) ? ay.i1 : ay.i0;
                wg[c] = ((c & 4) ? az.t : 1.f - az.t) * ((c & 2) ? ax.t : 1.f - ax.t) * ((c & 1) ? ay.t : 1.f - ay.t);
                const T* src = p1 + ((((size_t)b * d + zi) * h + xi) * w + yi) * C + c0;
                if constexpr (sizeof(T) == 2) {
                    const uint2 q = *reinterpret_cast<const uint2*>(src);
                    v[c][0] = __uint_as_float(q.x << 16); v[c][1] = __uint_as_float(q.x & 0xffff0000u);
                    v[c][2] = __uint_as_float(q.y << 16); v[c][3] = __uint_as_float(q.y & 0xffff0000u);
                } else {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
                }
            }
            float smp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) smp[k] += v[c][k] * wg[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += smp[k];
        }
        *reinterpret_cast<float4*>(out + (size_t)seg * C + c0) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    }
}
// backward: scatter-add of dfeats into an fp32 grid gradient (atomics; the gradient is sparse around the mask)
template <typename TG>
__global__ void trilinear_gather_bwd_kernel(const TG* __restrict__ dfeat, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                            float* __restrict__ dp1, int N, int d, int h, int w, int C, int Zr, int Xr, int Yr)
{
    const size_t total = (size_t)N * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), n = (int)(i / C);
        const int64_t f = idx[n];
        const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
        const int b = pt_batch[n];
        const TriAxis az = tri_axis(z, Zr, d), ax = tri_axis(x, Xr, h), ay = tri_axis(y, Yr, w);
        const float g = Elem<TG>::ld(dfeat + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zi = (k & 4) ? az.i1 : az.i0, xi = (k & 2) ? ax.i1 : ax.i0, yi = (k & 1) ? ay.i1 : ay.i0;
            const float wgt = ((k & 4) ? az.t : 1.f - az.t) * ((k & 2) ? ax.t : 1.f - ax.t) * ((k & 1) ? ay.t : 1.f - ay.t);
            if (wgt != 0.f) atomicAdd(dp1 + ((((size_t)b * d + zi) * h + xi) * w + yi) * C + c, g * wgt);
        }
    }
}

// active-set form of the gather backward: contributions are accumulated in a compact fp32 buffer [n1, C] addressed through
// map1 (rank of a coarse voxel in S1), then written once as T rows of the (zero-filled) dense gradient.
__global__ void trilinear_gather_bwd_rows_kernel(const float* __restrict__ dfeat, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                                 const int* __restrict__ map1, float* __restrict__ comp, int N, int d, int h, int w, int C,
                                                 int Zr, int Xr, int Yr)
{
    const size_t total = (size_t)N * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), n = (int)(i / C);
        const int64_t f = idx[n];
        const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
        const int b = pt_batch[n];
        const TriAxis az = tri_axis(z, Zr, d), ax = tri_axis(x, Xr, h), ay = tri_axis(y, Yr, w);
        const float g = dfeat[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zi = (k & 4) ? az.i1 : az.i0, xi = (k & 2) ? ax.i1 : ax.i0, yi = (k & 1) ? ay.i1 : ay.i0;
            const float wgt = ((k & 4) ? az.t : 1.f - az.t) * ((k & 2) ? ax.t : 1.f - ax.t) * ((k & 1) ? ay.t : 1.f - ay.t);
            if (wgt != 0.f) atomicAdd(comp + (size_t)map1[(((size_t)b * d + zi) * h + xi) * w + yi] * C + c, g * wgt);
        }
    }
}
// Deterministic (atomic-free) form: one wave per S1 voxel gathers the contributions of the occupied fine voxels whose
// trilinear corner set contains it.  fine_map[b][idx] = point index or -1.  Candidates per axis: the <= 5 fine coordinates with
// i*s in (c-1, c+1) (tri_range), each tested with tri_axis() itself, so the weights are the forward's bit for bit.
__global__ void fine_map_fill_kernel(const int64_t* __restrict__ idx, const int* __restrict__ pt_batch, int* __restrict__ fine_map, int N, size_t Vf)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) fine_map[(size_t)pt_batch[n] * Vf + (size_t)idx[n]] = n;
}
__device__ __forceinline__ void tri_range(int c, int n_out, int n_in, int& lo, int& hi) {
    if (n_out <= 1 || n_in <= 1) { lo = 0; hi = n_out - 1; return; }
    // fine coordinates i whose source coordinate i * s (tri_axis) lies strictly inside (c - 1, c + 1): i in ((c-1)/s, (c+1)/s), widened by
    // 0.01 against the rounding of the two products (coordinates < 2^16: errors ~1e-3 at most) — at most 5 per axis at a 2:1 ratio, so the
    // 125 candidates of a coarse voxel take two rounds of the 64 lanes (the former floor - 1 / ceil + 1 bracket: 7 per axis, six rounds)
    const float inv = (float)(n_out - 1) / (float)(n_in - 1);
    lo = max(0, (int)floorf((float)(c - 1) * inv - 0.01f) + 1);
    hi = min(n_out - 1, (int)ceilf((float)(c + 1) * inv + 0.01f) - 1);
}
__device__ __forceinline__ float tri_weight(int i, int c, int n_out, int n_in) {
    const TriAxis a = tri_axis(i, n_out, n_in);
    return (a.i0 == c ? 1.f - a.t : 0.f) + (a.i1 == c ? a.t : 0.f);
}
// SEG: the gradient of the gathered features is not materialised — the features went straight into the first voxel-average round of
// their pair (grid_downsample.py:6-94), so d(feats)[n] = g1[inv_seg[n]] * inv_cnt[n] with g1 the gradient of that round's output (one
// sixth of the rows: L2-resident): dfeat = g1 (the pairs' blocks one after the other), seg[grid] = (inv_seg, inv_cnt, first point, first
// g1 row) of the grid's pair.  The product is rounded to fp32 before it is weighted, exactly as the stored d(feats) was: bit-identical.
struct TriSegDesc { const uint32_t* inv_seg; const float* inv_cnt; long long point_start, row_off; };
static_assert(sizeof(TriSegDesc) == 32, "int64 [B][4] on the Python side");
template <typename T, int CPL, bool SEG = false>
__global__ __launch_bounds__(256) void trilinear_gather_bwd_gather_kernel(const float* __restrict__ dfeat, const int* __restrict__ fine_map,
                                                                          const int* __restrict__ rows1, int n1, T* __restrict__ dp1,
                                                                          int d, int h, int w, int Zr, int Xr, int Yr,
                                                                          const TriSegDesc* __restrict__ seg = nullptr)
{
    constexpr int C = CPL * 64;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= n1) return;
    const int v = rows1[wv];
    const int yc = v % w, xc = (v / w) % h, zc = (v / (w * h)) % d, b = v / (w * h * d);
    int zlo, zhi, xlo, xhi, ylo, yhi;
    tri_range(zc, Zr, d, zlo, zhi); tri_range(xc, Xr, h, xlo, xhi); tri_range(yc, Yr, w, ylo, yhi);
    const int nz = zhi - zlo + 1, nx = xhi - xlo + 1, ny = yhi - ylo + 1, total = nz * nx * ny;
    const size_t Vf = (size_t)Zr * Xr * Yr;
    float acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
    for (int base = 0; base < total; base += 64) {
        const int j = base + lane;
        int n = -1;
        float wgt = 0.f;
        if (j < total) {
            const int jy = j % ny, jx = (j / ny) % nx, jz = j / (ny * nx);
            const int z = zlo + jz, x = xlo + jx, y = ylo + jy;
            wgt = tri_weight(z, zc, Zr, d) * tri_weight(x, xc, Xr, h) * tri_weight(y, yc, Yr, w);
            if (wgt != 0.f) n = fine_map[(size_t)b * Vf + ((size_t)x * Yr + y) * Zr + z];
        }
        float scl = 1.f;
        if constexpr (SEG) {
            if (n >= 0) {
                const TriSegDesc sd = seg[b];
                const long long loc = (long long)n - sd.point_start;
                scl = sd.inv_cnt[loc];
                n = (int)((long long)sd.inv_seg[loc] + sd.row_off);
            }
        }
        unsigned long long m = __ballot(n >= 0);
        // four contributing points per round: their rows are requested together and added in the same (ascending) order — one row in
        // flight per wave made the kernel a chain of ~20 memory round trips per coarse voxel
        while (m) {
            constexpr int U = 4;
            int nn[U];
            float ww[U], ss[U], r[U][CPL];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = m != 0;                                   // wave-uniform
                const int l = ok[u] ? __ffsll((long long)m) - 1 : 0;
                if (ok[u]) m &= m - 1;
                nn[u] = __shfl(n, l, 64);
                ww[u] = __shfl(wgt, l, 64);
                if constexpr (SEG) ss[u] = __shfl(scl, l, 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
#pragma unroll
                    for (int k = 0; k < CPL; ++k) r[u][k] = dfeat[(size_t)nn[u] * C + lane + 64 * k];
                }
            if constexpr (SEG) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ok[u]) {
#pragma unroll
                        for (int k = 0; k < CPL; ++k) r[u][k] = r[u][k] * ss[u];
                    }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
#pragma unroll
                    for (int k = 0; k < CPL; ++k) acc[k] += r[u][k] * ww[u];
                }
        }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) Elem<T>::st(dp1 + (size_t)v * C + lane + 64 * k, acc[k]);
}
template <typename T>
__global__ void scatter_rows_cast_kernel(const float* __restrict__ comp, const int* __restrict__ rows, T* __restrict__ out, size_t total_gran, int C)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total_gran; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = (i >> 32) == 0 ? (size_t)((uint32_t)i / (uint32_t)CG) : i / CG;
        const int cg = (int)(i - r * CG);
        float v[G];
#pragma unroll
        for (int k = 0; k < G; ++k) v[k] = comp[r * C + (size_t)cg * G + k];
        Gran<T>::st(out + (size_t)rows[r] * C + (size_t)cg * G, v);
    }
}
// out[rows[r]][:] = 0: returns a dense buffer that only ever holds values on a row list to the all-zero state (the gradient
// buffers in front of the active-set convolutions: 1 GB each at 8 x 64^3 x 256 — zeroing the ~5 % of rows that were written instead
// of the whole buffer every step)
template <typename T>
__global__ void zero_rows_kernel(const int* __restrict__ rows, T* __restrict__ out, size_t total_gran, int C)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total_gran; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = (i >> 32) == 0 ? (size_t)((uint32_t)i / (uint32_t)CG) : i / CG;
        const int cg = (int)(i - r * CG);
        *reinterpret_cast<uint4*>(out + (size_t)rows[r] * C + (size_t)cg * G) = make_uint4(0, 0, 0, 0);
    }
}
// column sums over a row list: partial[chunk][c] = sum over rows[chunk*rpc .. ) of g[rows[i]][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_partial_kernel(const T* __restrict__ g, const int* __restrict__ rows, float* __restrict__ partial,
                                                                  int nrows, int C, int rows_per_chunk)
{
    constexpr int G = Gran<T>::G;
    const int CG = C / G;
    const int cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.y * cgs + (t % cgs), r0 = t / cgs;
    const int v0 = blockIdx.x * rows_per_chunk, v1 = min(v0 + rows_per_chunk, nrows);
    float s[G];
#pragma unroll
    for (int i = 0; i < G; ++i) s[i] = 0.f;
    if (r0 < rpi) {
        constexpr int UN = 4;                      // rows in flight per thread; added in list order
        for (int v = v0 + r0; v < v1; v += rpi * UN) {
            int rw[UN];
            uint4 q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) rw[u] = v + u * rpi < v1 ? rows[v + u * rpi] : -1;
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (rw[u] >= 0) q[u] = *reinterpret_cast<const uint4*>(g + (size_t)rw[u] * C + (size_t)cg * G);
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (rw[u] >= 0) {
                    float x[G];
                    Gran<T>::unpack(q[u], x);
#pragma unroll
                    for (int i = 0; i < G; ++i) s[i] += x[i];
                }
        }
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int i = 0; i < G; ++i) red[t][i] = s[i];
    __syncthreads();
    if (t < cgs) {
#pragma unroll
        for (int i = 0; i < G; ++i) { float a = 0.f; for (int r = 0; r < rpi; ++r) a += red[r * cgs + t][i]; partial[(size_t)blockIdx.x * C + (size_t)cg * G + i] = a; }
    }
}

// dst += src (gradient accumulation where a tensor feeds two consumers); fp32 add, one rounding — what torch's add does
template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, size_t total_gran)
{
    constexpr int G = Gran<T>::G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total_gran; i += (size_t)gridDim.x * blockDim.x) {
        float a[G], b[G];
        Gran<T>::ld(dst + i * G, a);
        Gran<T>::ld(src + i * G, b);
#pragma unroll
        for (int k = 0; k < G; ++k) a[k] += b[k];
        Gran<T>::st(dst + i * G, a);
    }
}
// ------------------------------------------------------------------------------------------------ input staging
// grids: B device pointers to fp32 [7, Z, X, Y] voxel grids (xyz | rgb | alpha, the reference's voxel_grid.pt layout).
// out [B, Z, X, Y, 8] (T): channels 0..3 = rgba (grid channels 3..6), 4..7 = 0  — the NDHWC input of the stem convolution
// (nerf_regtr.py:131-135 feeds grid[:, 3:]).
// inocc (optional, zero-filled by the caller): byte [B, Z, X] = 1 where the Y-row (b, z, x) holds any non-zero rgba value — the
// stem convolution skips output rows whose whole receptive field is flagged empty (dreg_conv_row_occupancy).
template <typename T>
__global__ void pack_rgba_kernel(const float* const* __restrict__ grids, T* __restrict__ out, size_t V, int B, uint8_t* __restrict__ inocc, int Yr)
{
    const size_t total = V * B;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / V);
        const size_t v = i - (size_t)b * V;
        const float* g = grids[b] + 3 * V + v;
        float c[8] = {g[0], g[V], g[2 * V], g[3 * V], 0.f, 0.f, 0.f, 0.f};
        if (inocc && (c[0] != 0.f || c[1] != 0.f || c[2] != 0.f || c[3] != 0.f)) inocc[i / Yr] = 1;
        if constexpr (sizeof(T) == 2) Gran<bf16_t>::st(reinterpret_cast<bf16_t*>(out) + i * 8, c);
        else { float lo[4] = {c[0], c[1], c[2], c[3]}, hi[4] = {0.f, 0.f, 0.f, 0.f}; Gran<float>::st(reinterpret_cast<float*>(out) + i * 8, lo); Gran<float>::st(reinterpret_cast<float*>(out) + i * 8 + 4, hi); }
    }
}
// sparse form of the same staging: vals fp32 [N,7] (xyz | rgb | alpha) of the occupied voxels idx[n] (flat (x*Yr + y)*Zr + z) of
// grid pt_batch[n]; out [B,Z,X,Y,8] must be zero-filled by the caller (the reference's voxel_grid.pt is zero outside the mask,
// eval_ngp_nerf.py:397-405)
template <typename T>
__global__ void scatter_rgba_kernel(const float* __restrict__ vals, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                    T* __restrict__ out, int N, int Zr, int Xr, int Yr, uint8_t* __restrict__ inocc)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t f = idx[n];
    const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
    const size_t V = (size_t)Zr * Xr * Yr, o = ((size_t)pt_batch[n] * V + ((size_t)z * Xr + x) * Yr + y) * 8;
    if (inocc) inocc[((size_t)pt_batch[n] * Zr + z) * Xr + x] = 1;
    const float* v = vals + (size_t)n * 7 + 3;
    float c[8] = {v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) Gran<bf16_t>::st(reinterpret_cast<bf16_t*>(out) + o, c);
    else { float lo[4] = {c[0], c[1], c[2], c[3]}, hi[4] = {0.f, 0.f, 0.f, 0.f}; Gran<float>::st(reinterpret_cast<float*>(out) + o, lo); Gran<float>::st(reinterpret_cast<float*>(out) + o + 4, hi); }
}
// xyz[n, c] = grids[pt_batch[n]][c, z, x, y] with (x, y, z) decoded from idx[n] = (x*Yr + y)*Zr + z  (nerf_regtr.py:144-147:
// grid[:3].permute(X, Y, Z)[mask])
__global__ void gather_xyz_kernel(const float* const* __restrict__ grids, const int64_t* __restrict__ idx, const int* __restrict__ pt_batch,
                                  float* __restrict__ xyz, int N, int Zr, int Xr, int Yr)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t f = idx[n];
    const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
    const size_t V = (size_t)Zr * Xr * Yr, o = ((size_t)z * Xr + x) * Yr + y;
    const float* g = grids[pt_batch[n]];
    xyz[(size_t)n * 3] = g[o]; xyz[(size_t)n * 3 + 1] = g[V + o]; xyz[(size_t)n * 3 + 2] = g[2 * V + o];
}

// rowocc[b, zo, ho] = any inocc[b, zi, hi] inside the kernel window of output row (zo, ho, *):  zi = zo*stride - pad + d, d < ksz
__global__ void conv_row_occupancy_kernel(const uint8_t* __restrict__ inocc, uint8_t* __restrict__ rowocc, int B, int Di, int Hi, int Do, int Ho,
                                          int ksz, int stride, int pad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Do * Ho) return;
    const int ho = i % Ho, zo = (i / Ho) % Do, b = i / (Ho * Do);
    uint8_t any = 0;
    for (int dz = 0; dz < ksz; ++dz) {
        const int zi = zo * stride - pad + dz;
        if ((unsigned)zi >= (unsigned)Di) continue;
        for (int dh = 0; dh < ksz; ++dh) {
            const int hi = ho * stride - pad + dh;
            if ((unsigned)hi < (unsigned)Hi) any |= inocc[((size_t)b * Di + zi) * Hi + hi];
        }
    }
    rowocc[i] = any;
}

// fp32 <-> T conversions (contiguous)
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) Elem<T>::st(out + i, in[i]);
}

// ------------------------------------------------------------------------------------------------ C ABI
// ------------------------------------------------------------------------------------------------ stem over a sparse volume
// The stem's convolution (5^3, stride 2, no bias) over a sparse input volume is non-zero on a short list R of output rows (~5 % at
// the benchmark's shells: dreg_conv_rows); behind it resnet3d.py:118-123 runs bn1 -> relu -> maxpool, and the FPN's finest lateral
// (feature_pyramid_net.py:97-103) reads the activation on its own row list S3.  Everything the dense passes did with the 8 x 64^3 x 64
// tensors follows from the lists:
//  * statistics: the zero rows add nothing to (sum x, sum x^2): sums over R, mean / variance over all V rows;
//  * pooling: a window without a listed row holds the constant relu(shift) (its first in-bounds tap wins the strict comparison);
//    the others run the 27 taps.  The raw x of the arg-max voxel is kept per pooled element (xam);
//  * the activation itself is written on S3 only (nobody else reads it);
//  * backward: every pooled gradient lands on exactly one input voxel, so (sum g, sum g xhat) over all input voxels is a sum over
//    the POOLED elements (mask and xhat from xam) plus the lateral's gradient on S3; dx is needed on R only (the stem's weight
//    gradient reads nothing else, and there is no gradient for the network input).
// Row lists: ascending int32 flat indices into [B, V]; a grid's segment is found by binary search.
__device__ __forceinline__ int rows_lower_bound(const int* __restrict__ rows, int n, long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)rows[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// [s, e) = grid b's segment of an ascending row list, found by the block's first wave: the two bounds side by side (lanes 0-31 / 32-63), 32
// probes per round (four rounds of dependent loads for 2^18 rows instead of the 2 x 18 of two scalar binary searches), published through LDS.
// Every thread of the block must call it (it ends with a barrier).
__device__ __forceinline__ void rows_segment(const int* __restrict__ rows, int n, long long V, int b, int& s, int& e)
{
    __shared__ int seg_bounds[2];
    if (threadIdx.x < 64) {
        const int half = threadIdx.x >> 5, l = threadIdx.x & 31;
        const long long key = (long long)(b + half) * V;
        int lo = 0, hi = n;                                   // first index with rows[i] >= key lies in [lo, hi]
        while (hi - lo > 0) {
            const int span = hi - lo;
            const int pos = lo + (int)(((long long)span * (l + 1)) / 33);      // 32 probes strictly inside or at lo .. hi - 1
            const int p = pos < hi ? pos : hi - 1;
            const bool less = (long long)rows[p] < key;
            const unsigned long long m = __ballot(less);
            const unsigned mm = (unsigned)(half ? (m >> 32) : (m & 0xffffffffull));
            const int cnt = __popc(mm);                       // probes are ascending, the predicate is monotone: the first cnt are true
            // new bracket: after the last true probe, up to the first false probe
            const int p_last_true = cnt > 0 ? __shfl(p, (half << 5) + cnt - 1, 64) : lo - 1;
            const int p_first_false = cnt < 32 ? __shfl(p, (half << 5) + cnt, 64) : hi;
            lo = p_last_true + 1;
            hi = p_first_false;
        }
        if (l == 0) seg_bounds[half] = lo;
    }
    __syncthreads();
    s = seg_bounds[0]; e = seg_bounds[1];
}
constexpr int SSTEM_NCH = 128;      // chunks a grid's list segment is split into (statistics passes)
// partial[b][chunk0 + chunk][c][2] over the chunk's piece of grid b's list segment.  MODE 0: (sum x, sum x^2) of x at the rows;
// MODE 1: (sum g, sum g*xhat), g = dl at the row masked by the ReLU (x*scale + shift > 0).  grid (SSTEM_NCH, B, slabs).
template <int MODE>
__device__ __forceinline__ void sstem_rows_sums(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dl, const int* __restrict__ rows, int n,
                                                const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift, float* __restrict__ partial,
                                                int V, int C, int chunk, int nch, int chunk0, int nch_total, int relu, float (*red)[17])
{
    constexpr int G = 8;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    int s, e;
    rows_segment(rows, n, V, b, s, e);
    const int len = e - s;
    const int i0 = s + (int)((long long)len * chunk / nch), i1 = s + (int)((long long)len * (chunk + 1) / nch);
    float s1[G], s2[G], mu[G], rs[G], sc[G], sh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        s1[i] = s2[i] = 0.f;
        if (MODE == 1) {
            const size_t pc = ((size_t)b * C + cg * G + i) * 2;
            mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1];
        }
    }
    if (r0 < rpi && cg < CG) {
        constexpr int UN = 4;                       // rows in flight per thread (added in row order)
        for (int v = i0 + r0; v < i1; v += rpi * UN) {
            uint4 xq[UN], gq[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < i1) {
                    const size_t off = (size_t)rows[v + u * rpi] * C + (size_t)cg * G;
                    xq[u] = *reinterpret_cast<const uint4*>(x + off);
                    if (MODE == 1) gq[u] = *reinterpret_cast<const uint4*>(dl + off);
                }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < i1) {
                    float xv[G], gv[G];
                    Gran<bf16_t>::unpack(xq[u], xv);
                    if (MODE == 1) Gran<bf16_t>::unpack(gq[u], gv);
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        if (MODE == 0) { s1[k] += xv[k]; s2[k] += xv[k] * xv[k]; }
                        else {
                            const float g = (relu && !((xv[k] * sc[k] + sh[k]) > 0.f)) ? 0.f : gv[k];
                            s1[k] += g; s2[k] += g * (xv[k] - mu[k]) * rs[k];
                        }
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) { red[t][i] = s1[i]; red[t][G + i] = s2[i]; }
    __syncthreads();
    if (t < cgs && cg < CG) {
        float a1[G], a2[G];
#pragma unroll
        for (int i = 0; i < G; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
        for (int r = 0; r < rpi; ++r)
#pragma unroll
            for (int i = 0; i < G; ++i) { a1[i] += red[r * cgs + t][i]; a2[i] += red[r * cgs + t][G + i]; }
        float* dst = partial + (((size_t)b * nch_total + chunk0 + chunk) * C + (size_t)cg * G) * 2;
#pragma unroll
        for (int i = 0; i < G; ++i) { dst[2 * i] = a1[i]; dst[2 * i + 1] = a2[i]; }
    }
}
__global__ __launch_bounds__(256) void sstem_stats_kernel(const bf16_t* __restrict__ x, const int* __restrict__ rows, int n, float* __restrict__ partial, int V, int C)
{
    __shared__ float red[256][17];
    sstem_rows_sums<0>(x, nullptr, rows, n, nullptr, nullptr, partial, V, C, blockIdx.x, gridDim.x, 0, gridDim.x, 0, red);
}
// pooled windows (3^3, stride 2, pad 1) that contain a listed row
__global__ void sstem_pool_mark_kernel(const int* __restrict__ rows, int n, uint8_t* __restrict__ pmask, int Di, int Hi, int Wi, int Do, int Ho, int Wo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = rows[i];
    const int ix = r % Wi; r /= Wi;
    const int iy = r % Hi; r /= Hi;
    const int iz = r % Di; const int b = r / Di;
    for (int oz = iz >> 1; oz <= ((iz + 1) >> 1) && oz < Do; ++oz)
        for (int oy = iy >> 1; oy <= ((iy + 1) >> 1) && oy < Ho; ++oy)
            for (int ox = ix >> 1; ox <= ((ix + 1) >> 1) && ox < Wo; ++ox) pmask[(((size_t)b * Do + oz) * Ho + oy) * Wo + ox] = 1;
}
// pooled = maxpool3(relu(bn(x))) as bn_relu_maxpool_fwd_kernel (same rounding, tap order, strict comparison), xam = raw x of the arg-max voxel
__global__ void sstem_pool_fwd_kernel(const bf16_t* __restrict__ x, const uint8_t* __restrict__ pmask, const float* __restrict__ scale_shift,
                                      bf16_t* __restrict__ y, uint8_t* __restrict__ arg, bf16_t* __restrict__ xam,
                                      int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu)
{
    constexpr int G = 8;
    const int CG = C / G;
    const size_t total = (size_t)B * Do * Ho * Wo * CG;        // < 2^32 (checked by the host): 32-bit index arithmetic
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r = (uint32_t)i;
        const int cg = (int)(r % (uint32_t)CG); r /= (uint32_t)CG;
        const uint32_t pv = r;                                   // pooled voxel
        const int ox = (int)(r % (uint32_t)Wo); r /= (uint32_t)Wo;
        const int oy = (int)(r % (uint32_t)Ho); r /= (uint32_t)Ho;
        const int oz = (int)(r % (uint32_t)Do); const int b = (int)(r / (uint32_t)Do);
        float sc[G], sh[G], best[G], xb[G];
        int bi[G];
        {
            const float4* ss = reinterpret_cast<const float4*>(scale_shift + ((size_t)b * C + (size_t)cg * G) * 2);    // (scale, shift) x 8: 64 bytes
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = ss[q]; sc[2 * q] = v.x; sh[2 * q] = v.y; sc[2 * q + 1] = v.z; sh[2 * q + 1] = v.w; }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) { best[k] = -INFINITY; bi[k] = 0; xb[k] = 0.f; }
        if (!pmask[pv]) {
            // every row of the window is zero: relu(0 * scale + shift) everywhere, the first in-bounds tap wins
            const int tap = ((oz == 0 ? 1 : 0) * 3 + (oy == 0 ? 1 : 0)) * 3 + (ox == 0 ? 1 : 0);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                float o = 0.f * sc[k] + sh[k];
                o = relu ? fmaxf(o, 0.f) : o;
                best[k] = bf2f(f2bf(o)); bi[k] = tap;
            }
        } else {
            // one z-plane of the window at a time: its (up to) nine granules are requested together, then compared in tap order.  Most
            // granules of a marked window are still all-zero (rows outside the list): their value is the constant relu(0 * scale + shift)
            float c0[G];
#pragma unroll
            for (int k = 0; k < G; ++k) { float o = 0.f * sc[k] + sh[k]; o = relu ? fmaxf(o, 0.f) : o; c0[k] = bf2f(f2bf(o)); }
            for (int dz = 0; dz < 3; ++dz) {
                const int z = oz * 2 - 1 + dz; if ((unsigned)z >= (unsigned)Di) continue;
                uint4 q[9];
                bool ok[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = oy * 2 - 1 + t / 3, xx = ox * 2 - 1 + t % 3;
                    ok[t] = (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
                    if (ok[t]) q[t] = *reinterpret_cast<const uint4*>(x + ((((size_t)b * Di + z) * Hi + yy) * Wi + xx) * C + (size_t)cg * G);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (!ok[t]) continue;
                    const int tap = dz * 9 + t;
                    if ((q[t].x | q[t].y | q[t].z | q[t].w) == 0u) {          // +0.0 in all eight channels
#pragma unroll
                        for (int k = 0; k < G; ++k)
                            if (c0[k] > best[k]) { best[k] = c0[k]; bi[k] = tap; xb[k] = 0.f; }
                        continue;
                    }
                    float v[G];
                    Gran<bf16_t>::unpack(q[t], v);
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        float o = v[k] * sc[k] + sh[k];
                        o = relu ? fmaxf(o, 0.f) : o;
                        o = bf2f(f2bf(o));
                        if (o > best[k]) { best[k] = o; bi[k] = tap; xb[k] = v[k]; }
                    }
                }
            }
        }
        Gran<bf16_t>::st(y + i * G, best);
        Gran<bf16_t>::st(xam + i * G, xb);
        *reinterpret_cast<uint2*>(arg + i * G) = make_uint2((uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24),
                                                            (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24));
    }
}
// a = relu(x * scale + shift) on a row list: grid (chunks, B, slabs) over the pieces of grid b's list segment, a thread keeps one channel
// granule's parameters in registers and walks down the piece's rows
constexpr int SSTEM_NCH_ROWS = 128;
__global__ __launch_bounds__(256) void sstem_apply_rows_kernel(const bf16_t* __restrict__ x, const int* __restrict__ rows, int n, const float* __restrict__ scale_shift,
                                                               bf16_t* __restrict__ a, int V, int C, int relu)
{
    constexpr int G = 8;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    int s, e;
    rows_segment(rows, n, V, b, s, e);
    if (r0 >= rpi || cg >= CG) return;
    const int len = e - s, nch = gridDim.x, chunk = blockIdx.x;
    const int i0 = s + (int)((long long)len * chunk / nch), i1 = s + (int)((long long)len * (chunk + 1) / nch);
    float sc[G], sh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { sc[i] = scale_shift[((size_t)b * C + cg * G + i) * 2]; sh[i] = scale_shift[((size_t)b * C + cg * G + i) * 2 + 1]; }
    constexpr int UN = 4;
    for (int v = i0 + r0; v < i1; v += rpi * UN) {
        uint4 xq[UN];
        size_t off[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < i1) { off[u] = (size_t)rows[v + u * rpi] * C + (size_t)cg * G; xq[u] = *reinterpret_cast<const uint4*>(x + off[u]); }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (v + u * rpi < i1) {
                float xv[G];
                Gran<bf16_t>::unpack(xq[u], xv);
#pragma unroll
                for (int k = 0; k < G; ++k) { const float o = xv[k] * sc[k] + sh[k]; xv[k] = relu ? fmaxf(o, 0.f) : o; }
                Gran<bf16_t>::st(a + off[u], xv);
            }
    }
}
// backward sums: chunks [0, nchp) walk the pooled elements of grid b (g = dp masked through xam, xhat from xam), chunks [nchp, nchp + SSTEM_NCH)
// the lateral's gradient on its row list (when there is one).  grid (nchp + SSTEM_NCH or nchp, B, slabs).
__global__ __launch_bounds__(256) void sstem_bwd_sums_kernel(const bf16_t* __restrict__ xam, const bf16_t* __restrict__ dp, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ dl, const int* __restrict__ rows_a, int n_a,
                                                             const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift, float* __restrict__ partial,
                                                             int V, int Vo, int C, int rpc, int nchp, int relu)
{
    constexpr int G = 8;
    __shared__ float red[256][17];
    const int chunk = blockIdx.x, nch_total = gridDim.x;
    if (chunk >= nchp) {
        sstem_rows_sums<1>(x, dl, rows_a, n_a, mean_rstd, scale_shift, partial, V, C, chunk - nchp, SSTEM_NCH, nchp, nch_total, relu, red);
        return;
    }
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    const int v0 = chunk * rpc, v1 = min(v0 + rpc, Vo);
    float s1[G], s2[G], mu[G], rs[G], sc[G], sh[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t pc = ((size_t)b * C + cg * G + i) * 2;
        mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1];
        s1[i] = s2[i] = 0.f;
    }
    if (r0 < rpi && cg < CG) {
        constexpr int UN = 4;
        for (int v = v0 + r0; v < v1; v += rpi * UN) {
            float xv[UN][G], gv[UN][G];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < v1) {
                    const size_t off = ((size_t)b * Vo + v + u * rpi) * C + (size_t)cg * G;
                    Gran<bf16_t>::ld(xam + off, xv[u]);
                    Gran<bf16_t>::ld(dp + off, gv[u]);
                }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (v + u * rpi < v1) {
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        const float g = (relu && !((xv[u][k] * sc[k] + sh[k]) > 0.f)) ? 0.f : gv[u][k];
                        s1[k] += g; s2[k] += g * (xv[u][k] - mu[k]) * rs[k];
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) { red[t][i] = s1[i]; red[t][G + i] = s2[i]; }
    __syncthreads();
    if (t < cgs && cg < CG) {
        float a1[G], a2[G];
#pragma unroll
        for (int i = 0; i < G; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
        for (int r = 0; r < rpi; ++r)
#pragma unroll
            for (int i = 0; i < G; ++i) { a1[i] += red[r * cgs + t][i]; a2[i] += red[r * cgs + t][G + i]; }
        float* dst = partial + (((size_t)b * nch_total + chunk) * C + (size_t)cg * G) * 2;
#pragma unroll
        for (int i = 0; i < G; ++i) { dst[2 * i] = a1[i]; dst[2 * i + 1] = a2[i]; }
    }
}
// dx = scale * (g - c1 - xhat * c2) on the convolution's row list; g = relu mask * (un-pooled dp [+ dl]).  grid (chunks, B, slabs) over the pieces
// of grid b's list segment, a thread keeps one channel granule's parameters in registers; the (at most eight) pooled windows that contain a
// voxel are enumerated statically, in the z, y, x order of unpool_gather, with all their loads in flight together
__global__ __launch_bounds__(256) void sstem_bwd_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dp, const uint8_t* __restrict__ arg, const bf16_t* __restrict__ dl,
                                                             const int* __restrict__ rows, int n, const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift,
                                                             const float* __restrict__ coef, bf16_t* __restrict__ dx, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu)
{
    constexpr int G = 8;
    const int V = Di * Hi * Wi;
    const int CG = C / G, cgs = CG < 256 ? CG : 256, rpi = 256 / cgs;
    const int t = threadIdx.x, cg = blockIdx.z * cgs + (t % cgs), r0 = t / cgs, b = blockIdx.y;
    int s, e;
    rows_segment(rows, n, V, b, s, e);
    if (r0 >= rpi || cg >= CG) return;
    const int len = e - s, nch = gridDim.x, chunk = blockIdx.x;
    const int i0 = s + (int)((long long)len * chunk / nch), i1 = s + (int)((long long)len * (chunk + 1) / nch);
    float mu[G], rs[G], sc[G], sh[G], c1[G], c2[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t pc = ((size_t)b * C + cg * G + i) * 2;
        mu[i] = mean_rstd[pc]; rs[i] = mean_rstd[pc + 1]; sc[i] = scale_shift[pc]; sh[i] = scale_shift[pc + 1]; c1[i] = coef[pc]; c2[i] = coef[pc + 1];
    }
    for (int v = i0 + r0; v < i1; v += rpi) {
        const size_t row = (size_t)rows[v];
        int r = (int)(row - (size_t)b * V);
        const int ix = r % Wi; r /= Wi;
        const int iy = r % Hi; const int iz = r / Hi;
        const size_t off = row * C + (size_t)cg * G;
        const uint4 xq = *reinterpret_cast<const uint4*>(x + off);
        uint4 lq = make_uint4(0u, 0u, 0u, 0u);
        if (dl) lq = *reinterpret_cast<const uint4*>(dl + off);
        // windows containing voxel i: per axis the one starting at floor(i/2) and, for odd i, the next one (ascending: unpool_gather's order)
        const int oz0 = iz >> 1, oy0 = iy >> 1, ox0 = ix >> 1;
        uint4 gq[8];
        uint2 aq[8];
        bool ok[8];
        int tp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int uz = u >> 2, uy = (u >> 1) & 1, ux = u & 1;
            const int oz = oz0 + uz, oy = oy0 + uy, ox = ox0 + ux;
            ok[u] = (!uz || (iz & 1)) && (!uy || (iy & 1)) && (!ux || (ix & 1)) && oz < Do && oy < Ho && ox < Wo;
            tp[u] = ((iz - 2 * oz + 1) * 3 + (iy - 2 * oy + 1)) * 3 + (ix - 2 * ox + 1);
            if (ok[u]) {
                const size_t o = ((((size_t)b * Do + oz) * Ho + oy) * Wo + ox) * C + (size_t)cg * G;
                gq[u] = *reinterpret_cast<const uint4*>(dp + o);
                aq[u] = *reinterpret_cast<const uint2*>(arg + o);
            }
        }
        float xv[G], g[G], lv[G];
        Gran<bf16_t>::unpack(xq, xv);
#pragma unroll
        for (int k = 0; k < G; ++k) g[k] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ok[u]) {
                float gg[G];
                Gran<bf16_t>::unpack(gq[u], gg);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const uint32_t a = ((k < 4 ? aq[u].x : aq[u].y) >> (8 * (k & 3))) & 0xffu;
                    if ((int)a == tp[u]) g[k] += gg[k];
                }
            }
        if (dl) {
            Gran<bf16_t>::unpack(lq, lv);
#pragma unroll
            for (int k = 0; k < G; ++k) g[k] += lv[k];
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (relu && !((xv[k] * sc[k] + sh[k]) > 0.f)) g[k] = 0.f;
            const float xh = (xv[k] - mu[k]) * rs[k];
            xv[k] = sc[k] * (g[k] - c1[k] - xh * c2[k]);
        }
        Gran<bf16_t>::st(dx + off, xv);
    }
}

static inline int nblocks(size_t total, int per = 256, int cap = 8192) {
    size_t b = (total + per - 1) / per;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}
// rows of one statistics chunk: enough chunks that B grids x chunks fill the chip even for the 8^3 / 4^3 levels of the ResNet
static inline int bn_rows_per_chunk(int V) { return V >= 262144 ? 512 : V >= 32768 ? 256 : V >= 4096 ? 128 : V >= 512 ? 32 : (V >= 8 ? 8 : V); }

extern "C" {

#ifdef DREG_PROBE
void dreg_bn_set_debug_skip(int mask) { g_bn_debug_skip = mask; }
void dreg_bn_set_store_g(int enable) { g_bn_store_g = enable ? 1 : 0; }
void dreg_bn_set_small_regs(int enable) { g_bn_small_regs = enable ? 1 : 0; }
void dreg_bn_set_small_max_voxels(int v) { g_bn_small_maxv = v; }
#endif
int dreg_bn_num_chunks(int V) { const int r = bn_rows_per_chunk(V); return (V + r - 1) / r; }

// Forward BatchNorm3d over B independent grids (per-grid statistics).  x,y,res: [B,V,C] (dtype 0 bf16 / 1 fp32).
// workspace: fp32 [B * chunks * C * 2].  scale_shift, mean_rstd: fp32 [B,C,2] outputs (saved for backward).
static int bn3d_fwd_impl(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                         int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, void* stream, float* var_keep, int* deferred,
                         int sums_rows_per_chunk = 0, const dreg_bn_extra* ex = nullptr);
int dreg_bn3d_fwd(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                  int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, void* stream)
{
    return bn3d_fwd_impl(x, res, y, gamma, beta, running_mean, running_var, scale_shift, mean_rstd, workspace, B, V, C, eps, momentum, train, relu, dtype, stream, nullptr, nullptr);
}
// The same with the running-statistics update of a SMALL training-mode layer left to the caller: the per-grid variances go to var_keep
// ([B][C] floats the caller keeps) and *deferred = 1; the caller later runs dreg_bn_running_update_batched over all such layers.
// Large layers (their finalize kernel updates the running statistics itself) and eval mode run as dreg_bn3d_fwd: *deferred = 0.
int dreg_bn3d_fwd_defer_update(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                               int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, float* var_keep, int* deferred, void* stream)
{
    return bn3d_fwd_impl(x, res, y, gamma, beta, running_mean, running_var, scale_shift, mean_rstd, workspace, B, V, C, eps, momentum, train, relu, dtype, stream, var_keep, deferred);
}
// Training-mode forward whose statistics pass already happened: workspace = the chunk sums [B][V / rows_per_chunk][C][2] left by
// dreg_conv3d_igemm_bnstats (rows_per_chunk as returned there, > 0, dividing V).  Finalize (+ running statistics) and apply only.
int dreg_bn3d_fwd_from_sums(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* sums,
                            int rows_per_chunk, int B, int V, int C, float eps, float momentum, int relu, int dtype, void* stream)
{
    if (rows_per_chunk <= 0) return DREG_EINVAL;
    return bn3d_fwd_impl(x, res, y, gamma, beta, running_mean, running_var, scale_shift, mean_rstd, sums, B, V, C, eps, momentum, 1, relu, dtype, stream, nullptr, nullptr,
                         rows_per_chunk);
}
// The general form: dreg_bn3d_fwd_defer_update (sums_rows_per_chunk = 0) or dreg_bn3d_fwd_from_sums (> 0, workspace = the sums, train implied) with
// the optional extras of include/dreg_nerf.h's dreg_bn_extra handed over as an ARGUMENT (no per-thread "next call" state in this library).
int dreg_bn3d_fwd_ex(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                     int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, float* var_keep, int* deferred,
                     int sums_rows_per_chunk, const dreg_bn_extra* ex, void* stream)
{
    if (sums_rows_per_chunk < 0) return DREG_EINVAL;
    return bn3d_fwd_impl(x, res, y, gamma, beta, running_mean, running_var, scale_shift, mean_rstd, workspace, B, V, C, eps, momentum,
                         sums_rows_per_chunk > 0 ? 1 : train, relu, dtype, stream, var_keep, deferred, sums_rows_per_chunk, ex);
}
int dreg_bn_small(int B, int V, int C, int dtype) { return bn_small_ok(B, V, C, dtype == 0 ? 8 : 4) ? 1 : 0; }
// descs_dev: n records of 48 bytes { const float* mean_rstd; const float* var; float* running_mean; float* running_var; int B, V, C, block0; }
// with block0 = sum of ceil(C / 256) of the records before; workgroups [block_base, block_base + nblocks) run.
int dreg_bn_running_update_batched(const void* descs_dev, int n, int block_base, int nblocks, float momentum, void* stream)
{
    if (n <= 0 || nblocks <= 0) return DREG_OK;
    hipLaunchKernelGGL(bn_tail_batched_kernel<0>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const BnTailDesc*)descs_dev, n, block_base, momentum, 0);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// the same table layout with { const float* sums; unused; float* dgamma; float* dbeta; ... }
int dreg_bn_param_grad_batched(const void* descs_dev, int n, int block_base, int nblocks, int accumulate, void* stream)
{
    if (n <= 0 || nblocks <= 0) return DREG_OK;
    hipLaunchKernelGGL(bn_tail_batched_kernel<1>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const BnTailDesc*)descs_dev, n, block_base, 0.f, accumulate);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// The next large-path forward call's residual is the INPUT of a ReLU-free BatchNorm whose (scale, shift) are res_scale_shift (that layer was
// run with y == nullptr: statistics + finalize only); consumed by that one call.  See bn_apply_cols_kernel.
// The next small-volume (register-resident) forward / backward call takes its x (forward) / dy (backward) as the sum of split-K slices:
// see bn_small_fwd_kernel.  Consumed by that one call; DREG_EINVAL if the call does not take the register-resident path.
struct BnSplitkIn { const float* part = nullptr; int nsplit = 0; size_t slice = 0; };
// 1: a [B,V,C] layer of this dtype takes the register-resident one-launch kernels (what dreg_bn_extra's split-K slices need)
int dreg_bn_small_in_regs(int B, int V, int C, int dtype)
{
    return (g_bn_small_regs && bn_small_ok(B, V, C, dtype == 0 ? 8 : 4) && (V == 8 * BNS_ROWS || V == BNS_ROWS)) ? 1 : 0;
}
static int bn3d_fwd_impl(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                         int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, void* stream, float* var_keep, int* deferred,
                         int sums_rows_per_chunk, const dreg_bn_extra* ex)
{
    hipStream_t st = (hipStream_t)stream;
    const float* res_ss = ex ? ex->res_scale_shift : nullptr;
    BnSplitkIn ski;
    if (ex) { ski.part = ex->splitk_part; ski.nsplit = ex->splitk_nsplit; ski.slice = ex->splitk_slice; }
    if (ski.part && !(train && sums_rows_per_chunk == 0 && dreg_bn_small_in_regs(B, V, C, dtype))) return DREG_EINVAL;
    if (deferred) *deferred = 0;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G) return DREG_EINVAL;
    const int rpc = bn_rows_per_chunk(V), nch = (V + rpc - 1) / rpc;
    const int CG = C / G, slabs = (CG + 255) / 256;
    // sums_rows_per_chunk > 0 (training mode): the workspace already holds the chunk sums [B][V / that][C][2] (written by the producing
    // convolution's epilogue: dreg_conv3d_igemm_bnstats) — no statistics pass over x
    const bool presummed = train && sums_rows_per_chunk > 0;
    if (presummed && V % sums_rows_per_chunk != 0) return DREG_EINVAL;
    if (train && !presummed && bn_small_ok(B, V, C, G)) {      // 16^3 / 8^3 / 4^3 levels: statistics + apply in one launch, running statistics in a tiny second one
        if (res_ss || !y) return DREG_EINVAL;                  // (the deferred-output forms are large-path only)
        const dim3 g1(CG / BNS_COLS, B);
        float* var = var_keep ? var_keep : workspace;   // [B][C] biased variances (the workspace holds >= B * chunks * C * 2 floats)
#define BNS_FWD(Tt, NRv) hipLaunchKernelGGL((bn_small_fwd_kernel<Tt, NRv>), g1, dim3(256), 0, st, (const Tt*)x, (const Tt*)res, (Tt*)y, gamma, beta, \
                                            scale_shift, mean_rstd, var, V, C, eps, relu, ski.part, ski.nsplit, ski.slice)
        const int nr = !g_bn_small_regs ? 0 : (V == 8 * BNS_ROWS ? 8 : (V == BNS_ROWS ? 1 : 0));
        if (dtype == 0) { if (nr == 8) BNS_FWD(bf16_t, 8); else if (nr == 1) BNS_FWD(bf16_t, 1); else BNS_FWD(bf16_t, 0); }
        else { if (nr == 8) BNS_FWD(float, 8); else if (nr == 1) BNS_FWD(float, 1); else BNS_FWD(float, 0); }
#undef BNS_FWD
        DREG_LAUNCH_CHECK();
        if (var_keep && deferred) { *deferred = 1; return DREG_OK; }
        hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean_rstd, var, running_mean, running_var, B, V, C, momentum);
        DREG_LAUNCH_CHECK();
        return DREG_OK;
    }
    if (train && V < 2) return DREG_EINVAL;      // torch raises for one value per channel
    if (train && !presummed && !(g_bn_debug_skip & 1)) {
        dim3 grid(nch, B, slabs);
        if (dtype == 0) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t*)x, nullptr, nullptr, nullptr, workspace, V, C, rpc, 0);
        else hipLaunchKernelGGL((bn_partial_kernel<float, 0>), grid, dim3(256), 0, st, (const float*)x, nullptr, nullptr, nullptr, workspace, V, C, rpc, 0);
        DREG_LAUNCH_CHECK();
    }
    if (B > BN_MAX_GRIDS) return DREG_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, gamma, beta, running_mean, running_var,
                       scale_shift, mean_rstd, B, presummed ? V / sums_rows_per_chunk : nch, C, V, eps, momentum, train);
    DREG_LAUNCH_CHECK();
    if (!y) return DREG_OK;        // statistics / scale / shift only: the consumer applies them (dreg_bn_extra.res_scale_shift)
    const dim3 agrid(nch, B, slabs);
    if (dtype == 0) hipLaunchKernelGGL((bn_apply_cols_kernel<bf16_t, 4>), agrid, dim3(256), 0, st, (const bf16_t*)x, scale_shift, (const bf16_t*)res, (bf16_t*)y, V, C, rpc, relu, res ? res_ss : nullptr);
    else hipLaunchKernelGGL((bn_apply_cols_kernel<float, 4>), agrid, dim3(256), 0, st, (const float*)x, scale_shift, (const float*)res, (float*)y, V, C, rpc, relu, res ? res_ss : nullptr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Backward of y = [relu](bn(x) [+ res]) in training mode.  dres may be null.  coef: fp32 [B,C,2] scratch.
// y may be null when the forward had NO residual: the ReLU mask is then recomputed from x (x*scale + shift > 0), one tensor
// read less in both passes.
static int bn3d_bwd_impl(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                         void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                         int B, int V, int C, int relu, int accumulate, int dtype, void* stream, float* sums_keep, int* deferred, const dreg_bn_extra* ex = nullptr);
int dreg_bn3d_bwd(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                  void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                  int B, int V, int C, int relu, int accumulate, int dtype, void* stream)
{
    return bn3d_bwd_impl(x, dy, y, scale_shift, mean_rstd, dx, dres, dgamma, dbeta, coef, workspace, B, V, C, relu, accumulate, dtype, stream, nullptr, nullptr);
}
// The same with dgamma / dbeta of a SMALL layer left to the caller: the per-grid sums go to sums_keep ([B][C][2] floats) and
// *deferred = 1 (dreg_bn_param_grad_batched finishes them); large layers run as dreg_bn3d_bwd: *deferred = 0.
int dreg_bn3d_bwd_defer_params(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                               void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                               int B, int V, int C, int relu, int accumulate, int dtype, float* sums_keep, int* deferred, void* stream)
{
    return bn3d_bwd_impl(x, dy, y, scale_shift, mean_rstd, dx, dres, dgamma, dbeta, coef, workspace, B, V, C, relu, accumulate, dtype, stream, sums_keep, deferred);
}
// ... and with dy handed over as un-summed split-K slices (ex->splitk_*: register-resident small path only)
int dreg_bn3d_bwd_ex(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                     void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                     int B, int V, int C, int relu, int accumulate, int dtype, float* sums_keep, int* deferred, const dreg_bn_extra* ex, void* stream)
{
    return bn3d_bwd_impl(x, dy, y, scale_shift, mean_rstd, dx, dres, dgamma, dbeta, coef, workspace, B, V, C, relu, accumulate, dtype, stream, sums_keep, deferred, ex);
}
static int bn3d_bwd_impl(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                         void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                         int B, int V, int C, int relu, int accumulate, int dtype, void* stream, float* sums_keep, int* deferred, const dreg_bn_extra* ex)
{
    hipStream_t st = (hipStream_t)stream;
    BnSplitkIn ski;
    if (ex) { ski.part = ex->splitk_part; ski.nsplit = ex->splitk_nsplit; ski.slice = ex->splitk_slice; }
    if (ski.part && !dreg_bn_small_in_regs(B, V, C, dtype)) return DREG_EINVAL;
    if (deferred) *deferred = 0;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G) return DREG_EINVAL;
    const int rpc = bn_rows_per_chunk(V), nch = (V + rpc - 1) / rpc;
    const int CG = C / G, slabs = (CG + 255) / 256;
    if (bn_small_ok(B, V, C, G)) {
        const dim3 g1(CG / BNS_COLS, B);
        float* sums = sums_keep ? sums_keep : coef;   // [B][C][2]: per-grid (sum g, sum g xhat)
#define BNS_BWD(Tt, NRv) hipLaunchKernelGGL((bn_small_bwd_kernel<Tt, NRv>), g1, dim3(256), 0, st, (const Tt*)x, (const Tt*)dy, (const Tt*)y, scale_shift, mean_rstd, \
                                            (Tt*)dx, (Tt*)dres, sums, V, C, relu, ski.part, ski.nsplit, ski.slice)
        const int nr = !g_bn_small_regs ? 0 : (V == 8 * BNS_ROWS ? 8 : (V == BNS_ROWS ? 1 : 0));
        if (dtype == 0) { if (nr == 8) BNS_BWD(bf16_t, 8); else if (nr == 1) BNS_BWD(bf16_t, 1); else BNS_BWD(bf16_t, 0); }
        else { if (nr == 8) BNS_BWD(float, 8); else if (nr == 1) BNS_BWD(float, 1); else BNS_BWD(float, 0); }
#undef BNS_BWD
        DREG_LAUNCH_CHECK();
        if (sums_keep && deferred) { *deferred = 1; return DREG_OK; }
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, dgamma, dbeta, B, C, accumulate);
        DREG_LAUNCH_CHECK();
        return DREG_OK;
    }
    dim3 grid(nch, B, slabs);
    // residual + ReLU layers (the last BatchNorm of a bottleneck): the masked gradient the statistics pass forms IS the residual branch's
    // gradient — it is stored there, and the apply pass reads it back instead of dy and y (one activation-sized read less, same values)
    const int g_stored = (dres && relu && y && dres != dy && dres != dx && g_bn_store_g && !(g_bn_debug_skip & 2)) ? 1 : 0;
    if (g_bn_debug_skip & 2) {}
    else if (dtype == 0) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean_rstd, workspace, V, C, rpc, relu, scale_shift, g_stored ? (bf16_t*)dres : nullptr);
    else hipLaunchKernelGGL((bn_partial_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (const float*)y, mean_rstd, workspace, V, C, rpc, relu, scale_shift, g_stored ? (float*)dres : nullptr);
    DREG_LAUNCH_CHECK();
    if (B > BN_MAX_GRIDS) return DREG_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, coef, dgamma, dbeta, B, nch, C, V, accumulate);
    DREG_LAUNCH_CHECK();
    const size_t tg = (size_t)B * V * CG;
    (void)tg;
    if (dtype == 0) hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean_rstd, scale_shift, coef, (bf16_t*)dx, (bf16_t*)dres, V, C, rpc, relu, g_stored);
    else hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (const float*)y, mean_rstd, scale_shift, coef, (float*)dx, (float*)dres, V, C, rpc, relu, g_stored);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Fused stem (bf16): statistics + finalize of dreg_bn3d_fwd, then pooled = maxpool3(relu(bn(x))) without the full-resolution
// activation in between.  x: [B,Di,Hi,Wi,C]; pooled: [B,Do,Ho,Wo,C]; argmax: uint8 per pooled element; no residual.
int dreg_bn_relu_maxpool_fwd(const void* x, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                             int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, float eps, float momentum, int train, int relu, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int V = Di * Hi * Wi;
    if (C % 8 || B > BN_MAX_GRIDS || (train && V < 2)) return DREG_EINVAL;
    const int rpc = bn_rows_per_chunk(V), nch = (V + rpc - 1) / rpc;
    const int CG = C / 8, slabs = (CG + 255) / 256;
    if (train) {
        hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 0>), dim3(nch, B, slabs), dim3(256), 0, st, (const bf16_t*)x, nullptr, nullptr, nullptr, workspace, V, C, rpc, 0);
        DREG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, gamma, beta, running_mean, running_var,
                       scale_shift, mean_rstd, B, nch, C, V, eps, momentum, train);
    DREG_LAUNCH_CHECK();
    const size_t total = (size_t)B * Do * Ho * Wo * CG;
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)x, scale_shift, (bf16_t*)pooled, argmax,
                       B, Di, Hi, Wi, Do, Ho, Wo, C, relu);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Its backward: dx = d(loss)/dx from the POOLED gradient dp; dgamma / dbeta as dreg_bn3d_bwd.  coef: fp32 [B,C,2] scratch,
// workspace as dreg_bn3d_fwd.
int dreg_bn_relu_maxpool_bwd(const void* x, const void* dp, const uint8_t* argmax, const float* scale_shift, const float* mean_rstd,
                             void* dx, float* dgamma, float* dbeta, float* coef, float* workspace,
                             int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu, int accumulate, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int V = Di * Hi * Wi;
    if (C % 8 || B > BN_MAX_GRIDS) return DREG_EINVAL;
    const int rpc = bn_rows_per_chunk(V), nch = (V + rpc - 1) / rpc;
    const int CG = C / 8, slabs = (CG + 255) / 256;
    const dim3 grid(nch, B, slabs);
    hipLaunchKernelGGL((bn_pool_bwd_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dp, argmax, mean_rstd, scale_shift, nullptr, workspace,
                       (bf16_t*)nullptr, Di, Hi, Wi, Do, Ho, Wo, C, rpc, relu);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, coef, dgamma, dbeta, B, nch, C, V, accumulate);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL((bn_pool_bwd_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dp, argmax, mean_rstd, scale_shift, coef, nullptr,
                       (bf16_t*)dx, Di, Hi, Wi, Do, Ho, Wo, C, rpc, relu);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Stem over a sparse volume (see the kernels above).  x: [B,Di,Hi,Wi,C] bf16, zero outside `rows` (ascending flat row indices, n of them);
// pooled / argmax as dreg_bn_relu_maxpool_fwd; xam: bf16 [B,Do,Ho,Wo,C] (raw x of the arg-max voxels, for the backward sums); pmask: B*Do*Ho*Wo
// bytes of scratch; act (optional): the dense-layout activation, written on rows_a only.  workspace: fp32 [B][64][C][2].
DREG_KNOB(int, g_sstem_pool_blocks, 8192);
#ifdef DREG_PROBE
void dreg_sstem_set_pool_blocks(int n) { g_sstem_pool_blocks = n > 0 ? n : 8192; }    // measurement (tools/bench_sparse_stem.py)
#endif
size_t dreg_sparse_stem_workspace_floats(int B, int Do, int Ho, int Wo, int C)
{
    const int Vo = Do * Ho * Wo;
    return (size_t)B * (dreg_bn_num_chunks(Vo) + SSTEM_NCH) * C * 2;
}
int dreg_sparse_stem_fwd(const void* x, const int* rows, int n, const int* rows_a, int n_a, void* act, void* pooled, uint8_t* argmax, void* xam, uint8_t* pmask,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, float eps, float momentum, int train, int relu, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int V = Di * Hi * Wi;
    if (C % 8 || B > BN_MAX_GRIDS || (train && V < 2) || n < 0 || n_a < 0 || (size_t)B * V > 0x7fffffffull) return DREG_EINVAL;
    const int CG = C / 8, slabs = (CG + 255) / 256;
    if (train) {
        hipLaunchKernelGGL(sstem_stats_kernel, dim3(SSTEM_NCH, B, slabs), dim3(256), 0, st, (const bf16_t*)x, rows, n, workspace, V, C);
        DREG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, gamma, beta, running_mean, running_var,
                       scale_shift, mean_rstd, B, SSTEM_NCH, C, V, eps, momentum, train);
    DREG_LAUNCH_CHECK();
    const size_t Po = (size_t)B * Do * Ho * Wo;
    if (Po * CG > 0xffffffffull) return DREG_EINVAL;
    if (hipMemsetAsync(pmask, 0, Po, st) != hipSuccess) return DREG_ELAUNCH;
    if (n > 0) hipLaunchKernelGGL(sstem_pool_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rows, n, pmask, Di, Hi, Wi, Do, Ho, Wo);
    hipLaunchKernelGGL(sstem_pool_fwd_kernel, dim3(nblocks(Po * CG, 256, g_sstem_pool_blocks)), dim3(256), 0, st, (const bf16_t*)x, pmask, scale_shift, (bf16_t*)pooled, argmax, (bf16_t*)xam,
                       B, Di, Hi, Wi, Do, Ho, Wo, C, relu);
    if (act && n_a > 0) hipLaunchKernelGGL(sstem_apply_rows_kernel, dim3(SSTEM_NCH_ROWS, B, slabs), dim3(256), 0, st, (const bf16_t*)x, rows_a, n_a, scale_shift, (bf16_t*)act, V, C, relu);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Its backward.  dp: gradient of `pooled`; dl (optional): gradient of the activation, non-zero on rows_a only (zero elsewhere); dx is
// written on `rows` only.  dgamma / dbeta as dreg_bn3d_bwd; coef: fp32 [B,C,2] scratch; workspace: dreg_sparse_stem_workspace_floats.
int dreg_sparse_stem_bwd(const void* x, const void* dp, const uint8_t* argmax, const void* xam, const void* dl, const int* rows_a, int n_a,
                         const int* rows, int n, const float* scale_shift, const float* mean_rstd, void* dx, float* dgamma, float* dbeta, float* coef, float* workspace,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu, int accumulate, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int V = Di * Hi * Wi, Vo = Do * Ho * Wo;
    if (C % 8 || B > BN_MAX_GRIDS || n < 0 || n_a < 0 || (size_t)B * V > 0x7fffffffull) return DREG_EINVAL;
    const int CG = C / 8, slabs = (CG + 255) / 256;
    const int rpc = bn_rows_per_chunk(Vo), nchp = (Vo + rpc - 1) / rpc;
    const bool lat = dl && n_a > 0;
    const int nch = nchp + (lat ? SSTEM_NCH : 0);
    hipLaunchKernelGGL(sstem_bwd_sums_kernel, dim3(nch, B, slabs), dim3(256), 0, st, (const bf16_t*)xam, (const bf16_t*)dp, (const bf16_t*)x, (const bf16_t*)dl, rows_a, n_a,
                       mean_rstd, scale_shift, workspace, V, Vo, C, rpc, nchp, relu);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * (B < 8 ? B : 8)), 0, st, workspace, coef, dgamma, dbeta, B, nch, C, V, accumulate);
    DREG_LAUNCH_CHECK();
    if (n > 0) hipLaunchKernelGGL(sstem_bwd_rows_kernel, dim3(SSTEM_NCH_ROWS, B, slabs), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dp, argmax, lat ? (const bf16_t*)dl : nullptr,
                                  rows, n, mean_rstd, scale_shift, coef, (bf16_t*)dx, Di, Hi, Wi, Do, Ho, Wo, C, relu);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_maxpool3d_fwd(const void* x, void* y, uint8_t* argmax, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    const size_t total = (size_t)B * Do * Ho * Wo * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, B, Di, Hi, Wi, Do, Ho, Wo, C);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, B, Di, Hi, Wi, Do, Ho, Wo, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// accumulate: dx += the un-pooled gradient (dx already holds another contribution: fp32 sum, one rounding)
int dreg_maxpool3d_bwd_acc(const void* dy, const uint8_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int accumulate,
                           int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    const size_t total = (size_t)B * Di * Hi * Wi * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, B, Di, Hi, Wi, Do, Ho, Wo, C, accumulate);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, B, Di, Hi, Wi, Do, Ho, Wo, C, accumulate);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_maxpool3d_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int dtype, void* stream)
{
    return dreg_maxpool3d_bwd_acc(dy, argmax, dx, B, Di, Hi, Wi, Do, Ho, Wo, C, 0, dtype, stream);
}

// Row-list form: out[rows] = sums of the 2^3 children in g; `out` elsewhere is left as it is (the caller keeps it zero: the gradient
// of an upsample-add whose fine side only holds values on an active set is zero off the parents of that set).
int dreg_downsample_sum_rows(const void* g, void* out, const int* rows, int nrows, int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int C, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G || nrows < 0) return DREG_EINVAL;
    if (nrows == 0) return DREG_OK;
    const size_t total = (size_t)nrows * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(downsample_sum_rows_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)out, rows, nrows, Df, Hf, Wf, Dc, Hc, Wc, C);
    else hipLaunchKernelGGL(downsample_sum_rows_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)g, (float*)out, rows, nrows, Df, Hf, Wf, Dc, Hc, Wc, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_downsample_sum(const void* g, void* out, int B, int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int C, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    const size_t total = (size_t)B * Dc * Hc * Wc * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(downsample_sum_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)out, B, Df, Hf, Wf, Dc, Hc, Wc, C);
    else hipLaunchKernelGGL(downsample_sum_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)g, (float*)out, B, Df, Hf, Wf, Dc, Hc, Wc, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

static inline int colsum_rows_per_chunk(size_t M) { size_t r = (M + 255) / 256; r = (r + 31) / 32 * 32; return (int)(r < 32 ? 32 : r); }
size_t dreg_colsum_workspace_bytes(size_t M, int C) { const size_t r = colsum_rows_per_chunk(M); return ((M + r - 1) / r) * C * sizeof(float); }
int dreg_colsum(const void* g, float* out, float* workspace, size_t M, int C, int accumulate, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G) return DREG_EINVAL;
    const int rpc = colsum_rows_per_chunk(M);
    const int nch = (int)((M + rpc - 1) / rpc);
    const int CG = C / G, slabs = (CG + 255) / 256;
    if (dtype == 0) hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(nch, slabs), dim3(256), 0, st, (const bf16_t*)g, workspace, M, C, rpc);
    else hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(nch, slabs), dim3(256), 0, st, (const float*)g, workspace, M, C, rpc);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, out, nch, C, accumulate);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Batched form.  descs_dev: n records of 56 bytes { const bf16* g [M][C]; float* out [C]; float* partial (dreg_colsum_workspace_bytes(M, C),
// one per record); int M, C, rpc (dreg_colsum_rows_per_chunk(M)), nch = ceil(M / rpc), pblock0 (sum of nch of the records before),
// fblock0 (sum of ceil(C / 4) before), accumulate, pad }; total_pblocks / total_fblocks = the sums over all records.  bf16, C % 8 == 0, C <= 2048.
int dreg_colsum_rows_per_chunk(size_t M) { return colsum_rows_per_chunk(M); }
int dreg_colsum_batched(const void* descs_dev, int n, int total_pblocks, int total_fblocks, void* stream)
{
    if (n <= 0 || total_pblocks <= 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_batched_kernel, dim3(total_pblocks), dim3(256), 0, st, (const ColsumDesc*)descs_dev, n);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_batched_kernel, dim3(total_fblocks), dim3(256), 0, st, (const ColsumDesc*)descs_dev, n);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// out[c] (+)= sum over the row list of g[rows[i]][c]  (bias gradient of an active-set convolution).  workspace as dreg_colsum(nrows, C).
int dreg_colsum_rows(const void* g, const int* rows, int nrows, float* out, float* workspace, int C, int accumulate, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G || nrows <= 0) return DREG_EINVAL;
    const int rpc = colsum_rows_per_chunk((size_t)nrows);
    const int nch = (nrows + rpc - 1) / rpc;
    const int CG = C / G, slabs = (CG + 255) / 256;
    if (dtype == 0) hipLaunchKernelGGL(colsum_rows_partial_kernel<bf16_t>, dim3(nch, slabs), dim3(256), 0, st, (const bf16_t*)g, rows, workspace, nrows, C, rpc);
    else hipLaunchKernelGGL(colsum_rows_partial_kernel<float>, dim3(nch, slabs), dim3(256), 0, st, (const float*)g, rows, workspace, nrows, C, rpc);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, out, nch, C, accumulate);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// p1: [B,d,h,w,C] (dtype), idx: int64 [N] flat fine-grid indices, pt_batch: int32 [N] grid id of every point.
// out: [N,C] fp32 (out_f32 = 1) or dtype.
int dreg_trilinear_gather_fwd(const void* p1, const int64_t* idx, const int* pt_batch, void* out, int N, int d, int h, int w, int C,
                              int Zr, int Xr, int Yr, int dtype, int out_f32, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    const size_t total = (size_t)N * (C / G);
    if (N == 0) return DREG_OK;
    if (dtype == 0) {
        if (out_f32) hipLaunchKernelGGL((trilinear_gather_fwd_kernel<bf16_t, float>), dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)p1, idx, pt_batch, (float*)out, N, d, h, w, C, Zr, Xr, Yr);
        else hipLaunchKernelGGL((trilinear_gather_fwd_kernel<bf16_t, bf16_t>), dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)p1, idx, pt_batch, (bf16_t*)out, N, d, h, w, C, Zr, Xr, Yr);
    } else hipLaunchKernelGGL((trilinear_gather_fwd_kernel<float, float>), dim3(nblocks(total)), dim3(256), 0, st, (const float*)p1, idx, pt_batch, (float*)out, N, d, h, w, C, Zr, Xr, Yr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// dfeat: [N,C] fp32 -> dp1_f32 [B,d,h,w,C] fp32 (must be zeroed by the caller), atomically accumulated.
int dreg_trilinear_gather_bwd(const float* dfeat, const int64_t* idx, const int* pt_batch, float* dp1_f32, int N, int d, int h, int w, int C,
                              int Zr, int Xr, int Yr, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return DREG_OK;
    const size_t total = (size_t)N * C;
    hipLaunchKernelGGL(trilinear_gather_bwd_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, dfeat, idx, pt_batch, dp1_f32, N, d, h, w, C, Zr, Xr, Yr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_cast_from_f32(const float* in, void* out, size_t n, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, in, (bf16_t*)out, n);
    else hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, in, (float*)out, n);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// grids: DEVICE array of B pointers to fp32 [7,Z,X,Y] grids -> out [B,Z,X,Y,8] (dtype): rgba + 4 zero channels.
// _occ: also inocc byte [B,Z,X] (may be null) = 1 for every Y-row with a non-zero value.
int dreg_pack_rgba_grids_occ(const void* grids, void* out, uint8_t* inocc, int B, int Z, int X, int Y, int dtype, void* stream)
{
    const size_t V = (size_t)Z * X * Y;
    if (B <= 0 || V == 0) return DREG_OK;
    if (inocc && hipMemsetAsync(inocc, 0, (size_t)B * Z * X, (hipStream_t)stream) != hipSuccess) return DREG_ELAUNCH;
    if (dtype == 0) hipLaunchKernelGGL(pack_rgba_kernel<bf16_t>, dim3(nblocks(V * B)), dim3(256), 0, (hipStream_t)stream, (const float* const*)grids, (bf16_t*)out, V, B, inocc, Y);
    else hipLaunchKernelGGL(pack_rgba_kernel<float>, dim3(nblocks(V * B)), dim3(256), 0, (hipStream_t)stream, (const float* const*)grids, (float*)out, V, B, inocc, Y);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_pack_rgba_grids(const void* grids, void* out, int B, int Z, int X, int Y, int dtype, void* stream)
{
    return dreg_pack_rgba_grids_occ(grids, out, nullptr, B, Z, X, Y, dtype, stream);
}
// Output-row occupancy of a convolution over a volume whose non-empty input rows are flagged in inocc (byte [B, Di, Hi]):
// rowocc byte [B, Do, Ho].  dreg_conv3d_igemm_occ / dreg_conv3d_wgrad_occ skip the rows flagged 0 (their result is exactly zero).
int dreg_conv_row_occupancy(const uint8_t* inocc, uint8_t* rowocc, int B, int Di, int Hi, int Do, int Ho, int ksz, int stride, int pad, void* stream)
{
    const int n = B * Do * Ho;
    if (n <= 0) return DREG_OK;
    hipLaunchKernelGGL(conv_row_occupancy_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, inocc, rowocc, B, Di, Hi, Do, Ho, ksz, stride, pad);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Sparse staging: out [B,Z,X,Y,8] (dtype) = zeros + rgba of the occupied voxels (vals fp32 [N,7], idx int64 [N], pt_batch int32 [N])
int dreg_pack_rgba_sparse_occ(const float* vals, const int64_t* idx, const int* pt_batch, void* out, uint8_t* inocc, int N, int B, int Z, int X, int Y,
                              int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)B * Z * X * Y * 8 * (dtype == 0 ? 2 : 4);
    if (bytes == 0) return DREG_OK;
    if (dreg_fill_zero(out, bytes, st) != DREG_OK) return DREG_ELAUNCH;
    if (inocc && hipMemsetAsync(inocc, 0, (size_t)B * Z * X, st) != hipSuccess) return DREG_ELAUNCH;
    if (N <= 0) return DREG_OK;
    if (dtype == 0) hipLaunchKernelGGL(scatter_rgba_kernel<bf16_t>, dim3((N + 255) / 256), dim3(256), 0, st, vals, idx, pt_batch, (bf16_t*)out, N, Z, X, Y, inocc);
    else hipLaunchKernelGGL(scatter_rgba_kernel<float>, dim3((N + 255) / 256), dim3(256), 0, st, vals, idx, pt_batch, (float*)out, N, Z, X, Y, inocc);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_pack_rgba_sparse(const float* vals, const int64_t* idx, const int* pt_batch, void* out, int N, int B, int Z, int X, int Y,
                          int dtype, void* stream)
{
    return dreg_pack_rgba_sparse_occ(vals, idx, pt_batch, out, nullptr, N, B, Z, X, Y, dtype, stream);
}
// xyz fp32 [N,3] of the occupied voxels: idx int64 [N] flat (x*Yr + y)*Zr + z, pt_batch int32 [N] grid ids
int dreg_gather_grid_xyz(const void* grids, const int64_t* idx, const int* pt_batch, float* xyz, int N, int Zr, int Xr, int Yr, void* stream)
{
    if (N <= 0) return DREG_OK;
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float* const*)grids, idx, pt_batch, xyz, N, Zr, Xr, Yr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Zero fill at HBM rate (hipMemsetAsync's fill kernel runs 256 workgroups: ~2 TB/s; the backward pass clears 134-268 MB gradient buffers).
// bytes % 16 == 0, p 16-byte aligned.
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4* __restrict__ p, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int dreg_fill_zero(void* p, size_t bytes, void* stream)
{
    if (bytes == 0) return DREG_OK;
    if ((bytes & 15) || ((uintptr_t)p & 15)) return hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) == hipSuccess ? DREG_OK : DREG_ELAUNCH;
    const size_t n16 = bytes >> 4;
    hipLaunchKernelGGL(fill_zero_kernel, dim3(nblocks(n16, 256, 16384)), dim3(256), 0, (hipStream_t)stream, (uint4*)p, n16);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Guard bands (debug mode of the executors: dreg_exec_opts.guard).  n bands of band_bytes (a multiple of 16) at base + offsets[i] are filled with the
// poison word 0xA5C3A5C3; dreg_guard_scan counts the bands that no longer hold it: result (device int64 [4], zeroed by the scan's own first kernel) =
// { bands with a changed byte, smallest such band index (n if none), byte offset of its first changed 16-byte word, changed words in total }.
__global__ void guard_fill_kernel(char* base, const unsigned long long* __restrict__ offsets, int band_bytes)
{
    uint4* p = reinterpret_cast<uint4*>(base + offsets[blockIdx.x]);
    for (int i = threadIdx.x; i < band_bytes / 16; i += blockDim.x) p[i] = make_uint4(0xA5C3A5C3u, 0xA5C3A5C3u, 0xA5C3A5C3u, 0xA5C3A5C3u);
}
__global__ void guard_reset_kernel(long long* result, int n) { result[0] = 0; result[1] = n; result[2] = 0; result[3] = 0; }
__global__ void guard_scan_kernel(const char* base, const unsigned long long* __restrict__ offsets, int band_bytes, long long* result)
{
    const uint4* p = reinterpret_cast<const uint4*>(base + offsets[blockIdx.x]);
    __shared__ int first, count;
    if (threadIdx.x == 0) { first = 0x7fffffff; count = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < band_bytes / 16; i += blockDim.x) {
        const uint4 v = p[i];
        if (v.x != 0xA5C3A5C3u || v.y != 0xA5C3A5C3u || v.z != 0xA5C3A5C3u || v.w != 0xA5C3A5C3u) { atomicMin(&first, i * 16); atomicAdd(&count, 1); }
    }
    __syncthreads();
    if (threadIdx.x == 0 && count) {
        atomicAdd((unsigned long long*)&result[0], 1ull);
        atomicAdd((unsigned long long*)&result[3], (unsigned long long)count);
        atomicMin(&result[1], (long long)blockIdx.x);
    }
}
__global__ void guard_first_kernel(const char* base, const unsigned long long* __restrict__ offsets, int band_bytes, long long* result, int n)
{
    const long long b = result[1];
    if (b >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + offsets[b]);
    for (int i = 0; i < band_bytes / 16; ++i) {
        const uint4 v = p[i];
        if (v.x != 0xA5C3A5C3u || v.y != 0xA5C3A5C3u || v.z != 0xA5C3A5C3u || v.w != 0xA5C3A5C3u) { result[2] = (long long)i * 16; return; }
    }
}
int dreg_guard_fill(void* base, const void* offsets_dev, int n, int band_bytes, void* stream)
{
    if (n <= 0) return DREG_OK;
    if (band_bytes <= 0 || (band_bytes & 15)) return DREG_EINVAL;
    hipLaunchKernelGGL(guard_fill_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (char*)base, (const unsigned long long*)offsets_dev, band_bytes);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_guard_scan(const void* base, const void* offsets_dev, int n, int band_bytes, void* result_dev, void* stream)
{
    if (band_bytes <= 0 || (band_bytes & 15) || !result_dev) return DREG_EINVAL;
    hipLaunchKernelGGL(guard_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)result_dev, n);
    if (n > 0) {
        hipLaunchKernelGGL(guard_scan_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const char*)base, (const unsigned long long*)offsets_dev, band_bytes, (long long*)result_dev);
        hipLaunchKernelGGL(guard_first_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const char*)base, (const unsigned long long*)offsets_dev, band_bytes, (long long*)result_dev, n);
    }
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// dst += src, n elements (n % 8 == 0 for bf16, % 4 for fp32)
int dreg_add_inplace(void* dst, const void* src, size_t n, int dtype, void* stream)
{
    const int G = dtype == 0 ? 8 : 4;
    if (n % G) return DREG_EINVAL;
    if (n == 0) return DREG_OK;
    if (dtype == 0) hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(nblocks(n / G)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, (const bf16_t*)src, n / G);
    else hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(nblocks(n / G)), dim3(256), 0, (hipStream_t)stream, (float*)dst, (const float*)src, n / G);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Active-set gather backward: dp1 [B,d,h,w,C] (dtype) is zero-filled here and receives the gradient on the S1 rows only.
// rows1 / n1 / map1: list, length and inverse map from dreg_active_sets; comp: fp32 [n1, C] scratch.
int dreg_trilinear_gather_bwd_rows(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1, const int* map1,
                                   float* comp, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int G = dtype == 0 ? 8 : 4;
    if (C % G) return DREG_EINVAL;
    const size_t dense_bytes = (size_t)B * d * h * w * C * (dtype == 0 ? 2 : 4);
    if (dreg_fill_zero(dp1, dense_bytes, st) != DREG_OK) return DREG_ELAUNCH;
    if (N == 0 || n1 == 0) return DREG_OK;
    if (hipMemsetAsync(comp, 0, (size_t)n1 * C * sizeof(float), st) != hipSuccess) return DREG_ELAUNCH;
    hipLaunchKernelGGL(trilinear_gather_bwd_rows_kernel, dim3(nblocks((size_t)N * C)), dim3(256), 0, st, dfeat, idx, pt_batch, map1, comp, N, d, h, w, C, Zr, Xr, Yr);
    DREG_LAUNCH_CHECK();
    const size_t tg = (size_t)n1 * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(scatter_rows_cast_kernel<bf16_t>, dim3(nblocks(tg)), dim3(256), 0, st, comp, rows1, (bf16_t*)dp1, tg, C);
    else hipLaunchKernelGGL(scatter_rows_cast_kernel<float>, dim3(nblocks(tg)), dim3(256), 0, st, comp, rows1, (float*)dp1, tg, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_zero_rows(void* buf, const int* rows, int nrows, int C, int dtype, void* stream)
{
    const int G = dtype == 0 ? 8 : 4;
    if (C % G || nrows < 0) return DREG_EINVAL;
    if (nrows == 0) return DREG_OK;
    const size_t tg = (size_t)nrows * (C / G);
    if (dtype == 0) hipLaunchKernelGGL(zero_rows_kernel<bf16_t>, dim3(nblocks(tg)), dim3(256), 0, (hipStream_t)stream, rows, (bf16_t*)buf, tg, C);
    else hipLaunchKernelGGL(zero_rows_kernel<float>, dim3(nblocks(tg)), dim3(256), 0, (hipStream_t)stream, rows, (float*)buf, tg, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Deterministic variant (no atomics): fine_map int32 [B*Zr*Xr*Yr] scratch.  C must be 64, 128, 192 or 256.
// zero_dense = 0: dp1 is already zero outside rows1 (a persistent buffer the caller cleans with dreg_zero_rows); only rows1 are written.
static int tri_bwd_gather(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                          int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                          int dtype, int zero_dense, void* stream, const void* seg = nullptr);
// Trilinear gather fused with the first voxel-average round of ONE pair (see gather_segment_mean_kernel): p1 [B,d,h,w,C] (dtype 0 bf16 / 1
// fp32), idx / pt_batch of ALL points, point_start = index of the pair's first point, order / starts / n_out = the round's plan
// (dreg_voxel_downsample_plan), out fp32 [M, C] (rows beyond *n_out untouched).  C a multiple of 4.
int dreg_gather_segment_mean(const void* p1, const int64_t* idx, const int* pt_batch, int point_start, const uint32_t* order, const uint32_t* starts,
                             const int* n_out, float* out, int M, int d, int h, int w, int C, int Zr, int Xr, int Yr, int dtype, void* stream)
{
    if (M <= 0) return DREG_OK;
    if (C % 4 || point_start < 0) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(gather_segment_mean_kernel<bf16_t>, dim3((M + 3) / 4), dim3(256), 0, st, (const bf16_t*)p1, idx, pt_batch, point_start, order, starts, n_out, out, d, h, w, C, Zr, Xr, Yr);
    else hipLaunchKernelGGL(gather_segment_mean_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, st, (const float*)p1, idx, pt_batch, point_start, order, starts, n_out, out, d, h, w, C, Zr, Xr, Yr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// The same with the gradient of the gathered features given through the first voxel-average round that consumed them: g1 fp32 [rows, C] =
// gradient of that round's outputs (the pairs' blocks one after the other), seg_descs: device int64 [B][4] = per grid (inv_seg pointer,
// inv_cnt pointer, index of its pair's first point, first g1 row of its pair) — see TriSegDesc.  Bit-identical to running
// dreg_voxel_downsample_bwd per pair and handing the concatenated result to dreg_trilinear_gather_bwd_gather[_rows_only].
int dreg_trilinear_gather_bwd_gather_seg(const float* g1, const void* seg_descs, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                         int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                                         int dtype, int zero_dense, void* stream)
{ return seg_descs ? tri_bwd_gather(g1, idx, pt_batch, rows1, n1, fine_map, dp1, N, B, d, h, w, C, Zr, Xr, Yr, dtype, zero_dense, stream, seg_descs) : DREG_EINVAL; }
int dreg_trilinear_gather_bwd_gather(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                     int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                                     int dtype, void* stream)
{ return tri_bwd_gather(dfeat, idx, pt_batch, rows1, n1, fine_map, dp1, N, B, d, h, w, C, Zr, Xr, Yr, dtype, 1, stream); }
int dreg_trilinear_gather_bwd_gather_rows_only(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                               int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                                               int dtype, void* stream)
{ return tri_bwd_gather(dfeat, idx, pt_batch, rows1, n1, fine_map, dp1, N, B, d, h, w, C, Zr, Xr, Yr, dtype, 0, stream); }
static int tri_bwd_gather(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                          int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                          int dtype, int zero_dense, void* stream, const void* seg)
{
    hipStream_t st = (hipStream_t)stream;
    if (C % 64 || C > 256) return DREG_EINVAL;
    const size_t dense_bytes = (size_t)B * d * h * w * C * (dtype == 0 ? 2 : 4);
    if (zero_dense && dreg_fill_zero(dp1, dense_bytes, st) != DREG_OK) return DREG_ELAUNCH;
    if (N == 0 || n1 == 0) return DREG_OK;
    const size_t Vf = (size_t)Zr * Xr * Yr;
    if (hipMemsetAsync(fine_map, 0xff, (size_t)B * Vf * sizeof(int), st) != hipSuccess) return DREG_ELAUNCH;
    hipLaunchKernelGGL(fine_map_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, st, idx, pt_batch, fine_map, N, Vf);
    DREG_LAUNCH_CHECK();
    const dim3 grid((n1 + 3) / 4);
#define TG_LAUNCH(T, CPL) do { if (seg) hipLaunchKernelGGL((trilinear_gather_bwd_gather_kernel<T, CPL, true>), grid, dim3(256), 0, st, dfeat, fine_map, rows1, n1, (T*)dp1, d, h, w, Zr, Xr, Yr, (const TriSegDesc*)seg); \
        else hipLaunchKernelGGL((trilinear_gather_bwd_gather_kernel<T, CPL, false>), grid, dim3(256), 0, st, dfeat, fine_map, rows1, n1, (T*)dp1, d, h, w, Zr, Xr, Yr, (const TriSegDesc*)nullptr); } while (0)
    const int cpl = C / 64;
    if (dtype == 0) { if (cpl == 1) TG_LAUNCH(bf16_t, 1); else if (cpl == 2) TG_LAUNCH(bf16_t, 2); else if (cpl == 3) TG_LAUNCH(bf16_t, 3); else TG_LAUNCH(bf16_t, 4); }
    else { if (cpl == 1) TG_LAUNCH(float, 1); else if (cpl == 2) TG_LAUNCH(float, 2); else if (cpl == 3) TG_LAUNCH(float, 3); else TG_LAUNCH(float, 4); }
#undef TG_LAUNCH
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ active sets of the FPN head
// S1 = coarse voxels touched by the trilinear gather (all 8 corners of every occupied fine voxel, tri_axis() above),
// S2 = S1 dilated by the 3^3 footprint, S3 = S2 dilated again.  Output: three ascending int32 row lists (flat indices into
// [B,d,h,w]) + their lengths, and map1[v] = rank of v in S1 (or -1).  One C call, no host round trip inside.
__global__ void aset_mark_kernel(const int64_t* __restrict__ idx, const int* __restrict__ pt_batch, uint8_t* __restrict__ f1,
                                 int N, int d, int h, int w, int Zr, int Xr, int Yr)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t f = idx[n];
    const int z = (int)(f % Zr), y = (int)((f / Zr) % Yr), x = (int)(f / ((int64_t)Zr * Yr));
    const int b = pt_batch[n];
    const TriAxis az = tri_axis(z, Zr, d), ax = tri_axis(x, Xr, h), ay = tri_axis(y, Yr, w);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int zi = (c & 4) ? az.i1 : az.i0, xi = (c & 2) ? ax.i1 : ax.i0, yi = (c & 1) ? ay.i1 : ay.i0;
        f1[(((size_t)b * d + zi) * h + xi) * w + yi] = 1;
    }
}
__global__ void aset_dilate_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int d, int h, int w)
{
    const size_t V = (size_t)B * d * h * w;
    const size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int yi = (int)(v % w), xi = (int)((v / w) % h), zi = (int)((v / ((size_t)w * h)) % d);
    const size_t base = v - ((size_t)zi * h + xi) * w - yi;
    uint8_t r = 0;
    for (int dz = -1; dz <= 1; ++dz) {
        const int zz = zi + dz;
        if (zz < 0 || zz >= d) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = xi + dx;
            if (xx < 0 || xx >= h) continue;
            const uint8_t* row = in + base + ((size_t)zz * h + xx) * w;
            r |= row[yi];
            if (yi > 0) r |= row[yi - 1];
            if (yi + 1 < w) r |= row[yi + 1];
        }
    }
    out[v] = r;
}
// 3-kernel stable compaction of three flag arrays at once (blockIdx.y = which); 2048 flags per block
constexpr int ASET_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void aset_count_kernel(const uint8_t* __restrict__ flags, int* __restrict__ blk_counts, size_t V, int nblk)
{
    const uint8_t* f = flags + (size_t)blockIdx.y * V;
    const size_t v0 = (size_t)blockIdx.x * ASET_PER_BLOCK + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c += (v0 + k < V && f[v0 + k]) ? 1 : 0;
    __shared__ int ws[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_counts[blockIdx.y * nblk + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void aset_scan_kernel(int* __restrict__ blk_counts, int* __restrict__ counts, int nblk)
{
    // exclusive scan of blk_counts[which][0..nblk) in place; one block of 1024 threads per list, serial chunks per thread
    int* bc = blk_counts + (size_t)blockIdx.x * nblk;
    __shared__ int part[1024];
    const int per = (nblk + 1023) / 1024, t = threadIdx.x;
    const int i0 = t * per, i1 = min(i0 + per, nblk);
    int s = 0;
    for (int i = i0; i < i1; ++i) s += bc[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = i0; i < i1; ++i) { const int c = bc[i]; bc[i] = run; run += c; }
    if (t == 1023) counts[blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(256) void aset_write_kernel(const uint8_t* __restrict__ flags, const int* __restrict__ blk_counts,
                                                         int* __restrict__ rows, int* __restrict__ map1, size_t V, int nblk)
{
    const int which = blockIdx.y;
    const uint8_t* f = flags + (size_t)which * V;
    int* out = rows + (size_t)which * V;
    const size_t v0 = (size_t)blockIdx.x * ASET_PER_BLOCK + threadIdx.x * 8;
    int c = 0;
    uint8_t fl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { fl[k] = (v0 + k < V) ? f[v0 + k] : 0; c += fl[k] ? 1 : 0; }
    // block exclusive scan of c (256 threads)
    __shared__ int sc[256];
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = blk_counts[which * nblk + blockIdx.x] + sc[threadIdx.x] - c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (v0 + k >= V) break;
        if (fl[k]) {
            out[pos] = (int)(v0 + k);
            if (which == 0 && map1) map1[v0 + k] = pos;
            ++pos;
        } else if (which == 0 && map1) map1[v0 + k] = -1;
    }
}
// level-2 flags: a coarse voxel of the next pyramid level is active when any of its (<= 8) children is
__global__ void aset_parent_kernel(const uint8_t* __restrict__ child, uint8_t* __restrict__ parent, int B, int d, int h, int w, int d2, int h2, int w2)
{
    const size_t V2 = (size_t)B * d2 * h2 * w2;
    const size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (v >= V2) return;
    const int y = (int)(v % w2), x = (int)((v / w2) % h2), z = (int)((v / ((size_t)w2 * h2)) % d2), b = (int)(v / ((size_t)w2 * h2 * d2));
    uint8_t r = 0;
    for (int dz = 0; dz < 2; ++dz) for (int dx = 0; dx < 2; ++dx) for (int dy = 0; dy < 2; ++dy) {
        const int zz = 2 * z + dz, xx = 2 * x + dx, yy = 2 * y + dy;
        if (zz < d && xx < h && yy < w) r |= child[(((size_t)b * d + zz) * h + xx) * w + yy];
    }
    parent[v] = r;
}
extern "C" {
// workspace: 3*V flag bytes + 3*nblk ints (see dreg_active_sets_workspace_bytes).  rows: int32 [3][V] (list k starts at
// rows + k*V); counts: device int32 [3]; map1: int32 [V] or NULL.
size_t dreg_active_sets_workspace_bytes(int B, int d, int h, int w)
{
    const size_t V = (size_t)B * d * h * w;
    const size_t nblk = (V + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK;
    return (3 * V + 255) / 256 * 256 + 3 * nblk * sizeof(int) + 256;
}
int dreg_active_sets(const int64_t* idx, const int* pt_batch, int N, int B, int Zr, int Xr, int Yr, int d, int h, int w,
                     int* rows, int* counts, int* map1, void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t V = (size_t)B * d * h * w;
    if (V == 0 || V > 0x7fffffffull || N < 0) return DREG_EINVAL;
    if (workspace_bytes < dreg_active_sets_workspace_bytes(B, d, h, w)) return DREG_EINVAL;
    const int nblk = (int)((V + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK);
    uint8_t* flags = (uint8_t*)workspace;
    int* blk = (int*)((char*)workspace + (3 * V + 255) / 256 * 256);
    if (hipMemsetAsync(flags, 0, V, st) != hipSuccess) return DREG_ELAUNCH;
    if (N > 0) {
        hipLaunchKernelGGL(aset_mark_kernel, dim3((N + 255) / 256), dim3(256), 0, st, idx, pt_batch, flags, N, d, h, w, Zr, Xr, Yr);
        DREG_LAUNCH_CHECK();
    }
    const unsigned nbv = (unsigned)((V + 255) / 256);
    hipLaunchKernelGGL(aset_dilate_kernel, dim3(nbv), dim3(256), 0, st, flags, flags + V, B, d, h, w);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_dilate_kernel, dim3(nbv), dim3(256), 0, st, flags + V, flags + 2 * V, B, d, h, w);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_count_kernel, dim3(nblk, 3), dim3(256), 0, st, flags, blk, V, nblk);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_scan_kernel, dim3(3), dim3(1024), 0, st, blk, counts, nblk);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_write_kernel, dim3(nblk, 3), dim3(256), 0, st, flags, blk, rows, map1, V, nblk);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Output voxels of a strided convolution over a sparse volume whose receptive field holds an occupied input voxel (everywhere else the
// result of a bias-free convolution is exactly zero): the stem of the feature network (5^3 taps, stride 2, pad 2: 128^3 -> 64^3) computes
// ~5 % of its rows.  idx / pt_batch as in dreg_active_sets (flat fine indices (x * Yr + y) * Zr + z); rows: ascending int32 flat indices
// into [B, d, h, w] (capacity V), count: device int32.  workspace: dreg_conv_rows_workspace_bytes.
size_t dreg_conv_rows_workspace_bytes(int B, int d, int h, int w)
{
    const size_t V = (size_t)B * d * h * w;
    const size_t nblk = (V + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK;
    return (V + 255) / 256 * 256 + nblk * sizeof(int) + 256;
}
}  // extern "C"
__global__ void conv_rows_mark_kernel(const int64_t* __restrict__ idx, const int* __restrict__ pt_batch, uint8_t* __restrict__ f,
                                      int N, int d, int h, int w, int Zr, int Xr, int Yr, int ksz, int stride, int pad)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t fl = idx[n];
    const int z = (int)(fl % Zr), y = (int)((fl / Zr) % Yr), x = (int)(fl / ((int64_t)Zr * Yr));
    const int b = pt_batch[n];
    // outputs o with o * stride - pad <= i <= o * stride - pad + ksz - 1
    auto lo = [&](int i) { const int t = i + pad - ksz + 1; return t <= 0 ? 0 : (t + stride - 1) / stride; };
    auto hi = [&](int i, int ext) { const int t = (i + pad) / stride; return t < ext ? t : ext - 1; };
    const int z0 = lo(z), z1 = hi(z, d), x0 = lo(x), x1 = hi(x, h), y0 = lo(y), y1 = hi(y, w);
    for (int zo = z0; zo <= z1; ++zo)
        for (int xo = x0; xo <= x1; ++xo)
            for (int yo = y0; yo <= y1; ++yo) f[(((size_t)b * d + zo) * h + xo) * w + yo] = 1;
}
extern "C" {
int dreg_conv_rows(const int64_t* idx, const int* pt_batch, int N, int B, int Zr, int Xr, int Yr, int d, int h, int w, int ksz, int stride, int pad,
                   int* rows, int* count, void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t V = (size_t)B * d * h * w;
    if (V == 0 || V > 0x7fffffffull || N < 0 || ksz < 1 || stride < 1 || pad < 0) return DREG_EINVAL;
    if (workspace_bytes < dreg_conv_rows_workspace_bytes(B, d, h, w)) return DREG_EINVAL;
    const int nblk = (int)((V + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK);
    uint8_t* flags = (uint8_t*)workspace;
    int* blk = (int*)((char*)workspace + (V + 255) / 256 * 256);
    if (hipMemsetAsync(flags, 0, V, st) != hipSuccess) return DREG_ELAUNCH;
    if (N > 0) hipLaunchKernelGGL(conv_rows_mark_kernel, dim3((N + 255) / 256), dim3(256), 0, st, idx, pt_batch, flags, N, d, h, w, Zr, Xr, Yr, ksz, stride, pad);
    hipLaunchKernelGGL(aset_count_kernel, dim3(nblk, 1), dim3(256), 0, st, flags, blk, V, nblk);
    hipLaunchKernelGGL(aset_scan_kernel, dim3(1), dim3(1024), 0, st, blk, count, nblk);
    hipLaunchKernelGGL(aset_write_kernel, dim3(nblk, 1), dim3(256), 0, st, flags, blk, rows, (int*)nullptr, V, nblk);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Second pyramid level of the active sets: child_flags = the S2 flags of dreg_active_sets (bytes [V, 2V) of ITS workspace, on
// [B,d,h,w]); A = parents of S2 on [B,d2,h2,w2] (where the next-coarser FPN map P2 is consumed by the nearest-x2 upsample-add),
// A2 = A dilated by 3^3 (where its lateral sum is consumed).  rows2 int32 [2][V2], counts2 device int32 [2].
size_t dreg_active_sets_level2_workspace_bytes(int B, int d2, int h2, int w2)
{
    const size_t V2 = (size_t)B * d2 * h2 * w2;
    const size_t nblk = (V2 + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK;
    return (2 * V2 + 255) / 256 * 256 + 2 * nblk * sizeof(int) + 256;
}
int dreg_active_sets_level2(const uint8_t* child_flags, int B, int d, int h, int w, int d2, int h2, int w2, int* rows2, int* counts2,
                            void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t V2 = (size_t)B * d2 * h2 * w2;
    if (V2 == 0 || V2 > 0x7fffffffull || d2 != (d + 1) / 2 || h2 != (h + 1) / 2 || w2 != (w + 1) / 2) return DREG_EINVAL;
    if (workspace_bytes < dreg_active_sets_level2_workspace_bytes(B, d2, h2, w2)) return DREG_EINVAL;
    const int nblk = (int)((V2 + ASET_PER_BLOCK - 1) / ASET_PER_BLOCK);
    uint8_t* flags = (uint8_t*)workspace;
    int* blk = (int*)((char*)workspace + (2 * V2 + 255) / 256 * 256);
    const unsigned nbv = (unsigned)((V2 + 255) / 256);
    hipLaunchKernelGGL(aset_parent_kernel, dim3(nbv), dim3(256), 0, st, child_flags, flags, B, d, h, w, d2, h2, w2);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_dilate_kernel, dim3(nbv), dim3(256), 0, st, flags, flags + V2, B, d2, h2, w2);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_count_kernel, dim3(nblk, 2), dim3(256), 0, st, flags, blk, V2, nblk);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_scan_kernel, dim3(2), dim3(1024), 0, st, blk, counts2, nblk);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aset_write_kernel, dim3(nblk, 2), dim3(256), 0, st, flags, blk, rows2, (int*)nullptr, V2, nblk);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
}

// Row-wise / small kernels of the point-set half (gfx950): LayerNorm (+position embedding) forward/backward,
// sine position embedding, overlap head (GEMV + sigmoid), ReLU backward, weighted Kabsch (3x3 SVD on device),
// voxel-average downsampling (sorted keys + segmented mean) and the flat AdamW / gradient-norm kernels.
//
// Reference call sites: conerf/register/transformer.py:238-293 (LayerNorm, with_pos_embed),
// conerf/register/position_embedding.py:30-53, conerf/register/nerf_regtr.py:384-387 (overlap),
// conerf/register/se3.py:89-140 (Kabsch), conerf/register/grid_downsample.py:6-44 (MinkowskiEngine average),
// train_nerf_regtr.py:96-102,232-239 (clip_grad_norm_ + AdamW).
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

// ------------------------------------------------------------------------------------------------ LayerNorm (C = 256)
// y = LN(x) * g + b [+ pe]; one wave per row, 4 channels per lane.  Saves (mean, rstd) per row.
// four consecutive elements as ONE load / store (8 bytes of bf16, 16 of fp32)
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u); v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) { *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3])); }
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                            const float* __restrict__ pe, TO* __restrict__ y, float* __restrict__ stats, int N, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / 256.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4), bb = *reinterpret_cast<const float4*>(b + lane * 4);
    float o[4] = {d0 * rstd * gg.x + bb.x, d1 * rstd * gg.y + bb.y, d2 * rstd * gg.z + bb.z, d3 * rstd * gg.w + bb.w};
    if (pe) { const float4 p = *reinterpret_cast<const float4*>(pe + (size_t)row * 256 + lane * 4); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
    st4(y + (size_t)row * 256 + lane * 4, o);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// The final norm of the encoder is applied to the six layer outputs twice (nerf_regtr.py:170-206: `cond` = LN(x) in fp32 for the heads
// and the losses, LN(x) + pe in the compute dtype for the decoder's projections): both from ONE read of x.  pe has pe_rows rows and is
// indexed by row % pe_rows (the six layers share the key points' embedding).  Each output is bit-identical to layernorm_fwd_kernel's.
__global__ __launch_bounds__(256) void layernorm_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                             const float* __restrict__ pe, int pe_rows, float* __restrict__ y32, bf16_t* __restrict__ y16,
                                                             float* __restrict__ stats, int N, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / 256.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4), bb = *reinterpret_cast<const float4*>(b + lane * 4);
    float o[4] = {d0 * rstd * gg.x + bb.x, d1 * rstd * gg.y + bb.y, d2 * rstd * gg.z + bb.z, d3 * rstd * gg.w + bb.w};
    if (y32) st4(y32 + (size_t)row * 256 + lane * 4, o);
    if (y16) {
        if (pe) { const float4 p = *reinterpret_cast<const float4*>(pe + (size_t)(row % pe_rows) * 256 + lane * 4); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
        st4(y16 + (size_t)row * 256 + lane * 4, o);
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));  partial dgamma/dbeta per block of rows
template <typename TG>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const TG* __restrict__ dy, const float* __restrict__ g,
                                                            const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ part,
                                                            int N, int rows_per_block, const float* dx_add, const float* dx_add2 = nullptr,
                                                            bf16_t* __restrict__ dx_bf = nullptr, const bf16_t* __restrict__ dy2 = nullptr)
{
    // dx = (LayerNorm-backward(dy [+ dy2]) + dx_add) + dx_add2 (both optional, either may alias dx); dx_bf (optional): the same values
    // rounded to bf16 — the operand of the GEMMs that consume this gradient (no separate cast launch); dy2 (optional, bf16): a second
    // upstream gradient of the SAME LayerNorm application (the backward is linear in dy), added to dy in fp32 before anything else
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
    float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    const int r0 = blockIdx.x * rows_per_block;
    // a wave's rows four at a time: every load of the four rows is issued before the first row's reductions (the rows were a chain of
    // load -> two wave reductions -> store, one row's loads in flight); per-lane dgamma / dbeta sums stay in row order
    const int rend = min(r0 + rows_per_block, N);
    const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
    for (int rb = r0 + wave; rb < rend; rb += 16) {
        constexpr int U = 4;
        float4 v[U], old[U], old2[U];
        float d[U][4], mean[U], rstd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rb + 4 * u;
            if (row < rend) {
                v[u] = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
                ld4(dy + (size_t)row * 256 + lane * 4, d[u]);
                if (dy2) { float e2[4]; ld4(dy2 + (size_t)row * 256 + lane * 4, e2); d[u][0] += e2[0]; d[u][1] += e2[1]; d[u][2] += e2[2]; d[u][3] += e2[3]; }
                mean[u] = stats[2 * row]; rstd[u] = stats[2 * row + 1];
                if (dx_add) old[u] = *reinterpret_cast<const float4*>(dx_add + (size_t)row * 256 + lane * 4);
                if (dx_add2) old2[u] = *reinterpret_cast<const float4*>(dx_add2 + (size_t)row * 256 + lane * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rb + 4 * u;
            if (row < rend) {
                const float xh[4] = {(v[u].x - mean[u]) * rstd[u], (v[u].y - mean[u]) * rstd[u], (v[u].z - mean[u]) * rstd[u], (v[u].w - mean[u]) * rstd[u]};
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { s1 += d[u][i] * gv[i]; s2 += d[u][i] * gv[i] * xh[i]; dg[i] += d[u][i] * xh[i]; db[i] += d[u][i]; }
                s1 = wave_sum(s1) * (1.f / 256.f); s2 = wave_sum(s2) * (1.f / 256.f);
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = rstd[u] * (d[u][i] * gv[i] - s1 - xh[i] * s2);
                if (dx_add) { o[0] += old[u].x; o[1] += old[u].y; o[2] += old[u].z; o[3] += old[u].w; }
                if (dx_add2) { o[0] += old2[u].x; o[1] += old2[u].y; o[2] += old2[u].z; o[3] += old2[u].w; }
                *reinterpret_cast<float4*>(dx + (size_t)row * 256 + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
                if (dx_bf) st4(dx_bf + (size_t)row * 256 + lane * 4, o);
            }
        }
    }
    __shared__ float red[4][512];
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[wave][lane * 4 + i] = dg[i]; red[wave][256 + lane * 4 + i] = db[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) part[(size_t)blockIdx.x * 512 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ void layernorm_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ dg, float* __restrict__ db, int nblk, int accumulate)
{
    // one wave per column (lanes stride over the block partials, fixed order -> deterministic)
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= 512) return;
    double s = 0.0;
#pragma unroll 8                                     // ~10 partials per lane: all in flight, added in order
    for (int k = lane; k < nblk; k += 64) s += part[(size_t)k * 512 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        float* dst = c < 256 ? dg + c : db + (c - 256);
        *dst = accumulate ? *dst + (float)s : (float)s;
    }
}

// The dgamma / dbeta sums of ALL LayerNorm applications of a backward pass in one launch (the point-set executor keeps every
// application's block partials until its pass ends): 128 workgroups per record, the same column sums in the same order.
struct LnFinalDesc { const float* part; float* dg; float* db; int nblk, accumulate; };
static_assert(sizeof(LnFinalDesc) == 32, "LayerNorm final record: 32 bytes");
__global__ void layernorm_bwd_final_batched_kernel(const LnFinalDesc* __restrict__ descs)
{
    const LnFinalDesc d = descs[blockIdx.x >> 7];
    const int c = (blockIdx.x & 127) * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= 512) return;
    double s = 0.0;
#pragma unroll 8
    for (int k = lane; k < d.nblk; k += 64) s += d.part[(size_t)k * 512 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        float* dst = c < 256 ? d.dg + c : d.db + (c - 256);
        *dst = d.accumulate ? *dst + (float)s : (float)s;
    }
}

// ------------------------------------------------------------------------------------------------ sine position embedding
// 84 features per coordinate: feature i of coordinate c = (i even ? sin : cos)(x_c * 2*pi*scale / T^(2*floor(i/2)/84)); 252..255 = 0
__global__ void posenc_sine_kernel(const float* __restrict__ xyz, float* __restrict__ pe, int N, float scale2pi, float log_temp)
{
    const size_t total = (size_t)N * 256;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 255), n = (int)(i >> 8);
        float v = 0.f;
        if (c < 252) {
            const int coord = c / 84, f = c - coord * 84;
            const float dim_t = powf(expf(log_temp), (float)(2 * (f >> 1)) / 84.f);
            const float arg = (xyz[(size_t)n * 3 + coord] * scale2pi) / dim_t;
            v = (f & 1) ? cosf(arg) : sinf(arg);
        }
        pe[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ overlap head
// s[n] = sigmoid(f[n,:] . w + b)   (one wave per row);  backward: dlogit = gy*s*(1-s); df = dlogit*w; dw/db partials
__global__ __launch_bounds__(256) void overlap_fwd_kernel(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ s, int N)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4 v = *reinterpret_cast<const float4*>(f + (size_t)row * 256 + lane * 4);
    const float4 ww = *reinterpret_cast<const float4*>(w + lane * 4);
    const float z = wave_sum(v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w) + b[0];
    if (lane == 0) s[row] = 1.f / (1.f + expf(-z));
}
__global__ __launch_bounds__(256) void overlap_bwd_kernel(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ s,
                                                          const float* __restrict__ gy, float* __restrict__ df, float* __restrict__ part,
                                                          int N, int rows_per_block, const float* df_add = nullptr)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 ww = *reinterpret_cast<const float4*>(w + lane * 4);
    float dw[4] = {0, 0, 0, 0}, dbias = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    for (int row = r0 + wave; row < min(r0 + rows_per_block, N); row += 4) {
        const float sv = s[row], dl = gy[row] * sv * (1.f - sv);
        const float4 v = *reinterpret_cast<const float4*>(f + (size_t)row * 256 + lane * 4);
        float4 o = make_float4(dl * ww.x, dl * ww.y, dl * ww.z, dl * ww.w);
        if (df_add) {           // the other gradient of f (f feeds the losses directly as well; may be df itself): df = df_add + dlogit * w
            const float4 p = *reinterpret_cast<const float4*>(df_add + (size_t)row * 256 + lane * 4);
            o.x = p.x + o.x; o.y = p.y + o.y; o.z = p.z + o.z; o.w = p.w + o.w;
        }
        *reinterpret_cast<float4*>(df + (size_t)row * 256 + lane * 4) = o;
        dw[0] += dl * v.x; dw[1] += dl * v.y; dw[2] += dl * v.z; dw[3] += dl * v.w;
        dbias += dl;
    }
    __shared__ float red[4][257];
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = dw[i];
    if (lane == 0) red[wave][256] = dbias;
    __syncthreads();
    for (int c = threadIdx.x; c < 257; c += 256) part[(size_t)blockIdx.x * 257 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ void overlap_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nblk, int accumulate = 0)
{
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= 257) return;
    double s = 0.0;
    for (int k = lane; k < nblk; k += 64) s += part[(size_t)k * 257 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) { float* dst = c < 256 ? dw + c : db; *dst = accumulate ? *dst + (float)s : (float)s; }
}

// ------------------------------------------------------------------------------------------------ elementwise helpers
// g_out = (y > 0) ? g : 0, cast to TO
template <typename TY, typename TG, typename TO>
__global__ void relu_bwd_kernel(const TY* __restrict__ y, const TG* __restrict__ g, TO* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        Elem<TO>::st(out + i, Elem<TY>::ld(y + i) > 0.f ? Elem<TG>::ld(g + i) : 0.f);
}

// ------------------------------------------------------------------------------------------------ weighted Kabsch
// One workgroup per problem: T with T*a = b minimising sum w |R a + t - b|^2.  a,b [P,N,3], w [P,N] -> out [P,3,4].
// Covariance accumulated in fp64; 3x3 SVD by two-sided Jacobi on H^T H (fp64); R = V U^T with the reflection fix.
__device__ void jacobi_eig3(double A[3][3], double V[3][3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
}
// row accessors of one Kabsch problem: dense [N,3] arrays, or the (src | tgt) rows of a pair inside the shared row space
struct KabschDense {
    const float* A; const float* Bp; const float* Wp;
    __device__ __forceinline__ float a(int i, int c) const { return A[i * 3 + c]; }
    __device__ __forceinline__ float b(int i, int c) const { return Bp[i * 3 + c]; }
    __device__ __forceinline__ float w(int i) const { return Wp[i]; }
};
struct KabschPair {   // a = cat[src_xyz, tgt_corr], b = cat[src_corr, tgt_xyz], w = cat[src_ov, tgt_ov]  (nerf_regtr.py:226-236)
    const float* xyz; const float* corr; const float* ov; int s0, ns, t0;
    __device__ __forceinline__ float a(int i, int c) const { return i < ns ? xyz[(size_t)(s0 + i) * 3 + c] : corr[(size_t)(t0 + i - ns) * 3 + c]; }
    __device__ __forceinline__ float b(int i, int c) const { return i < ns ? corr[(size_t)(s0 + i) * 3 + c] : xyz[(size_t)(t0 + i - ns) * 3 + c]; }
    __device__ __forceinline__ float w(int i) const { return i < ns ? ov[s0 + i] : ov[t0 + i - ns]; }
};
template <typename Acc> __device__ void kabsch_body(const Acc& X, int N, float* __restrict__ o, float eps);
__global__ __launch_bounds__(256) void kabsch_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                                                     float* __restrict__ out, int N, float eps)
{
    const int p = blockIdx.x;
    KabschDense X{a + (size_t)p * N * 3, b + (size_t)p * N * 3, w + (size_t)p * N};
    kabsch_body(X, N, out + (size_t)p * 12, eps);
}
// one block per (pair, layer): probs int32 [P][4] = (s0, ns, t0, nt) rows of the shared row space; corr [L,R,3], ov [L,R]
__global__ __launch_bounds__(256) void kabsch_pairs_kernel(const float* __restrict__ xyz, const float* __restrict__ corr, const float* __restrict__ ov,
                                                           const int* __restrict__ probs, float* __restrict__ out, int L, int R, float eps)
{
    const int p = blockIdx.x / L, l = blockIdx.x % L;
    const int* pr = probs + p * 4;
    KabschPair X{xyz, corr + (size_t)l * R * 3, ov + (size_t)l * R, pr[0], pr[1], pr[2]};
    kabsch_body(X, pr[1] + pr[3], out + (size_t)blockIdx.x * 12, eps);
}
template <typename Acc> __device__ void kabsch_body(const Acc& X, int N, float* __restrict__ o, float eps)
{
    const int t = threadIdx.x;
    __shared__ double red[9][256];
    __shared__ double tot[9];
    // Sums of NV quantities over the 256 threads with TWO barriers: the additions are exactly those of a halving tree over red[0..255]
    // (s = 128, 64 across the waves: (a + c) + (b + d); s = 32 .. 1 inside wave 0 as shuffles), which took nine barriers per quantity.
    auto block_sums = [&](double* v, int NV) {
        for (int q = 0; q < NV; ++q) red[q][t] = v[q];
        __syncthreads();
        if (t < 64) {
            for (int q = 0; q < NV; ++q) {
                double a = (red[q][t] + red[q][t + 128]) + (red[q][t + 64] + red[q][t + 192]);
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
                if (t == 0) tot[q] = a;
            }
        }
        __syncthreads();
        for (int q = 0; q < NV; ++q) v[q] = tot[q];
        __syncthreads();
    };
    double sw = 0.0;
#pragma unroll 4
    for (int i = t; i < N; i += 256) sw += X.w(i);
    block_sums(&sw, 1);
    const double norm = fmax(sw, (double)eps);
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
#pragma unroll 2
    for (int i = t; i < N; i += 256) {
        const double wn = X.w(i) / norm;
        for (int c = 0; c < 3; ++c) { ca[c] += wn * X.a(i, c); cb[c] += wn * X.b(i, c); }
    }
    {
        double v6[6] = {ca[0], ca[1], ca[2], cb[0], cb[1], cb[2]};
        block_sums(v6, 6);
        for (int c = 0; c < 3; ++c) { ca[c] = v6[c]; cb[c] = v6[3 + c]; }
    }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 2
    for (int i = t; i < N; i += 256) {
        const double wn = X.w(i) / norm;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] += (X.a(i, r) - ca[r]) * (X.b(i, c) - cb[c]) * wn;
    }
    block_sums(&H[0][0], 9);
    if (t == 0) {
        // H = U S V^T.  Eigen-decompose H^T H = V S^2 V^T, sort descending, U = H V S^-1 (last column by cross product).
        double HtH[3][3], V[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += H[k][i] * H[k][j]; HtH[i][j] = s; }
        jacobi_eig3(HtH, V);
        double ev[3] = {HtH[0][0], HtH[1][1], HtH[2][2]};
        int idx[3] = {0, 1, 2};
        for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (ev[idx[j]] > ev[idx[i]]) { int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; }
        double Vs[3][3], U[3][3];
        for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) Vs[k][j] = V[k][idx[j]];
        for (int j = 0; j < 2; ++j) {
            double n2 = 0;
            for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += H[i][k] * Vs[k][j]; U[i][j] = s; n2 += s * s; }
            const double n = sqrt(n2);
            for (int i = 0; i < 3; ++i) U[i][j] = n > 0 ? U[i][j] / n : (i == j);
        }
        // third singular vectors from cross products so that det(U), det(V) are consistent with the two computed columns
        double u2[3] = {U[1][0] * U[2][1] - U[2][0] * U[1][1], U[2][0] * U[0][1] - U[0][0] * U[2][1], U[0][0] * U[1][1] - U[1][0] * U[0][1]};
        double v2[3] = {Vs[1][0] * Vs[2][1] - Vs[2][0] * Vs[1][1], Vs[2][0] * Vs[0][1] - Vs[0][0] * Vs[2][1], Vs[0][0] * Vs[1][1] - Vs[1][0] * Vs[0][1]};
        // R = V diag(1,1,d) U^T with d = +1 here gives det(R) = +1 by construction (both bases right-handed),
        // which is exactly the reference's "flip the last column of V when det(V U^T) < 0" rule (se3.py:128-134).
        double R[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + v2[i] * u2[j];
        for (int i = 0; i < 3; ++i) {
            double tr = cb[i];
            for (int j = 0; j < 3; ++j) { o[i * 4 + j] = (float)R[i][j]; tr -= R[i][j] * ca[j]; }
            o[i * 4 + 3] = (float)tr;
        }
    }
}

// ------------------------------------------------------------------------------------------------ voxel-average downsample
// frozen (may be null): uint8 per batch id; the rows of a frozen batch keep their identity — key = (batch, row index), every row a cell of its own, so the
// round passes them through unchanged (a mean over one row) and in order.  It lets ONE launch set serve all pairs of a step although the reference's stopping
// rule is per pair (grid_downsample.py:83-94: a pair whose point count has dropped to <= 3,000 takes no further round).
__global__ void voxel_keys_kernel(const float* __restrict__ pts, const int* __restrict__ pt_batch, uint64_t* __restrict__ keys,
                                  uint32_t* __restrict__ vals, int N, float dl, int* __restrict__ err, const uint8_t* __restrict__ frozen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint64_t key = (uint64_t)(uint32_t)pt_batch[i] << 48;
    if (frozen != nullptr && frozen[pt_batch[i]]) {
        keys[i] = key | (uint64_t)(uint32_t)i; vals[i] = (uint32_t)i;
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float q = floorf(pts[(size_t)i * 3 + c] / dl);   // true division, as the CPU oracle
        const int ci = (int)q + 32768;
        if (ci < 0 || ci > 65535) atomicExch(err, 1);
        key |= (uint64_t)(uint32_t)(ci & 0xffff) << (32 - 16 * c);
    }
    keys[i] = key; vals[i] = (uint32_t)i;
}
__global__ void segment_heads_kernel(const uint64_t* __restrict__ keys, uint32_t* __restrict__ head, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// seg[i] = inclusive_scan(head)[i] - 1 ; starts[seg] = i for heads ; counts per batch
__global__ void segment_starts_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head, const uint32_t* __restrict__ scan,
                                      uint32_t* __restrict__ starts, int* __restrict__ batch_counts, int* __restrict__ nseg, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_head = i < N && head[i];
    const int b = is_head ? (int)(keys[i] >> 48) : -1;
    if (is_head) starts[scan[i] - 1] = (uint32_t)i;
    if (i == N - 1) { *nseg = (int)scan[i]; starts[scan[i]] = (uint32_t)N; }
    // one atomic per (wave, batch id) instead of one per segment head: keys are sorted, a wave sees very few distinct ids
    unsigned long long todo = __ballot(is_head);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b0 = __shfl(b, leader, 64);
        const unsigned long long same = __ballot(is_head && b == b0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(batch_counts + b0, (int)__popcll(same));
        todo &= ~same;
    }
}
// ------------------------------------------------------------------------------------------------ own sort + segments (one workgroup)
// The voxel keys of one downsample round are a few 10^4 64-bit values (both point sets of a pair): keys, a stable LSD radix sort,
// segment heads, their scan, the segment starts and the per-batch counts in ONE launch of one 1,024-thread workgroup — it replaces
// voxel_keys + rocPRIM radix_sort_pairs (7-9 launches) + segment_heads + rocPRIM inclusive_scan (2) + segment_starts, ~15 launches per
// round and 73 library launches per training step (grid_downsample.py:24-36 is one MinkowskiEngine call in the reference).
// 4-bit digits; digits in which all keys agree are skipped (coordinates of a NeRF block span a few dozen cells: typically 6-7 of the 16
// passes run).  Stable: equal keys keep ascending original indices, exactly the order rocPRIM's stable sort gives (bit-identical
// downstream results).
constexpr int VS_THREADS = 1024, VS_WAVES = VS_THREADS / 64, VS_MAX_N = 131072;
// barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store (s_waitcnt vmcnt(0)), which put the
// latency of a round's scattered stores in front of every one of its three barriers (940 us per sort on one CU)
#define VS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(1024) void voxel_sort_segments_kernel(const float* __restrict__ pts, const int* __restrict__ pt_batch, float dl,
                                                                   uint64_t* keysA, uint64_t* keysB, uint32_t* valsA, uint32_t* valsB,
                                                                   uint32_t* __restrict__ starts, int* __restrict__ batch_counts, int* __restrict__ nseg,
                                                                   int* __restrict__ err, int N, int nbatch)
{
    // Elements are visited 1,024 at a time (index = 1024 j + thread: coalesced loads).  Stable rank of an element inside a round = elements
    // of the same digit in lower lanes of its wave (ballot) + in lower waves (16 x 16 table in LDS); a digit's running base is carried
    // from round to round.
    constexpr int E = 4;                            // elements per thread and round: index = 4096 j + 1024 e + thread
    __shared__ uint32_t wcnt[E][VS_WAVES][16];      // per (sub-block, wave, digit) count of the current round
    __shared__ uint32_t wpre[E][VS_WAVES][16];      // exclusive prefix over the waves of a sub-block
    __shared__ uint32_t etot[E][16];                // per (sub-block, digit) count of the round
    __shared__ uint32_t hsum[2][16];                // per-digit count of the whole array: this pass / the next one (built while this one scatters)
    __shared__ unsigned long long s_or;
    __shared__ uint64_t s_k0;
    __shared__ int s_err;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (t == 0) { s_or = 0ull; s_err = 0; }
    if (t < 32) hsum[t >> 4][t & 15] = 0;
    for (int b = t; b < nbatch; b += VS_THREADS) batch_counts[b] = 0;
    __syncthreads();
    // ---- keys (voxel_keys_kernel's arithmetic), original indices, and the bits in which the keys differ
    unsigned long long diff = 0ull;
    uint64_t key0 = 0;
    for (int i = t; i < N; i += VS_THREADS) {
        uint64_t key = (uint64_t)(uint32_t)(pt_batch[i] & 0xffff) << 48;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = floorf(pts[(size_t)i * 3 + c] / dl);
            const int ci = (int)q + 32768;
            if (ci < 0 || ci > 65535) s_err = 1;
            key |= (uint64_t)(uint32_t)(ci & 0xffff) << (32 - 16 * c);
        }
        if (i == t) key0 = key;
        diff |= key ^ key0;
        keysA[i] = key; valsA[i] = (uint32_t)i;
    }
    if (t == 0) s_k0 = key0;
    __syncthreads();
    if (t < N) diff |= key0 ^ s_k0;               // a thread's diff is relative to its own first key
    if (diff) atomicOr(&s_or, diff);
    __syncthreads();
    const unsigned long long mask = s_or;
    if (t == 0) *err = s_err;
    const int rounds = (N + E * VS_THREADS - 1) / (E * VS_THREADS);
    uint64_t* ksrc = keysA; uint64_t* kdst = keysB;
    uint32_t* vsrc = valsA; uint32_t* vdst = valsB;
    // digits that take part, in order; the histogram of the first one needs a walk of its own
    int digs[16], nd = 0;
    for (int d = 0; d < 16; ++d) if ((mask >> (4 * d)) & 15ull) digs[nd++] = d;
    auto add_hist = [&](uint32_t (&cnt)[16], int which) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint32_t v = cnt[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && v) atomicAdd(&hsum[which][k], v);
        }
    };
    if (nd > 0) {
        const int sh = 4 * digs[0];
        uint32_t cnt[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) cnt[k] = 0;
        for (int i = t; i < N; i += VS_THREADS) {
            const int dg = (int)((ksrc[i] >> sh) & 15ull);
#pragma unroll
            for (int k = 0; k < 16; ++k) cnt[k] += (dg == k) ? 1u : 0u;
        }
        add_hist(cnt, 0);
    }
    __syncthreads();
    for (int pi = 0; pi < nd; ++pi) {
        const int sh = 4 * digs[pi];
        const int cur = pi & 1, nxt = cur ^ 1;
        const bool has_next = pi + 1 < nd;
        const int sh_n = has_next ? 4 * digs[pi + 1] : 0;
        uint32_t base[16], ncnt[16];
        {
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { base[k] = run; run += hsum[cur][k]; ncnt[k] = 0; }
        }
        for (int j = 0; j < rounds; ++j) {
            uint64_t key[E]; uint32_t v[E]; int dg[E]; uint32_t myrank[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = (j * E + e) * VS_THREADS + t;
                const bool live = i < N;
                key[e] = 0; v[e] = 0;
                if (live) { key[e] = ksrc[i]; v[e] = vsrc[i]; }
                dg[e] = live ? (int)((key[e] >> sh) & 15ull) : -1;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                myrank[e] = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const unsigned long long m = __ballot(dg[e] == k);
                    if (dg[e] == k) myrank[e] = (uint32_t)__popcll(m & lt);
                    if (lane == k) wcnt[e][wave][k] = (uint32_t)__popcll(m);
                }
                if (has_next && dg[e] >= 0) {
                    const int dn = (int)((key[e] >> sh_n) & 15ull);
#pragma unroll
                    for (int k = 0; k < 16; ++k) ncnt[k] += (dn == k) ? 1u : 0u;
                }
            }
            VS_LDS_BARRIER();
            {                                         // thread (e, w, k): elements of digit k in the waves below w of sub-block e
                const int e = t >> 8, w = (t >> 4) & 15, k = t & 15;
                uint32_t p = 0;
                for (int w2 = 0; w2 < w; ++w2) p += wcnt[e][w2][k];
                wpre[e][w][k] = p;
                if (w == VS_WAVES - 1) etot[e][k] = p + wcnt[e][w][k];
            }
            VS_LDS_BARRIER();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (dg[e] >= 0) {
                    uint32_t b0 = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) if (dg[e] == k) b0 = base[k];
                    for (int e2 = 0; e2 < e; ++e2) b0 += etot[e2][dg[e]];
                    const uint32_t pos = b0 + wpre[e][wave][dg[e]] + myrank[e];
                    kdst[pos] = key[e]; vdst[pos] = v[e];
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) base[k] += etot[0][k] + etot[1][k] + etot[2][k] + etot[3][k];
            VS_LDS_BARRIER();                         // wcnt / wpre / etot are rewritten by the next round (the scattered stores stay in flight)
        }
        if (t < 16) hsum[cur][t] = 0;                 // this pass's histogram becomes the buffer of the pass after next
        if (has_next) add_hist(ncnt, nxt);
        __syncthreads();
        { uint64_t* a = ksrc; ksrc = kdst; kdst = a; uint32_t* b = vsrc; vsrc = vdst; vdst = b; }
    }
    // the result must end in (keysB, valsB): an even number of passes left it in A
    if (ksrc == keysA) {
        for (int i = t; i < N; i += VS_THREADS) { keysB[i] = keysA[i]; valsB[i] = valsA[i]; }
        __syncthreads();
    }
    // ---- segments of equal keys: heads, their ranks (same round / ballot scheme with one class), starts, per-batch segment counts
    uint32_t hbase = 0;
    for (int j = 0; j < (N + VS_THREADS - 1) / VS_THREADS; ++j) {
        const int i = j * VS_THREADS + t;
        const bool live = i < N;
        const uint64_t key = live ? keysB[i] : 0;
        const bool is_head = live && (i == 0 || key != keysB[i - 1]);
        const unsigned long long m = __ballot(is_head);
        if (lane == 0) wcnt[0][wave][0] = (uint32_t)__popcll(m);
        __syncthreads();
        if (t < VS_WAVES) {
            uint32_t p = 0;
            for (int w2 = 0; w2 < t; ++w2) p += wcnt[0][w2][0];
            wpre[0][t][0] = p;
            if (t == VS_WAVES - 1) etot[0][0] = p + wcnt[0][t][0];
        }
        __syncthreads();
        if (is_head) starts[hbase + wpre[0][wave][0] + (uint32_t)__popcll(m & lt)] = (uint32_t)i;
        // one atomic per (wave, batch id): keys are sorted, a wave sees very few distinct ids
        const int b = is_head ? (int)(key >> 48) : -1;
        unsigned long long todo = m;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int b0 = __shfl(b, leader, 64);
            const unsigned long long same = __ballot(is_head && b == b0);
            if (lane == leader) atomicAdd(batch_counts + b0, (int)__popcll(same));
            todo &= ~same;
        }
        hbase += etot[0][0];
        __syncthreads();
    }
    if (t == 0) { starts[hbase] = (uint32_t)N; *nseg = (int)hbase; }
}

// one wave per output row: mean over the segment's members (ascending original index = stable sort order)
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ pts, const float* __restrict__ feats, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ starts, const int* __restrict__ nseg,
                                                           float* __restrict__ out_pts, float* __restrict__ out_feats, uint32_t* __restrict__ inv_seg,
                                                           float* __restrict__ inv_cnt, int C)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= *nseg) return;
    const uint32_t s0 = starts[seg], s1 = starts[seg + 1];
    const float inv = 1.f / (float)(s1 - s0);
    if (feats != nullptr)
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 4
            for (uint32_t j = s0; j < s1; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(feats + (size_t)order[j] * C + c0);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(out_feats + (size_t)seg * C + c0) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        }
    if (pts == nullptr) return;
    if (lane < 3) {
        float acc = 0.f;
        for (uint32_t j = s0; j < s1; ++j) acc += pts[(size_t)order[j] * 3 + lane];
        out_pts[(size_t)seg * 3 + lane] = acc * inv;
    }
    for (uint32_t j = s0 + lane; j < s1; j += 64) { inv_seg[order[j]] = (uint32_t)seg; inv_cnt[order[j]] = inv; }
}
// backward: dfeat_in[i] = dfeat_out[seg(i)] / count
__global__ void segment_mean_bwd_kernel(const float* __restrict__ gout, const uint32_t* __restrict__ inv_seg, const float* __restrict__ inv_cnt,
                                        float* __restrict__ gin, int N, int C)
{
    const size_t total = (size_t)N * (C / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4)), n = (int)(i / (C / 4));
        const float s = inv_cnt[n];
        const float4 g = *reinterpret_cast<const float4*>(gout + (size_t)inv_seg[n] * C + c4 * 4);
        *reinterpret_cast<float4*>(gin + (size_t)n * C + c4 * 4) = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
    }
}

// ------------------------------------------------------------------------------------------------ optimizer
// partial sum of squares of a flat fp32 gradient buffer
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ part, size_t n)
{
    float s = 0.f;
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) { const float4 v = *reinterpret_cast<const float4*>(g + i); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        else for (size_t k = i; k < n; ++k) s += g[k] * g[k];
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, float* __restrict__ norm_out, int nblk)
{
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) s += part[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) norm_out[0] = (float)sqrt(s);
}
// torch.optim.AdamW step on flat buffers with clip_grad_norm_ folded in:
//   g *= min(1, max_norm / (norm + 1e-6));  p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ norm,
                             size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float max_norm)
{
    float clip = 1.f;
    if (max_norm > 0.f) { clip = max_norm / (norm[0] + 1e-6f); clip = clip < 1.f ? clip : 1.f; }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        g[i] = gi;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static inline int nblocks(size_t total, int per = 256, int cap = 4096) {
    size_t b = (total + per - 1) / per;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

extern "C" {

int dreg_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* pe, void* y, float* stats,
                       int N, int C, float eps, int out_dtype, void* stream)
{
    if (C != 256) return DREG_EINVAL;
    if (N == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == 0) hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, st, x, gamma, beta, pe, (bf16_t*)y, stats, N, eps);
    else hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, st, x, gamma, beta, pe, (float*)y, stats, N, eps);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_layernorm_bwd_add(const float* x, const void* dy, const float* gamma, const float* stats, float* dx, const float* dx_add, float* dgamma,
                           float* dbeta, float* workspace, int N, int C, int g_dtype, int accumulate_w, void* stream);
// rows per workgroup of the backward kernel: 16 (4 per wave) puts ~600 workgroups on the chip for the ~10^4 key points of a step
// (64 left 60 % of the CUs empty and every wave walking 16 dependent rows)
constexpr int LN_BWD_ROWS = 16;
size_t dreg_layernorm_bwd_workspace_bytes(int N) { return (size_t)((N + LN_BWD_ROWS - 1) / LN_BWD_ROWS) * 512 * sizeof(float); }
// dy in g_dtype (0 bf16 / 1 fp32); dx fp32 (accumulated into when accumulate_dx); dgamma/dbeta fp32 (accumulated when accumulate_w)
int dreg_layernorm_bwd(const float* x, const void* dy, const float* gamma, const float* stats, float* dx, float* dgamma, float* dbeta,
                       float* workspace, int N, int C, int g_dtype, int accumulate_dx, int accumulate_w, void* stream)
{
    return dreg_layernorm_bwd_add(x, dy, gamma, stats, dx, accumulate_dx ? dx : nullptr, dgamma, dbeta, workspace, N, C, g_dtype, accumulate_w, stream);
}
// the same with an explicit addend: dx = LayerNorm-backward(dy) + dx_add  (dx_add fp32 [N,256] or null; may be dx itself).  Folds the
// gradient of a residual branch that by-passes the LayerNorm (transformer.py:238-293: x -> LN -> ... + x) into this launch.
int dreg_layernorm_bwd_add(const float* x, const void* dy, const float* gamma, const float* stats, float* dx, const float* dx_add, float* dgamma,
                           float* dbeta, float* workspace, int N, int C, int g_dtype, int accumulate_w, void* stream)
{
    if (C != 256) return DREG_EINVAL;
    if (N == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (N + LN_BWD_ROWS - 1) / LN_BWD_ROWS;
    if (g_dtype == 0) hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, x, (const bf16_t*)dy, gamma, stats, dx, workspace, N, LN_BWD_ROWS, dx_add);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nblk), dim3(256), 0, st, x, (const float*)dy, gamma, stats, dx, workspace, N, LN_BWD_ROWS, dx_add);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3(128), dim3(256), 0, st, workspace, dgamma, dbeta, nblk, accumulate_w);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// y32 (fp32, no pe) and / or y16 (bf16, + pe[row % pe_rows]) of the same LayerNorm from one read of x (either output may be null)
int dreg_layernorm_fwd2(const float* x, const float* gamma, const float* beta, const float* pe, int pe_rows, float* y32, void* y16, float* stats,
                        int N, int C, float eps, void* stream)
{
    if (C != 256 || (pe && pe_rows <= 0)) return DREG_EINVAL;
    if (N == 0) return DREG_OK;
    hipLaunchKernelGGL(layernorm_fwd2_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, pe, pe_rows, y32, (bf16_t*)y16, stats, N, eps);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// The row pass of the LayerNorm backward alone: dx = (LN-backward(dy [+ dy2]) + dx_add) + dx_add2, optional bf16 copy dx_bf16, block
// partials of dgamma / dbeta into `part` (dreg_layernorm_bwd_workspace_bytes(N), kept by the caller until dreg_layernorm_bwd_final_batched).
int dreg_layernorm_bwd_parts(const float* x, const void* dy, const void* dy2_bf16, const float* gamma, const float* stats, float* dx, const float* dx_add,
                             const float* dx_add2, void* dx_bf16, float* part, int N, int C, int g_dtype, void* stream)
{
    if (C != 256) return DREG_EINVAL;
    if (N == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (N + LN_BWD_ROWS - 1) / LN_BWD_ROWS;
    if (g_dtype == 0) hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, x, (const bf16_t*)dy, gamma, stats, dx, part, N, LN_BWD_ROWS, dx_add, dx_add2, (bf16_t*)dx_bf16, (const bf16_t*)dy2_bf16);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nblk), dim3(256), 0, st, x, (const float*)dy, gamma, stats, dx, part, N, LN_BWD_ROWS, dx_add, dx_add2, (bf16_t*)dx_bf16, (const bf16_t*)dy2_bf16);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_layernorm_bwd_blocks(int N) { return (N + LN_BWD_ROWS - 1) / LN_BWD_ROWS; }
// descs_dev: n records of 32 bytes { const float* part; float* dgamma; float* dbeta; int nblk (dreg_layernorm_bwd_blocks); int accumulate; }
int dreg_layernorm_bwd_final_batched(const void* descs_dev, int n, void* stream)
{
    if (n <= 0) return DREG_OK;
    hipLaunchKernelGGL(layernorm_bwd_final_batched_kernel, dim3(128 * n), dim3(256), 0, (hipStream_t)stream, (const LnFinalDesc*)descs_dev);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// dreg_overlap_bwd with accumulation: df = df_add + dlogit * w (df_add fp32 [N,256] or null; may be df), dw / db += when accumulate_w
int dreg_overlap_bwd_acc(const float* f, const float* w, const float* s, const float* gy, float* df, const float* df_add, float* dw, float* db,
                         int accumulate_w, float* workspace, int N, void* stream)
{
    if (N == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (N + 63) / 64;
    hipLaunchKernelGGL(overlap_bwd_kernel, dim3(nblk), dim3(256), 0, st, f, w, s, gy, df, workspace, N, 64, df_add);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(overlap_bwd_final_kernel, dim3(65), dim3(256), 0, st, workspace, dw, db, nblk, accumulate_w);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_posenc_sine(const float* xyz, float* pe, int N, float scale, float temperature, void* stream)
{
    if (N == 0) return DREG_OK;
    hipLaunchKernelGGL(posenc_sine_kernel, dim3(nblocks((size_t)N * 256)), dim3(256), 0, (hipStream_t)stream, xyz, pe, N,
                       scale * 6.283185307179586f, logf(temperature));
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_overlap_fwd(const float* f, const float* w, const float* b, float* s, int N, void* stream)
{
    if (N == 0) return DREG_OK;
    hipLaunchKernelGGL(overlap_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, f, w, b, s, N);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
size_t dreg_overlap_bwd_workspace_bytes(int N) { return (size_t)((N + 63) / 64) * 257 * sizeof(float); }
int dreg_overlap_bwd(const float* f, const float* w, const float* s, const float* gy, float* df, float* dw, float* db,
                     float* workspace, int N, void* stream)
{
    if (N == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (N + 63) / 64;
    hipLaunchKernelGGL(overlap_bwd_kernel, dim3(nblk), dim3(256), 0, st, f, w, s, gy, df, workspace, N, 64);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(overlap_bwd_final_kernel, dim3(65), dim3(256), 0, st, workspace, dw, db, nblk);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// out = (y > 0 ? g : 0) cast to out_dtype.  y_dtype / g_dtype / out_dtype: 0 bf16, 1 fp32
int dreg_relu_bwd(const void* y, const void* g, void* out, size_t n, int y_dtype, int g_dtype, int out_dtype, void* stream)
{
    if (n == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = nblocks(n);
#define RB(TY, TG, TO) hipLaunchKernelGGL((relu_bwd_kernel<TY, TG, TO>), dim3(nb), dim3(256), 0, st, (const TY*)y, (const TG*)g, (TO*)out, n)
    const int code = y_dtype * 4 + g_dtype * 2 + out_dtype;
    switch (code) {
        case 0: RB(bf16_t, bf16_t, bf16_t); break; case 1: RB(bf16_t, bf16_t, float); break;
        case 2: RB(bf16_t, float, bf16_t); break;  case 3: RB(bf16_t, float, float); break;
        case 4: RB(float, bf16_t, bf16_t); break;  case 5: RB(float, bf16_t, float); break;
        case 6: RB(float, float, bf16_t); break;   default: RB(float, float, float); break;
    }
#undef RB
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_weighted_kabsch(const float* a, const float* b, const float* w, float* out, int P, int N, float eps, void* stream)
{
    if (P == 0) return DREG_OK;
    hipLaunchKernelGGL(kabsch_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, a, b, w, out, N, eps);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// The same solve for every (pair, decoder layer) of a step in one launch, reading the shared row space directly:
// a = cat[src_xyz, tgt_corr], b = cat[src_corr, tgt_xyz], w = cat[src_ov, tgt_ov] (nerf_regtr.py:226-236) without the copies.
// xyz [R,3], corr [L,R,3], ov [L,R], probs int32 [P][4] = (s0, ns, t0, nt); out [P,L,3,4].
int dreg_weighted_kabsch_pairs(const float* xyz, const float* corr, const float* ov, const int* probs, float* out, int P, int L, int R,
                               float eps, void* stream)
{
    if (P == 0 || L == 0) return DREG_OK;
    hipLaunchKernelGGL(kabsch_pairs_kernel, dim3(P * L), dim3(256), 0, (hipStream_t)stream, xyz, corr, ov, probs, out, L, R, eps);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Voxel-average downsample of (xyz | feat) rows sharing (batch, floor(p/dl)); output rows ordered by (batch, ix, iy, iz).
// Outputs sized for the worst case N rows; n_out (device int) and batch_counts (device int[nbatch], zeroed here) report
// the sizes; inv_seg / inv_cnt (per input row) feed the backward pass.  err: device int, set to 1 on coordinate overflow.
size_t dreg_voxel_downsample_workspace_bytes(int N)
{
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)N);
    size_t scan_bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)N, rocprim::plus<uint32_t>());
    const size_t tmp = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    const size_t a = ((size_t)N * 8 + 255) / 256 * 256, b4 = ((size_t)(N + 1) * 4 + 255) / 256 * 256;
    return 2 * a + 5 * b4 + tmp + 256;
}
DREG_KNOB(int, g_own_sort, 0);     // tuning (include/dreg_nerf_probe.h): 1 = the one-workgroup kernel above for <= 131,072 keys.  OFF by default: measured 0.45 ms (9 k keys) to
                               // 1.45 ms (38 k keys) per round on its single CU against ~0.07 ms for rocPRIM's chip-wide sort + scan (round 4, rocprofv3); the rounds' host
                               // syncs then put the geometry phase on the critical path (19.2 vs 18.2 ms per step).  Results are identical (tested both ways).
#ifdef DREG_PROBE
void dreg_voxel_set_own_sort(int on) { g_own_sort = on ? 1 : 0; }
#endif
static int voxel_downsample_impl(const float* pts, const float* feats, const int* pt_batch, float* out_pts, float* out_feats,
                                 int* n_out, int* batch_counts, uint32_t* inv_seg, float* inv_cnt, int* err,
                                 uint32_t* order_out, uint32_t* starts_out,
                                 void* workspace, size_t workspace_bytes, int N, int C, int nbatch, float dl, void* stream, const uint8_t* frozen = nullptr)
{
    if (N <= 0 || C % 4) return DREG_EINVAL;
    if (workspace_bytes < dreg_voxel_downsample_workspace_bytes(N)) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t a = ((size_t)N * 8 + 255) / 256 * 256, b4 = ((size_t)(N + 1) * 4 + 255) / 256 * 256;
    char* w = (char*)workspace;
    uint64_t* keys = (uint64_t*)w; w += a;
    uint64_t* keys_s = (uint64_t*)w; w += a;
    uint32_t* vals = (uint32_t*)w; w += b4;
    uint32_t* order = (uint32_t*)w; w += b4;
    uint32_t* head = (uint32_t*)w; w += b4;
    uint32_t* scan = (uint32_t*)w; w += b4;
    uint32_t* starts = (uint32_t*)w; w += b4;
    if (order_out) order = order_out;
    if (starts_out) starts = starts_out;
    void* tmp = w;
    size_t tmp_bytes = workspace_bytes - (size_t)(w - (char*)workspace);
    if (g_own_sort && N <= VS_MAX_N && frozen == nullptr) {
        // keys, sort, segments in one launch (voxel_sort_segments_kernel); `scan` / `head` stay unused
        hipLaunchKernelGGL(voxel_sort_segments_kernel, dim3(1), dim3(VS_THREADS), 0, st, pts, pt_batch, dl, keys, keys_s, vals, order, starts, batch_counts, n_out, err, N, nbatch);
        DREG_LAUNCH_CHECK();
        hipLaunchKernelGGL(segment_mean_kernel, dim3((N + 3) / 4), dim3(256), 0, st, pts, feats, order, starts, n_out, out_pts, out_feats, inv_seg, inv_cnt, C);
        DREG_LAUNCH_CHECK();
        return DREG_OK;
    }
    if (hipMemsetAsync(batch_counts, 0, sizeof(int) * nbatch, st) != hipSuccess) return DREG_ELAUNCH;
    if (hipMemsetAsync(err, 0, sizeof(int), st) != hipSuccess) return DREG_ELAUNCH;
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(voxel_keys_kernel, dim3(nb), dim3(256), 0, st, pts, pt_batch, keys, vals, N, dl, err, frozen);
    DREG_LAUNCH_CHECK();
    size_t sb = tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, sb, keys, keys_s, vals, order, (size_t)N, 0, 64, st) != hipSuccess) return DREG_ELAUNCH;
    hipLaunchKernelGGL(segment_heads_kernel, dim3(nb), dim3(256), 0, st, keys_s, head, N);
    DREG_LAUNCH_CHECK();
    sb = tmp_bytes;
    if (rocprim::inclusive_scan(tmp, sb, head, scan, (size_t)N, rocprim::plus<uint32_t>(), st) != hipSuccess) return DREG_ELAUNCH;
    hipLaunchKernelGGL(segment_starts_kernel, dim3(nb), dim3(256), 0, st, keys_s, head, scan, starts, batch_counts, n_out, N);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(segment_mean_kernel, dim3((N + 3) / 4), dim3(256), 0, st, pts, feats, order, starts, n_out, out_pts, out_feats, inv_seg, inv_cnt, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_voxel_downsample_fwd(const float* pts, const float* feats, const int* pt_batch, float* out_pts, float* out_feats,
                              int* n_out, int* batch_counts, uint32_t* inv_seg, float* inv_cnt, int* err,
                              void* workspace, size_t workspace_bytes, int N, int C, int nbatch, float dl, void* stream)
{
    if (feats == nullptr) return DREG_EINVAL;
    return voxel_downsample_impl(pts, feats, pt_batch, out_pts, out_feats, n_out, batch_counts, inv_seg, inv_cnt, err, nullptr, nullptr,
                                 workspace, workspace_bytes, N, C, nbatch, dl, stream);
}
// The same downsample split in two, so that everything whose size is data dependent can run before the feature network:
// plan = cells, sort, segments and averaged points (needs only xyz); order uint32 [N], starts uint32 [N+1] are kept by the caller.
int dreg_voxel_downsample_plan(const float* pts, const int* pt_batch, float* out_pts, int* n_out, int* batch_counts,
                               uint32_t* inv_seg, float* inv_cnt, int* err, uint32_t* order, uint32_t* starts,
                               void* workspace, size_t workspace_bytes, int N, int nbatch, float dl, void* stream)
{
    if (!order || !starts) return DREG_EINVAL;
    return voxel_downsample_impl(pts, nullptr, pt_batch, out_pts, nullptr, n_out, batch_counts, inv_seg, inv_cnt, err, order, starts,
                                 workspace, workspace_bytes, N, 0, nbatch, dl, stream);
}
// plan with per-batch pass-through flags (frozen uint8 [nbatch] on the device, may be null): see voxel_keys_kernel
int dreg_voxel_downsample_plan_frozen(const float* pts, const int* pt_batch, float* out_pts, int* n_out, int* batch_counts,
                                      uint32_t* inv_seg, float* inv_cnt, int* err, uint32_t* order, uint32_t* starts,
                                      void* workspace, size_t workspace_bytes, int N, int nbatch, float dl, const uint8_t* frozen, void* stream)
{
    if (!order || !starts || nbatch > 65535) return DREG_EINVAL;
    return voxel_downsample_impl(pts, nullptr, pt_batch, out_pts, nullptr, n_out, batch_counts, inv_seg, inv_cnt, err, order, starts,
                                 workspace, workspace_bytes, N, 0, nbatch, dl, stream, frozen);
}
// apply = segment means of a feature matrix over a plan: out_feats [M,C] fp32, M = the plan's n_out (device int, rows beyond it untouched)
int dreg_voxel_segment_mean(const float* feats, const uint32_t* order, const uint32_t* starts, const int* n_out, float* out_feats,
                            int M, int C, void* stream)
{
    if (M <= 0) return DREG_OK;
    if (C % 4) return DREG_EINVAL;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, feats, order, starts, n_out,
                       (float*)nullptr, out_feats, (uint32_t*)nullptr, (float*)nullptr, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_voxel_downsample_bwd(const float* gout, const uint32_t* inv_seg, const float* inv_cnt, float* gin, int N, int C, void* stream)
{
    if (N == 0) return DREG_OK;
    hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3(nblocks((size_t)N * (C / 4))), dim3(256), 0, (hipStream_t)stream, gout, inv_seg, inv_cnt, gin, N, C);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// norm_out[0] = ||g||_2 over a flat fp32 buffer.  workspace: fp32 [1024].
int dreg_grad_norm(const float* g, float* norm_out, float* workspace, size_t n, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int nb = nblocks((n + 3) / 4, 256, 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, workspace, n);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, workspace, norm_out, nb);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// One AdamW step (torch.optim.AdamW semantics, step count `step` >= 1) over flat fp32 p/g/m/v with the gradient first
// scaled by clip_grad_norm_(max_norm) using the device-resident norm (max_norm <= 0: no clipping).
int dreg_adamw_step(float* p, float* g, float* m, float* v, const float* norm, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float max_norm, void* stream)
{
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, norm, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, sqrtf(bc2), max_norm);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

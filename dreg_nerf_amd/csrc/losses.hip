// Training losses of the registration network for all pairs of a step at once (gfx950): the overlap BCE, the label
// consistency smooth-L1, the correspondence L1 and the InfoNCE feature loss, with their input gradients.
//
// Reference: train_nerf_regtr.py:186-229 (assembly; last decoder layer only; weights 1 / 1 / 0.1 / 1),
// conerf/loss/correspondence_loss.py:16-51, conerf/loss/feature_loss.py:24-73.  The reference's quirks are kept:
// BCEWithLogits(input = labels, target = prediction), and the [nl,N,1] x [N] broadcast of the correspondence loss, which
// makes it sum_j err_j * (sum w / max(sum w, eps)).
//
// Row space: R rows = every pair's (src | tgt) key points; probs int32 [P][4] = (s0, ns, t0, nt).  All reductions run in
// a fixed order inside one block per segment: results are deterministic.
#include "common.h"

__device__ __forceinline__ double block_sum_d(double v, double* red)
{
    const int t = threadIdx.x;
    red[t] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    const double r = red[0]; __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------------ point losses
// grid = 2P blocks: block (p, side).  gt, tilde: [L,R] labels; ov: [R] last-layer overlap prediction; corr: [R,3] last-layer
// correspondences; xyz [R,3]; pose [P,4,4].  partial [2P][4] = (bce sum, smooth-l1 sum, err sum, label sum).
// d_ov [R], d_corr [R,3]: gradients of the MEAN-over-pairs total (weights folded in).
__global__ __launch_bounds__(256) void reg_point_losses_kernel(
    const float* __restrict__ gt, const float* __restrict__ tilde, const float* __restrict__ ov, const float* __restrict__ corr,
    const float* __restrict__ xyz, const float* __restrict__ pose, const int* __restrict__ probs,
    float* __restrict__ partial, float* __restrict__ d_ov, float* __restrict__ d_corr,
    int L, int R, int P, int robust, float eps, float w_overlap, float w_corr)
{
    __shared__ double red[256];
    const int p = blockIdx.x >> 1, side = blockIdx.x & 1, t = threadIdx.x;
    const int s0 = probs[p * 4], ns = probs[p * 4 + 1], t0 = probs[p * 4 + 2], nt = probs[p * 4 + 3];
    const int r0 = side ? t0 : s0, n = side ? nt : ns;
    // rigid map of this side: src uses the pose, tgt its inverse (R^T, -R^T t)
    float Rm[3][3], tv[3];
    const float* Pm = pose + (size_t)p * 16;
    if (!side) {
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Rm[i][j] = Pm[i * 4 + j]; tv[i] = Pm[i * 4 + 3]; }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Rm[i][j] = Pm[j * 4 + i];
            tv[i] = -(Pm[0 * 4 + i] * Pm[0 * 4 + 3] + Pm[1 * 4 + i] * Pm[1 * 4 + 3] + Pm[2 * 4 + i] * Pm[2 * 4 + 3]);
        }
    }
    double bce = 0.0, sl1 = 0.0, err = 0.0, wsum = 0.0;
    const float inv_n = 1.f / (float)(ns + nt);
    for (int i = t; i < n; i += 256) {
        const int row = r0 + i;
        const float x = gt[(size_t)(L - 1) * R + row], tt = ov[row];
        bce += (double)(fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x))));
        d_ov[row] = -x * inv_n * w_overlap / (float)P;
        for (int l = 0; l < L; ++l) {
            const float g = gt[(size_t)l * R + row];
            if (tilde) {     // null: the label-consistency term is formed later by dreg_nerf_cont_deferred (it carries no gradient)
                const float d = g - tilde[(size_t)l * R + row], a = fabsf(d);
                sl1 += (double)(a < 1.f ? 0.5f * d * d : a - 0.5f);
            }
            wsum += (double)g;
        }
        float e = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float tx = xyz[(size_t)row * 3] * Rm[c][0] + xyz[(size_t)row * 3 + 1] * Rm[c][1] + xyz[(size_t)row * 3 + 2] * Rm[c][2] + tv[c];
            float d = corr[(size_t)row * 3 + c] - tx;
            if (robust) d = sqrtf((d / 0.5f) * (d / 0.5f) + 1.f) - 1.f;
            e += fabsf(d);
        }
        err += (double)e;
    }
    bce = block_sum_d(bce, red); sl1 = block_sum_d(sl1, red); err = block_sum_d(err, red); wsum = block_sum_d(wsum, red);
    if (t == 0) {
        float* o = partial + (size_t)blockIdx.x * 4;
        o[0] = (float)bce; o[1] = (float)sl1; o[2] = (float)err; o[3] = (float)wsum;
    }
    const float factor = (float)wsum / fmaxf((float)wsum, eps) * w_corr / (float)P;
    for (int i = t; i < n; i += 256) {
        const int row = r0 + i;
        for (int c = 0; c < 3; ++c) {
            const float tx = xyz[(size_t)row * 3] * Rm[c][0] + xyz[(size_t)row * 3 + 1] * Rm[c][1] + xyz[(size_t)row * 3 + 2] * Rm[c][2] + tv[c];
            const float d = corr[(size_t)row * 3 + c] - tx;
            float g;
            if (robust) {
                // |ph(d)| with ph(d) = sqrt((d/0.5)^2 + 1) - 1 >= 0: derivative (d / 0.25) / sqrt((d/0.5)^2 + 1)
                g = (d / 0.25f) / sqrtf((d / 0.5f) * (d / 0.5f) + 1.f);
            } else g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            d_corr[(size_t)row * 3 + c] = g * factor;
        }
    }
}

// ------------------------------------------------------------------------------------------------ InfoNCE
// pass 1 (needs only coordinates): nearest target of every transformed source point, its distance test, the pair's positive
// count (a sum of 0/1: exact in fp32, order independent; formed by a second tiny launch).  One wave per source row.  nn [Rs] (index inside the pair's
// target set; the lowest index on ties), mask [Rs] (0/1), count [P] (zeroed by the caller); rows indexed by their position in
// the concatenated SOURCE row list (src_off[p] = sum of ns of earlier pairs).
constexpr int INFONCE_LDS_PTS = 2048;     // target points staged per workgroup (24 KB); larger sets are read from global memory
__global__ __launch_bounds__(256) void infonce_nn_kernel(const float* __restrict__ xyz, const float* __restrict__ pose, const int* __restrict__ probs,
                                                         const int* __restrict__ src_off, int* __restrict__ nn, float* __restrict__ mask,
                                                         float* __restrict__ count, int P, int total_src, float r_p)
{
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    // the target points of the workgroup's pair once into LDS (its four source rows almost always share the pair): the loop below
    // was a chain of ~19 L2 round trips per wave (95 % of the wave cycles parked)
    __shared__ float sx[INFONCE_LDS_PTS * 3];
    int p0 = 0;
    while (p0 + 1 < P && src_off[p0 + 1] <= (int)blockIdx.x * 4) ++p0;
    const int t00 = probs[p0 * 4 + 2], nt0 = probs[p0 * 4 + 3];
    const bool staged = nt0 <= INFONCE_LDS_PTS;
    if (staged)
        for (int k = threadIdx.x; k < nt0 * 3; k += 256) sx[k] = xyz[(size_t)t00 * 3 + k];
    __syncthreads();
    if (wv >= total_src) return;
    int p = p0;
    while (p + 1 < P && src_off[p + 1] <= wv) ++p;
    const int i = wv - src_off[p];
    const int s0 = probs[p * 4], t0 = probs[p * 4 + 2], nt = probs[p * 4 + 3];
    const float* Pm = pose + (size_t)p * 16;
    const float* s = xyz + (size_t)(s0 + i) * 3;
    float a[3];
    for (int c = 0; c < 3; ++c) a[c] = s[0] * Pm[c * 4] + s[1] * Pm[c * 4 + 1] + s[2] * Pm[c * 4 + 2] + Pm[c * 4 + 3];
    float best = 3.4e38f; int bj = 0x7fffffff;
    const float* tb = (staged && p == p0) ? sx : xyz + (size_t)t0 * 3;      // same values either way
#pragma unroll 4                                     // the four iterations' coordinate loads are independent: in flight together
    for (int j = lane; j < nt; j += 64) {
        const float* q = tb + (size_t)j * 3;
        const float dx = a[0] - q[0], dy = a[1] - q[1], dz = a[2] - q[2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) { best = d2; bj = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float b2 = __shfl_xor(best, o, 64);
        const int j2 = __shfl_xor(bj, o, 64);
        if (b2 < best || (b2 == best && j2 < bj)) { best = b2; bj = j2; }
    }
    if (lane == 0) {
        nn[wv] = bj;
        const float m = sqrtf(best) < r_p ? 1.f : 0.f;
        mask[wv] = m;
        // the pair's positive count is summed by infonce_count_kernel: 4,876 float atomics on four addresses were most of this kernel
    }
}

// pass 2: one wave per source row over the pair's logits block (row-major [ns][nt] at logit_off[p]).  Included columns:
// dist >= r_n, plus the nearest neighbour.  loss_row = -logit[nn] + logsumexp(included); dlogits (in place) =
// mask/count * (softmax(included) - onehot(nn)) * scale.
__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ logits, const float* __restrict__ xyz, const float* __restrict__ pose,
                                                           const int* __restrict__ probs, const int* __restrict__ src_off, const long long* __restrict__ logit_off,
                                                           const int* __restrict__ nn, const float* __restrict__ mask, const float* __restrict__ count,
                                                           float* __restrict__ loss_row, int P, int total_src, float r_n, float scale, int write_grad)
{
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    __shared__ float sx[INFONCE_LDS_PTS * 3];          // as in infonce_nn_kernel
    int p0 = 0;
    while (p0 + 1 < P && src_off[p0 + 1] <= (int)blockIdx.x * 4) ++p0;
    const int t00 = probs[p0 * 4 + 2], nt0 = probs[p0 * 4 + 3];
    const bool staged = nt0 <= INFONCE_LDS_PTS;
    if (staged)
        for (int k = threadIdx.x; k < nt0 * 3; k += 256) sx[k] = xyz[(size_t)t00 * 3 + k];
    __syncthreads();
    if (wv >= total_src) return;
    int p = p0;
    while (p + 1 < P && src_off[p + 1] <= wv) ++p;
    const int i = wv - src_off[p];
    const int s0 = probs[p * 4], t0 = probs[p * 4 + 2], nt = probs[p * 4 + 3];
    const float* Pm = pose + (size_t)p * 16;
    const float* s = xyz + (size_t)(s0 + i) * 3;
    float a[3];
    for (int c = 0; c < 3; ++c) a[c] = s[0] * Pm[c * 4] + s[1] * Pm[c * 4 + 1] + s[2] * Pm[c * 4 + 2] + Pm[c * 4 + 3];
    const float* tb = (staged && p == p0) ? sx : xyz + (size_t)t0 * 3;
    float* lg = logits + logit_off[p] + (size_t)i * ((nt + 3) & ~3);      // rows padded to four floats (the GEMMs' vector loads)
    const int jn = nn[wv];
    // online logsumexp over the included columns
    float mx = -3.4e38f, se = 0.f;
#pragma unroll 4
    for (int j = lane; j < nt; j += 64) {
        const float* q = tb + (size_t)j * 3;
        const float dx = a[0] - q[0], dy = a[1] - q[1], dz = a[2] - q[2];
        const bool inc = (j == jn) || !(sqrtf(dx * dx + dy * dy + dz * dz) < r_n);
        const float v = lg[j];                       // unconditional: the load does not wait for the distance test
        if (inc) {
            if (v > mx) { se = se * expf(mx - v) + 1.f; mx = v; } else se += expf(v - mx);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(se, o, 64);
        const float mm = fmaxf(mx, m2);
        se = se * expf(mx - mm) + s2 * expf(m2 - mm);
        mx = mm;
    }
    const float lse = mx + logf(se);
    const float lnn = lg[jn];
    if (lane == 0) loss_row[wv] = mask[wv] * (lse - lnn);
    if (!write_grad) return;
    const float cnt = count[p];
    const float k = (mask[wv] > 0.f && cnt > 0.f) ? scale / cnt : 0.f;
#pragma unroll 4
    for (int j = lane; j < nt; j += 64) {
        const float* q = tb + (size_t)j * 3;
        const float dx = a[0] - q[0], dy = a[1] - q[1], dz = a[2] - q[2];
        const bool inc = (j == jn) || !(sqrtf(dx * dx + dy * dy + dz * dz) < r_n);
        const float v = lg[j];
        float g = inc ? expf(v - lse) : 0.f;
        if (j == jn) g -= 1.f;
        lg[j] = g * k;
    }
}

// final: out[5] = mean over pairs of (overlap, nerf_cont, feature, corr, total)
__global__ __launch_bounds__(256) void reg_losses_final_kernel(const float* __restrict__ partial, const float* __restrict__ loss_row, const float* __restrict__ count,
                                                               const int* __restrict__ probs, const int* __restrict__ src_off, float* __restrict__ out,
                                                               int P, int L, float eps, float w_overlap, float w_cont, float w_feat, float w_corr)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    double acc[4] = {0, 0, 0, 0};
    for (int p = 0; p < P; ++p) {
        const int ns = probs[p * 4 + 1], nt = probs[p * 4 + 3];
        double f = 0.0;
        for (int i = t; i < ns; i += 256) f += (double)loss_row[src_off[p] + i];
        f = block_sum_d(f, red);
        if (t == 0) {
            const float* a = partial + (size_t)(2 * p) * 4;
            const float* b = a + 4;
            const double n = (double)(ns + nt);
            acc[0] += ((double)a[0] + (double)b[0]) / n;
            acc[1] += ((double)a[1] + (double)b[1]) / (n * L);
            acc[2] += f / (double)count[p];           // 0/0 -> nan, as loss[mask].sum() / mask.sum() of the reference
            acc[3] += (double)a[2] * ((double)a[3] / fmax((double)a[3], (double)eps)) + (double)b[2] * ((double)b[3] / fmax((double)b[3], (double)eps));
        }
    }
    if (t == 0) {
        for (int k = 0; k < 4; ++k) out[k] = (float)(acc[k] / P);
        out[4] = w_overlap * out[0] + w_cont * out[1] + w_feat * out[2] + w_corr * out[3];
    }
}

// The label-consistency term ('nerf_cont', train_nerf_regtr.py:198-201) formed AFTER the other three: its 'tilde' labels — the visibility of the
// PREDICTED correspondences, six point sets per block — feed nothing that carries a gradient, so the training step marches them on a
// side stream while backward runs and completes the step's loss values here.  Same arithmetic, in the same order, as reg_point_losses_kernel /
// reg_losses_final_kernel with tilde given: partial[.][1] and out[1], out[4] come out bit-identical.  grid = 2P blocks, then one single-thread launch.
__global__ __launch_bounds__(256) void nerf_cont_partial_kernel(const float* __restrict__ gt, const float* __restrict__ tilde, const int* __restrict__ probs,
                                                                float* __restrict__ partial, int L, int R)
{
    __shared__ double red[256];
    const int p = blockIdx.x >> 1, side = blockIdx.x & 1, t = threadIdx.x;
    const int r0 = side ? probs[p * 4 + 2] : probs[p * 4], n = side ? probs[p * 4 + 3] : probs[p * 4 + 1];
    double sl1 = 0.0;
    for (int i = t; i < n; i += 256) {
        const int row = r0 + i;
        for (int l = 0; l < L; ++l) {
            const float g = gt[(size_t)l * R + row], d = g - tilde[(size_t)l * R + row], a = fabsf(d);
            sl1 += (double)(a < 1.f ? 0.5f * d * d : a - 0.5f);
        }
    }
    sl1 = block_sum_d(sl1, red);
    if (t == 0) partial[(size_t)blockIdx.x * 4 + 1] = (float)sl1;
}
__global__ void nerf_cont_final_kernel(const float* __restrict__ partial, const int* __restrict__ probs, float* __restrict__ out, int P, int L,
                                       float w_overlap, float w_cont, float w_feat, float w_corr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double acc = 0.0;
    for (int p = 0; p < P; ++p) {
        const float* a = partial + (size_t)(2 * p) * 4;
        const float* b = a + 4;
        const double n = (double)(probs[p * 4 + 1] + probs[p * 4 + 3]);
        acc += ((double)a[1] + (double)b[1]) / (n * L);
    }
    out[1] = (float)(acc / P);
    out[4] = w_overlap * out[0] + w_cont * out[1] + w_feat * out[2] + w_corr * out[3];
}

// ------------------------------------------------------------------------------------------------ InfoNCE GEMMs (round 5)
// The feature loss's matrix products (feature_loss.py:24-60: logits = f_a (triu(W) + triu(W)^T) f_p^T per pair, and their gradients) as
// BATCHED fp32 GEMMs on the exact-fp32 MFMA (16 x 16 x 4), one launch per dependency level over a descriptor table instead of one library
// GEMM per pair and product (14 rocBLAS launches + ~40 small ATen launches per step in round 4).
//   C[M x N] = op(A)[M x K] . op(B)[K x N],  row-major;  tA = 0: A[m * lda + k], 1: A[k * lda + m];  tB = 0: B[k * ldb + n], 1: B[n * ldb + k].
// The operands are addressed as (base pointer id, element offset): the table depends on the step's segment lengths only and is built once
// per row-space table; the base pointers (this step's tensors) are kernel arguments.  64 x 64 tiles, 16-wide K steps through LDS ([k][m] /
// [k][n] images, rows 80 floats apart: the four k rows of an MFMA operand read land in disjoint bank groups), register prefetch of the next
// K step.  Every output element adds its K steps in ascending order: deterministic.
struct GemmDesc { int a_id, b_id, c_id, tA, tB, M, N, K, lda, ldb, ldc, tile0; long long a_off, b_off, c_off; };
static_assert(sizeof(GemmDesc) == 72, "descriptor layout is part of the ABI (dreg_gemm_f32_desc_bytes)");
struct GemmBases { const float* p[8]; };
constexpr int GT = 64, GK = 16, GLD = 80;

__global__ __launch_bounds__(256) void gemm_f32_batched_kernel(const GemmDesc* __restrict__ descs, int n, GemmBases bases)
{
    __shared__ float sA[GK][GLD], sB[GK][GLD];
    int d = 0;
    while (d + 1 < n && descs[d + 1].tile0 <= (int)blockIdx.x) ++d;
    const GemmDesc D = descs[d];
    const int tiles_n = (D.N + GT - 1) / GT;
    const int tl = (int)blockIdx.x - D.tile0, tm = tl / tiles_n, tn = tl - tm * tiles_n;
    const int m0 = tm * GT, n0 = tn * GT;
    const float* A = bases.p[D.a_id] + D.a_off;
    const float* B = bases.p[D.b_id] + D.b_off;
    float* C = const_cast<float*>(bases.p[D.c_id]) + D.c_off;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;                   // wave tile 32 x 32
    const int fr = lane & 15, kg = lane >> 4;
    // staging roles.  contiguous-along-K operand (tA = 0 / tB = 1): thread = (row r = t / 4, k4 = 4 (t % 4)), four consecutive k;
    // contiguous-along-M/N operand (tA = 1 / tB = 0): thread = (k = t / 16, c4 = 4 (t % 16)), four consecutive rows / columns
    float ra[4], rb[4];
    auto ldg = [&](const float* P, int tr, int rows_total, int row0, int ld, int k0, float (&r)[4]) {
        // tr = 1: P[k * ld + row] (row-contiguous); tr = 0: P[row * ld + k] (k-contiguous)
        if (tr) {
            const int k = k0 + (t >> 4), c = row0 + 4 * (t & 15);
            const float* q = P + (size_t)k * ld + c;
            const bool kv = k < D.K;
            if (kv && c + 3 < rows_total && ((ld | c) & 3) == 0 && (((uintptr_t)P) & 15) == 0) { const float4 v = *reinterpret_cast<const float4*>(q); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = (kv && c + e < rows_total) ? q[e] : 0.f;
            }
        } else {
            const int row = row0 + (t >> 2), k = k0 + 4 * (t & 3);
            const float* q = P + (size_t)row * ld + k;
            const bool rv = row < rows_total;
            if (rv && k + 3 < D.K && ((ld | k) & 3) == 0 && (((uintptr_t)P) & 15) == 0) { const float4 v = *reinterpret_cast<const float4*>(q); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = (rv && k + e < D.K) ? q[e] : 0.f;
            }
        }
    };
    auto sts = [&](float (*S)[GLD], int tr, const float (&r)[4]) {
        if (tr) *reinterpret_cast<float4*>(&S[t >> 4][4 * (t & 15)]) = make_float4(r[0], r[1], r[2], r[3]);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) S[4 * (t & 3) + e][t >> 2] = r[e];
        }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nk = (D.K + GK - 1) / GK;
    ldg(A, D.tA, D.M, m0, D.lda, 0, ra);
    ldg(B, !D.tB, D.N, n0, D.ldb, 0, rb);
    for (int ks = 0; ks < nk; ++ks) {
        __syncthreads();                                       // the previous step's fragment reads are done
        sts(sA, D.tA, ra);
        sts(sB, !D.tB, rb);
        __syncthreads();
        if (ks + 1 < nk) { ldg(A, D.tA, D.M, m0, D.lda, (ks + 1) * GK, ra); ldg(B, !D.tB, D.N, n0, D.ldb, (ks + 1) * GK, rb); }
#pragma unroll
        for (int q = 0; q < GK / 4; ++q) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = sA[q * 4 + kg][wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = sB[q * 4 + kg][wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C/D layout of the 16 x 16 MFMA: column = lane & 15, rows = (lane >> 4) * 4 + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 32 + i * 16 + kg * 4 + r;
            if (m >= D.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nn_ = n0 + wn * 32 + j * 16 + fr;
                if (nn_ < D.N) C[(size_t)m * D.ldc + nn_] = acc[i][j][r];
            }
        }
}
// Wsym = triu(W) + triu(W)^T (feature_loss.py:43-44), [E][E]
__global__ void infonce_wsym_kernel(const float* __restrict__ W, float* __restrict__ out, int E)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * E) return;
    const int r = i / E, c = i - r * E;
    out[i] = (r <= c ? W[(size_t)r * E + c] : 0.f) + (c <= r ? W[(size_t)c * E + r] : 0.f);
}

// The stand-in {0,1} visibility labels of data WITHOUT NeRF blocks on disk (synthetic scenes: dreg_nerf_amd/synth.py synthetic_overlap_gt — bench.py's default
// workload, the tests): the half-space test 1[x + 0.31 y - 0.17 z > 0.0123] of the key points (gt, the same row for all L layers) and of the L layers'
// predicted correspondences (tilde), with the roundings of the element-wise fp32 formula (no contraction).
__device__ __forceinline__ float halfspace_label(const float* __restrict__ p)
{
    const float s = __fsub_rn(__fadd_rn(p[0], __fmul_rn(0.31f, p[1])), __fmul_rn(0.17f, p[2]));
    return s > 0.0123f ? 1.f : 0.f;
}
__global__ __launch_bounds__(256) void halfspace_labels_kernel(const float* __restrict__ xyz, const float* __restrict__ corr, float* __restrict__ gt,
                                                               float* __restrict__ tilde, int L, int R)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float g = halfspace_label(xyz + (size_t)r * 3);
    for (int l = 0; l < L; ++l) {
        gt[(size_t)l * R + r] = g;
        tilde[(size_t)l * R + r] = halfspace_label(corr + ((size_t)l * R + r) * 3);
    }
}

extern "C" {

int dreg_halfspace_labels(const float* xyz, const float* corr, float* gt, float* tilde, int L, int R, void* stream)
{
    if (R <= 0 || L <= 0) return DREG_OK;
    if (!xyz || !corr || !gt || !tilde) return DREG_EINVAL;
    hipLaunchKernelGGL(halfspace_labels_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz, corr, gt, tilde, L, R);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_reg_point_losses(const float* gt, const float* tilde, const float* ov, const float* corr, const float* xyz, const float* pose,
                          const int* probs, float* partial, float* d_ov, float* d_corr, int L, int R, int P, int robust, float eps,
                          float w_overlap, float w_corr, void* stream)
{
    if (P <= 0) return DREG_OK;
    hipLaunchKernelGGL(reg_point_losses_kernel, dim3(2 * P), dim3(256), 0, (hipStream_t)stream, gt, tilde, ov, corr, xyz, pose, probs, partial, d_ov, d_corr,
                       L, R, P, robust, eps, w_overlap, w_corr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// count[p] = number of positives of pair p (sum of its 0/1 mask entries: exact in fp32 in any order); one workgroup per pair
__global__ __launch_bounds__(256) void infonce_count_kernel(const float* __restrict__ mask, const int* __restrict__ probs, const int* __restrict__ src_off,
                                                            float* __restrict__ count)
{
    __shared__ float red[256];
    const int p = blockIdx.x, t = threadIdx.x, ns = probs[p * 4 + 1];
    float c = 0.f;
#pragma unroll 4
    for (int i = t; i < ns; i += 256) c += mask[src_off[p] + i];
    red[t] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    if (t == 0) count[p] = red[0];
}
int dreg_infonce_nn(const float* xyz, const float* pose, const int* probs, const int* src_off, int* nn, float* mask, float* count, int P,
                    int total_src, float r_p, void* stream)
{
    if (P <= 0 || total_src <= 0) return DREG_OK;
    hipLaunchKernelGGL(infonce_nn_kernel, dim3((total_src + 3) / 4), dim3(256), 0, (hipStream_t)stream, xyz, pose, probs, src_off, nn, mask, count,
                       P, total_src, r_p);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(infonce_count_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, mask, probs, src_off, count);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// logits: every pair's [ns][nt] block at element offset logit_off[p]; overwritten by the gradient when write_grad.
// scale: weight of the feature loss in the mean-over-pairs total (w_feat / P).
int dreg_infonce_rows(float* logits, const float* xyz, const float* pose, const int* probs, const int* src_off, const long long* logit_off,
                      const int* nn, const float* mask, const float* count, float* loss_row, int P, int total_src, float r_n, float scale,
                      int write_grad, void* stream)
{
    if (P <= 0 || total_src <= 0) return DREG_OK;
    hipLaunchKernelGGL(infonce_rows_kernel, dim3((total_src + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, xyz, pose, probs, src_off, logit_off,
                       nn, mask, count, loss_row, P, total_src, r_n, scale, write_grad);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_reg_losses_final(const float* partial, const float* loss_row, const float* count, const int* probs, const int* src_off, float* out,
                          int P, int L, float eps, float w_overlap, float w_cont, float w_feat, float w_corr, void* stream)
{
    if (P <= 0) return DREG_OK;
    hipLaunchKernelGGL(reg_losses_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, loss_row, count, probs, src_off, out, P, L, eps,
                       w_overlap, w_cont, w_feat, w_corr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// completes partial[.][1], out[1] (nerf_cont) and out[4] (total) of a step whose dreg_reg_point_losses ran with tilde = null
int dreg_nerf_cont_deferred(const float* gt, const float* tilde, const int* probs, float* partial, float* out, int P, int L, int R,
                            float w_overlap, float w_cont, float w_feat, float w_corr, void* stream)
{
    if (P <= 0) return DREG_OK;
    if (!gt || !tilde || !probs || !partial || !out) return DREG_EINVAL;
    hipLaunchKernelGGL(nerf_cont_partial_kernel, dim3(2 * P), dim3(256), 0, (hipStream_t)stream, gt, tilde, probs, partial, L, R);
    hipLaunchKernelGGL(nerf_cont_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, probs, out, P, L, w_overlap, w_cont, w_feat, w_corr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Batched fp32 GEMMs over a descriptor table (see gemm_f32_batched_kernel).  descs_dev: n records of dreg_gemm_f32_desc_bytes() bytes
//   { int a_id, b_id, c_id, tA, tB, M, N, K, lda, ldb, ldc, tile0; int64 a_off, b_off, c_off }
// with tile0 = the number of 64 x 64 output tiles of the records before (total_tiles = that of all n); bases: 8 device pointers the ids index.
int dreg_gemm_f32_desc_bytes(void) { return (int)sizeof(GemmDesc); }
int dreg_gemm_f32_batched(const void* descs_dev, int n, int total_tiles, const float* const* bases8, void* stream)
{
    if (n <= 0 || total_tiles <= 0) return DREG_OK;
    GemmBases b;
    for (int i = 0; i < 8; ++i) b.p[i] = bases8[i];
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const GemmDesc*)descs_dev, n, b);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_infonce_wsym(const float* W, float* out, int E, void* stream)
{
    hipLaunchKernelGGL(infonce_wsym_kernel, dim3((E * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, out, E);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

// Flash-style softmax attention on gfx950 MFMA for the RegTR point-set transformer.
//
// Replaces the un-fused nn.MultiheadAttention path (need_weights=True materialises [1,N,N] weights;
// conerf/register/transformer.py:242-281) and the correspondence decoder's einsum/softmax/einsum
// (conerf/register/nerf_regtr.py:289-306).  Two instantiations:
//   D = 32  : 8-head attention, V is [Nk, 32] per head, output in the operand dtype.
//   D = 256 : single-head correspondence attention over the 6 stacked layer outputs, V = xyz (3 fp32 columns,
//             accumulated on the VALU in fp32 so key-point coordinates never round to bf16).
// One workgroup = 4 waves = 64 query (or key) rows of one head; each wave owns 16 rows (one per lane column); the other operand is
// streamed through LDS in 64-row tiles (next tile prefetched into registers).  Score tiles are computed transposed (one 16x16x32
// MFMA per 32 channels) so that their C layout is the B-operand layout of the second product: probabilities never touch LDS (see
// "transposed formulation" below).  Strided operands (V, K, Q, dO as [row][channel]) are read with ds_read_b64_tr_b16 (bf16) or
// gathered (fp32).  fp32 operands use the exact 16x16x4 f32 MFMA with the k-permutation "lane group g owns k = 8g..8g+7".
#include "common.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8_t v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ f32x4_t mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mma(const Frag<float>& a, const Frag<float>& b, f32x4_t c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], c, 0, 0, 0);
    return c;
}
// 8 consecutive elements [row][k0 .. k0+7] of a row-major LDS tile
__device__ __forceinline__ Frag<bf16_t> frag_row(const char* base, int rs, int row, int k0, bf16_t) {
    Frag<bf16_t> f; f.v = *reinterpret_cast<const bf16x8_t*>(base + row * rs + k0 * 2); return f;
}
__device__ __forceinline__ Frag<float> frag_row(const char* base, int rs, int row, int k0, float) {
    Frag<float> f;
    float4 lo = *reinterpret_cast<const float4*>(base + row * rs + k0 * 4);
    float4 hi = *reinterpret_cast<const float4*>(base + row * rs + k0 * 4 + 16);
    f.v[0] = lo.x; f.v[1] = lo.y; f.v[2] = lo.z; f.v[3] = lo.w; f.v[4] = hi.x; f.v[5] = hi.y; f.v[6] = hi.z; f.v[7] = hi.w;
    return f;
}
// same from global memory (row pointer given)
__device__ __forceinline__ Frag<bf16_t> frag_glob(const bf16_t* p, bool valid) {
    Frag<bf16_t> f;
    if (valid) f.v = *reinterpret_cast<const bf16x8_t*>(p); else f.v = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    return f;
}
__device__ __forceinline__ Frag<float> frag_glob(const float* p, bool valid) {
    Frag<float> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = valid ? p[e] : 0.f;
    return f;
}
struct AttnArgs {
    const void *q, *k, *v;     // operand type T;  v = fp32 xyz [Nk,3] when XYZ
    void* o;                   // T [Nq, ...] or fp32 [H, Nq, 3] when XYZ
    float* lse;                // [H, Nq]
    int Nq, Nk;
    int ldq, ldk, ldv, ldo;    // row strides (elements)
    long hq, hk, hv, ho;       // per-head (or per-layer) element offsets
    float scale;
    // backward
    const void* dout;          // T [Nq,...] (ldo/ho) or fp32 [H,Nq,3]
    float* dvec;               // [H, Nq]   rowsum(dO * O)
    void *dq, *dk, *dv;        // outputs, operand type T, same strides as q/k/v
    // variable-length batching: blockIdx.z selects a problem {q_start, q_len, kv_start, kv_len} (rows of the shared tensors)
    const int4* probs;
    int Rq;                    // total query rows (row stride of lse / dvec / xyz-outputs per head)
};

struct Prob { int Nq, Nk; long q0, k0; };
__device__ __forceinline__ Prob get_prob(const AttnArgs& a) {
    Prob p; p.Nq = a.Nq; p.Nk = a.Nk; p.q0 = 0; p.k0 = 0;
    if (a.probs) { const int4 t = a.probs[blockIdx.z]; p.q0 = t.x; p.Nq = t.y; p.k0 = t.z; p.Nk = t.w; }
    return p;
}

// Tiles are register-staged: fetch the NEXT tile's granules into registers while the current tile (already in LDS) is being consumed,
// store them after the block has finished reading the current one — the global-memory latency of a tile hides behind a tile of MFMAs.
// Row padding of the LDS tiles.  bf16: 32 bytes — rows then start 32 bytes apart modulo 256, so the eight consecutive rows x 32 bytes a
// half-wave of ds_read_b64_tr_b16 touches tile the 64 banks exactly (with 16 bytes of padding the 4th row wrapped onto the 1st and the
// 5th overlapped the 2nd: SQ_LDS_BANK_CONFLICT was 46 % of the LDS cycles); the 16-byte row reads stay conflict-free.
template <typename T> __host__ __device__ constexpr int row_pad() { return sizeof(T) == 2 ? 32 : 16; }
// Two LDS buffers per operand and ONE barrier per tile: after the barrier of tile k (everyone has finished tile k-1) the registers holding
// tile k+1 are stored into the other buffer and tile k+2 is requested; tile k is then consumed from its buffer.
template <typename T, int D>
struct TileRegs {
    static constexpr int GPR = D * sizeof(T) / 16, NL = (64 * GPR + 255) / 256;
    uint4 v[NL];
    __device__ __forceinline__ void fetch(const T* src, long ld, int row0, int nrows_valid, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + i * 256, r = q / GPR, g = q % GPR;
            v[i] = make_uint4(0, 0, 0, 0);
            if (q < 64 * GPR && row0 + r < nrows_valid) v[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src + (long)(row0 + r) * ld) + g * 16);
        }
    }
    __device__ __forceinline__ void store(char* dst, int rs, int tid) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + i * 256, r = q / GPR, g = q % GPR;
            if (q < 64 * GPR) *reinterpret_cast<uint4*>(dst + r * rs + g * 16) = v[i];
        }
    }
};
// ------------------------------------------------------------------------------------------------ transposed formulation
// All three kernels compute the score tile TRANSPOSED with respect to the operand they stream, so that the MFMA C layout of the
// scores (lane (fr, kg) holds rows kg*4..kg*4+3 of column fr) IS the B-operand layout of the second product (lane (fr, kg) holds
// eight k-slots of column fr): two C tiles are converted in registers and fed straight back — no LDS round trip for the
// probabilities, and every softmax quantity (running max, sum, lse, D) is one value per lane instead of four.  The eight k-slots
// of lane group kg are rows kg*4..kg*4+3 of the first 16-row tile and of the second; the other operand of the second product is
// read with the same row permutation (frag_colp).
__device__ __forceinline__ Frag<bf16_t> frag_regs(const float (&v)[8], bf16_t) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    const u32x4_t w = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
    Frag<bf16_t> f; f.v = __builtin_bit_cast(bf16x8_t, w); return f;
}
__device__ __forceinline__ Frag<float> frag_regs(const float (&v)[8], float) {
    Frag<float> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = v[e];
    return f;
}
// element j of the result = tile[j < 4 ? row_lo + j : row_hi + j - 4][cb + (lane & 15)]   (row_lo / row_hi already per lane group)
__device__ __forceinline__ Frag<bf16_t> frag_colp(const char* base, int rs, int row_lo, int row_hi, int cb, int lane, bf16_t) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
    const int fi = lane & 15;
    const int off = (fi >> 2) * rs + (cb + (fi & 3) * 4) * 2;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base + row_lo * rs + off));
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base + row_hi * rs + off));
    Frag<bf16_t> f; f.v = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
__device__ __forceinline__ Frag<float> frag_colp(const char* base, int rs, int row_lo, int row_hi, int cb, int lane, float) {
    const int fi = lane & 15;
    Frag<float> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = *reinterpret_cast<const float*>(base + (e < 4 ? row_lo + e : row_hi + e - 4) * rs + (cb + fi) * 4);
    return f;
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) { *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(a, b), f2bf2(c, d)); }
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ float group4x_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }   // over the 4 lane groups
__device__ __forceinline__ float group4x_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// ------------------------------------------------------------------------------------------------ forward
// wave = 16 queries (lane's query: q0 + fr); S^T = K Q^T per 16-key tile, O^T = V^T P^T.
template <typename T, int D, bool XYZ>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a)
{
    constexpr int DV = 32;
    constexpr int KRS = D * sizeof(T) + row_pad<T>(), VRS = XYZ ? 16 : DV * sizeof(T) + row_pad<T>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = 64 * KRS + 64 * VRS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y;
    const Prob pb = get_prob(a);
    if (blockIdx.x * 64 >= pb.Nq) return;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const T* Q = (const T*)a.q + h * a.hq + pb.q0 * a.ldq;
    const T* K = (const T*)a.k + h * a.hk + pb.k0 * a.ldk;
    const int fr = lane & 15, kg = lane >> 4;

    Frag<T> qf[D / 32];       // B operand Q^T: column q = fr, k-slots kg*8..
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[kk] = frag_glob(Q + (long)(q0 + fr) * a.ldq + kk * 32 + kg * 8, q0 + fr < pb.Nq);
    const float c2 = a.scale * LOG2E;          // scores in the log2 domain
    float m = -INFINITY, l = 0.f;              // running max / this lane's share of the running sum, query q0 + fr
    f32x4_t o[2];                              // O^T[dv = u*16 + kg*4 + r][q = fr]
    float o3[3] = {0.f, 0.f, 0.f};
    o[0] = o[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rk;
    TileRegs<T, DV> rv;
    float4 rx = make_float4(0, 0, 0, 0);
    const T* Vp = XYZ ? nullptr : (const T*)a.v + h * a.hv + pb.k0 * a.ldv;
    auto fetch = [&](int k0) {
        rk.fetch(K, a.ldk, k0, pb.Nk, tid);
        if constexpr (XYZ) {
            const float* x = (const float*)a.v + pb.k0 * 3;
            rx = make_float4(0, 0, 0, 0);
            if (tid < 64 && k0 + tid < pb.Nk) rx = make_float4(x[(long)(k0 + tid) * 3], x[(long)(k0 + tid) * 3 + 1], x[(long)(k0 + tid) * 3 + 2], 0.f);
        } else {
            rv.fetch(Vp, a.ldv, k0, pb.Nk, tid);
        }
    };
    auto stage = [&](int buf) {
        char* dK = smem + buf * BUF;
        rk.store(dK, KRS, tid);
        if constexpr (XYZ) { if (tid < 64) *reinterpret_cast<float4*>(dK + 64 * KRS + tid * 16) = rx; }
        else rv.store(dK + 64 * KRS, VRS, tid);
    };
    fetch(0);
    stage(0);
    if (64 < pb.Nk) fetch(64);
    for (int k0 = 0, buf = 0; k0 < pb.Nk; k0 += 64, buf ^= 1) {
        __syncthreads();
        if (k0 + 64 < pb.Nk) { stage(buf ^ 1); if (k0 + 128 < pb.Nk) fetch(k0 + 128); }
        const char* sK = smem + buf * BUF;
        const char* sV = sK + 64 * KRS;
        f32x4_t s[4];          // s[t][r] = score(q = fr, key = k0 + t*16 + kg*4 + r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < D / 32; ++kk) s[t] = mma(frag_row(sK, KRS, t * 16 + fr, kk * 32 + kg * 8, T()), qf[kk], s[t]);
        }
        // The loop is VALU-bound (16 scores per lane and tile): the key mask runs in the last tile only (a wave-uniform branch), the
        // maximum is taken over the raw scores (c2 > 0) and the scale folds into the exponent's fused multiply-add.
        if (k0 + 64 > pb.Nk) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + t * 16 + kg * 4 + r >= pb.Nk) s[t][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
        mx = group4x_max(mx) * c2;                // every tile holds at least one real key: finite
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c2, -mn)); sum += s[t][r]; }
        l = l * alpha + sum;
        m = mn;
        if constexpr (XYZ) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o3[c] *= alpha;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 x = *reinterpret_cast<const float4*>(sV + (t * 16 + kg * 4 + r) * 16);
                    o3[0] += s[t][r] * x.x; o3[1] += s[t][r] * x.y; o3[2] += s[t][r] * x.z;
                }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float pv[8] = {s[2 * ks][0], s[2 * ks][1], s[2 * ks][2], s[2 * ks][3], s[2 * ks + 1][0], s[2 * ks + 1][1], s[2 * ks + 1][2], s[2 * ks + 1][3]};
                const Frag<T> pf = frag_regs(pv, T());
#pragma unroll
                for (int u = 0; u < 2; ++u) o[u] = mma(frag_colp(sV, VRS, ks * 32 + kg * 4, ks * 32 + 16 + kg * 4, u * 16, lane, T()), pf, o[u]);
            }
        }
    }
    l = group4x_sum(l);
    const float inv = 1.f / l;
    const int qr = q0 + fr;
    if constexpr (XYZ) {
        const float x = group4x_sum(o3[0]), y = group4x_sum(o3[1]), z = group4x_sum(o3[2]);
        if (kg == 0 && qr < pb.Nq) {
            float* op = (float*)a.o + ((long)h * a.Rq + pb.q0 + qr) * 3;
            op[0] = x * inv; op[1] = y * inv; op[2] = z * inv;
        }
    } else if (qr < pb.Nq) {
        T* op = (T*)a.o + h * a.ho + (pb.q0 + qr) * a.ldo + kg * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) store4(op + u * 16, o[u][0] * inv, o[u][1] * inv, o[u][2] * inv, o[u][3] * inv);
    }
    if (kg == 0 && qr < pb.Nq) a.lse[(long)h * a.Rq + pb.q0 + qr] = (m + __log2f(l)) * LN2;
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// wave = 16 queries; S^T and dP^T per 16-key tile, dQ^T = K^T dS^T.
template <typename T, int D, bool XYZ>
__global__ __launch_bounds__(256, (D > 32 ? 2 : 1)) void attn_bwd_dq_kernel(AttnArgs a)
{
    constexpr int DV = 32;
    constexpr int KRS = D * sizeof(T) + row_pad<T>(), VRS = XYZ ? 16 : DV * sizeof(T) + row_pad<T>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = 64 * KRS + 64 * VRS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y;
    const Prob pb = get_prob(a);
    if (blockIdx.x * 64 >= pb.Nq) return;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const T* Q = (const T*)a.q + h * a.hq + pb.q0 * a.ldq;
    const T* K = (const T*)a.k + h * a.hk + pb.k0 * a.ldk;
    const int fr = lane & 15, kg = lane >> 4;
    const int qr = q0 + fr;
    const bool qv = qr < pb.Nq;

    Frag<T> qf[D / 32];
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[kk] = frag_glob(Q + (long)qr * a.ldq + kk * 32 + kg * 8, qv);
    // per-lane quantities of query qr: lse (log2 domain), D = rowsum(dO * O), dO
    const float lse2 = qv ? a.lse[(long)h * a.Rq + pb.q0 + qr] * LOG2E : 0.f;
    float dvec, do3[3] = {0.f, 0.f, 0.f};
    Frag<T> dof;  // dO^T as B operand (column q = fr)
    if constexpr (!XYZ) {
        const T* dO = (const T*)a.dout + h * a.ho + pb.q0 * a.ldo;
        const T* O = (const T*)a.o + h * a.ho + pb.q0 * a.ldo;
        dof = frag_glob(dO + (long)qr * a.ldo + kg * 8, qv);
        const Frag<T> of = frag_glob(O + (long)qr * a.ldo + kg * 8, qv);
        float part = 0.f;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) part += bf2f((bf16_t)dof.v[e]) * bf2f((bf16_t)of.v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) part += dof.v[e] * of.v[e];
        }
        dvec = group4x_sum(part);
    } else {
        const float* dO = (const float*)a.dout + ((long)h * a.Rq + pb.q0 + qr) * 3;
        const float* O = (const float*)a.o + ((long)h * a.Rq + pb.q0 + qr) * 3;
        dvec = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { do3[c] = qv ? dO[c] : 0.f; dvec += do3[c] * (qv ? O[c] : 0.f); }
    }
    if (kg == 0 && qv) a.dvec[(long)h * a.Rq + pb.q0 + qr] = dvec;
    const float c2 = a.scale * LOG2E;
    f32x4_t dq[D / 16];        // dQ^T[d = u*16 + kg*4 + r][q = fr]
#pragma unroll
    for (int u = 0; u < D / 16; ++u) dq[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rk;
    TileRegs<T, DV> rv;
    float4 rx = make_float4(0, 0, 0, 0);
    const T* Vp = XYZ ? nullptr : (const T*)a.v + h * a.hv + pb.k0 * a.ldv;
    auto fetch = [&](int k0) {
        rk.fetch(K, a.ldk, k0, pb.Nk, tid);
        if constexpr (XYZ) {
            const float* x = (const float*)a.v + pb.k0 * 3;
            rx = make_float4(0, 0, 0, 0);
            if (tid < 64 && k0 + tid < pb.Nk) rx = make_float4(x[(long)(k0 + tid) * 3], x[(long)(k0 + tid) * 3 + 1], x[(long)(k0 + tid) * 3 + 2], 0.f);
        } else {
            rv.fetch(Vp, a.ldv, k0, pb.Nk, tid);
        }
    };
    auto stage = [&](int buf) {
        char* dK = smem + buf * BUF;
        rk.store(dK, KRS, tid);
        if constexpr (XYZ) { if (tid < 64) *reinterpret_cast<float4*>(dK + 64 * KRS + tid * 16) = rx; }
        else rv.store(dK + 64 * KRS, VRS, tid);
    };
    fetch(0);
    stage(0);
    if (64 < pb.Nk) fetch(64);
    for (int k0 = 0, buf = 0; k0 < pb.Nk; k0 += 64, buf ^= 1) {
        __syncthreads();
        if (k0 + 64 < pb.Nk) { stage(buf ^ 1); if (k0 + 128 < pb.Nk) fetch(k0 + 128); }
        const char* sK = smem + buf * BUF;
        const char* sV = sK + 64 * KRS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float dsv[8];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * ks + tt;
                f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < D / 32; ++kk) s = mma(frag_row(sK, KRS, t * 16 + fr, kk * 32 + kg * 8, T()), qf[kk], s);
                f32x4_t dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if constexpr (!XYZ) dp = mma(frag_row(sV, VRS, t * 16 + fr, kg * 8, T()), dof, dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = t * 16 + kg * 4 + r;
                    // keys past Nk need no mask: their K (and V / xyz) rows are zero in LDS, so whatever dS they get multiplies zero
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lse2));
                    float dpv = dp[r];
                    if constexpr (XYZ) {
                        const float4 x = *reinterpret_cast<const float4*>(sV + kr * 16);
                        dpv = do3[0] * x.x + do3[1] * x.y + do3[2] * x.z;
                    }
                    dsv[tt * 4 + r] = p * (dpv - dvec);      // the softmax scale is applied once, to dQ
                }
            }
            const Frag<T> df = frag_regs(dsv, T());
#pragma unroll
            for (int u = 0; u < D / 16; ++u) {
                dq[u] = mma(frag_colp(sK, KRS, ks * 32 + kg * 4, ks * 32 + 16 + kg * 4, u * 16, lane, T()), df, dq[u]);
                if (D > 32 && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep at most four operand fragments in flight (registers)
            }
        }
    }
    if (qv) {
        T* dQ = (T*)a.dq + h * a.hq + (pb.q0 + qr) * a.ldq + kg * 4;
#pragma unroll
        for (int u = 0; u < D / 16; ++u) store4(dQ + u * 16, dq[u][0] * a.scale, dq[u][1] * a.scale, dq[u][2] * a.scale, dq[u][3] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// wave = 16 keys (lane's key: key0 + fr); S and dP per 16-query tile, dK^T = Q^T dS, dV^T = dO^T P.
template <typename T, int D, bool XYZ>
__global__ __launch_bounds__(256, (D > 32 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnArgs a)
{
    constexpr int DV = 32;
    constexpr int QRS = D * sizeof(T) + row_pad<T>(), ORS = XYZ ? 16 : DV * sizeof(T) + row_pad<T>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = 64 * QRS + 64 * ORS + 128 * (int)sizeof(float);   // Q tile | dO tile | lse * log2(e) [64], dvec [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y;
    const Prob pb = get_prob(a);
    if (blockIdx.x * 64 >= pb.Nk) return;
    const int key0 = blockIdx.x * 64 + wave * 16;
    const T* Q = (const T*)a.q + h * a.hq + pb.q0 * a.ldq;
    const T* K = (const T*)a.k + h * a.hk + pb.k0 * a.ldk;
    const int fr = lane & 15, kg = lane >> 4;
    const int kr = key0 + fr;
    const bool kv = kr < pb.Nk;

    Frag<T> kf[D / 32], vf;     // B operands K^T / V^T: column key = fr
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) kf[kk] = frag_glob(K + (long)kr * a.ldk + kk * 32 + kg * 8, kv);
    float x3[3] = {0.f, 0.f, 0.f};
    if constexpr (!XYZ) {
        vf = frag_glob((const T*)a.v + h * a.hv + (pb.k0 + (long)kr) * a.ldv + kg * 8, kv);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) x3[c] = kv ? ((const float*)a.v)[(pb.k0 + kr) * 3 + c] : 0.f;
    }
    const float c2 = a.scale * LOG2E;
    f32x4_t dk[D / 16], dvv[2];   // dK^T[d = u*16 + kg*4 + r][key = fr], dV^T likewise
#pragma unroll
    for (int u = 0; u < D / 16; ++u) dk[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dvv[0] = dvv[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rq;
    TileRegs<T, DV> ro;
    float4 rg = make_float4(0, 0, 0, 0);
    float rl = 0.f, rd = 0.f;
    const T* dOp = XYZ ? nullptr : (const T*)a.dout + h * a.ho + pb.q0 * a.ldo;
    auto fetch = [&](int q0) {
        rq.fetch(Q, a.ldq, q0, pb.Nq, tid);
        if constexpr (XYZ) {
            rg = make_float4(0, 0, 0, 0);
            if (tid < 64 && q0 + tid < pb.Nq) {
                const float* g = (const float*)a.dout + ((long)h * a.Rq + pb.q0 + q0 + tid) * 3;
                rg = make_float4(g[0], g[1], g[2], 0.f);
            }
        } else {
            ro.fetch(dOp, a.ldo, q0, pb.Nq, tid);
        }
        rl = rd = 0.f;
        if (tid < 64 && q0 + tid < pb.Nq) { rl = a.lse[(long)h * a.Rq + pb.q0 + q0 + tid] * LOG2E; rd = a.dvec[(long)h * a.Rq + pb.q0 + q0 + tid]; }
    };
    auto stage = [&](int buf) {
        char* dQ = smem + buf * BUF;
        rq.store(dQ, QRS, tid);
        if constexpr (XYZ) { if (tid < 64) *reinterpret_cast<float4*>(dQ + 64 * QRS + tid * 16) = rg; }
        else ro.store(dQ + 64 * QRS, ORS, tid);
        float* dL = reinterpret_cast<float*>(dQ + 64 * QRS + 64 * ORS);
        if (tid < 64) { dL[tid] = rl; dL[64 + tid] = rd; }
    };
    fetch(0);
    stage(0);
    if (64 < pb.Nq) fetch(64);
    for (int q0 = 0, buf = 0; q0 < pb.Nq; q0 += 64, buf ^= 1) {
        __syncthreads();
        if (q0 + 64 < pb.Nq) { stage(buf ^ 1); if (q0 + 128 < pb.Nq) fetch(q0 + 128); }
        const char* sQ = smem + buf * BUF;
        const char* sO = sQ + 64 * QRS;
        const float* sL = reinterpret_cast<const float*>(sO + 64 * ORS);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float pv[8], dsv[8];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * ks + tt;
                // S[q][key] : rows = queries t*16 + kg*4 + r, column = this lane's key
                f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < D / 32; ++kk) s = mma(frag_row(sQ, QRS, t * 16 + fr, kk * 32 + kg * 8, T()), kf[kk], s);
                f32x4_t dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if constexpr (!XYZ) dp = mma(frag_row(sO, ORS, t * 16 + fr, kg * 8, T()), vf, dp);
                const float4 l4 = *reinterpret_cast<const float4*>(sL + t * 16 + kg * 4);
                const float4 d4 = *reinterpret_cast<const float4*>(sL + 64 + t * 16 + kg * 4);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = t * 16 + kg * 4 + r;
                    // no masks: query rows past Nq are zero in LDS (Q, dO, lse, D), so their p multiplies zero; key columns past Nk are not stored
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lq[r]));
                    float dpv = dp[r];
                    if constexpr (XYZ) {
                        const float4 g = *reinterpret_cast<const float4*>(sO + ql * 16);
                        dpv = x3[0] * g.x + x3[1] * g.y + x3[2] * g.z;
                    }
                    pv[tt * 4 + r] = p;
                    dsv[tt * 4 + r] = p * (dpv - dq_[r]);    // the softmax scale is applied once, to dK
                }
            }
            const Frag<T> sf = frag_regs(dsv, T());
#pragma unroll
            for (int u = 0; u < D / 16; ++u) {
                dk[u] = mma(frag_colp(sQ, QRS, ks * 32 + kg * 4, ks * 32 + 16 + kg * 4, u * 16, lane, T()), sf, dk[u]);
                if (D > 32 && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!XYZ) {
                const Frag<T> pf = frag_regs(pv, T());
#pragma unroll
                for (int u = 0; u < 2; ++u) dvv[u] = mma(frag_colp(sO, ORS, ks * 32 + kg * 4, ks * 32 + 16 + kg * 4, u * 16, lane, T()), pf, dvv[u]);
            }
        }
    }
    if (kv) {
        T* dK = (T*)a.dk + h * a.hk + (pb.k0 + (long)kr) * a.ldk + kg * 4;
#pragma unroll
        for (int u = 0; u < D / 16; ++u) store4(dK + u * 16, dk[u][0] * a.scale, dk[u][1] * a.scale, dk[u][2] * a.scale, dk[u][3] * a.scale);
        if constexpr (!XYZ) {
            T* dV = (T*)a.dv + h * a.hv + (pb.k0 + (long)kr) * a.ldv + kg * 4;
            store4(dV, dvv[0][0], dvv[0][1], dvv[0][2], dvv[0][3]);
            store4(dV + 16, dvv[1][0], dvv[1][1], dvv[1][2], dvv[1][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
template <typename T, int D, bool XYZ>
static int launch_attn(const AttnArgs& a, int H, int mode, hipStream_t st, int nprob = 1, int maxq = 0, int maxk = 0)
{
    const int gq = ((a.probs ? maxq : a.Nq) + 63) / 64, gk = ((a.probs ? maxk : a.Nk) + 63) / 64;
    constexpr int DV = 32;
    const size_t krs = D * sizeof(T) + row_pad<T>(), vrs = XYZ ? 16 : DV * sizeof(T) + row_pad<T>();
    if (mode == 0) {
        const size_t lds = 2 * (64 * krs + 64 * vrs);
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, XYZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_fwd_kernel<T, D, XYZ>), dim3(gq, H, nprob), dim3(256), lds, st, a);
    } else if (mode == 1) {
        const size_t lds = 2 * (64 * krs + 64 * vrs);
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, D, XYZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, XYZ>), dim3(gq, H, nprob), dim3(256), lds, st, a);
    } else {
        const size_t lds = 2 * (64 * krs + 64 * vrs + 128 * sizeof(float));
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, D, XYZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, XYZ>), dim3(gk, H, nprob), dim3(256), lds, st, a);
    }
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

static int dispatch_attn(const AttnArgs& a, int H, int D, int xyz, int dtype, int mode, hipStream_t st, int nprob = 1, int maxq = 0, int maxk = 0)
{
    if (!a.probs && (a.Nq <= 0 || a.Nk <= 0)) return DREG_EINVAL;
    if (a.probs && (nprob <= 0 || maxq <= 0 || maxk <= 0)) return DREG_EINVAL;
    if (D == 32 && !xyz) return dtype == 0 ? launch_attn<bf16_t, 32, false>(a, H, mode, st, nprob, maxq, maxk) : launch_attn<float, 32, false>(a, H, mode, st, nprob, maxq, maxk);
    if (D == 256 && xyz) return dtype == 0 ? launch_attn<bf16_t, 256, true>(a, H, mode, st, nprob, maxq, maxk) : launch_attn<float, 256, true>(a, H, mode, st, nprob, maxq, maxk);
    return DREG_EINVAL;
}

extern "C" {

// Multi-head attention core, head dim 32: o[:, h*32:(h+1)*32] = softmax(scale * q_h k_h^T) v_h.
// q [Nq, ldq], k [Nk, ldk], v [Nk, ldv], o [Nq, ldo] (dtype 0 bf16 / 1 fp32, heads at column offset h*32); lse fp32 [H, Nq].
int dreg_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int Nq, int Nk, int H,
                 int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.Nq = Nq; a.Nk = Nk;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.hq = a.hk = a.hv = a.ho = 32; a.scale = scale; a.Rq = Nq;
    return dispatch_attn(a, H, 32, 0, dtype, 0, (hipStream_t)stream);
}
// Gradients dq, dk, dv (same layouts/strides as q, k, v) from dout (layout of o).  dvec: fp32 [H, Nq] scratch.
int dreg_mha_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 float* dvec, void* dq, void* dk, void* dv, int Nq, int Nk, int H,
                 int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = (void*)o; a.lse = (float*)lse; a.Nq = Nq; a.Nk = Nk;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.hq = a.hk = a.hv = a.ho = 32; a.scale = scale;
    a.dout = dout; a.dvec = dvec; a.dq = dq; a.dk = dk; a.dv = dv; a.Rq = Nq;
    int rc = dispatch_attn(a, H, 32, 0, dtype, 1, (hipStream_t)stream);
    if (rc) return rc;
    return dispatch_attn(a, H, 32, 0, dtype, 2, (hipStream_t)stream);
}
// Correspondence attention over L stacked layers: out[l] = softmax(scale * q[l] k[l]^T) xyz.
// q [L,Nq,256], k [L,Nk,256] (dtype), xyz fp32 [Nk,3], out fp32 [L,Nq,3], lse fp32 [L,Nq].
int dreg_corr_attention_fwd(const void* q, const void* k, const float* xyz, float* out, float* lse, int L, int Nq, int Nk,
                            float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = xyz; a.o = out; a.lse = lse; a.Nq = Nq; a.Nk = Nk;
    a.ldq = a.ldk = 256; a.hq = (long)Nq * 256; a.hk = (long)Nk * 256; a.scale = scale; a.Rq = Nq;
    return dispatch_attn(a, L, 256, 1, dtype, 0, (hipStream_t)stream);
}
int dreg_corr_attention_bwd(const void* q, const void* k, const float* xyz, const float* out, const float* dout,
                            const float* lse, float* dvec, void* dq, void* dk, int L, int Nq, int Nk,
                            float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = xyz; a.o = (void*)out; a.lse = (float*)lse; a.Nq = Nq; a.Nk = Nk;
    a.ldq = a.ldk = 256; a.hq = (long)Nq * 256; a.hk = (long)Nk * 256; a.scale = scale;
    a.dout = dout; a.dvec = dvec; a.dq = dq; a.dk = dk; a.Rq = Nq;
    int rc = dispatch_attn(a, L, 256, 1, dtype, 1, (hipStream_t)stream);
    if (rc) return rc;
    return dispatch_attn(a, L, 256, 1, dtype, 2, (hipStream_t)stream);
}

// ---- variable-length batched forms: `probs` is a device array of nprob x {q_start, q_len, kv_start, kv_len} (int32) selecting
// row ranges of shared [R, ld] tensors (all pairs of a step in one launch); max_q / max_k bound the per-problem lengths.
int dreg_mha_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* probs, int nprob,
                        int max_q, int max_k, int R, int H, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.probs = (const int4*)probs; a.Rq = R;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.hq = a.hk = a.hv = a.ho = 32; a.scale = scale;
    return dispatch_attn(a, H, 32, 0, dtype, 0, (hipStream_t)stream, nprob, max_q, max_k);
}
int dreg_mha_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        float* dvec, void* dq, void* dk, void* dv, const int* probs, int nprob, int max_q, int max_k, int R, int H,
                        int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = (void*)o; a.lse = (float*)lse; a.probs = (const int4*)probs; a.Rq = R;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.hq = a.hk = a.hv = a.ho = 32; a.scale = scale;
    a.dout = dout; a.dvec = dvec; a.dq = dq; a.dk = dk; a.dv = dv;
    int rc = dispatch_attn(a, H, 32, 0, dtype, 1, (hipStream_t)stream, nprob, max_q, max_k);
    if (rc) return rc;
    return dispatch_attn(a, H, 32, 0, dtype, 2, (hipStream_t)stream, nprob, max_q, max_k);
}
// q, k: [L, R, 256]; xyz fp32 [R, 3]; out fp32 [L, R, 3]; lse fp32 [L, R]
int dreg_corr_attention_varlen_fwd(const void* q, const void* k, const float* xyz, float* out, float* lse, const int* probs, int nprob,
                                   int max_q, int max_k, int L, int R, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = xyz; a.o = out; a.lse = lse; a.probs = (const int4*)probs; a.Rq = R;
    a.ldq = a.ldk = 256; a.hq = a.hk = (long)R * 256; a.scale = scale;
    return dispatch_attn(a, L, 256, 1, dtype, 0, (hipStream_t)stream, nprob, max_q, max_k);
}
int dreg_corr_attention_varlen_bwd(const void* q, const void* k, const float* xyz, const float* out, const float* dout,
                                   const float* lse, float* dvec, void* dq, void* dk, const int* probs, int nprob,
                                   int max_q, int max_k, int L, int R, float scale, int dtype, void* stream)
{
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = xyz; a.o = (void*)out; a.lse = (float*)lse; a.probs = (const int4*)probs; a.Rq = R;
    a.ldq = a.ldk = 256; a.hq = a.hk = (long)R * 256; a.scale = scale;
    a.dout = dout; a.dvec = dvec; a.dq = dq; a.dk = dk;
    int rc = dispatch_attn(a, L, 256, 1, dtype, 1, (hipStream_t)stream, nprob, max_q, max_k);
    if (rc) return rc;
    return dispatch_attn(a, L, 256, 1, dtype, 2, (hipStream_t)stream, nprob, max_q, max_k);
}

}  // extern "C"

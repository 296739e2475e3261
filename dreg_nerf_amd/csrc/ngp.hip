// Instant-NGP dense query on gfx950: multiresolution hash-grid encoding fused with the density MLP, and the
// 18-direction colour MLP, one wavefront per 64 samples, MLP layers on fp16 MFMA (16x16x32) with fp32 accumulation.
//
// Replaces the tiny-cuda-nn calls of the reference (arithmetic lives upstream, git master, unpinned):
//   NGPradianceField.query_density   conerf/radiance_fields/ngp.py:148-176  (HashGrid L=16 F=2 T=2^19 Nmin=16 b=1.4472692
//                                    + FullyFusedMLP 32->64->16, config ngp.py:92-110)
//   NGPradianceField.query_rgb       ngp.py:178-193  (SH degree 4 + FullyFusedMLP 32->64->64->16, sigmoid, ngp.py:112-146)
//   called 1 + 18 times per grid by SampleGrid.query_radiance_and_density_from_camera (conerf/register/sample_grid.py:321-341).
// Build-side specification: table and weights are fp16 (tcnn's inference precision), trilinear
// interpolation in fp32 rounded to fp16, layer inputs fp16, accumulation fp32 (tcnn accumulates in fp16 — unpinned).
// The 18 viewing directions are constants of the caller, so the SH half of the colour net's first layer collapses to one
// bias vector per direction (c_k = W1[:, :16] sh_k, computed by the caller) and the geometry half is computed once per point.
#include "common.h"

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct NgpLevels {
    uint32_t offset[16];   // first entry of the level (entries of 2 features)
    uint32_t size[16];     // entries in the level
    uint32_t res[16];
    float scale[16];
    uint32_t hashed[16];
};

__device__ __forceinline__ uint32_t grid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t size, uint32_t hashed) {
    uint32_t idx = hashed ? (x ^ (y * 2654435761u) ^ (z * 805459861u)) : (x + y * res + z * res * res);
    return idx % size;
}

__device__ __forceinline__ f16x8_t ldsfrag(const char* base, int rs, int row, int k0) {
    return *reinterpret_cast<const f16x8_t*>(base + row * rs + k0 * 2);
}

// Every wave owns its 64 points and its own slice of LDS (sX / sH of wave w): what one layer writes is read back by the SAME wave.  LDS
// instructions of a wave execute in issue order, so no workgroup barrier is needed between the layers — only the compiler must not
// move the reads above the writes.  (Six __syncthreads per direction kept the four waves of a workgroup in lockstep.)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Hidden activations go back to LDS between the layers.  The products are formed TRANSPOSED (weights as the MFMA's first operand), so a
// lane holds four consecutive hidden units of one point: one 8-byte LDS write of four fp16 instead of four 2-byte writes (the layers
// are 16-64 MFMAs each; 64 scalar LDS writes per lane and layer were most of the kernel).  Same products, same sums: bit-identical.
__device__ __forceinline__ void store_relu4(char* sH, int row_stride, int point, int hidden, const f32x4_t& v, float b0 = 0.f, float b1 = 0.f, float b2 = 0.f, float b3 = 0.f)
{
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t h = {(_Float16)fmaxf(v[0] + b0, 0.f), (_Float16)fmaxf(v[1] + b1, 0.f), (_Float16)fmaxf(v[2] + b2, 0.f), (_Float16)fmaxf(v[3] + b3, 0.f)};
    *reinterpret_cast<f16x4_t*>(sH + point * row_stride + hidden * 2) = h;
}

// The same with the ReLU on the packed-half VALU: round the four sums to fp16 first (round-to-nearest, as the plain cast), then ONE
// v_pk_max_f16 per pair — relu(rn(x)) == rn(relu(x)), so the stored bits are those of store_relu4.  (The library is built without
// packed-fp32 VALU instructions; packed fp16 ones are not affected: DESIGN.md, co-execution fault.)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_relu4_pk(char* sH, int row_stride, int point, int hidden, const f32x4_t& v)
{
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x2_t z = {(_Float16)0.f, (_Float16)0.f};
    f16x2_t a = {(_Float16)v[0], (_Float16)v[1]}, b = {(_Float16)v[2], (_Float16)v[3]};
    a = __builtin_elementwise_max(a, z);
    b = __builtin_elementwise_max(b, z);
    const f16x4_t h = {a[0], a[1], b[0], b[1]};
    *reinterpret_cast<f16x4_t*>(sH + point * row_stride + hidden * 2) = h;
}

// ------------------------------------------------------------------------------------------------ density
// x world [Np,3] fp32 -> density fp32 [Np] (= exp(h0 - 1) * inside), raw fp16 [Np,16] (h0 | 15 geometry features)
// one wave per workgroup: the waves are independent (no workgroup barrier), and 5,081 small workgroups for a 325 k-point block fill
// the chip evenly where 1,270 four-wave ones left a quarter-full second round
// unit-cube coordinates of point p (clamped) and whether it lies strictly inside (ngp.py:157-167)
__device__ __forceinline__ bool ngp_unit_coords(const float* __restrict__ x, int p, int Np, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                                                int contract, float (&u)[3])
{
    u[0] = u[1] = u[2] = 0.f;
    bool inside = false;
    if (p < Np) {
        const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
        inside = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) u[c] = (x[(size_t)p * 3 + c] - lo[c]) / (hi[c] - lo[c]);
        if (contract) {
            // contract_to_unisphere (conerf/radiance_fields/ngp.py:41-63): the aabb maps to [-1,1]^3, points of norm > 1 are pulled
            // onto the shell (2 - 1/|x|) x/|x| of radius < 2, and [-2,2]^3 maps to [0,1]^3
            float v[3], m2 = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { v[c] = u[c] * 2.f - 1.f; m2 += v[c] * v[c]; }
            const float mag = sqrtf(m2);
            if (mag > 1.f) {
                const float sc = (2.f - 1.f / mag) / mag;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] *= sc;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c] = v[c] / 4.f + 0.5f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            inside = inside && (u[c] > 0.f) && (u[c] < 1.f);
            u[c] = fminf(fmaxf(u[c], 0.f), 1.f);
        }
    }
    return inside;
}
// trilinear interpolation of level l's two features at unit coordinates u (8 corner gathers of 4 bytes)
__device__ __forceinline__ void ngp_level_features(const _Float16* __restrict__ table, const NgpLevels& lv, int l, const float (&u)[3], float& f0, float& f1)
{
    const float sc = lv.scale[l];
    const uint32_t res = lv.res[l], size = lv.size[l], hashed = lv.hashed[l];
    const _Float16* tl = table + (size_t)lv.offset[l] * 2;
    float pos[3], w[3];
    uint32_t g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { pos[c] = u[c] * sc + 0.5f; const float fl = floorf(pos[c]); g[c] = (uint32_t)fl; w[c] = pos[c] - fl; }
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const uint32_t cx = g[0] + (corner & 1), cy = g[1] + ((corner >> 1) & 1), cz = g[2] + ((corner >> 2) & 1);
        const float wt = ((corner & 1) ? w[0] : 1.f - w[0]) * ((corner & 2) ? w[1] : 1.f - w[1]) * ((corner & 4) ? w[2] : 1.f - w[2]);
        const uint32_t idx = grid_index(cx, cy, cz, res, size, hashed);
        const uint32_t pr = *reinterpret_cast<const uint32_t*>(tl + (size_t)idx * 2);
        union { uint32_t u32; _Float16 h[2]; } cv; cv.u32 = pr;
        f0 += wt * (float)cv.h[0]; f1 += wt * (float)cv.h[1];
    }
}

// Hash-grid encoding with the table held in L2: the 16 levels are 25 MB (fits the Infinity Cache, not one XCD's 4 MB L2), and a wave
// that walks all levels pulls a 128-byte line through the fabric for almost every 4-byte corner (807 MB per 325 k-point block, 5 TB/s:
// what bounded the fused kernel).  Workgroups are dealt round-robin to the 8 XCDs, so workgroup b works for XCD b & 7 — and takes ONLY
// the two levels (b & 7) and 15 - (b & 7) (a small dense level with a hashed one: <= 3.5 MB; three XCDs hold two hashed levels, 4 MB)
// for its share of the points: every XCD's L2 keeps its own levels, the corner gathers become L2 hits.  feat: fp16 [16][Np][2].
__global__ __launch_bounds__(64) void ngp_encode_xcd_kernel(const float* __restrict__ x, const _Float16* __restrict__ table, _Float16* __restrict__ feat,
                                                             NgpLevels lv, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2, int Np,
                                                             int contract, int waves_per_xcd, const int* __restrict__ order, int x_in_slot_order)
{
    const int lane = threadIdx.x & 63;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int la = xcd, lb = 15 - xcd;
    const int nchunk = (Np + 63) / 64;
    for (int chunk = slot; chunk < nchunk; chunk += waves_per_xcd) {
        const int j = chunk * 64 + lane;                               // slot of this lane (feat is kept in slot order)
        const int p = (order && !x_in_slot_order && j < Np) ? order[j] : j;   // where this slot's coordinates are
        float u[3];
        (void)ngp_unit_coords(x, p, Np, lo0, lo1, lo2, hi0, hi1, hi2, contract, u);
        float a0, a1, b0, b1;
        ngp_level_features(table, lv, la, u, a0, a1);
        ngp_level_features(table, lv, lb, u, b0, b1);
        if (j < Np) {
            union { uint32_t u32; _Float16 h[2]; } va, vb;
            va.h[0] = (_Float16)a0; va.h[1] = (_Float16)a1; vb.h[0] = (_Float16)b0; vb.h[1] = (_Float16)b1;
            reinterpret_cast<uint32_t*>(feat)[(size_t)la * Np + j] = va.u32;
            reinterpret_cast<uint32_t*>(feat)[(size_t)lb * Np + j] = vb.u32;
        }
    }
}

// optional epilogue of the density kernel (SampleGrid.query_dense): alpha[n] = clip(1 - exp(-delta * density)), keep[n] = density > thre
// (counting the kept cells per row here with atomics doubled the launch's 17 us; dreg_grid_write_kept counts them with a wave per row)
struct NgpKeepEpi { float* alpha; uint8_t* keep; float delta, thre; };
// PRE: the level features come from ngp_encode_xcd_kernel's buffer (feat) instead of being gathered here
template <int UNR, bool PRE = false>
__global__ __launch_bounds__(64) void ngp_density_kernel(const float* __restrict__ x, const _Float16* __restrict__ table,
                                                          const _Float16* __restrict__ w1, const _Float16* __restrict__ w2,
                                                          float* __restrict__ density, _Float16* __restrict__ raw,
                                                          NgpLevels lv, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2, int Np,
                                                          int contract, const _Float16* __restrict__ feat = nullptr, const int* __restrict__ order = nullptr,
                                                          int x_in_slot_order = 0, NgpKeepEpi ke = NgpKeepEpi{})
{
    constexpr int XRS = 32 * 2 + 16, HRS = 64 * 2 + 16;
    // per wave: ONE 64 x 64 fp16 tile (the encoded input X lives in its first 5 KB until the first layer has read it) + 64 flags:
    // 9.5 KB per wave, 38 KB per workgroup -> four workgroups per CU (the gathers of the 16 levels are latency: occupancy hides them)
    __shared__ __attribute__((aligned(16))) char smem[64 * HRS + 64 * 4 + 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* sH = smem;
    char* sX = sH;
    float* sSel = reinterpret_cast<float*>(sH + 64 * HRS);
    int* sPt = reinterpret_cast<int*>(sH + 64 * HRS + 64 * 4);            // the point each slot of this wave works on
    const int p0 = (int)blockIdx.x * 64;
    const int jslot = p0 + lane;                                         // slot; the point is order[slot] (identity without an order)
    const int p = (order && jslot < Np) ? order[jslot] : jslot;
    float u[3];
    const bool inside = ngp_unit_coords(x, x_in_slot_order ? jslot : p, Np, lo0, lo1, lo2, hi0, hi1, hi2, contract, u);
    sSel[lane] = inside ? 1.f : 0.f;
    sPt[lane] = p;
    if constexpr (PRE) {
        uint32_t fv[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) fv[l] = jslot < Np ? reinterpret_cast<const uint32_t*>(feat)[(size_t)l * Np + jslot] : 0u;
        uint32_t* xr = reinterpret_cast<uint32_t*>(sX + lane * XRS);
#pragma unroll
        for (int l = 0; l < 16; ++l) xr[l] = fv[l];
    } else {
        // UNR levels at a time: 8 * UNR corner gathers of a lane in flight (measured: no difference between 1 and 8 — the kernel is bound by
        // the lines it pulls through the fabric, not by the latency of a level)
#pragma unroll UNR
        for (int l = 0; l < 16; ++l) {
            float f0, f1;
            ngp_level_features(table, lv, l, u, f0, f1);
            _Float16* xr = reinterpret_cast<_Float16*>(sX + lane * XRS);
            xr[2 * l] = (_Float16)f0; xr[2 * l + 1] = (_Float16)f1;
        }
    }
    wave_sync();
    const int fr = lane & 15, kg = lane >> 4;
    f32x4_t acc[4][4];
    {
        f16x8_t bf[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bf[cb] = *reinterpret_cast<const f16x8_t*>(w1 + (cb * 16 + fr) * 32 + kg * 8);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const f16x8_t af = ldsfrag(sX, XRS, rb * 16 + fr, kg * 8);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[cb], af, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) store_relu4(sH, HRS, rb * 16 + fr, cb * 16 + kg * 4, acc[rb][cb]);
    wave_sync();
    f16x8_t w2f[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) w2f[kb] = *reinterpret_cast<const f16x8_t*>(w2 + fr * 64 + kb * 32 + kg * 8);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ldsfrag(sH, HRS, rb * 16 + fr, kb * 32 + kg * 8), w2f[kb], o, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rb * 16 + kg * 4 + r;
            if (p0 + row < Np) {
                const _Float16 hv = (_Float16)o[r];
                const int pt = sPt[row];
                raw[(size_t)pt * 16 + fr] = hv;
                if (fr == 0) {
                    const float d = __expf((float)hv - 1.f) * sSel[row];
                    density[pt] = d;
                    if (ke.alpha) {                     // the dense query's alpha / density mask from the same launch (ngp_alpha_keep_kernel's arithmetic)
                        ke.alpha[pt] = fminf(fmaxf(1.f - __expf(-ke.delta * d), 0.f), 1.f);
                        const bool kp = d > ke.thre;
                        ke.keep[pt] = kp ? 1 : 0;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ colour, 18 directions
// raw fp16 [Np,16] (col 0 ignored, cols 1..15 geometry features) -> rgb fp32 [Np,3] = mean_k sigmoid(net(sh_k | feat | 1))
// w1 fp16 [64][32], w2 fp16 [64][64], w3 fp16 [16][64]; dirbias fp32 [ndir][64] = W1[:, :16] . fp16(sh_k)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void ngp_rgb_kernel(const _Float16* __restrict__ raw, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2,
                                                      const _Float16* __restrict__ w3, const float* __restrict__ dirbias,
                                                      float* __restrict__ rgb, int ndir, int Np, const float* __restrict__ dirs = nullptr)
{
    constexpr int XRS = 32 * 2 + 16, HRS = 64 * 2 + 16;
    // per wave ONE 64 x 64 fp16 tile: X (prologue only) aliases it, and the second layer overwrites the first layer's activations in
    // place — a 16-row block is read into registers in full before its outputs are written.  9 KB per wave instead of 23.5 KB: four
    // workgroups per CU instead of one (one wave per SIMD had nothing to hide the LDS round trips and MFMA dependency chains with).
    __shared__ __attribute__((aligned(16))) char smem[64 * HRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* sH1 = smem;
    char* sH2 = sH1;
    char* sX = sH1;
    const int p0 = (int)blockIdx.x * 64;
    const int p = p0 + lane;
    {   // X = (0 x16 | feat[1..15] | 1); with per-point directions (dirs != null, ndir == 1) the first 16 columns carry fp16(SH4(dir))
        // and the first layer is the full 32-wide product instead of "geometry half + per-direction bias"
        _Float16* xr = reinterpret_cast<_Float16*>(sX + lane * XRS);
        if (dirs && p < Np) {
            const float x = dirs[(size_t)p * 3], y = dirs[(size_t)p * 3 + 1], z = dirs[(size_t)p * 3 + 2];
            const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
            const float sh[16] = {0.28209479177387814f, -0.48860251190291987f * y, 0.48860251190291987f * z, -0.48860251190291987f * x,
                                  1.0925484305920792f * xy, -1.0925484305920792f * yz, 0.94617469575755997f * z2 - 0.31539156525251999f,
                                  -1.0925484305920792f * xz, 0.54627421529603959f * x2 - 0.54627421529603959f * y2,
                                  0.59004358992664352f * y * (-3.0f * x2 + y2), 2.8906114426405538f * xy * z,
                                  0.45704579946446572f * y * (1.0f - 5.0f * z2), 0.3731763325901154f * z * (5.0f * z2 - 3.0f),
                                  0.45704579946446572f * x * (1.0f - 5.0f * z2), 1.4453057213202769f * z * (x2 - y2),
                                  0.59004358992664352f * x * (-x2 + 3.0f * y2)};
#pragma unroll
            for (int j = 0; j < 16; ++j) xr[j] = (_Float16)sh[j];
        } else
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = (_Float16)0.f;
#pragma unroll
        for (int j = 0; j < 15; ++j) xr[16 + j] = p < Np ? raw[(size_t)p * 16 + 1 + j] : (_Float16)0.f;
        xr[31] = (_Float16)1.f;
    }
    wave_sync();
    const int fr = lane & 15, kg = lane >> 4;
    f32x4_t base[4][4];
    {
        f16x8_t bf[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bf[cb] = *reinterpret_cast<const f16x8_t*>(w1 + (cb * 16 + fr) * 32 + kg * 8);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const f16x8_t af = ldsfrag(sX, XRS, rb * 16 + fr, kg * 8);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) base[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[cb], af, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
    }
    f16x8_t w3f[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) w3f[kb] = *reinterpret_cast<const f16x8_t*>(w3 + fr * 64 + kb * 32 + kg * 8);
    // Round 3: the loop was VALU-bound (~480 VALU instructions against 40 MFMAs per direction).  Three changes, same arithmetic:
    //  * the direction's bias joins on the MATRIX pipe:  base + c_k 1^T  is one more MFMA per tile whose first operand holds the bias
    //    split into two halves (c = hi + lo, both fp16: exact to 2^-22) in its K columns 0 and 1 and whose second operand is ones there
    //    — 16 MFMAs instead of 128 v_add_f32 per direction;
    //  * ReLU on the packed-half VALU after the fp16 rounding (store_relu4_pk);
    //  * the sigmoid on DENSE lanes: the output tile has its 3 colour channels in 3 of every 16 lanes, so the 64 x 3 pre-activations
    //    go through a 1 KB LDS patch and every lane finishes the three channels of ONE point (27 instead of 144 VALU per direction).
    f16x8_t ones = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    if (kg == 0) { ones[0] = (_Float16)1.f; ones[1] = (_Float16)1.f; }
    __shared__ __attribute__((aligned(16))) float sO[64 * 4];
    float sum3[3] = {0.f, 0.f, 0.f};

#pragma unroll 1
    for (int k = 0; k < ndir; ++k) {
        f16x8_t bfr[4];      // first operand of the bias product for this lane's row (hidden unit cb * 16 + fr): (hi, lo, 0, ...) in the kg == 0 lanes
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const float c = (dirs || kg != 0) ? 0.f : dirbias[k * 64 + cb * 16 + fr];
            const _Float16 hi = (_Float16)c, lo = (_Float16)(c - (float)hi);
            bfr[cb] = (f16x8_t){hi, lo, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        }
        wave_sync();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                store_relu4_pk(sH1, HRS, rb * 16 + fr, cb * 16 + kg * 4, __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[cb], ones, base[rb][cb], 0, 0, 0));
        wave_sync();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f16x8_t af[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) af[kb] = ldsfrag(sH1, HRS, rb * 16 + fr, kb * 32 + kg * 8);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4_t h = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    h = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8_t*>(w2 + (cb * 16 + fr) * 64 + kb * 32 + kg * 8), af[kb], h, 0, 0, 0);
                store_relu4_pk(sH2, HRS, rb * 16 + fr, cb * 16 + kg * 4, h);
            }
        }
        wave_sync();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ldsfrag(sH2, HRS, rb * 16 + fr, kb * 32 + kg * 8), w3f[kb], o, 0, 0, 0);
            if (fr < 4) {        // channel fr of points rb * 16 + kg * 4 + r (channel 3 is padding: written, never read)
#pragma unroll
                for (int r = 0; r < 4; ++r) sO[(rb * 16 + kg * 4 + r) * 4 + fr] = o[r];
            }
        }
        wave_sync();
        const float4 pre = *reinterpret_cast<const float4*>(sO + lane * 4);      // this lane's point: pre-activations of its three channels
        const float pv[3] = {pre.x, pre.y, pre.z};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float hv = (float)(_Float16)pv[ch];
            // v_rcp_f32 (1 ulp) instead of the 12-instruction IEEE division: the result is rounded to fp16 right after
            sum3[ch] += (float)(_Float16)__builtin_amdgcn_rcpf(1.f + __expf(-hv));
        }
    }
    if (p < Np) {
        const float inv = 1.f / (float)ndir;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rgb[(size_t)p * 3 + ch] = sum3[ch] * inv;
    }
}

// ------------------------------------------------------------------------------------------------ colour, K shared directions, round 3
// The same network as ngp_rgb_kernel for the shared-direction case (dirs == nullptr), restructured around what bounded it:
//  * 64-point waves left a 38 % tail (5,078 waves of ~55 us on 4,096 wave slots) and spilled 16 of their 64 fp32 first-layer tiles;
//    here a wave owns 16-point chunks (ONE 16 x 64 tile: 16 accumulator registers) and walks  chunk = block, block + grid, ...  with
//    grid = the chip's wave slots: 20,313 chunks of a 325 k-point block are 4.96 per wave — no tail, no spills;
//  * every weight fragment (w1: 4, w2: 8, w3: 2 MFMA operands) stays in registers for the whole kernel (the 64-point form re-read
//    w2 from L1 for every tile and direction: 32 loads per direction);
//  * the direction's bias joins on the matrix pipe (hi + lo halves against a ones operand: packed by ngp_dir_bias_kernel), ReLU on the
//    packed-half VALU, the sigmoid on dense lanes (lane = (point, channel) of the chunk): ~50 VALU per chunk and direction.
// biaspk: uint32 [ndir][64] = fp16 (hi | lo << 16) of c_k[j];  same sums in the same order as ngp_rgb_kernel up to the bias split (2^-22).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void ngp_rgb_chunks_kernel(
    const _Float16* __restrict__ raw, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2, const _Float16* __restrict__ w3,
    const uint32_t* __restrict__ biaspk, float* __restrict__ rgb, int ndir, int Np)
{
    constexpr int XRS = 32 * 2 + 16, HRS = 64 * 2 + 16;
    __shared__ __attribute__((aligned(16))) char sH[16 * HRS];      // one 16 x 64 fp16 tile (X aliases its first rows)
    __shared__ __attribute__((aligned(16))) float sO[16 * 4];
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, kg = lane >> 4;
    f16x8_t w1f[4], w2f[4][2], w3f[2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        w1f[cb] = *reinterpret_cast<const f16x8_t*>(w1 + (cb * 16 + fr) * 32 + kg * 8);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) w2f[cb][kb] = *reinterpret_cast<const f16x8_t*>(w2 + (cb * 16 + fr) * 64 + kb * 32 + kg * 8);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) w3f[kb] = *reinterpret_cast<const f16x8_t*>(w3 + fr * 64 + kb * 32 + kg * 8);
    f16x8_t ones = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    if (kg == 0) { ones[0] = (_Float16)1.f; ones[1] = (_Float16)1.f; }
    const int nchunks = (Np + 15) / 16;
    const float inv = 1.f / (float)ndir;
    const int opt = lane / 3, och = lane - opt * 3;                  // dense output role: lanes 0..47 = (point, channel) of the chunk
#pragma unroll 1
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int p0 = chunk * 16;
        wave_sync();
        {   // X = (0 x16 | feat[1..15] | 1): lane (point fr, quarter kg) writes columns kg*4 .. +3 of both halves
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
            const int p = p0 + fr;
            f16x4_t f = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (p < Np) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kg * 4 + e < 15) f[e] = raw[(size_t)p * 16 + 1 + kg * 4 + e];
            }
            if (kg == 3) f[3] = (_Float16)1.f;
            *reinterpret_cast<f16x4_t*>(sH + fr * XRS + kg * 8) = (f16x4_t){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<f16x4_t*>(sH + fr * XRS + 32 + kg * 8) = f;
        }
        wave_sync();
        f32x4_t base[4];
        {
            const f16x8_t af = ldsfrag(sH, XRS, fr, kg * 8);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) base[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[cb], af, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        float sum = 0.f;
#pragma unroll 1
        for (int k = 0; k < ndir; ++k) {
            f16x8_t bfr[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const uint32_t pk = kg == 0 ? biaspk[k * 64 + cb * 16 + fr] : 0u;
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                bfr[cb] = __builtin_bit_cast(f16x8_t, (u32x4_t){pk, 0u, 0u, 0u});
            }
            wave_sync();
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                store_relu4_pk(sH, HRS, fr, cb * 16 + kg * 4, __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[cb], ones, base[cb], 0, 0, 0));
            wave_sync();
            f16x8_t af[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) af[kb] = ldsfrag(sH, HRS, fr, kb * 32 + kg * 8);
            f32x4_t h[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                h[cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) h[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[cb][kb], af[kb], h[cb], 0, 0, 0);
            }
            wave_sync();                                  // every lane has read its layer-1 fragments: the tile may be overwritten
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) store_relu4_pk(sH, HRS, fr, cb * 16 + kg * 4, h[cb]);
            wave_sync();
            f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ldsfrag(sH, HRS, fr, kb * 32 + kg * 8), w3f[kb], o, 0, 0, 0);
            if (fr < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sO[(kg * 4 + r) * 4 + fr] = o[r];
            }
            wave_sync();
            const float hv = (float)(_Float16)sO[(lane < 48 ? opt : 0) * 4 + och];
            sum += (float)(_Float16)__builtin_amdgcn_rcpf(1.f + __expf(-hv));
        }
        if (lane < 48 && p0 + opt < Np) rgb[(size_t)p0 * 3 + lane] = sum * inv;
    }
}

// grid[idx[n]] = (xyz, rgb, alpha) for kept points  (eval_ngp_nerf.py:397-405)
__global__ void grid_scatter7_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb, const float* __restrict__ alpha,
                                     const int64_t* __restrict__ idx, const uint8_t* __restrict__ keep, float* __restrict__ grid, int Np)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np || !keep[n]) return;
    float* g = grid + (size_t)idx[n] * 7;
    g[0] = xyz[(size_t)n * 3]; g[1] = xyz[(size_t)n * 3 + 1]; g[2] = xyz[(size_t)n * 3 + 2];
    g[3] = rgb[(size_t)n * 3]; g[4] = rgb[(size_t)n * 3 + 1]; g[5] = rgb[(size_t)n * 3 + 2];
    g[6] = alpha[n];
}

// occupied-cell sample positions: world = lo + (cell + jitter) / res * (hi - lo)  (sample_grid.py:226-242, AABB contraction)
// order / world_slot (optional, together): thread j takes point order[j] and ALSO writes its position to world_slot[j] — the coordinates
// in the order the density query's lanes take them (contiguous reads there instead of eight scattered passes)
__global__ void grid_sample_points_kernel(const int64_t* __restrict__ idx, const float* __restrict__ jitter, float* __restrict__ world,
                                          int rx, int ry, int rz, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2, int Np,
                                          const int* __restrict__ order, float* __restrict__ world_slot)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    const int n = order ? order[j] : j;
    const int64_t f = idx[n];
    const int z = (int)(f % rz), y = (int)((f / rz) % ry), xx = (int)(f / ((int64_t)rz * ry));
    const float u0 = ((float)xx + jitter[(size_t)n * 3]) / (float)rx, u1 = ((float)y + jitter[(size_t)n * 3 + 1]) / (float)ry,
                u2 = ((float)z + jitter[(size_t)n * 3 + 2]) / (float)rz;
    const float w0 = u0 * (hi0 - lo0) + lo0, w1 = u1 * (hi1 - lo1) + lo1, w2 = u2 * (hi2 - lo2) + lo2;
    world[(size_t)n * 3] = w0; world[(size_t)n * 3 + 1] = w1; world[(size_t)n * 3 + 2] = w2;
    if (world_slot) { world_slot[(size_t)j * 3] = w0; world_slot[(size_t)j * 3 + 1] = w1; world_slot[(size_t)j * 3 + 2] = w2; }
}

// ---- lane order of a block's dense query.  The occupied cells arrive as ascending flat indices (x * ry + y) * rz + z: consecutive points
// differ in z — the SLOWEST axis of the hash grid's tables (dense levels: x + y * res + z * res^2; hashed: x ^ y * P1 ^ z * P2), so the
// 64 corner addresses of a gather instruction fall into 64 different cache lines and the kernel is bound by the texture unit's line
// rate.  With the lanes of a wave running along x the same gathers touch a few lines (155 -> 112 us for 325 k points; with the XCD-
// resident level pairs 86).  order[j] = index (in the ascending list) of the j-th occupied cell in (z, y, x)-major enumeration.
// Three small launches per block: column prefix counts along x, an exclusive scan over the ry * rz columns, one pass over the points.
// pre[f] = occupied cells with smaller x in f's (y, z) column, cnt[z][y] = the column's total.  A workgroup = NS x-segments of 16 cells
// x COLS consecutive columns: every thread has its 16 loads in flight at once (a thread walking a whole column is a chain of rx
// dependent steps: 3x the time of the density kernel's gain), the segments are joined through LDS.
__device__ __forceinline__ void grid_xprefix(const uint8_t* __restrict__ binary, uint16_t* __restrict__ pre, int* __restrict__ cnt, int rx, int ry, int rz,
                                             int NS, int COLS, int block)
{
    __shared__ int seg[256];
    const int t = threadIdx.x, s_ = t / COLS, c = t - s_ * COLS;          // segment, column within the workgroup (columns fastest: coalesced)
    const int col = block * COLS + c;                                      // column (y, z), z fastest
    const bool live = s_ < NS && col < ry * rz;
    const size_t plane = (size_t)ry * rz;
    uint8_t b[16];
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int x = s_ * 16 + i; b[i] = (live && x < rx) ? binary[(size_t)col + (size_t)x * plane] : 0; }
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += b[i] ? 1 : 0;
    seg[t] = live ? tot : 0;
    __syncthreads();
    int run = 0;
    for (int q = 0; q < s_; ++q) run += seg[q * COLS + c];
    if (live) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int x = s_ * 16 + i;
            if (x < rx) { pre[(size_t)col + (size_t)x * plane] = (uint16_t)run; run += b[i] ? 1 : 0; }
        }
        if (s_ == NS - 1) { const int y = col / rz, z = col - y * rz; cnt[z * ry + y] = run; }   // columns in (z, y) order: the order of the enumeration
    }
}
__global__ __launch_bounds__(256) void grid_xprefix_kernel(const uint8_t* __restrict__ binary, uint16_t* __restrict__ pre, int* __restrict__ cnt, int rx, int ry, int rz,
                                                            int NS, int COLS)
{
    grid_xprefix(binary, pre, cnt, rx, ry, rz, NS, COLS, (int)blockIdx.x);
}
// exclusive scan in place, one workgroup: rounds of 1024 x 16 entries — a thread's 16 values are loaded at once, scanned in registers,
// the threads' totals by wave shuffles + one LDS exchange (a loop of dependent loads / block-wide sync steps took 25 us for 16 k entries)
__global__ __launch_bounds__(1024) void grid_colscan_kernel(int* __restrict__ cnt, int n)
{
    __shared__ int wtot[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int carry = 0;
    for (int r0 = 0; r0 < n; r0 += 1024 * 16) {
        const int lo = r0 + t * 16;
        int v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lo + i < n ? cnt[lo + i] : 0;
        int s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = v[i]; v[i] = s; s += c; }        // exclusive within the thread
        int inc = s;                                                                     // inclusive scan of the threads' totals within the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; ++q) wbase += wtot[q];
        const int base = carry + wbase + inc - s;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (lo + i < n) cnt[lo + i] = base + v[i];
        if (t == 1023) carry_s = base + s;
        __syncthreads();
        carry = carry_s;
    }
}
__global__ void grid_xorder_kernel(const int64_t* __restrict__ idx, const uint16_t* __restrict__ pre, const int* __restrict__ base, int* __restrict__ order,
                                   int ry, int rz, int Np)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np) return;
    const int64_t f = idx[n];
    const int z = (int)(f % rz), y = (int)((f / rz) % ry);
    order[base[z * ry + y] + pre[f]] = n;
}

template <typename T> __global__ void f32_to_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (_Float16)in[i];
}

DREG_KNOB(int, g_ngp_density_unroll, 1);   // tuning (include/dreg_nerf_probe.h): hash-grid levels whose corner gathers are in flight together (1, 2, 4, 8)
static void ngp_density_launch(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw, const NgpLevels& lv,
                               const float* aabb, int Np, int contract, void* stream, const int* order = nullptr, int xslot = 0, NgpKeepEpi ke = NgpKeepEpi{})
{
#define NGP_D(U) hipLaunchKernelGGL(ngp_density_kernel<U>, dim3((Np + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, (const _Float16*)table, \
                                    (const _Float16*)w1, (const _Float16*)w2, density, (_Float16*)raw, lv, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], Np, contract, nullptr, order, xslot, ke)
    if (g_ngp_density_unroll == 8) NGP_D(8); else if (g_ngp_density_unroll == 4) NGP_D(4); else if (g_ngp_density_unroll == 2) NGP_D(2); else NGP_D(1);
#undef NGP_D
}
DREG_KNOB(int, g_ngp_xcd_levels, 1);       // tuning (include/dreg_nerf_probe.h): with a workspace, encode per XCD-resident level pair first (ngp_encode_xcd_kernel)
// two launches: the level features of all points into `feat` (fp16 [16][Np][2], XCD-partitioned levels), then the density MLP over them
static void ngp_density_launch_xcd(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw, const NgpLevels& lv,
                                   const float* aabb, int Np, int contract, void* feat, void* stream, const int* order, int xslot, NgpKeepEpi ke = NgpKeepEpi{})
{
    const int nchunk = (Np + 63) / 64;
    int wpx = 512;                      // waves per XCD: 32 CUs x 16 resident one-wave workgroups
    if (wpx > nchunk) wpx = nchunk;
    hipLaunchKernelGGL(ngp_encode_xcd_kernel, dim3(8 * wpx), dim3(64), 0, (hipStream_t)stream, x, (const _Float16*)table, (_Float16*)feat, lv,
                       aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], Np, contract, wpx, order, xslot);
    hipLaunchKernelGGL((ngp_density_kernel<1, true>), dim3(nchunk), dim3(64), 0, (hipStream_t)stream, x, (const _Float16*)table,
                       (const _Float16*)w1, (const _Float16*)w2, density, (_Float16*)raw, lv, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], Np, contract,
                       (const _Float16*)feat, order, xslot, ke);
}
DREG_KNOB(int, g_ngp_rgb_chunks, 1);     // tuning (include/dreg_nerf_probe.h): shared-direction colour queries run the 16-point-chunk kernel (0: the 64-point kernel)

extern "C" {

// level table of the HashGrid (n_levels 16, base 16, per_level_scale b, log2_hashmap_size): fills 5 x 16 host arrays,
// returns the total number of table entries (of 2 features).
uint32_t dreg_ngp_level_table(float per_level_scale, int log2_hashmap_size, int base_resolution,
                              uint32_t* offset, uint32_t* size, uint32_t* res, float* scale, uint32_t* hashed)
{
    uint32_t off = 0;
    const uint32_t cap = 1u << log2_hashmap_size;
    for (int l = 0; l < 16; ++l) {
        const float sc = exp2f((float)l * log2f(per_level_scale)) * (float)base_resolution - 1.0f;
        const uint32_t r = (uint32_t)ceilf(sc) + 1u;
        uint64_t n = (uint64_t)r * r * r;
        n = (n + 7) / 8 * 8;
        const uint32_t sz = n > cap ? cap : (uint32_t)n;
        offset[l] = off; size[l] = sz; res[l] = r; scale[l] = sc;
        hashed[l] = ((uint64_t)r * r * r > sz) ? 1u : 0u;
        off += sz;
    }
    return off;
}

int dreg_f32_to_f16(const float* in, void* out, size_t n, void* stream)
{
    if (n == 0) return DREG_OK;
    size_t b = (n + 255) / 256; if (b > 4096) b = 4096;
    hipLaunchKernelGGL(f32_to_f16_kernel<float>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, in, (_Float16*)out, n);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// x fp32 [Np,3] world; table fp16 [entries,2]; w1 fp16 [64,32]; w2 fp16 [16,64]; levels: 5 x 16 arrays as produced by
// dreg_ngp_level_table; aabb = (lo xyz, hi xyz).  Outputs density fp32 [Np], raw fp16 [Np,16].
int dreg_ngp_density_fwd(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                         const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                         const float* aabb, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    NgpLevels lv;
    for (int l = 0; l < 16; ++l) { lv.offset[l] = offset[l]; lv.size[l] = size[l]; lv.res[l] = res[l]; lv.scale[l] = scale[l]; lv.hashed[l] = hashed[l]; }
    ngp_density_launch(x, table, w1, w2, density, raw, lv, aabb, Np, 0, stream);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// the same for an unbounded scene (NGPradianceField(unbounded=True), conerf/radiance_fields/ngp.py:41-63,163-167): positions go through
// the unisphere contraction of the aabb before the hash grid; contract = 0 is dreg_ngp_density_fwd.
int dreg_ngp_density_fwd_contract(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                                  const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                  const float* aabb, int Np, int contract, void* stream)
{
    if (Np == 0) return DREG_OK;
    NgpLevels lv;
    for (int l = 0; l < 16; ++l) { lv.offset[l] = offset[l]; lv.size[l] = size[l]; lv.res[l] = res[l]; lv.scale[l] = scale[l]; lv.hashed[l] = hashed[l]; }
    ngp_density_launch(x, table, w1, w2, density, raw, lv, aabb, Np, contract, stream);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// order (optional, device int32 [Np], a permutation of 0..Np-1): slot j of the launch works on point order[j] — outputs stay where the
// point is; what changes is which points share a wave (dreg_grid_x_order: lanes along the tables' fastest axis).
// The same with a caller-owned scratch buffer of dreg_ngp_density_workspace_bytes(Np) bytes: the hash-grid encoding then runs as its own
// launch with every XCD working on two levels that stay in its L2 (ngp_encode_xcd_kernel), the MLP as a second one.  Same outputs, bit
// for bit.  workspace = null (or too small): the fused kernel.
size_t dreg_ngp_density_workspace_bytes(int Np) { return Np > 0 ? (size_t)16 * Np * 2 * sizeof(_Float16) : 0; }
int dreg_ngp_density_fwd_ws(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                            const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                            const float* aabb, int Np, int contract, void* workspace, size_t workspace_bytes, const int* order, int x_in_slot_order, void* stream)
{
    if (Np == 0) return DREG_OK;
    NgpLevels lv;
    for (int l = 0; l < 16; ++l) { lv.offset[l] = offset[l]; lv.size[l] = size[l]; lv.res[l] = res[l]; lv.scale[l] = scale[l]; lv.hashed[l] = hashed[l]; }
    if (g_ngp_xcd_levels && workspace && workspace_bytes >= dreg_ngp_density_workspace_bytes(Np)) ngp_density_launch_xcd(x, table, w1, w2, density, raw, lv, aabb, Np, contract, workspace, stream, order, order ? x_in_slot_order : 0);
    else ngp_density_launch(x, table, w1, w2, density, raw, lv, aabb, Np, contract, stream, order, order ? x_in_slot_order : 0);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
#ifdef DREG_PROBE
void dreg_ngp_set_xcd_levels(int on) { g_ngp_xcd_levels = on ? 1 : 0; }
#endif
// colour for ONE viewing direction per point (NGPradianceField.query_rgb(dir, embedding) / forward(positions, directions),
// conerf/radiance_fields/ngp.py:178-208): dirs fp32 [Np,3] as passed to query_rgb, raw fp16 [Np,16] -> rgb fp32 [Np,3].
int dreg_ngp_rgb_dir_fwd(const void* raw, const void* w1, const void* w2, const void* w3, const float* dirs, float* rgb, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    hipLaunchKernelGGL(ngp_rgb_kernel, dim3((Np + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const _Float16*)raw,
                       (const _Float16*)w1, (const _Float16*)w2, (const _Float16*)w3, (const float*)nullptr, rgb, 1, Np, dirs);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// raw fp16 [Np,16] -> rgb fp32 [Np,3], mean over ndir directions; dirbias fp32 [ndir,64] on the device.
// c[k][j] = sum_i fp16(W1[j][i]) * fp16(SH4(dir_k)[i]), i < 16: the direction half of the colour net's first layer for K shared viewing
// directions (sample_grid.py:332-337 evaluates the field from 18 of them) — one launch instead of ~30 elementwise torch launches.
__global__ void ngp_dir_bias_kernel(const float* __restrict__ dirs, const _Float16* __restrict__ w1, float* __restrict__ out, int K)
{
    const int k = blockIdx.x, j = threadIdx.x;       // 64 threads: one output unit each
    if (k >= K || j >= 64) return;
    const float x = dirs[k * 3], y = dirs[k * 3 + 1], z = dirs[k * 3 + 2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float sh[16] = {0.28209479177387814f, -0.48860251190291987f * y, 0.48860251190291987f * z, -0.48860251190291987f * x,
                          1.0925484305920792f * xy, -1.0925484305920792f * yz, 0.94617469575755997f * z2 - 0.31539156525251999f,
                          -1.0925484305920792f * xz, 0.54627421529603959f * x2 - 0.54627421529603959f * y2,
                          0.59004358992664352f * y * (-3.0f * x2 + y2), 2.8906114426405538f * xy * z,
                          0.45704579946446572f * y * (1.0f - 5.0f * z2), 0.3731763325901154f * z * (5.0f * z2 - 3.0f),
                          0.45704579946446572f * x * (1.0f - 5.0f * z2), 1.4453057213202769f * z * (x2 - y2),
                          0.59004358992664352f * x * (-x2 + 3.0f * y2)};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += (float)w1[j * 32 + i] * (float)(_Float16)sh[i];
    out[k * 64 + j] = acc;
    // the same value split into two fp16 halves (hi = rn(c), lo = rn(c - hi)), packed: the operand ngp_rgb_chunks_kernel feeds to the MFMA
    const _Float16 hi = (_Float16)acc, lo = (_Float16)(acc - (float)hi);
    reinterpret_cast<uint32_t*>(out + (size_t)K * 64)[k * 64 + j] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
}
int dreg_ngp_dir_bias(const float* dirs, const void* w1, float* out, int K, void* stream)
{
    if (K <= 0) return DREG_OK;
    hipLaunchKernelGGL(ngp_dir_bias_kernel, dim3(K), dim3(64), 0, (hipStream_t)stream, dirs, (const _Float16*)w1, out, K);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// alpha[n] = clip(1 - exp(-delta * density[n]), 0, 1), keep[n] = density[n] > threshold (sample_grid.py:338-341) in one pass
__global__ void ngp_alpha_keep_kernel(const float* __restrict__ density, float* __restrict__ alpha, uint8_t* __restrict__ keep, int N, float delta, float thre)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float d = density[n];
    alpha[n] = fminf(fmaxf(1.f - __expf(-delta * d), 0.f), 1.f);
    keep[n] = d > thre ? 1 : 0;
}
int dreg_ngp_alpha_keep(const float* density, float* alpha, uint8_t* keep, int N, float delta, float threshold, void* stream)
{
    if (N <= 0) return DREG_OK;
    hipLaunchKernelGGL(ngp_alpha_keep_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, density, alpha, keep, N, delta, threshold);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_ngp_rgb_mean_fwd(const void* raw, const void* w1, const void* w2, const void* w3, const float* dirbias, float* rgb,
                          int ndir, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    // dirbias: fp32 [ndir][64] followed by the packed fp16 halves uint32 [ndir][64] (dreg_ngp_dir_bias writes both: 2 * ndir * 64 words)
    if (g_ngp_rgb_chunks) {
        const int nchunks = (Np + 15) / 16;
        const int slots = 256 * 16;                      // wave slots of the chip at four waves per SIMD
        hipLaunchKernelGGL(ngp_rgb_chunks_kernel, dim3(nchunks < slots ? nchunks : slots), dim3(64), 0, (hipStream_t)stream, (const _Float16*)raw,
                           (const _Float16*)w1, (const _Float16*)w2, (const _Float16*)w3, reinterpret_cast<const uint32_t*>(dirbias + (size_t)ndir * 64), rgb, ndir, Np);
    } else
    hipLaunchKernelGGL(ngp_rgb_kernel, dim3((Np + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const _Float16*)raw,
                       (const _Float16*)w1, (const _Float16*)w2, (const _Float16*)w3, dirbias, rgb, ndir, Np);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
#ifdef DREG_PROBE
void dreg_ngp_set_rgb_chunks(int on) { g_ngp_rgb_chunks = on ? 1 : 0; }
void dreg_ngp_set_density_unroll(int n) { g_ngp_density_unroll = (n == 2 || n == 4 || n == 8) ? n : 1; }
#endif

int dreg_grid_scatter7(const float* xyz, const float* rgb, const float* alpha, const int64_t* idx, const uint8_t* keep,
                       float* grid, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    hipLaunchKernelGGL(grid_scatter7_kernel, dim3((Np + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz, rgb, alpha, idx, keep, grid, Np);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// order[j] = position in idx (ascending flat indices (x * ry + y) * rz + z of the occupied cells of `binary`, byte [rx][ry][rz]) of the
// j-th occupied cell in (z, y, x)-major enumeration, i.e. with x running fastest.  workspace: dreg_grid_x_order_workspace_bytes bytes.
size_t dreg_grid_x_order_workspace_bytes(int rx, int ry, int rz) { return (size_t)rx * ry * rz * sizeof(uint16_t) + (size_t)ry * rz * sizeof(int) + 256; }
int dreg_grid_x_order(const uint8_t* binary, const int64_t* idx, int* order, void* workspace, size_t workspace_bytes, int rx, int ry, int rz, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    if (!binary || !idx || !order || !workspace || workspace_bytes < dreg_grid_x_order_workspace_bytes(rx, ry, rz) || rx > 65535) return DREG_EINVAL;
    uint16_t* pre = (uint16_t*)workspace;
    int* cnt = (int*)((char*)workspace + (((size_t)rx * ry * rz * sizeof(uint16_t) + 255) / 256) * 256);
    hipStream_t st = (hipStream_t)stream;
    const int NS = (rx + 15) / 16;                  // x-segments per column
    if (NS > 256) return DREG_EINVAL;
    int COLS = 1;
    while (COLS * 2 * NS <= 256) COLS *= 2;
    hipLaunchKernelGGL(grid_xprefix_kernel, dim3((ry * rz + COLS - 1) / COLS), dim3(256), 0, st, binary, pre, cnt, rx, ry, rz, NS, COLS);
    hipLaunchKernelGGL(grid_colscan_kernel, dim3(1), dim3(1024), 0, st, cnt, ry * rz);
    hipLaunchKernelGGL(grid_xorder_kernel, dim3((Np + 255) / 256), dim3(256), 0, st, idx, pre, cnt, order, ry, rz, Np);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_grid_sample_points(const int64_t* idx, const float* jitter, float* world, int rx, int ry, int rz, const float* aabb,
                            int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    hipLaunchKernelGGL(grid_sample_points_kernel, dim3((Np + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, jitter, world,
                       rx, ry, rz, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], Np, (const int*)nullptr, (float*)nullptr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// the same, and world_slot[j] = world[order[j]] (the positions in the lane order of dreg_grid_x_order) from the same launch
int dreg_grid_sample_points_ordered(const int64_t* idx, const float* jitter, const int* order, float* world, float* world_slot,
                                    int rx, int ry, int rz, const float* aabb, int Np, void* stream)
{
    if (Np == 0) return DREG_OK;
    if (!order || !world_slot) return DREG_EINVAL;
    hipLaunchKernelGGL(grid_sample_points_kernel, dim3((Np + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, jitter, world,
                       rx, ry, rz, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], Np, order, world_slot);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ the dense query's cell lists in three launches
// SampleGrid.query_dense needs, from the block's occupancy volume `binary` [rx][ry][rz]: the ascending list of occupied flat indices
// (sample_grid.py:223-228: torch.nonzero), the jittered world position of each (:229-242), and — for the hash grid's sake — the x-fastest
// lane order (dreg_grid_x_order).  torch.nonzero alone is five launches and a host sync; here
//   1. grid_occupied_count_kernel: the x-prefix counts of dreg_grid_x_order PLUS, in extra workgroups, the occupied cells of every (x, y)
//      row (a row = rz consecutive bytes) and a cleared kept-cells-per-row table;
//   2. grid_scan2_kernel: workgroup 0 scans the (z, y) column counts, workgroup 1 the (x, y) row counts and leaves the total N;
//      (the host reads N: the one sync, needed to size the outputs)
//   3. grid_occupied_build_kernel: a wave per row ranks its occupied cells by ballot: n = row base + rank (ascending order), slot j =
//      column base + x-prefix; writes indices[n], order[j], world[n], world_slot[j] with grid_sample_points_kernel's arithmetic.
// After the query, dreg_grid_write_kept turns keep[] into voxel_mask (ascending kept indices) and voxel_grid the same way: the density
// kernel's epilogue counted the kept cells per row, one scan gives the rows' offsets, a wave per row ranks its kept points.
__device__ __forceinline__ void grid_rowcount(const uint8_t* __restrict__ binary, int* __restrict__ rowcnt, int nrows, int rz, int block)
{
    const int r = block * 256 + threadIdx.x;
    if (r >= nrows) return;
    const uint8_t* b = binary + (size_t)r * rz;
    int c = 0;
    int z = 0;
    if ((rz & 15) == 0 && (((uintptr_t)b) & 15) == 0) {
        for (; z < rz; z += 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(b + z);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)         // non-zero bytes of the word: bit 7 of (low seven bits + 0x7f) | byte
                c += __popc((((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k]) & 0x80808080u);
        }
    } else {
        for (; z < rz; ++z) c += b[z] ? 1 : 0;
    }
    rowcnt[r] = c;
}
// one launch: workgroups [0, nprefix) are grid_xprefix_kernel's, the rest count the rows
__global__ __launch_bounds__(256) void grid_occupied_count_kernel(const uint8_t* __restrict__ binary, uint16_t* __restrict__ pre, int* __restrict__ colcnt,
                                                                   int* __restrict__ rowcnt, int rx, int ry, int rz, int NS, int COLS, int nprefix)
{
    if ((int)blockIdx.x < nprefix) grid_xprefix(binary, pre, colcnt, rx, ry, rz, NS, COLS, (int)blockIdx.x);
    else grid_rowcount(binary, rowcnt, rx * ry, rz, (int)blockIdx.x - nprefix);
}
// kept points per (x, y) row: a wave per row over keep[rowbase[r] .. rowbase[r + 1])
__global__ __launch_bounds__(256) void grid_keeprow_kernel(const int* __restrict__ rowbase, const uint8_t* __restrict__ keep, int* __restrict__ keeprow, int nrows, int Np)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int nb = rowbase[r], ne = r + 1 < nrows ? rowbase[r + 1] : Np;
    int c = 0;
    for (int q = nb; q < ne; q += 64) c += __popcll(__ballot(q + lane < ne && keep[q + lane] != 0));
    if (lane == 0) keeprow[r] = c;
}
// exclusive scans in place, one workgroup each: blockIdx 0 -> (a0, n0), blockIdx 1 -> (a1, n1); totals[blockIdx] = the array's sum
__global__ __launch_bounds__(1024) void grid_scan2_kernel(int* __restrict__ a0, int n0, int* __restrict__ a1, int n1, int* __restrict__ totals)
{
    __shared__ int wtot[16];
    __shared__ int carry_s;
    int* cnt = blockIdx.x == 0 ? a0 : a1;
    const int n = blockIdx.x == 0 ? n0 : n1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int carry = 0;
    for (int r0 = 0; r0 < n; r0 += 1024 * 16) {
        const int lo = r0 + t * 16;
        int v[16];
        const bool whole = lo + 16 <= n && ((uintptr_t)(cnt + lo) & 15) == 0;      // a thread's 16 entries as four 16-byte loads (scalar loads at a
        if (whole) {                                                                 // 64-byte lane stride are 16 x 64 cache-line requests per wave)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 q = *reinterpret_cast<const int4*>(cnt + lo + 4 * k);
                v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lo + i < n ? cnt[lo + i] : 0;
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = v[i]; v[i] = s; s += c; }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; ++q) wbase += wtot[q];
        const int base = carry + wbase + inc - s;
        if (whole) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<int4*>(cnt + lo + 4 * k) = make_int4(base + v[4 * k], base + v[4 * k + 1], base + v[4 * k + 2], base + v[4 * k + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) if (lo + i < n) cnt[lo + i] = base + v[i];
        }
        if (t == 1023) carry_s = base + s;
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
    if (t == 0 && totals) totals[blockIdx.x] = carry;
}
__global__ __launch_bounds__(256) void grid_occupied_build_kernel(const uint8_t* __restrict__ binary, const uint16_t* __restrict__ pre, const int* __restrict__ colbase,
                                                                   const int* __restrict__ rowbase, const float* __restrict__ jitter,
                                                                   int64_t* __restrict__ indices, int* __restrict__ order, float* __restrict__ world,
                                                                   float* __restrict__ world_slot, int rx, int ry, int rz, int N,
                                                                   float lo0, float lo1, float lo2, float hi0, float hi1, float hi2)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);            // row (x, y)
    if (r >= rx * ry) return;
    const int xx = r / ry, y = r - xx * ry;
    int n0 = rowbase[r];
    for (int z0 = 0; z0 < rz; z0 += 64) {
        const int z = z0 + lane;
        const int64_t f = (int64_t)r * rz + z;
        const bool occ = z < rz && binary[f] != 0;
        const unsigned long long m = __ballot(occ);
        if (occ) {
            const int n = n0 + __popcll(m & ((1ull << lane) - 1ull));
            const int j = colbase[z * ry + y] + (int)pre[f];
            // N is the CALLER's count of occupied cells (the size of every output): a caller that passed a wrong one — the pipeline takes it from the host
            // side of the checkpoint load instead of reading totals[0] back — must not write outside its buffers (it compares totals[0] with N later)
            if (n < N && j < N) {
                const float u0 = ((float)xx + jitter[(size_t)n * 3]) / (float)rx, u1 = ((float)y + jitter[(size_t)n * 3 + 1]) / (float)ry,
                            u2 = ((float)z + jitter[(size_t)n * 3 + 2]) / (float)rz;
                const float w0 = u0 * (hi0 - lo0) + lo0, w1 = u1 * (hi1 - lo1) + lo1, w2 = u2 * (hi2 - lo2) + lo2;
                indices[n] = f;
                order[j] = n;
                world[(size_t)n * 3] = w0; world[(size_t)n * 3 + 1] = w1; world[(size_t)n * 3 + 2] = w2;
                world_slot[(size_t)j * 3] = w0; world_slot[(size_t)j * 3 + 1] = w1; world_slot[(size_t)j * 3 + 2] = w2;
            }
        }
        n0 += __popcll(m);
    }
}
// a wave per (x, y) row: the row's points are n in [rowbase[r], rowbase[r + 1]) (ascending cells); the kept ones go to
// mask[keepbase[r] + rank] and to the voxel grid (grid_scatter7_kernel's writes)
__global__ __launch_bounds__(256) void grid_write_kept_kernel(const int* __restrict__ rowbase, const int* __restrict__ keepbase, const uint8_t* __restrict__ keep,
                                                               const int64_t* __restrict__ indices, const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                               const float* __restrict__ alpha, int64_t* __restrict__ mask, float* __restrict__ grid,
                                                               int nrows, int Np)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int nb = rowbase[r], ne = r + 1 < nrows ? rowbase[r + 1] : Np;
    int m0 = keepbase[r];
    for (int q = nb; q < ne; q += 64) {
        const int n = q + lane;
        const bool kp = n < ne && keep[n] != 0;
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const int64_t f = indices[n];
            mask[m0 + __popcll(m & ((1ull << lane) - 1ull))] = f;
            float* g = grid + (size_t)f * 7;
            g[0] = xyz[(size_t)n * 3]; g[1] = xyz[(size_t)n * 3 + 1]; g[2] = xyz[(size_t)n * 3 + 2];
            g[3] = rgb[(size_t)n * 3]; g[4] = rgb[(size_t)n * 3 + 1]; g[5] = rgb[(size_t)n * 3 + 2];
            g[6] = alpha[n];
        }
        m0 += __popcll(m);
    }
}

extern "C" {

// workspace layout (bytes, 256-aligned parts): pre uint16 [rx*ry*rz] | colcnt int [ry*rz] | rowcnt int [rx*ry] | keeprow int [rx*ry] | totals int [4]
static size_t occ_align(size_t v) { return (v + 255) / 256 * 256; }
size_t dreg_grid_occupied_workspace_bytes(int rx, int ry, int rz)
{
    return occ_align((size_t)rx * ry * rz * sizeof(uint16_t)) + occ_align((size_t)ry * rz * sizeof(int)) + 2 * occ_align((size_t)rx * ry * sizeof(int)) + 256;
}
struct OccWs { uint16_t* pre; int* colcnt; int* rowcnt; int* keeprow; int* totals; };
static OccWs occ_ws(void* workspace, int rx, int ry, int rz)
{
    char* p = (char*)workspace;
    OccWs w;
    w.pre = (uint16_t*)p; p += occ_align((size_t)rx * ry * rz * sizeof(uint16_t));
    w.colcnt = (int*)p; p += occ_align((size_t)ry * rz * sizeof(int));
    w.rowcnt = (int*)p; p += occ_align((size_t)rx * ry * sizeof(int));
    w.keeprow = (int*)p; p += occ_align((size_t)rx * ry * sizeof(int));
    w.totals = (int*)p;
    return w;
}
// device address of the int32 pair (N = occupied cells, n_keep = kept cells once dreg_grid_write_kept has run) inside the workspace
void* dreg_grid_occupied_totals(void* workspace, int rx, int ry, int rz) { return occ_ws(workspace, rx, ry, rz).totals; }
// steps 1 + 2 (see above): after it the int at dreg_grid_occupied_totals()[0] is N
int dreg_grid_occupied_count(const uint8_t* binary, void* workspace, size_t workspace_bytes, int rx, int ry, int rz, void* stream)
{
    if (!binary || !workspace || workspace_bytes < dreg_grid_occupied_workspace_bytes(rx, ry, rz) || rx > 65535 || rx <= 0 || ry <= 0 || rz <= 0) return DREG_EINVAL;
    const OccWs w = occ_ws(workspace, rx, ry, rz);
    hipStream_t st = (hipStream_t)stream;
    const int NS = (rx + 15) / 16;
    if (NS > 256) return DREG_EINVAL;
    int COLS = 1;
    while (COLS * 2 * NS <= 256) COLS *= 2;
    const int nprefix = (ry * rz + COLS - 1) / COLS;
    hipLaunchKernelGGL(grid_occupied_count_kernel, dim3(nprefix + (rx * ry + 255) / 256), dim3(256), 0, st, binary, w.pre, w.colcnt, w.rowcnt, rx, ry, rz, NS, COLS, nprefix);
    hipLaunchKernelGGL(grid_scan2_kernel, dim3(2), dim3(1024), 0, st, w.colcnt, ry * rz, w.rowcnt, rx * ry, w.totals);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// step 3: indices int64 [N], order int32 [N], world / world_slot fp32 [N,3]; jitter fp32 [N,3] indexed like indices; aabb = (lo xyz, hi xyz)
int dreg_grid_occupied_build(const uint8_t* binary, void* workspace, const float* jitter, const float* aabb, int64_t* indices, int* order,
                             float* world, float* world_slot, int rx, int ry, int rz, int N, void* stream)
{
    if (N == 0) return DREG_OK;
    if (!binary || !workspace || !jitter || !indices || !order || !world || !world_slot) return DREG_EINVAL;
    const OccWs w = occ_ws(workspace, rx, ry, rz);
    hipLaunchKernelGGL(grid_occupied_build_kernel, dim3((rx * ry + 3) / 4), dim3(256), 0, (hipStream_t)stream, binary, w.pre, w.colcnt, w.rowcnt, jitter,
                       indices, order, world, world_slot, rx, ry, rz, N, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5]);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// dreg_ngp_density_fwd_ws with the dense query's alpha / keep (dreg_ngp_alpha_keep's arithmetic) written by the same launch and the kept
// cells counted per (x, y) row into the workspace of dreg_grid_occupied_count (idx = the points' flat cell indices)
int dreg_ngp_density_keep_fwd_ws(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                                 const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                 const float* aabb, int Np, int contract, void* workspace, size_t workspace_bytes, const int* order, int x_in_slot_order,
                                 float* alpha, uint8_t* keep, float delta, float threshold, void* stream)
{
    if (Np == 0) return DREG_OK;
    if (!alpha || !keep) return DREG_EINVAL;
    NgpLevels lv;
    for (int l = 0; l < 16; ++l) { lv.offset[l] = offset[l]; lv.size[l] = size[l]; lv.res[l] = res[l]; lv.scale[l] = scale[l]; lv.hashed[l] = hashed[l]; }
    NgpKeepEpi ke{alpha, keep, delta, threshold};
    if (g_ngp_xcd_levels && workspace && workspace_bytes >= dreg_ngp_density_workspace_bytes(Np))
        ngp_density_launch_xcd(x, table, w1, w2, density, raw, lv, aabb, Np, contract, workspace, stream, order, order ? x_in_slot_order : 0, ke);
    else ngp_density_launch(x, table, w1, w2, density, raw, lv, aabb, Np, contract, stream, order, order ? x_in_slot_order : 0, ke);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// voxel_mask (ascending kept flat indices, int64, capacity Np) and voxel_grid[idx] = (xyz, rgb, alpha) of the kept points from the per-row
// kept counts the density launch left in the workspace; afterwards dreg_grid_occupied_totals()[1] = number of kept cells.  grid must be zeroed.
int dreg_grid_write_kept(void* occ_workspace, const float* xyz, const float* rgb, const float* alpha, const int64_t* indices, const uint8_t* keep,
                         int64_t* mask, float* grid, int rx, int ry, int rz, int Np, void* stream)
{
    if (!occ_workspace || (Np > 0 && (!mask || !grid || !keep || !indices))) return DREG_EINVAL;
    const OccWs w = occ_ws(occ_workspace, rx, ry, rz);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grid_keeprow_kernel, dim3((rx * ry + 3) / 4), dim3(256), 0, st, w.rowcnt, keep, w.keeprow, rx * ry, Np);
    hipLaunchKernelGGL(grid_scan2_kernel, dim3(1), dim3(1024), 0, st, w.keeprow, rx * ry, (int*)nullptr, 0, w.totals + 1);
    if (Np > 0)
        hipLaunchKernelGGL(grid_write_kept_kernel, dim3((rx * ry + 3) / 4), dim3(256), 0, st, w.rowcnt, w.keeprow, keep, indices, xyz, rgb, alpha, mask, grid, rx * ry, Np);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

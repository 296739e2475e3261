// 3-D convolution as implicit GEMM on gfx950 MFMA (bf16 16x16x32 / exact-f32 16x16x4).
//
// Replaces the cuDNN conv3d calls behind nn.Conv3d on the reference hot path
// (conerf/model/resnet3d.py:79-84,120,143-147; conerf/model/feature_pyramid_net.py:47-56) and, with
// ksz = 1, the cuBLAS GEMMs behind nn.Linear (conerf/register/transformer.py:131-133,
// conerf/register/nerf_regtr.py:268-270).
//
// Layouts (HBM):
//   activations  NDHWC  [B, D, H, W, C]   (channels contiguous; C a power of two, >= 8)
//   packed weight for the gather form: [Cout][tap][Cin] with the K = tap*Cin+ci axis padded to a
//   multiple of 128 bytes (zero filled) -> both MFMA operands are K-contiguous.
//   One kernel serves forward (coordinate o*s - p + d) and data-gradient (coordinate (i + p - d)/s,
//   valid only when divisible) through the (sn, dsign, off, sd) coordinate map.
//
// Tile: 128 output voxels x BN output channels per 256-thread workgroup (4 waves, 2x2, each 64 x BN/2),
// 128 bytes of K per step, register-staged double-buffered LDS with a 16-byte-slot XOR swizzle
// (slot ^= (row>>1)&7) so the ds_read_b128 fragment reads of 16 consecutive rows are conflict free.
#include "common.h"
#include "../../include/dreg_nerf.h"   // signature check of every entry point defined here
#include <cstring>
extern "C" int dreg_fill_zero(void* p, size_t bytes, void* stream);   // fpn_ops.hip (include/dreg_nerf.h)

struct ConvGeom {
    int B, Di, Hi, Wi, Cin, log2Cin;  // gathered operand [B,Di,Hi,Wi,Cin]; ksz==1: log2Cin=30 (tap always 0)
    int Cmask;                        // (1<<log2Cin)-1
    int Do, Ho, Wo, Cout;             // row space [B,Do,Ho,Wo] x Cout
    int ksz, ntaps;
    int sn, dsign, off, sd;           // gathered coord = (o*sn + off + dsign*d) / sd
    uint32_t M;                       // B*Do*Ho*Wo
    uint32_t magW, magH, magD;        // magic reciprocals of Wo, Ho, Do
    int Kpad;                         // padded K (elements) = packed weight row stride
};

__device__ __forceinline__ void tap_decode(int tap, int ksz, int& dz, int& dy, int& dx) {
    if (ksz == 1) { dz = dy = dx = 0; }
    else if (ksz == 3) { dz = tap / 9; int r = tap - dz * 9; dy = r / 3; dx = r - dy * 3; }
    else { int k2 = ksz * ksz; dz = tap / k2; int r = tap - dz * k2; dy = r / ksz; dx = r - dy * ksz; }
}

__device__ __forceinline__ void vox_decode(uint32_t m, const ConvGeom& g, int& b, int& z, int& y, int& x) {
    uint32_t q1 = fdiv(m, g.magW); x = (int)(m - q1 * g.Wo);
    uint32_t q2 = fdiv(q1, g.magH); y = (int)(q1 - q2 * g.Ho);
    uint32_t q3 = fdiv(q2, g.magD); z = (int)(q2 - q3 * g.Do);
    b = (int)q3;
}

__device__ __forceinline__ uint32_t swz(int row, int slot) { return (uint32_t)row * 128u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    uint32_t xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient
// ------------------------------------------------------------------------------------------------
// ATR (measurement build only, tools/bench_bn_conv_fuse.py — round-5 review item 1): the A operand is transformed on its way from global memory to
// LDS, a = relu(x * scale[b, ci] + shift[b, ci]) rounded to bf16 — a training-mode BatchNorm + ReLU applied INSIDE the 1^3 convolution that consumes
// it (bn2 -> relu -> conv3 of a bottleneck, resnet3d.py:99-104) instead of as a pass of its own.  a_ss: fp32 [B][Cin][2].
template <typename T, typename TO, int BN, bool ATR = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const T* __restrict__ in, const T* __restrict__ wt, TO* __restrict__ out,
    const float* __restrict__ bias, const TO* __restrict__ addend, ConvGeom g,
    int relu, int Da, int Ha, int Wa, int add_shift, int tilesN, const uint8_t* __restrict__ rowocc = nullptr,
    const int* __restrict__ rowlist = nullptr, uint32_t nrows = 0,     // rowlist (optional): only output voxels rowlist[0 .. nrows) are computed
    const float* __restrict__ a_ss = nullptr)
{
    constexpr int BM = 128;
    constexpr int G = 16 / sizeof(T);
    constexpr int BKe = 128 / sizeof(T);
    constexpr int WN = BN / 2;
    constexpr int TM = 4, TN = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tile_m = lid / tilesN, tile_n = lid - tile_m * tilesN;
    const uint32_t m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int gq = t & 7, rbase = t >> 3;

    // rowocc (host guarantees: forward gather, no bias / addend, BM % Wo == 0): byte per output W-row, 0 = its whole receptive field
    // is zero, so the result is exactly zero: tiles made of such rows are zero-filled without touching the operands
    if (rowocc) {
        const uint32_t nrow = BM / (uint32_t)g.Wo, row0 = m0 / (uint32_t)g.Wo, rows_total = g.M / (uint32_t)g.Wo;
        bool any = false;
        for (uint32_t r = 0; r < nrow; ++r) any = any || (row0 + r < rows_total && rowocc[row0 + r] != 0);
        if (!any) {
            const size_t base = (size_t)m0 * g.Cout + n0;
            for (int i = t; i < BM * (BN / G); i += 256) {
                const int r = i / (BN / G), c = (i - r * (BN / G)) * G;
                if (m0 + r < g.M) {
                    if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint4*>(out + base + (size_t)r * g.Cout + c) = make_uint4(0, 0, 0, 0);
                    else { for (int e = 0; e < G; ++e) out[base + (size_t)r * g.Cout + c + e] = (TO)0; }
                }
            }
            return;
        }
    }

    int zb[4], yb[4], xb[4];
    uint32_t vb[4];
    bool mv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t m = m0 + rbase + 32 * i;
        mv[i] = m < (rowlist ? nrows : g.M);
        if (rowlist) m = mv[i] ? (uint32_t)rowlist[m] : 0u;
        int b, z, y, x;
        vox_decode(mv[i] ? m : 0, g, b, z, y, x);
        zb[i] = z * g.sn + g.off; yb[i] = y * g.sn + g.off; xb[i] = x * g.sn + g.off;
        vb[i] = (uint32_t)b * (uint32_t)(g.Di * g.Hi * g.Wi);
    }

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[NB];
    const int nk = g.Kpad / BKe;

    auto load_g = [&](int k) {
        const int kel = k * BKe + gq * G;
        const int tap = kel >> g.log2Cin, ci = kel & g.Cmask;
        int dz, dy, dx;
        tap_decode(tap, g.ksz, dz, dy, dx);
        dz *= g.dsign; dy *= g.dsign; dx *= g.dsign;
        const bool tv = tap < g.ntaps && ci < g.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int z = zb[i] + dz, y = yb[i] + dy, x = xb[i] + dx;
            bool v = mv[i] && tv;
            if (g.sd == 2) { v = v && !((z | y | x) & 1); z >>= 1; y >>= 1; x >>= 1; }
            v = v && (unsigned)z < (unsigned)g.Di && (unsigned)y < (unsigned)g.Hi && (unsigned)x < (unsigned)g.Wi;
            uint32_t vox = vb[i] + (uint32_t)((z * g.Hi + y) * g.Wi + x);
            const T* p = in + (size_t)vox * g.Cin + ci;
            ra[i] = v ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
            if constexpr (ATR && sizeof(T) == 2) {
                if (v) {        // (padding stays zero: the transform applies to voxels that exist)
                    const float* ss = a_ss + ((size_t)(vb[i] / (uint32_t)(g.Di * g.Hi * g.Wi)) * g.Cin + ci) * 2;
                    const uint32_t w4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                    uint32_t o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 q = *reinterpret_cast<const float4*>(ss + 4 * e);      // (scale, shift) of channels ci + 2e, ci + 2e + 1
                        const float lo = fmaxf(__uint_as_float(w4[e] << 16) * q.x + q.y, 0.f), hi = fmaxf(__uint_as_float(w4[e] & 0xffff0000u) * q.z + q.w, 0.f);
                        o4[e] = f2bf2(lo, hi);
                    }
                    ra[i] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const T* p = wt + (size_t)(n0 + rbase + 32 * i) * g.Kpad + (size_t)k * BKe + gq * G;
            rb[i] = *reinterpret_cast<const uint4*>(p);
        }
    };
    auto store_l = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sA + swz(rbase + 32 * i, gq)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4*>(sB + swz(rbase + 32 * i, gq)) = rb[i];
    };
    auto compute = [&](int buf) {
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
        const int fr = lane & 15, kg = lane >> 4;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sA + swz(wm * 64 + i * 16 + fr, ks * 4 + kg));
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(sB + swz(wn * WN + j * 16 + fr, ks * 4 + kg));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // exact f32: lane group kg owns k = kg*8 + j; MFMA j multiplies the j-th element of every group.
            float bfv[TN][8];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WN + j * 16 + fr;
                float4 lo = *reinterpret_cast<const float4*>(sB + swz(row, 2 * kg));
                float4 hi = *reinterpret_cast<const float4*>(sB + swz(row, 2 * kg + 1));
                bfv[j][0] = lo.x; bfv[j][1] = lo.y; bfv[j][2] = lo.z; bfv[j][3] = lo.w;
                bfv[j][4] = hi.x; bfv[j][5] = hi.y; bfv[j][6] = hi.z; bfv[j][7] = hi.w;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                float4 lo = *reinterpret_cast<const float4*>(sA + swz(row, 2 * kg));
                float4 hi = *reinterpret_cast<const float4*>(sA + swz(row, 2 * kg + 1));
                float av[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bfv[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };

    load_g(0);
    store_l(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        if (k + 1 < nk) load_g(k + 1);
        compute(k & 1);
        if (k + 1 < nk) store_l((k + 1) & 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    const int col_l = lane & 15, rowq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t m = m0 + wm * 64 + i * 16 + rowq + r;
            if (m >= (rowlist ? nrows : g.M)) continue;
            if (rowlist) m = (uint32_t)rowlist[m];
            size_t arow = 0;
            if (addend) {
                int b, z, y, x;
                vox_decode(m, g, b, z, y, x);
                arow = ((size_t)((b * Da + (z >> add_shift)) * Ha + (y >> add_shift)) * Wa + (x >> add_shift)) * g.Cout;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                float v = acc[i][j][r];
                if (bias) v += bias[n];
                if (addend) { if (relu == 2) v = Elem<TO>::ld(addend + arow + n) > 0.f ? v : 0.f; else v += Elem<TO>::ld(addend + arow + n); }
                if (relu == 1) v = fmaxf(v, 0.f);
                Elem<TO>::st(out + (size_t)m * g.Cout + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward / stride-1 data gradient, bf16, direct-to-LDS staging (buffer_load ... lds, 16 B per lane).
// Requirements: Cin % 64 == 0, ntaps <= 32, sd == 1.  The LDS image is lane-linear per wave-instruction (8 rows x 8
// slots), so the XOR swizzle is applied on the SOURCE side: physical slot s of row r fetches logical granule
// s ^ ((r>>1)&7) of that row's 128-byte K chunk.  Padding taps and rows beyond M point their lane at an out-of-range
// buffer offset: the hardware bounds check of the raw buffer returns zeros into LDS.  Per-row tap validity is a
// 27-bit mask computed once per block, so the K loop spends 3 VALU ops per row on addressing.
// ------------------------------------------------------------------------------------------------
// DBG (tools/igemm_phase_probe.py, measurement only): s_memtime stamps around the four phases of a K step, summed per wave into g_igemm_dbg
#ifdef DREG_PROBE
__device__ unsigned long long g_igemm_dbg[8];
#endif
__device__ __forceinline__ unsigned long long dbg_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
// AP (round 3): the 128-row tiles on EIGHT waves in two anti-phase groups, for launches that put at most one or two workgroups on a CU
// (layer2-4 of the ResNet, the point-set half's linear layers: ~150 launches per training step).  With four waves and one workgroup
// per CU nothing overlaps: per 64-channel K step a wave waits for the stage, issues 8 direct-to-LDS pieces (~600 cycles), reads 16
// fragments and only then runs its 32 MFMAs (512 cycles of a ~1,700-cycle step: tools/igemm_phase_probe.py).  Here waves 0-3 and 4-7
// (w and w + 4 share a SIMD) each own a 64 x (BN/4) quarter-strip of the SAME tile and run one barrier apart over a ring of four
// stages: while one group reads its fragments of stage k and issues its half of the pieces of stage k + 3, the other runs its MFMAs
// of the stage it read a phase earlier.  Every accumulator still sums the K steps in order: results are bit-identical to the
// four-wave kernel, so the choice may depend on the launch size.
template <typename TO, int BM, int BN, int DBG = 0, int AP = 0>
__global__ __launch_bounds__((BN == 256 || AP) ? 512 : 256) void conv_igemm_glds_kernel(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ wt, TO* __restrict__ out,
    const float* __restrict__ bias, const TO* __restrict__ addend, ConvGeom g,
    int relu, int Da, int Ha, int Wa, int add_shift, int tilesN, uint32_t in_bytes, uint32_t wt_bytes,
    const int* __restrict__ rowlist, uint32_t nrows, int ksplit, int nstage, float* __restrict__ bn_part)
{
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int NW = (BN == 256 || AP) ? 8 : 4, WAVES_N = NW / 2;      // waves: 2 (M) x WAVES_N (N), (BM/2) x (BN/WAVES_N) each
    // AP with the 256 x 256 tile: stages of 32 channels (64-byte rows, 16 rows per direct-to-LDS piece) so that a ring of four fits LDS
    constexpr bool AP256 = AP && BM == 256;
    constexpr int BKe = AP256 ? 32 : 64, WMt = BM / 2, WN = BN / WAVES_N, TM = WMt / 16, TN = WN / 16;
    constexpr int RPP = AP256 ? 16 : 8, GPR = 64 / RPP;                  // rows per piece, 16-byte granules per row
    constexpr int LOGK = AP256 ? 5 : 6;
    constexpr int IA = (BM / RPP) / NW, IBW = (BN / RPP) / NW;           // wave-instructions per wave per stage
    constexpr int A_BYTES = BM * BKe * 2, B_BYTES = BN * BKe * 2, STAGE = A_BYTES + B_BYTES;
    auto gswz = [](int r) -> int { return AP256 ? (((r >> 3) & 1) * 3) : ((r >> 1) & 7); };   // granule XOR of LDS row r (see the fragment reads)
    constexpr uint32_t OOB = 0x7fffff00u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tile_m = lid / tilesN, tile_n = lid - tile_m * tilesN;
    const uint32_t m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, wt_bytes, 0x00020000);

    // staging roles: wave-instruction i of this wave covers tile rows (wave*IA + i)*8 + (lane>>3), physical slot lane&7
    uint32_t abase[IA], amask[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int r = (wave * IA + i) * RPP + lane / GPR;
        const int gl = (lane % GPR) ^ gswz(r);
        const uint32_t ri = m0 + r;            // row of the (possibly sparse) row space
        const bool rv = ri < nrows;
        const uint32_t m = rowlist ? (rv ? (uint32_t)rowlist[ri] : 0u) : ri;   // output voxel
        if (g.ksz == 1 && g.sn == 1 && g.sd == 1 && g.off == 0) {
            // 1^3 stride 1 (the linear layers, two of three bottleneck convolutions): the gathered voxel IS the output voxel
            amask[i] = rv ? 1u : 0u;
            abase[i] = m * (uint32_t)(g.Cin * 2) + (uint32_t)(gl * 16);
            continue;
        }
        int b, z, y, x;
        vox_decode(rv ? m : 0, g, b, z, y, x);
        const int zb = z * g.sn + g.off, yb = y * g.sn + g.off, xb = x * g.sn + g.off;
        uint32_t mask = 0;
        if (rv) {
            if (g.ksz == 3) {
                // the 27-tap validity mask is the outer product of three 3-bit per-axis masks (9 bounds tests instead of 81 and no
                // 27-iteration loop: ~1,300 instructions of prologue per workgroup became ~100)
                uint32_t vz = 0, vy = 0, vx = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    vz |= ((unsigned)(zb + d * g.dsign) < (unsigned)g.Di ? 1u : 0u) << d;
                    vy |= ((unsigned)(yb + d * g.dsign) < (unsigned)g.Hi ? 1u : 0u) << d;
                    vx |= ((unsigned)(xb + d * g.dsign) < (unsigned)g.Wi ? 1u : 0u) << d;
                }
                uint32_t m9 = 0;                                   // bit dy*3 + dx
#pragma unroll
                for (int d = 0; d < 3; ++d) m9 |= ((vy >> d) & 1u) ? (vx << (3 * d)) : 0u;
#pragma unroll
                for (int d = 0; d < 3; ++d) mask |= ((vz >> d) & 1u) ? (m9 << (9 * d)) : 0u;
            } else {
                for (int tap = 0; tap < g.ntaps; ++tap) {
                    int dz, dy, dx;
                    tap_decode(tap, g.ksz, dz, dy, dx);
                    const int zz = zb + dz * g.dsign, yy = yb + dy * g.dsign, xx = xb + dx * g.dsign;
                    if ((unsigned)zz < (unsigned)g.Di && (unsigned)yy < (unsigned)g.Hi && (unsigned)xx < (unsigned)g.Wi) mask |= 1u << tap;
                }
            }
        }
        amask[i] = mask;
        // byte offset of the d = 0 corner (may be "negative": only used together with a valid tap, where the sum is in range)
        abase[i] = (uint32_t)(((int)(b * g.Di + zb) * g.Hi + yb) * g.Wi + xb) * (uint32_t)(g.Cin * 2) + (uint32_t)(gl * 16);
    }
    uint32_t bbase[IBW];
#pragma unroll
    for (int i = 0; i < IBW; ++i) {
        const int r = (wave * IBW + i) * RPP + lane / GPR;
        const int gl = (lane % GPR) ^ gswz(r);
        bbase[i] = (uint32_t)(n0 + r) * (uint32_t)(g.Kpad * 2) + (uint32_t)(gl * 16);
    }
    // keep the per-lane bases as VALUES in registers: left alone, the compiler re-derives each abase from its factors inside the K loop
    // (a v_mul_lo_u32 and three more VALU operations per direct-to-LDS load in front of the select; the issue phase is ~40 % of a K step
    // of an under-filled launch: tools/igemm_phase_probe.py)
#pragma unroll
    for (int i = 0; i < IA; ++i) asm volatile("" : "+v"(abase[i]), "+v"(amask[i]));
#pragma unroll
    for (int i = 0; i < IBW; ++i) asm volatile("" : "+v"(bbase[i]));
    const int chunks = g.Cin / BKe;            // K steps per tap
    const int nk = g.ntaps * chunks;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- requests.  K steps are requested IN ORDER, so the gather address of a piece is kept as  per-lane offset (fixed for a tap) +
    // scalar offset (tap offset + channel chunk, the instruction's soffset operand):  a K step costs its 2 x (IA + IBW) M0 / load
    // instructions and two scalar adds; the tap decode and the per-row validity select (4 VALU per piece) run once per tap, behind a real
    // branch.  (The issue phase was ~600 of ~1,700 cycles of a K step: a wave issues about one instruction per 5 cycles whatever it is.)
    // Offsets are biased so that both operands of the hardware's address sum are non-negative: the descriptor starts X + Y bytes before
    // the tensor, the lane offset carries + X (the d = 0 corner of a row may lie up to X bytes before it: forward padding), the scalar
    // offset + Y (data-gradient taps walk backwards); a valid lane's sum is inside the tensor, an invalid lane's offset is out of range.
    const int pneg = g.off < 0 ? -g.off : 0;
    const uint32_t biasX = (uint32_t)(((pneg * g.Hi + pneg) * g.Wi + pneg) * g.Cin * 2);
    const uint32_t biasY = g.dsign < 0 ? (uint32_t)(((g.ksz - 1) * ((g.Hi + 1) * g.Wi + 1)) * g.Cin * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rs_inb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)in - (biasX + biasY)), 0, in_bytes + biasX + biasY, 0x00020000);
#pragma unroll
    for (int i = 0; i < IA; ++i) { abase[i] += biasX; asm volatile("" : "+v"(abase[i])); }
    const int step_x = g.dsign * g.Cin * 2, step_y = step_x * (g.Wi - (g.ksz - 1)), step_z = step_x * ((g.Hi - (g.ksz - 1)) * g.Wi - (g.ksz - 1));
    int ck = 0, cc = 0, ctap = 0, cdx = 0, cdy = 0, cdz = 0;    // cursor: next K step, its channel chunk, tap and tap coordinates
    uint32_t ctoff = 0;                                         // biased byte offset of the cursor's tap
    uint32_t voffA[IA];
    auto tap_select = [&]() {
#pragma unroll
        for (int i = 0; i < IA; ++i) voffA[i] = ((amask[i] >> ctap) & 1u) ? abase[i] : OOB;
    };
    auto cursor_init = [&](int k) {
        ck = k;
        ctap = g.ksz == 1 ? 0 : (k >> (g.log2Cin - LOGK));      // k^3 layers have power-of-two channel counts (fill_geom): a shift
        cc = k - ctap * chunks;
        tap_decode(ctap, g.ksz, cdz, cdy, cdx);
        ctoff = (uint32_t)(((cdz * g.Hi + cdy) * g.Wi + cdx) * step_x) + biasY;
        tap_select();
    };
    auto issue = [&](int k, int buf) {
        (void)k;                                                // == ck: requests are sequential
        char* sA = smem + buf * STAGE + wave * IA * 1024;
        char* sB = smem + buf * STAGE + A_BYTES + wave * IBW * 1024;
        const uint32_t soA = ctoff + (uint32_t)(cc * BKe * 2), soB = (uint32_t)(ck * BKe * 2);
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_inb, (lds_ptr_t)(sA + i * 1024), 16, (int)voffA[i], (int)soA, 0, 0);
#pragma unroll
        for (int i = 0; i < IBW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lds_ptr_t)(sB + i * 1024), 16, (int)bbase[i], (int)soB, 0, 0);
        ++ck;
        if (++cc == chunks) {                                   // next tap (the empty volatile asm keeps this a real branch)
            asm volatile("" ::: "memory");
            cc = 0; ++ctap;
            if (++cdx < g.ksz) ctoff += (uint32_t)step_x;
            else { cdx = 0; if (++cdy < g.ksz) ctoff += (uint32_t)step_y; else { cdy = 0; ++cdz; ctoff += (uint32_t)step_z; } }
            tap_select();
        }
    };
    auto compute = [&](int buf) {
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
        const int fr = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sA + swz(wm * WMt + i * 16 + fr, ks * 4 + kg));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(sB + swz(wn * WN + j * 16 + fr, ks * 4 + kg));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // C^T: a lane holds 4 consecutive channels of one row
        }
    };

    // split-K (small row spaces: blockIdx.y = K slice, raw fp32 partial tiles at out + slice*M*Cout, finished by
    // splitk_reduce_kernel); ksplit == 1 is the plain kernel
    // (32-bit: nk <= 27 * 32 K steps, <= 16 slices; a 64-bit division by a run-time value is ~150 scalar instructions, twice per workgroup)
    const int k_begin = ksplit > 1 ? (int)((uint32_t)nk * blockIdx.y / (uint32_t)ksplit) : 0;
    const int k_end = ksplit > 1 ? (int)((uint32_t)nk * (blockIdx.y + 1) / (uint32_t)ksplit) : nk;
    if (ksplit > 1) out += (size_t)blockIdx.y * g.M * g.Cout;
    // ring of nstage (2..4) LDS stages: stage k is consumed while the loads of up to nstage-1 later stages are in flight
    constexpr int LPS = IA + IBW;                      // direct-to-LDS loads per wave per stage (vmcnt retires them in order)
    cursor_init(k_begin);
    if constexpr (AP) {
        static_assert(((BM == 128 && (BN == 128 || BN == 64)) || (BM == 256 && BN == 256)) && !DBG && STAGE <= 65535, "anti-phase form: 128-row tiles, or 256 x 256 with 32-channel stages");
        constexpr int KS = AP256 ? 1 : 2;            // 32-channel fragment groups per stage
        const int nku = k_end - k_begin;
        if (nku > 0) {
            const int grp = wave >> 2;
            const int fr = lane & 15, kg = lane >> 4;
            const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
            // fragment addresses in ring slot 0 (or 2): fixed per lane; the odd slots are + STAGE as an immediate offset
            // 64-byte rows (AP256): slot = granule ^ 3 * ((row >> 3) & 1) — a ds_read_b128 is served in four groups of 16 lanes
            // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... : fragment rows q, q+4, q+8, q+12 of a 16-row block meet in one group with
            // K granules (g, g^1, g^1, g)), and with this XOR the 16 lanes of every group touch 16 different 16-byte slots of a 256-byte line
            auto fswz = [&](int row, int slot) -> uint32_t {
                if constexpr (AP256) return (uint32_t)row * 64u + (uint32_t)((slot ^ gswz(row)) << 4);
                else return swz(row, slot);
            };
            uint32_t fa[KS][TM], fb[KS][TN];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[ks][i] = lds0 + fswz(wm * WMt + i * 16 + fr, ks * 4 + kg);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[ks][j] = lds0 + A_BYTES + fswz(wn * WN + j * 16 + fr, ks * 4 + kg);
            }
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            i32x4_t af[KS][TM], bf[KS][TN];
            auto rd1 = [&](uint32_t addr, int odd) -> i32x4_t {
                i32x4_t v;
                if (odd) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(STAGE));
                else asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
                return v;
            };
            auto rd = [&](int odd) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[ks][i] = rd1(fa[ks][i], odd);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[ks][j] = rd1(fb[ks][j], odd);
                }
            };
            auto mm = [&]() {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[ks][j]), __builtin_bit_cast(bf16x8_t, af[ks][i]), acc[i][j], 0, 0, 0);
            };
            // prologue: stages 0..2 requested; stage 0 landed (own pieces: counted wait; the other waves': the barrier)
            for (int p = 0; p < 3 && p < nku; ++p) issue(k_begin + p, p);
            if (nku >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (nku == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (grp) __builtin_amdgcn_s_barrier();           // the second group runs one phase behind
            auto unit = [&](int u, int slot) {
                // load half: fragments of stage u, then stage u + 3 into the ring slot of stage u - 1 (read by both groups two barriers ago)
                rd(slot & 1);
                if (u + 3 < nku) {
                    issue(k_begin + u + 3, (slot + 3) & 3);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");   // this wave's pieces of stage u + 1 have landed
                } else {
                    if (u + 2 < nku) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // MFMA half
                __builtin_amdgcn_s_setprio(1);
                mm();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            auto shift = [&](uint32_t d) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[ks][i] += d;
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[ks][j] += d;
                }
            };
            for (int u = 0; u < nku; u += 4) {
                unit(u, 0);
                if (u + 1 < nku) unit(u + 1, 1);
                shift(2u * STAGE);
                if (u + 2 < nku) unit(u + 2, 2);
                if (u + 3 < nku) unit(u + 3, 3);
                shift(0u - 2u * STAGE);
            }
            if (!grp) __builtin_amdgcn_s_barrier();
        }
    } else {
    for (int s = 0; s < nstage - 1; ++s)
        if (k_begin + s < k_end) issue(k_begin + s, s);
    int cbuf = 0, ibuf = nstage - 1;                   // stage being consumed / stage the next issue goes to
    unsigned long long dsum[4] = {0, 0, 0, 0}, dt0 = 0;
    for (int k = k_begin; k < k_end; ++k) {
        const int ahead = min(nstage - 2, k_end - 1 - k);   // later stages already issued
        if constexpr (DBG) dt0 = dbg_now();
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DBG) { const unsigned long long t_ = dbg_now(); dsum[0] += t_ - dt0; dt0 = t_; }
        __syncthreads();                               // stage k landed for every wave; everyone is done with stage k-1
        if constexpr (DBG) { const unsigned long long t_ = dbg_now(); dsum[1] += t_ - dt0; dt0 = t_; }
        if (k + nstage - 1 < k_end) issue(k + nstage - 1, ibuf);
        if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = dbg_now(); dsum[2] += t_ - dt0; dt0 = t_; }
        compute(cbuf);
        if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = dbg_now(); dsum[3] += t_ - dt0; }
        cbuf = cbuf + 1 == nstage ? 0 : cbuf + 1;
        ibuf = ibuf + 1 == nstage ? 0 : ibuf + 1;
    }
    if constexpr (DBG) {
#ifdef DREG_PROBE
        if (lane == 0) {
            for (int q = 0; q < 4; ++q) atomicAdd(&g_igemm_dbg[q], dsum[q]);
            atomicAdd(&g_igemm_dbg[4], (unsigned long long)(k_end - k_begin));   // K steps x waves
            atomicAdd(&g_igemm_dbg[5], 1ull);                                    // waves
        }
#endif
    }
    }   // !AP

    // Epilogue through LDS, one pass per wave row (wm): accumulators -> fp32 tile [BM/2][BN] -> coalesced 16-byte rows with bias /
    // addend / ReLU applied in fp32.  The products are formed transposed (weights as the MFMA's first operand), so a lane holds FOUR
    // CONSECUTIVE CHANNELS of one output row per MFMA tile: one ds_write_b128 per tile instead of four ds_write_b32 (16 instead of 64
    // LDS writes per lane and pass — the epilogue is a third of a short 1^3-convolution workgroup's life).  16-byte granules are XOR-ed
    // with the row's low four bits: the 16 lanes of a write phase (16 rows, same channels) land in 16 different granules.
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN / 8;                    // 8-column chunks per row
    constexpr int NTHR = NW * 64;
    // bn_part (bf16 output, dense rows; the BatchNorm behind this convolution): sums of the STORED (rounded) values and of their squares
    // per 128-row chunk and channel, [chunk][Cout][2] — the layout of bn_partial_kernel's chunk sums with 128 rows per chunk, so the
    // BatchNorm's statistics pass (one more read of this tensor) is not launched.  Which tile shape / wave count a launch gets depends
    // on how many grids share it, and a grid's statistics must not: EVERY form adds a chunk's rows in the same order —
    //     sum_{j = 0..15} ( (c[j] + c[j + 32]) + (c[j + 16] + c[j + 48]) ),   c[q] = x[q] + x[q + 64]
    // A thread always works on the same 8 columns (NTHR % CPR == 0) and on rows NG apart (NG = 16, 32 or 64): it owns the 64 / NG
    // classes q = its row group + i * NG, keeps one accumulator per class (static register indices: the row loop below is fully
    // unrolled), combines its own classes as the bracketing above says, and the rest of the bracket is formed across threads in LDS.
    constexpr int NG = NTHR / CPR, NCLS = 64 / NG, ITER = WMt / NG;
    static_assert(NTHR % CPR == 0 && (NG == 16 || NG == 32 || NG == 64) && WMt % 64 == 0, "BatchNorm sums: row classes modulo 64");
    float bs1[NCLS][8], bs2[NCLS][8];
#pragma unroll
    for (int i = 0; i < NCLS; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs1[i][e] = 0.f; bs2[i][e] = 0.f; }
    float bias8[8];                                  // a thread's 8 columns never change: its bias values are loaded once
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = bias ? bias[n0 + (t % CPR) * 8 + e] : 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if (wm == pass) {
            const int rl = lane & 15, gq = lane >> 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 16 + rl;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gr = ((wn * WN + j * 16) >> 2) + gq;
                    *reinterpret_cast<f32x4_t*>(sC + row * BN + ((gr ^ rl) << 2)) = acc[i][j];
                }
            }
        }
        __syncthreads();
        // The addend (and the row list) are fetched for EPF rows at a time before any of them is used: one row per round trip made the
        // accumulate-epilogues of the 1^3 data gradients a chain of 32 dependent global loads per tile (2.5 TB/s on an HBM-bound layer).
        // (128-row tiles only: the 256 x 256 tile has no registers left — its BatchNorm sums went to scratch and the forward launches slowed down)
        constexpr int EPF = (sizeof(TO) == 2 && ITER % 4 == 0 && BM == 128) ? 4 : 1;
#pragma unroll
        for (int it0 = 0; it0 < ITER; it0 += EPF) {
        uint32_t pm[EPF];
        uint4 pq[EPF];
#pragma unroll
        for (int u = 0; u < EPF; ++u) {
            const int lrow = (t + (it0 + u) * NTHR) / CPR;
            const int row = pass * WMt + lrow;
            pm[u] = (m0 + row < nrows) ? (rowlist ? (uint32_t)rowlist[m0 + row] : m0 + row) : 0xffffffffu;
        }
        if constexpr (EPF > 1) {
            if (addend && add_shift >= 0) {
#pragma unroll
                for (int u = 0; u < EPF; ++u) {
                    pq[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (pm[u] != 0xffffffffu) {
                        int b, z, y, x;
                        vox_decode(pm[u], g, b, z, y, x);
                        pq[u] = *reinterpret_cast<const uint4*>(addend + ((size_t)((b * Da + (z >> add_shift)) * Ha + (y >> add_shift)) * Wa + (x >> add_shift)) * g.Cout + n0 + (t % CPR) * 8);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < EPF; ++u) {
            const int it = it0 + u;
            const int c = t + it * NTHR;
            const int lrow = c / CPR, cc = (c - lrow * CPR) * 8;
            const int row = pass * WMt + lrow;
            if (pm[u] == 0xffffffffu) continue;
            const uint32_t m = pm[u];
            const int g0 = cc >> 2, sw = lrow & 15;
            const float4 lo = *reinterpret_cast<const float4*>(sC + lrow * BN + ((g0 ^ sw) << 2));
            const float4 hi = *reinterpret_cast<const float4*>(sC + lrow * BN + (((g0 + 1) ^ sw) << 2));
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int n = n0 + cc;
            if (bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            }
            if (addend && add_shift >= 0) {
                int b, z, y, x;
                vox_decode(m, g, b, z, y, x);
                const TO* ap = addend + ((size_t)((b * Da + (z >> add_shift)) * Ha + (y >> add_shift)) * Wa + (x >> add_shift)) * g.Cout + n;
                // the row's 8 addend values in ONE 16-byte load (n and Cout are multiples of 8: 16-byte aligned on a 16-byte aligned tensor);
                // eight 2-byte loads per thread and row made the accumulate-epilogue of the 1^3 data gradients latency-bound
                float a8[8];
                if constexpr (sizeof(TO) == 2) {
                    const uint4 q = EPF > 1 ? pq[u] : *reinterpret_cast<const uint4*>(ap);
                    const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a8[2 * e] = __uint_as_float(w4[e] << 16); a8[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
                } else {
                    const float4 q0 = *reinterpret_cast<const float4*>(ap), q1 = *reinterpret_cast<const float4*>(ap + 4);
                    a8[0] = q0.x; a8[1] = q0.y; a8[2] = q0.z; a8[3] = q0.w; a8[4] = q1.x; a8[5] = q1.y; a8[6] = q1.z; a8[7] = q1.w;
                }
                if (relu == 2) {
                    // ReLU-backward mask (the data gradient of a layer whose forward epilogue applied ReLU): the "addend" is that
                    // layer's stored activation, and the gradient passes where it is positive (transformer.py:291 linear1 -> relu)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = a8[e] > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += a8[e];
                }
            }
            if (relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            TO* dst = out + (size_t)m * g.Cout + n;
            if (add_shift < 0) {
                // stride-2 data gradient as a 2^3-tap convolution over dOut with 8 x Cin "parity class" output channels:
                // channel block p = (pz,py,px) of coarse voxel (z,y,x) is voxel (2z+pz, 2y+py, 2x+px) of the [B,Da,Ha,Wa,Cin] result
                const int cin = -add_shift, p = n / cin, ci = n - p * cin;
                int b, z, y, x;
                vox_decode(m, g, b, z, y, x);
                const int zf = 2 * z + (p >> 2), yf = 2 * y + ((p >> 1) & 1), xf = 2 * x + (p & 1);
                if (zf >= Da || yf >= Ha || xf >= Wa) continue;
                dst = out + ((size_t)((b * Da + zf) * Ha + yf) * Wa + xf) * cin + ci;
                if constexpr (sizeof(TO) == 2) {
                    if (addend) {      // accumulating form (addend == out): a second contribution to an existing gradient, added in fp32 at the
                                       // voxel the class lands on — one rounding, no temporary + add pass (dreg_conv3d_dgrad_s2_acc)
                        const uint4 q = *reinterpret_cast<const uint4*>(dst);
                        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w4[e] << 16); v[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u); }
                    }
                }
            }
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                if (bn_part) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo_ = __uint_as_float(w[e] << 16), hi_ = __uint_as_float(w[e] & 0xffff0000u);
                        bs1[it % NCLS][2 * e] += lo_; bs2[it % NCLS][2 * e] += lo_ * lo_;
                        bs1[it % NCLS][2 * e + 1] += hi_; bs2[it % NCLS][2 * e + 1] += hi_ * hi_;
                    }
                }
            }
        }
        }
        if constexpr (sizeof(TO) == 2) {
            // a 128-row chunk is complete after both passes of a 128-row tile, after EACH pass (128 rows) of a 256-row tile
            if (bn_part && (WMt == 128 || pass == 1)) {
                float* red = reinterpret_cast<float*>(smem);            // [NG row groups][BN][2]
                __syncthreads();
                const int chunkc = t % CPR, grp = t / CPR;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float p1, p2;                                        // this thread's classes, combined as the fixed order says
                    if constexpr (NCLS == 4) { p1 = (bs1[0][e] + bs1[2][e]) + (bs1[1][e] + bs1[3][e]); p2 = (bs2[0][e] + bs2[2][e]) + (bs2[1][e] + bs2[3][e]); }
                    else if constexpr (NCLS == 2) { p1 = bs1[0][e] + bs1[1][e]; p2 = bs2[0][e] + bs2[1][e]; }
                    else { p1 = bs1[0][e]; p2 = bs2[0][e]; }
                    red[((grp * BN) + chunkc * 8 + e) * 2] = p1;
                    red[((grp * BN) + chunkc * 8 + e) * 2 + 1] = p2;
                }
#pragma unroll
                for (int i = 0; i < NCLS; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { bs1[i][e] = 0.f; bs2[i][e] = 0.f; }
                __syncthreads();
                const size_t chunk = WMt == 128 ? (size_t)2 * tile_m + pass : (size_t)tile_m;
                for (int col = t; col < BN; col += NTHR) {
                    float a = 0.f, b = 0.f;
                    auto rd = [&](int q, int w) -> float { return red[(q * BN + col) * 2 + w]; };
                    for (int jq = 0; jq < 16; ++jq) {
                        if constexpr (NG == 16) { a += rd(jq, 0); b += rd(jq, 1); }
                        else if constexpr (NG == 32) { a += rd(jq, 0) + rd(jq + 16, 0); b += rd(jq, 1) + rd(jq + 16, 1); }
                        else { a += (rd(jq, 0) + rd(jq + 32, 0)) + (rd(jq + 16, 0) + rd(jq + 48, 0)); b += (rd(jq, 1) + rd(jq + 32, 1)) + (rd(jq + 16, 1) + rd(jq + 48, 1)); }
                    }
                    float* o = bn_part + (chunk * g.Cout + n0 + col) * 2;
                    o[0] = a; o[1] = b;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: part[s][co][n] = sum_{voxels of split s} gout[m][co] * in[gather(m, tap(n))][ci(n)]
// rows = co (128 / block), cols = packed K index n (BNC / block), reduction over voxels in steps of 32.
// Both operands are reduction-strided in memory ([voxel][channel]); bf16 fragments are formed with the
// LDS transpose read ds_read_b64_tr_b16 (TR = true) or by eight 16-bit reads (TR = false, the checker).
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BNC, bool TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const T* __restrict__ gout, const T* __restrict__ in, float* __restrict__ part,
    ConvGeom g, int tilesCol, uint32_t vox_per_split, const uint8_t* __restrict__ rowocc = nullptr)
{
    constexpr int KV = 32;
    constexpr int G = 16 / sizeof(T);
    constexpr int RSA = BM * sizeof(T), RSB = BNC * sizeof(T);  // LDS row strides (bytes)
    constexpr int GPA = RSA / 16, GPB = RSB / 16;                // granules per row
    constexpr int A_BYTES = KV * RSA, B_BYTES = KV * RSB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = KV * GPA / 256, NBG = KV * GPB / 256;     // granules per thread
    constexpr int WN = BNC / 2, WM = BM / 2, TM = WM / 16, TN = WN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tile_r = blockIdx.x / tilesCol, tile_c = blockIdx.x - tile_r * tilesCol;
    const int co0 = tile_r * BM, n0 = tile_c * BNC;
    const uint32_t v_begin = blockIdx.y * vox_per_split;
    const uint32_t v_end = min(v_begin + vox_per_split, g.M);

    // B-side per-thread constants: one voxel row per thread per step, NBG granules in that row
    // thread t -> row rB = t / (GPB / NBG) ... keep it simple: granule id q = t + 256*i, row = q / GPB, gc = q % GPB
    int b_dz[NBG], b_dy[NBG], b_dx[NBG], b_ci[NBG];
    bool b_tv[NBG];
#pragma unroll
    for (int i = 0; i < NBG; ++i) {
        const int q = t + 256 * i, gc = q % GPB;
        const int n = n0 + gc * G;
        const int tap = n >> g.log2Cin;
        b_ci[i] = n & g.Cmask;
        tap_decode(tap, g.ksz, b_dz[i], b_dy[i], b_dx[i]);
        b_dz[i] *= g.dsign; b_dy[i] *= g.dsign; b_dx[i] *= g.dsign;
        b_tv[i] = tap < g.ntaps && b_ci[i] < g.Cin;
    }

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[NA], rb[NBG];
    auto load_g = [&](uint32_t v0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = t + 256 * i, row = q / GPA, ga = q % GPA;
            const uint32_t m = v0 + row;
            const T* p = gout + (size_t)m * g.Cout + co0 + ga * G;
            ra[i] = (m < v_end) ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NBG; ++i) {
            const int q = t + 256 * i, row = q / GPB;
            const uint32_t m = v0 + row;
            bool v = (m < v_end) && b_tv[i];
            int b, z, y, x;
            vox_decode(v ? m : 0, g, b, z, y, x);
            z = z * g.sn + g.off + b_dz[i]; y = y * g.sn + g.off + b_dy[i]; x = x * g.sn + g.off + b_dx[i];
            if (g.sd == 2) { v = v && !((z | y | x) & 1); z >>= 1; y >>= 1; x >>= 1; }
            v = v && (unsigned)z < (unsigned)g.Di && (unsigned)y < (unsigned)g.Hi && (unsigned)x < (unsigned)g.Wi;
            const uint32_t vox = (uint32_t)b * (uint32_t)(g.Di * g.Hi * g.Wi) + (uint32_t)((z * g.Hi + y) * g.Wi + x);
            const T* p = in + (size_t)vox * g.Cin + b_ci[i];
            rb[i] = v ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_l = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int q = t + 256 * i; *reinterpret_cast<uint4*>(sA + q * 16) = ra[i]; }
#pragma unroll
        for (int i = 0; i < NBG; ++i) { const int q = t + 256 * i; *reinterpret_cast<uint4*>(sB + q * 16) = rb[i]; }
    };
    auto compute = [&](int buf) {
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
        const int fi = lane & 15, kb = (lane >> 4) * 8;
        if constexpr (sizeof(T) == 2) {
            bf16x8_t af[TM], bf[TN];
            if constexpr (TR) {
                // 16-lane group reads a [4 voxel][16 channel] block; lane fi supplies row fi>>2, 8 bytes at column (fi&3)*4
                typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int cb = wm * WM + i * 16;
                    const char* p0 = sA + (kb + (fi >> 2)) * RSA + (cb + (fi & 3) * 4) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 4 * RSA));
                    af[i] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cb = wn * WN + j * 16;
                    const char* p0 = sB + (kb + (fi >> 2)) * RSB + (cb + (fi & 3) * 4) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 4 * RSB));
                    bf[j] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        af[i][e] = *reinterpret_cast<const short*>(sA + (kb + e) * RSA + (wm * WM + i * 16 + fi) * 2);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        bf[j][e] = *reinterpret_cast<const short*>(sB + (kb + e) * RSB + (wn * WN + j * 16 + fi) * 2);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        } else {
            float bfv[TN][8];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    bfv[j][e] = *reinterpret_cast<const float*>(sB + (kb + e) * RSB + (wn * WN + j * 16 + fi) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float av[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    av[e] = *reinterpret_cast<const float*>(sA + (kb + e) * RSA + (wm * WM + i * 16 + fi) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bfv[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (v_begin < v_end) {
        const int nk = (int)((v_end - v_begin + KV - 1) / KV);
        // rowocc (host guarantees Wo % KV == 0 and vox_per_split % KV == 0): a 32-voxel step lies in one output W-row; rows flagged 0
        // gather only zeros, so their products add exactly nothing and the step is skipped
        auto next = [&](int k) {
            if (rowocc)
                while (k < nk && rowocc[(v_begin + (uint32_t)k * KV) / (uint32_t)g.Wo] == 0) ++k;
            return k;
        };
        int k = next(0), buf = 0;
        if (k < nk) {
            load_g(v_begin + (uint32_t)k * KV);
            store_l(0);
        }
        __syncthreads();
        while (k < nk) {
            const int kn = next(k + 1);
            if (kn < nk) load_g(v_begin + (uint32_t)kn * KV);
            compute(buf);
            if (kn < nk) store_l(buf ^ 1);
            __syncthreads();
            k = kn; buf ^= 1;
        }
    }
    float* dst = part + (size_t)blockIdx.y * g.Cout * g.Kpad;
    const int col_l = lane & 15, rowq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * WM + i * 16 + rowq + r;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                dst[(size_t)co * g.Kpad + n0 + wn * WN + j * 16 + col_l] = acc[i][j][r];
        }
}

// bf16 weight gradient with direct-to-LDS staging (buffer_load ... lds).  Tiles are [64 voxels][channels] row-major and
// lane-linear per wave-instruction; rows past the split's end and padding taps use the buffer bounds check to zero-fill.
// Bank conflicts of the transpose reads (a 16-lane group touches 4 voxel rows x 32 B at the same column) are removed by an
// XOR of the 16-byte granule index with f(row), applied on the SOURCE side of the DMA and on the read address:
// 256-byte rows: f = ((row&3) + 4*((row>>3)&1)) * 2;  128-byte rows: f = (((row>>1)&1) + 2*((row>>3)&1)) * 2.
// (ds_read_b64_tr_b16 services a 32-lane half-wave per cycle: voxel rows r..r+3 and r+8..r+11 must land on disjoint banks)
#ifdef DREG_PROBE
__device__ unsigned long long g_wgrad_dbg[8];
#endif
//   // AP == 2 (tools/wgrad_phase_probe.py, measurement only): cycles per phase, summed over waves
template <int GP> __device__ __forceinline__ int wg_swz(int row) {   // rows of >= 256 bytes (GP >= 16): the XOR acts on the low four granule bits
    return GP >= 16 ? (((row & 3) + 4 * ((row >> 3) & 1)) << 1) : ((((row >> 1) & 1) + 2 * ((row >> 3) & 1)) << 1);
}

// NW = 4: 2 x 2 waves (128 x 128 / 128 x 64 / 64 x 128 / 64 x 64 tiles).  NW = 8: 2 x 4 waves on a 256 x 256 tile — per 64-voxel stage it
// moves 64 KB for 8.4 MFLOP instead of 32 KB for 2.1 MFLOP: the 128-square tile needs ~62 B/clk of direct-to-LDS traffic at the MFMA
// roof, which is the whole L2 -> LDS path of a CU (tools/hw_probe/l2_stream.hip: 129 GB/s per CU) and the reason it stops at 0.7 PFLOP/s.
// ABL (tools/bench_wgrad.py --ablate, wrong results): 1 no MFMA, 2 no fragment reads, 3 no direct-to-LDS loads — what bounds the loop
// KV_ / NS_: voxels per stage and LDS ring depth.  The loop is bound by what the ring keeps in flight (a stage is requested one
// iteration before it is consumed; the L2 / HBM gather latency is ~2 us under load): (NS-1) stages x workgroups per CU.  64-voxel
// stages x 2 keep 64 KB per CU in flight for either tile; 32-voxel stages x 4 or 5 keep 96 / 128 KB in the same LDS footprint.
// PIPE: the fragment reads of the next group of MFMAs are issued before the current group's MFMAs (two register slots per operand),
// so the LDS read phase of a wave runs under its own MFMAs instead of in front of them; same products, same accumulation order.
// AP (round 3, NW = 8, KV = 32, NS = 4): ANTI-PHASE wave groups.  Waves 0-3 and 4-7 (wave w and w + 4 share a SIMD) run one barrier
// apart over 32-voxel units: in every phase one group issues its 4 direct-to-LDS pieces for the unit three ahead and reads the 24
// transposed fragments of its next unit, while the other group runs the 32 MFMAs of the unit it read one phase earlier at raised
// priority; then they swap.  Same products in the same order as the lockstep loop (bit-identical), but a SIMD's matrix pipe always
// has one of its two waves in the MFMA half instead of both loading, then both multiplying behind one barrier per stage.
// GRP (round 4): ONE launch for the weight gradients of many layers (the 36 linear layers of the point-set half: 4 - 12 weight tiles x 16
// splits each, ~25 us per launch for ~4 GFLOP — launch- and fill-bound one at a time).  The kernel parameters of a workgroup's layer come
// from a descriptor table (rowocc carries the table, rows_fast its length; binary search over block0 with a wave-uniform index); the
// body, the tile / split arithmetic and therefore every partial sum are those of the layer's own launch.
struct WgradGroupDesc {
    const bf16_t* gout; const bf16_t* in; float* part;
    ConvGeom g;
    int tilesCol, tiles, nsplit;
    uint32_t vps, gbytes, ibytes, nrows;
    int block0;
};
static_assert(sizeof(WgradGroupDesc) == 144, "descriptor layout is part of the ABI (dreg_wgrad_group_desc_bytes)");
template <int BM, int BNC, bool ROWS, int NW = 4, int ABL = 0, int KV_ = 64, int NS_ = 2, int PIPE = 0, int AP = 0, bool GRP = false>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_glds_kernel(
    const bf16_t* __restrict__ gout, const bf16_t* __restrict__ in, float* __restrict__ part,
    ConvGeom g, int tilesCol, int tiles, int nsplit, uint32_t vox_per_split, uint32_t gout_bytes, uint32_t in_bytes,
    const int* __restrict__ rowlist, uint32_t nrows, const uint8_t* __restrict__ rowocc = nullptr, int rows_fast = 0)
{
    uint32_t bid = blockIdx.x;
    if constexpr (GRP) {
        const WgradGroupDesc* D = reinterpret_cast<const WgradGroupDesc*>(rowocc);
        int lo = 0, hi = rows_fast - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (D[mid].block0 <= (int)bid) lo = mid; else hi = mid - 1;
        }
        const WgradGroupDesc d = D[lo];
        gout = d.gout; in = d.in; part = d.part; g = d.g; tilesCol = d.tilesCol; tiles = d.tiles; nsplit = d.nsplit;
        vox_per_split = d.vps; gout_bytes = d.gbytes; in_bytes = d.ibytes; nrows = d.nrows;
        rowocc = nullptr; rows_fast = 0; rowlist = nullptr;
        bid -= (uint32_t)d.block0;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
    constexpr int KV = KV_, NS = NS_;                     // voxels per stage, LDS ring depth (prefetch distance NS-1)
    constexpr int RSA = BM * 2, RSB = BNC * 2;
    constexpr int GPA = RSA / 16, GPB = RSB / 16;          // granules per row (16 or 8)
    constexpr int A_BYTES = KV * RSA, B_BYTES = KV * RSB, STAGE = A_BYTES + B_BYTES;
    constexpr int IA = A_BYTES / 1024 / NW, IB = B_BYTES / 1024 / NW;   // wave-instructions per wave per tile
    constexpr int LPS = IA + IB;                           // DMA instructions per wave per stage (for counted vmcnt)
    static_assert(IA >= 1 && IB >= 1, "stage too small for the wave count");
    constexpr int RPA = 64 / GPA, RPB = 64 / GPB;          // rows per wave-instruction
    constexpr int WAVES_N = NW / 2;                       // waves: 2 (rows of the tile) x WAVES_N (columns)
    constexpr int WN = BNC / WAVES_N, WM = BM / 2, TM = WM / 16, TN = WN / 16;
    constexpr uint32_t OOB = 0x7fffff00u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware placement (block b runs on XCD b % 8): every XCD owns the voxel splits s == xcd (mod 8) and runs all
    // (co, n) tiles of a split back to back, so both streamed operands are fetched from HBM by ONE L2 and re-used there.
    uint32_t split, tile;
    if ((nsplit & 7) == 0) { const uint32_t xcd = bid & 7, j = bid >> 3; split = (j / tiles) * 8 + xcd; tile = j % tiles; }
    else { split = bid / tiles; tile = bid - split * tiles; }
    const uint32_t tile_r = tile / tilesCol, tile_c = tile - tile_r * tilesCol;
    const int co0 = tile_r * BM, n0 = tile_c * BNC;
    const uint32_t v_begin = split * vox_per_split;
    const uint32_t v_end = min(v_begin + vox_per_split, nrows);   // positions in the row space (dense: voxels; sparse: list entries)
    // ROWS: this block's slice of the row list is copied to LDS up front (plain loads inside the K loop would make hipcc
    // drain the DMA queue with vmcnt(0) every step); it sits behind the NS stages.
    // rows_fast (stride 1, output volume = input volume: the two head convolutions): the slice also keeps every row's (z, y, x) packed
    // in 30 bits.  The gathered voxel of output voxel m at tap d is m + a lane constant, and the bounds test reads the packed
    // coordinates: no voxel decode (three magic divisions) per load in the K loop — a third of the loop's VALU work.
    const int* srow = reinterpret_cast<const int*>(smem + NS * STAGE);
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    const u32x2_t* srow2 = reinterpret_cast<const u32x2_t*>(smem + NS * STAGE);   // rows_fast: (row, packed z|y|x) pairs
    if constexpr (ROWS && AP >= 3) {
        // anti-phase row-list form: (row index, border flags) pairs — z == 0: 1, z == D-1: 2, y == 0: 4, y == H-1: 8, x == 0: 16, x == W-1: 32,
        // every real row: 256 — padded with sentinel entries (flag 128, an out-of-range row) to whole 32-row units plus the four units the
        // preparation runs ahead: the K loop tests  (lane's tap mask & row flags) == 0  and never decodes a voxel or compares a count
        u32x2_t* w2_ = reinterpret_cast<u32x2_t*>(smem + NS * STAGE);
        const uint32_t nreal = v_begin < v_end ? v_end - v_begin : 0u;
        const uint32_t npad = ((nreal + KV - 1) / KV + 4) * KV;
        const uint32_t m_sent = 0x7fffff00u / (uint32_t)((g.Cout > g.Cin ? g.Cout : g.Cin) * 2) + 1u;
        for (uint32_t i = t; i < npad; i += NW * 64) {
            if (i < nreal) {
                const int m = rowlist[v_begin + i];
                int b, z, y, x;
                vox_decode((uint32_t)m, g, b, z, y, x);
                const uint32_t fl = (z == 0 ? 1u : 0u) | (z == g.Do - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == g.Ho - 1 ? 8u : 0u) |
                                    (x == 0 ? 16u : 0u) | (x == g.Wo - 1 ? 32u : 0u) | 256u;
                w2_[i] = (u32x2_t){(uint32_t)m, fl};
            } else w2_[i] = (u32x2_t){m_sent, 128u};
        }
        __syncthreads();
    } else
    if constexpr (ROWS) {
        int* w_ = reinterpret_cast<int*>(smem + NS * STAGE);
        u32x2_t* w2_ = reinterpret_cast<u32x2_t*>(smem + NS * STAGE);
        for (uint32_t i = v_begin + t; i < v_end; i += NW * 64) {
            const int m = rowlist[i];
            if (rows_fast) {
                int b, z, y, x;
                vox_decode((uint32_t)m, g, b, z, y, x);
                w2_[i - v_begin] = (u32x2_t){(uint32_t)m, ((uint32_t)z << 20) | ((uint32_t)y << 10) | (uint32_t)x};
            } else w_[i - v_begin] = m;
        }
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)gout, 0, gout_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);

    // per wave-instruction lane constants (the tile row of instruction j is j*RP + lane/GP; j = wave*I + i)
    const int ra = lane / GPA, rb = lane / GPB;
    uint32_t a_col[IA];
    int b_dz[IB], b_dy[IB], b_dx[IB], b_ci[IB];
    bool b_tv[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wave * IA + i) * RPA + ra;
        a_col[i] = (uint32_t)(co0 + (((lane % GPA) ^ wg_swz<GPA>(row)) * 8)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = (wave * IB + i) * RPB + rb;
        const int n = n0 + (((lane % GPB) ^ wg_swz<GPB>(row)) * 8);
        const int tap = n >> g.log2Cin;
        b_ci[i] = n & g.Cmask;
        tap_decode(tap, g.ksz, b_dz[i], b_dy[i], b_dx[i]);
        b_dz[i] *= g.dsign; b_dy[i] *= g.dsign; b_dx[i] *= g.dsign;
        b_tv[i] = tap < g.ntaps && b_ci[i] < g.Cin;
    }

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Row-aligned stages (dense stride-1 launches whose 64-voxel stages are whole x-rows of the volume: 64 % Wo == 0, Ho * Wo % 64 == 0 —
    // every layer of the 64^3 / 32^3 / 16^3 levels): the stage fixes (b, z, y0) for the whole workgroup, and what a lane adds — its
    // row inside the stage, its tap — never changes.  The gather address of a lane is then  uniform stage base + lane constant, and
    // only the bounds test needs the stage's z / y: ~8 VALU instructions per load instead of a voxel decode with carries (~30).  The
    // address arithmetic of the general path was a third of the kernel's time (tools/bench_wgrad_ablate.py).
    // (a stage may also be a PART of an x-row — Wo a multiple of the stage length: then the stage's first x joins the base and the
    // x bounds test moves into the loop)
    const bool aligned = !ROWS && g.sn == 1 && g.sd == 1 && g.dsign == 1 &&
                         (((KV % g.Wo) == 0 && ((g.Ho * g.Wo) % KV) == 0) || (g.Wo % KV) == 0) &&
                         (uint64_t)g.B * g.Di * g.Hi * g.Wi * g.Cin * 2 < 0x7fffff00ull;
    int al_dz[IB], al_dy[IB], al_x[IB];
    uint32_t al_off[IB];
    uint32_t al_l[IB];
    if (ROWS && rows_fast) {
#pragma unroll
        for (int i = 0; i < IB; ++i) {    // the same registers: tap offsets per axis and the byte offset of the tap relative to the output voxel
            al_dz[i] = g.off + b_dz[i]; al_dy[i] = g.off + b_dy[i]; al_x[i] = g.off + b_dx[i];
            al_off[i] = (uint32_t)((((al_dz[i] * g.Hi + al_dy[i]) * g.Wi + al_x[i]) * g.Cin + b_ci[i]) * 2);
        }
    }
    if (aligned) {
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int l = (wave * IB + i) * RPB + rb;              // voxel of this lane inside a stage
            const int ly = l / g.Wo, lx = l - ly * g.Wo;
            al_l[i] = b_tv[i] ? (uint32_t)l : 0x7fffffffu;         // an invalid column (K padding) never passes the "inside the split" test
            al_dz[i] = g.off + b_dz[i];
            al_dy[i] = ly + g.off + b_dy[i];
            al_x[i] = lx + g.off + b_dx[i];
            // bytes relative to the stage's first voxel (b, z, y0, x0) of the gathered operand; may be negative: added modulo 2^32
            al_off[i] = (uint32_t)((((al_dz[i] * g.Hi + al_dy[i]) * g.Wi + al_x[i]) * g.Cin + b_ci[i]) * 2);
        }
    }

    auto issue = [&](uint32_t v0, int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
        if constexpr (ROWS) {
            if (rows_fast) {
                // every LDS read of the stage first (clamped index: no branch), then the direct-to-LDS loads: the reads' latency is paid
                // once per stage, not once per load (the compiler keeps an LDS read behind every direct-to-LDS load issued before it)
                uint32_t am[IA];
                u32x2_t bm[IB];
                const uint32_t last = v_end - 1 - v_begin;
#pragma unroll
                for (int i = 0; i < IA; ++i) am[i] = srow2[min(v0 + (wave * IA + i) * RPA + ra - v_begin, last)][0];
#pragma unroll
                for (int i = 0; i < IB; ++i) bm[i] = srow2[min(v0 + ((wave * IB) + i) * RPB + rb - v_begin, last)];
#pragma unroll
                for (int i = 0; i < IA; ++i) {
                    const int j = wave * IA + i;
                    const uint32_t vi = v0 + j * RPA + ra;
                    const uint32_t voff = (vi < v_end) ? am[i] * (uint32_t)(g.Cout * 2) + a_col[i] : OOB;
                    if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(sA + j * 1024), 16, (int)voff, 0, 0, 0);
                    else asm volatile("" :: "v"(voff));
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int j = wave * IB + i;
                    const uint32_t vi = v0 + j * RPB + rb, p = bm[i][1];
                    const bool v = vi < v_end && b_tv[i] && (unsigned)((int)(p >> 20) + al_dz[i]) < (unsigned)g.Di &&
                                   (unsigned)((int)((p >> 10) & 1023u) + al_dy[i]) < (unsigned)g.Hi && (unsigned)((int)(p & 1023u) + al_x[i]) < (unsigned)g.Wi;
                    const uint32_t voff = v ? bm[i][0] * (uint32_t)(g.Cin * 2) + al_off[i] : OOB;
                    if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sB + j * 1024), 16, (int)voff, 0, 0, 0);
                    else asm volatile("" :: "v"(voff));
                }
                return;
            }
        }
        if (AP && aligned) {
            // anti-phase loop: every address of the unit first (independent chains the scheduler can interleave), then the pieces back to
            // back — with an address chain in front of each piece a piece cost ~200 cycles of the load half (tools/wgrad_phase_probe.py)
            uint32_t voa[IA], vob[IB];
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const uint32_t vi = v0 + (wave * IA + i) * RPA + ra;
                voa[i] = (vi < v_end) ? vi * (uint32_t)(g.Cout * 2) + a_col[i] : OOB;
            }
            int sb, sz, sy, sx;
            vox_decode(v0, g, sb, sz, sy, sx);
            const uint32_t base = (uint32_t)(((sb * g.Di + sz) * g.Hi + sy) * g.Wi + sx) * (uint32_t)(g.Cin * 2);
            const uint32_t left = v_end - v0;
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const bool v = al_l[i] < left && (unsigned)(sz + al_dz[i]) < (unsigned)g.Di && (unsigned)(sy + al_dy[i]) < (unsigned)g.Hi &&
                               (unsigned)(sx + al_x[i]) < (unsigned)g.Wi;
                vob[i] = v ? base + al_off[i] : OOB;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(sA + (wave * IA + i) * 1024), 16, (int)voa[i], 0, 0, 0);
                else asm volatile("" :: "v"(voa[i]));
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sB + (wave * IB + i) * 1024), 16, (int)vob[i], 0, 0, 0);
                else asm volatile("" :: "v"(vob[i]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int j = wave * IA + i;
            const uint32_t vi = v0 + j * RPA + ra;
            uint32_t m = vi;
            if constexpr (ROWS) m = vi < v_end ? (uint32_t)srow[vi - v_begin] : 0u;
            const uint32_t voff = (vi < v_end) ? m * (uint32_t)(g.Cout * 2) + a_col[i] : OOB;
            if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(sA + j * 1024), 16, (int)voff, 0, 0, 0);
            else asm volatile("" :: "v"(voff));
        }
        if (aligned) {
            int sb, sz, sy, sx;
            vox_decode(v0, g, sb, sz, sy, sx);                       // wave-uniform: v0 starts an x-row or a KV-aligned part of one
            const uint32_t base = (uint32_t)(((sb * g.Di + sz) * g.Hi + sy) * g.Wi + sx) * (uint32_t)(g.Cin * 2);
            const uint32_t left = v_end - v0;
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int j = wave * IB + i;
                const bool v = al_l[i] < left && (unsigned)(sz + al_dz[i]) < (unsigned)g.Di && (unsigned)(sy + al_dy[i]) < (unsigned)g.Hi &&
                               (unsigned)(sx + al_x[i]) < (unsigned)g.Wi;
                const uint32_t voff = v ? base + al_off[i] : OOB;
                if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sB + j * 1024), 16, (int)voff, 0, 0, 0);
                else asm volatile("" :: "v"(voff));
            }
            return;
        }
        const uint32_t mb = v0 + (wave * IB) * RPB + rb;
        // one full voxel decode per K step (row of instruction 0); the wave's other rows are +RPB, +2*RPB ... voxels
        // further along x with at most one carry when Wo >= IB*RPB (else every row is decoded in full)
        int b0, z0, y0, x0;
        vox_decode((!ROWS && mb < g.M) ? mb : 0, g, b0, z0, y0, x0);
        const bool fast = !ROWS && g.Wo >= IB * RPB;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int j = wave * IB + i;
            const uint32_t vi = mb + i * RPB;
            uint32_t m = vi;
            if constexpr (ROWS) m = vi < v_end ? (uint32_t)srow[vi - v_begin] : 0u;
            bool v = (vi < v_end) && b_tv[i];
            int b = b0, z = z0, y = y0, x = x0 + i * RPB;
            if (fast) {
                if (x >= g.Wo) { x -= g.Wo; if (++y >= g.Ho) { y = 0; if (++z >= g.Do) { z = 0; ++b; } } }
            } else {
                vox_decode(v ? m : 0, g, b, z, y, x);
            }
            z = z * g.sn + g.off + b_dz[i]; y = y * g.sn + g.off + b_dy[i]; x = x * g.sn + g.off + b_dx[i];
            if (g.sd == 2) { v = v && !((z | y | x) & 1); z >>= 1; y >>= 1; x >>= 1; }
            v = v && (unsigned)z < (unsigned)g.Di && (unsigned)y < (unsigned)g.Hi && (unsigned)x < (unsigned)g.Wi;
            const uint32_t vox = (uint32_t)b * (uint32_t)(g.Di * g.Hi * g.Wi) + (uint32_t)((z * g.Hi + y) * g.Wi + x);
            const uint32_t voff = v ? (vox * (uint32_t)g.Cin + (uint32_t)b_ci[i]) * 2u : OOB;
            if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sB + j * 1024), 16, (int)voff, 0, 0, 0);
            else asm volatile("" :: "v"(voff));
        }
    };
    // The transpose reads go through inline asm: hipcc otherwise orders ds_read_b64_tr_b16 behind every pending LDS-DMA with an
    // s_waitcnt vmcnt(0), which would drain the stages prefetched above.  All 16 reads of a step are issued, then one
    // lgkmcnt(0) + sched_barrier (an MFMA may not be hoisted above the wait: it only touches registers).
    typedef __attribute__((ext_vector_type(2))) int i32x2_t;
    auto tr_read = [&](uint32_t addr) -> i32x2_t {
        i32x2_t v;
        if constexpr (ABL != 2) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        else { v = (i32x2_t){(int)addr, (int)addr}; asm volatile("" : "+v"(v)); }
        return v;
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto compute = [&](int buf) {
        const uint32_t sA = lds_base + buf * STAGE;
        const uint32_t sB = sA + A_BYTES;
        const int fi = lane & 15;
        if constexpr (PIPE) {
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            constexpr int G = TM >= 4 ? 2 : 1, NG = TM / G, KS = KV / 32;
            i32x2_t alo[2][G], ahi[2][G], blo[2][TN], bhi[2][TN];
            auto rdA = [&](int ks, int grp, int slot) {
                const int kb = ks * 32 + (lane >> 4) * 8, r0 = kb + (fi >> 2), r1 = r0 + 4;
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int gi = (wm * WM + (grp * G + i) * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    alo[slot][i] = tr_read(sA + r0 * RSA + ((gi ^ wg_swz<GPA>(r0)) << 4) + o8);
                    ahi[slot][i] = tr_read(sA + r1 * RSA + ((gi ^ wg_swz<GPA>(r1)) << 4) + o8);
                }
            };
            auto rdB = [&](int ks, int slot) {
                const int kb = ks * 32 + (lane >> 4) * 8, r0 = kb + (fi >> 2), r1 = r0 + 4;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gi = (wn * WN + j * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    blo[slot][j] = tr_read(sB + r0 * RSB + ((gi ^ wg_swz<GPB>(r0)) << 4) + o8);
                    bhi[slot][j] = tr_read(sB + r1 * RSB + ((gi ^ wg_swz<GPB>(r1)) << 4) + o8);
                }
            };
            rdB(0, 0); rdA(0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int grp = 0; grp < NG; ++grp) {
                    const int as = (ks * NG + grp) & 1, bs = ks & 1;
                    if (grp + 1 < NG) rdA(ks, grp + 1, as ^ 1);
                    else if (ks + 1 < KS) { rdB(ks + 1, bs ^ 1); rdA(ks + 1, 0, as ^ 1); }
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8_t af[G], bf[TN];
#pragma unroll
                    for (int i = 0; i < G; ++i) { i32x4_t t4 = {alo[as][i][0], alo[as][i][1], ahi[as][i][0], ahi[as][i][1]}; af[i] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
                    for (int j = 0; j < TN; ++j) { i32x4_t t4 = {blo[bs][j][0], blo[bs][j][1], bhi[bs][j][0], bhi[bs][j][1]}; bf[j] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
                    for (int i = 0; i < G; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[grp * G + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[grp * G + i][j], 0, 0, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KV / 32; ++ks) {
            const int kb = ks * 32 + (lane >> 4) * 8;
            const int r0 = kb + (fi >> 2), r1 = r0 + 4;
            i32x2_t alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gi = (wm * WM + i * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                alo[i] = tr_read(sA + r0 * RSA + ((gi ^ wg_swz<GPA>(r0)) << 4) + o8);
                ahi[i] = tr_read(sA + r1 * RSA + ((gi ^ wg_swz<GPA>(r1)) << 4) + o8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gi = (wn * WN + j * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                blo[j] = tr_read(sB + r0 * RSB + ((gi ^ wg_swz<GPB>(r0)) << 4) + o8);
                bhi[j] = tr_read(sB + r1 * RSB + ((gi ^ wg_swz<GPB>(r1)) << 4) + o8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { typedef __attribute__((ext_vector_type(4))) int i32x4_t; i32x4_t t4 = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]}; af[i] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { typedef __attribute__((ext_vector_type(4))) int i32x4_t; i32x4_t t4 = {blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]}; bf[j] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (ABL != 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    else asm volatile("" :: "v"(af[i]), "v"(bf[j]));
                }
        }
    };

    if constexpr (AP) {
        // AP: 1 general anti-phase loop, 3 FAST dense form; 2 / 4 = the same with s_memtime stamps (tools/wgrad_phase_probe.py).
        // What the probe showed: a wave issues roughly one instruction per 5 cycles, whatever the instruction is — the load half of the
        // first anti-phase version took ~1,400 cycles per unit for 24 fragment reads + 4 pieces because it was ~200 INSTRUCTIONS long
        // (an address add in front of every read, a voxel decode and three compares per piece), against 620 cycles for the 32 MFMAs
        // of the other group.  FAST (dense stride-1 same-volume layers whose 32-voxel units are parts of one x-row; host-checked):
        //  * fragment-read addresses live in 24 registers for the whole kernel; the ring slot is an immediate offset (0 / 32 KB) plus
        //    one +-64 KB step of those registers every second unit (the unit loop is unrolled by four);
        //  * a piece's source offset is  uniform unit base + lane constant,  its bounds test  (lane mask & unit mask) == 0  with the
        //    unit's (z, y, x-phase) border flags in one scalar mask advanced incrementally — no decode, no per-axis compare.
        constexpr bool FAST = AP >= 3, STAMP = (AP == 2 || AP == 4);
        static_assert(NW == 8 && KV == 32 && NS == 4 && PIPE == 0 && LPS <= 15, "anti-phase loop: 8 waves, four 32-voxel units in the ring");
        if (v_begin < v_end) {
            const int nk = (int)((v_end - v_begin + KV - 1) / KV);
            const int grp = wave >> 2;
            const int fi = lane & 15;
            i32x2_t alo[TM], ahi[TM], blo[TN], bhi[TN];
            // ---- fragment reads
            // FAST: LDS byte addresses of this lane's reads of fragment row r0 in ring slot 0 (or 2).  Row r1 = r0 + 4 has the same
            // swizzle (r0 and r0 + 4 agree in row & 3 and in row >> 3), so its read is the same address + 4 rows as an immediate offset,
            // and the odd ring slots are + 32 KB as an immediate as well: 12 address registers serve all 24 reads of every unit.
            uint32_t fa[TM + TN];
            if constexpr (FAST) {
                static_assert(RSA == RSB, "one row stride for both operands");
                const int kb = (lane >> 4) * 8;
                const int r0 = kb + (fi >> 2);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gi = (wm * WM + i * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    fa[i] = lds_base + r0 * RSA + ((gi ^ wg_swz<GPA>(r0)) << 4) + o8;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gi = (wn * WN + j * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    fa[TM + j] = lds_base + A_BYTES + r0 * RSB + ((gi ^ wg_swz<GPB>(r0)) << 4) + o8;
                }
            }
            auto tr_read_o = [&](uint32_t addr, int which) -> i32x2_t {   // which: 0 r0 / even slot, 1 r1 / even, 2 r0 / odd, 3 r1 / odd (constant after inlining)
                i32x2_t v;
                if constexpr (ABL != 2) {
                    if (which == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
                    else if (which == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(4 * RSA));
                    else if (which == 2) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(STAGE));
                    else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(STAGE + 4 * RSA));
                } else { v = (i32x2_t){(int)addr, (int)addr}; asm volatile("" : "+v"(v)); }
                return v;
            };
            auto rd = [&](int buf) {
                if constexpr (FAST) {
                    const int odd = (buf & 1) * 2;
#pragma unroll
                    for (int i = 0; i < TM; ++i) { alo[i] = tr_read_o(fa[i], odd); ahi[i] = tr_read_o(fa[i], odd + 1); }
#pragma unroll
                    for (int j = 0; j < TN; ++j) { blo[j] = tr_read_o(fa[TM + j], odd); bhi[j] = tr_read_o(fa[TM + j], odd + 1); }
                    return;
                }
                const uint32_t sA = lds_base + buf * STAGE;
                const uint32_t sB = sA + A_BYTES;
                const int kb = (lane >> 4) * 8;
                const int r0 = kb + (fi >> 2), r1 = r0 + 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gi = (wm * WM + i * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    alo[i] = tr_read(sA + r0 * RSA + ((gi ^ wg_swz<GPA>(r0)) << 4) + o8);
                    ahi[i] = tr_read(sA + r1 * RSA + ((gi ^ wg_swz<GPA>(r1)) << 4) + o8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gi = (wn * WN + j * 16) / 8 + ((fi & 3) >> 1), o8 = (fi & 1) * 8;
                    blo[j] = tr_read(sB + r0 * RSB + ((gi ^ wg_swz<GPB>(r0)) << 4) + o8);
                    bhi[j] = tr_read(sB + r1 * RSB + ((gi ^ wg_swz<GPB>(r1)) << 4) + o8);
                }
            };
            auto mm = [&]() {
                typedef __attribute__((ext_vector_type(4))) int i32x4_t;
                bf16x8_t af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) { i32x4_t t4 = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]}; af[i] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
                for (int j = 0; j < TN; ++j) { i32x4_t t4 = {blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]}; bf[j] = __builtin_bit_cast(bf16x8_t, t4); }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (ABL != 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                        else asm volatile("" :: "v"(af[i]), "v"(bf[j]));
                    }
            };
            // ---- pieces.  FAST: lane constants + an incrementally advanced scalar cursor (the NEXT unit to request)
            uint32_t pa_off[IA], pb_off[IB], pb_m[IB];
            int cz = 0, cy = 0, cx = 0;
            uint32_t lp = 0;                               // row-list form: LDS address of this lane's first (row, flags) pair of the cursor's unit
            u32x2_t lm[IA];
            if constexpr (FAST && ROWS) {
                static_assert(IA == IB && RPA == RPB, "one row per lane for both operands");
                lp = lds_base + NS * STAGE + (uint32_t)(wave * IA * RPA + ra) * 8u;
#pragma unroll
                for (int i = 0; i < IA; ++i) pa_off[i] = a_col[i];
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int dz = g.off + b_dz[i], dy = g.off + b_dy[i], dx = g.off + b_dx[i];
                    pb_off[i] = (uint32_t)((((dz * g.Hi + dy) * g.Wi + dx) * g.Cin + b_ci[i]) * 2);   // relative to the row's own voxel, mod 2^32
                    pb_m[i] = (dz < 0 ? 1u : 0u) | (dz > 0 ? 2u : 0u) | (dy < 0 ? 4u : 0u) | (dy > 0 ? 8u : 0u) |
                              (dx < 0 ? 16u : 0u) | (dx > 0 ? 32u : 0u) | 128u | (b_tv[i] ? 0u : 256u);
                }
            } else
            if constexpr (FAST) {
#pragma unroll
                for (int i = 0; i < IA; ++i) pa_off[i] = (uint32_t)((wave * IA + i) * RPA + ra) * (uint32_t)(g.Cout * 2) + a_col[i];
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int l = (wave * IB + i) * RPB + rb;              // voxel of this lane inside a unit = its x offset in the row part
                    const int dz = g.off + b_dz[i], dy = g.off + b_dy[i], dx = g.off + b_dx[i];
                    pb_off[i] = (uint32_t)((((dz * g.Hi + dy) * g.Wi + (l + dx)) * g.Cin + b_ci[i]) * 2);   // relative to the unit's first voxel, mod 2^32
                    pb_m[i] = (dz < 0 ? 1u : 0u) | (dz > 0 ? 2u : 0u) | (dy < 0 ? 4u : 0u) | (dy > 0 ? 8u : 0u) |
                              (l + dx < 0 ? 16u : 0u) | (l + dx >= KV ? 32u : 0u) | (b_tv[i] ? 0u : 64u);
                }
                int cb;
                vox_decode(v_begin, g, cb, cz, cy, cx);
            }
            // the source offsets of the NEXT request are computed one half earlier (prep_fast: inside the MFMA half, where the wave has
            // ~70 idle issue slots between its 32 MFMAs) and only handed to the direct-to-LDS instructions in the load half
            uint32_t voa[IA], vob[IB];
            uint32_t cur_a = v_begin * (uint32_t)(g.Cout * 2), cur_b = v_begin * (uint32_t)(g.Cin * 2);   // byte offsets of the cursor's first voxel
            uint32_t m_zy = 0;                             // border flags of the cursor's (z, y): recomputed only when the x-row changes
            auto zy_flags = [&]() { m_zy = (cz == 0 ? 1u : 0u) | (cz == g.Do - 1 ? 2u : 0u) | (cy == 0 ? 4u : 0u) | (cy == g.Ho - 1 ? 8u : 0u) | 64u; };
            if constexpr (FAST) zy_flags();
            // scalar half of the preparation (border mask + byte offsets of the cursor's unit, cursor advanced) and its VALU half
            uint32_t nx_mask = 0, nx_a = 0, nx_b = 0;
            auto prep_scalar = [&]() {
                nx_mask = m_zy | (cx == 0 ? 16u : 0u) | (cx + KV == g.Wo ? 32u : 0u);
                nx_a = cur_a; nx_b = cur_b;
                cur_a += (uint32_t)(KV * 2) * (uint32_t)g.Cout; cur_b += (uint32_t)(KV * 2) * (uint32_t)g.Cin; cx += KV;
                if (cx == g.Wo) {                             // a new x-row: every Wo / 32 units.  The empty volatile asm keeps this a real (rarely
                    asm volatile("" ::: "memory");            // taken) branch: if-converted it was ~20 scalar selects in every load half
                    cx = 0; if (++cy == g.Ho) { cy = 0; if (++cz == g.Do) cz = 0; } zy_flags();
                }
            };
            auto prep_fast = [&]() {
#pragma unroll
                for (int i = 0; i < IA; ++i) voa[i] = nx_a + pa_off[i];
#pragma unroll
                for (int i = 0; i < IB; ++i) vob[i] = (pb_m[i] & nx_mask) == 0u ? nx_b + pb_off[i] : OOB;
            };
            auto list_read = [&]() {                       // (row, flags) of this lane's rows in the cursor's unit; cursor -> next unit
#pragma unroll
                for (int i = 0; i < IA; ++i) {
                    if (i == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(lm[i]) : "v"(lp));
                    else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lm[i]) : "v"(lp), "n"(RPA * 8));
                }
                static_assert(IA <= 2, "list_read: two rows per lane");
                lp += (uint32_t)KV * 8u;
            };
            auto prep_rows = [&]() {
#pragma unroll
                for (int i = 0; i < IA; ++i) voa[i] = __umul24(lm[i][0], (uint32_t)(g.Cout * 2)) + pa_off[i];
#pragma unroll
                for (int i = 0; i < IB; ++i) vob[i] = (pb_m[i] & lm[i][1]) == 0u ? __umul24(lm[i][0], (uint32_t)(g.Cin * 2)) + pb_off[i] : OOB;
            };
            auto issue_fast = [&](int buf) {
                char* sA = smem + buf * STAGE;
                char* sB = sA + A_BYTES;
#pragma unroll
                for (int i = 0; i < IA; ++i) {
                    if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(sA + (wave * IA + i) * 1024), 16, (int)voa[i], 0, 0, 0);
                    else asm volatile("" :: "v"(voa[i]));
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    if constexpr (ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sB + (wave * IB + i) * 1024), 16, (int)vob[i], 0, 0, 0);
                    else asm volatile("" :: "v"(vob[i]));
                }
            };
            auto request = [&](int u) {                    // unit u -> ring slot u & 3 (units are requested in order)
                if constexpr (FAST) issue_fast(u & 3);
                else issue(v_begin + (uint32_t)u * KV, u & 3);
            };
            // prologue: units 0..2 requested; unit 0 has landed (own pieces: counted wait; the others': the barrier)
            for (int p = 0; p < 3 && p < nk; ++p) {
                if constexpr (FAST && ROWS) { list_read(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prep_rows(); }
                else if constexpr (FAST) { prep_scalar(); prep_fast(); }
                request(p);
            }
            if constexpr (FAST && ROWS) {
                for (int p = nk < 3 ? nk : 3; p < 3; ++p) list_read();     // a split of fewer than three units: keep the cursor in step
                list_read(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prep_rows();   // offsets of unit 3
            } else if constexpr (FAST) { prep_scalar(); prep_fast(); }   // offsets of unit 3, requested in the first load half
            if (nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (grp) __builtin_amdgcn_s_barrier();           // the second group runs one phase behind
            unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
            auto stamp = [&](int q) {
                if constexpr (STAMP) {
                    unsigned long long tnow;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow) :: "memory");
                    if (q >= 0) dsum[q] += tnow - tprev;
                    tprev = tnow;
                }
            };
            stamp(-1);
            auto unit = [&](int u, int slot) {
                // ---- load half: fragment reads of unit u, then unit u + 3 into the ring slot of unit u - 1 (both groups have read it)
                rd(slot);
                if constexpr (FAST && ROWS) list_read();    // rows of unit u + 4: back with the fragments, consumed in the MFMA half
                stamp(0);
                // this wave's pieces of unit u + 1 must have landed before the barrier that closes this phase: all but the two youngest
                // units in flight (steady state: one compare), fewer at the end of the split
                if (u + 3 < nk) {
                    request(u + 3);
                    stamp(1);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                } else {
                    stamp(1);
                    if (u + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp(2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                stamp(3);
                // ---- MFMA half
                __builtin_amdgcn_s_setprio(1);               // measured against no raise: 1.40 vs 1.35 PFLOP/s (and 1.30 vs 0.86 on the 64 -> 256 layer)
                // offsets of unit u + 4 (scalar cursor + ~12 VALU), scheduled among the MFMAs: the MFMA half is paced by the matrix pipe and
                // has idle issue slots, the load half is bound by its instruction count (the same scalar work at the end of the load
                // half measured 1.11 instead of 1.20 PFLOP/s)
                if constexpr (FAST && ROWS) prep_rows();
                else if constexpr (FAST) { prep_scalar(); prep_fast(); }
                mm();
                if constexpr (FAST) {
#pragma unroll
                    for (int q = 0; q < 14; ++q) {            // ONE non-MFMA instruction per gap between MFMAs (two per gap stretched the
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA cadence from 16 to ~20 cycles: tools/wgrad_phase_probe.py)
                        __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                stamp(4);
                __builtin_amdgcn_s_barrier();
                stamp(5);
            };
            if constexpr (FAST) {
                for (int u = 0; u < nk; u += 4) {
                    unit(u, 0);
                    if (u + 1 < nk) unit(u + 1, 1);
#pragma unroll
                    for (int k = 0; k < TM + TN; ++k) fa[k] += 2u * STAGE;
                    if (u + 2 < nk) unit(u + 2, 2);
                    if (u + 3 < nk) unit(u + 3, 3);
#pragma unroll
                    for (int k = 0; k < TM + TN; ++k) fa[k] -= 2u * STAGE;
                }
            } else {
                for (int u = 0; u < nk; ++u) unit(u, u & 3);
            }
            if (!grp) __builtin_amdgcn_s_barrier();
            if constexpr (STAMP) {
#ifdef DREG_PROBE
                if (lane == 0) {
                    for (int q = 0; q < 6; ++q) atomicAdd(&g_wgrad_dbg[q], dsum[q]);
                    atomicAdd(&g_wgrad_dbg[6], (unsigned long long)nk);
                    atomicAdd(&g_wgrad_dbg[7], 1ull);
                }
#endif
            }
        }
    } else
    if (v_begin < v_end) {
        // NS-deep LDS ring, loads issued 3 stages ahead; stage k is consumed after a COUNTED wait (the two younger stages stay
        // in flight across the barrier) + a raw s_barrier (a __syncthreads() here would drain the DMA queue with vmcnt(0)).
        const int nk = (int)((v_end - v_begin + KV - 1) / KV);
        if (rowocc) {
            // rowocc (dense launches with Wo % KV == 0): a 64-voxel stage lies in one output W-row; a row flagged 0 gathers only
            // zeros, its products add exactly nothing, and the stage is skipped.  The flags of this block's stages go to LDS first
            // (a global load inside the K loop would drain the DMA queue).
            uint8_t* socc = reinterpret_cast<uint8_t*>(smem + NS * STAGE);
            for (int i = t; i < nk; i += NW * 64) socc[i] = rowocc[(v_begin + (uint32_t)i * KV) / (uint32_t)g.Wo];
            __syncthreads();
            auto next = [&](int k) { while (k < nk && socc[k] == 0) ++k; return k; };
            int k = next(0), buf = 0;
            if (k < nk) issue(v_begin + (uint32_t)k * KV, 0);
            while (k < nk) {
                const int kn = next(k + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kn < nk) issue(v_begin + (uint32_t)kn * KV, buf ^ 1);
                compute(buf);
                k = kn; buf ^= 1;
            }
        } else {
        static_assert(NS >= 2 && NS <= 5 && 3 * LPS <= 63, "ring depth / counted wait range");
        for (int p = 0; p < NS - 1 && p < nk; ++p) issue(v_begin + (uint32_t)p * KV, p);
        int cur = 0, nxt = NS - 1;                       // ring positions of stage k and of stage k + NS - 1
        for (int k = 0; k < nk; ++k) {
            const int ahead = min(nk - 1 - k, NS - 2);   // stages issued after stage k that may stay in flight
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k + NS - 1 < nk) issue(v_begin + (uint32_t)(k + NS - 1) * KV, nxt);
            compute(cur);
            cur = cur + 1 == NS ? 0 : cur + 1;
            nxt = nxt + 1 == NS ? 0 : nxt + 1;
        }
        }
    }
    float* dst = part + (size_t)split * g.Cout * g.Kpad;
    const int col_l = lane & 15, rowq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * WM + i * 16 + rowq + r;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                if (n < g.Kpad) dst[(size_t)co * g.Kpad + n] = acc[i][j][r];    // the last column tile may be ragged (Kpad = 27 * 64)
            }
        }
}

// Sum the split partials [nsplit][Cout][Kpad] (K = tap*Cin + ci) into the torch-layout gradient [Cout][Cin_real][ntaps] (fp32),
// optionally accumulating.  One kernel body serves a single layer (descriptor by value) and the batched form (descriptor table:
// the partials of MANY layers, each in its own workspace, summed by one launch).
//  * k^3 layers: a workgroup owns (co, a chunk of input channels): it sums the splits of the [ntaps][chunk] patch of the packed row
//    (contiguous runs per tap), turns it through LDS and writes chunk * ntaps CONTIGUOUS floats of the torch-layout row;
//  * 1^3 layers need no transposition: contiguous elements, float4 lanes;
//  * the sum over the splits is what takes the time for the layers with few weights and hundreds of splits (a handful of
//    workgroups walking 512 partials each): there the workgroup covers fewer elements and its threads form KG groups that take
//    every KG-th split, combined through LDS in group order — the result does not depend on the launch shape, only on nsplit.
struct WgradReduceDesc {
    const float* part; float* dw;
    int nsplit, Cout, Kpad, ntaps, Cin, Cin_real, accumulate, block0;
};
static_assert(sizeof(WgradReduceDesc) == 48, "descriptor layout is part of the ABI");
constexpr int WR_STAGE = 8192;                              // floats of LDS behind one workgroup
__host__ __device__ static inline int wr_epb(int nsplit) { return nsplit < 8 ? 2048 : (nsplit < 64 ? 512 : 128); }   // 1^3: elements per workgroup
__host__ __device__ static inline int wr_chunk(int nsplit) { return nsplit < 32 ? 64 : 16; }                          // k^3: input channels per workgroup
static inline int wgrad_reduce_blocks(int Cout, int Cin_real, int ntaps, int nsplit)
{
    if (ntaps == 1) return (int)(((size_t)Cout * Cin_real + wr_epb(nsplit) - 1) / wr_epb(nsplit));
    return Cout * ((Cin_real + wr_chunk(nsplit) - 1) / wr_chunk(nsplit));
}
__device__ __forceinline__ float wr_sum(const float* p, size_t slab, int k0, int kstep, int nsplit)
{
    float a = 0.f;
#pragma unroll 4
    for (int k = k0; k < nsplit; k += kstep) a += p[(size_t)k * slab];
    return a;
}
__device__ __forceinline__ void wgrad_reduce_block(const WgradReduceDesc& d, int r, float* stage)
{
    const int t = threadIdx.x;
    const size_t slab = (size_t)d.Cout * d.Kpad;
    // accumulate bit 1: the slices actually written are counted behind them (row-list launches: wgrad_row_splits); the partition of the
    // work below follows the table's nsplit, the sums the written count
    int nwritten = d.nsplit;
    if (d.accumulate & 2) {
        const int e = *reinterpret_cast<const int*>(d.part + (size_t)d.nsplit * slab);
        nwritten = e < 1 ? 1 : (e < d.nsplit ? e : d.nsplit);
    }
    const bool accum = (d.accumulate & 1) != 0;
    if (d.ntaps == 1) {
        const int epb = wr_epb(d.nsplit), KG = 2048 / epb, EL = 256 / KG, g = t / EL, l = t - g * EL;
        const size_t total = (size_t)d.Cout * d.Cin_real;
        const bool vec = (d.Cin_real & 3) == 0;
        const int per = vec ? 4 : 1, passes = epb / (EL * per);
        for (int ps = 0; ps < passes; ++ps) {
            const size_t i = (size_t)r * epb + (size_t)(ps * EL + l) * per;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                const int co = (int)(i / d.Cin_real), ci = (int)(i - (size_t)co * d.Cin_real);
                const float* src = d.part + (size_t)co * d.Kpad + ci;
                if (vec) {
#pragma unroll 4
                    for (int k = g; k < nwritten; k += KG) {
                        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * slab);
                        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                    }
                } else a.x = wr_sum(src, slab, g, KG, nwritten);
            }
            if (KG > 1) {
                __syncthreads();
                reinterpret_cast<float4*>(stage)[g * EL + l] = a;
                __syncthreads();
                if (g == 0) {
                    a = reinterpret_cast<float4*>(stage)[l];
                    for (int q = 1; q < KG; ++q) { const float4 v = reinterpret_cast<float4*>(stage)[q * EL + l]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
                }
            }
            if (g == 0 && i < total) {
                if (vec) {
                    float4* o = reinterpret_cast<float4*>(d.dw + i);
                    if (accum) { const float4 q = *o; a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w; }
                    *o = a;
                } else d.dw[i] = accum ? d.dw[i] + a.x : a.x;
            }
        }
        return;
    }
    const int cc = wr_chunk(d.nsplit);
    const int nchunk = (d.Cin_real + cc - 1) / cc;
    const int co = r / nchunk, ci0 = (r - co * nchunk) * cc;
    const int nci = min(cc, d.Cin_real - ci0);
    const int chp = d.Cin < cc ? d.Cin : cc;                  // padded channels of the patch row (a power of two for k > 1)
    const int sh = 31 - __builtin_clz(chp);
    const int ne = d.ntaps << sh;                             // patch elements, ordered (tap, channel)
    const int no = nci * d.ntaps;                             // output elements, ordered (channel, tap)
    const int KG = (d.nsplit >= 8 && 4 * no <= WR_STAGE) ? 4 : 1, EL = 256 / KG, g = t / EL, l = t - g * EL;
    const float* src = d.part + (size_t)co * d.Kpad + ci0;
    if ((chp & 3) == 0 && (d.Cin & 3) == 0 && (d.Kpad & 3) == 0) {
        // four channels of a tap per thread and split (16-byte loads; every element's splits are still added in ascending order: the sums
        // are those of the scalar form below, bit for bit — the scalar form issued four times the loads and ran at 1.8 TB/s)
        for (int e = l; e < (ne >> 2); e += EL) {
            const int tap = (e << 2) >> sh, cl = (e << 2) & (chp - 1);
            if (cl >= nci) continue;
            const float* p = src + (size_t)tap * d.Cin + cl;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int k = g; k < nwritten; k += KG) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * slab);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            float* q = stage + g * no + cl * d.ntaps + tap;
            q[0] = a.x;
            if (cl + 1 < nci) q[d.ntaps] = a.y;
            if (cl + 2 < nci) q[2 * d.ntaps] = a.z;
            if (cl + 3 < nci) q[3 * d.ntaps] = a.w;
        }
    } else
    for (int e = l; e < ne; e += EL) {
        const int tap = e >> sh, cl = e & (chp - 1);
        if (cl < nci) stage[g * no + cl * d.ntaps + tap] = wr_sum(src + (size_t)tap * d.Cin + cl, slab, g, KG, nwritten);   // lanes walk cl: stride ntaps (odd)
    }
    __syncthreads();
    float* o = d.dw + ((size_t)co * d.Cin_real + ci0) * d.ntaps;
    for (int j = t; j < no; j += 256) {
        float a = stage[j];
        for (int q = 1; q < KG; ++q) a += stage[q * no + j];
        o[j] = accum ? o[j] + a : a;
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceDesc d)
{
    __shared__ __align__(16) float stage[WR_STAGE];
    wgrad_reduce_block(d, (int)blockIdx.x, stage);
}
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgradReduceDesc* __restrict__ descs, int n, int block_base)
{
    __shared__ __align__(16) float stage[WR_STAGE];
    const int blk = (int)blockIdx.x + block_base;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= blk) lo = mid; else hi = mid - 1;
    }
    const WgradReduceDesc d = descs[lo];
    wgrad_reduce_block(d, blk - d.block0, stage);
}

// torch weight [Cout][Cin_real][ntaps] fp32 -> gather-form pack [Cout][Kpad] (K = tap*Cin + ci), zero padded.
// finish of a split-K convolution: out[i] = cast([relu](sum_s part[s][i] + bias[i % Cout])), 8 channels per thread
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, TO* __restrict__ out, const float* __restrict__ bias,
                                     size_t total8, size_t slice, int Cout, int nsplit, int relu)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll 4                                 // the slices' loads are independent: four slices in flight, added in slice order
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 a = *reinterpret_cast<const float4*>(part + sp * slice + i * 8), b = *reinterpret_cast<const float4*>(part + sp * slice + i * 8 + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        const int n = (int)((i * 8) % (size_t)Cout);
        if (bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(out + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

template <typename T>
__global__ void pack_weight_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin_real,
                                       int ntaps, int Cin, int log2Cin, int Kpad)
{
    const size_t total = (size_t)Cout * Kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), co = (int)(i / Kpad);
        const int tap = k / Cin, ci = k - tap * Cin;
        float v = (tap < ntaps && ci < Cin_real) ? w[((size_t)co * Cin_real + ci) * ntaps + tap] : 0.f;
        Elem<T>::st(out + i, v);
    }
}
// torch weight -> "parity class" pack of the stride-2 data gradient (pad = (ksz-1)/2, ksz 1 or 3): row (p, ci), p = 4pz+2py+px,
// K' = j*Cout + co over the 2^3 taps j = 4jz+2jy+jx (ksz 3) or the single tap (ksz 1, class 0 only);
// element = W[co][ci][d] with d_a = p_a + pad - 2 j_a when 0 <= d_a < ksz on every axis, else 0.
__device__ __forceinline__ int s2_class_tap(int p, int j, int ksz)
{
    if (ksz == 1) return (p == 0 && j == 0) ? 0 : -1;
    const int pad = ksz >> 1;
    const int dz = (p >> 2) + pad - 2 * (j >> 2), dy = ((p >> 1) & 1) + pad - 2 * ((j >> 1) & 1), dx = (p & 1) + pad - 2 * (j & 1);
    if ((unsigned)dz >= (unsigned)ksz || (unsigned)dy >= (unsigned)ksz || (unsigned)dx >= (unsigned)ksz) return -1;
    return (dz * ksz + dy) * ksz + dx;
}
template <typename T>
__global__ void pack_weight_s2class_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin_real, int ksz, int Kpad)
{
    const int ntaps = ksz * ksz * ksz, nj = ksz == 1 ? 1 : 8, ncls = ksz == 1 ? 1 : 8;
    const size_t total = (size_t)ncls * Cin_real * Kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), r = (int)(i / Kpad);
        const int p = r / Cin_real, ci = r - p * Cin_real;
        const int j = k / Cout, co = k - j * Cout;
        const int tap = j < nj ? s2_class_tap(p, j, ksz) : -1;
        Elem<T>::st(out + i, tap >= 0 ? w[((size_t)co * Cin_real + ci) * ntaps + tap] : 0.f);
    }
}
// All weight packs of a step in ONE launch (they are re-made after every optimizer update).  One block per packed row
// (block -> descriptor by binary search over the row prefix).  bf16 packs take one of three coalesced routes:
//   1^3 forward pack      : the row is a contiguous fp32 span -> 8 elements per thread, 16-byte stores;
//   1^3 data-gradient pack: a [Cout][Cin] -> [Cin][Cout] transpose; the first block of every 32 rows moves the band through
//                           64x32 LDS tiles (the other 31 blocks exit);
//   k^3 packs             : 64 channels x taps at a time through LDS (coalesced reads: one contiguous span for the forward
//                           pack, Cout spans of `taps` floats for the data-gradient / stride-2 class packs), then one
//                           16-byte store per (tap, 8 channels).
// fp32 packs (exact-f32 parity mode) read global memory element by element.  Same element maps as the single-layer kernels.
struct PackDesc {
    const float* w; void* out;
    int Cout, Cin_real, Cin, ntaps, for_dgrad, Kpad, dtype, block0;
};
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackDesc* __restrict__ descs, int n, const int* __restrict__ row_desc)
{
    extern __shared__ __align__(16) float stage[];
    __shared__ float tile[64][33];
    int lo = 0, hi = n - 1;
    if (row_desc) lo = row_desc[blockIdx.x];             // one load instead of a chain of log2(n) dependent ones
    else while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = descs[lo];
    int r = (int)blockIdx.x - d.block0;                 // co (forward pack), ci (data-gradient pack) or (class, ci)
    const int t = threadIdx.x;
    const int inner = d.for_dgrad ? d.Cout : d.Cin;     // channels per tap in the packed row
    const int nreal = d.for_dgrad ? d.Cout : d.Cin_real;
    const int ksz = d.ntaps == 1 ? 1 : (d.ntaps == 27 ? 3 : 5);
    if (d.dtype != 0 || (inner & 7)) {                   // generic route
        const int cls = d.for_dgrad == 2 ? r / d.Cin_real : 0;
        if (d.for_dgrad == 2) r -= cls * d.Cin_real;
        const size_t row_o = ((size_t)cls * d.Cin_real + r) * d.Kpad;
        for (int k = t; k < d.Kpad; k += 256) {
            int tap = k / inner;
            const int c = k - tap * inner;
            if (d.for_dgrad == 2) tap = tap < (ksz == 1 ? 1 : 8) ? s2_class_tap(cls, tap, ksz) : -1;
            float v = 0.f;
            if (tap >= 0 && tap < d.ntaps && c < nreal)
                v = d.for_dgrad ? d.w[((size_t)c * d.Cin_real + r) * d.ntaps + tap] : d.w[((size_t)r * d.Cin_real + c) * d.ntaps + tap];
            if (d.dtype == 0) Elem<bf16_t>::st((bf16_t*)d.out + row_o + k, v);
            else ((float*)d.out)[row_o + k] = v;
        }
        return;
    }
    bf16_t* out = (bf16_t*)d.out;
    if (d.ntaps == 1 && !d.for_dgrad) {                  // contiguous fp32 row -> bf16 row
        const float* src = d.w + (size_t)r * d.Cin_real;
        for (int k = t * 8; k < d.Kpad; k += 2048) {
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = k + 2 * e < d.Cin_real ? src[k + 2 * e] : 0.f, v1 = k + 2 * e + 1 < d.Cin_real ? src[k + 2 * e + 1] : 0.f;
                pk[e] = f2bf2(v0, v1);
            }
            *reinterpret_cast<uint4*>(out + (size_t)r * d.Kpad + k) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        return;
    }
    if (d.ntaps == 1) {                                  // [Cout][Cin] -> [Cin][Kpad] (data-gradient pack and class-0 pack)
        // the 32 workgroups of a 32-row group share its 64-column chunks (one chunk each for Cout = 2,048) instead of one of them
        // walking all of them: the pack is latency-bound, so what counts is how many 8 KB tiles are in flight
        const int sub = r & 31;
        r -= sub;
        const int nb = min(32, d.Cin_real - r);          // workgroups of this group (a ragged last group has fewer)
        const int ci_l = t & 31, ci_w = t >> 3, cg = t & 7;
        for (int co0 = sub * 64; co0 < d.Kpad; co0 += nb * 64) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int co = co0 + (t >> 5) + 8 * it;
                tile[(t >> 5) + 8 * it][ci_l] = (co < d.Cout && r + ci_l < d.Cin_real) ? d.w[(size_t)co * d.Cin_real + r + ci_l] : 0.f;
            }
            __syncthreads();
            if (r + ci_w < d.Cin_real) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pk[e] = f2bf2(tile[cg * 8 + 2 * e][ci_w], tile[cg * 8 + 2 * e + 1][ci_w]);
                *reinterpret_cast<uint4*>(out + (size_t)(r + ci_w) * d.Kpad + co0 + cg * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
            __syncthreads();
        }
        return;
    }
    // k^3 packs: CH channels x taps per pass (256 for 3^3: 27 KB of the fp32 row in flight per workgroup and pass — the pack is
    // latency-bound — 64 for the 5^3 stem)
    const int cls = d.for_dgrad == 2 ? r / d.Cin_real : 0;
    if (d.for_dgrad == 2) r -= cls * d.Cin_real;
    const size_t row_o = ((size_t)cls * d.Cin_real + r) * d.Kpad;
    const int ntd = d.for_dgrad == 2 ? 8 : d.ntaps;     // taps of the packed row
    const int CH = d.ntaps <= 27 ? 256 : 64;
    const int ngr = CH >> 3;                             // 8-channel granules per pass
    for (int c0 = 0; c0 < inner; c0 += CH) {
        const int nc = min(CH, nreal - c0);              // real channels in this chunk (<= 0: padding only)
        if (nc > 0) {
            if (!d.for_dgrad) {
                const float* src = d.w + ((size_t)r * d.Cin_real + c0) * d.ntaps;
                const int n = nc * d.ntaps;
                if (((size_t)src & 15) == 0) {
                    for (int i = t * 4; i + 3 < n; i += 1024) *reinterpret_cast<float4*>(stage + i) = *reinterpret_cast<const float4*>(src + i);
                    for (int i = (n & ~3) + t; i < n; i += 256) stage[i] = src[i];
                } else
                    for (int i = t; i < n; i += 256) stage[i] = src[i];
            } else {
                for (int cl = t >> 5; cl < nc; cl += 8)
                    for (int tap = t & 31; tap < d.ntaps; tap += 32)
                        stage[cl * d.ntaps + tap] = d.w[((size_t)(c0 + cl) * d.Cin_real + r) * d.ntaps + tap];
            }
        }
        __syncthreads();
        for (int it = t; it < ngr * ntd; it += 256) {
            const int g = it % ngr, tp = it / ngr;
            if (c0 + g * 8 >= inner) continue;
            const int tap = d.for_dgrad == 2 ? s2_class_tap(cls, tp, ksz) : tp;
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cl = g * 8 + 2 * e;
                const float v0 = (tap >= 0 && cl < nc) ? stage[cl * d.ntaps + tap] : 0.f;
                const float v1 = (tap >= 0 && cl + 1 < nc) ? stage[(cl + 1) * d.ntaps + tap] : 0.f;
                pk[e] = f2bf2(v0, v1);
            }
            *reinterpret_cast<uint4*>(out + row_o + (size_t)tp * inner + c0 + g * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        __syncthreads();
    }
    for (int k = ntd * inner + t * 8; k < d.Kpad; k += 2048) *reinterpret_cast<uint4*>(out + row_o + k) = make_uint4(0, 0, 0, 0);
}
// torch weight -> data-gradient pack [Cin_real][Kpad'] with K' = tap*Cout + co
template <typename T>
__global__ void pack_weight_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin_real,
                                         int ntaps, int log2Cout, int Kpad)
{
    const size_t total = (size_t)Cin_real * Kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), ci = (int)(i / Kpad);
        const int tap = k / Cout, co = k - tap * Cout;
        float v = (tap < ntaps) ? w[((size_t)co * Cin_real + ci) * ntaps + tap] : 0.f;
        Elem<T>::st(out + i, v);
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int fill_geom(ConvGeom& g, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                     int ksz, int stride, int pad, int transposed, int esize)
{
    if ((ksz != 1 && !is_pow2(Cin)) || (Cin * esize) % 16 != 0 || !(ksz == 1 || ksz == 2 || ksz == 3 || ksz == 5) || !(stride == 1 || stride == 2)) return DREG_EINVAL;
    g.B = B; g.Di = Di; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.log2Cin = (ksz == 1) ? 30 : ilog2(Cin);
    g.Cmask = (1 << g.log2Cin) - 1;
    g.Do = Do; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout;
    g.ksz = ksz; g.ntaps = ksz * ksz * ksz;
    if (!transposed) { g.sn = stride; g.dsign = 1; g.off = -pad; g.sd = 1; }
    else { g.sn = 1; g.dsign = -1; g.off = pad; g.sd = stride; }
    const uint64_t M = (uint64_t)B * Do * Ho * Wo;
    const uint64_t dmax = (uint64_t)(Wo > Ho ? (Wo > Do ? Wo : Do) : (Ho > Do ? Ho : Do));
    if (M * dmax >= (1ull << 32)) return DREG_EINVAL;
    if ((uint64_t)B * Di * Hi * Wi * Cin * esize >= (1ull << 40)) return DREG_EINVAL;
    g.M = (uint32_t)M;
    g.magW = host_magic(Wo); g.magH = host_magic(Ho); g.magD = host_magic(Do);
    const int bke = 128 / esize;
    g.Kpad = ((g.ntaps * Cin + bke - 1) / bke) * bke;
    return DREG_OK;
}

DREG_KNOB(int, g_use_glds, 1);
DREG_KNOB(int, g_wgrad_pipe, 0);    // tuning (include/dreg_nerf_probe.h): the dense 8-wave weight-gradient tile reads its fragments one MFMA group ahead (measured: no gain)
DREG_KNOB(int, g_wgrad_ring, 3);    // tuning (include/dreg_nerf_probe.h): dense 8-wave weight-gradient tile: 3 (default) anti-phase wave groups over four 32-voxel units, 0 lockstep over two 64-voxel stages, 1 / 2 lockstep over four / five 32-voxel stages, 4..8 probes
DREG_KNOB(int, g_rows_fast, 1);     // tuning (include/dreg_nerf_probe.h): row-list weight gradients keep packed coordinates in LDS (no voxel decode per load) and use the 8-wave tile
DREG_KNOB(int, g_wgrad_big, 3);     // tuning (include/dreg_nerf_probe.h): 256-row weight-gradient tiles for large dense layers (1: 256 x 128 / 4 waves, 3: 256 x 256 / 8 waves)

// K slices of a small bf16 stride-1-gather convolution (0/1 = no split): fill the chip when the 128-row tiling leaves most CUs idle
static int conv_ksplit(const ConvGeom& g, bool has_addend)
{
    if (!g_use_glds || g_use_glds == 5 || has_addend || g.sd != 1 || g.Cin % 64 != 0 || g.ntaps > 32) return 1;
    const int bn = g.Cout % 128 == 0 ? 128 : (g.Cout % 64 == 0 ? 64 : 0);
    if (!bn) return 1;
    // the slice count must not depend on how many grids share the launch (a pair's result is independent of its batch
    // mates, bit for bit): it is derived from the per-grid voxel count at a nominal batch of 8 grids
    const long vox = (long)g.Do * g.Ho * g.Wo;
    const long rows_nominal = vox == 1 ? (long)g.M : 8L * vox;   // 1x1x1 volumes = linear layers: rows are the batch dimension
    const long tiles = ((rows_nominal + 127) / 128) * (g.Cout / bn);
    const int nk = g.ntaps * (g.Cin / 64);
    if (tiles >= 128 || nk < 32) return 1;   // short K: the second (reduce) launch costs more than the idle CUs
    long s = (256 + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

// LDS stages of the direct-to-LDS kernel.  Measured (tools/bench_small_conv.py): a deeper ring (3-4 stages) does not help even when a
// launch leaves one block per CU — the K step (~1,000 cycles for 32 MFMAs per wave) is bound by the issue cost of its 8 direct-to-LDS
// pieces per wave, not by their latency.  The default stays 2; dreg_conv_set_glds_stages forces 2..4 for experiments.
DREG_KNOB(int, g_glds_stages, 0);
static inline int glds_stages(long blocks) { (void)blocks; return g_glds_stages ? g_glds_stages : 2; }

DREG_KNOB(int, g_igemm_probe, 0);         // measurement only (tools/igemm_phase_probe.py): the 128 x 128 kernel with s_memtime stamps
DREG_KNOB(int, g_narrow_thr, 224);        // tile count below which a launch takes the narrower tiles
DREG_KNOB(int, g_pointwise_rmw_cin, 128);  // tuning (include/dreg_nerf_probe.h): see igemm_choose
DREG_KNOB(int, g_igemm_ap256, 1);         // tuning (include/dreg_nerf_probe.h): the 256 x 256 tile of large launches runs in its anti-phase form (32-channel stages)
// A split-K launch may leave its partials un-summed for the consumer to sum (the small-volume BatchNorm kernels read the fp32 slices
// directly: dreg_bn_extra.splitk_part of dreg_bn3d_fwd_ex / _bwd_ex) — asked for per call through dreg_conv3d_igemm_defer's out arguments (no
// per-thread "armed" state), taken when that launch is split-K with bf16 output and no bias / ReLU.
struct SplitkDefer { int* nsplit; size_t* slice; };   // out arguments of dreg_conv3d_igemm_defer (null = always reduce)
DREG_KNOB(int, g_igemm_ap, 256);          // tuning (include/dreg_nerf_probe.h): launches of at most this many 128-row tiles take the eight-wave anti-phase form (0: never)
DREG_KNOB(int, g_narrow_small, 2);        // tuning (include/dreg_nerf_probe.h): 128 x 64 tiles for launches of < 224 128 x 128 tiles
// Which kernel instantiation a bf16 / fp32 convolution launch runs (ONE rule set: launch_conv dispatches on it and
// dreg_conv3d_igemm_variant reports it, so profiler labels name the launched template arguments — the row rocprofv3 prints).
struct IgemmChoice { int kind, bm, bn, ap, ksplit; };   // kind 0: direct-to-LDS kernel, 1: register-staged kernel, -1: unsupported
static IgemmChoice igemm_choose(const ConvGeom& g, uint32_t nrows, bool rowlist, bool has_ws, bool has_addend, int esize)
{
    IgemmChoice c{1, 128, g.Cout % 128 == 0 ? 128 : 64, 0, 1};
    if (esize == 2) {
        const uint64_t in_bytes = (uint64_t)g.B * g.Di * g.Hi * g.Wi * g.Cin * 2, wt_bytes = (uint64_t)g.Cout * g.Kpad * 2;
        const bool fits = in_bytes < 0x7fffff00ull && wt_bytes < 0x7fffff00ull;
        const int ksplit = (has_ws && !rowlist) ? conv_ksplit(g, has_addend) : 1;
        if (ksplit > 1 && fits) {
            const int tm_ = (g.M + 127) / 128;
            c.kind = 0; c.ksplit = ksplit; c.bn = g.Cout % 128 == 0 ? 128 : 64;
            c.ap = (g_igemm_ap && tm_ * (g.Cout / c.bn) * ksplit <= g_igemm_ap) ? 1 : 0;
            return c;
        }
        if (g_use_glds && g.sd == 1 && g.Cin % 64 == 0 && g.ntaps <= 32 && fits) {
            c.kind = 0;
            const uint32_t tm_ = (nrows + 127) / 128;
            // (1^3 layers with few input channels and an addend — the accumulating data gradients of layer 1 / 2's first convolutions — are
            //  HBM-bound read-modify-write passes: the 128-row tile, two workgroups per CU and the addend fetched four rows ahead)
            const bool rmw_pointwise = g.ntaps == 1 && has_addend && g.Cin <= g_pointwise_rmw_cin;
            if (g.Cout % 256 == 0 && (g_use_glds == 1 || g_use_glds == 4 || g_use_glds == 5) && nrows >= 65536 && !rmw_pointwise) { c.bm = 256; c.bn = 256; c.ap = g_igemm_ap256 ? 1 : 0; }
            else if (g.Cout % 256 == 0 && g_use_glds == 3 && nrows >= 65536) { c.bm = 128; c.bn = 256; }
            else if (g_igemm_ap && g.Cout % 128 == 0 && tm_ * (g.Cout / 128) <= (uint32_t)g_igemm_ap) { c.bn = 128; c.ap = 1; }
            else if (g_igemm_ap && g.Cout % 128 != 0 && g.Cout % 64 == 0 && tm_ * (g.Cout / 64) <= (uint32_t)g_igemm_ap) { c.bn = 64; c.ap = 1; }
            else if (g.Cout % 128 == 0 && !(g_narrow_small && tm_ * (g.Cout / 128) < (uint32_t)g_narrow_thr)) c.bn = 128;
            else if (g.Cout % 64 == 0) c.bn = 64;
            else c.kind = -1;
            return c;
        }
    }
    if (g.Cout % 128 != 0 && g.Cout % 64 != 0) c.kind = -1;      // (row lists outside the direct-to-LDS shapes: the register-staged kernel takes them too)
    return c;
}

template <typename T, typename TO>
static int launch_conv(const void* in, const void* wt, void* out, const float* bias, const void* addend,
                       const ConvGeom& g, int relu, int Da, int Ha, int Wa, int add_shift, hipStream_t st,
                       const int* rowlist = nullptr, uint32_t nrows_in = 0, float* ks_ws = nullptr, size_t ks_ws_bytes = 0,
                       const uint8_t* rowocc = nullptr, float* bn_part = nullptr, const SplitkDefer* defer = nullptr)
{
    // output-row occupancy is honoured by the register-staged kernel for plain forward gathers whose tiles are whole W-rows
    if (rowocc && (rowlist || bias || addend || relu || g.dsign != 1 || g.sd != 1 || g.Wo <= 0 || 128 % g.Wo != 0 || g.M % (uint32_t)g.Wo != 0)) rowocc = nullptr;
    const uint32_t nrows = rowlist ? nrows_in : g.M;
    const int tilesM = (nrows + 127) / 128;
    if (rowlist && nrows == 0) return DREG_OK;
    if constexpr (sizeof(T) == 2) {
        const uint64_t in_bytes = (uint64_t)g.B * g.Di * g.Hi * g.Wi * g.Cin * 2, wt_bytes = (uint64_t)g.Cout * g.Kpad * 2;
        const int ksplit = (ks_ws && !rowlist) ? conv_ksplit(g, addend != nullptr) : 1;
        if (ksplit > 1 && in_bytes < 0x7fffff00ull && wt_bytes < 0x7fffff00ull) {
            const size_t slice = (size_t)g.M * g.Cout;
            if (ks_ws_bytes < slice * ksplit * sizeof(float)) return DREG_EINVAL;
            const int tm_ = (g.M + 127) / 128;
            if (g.Cout % 128 == 0 && g_igemm_ap && tm_ * (g.Cout / 128) * ksplit <= g_igemm_ap) {
                (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<float, 128, 128, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128);
                hipLaunchKernelGGL((conv_igemm_glds_kernel<float, 128, 128, 0, 1>), dim3(tm_ * (g.Cout / 128), ksplit), dim3(512), (size_t)4 * 256 * 128, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, ks_ws, nullptr, nullptr, g, 0, 0, 0, 0, 0, g.Cout / 128,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, nullptr, g.M, ksplit, 4, nullptr);
            } else if (g.Cout % 128 != 0 && g_igemm_ap && tm_ * (g.Cout / 64) * ksplit <= g_igemm_ap) {
                (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<float, 128, 64, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 192 * 128);
                hipLaunchKernelGGL((conv_igemm_glds_kernel<float, 128, 64, 0, 1>), dim3(tm_ * (g.Cout / 64), ksplit), dim3(512), (size_t)4 * 192 * 128, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, ks_ws, nullptr, nullptr, g, 0, 0, 0, 0, 0, g.Cout / 64,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, nullptr, g.M, ksplit, 4, nullptr);
            } else
            if (g.Cout % 128 == 0) {
                const int ns = glds_stages(tm_ * (g.Cout / 128) * ksplit);
                if (ns > 2) (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<float, 128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, ns * 256 * 128);
                hipLaunchKernelGGL((conv_igemm_glds_kernel<float, 128, 128>), dim3(tm_ * (g.Cout / 128), ksplit), dim3(256), (size_t)ns * 256 * 128, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, ks_ws, nullptr, nullptr, g, 0, 0, 0, 0, 0, g.Cout / 128,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, nullptr, g.M, ksplit, ns, nullptr);
            } else {
                const int ns = glds_stages(tm_ * (g.Cout / 64) * ksplit);
                if (ns > 2) (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<float, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, ns * 192 * 128);
                hipLaunchKernelGGL((conv_igemm_glds_kernel<float, 128, 64>), dim3(tm_ * (g.Cout / 64), ksplit), dim3(256), (size_t)ns * 192 * 128, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, ks_ws, nullptr, nullptr, g, 0, 0, 0, 0, 0, g.Cout / 64,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, nullptr, g.M, ksplit, ns, nullptr);
            }
            DREG_LAUNCH_CHECK();
            if (defer && !bias && !relu && sizeof(TO) == 2) {      // the consumer sums the slices (dreg_conv3d_igemm_defer's out arguments tell it)
                *defer->nsplit = ksplit; *defer->slice = slice;
                return DREG_OK;
            }
            const size_t total8 = slice / 8;
            const int nb = (int)((total8 + 255) / 256 > 2048 ? 2048 : (total8 + 255) / 256);
            hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3(nb), dim3(256), 0, st, ks_ws, (TO*)out, bias, total8, slice, g.Cout, ksplit, relu);
            DREG_LAUNCH_CHECK();
            return DREG_OK;
        }
        if (g_use_glds && g.sd == 1 && g.Cin % 64 == 0 && g.ntaps <= 32 && in_bytes < 0x7fffff00ull && wt_bytes < 0x7fffff00ull) {
#define GL_LAUNCH(BMv, BNv, NT) do { \
                const int tm_ = (nrows + BMv - 1) / BMv, tn_ = g.Cout / BNv; \
                const int ns_ = BNv == 256 ? 2 : glds_stages(tm_ * tn_); \
                const size_t lds_ = (size_t)ns_ * (BMv + BNv) * 128; \
                if (lds_ > 65536) (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<TO, BMv, BNv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
                hipLaunchKernelGGL((conv_igemm_glds_kernel<TO, BMv, BNv>), dim3(tm_ * tn_), dim3(NT), lds_, st, \
                                   (const bf16_t*)in, (const bf16_t*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tn_, \
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, rowlist, nrows, 1, ns_, bn_part); } while (0)
            const IgemmChoice ch = igemm_choose(g, nrows, rowlist != nullptr, false, addend != nullptr, 2);
            if (ch.bm == 256 && ch.ap) {
                const int tm_ = (nrows + 255) / 256, tn_ = g.Cout / 256;
                const size_t lds_ = (size_t)4 * (256 + 256) * 64;
                (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<TO, 256, 256, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);
                hipLaunchKernelGGL((conv_igemm_glds_kernel<TO, 256, 256, 0, 1>), dim3(tm_ * tn_), dim3(512), lds_, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tn_,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, rowlist, nrows, 1, 4, bn_part);
            }
            else if (ch.bm == 256) GL_LAUNCH(256, 256, 512);
            else if (ch.bn == 256) GL_LAUNCH(128, 256, 512);
            // fewer 128 x 128 tiles than CUs (the point-set half's linear layers: ~77 row tiles x 2): half-width tiles put twice as
            // many workgroups on the chip
            else if (g_igemm_probe && g.Cout % 128 == 0 && sizeof(TO) == 2) {
                const int tm_ = (nrows + 127) / 128, tn_ = g.Cout / 128;
                hipLaunchKernelGGL((conv_igemm_glds_kernel<TO, 128, 128, 1>), dim3(tm_ * tn_), dim3(256), (size_t)2 * 256 * 128, st,
                                   (const bf16_t*)in, (const bf16_t*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tn_,
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, rowlist, nrows, 1, 2, bn_part);
            }
#define GL_LAUNCH_AP(BNv) do { \
                const int tm_ = (nrows + 127) / 128, tn_ = g.Cout / BNv; \
                const size_t lds_ = (size_t)4 * (128 + BNv) * 128; \
                (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<TO, 128, BNv, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
                hipLaunchKernelGGL((conv_igemm_glds_kernel<TO, 128, BNv, 0, 1>), dim3(tm_ * tn_), dim3(512), lds_, st, \
                                   (const bf16_t*)in, (const bf16_t*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tn_, \
                                   (uint32_t)in_bytes, (uint32_t)wt_bytes, rowlist, nrows, 1, 4, bn_part); } while (0)
            // launches that put at most one or two workgroups on a CU: the eight-wave anti-phase form of the same tile (bit-identical)
            else if (ch.kind == 0 && ch.ap && ch.bn == 128) GL_LAUNCH_AP(128);
            else if (ch.kind == 0 && ch.ap && ch.bn == 64) GL_LAUNCH_AP(64);
            else if (ch.kind == 0 && ch.bn == 128) GL_LAUNCH(128, 128, 256);
            else if (ch.kind == 0 && ch.bn == 64) GL_LAUNCH(128, 64, 256);
            else return DREG_EINVAL;
#undef GL_LAUNCH
#undef GL_LAUNCH_AP
            DREG_LAUNCH_CHECK();
            return DREG_OK;
        }
    }
    // (row lists: the direct-to-LDS kernels above; shapes they do not take — the stem: 5^3 taps, stride 2, 8 input channels — run the
    //  register-staged kernel on the list)
    if (g.Cout % 128 == 0) {
        const int tilesN = g.Cout / 128;
        const size_t lds = 2 * (128 + 128) * 128;
        hipLaunchKernelGGL((conv_igemm_kernel<T, TO, 128>), dim3(tilesM * tilesN), dim3(256), lds, st,
                           (const T*)in, (const T*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tilesN, rowocc, rowlist, nrows);
    } else if (g.Cout % 64 == 0) {
        const int tilesN = g.Cout / 64;
        const size_t lds = 2 * (128 + 64) * 128;
        hipLaunchKernelGGL((conv_igemm_kernel<T, TO, 64>), dim3(tilesM * tilesN), dim3(256), lds, st,
                           (const T*)in, (const T*)wt, (TO*)out, bias, (const TO*)addend, g, relu, Da, Ha, Wa, add_shift, tilesN, rowocc, rowlist, nrows);
    } else return DREG_EINVAL;
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

extern "C" {

int dreg_conv3d_igemm_occ(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                          int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                          int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                          int dtype, int out_f32, void* workspace, size_t workspace_bytes, const uint8_t* rowocc, void* stream);

// dtype: 0 = bf16 activations/weights (fp32 accumulate), 1 = fp32 (exact-f32 MFMA).  out_f32: bf16 inputs, fp32 output.
// transposed = 0: out[b,o,:] = sum_d in[b, o*stride - pad + d, :] . W[:, d, :]           (forward)
// transposed = 1: out[b,i,:] = sum_d in[b, (i + pad - d)/stride, :] . W'[:, d, :]       (data gradient; "in" = dOut)
// addend (optional, same dtype as out): [B, Da, Ha, Wa, Cout] added with nearest x2 upsampling (FPN top-down path),
// or element-wise when add_same = 1 (residual connections of the transformer, Da,Ha,Wa = Do,Ho,Wo).
// workspace (optional): fp32 scratch of dreg_conv3d_igemm_workspace_bytes(...) bytes; with it, small row spaces (the 8^3 / 4^3
// levels of the ResNet) run split-K over blockIdx.y and are finished by a reduce + bias/ReLU/cast pass.
size_t dreg_conv3d_igemm_workspace_bytes(int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                                         int ksz, int stride, int pad, int transposed, int has_addend, int dtype)
{
    if (dtype != 0) return 0;
    ConvGeom g;
    if (fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, 2)) return 0;
    const int s = conv_ksplit(g, has_addend != 0);
    return s > 1 ? (size_t)s * g.M * Cout * sizeof(float) : 0;
}
int dreg_conv3d_igemm_ws(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                         int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                         int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                         int dtype, int out_f32, void* workspace, size_t workspace_bytes, void* stream)
{
    return dreg_conv3d_igemm_occ(in, wt_packed, out, bias, addend, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, relu,
                                 Da, Ha, Wa, add_same, dtype, out_f32, workspace, workspace_bytes, nullptr, stream);
}
// the same with output-row occupancy flags rowocc byte [B, Do, Ho] (dreg_conv_row_occupancy; may be null): W-rows flagged 0 have an
// all-zero receptive field and are written as zeros without being computed (forward, no bias / addend / ReLU) — same result.
int dreg_conv3d_igemm_occ(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                          int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                          int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                          int dtype, int out_f32, void* workspace, size_t workspace_bytes, const uint8_t* rowocc, void* stream)
{
    const int add_shift = add_same ? 0 : 1;
    ConvGeom g;
    const int es = dtype == 0 ? 2 : 4;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, es);
    if (rc) return rc;
    if (g.M == 0) return DREG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (out_f32) return launch_conv<bf16_t, float>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_shift, st, nullptr, 0, (float*)workspace, workspace_bytes, rowocc);
        return launch_conv<bf16_t, bf16_t>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_shift, st, nullptr, 0, (float*)workspace, workspace_bytes, rowocc);
    }
    return launch_conv<float, float>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_shift, st, nullptr, 0, nullptr, 0, rowocc);
}
// Forward bf16 convolution that also leaves the BatchNorm statistics of its OUTPUT behind: bn_partial [B][V / *rows_per_chunk][Cout][2]
// fp32 = per-chunk (sum, sum of squares) of the stored bf16 values per grid and channel (V = Do*Ho*Wo), the input of
// dreg_bn3d_fwd_from_sums.  Which launches can do it follows the dispatch rules (the direct-to-LDS kernel without split-K, V a
// multiple of its row tile): *rows_per_chunk = 128 when the sums were written, 0 when not (the caller then runs the ordinary
// BatchNorm, whose first pass re-reads the tensor).  Same output as dreg_conv3d_igemm_ws, bit for bit; the sums of a grid do not
// depend on the tile shape the launch's size selects (every form adds a chunk's rows in one fixed order).
DREG_KNOB(int, g_bn_stats_epilogue, 1);   // probe knob (include/dreg_nerf_probe.h)
#ifdef DREG_PROBE
void dreg_conv_set_bn_stats_epilogue(int on) { g_bn_stats_epilogue = on ? 1 : 0; }
#endif
int dreg_conv3d_igemm_bnstats(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                              int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                              int ksz, int stride, int pad, int relu, int Da, int Ha, int Wa, int add_same,
                              void* workspace, size_t workspace_bytes, float* bn_partial, int* rows_per_chunk, void* stream)
{
    if (!rows_per_chunk) return DREG_EINVAL;
    *rows_per_chunk = 0;
    ConvGeom g;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, 0, 2);
    if (rc) return rc;
    if (g.M == 0) return DREG_OK;
    const IgemmChoice c = igemm_choose(g, g.M, false, workspace != nullptr, addend != nullptr, 2);
    const int V = Do * Ho * Wo;
    const bool emit = g_bn_stats_epilogue && bn_partial && c.kind == 0 && c.ksplit == 1 && V % c.bm == 0 && V % 128 == 0 && ((c.bm == 128 && c.bn <= 128) || (c.bm == 256 && c.bn == 256));   // (LDS behind the 128 x 256 tile is too small for the class sums)
    rc = launch_conv<bf16_t, bf16_t>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_same ? 0 : 1, (hipStream_t)stream, nullptr, 0,
                                     (float*)workspace, workspace_bytes, nullptr, emit ? bn_partial : nullptr);
    if (rc == DREG_OK && emit) *rows_per_chunk = 128;     // every tile shape writes 128-row chunk sums in one canonical order
    return rc;
}
int dreg_conv3d_igemm(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                      int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                      int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                      int dtype, int out_f32, void* stream)
{
    return dreg_conv3d_igemm_ws(in, wt_packed, out, bias, addend, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, relu,
                                Da, Ha, Wa, add_same, dtype, out_f32, nullptr, 0, stream);
}

static thread_local bool g_s2_accumulate_call = false;   // set by dreg_conv3d_dgrad_s2_acc around its call of dreg_conv3d_dgrad_s2
// Data gradient of a stride-2 convolution (ksz 3 / pad 1 or ksz 1 / pad 0; bf16) without the 7/8 structurally-zero taps of the
// gather form: one 2^3-tap convolution over dOut [B,Do,Ho,Wo,Cout] whose 8 x Cin output channels are the 8 parity classes
// of dIn [B,Di,Hi,Wi,Cin] (written in place by the epilogue).  wt_class_packed: dreg_pack_conv_weight(..., for_dgrad = 2).
int dreg_conv3d_dgrad_s2(const void* gout, const void* wt_class_packed, void* din, int B, int Di, int Hi, int Wi, int Cin,
                         int Do, int Ho, int Wo, int Cout, int ksz, int pad, void* stream)
{
    if (!g_use_glds) return DREG_EINVAL;   // served by the direct-to-LDS kernel only (callers check dreg_conv_get_glds)
    if (!((ksz == 3 && pad == 1) || (ksz == 1 && pad == 0)) || Cout % 64 != 0 || Cin % 8 != 0) return DREG_EINVAL;
    if (Do != (Di + 2 * pad - ksz) / 2 + 1 || Ho != (Hi + 2 * pad - ksz) / 2 + 1 || Wo != (Wi + 2 * pad - ksz) / 2 + 1) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Dc = (Di + 1) / 2, Hc = (Hi + 1) / 2, Wc = (Wi + 1) / 2;     // class lattice
    const int ncls = ksz == 1 ? 1 : 8;
    if ((ncls * Cin) % 64 != 0) return DREG_EINVAL;
    ConvGeom g;
    // class convolution: rows [B,Dc,Hc,Wc], gathered operand dOut (Cin of the geometry = Cout), taps {0,+1}^3 (or the single tap)
    int rc = fill_geom(g, B, Do, Ho, Wo, Cout, Dc, Hc, Wc, ncls * Cin, ksz == 1 ? 1 : 2, 1, 0, 0, 2);
    if (rc) return rc;
    if (g.M == 0) return DREG_OK;
    if (ksz == 1 && !g_s2_accumulate_call && dreg_fill_zero(din, (size_t)B * Di * Hi * Wi * Cin * 2, st) != DREG_OK) return DREG_ELAUNCH;   // (own fill kernel: HBM rate; hipMemsetAsync's runs 256 workgroups)
    return launch_conv<bf16_t, bf16_t>(gout, wt_class_packed, din, nullptr, g_s2_accumulate_call ? (const bf16_t*)din : nullptr, g, 0, Di, Hi, Wi, -Cin, st);
}
// The same ADDED to an existing dIn (a tensor with a second gradient contribution): in fp32 in the epilogue, one rounding; a 1^3 layer
// touches the even-coordinate voxels only (the others keep their value: no fill).
int dreg_conv3d_dgrad_s2_acc(const void* gout, const void* wt_class_packed, void* din, int B, int Di, int Hi, int Wi, int Cin,
                             int Do, int Ho, int Wo, int Cout, int ksz, int pad, void* stream)
{
    g_s2_accumulate_call = true;
    const int rc = dreg_conv3d_dgrad_s2(gout, wt_class_packed, din, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, pad, stream);
    g_s2_accumulate_call = false;
    return rc;
}

// 1 (default): bf16 stride-1 convolutions use the direct-to-LDS kernel (8-wave 256x256 tile when Cout % 256 == 0 and the row
// space is large, else 128 x {128|64}); 2: 128-row tiles only; 3: the 8-wave 128x256
// tile when Cout % 256 == 0 (measured equal to 128x128 in round 1); 4: the 8-wave 256x256 tile (128x64 per wave);
// 5: as 1 but never split-K; 0: always the register-staged kernel (A/B checks).
// dreg_conv3d_igemm_occ (bf16 in / out) that may leave a split-K launch's fp32 slices [*sk_nsplit][M * Cout] UN-SUMMED in its workspace for
// the consumer to sum (the register-resident BatchNorm kernels: dreg_bn3d_fwd_ex / dreg_bn3d_bwd_ex): *sk_nsplit = 0 when the launch
// finished its output itself (not split-K, or a bias / ReLU epilogue), else the slice count, *sk_slice = the slice length in elements.
int dreg_conv3d_igemm_defer(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                            int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                            int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                            void* workspace, size_t workspace_bytes, const uint8_t* rowocc, int* sk_nsplit, size_t* sk_slice, void* stream)
{
    if (!sk_nsplit || !sk_slice) return DREG_EINVAL;
    *sk_nsplit = 0; *sk_slice = 0;
    ConvGeom g;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, 2);
    if (rc) return rc;
    if (g.M == 0) return DREG_OK;
    const SplitkDefer d{sk_nsplit, sk_slice};
    return launch_conv<bf16_t, bf16_t>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_same ? 0 : 1, (hipStream_t)stream, nullptr, 0,
                                       (float*)workspace, workspace_bytes, rowocc, nullptr, &d);
}
#ifdef DREG_PROBE
void dreg_conv_set_glds(int enable) { g_use_glds = enable; }
void dreg_conv_set_wgrad_big(int enable) { g_wgrad_big = enable; }
// MEASUREMENT (tools/bench_bn_conv_fuse.py): out = conv1x1(relu(in * scale + shift)) with the BatchNorm + ReLU applied on the A load of the
// register-staged kernel (bf16, dense rows, Cout % 128 == 0); a_scale_shift fp32 [B][Cin][2]
int dreg_conv1_bnrelu_a_probe(const void* in, const void* wt_packed, void* out, const float* a_scale_shift, int B, int D, int H, int W, int Cin, int Cout, void* stream)
{
    ConvGeom g;
    int rc = fill_geom(g, B, D, H, W, Cin, D, H, W, Cout, 1, 1, 0, 0, 2);
    if (rc || Cout % 128 != 0 || Cin % 8 != 0) return DREG_EINVAL;
    const int tilesM = (g.M + 127) / 128, tilesN = Cout / 128;
    hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, bf16_t, 128, true>), dim3(tilesM * tilesN), dim3(256), 2 * (128 + 128) * 128, (hipStream_t)stream,
                       (const bf16_t*)in, (const bf16_t*)wt_packed, (bf16_t*)out, nullptr, nullptr, g, 0, 0, 0, 0, 0, tilesN, nullptr, nullptr, 0u, a_scale_shift);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
#endif
// measurement only: enable = 1 routes bf16 launches with Cout % 128 == 0 to the instrumented 128 x 128 kernel; read returns
// { cycles waiting for the stage's loads, in the barrier, issuing the next stage, in fragment reads + MFMAs; K steps x waves; waves } and clears them
// Which kernel a convolution launch of this shape runs (the rules launch_conv applies; for profiler labels):
//   kind * 100000000 + BM * 100000 + BN * 100 + AP * 10 + (split-K ? 1 : 0);   kind 0 = conv_igemm_glds_kernel, 1 = conv_igemm_kernel, negative = unsupported.
// nrows: 0 = dense, else the row-list length; has_ws: the caller passes a split-K workspace (dreg_conv3d_igemm_ws); dtype 0 bf16, 1 fp32.
int dreg_conv3d_igemm_variant(int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksz, int stride, int pad,
                              int transposed, int nrows, int has_ws, int has_addend, int dtype)
{
    ConvGeom g;
    if (fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, dtype == 0 ? 2 : 4)) return -1;
    const IgemmChoice c = igemm_choose(g, nrows > 0 ? (uint32_t)nrows : g.M, nrows > 0, has_ws != 0, has_addend != 0, dtype == 0 ? 2 : 4);
    if (c.kind < 0) return -1;
    return c.kind * 100000000 + c.bm * 100000 + c.bn * 100 + c.ap * 10 + (c.ksplit > 1 ? 1 : 0);
}
#ifdef DREG_PROBE
void dreg_conv_igemm_probe(int enable) { g_igemm_probe = enable; }
void dreg_conv_set_igemm_ap(int max_tiles) { g_igemm_ap = max_tiles > 0 ? max_tiles : 0; }
void dreg_conv_set_igemm_ap256(int on) { g_igemm_ap256 = on ? 1 : 0; }
void dreg_conv_set_pointwise_rmw_cin(int max_cin) { g_pointwise_rmw_cin = max_cin > 0 ? max_cin : 0; }
int dreg_conv_igemm_probe_read(unsigned long long* out6)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return DREG_ELAUNCH;
    if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_igemm_dbg), 6 * sizeof(unsigned long long)) != hipSuccess) return DREG_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_dbg), z, sizeof(z)) != hipSuccess) return DREG_ELAUNCH;
    return DREG_OK;
}
int dreg_conv_wgrad_probe_read(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return DREG_ELAUNCH;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wgrad_dbg), 8 * sizeof(unsigned long long)) != hipSuccess) return DREG_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_dbg), z, sizeof(z)) != hipSuccess) return DREG_ELAUNCH;
    return DREG_OK;
}
void dreg_conv_set_wgrad_rows_fast(int enable) { g_rows_fast = enable ? 1 : 0; }
void dreg_conv_set_wgrad_ring(int mode) { g_wgrad_ring = mode; }
void dreg_conv_set_wgrad_pipe(int enable) { g_wgrad_pipe = enable ? 1 : 0; }
#endif
int dreg_conv_get_glds(void) { return g_use_glds; }

// K padding of the packed weight row for (ntaps, Cin) at dtype.
int dreg_conv3d_kpad(int ksz, int Cin, int dtype) {
    const int bke = dtype == 0 ? 64 : 32;
    return ((ksz * ksz * ksz * Cin + bke - 1) / bke) * bke;
}

int dreg_pack_conv_weight(const float* w, void* out, int Cout, int Cin_real, int Cin, int ksz, int for_dgrad,
                          int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int ntaps = ksz * ksz * ksz;
    if (!for_dgrad) {
        if ((ksz != 1 && !is_pow2(Cin)) || Cin < Cin_real) return DREG_EINVAL;
        const int Kpad = dreg_conv3d_kpad(ksz, Cin, dtype);
        const size_t total = (size_t)Cout * Kpad;
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        if (dtype == 0) hipLaunchKernelGGL(pack_weight_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w, (bf16_t*)out, Cout, Cin_real, ntaps, Cin, ilog2(Cin), Kpad);
        else hipLaunchKernelGGL(pack_weight_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)out, Cout, Cin_real, ntaps, Cin, ilog2(Cin), Kpad);
    } else if (for_dgrad == 2) {
        if (!(ksz == 1 || ksz == 3) || Cout % 64 != 0) return DREG_EINVAL;
        const int Kpad = dreg_conv3d_kpad(ksz == 1 ? 1 : 2, Cout, dtype);
        const size_t total = (size_t)(ksz == 1 ? 1 : 8) * Cin_real * Kpad;
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        if (dtype == 0) hipLaunchKernelGGL(pack_weight_s2class_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w, (bf16_t*)out, Cout, Cin_real, ksz, Kpad);
        else hipLaunchKernelGGL(pack_weight_s2class_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)out, Cout, Cin_real, ksz, Kpad);
    } else {
        if (ksz != 1 && !is_pow2(Cout)) return DREG_EINVAL;
        const int Kpad = dreg_conv3d_kpad(ksz, Cout, dtype);
        const size_t total = (size_t)Cin_real * Kpad;
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        if (dtype == 0) hipLaunchKernelGGL(pack_weight_dgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w, (bf16_t*)out, Cout, Cin_real, ntaps, ilog2(Cout), Kpad);
        else hipLaunchKernelGGL(pack_weight_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)out, Cout, Cin_real, ntaps, ilog2(Cout), Kpad);
    }
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Batched form of dreg_pack_conv_weight: descs = DEVICE array of n 48-byte records
//   { const float* w; void* out; int Cout, Cin_real, inner (Cout for dgrad packs, padded Cin otherwise), ksz^3, for_dgrad, Kpad,
//     dtype, row0 }
// with row0 = exclusive prefix of the packed row counts (Cin_real [x8 for 3^3 class packs] for dgrad packs, Cout otherwise),
// total_rows its sum and stage_floats = max over the bf16 records with ksz > 1 of min(channels, 64) * ksz^3 (LDS staging).
// row_desc (optional, device int32 [total_rows]): record index of every packed row — saves the per-block binary search.
int dreg_pack_conv_weights_batched(const void* descs, int n, int total_rows, int stage_floats, const int* row_desc, void* stream)
{
    static_assert(sizeof(PackDesc) == 48, "descriptor layout is part of the ABI");
    if (n <= 0 || total_rows <= 0) return DREG_OK;
    if (stage_floats < 0 || (size_t)stage_floats * 4 > 48 * 1024) return DREG_EINVAL;
    // stage_floats > 0 says "k^3 packs present": the kernel stages 256 channels x 27 taps (or 64 x 125) per pass
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(total_rows), dim3(256), (size_t)(stage_floats > 0 ? 64 * 125 : 1) * 4, (hipStream_t)stream,
                       (const PackDesc*)descs, n, row_desc);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// number of voxel splits the weight-gradient kernel will use (pure function of the shape)
DREG_KNOB(int, g_force_wgrad_splits, 0);
// Workgroups the split choice aims for.  Rounds 1-5 used 3,072 (12 per CU: the weight-gradient kernel ALONE is fastest with ~1,024 voxels per split), but every split
// writes a Cout x Kpad fp32 slab that the batched sum reads back — 1.8 GB written + read per step at 3,072, more HBM time than the MFMA work of the 16^3 / 32^3
// layers, taken from a main stream whose 1^3 layers and BatchNorm passes are HBM-bound too.  On the whole step (tools/ab_step.py dreg_conv_set_wgrad_target_blocks,
// values alternated in one process, profiles/r06_ab_wgrad_target_blocks.txt): 3,072 16.10 ms, 768 16.02, 512 15.90, 384 15.87, 256 15.92, 128 16.05 on one box;
// 3,072 16.20, 640 16.07, 512 15.97, 448 15.99, 384 15.93, 320 15.97 on another.
DREG_KNOB(int, g_wgrad_target_blocks, 384);
#ifdef DREG_PROBE
void dreg_conv_set_narrow_small(int on) { if (on >= 10) { g_narrow_small = 2; g_narrow_thr = on; } else { g_narrow_small = on; g_narrow_thr = 224; } }   // 0 off, 1 forward / data gradient only, 2 weight gradients too; >= 10: mode 2 with this tile-count threshold
// tuning knob: workgroups the automatic voxel-split choice of the weight-gradient kernels aims for
void dreg_conv_set_glds_stages(int stages) { g_glds_stages = (stages >= 2 && stages <= 4) ? stages : 0; }
void dreg_conv_set_wgrad_target_blocks(int blocks) { g_wgrad_target_blocks = blocks > 0 ? blocks : 384; }
// tuning / test knob: force the number of voxel splits of the weight-gradient kernels (0 = automatic)
void dreg_conv_set_wgrad_splits(int splits) { g_force_wgrad_splits = splits; }
#endif
int dreg_conv3d_wgrad_splits(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int dtype) {
    if (g_force_wgrad_splits > 0) {
        const long Mf = (long)B * Do * Ho * Wo;
        long f = g_force_wgrad_splits;
        if (f > (Mf + 63) / 64) f = (Mf + 63) / 64;
        return (int)(f > 64 ? 64 : (f < 1 ? 1 : f));
    }
    const int es = dtype == 0 ? 2 : 4;
    const int bke = 128 / es;
    const int Kpad = ((ksz * ksz * ksz * Cin + bke - 1) / bke) * bke;
    const int bnc = (Kpad % 128 == 0 || (dtype == 0 && Kpad > 128)) ? 128 : 64;
    const long tiles = (long)((Cout % 128 == 0) ? Cout / 128 : Cout / 64) * ((Kpad + bnc - 1) / bnc);
    const long M = (long)B * Do * Ho * Wo;
    long s = (g_wgrad_target_blocks + tiles - 1) / tiles;
    // enough voxels per split that the fp32 partial tile written per block (64 KB) stays small next to its MFMA work:
    // measured optimum on MI355X is ~1024 voxels for 3^3 taps, ~512 for 1^3 (tools/bench_small_conv.py)
    const long vmin = ksz == 1 ? 512 : 1024;
    const long maxs = (M + vmin - 1) / vmin;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    // layers with a handful of weight tiles (the 1^3 convolutions of the 32^3 level: 1-2 tiles, 262,144 voxels to stream) need more
    // than 64 splits to occupy the chip at all: up to 512 while the launch stays under 1,024 workgroups (partials <= 34 MB)
    const long cap = (tiles * 64 >= 1024 || dtype != 0) ? 64 : ((1024 + tiles - 1) / tiles > 512 ? 512 : (1024 + tiles - 1) / tiles);
    if (s > cap) s = cap;
    if (s > 8) s = (s + 7) / 8 * 8;   // whole multiples of the 8 XCDs (see the block placement of the glds kernel)
    // layers that take the 8-wave 256 x 256 tile (one workgroup per CU): the launch is tiles256 x splits workgroups over 256 CUs, and a last
    // round that is mostly empty is pure loss (27 tiles x 32 splits = 3.375 rounds ran as 4: 84 %; x 56 = 5.9 rounds: 98 %).  Among the
    // multiples of 8 up to 64 with >= 2,048 voxels per split take the count with the fullest last round (ties: the larger).
    if (dtype == 0 && Cout % 256 == 0 && (Kpad % 256 == 0 || Kpad >= 1024) && M >= 65536) {
        const long t256 = (long)(Cout / 256) * ((Kpad + 255) / 256);
        long best = s;
        double best_eff = 0.0;
        for (long c = 24; c <= 64; c += 8) {
            if (M / c < 2048) break;
            const double rounds = (double)(t256 * c) / 256.0;
            const double eff = rounds / (double)((t256 * c + 255) / 256);
            if (eff >= best_eff - 1e-9) { best_eff = eff; best = c; }
        }
        if (best_eff > 0.0) s = best;
    }
    return (int)s;
}
size_t dreg_conv3d_wgrad_workspace_bytes(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int dtype) {
    const int es = dtype == 0 ? 2 : 4;
    const int bke = 128 / es;
    const int Kpad = ((ksz * ksz * ksz * Cin + bke - 1) / bke) * bke;
    // + 256: one int behind the slices, the number of slices a row-list launch actually wrote (see wgrad_row_splits)
    return (size_t)dreg_conv3d_wgrad_splits(B, Do, Ho, Wo, Cin, Cout, ksz, dtype) * Cout * Kpad * sizeof(float) + 256;
}
// Voxel splits of a ROW-LIST weight gradient: dreg_conv3d_wgrad_splits sizes the split count (and the workspace) for the dense volume,
// but an active set holds 4 - 15 % of it — 56 splits of a 28 k-row list are 500 rows each, 1,512 workgroups that spend their life in
// the prologue and the 256 KB partial tile they write (0.3 PFLOP/s).  The launch uses the first s <= smax slices; the deferred sum reads
// s from the int behind the slices (descriptor flag, WgradReduceDesc::accumulate bit 1).
//  * 8-wave 256 x 256 tile (one workgroup per CU; its slice of the (row, flags) list sits in LDS: <= 3,904 rows per split): among the
//    counts with >= 2,048 rows per split the one whose last round of 256 workgroups is fullest (ties: fewer splits = fewer partials);
//  * the four-wave tiles: at least 1,024 (3^3) / 512 (1^3) rows per split, as in the dense rule.
DREG_KNOB(bool, g_row_splits, true);      // tuning (include/dreg_nerf_probe.h): 0 = row lists use the dense volume's split count
#ifdef DREG_PROBE
extern "C" void dreg_conv_set_row_splits(int on) { g_row_splits = on != 0; }
#endif
static int wgrad_row_splits(int Cout, int Kpad, int ksz, uint32_t nrows, int smax, bool tile256)
{
    if (smax <= 1 || !g_row_splits) return smax;
    if (tile256) {
        const long t256 = (long)(Cout / 256) * ((Kpad + 255) / 256);
        const long vmax = (((long)160 * 1024 - (long)4 * 32 * 512 * 2 - 5 * 32 * 8) / 8) / 64 * 64;
        long cmin = ((long)nrows + vmax - 1) / vmax;
        if (cmin < 1) cmin = 1;
        if (cmin >= smax) return smax;
        long cmax = (long)nrows / 2048;
        if (cmax > smax) cmax = smax;
        if (cmax < cmin) cmax = cmin;
        long best = cmin;
        double best_eff = -1.0;
        for (long c = cmin; c <= cmax; ++c) {
            const double eff = (double)(t256 * c) / 256.0 / (double)((t256 * c + 255) / 256);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
        }
        return (int)best;
    }
    const long vmin = ksz == 1 ? 512 : 1024;
    long s = ((long)nrows + vmin - 1) / vmin;
    if (s < 1) s = 1;
    return (int)(s < smax ? s : smax);
}

// dW[Cout][Cin_real][ntaps] (torch layout, fp32) (+)= sum_m gout[m][:]^T x gathered in[m][tap][:]
// gout: [B,Do,Ho,Wo,Cout], in: [B,Di,Hi,Wi,Cin] (same dtype).  use_tr: 1 = LDS transpose reads (bf16 only).
static int wgrad_impl(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                      int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                      int ksz, int stride, int pad, int accumulate, int dtype, int use_tr, void* stream,
                      const int* rowlist, uint32_t nrows_list, const uint8_t* rowocc = nullptr, bool defer_reduce = false)
{
    ConvGeom g;
    const int es = dtype == 0 ? 2 : 4;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, 0, es);
    if (rc) return rc;
    if (Cout % 128 != 0 && Cout != 64) return DREG_EINVAL;
    if (workspace_bytes < dreg_conv3d_wgrad_workspace_bytes(B, Do, Ho, Wo, Cin, Cout, ksz, dtype)) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nrows = rowlist ? nrows_list : g.M;
    if (rowlist && !(dtype == 0 && use_tr && g_use_glds)) return DREG_EINVAL;
    const int smax = dreg_conv3d_wgrad_splits(B, Do, Ho, Wo, Cin, Cout, ksz, dtype);
    const bool row_tile256 = rowlist && g.sn == 1 && g.sd == 1 && g.dsign == 1 && Di == Do && Hi == Ho && Wi == Wo && Do < 1024 && Ho < 1024 && Wo < 1024 &&
                             g_rows_fast && g_wgrad_big == 3 && g_wgrad_ring == 3 && ksz <= 3 && Cout % 256 == 0 && (g.Kpad % 256 == 0 || g.Kpad >= 1024) && nrows >= 16384;
    const int nsplit = rowlist ? wgrad_row_splits(Cout, g.Kpad, ksz, nrows, smax, row_tile256) : smax;
    uint32_t vps = (uint32_t)((nrows + nsplit - 1) / nsplit);
    vps = ((vps + 63) / 64) * 64;
    if (vps == 0) vps = 64;
    int bm = (Cout % 128 == 0) ? 128 : 64;
    const uint64_t gbytes = (uint64_t)g.M * Cout * 2, ibytes = (uint64_t)B * Di * Hi * Wi * Cin * 2;
    const bool glds_path = dtype == 0 && use_tr && g_use_glds && gbytes < 0x7fffff00ull && ibytes < 0x7fffff00ull;
    // 128 columns per tile also when Kpad is an odd multiple of 64 (3^3 taps x 64 channels = 1,728): the direct-to-LDS kernel
    // masks the ragged last tile (columns >= Kpad gather zeros and are not stored); the older kernels need exact tiles
    int bnc = (g.Kpad % 128 == 0 || (glds_path && g.Kpad > 128)) ? 128 : 64;
    // launches that leave most CUs empty (the point-set half's linear layers: 4 weight tiles x 16 splits) take 64-wide tiles: up to
    // four times the workgroups, the same per-element accumulation order
    if (g_narrow_small >= 2 && dtype == 0 && (Cout / bm) * ((g.Kpad + bnc - 1) / bnc) * nsplit < g_narrow_thr) {
        if (bnc == 128) bnc = 64;
        if (bm == 128 && (Cout / bm) * (g.Kpad / bnc) * nsplit < g_narrow_thr) bm = 64;
    }
    const int tilesRow = Cout / bm;
    const int tilesCol = (g.Kpad + bnc - 1) / bnc;
    dim3 grid(tilesRow * tilesCol, nsplit);
    const size_t lds = (size_t)2 * 32 * (bm + bnc) * es;
    float* part = (float*)workspace;
    if (rowocc && (rowlist || Wo % 64 != 0)) rowocc = nullptr;   // the flags are per output W-row: stages must not straddle rows
#define WG_LAUNCH(T, BMv, BNv, TRv) hipLaunchKernelGGL((conv_wgrad_kernel<T, BMv, BNv, TRv>), grid, dim3(256), lds, st, (const T*)gout, (const T*)in, part, g, tilesCol, vps, rowocc)
#define WG_DISPATCH(T, TRv) do { \
        if (bm == 128 && bnc == 128) WG_LAUNCH(T, 128, 128, TRv); \
        else if (bm == 128 && bnc == 64) WG_LAUNCH(T, 128, 64, TRv); \
        else if (bm == 64 && bnc == 128) WG_LAUNCH(T, 64, 128, TRv); \
        else WG_LAUNCH(T, 64, 64, TRv); } while (0)
    if (glds_path) {
#define WGG_(BMv, BNv, KVv, NSv, PPv) do { if (rowlist) hipLaunchKernelGGL((conv_wgrad_glds_kernel<BMv, BNv, true, 4, 0, KVv, NSv, PPv>), dim3(tilesRow * tilesCol * nsplit), dim3(256), (size_t)2 * 64 * (bm + bnc) * 2 + (size_t)vps * (rows_fast ? 8 : 4), st, (const bf16_t*)gout, (const bf16_t*)in, part, g, tilesCol, tilesRow * tilesCol, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr, rows_fast); \
        else hipLaunchKernelGGL((conv_wgrad_glds_kernel<BMv, BNv, false, 4, 0, KVv, NSv, PPv>), dim3(tilesRow * tilesCol * nsplit), dim3(256), (size_t)2 * 64 * (bm + bnc) * 2 + (rowocc ? (size_t)(vps / KVv + 16) : 0), st, (const bf16_t*)gout, (const bf16_t*)in, part, g, tilesCol, tilesRow * tilesCol, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, rowocc); } while (0)
#define WGG(BMv, BNv) WGG_(BMv, BNv, 64, 2, 0)
        if (rowlist && vps > 20480) return DREG_EINVAL;   // the row-list slice must fit in LDS behind the stages (caller falls back to dense)
        // row lists over a stride-1 same-size volume (what the active-set head launches): 8 bytes per row in LDS (index + packed
        // coordinates) when that fits, and the 8-wave 256 x 256 tile for the 256 -> 256 layers
        const bool same_vol = g.sn == 1 && g.sd == 1 && g.dsign == 1 && Di == Do && Hi == Ho && Wi == Wo && Do < 1024 && Ho < 1024 && Wo < 1024;
        const bool rows256 = rowlist && same_vol && g_rows_fast && g_wgrad_big == 3 && Cout % 256 == 0 && g.Kpad % 256 == 0 && nrows >= 65536 &&
                             (size_t)2 * 64 * 512 * 2 + (size_t)vps * 8 <= (size_t)160 * 1024;
        const int rows_fast = (rowlist && same_vol && g_rows_fast && (rows256 || (size_t)2 * 64 * (bm + bnc) * 2 + (size_t)vps * 8 <= (size_t)160 * 1024)) ? 1 : 0;
        if (rowlist) {
            const int ldsr = 2 * 64 * (bm + bnc) * 2 + (int)vps * 4;
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<128, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<64, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<64, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);

            (void)ldsr;
        }
        // anti-phase wave groups on the row list ((row, border flags) pairs in LDS; lean load half as in the dense form).  Also for
        // lists of >= 16,384 rows (the 32^3 level of the active-set head: 28 k rows ran at 0.26 PFLOP/s on the four-wave tile) and for a K
        // extent that is not a multiple of 256 (27 taps x 64 channels: ragged last column tile, as in the dense form)
        const bool rows256_ap = rowlist && same_vol && g_rows_fast && g_wgrad_big == 3 && g_wgrad_ring == 3 && ksz <= 3 && Cout % 256 == 0 &&
                                (g.Kpad % 256 == 0 || g.Kpad >= 1024) && nrows >= 16384 &&
                                (size_t)4 * 32 * 512 * 2 + ((size_t)vps + 5 * 32) * 8 <= (size_t)160 * 1024;
        if (rows256_ap) {
            const int tiles256 = (Cout / 256) * ((g.Kpad + 255) / 256);
            const size_t l_ = (size_t)4 * 32 * 512 * 2 + ((size_t)vps + 5 * 32) * 8;
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, true, 8, 0, 32, 4, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, true, 8, 0, 32, 4, 0, 3>), dim3(tiles256 * nsplit), dim3(512), l_, st, (const bf16_t*)gout,
                               (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr, 1);
        } else
        if (rows256) {
            const int tiles256 = (Cout / 256) * (g.Kpad / 256);
            const size_t l_ = (size_t)2 * 64 * 512 * 2 + (size_t)vps * 8;
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, true, 8>), dim3(tiles256 * nsplit), dim3(512), l_, st, (const bf16_t*)gout,
                               (const bf16_t*)in, part, g, g.Kpad / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr, 1);
        } else
        // (a K extent that is not a multiple of 256 — 27 taps x 64 channels = 1,728 — takes a ragged last column tile: its columns >= Kpad
        //  gather zeros and are not stored)
        if (!rowlist && !rowocc && g_wgrad_big && Cout % 256 == 0 && (g.Kpad % 256 == 0 || (g.Kpad >= 1024 && g_wgrad_big == 3 && g_wgrad_ring >= 3 && !g_wgrad_pipe)) && nrows >= 65536) {
            // large dense layers: 256-row tiles.  Default (3): the 8-wave 256 x 256 tile (0.90 PFLOP/s on 256 -> 256 @64^3 alone); 1: 4
            // waves on 256 x 128 with 32-voxel stages — 48 KB of LDS, two independent workgroups per CU: 0.93 alone, but no faster
            // inside the step, where the data-gradient stream shares the CUs (tools/ab_step.py: 43.25 vs 43.16 ms, dense head)
            // large dense layers: the 8-wave 256 x 256 tile
            const int tiles256 = (Cout / 256) * ((g.Kpad + 255) / 256);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 * 2);
            if (g_wgrad_big == 1) {
                const int t2 = (Cout / 256) * (g.Kpad / 128);
                const size_t l_ = (size_t)2 * 32 * 384 * 2;
                uint32_t vps2 = ((vps + 31) / 32) * 32;
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 128, false, 4, 0, 32>), dim3(t2 * nsplit), dim3(256), l_, st, (const bf16_t*)gout,
                                   (const bf16_t*)in, part, g, g.Kpad / 128, t2, nsplit, vps2, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr);
            } else if (g_wgrad_big >= 11 && g_wgrad_big <= 13) {   // ablations of the 8-wave kernel
                const size_t l_ = (size_t)2 * 64 * 512 * 2;
#define WG_ABL(A) do { (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_); \
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8, A>), dim3(tiles256 * nsplit), dim3(512), l_, st, (const bf16_t*)gout, \
                                   (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, rowocc); } while (0)
                if (g_wgrad_big == 11) WG_ABL(1); else if (g_wgrad_big == 12) WG_ABL(2); else WG_ABL(3);
#undef WG_ABL
            } else if (g_wgrad_pipe && !g_wgrad_ring) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8, 0, 64, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8, 0, 64, 2, 1>), dim3(tiles256 * nsplit), dim3(512), (size_t)2 * 64 * 512 * 2 + (rowocc ? (size_t)(vps / 64 + 16) : 0), st, (const bf16_t*)gout,
                                   (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, rowocc);
            } else if (g_wgrad_ring == 2) {   // five 32-voxel stages: the whole 160 KB of LDS, four stages in flight
                (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8, 0, 32, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8, 0, 32, 5>), dim3(tiles256 * nsplit), dim3(512), (size_t)5 * 32 * 512 * 2, st, (const bf16_t*)gout,
                                   (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr);
            } else if (g_wgrad_ring >= 3 && g_wgrad_ring <= 8) {
                // anti-phase wave groups over a ring of four 32-voxel units.  3: product form (FAST when the layer qualifies, else the general
                // loop); 8: the general loop; 4..7 measurement only: FAST with s_memtime stamps, 5..7 (wrong results) without fragment
                // reads / without pieces / without MFMAs
                const bool fast_ok = ksz <= 3 && g.sn == 1 && g.sd == 1 && g.dsign == 1 && Di == Do && Hi == Ho && Wi == Wo && (Wo % 32) == 0 &&
                                     (g.M % 32u) == 0 && (vps % 32u) == 0;
#define WG_AP(A, APv) do { (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8, A, 32, 4, 0, APv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8, A, 32, 4, 0, APv>), dim3(tiles256 * nsplit), dim3(512), (size_t)4 * 32 * 512 * 2, st, (const bf16_t*)gout, \
                                   (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr, 0); } while (0)
                if (g_wgrad_ring == 8 || !fast_ok) WG_AP(0, 1);
                else if (g_wgrad_ring == 3) WG_AP(0, 3);
                else if (g_wgrad_ring == 4) WG_AP(0, 4);
                else if (g_wgrad_ring == 5) WG_AP(2, 4);
                else if (g_wgrad_ring == 6) WG_AP(3, 4);
                else WG_AP(1, 4);
#undef WG_AP
            } else if (g_wgrad_ring == 1) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<256, 256, false, 8, 0, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8, 0, 32, 4>), dim3(tiles256 * nsplit), dim3(512), (size_t)4 * 32 * 512 * 2, st, (const bf16_t*)gout,
                                   (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, nullptr);
            } else
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 256, false, 8>), dim3(tiles256 * nsplit), dim3(512), (size_t)2 * 64 * 512 * 2, st, (const bf16_t*)gout,
                               (const bf16_t*)in, part, g, (g.Kpad + 255) / 256, tiles256, nsplit, vps, (uint32_t)gbytes, (uint32_t)ibytes, rowlist, nrows, rowocc);
        } else
        if (bm == 128 && bnc == 128) WGG(128, 128); else if (bm == 128 && bnc == 64) WGG(128, 64);
        else if (bm == 64 && bnc == 128) WGG(64, 128); else WGG(64, 64);
#undef WGG
    } else
    if (dtype == 0) { if (use_tr) WG_DISPATCH(bf16_t, true); else WG_DISPATCH(bf16_t, false); }
    else WG_DISPATCH(float, false);
#undef WG_DISPATCH
#undef WG_LAUNCH
    DREG_LAUNCH_CHECK();
    if (defer_reduce) {                  // the caller sums the splits later (dreg_wgrad_reduce_batched)
        // row lists: how many slices were written, behind the slices (a 4-byte fill on the same stream)
        if (rowlist && hipMemsetD32Async((hipDeviceptr_t)(part + (size_t)smax * Cout * g.Kpad), nsplit, 1, st) != hipSuccess) return DREG_ELAUNCH;
        return DREG_OK;
    }
    if (g.ntaps > 1 && (size_t)g.ntaps * (Cin < 64 ? Cin : 64) > (size_t)WR_STAGE) return DREG_EINVAL;
    WgradReduceDesc rd{part, dw, nsplit, Cout, g.Kpad, g.ntaps, Cin, Cin_real, accumulate, 0};
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(Cout, Cin_real, g.ntaps, nsplit)), dim3(256), 0, st, rd);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// ---- weight gradients of MANY linear layers in one launch per tile shape (see GRP above).
// dreg_linear_wgrad_group_fill writes the 144-byte descriptor of y = x W^T (x: bf16 [rows, Cin], gout: bf16 [rows, Cout], split partials
// [nsplit][Cout][Kpad] fp32 into `workspace` exactly as dreg_conv3d_wgrad_partials(..., B = rows, 1^3) would leave them) into host memory
// and returns its tile shape (*variant = BM * 1000 + BNC) and workgroup count; block0 is left to the caller (exclusive prefix of the
// workgroup counts inside one variant's table).  DREG_EINVAL: the layer does not take the four-wave direct-to-LDS kernel (caller falls back).
int dreg_wgrad_group_desc_bytes() { return (int)sizeof(WgradGroupDesc); }
int dreg_conv3d_wgrad_group_fill(void* desc_host, const void* gout, const void* in, void* workspace, size_t workspace_bytes,
                                 int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksz, int stride, int pad,
                                 int* variant, int* nblocks);
int dreg_linear_wgrad_group_fill(void* desc_host, const void* gout, const void* in, void* workspace, size_t workspace_bytes,
                                 int rows, int Cin, int Cout, int* variant, int* nblocks)
{
    return dreg_conv3d_wgrad_group_fill(desc_host, gout, in, workspace, workspace_bytes, rows, 1, 1, 1, Cin, 1, 1, 1, Cout, 1, 1, 0, variant, nblocks);
}
// The same for a dense convolution layer (no row list, no occupancy flags): gout bf16 [B,Do,Ho,Wo,Cout], in bf16 [B,Di,Hi,Wi,Cin]; the
// descriptor makes the grouped launch write exactly the partials dreg_conv3d_wgrad_partials leaves for the layer.
int dreg_conv3d_wgrad_group_fill(void* desc_host, const void* gout, const void* in, void* workspace, size_t workspace_bytes,
                                 int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksz, int stride, int pad,
                                 int* variant, int* nblocks)
{
    ConvGeom g;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, 0, 2);
    if (rc) return rc;
    if ((Cout % 128 != 0 && Cout != 64) || !g_use_glds) return DREG_EINVAL;
    if (workspace_bytes < dreg_conv3d_wgrad_workspace_bytes(B, Do, Ho, Wo, Cin, Cout, ksz, 0)) return DREG_EINVAL;
    const uint32_t nrows = g.M;
    const int nsplit = dreg_conv3d_wgrad_splits(B, Do, Ho, Wo, Cin, Cout, ksz, 0);
    uint32_t vps = (uint32_t)((nrows + nsplit - 1) / nsplit);
    vps = ((vps + 63) / 64) * 64;
    if (vps == 0) vps = 64;
    int bm = (Cout % 128 == 0) ? 128 : 64;
    const uint64_t gbytes = (uint64_t)g.M * Cout * 2, ibytes = (uint64_t)B * Di * Hi * Wi * Cin * 2;
    if (gbytes >= 0x7fffff00ull || ibytes >= 0x7fffff00ull) return DREG_EINVAL;
    int bnc = (g.Kpad % 128 == 0 || g.Kpad > 128) ? 128 : 64;
    if (g_narrow_small >= 2 && (Cout / bm) * ((g.Kpad + bnc - 1) / bnc) * nsplit < g_narrow_thr) {      // the rules of wgrad_impl
        if (bnc == 128) bnc = 64;
        if (bm == 128 && (Cout / bm) * (g.Kpad / bnc) * nsplit < g_narrow_thr) bm = 64;
    }
    if (g_wgrad_big && Cout % 256 == 0 && (g.Kpad % 256 == 0 || g.Kpad >= 1024) && nrows >= 65536) return DREG_EINVAL;   // (takes the 8-wave tile)
    const int tilesRow = Cout / bm, tilesCol = (g.Kpad + bnc - 1) / bnc;
    WgradGroupDesc d{};
    d.gout = (const bf16_t*)gout; d.in = (const bf16_t*)in; d.part = (float*)workspace; d.g = g;
    d.tilesCol = tilesCol; d.tiles = tilesRow * tilesCol; d.nsplit = nsplit; d.vps = vps; d.gbytes = (uint32_t)gbytes; d.ibytes = (uint32_t)ibytes;
    d.nrows = nrows; d.block0 = 0;
    std::memcpy(desc_host, &d, sizeof(d));
    *variant = bm * 1000 + bnc;
    *nblocks = tilesRow * tilesCol * nsplit;
    return DREG_OK;
}
// descs_dev: n descriptors of ONE variant in device memory with ascending block0; total_blocks = the sum of their workgroup counts
int dreg_wgrad_group_launch(const void* descs_dev, int n, int variant, int total_blocks, void* stream)
{
    if (n <= 0 || total_blocks <= 0) return DREG_OK;
    if (!descs_dev) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int bm = variant / 1000, bnc = variant % 1000;
    const ConvGeom g0{};
#define WGRP(BMv, BNv) hipLaunchKernelGGL((conv_wgrad_glds_kernel<BMv, BNv, false, 4, 0, 64, 2, 0, 0, true>), dim3(total_blocks), dim3(256), (size_t)2 * 64 * (BMv + BNv) * 2, st, \
        (const bf16_t*)nullptr, (const bf16_t*)nullptr, (float*)nullptr, g0, 0, 0, 0, 0u, 0u, 0u, (const int*)nullptr, 0u, (const uint8_t*)descs_dev, n)
    if (bm == 128 && bnc == 128) WGRP(128, 128); else if (bm == 128 && bnc == 64) WGRP(128, 64);
    else if (bm == 64 && bnc == 128) WGRP(64, 128); else if (bm == 64 && bnc == 64) WGRP(64, 64);
    else return DREG_EINVAL;
#undef WGRP
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

int dreg_conv3d_wgrad(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                      int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                      int ksz, int stride, int pad, int accumulate, int dtype, int use_tr, void* stream)
{
    return wgrad_impl(gout, in, dw, workspace, workspace_bytes, B, Di, Hi, Wi, Cin, Cin_real, Do, Ho, Wo, Cout, ksz, stride, pad,
                      accumulate, dtype, use_tr, stream, nullptr, 0);
}
// the same with output-row occupancy flags (dreg_conv_row_occupancy): output W-rows flagged 0 have an all-zero receptive field in
// `in`, so they are left out of the reduction — same result, bit for bit.  rowocc may be null.
int dreg_conv3d_wgrad_occ(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                          int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                          int ksz, int stride, int pad, int accumulate, int dtype, int use_tr, const uint8_t* rowocc, void* stream)
{
    return wgrad_impl(gout, in, dw, workspace, workspace_bytes, B, Di, Hi, Wi, Cin, Cin_real, Do, Ho, Wo, Cout, ksz, stride, pad,
                      accumulate, dtype, use_tr, stream, nullptr, 0, rowocc);
}

// ---- active-set ("row list") forms, bf16, stride 1: only the output voxels rows[0..nrows) (ascending int32 flat indices
// b*Do*Ho*Wo + ...) are computed / reduced over; the other rows of `out` are left untouched.  Used for the two FPN head
// convolutions whose outputs are consumed only around the occupied voxels (nerf_regtr.py:138-147).
int dreg_conv3d_igemm_rows(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                           const int* rows, int nrows,
                           int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                           int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                           int out_f32, void* stream)
{
    ConvGeom g;
    int rc = fill_geom(g, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksz, stride, pad, transposed, 2);
    if (rc) return rc;
    if (!rows || nrows < 0) return DREG_EINVAL;
    const int add_shift = add_same ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (out_f32) return launch_conv<bf16_t, float>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_shift, st, rows, (uint32_t)nrows);
    return launch_conv<bf16_t, bf16_t>(in, wt_packed, out, bias, addend, g, relu, Da, Ha, Wa, add_shift, st, rows, (uint32_t)nrows);
}
// Deferred form: only the split partials are written ([nsplit][Cout][Kpad] fp32 in `workspace`, which the caller keeps until it has
// run dreg_wgrad_reduce_batched over it).  rows / rowocc as in the _rows / _occ forms (both may be null).
int dreg_conv3d_wgrad_partials(const void* gout, const void* in, void* workspace, size_t workspace_bytes, const int* rows, int nrows,
                               int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                               int ksz, int stride, int pad, const uint8_t* rowocc, void* stream)
{
    if (rows && nrows < 0) return DREG_EINVAL;
    return wgrad_impl(gout, in, nullptr, workspace, workspace_bytes, B, Di, Hi, Wi, Cin, Cin_real, Do, Ho, Wo, Cout, ksz, stride, pad,
                      0, 0, 1, stream, rows, rows ? (uint32_t)nrows : 0, rows ? nullptr : rowocc, true);
}
// Workgroups the batched reduce spends on one layer: descriptor i starts at block0 = sum of the counts of the descriptors before it.
int dreg_wgrad_reduce_blocks(int Cout, int Cin_real, int ksz, int nsplit) { return wgrad_reduce_blocks(Cout, Cin_real, ksz * ksz * ksz, nsplit); }
// `accumulate`: bit 0 = add to dw (else overwrite); bit 1 = the workspace was written by a ROW-LIST launch of dreg_conv3d_wgrad_partials, which
// uses nwritten <= nsplit slices and stores nwritten as an int behind the nsplit-th slice: the sum covers those only.
// descs_dev: n descriptors {part, dw, nsplit, Cout, Kpad, ntaps, Cin, Cin_real, accumulate, block0} (48 bytes each, device memory,
// ascending block0); blocks [block_base, block_base + nblocks) are launched, so a sub-range of a long table can be reduced on its own.
int dreg_wgrad_reduce_batched(const void* descs_dev, int n, int block_base, int nblocks, void* stream)
{
    if (n <= 0 || nblocks <= 0) return DREG_OK;
    if (!descs_dev || block_base < 0) return DREG_EINVAL;
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const WgradReduceDesc*)descs_dev, n, block_base);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
// Which bf16 weight-gradient kernel a launch of this shape runs (the same rules as wgrad_impl; for the profiler's labels):
// returns BM * 1000 + BNC (256256 = the 8-wave 256 x 256 tile).  rows: row-list launch; occ: launch with output-row occupancy flags.
int dreg_conv3d_wgrad_variant(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int rows, int nrows, int occ)
{
    const int Kpad = dreg_conv3d_kpad(ksz, Cin, 0);
    const int smax = dreg_conv3d_wgrad_splits(B, Do, Ho, Wo, Cin, Cout, ksz, 0);
    const int nsplit = rows ? wgrad_row_splits(Cout, Kpad, ksz, (uint32_t)nrows, smax, g_rows_fast && g_wgrad_big == 3 && g_wgrad_ring == 3 && ksz <= 3 && Cout % 256 == 0 &&
                                               (Kpad % 256 == 0 || Kpad >= 1024) && nrows >= 16384 && Do < 1024 && Ho < 1024 && Wo < 1024) : smax;
    const long M = (long)B * Do * Ho * Wo;
    if (!rows && !occ && g_wgrad_big && Cout % 256 == 0 && (Kpad % 256 == 0 || (Kpad >= 1024 && g_wgrad_big == 3 && g_wgrad_ring >= 3 && !g_wgrad_pipe)) && (rows ? nrows : M) >= 65536) return g_wgrad_big == 1 ? 256128 : 256256;
    if (rows && g_rows_fast && g_wgrad_big == 3 && g_wgrad_ring == 3 && ksz <= 3 && Cout % 256 == 0 && (Kpad % 256 == 0 || Kpad >= 1024) && nrows >= 16384) {   // anti-phase row-list form
        uint32_t vps = (uint32_t)((nrows + nsplit - 1) / nsplit);
        vps = ((vps + 63) / 64) * 64;
        if ((size_t)4 * 32 * 512 * 2 + ((size_t)vps + 5 * 32) * 8 <= (size_t)160 * 1024) return 256256;
    }
    if (rows && g_rows_fast && g_wgrad_big == 3 && ksz == 3 && Cout % 256 == 0 && Kpad % 256 == 0 && nrows >= 65536) {   // stride 1, same-size volume assumed
        uint32_t vps = (uint32_t)((nrows + nsplit - 1) / nsplit);
        vps = ((vps + 63) / 64) * 64;
        if ((size_t)2 * 64 * 512 * 2 + (size_t)vps * 8 <= (size_t)160 * 1024) return 256256;
    }
    int bm = (Cout % 128 == 0) ? 128 : 64;
    int bnc = (Kpad % 128 == 0 || Kpad > 128) ? 128 : 64;
    if (g_narrow_small >= 2 && (Cout / bm) * ((Kpad + bnc - 1) / bnc) * nsplit < g_narrow_thr) {
        if (bnc == 128) bnc = 64;
        if (bm == 128 && (Cout / bm) * (Kpad / bnc) * nsplit < g_narrow_thr) bm = 64;
    }
    return bm * 1000 + bnc;
}
int dreg_conv3d_wgrad_rows(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                           const int* rows, int nrows,
                           int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                           int ksz, int stride, int pad, int accumulate, void* stream)
{
    if (!rows || nrows < 0) return DREG_EINVAL;
    return wgrad_impl(gout, in, dw, workspace, workspace_bytes, B, Di, Hi, Wi, Cin, Cin_real, Do, Ho, Wo, Cout, ksz, stride, pad,
                      accumulate, 0, 1, stream, rows, (uint32_t)nrows);
}

}  // extern "C"

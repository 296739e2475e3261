// Dense 3x3x3 / stride-1 / pad-1 convolution (forward and, with the flipped-tap weight pack, its data gradient) for bf16 NDHWC
// activations with Cout = 256: the two dominant layers of the reference's FeaturePyramid_v1 head, upsample_transform_1
// (256 -> 256 at 64^3: 67 % of the network's FLOPs) and pyramid_transformation_1 (64 -> 256)
// (conerf/model/feature_pyramid_net.py:47-56,97-103; cuDNN conv3d in the reference).
//
// Why a second convolution kernel: the implicit-GEMM kernel of conv.hip re-gathers the A tile (256 voxels x 64 channels) from
// L2 for every one of the 27 taps — 64 KB of direct-to-LDS pieces per K step and workgroup, whose ISSUE alone costs what the
// step's MFMAs cost (0.41 of the MFMA roof, 5x the compulsory HBM bytes; profiles/r01_j).  Here a workgroup owns a
// 4 x 8 x 8 box of output voxels and stages its 6 x 10 x 10 input halo ONCE per 32-channel chunk (37.5 KB); all 27 taps read
// it through shifted ds_read_b128 windows.  Only the weights stream per tap (16 KB per unit = one tap x 32 channels x 256
// output channels, contiguous in the [chunk][tap][cout][32] pack).  Direct-to-LDS traffic per unit drops from 32 KB to ~17 KB.
//
// Schedule (8 waves = 2 (M: two z-planes each) x 4 (N: 64 output channels each), v_mfma_f32_32x32x16_bf16, 16 per wave and unit):
// the two wave groups (waves 0-3 / 4-7; wave w and w+4 share a SIMD) run in ANTI-PHASE, one barrier apart: while one group
// issues its 12 fragment reads and its direct-to-LDS pieces for a later unit, the other runs its 16 MFMAs at raised priority,
// then they swap (the guide's 8-phase idea with a unit as the phase).  Counted vmcnt only: the weight ring is RING units deep,
// a unit is issued AHEAD units before it is read, waited for one unit before it is read and never drained.
//
// LDS (147,456 B): two halo buffers of 40 KiB (chunk parity) | weight ring 4 x 16 KiB (re-used by the epilogue as the fp32
// staging tile).  Both images are written lane-linearly by the DMA, the 16-byte-slot XOR swizzles sit on the SOURCE address:
//   halo voxel hv = (hz*10 + hy)*10 + hx, 64 B each: slot s of voxel hv holds channel granule s ^ (hy & 3)
//   weight row co (64 B):                               slot s holds granule s ^ ((co >> 2) & 3)
// which makes every ds_read_b128 of a 32-row MFMA fragment conflict free for all 27 window shifts (tools/lds_conflicts.py).
#include "common.h"
#include <utility>

namespace halo {
constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HVOX = HZ * HY * HX;   // 600
constexpr int CK = 32;                                  // channels per unit
constexpr int HPW = 5;                                  // DMA pieces of 16 voxels x 64 B per wave: 40 >= ceil(600 / 16) = 38
constexpr int HBUF = 8 * HPW * 1024;                    // 40,960 B
constexpr int UNIT = 256 * CK * 2;                      // 16,384 B of weights per (chunk, tap)
constexpr uint32_t OOB = 0x7fffff00u;                   // out-of-range buffer offset: the DMA writes zeros
}

struct HaloGeom {
    int B, D, H, W, Cin, nchunks;
    int tilesY, tilesX, tilesPerGrid;        // tiles along y, x; tiles per grid
    int Da, Ha, Wa, add_shift;               // addend geometry (nearest x2 when add_shift = 1, same size when 0)
};

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ uint32_t halo_xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

#define HALO_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

// The 27 taps of a chunk are unrolled: every LDS offset, the halo-issue slot and every vmcnt count is an immediate and the load half
// is straight-line code (a first version looped over the units with scalar branches for the tap coordinates, the piece selection and
// the wait count: ~300 of its ~1,000 load-half cycles per unit were control flow).
// ABL (timing experiments only, wrong results): 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no MFMAs
template <typename TO, bool STAGGER, int ABL = 0, bool SPLITDS = true, bool PROF = false>
__global__ __launch_bounds__(512) void conv3_halo_kernel(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ wpk, TO* __restrict__ out,
    const float* __restrict__ bias, const TO* __restrict__ addend, HaloGeom g, uint32_t in_bytes, uint32_t wt_bytes,
    unsigned long long* __restrict__ prof = nullptr)
{
    using namespace halo;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int AHEAD = 2, RING = 4;       // unit u+AHEAD is issued in the load half of unit u into the slot unit u-2 was read from
    constexpr int RING_OFF = 2 * HBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t tile = halo_xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / g.tilesPerGrid;
    int rem = tile - b * g.tilesPerGrid;
    const int tz = rem / (g.tilesY * g.tilesX);
    rem -= tz * (g.tilesY * g.tilesX);
    const int ty = rem / g.tilesX, tx = rem - ty * g.tilesX;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, wt_bytes, 0x00020000);

    // ---- halo staging roles: this wave moves pieces wave, wave+8, ..., wave+32 (16 halo voxels each; lane -> voxel lane>>2, slot
    // lane&3); pieces 38, 39 and the voxels past 600 exist only as zero-filled padding of the buffer
    uint32_t hoff[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int hv = (wave + 8 * j) * 16 + (lane >> 2);
        const int hz = hv / (HY * HX), r2 = hv - hz * (HY * HX), hy = r2 / HX, hx = r2 - hy * HX;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool v = hv < HVOX && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        const uint32_t vox = (uint32_t)(((b * g.D + gz) * g.H + gy) * g.W + gx);
        hoff[j] = v ? vox * (uint32_t)(g.Cin * 2) + (uint32_t)((((lane & 3) ^ (hy & 3))) << 4) : OOB;
    }
    // ---- weight staging role: pieces 2*wave, 2*wave+1 of every unit (rows 32*wave .. +31); lane -> row lane>>2, slot lane&3.
    // The unit's byte offset goes into the VGPR offset (the buffer bounds check covers it: units past the pack read as zeros).
    const uint32_t wlane = (uint32_t)(wave * 2048 + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));

    // ---- fragment read roles (32x32x16: lane -> row lane&31, k-half lane>>5)
    const int fr = lane & 31, fq = lane >> 5;
    const int fx = fr & 7, fy = fr >> 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds0 + (uint32_t)((((2 * wm) * HY + fy) * HX + fx) * 64);
    uint32_t a_sw[3];                                     // per dy: lane base + swizzled slot of k-half 0 (k-half 1: ^ 32)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) a_sw[dy] = (uint32_t)((fq ^ ((fy + dy) & 3)) << 4);
    const uint32_t b_sw0 = (uint32_t)((fq ^ ((fr >> 2) & 3)) << 4);
    const uint32_t b_lane0 = lds0 + RING_OFF + (uint32_t)((wn * 64 + fr) * 64) + b_sw0;
    const uint32_t b_lane1 = lds0 + RING_OFF + (uint32_t)((wn * 64 + fr) * 64) + (b_sw0 ^ 32u);

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto issue_halo = [&](int chunk, bool real) {
        char* dst = smem + (chunk & 1) * HBUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 8192), 16, (int)(real ? hoff[j] : OOB), chunk * (CK * 2), 0, 0);
    };
    auto issue_unit = [&](int ring_w, int src_off) {       // ring_w: byte offset of the destination slot, src_off: byte offset of the unit
        char* dst = smem + RING_OFF + ring_w + wave * 2048;
        const int vo = (int)(wlane + (uint32_t)src_off);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        // the instruction offset of an LDS-DMA load is added to the memory address AND to the LDS address (M0 base + offset + 16 * lane)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lds_ptr_t)dst, 16, vo, 0, 1024, 0);
    };

    // ---- prologue: halo of chunk 0, units 0 and 1; unit 0 and the halo have landed for this wave once only unit 1 is outstanding
    issue_halo(0, true);
    issue_unit(0, 0);
    issue_unit(UNIT, UNIT);
    wait_vmcnt<2>();
    __builtin_amdgcn_s_barrier();
    if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind group 0

    unsigned long long pt[5] = {0, 0, 0, 0, 0}, tprev = 0;     // PROF: shader-clock sums per wave (tools/bench_conv_halo.py --prof)
    auto stamp = [&](int k) {
        if constexpr (PROF) {
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow) :: "memory");
            if (k >= 0) pt[k] += tnow - tprev;
            tprev = tnow;
        }
    };
    stamp(-1);

    int ring_r = 0;                       // slot (byte offset) unit u is read from; unit u+2 goes to ring_r + 2 slots
    int src_off = 2 * UNIT;               // byte offset of unit u+2 in the pack
#pragma unroll 1
    for (int c = 0; c < g.nchunks; ++c) {
        const uint32_t hbase = (uint32_t)((c & 1) * HBUF);
        const bool next_real = c + 1 < g.nchunks;
        static_for(std::make_integer_sequence<int, 27>{}, [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int dz = T / 9, dy = (T / 3) % 3, dx = T % 3;
            constexpr int OA = ((dz * HY + dy) * HX + dx) * 64;           // window shift of this tap inside the halo
            // ================= load half: fragments of unit u (k-half 0; SPLITDS: k-half 1 rides in the compute half), then the
            // direct-to-LDS pieces of unit u+2 — issued HERE, under the other group's MFMAs (between this wave's own MFMAs each piece
            // stalls its in-order instruction stream for ~100 cycles and the matrix pipe runs dry: measured 6 % slower)
            const uint32_t aA0 = a_lane + hbase + a_sw[dy], aA1 = a_lane + hbase + (a_sw[dy] ^ 32u);
            const uint32_t aB0 = b_lane0 + (uint32_t)ring_r, aB1 = b_lane1 + (uint32_t)ring_r;
            i32x4_t a00, a01, a10, a11, a20, a21, a30, a31, b00, b01, b10, b11;   // [tile][k-half]
            if (!(ABL & 2)) {
                HALO_DSR(b00, aB0, 0);         HALO_DSR(b10, aB0, 2048);
                HALO_DSR(a00, aA0, OA);        HALO_DSR(a10, aA0, OA + 2560);  HALO_DSR(a20, aA0, OA + 6400);  HALO_DSR(a30, aA0, OA + 8960);
                if (!SPLITDS) {
                    HALO_DSR(b01, aB1, 0);     HALO_DSR(b11, aB1, 2048);
                    HALO_DSR(a01, aA1, OA);    HALO_DSR(a11, aA1, OA + 2560);  HALO_DSR(a21, aA1, OA + 6400);  HALO_DSR(a31, aA1, OA + 8960);
                }
            } else {
                asm volatile("" : "=v"(a00), "=v"(a01), "=v"(a10), "=v"(a11), "=v"(a20), "=v"(a21), "=v"(a30), "=v"(a31), "=v"(b00), "=v"(b01), "=v"(b10), "=v"(b11)
                             : "v"(aA0), "v"(aA1), "v"(aB0), "v"(aB1));
            }
            // vmcnt retires loads IN ORDER, so a wait for a weight unit also waits for every older piece.  The halo pieces of the next
            // chunk are mostly HBM misses; issued AFTER this half's weight pieces (tap 1: the other halo buffer's last reader finished
            // a barrier ago) they are only older than the units issued from the next load half on, i.e. they have two units to land
            // (and in the last chunk they are turned into zero fills so that the counts below stay constants).
            if (!(ABL & 1)) {
                issue_unit((ring_r + 2 * UNIT) & (RING * UNIT - 1), src_off);
                if (T == 1) issue_halo(c + 1, next_real);
            }
            if (PROF) stamp(0);
            // unit u+1 (issued one load half ago) has landed for this wave when only the pieces issued after it are outstanding
            if (!(ABL & 1)) wait_vmcnt<2 + (T == 1 ? HPW : 0) + (T == 2 ? HPW : 0)>();
            if (PROF) stamp(1);
            ring_r = (ring_r + UNIT) & (RING * UNIT - 1);
            src_off += UNIT;
            __builtin_amdgcn_s_barrier();
            if (SPLITDS && !(ABL & 2))
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a00), "+v"(a10), "+v"(a20), "+v"(a30), "+v"(b00), "+v"(b10));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11), "+v"(a20), "+v"(a21), "+v"(a30), "+v"(a31),
                               "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11));
            __builtin_amdgcn_sched_barrier(0);
            if (PROF) stamp(2);
            // ================= compute half
            __builtin_amdgcn_s_setprio(1);
#define HALO_MM(i, j, A, Bv) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, Bv), acc[i][j], 0, 0, 0)
#define HALO_SB() __builtin_amdgcn_sched_barrier(0)
            if (ABL & 4) {
                asm volatile("" :: "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31), "v"(b00), "v"(b01), "v"(b10), "v"(b11));
            } else if (SPLITDS && !(ABL & 2)) {
                // the second k-half's six fragment reads ride in the issue gaps of the first eight MFMAs (an MFMA occupies the matrix
                // pipe for 32 cycles, a ds_read_b128 the wave's issue slot for a few)
                HALO_MM(0, 0, a00, b00); HALO_SB(); HALO_DSR(b01, aB1, 0);         HALO_SB();
                HALO_MM(1, 0, a10, b00); HALO_SB(); HALO_DSR(b11, aB1, 2048);      HALO_SB();
                HALO_MM(2, 0, a20, b00); HALO_SB(); HALO_DSR(a01, aA1, OA);        HALO_SB();
                HALO_MM(3, 0, a30, b00); HALO_SB(); HALO_DSR(a11, aA1, OA + 2560); HALO_SB();
                HALO_MM(0, 1, a00, b10); HALO_SB(); HALO_DSR(a21, aA1, OA + 6400); HALO_SB();
                HALO_MM(1, 1, a10, b10); HALO_SB(); HALO_DSR(a31, aA1, OA + 8960); HALO_SB();
                HALO_MM(2, 1, a20, b10); HALO_MM(3, 1, a30, b10);
                HALO_SB();
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01), "+v"(a11), "+v"(a21), "+v"(a31), "+v"(b01), "+v"(b11));
                HALO_SB();
                HALO_MM(0, 0, a01, b01); HALO_MM(1, 0, a11, b01); HALO_MM(2, 0, a21, b01); HALO_MM(3, 0, a31, b01);
                HALO_MM(0, 1, a01, b11); HALO_MM(1, 1, a11, b11); HALO_MM(2, 1, a21, b11); HALO_MM(3, 1, a31, b11);
            } else {
                HALO_MM(0, 0, a00, b00); HALO_MM(1, 0, a10, b00); HALO_MM(2, 0, a20, b00); HALO_MM(3, 0, a30, b00);
                HALO_MM(0, 1, a00, b10); HALO_MM(1, 1, a10, b10); HALO_MM(2, 1, a20, b10); HALO_MM(3, 1, a30, b10);
                HALO_MM(0, 0, a01, b01); HALO_MM(1, 0, a11, b01); HALO_MM(2, 0, a21, b01); HALO_MM(3, 0, a31, b01);
                HALO_MM(0, 1, a01, b11); HALO_MM(1, 1, a11, b11); HALO_MM(2, 1, a21, b11); HALO_MM(3, 1, a31, b11);
            }
#undef HALO_MM
#undef HALO_SB
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (PROF) stamp(3);
            __builtin_amdgcn_s_barrier();
            if (PROF) stamp(4);
        });
    }
    if constexpr (PROF) {
        if (prof && blockIdx.x < 64 && lane == 0)
            for (int k = 0; k < 5; ++k) prof[(blockIdx.x * 8 + wave) * 5 + k] = pt[k];
    }
    if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: one z-plane (64 voxels x 256 channels, fp32) at a time through the weight ring, then coalesced rows with
    // bias / addend applied in fp32
    float* sC = reinterpret_cast<float*>(smem + RING_OFF);
#pragma unroll
    for (int p = 0; p < 4; ++p) {            // unrolled: the accumulator tiles are indexed at compile time
        if (wm == (p >> 1)) {
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                const int i = 2 * (p & 1) + ih;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = ih * 32 + (e & 3) + 8 * (e >> 2) + 4 * fq;
                        sC[row * 256 + wn * 64 + j * 32 + fr] = acc[i][j][e];
                    }
            }
        }
        __syncthreads();
        const int z = z0 + p;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cidx = t + 512 * it;
            const int row = cidx >> 5, c8 = (cidx & 31) * 8;
            const int y = y0 + (row >> 3), x = x0 + (row & 7);
            const float4 lo = *reinterpret_cast<const float4*>(sC + row * 256 + c8), hi = *reinterpret_cast<const float4*>(sC + row * 256 + c8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias[c8 + e];
            }
            if (addend) {
                const TO* ap = addend + ((size_t)((b * g.Da + (z >> g.add_shift)) * g.Ha + (y >> g.add_shift)) * g.Wa + (x >> g.add_shift)) * 256 + c8;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += Elem<TO>::ld(ap + e);
            }
            TO* dst = out + ((size_t)((b * g.D + z) * g.H + y) * g.W + x) * 256 + c8;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same idea for 64 output channels: the three 64 -> 64 3^3 convolutions of layer1's bottlenecks at 32^3 (ResNet50 conv2 of
// conerf/model/resnet3d.py:76-113, Bottleneck; forward and data gradient).  In the implicit GEMM they gather 27 x 128 B per output row
// through L2 and run at 0.29 of the MFMA roof.  With 64 output channels a wave can own ALL of them, so the 4 waves of a workgroup
// split M only: a wave owns two z-planes (128 voxels) x 64 channels — the same 12 fragment reads per 16 MFMAs as the 256-channel
// kernel — and a workgroup an 8 x 8 x 8 box whose 10^3 halo (one 32-channel chunk: 62.5 KiB) sits in LDS.
// LDS 76 KiB (halo 64 KiB | weight ring 3 x 4 KiB): TWO workgroups per CU, so one workgroup's exposed halo load (there is no room for
// a second halo buffer) and epilogue run under the other's MFMAs; the two are not synchronised, which also takes the place of the
// anti-phase groups above.  One barrier per unit: at the top of unit u every wave has waited for its piece of unit u (vmcnt 1: only
// unit u+1 may be outstanding) — behind the barrier unit u is complete and unit u-1 has been read by everyone, so unit u+2 goes into
// unit u-1's slot.  27 % 3 == 0: ring slots are compile-time constants of the tap.
namespace halo64 {
constexpr int TZ = 8, TY = 8, TX = 8;
constexpr int HY = TY + 2, HX = TX + 2, HVOX = (TZ + 2) * HY * HX;           // 1000
constexpr int CK = 32;
constexpr int HPW = 16;                                 // DMA pieces of 16 voxels x 64 B per wave: 64 >= ceil(1000 / 16) = 63
constexpr int HBUF = 4 * HPW * 1024;                    // 65,536 B
constexpr int UNIT = 64 * CK * 2;                       // 4,096 B of weights per (chunk, tap): one 1 KiB piece per wave
constexpr int RING = 3;
constexpr int LDS = HBUF + RING * UNIT;                 // 77,824 B
}

// STATS: the BatchNorm behind this convolution gets its chunk sums from here (the layout of dreg_conv3d_igemm_bnstats: [B][V / 128][64][2] sums of the
// STORED values and of their squares; a chunk is one wave's 128 voxels — two z-planes of the box —, chunk index tile * 4 + wave).  The order of
// every addition is fixed by the box geometry alone, never by how many grids share the launch.
// (TO: the output type — bf16 only; it keeps the instantiation's name in step with conv3_halo_kernel<TO, ...> for the profile labels)
template <typename TO, bool PF, bool STATS>
__global__ __launch_bounds__(256, 2) void conv3_halo64_kernel(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ wpk, TO* __restrict__ out,
    const float* __restrict__ bias, const TO* __restrict__ addend, HaloGeom g, uint32_t in_bytes, uint32_t wt_bytes,
    float* __restrict__ bn_part)
{
    using namespace halo64;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t tile = halo_xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / g.tilesPerGrid;
    int rem = tile - b * g.tilesPerGrid;
    const int tz = rem / (g.tilesY * g.tilesX);
    rem -= tz * (g.tilesY * g.tilesX);
    const int ty = rem / g.tilesX, tx = rem - ty * g.tilesX;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, wt_bytes, 0x00020000);

    // halo staging: this wave moves pieces wave, wave+4, ..., wave+60 (lane -> voxel lane>>2 of the piece, slot lane&3); same image as above
    uint32_t hoff[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int hv = (wave + 4 * j) * 16 + (lane >> 2);
        const int hz = hv / (HY * HX), r2 = hv - hz * (HY * HX), hy = r2 / HX, hx = r2 - hy * HX;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool v = hv < HVOX && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        const uint32_t vox = (uint32_t)(((b * g.D + gz) * g.H + gy) * g.W + gx);
        hoff[j] = v ? vox * (uint32_t)(g.Cin * 2) + (uint32_t)((((lane & 3) ^ (hy & 3))) << 4) : halo::OOB;
    }
    // weight staging: piece `wave` of every unit (rows 16*wave .. +15; lane -> row lane>>2, slot lane&3)
    const uint32_t wlane = (uint32_t)(wave * 1024 + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));

    const int fr = lane & 31, fq = lane >> 5;
    const int fx = fr & 7, fy = fr >> 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds0 + (uint32_t)((((2 * wave) * HY + fy) * HX + fx) * 64);
    uint32_t aA0[3], aA1[3];                               // per dy: k-half 0 / 1 address of this lane's row
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const uint32_t sw = (uint32_t)((fq ^ ((fy + dy) & 3)) << 4);
        aA0[dy] = a_lane + sw;
        aA1[dy] = a_lane + (sw ^ 32u);
    }
    const uint32_t b_sw0 = (uint32_t)((fq ^ ((fr >> 2) & 3)) << 4);
    const uint32_t aB0 = lds0 + HBUF + (uint32_t)(fr * 64) + b_sw0;
    const uint32_t aB1 = lds0 + HBUF + (uint32_t)(fr * 64) + (b_sw0 ^ 32u);

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto issue_halo = [&](int chunk) {
        char* dst = smem + wave * 1024;
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 4096), 16, (int)hoff[j], chunk * (CK * 2), 0, 0);
    };
    auto issue_unit = [&](int ring_w, int src_off) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lds_ptr_t)(smem + HBUF + ring_w + wave * 1024), 16, (int)(wlane + (uint32_t)src_off), 0, 0, 0);
    };

    issue_halo(0);
    issue_unit(0, 0);
    issue_unit(UNIT, UNIT);
    int src_off = 2 * UNIT;               // byte offset of unit u+2 in the pack (units past the pack read as zeros: buffer bounds)
    // PF: the halo does not change inside a chunk, so tap T+1's four A fragments (k-half 0) are read under the last MFMAs of tap T; behind the
    // barrier of tap T+1 only its two weight fragments are on the wave's critical path
    i32x4_t af[4];
#pragma unroll 1
    for (int c = 0; c < g.nchunks; ++c) {
        static_for(std::make_integer_sequence<int, 27>{}, [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int dz = T / 9, dy = (T / 3) % 3, dx = T % 3;
            constexpr int OA = ((dz * HY + dy) * HX + dx) * 64;
            constexpr int Tn = T < 26 ? T + 1 : 0, dyn = (Tn / 3) % 3;
            constexpr int OAn = (((Tn / 9) * HY + dyn) * HX + Tn % 3) * 64;
            constexpr int OB = (T % RING) * UNIT, OW = ((T + 2) % RING) * UNIT;
            if (T == 0) wait_vmcnt<0>(); else wait_vmcnt<1>();     // (tap 0: the chunk's halo pieces are the youngest loads)
            __builtin_amdgcn_s_barrier();
            i32x4_t a01, a11, a21, a31, b00, b01, b10, b11;        // [tile][k-half]
            HALO_DSR(b00, aB0, OB);            HALO_DSR(b10, aB0, OB + 2048);
            if (!PF || T == 0) {
                HALO_DSR(af[0], aA0[dy], OA);  HALO_DSR(af[1], aA0[dy], OA + 2560);  HALO_DSR(af[2], aA0[dy], OA + 6400);  HALO_DSR(af[3], aA0[dy], OA + 8960);
            }
            issue_unit(OW, src_off);
            src_off += UNIT;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(b00), "+v"(b10));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#define HALO_MM(i, j, A, Bv) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, Bv), acc[i][j], 0, 0, 0)
#define HALO_SB() __builtin_amdgcn_sched_barrier(0)
            // the second k-half's six fragment reads ride in the issue gaps of the first eight MFMAs
            HALO_MM(0, 0, af[0], b00); HALO_SB(); HALO_DSR(b01, aB1, OB);            HALO_SB();
            HALO_MM(1, 0, af[1], b00); HALO_SB(); HALO_DSR(b11, aB1, OB + 2048);     HALO_SB();
            HALO_MM(2, 0, af[2], b00); HALO_SB(); HALO_DSR(a01, aA1[dy], OA);        HALO_SB();
            HALO_MM(3, 0, af[3], b00); HALO_SB(); HALO_DSR(a11, aA1[dy], OA + 2560); HALO_SB();
            HALO_MM(0, 1, af[0], b10); HALO_SB(); HALO_DSR(a21, aA1[dy], OA + 6400); HALO_SB();
            HALO_MM(1, 1, af[1], b10); HALO_SB(); HALO_DSR(a31, aA1[dy], OA + 8960); HALO_SB();
            HALO_MM(2, 1, af[2], b10); HALO_MM(3, 1, af[3], b10);
            HALO_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01), "+v"(a11), "+v"(a21), "+v"(a31), "+v"(b01), "+v"(b11));
            HALO_SB();
            if (PF && T < 26) {
                HALO_MM(0, 0, a01, b01); HALO_MM(1, 0, a11, b01); HALO_SB(); HALO_DSR(af[0], aA0[dyn], OAn);        HALO_SB();
                HALO_MM(2, 0, a21, b01); HALO_SB();                          HALO_DSR(af[1], aA0[dyn], OAn + 2560); HALO_SB();
                HALO_MM(3, 0, a31, b01); HALO_SB();                          HALO_DSR(af[2], aA0[dyn], OAn + 6400); HALO_SB();
                HALO_MM(0, 1, a01, b11); HALO_SB();                          HALO_DSR(af[3], aA0[dyn], OAn + 8960); HALO_SB();
                HALO_MM(1, 1, a11, b11); HALO_MM(2, 1, a21, b11); HALO_MM(3, 1, a31, b11);
            } else {
                HALO_MM(0, 0, a01, b01); HALO_MM(1, 0, a11, b01); HALO_MM(2, 0, a21, b01); HALO_MM(3, 0, a31, b01);
                HALO_MM(0, 1, a01, b11); HALO_MM(1, 1, a11, b11); HALO_MM(2, 1, a21, b11); HALO_MM(3, 1, a31, b11);
            }
#undef HALO_MM
#undef HALO_SB
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (c + 1 < g.nchunks) {
            __builtin_amdgcn_s_barrier();          // every wave is done with this chunk's halo
            issue_halo(c + 1);
        }
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: two passes of 256 voxels x 64 channels (fp32) through the halo buffer, then coalesced 16-byte stores with bias /
    // addend applied in fp32.  Row r of a pass = (wave & 1) * 128 + tile * 32 + row of the MFMA tile; the two 32-channel halves of rows
    // with bit 2 set are swapped (the MFMA's two k-groups of lanes hold rows 4 apart: without the swap they hit the same banks).
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if ((wave >> 1) == p) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wave & 1) * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fq;
                        sC[row * 64 + ((j * 32 + fr) ^ (fq << 5))] = acc[i][j][e];
                    }
        }
        __syncthreads();
        float bs1[2][8], bs2[2][8];                  // STATS: this thread's 8 channels, rows t/8 + 32 * it: it < 4 -> the pass's first chunk, else its second
        if constexpr (STATS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) { bs1[h][e] = 0.f; bs2[h][e] = 0.f; }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int cidx = t + 256 * it;
            const int row = cidx >> 3, c8 = (cidx & 7) * 8;
            const int i = (row >> 5) & 3, r = row & 31;
            const int z = z0 + 2 * (2 * p + (row >> 7)) + (i >> 1), y = y0 + 4 * (i & 1) + (r >> 3), x = x0 + (r & 7);
            const float* sp = sC + row * 64 + (c8 ^ (((row >> 2) & 1) << 5));
            const float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias[c8 + e];
            }
            if (addend) {
                const bf16_t* ap = addend + ((size_t)((b * g.Da + (z >> g.add_shift)) * g.Ha + (y >> g.add_shift)) * g.Wa + (x >> g.add_shift)) * 64 + c8;
                const uint4 q = *reinterpret_cast<const uint4*>(ap);
                const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w4[e] << 16); v[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u); }
            }
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<uint4*>(out + ((size_t)((b * g.D + z) * g.H + y) * g.W + x) * 64 + c8) = make_uint4(w[0], w[1], w[2], w[3]);
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = __uint_as_float(w[e] << 16), c = __uint_as_float(w[e] & 0xffff0000u);
                    bs1[it >> 2][2 * e] += a;      bs2[it >> 2][2 * e] += a * a;
                    bs1[it >> 2][2 * e + 1] += c;  bs2[it >> 2][2 * e + 1] += c * c;
                }
            }
        }
        if constexpr (STATS) {
            // the 32 threads with the same t & 7 hold a chunk's 128 rows: 8 lanes of each wave (xor 8, 16, 32), then the four waves in order
            float* red = reinterpret_cast<float*>(smem + HBUF);              // [wave][chunk of the pass][64 channels][2] (the weight ring is idle)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = bs1[h][e], c = bs2[h][e];
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
                    if (lane < 8) { red[((wave * 2 + h) * 64 + lane * 8 + e) * 2] = a; red[((wave * 2 + h) * 64 + lane * 8 + e) * 2 + 1] = c; }
                }
            __syncthreads();
            {
                const int h = t >> 7, cw = t & 127;                          // (channel, which) of chunk h
                const float sum = ((red[(0 * 2 + h) * 128 + cw] + red[(1 * 2 + h) * 128 + cw]) + red[(2 * 2 + h) * 128 + cw]) + red[(3 * 2 + h) * 128 + cw];
                bn_part[((size_t)tile * 4 + 2 * p + h) * 128 + cw] = sum;
            }
        }
        __syncthreads();
    }
}

// torch weight [Cout][Cin][3][3][3] fp32 -> [Cin/32][27][256 rows][32] bf16.
//   forward      : row = co,  column = ci within the chunk, tap as stored:            pack[c][t][co][k] = W[co][32c+k][t]
//   data gradient: row = ci,  column = co within the chunk, tap flipped (26 - t):     pack[c][t][ci][k] = W[32c+k][ci][26-t]
// (dIn[v][ci] = sum_{t,co} dOut[v + 1 - d(t)][co] W[co][ci][t] = sum_{t'} sum_co dOut[v - 1 + d(t')][co] W[co][ci][26-t'])
// experiments only (include/dreg_nerf_probe.h): 0 = anti-phase groups (default); 1 = lockstep; 3 = all 12 fragment reads in the load half;
// 5 = profiled; 1x = ablations; -1 = dreg_conv3_halo_use() answers 0 (the implicit-GEMM kernel serves every shape: A/B tests)
DREG_KNOB(int, g_halo_variant, 0);
// 64-output-channel kernel: 1 = on (default), 0 = dreg_conv3_halo_use() answers 0 for Cout = 64 (A/B tests), 2 = without the prefetch of the next tap's A fragments (53.3 -> 54.6 us on 32^3 x 8, 64 -> 64)
DREG_KNOB(int, g_halo64, 1);

__global__ __launch_bounds__(256) void pack_weight_halo_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int transposed)
{
    const int rows = transposed ? Cin : Cout, red = transposed ? Cout : Cin;   // rows must be 256
    const int unit = blockIdx.x;                 // (chunk, tap)
    const int c = unit / 27, tp = unit - c * 27;
    for (int i = threadIdx.x; i < rows * 4; i += 256) {
        const int row = i >> 2, k8 = (i & 3) * 8;
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = c * 32 + k8 + 2 * e + h;
                v[h] = k < red ? (transposed ? w[((size_t)k * Cin + row) * 27 + (26 - tp)] : w[((size_t)row * Cin + k) * 27 + tp]) : 0.f;
            }
            pk[e] = f2bf2(v[0], v[1]);
        }
        *reinterpret_cast<uint4*>(out + ((size_t)unit * rows + row) * 32 + k8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}


DREG_KNOB(unsigned long long*, g_halo_prof, nullptr);

extern "C" {

#ifdef DREG_PROBE
void dreg_conv3_halo_set_variant(int v) { g_halo_variant = v; }
void dreg_conv3_halo64_set(int v) { g_halo64 = v; }
void dreg_conv3_halo_set_prof(void* buf) { g_halo_prof = (unsigned long long*)buf; }   // 64 blocks x 8 waves x 5 u64 (variant 5)
#endif

// 1 when (shape) is served by a halo kernel: 3^3 / stride 1 / pad 1, Cin % 32 == 0, operands below 2 GiB (32-bit buffer offsets) and
// 256 output channels with the volume divisible by the 4 x 8 x 8 box, or 64 output channels with the volume divisible by 8 x 8 x 8
int dreg_conv3_halo_supported(int B, int D, int H, int W, int Cin, int Cout)
{
    if (Cin % 32 != 0 || Cin < 32) return 0;
    if (Cout == 256) { if (D % halo::TZ || H % halo::TY || W % halo::TX) return 0; }
    else if (Cout == 64) { if (D % halo64::TZ || H % halo64::TY || W % halo64::TX) return 0; }
    else return 0;
    if ((uint64_t)B * D * H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffff00ull) return 0;
    return 1;
}

// 1 when a bf16 convolution (ksz, stride, pad; Cin -> Cout over [B,D,H,W]) should run on a halo kernel: supported shape and enough
// boxes to fill the chip (a 4 x 8 x 8 box of the 256-channel kernel keeps a CU busy for ~100 us; small volumes are better served by the
// split-K implicit GEMM).  The decision depends on the per-grid shape only, never on B: a pair's result does not depend on its batch mates.
int dreg_conv3_halo_use(int B, int D, int H, int W, int Cin, int Cout, int ksz, int stride, int pad)
{
    if (g_halo_variant < 0 || ksz != 3 || stride != 1 || pad != 1 || !dreg_conv3_halo_supported(B, D, H, W, Cin, Cout)) return 0;
    if (Cout == 64) return g_halo64 && (D / halo64::TZ) * (H / halo64::TY) * (W / halo64::TX) >= 64;      // >= 32^3 per grid
    return (D / halo::TZ) * (H / halo::TY) * (W / halo::TX) >= 128;      // >= 32^3 per grid
}

size_t dreg_conv3_halo_pack_bytes(int Cin_red) { return (size_t)(Cin_red / 32) * 27 * halo::UNIT; }
size_t dreg_conv3_halo_pack_bytes_n(int Cin_red, int rows) { return (size_t)(Cin_red / 32) * 27 * rows * halo::CK * 2; }

// w: torch layout fp32 [Cout][Cin][27]; transposed = 0: forward pack (Cout must be 256 or 64), 1: data-gradient pack (Cin must be 256 or 64)
int dreg_pack_conv_weight_halo(const float* w, void* out, int Cout, int Cin, int transposed, void* stream)
{
    const int rows = transposed ? Cin : Cout, red = transposed ? Cout : Cin;
    if ((rows != 256 && rows != 64) || red % 32 != 0) return DREG_EINVAL;
    hipLaunchKernelGGL(pack_weight_halo_kernel, dim3((red / 32) * 27), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)out, Cout, Cin, transposed);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// out[b,z,y,x,:] = bias + addend(...) + sum_{d in 3^3} in[b, (z,y,x) - 1 + d, :] . pack[., d, :, .]     (bf16 in/out, fp32 accumulate)
// in [B,D,H,W,Cin] bf16, wpk from dreg_pack_conv_weight_halo, out [B,D,H,W,256] bf16 (out_f32: fp32), bias fp32 [256] or null,
// addend [B,Da,Ha,Wa,256] (same dtype as out) added with nearest x2 upsampling (add_same = 0) or element-wise (add_same = 1), or null.
int dreg_conv3_halo(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                    int B, int D, int H, int W, int Cin, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream)
{
    using namespace halo;
    if (!dreg_conv3_halo_supported(B, D, H, W, Cin, 256)) return DREG_EINVAL;
    HaloGeom g;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.nchunks = Cin / CK;
    g.tilesY = H / TY; g.tilesX = W / TX; g.tilesPerGrid = (D / TZ) * g.tilesY * g.tilesX;
    g.Da = Da; g.Ha = Ha; g.Wa = Wa; g.add_shift = add_same ? 0 : 1;
    const uint32_t ntiles = (uint32_t)B * g.tilesPerGrid;
    if (ntiles == 0) return DREG_OK;
    const uint32_t in_bytes = (uint32_t)((uint64_t)B * D * H * W * Cin * 2), wt_bytes = (uint32_t)dreg_conv3_halo_pack_bytes(Cin);
    hipStream_t st = (hipStream_t)stream;
#define HALO_LAUNCH(TOt, SG, AB, SD, PR) do { \
        const int lds_ = 2 * HBUF + 4 * UNIT; \
        (void)hipFuncSetAttribute((const void*)conv3_halo_kernel<TOt, SG, AB, SD, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); \
        hipLaunchKernelGGL((conv3_halo_kernel<TOt, SG, AB, SD, PR>), dim3(ntiles), dim3(512), lds_, st, (const bf16_t*)in, (const bf16_t*)wpk, (TOt*)out, \
                           bias, (const TOt*)addend, g, in_bytes, wt_bytes, g_halo_prof); } while (0)
    if (out_f32) HALO_LAUNCH(float, true, 0, true, false);
    else if (g_halo_variant == 1) HALO_LAUNCH(bf16_t, false, 0, true, false);
    else if (g_halo_variant == 3) HALO_LAUNCH(bf16_t, true, 0, false, false);
    else if (g_halo_variant == 5) HALO_LAUNCH(bf16_t, true, 0, false, true);
    else if (g_halo_variant == 11) HALO_LAUNCH(bf16_t, true, 1, true, false);
    else if (g_halo_variant == 12) HALO_LAUNCH(bf16_t, true, 2, true, false);
    else if (g_halo_variant == 13) HALO_LAUNCH(bf16_t, true, 3, true, false);
    else if (g_halo_variant == 14) HALO_LAUNCH(bf16_t, true, 4, true, false);
    else if (g_halo_variant == 16) HALO_LAUNCH(bf16_t, true, 6, true, false);
    else HALO_LAUNCH(bf16_t, true, 0, true, false);
#undef HALO_LAUNCH
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// The same for Cout output channels (256: dreg_conv3_halo; 64: the 8 x 8 x 8-box kernel, bf16 output only).
// dreg_conv3_halo_n_bnstats: the forward that also leaves the BatchNorm statistics of its output behind, like dreg_conv3d_igemm_bnstats: bn_partial
// [B][V / *rows_per_chunk][Cout][2]; *rows_per_chunk = 0 when the kernel has no such epilogue (Cout = 256: the BatchNorm runs its own pass).
static int halo_n_impl(const void* in, const void* wpk, void* out, const float* bias, const void* addend, int B, int D, int H, int W, int Cin, int Cout,
                       int Da, int Ha, int Wa, int add_same, int out_f32, float* bn_partial, int* rows_per_chunk, void* stream)
{
    if (rows_per_chunk) *rows_per_chunk = 0;
    if (Cout == 256) return dreg_conv3_halo(in, wpk, out, bias, addend, B, D, H, W, Cin, Da, Ha, Wa, add_same, out_f32, stream);
    using namespace halo64;
    if (Cout != 64 || out_f32 || !dreg_conv3_halo_supported(B, D, H, W, Cin, 64)) return DREG_EINVAL;
    HaloGeom g;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.nchunks = Cin / CK;
    g.tilesY = H / TY; g.tilesX = W / TX; g.tilesPerGrid = (D / TZ) * g.tilesY * g.tilesX;
    g.Da = Da; g.Ha = Ha; g.Wa = Wa; g.add_shift = add_same ? 0 : 1;
    const uint32_t ntiles = (uint32_t)B * g.tilesPerGrid;
    if (ntiles == 0) return DREG_OK;
    const uint32_t in_bytes = (uint32_t)((uint64_t)B * D * H * W * Cin * 2), wt_bytes = (uint32_t)dreg_conv3_halo_pack_bytes_n(Cin, 64);
    hipStream_t st = (hipStream_t)stream;
#define HALO64_LAUNCH(SD, STv) do { \
        (void)hipFuncSetAttribute((const void*)conv3_halo64_kernel<bf16_t, SD, STv>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
        hipLaunchKernelGGL((conv3_halo64_kernel<bf16_t, SD, STv>), dim3(ntiles), dim3(256), LDS, st, (const bf16_t*)in, (const bf16_t*)wpk, (bf16_t*)out, \
                           bias, (const bf16_t*)addend, g, in_bytes, wt_bytes, bn_partial); } while (0)
    if (g_halo64 == 2) { if (bn_partial) HALO64_LAUNCH(false, true); else HALO64_LAUNCH(false, false); }
    else { if (bn_partial) HALO64_LAUNCH(true, true); else HALO64_LAUNCH(true, false); }
#undef HALO64_LAUNCH
    DREG_LAUNCH_CHECK();
    if (rows_per_chunk && bn_partial) *rows_per_chunk = 128;
    return DREG_OK;
}
int dreg_conv3_halo_n(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                      int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream)
{
    return halo_n_impl(in, wpk, out, bias, addend, B, D, H, W, Cin, Cout, Da, Ha, Wa, add_same, out_f32, nullptr, nullptr, stream);
}
int dreg_conv3_halo_n_bnstats(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                              int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, float* bn_partial, int* rows_per_chunk, void* stream)
{
    if (!rows_per_chunk) return DREG_EINVAL;
    return halo_n_impl(in, wpk, out, bias, addend, B, D, H, W, Cin, Cout, Da, Ha, Wa, add_same, 0, bn_partial, rows_per_chunk, stream);
}

}  // extern "C"

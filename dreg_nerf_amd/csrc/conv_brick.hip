// Active-set ("row list") 3x3x3 / stride-1 / pad-1 convolution with staged-neighbourhood reuse: the two FPN head layers and their data
// gradients on the voxels around the occupied surface (conerf/model/feature_pyramid_net.py:47-56,97-103 upsample_transform_{1,2},
// pyramid_transformation_1; cuDNN conv3d over the whole volume in the reference).
//
// Why a third convolution kernel.  The row-list form of the implicit-GEMM kernel (conv.hip) gathers, for every output row and every one
// of the 27 taps, the 512 bytes of that tap's input voxel from L2 into LDS: 27 x 512 B per row, 3.1x the compulsory HBM bytes and
// 0.29-0.5 of the MFMA roof (0.19 for the 64-output-channel data gradient: 64 FLOP per gathered byte).  Neighbouring output voxels
// share 18 of their 27 input voxels.  Here a workgroup owns a TILE of up to 256 output rows that lie close together (rows are taken in
// brick-major order: 8 x 8 x 8 bricks), stages the UNION of their 3^3 neighbourhoods once per 16-channel chunk (<= 1,264 voxels x 32 B,
// ~3-4 staged voxels per row instead of 27) and runs all 27 taps from LDS through a per-(row, tap) table of LDS offsets.  Only the
// weights stream per tap, as in conv_halo.hip (whose schedule this kernel keeps: eight waves in two anti-phase groups, one barrier
// apart; a phase = 16 KB of weights = 16 MFMAs per wave; counted vmcnt, weight ring four phases deep).
//
// Tiles are built per step from the active-set flag volumes by dreg_brick_tiles_build (geometry stream): brick-major compaction of the
// rows, then one workgroup per candidate tile forms the neighbourhood union in an LDS bitmap of the (padded) grid, ranks the set bits
// (slot = rank: raster order, so rows that are neighbours in x read neighbouring slots) and writes the halo list and the offset table.  A
// candidate whose union exceeds the LDS capacity is halved (down to 32 rows: 32 x 27 < 1,023 always fits).
//
// LDS (147,456 B; + 14 KiB for the offset table of the 256-channel form): two halo buffers of 40 KiB (chunk parity; slot s = 32 B = 16 channels, 16-byte granule q of slot s at q ^ ((s >> 3) & 1))
// | weight ring 4 x 16 KiB (row r of a unit = 32 B, granule q at q ^ ((r >> 3) & 1)); both images are written lane-linearly by
// buffer_load ... lds with the XOR on the SOURCE address.  With 32-byte records every 16-lane group of a ds_read_b128 covers 16
// distinct 16-byte bank groups when its 16 rows read 16 consecutive slots.
#include "common.h"
#include <utility>

namespace brick {
constexpr int TROWS = 256;                  // output rows of a tile
constexpr int HCAP = 1280;                  // halo slots per tile = 16 colour classes x 80; the last stripe (slots 1264 + c) is always zero
constexpr int NCLS = 16, CLS_CAP = HCAP / NCLS - 1;      // staged voxels per colour class (79)
constexpr int ZSLOT = HCAP - NCLS;          // first zero slot (colour 0); padding taps / padded rows read the zero slot of their colour
// Slot numbering.  false (default): slot = raster rank of the staged voxel (dense: every tile holds up to 1,263 voxels).  true: slot
// congruent to the voxel's colour (x mod 8, y mod 2) modulo 16 — fewer LDS bank-group collisions of the fragment reads, but a tile ends
// when ONE colour class is full: +20 % tiles on the shell-R sets and no faster (tools/bench_conv_brick.py; the reads are not what bounds it)
constexpr bool COLOUR_SLOTS = false;
constexpr int CK = 16;                      // channels per chunk
constexpr int TAPS = 28;                    // 27 + one zero tap: a chunk is a whole number of phases
constexpr int HBUF = HCAP * CK * 2;         // 40,960 B
constexpr int PHASE_BYTES = 16384;          // weight ring slot
constexpr int RING = 4;
constexpr int HPW = 5;                      // halo DMA pieces (32 slots each) per wave and chunk
constexpr uint32_t OOB = 0x7fffff00u;
constexpr int BS = 8;                       // brick edge (rows are ordered brick by brick)
// LDS byte offset of granule 0 of a slot (granule q of slot s sits at q ^ ((s >> 3) & 1))
__host__ __device__ constexpr inline uint16_t slot_off(int s) { return (uint16_t)(s * 32 + (((s >> 3) & 1) << 4)); }
// colour of a voxel: (x mod 8, y mod 2).  Slot numbers are congruent to the colour modulo 16, so two lanes of a ds_read_b128's 16-lane
// group collide only when their rows' neighbours share (x mod 8, y mod 2) — rows in brick raster order rarely do (the dense kernel's
// 2 x 8 fragment rows are exactly one voxel of every colour); a shift by a tap offset permutes the colours, so this holds for all 27 taps
__device__ __forceinline__ int colour(int y, int x) { return (x & 7) | ((y & 1) << 3); }
}

struct BrickGeom {
    int B, D, H, W, Cin, nchunks;           // input tensor [B,D,H,W,Cin] (= output grid), Cin = reduction channels
    int Da, Ha, Wa, add_shift;              // addend geometry (nearest x2 when add_shift = 1), 0 = no addend
};
struct BrickTile { int row0, nrows, nhalo, pad; };

typedef __attribute__((ext_vector_type(4))) int bi32x4_t;

template <int N> __device__ __forceinline__ void bk_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int... I, typename F> __device__ __forceinline__ void bk_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
#define BK_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
// s_waitcnt lgkmcnt(CNT) tied to the fragment registers it releases (the compiler knows nothing of the asm-issued LDS reads: without the
// "+v" operands it may move an MFMA that consumes a fragment above the wait)
template <int CNT, int FM, int FN> __device__ __forceinline__ void bk_wait_frags(bi32x4_t (&a)[FM], bi32x4_t (&b)[FN])
{
    if constexpr (FM == 4 && FN == 2)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(CNT));
    else {
        static_assert(FM == 2 && FN == 1, "fragment shapes of the two instantiations");
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(CNT));
    }
}

// NOUT = 256: waves 2 (M) x 4 (N), 4 x 2 fragments of 32 x 32, two taps per phase.  NOUT = 64: waves 4 (M) x 2 (N), 2 x 1 fragments, seven
// taps per phase.  Either way a phase is 14-16 v_mfma_f32_32x32x16_bf16 per wave behind <= 16 KB of weights.
template <int NOUT, typename TO>
__global__ __launch_bounds__(512) void conv3_brick_kernel(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ wpk, TO* __restrict__ out, const float* __restrict__ bias, const TO* __restrict__ addend,
    BrickGeom g, const BrickTile* __restrict__ tiles, const int* __restrict__ halo_vox, const uint16_t* __restrict__ nbr,
    const int* __restrict__ rows_sorted, uint32_t in_bytes, uint32_t wt_bytes)
{
    using namespace brick;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int MW = NOUT == 256 ? 2 : 4, NW = 8 / MW;
    constexpr int FM = TROWS / (32 * MW), FN = NOUT / (32 * NW);
    constexpr int UPP = NOUT == 256 ? 2 : 7;                 // taps per phase
    constexpr int PPC = TAPS / UPP;                          // phases per chunk
    constexpr int UNIT = NOUT * CK * 2;                      // bytes of one tap's weights
    constexpr int PH_SRC = UPP * UNIT;                       // bytes of a phase in the pack (<= PHASE_BYTES)
    constexpr int WPIECES = (PH_SRC + 8191) / 8192;          // DMA instructions per wave and phase
    constexpr int RING_OFF = 2 * HBUF;
    static_assert(PH_SRC <= PHASE_BYTES && TAPS % UPP == 0, "phase geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / NW, wn = wave % NW;
    const int grp = wave >> 2;                               // anti-phase group (waves w and w + 4 share a SIMD)
    const BrickTile tile = tiles[blockIdx.x];
    if (tile.nrows <= 0) return;
    const int fr = lane & 31, fq = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, wt_bytes, 0x00020000);

    // ---- halo staging role: pieces wave, wave + 8, ... (32 slots each; lane -> slot lane >> 1, granule lane & 1)
    uint32_t hoff[HPW];
    const int* hv = halo_vox + (size_t)blockIdx.x * HCAP;
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int slot = (wave + 8 * j) * 32 + (lane >> 1);
        const int vox = slot < tile.nhalo ? hv[slot] : -1;        // slots past the tile's last used stripe (incl. the zero stripe) are zero-filled
        hoff[j] = vox >= 0 ? (uint32_t)vox * (uint32_t)(g.Cin * 2) + (uint32_t)((((lane & 1) ^ ((slot >> 3) & 1))) << 4) : OOB;
    }
    // ---- weight staging role: lane -> row (o / 32) % NOUT of the phase image at byte o = piece * 8192 + wave * 1024 + lane * 16
    uint32_t wlane[WPIECES];
#pragma unroll
    for (int pc = 0; pc < WPIECES; ++pc) {
        const uint32_t o = (uint32_t)(pc * 8192 + wave * 1024 + lane * 16);
        const uint32_t row = (o >> 5) % NOUT;
        wlane[pc] = o < (uint32_t)PH_SRC ? (o & ~31u) + ((((o >> 4) & 1) ^ ((row >> 3) & 1)) << 4) : OOB;
    }
    // ---- the (row, tap) -> LDS offset table (28 uint16 per row).  NOUT = 64: this lane's FM rows in registers (28 dwords).  NOUT = 256:
    // four rows per lane would be 56 registers next to 128 accumulators — the tile's table is copied to LDS instead and a phase's two taps
    // (one dword per row) are read one phase ahead
    constexpr bool NB_LDS = NOUT == 256;
    constexpr int NBT_OFF = RING_OFF + RING * PHASE_BYTES;
    uint32_t nb[NB_LDS ? 1 : FM][NB_LDS ? 1 : TAPS / 2];
    const uint16_t* nbt = nbr + (size_t)blockIdx.x * (TROWS * TAPS);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t nbc[FM], nbn[FM], nb_row[FM];
    if constexpr (NB_LDS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(nbt);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem + NBT_OFF);
#pragma unroll
        for (int k = 0; k < (TROWS * TAPS / 2) / 512; ++k) dst[t + 512 * k] = src[t + 512 * k];
#pragma unroll
        for (int i = 0; i < FM; ++i) { nb_row[i] = lds0 + NBT_OFF + (uint32_t)((wm * (32 * FM) + i * 32 + fr) * (TAPS * 2)); nbc[i] = 0; nbn[i] = 0; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FM; ++i) nbc[i] = *reinterpret_cast<const uint32_t*>(smem + NBT_OFF + (wm * (32 * FM) + i * 32 + fr) * (TAPS * 2));
    } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(nbt + (size_t)(wm * (32 * FM) + i * 32 + fr) * TAPS);
#pragma unroll
            for (int k = 0; k < TAPS / 2; ++k) nb[i][k] = src[k];
            nbc[i] = nbn[i] = nb_row[i] = 0;
        }
    }
    const uint32_t fq16 = (uint32_t)(fq << 4);
    uint32_t b_lane[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = wn * (32 * FN) + j * 32 + fr;
        b_lane[j] = lds0 + RING_OFF + (uint32_t)(row * 32) + (uint32_t)(((fq ^ ((row >> 3) & 1))) << 4);
    }

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto issue_halo = [&](int chunk, bool real) {
        char* dst = smem + (chunk & 1) * HBUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 8192), 16, (int)(real ? hoff[j] : OOB), chunk * (CK * 2), 0, 0);
    };
    auto issue_phase = [&](int ring_w, int src_off) {       // ring_w: byte offset of the ring slot, src_off: byte offset of the phase in the pack
        char* dst = smem + RING_OFF + ring_w + wave * 1024;
#pragma unroll
        for (int pc = 0; pc < WPIECES; ++pc)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lds_ptr_t)(dst + pc * 8192), 16, (int)(wlane[pc] == OOB ? OOB : wlane[pc] + (uint32_t)src_off), 0, 0, 0);
    };

    // ---- prologue: halo of chunk 0, phases 0 and 1
    issue_halo(0, true);
    issue_phase(0, 0);
    issue_phase(PHASE_BYTES, PH_SRC);
    bk_wait_vmcnt<WPIECES>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind group 0

    int ring_r = 0;
    int src_off = 2 * PH_SRC;
#pragma unroll 1
    for (int c = 0; c < g.nchunks; ++c) {
        const uint32_t hbase = lds0 + (uint32_t)((c & 1) * HBUF);
        const bool next_real = c + 1 < g.nchunks;
        bk_static_for(std::make_integer_sequence<int, PPC>{}, [&](auto pc_) {
            constexpr int PH = decltype(pc_)::value;
            // ================= load half: fragments of the first taps of this phase, then the direct-to-LDS pieces of phase u + 2
            const uint32_t bbase = (uint32_t)ring_r;
            bi32x4_t a[UPP][FM], b[UPP][FN];
            auto a_addr = [&](int i, int tap) -> uint32_t {
                uint32_t w;
                if constexpr (NB_LDS) w = nbc[i]; else w = nb[i][tap >> 1];     // NB_LDS: nbc holds exactly this phase's two taps
                return hbase + (((tap & 1) ? (w >> 16) : (w & 0xffffu)) ^ fq16);
            };
            constexpr int EARLY = NOUT == 256 ? 1 : 3;        // taps whose fragments are read in the load half; the rest ride in the compute half
#pragma unroll
            for (int k = 0; k < EARLY; ++k) {
#pragma unroll
                for (int j = 0; j < FN; ++j) BK_DSR(b[k][j], b_lane[j] + bbase + (uint32_t)(k * UNIT), 0);
#pragma unroll
                for (int i = 0; i < FM; ++i) BK_DSR(a[k][i], a_addr(i, PH * UPP + k), 0);
            }
            issue_phase((ring_r + 2 * PHASE_BYTES) & (RING * PHASE_BYTES - 1), src_off);
            if (PH == 1) issue_halo(c + 1, next_real);
            bk_wait_vmcnt<WPIECES + (PH == 1 ? HPW : 0) + (PH == 2 ? HPW : 0)>();
            ring_r = (ring_r + PHASE_BYTES) & (RING * PHASE_BYTES - 1);
            src_off += PH_SRC;
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int k = 0; k < EARLY; ++k) bk_wait_frags<0, FM, FN>(a[k], b[k]);
            __builtin_amdgcn_sched_barrier(0);
            // ================= compute half
            __builtin_amdgcn_s_setprio(1);
#define BK_MM(i, j, A, Bv) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, Bv), acc[i][j], 0, 0, 0)
#pragma unroll
            for (int k = 0; k < UPP; ++k) {
                // the fragments of tap k + EARLY ride in the issue gaps of tap k's MFMAs: one or two reads behind every MFMA, the first MFMA
                // first (a compute half that opens with its reads leaves the matrix pipe idle while the other group is in its load half)
                constexpr int NR = FM + FN, NM = FM * FN, RPM = (NR + NM - 1) / NM;
                const bool ride = k + EARLY < UPP;
                int rd = 0;
#pragma unroll
                for (int mi = 0; mi < NM; ++mi) {
                    const int j = mi / FM, i = mi % FM;
                    BK_MM(i, j, a[k][i], b[k][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (NB_LDS && k == 0 && mi < FM) {
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(nbn[mi]) : "v"(nb_row[mi]), "n"(4 * ((PH + 1) % PPC)));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ride) {
#pragma unroll
                        for (int q = 0; q < RPM; ++q, ++rd) {
                            if (rd < FN) BK_DSR(b[k + EARLY][rd], b_lane[rd] + bbase + (uint32_t)((k + EARLY) * UNIT), 0);
                            else if (rd < NR) BK_DSR(a[k + EARLY][rd - FN], a_addr(rd - FN, PH * UPP + k + EARLY), 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (k + 1 < UPP) {
                    // tap k + 1's fragments were issued EARLY taps ago: at most the younger taps' reads may still be in flight
                    if (k + 1 + EARLY <= UPP) bk_wait_frags<(EARLY - 1) * (FM + FN), FM, FN>(a[k + 1], b[k + 1]);
                    else bk_wait_frags<0, FM, FN>(a[k + 1], b[k + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef BK_MM
            if constexpr (NB_LDS) {
                // every LDS read of this half has been waited for above (lgkmcnt(0) before the last tap's MFMAs)
                asm volatile("" : "+v"(nbn[0]), "+v"(nbn[1]), "+v"(nbn[2]), "+v"(nbn[3]));
#pragma unroll
                for (int i = 0; i < FM; ++i) nbc[i] = nbn[i];
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        });
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's last barrier
    bk_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: 64 rows x NOUT channels (fp32) at a time through the weight ring, then coalesced rows with bias / addend in fp32
    float* sC = reinterpret_cast<float*>(smem + RING_OFF);
    const int* rsort = rows_sorted + tile.row0;
    const int V = g.D * g.H * g.W;
    constexpr int PASSES = TROWS / 64;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        constexpr int FPP = 2;                               // fragments (32 rows) per pass
        if (wm == (p * FPP) / FM) {
#pragma unroll
            for (int ih = 0; ih < FPP; ++ih) {
                const int i = (p * FPP + ih) % FM;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = ih * 32 + (e & 3) + 8 * (e >> 2) + 4 * fq;
                        sC[row * NOUT + wn * (32 * FN) + j * 32 + fr] = acc[i][j][e];
                    }
            }
        }
        __syncthreads();
        constexpr int CPR = NOUT / 8;                        // 8-channel chunks per row
#pragma unroll
        for (int it = 0; it < (64 * CPR) / 512; ++it) {
            const int cidx = t + 512 * it;
            const int row = cidx / CPR, c8 = (cidx % CPR) * 8;
            const int trow = p * 64 + row;
            if (trow < tile.nrows) {
                const int vox = rsort[trow];
                const float4 lo = *reinterpret_cast<const float4*>(sC + row * NOUT + c8), hi = *reinterpret_cast<const float4*>(sC + row * NOUT + c8 + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias[c8 + e];
                }
                if (addend) {
                    const int bb = vox / V, r1 = vox - bb * V, z = r1 / (g.H * g.W), r2 = r1 - z * (g.H * g.W), y = r2 / g.W, x = r2 - y * g.W;
                    const TO* ap = addend + ((size_t)((bb * g.Da + (z >> g.add_shift)) * g.Ha + (y >> g.add_shift)) * g.Wa + (x >> g.add_shift)) * NOUT + c8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += Elem<TO>::ld(ap + e);
                }
                TO* dst = out + (size_t)vox * NOUT + c8;
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
        }
        __syncthreads();
    }
}

// torch weight [Cout][Cin][3][3][3] fp32 -> [red / 16 chunks][28 taps][NOUT rows][16] bf16 (tap 27 = zeros).
//   forward      : rows = co (NOUT = Cout), red = ci:   pack[c][t][co][k] = W[co][16c+k][t]
//   data gradient: rows = ci (NOUT = Cin),  red = co:   pack[c][t][ci][k] = W[16c+k][ci][26-t]
__global__ __launch_bounds__(256) void pack_weight_brick_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int transposed)
{
    const int rows = transposed ? Cin : Cout, red = transposed ? Cout : Cin;
    const int unit = blockIdx.x;                 // (chunk, tap)
    const int c = unit / brick::TAPS, tp = unit - c * brick::TAPS;
    for (int i = threadIdx.x; i < rows * 2; i += 256) {
        const int row = i >> 1, k8 = (i & 1) * 8;
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = c * 16 + k8 + 2 * e + h;
                v[h] = (k < red && tp < 27) ? (transposed ? w[((size_t)k * Cin + row) * 27 + (26 - tp)] : w[((size_t)row * Cin + k) * 27 + tp]) : 0.f;
            }
            pk[e] = f2bf2(v[0], v[1]);
        }
        *reinterpret_cast<uint4*>(out + ((size_t)unit * rows + row) * 16 + k8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

// ------------------------------------------------------------------------------------------------ tile construction
// flags: uint8 [B,D,H,W] (an active-set flag volume of dreg_active_sets).  Bricks of 8^3 cells in raster order per grid.
__global__ __launch_bounds__(512) void brick_count_kernel(const uint8_t* __restrict__ flags, int* __restrict__ cnt, int B, int D, int H, int W, int nbz, int nby, int nbx)
{
    using namespace brick;
    const int bid = blockIdx.x;                  // ((b * nbz + bz) * nby + by) * nbx + bx
    int r = bid;
    const int bx = r % nbx; r /= nbx;
    const int by = r % nby; r /= nby;
    const int bz = r % nbz; const int b = r / nbz;
    const int l = threadIdx.x, lz = l >> 6, ly = (l >> 3) & 7, lx = l & 7;
    const int z = bz * BS + lz, y = by * BS + ly, x = bx * BS + lx;
    const int f = (z < D && y < H && x < W) ? (flags[(((size_t)b * D + z) * H + y) * W + x] ? 1 : 0) : 0;
    const unsigned long long m = __ballot(f);
    __shared__ int wc[8];
    if ((l & 63) == 0) wc[l >> 6] = __popcll(m);
    __syncthreads();
    if (l == 0) { int s = 0; for (int k = 0; k < 8; ++k) s += wc[k]; cnt[bid] = s; }
}
// one workgroup: exclusive scan of the brick counts (in place), per-grid row ranges and candidate-tile ranges.
// meta: [0] = total rows, [1] = number of candidate tiles, [2] = tiles emitted (zeroed here), [3] = overflow flag (zeroed here);
// grid_row0 [B+1], grid_cand0 [B+1].
__global__ __launch_bounds__(1024) void brick_scan_kernel(int* __restrict__ cnt, int nbricks, int bricks_per_grid, int B, int* __restrict__ meta,
                                                          int* __restrict__ grid_row0, int* __restrict__ grid_cand0)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nbricks + 1023) / 1024;
    const int i0 = t * per, i1 = min(i0 + per, nbricks);
    int s = 0;
    for (int i = i0; i < i1; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = i0; i < i1; ++i) { const int c = cnt[i]; cnt[i] = run; run += c; }
    __syncthreads();
    if (t == 0) {
        const int total = part[1023];
        int cand = 0;
        for (int b = 0; b <= B; ++b) {
            const int r0 = b < B ? cnt[b * bricks_per_grid] : total;
            grid_row0[b] = r0;
        }
        for (int b = 0; b < B; ++b) {
            grid_cand0[b] = cand;
            cand += (grid_row0[b + 1] - grid_row0[b] + brick::TROWS - 1) / brick::TROWS;
        }
        grid_cand0[B] = cand;
        meta[0] = total; meta[1] = cand; meta[2] = 0; meta[3] = 0;
    }
}
__global__ __launch_bounds__(512) void brick_write_kernel(const uint8_t* __restrict__ flags, const int* __restrict__ off, int* __restrict__ rows_sorted,
                                                          int B, int D, int H, int W, int nbz, int nby, int nbx)
{
    using namespace brick;
    const int bid = blockIdx.x;
    int r = bid;
    const int bx = r % nbx; r /= nbx;
    const int by = r % nby; r /= nby;
    const int bz = r % nbz; const int b = r / nbz;
    const int l = threadIdx.x, lz = l >> 6, ly = (l >> 3) & 7, lx = l & 7;
    const int z = bz * BS + lz, y = by * BS + ly, x = bx * BS + lx;
    const size_t vox = (((size_t)b * D + z) * H + y) * W + x;
    const int f = (z < D && y < H && x < W) ? (flags[vox] ? 1 : 0) : 0;
    const unsigned long long m = __ballot(f);
    __shared__ int wc[8];
    if ((l & 63) == 0) wc[l >> 6] = __popcll(m);
    __syncthreads();
    int base = off[bid];
    for (int k = 0; k < (l >> 6); ++k) base += wc[k];
    if (f) rows_sorted[base + __popcll(m & ((1ull << (l & 63)) - 1ull))] = (int)vox;
}

// One workgroup (256 threads) per candidate tile = up to 256 consecutive rows of one grid in brick-major order.
constexpr int BRICK_MAX_BITMAP_WORDS = 9216;     // (64 + 2)^3 bits = 8,985 words
__global__ __launch_bounds__(256) void brick_tiles_kernel(const int* __restrict__ rows_sorted, const int* __restrict__ grid_row0, const int* __restrict__ grid_cand0,
                                                          int B, int D, int H, int W, int max_tiles, int* __restrict__ meta, BrickTile* __restrict__ tiles,
                                                          int* __restrict__ halo_vox, uint16_t* __restrict__ nbr)
{
    using namespace brick;
    __shared__ uint32_t bm[BRICK_MAX_BITMAP_WORDS];
    __shared__ int scan[256];
    __shared__ int s_lo, s_hi, s_total, s_slot;
    __shared__ int cls[NCLS];
    __shared__ uint16_t slot_of[HCAP];
    __shared__ int work[16][2];
    __shared__ int nwork;
    const int t = threadIdx.x;
    const int cand = blockIdx.x;
    // which grid
    int b = 0;
    while (b + 1 < B && grid_cand0[b + 1] <= cand) ++b;
    const int k = cand - grid_cand0[b];
    const int r_begin = grid_row0[b] + k * TROWS;
    const int r_end = min(r_begin + TROWS, grid_row0[b + 1]);
    if (r_begin >= r_end) return;
    const int Hp = H + 2, Wp = W + 2, plane = Hp * Wp;
    const int V = D * H * W;
    if (t == 0) { work[0][0] = r_begin; work[0][1] = r_end - r_begin; nwork = 1; }
    __syncthreads();
    for (int wi = 0; wi < nwork; ++wi) {
        const int r0 = work[wi][0], n = work[wi][1];
        // this thread's row
        int z = 0, y = 0, x = 0;
        const bool live = t < n;
        if (live) { const int v = rows_sorted[r0 + t] - b * V; z = v / (H * W); const int r2 = v - z * (H * W); y = r2 / W; x = r2 - y * W; }
        // z range of the rows -> bitmap word range of planes [zlo - 1, zhi + 1] (padded coordinates: plane index z + 1)
        if (t == 0) { s_lo = 1 << 30; s_hi = -1; }
        __syncthreads();
        if (live) { atomicMin(&s_lo, z); atomicMax(&s_hi, z); }
        __syncthreads();
        const int bit_lo = s_lo * plane, bit_hi = (s_hi + 3) * plane;          // padded planes s_lo .. s_hi + 2
        const int w_lo = bit_lo >> 5, w_hi = (bit_hi + 31) >> 5;
        const int nw = w_hi - w_lo;
        for (int i = t; i < nw; i += 256) bm[w_lo + i] = 0u;
        __syncthreads();
        if (live) {
            for (int dz = 0; dz < 3; ++dz) for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) {
                const int zz = z + dz - 1, yy = y + dy - 1, xx = x + dx - 1;
                if ((unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    const int bit = ((zz + 1) * Hp + (yy + 1)) * Wp + (xx + 1);
                    atomicOr(&bm[bit >> 5], 1u << (bit & 31));
                }
            }
        }
        __syncthreads();
        // ranks: every thread owns a contiguous run of words
        const int per = (nw + 255) / 256;
        const int i0 = min(t * per, nw), i1 = min(i0 + per, nw);
        int c = 0;
        for (int i = i0; i < i1; ++i) c += __popc(bm[w_lo + i]);
        scan[t] = c;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = t >= o ? scan[t - o] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        if (t == 255) s_total = scan[255];
        __syncthreads();
        const int total = s_total;
        // staged voxels per colour class (slot = 16 * rank within the class + colour): a first pass counts, so that a candidate whose
        // fullest class exceeds the capacity is halved BEFORE a slot is taken
        if (t < NCLS) cls[t] = 0;
        __syncthreads();
        for (int i = i0; i < i1; ++i) {
            uint32_t w = bm[w_lo + i];
            while (w) {
                const int bpos = __ffs(w) - 1;
                w &= w - 1;
                const int bit = ((w_lo + i) << 5) + bpos;
                const int r2 = bit % plane, yy = r2 / Wp - 1, xx = r2 % Wp - 1;
                atomicAdd(&cls[colour(yy, xx)], 1);
            }
        }
        __syncthreads();
        int cmax = 0;
        for (int cc = 0; cc < NCLS; ++cc) cmax = max(cmax, cls[cc]);
        __syncthreads();
        if ((COLOUR_SLOTS ? cmax > CLS_CAP : total > ZSLOT) && n > 32) {
            // too many staged voxels of one colour for the LDS of the convolution kernel: two halves (fragment granularity), tried again
            if (t == 0) {
                const int h = ((n / 2) + 31) / 32 * 32;
                const int q = nwork;
                if (q + 2 <= 16) { work[q][0] = r0; work[q][1] = h; work[q + 1][0] = r0 + h; work[q + 1][1] = n - h; nwork = q + 2; }
                else meta[3] = 1;
            }
            __syncthreads();
            continue;
        }
        if (t == 0) {
            const int s = atomicAdd(&meta[2], 1);
            s_slot = s < max_tiles ? s : -1;
            if (s >= max_tiles || (COLOUR_SLOTS ? cmax > CLS_CAP : total > ZSLOT)) meta[3] = 1;
        }
        if (t < NCLS) cls[t] = 0;
        __syncthreads();
        const int slot = s_slot;
        if (slot < 0) { __syncthreads(); continue; }
        int* hv = halo_vox + (size_t)slot * HCAP;
        for (int i = t; i < HCAP; i += 256) hv[i] = -1;
        __syncthreads();
        // slots: rank (raster order, from the scan) -> 16 * (arrival within its colour class) + colour
        {
            int rk = scan[t] - c;
            for (int i = i0; i < i1; ++i) {
                uint32_t w = bm[w_lo + i];
                while (w) {
                    const int bpos = __ffs(w) - 1;
                    w &= w - 1;
                    const int bit = ((w_lo + i) << 5) + bpos;
                    const int zz = bit / plane - 1, r2 = bit % plane, yy = r2 / Wp - 1, xx = r2 % Wp - 1;
                    const int cc = colour(yy, xx);
                    const int sl = COLOUR_SLOTS ? NCLS * atomicAdd(&cls[cc], 1) + cc : rk;
                    slot_of[rk++] = (uint16_t)sl;
                    hv[sl] = b * V + (zz * H + yy) * W + xx;
                }
            }
        }
        __syncthreads();
        // the table: rank of a bit = ranks of the words before it in its owner's run + popcount below it.  Owner of word i = i / per.
        {
            uint16_t* row_out = nbr + ((size_t)slot * TROWS + t) * TAPS;
            if (live) {
                for (int tp = 0; tp < 27; ++tp) {
                    const int dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
                    const int zz = z + dz - 1, yy = y + dy - 1, xx = x + dx - 1;
                    uint16_t o = slot_off(ZSLOT + colour(yy, xx));
                    if ((unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                        const int bit = ((zz + 1) * Hp + (yy + 1)) * Wp + (xx + 1);
                        const int wi_ = (bit >> 5) - w_lo;
                        const int owner = wi_ / per;
                        int rk = owner ? scan[owner - 1] : 0;
                        for (int q = owner * per; q < wi_; ++q) rk += __popc(bm[w_lo + q]);
                        rk += __popc(bm[w_lo + wi_] & ((1u << (bit & 31)) - 1u));
                        o = slot_off(slot_of[rk]);
                    }
                    row_out[tp] = o;
                }
                row_out[27] = slot_off(ZSLOT + colour(y, x));
            } else {
                for (int tp = 0; tp < TAPS; ++tp) row_out[tp] = slot_off(ZSLOT + (t & 15));
            }
        }
        if (t == 0) s_total = COLOUR_SLOTS ? NCLS * cmax : total;      // slots in use: [0, 16 * fullest class) / [0, staged voxels)
        __syncthreads();
        const int nslots = s_total;
        if (t == 0) { BrickTile tl; tl.row0 = r0; tl.nrows = n; tl.nhalo = nslots; tl.pad = total; tiles[slot] = tl; }
        __syncthreads();
    }
}

extern "C" {

// 1 when a [B,D,H,W] grid can be tiled (the builder's LDS bitmap covers the padded grid) and the operands fit 32-bit buffer offsets
int dreg_brick_supported(int B, int D, int H, int W, int Cin, int Cout)
{
    if ((Cout != 256 && Cout != 64) || Cin % 16 != 0 || Cin < 16) return 0;
    if ((size_t)(D + 2) * (H + 2) * (W + 2) > (size_t)BRICK_MAX_BITMAP_WORDS * 32) return 0;
    if ((uint64_t)B * D * H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffff00ull) return 0;
    return 1;
}
size_t dreg_conv3_brick_pack_bytes(int rows, int red) { return (size_t)(red / 16) * brick::TAPS * rows * 32; }
// w: torch layout fp32 [Cout][Cin][27]; transposed = 0: forward pack (rows = Cout in {256, 64}), 1: data-gradient pack (rows = Cin in {256, 64})
int dreg_pack_conv_weight_brick(const float* w, void* out, int Cout, int Cin, int transposed, void* stream)
{
    const int rows = transposed ? Cin : Cout, red = transposed ? Cout : Cin;
    if ((rows != 256 && rows != 64) || red % 16 != 0) return DREG_EINVAL;
    hipLaunchKernelGGL(pack_weight_brick_kernel, dim3((red / 16) * brick::TAPS), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)out, Cout, Cin, transposed);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// Tile tables of one active set.  max_tiles: capacity of the tables (a candidate tile = 256 rows may split: 2 * ceil(rows / 256) + B + 8
// never overflowed in the sweeps of tools/bench_conv_brick.py; on overflow meta[3] = 1 and the caller keeps the row-list kernel).
//   workspace: int32 [nbricks + 2 * (B + 1)] (brick offsets, per-grid ranges)
//   meta int32 [4] = (rows, candidate tiles, tiles emitted, overflow); rows_sorted int32 [>= rows]; tiles [max_tiles] x 16 B;
//   halo_vox int32 [max_tiles][1280]; nbr uint16 [max_tiles][256][28].
size_t dreg_brick_tiles_workspace_bytes(int B, int D, int H, int W)
{
    const size_t nb = (size_t)B * ((D + 7) / 8) * ((H + 7) / 8) * ((W + 7) / 8);
    return (nb + 2 * (size_t)(B + 1) + 64) * sizeof(int);
}
int dreg_brick_tiles_build(const uint8_t* flags, int B, int D, int H, int W, int max_rows, int max_tiles, void* workspace, size_t workspace_bytes,
                           int* meta, int* rows_sorted, void* tiles, int* halo_vox, void* nbr, void* stream)
{
    using namespace brick;
    if (!dreg_brick_supported(B, D, H, W, 16, 256) || workspace_bytes < dreg_brick_tiles_workspace_bytes(B, D, H, W)) return DREG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nbz = (D + 7) / 8, nby = (H + 7) / 8, nbx = (W + 7) / 8;
    const int nbricks = B * nbz * nby * nbx;
    int* cnt = (int*)workspace;
    int* grid_row0 = cnt + nbricks;
    int* grid_cand0 = grid_row0 + (B + 1);
    hipLaunchKernelGGL(brick_count_kernel, dim3(nbricks), dim3(512), 0, st, flags, cnt, B, D, H, W, nbz, nby, nbx);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(brick_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, nbricks, nbz * nby * nbx, B, meta, grid_row0, grid_cand0);
    DREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(brick_write_kernel, dim3(nbricks), dim3(512), 0, st, flags, cnt, rows_sorted, B, D, H, W, nbz, nby, nbx);
    DREG_LAUNCH_CHECK();
    // candidate tiles: at most ceil(max_rows / 256) + B (one partial tile per grid); workgroups past the real count return at once
    const int max_cand = (max_rows + TROWS - 1) / TROWS + B;
    hipLaunchKernelGGL(brick_tiles_kernel, dim3(max_cand), dim3(256), 0, st, rows_sorted, grid_row0, grid_cand0, B, D, H, W, max_tiles, meta,
                       (BrickTile*)tiles, halo_vox, (uint16_t*)nbr);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// out[rows] = bias + addend(...) + 3^3 convolution of `in` at the tiled rows (all other rows of out untouched).
// in [B,D,H,W,Cin] bf16, wpk from dreg_pack_conv_weight_brick (rows = Cout), out [B,D,H,W,Cout] bf16 (out_f32: fp32), Cout in {256, 64};
// addend [B,Da,Ha,Wa,Cout] added with nearest x2 upsampling (add_same = 0) or element-wise (add_same = 1), or null.
int dreg_conv3_brick(const void* in, const void* wpk, void* out, const float* bias, const void* addend, const void* tiles, int ntiles,
                     const int* halo_vox, const void* nbr, const int* rows_sorted,
                     int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream)
{
    using namespace brick;
    if (!dreg_brick_supported(B, D, H, W, Cin, Cout)) return DREG_EINVAL;
    if (ntiles <= 0) return DREG_OK;
    BrickGeom g;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.nchunks = Cin / CK;
    g.Da = Da; g.Ha = Ha; g.Wa = Wa; g.add_shift = add_same ? 0 : 1;
    const uint32_t in_bytes = (uint32_t)((uint64_t)B * D * H * W * Cin * 2), wt_bytes = (uint32_t)dreg_conv3_brick_pack_bytes(Cout, Cin);
    hipStream_t st = (hipStream_t)stream;
    const int lds = 2 * HBUF + RING * PHASE_BYTES + (Cout == 256 ? TROWS * TAPS * 2 : 0);
#define BK_LAUNCH(NO, TOt) do { \
        (void)hipFuncSetAttribute((const void*)conv3_brick_kernel<NO, TOt>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv3_brick_kernel<NO, TOt>), dim3(ntiles), dim3(512), lds, st, (const bf16_t*)in, (const bf16_t*)wpk, (TOt*)out, bias, \
                           (const TOt*)addend, g, (const BrickTile*)tiles, halo_vox, (const uint16_t*)nbr, rows_sorted, in_bytes, wt_bytes); } while (0)
    if (Cout == 256) { if (out_f32) BK_LAUNCH(256, float); else BK_LAUNCH(256, bf16_t); }
    else { if (out_f32) BK_LAUNCH(64, float); else BK_LAUNCH(64, bf16_t); }
#undef BK_LAUNCH
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

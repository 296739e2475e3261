// Surface-field visibility of query points from a NeRF block's training cameras (row N1 of SURVEY.md §8f), gfx950.
//
// Replaces, in ONE kernel, the reference's chain for every (camera, point) ray
//   nerfacc.ray_aabb_intersect -> nerfacc _C.ray_marching (two passes, sample list in HBM) -> tcnn density at every sample ->
//   CUB segmented exclusive cumprod (transmittance) -> torch_scatter.scatter_max(alpha*T) -> >= cut_off -> max over cameras
//   (conerf/utils/nerfacc_utils.py:84-222, conerf/loss/confidence_loss.py:56-160, conerf/register/sample_grid.py:244-318).
// One lane = one ray, 64 rays per wave march in lock step: every lane advances to its next lattice sample
// t_mid = t_min + (n + 1/2) dt that falls in an occupied cell of the 128^3 binary grid (empty cells are skipped to the
// cell's exit, as nerfacc does), the wave evaluates the 64 densities together (hash-grid gather per lane, 32->64->16 MLP on
// fp16 MFMA through LDS, as ngp_density_kernel), then each lane updates T and max(alpha*T).  Only the binary label is
// needed, so a ray stops as soon as the label is decided: hit (max >= cut_off), or T < max(cut_off, early_stop_eps)
// (alpha*T <= T can no longer reach cut_off) — no sample list, no later samples evaluated.  Labels of a point are OR-ed
// over cameras with one atomic per hit.  nerfacc 0.3.5 / tcnn are absent from the reference tree: parity unpinned.
#include "common.h"
#include <cstring>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct NgpLevelsV {
    uint32_t offset[16], size[16], res[16];
    float scale[16];
    uint32_t hashed[16];
};
struct VisArgs {
    const float* cams;     // [Nc,3] camera centres
    const float* pts;      // [Np,3]
    const uint8_t* binary; // [rx,ry,rz] occupancy
    int* label;            // [Np] OR over cameras
    const _Float16 *table, *w1, *w2;
    NgpLevelsV lv;
    float roi[6], scene[6], model[6];
    int rx, ry, rz, Nc, Np;
    float dt, cut_off, early_eps, alpha_thre;
    int max_steps;
    const uint32_t* coarse;   // optional: one bit per 4^3 block of `binary` (dreg_occupancy_coarse_bits), <= 32,768 bits; null = none
    int cx, cy, cz;           // its extents: ceil(r / 4)
    unsigned long long* queue;   // persistent forms: this block's ray counter (zeroed by the caller)
};

__device__ __forceinline__ uint32_t vgrid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t size, uint32_t hashed) {
    uint32_t idx = hashed ? (x ^ (y * 2654435761u) ^ (z * 805459861u)) : (x + y * res + z * res * res);
    return idx % size;
}

__global__ __launch_bounds__(64) void surface_visibility_kernel(VisArgs a)
{
    constexpr int XRS = 32 * 2 + 16, HRS = 64 * 2 + 16;
    __shared__ __attribute__((aligned(16))) char smem[64 * XRS + 64 * HRS + 64 * 4];
    char* sX = smem;
    char* sH = sX + 64 * XRS;
    float* sOut = reinterpret_cast<float*>(sH + 64 * HRS);
    const int lane = threadIdx.x;
    const long ray = (long)blockIdx.x * 64 + lane;
    const long nrays = (long)a.Nc * a.Np;
    bool done = ray >= nrays;
    const int c = done ? 0 : (int)(ray / a.Np), p = done ? 0 : (int)(ray - (long)c * a.Np);
    float o[3], d[3], tmax = 0.f, tmin = 0.f;
    {
        float n2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = a.cams[c * 3 + k]; d[k] = a.pts[(long)p * 3 + k] - o[k]; n2 += d[k] * d[k]; }
        tmax = sqrtf(n2);
        const float inv = tmax > 0.f ? 1.f / tmax : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] *= inv;
        // slab test against the scene aabb; marching starts at max(near, 0)
        float near = -1e30f, far = 1e30f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float id = 1.f / d[k];
            float t0 = (a.scene[k] - o[k]) * id, t1 = (a.scene[3 + k] - o[k]) * id;
            if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; }
            near = fmaxf(near, t0); far = fminf(far, t1);
        }
        if (!(near <= far) || far <= 0.f || tmax <= 0.f) done = true;
        tmin = fmaxf(near, 0.f);
    }
    float T = 1.f, best = 0.f;
    int n = 0;
    const float roi_ext[3] = {a.roi[3] - a.roi[0], a.roi[4] - a.roi[1], a.roi[5] - a.roi[2]};
    const int rdim[3] = {a.rx, a.ry, a.rz};
    const float stop_T = fmaxf(a.cut_off, a.early_eps);
    const int fr = lane & 15, kg = lane >> 4;
    f16x8_t w1f[4], w2f[2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) w1f[cb] = *reinterpret_cast<const f16x8_t*>(a.w1 + (cb * 16 + fr) * 32 + kg * 8);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) w2f[kb] = *reinterpret_cast<const f16x8_t*>(a.w2 + fr * 64 + kb * 32 + kg * 8);

    for (int iter = 0; iter < a.max_steps; ++iter) {
        // ---- advance to the next lattice sample inside an occupied cell
        bool have = false;
        float x[3] = {0.f, 0.f, 0.f};
        for (int guard = 0; !done && !have && guard < 4096; ++guard) {
            const float tm = tmin + ((float)n + 0.5f) * a.dt;
            if (tm >= tmax) { done = true; break; }
            float u[3];
            bool inside = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) { x[k] = o[k] + tm * d[k]; u[k] = (x[k] - a.roi[k]) / roi_ext[k]; inside = inside && u[k] >= 0.f && u[k] <= 1.f; }
            int ci[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ci[k] = min(max((int)floorf(u[k] * (float)rdim[k]), 0), rdim[k] - 1);
            const bool occ = inside && a.binary[((long)ci[0] * a.ry + ci[1]) * a.rz + ci[2]] != 0;
            if (occ) { have = true; break; }
            // skip to the exit of this cell (or, outside the roi, just step): smallest positive distance to a cell face
            float texit = 1e30f;
            if (inside) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (d[k] != 0.f) {
                        const float face = a.roi[k] + (float)(ci[k] + (d[k] > 0.f ? 1 : 0)) * roi_ext[k] / (float)rdim[k];
                        texit = fminf(texit, fmaxf((face - x[k]) / d[k], 0.f));
                    }
                }
            } else texit = 0.f;
            const int skip = (int)floorf(texit / a.dt - 1e-3f);   // lattice points strictly inside the remaining empty stretch (conservative)
            n += 1 + max(skip, 0);
        }
        if (!__any(have)) break;
        // ---- density of the 64 samples (as ngp_density_kernel)
        float u[3];
        bool inside_m = have;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u[k] = (x[k] - a.model[k]) / (a.model[3 + k] - a.model[k]);
            inside_m = inside_m && u[k] > 0.f && u[k] < 1.f;
            u[k] = fminf(fmaxf(u[k], 0.f), 1.f);
        }
#pragma unroll 1
        for (int l = 0; l < 16; ++l) {
            float f0 = 0.f, f1 = 0.f;
            if (have) {
                const float sc = a.lv.scale[l];
                const uint32_t res = a.lv.res[l], size = a.lv.size[l], hashed = a.lv.hashed[l];
                const _Float16* tl = a.table + (size_t)a.lv.offset[l] * 2;
                float w[3];
                uint32_t g[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float pos = u[k] * sc + 0.5f; const float fl = floorf(pos); g[k] = (uint32_t)fl; w[k] = pos - fl; }
#pragma unroll
                for (int corner = 0; corner < 8; ++corner) {
                    const uint32_t cx = g[0] + (corner & 1), cy = g[1] + ((corner >> 1) & 1), cz = g[2] + ((corner >> 2) & 1);
                    const float wt = ((corner & 1) ? w[0] : 1.f - w[0]) * ((corner & 2) ? w[1] : 1.f - w[1]) * ((corner & 4) ? w[2] : 1.f - w[2]);
                    union { uint32_t u32; _Float16 h[2]; } cv;
                    cv.u32 = *reinterpret_cast<const uint32_t*>(tl + (size_t)vgrid_index(cx, cy, cz, res, size, hashed) * 2);
                    f0 += wt * (float)cv.h[0]; f1 += wt * (float)cv.h[1];
                }
            }
            _Float16* xr = reinterpret_cast<_Float16*>(sX + lane * XRS);
            xr[2 * l] = (_Float16)f0; xr[2 * l + 1] = (_Float16)f1;
        }
        __syncthreads();
        f32x4_t acc[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const f16x8_t af = *reinterpret_cast<const f16x8_t*>(sX + (rb * 16 + fr) * XRS + kg * 16);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, w1f[cb], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    reinterpret_cast<_Float16*>(sH + (rb * 16 + kg * 4 + r) * HRS)[cb * 16 + fr] = (_Float16)fmaxf(acc[rb][cb][r], 0.f);
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4_t ov = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                ov = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8_t*>(sH + (rb * 16 + fr) * HRS + (kb * 32 + kg * 8) * 2), w2f[kb], ov, 0, 0, 0);
            if (fr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sOut[rb * 16 + kg * 4 + r] = (float)(_Float16)ov[r];
            }
        }
        __syncthreads();
        if (have) {
            const float sigma = inside_m ? __expf(sOut[lane] - 1.f) : 0.f;
            const float alpha = 1.f - __expf(-sigma * a.dt);
            const bool vis = T >= a.early_eps && (a.alpha_thre <= 0.f || alpha >= a.alpha_thre);
            if (vis) best = fmaxf(best, alpha * T);
            T *= (1.f - alpha);
            ++n;
            if (best >= a.cut_off || T < stop_T) done = true;
        }
        __syncthreads();
    }
    if (ray < nrays && best >= a.cut_off) atomicOr(a.label + p, 1);
}

// ------------------------------------------------------------------------------------------------
// Persistent form (round 3).  The lock-step kernel above gives every wave 64 rays and runs until the LONGEST of them is decided: the
// others' lanes idle through the hash-grid gathers and the MLP of every remaining iteration, and a point already seen from one camera
// is marched from all the others.  Here a wave keeps its 64 lanes full: a lane whose ray is decided takes the next ray from a queue
// (one atomic per wave and refill), and a ray whose point already carries the label is not marched at all (rays are queued camera by
// camera, so the later cameras mostly find the label set).  Per-ray arithmetic is unchanged: the labels are the same.
// The MLP stores its hidden layer as in ngp_density_kernel: products formed transposed (weights as the MFMA's first operand), so a lane
// holds four consecutive hidden units of one sample and writes them with one 8-byte LDS store instead of 64 two-byte ones.
#ifdef DREG_PROBE
__device__ long g_pass_bound = 1L << 22;   // passes of the march loop per wave (test hook of the measurement build: dreg_visibility_set_pass_bound)
#else
static constexpr long g_pass_bound = 1L << 22;
#endif
__device__ __forceinline__ void vwave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void vis_march_queue(const VisArgs& a, unsigned long long* __restrict__ queue)
{
    constexpr int XRS = 32 * 2 + 16, HRS = 64 * 2 + 16;
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    __shared__ __attribute__((aligned(16))) char smem[64 * XRS + 64 * HRS + 64 * 4];
    char* sX = smem;
    char* sH = sX + 64 * XRS;
    float* sOut = reinterpret_cast<float*>(sH + 64 * HRS);
    const int lane = threadIdx.x;
    // the coarse occupancy (one bit per 4^3 cells) in LDS: a ray walks the empty space of a block — from the aabb's face to the surface —
    // in coarse cells looked up here (~100 cycles) instead of fine cells looked up in global memory (a dependent load of 1-2 us each);
    // every pass of the loop below waits for its slowest lane, and a freshly queued ray's walk WAS that lane
    __shared__ uint32_t sCoarse[1024];
    const bool use_coarse = a.coarse != nullptr;
    if (use_coarse) {
        const int nw = (a.cx * a.cy * a.cz + 31) / 32;
        for (int i = lane; i < nw; i += 64) sCoarse[i] = a.coarse[i];
    }
    __syncthreads();
    const unsigned long long nrays = (unsigned long long)a.Nc * (unsigned long long)a.Np;
    const float roi_ext[3] = {a.roi[3] - a.roi[0], a.roi[4] - a.roi[1], a.roi[5] - a.roi[2]};
    const int rdim[3] = {a.rx, a.ry, a.rz};
    const float stop_T = fmaxf(a.cut_off, a.early_eps);
    const int fr = lane & 15, kg = lane >> 4;
    f16x8_t w1f[4], w2f[2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) w1f[cb] = *reinterpret_cast<const f16x8_t*>(a.w1 + (cb * 16 + fr) * 32 + kg * 8);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) w2f[kb] = *reinterpret_cast<const f16x8_t*>(a.w2 + fr * 64 + kb * 32 + kg * 8);

    // per-lane ray state
    bool active = false, exhausted = false;
    int p = 0, n = 0;
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f}, tmax = 0.f, tmin = 0.f, T = 1.f, best = 0.f;

    bool finished = false;
    const long pass_bound = g_pass_bound;
    for (long it = 0; it < pass_bound; ++it) {                    // (bound: a safety net, never reached — every pass advances every live ray)
        // ---- refill: lanes without a ray take the next ones from the queue, skipping rays whose point is already labelled
        for (int tries = 0; tries < 1024; ++tries) {
            const bool need = !active && !exhausted;
            const unsigned long long mask = __ballot(need);
            if (!mask) break;
            unsigned long long base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) base = atomicAdd(queue, (unsigned long long)__popcll(mask));
            base = __shfl(base, leader, 64);
            if (need) {
                const unsigned long long ray = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
                if (ray >= nrays) exhausted = true;
                else {
                    const int c = (int)(ray / (unsigned long long)a.Np);
                    p = (int)(ray - (unsigned long long)c * (unsigned long long)a.Np);
                    if (__hip_atomic_load(a.label + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {      // not yet seen from an earlier camera (device-scope load: other waves set it)
                        float n2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) { o[k] = a.cams[c * 3 + k]; d[k] = a.pts[(long)p * 3 + k] - o[k]; n2 += d[k] * d[k]; }
                        tmax = sqrtf(n2);
                        const float inv = tmax > 0.f ? 1.f / tmax : 0.f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) d[k] *= inv;
                        float near = -1e30f, far = 1e30f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float id = 1.f / d[k];
                            float t0 = (a.scene[k] - o[k]) * id, t1 = (a.scene[3 + k] - o[k]) * id;
                            if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; }
                            near = fmaxf(near, t0); far = fminf(far, t1);
                        }
                        if ((near <= far) && far > 0.f && tmax > 0.f) { active = true; tmin = fmaxf(near, 0.f); T = 1.f; best = 0.f; n = 0; }
                    }
                }
            }
        }
        if (!__any(active)) {
            if (__all(exhausted)) { finished = true; break; }        // the queue is empty and nothing is in flight
            continue;                                                // (a long run of skipped rays used up this pass's refill rounds)
        }
        // ---- advance every live ray to its next lattice sample inside an occupied cell
        bool have = false;
        float x[3] = {0.f, 0.f, 0.f};
        for (int guard = 0; active && !have && guard < 4096; ++guard) {
            const float tm = tmin + ((float)n + 0.5f) * a.dt;
            if (tm >= tmax) { active = false; break; }
            float u[3];
            bool inside = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) { x[k] = o[k] + tm * d[k]; u[k] = (x[k] - a.roi[k]) / roi_ext[k]; inside = inside && u[k] >= 0.f && u[k] <= 1.f; }
            int ci[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ci[k] = min(max((int)floorf(u[k] * (float)rdim[k]), 0), rdim[k] - 1);
            // an empty COARSE cell holds no occupied fine cell: skip to ITS exit (the same lattice samples are visited as cell by cell:
            // only samples inside empty fine cells are passed over, and the skip count is the same conservative floor)
            int cell_lo[3] = {ci[0], ci[1], ci[2]}, cell_w = 1;
            bool occ = false;
            if (inside) {
                bool coarse_empty = false;
                if (use_coarse) {
                    const int b = ((ci[0] >> 2) * a.cy + (ci[1] >> 2)) * a.cz + (ci[2] >> 2);
                    coarse_empty = ((sCoarse[b >> 5] >> (b & 31)) & 1u) == 0u;
                }
                if (coarse_empty) { cell_lo[0] = ci[0] & ~3; cell_lo[1] = ci[1] & ~3; cell_lo[2] = ci[2] & ~3; cell_w = 4; }
                else occ = a.binary[((long)ci[0] * a.ry + ci[1]) * a.rz + ci[2]] != 0;
            }
            if (occ) { have = true; break; }
            float texit = 1e30f;
            if (inside) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (d[k] != 0.f) {
                        const int hi = min(cell_lo[k] + cell_w, rdim[k]);                // (a ragged last coarse cell ends at the grid's face)
                        const float face = a.roi[k] + (float)(d[k] > 0.f ? hi : cell_lo[k]) * roi_ext[k] / (float)rdim[k];
                        texit = fminf(texit, fmaxf((face - x[k]) / d[k], 0.f));
                    }
                }
            } else texit = 0.f;
            const int skip = (int)floorf(texit / a.dt - 1e-3f);
            n += 1 + max(skip, 0);
        }
        if (!__any(have)) continue;
        // ---- density of the 64 samples
        float u[3];
        bool inside_m = have;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u[k] = (x[k] - a.model[k]) / (a.model[3 + k] - a.model[k]);
            inside_m = inside_m && u[k] > 0.f && u[k] < 1.f;
            u[k] = fminf(fmaxf(u[k], 0.f), 1.f);
        }
#pragma unroll 2
        for (int l = 0; l < 16; ++l) {
            float f0 = 0.f, f1 = 0.f;
            if (have) {
                const float sc = a.lv.scale[l];
                const uint32_t res = a.lv.res[l], size = a.lv.size[l], hashed = a.lv.hashed[l];
                const _Float16* tl = a.table + (size_t)a.lv.offset[l] * 2;
                float w[3];
                uint32_t g[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float pos = u[k] * sc + 0.5f; const float fl = floorf(pos); g[k] = (uint32_t)fl; w[k] = pos - fl; }
#pragma unroll
                for (int corner = 0; corner < 8; ++corner) {
                    const uint32_t cx = g[0] + (corner & 1), cy = g[1] + ((corner >> 1) & 1), cz = g[2] + ((corner >> 2) & 1);
                    const float wt = ((corner & 1) ? w[0] : 1.f - w[0]) * ((corner & 2) ? w[1] : 1.f - w[1]) * ((corner & 4) ? w[2] : 1.f - w[2]);
                    union { uint32_t u32; _Float16 h[2]; } cv;
                    cv.u32 = *reinterpret_cast<const uint32_t*>(tl + (size_t)vgrid_index(cx, cy, cz, res, size, hashed) * 2);
                    f0 += wt * (float)cv.h[0]; f1 += wt * (float)cv.h[1];
                }
            }
            _Float16* xr = reinterpret_cast<_Float16*>(sX + lane * XRS);
            xr[2 * l] = (_Float16)f0; xr[2 * l + 1] = (_Float16)f1;
        }
        vwave_sync();
        f32x4_t acc[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const f16x8_t af = *reinterpret_cast<const f16x8_t*>(sX + (rb * 16 + fr) * XRS + kg * 16);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[cb], af, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const f32x4_t v = acc[rb][cb];       // hidden units cb*16 + kg*4 .. +3 of sample rb*16 + fr
                const f16x4_t h = {(_Float16)fmaxf(v[0], 0.f), (_Float16)fmaxf(v[1], 0.f), (_Float16)fmaxf(v[2], 0.f), (_Float16)fmaxf(v[3], 0.f)};
                *reinterpret_cast<f16x4_t*>(sH + (rb * 16 + fr) * HRS + (cb * 16 + kg * 4) * 2) = h;
            }
        vwave_sync();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4_t ov = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                ov = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8_t*>(sH + (rb * 16 + fr) * HRS + (kb * 32 + kg * 8) * 2), w2f[kb], ov, 0, 0, 0);
            if (fr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sOut[rb * 16 + kg * 4 + r] = (float)(_Float16)ov[r];
            }
        }
        vwave_sync();
        if (have) {
            const float sigma = inside_m ? __expf(sOut[lane] - 1.f) : 0.f;
            const float alpha = 1.f - __expf(-sigma * a.dt);
            const bool vis = T >= a.early_eps && (a.alpha_thre <= 0.f || alpha >= a.alpha_thre);
            if (vis) best = fmaxf(best, alpha * T);
            T *= (1.f - alpha);
            ++n;
            if (best >= a.cut_off) { atomicOr(a.label + p, 1); active = false; }
            else if (T < stop_T) active = false;
            // another camera's ray may have labelled this point in the meantime: its label is decided, this ray need not finish (a call's
            // duration is its LONGEST ray — typically a camera on the far side marching through the whole block — not its average one)
            else if ((it & 3) == 3 && __hip_atomic_load(a.label + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) active = false;
        }
        vwave_sync();
    }
    // the bound was reached with rays still queued or in flight: their points stay unlabelled — say so in bit 63 of the ray counter
    // (the caller reads it back: dreg_nerf_amd/visibility.py raises on the next call)
    if (!finished && lane == 0) atomicOr(queue, 1ull << 63);
}

__global__ __launch_bounds__(64) void surface_visibility_persistent_kernel(VisArgs a, unsigned long long* __restrict__ queue)
{
    vis_march_queue(a, queue);
}
// Several blocks in ONE launch (a training step asks for the labels of 8 blocks: one pair of point sets each).  A call's duration is
// its longest ray, not its average one, so eight launches cost eight tails; here every wave works through ALL blocks' queues, starting at
// block (workgroup index mod n): the blocks' tails overlap.  A wave serves one block at a time (its MLP weights live in registers).
__global__ __launch_bounds__(64) void surface_visibility_multi_kernel(const VisArgs* __restrict__ descs, int n)
{
    for (int i = 0; i < n; ++i) {
        const int b = ((int)blockIdx.x + i) % n;
        vis_march_queue(descs[b], descs[b].queue);
        __syncthreads();
    }
}

// one bit per 4^3 block of the occupancy volume: set when any of its cells is occupied (bits must be zeroed by the caller)
__global__ void occupancy_coarse_bits_kernel(const uint8_t* __restrict__ binary, uint32_t* __restrict__ bits, int rx, int ry, int rz, int cx, int cy, int cz)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cx * cy * cz) return;
    const int z0 = (b % cz) * 4, y0 = ((b / cz) % cy) * 4, x0 = (b / (cz * cy)) * 4;
    bool any = false;
    for (int dx = 0; dx < 4 && x0 + dx < rx; ++dx)
        for (int dy = 0; dy < 4 && y0 + dy < ry; ++dy)
            for (int dz = 0; dz < 4 && z0 + dz < rz; ++dz) any = any || binary[((long)(x0 + dx) * ry + y0 + dy) * rz + z0 + dz] != 0;
    if (any) atomicOr(bits + (b >> 5), 1u << (b & 31));
}

extern "C" {

// bits: uint32 [ceil(ceil(rx/4) * ceil(ry/4) * ceil(rz/4) / 32)], zeroed by the caller; bit ((x/4) * cy + y/4) * cz + z/4
int dreg_occupancy_coarse_bits(const uint8_t* binary, uint32_t* bits, int rx, int ry, int rz, void* stream)
{
    const int cx = (rx + 3) / 4, cy = (ry + 3) / 4, cz = (rz + 3) / 4;
    if (!binary || !bits) return DREG_EINVAL;
    hipLaunchKernelGGL(occupancy_coarse_bits_kernel, dim3((cx * cy * cz + 255) / 256), dim3(256), 0, (hipStream_t)stream, binary, bits, rx, ry, rz, cx, cy, cz);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

// label[p] = OR over cameras of (max_samples alpha*T >= cut_off) along the ray camera -> point p (must be zeroed by the caller).
// table/w1/w2: fp16 inference copies of mlp_base.params; level arrays / aabbs are HOST pointers (16 entries / 6 floats).
int dreg_surface_visibility(const float* cams, const float* pts, const uint8_t* binary, int* label,
                            const void* table, const void* w1, const void* w2,
                            const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                            const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                            int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                            float alpha_thre, void* stream)
{
    if ((long)Nc * Np == 0) return DREG_OK;
    VisArgs a;
    a.cams = cams; a.pts = pts; a.binary = binary; a.label = label;
    a.table = (const _Float16*)table; a.w1 = (const _Float16*)w1; a.w2 = (const _Float16*)w2;
    for (int l = 0; l < 16; ++l) { a.lv.offset[l] = offset[l]; a.lv.size[l] = size[l]; a.lv.res[l] = res[l]; a.lv.scale[l] = scale[l]; a.lv.hashed[l] = hashed[l]; }
    for (int k = 0; k < 6; ++k) { a.roi[k] = roi_aabb[k]; a.scene[k] = scene_aabb[k]; a.model[k] = model_aabb[k]; }
    a.rx = rx; a.ry = ry; a.rz = rz; a.Nc = Nc; a.Np = Np;
    a.dt = render_step_size; a.cut_off = cut_off; a.early_eps = early_stop_eps; a.alpha_thre = alpha_thre;
    a.max_steps = 1 << 16;
    a.coarse = nullptr; a.cx = a.cy = a.cz = 0; a.queue = nullptr;
    const long nrays = (long)Nc * Np;
    hipLaunchKernelGGL(surface_visibility_kernel, dim3((unsigned)((nrays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
static void vis_fill(VisArgs& a, const float* cams, const float* pts, const uint8_t* binary, int* label, const void* table, const void* w1, const void* w2,
                     const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                     const float* roi_aabb, const float* scene_aabb, const float* model_aabb, int rx, int ry, int rz, int Nc, int Np,
                     float render_step_size, float cut_off, float early_stop_eps, float alpha_thre, void* queue, const uint32_t* coarse_bits)
{
    a.cams = cams; a.pts = pts; a.binary = binary; a.label = label;
    a.table = (const _Float16*)table; a.w1 = (const _Float16*)w1; a.w2 = (const _Float16*)w2;
    for (int l = 0; l < 16; ++l) { a.lv.offset[l] = offset[l]; a.lv.size[l] = size[l]; a.lv.res[l] = res[l]; a.lv.scale[l] = scale[l]; a.lv.hashed[l] = hashed[l]; }
    for (int k = 0; k < 6; ++k) { a.roi[k] = roi_aabb[k]; a.scene[k] = scene_aabb[k]; a.model[k] = model_aabb[k]; }
    a.rx = rx; a.ry = ry; a.rz = rz; a.Nc = Nc; a.Np = Np;
    a.dt = render_step_size; a.cut_off = cut_off; a.early_eps = early_stop_eps; a.alpha_thre = alpha_thre;
    a.max_steps = 1 << 16;
    a.cx = (rx + 3) / 4; a.cy = (ry + 3) / 4; a.cz = (rz + 3) / 4;
    a.coarse = ((long)a.cx * a.cy * a.cz <= 32768) ? coarse_bits : nullptr;      // (the kernel keeps the bits in 4 KB of LDS)
    a.queue = (unsigned long long*)queue;
}
// Descriptor-table form for several blocks per launch.  The caller owns the table: n records of dreg_surface_visibility_desc_bytes() bytes,
// filled ON THE HOST by dreg_surface_visibility_fill_desc (same arguments as dreg_surface_visibility_queue, one block each), copied to
// the device by the caller, then passed to dreg_surface_visibility_multi.  Labels / queues of every block zeroed by the caller.
size_t dreg_surface_visibility_desc_bytes() { return sizeof(VisArgs); }
int dreg_surface_visibility_fill_desc(void* host_desc, const float* cams, const float* pts, const uint8_t* binary, int* label,
                                      const void* table, const void* w1, const void* w2,
                                      const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                      const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                                      int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                                      float alpha_thre, void* queue, const uint32_t* coarse_bits)
{
    if (!host_desc || !queue) return DREG_EINVAL;
    VisArgs a;
    vis_fill(a, cams, pts, binary, label, table, w1, w2, offset, size, res, scale, hashed, roi_aabb, scene_aabb, model_aabb, rx, ry, rz, Nc, Np,
             render_step_size, cut_off, early_stop_eps, alpha_thre, queue, coarse_bits);
    memcpy(host_desc, &a, sizeof(a));
    return DREG_OK;
}
DREG_KNOB(int, g_vis_waves, 4096);        // tuning (include/dreg_nerf_probe.h): one-wave workgroups of the persistent launches (256 CUs x 16)
// max_waves (0 = the default 4096 = 16 per CU): one-wave workgroups of the persistent launch.  A workgroup holds 18.7 KB of LDS for as long as the launch
// runs, so a full-width launch leaves a CU's LDS to nobody else; a BACKGROUND launch — the training step's gradient-free 'tilde' labels, marched on a side
// stream while backward runs (train_step.TrainStep.split_labels) — asks for one or two waves per CU and takes proportionally longer.  Same labels.
int dreg_surface_visibility_multi_waves(const void* descs_dev, int n, long total_rays, int max_waves, void* stream)
{
    if (n <= 0 || total_rays <= 0) return DREG_OK;
    if (!descs_dev || max_waves < 0) return DREG_EINVAL;
    long waves = (total_rays + 63) / 64;
    if (waves > g_vis_waves) waves = g_vis_waves;
    if (max_waves > 0 && waves > max_waves) waves = max_waves;
    hipLaunchKernelGGL(surface_visibility_multi_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, (const VisArgs*)descs_dev, n);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}
int dreg_surface_visibility_multi(const void* descs_dev, int n, long total_rays, void* stream)
{
    return dreg_surface_visibility_multi_waves(descs_dev, n, total_rays, 0, stream);
}
// The same labels from the persistent kernel (lanes refilled from a ray queue; rays of points that already carry the label are not
// marched).  queue: 8 bytes of device memory the CALLER has zeroed on this stream (the ray counter).
#ifdef DREG_PROBE
void dreg_visibility_set_waves(int n) { g_vis_waves = n > 0 ? n : 4096; }
// Test hook: passes of the persistent kernels' march loop per wave (0 = the default 2^22, never reached by a real extraction).  A launch
// that hits the bound sets bit 63 of its ray counter(s) — the caller's `queue` words — instead of silently leaving points unlabelled.
int dreg_visibility_set_pass_bound(long passes)
{
    const long v = passes > 0 ? passes : (1L << 22);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pass_bound), &v, sizeof(v));
}
#endif
int dreg_surface_visibility_queue(const float* cams, const float* pts, const uint8_t* binary, int* label,
                                  const void* table, const void* w1, const void* w2,
                                  const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                  const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                                  int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                                  float alpha_thre, void* queue, const uint32_t* coarse_bits, void* stream)
{
    if ((long)Nc * Np == 0) return DREG_OK;
    if (!queue) return DREG_EINVAL;
    VisArgs a;
    vis_fill(a, cams, pts, binary, label, table, w1, w2, offset, size, res, scale, hashed, roi_aabb, scene_aabb, model_aabb, rx, ry, rz, Nc, Np,
             render_step_size, cut_off, early_stop_eps, alpha_thre, queue, coarse_bits);
    const long nrays = (long)Nc * Np;
    long waves = (nrays + 63) / 64;
    if (waves > g_vis_waves) waves = g_vis_waves;
    hipLaunchKernelGGL(surface_visibility_persistent_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, a, (unsigned long long*)queue);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

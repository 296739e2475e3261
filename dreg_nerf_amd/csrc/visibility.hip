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

extern "C" {

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
    const long nrays = (long)Nc * Np;
    hipLaunchKernelGGL(surface_visibility_kernel, dim3((unsigned)((nrays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    DREG_LAUNCH_CHECK();
    return DREG_OK;
}

}  // extern "C"

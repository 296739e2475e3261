"""ORACLE (test infrastructure, not product): CPU restatement of the surface-field visibility labels (row N1).

PARITY UNPINNED: the marching lives in nerfacc 0.3.5 (pinned in scripts/env/install.sh:23 but absent here) and the density in
tiny-cuda-nn.  Restated from the reference's call sites — conerf/utils/nerfacc_utils.py:168-220 (t_min from ray_aabb_intersect,
caller-supplied t_max = |p - o|, alphas = 1 - exp(-sigma * (t_end - t_start)), transmittance = exclusive cumprod, visibility =
T >= early_stop_eps [& alpha >= alpha_thre]), conerf/loss/confidence_loss.py:137-156 (max of alpha*T per ray, >= cut_off, max over
cameras) — with nerfacc's published marching rule: constant step dt, samples at the lattice midpoints t_min + (n + 1/2) dt whose
cell of the binary grid is occupied (skipping empty cells does not change which lattice points are sampled)."""
import numpy as np
import torch

from . import ngp_oracle as N


def surface_visibility(cams, pts, binary, roi_aabb, scene_aabb, model_aabb, mlp_base_params, dt, cut_off=0.5,
                       early_stop_eps=1e-4, alpha_thre=0.0):
    """cams [Nc,3], pts [Np,3] torch fp32; binary bool [rx,ry,rz].  Returns (labels int [Np], best float [Nc,Np])."""
    rx, ry, rz = binary.shape
    res = torch.tensor([rx, ry, rz], dtype=torch.float32)
    roi_lo, roi_ext = roi_aabb[:3], roi_aabb[3:] - roi_aabb[:3]
    Nc, Np = cams.shape[0], pts.shape[0]
    best = torch.zeros(Nc, Np)
    for c in range(Nc):
        o = cams[c]
        d = pts - o
        tmax = d.norm(dim=-1)
        d = d / tmax[:, None]
        inv = 1.0 / d
        t0 = (scene_aabb[:3] - o) * inv
        t1 = (scene_aabb[3:] - o) * inv
        near = torch.minimum(t0, t1).max(dim=-1).values
        far = torch.maximum(t0, t1).min(dim=-1).values
        hit = (near <= far) & (far > 0)
        tmin = near.clamp(min=0.0)
        nmax = int(torch.ceil(((tmax - tmin) / dt).max()).item()) + 1
        n = torch.arange(nmax, dtype=torch.float32)
        tm = tmin[:, None] + (n[None, :] + 0.5) * np.float32(dt)          # [Np, nmax]
        valid = (tm < tmax[:, None]) & hit[:, None]
        x = o + tm[..., None] * d[:, None, :]
        u = (x - roi_lo) / roi_ext
        inside = ((u >= 0) & (u <= 1)).all(-1)
        ci = torch.floor(u * res).long()
        ci = torch.minimum(torch.maximum(ci, torch.zeros(3, dtype=torch.long)), torch.tensor([rx - 1, ry - 1, rz - 1]))
        occ = binary[ci[..., 0], ci[..., 1], ci[..., 2]] & inside & valid
        idx = torch.nonzero(occ)
        sigma = torch.zeros(Np, nmax)
        if idx.shape[0] > 0:
            dens, _ = N.query_density(x[idx[:, 0], idx[:, 1]], model_aabb, mlp_base_params)
            sigma[idx[:, 0], idx[:, 1]] = dens
        alpha = (1.0 - torch.exp(-sigma * np.float32(dt))) * occ
        T = torch.cumprod(torch.cat([torch.ones(Np, 1), 1.0 - alpha[:, :-1]], dim=1), dim=1)
        vis = occ & (T >= early_stop_eps)
        if alpha_thre > 0:
            vis = vis & (alpha >= alpha_thre)
        best[c] = torch.where(vis, alpha * T, torch.zeros(())).max(dim=1).values
    return (best >= cut_off).any(dim=0).int(), best

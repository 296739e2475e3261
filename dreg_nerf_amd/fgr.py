"""Fast Global Registration baseline (row N4 of SURVEY.md §8f) without open3d.

The reference compares RegTR's pose with open3d's FGR on the two blocks' voxel point clouds and writes
``fgr_metrics_{split}.json`` (eval_nerf_regtr.py:303-311, 261-273; conerf/geometry/global_registration.py:21-116:
voxel_down_sample(0.05) -> estimate_normals(radius 0.1, max_nn 30) -> compute_fpfh_feature(radius 0.25, max_nn 100) ->
registration_fgr_based_on_feature_matching(maximum_correspondence_distance = 0.5)).  open3d is not in this image, so this is a
restatement of the PUBLISHED algorithms — FPFH (Rusu et al., ICRA 2009) and FGR (Zhou et al., ECCV 2016), with open3d's default
options — in plain torch: **parity unpinned** (no open3d output to compare with; the tests check that a known rigid motion is
recovered).  It is a baseline for the metrics file, not part of the hot path: dense O(N^2) neighbour search on a few thousand
down-sampled points, on whatever device the points live."""
import math
import time
from typing import Tuple

import torch


def voxel_down_sample(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """Mean of the points of every occupied voxel (open3d PointCloud.voxel_down_sample)."""
    lo = points.min(dim=0).values - 0.5 * voxel_size
    key = torch.floor((points - lo) / voxel_size).long()
    dims = key.max(dim=0).values + 1
    flat = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    uniq, inv = torch.unique(flat, return_inverse=True)
    out = torch.zeros(uniq.numel(), 3, dtype=points.dtype, device=points.device).index_add_(0, inv, points)
    cnt = torch.zeros(uniq.numel(), dtype=points.dtype, device=points.device).index_add_(0, inv, torch.ones_like(flat, dtype=points.dtype))
    return out / cnt[:, None]


def _neighbours(points: torch.Tensor, radius: float, max_nn: int):
    """Hybrid search (radius AND at most max_nn nearest, self included): indices [N,K], squared distances [N,K], valid [N,K]."""
    d2 = torch.cdist(points, points).square()
    k = min(max_nn, points.shape[0])
    dist, idx = torch.topk(d2, k, dim=1, largest=False)
    return idx, dist, dist <= radius * radius


def estimate_normals(points: torch.Tensor, radius: float, max_nn: int = 30) -> torch.Tensor:
    """Eigenvector of the smallest eigenvalue of the neighbourhood covariance; oriented away from the cloud's centroid so that
    the two clouds of a pair (related by a rigid motion) get consistent signs."""
    idx, _, ok = _neighbours(points, radius, max_nn)
    nb = points[idx]                                           # [N,K,3]
    w = ok.to(points.dtype)[..., None]
    n = w.sum(dim=1).clamp_min(1.0)
    mean = (nb * w).sum(dim=1) / n
    c = (nb - mean[:, None]) * w
    cov = c.transpose(1, 2) @ c / n[..., None]
    _, vec = torch.linalg.eigh(cov.double())
    nrm = vec[..., 0].to(points.dtype)
    out = points - points.mean(dim=0)
    sign = torch.where((nrm * out).sum(dim=1) < 0, -1.0, 1.0).to(points.dtype)
    return nrm * sign[:, None]


def compute_fpfh(points: torch.Tensor, normals: torch.Tensor, radius: float, max_nn: int = 100) -> torch.Tensor:
    """33-bin FPFH [N,33]: SPFH histograms of (f3 = atan2(w.n2, u.n2), f1 = v.n2, f2 = u.d) in 11 bins each, then the
    1/d^2-weighted sum of the neighbours' SPFH, each third normalised to 100, plus the point's own SPFH (open3d's layout)."""
    N = points.shape[0]
    idx, d2, ok = _neighbours(points, radius, max_nn)
    ok = ok & (d2 > 0)                                         # the point itself is not its own neighbour
    p1, n1 = points[:, None], normals[:, None]
    p2, n2 = points[idx], normals[idx]
    dp = p2 - p1
    dist = d2.clamp_min(1e-20).sqrt()
    # the point whose normal makes the smaller angle with the connecting line is the frame's origin (PCL / open3d rule)
    a1 = (n1 * dp).sum(-1) / dist
    a2 = -(n2 * dp).sum(-1) / dist
    swap = a1.abs().acos() > a2.abs().acos()
    u = torch.where(swap[..., None], n2, n1.expand_as(n2))
    nt = torch.where(swap[..., None], n1.expand_as(n2), n2)
    d = torch.where(swap[..., None], -dp, dp)
    f2 = (u * d).sum(-1) / dist
    v = torch.cross(d, u, dim=-1)
    vn = v.norm(dim=-1, keepdim=True)
    ok = ok & (vn[..., 0] > 0)
    v = v / vn.clamp_min(1e-20)
    w = torch.cross(u, v, dim=-1)
    f1 = (v * nt).sum(-1)
    f3 = torch.atan2((w * nt).sum(-1), (u * nt).sum(-1))
    b3 = ((f3 + math.pi) / (2 * math.pi) * 11).floor().clamp(0, 10).long()
    b1 = ((f1 + 1.0) * 0.5 * 11).floor().clamp(0, 10).long()
    b2 = ((f2 + 1.0) * 0.5 * 11).floor().clamp(0, 10).long()
    k = ok.sum(dim=1).clamp_min(1).to(points.dtype)
    inc = (100.0 / k)[:, None] * ok.to(points.dtype)
    spfh = torch.zeros(N, 33, dtype=points.dtype, device=points.device)
    rows = torch.arange(N, device=points.device)[:, None].expand_as(b1)
    for off, b in ((0, b3), (11, b1), (22, b2)):
        spfh.index_put_((rows.reshape(-1), (b + off).reshape(-1)), inc.reshape(-1), accumulate=True)
    wgt = torch.where(ok, 1.0 / d2.clamp_min(1e-20), torch.zeros_like(d2))
    acc = (spfh[idx] * wgt[..., None]).sum(dim=1)              # [N,33]
    thirds = acc.view(N, 3, 11)
    s = thirds.sum(dim=2, keepdim=True)
    thirds = torch.where(s > 0, thirds * (100.0 / s.clamp_min(1e-20)), thirds)
    return thirds.reshape(N, 33) + spfh


def _nearest(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.cdist(a, b).argmin(dim=1)


def _correspondences(fs, ft, ps, pt, tuple_scale: float, max_tuples: int, gen: torch.Generator):
    """Cross-checked feature matches, then FGR's tuple test (random triplets whose three edge-length ratios all lie within
    [tuple_scale, 1/tuple_scale])."""
    ij, ji = _nearest(fs, ft), _nearest(ft, fs)
    i = torch.arange(fs.shape[0], device=fs.device)
    keep = ji[ij] == i
    src, tgt = i[keep], ij[keep]
    if src.numel() < 10:                                       # too few mutual matches: the union of both directions (open3d does the same)
        j = torch.arange(ft.shape[0], device=fs.device)
        src, tgt = torch.cat([i, ji]), torch.cat([ij, j])
    n = src.numel()
    if n >= 3 and max_tuples > 0:
        t = torch.randint(0, n, (max_tuples * 100, 3), generator=gen, device="cpu").to(fs.device)
        a, b = ps[src[t]], pt[tgt[t]]                          # [T,3,3]
        ok = torch.ones(t.shape[0], dtype=torch.bool, device=fs.device)
        for x, y in ((0, 1), (0, 2), (1, 2)):
            ls, lt = (a[:, x] - a[:, y]).norm(dim=1), (b[:, x] - b[:, y]).norm(dim=1)
            ok &= (ls * tuple_scale < lt) & (lt * tuple_scale < ls)
        sel = t[ok][:max_tuples].reshape(-1)
        if sel.numel() >= 3:
            src, tgt = src[sel], tgt[sel]
    return src, tgt


def _skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def fast_global_registration(ps: torch.Tensor, pt: torch.Tensor, fs: torch.Tensor, ft: torch.Tensor,
                             maximum_correspondence_distance: float = 0.5, division_factor: float = 1.4, iteration_number: int = 64,
                             tuple_scale: float = 0.95, maximum_tuple_count: int = 1000, decrease_mu: bool = True, seed: int = 0) -> torch.Tensor:
    """4x4 transformation taking the source points onto the target points (Zhou et al. 2016 with open3d's option defaults)."""
    dt, dev = torch.float64, ps.device
    ps, pt = ps.to(dt), pt.to(dt)
    gen = torch.Generator().manual_seed(seed)
    src, tgt = _correspondences(fs, ft, ps, pt, tuple_scale, maximum_tuple_count, gen)
    # normalisation: centre both clouds, one common scale (the larger radius)
    ms, mt = ps.mean(dim=0), pt.mean(dim=0)
    scale = float(max((ps - ms).norm(dim=1).max(), (pt - mt).norm(dim=1).max()))
    a, b = (ps[src] - ms) / scale, (pt[tgt] - mt) / scale
    T = torch.eye(4, dtype=dt, device=dev)
    mu = 1.0
    limit = maximum_correspondence_distance / scale
    for it in range(iteration_number):
        q = a @ T[:3, :3].T + T[:3, 3]
        r = q - b
        l = (mu / (mu + (r * r).sum(dim=1))).square()          # line-process weights
        # r(xi) ~ r + [-[q]x | I] xi  ->  6x6 normal equations
        J = torch.cat([-_skew(q), torch.eye(3, dtype=dt, device=dev).expand(q.shape[0], 3, 3)], dim=2)   # [M,3,6]
        JtJ = (J.transpose(1, 2) @ (J * l[:, None, None])).sum(dim=0)
        Jtr = (J.transpose(1, 2) @ (r * l[:, None])[..., None]).sum(dim=0)[:, 0]
        xi = -torch.linalg.solve(JtJ + 1e-12 * torch.eye(6, dtype=dt, device=dev), Jtr)
        th = xi[:3].norm()
        K = _skew(xi[:3] / th.clamp_min(1e-20))
        R = torch.eye(3, dtype=dt, device=dev) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
        D = torch.eye(4, dtype=dt, device=dev)
        D[:3, :3], D[:3, 3] = R, xi[3:]
        T = D @ T
        if decrease_mu and it % 4 == 0 and mu > limit:
            mu /= division_factor
    # undo the normalisation:  x_t = scale * (R (x_s - ms)/scale + t) + mt
    out = torch.eye(4, dtype=dt, device=dev)
    out[:3, :3] = T[:3, :3]
    out[:3, 3] = scale * T[:3, 3] + mt - T[:3, :3] @ ms
    return out


def preprocess_point_cloud(points: torch.Tensor, voxel_size: float = 0.05):
    """(down-sampled points, FPFH) with the reference's radii (global_registration.py:21-35)."""
    down = voxel_down_sample(points.float(), voxel_size)
    normals = estimate_normals(down, voxel_size * 2, 30)
    return down, compute_fpfh(down, normals, voxel_size * 5, 100)


def run_registration(source_points: torch.Tensor, target_points: torch.Tensor, voxel_size: float = 0.05) -> Tuple[torch.Tensor, float]:
    """The reference's run_registration(..., method='fast') on point tensors [N,3] instead of PLY paths (the PLY it reads is the
    block's voxel point cloud = voxel_grid[...,:3] at voxel_mask): returns (4x4 float64 transformation, seconds of the FGR call)."""
    sd, sf = preprocess_point_cloud(source_points, voxel_size)
    td, tf = preprocess_point_cloud(target_points, voxel_size)
    if source_points.is_cuda:
        torch.cuda.synchronize()
    t0 = time.time()
    T = fast_global_registration(sd, td, sf, tf, maximum_correspondence_distance=voxel_size * 10)
    if source_points.is_cuda:
        torch.cuda.synchronize()
    return T, time.time() - t0

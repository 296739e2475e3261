"""Canonical synthetic inputs ("shell-R").

A NeRF block's ``voxel_grid.pt`` is float32 ``[X,Y,Z,7]`` = (world xyz, mean rgb, alpha) at
occupied voxels and zero elsewhere, with ``voxel_mask.pt`` the ascending int64 flat indices
(reference writer: eval_ngp_nerf.py:383-412; loader: conerf/datasets/register/dataset.py:244-248).
"""
from typing import Tuple

import torch


def shell_grid(res: int, seed: int, r0: float, r1: float, half: float = 1.5,
               pose: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (voxel_grid [res,res,res,7] fp32, voxel_mask int64).  Occupied iff
    r0 < |centre| < r1; rgb then alpha drawn from manual_seed(seed).  ``pose`` ([3|4,4]) is
    applied to the stored xyz (the block's own frame differs from the world frame by it)."""
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * (2 * half) - half
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    ctr = torch.stack([X, Y, Z], dim=-1)
    rad = ctr.norm(dim=-1)
    occ = (rad > r0) & (rad < r1)
    mask = torch.nonzero(occ.flatten())[:, 0]
    n = mask.shape[0]
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(n, 3, generator=g)
    alpha = torch.rand(n, generator=g)
    xyz = ctr.reshape(-1, 3)[mask]
    if pose is not None:
        xyz = xyz @ pose[:3, :3].T + pose[:3, 3]
    grid = torch.zeros(res * res * res, 7, dtype=torch.float32)
    grid[mask] = torch.cat([xyz, rgb, alpha[:, None]], dim=1)
    return grid.view(res, res, res, 7), mask


def shell_radii(res: int) -> Tuple[float, float]:
    """Config 1 (32^3): (0.8, 0.9) -> 1040 voxels/side.  128^3: (0.8, 0.83) -> 19,176."""
    return (0.8, 0.9) if res <= 32 else (0.8, 0.83)


def fixed_pose(variant: int = 0) -> torch.Tensor:
    """A fixed SE(3) (rotation ~17 deg about a skew axis, small translation) for RRE/RTE checks.  variant 1: another axis, ~26 deg,
    another translation (the second pinned training step: tests/golden/train64_wc_b.npz)."""
    ax = torch.tensor([0.3, -0.5, 0.8] if variant == 0 else [-0.6, 0.2, 0.35])
    ax = ax / ax.norm()
    ang = 0.3 if variant == 0 else -0.45
    K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = torch.eye(3) + torch.sin(torch.tensor(ang)) * K + (1 - torch.cos(torch.tensor(ang))) * (K @ K)
    T = torch.eye(4)
    T[:3, :3] = R
    T[:3, 3] = torch.tensor([0.05, -0.03, 0.02] if variant == 0 else [-0.04, 0.06, 0.03])
    return T


def shell_pair(res: int, seed_src: int = 1, seed_tgt: int = 2, pose: torch.Tensor = None) -> dict:
    """One registration sample in the layout NeRFRegDataset.__getitem__ returns
    (dataset.py:221-275): grids permuted to [1,7,Z,X,Y], masks int64, pose [1,4,4]."""
    r0, r1 = shell_radii(res)
    if pose is None:
        pose = torch.eye(4)
    gs, ms = shell_grid(res, seed_src, r0, r1)
    gt, mt = shell_grid(res, seed_tgt, r0, r1, pose=pose)
    return {
        "src_xyz_rgba": gs.permute(3, 2, 0, 1).unsqueeze(0).contiguous(),
        "tgt_xyz_rgba": gt.permute(3, 2, 0, 1).unsqueeze(0).contiguous(),
        "src_mask": ms, "tgt_mask": mt,
        "pose": pose[None].clone(),
        "src_nerf_path": "", "tgt_nerf_path": "",
    }


def synthetic_overlap_gt(kp: torch.Tensor, nl: int = 6) -> torch.Tensor:
    """Deterministic stand-in for the ray-marched visibility labels: a half-space test
    1[x + 0.31 y - 0.17 z > 0.0123] broadcast to [nl, N, 1].  (A radial threshold inside the shell would
    flip labels under 1e-7 perturbations of the key points; the plane keeps the labelling well conditioned.)"""
    s = kp[..., 0] + 0.31 * kp[..., 1] - 0.17 * kp[..., 2]
    return (s > 0.0123).to(kp.dtype)[None, :, None].expand(nl, -1, -1).contiguous()

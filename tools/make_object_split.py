#!/usr/bin/env python3
"""A stand-in registration split with LEARNABLE content, in the reference's directory layout (the Objaverse blocks and the trained checkpoint are
external downloads: SURVEY.md 8(d) 'Data availability').

Every scene is a random asymmetric object (a union of ellipsoids) seen by two NeRF blocks, each in its own frame: block k's frame is T_k applied to
the world frame (world_frame_transforms.json holds T_k, the dataset's ground truth is T_t T_s^-1: conerf/datasets/register/dataset.py:100-134,221-275).
A block is a NeRF checkpoint in the reference's format (train_ngp_nerf.py:187-209) whose OCCUPANCY GRID is the object's surface shell in the block's
frame and whose NGP weights are generated (opaque surfaces, constant colour: no NeRF is trained here — out of scope); its cameras look at the object
from one side, so the two blocks see overlapping but different parts of the surface.  Grid extraction (eval_ngp_nerf.py), training with labels
ray-marched from the blocks (train_nerf_regtr.py) and evaluation (eval_nerf_regtr.py) then run on it exactly as on real data.

  python tools/make_object_split.py --root /dev/shm/objsplit --train 128 --test 16
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AABB = [-1.5] * 3 + [1.5] * 3


def random_object(gen):
    """Ellipsoids (centre, rotation, radii): one body and four to six distinct limbs, all inside the unit ball."""
    n = 5 + int(torch.randint(0, 3, (1,), generator=gen))
    parts = []
    for i in range(n):
        c = (torch.rand(3, generator=gen) - 0.5) * (0.5 if i == 0 else 1.1)
        r = (0.35 + 0.25 * torch.rand(3, generator=gen)) if i == 0 else (0.12 + 0.3 * torch.rand(3, generator=gen))
        q = torch.randn(4, generator=gen)
        q = q / q.norm()
        w, x, y, z = q.tolist()
        R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        parts.append((c, R, r))
    return parts


def object_distance(p, parts):
    """Approximate signed distance of points [N,3] to the union of the ellipsoids (negative inside)."""
    d = None
    for c, R, r in parts:
        q = (p - c) @ R                       # into the ellipsoid's axes
        k = (q / r).norm(dim=-1)
        di = (k - 1.0) * r.min()
        d = di if d is None else torch.minimum(d, di)
    return d


def small_se3(std, gen):
    from dreg_nerf_amd.dataset import _small_se3
    return _small_se3(std, gen)


def write_block(path, parts, T, cams_dir, base_params, color_params, res, ncam, gen, thick=(-0.05, 0.02)):
    from dreg_nerf_amd import ngp
    dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")       # (plain torch arithmetic of the GENERATOR, not of the product path)
    c = (torch.arange(res, dtype=torch.float32, device=dev) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    local = torch.stack([X, Y, Z], -1).reshape(-1, 3)
    Ti = torch.linalg.inv(T).to(dev)
    world = local @ Ti[:3, :3].T + Ti[:3, 3]
    d = object_distance(world, [(a.to(dev), b_.to(dev), c_.to(dev)) for a, b_, c_ in parts]).view(res, res, res)
    binary = ((d > thick[0]) & (d < thick[1])).cpu()
    # cameras on a sphere of radius 3 around the block, within ~100 degrees of this block's viewing side
    v = torch.nn.functional.normalize(torch.randn(4 * ncam, 3, generator=gen), dim=-1)
    v = v[(v @ cams_dir) > -0.2][:ncam]
    poses = torch.eye(4)[None].repeat(v.shape[0], 1, 1)
    poses[:, :3, 3] = v * 3.0
    occ_state = {"_roi_aabb": torch.tensor(AABB), "resolution": torch.tensor([res] * 3, dtype=torch.int32), "occs": torch.zeros(res ** 3), "_binary": binary}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"step": 1, "model": {"aabb": torch.tensor(AABB), "mlp_base.params": base_params, "color_mlp.params": color_params}, "occupancy_grid": occ_state,
                "aabb": AABB, "unbounded": False, "near_plane": None, "far_plane": None, "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB,
                "render_step_size": 3 * 3 ** 0.5 / 1024, "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": 0}, path)
    return int(binary.sum())


def make_split(root, json_dir, n_train, n_test, res=128, ncam=24, pose_std=0.15, seed=0, dataset="objaverse"):
    from dreg_nerf_amd import ngp
    gen = torch.Generator().manual_seed(seed)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():                # ONE set of generated NGP weights for all blocks: the geometry lives in the occupancy grids
        f.mlp_base.params[:3072] = torch.randn(3072, generator=gen) * 3.0
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=gen)
        f.color_mlp.params[6144:] = 0.0                                          # constant colour (the two blocks' hash grids share no texture)
    base, col = f.mlp_base.params.detach().clone(), f.color_mlp.params.detach().clone()
    os.makedirs(json_dir, exist_ok=True)
    names = {"train": [f"obj_train_{i:04d}" for i in range(n_train)], "test": [f"obj_test_{i:04d}" for i in range(n_test)]}
    ids = {sp: [f"uid_{n}" for n in ns] for sp, ns in names.items()}
    json.dump({dataset: ids}, open(os.path.join(json_dir, "objaverse.json"), "w"))
    json.dump({f"uid_{n}": n for ns in names.values() for n in ns}, open(os.path.join(json_dir, "obj_id_names.json"), "w"))
    # a second set of split files whose 'test' split is the first 16 TRAINING scenes: the in-distribution evaluation of a trained checkpoint
    sub = json_dir.rstrip("/") + "_trainsub"
    os.makedirs(sub, exist_ok=True)
    json.dump({dataset: {"train": [], "test": ids["train"][:16]}}, open(os.path.join(sub, "objaverse.json"), "w"))
    json.dump({f"uid_{n}": n for ns in names.values() for n in ns}, open(os.path.join(sub, "obj_id_names.json"), "w"))
    occ = []
    for sp, ns in names.items():
        for name in ns:
            parts = random_object(gen)
            side = torch.nn.functional.normalize(torch.randn(3, generator=gen), dim=0)
            ortho = torch.nn.functional.normalize(torch.linalg.cross(side, torch.nn.functional.normalize(torch.randn(3, generator=gen), dim=0)), dim=0)
            dirs = [side, torch.nn.functional.normalize(0.3 * side + ortho, dim=0)]          # the two blocks look from ~70 degrees apart
            tf = {}
            for k in range(2):
                T = small_se3(pose_std, gen)
                tf[str(k)] = T.tolist()
                occ.append(write_block(os.path.join(root, dataset, "nerf_models", name, f"block_{k}", "model.pth"), parts, T, T[:3, :3] @ dirs[k], base, col, res, ncam, gen))
            d = os.path.join(root, dataset, "images", name)
            os.makedirs(d, exist_ok=True)
            json.dump(tf, open(os.path.join(d, "world_frame_transforms.json"), "w"))
    return names, occ


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--json_dir", default="")
    ap.add_argument("--train", type=int, default=128)
    ap.add_argument("--test", type=int, default=16)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--cams", type=int, default=24)
    ap.add_argument("--pose_std", type=float, default=0.15)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    names, occ = make_split(a.root, a.json_dir or os.path.join(a.root, "json"), a.train, a.test, a.res, a.cams, a.pose_std, a.seed)
    print(f"{len(names['train'])} train + {len(names['test'])} test scenes under {a.root}; occupied cells per block: min {min(occ)} mean {sum(occ) / len(occ):.0f} max {max(occ)}")

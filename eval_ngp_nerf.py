#!/usr/bin/env python3
"""Voxel-grid extraction from trained NeRF blocks on MI355X — drop-in for the `sample_points` part of the reference's
eval_ngp_nerf.py (:336-451, `--multi_blocks`): for every <root>/<dataset>/nerf_models/<scene>/block_k/model.pth write
voxel_grid.pt / voxel_mask.pt next to it.  Blocks are independent: ranks take blocks round-robin (replicas only)."""
import glob
import os

import torch

from dreg_nerf_amd import ngp
from dreg_nerf_amd.config import config_parser


@torch.no_grad()
def extract_block(ckpt_path: str, dev, density_thre: float = 0.7):
    ngp.install_pickle_shims()
    state = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    field = ngp.NGPradianceField(state["aabb"], unbounded=bool(state.get("unbounded", False)))
    field.load_state_dict(state["model"], strict=False)
    field = field.to(dev).eval()
    res = int(state.get("grid_resolution", 128))
    occ = state["occupancy_grid"]
    binary = occ["_binary"] if "_binary" in occ else occ["binary"]
    sg = ngp.SampleGrid(state["aabb"], res, state.get("contraction_type", ngp.ContractionType.AABB)).to(dev)
    sg.set_binary_fields(binary.to(dev).view(res, res, res))
    world, rgb, alpha, idx, dmask, smask = sg.query_radiance_and_density_from_camera(field, None, state, dev, density_thre)
    grid, mask = ngp.build_voxel_grid(world, rgb, alpha, idx, dmask & smask, res)
    ngp.save_voxel_grid(os.path.dirname(ckpt_path), grid, mask)
    return int(mask.shape[0])


def main():
    cfg = config_parser()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", cfg.local_rank)))
    torch.cuda.set_device(dev)
    pattern = os.path.join(cfg.root_dir, cfg.dataset, "nerf_models", cfg.scene or "*", "block_*", "model.pth")
    for i, path in enumerate(sorted(glob.glob(pattern))):
        if i % world == rank:
            n = extract_block(path, dev)
            print(f"[rank {rank}] {path}: {n} voxels kept", flush=True)


if __name__ == "__main__":
    main()

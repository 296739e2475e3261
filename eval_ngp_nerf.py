#!/usr/bin/env python3
"""Voxel-grid extraction from trained NeRF blocks on MI355X — drop-in for the `sample_points` part of the reference's
eval_ngp_nerf.py (:336-451, `--multi_blocks`): for every <root>/<dataset>/nerf_models/<scene>/block_k/model.pth write
voxel_grid.pt / voxel_mask.pt / voxel_point_cloud.ply and their density_voxel_* twins next to it.  Blocks are independent: ranks take blocks round-robin (replicas only)."""
import glob
import os

import torch

from dreg_nerf_amd import ngp
from dreg_nerf_amd.checkpoint import CheckPointManager
from dreg_nerf_amd.config import config_parser


@torch.no_grad()
def extract_block(ckpt_path: str, dev, density_thre: float = 0.7):
    # the reference's two-pass load (eval_ngp_nerf.py:63-115): meta data, then the modules constructed from it
    meta = {k: None for k in ("aabb", "unbounded", "grid_resolution", "contraction_type",
                              "render_step_size", "alpha_thre", "cone_angle", "camera_poses")}
    mgr = CheckPointManager(verbose=False)
    mgr.load_no_config(ckpt_path, meta_data=meta, map_location="cpu")
    field = ngp.NGPradianceField(meta["aabb"], unbounded=bool(meta["unbounded"]))
    occ = ngp.OccupancyGrid(meta["aabb"], meta["grid_resolution"], meta["contraction_type"])
    mgr.load_no_config(ckpt_path, models={"model": field, "occupancy_grid": occ}, map_location="cpu")
    field = field.to(dev).eval()
    sg = ngp.SampleGrid(meta["aabb"], meta["grid_resolution"], meta["contraction_type"]).to(dev)
    sg.set_binary_fields(occ.binary.to(dev))
    res = int(sg.resolution[0])
    state = meta
    world, rgb, alpha, idx, dmask, smask = sg.query_radiance_and_density_from_camera(field, None, state, dev, density_thre)
    out_dir = os.path.dirname(ckpt_path)
    # the density-field twins first (eval_ngp_nerf.py:350-381), then the surface AND density set the registration dataset reads (:383-412)
    dgrid, dmask_idx = ngp.build_voxel_grid(world, rgb, alpha, idx, dmask, res)
    ngp.save_voxel_grid(out_dir, dgrid, dmask_idx, prefix="density_voxel", points=world[dmask], colors=rgb[dmask])
    keep = dmask & smask
    grid, mask = ngp.build_voxel_grid(world, rgb, alpha, idx, keep, res)
    ngp.save_voxel_grid(out_dir, grid, mask, points=world[keep], colors=rgb[keep])
    from dreg_nerf_amd import visibility
    visibility.OVERRUN.check(wait=True)      # a surface-label launch that hit its pass bound is an error of THIS block, raised before the next one
    return int(mask.shape[0])


def main():
    cfg = config_parser()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", cfg.local_rank)))
    torch.cuda.set_device(dev)
    pattern = os.path.join(cfg.root_dir, cfg.dataset, "nerf_models", cfg.scene or "*", "block_*", "model.pth")
    mine = [p for i, p in enumerate(sorted(glob.glob(pattern))) if i % world == rank]
    if os.environ.get("DREG_SERIAL_EXTRACT") == "1":       # the block-at-a-time form (what the reference does); same files, byte for byte
        for path in mine:
            print(f"[rank {rank}] {path}: {extract_block(path, dev)} voxels kept", flush=True)
        return
    # checkpoint reads, queries and file writes of different blocks overlapped (dreg_nerf_amd/eval_pipeline.py)
    from dreg_nerf_amd.eval_pipeline import ExtractionPipeline
    with ExtractionPipeline(dev) as pipe:
        done = []
        for ex in pipe.run(mine):
            done.append(ex)
            while len(done) > 4:            # report a few blocks behind the GPU: reading a count back waits for that block's query only
                e = done.pop(0)
                print(f"[rank {rank}] {e.path}: {e.kept()} voxels kept", flush=True)
        for e in done:
            print(f"[rank {rank}] {e.path}: {e.kept()} voxels kept", flush=True)


if __name__ == "__main__":
    main()

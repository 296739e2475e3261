#!/usr/bin/env python3
"""Evaluate NeRF registration on MI355X — drop-in for the metric part of the reference's eval_nerf_regtr.py
(:224-301; with --dump_outputs also its per-scene transformation_est.json and PLY point clouds, :313-438): per-scene RRE/RTE + forward time (with a device sync, unlike the reference — quirk Q13) written to
<root>/eval/<expname>/<dataset>/metrics_<split>.json with the reference's schema.  Scenes are sharded over ranks
when launched with torch.distributed.run (replicas only, results gathered on rank 0)."""
import json
import os
import time

import torch
import torch.distributed as dist

from dreg_nerf_amd import fgr, vis_dump
from dreg_nerf_amd import losses as LS
from dreg_nerf_amd.checkpoint import CheckPointManager
from dreg_nerf_amd.config import config_parser
from dreg_nerf_amd.dataset import NeRFRegDataset, SyntheticRegDataset
from dreg_nerf_amd.regtr import NeRFRegTr


def _points(data, side):
    """World coordinates of the occupied voxels of one block = the voxel point cloud the reference reads from its PLY."""
    if side + "_sparse" in data:
        return data[side + "_sparse"].vals[:, :3].float()
    g = data[side + "_xyz_rgba"]
    g = g.squeeze(0) if g.dim() == 6 else g
    m = data[side + "_mask"]
    m = m.squeeze(0) if m.dim() == 2 else m
    return g[:, :3].permute(0, 3, 4, 2, 1).reshape(-1, 3)[m].float()      # same flattening as nerf_regtr.py:144-147


def main():
    cfg = config_parser()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", cfg.local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    split = "test"
    ds = SyntheticRegDataset(cfg.synthetic, cfg.synthetic_res, split) if cfg.synthetic > 0 else \
        NeRFRegDataset(cfg.root_dir, cfg.json_dir, cfg.dataset, split, sparse=True, device=dev)
    model = NeRFRegTr(cfg.position_embedding_type, cfg.position_embedding_dim, cfg.position_embedding_scaling,
                      cfg.num_downsample, precision=cfg.precision).to(dev).eval()
    ckpt_path = cfg.ckpt_path or os.path.join(cfg.root_dir, "out", cfg.expname, "model.pth")
    if CheckPointManager(verbose=rank == 0).load_no_config(ckpt_path, models={"model": model}, map_location=dev) == 0 and not os.path.exists(ckpt_path):
        print(f"[WARNING] no checkpoint at {ckpt_path}: evaluating random-init weights", flush=True)
    rows, fgr_rows = {}, {}
    with torch.no_grad():
        for i in range(rank, len(ds), world):
            data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ds[i].items()}
            torch.cuda.synchronize()
            t0 = time.time()
            pred = model(data)
            torch.cuda.synchronize()
            dt = time.time() - t0
            model.check_inputs()      # a grid with values outside its voxel_mask (the row-list stem would have dropped them) is an error of THIS scene
            err = LS.evaluate_camera_alignment(pred["pose"][-1], data["pose"])
            rows[data["scene"]] = {"R_mean": float(err["R_error_mean"]), "t_mean": float(err["t_error_mean"]),
                                   "R_med": float(err["R_error_med"]), "t_med": float(err["t_error_med"]), "time": dt}
            if cfg.dump_outputs:   # the reference's per-scene files (eval_nerf_regtr.py:313-321, 369-438; no videos / camera-pose dumps)
                cams = [None, None]
                sp, tp = data.get("src_nerf_path", ""), data.get("tgt_nerf_path", "")
                if sp and tp and os.path.exists(sp) and os.path.exists(tp):       # camera_poses of the two NeRF blocks (train_ngp_nerf.py:187-209)
                    from dreg_nerf_amd.visibility import load_block
                    cams = [load_block(q, dev)[2]["camera_poses"] for q in (sp, tp)]
                vis_dump.dump_scene_outputs(os.path.join(cfg.root_dir, "eval", cfg.expname, cfg.dataset or "synthetic", str(data["scene"])), pred, data["pose"],
                                            cams[0], cams[1])
            if cfg.fgr_baseline:   # the reference's baseline on the two voxel point clouds (global_registration.py:96-116)
                T, sec = fgr.run_registration(_points(data, "src"), _points(data, "tgt"))
                e = LS.evaluate_camera_alignment(T[None].float(), data["pose"])
                fgr_rows[data["scene"]] = {"R_mean": float(e["R_error_mean"]), "t_mean": float(e["t_error_mean"]),
                                           "R_med": float(e["R_error_med"]), "t_med": float(e["t_error_med"]), "time": sec}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = {k: v for g in gathered for k, v in g.items()}
        dist.all_gather_object(gathered, fgr_rows)
        fgr_rows = {k: v for g in gathered for k, v in g.items()}
    if rank == 0:
        out = dict(rows)
        out["R_mean"] = sum(r["R_mean"] for r in rows.values()) / max(len(rows), 1)
        out["t_mean"] = sum(r["t_mean"] for r in rows.values()) / max(len(rows), 1)
        d = os.path.join(cfg.root_dir, "eval", cfg.expname, cfg.dataset or "synthetic")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"metrics_{split}.json"), "w") as f:
            json.dump(out, f, indent=2)
        print(f"{len(rows)} scenes: R_mean={out['R_mean']:.3f} deg, t_mean={out['t_mean']:.4f} -> {d}/metrics_{split}.json", flush=True)
        if fgr_rows:
            fo = dict(fgr_rows)
            fo["R_mean"] = sum(r["R_mean"] for r in fgr_rows.values()) / len(fgr_rows)
            fo["t_mean"] = sum(r["t_mean"] for r in fgr_rows.values()) / len(fgr_rows)
            with open(os.path.join(d, f"fgr_metrics_{split}.json"), "w") as f:
                json.dump(fo, f, indent=4)
            print(f"FGR baseline: R_mean={fo['R_mean']:.3f} deg, t_mean={fo['t_mean']:.4f} -> {d}/fgr_metrics_{split}.json", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

from dreg_nerf_amd import eval_shard as ES
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


def _row(err, dt):
    return {"R_mean": float(err["R_error_mean"]), "t_mean": float(err["t_error_mean"]), "R_med": float(err["R_error_med"]), "t_med": float(err["t_error_med"]), "time": dt}


def init_distributed(local_rank: int):
    """One process per GPU under torch.distributed.run: RCCL ('nccl' on ROCm).  DREG_EVAL_BACKEND=gloo with DREG_EVAL_ONE_GPU=1 is the test hook bench.py
    has too — several ranks on the one GPU of a test box exercise the sharding and the gather (tests/test_hip_eval_pipeline.py); never a measurement."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("DREG_EVAL_BACKEND", "nccl")
    if os.environ.get("DREG_EVAL_ONE_GPU") == "1":
        local_rank = 0
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return local_rank


def main():
    cfg = config_parser()
    import random
    import numpy as np
    random.seed(cfg.seed)            # setup_seed(config.seed) of the reference (eval_nerf_regtr.py:462, conerf/utils/utils.py:21-26): the block order of
    np.random.seed(cfg.seed)         # every scene (quirk Q15) and, with --extract_grids, the extraction's jitter (quirk Q11) are reproducible
    torch.manual_seed(cfg.seed)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", cfg.local_rank))
    if world > 1:
        local_rank = init_distributed(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    split = "test"
    ds = SyntheticRegDataset(cfg.synthetic, cfg.synthetic_res, split) if cfg.synthetic > 0 else \
        NeRFRegDataset(cfg.root_dir, cfg.json_dir, cfg.dataset, split, sparse=True, device=dev,
                       require_grids=not cfg.extract_grids)
    model = NeRFRegTr(cfg.position_embedding_type, cfg.position_embedding_dim, cfg.position_embedding_scaling,
                      cfg.num_downsample, precision=cfg.precision).to(dev).eval()
    ckpt_path = cfg.ckpt_path or os.path.join(cfg.root_dir, "out", cfg.expname, "model.pth")
    if CheckPointManager(verbose=rank == 0).load_no_config(ckpt_path, models={"model": model}, map_location=dev) == 0 and not os.path.exists(ckpt_path):
        print(f"[WARNING] no checkpoint at {ckpt_path}: evaluating random-init weights", flush=True)
    rows, fgr_rows = {}, {}
    per_scene_extras = cfg.dump_outputs or cfg.fgr_baseline
    mine = ES.my_scenes(len(ds), rank, world)
    # every rank consumes the block-order draws of ALL scenes in scene order: a scene's source / target assignment is then the one-rank run's,
    # whatever the rank count — the gathered metrics file does not depend on the sharding (dreg_nerf_amd/eval_shard.py)
    order_of = ES.block_orders(ds) if cfg.synthetic == 0 else {}
    if cfg.extract_grids and cfg.synthetic == 0:
        # BASELINE.json configs[4] in one process: this rank's scenes go through grid extraction (the reference's eval_ngp_nerf.py files are written) and are
        # registered from the device-resident grids, extraction of later scenes overlapping registration of earlier ones (dreg_nerf_amd/eval_pipeline.py)
        from dreg_nerf_amd.eval_pipeline import extract_and_register
        scenes = []
        for i in mine:
            sm = ds.meta[i]
            ids = order_of[i]                        # which block is the source: as the dataset draws it (quirk Q15)
            s, t = sm["blocks"][ids[0]], sm["blocks"][ids[1]]
            scenes.append((sm["scene"], os.path.join(s["dir"], "model.pth"), os.path.join(t["dir"], "model.pth"), t["transform"] @ torch.linalg.inv(s["transform"])))
        got, tm = extract_and_register(scenes, model, dev, batch_pairs=max(cfg.eval_batch, 1))
        rows = {k: {q: v[q] for q in ("R_mean", "t_mean", "R_med", "t_med", "time")} for k, v in got.items()}
        if rank == 0:
            print(f"extracted {tm['blocks']} blocks ({tm['bytes_written'] / 1e9:.2f} GB of grid files) and registered {len(rows)} pairs", flush=True)
        mine = []
    with torch.no_grad():
        step = 1 if per_scene_extras else max(cfg.eval_batch, 1)
        for b0 in range(0, len(mine), step):
            # `eval_batch` pairs per call (NeRFRegTr.forward_batch; the reference evaluates one pair per call — --eval_batch 1): samples are drawn in scene order on
            # this thread, so the block order a scene gets (quirk Q15) does not depend on the batch size
            batch = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in (ds.get(i, block_order=order_of[i]) if i in order_of else ds[i]).items()}
                     for i in mine[b0:b0 + step]]
            torch.cuda.synchronize()
            t0 = time.time()
            preds = model.forward_batch(batch)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / len(batch)     # forward time per pair (with a device sync, unlike the reference — quirk Q13)
            model.check_inputs()      # a grid with values outside its voxel_mask (the row-list stem would have dropped them) is an error of THESE scenes
            for data, pred in zip(batch, preds):
                rows[data["scene"]] = _row(LS.evaluate_camera_alignment(pred["pose"][-1], data["pose"]), dt)
            if not per_scene_extras:
                continue
            data, pred = batch[0], preds[0]
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
    rows, fgr_rows = ES.gather_rows(rows, world), ES.gather_rows(fgr_rows, world)
    if rank == 0:
        out = ES.summary(rows)
        d = os.path.join(cfg.root_dir, "eval", cfg.expname, cfg.dataset or "synthetic")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"metrics_{split}.json"), "w") as f:
            json.dump(out, f, indent=2)
        print(f"{len(rows)} scenes: R_mean={out['R_mean']:.3f} deg, t_mean={out['t_mean']:.4f} -> {d}/metrics_{split}.json", flush=True)
        if fgr_rows:
            fo = ES.summary(fgr_rows)
            with open(os.path.join(d, f"fgr_metrics_{split}.json"), "w") as f:
                json.dump(fo, f, indent=4)
            print(f"FGR baseline: R_mean={fo['R_mean']:.3f} deg, t_mean={fo['t_mean']:.4f} -> {d}/fgr_metrics_{split}.json", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

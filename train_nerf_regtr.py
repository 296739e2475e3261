#!/usr/bin/env python3
"""Train the NeRF registration network on MI355X — drop-in for the reference's train_nerf_regtr.py
(flags: conerf/utils/config.py; loop: train_nerf_regtr.py:124-169; step: dreg_nerf_amd/train_step.py).

Single GPU:  python train_nerf_regtr.py --root_dir <root> --json_dir <json> --dataset objaverse --expname regtr
8 GPUs:      python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_nerf_regtr.py ... (plain DDP over RCCL)
No data:     python train_nerf_regtr.py --synthetic 16 --synthetic_res 128 --epochs 1 --root_dir /tmp/dreg
"""
import os
import random
import time

import torch
import torch.distributed as dist

from dreg_nerf_amd import losses as LS
from dreg_nerf_amd.checkpoint import CheckPointManager
from dreg_nerf_amd.config import config_parser
from dreg_nerf_amd.dataset import NeRFRegDataset, PrefetchLoader, SyntheticRegDataset
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
from dreg_nerf_amd.optim import broadcast_buffers


def to_device(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


@torch.no_grad()
def validate(model, dataset, dev, frac=0.2, rank=0, world=1):
    """train_nerf_regtr.py:258-291: RRE/RTE on the first 20 % of the validation scenes; score = n / sum(R_mean).
    With several ranks EVERY rank calls this at the same iteration and evaluates scenes rank, rank + world, ...; the two error sums
    are added over the ranks (one small all-reduce), so no rank waits in the next step's gradient exchange while rank 0 validates."""
    if world > 1:
        # plain DDP averages gradients, not BatchNorm running statistics: every rank evaluates (and rank 0 later saves) rank 0's buffers
        broadcast_buffers(model, 0)
    model.eval()
    n = max(1, int(len(dataset) * frac))
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    for i in range(rank, n, world):
        data = to_device(dataset[i], dev)
        pred = model(data)
        err = LS.evaluate_camera_alignment(pred["pose"][-1], data["pose"])
        sums[0] += err["R_error_mean"].double()
        sums[1] += err["t_error_mean"].double()
    if world > 1:
        dist.all_reduce(sums)
    model.train()
    r_sum, t_sum = (float(v) for v in sums.tolist())
    return n / max(r_sum, 1e-9), r_sum / n, t_sum / n


def main():
    # The loader thread (dataset.PrefetchLoader) and this thread share the interpreter lock; with CPython's default 5 ms switch
    # interval the launching thread can sit out several milliseconds per hand-over while the GPU runs dry.
    import sys
    sys.setswitchinterval(float(os.environ.get("DREG_SWITCH_INTERVAL", "0.0005")))
    cfg = config_parser()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", cfg.local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    random.seed(cfg.seed + rank)
    torch.manual_seed(cfg.seed)

    if cfg.synthetic > 0:
        train_ds, val_ds = SyntheticRegDataset(cfg.synthetic, cfg.synthetic_res, "train"), SyntheticRegDataset(max(2, cfg.synthetic // 4), cfg.synthetic_res, "test")
    else:
        train_ds = NeRFRegDataset(cfg.root_dir, cfg.json_dir, cfg.dataset, "train", sparse=True, device=dev)
        val_ds = NeRFRegDataset(cfg.root_dir, cfg.json_dir, cfg.dataset, "test", sparse=True, device=dev)
    model = NeRFRegTr(cfg.position_embedding_type, cfg.position_embedding_dim, cfg.position_embedding_scaling,
                      cfg.num_downsample, precision=cfg.precision).to(dev).train()
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    ts = TrainStep(model, lr=cfg.lr, robust_loss=cfg.robust_loss, finetune=cfg.finetune)
    save_dir = os.path.join(cfg.root_dir, "out", cfg.expname)
    ckpt = CheckPointManager(save_dir if rank == 0 else None, verbose=rank == 0)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    models = {"model": model, "feature_loss": ts.feature_loss}
    loader = CheckPointManager(verbose=rank == 0)     # every rank reads; only rank 0 owns the index file and writes
    loader.set_save_path(save_dir)
    start = loader.load(cfg, models=models,
                        optimizers=None if cfg.no_load_opt else {"optimizer": ts.optimizer},
                        schedulers=None if cfg.no_load_scheduler else {"scheduler": ts.scheduler}, map_location=dev)
    iteration = 0 if cfg.finetune else start
    per_step = cfg.pairs_per_step
    log = open(os.path.join(save_dir, "log.txt"), "a") if rank == 0 else None
    score = 0.0
    for epoch in range(cfg.epochs):
        ids = list(range(len(train_ds)))
        random.Random(cfg.seed + epoch).shuffle(ids)
        # every rank must run the same number of steps (each step ends in a blocking gradient all-reduce, and the step count drives
        # StepLR and the checkpoint cadence): keep a multiple of world * pairs_per_step scenes of this epoch's permutation
        usable = (len(ids) // (world * per_step)) * world * per_step
        if usable == 0:
            raise SystemExit(f"{len(ids)} training scenes < world ({world}) x pairs_per_step ({per_step}): nothing to train on")
        ids = ids[:usable][rank::world]
        # on-disk data: samples are read, uploaded and augmented two steps ahead on a loader thread / stream
        # (the loader shares the model's geometry stream: no fifth HIP stream next to the step's four — NeRFRegTr.geometry_stream)
        loader = PrefetchLoader(train_ds, ids, dev, depth=2 * per_step, stream=None if os.environ.get("DREG_LOADER_OWN_STREAM") == "1" else model.geometry_stream(dev)) if cfg.synthetic == 0 else None
        t_epoch, n_pairs = time.time(), 0
        t_wait = t_issue = 0.0
        for b in range(0, len(ids) - per_step + 1, per_step):
            t0 = time.perf_counter()
            batch = [next(loader) for _ in range(per_step)] if loader is not None else [to_device(train_ds[i], dev) for i in ids[b:b + per_step]]
            t1 = time.perf_counter()
            out = ts.step(batch)
            t_wait += t1 - t0
            t_issue += time.perf_counter() - t1
            iteration += 1
            n_pairs += per_step
            if rank == 0 and iteration % cfg.n_tensorboard == 0:
                pred, data = ts.last_preds[0], batch[0]
                err = LS.evaluate_camera_alignment(pred["pose"][-1].detach(), data["pose"])
                msg = f"it {iteration} " + " ".join(f"{k}={float(v):.4f}" for k, v in out["losses"].items()) + \
                      f" R={float(err['R_error_mean']):.3f}deg t={float(err['t_error_mean']):.4f} lr={ts.scheduler.get_last_lr()[0]:.2e}"
                print(msg, flush=True)
                log.write(msg + "\n"); log.flush()
            if iteration % cfg.n_validation == 0:
                score, r, t = validate(model, val_ds, dev, rank=rank, world=world)
                if rank == 0:
                    print(f"val it {iteration}: R_mean={r:.3f} t_mean={t:.4f}", flush=True)
            if iteration % cfg.n_checkpoint == 0 and world > 1:
                broadcast_buffers(model, 0)     # checkpoint time: all ranks continue from the statistics that are saved
            if iteration % cfg.n_checkpoint == 0 and rank == 0:
                ckpt.save(models, {"optimizer": ts.optimizer}, iteration, schedulers={"scheduler": ts.scheduler}, score=score)
        torch.cuda.synchronize()
        if rank == 0:
            dt = time.time() - t_epoch
            msg = (f"epoch {epoch}: {n_pairs} pairs on this rank in {dt:.2f}s = {n_pairs * world / dt:.1f} pairs/s over {world} GPU(s), input pipeline included "
                   f"(this thread: {t_wait:.2f}s waiting for samples, {t_issue:.2f}s issuing steps)")
            print(msg, flush=True)
            log.write(msg + "\n"); log.flush()
            from dreg_nerf_amd import train_step as _TS
            if _TS.STEP_TIMERS is not None:
                print("  host seconds by phase: " + ", ".join(f"{k} {v:.2f}" for k, v in _TS.STEP_TIMERS.items()), flush=True)
                _TS.STEP_TIMERS.clear()
                from dreg_nerf_amd import visibility as _V
                print(f"  label descriptor staging waits: {_V.STAGING_WAIT[0]:.2f}s", flush=True)
                _V.STAGING_WAIT[0] = 0.0
    score, r, t = validate(model, val_ds, dev, rank=rank, world=world)
    if rank == 0:
        print(f"final val: R_mean={r:.3f} t_mean={t:.4f}", flush=True)
        ckpt.save(models, {"optimizer": ts.optimizer}, iteration, schedulers={"scheduler": ts.scheduler}, score=score)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""How eval_nerf_regtr.py spreads scenes over ranks ('RegTR eval': independent scenes, replicas only, rows gathered on rank 0).
No device code here: the CPU tests drive it with two gloo ranks (tests/test_abi_and_ddp.py)."""
from typing import Dict, List

import torch.distributed as dist


def my_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Scene indices of this rank: rank, rank + world, ..."""
    return list(range(rank, n_scenes, world))


def block_orders(dataset) -> Dict[int, list]:
    """The source / target block order of EVERY scene, drawn in scene order on every rank (quirk Q15: the reference's dataset shuffles a scene's block ids per
    access from the process's Python RNG, which eval_nerf_regtr.py seeds like the reference, setup_seed(config.seed)).  Drawing all of them — not only this
    rank's — makes a scene's assignment the one-rank run's whatever the rank count: the gathered metrics file does not depend on the sharding."""
    return {i: dataset.draw_block_order(i) for i in range(len(dataset))} if hasattr(dataset, "draw_block_order") else {}


def gather_rows(rows: dict, world: int) -> dict:
    """Union of the ranks' {scene: row} dictionaries on every rank (all_gather_object: a few hundred bytes per scene), in scene-name order of arrival by rank."""
    if world <= 1:
        return dict(rows)
    gathered = [None] * world
    dist.all_gather_object(gathered, rows)
    out = {}
    for g in gathered:
        for k, v in g.items():
            if k in out:
                raise RuntimeError(f"scene {k} was evaluated by two ranks")
            out[k] = v
    return out


def summary(rows: dict) -> dict:
    """The reference's file layout (eval_nerf_regtr.py:240-257): per-scene rows + R_mean / t_mean over the scenes."""
    out = dict(rows)
    out["R_mean"] = sum(r["R_mean"] for r in rows.values()) / max(len(rows), 1)
    out["t_mean"] = sum(r["t_mean"] for r in rows.values()) / max(len(rows), 1)
    return out

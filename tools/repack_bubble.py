#!/usr/bin/env python3
"""How much of a training step is the weight repack that follows the optimizer update?  Steps timed as usual, then with both repacks
(trunk executor + point-set half) turned into no-ops (stale packs: timing only).  The difference is the upper bound of what a faster
or better-hidden repack can give.  usage: python tools/repack_bubble.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, synth, trunk_exec
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(5): ts.step(batch)
torch.cuda.synchronize()
orig_all, orig_exec = ops.repack_all, trunk_exec.TrunkExecutor.repack_after_update


def timed(n=15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ts.step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"with": [], "without": []}
for rnd in range(4):
    ops.repack_all, trunk_exec.TrunkExecutor.repack_after_update = orig_all, orig_exec
    ts.step(batch); res["with"].append(timed())
    def fake_all(device=None):            # mark every cached pack fresh without launching anything
        for key, ent in list(ops._pack_cache.items()):
            base = ent[0]()
            if base is not None:
                ops._pack_cache[key] = (ent[0], (base._version, ops._weight_generation)) + ent[2:]

    def fake_exec(self):
        self.pack_stamp = (ops._weight_generation, sum(t._version for t in self._param_refs))

    ops.repack_all = fake_all
    trunk_exec.TrunkExecutor.repack_after_update = fake_exec
    ts.step(batch); res["without"].append(timed())
for k, v in res.items():
    print(f"{k:8s} repack: " + " ".join(f"{x:.2f}" for x in v) + f"  ms/step, best {min(v):.2f}")

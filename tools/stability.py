"""Soak test: many optimizer steps over changing synthetic pairs (row counts change every step); prints loss trend, peak memory,
NaN check.  usage: python tools/stability.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model, lr=1e-4)
pose = synth.fixed_pose()
pool = []
def make(i, j, r0, r1):
    g, mk = synth.shell_grid(128, 1 + 2 * i + j, r0, r1)
    return g.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mk
EDGE = os.environ.get("SOAK_EDGE")
for i in range(12):      # 12 different pairs (different shell radii -> different row counts), cycled in groups
    d = {"pose": pose[None].clone(), "src_nerf_path": "", "tgt_nerf_path": ""}
    for j, side in enumerate(("src", "tgt")):
        r0, r1 = 0.76 + 0.01 * (i % 5), 0.81 + 0.012 * ((i + j) % 4)
        if EDGE and i % 6 == 1:
            r0, r1 = 0.3, 1.2          # a thick shell: > 20 % of the volume occupied -> the dense head path for that step
        if EDGE and i % 6 == 4:
            r0, r1 = 0.20, 0.215       # a tiny shell: a few hundred voxels, fewer points than one attention tile after the voxel rounds
        d[side + "_xyz_rgba"], d[side + "_mask"] = make(i, j, r0, r1)
    pool.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
losses, t0 = [], time.perf_counter()
for s in range(steps):
    nb = (1 + s % 4) if EDGE else 4          # 1..4 pairs per step in the edge-case run
    batch = [pool[(4 * s + j) % len(pool)] for j in range(nb)]
    out = ts.step(batch)
    losses.append(out["losses"]["total"])
    if os.environ.get("SOAK_SYNC"):
        torch.cuda.synchronize()
        print("step", s, "ok", [(int(b["src_mask"].shape[-1]), int(b["tgt_mask"].shape[-1])) for b in batch], {k: round(float(v), 4) for k, v in out["losses"].items()}, flush=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
if os.environ.get("SOAK_ALLOC"):
    # steady-state check: a second pass over the same pool — how many device allocations (hipMalloc: synchronises) does a step with
    # CHANGING row counts still trigger, and what does it cost next to a fixed batch?
    st0 = torch.cuda.memory_stats()
    t1 = time.perf_counter()
    for s in range(steps):
        ts.step([pool[(4 * s + j) % len(pool)] for j in range(4)])
    torch.cuda.synchronize()
    dt_var = (time.perf_counter() - t1) / steps
    st1 = torch.cuda.memory_stats()
    t1 = time.perf_counter()
    for s in range(steps):
        ts.step(pool[:4])
    torch.cuda.synchronize()
    dt_fix = (time.perf_counter() - t1) / steps
    st2 = torch.cuda.memory_stats()
    k = "num_device_alloc"
    print(f"changing pairs: {1e3 * dt_var:.2f} ms/step, {st1[k] - st0[k]} device allocations, {st1['num_alloc_retries'] - st0['num_alloc_retries']} retries; "
          f"fixed batch: {1e3 * dt_fix:.2f} ms/step, {st2[k] - st1[k]} device allocations")
ls = [float(x) for x in losses]
assert all(l == l and abs(l) < 1e9 for l in ls), "non-finite loss"
k = max(1, steps // 10)
print(f"{steps} steps in {el:.1f} s ({1e3 * el / steps:.1f} ms/step); loss first {k}: {sum(ls[:k]) / k:.3f}  last {k}: {sum(ls[-k:]) / k:.3f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")

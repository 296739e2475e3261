"""How host-bound is one training step?  Times (a) the wall time per step, (b) the host time until step() returns,
(c) the same with the HIP-event KernelTimer enabled (what bench.py's timed region carries)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
pose = synth.fixed_pose()
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=pose)
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})

def run(n, prof):
    for _ in range(2):
        ts.step(batch)
    torch.cuda.synchronize()
    ops.PROFILER = ops.KernelTimer() if prof else None
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        ts.step(batch)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ops.PROFILER = None
    return 1e3 * el / n, 1e3 * host / n

for active in (True, False):
    model.active_set = active
    for prof in (False, True):
        w, h = run(8, prof)
        print(f"active_set={active} event_timer={prof}: wall {w:.1f} ms/step, host-until-return {h:.1f} ms/step", flush=True)

# host cost alone: the same launch sequence at 32^3, where the GPU work is negligible
import cProfile, pstats
for res in (64,):
    batch = []
    for i in range(4):
        d = synth.shell_pair(res, 1 + 2 * i, 2 + 2 * i, pose=pose)
        batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
    model.active_set = True
    w, h = run(8, False)
    print(f"res {res}: wall {w:.1f} ms/step, host {h:.1f}", flush=True)
batch = []
for i in range(4):
    d = synth.shell_pair(64, 1 + 2 * i, 2 + 2 * i, pose=pose)
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ts.step(batch)
torch.cuda.synchronize()
pr.disable()
st=pstats.Stats(pr); st.sort_stats("cumtime").print_stats("dreg_nerf_amd|run_backward|apply", 60)

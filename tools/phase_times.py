"""GPU-timeline duration of the phases of a step (torch events on the main stream), un-profiled."""
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
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
# monkeypatch phase boundaries
orig_fpn = model.fpn
def fpn(x, rows=None, *a, **kw):
    mark("fpn_fwd_start"); y = orig_fpn(x, rows, *a, **kw); mark("fpn_fwd_end"); return y
model.fpn = fpn
from dreg_nerf_amd import trunk_exec
orig_bwd = trunk_exec.TrunkExecutor.backward
def bwd(self, x, rows, g, generation=None):
    mark("fpn_bwd_start"); orig_bwd(self, x, rows, g, generation); mark("fpn_bwd_end")
trunk_exec.TrunkExecutor.backward = bwd
orig_fb = model.forward_batch
def fb(b):
    o = orig_fb(b); mark("forward_batch_end"); return o
model.forward_batch = fb
orig_geo = model._geometry
def geo(b, d):
    mark("geo_start(side)"); o = orig_geo(b, d); mark("geo_end(side)"); return o
model._geometry = geo
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
acc = {}
N = 8
NOSYNC = "--nosync" in sys.argv      # steady state: the host runs ahead, steps are not separated by a device synchronisation
allm = []
for _ in range(N):
    marks.clear()
    mark("step_start"); ts.step(batch); mark("step_end")
    if not NOSYNC: torch.cuda.synchronize()
    allm.append(list(marks))
torch.cuda.synchronize()
for ms in allm[2:]:
    base = ms[0][1]
    for (n1, e1) in ms[1:]:
        acc[("step_start", n1)] = acc.get(("step_start", n1), 0.0) + base.elapsed_time(e1)
for k, v in acc.items(): print(f"{k[0]:20s} -> {k[1]:20s} {v/(N-2):7.2f} ms")
# fwd trunk = fpn_fwd_end - fpn_fwd_start; point-set half fwd + losses + its backward = fpn_bwd_start - fpn_fwd_end;
# trunk backward = fpn_bwd_end - fpn_bwd_start; reduce join + clip + AdamW + repack = step_end - fpn_bwd_end

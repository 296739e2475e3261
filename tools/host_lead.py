#!/usr/bin/env python3
"""How far ahead of the GPU is the host at the phase boundaries of a steady-state training step?  At each boundary (after forward_batch,
after the losses, after backward, after the optimizer) a HIP event is recorded on the main stream and the host clock is read; after the
run, lead = (time the GPU reached the event) - (time the host recorded it).  A lead near zero at the end of a phase means the GPU ran dry
there: the host, not the kernels, paced that phase.  usage: python tools/host_lead.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth, train_step as TS
from dreg_nerf_amd.regtr import NeRFRegTr
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TS.TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((name, time.perf_counter(), ev))
orig_fb, orig_bw, orig_opt = model.forward_batch, ts._backward, ts.optimizer.step
def fb(b):
    mark("step_start"); r = orig_fb(b); mark("forward_issued"); return r
def bw(total, d):
    mark("losses_issued"); orig_bw(total, d); mark("backward_issued")
def opt():
    r = orig_opt(); mark("optimizer_issued"); return r
model.forward_batch, ts._backward, ts.optimizer.step = fb, bw, opt
for _ in range(3): ts.step(batch)
torch.cuda.synchronize(); marks.clear()
base = torch.cuda.Event(enable_timing=True); base.record(); t_base = time.perf_counter()
torch.cuda.synchronize()
N = 8
for _ in range(N): ts.step(batch)
torch.cuda.synchronize()
rows = {}
for name, th, ev in marks:
    rows.setdefault(name, []).append((1e3 * (th - t_base), base.elapsed_time(ev)))
per = len(marks) // N
print(f"{N} steps; per boundary: host time and GPU time since the start of the run (ms, last 4 steps), lead = GPU - host")
for i in range(N - 4, N):
    print("step", i, "  ".join(f"{n}: host {rows[n][i][0]:7.2f} gpu {rows[n][i][1]:7.2f} lead {rows[n][i][1] - rows[n][i][0]:6.2f}" for n in rows))

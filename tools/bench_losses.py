"""The loss phase of a training step on its own (csrc/losses.hip): regtr_losses forward + backward on the benchmark's row space (4 pairs, ~1,200 key points
per side), kernel launches counted through a torch profiler trace.  usage: python tools/bench_losses.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import attn_ops as A, fused_losses as FL, losses as LS
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
segs = [(1217, 1221), (1198, 1240), (1225, 1209), (1230, 1212)]
R = sum(a + b for a, b in segs)
tab = A.ProblemTable(segs, dev)
xyz = ((torch.rand(R, 3, generator=g) - 0.5) * 1.6).to(dev)
cond = (torch.randn(6, R, 256, generator=g) * 0.3).to(dev).requires_grad_(True)
corr = (xyz.cpu()[None] + 0.1 * torch.randn(6, R, 3, generator=g)).to(dev).requires_grad_(True)
ov = torch.sigmoid(torch.randn(6, R, 1, generator=g)).to(dev).requires_grad_(True)
gt = (torch.rand(6, R, generator=g) > 0.4).float().to(dev)
tilde = (torch.rand(6, R, generator=g) > 0.5).float().to(dev)
poses = torch.eye(4)[None].repeat(4, 1, 1).to(dev)
fl = LS.InfoNCELoss(256, 0.2, 0.4).to(dev)
bt = {"cond": cond, "corr": corr, "ov": ov, "xyz": xyz, "tab": tab}
def step():
    out = FL.regtr_losses(bt, poses, fl, gt, tilde, False)
    out["total"].backward()
    cond.grad = corr.grad = ov.grad = None
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): step()
e1.record(); torch.cuda.synchronize()
print(f"loss phase forward + backward: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per step (GPU time between events, host running ahead)")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(f"{len(ks)} device launches per step, {sum(e.cuda_time for e in ks):.0f} us of kernel time:")
agg = {}
for e in ks:
    a = agg.setdefault(e.name[:70], [0, 0.0]); a[0] += 1; a[1] += e.cuda_time
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {n:3d} x {k:70s} {t:8.1f} us")

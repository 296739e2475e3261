"""Every torch (aten) operator of one training step that launches device work, grouped by operator and by the line of
dreg_nerf_amd that called it: the small launches between the HIP kernels.  usage: python tools/aten_launches.py"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train(); ts = TrainStep(model)
pose = synth.fixed_pose(); batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=pose)
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(2): ts.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(batch)
    torch.cuda.synchronize()
evs = prof.events()
# device launches per CPU op: an op "launches" if a kernel/memcpy/memset event is attributed to it and to none of its children
cnt = collections.Counter(); where = collections.defaultdict(collections.Counter)
nk = 0
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if any(c.kernels for c in e.cpu_children):
        continue
    name = e.name
    n = len(e.kernels)
    nk += n
    frame = "?"
    for fr in (e.stack or []):
        if "dreg_nerf_amd" in fr and "torch/" not in fr:
            frame = fr.split("dreg_nerf_amd/")[-1][:70]
            break
    cnt[name] += n
    where[name][frame] += n
print(f"{nk} device launches attributed to CPU ops")
for name, c in cnt.most_common(60):
    print(f"{c:5d}  {name}")
    for fr, k in where[name].most_common(6):
        print(f"          {k:4d}  {fr}")

"""Which torch (aten) operators still run inside one training step, with their shapes: anything large here is traffic outside the
HIP kernels (autograd's slice / accumulate bookkeeping, casts, concatenations).  usage: python tools/aten_ops.py [min_elements]"""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    ts.step(batch)
torch.cuda.synchronize()
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cnt = collections.Counter(); size = {}
skip = ("aten::empty", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::reshape", "aten::detach", "aten::t", "aten::transpose",
        "aten::_unsafe_view", "aten::expand", "aten::permute", "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::narrow", "aten::empty_like", "aten::empty_strided",
        "aten::split", "aten::split_with_sizes", "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::contiguous", "aten::to")
for e in prof.events():
    if not e.name.startswith("aten::") or e.name in skip: continue
    n = 0
    for sh in (e.input_shapes or []):
        if sh:
            k = 1
            for d in sh: k *= d
            n = max(n, k)
    if n >= thr:
        key = (e.name, str(e.input_shapes)[:90]); cnt[key] += 1; size[key] = n
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1] * size[kv[0]])[:40]:
    print(f"{c:4d} x {size[key] / 1e6:8.2f} M elements  {key[0]:24s} {key[1]}")

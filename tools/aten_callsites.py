"""Which lines of dreg_nerf_amd issue torch (aten) device operators in one training step: a TorchDispatchMode that records the first
dreg_nerf_amd frame of every aten call touching a CUDA tensor (the torch profiler has no Python stacks in this build).  Operators inside
the autograd engine's own nodes (accumulations, views) are not Python calls and show up under the frame that called backward().
usage: python tools/aten_callsites.py"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train(); ts = TrainStep(model)
pose = synth.fixed_pose(); batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=pose)
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(2): ts.step(batch)
torch.cuda.synchronize()
NOLAUNCH = ("view", "reshape", "detach", "alias", "expand", "slice", "select", "unsqueeze", "squeeze", "permute", "transpose", "t.default", "as_strided",
            "empty", "_unsafe_view", "split", "unbind", "is_", "size", "stride", "sym_", "_local_scalar", "lift_fresh", "record_stream", "narrow", "chunk", "view_as")
cnt = collections.Counter()


def is_cuda(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in NOLAUNCH):
            return out
        flat = list(args) + list((kwargs or {}).values()) + (list(out) if isinstance(out, (tuple, list)) else [out])
        if any(is_cuda(a) or (isinstance(a, (list, tuple)) and any(is_cuda(b) for b in a)) for a in flat):
            fr = "?"
            for f in reversed(traceback.extract_stack(limit=40)):
                if "dreg_nerf_amd" in f.filename:
                    fr = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
                    break
            cnt[(name, fr)] += 1
        return out


with torch.autograd.set_multithreading_enabled(False), Mode():   # backward on this thread: the mode (thread-local) sees it too
    ts.step(batch)
torch.cuda.synchronize()
tot = sum(cnt.values())
print(f"{tot} aten calls on device tensors in one step (autograd multithreading off: backward traced too)")
for (name, fr), c in cnt.most_common(70):
    print(f"{c:4d}  {name:38s} {fr}")

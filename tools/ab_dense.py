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
def run(n):
    for _ in range(3):
        ts.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(batch)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
seq = sys.argv[1] if len(sys.argv) > 1 else "ADAD"
from dreg_nerf_amd import lib as L
if len(sys.argv) > 2:
    L.load().dreg_conv_set_wgrad_target_blocks(int(sys.argv[2]))
for c in seq:
    if c == "E":
        torch.cuda.empty_cache(); print("empty_cache"); continue
    model.active_set = c == "A"
    print(f"mode {c}: {run(6):.1f} ms/step  reserved={torch.cuda.memory_reserved()/2**30:.1f} GiB alloc={torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)

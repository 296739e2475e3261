"""Forward-only registration throughput (the eval_nerf_regtr.py path: model.eval(), no gradients) for 1 and 4 pairs per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
dev = torch.device("cuda"); torch.manual_seed(0)
m = NeRFRegTr(precision="bf16").to(dev).eval()
pairs = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    pairs.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
with torch.no_grad():
    for nb in (1, 4):
        batch = pairs[:nb]
        for _ in range(3): m.forward_batch(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): m.forward_batch(batch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{nb} pair(s) per call: {1e3 * dt:.2f} ms per call = {nb / dt:.1f} pairs/s")

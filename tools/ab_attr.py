"""A/B of one boolean attribute of the model (NeRFRegTr.<attr>) on the whole training step, values alternated inside one process.
usage: python tools/ab_attr.py batched_subsample [--rounds 3] [--steps 12]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
args = sys.argv[1:]
attr = args[0]
rounds = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 3
steps = int(args[args.index("--steps") + 1]) if "--steps" in args else 12
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
res = {False: [], True: []}
for r in range(rounds):
    for v in ((False, True) if r % 2 == 0 else (True, False)):
        setattr(model, attr, v)
        for _ in range(2): ts.step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): ts.step(batch)
        torch.cuda.synchronize(); res[v].append(1e3 * (time.perf_counter() - t0) / steps)
for v in (False, True):
    print(f"{attr}={v}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  ms/step, best {min(res[v]):.2f}")

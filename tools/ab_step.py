"""A/B of one tuning setter (include/dreg_nerf_tuning.h) on the whole training step, alternating the values inside one process so that
clock / thermal drift cancels.  usage: python tools/ab_step.py dreg_conv_set_narrow_small 0 1 2 [--dense] [--rounds 3] [--steps 12]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
args = sys.argv[1:]
dense = "--dense" in args
rounds = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 3
steps = int(args[args.index("--steps") + 1]) if "--steps" in args else 12
setter = args[0]
vals = []
for i, a in enumerate(args[1:], 1):
    if a.lstrip("-").isdigit() and args[i - 1] not in ("--rounds", "--steps"):
        vals.append(int(a))
lib = L.load()
fn = getattr(lib, setter)
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train(); model.active_set = not dense
ts = TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
res = {v: [] for v in vals}
for r in range(rounds):
    for v in (vals if r % 2 == 0 else vals[::-1]):
        fn(v)
        model.__dict__.pop('_trunk_cache', None)      # setters read at executor creation take effect
        for _ in range(2): ts.step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): ts.step(batch)
        torch.cuda.synchronize(); res[v].append(1e3 * (time.perf_counter() - t0) / steps)
for v in vals:
    print(f"{setter}({v}): " + " ".join(f"{t:.2f}" for t in res[v]) + f"  ms/step, best {min(res[v]):.2f}")

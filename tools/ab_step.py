"""A/B of one switch on the whole training step, alternating the values inside one process so that clock / thermal drift cancels.
The switch is either a creation option of the trunk executor (dreg_exec_opts of include/dreg_nerf.h: `opt:sparse_stem`), the point-set executor's
per-handle `ps:group_wgrad`, or a process-global kernel-variant setter of the MEASUREMENT build (include/dreg_nerf_probe.h: `dreg_conv_set_narrow_small`),
in which case the whole run uses libdreg_nerf_hip_probe.so.
usage: python tools/ab_step.py opt:sparse_stem 0 1 | ps:group_wgrad 0 1 | dreg_conv_set_narrow_small 0 1 2 [--dense] [--rounds 3] [--steps 12]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, pointset_exec, synth, trunk_exec
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
if setter.startswith("opt:"):
    def fn(v): trunk_exec.OPTS[setter[4:]] = v
elif setter == "ps:group_wgrad":
    def fn(v): pointset_exec.GROUP_WGRAD = bool(v)
elif setter == "aux:cumask":      # 0 = the default low-priority second stream; 1..7 = CU masks of trunk_exec.aux_stream
    def fn(v):
        trunk_exec.AUX_CU_MASK = v
        trunk_exec._AUX.clear()
        ts._pg_stream = None
else:
    fn = getattr(L.use_probe(), setter)
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
        model.__dict__.pop('_trunk_cache', None)      # creation options / shape-dependent choices made at executor creation take effect
        model.__dict__.pop('_ps_cache', None)
        for _ in range(2): ts.step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): ts.step(batch)
        torch.cuda.synchronize(); res[v].append(1e3 * (time.perf_counter() - t0) / steps)
for v in vals:
    print(f"{setter}({v}): " + " ".join(f"{t:.2f}" for t in res[v]) + f"  ms/step, best {min(res[v]):.2f}")

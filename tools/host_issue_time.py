"""How far ahead of the GPU the host runs in the training step: the host time of step() (no synchronisation inside the timed loop)
against the wall time per step.  DREG_NO_GPU_WORK is not needed: the host time is measured with the stream kept busy, and a
second run after a long sleep-free warm queue shows whether issuing alone could sustain the rate.
usage: python tools/host_issue_time.py [--dense] [--steps 24]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
args = sys.argv[1:]
steps = int(args[args.index("--steps") + 1]) if "--steps" in args else 24
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train(); model.active_set = "--dense" not in args
ts = TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
for _ in range(4): ts.step(batch)
torch.cuda.synchronize()
host = []
t00 = time.perf_counter()
for _ in range(steps):
    t0 = time.perf_counter(); ts.step(batch); host.append(1e3 * (time.perf_counter() - t0))
t_issue = 1e3 * (time.perf_counter() - t00)
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t00)
print(f"host time of step(): {' '.join(f'{h:.1f}' for h in host)} ms")
print(f"issue {t_issue / steps:.2f} ms/step, wall {wall / steps:.2f} ms/step, GPU still busy {wall - t_issue:.2f} ms after the last step() returned")
# phases of one step on the host (each phase synchronised: the sum is not the step time; the host share of each phase is what matters)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(6): ts.step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(28)

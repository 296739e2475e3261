"""A/B in one process: the training step with labels from NeRF blocks, unsplit vs split at several background-launch widths (review item 2, round 5)."""
import json, os, sys, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dreg_nerf_amd import synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
pose = synth.fixed_pose()
td, paths = bench.write_generated_blocks(8, 50, 0)
base, batch = [], []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=pose)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
    base.append(d)
    batch.append(dict(d, src_nerf_path=paths[2 * i], tgt_nerf_path=paths[2 * i + 1]))

def run(b, n=12):
    for _ in range(3):
        ts.step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(b)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
out = {}
ts.split_labels = False
out["synthetic_labels_ms"] = run(base)
out["unsplit_ms"] = run(batch)
ts.split_labels = True
ts.label_waves = (128, 128)
for dbg in ("", "skip_tilde", "skip_kp", "skip_kp,skip_tilde", "kp_geo", ""):
    os.environ["DREG_LABEL_DEBUG"] = dbg
    out[f"split_128_{dbg or 'default'}_ms"] = run(batch)
os.environ["DREG_LABEL_DEBUG"] = "kp_geo"
for w in ((0, 128), (512, 128), (128, 96), (128, 64)):
    ts.label_waves = w
    out[f"kp_geo_waves{w[0]}_{w[1]}_ms"] = run(batch)
os.environ["DREG_LABEL_DEBUG"] = ""
ts.split_labels = False
out["unsplit_again_ms"] = run(batch)
out["synthetic_again_ms"] = run(base)
shutil.rmtree(td, ignore_errors=True)
print(json.dumps(out, indent=1))

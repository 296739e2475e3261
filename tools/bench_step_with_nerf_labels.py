#!/usr/bin/env python3
"""One training step whose overlap labels come from the pairs' NeRF blocks (train_nerf_regtr.py:186-199: surface-field visibility of
the key points and of the predicted correspondences, four ray-march calls per pair through csrc/visibility.hip) against the same step
with synthetic labels (bench.py).  Blocks: generated NGP weights, a shell-shaped 128^3 occupancy grid, NCAM training cameras on a
sphere; 4 pairs per step.  usage: python tools/bench_step_with_nerf_labels.py [NCAM] [WSCALE]"""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ngp, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
NCAM = int(sys.argv[1]) if len(sys.argv) > 1 else 50
WSCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0      # density MLP weight scale: 3 = opaque surfaces (rays end within a few samples, as in a trained block), 0.5 = thin fog
dev = torch.device("cuda", 0)
AABB = [-1.5] * 3 + [1.5] * 3
res = 128
g = torch.Generator().manual_seed(0)
td = tempfile.mkdtemp(prefix="dreg_nl_")
c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
paths = []
for b in range(8):
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * WSCALE
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
    occ = ngp.OccupancyGrid(AABB, res)
    occ._binary.copy_((rad > 0.75) & (rad < 0.88))
    poses = torch.eye(4)[None].repeat(NCAM, 1, 1)
    poses[:, :3, 3] = torch.nn.functional.normalize(torch.randn(NCAM, 3, generator=g), dim=-1) * 3.0
    p = os.path.join(td, f"block_{b}.pth")
    torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": AABB, "unbounded": False, "near_plane": None, "far_plane": None,
                "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB, "render_step_size": 3 * 3 ** 0.5 / 1024,
                "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": b}, p)
    paths.append(p)
torch.manual_seed(3407)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
batch = []
for i in range(4):
    d = synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})


def timed(n=10):
    for _ in range(3): ts.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ts.step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from dreg_nerf_amd import visibility
a = timed()
for i, d in enumerate(batch):
    d["src_nerf_path"], d["tgt_nerf_path"] = paths[2 * i], paths[2 * i + 1]
visibility.PERSISTENT = False
b0 = timed()
visibility.PERSISTENT = True
b = timed()
kp = ts.last_preds[0]["src_kp"][0].shape[0]
print(f"(lock-step visibility kernel: {b0:.2f} ms per step)")
print(f"synthetic labels {a:.2f} ms per step of 4 pairs; labels from the NeRF blocks ({NCAM} cameras, ~{kp} key points per cloud, 7 x 2 point sets per pair, field weight scale {WSCALE}) {b:.2f} ms -> {4e3 / b:.1f} pairs/s")

#!/usr/bin/env python3
"""End-to-end training throughput of train_nerf_regtr.py on ON-DISK blocks in the reference's directory layout (input pipeline
included: disk -> sparse block -> H2D -> on-device augmentation through dataset.PrefetchLoader), next to bench.py's synthetic,
HBM-resident number.  8 real shell-R scenes (2 blocks of 128^3, dense voxel_grid.pt = 58.7 MB each as the reference writes them) are
exposed under 256 scene names (symlinks; IO_NAMES); epoch 0 builds the voxel_sparse.pt caches, later epochs are the steady state.
IO_NERF=1 also writes every block's NeRF checkpoint (generated weights, 50 cameras): the overlap labels of the step then come from ray
marching the blocks (four label sets per pair) and the blocks themselves go through visibility's cache / the loader's prefetch."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreg_nerf_amd import synth  # noqa: E402
from dreg_nerf_amd.dataset import _small_se3  # noqa: E402


def main():
    res, n_real, n_names, per_step = 128, 8, int(os.environ.get("IO_NAMES", "256")), 4
    root = tempfile.mkdtemp(prefix="dreg_io_")
    jdir = os.path.join(root, "json")
    os.makedirs(jdir)
    t0 = time.time()
    for i in range(n_real):
        tf = {}
        for k in range(2):
            pose = _small_se3(0.2, torch.Generator().manual_seed(10 * i + k)) if k else torch.eye(4)
            g, m = synth.shell_grid(res, 2 * i + k + 1, *synth.shell_radii(res), pose=pose)
            d = os.path.join(root, "objaverse", "nerf_models", f"real_{i}", f"block_{k}")
            os.makedirs(d)
            torch.save(g, os.path.join(d, "voxel_grid.pt"))
            torch.save(m, os.path.join(d, "voxel_mask.pt"))
            if os.environ.get("IO_NERF"):      # the block's NeRF checkpoint too (train_ngp_nerf.py:187-209): the step's overlap labels then come from ray marching it
                from dreg_nerf_amd import ngp
                gen = torch.Generator().manual_seed(100 + 2 * i + k)
                f = ngp.NGPradianceField([-1.5] * 3 + [1.5] * 3)
                with torch.no_grad():
                    f.mlp_base.params[:3072] = torch.randn(3072, generator=gen) * 3.0            # opaque surfaces: rays end within a few samples
                    f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=gen)
                occ = ngp.OccupancyGrid([-1.5] * 3 + [1.5] * 3, res)
                binary = torch.zeros(res ** 3, dtype=torch.bool)
                binary[m] = True
                occ._binary.copy_(binary.view(res, res, res))
                cams = torch.eye(4)[None].repeat(50, 1, 1)
                cams[:, :3, 3] = torch.nn.functional.normalize(torch.randn(50, 3, generator=gen), dim=-1) * 3.0
                torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": [-1.5] * 3 + [1.5] * 3, "unbounded": False,
                            "near_plane": None, "far_plane": None, "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB,
                            "render_step_size": 3 * 3 ** 0.5 / 1024, "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": cams, "block_id": k},
                           os.path.join(d, "model.pth"))
            tf[str(k)] = pose.tolist()
        os.makedirs(os.path.join(root, "objaverse", "images", f"real_{i}"))
        json.dump(tf, open(os.path.join(root, "objaverse", "images", f"real_{i}", "world_frame_transforms.json"), "w"))
    names = {}
    for j in range(n_names):
        nm = f"Scene_{j:03d}"
        for sub in ("nerf_models", "images"):
            os.symlink(os.path.join(root, "objaverse", sub, f"real_{j % n_real}"), os.path.join(root, "objaverse", sub, nm))
        names[f"uid{j:03d}"] = nm
    json.dump({"objaverse": {"train": list(names), "test": list(names)[:2]}}, open(os.path.join(jdir, "objaverse.json"), "w"))
    json.dump(names, open(os.path.join(jdir, "obj_id_names.json"), "w"))
    print(f"wrote {2 * n_real} dense blocks ({2 * n_real * 58.7:.0f} MB) in {time.time() - t0:.1f}s", flush=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_nerf_regtr.py"), "--root_dir", root, "--json_dir", jdir, "--dataset", "objaverse",
                        "--expname", "io", "--epochs", "4", "--pairs_per_step", str(per_step), "--n_tensorboard", "100000", "--n_validation", "100000",
                        "--n_checkpoint", "100000"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    print(r.stdout[-3000:])
    if r.returncode != 0:
        print(r.stderr[-3000:])
        raise SystemExit(r.returncode)
    rates = [float(x) for x in re.findall(r"= ([0-9.]+) pairs/s", r.stdout)]
    print(json.dumps({"metric": "train_nerf_regtr_on_disk_pairs_per_sec_128", "epochs_pairs_per_s": rates, "steady_state": max(rates[1:]) if len(rates) > 1 else None,
                      "nerf_labels": bool(os.environ.get("IO_NERF")), "config": f"{n_names} scene names over {n_real} real scenes x 2 blocks at {res}^3, pairs_per_step {per_step}, sparse cache after epoch 0, 1 GPU"}))


if __name__ == "__main__":
    main()

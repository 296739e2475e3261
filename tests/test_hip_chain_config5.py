"""GPU: BASELINE.json configs[4] as one chain on a synthetic stand-in split (the Objaverse blocks and the trained checkpoint are external
downloads): generated NGP blocks -> eval_ngp_nerf.py (grid extraction) -> eval_nerf_regtr.py (registration + RRE/RTE) over a split
described by the REFERENCE's split files (objaverse.json + obj_id_names.json), then every row of metrics_test.json is re-derived on the
CPU by the reference-pinned oracle from the extracted voxel_grid.pt / voxel_mask.pt and the ground-truth block transforms.
Reference flow: eval_ngp_nerf.py:336-451 -> eval_nerf_regtr.py:224-301 (metrics: :24-65)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from dreg_nerf_amd import ngp, params, synth  # noqa: E402
from dreg_nerf_amd.dataset import _small_se3  # noqa: E402
from oracle import regtr_oracle as O  # noqa: E402

AABB = [-1.5] * 3 + [1.5] * 3


def _run(args, timeout=900):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _make_block(path, seed, RES=32, shell=(0.55, 1.05)):
    """A NeRF block checkpoint in the reference's format (train_ngp_nerf.py:187-209) with generated weights: a thick occupancy shell,
    a random hash grid / MLP (density = exp(h0 - 1) with a heavy tail: some cells opaque enough to be surfaces), six cameras around the block."""
    g = torch.Generator().manual_seed(seed)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 1.0     # |h0| large enough that every block keeps ~1e3 voxels
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    c = (torch.arange(RES, dtype=torch.float32) + 0.5) / RES * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    binary = (rad > shell[0]) & (rad < shell[1])
    occ = ngp.OccupancyGrid(AABB, RES)
    occ._binary.copy_(binary)
    cams = torch.eye(4)[None].repeat(6, 1, 1)
    cams[:, :3, 3] = torch.tensor([[2.5, 0, 0], [-2.5, 0, 0], [0, 2.5, 0], [0, -2.5, 0], [0, 0, 2.5], [0, 0, -2.5]])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": AABB, "unbounded": False,
                "near_plane": None, "far_plane": None, "grid_resolution": RES, "contraction_type": ngp.ContractionType.AABB,
                "render_step_size": 0.005, "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": cams, "block_id": 0}, path)
    return int(binary.sum())


def test_extract_then_register_split_matches_oracle(tmp_path):
    _chain(tmp_path, 32, (0.55, 1.05), 3000, None)


def test_extract_then_register_at_the_baseline_size_128(tmp_path):
    """BASELINE.json configs[4] at the BASELINE resolution: four generated scenes of two 128^3 blocks each (a thin occupancy shell: ~4e4 occupied cells
    per block, as on the benchmark's grids) through grid extraction -> registration -> metrics_test.json; the fp32-mode leg is re-derived by the
    reference-pinned CPU oracle on EVERY scene (round 6: eight 128^3 oracle forwards take ~1 min on the collection box's 16 cores; rounds 3-5 re-derived
    one), the bf16 leg — the script's default precision — is held against the fp32-mode leg on every scene."""
    _chain(tmp_path, 128, (0.75, 0.88), 30000, None)


def _chain(tmp_path, RES, shell, min_occ, oracle_scenes):
    root, jdir = tmp_path / "root", tmp_path / "json"
    jdir.mkdir()
    scenes = {f"uid{i:02d}": f"Scene_{i:02d}" for i in range(4)}
    # the reference's split files: object uids per split, uid -> scene directory name (dataset.py:194-216)
    json.dump({"objaverse": {"train": [], "test": list(scenes)}}, open(jdir / "objaverse.json", "w"))
    json.dump(scenes, open(jdir / "obj_id_names.json", "w"))
    gt = {}
    for i, name in enumerate(scenes.values()):
        (root / "objaverse" / "images" / name).mkdir(parents=True)
        tf = {}
        for k in range(2):
            n_occ = _make_block(str(root / "objaverse" / "nerf_models" / name / f"block_{k}" / "model.pth"), 100 * i + k, RES, shell)
            assert n_occ > min_occ
            tf[str(k)] = _small_se3(0.2, torch.Generator().manual_seed(7 * i + k)).tolist()
        json.dump(tf, open(root / "objaverse" / "images" / name / "world_frame_transforms.json", "w"))
        gt[name] = {int(k): torch.tensor(v) for k, v in tf.items()}
    # stage 1: grid extraction for all 8 blocks
    out = _run(["eval_ngp_nerf.py", "--root_dir", str(root), "--dataset", "objaverse", "--multi_blocks"])
    assert out.count("voxels kept") == 8
    # stage 2: registration over the test split with a fixed checkpoint in the reference's format
    sd = params.synth_state_dict(0)
    os.makedirs(root / "out" / "chain", exist_ok=True)
    torch.save({"step": 1, "model": sd}, str(root / "out" / "chain" / "model.pth"))
    _run(["eval_nerf_regtr.py", "--root_dir", str(root), "--json_dir", str(jdir), "--dataset", "objaverse", "--expname", "chain",
          "--precision", "fp32", "--dump_outputs"])
    m = json.load(open(root / "eval" / "chain" / "objaverse" / "metrics_test.json"))
    assert set(m) == set(scenes.values()) | {"R_mean", "t_mean"}
    # the reference's per-scene files (eval_nerf_regtr.py:313-321, 369-438): estimated transformation + registration point clouds
    from dreg_nerf_amd import vis_dump
    for name in scenes.values():
        sdir = root / "eval" / "chain" / "objaverse" / name
        T = np.array(json.load(open(sdir / "transformation_est.json"))["transformation"])
        assert T.shape == (4, 4) and np.allclose(T[3], [0, 0, 0, 1]) and abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-3
        xyz, rgb = vis_dump.read_ply(str(sdir / "noisy_point_cloud_pred.ply"))
        ns = vis_dump.read_ply(str(sdir / "src_xyz.ply"))[0].shape[0]
        nt = vis_dump.read_ply(str(sdir / "tgt_xyz.ply"))[0].shape[0]
        assert xyz.shape == (ns + nt, 3) and (rgb[:ns] == [255, 0, 0]).all() and (rgb[ns:] == [0, 255, 0]).all()
        src = vis_dump.read_ply(str(sdir / "src_xyz.ply"))[0]
        assert np.allclose(xyz[:ns], src @ T[:3, :3].T + T[:3, 3], atol=1e-5)
        # the blocks' NeRF checkpoints are on disk here: the camera-pose dumps (eval_nerf_regtr.py:330-343) are written too
        un, al = torch.load(str(sdir / "unaligned_poses.pt")), torch.load(str(sdir / "aligned_poses_pred.pt"))
        assert un.shape == al.shape and un.shape[1:] == (4, 4) and (sdir / "aligned_poses_gt.pt").exists()
        assert torch.allclose(al[0], torch.from_numpy(T).float() @ un[0], atol=1e-5)
    # stage 3: every row re-derived by the oracle (CPU, fp32, eval-mode BatchNorm) from the files stage 1 wrote
    r_all, t_all = [], []
    for si, name in enumerate(scenes.values()):
        if oracle_scenes is not None and si >= oracle_scenes:
            r_all.append(m[name]["R_mean"])
            t_all.append(m[name]["t_mean"])
            continue
        blocks = {}
        for k in range(2):
            d = root / "objaverse" / "nerf_models" / name / f"block_{k}"
            grid, mask = torch.load(str(d / "voxel_grid.pt")), torch.load(str(d / "voxel_mask.pt"))
            assert grid.shape == (RES, RES, RES, 7) and mask.dtype == torch.int64 and mask.numel() > 50
            assert torch.all(mask[1:] > mask[:-1]) and torch.count_nonzero(grid.reshape(-1, 7)[mask][:, :3]) > 0
            blocks[k] = (grid.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mask)          # loader layout (dataset.py:244-248)
        cands = []
        for s, t in ((0, 1), (1, 0)):       # the dataset shuffles which block is the source (quirk Q15): either order is legitimate
            pose = (gt[name][t] @ torch.linalg.inv(gt[name][s]))[None]
            data = {"src_xyz_rgba": blocks[s][0], "tgt_xyz_rgba": blocks[t][0], "src_mask": blocks[s][1], "tgt_mask": blocks[t][1], "pose": pose}
            with torch.no_grad():
                pred = O.regtr_forward(params.clone_state_dict(sd), data, train=False)
            rre, rte = O.rre_rte(pred["pose"][-1], pose)
            cands.append((float(rre[0]), float(rte[0])))
        row = m[name]
        assert set(row) == {"R_mean", "t_mean", "R_med", "t_med", "time"}
        err = min(abs(row["R_mean"] - r) / max(r, 1.0) + abs(row["t_mean"] - t) / max(t, 1e-2) for r, t in cands)
        assert err < 2e-3, (name, row, cands)       # RRE in degrees / RTE of the HIP chain vs the oracle on the same grids
        assert row["R_med"] == pytest.approx(row["R_mean"]) and row["time"] > 0
        r_all.append(row["R_mean"])
        t_all.append(row["t_mean"])
    assert m["R_mean"] == pytest.approx(float(np.mean(r_all)), rel=1e-6) and m["t_mean"] == pytest.approx(float(np.mean(t_all)), rel=1e-6)
    # stage 4: the bf16 leg — eval_nerf_regtr.py's DEFAULT precision — on the well-conditioned weight profile (params.PROFILES["wc"]: at the
    # reference's random initialisation the pose of a bf16 evaluation is not determined to better than ~0.3, see test_hip_pinned_step.py),
    # against the fp32 leg of the same script on the same checkpoint: same scenes, same block order (the script seeds the shuffle)
    sd_wc = params.synth_state_dict(0, profile="wc")
    rows = {}
    for prec in ("fp32", "bf16"):
        os.makedirs(root / "out" / f"chain_wc_{prec}", exist_ok=True)
        torch.save({"step": 1, "model": sd_wc}, str(root / "out" / f"chain_wc_{prec}" / "model.pth"))
        _run(["eval_nerf_regtr.py", "--root_dir", str(root), "--json_dir", str(jdir), "--dataset", "objaverse", "--expname", f"chain_wc_{prec}"] +
             (["--precision", "fp32"] if prec == "fp32" else []))
        rows[prec] = json.load(open(root / "eval" / f"chain_wc_{prec}" / "objaverse" / "metrics_test.json"))
    for name in scenes.values():
        a, b = rows["fp32"][name], rows["bf16"][name]
        assert abs(a["R_mean"] - b["R_mean"]) <= 0.02 * max(a["R_mean"], 1.0) + 0.05, (name, a, b)       # degrees
        assert abs(a["t_mean"] - b["t_mean"]) <= 2e-2, (name, a, b)                                      # the absolute pose bound of test_hip_pinned_step.py
    assert abs(rows["fp32"]["R_mean"] - rows["bf16"]["R_mean"]) <= 0.05 + 0.02 * rows["fp32"]["R_mean"]

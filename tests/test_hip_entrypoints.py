"""GPU: the drop-in entry points run end to end on synthetic data: train (2 steps) -> checkpoint in the reference
format -> eval metrics JSON with the reference schema; NeRF block checkpoint -> voxel_grid.pt / voxel_mask.pt."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_train_then_eval_synthetic(tmp_path):
    root = str(tmp_path)
    _run(["train_nerf_regtr.py", "--synthetic", "2", "--synthetic_res", "64", "--epochs", "1", "--root_dir", root,
          "--expname", "t", "--n_tensorboard", "1", "--n_checkpoint", "1000"])
    ck = torch.load(os.path.join(root, "out", "t", "model.pth"), map_location="cpu", weights_only=False)
    assert set(["step", "model", "feature_loss", "optimizer", "scheduler"]) <= set(ck.keys())
    assert len(ck["model"]) == 772 and ck["feature_loss"]["W"].shape == (256, 256)
    assert os.path.exists(os.path.join(root, "out", "t", "model_best.pth")) and os.path.exists(os.path.join(root, "out", "t", "checkpoints.txt"))
    out = _run(["eval_nerf_regtr.py", "--synthetic", "2", "--synthetic_res", "64", "--root_dir", root, "--expname", "t", "--fgr_baseline"])
    m = json.load(open(os.path.join(root, "eval", "t", "synthetic", "metrics_test.json")))
    assert "R_mean" in m and "t_mean" in m and "shell_0000" in m and set(m["shell_0000"]) == {"R_mean", "t_mean", "R_med", "t_med", "time"}
    # the FGR baseline file of the reference (eval_nerf_regtr.py:261-273), same schema (a sphere shell has no unique pose: only the format is checked)
    fm = json.load(open(os.path.join(root, "eval", "t", "synthetic", "fgr_metrics_test.json")))
    assert set(fm) == set(m) and set(fm["shell_0000"]) == {"R_mean", "t_mean", "R_med", "t_med", "time"}


def _occ_state(ngp, binary, res):
    occ = ngp.OccupancyGrid([-1.5] * 3 + [1.5] * 3, res)      # the four persistent buffers a nerfacc 0.3.5 grid stores
    occ._binary.copy_(binary)
    return occ.state_dict()


def test_grid_extraction_from_block_checkpoint(tmp_path):
    from dreg_nerf_amd import ngp
    res = 32
    d = tmp_path / "objaverse" / "nerf_models" / "sceneA" / "block_0"
    d.mkdir(parents=True)
    f = ngp.NGPradianceField([-1.5] * 3 + [1.5] * 3)
    with torch.no_grad():
        g = torch.Generator().manual_seed(0)
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.4
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    binary = torch.rand(res, res, res, generator=torch.Generator().manual_seed(1)) < 0.05
    cam_poses = torch.eye(4)[None].repeat(6, 1, 1)   # six cameras around the block, outside the aabb
    cam_poses[:, :3, 3] = torch.tensor([[2.5, 0, 0], [-2.5, 0, 0], [0, 2.5, 0], [0, -2.5, 0], [0, 0, 2.5], [0, 0, -2.5]])
    torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": _occ_state(ngp, binary, res),
                "aabb": [-1.5] * 3 + [1.5] * 3, "unbounded": False, "near_plane": None, "far_plane": None, "grid_resolution": res,
                "contraction_type": ngp.ContractionType.AABB, "render_step_size": 0.005, "alpha_thre": 0.0, "cone_angle": 0.0,
                "camera_poses": cam_poses, "block_id": 0}, str(d / "model.pth"))
    _run(["eval_ngp_nerf.py", "--root_dir", str(tmp_path), "--dataset", "objaverse", "--multi_blocks"])
    grid = torch.load(str(d / "voxel_grid.pt"))
    mask = torch.load(str(d / "voxel_mask.pt"))
    assert grid.shape == (res, res, res, 7) and mask.dtype == torch.int64
    assert torch.all(binary.flatten()[mask]) and torch.all(mask[1:] > mask[:-1])
    # kept voxels = density mask AND surface-visible from a camera: a strict subset of the occupied cells here
    assert 0 < mask.numel() < int(binary.sum())
    # the reference's other outputs of sample_points (eval_ngp_nerf.py:350-394): the density-mask twins and both point clouds
    from dreg_nerf_amd.vis_dump import read_ply
    dgrid, dmask = torch.load(str(d / "density_voxel_grid.pt")), torch.load(str(d / "density_voxel_mask.pt"))
    assert dgrid.shape == grid.shape and dmask.dtype == torch.int64 and torch.all(dmask[1:] > dmask[:-1])
    assert set(mask.tolist()) <= set(dmask.tolist()) and torch.all(binary.flatten()[dmask])       # surface AND density is a subset of density
    assert torch.equal(dgrid.reshape(-1, 7)[mask], grid.reshape(-1, 7)[mask])                        # same samples where both keep a voxel
    others = torch.ones(res ** 3, dtype=torch.bool)
    others[dmask] = False
    assert float(dgrid.reshape(-1, 7)[others].abs().max()) == 0.0
    for name, g7, m in (("voxel_point_cloud.ply", grid, mask), ("density_voxel_point_cloud.ply", dgrid, dmask)):
        xyz, rgb = read_ply(str(d / name))
        rows = g7.reshape(-1, 7)[m]
        assert xyz.shape == (m.numel(), 3) and rgb.shape == (m.numel(), 3)
        assert torch.allclose(torch.from_numpy(xyz).float(), rows[:, :3], atol=1e-6)
        assert torch.equal(torch.from_numpy(rgb).long(), torch.clamp(torch.round(rows[:, 3:6].double() * 255.0), 0, 255).long())


def test_bench_eval_and_ngp_lines():
    """`bench.py --eval` (forward-only registration, RRE / RTE against the known pose) and `bench.py --ngp` (grid extraction) print one
    JSON line each with the contract's keys."""
    import json
    for flags, metric in ((["--eval", "--res", "64", "--pairs", "1", "--steps", "2", "--warmup", "1"], "nerf_pairs_per_sec_regtr_eval_forward_128"),
                          (["--nerf-labels", "--res", "64", "--pairs", "1", "--steps", "2", "--warmup", "1"], "nerf_pairs_per_sec_regtr_fwd_bwd_128_labels_from_nerf_blocks"),
                          (["--ngp", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], "ngp_grid_extraction_blocks_per_sec_128")):
        out = _run(["bench.py"] + flags, timeout=900)
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out[-2000:]
        d = json.loads(lines[0])
        assert d["metric"] == metric and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["higher_is_better"] is True
        assert "workload" in d["config"] and d["data"] == "synthetic"
        if "--eval" in flags:
            assert 0.0 <= d["rre_deg_mean"] <= 180.0 and d["rte_mean"] >= 0.0
        elif "--ngp" in flags:
            assert d["roofline"]["density_kernel"]["avg_launch_ms"] > 0

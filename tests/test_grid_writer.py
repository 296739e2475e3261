"""CPU: the host side of the evaluation pipeline's writer processes (dreg_nerf_amd/grid_writer.py) and the closed-form rigid transforms of the dataset's
augmentation — no GPU involved.  The files a writer process produces from a shared-memory mapping must be the bytes torch.save gives for an ordinary tensor
with the same values (eval_ngp_nerf.py:350-412 writes them with torch.save / open3d)."""
import filecmp
import multiprocessing as mp
import os

import numpy as np
import pytest
import torch

from dreg_nerf_amd import grid_writer as GW
from dreg_nerf_amd.dataset import _rigid, _rigid_inverse, _small_se3
from dreg_nerf_amd.vis_dump import read_ply, write_ply

SHM = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None


@pytest.mark.skipif(SHM is None, reason="/dev/shm not writable")
def test_files_from_a_shared_mapping_equal_torch_save_of_plain_tensors(tmp_path):
    res, cap, n = 8, 16384, 300
    g = torch.Generator().manual_seed(0)
    grid = torch.zeros(res, res, res, 7)
    grid.view(-1, 7)[::5] = torch.rand(len(range(0, res ** 3, 5)), 7, generator=g)
    seg = GW.Segment(f"{SHM}/dreg_test_{os.getpid()}_g", grid.numel() * 4, create=True)
    off, total = GW.small_layout(cap)
    small = GW.Segment(f"{SHM}/dreg_test_{os.getpid()}_s", total, create=True)
    try:
        seg.tensor(torch.float32, (res, res, res, 7)).copy_(grid)
        world, rgb = torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g)
        dkeep = (torch.rand(n, generator=g) < 0.6).to(torch.uint8)
        keep = dkeep & (torch.rand(n, generator=g) < 0.5).to(torch.uint8)
        dmask, mask = torch.arange(n, dtype=torch.int64)[dkeep.bool()] * 3, torch.arange(n, dtype=torch.int64)[keep.bool()] * 3
        kd, k = dmask.numel(), mask.numel()
        small.tensor(torch.float32, (cap, 3), off["world"])[:n] = world
        small.tensor(torch.float32, (cap, 3), off["rgb"])[:n] = rgb
        small.tensor(torch.int64, (cap,), off["dmask"])[:kd] = dmask
        small.tensor(torch.int64, (cap,), off["mask"])[:k] = mask
        small.tensor(torch.uint8, (cap,), off["dkeep"])[:n] = dkeep
        small.tensor(torch.uint8, (cap,), off["keep"])[:n] = keep
        # through a spawned worker process, as the pipeline runs them
        ctx = mp.get_context("spawn")
        jobs, done = ctx.Queue(), ctx.Queue()
        pr = ctx.Process(target=GW.worker_main, args=(jobs, done), daemon=True)
        pr.start()
        a = tmp_path / "worker"
        a.mkdir()
        jobs.put(("grid", 1, seg.path, seg.nbytes, res, str(a / "voxel_grid.pt")))
        jobs.put(("small", 1, small.path, small.nbytes, cap, n, kd, k, str(a)))
        jobs.put(None)
        msgs = [done.get(timeout=120) for _ in range(3)]
        pr.join(timeout=30)
        assert msgs[0][0] == "ready" and all(m[0] == "done" and m[5] is None for m in msgs[1:]), msgs
        assert msgs[1][3] == grid.numel() * 4
        # the plain form
        b = tmp_path / "plain"
        b.mkdir()
        torch.save(grid.clone(), str(b / "voxel_grid.pt"))
        torch.save(dmask.clone(), str(b / "density_voxel_mask.pt"))
        torch.save(mask.clone(), str(b / "voxel_mask.pt"))
        write_ply(str(b / "density_voxel_point_cloud.ply"), world[dkeep.bool()].numpy(), rgb[dkeep.bool()].numpy())
        write_ply(str(b / "voxel_point_cloud.ply"), world[keep.bool()].numpy(), rgb[keep.bool()].numpy())
        for f in ("voxel_grid.pt", "voxel_mask.pt", "density_voxel_mask.pt", "voxel_point_cloud.ply", "density_voxel_point_cloud.ply"):
            assert filecmp.cmp(str(a / f), str(b / f), shallow=False), f
        assert torch.equal(torch.load(str(a / "voxel_grid.pt")), grid) and torch.equal(torch.load(str(a / "voxel_mask.pt")), mask)
        xyz, col = read_ply(str(a / "voxel_point_cloud.ply"))
        assert xyz.shape == (k, 3) and col.shape == (k, 3)
    finally:
        for s_ in (seg, small):
            os.unlink(s_.path)


def test_small_layout_is_aligned_and_disjoint():
    for cap in (16384, 1 << 20):
        off, total = GW.small_layout(cap)
        spans = sorted((o, o + per * cap) for (name, per), o in zip((("world", 12), ("rgb", 12), ("dmask", 8), ("mask", 8), ("dkeep", 1), ("keep", 1)), (off[n] for n in ("world", "rgb", "dmask", "mask", "dkeep", "keep"))))
        assert all(a % 256 == 0 for a, _ in spans) and all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1)) and spans[-1][1] <= total


def test_closed_form_rigid_inverse_and_composition_match_linalg():
    """dataset.augment_sparse composes / inverts its rigid transforms in closed form (no solver call on the device); conerf/datasets/register/dataset.py:277-331
    does inv(Tc) @ perturb @ Tc and torch.linalg.inv."""
    g = torch.Generator().manual_seed(3)
    for _ in range(5):
        T = _small_se3(0.3, g).float()
        np.testing.assert_allclose(_rigid_inverse(T).numpy(), torch.linalg.inv(T).numpy(), atol=2e-7)
        np.testing.assert_allclose(_rigid_inverse(T[None]).numpy(), torch.linalg.inv(T)[None].numpy(), atol=2e-7)
        c = torch.randn(3, generator=g)
        Tc = torch.eye(4)
        Tc[:3, 3] = -c
        want = torch.linalg.inv(Tc) @ T @ Tc
        R, t = T[:3, :3], T[:3, 3]
        np.testing.assert_allclose(_rigid(R, t + c - R @ c).numpy(), want.numpy(), atol=5e-7)

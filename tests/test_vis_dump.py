"""CPU: the per-scene evaluation outputs (dreg_nerf_amd/vis_dump.py; reference eval_nerf_regtr.py:313-438) — file set, point counts,
colours and the transformed coordinates of the PLY point clouds."""
import json
import os

import numpy as np
import torch

from dreg_nerf_amd import vis_dump


def test_scene_outputs(tmp_path):
    g = torch.Generator().manual_seed(0)
    ns, nt = 37, 29
    src, tgt = torch.randn(ns, 3, generator=g), torch.randn(nt, 3, generator=g)
    pose = torch.eye(4)
    pose[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    pose[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
    pred_pose = pose[:3].clone()
    pred_pose[:3, 3] += 0.01
    ov_s, ov_t = torch.rand(6, ns, 1, generator=g), torch.rand(6, nt, 1, generator=g)
    pred = {"pose": pred_pose[None, None].expand(6, 1, 3, 4).clone(), "src_kp": [src], "tgt_kp": [tgt],
            "src_kp_warped": [torch.randn(6, ns, 3, generator=g)], "tgt_kp_warped": [torch.randn(6, nt, 3, generator=g)],
            "src_overlap": [ov_s], "tgt_overlap": [ov_t]}
    d = str(tmp_path / "scene0")
    cs, ct = torch.randn(5, 4, 4, generator=g), torch.randn(3, 4, 4, generator=g)
    vis_dump.dump_scene_outputs(d, pred, pose[None], cs, ct)
    for f_, T_ in (("unaligned_poses.pt", torch.eye(4)), ("aligned_poses_gt.pt", pose), ("aligned_poses_pred.pt", torch.cat([pred_pose, torch.tensor([[0.0, 0, 0, 1]])]))):
        P = torch.load(os.path.join(d, f_))
        assert P.shape == (8, 4, 4) and torch.allclose(P[:5], T_ @ cs, atol=1e-6) and torch.equal(P[5:], ct)
        os.remove(os.path.join(d, f_))
    names = {"transformation_est.json", "src_xyz.ply", "tgt_xyz.ply", "src_kp_warped.ply", "tgt_kp_warped.ply", "all_src_xyz.ply", "all_tgt_xyz.ply",
             "noisy_point_cloud_pred.ply", "point_cloud_pred.ply", "noisy_point_cloud_gt.ply", "point_cloud_gt.ply"}
    assert set(os.listdir(d)) == names
    T = np.array(json.load(open(os.path.join(d, "transformation_est.json")))["transformation"])
    np.testing.assert_allclose(T[:3], pred_pose.numpy(), atol=1e-7)
    np.testing.assert_allclose(T[3], [0, 0, 0, 1])
    xyz, rgb = vis_dump.read_ply(os.path.join(d, "src_xyz.ply"))
    assert rgb is None
    np.testing.assert_allclose(xyz, src.numpy().astype(np.float64))
    xyz, rgb = vis_dump.read_ply(os.path.join(d, "all_src_xyz.ply"))
    assert xyz.shape == (ns + nt, 3) and (rgb[:ns] == [255, 0, 0]).all() and (rgb[ns:] == [0, 255, 0]).all()
    np.testing.assert_allclose(xyz[ns:], pred["tgt_kp_warped"][0][-1].numpy().astype(np.float64))
    xyz, rgb = vis_dump.read_ply(os.path.join(d, "noisy_point_cloud_gt.ply"))
    np.testing.assert_allclose(xyz[:ns], (src @ pose[:3, :3].T + pose[:3, 3]).numpy(), atol=1e-6)
    assert (rgb == [255, 0, 0]).all()
    keep = (torch.cat([ov_s[-1], ov_t[-1]]) >= 0.5).squeeze(-1).numpy()
    xyz, rgb = vis_dump.read_ply(os.path.join(d, "point_cloud_pred.ply"))
    assert xyz.shape[0] == int(keep.sum()) and (rgb == [0, 255, 0]).all()
    full, _ = vis_dump.read_ply(os.path.join(d, "noisy_point_cloud_pred.ply"))
    np.testing.assert_allclose(xyz, full[keep])
    head = open(os.path.join(d, "all_tgt_xyz.ply"), "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\n") and "property double x" in head and "property uchar red" in head

"""Per-scene outputs of the reference's evaluation besides the metrics (eval_nerf_regtr.py:313-438): the estimated transformation as
JSON and the point clouds of the registration as PLY files with the reference's names and colours.  The reference writes them with
open3d.io.write_point_cloud (absent here); the files below use the layout open3d's writer produces for a point cloud — binary
little-endian, double x / y / z, uchar red / green / blue when coloured — so the same viewers and scripts read them.  The camera-pose
dumps (:330-343) are written when the blocks' NeRF checkpoints (their camera_poses meta data) are on disk; the rendered videos need the
reference's NeRF renderer and are not produced."""
import json
import os

import numpy as np
import torch


def write_ply(path: str, xyz, rgb=None) -> None:
    """xyz [N,3] float; rgb [N,3] in [0,1] (open3d's colour convention) or None."""
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(-1, 3))
    n = xyz.shape[0]
    props = "property double x\nproperty double y\nproperty double z\n"
    if rgb is not None:
        rgb = np.asarray(rgb, dtype=np.float64).reshape(-1, 3)
        assert rgb.shape[0] == n
        props += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
        rec = np.empty(n, dtype=[("p", "<f8", 3), ("c", "u1", 3)])
        rec["p"] = xyz
        rec["c"] = np.clip(np.round(rgb * 255.0), 0, 255).astype(np.uint8)
        body = rec.tobytes()
    else:
        body = xyz.astype("<f8").tobytes()
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment Created by dreg_nerf_amd (open3d point-cloud layout)\n"
                 f"element vertex {n}\n{props}end_header\n").encode("ascii"))
        f.write(body)


def read_ply(path: str):
    """Reader for the files above (tests): returns (xyz float64 [N,3], rgb uint8 [N,3] or None)."""
    with open(path, "rb") as f:
        header = b""
        while not header.endswith(b"end_header\n"):
            header += f.readline()
        lines = header.decode("ascii").splitlines()
        n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        colored = any(l.startswith("property uchar") for l in lines)
        if colored:
            rec = np.frombuffer(f.read(), dtype=[("p", "<f8", 3), ("c", "u1", 3)], count=n)
            return rec["p"].copy(), rec["c"].copy()
        return np.frombuffer(f.read(), dtype="<f8", count=3 * n).reshape(n, 3).copy(), None


def _se3(pose, xyz):
    return xyz @ pose[:3, :3].T + pose[:3, 3]


def dump_camera_poses(out_dir: str, pose4: torch.Tensor, pose_gt4: torch.Tensor, src_cams: torch.Tensor, tgt_cams: torch.Tensor) -> None:
    """eval_nerf_regtr.py:330-343: the two blocks' training cameras [N,4,4] unaligned, aligned by the ground-truth and by the estimated
    transformation (source cameras moved into the target frame)."""
    src_cams, tgt_cams = src_cams.float().cpu(), tgt_cams.float().cpu()
    torch.save(torch.cat([src_cams, tgt_cams], dim=0), os.path.join(out_dir, "unaligned_poses.pt"))
    torch.save(torch.cat([pose_gt4.float().cpu() @ src_cams, tgt_cams], dim=0), os.path.join(out_dir, "aligned_poses_gt.pt"))
    torch.save(torch.cat([pose4.float().cpu() @ src_cams, tgt_cams], dim=0), os.path.join(out_dir, "aligned_poses_pred.pt"))


def dump_scene_outputs(out_dir: str, pred: dict, pose_gt: torch.Tensor, src_cams=None, tgt_cams=None) -> None:
    """pred: the model's output dict for ONE pair (List(B=1) members, pose [6,1,3,4]); pose_gt [1,4,4]; src_cams / tgt_cams: the blocks'
    camera_poses meta data [N,4,4] when their NeRF checkpoints are on disk (then the three pose files are written too)."""
    os.makedirs(out_dir, exist_ok=True)
    pred_pose = pred["pose"][-1][0].detach().float().cpu()            # [3,4]
    pose4 = torch.cat([pred_pose, torch.tensor([[0.0, 0.0, 0.0, 1.0]])])
    if src_cams is not None and tgt_cams is not None:
        dump_camera_poses(out_dir, pose4, pose_gt[0].detach(), torch.as_tensor(src_cams), torch.as_tensor(tgt_cams))
    with open(os.path.join(out_dir, "transformation_est.json"), "w") as f:
        f.write(json.dumps({"transformation": pose4.numpy().tolist()}, indent=4))
    red, green = np.array([[1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0]])
    src, tgt = pred["src_kp"][0].detach().float().cpu(), pred["tgt_kp"][0].detach().float().cpu()
    src_w, tgt_w = pred["src_kp_warped"][0][-1].detach().float().cpu(), pred["tgt_kp_warped"][0][-1].detach().float().cpu()
    write_ply(os.path.join(out_dir, "src_xyz.ply"), src.numpy())
    write_ply(os.path.join(out_dir, "tgt_xyz.ply"), tgt.numpy())
    write_ply(os.path.join(out_dir, "src_kp_warped.ply"), src_w.numpy())
    write_ply(os.path.join(out_dir, "tgt_kp_warped.ply"), tgt_w.numpy())
    two = lambda a, b: np.concatenate([np.repeat(red, a, axis=0), np.repeat(green, b, axis=0)], axis=0)
    write_ply(os.path.join(out_dir, "all_src_xyz.ply"), torch.cat([src, tgt_w]).numpy(), two(src.shape[0], tgt_w.shape[0]))
    write_ply(os.path.join(out_dir, "all_tgt_xyz.ply"), torch.cat([src_w, tgt]).numpy(), two(src_w.shape[0], tgt.shape[0]))
    ov = torch.cat([pred["src_overlap"][0], pred["tgt_overlap"][0]], dim=-2)[-1].detach().float().cpu()   # [Ns+Nt,1], last layer
    keep = (ov >= 0.5).squeeze(-1).numpy()
    xyz_pred = torch.cat([_se3(pred_pose, src), tgt]).numpy()
    write_ply(os.path.join(out_dir, "noisy_point_cloud_pred.ply"), xyz_pred, two(src.shape[0], tgt.shape[0]))
    write_ply(os.path.join(out_dir, "point_cloud_pred.ply"), xyz_pred[keep], np.repeat(green, int(keep.sum()), axis=0))
    xyz_gt = torch.cat([_se3(pose_gt[0].detach().float().cpu(), src), tgt]).numpy()
    write_ply(os.path.join(out_dir, "noisy_point_cloud_gt.ply"), xyz_gt, np.repeat(red, xyz_gt.shape[0], axis=0))
    write_ply(os.path.join(out_dir, "point_cloud_gt.ply"), xyz_gt[keep], np.repeat(red, int(keep.sum()), axis=0))

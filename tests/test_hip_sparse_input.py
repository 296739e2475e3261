"""GPU: the sparse input form (N2 of SURVEY §8f: dataset.SparseBlock, dreg_pack_rgba_sparse) gives the dense path's results bit
for bit; the sparse augmentation equals the reference-form dense augmentation for the same random draws."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import dataset as DS, synth  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402

DEV = "cuda:0"


def _pair(res=64):
    d = synth.shell_pair(res, 3, 4, pose=synth.fixed_pose())
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}


def _sparse_of(d):
    out = {"pose": d["pose"]}
    for s in ("src", "tgt"):
        g = d[s + "_xyz_rgba"][0]                                   # [7,Z,X,Y]
        dense_xyz7 = g.permute(2, 3, 1, 0).contiguous()             # [X,Y,Z,7] = the voxel_grid.pt layout
        out[s + "_sparse"] = DS.SparseBlock.from_dense(dense_xyz7, d[s + "_mask"])
    return out


def test_sparse_blocks_are_lossless_and_give_identical_outputs():
    torch.manual_seed(0)
    m = NeRFRegTr(precision="bf16").to(DEV).eval()
    d = _pair()
    sp = _sparse_of(d)
    for s in ("src", "tgt"):
        assert torch.equal(sp[s + "_sparse"].dense(), d[s + "_xyz_rgba"])
    with torch.no_grad():
        a = m.forward_batch([d])[0]
        b = m.forward_batch([sp])[0]
    for k in ("src_kp", "tgt_kp", "src_kp_warped", "src_overlap", "pose"):
        va, vb = (a[k][0], b[k][0]) if isinstance(a[k], list) else (a[k], b[k])
        assert torch.equal(va, vb), k


def test_sparse_augmentation_matches_dense_reference_form():
    d = _pair(32)
    sp = _sparse_of(d)
    g = torch.Generator().manual_seed(4)
    draws = {"noise_src": torch.randn(d["src_mask"].shape[0], 3, generator=g) * 0.005,
             "noise_tgt": torch.randn(d["tgt_mask"].shape[0], 3, generator=g) * 0.005,
             "perturb": DS._small_se3(0.1, g), "perturb_source": False, "swap": True}
    sp["pose"] = sp["pose"][0]
    out = DS.augment_sparse(sp, draws=draws)
    # the same steps in the reference's dense form (dataset.py:277-331), in fp64 on the CPU
    dd = {k: (v.detach().cpu().double() if torch.is_tensor(v) and v.is_floating_point() else (v.cpu() if torch.is_tensor(v) else v)) for k, v in d.items()}
    flat = {}
    for s in ("src", "tgt"):
        gx = dd[s + "_xyz_rgba"][0, :3].permute(2, 3, 1, 0).reshape(-1, 3).clone()
        gx[dd[s + "_mask"]] += draws["noise_" + s].double()
        flat[s] = gx
    c = flat["tgt"].mean(dim=0)
    Tc = torch.eye(4, dtype=torch.float64)
    Tc[:3, 3] = -c
    P = torch.linalg.inv(Tc) @ draws["perturb"].double() @ Tc
    mt = dd["tgt_mask"]
    flat["tgt"][mt] = flat["tgt"][mt] @ P[:3, :3].T + P[:3, 3]
    pose = P @ dd["pose"][0]
    pose = torch.linalg.inv(pose)          # swap
    np.testing.assert_allclose(out["pose"].cpu().numpy(), pose.numpy(), atol=2e-6)
    # after the swap: new src = old tgt
    np.testing.assert_allclose(out["src_sparse"].vals[:, :3].cpu().numpy(), flat["tgt"][mt].numpy(), atol=2e-6)
    np.testing.assert_allclose(out["tgt_sparse"].vals[:, :3].cpu().numpy(), flat["src"][dd["src_mask"]].numpy(), atol=2e-6)
    assert torch.equal(out["src_sparse"].idx.cpu(), mt)


def test_on_disk_dataset_sparse_mode_equals_dense_mode(tmp_path):
    """NeRFRegDataset over the reference's directory layout: sparse samples (cached voxel_sparse.pt, 100x less H2D) feed the network
    the same values as the dense samples."""
    import json
    import os
    root, jdir = tmp_path / "root", tmp_path / "json"
    scene = "scene_a"
    (root / "objaverse" / "images" / scene).mkdir(parents=True)
    jdir.mkdir()
    json.dump({"train": [scene], "test": [scene]}, open(jdir / "objaverse.json", "w"))
    tf = {}
    for k in range(2):
        g, m = synth.shell_grid(32, 11 + k, 0.5, 0.7)            # [X,Y,Z,7], mask
        bd = root / "objaverse" / "nerf_models" / scene / f"block_{k}"
        bd.mkdir(parents=True)
        torch.save(g, bd / "voxel_grid.pt")
        torch.save(m, bd / "voxel_mask.pt")
        T = torch.eye(4)
        T[:3, 3] = torch.tensor([0.01 * k, 0.0, -0.02 * k])
        tf[str(k)] = T.tolist()
    json.dump(tf, open(root / "objaverse" / "images" / scene / "world_frame_transforms.json", "w"))
    import random
    dense = DS.NeRFRegDataset(str(root), str(jdir), "objaverse", "test")
    sparse = DS.NeRFRegDataset(str(root), str(jdir), "objaverse", "test", sparse=True, device=torch.device(DEV))
    assert len(dense) == len(sparse) == 1
    random.seed(5)
    a = dense[0]
    random.seed(5)
    b = sparse[0]
    assert os.path.exists(root / "objaverse" / "nerf_models" / scene / "block_0" / "voxel_sparse.pt")
    assert a["block_list"] == b["block_list"]
    torch.manual_seed(0)
    m = NeRFRegTr(precision="bf16").to(DEV).eval()
    a = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in a.items()}
    with torch.no_grad():
        oa, ob = m.forward(a), m.forward(b)
    assert torch.equal(oa["pose"], ob["pose"]) and torch.equal(oa["src_kp"][0], ob["src_kp"][0])
    assert torch.allclose(a["pose"], b["pose"].to(a["pose"].device))

"""GPU: the training step with its NeRF-block labels split in two launches (train_step.TrainStep.split_labels — key-point labels marched right behind the
geometry phase, the gradient-free 'tilde' labels marched on the label stream under backward and 'nerf_cont' / 'total' completed there) against the step
with every label in front of the loss.  Reference: train_nerf_regtr.py:186-201 (the 'tilde' scores feed only nerf_cont, which has no gradient: SURVEY.md
quirk Q4).  Same labels, same loss values, same parameter update — bit for bit."""
import os
import shutil
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from dreg_nerf_amd import fused_losses as FL  # noqa: E402
from dreg_nerf_amd import params, synth  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402
from dreg_nerf_amd.train_step import TrainStep  # noqa: E402


def _blocks(n, ncam, seed):
    sys.path.insert(0, ROOT)
    import bench
    return bench.write_generated_blocks(n, ncam, seed)


def _run(split, paths, steps=2):
    m = NeRFRegTr(precision="bf16")
    m.load_state_dict(params.synth_state_dict(0), strict=True)
    m = m.cuda().train()
    ts = TrainStep(m)
    ts.split_labels = split
    with torch.no_grad():
        ts.feature_loss.W.copy_((0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(5))).cuda())
    batch = []
    for i in range(2):
        d = synth.shell_pair(64, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
        d["src_nerf_path"], d["tgt_nerf_path"] = paths[2 * i], paths[2 * i + 1]
        batch.append(d)
    rec = []
    for _ in range(steps):
        out = ts.step(batch)
        torch.cuda.synchronize()
        rec.append(({k: v.detach().float().cpu().clone() for k, v in out["losses"].items()}, float(out["grad_norm"])))
    flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in m.parameters()])
    ts.close()
    return rec, flat


def test_split_label_step_equals_the_step_with_all_labels_in_front_of_the_loss():
    td, paths = _blocks(4, 12, 3)
    try:
        a, pa = _run(False, paths)
        b, pb = _run(True, paths)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    for (la, ga), (lb, gb) in zip(a, b):
        assert set(la) == set(lb) == {"overlap", "nerf_cont", "feature", "corr", "total"}
        for k in la:
            assert torch.equal(la[k], lb[k]), (k, la[k], lb[k])
        assert ga == gb
        assert float(la["total"]) >= float(la["nerf_cont"]) >= 0
    assert torch.equal(pa, pb)


def test_deferred_nerf_cont_is_the_one_call_value():
    """csrc/losses.hip: dreg_reg_point_losses(tilde = NULL) + dreg_nerf_cont_deferred == the one-call form, on random labels."""
    g = torch.Generator().manual_seed(1)
    segs = [(300, 280), (257, 511)]
    from dreg_nerf_amd import attn_ops as A
    from dreg_nerf_amd import losses as LS
    tab = A.ProblemTable(segs, torch.device("cuda"))
    R = sum(a + b for a, b in segs)
    bt = {"cond": torch.randn(6, R, 256, generator=g).cuda(), "corr": torch.randn(6, R, 3, generator=g).cuda(), "ov": torch.rand(6, R, 1, generator=g).cuda(),
          "xyz": torch.randn(R, 3, generator=g).cuda(), "tab": tab}
    gt = (torch.rand(6, R, generator=g) < 0.5).float().cuda()
    tilde = (torch.rand(6, R, generator=g) < 0.5).float().cuda()
    poses = torch.eye(4)[None].repeat(2, 1, 1).cuda()
    fl = LS.InfoNCELoss().cuda()
    one = FL.regtr_losses(bt, poses, fl, gt, tilde)
    d = {}
    two = FL.regtr_losses(bt, poses, fl, gt, None, defer=d)
    assert float(two["nerf_cont"]) == 0.0
    FL.finish_nerf_cont(d, tilde)
    torch.cuda.synchronize()
    for k in one:
        assert torch.equal(one[k].detach(), two[k].detach()), k
    assert float(one["nerf_cont"]) > 0.1
    with pytest.raises(ValueError):
        FL.regtr_losses(bt, poses, fl, gt, None)

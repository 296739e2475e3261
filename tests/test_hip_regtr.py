"""GPU: the assembled network (HIP path, fp32 exact-MFMA mode) against the reference-generated golden
vectors and the oracle; bf16 mode sanity (loose tolerances)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import losses as LS, params, synth  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402


def _to(data, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}


def _model(precision, train):
    m = NeRFRegTr(precision=precision)
    m.load_state_dict(params.synth_state_dict(0), strict=True)
    m = m.cuda()
    m.train(train)
    return m


def test_fpn_and_e2e_eval32_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e_eval32.npz"))
    m = _model("fp32", False)
    data = synth.shell_pair(32, 1, 2, pose=synth.fixed_pose())
    with torch.no_grad():
        p1 = m.fpn(m.pack_grids([data["src_xyz_rgba"].cuda()], torch.float32))
        out = m(_to(data, "cuda"))
    p1_ncdhw = p1.permute(0, 4, 1, 2, 3).contiguous().cpu()
    np.testing.assert_allclose(p1_ncdhw.flatten()[g["p1_idx"]].numpy(), g["p1_val"], rtol=1e-3, atol=2e-4 * float(g["p1_absmean"]))
    assert out["src_kp"][0].shape[0] == int(g["n_src"]) and out["tgt_kp"][0].shape[0] == int(g["n_tgt"])
    np.testing.assert_allclose(out["src_kp"][0].cpu().numpy(), g["src_kp"], atol=1e-6)
    np.testing.assert_allclose(out["src_kp_warped"][0][-1].cpu().numpy(), g["src_kp_warped_last"], atol=2e-4)
    np.testing.assert_allclose(out["tgt_overlap"][0][-1].cpu().numpy(), g["tgt_overlap_last"], atol=2e-4)
    # north-star tolerance: rotation / translation within 1e-4 of the reference on identical inputs
    np.testing.assert_allclose(out["pose"].cpu().numpy(), g["pose"], atol=1e-4)
    rre, rte = LS.rre_rte(out["pose"][-1].cpu(), data["pose"])
    np.testing.assert_allclose(rre.numpy(), g["rre"], atol=1e-2)
    np.testing.assert_allclose(rte.numpy(), g["rte"], atol=1e-4)


# Train-mode forward at 64^3 from the REFERENCE'S OWN initialisation: the pose of this case is ill-conditioned (the same case in bf16 moves the pose by 0.33,
# tests/test_hip_pinned_step.py bf16_64_active_exec; at random init the correspondences are near-uniform averages and the Kabsch solve amplifies
# rounding).  The north-star tolerance 1e-4 is asserted where the problem is well posed: eval-mode e2e_eval32 above (atol 1e-4) and the 128^3 fp32 step of
# tests/test_hip_pinned_step.py (TOL_FP32["pose"] = 1e-4, measured 1.7e-5).  Here the bound is 2x the measured 1.21e-4 (printed by the test).
POSE_TOL_TRAIN64 = 2.5e-4


def test_train_step_64_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "train64.npz"))
    m = _model("fp32", True)
    data = synth.shell_pair(64, 1, 2, pose=synth.fixed_pose())
    pred = m(_to(data, "cuda"))
    assert pred["src_kp"][0].shape[0] == int(g["n_src"])
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
    with torch.no_grad():
        s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(6)])
        t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(6)])
    fl = LS.InfoNCELoss().cuda()
    with torch.no_grad():
        fl.W.copy_((0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(int(g["W_seed"])))).cuda())
    losses = LS.training_losses(pred, data["pose"].cuda(), fl, s_gt, t_gt, s_tl, t_tl, robust=False)
    for k in ("overlap", "nerf_cont", "feature", "corr", "total"):
        # 'feature' thresholds pairwise distances (r_p, r_n): one anchor flipping moves it by ~1/N
        np.testing.assert_allclose(float(losses[k].detach()), float(g["loss_" + k]), rtol=5e-3 if k == "feature" else 1e-3)
    pose_err = float(np.abs(pred["pose"].detach().cpu().numpy() - g["pose"]).max())
    print(f"fp32 train-mode pose at 64^3: max |diff| to the reference-generated golden = {pose_err:.3e}")
    assert pose_err < POSE_TOL_TRAIN64, f"pose differs from the reference's by {pose_err:.3e} (bound {POSE_TOL_TRAIN64:.0e})"
    # the pose head has no backward here (se3.py:89-140 is inside autograd in the reference, but no loss of train_nerf_regtr.py:186-229 reads `pose`):
    # it is returned detached, explicitly — a pose loss added on top must use the correspondences / overlap scores, which do carry gradients
    assert pred["pose"].requires_grad is False and pred["src_kp_warped"][0].requires_grad and pred["src_overlap"][0].requires_grad
    losses["total"].backward()
    named = dict(m.named_parameters())
    groups = {"resnet": "fpn3d.backbone_net.", "fpn_head": "fpn3d.feature_pyramid.",
              "transformer": "transformer_encoder.", "decoder": "correspondence_decoder."}
    for name, pref in groups.items():
        sq = sum(float(p.grad.double().pow(2).sum()) for k, p in named.items() if k.startswith(pref) and p.grad is not None)
        np.testing.assert_allclose(sq ** 0.5, float(g["gnorm64_" + name]), rtol=2e-2)
    # gradient probes: distance to the fp64 truth must be of the order of the fp32 reference's own distance
    for key in g.files:
        if key.startswith("gidx/"):
            k = key[5:]
            got = named[k].grad.flatten().cpu()[g[key]].double().numpy()
            ref32, ref64 = g["gval/" + k].astype(np.float64), g["gval64/" + k]
            scale = np.linalg.norm(ref64)
            err_ref = np.linalg.norm(ref32 - ref64) / scale
            err_got = np.linalg.norm(got - ref64) / scale
            assert err_got <= max(4 * err_ref, 2e-3), (k, err_got, err_ref)
    np.testing.assert_allclose(m.state_dict()["fpn3d.backbone_net.bn1.running_mean"][:16].cpu().numpy(),
                               g["bn_running_mean_probe"], rtol=1e-3, atol=1e-5)


def test_e2e_eval32_bf16_close_to_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e_eval32.npz"))
    m = _model("bf16", False)
    data = synth.shell_pair(32, 1, 2, pose=synth.fixed_pose())
    with torch.no_grad():
        out = m(_to(data, "cuda"))
    assert out["src_kp"][0].shape[0] == int(g["n_src"])
    err = np.abs(out["pose"][-1].cpu().numpy() - g["pose"][-1]).max()
    assert err < 0.05, err


def _head_modes(dense_kernel):
    """(key-point predictions, pose, gradients) of the dense and the active-set head on one pair; dense_kernel: 'igemm' forces the
    implicit-GEMM kernel on the dense path (the measurement build: include/dreg_nerf_probe.h), 'halo' is the product default."""
    from dreg_nerf_amd import lib as L
    data = synth.shell_pair(64, 1, 2, pose=synth.fixed_pose())
    res = {}
    import contextlib
    with (L.probe() if dense_kernel == "igemm" else contextlib.nullcontext()) as pr:
        if pr is not None:
            pr.set("dreg_conv3_halo_set_variant", -1, 0)
        for mode in (False, True):
            m = _model("bf16", True)
            m.active_set = mode
            pred = m(_to(data, "cuda"))
            loss = (pred["src_kp_warped"][0] ** 2).sum() + pred["tgt_overlap"][0].sum() + (pred["src_feats"][0][-1] ** 2).mean()
            loss.backward()
            named = dict(m.named_parameters())
            res[mode] = (pred["src_kp_warped"][0].detach().cpu(), pred["pose"].cpu(),
                         {k: named[k].grad.float().cpu() for k in ("fpn3d.feature_pyramid.upsample_transform_1.weight",
                                                                  "fpn3d.feature_pyramid.pyramid_transformation_1.weight",
                                                                  "fpn3d.feature_pyramid.pyramid_transformation_1.bias",
                                                                  "fpn3d.backbone_net.conv1.weight",
                                                                  "fpn3d.backbone_net.layer2.0.conv2.weight")})
    return res


def test_active_set_head_equals_dense_head():
    """bf16: evaluating the two FPN head convolutions on the active set (rows around the occupied voxels) gives the same
    outputs and parameter gradients as the dense 64^3 evaluation.  With the SAME convolution kernel on both paths the forward
    is bit-identical: the row lists leave out nothing that is consumed."""
    res = _head_modes("igemm")
    assert torch.equal(res[False][0], res[True][0])       # forward: bit-identical key-point predictions
    assert torch.equal(res[False][1], res[True][1])
    # gradients: the head's own parameters agree to reduction-order noise; deep ResNet gradients additionally see the
    # run-to-run noise of the fp32 atomics in the trilinear-gather backward amplified by train-mode BatchNorm
    for k in res[False][2]:
        a, b = res[False][2][k], res[True][2][k]
        tol = 2e-3 if "feature_pyramid" in k else 1e-1
        assert (a - b).norm() <= tol * a.norm() + 1e-6, (k, float((a - b).norm()), float(a.norm()))


def test_active_set_head_vs_halo_kernel_dense_head():
    """The product's dense path runs the 3^3 head convolutions on the halo kernel, which accumulates the same products in another
    order (32-channel chunks outermost): P1 differs from the row-list kernels' by fp32 round-off before its single bf16 rounding, and
    the network's outputs follow within bf16 noise."""
    res = _head_modes("halo")
    d = (res[False][0] - res[True][0]).abs().max().item()
    assert d <= 2e-2 * res[False][0].abs().max().item(), d
    assert (res[False][1] - res[True][1]).abs().max().item() < 5e-2
    for k in res[False][2]:
        a, b = res[False][2][k], res[True][2][k]
        tol = 2e-2 if "feature_pyramid" in k else 3e-1
        assert (a - b).norm() <= tol * a.norm() + 1e-6, (k, float((a - b).norm()), float(a.norm()))


def test_full_size_batch_invariance_128():
    """BASELINE size (128^3, bf16, train-mode BatchNorm): a pair's outputs do not depend on which other pairs share the
    step — the reference runs one grid per BatchNorm call (nerf_regtr.py:135) and the batched kernels must keep that."""
    m = _model("bf16", True)
    a = _to(synth.shell_pair(128, 1, 2, pose=synth.fixed_pose()), "cuda")
    b = _to(synth.shell_pair(128, 3, 4, pose=synth.fixed_pose()), "cuda")
    with torch.no_grad():
        alone = m.forward_batch([a])[0]
        both = m.forward_batch([b, a])[1]
    assert alone["src_kp"][0].shape[0] > 1000 and alone["src_kp"][0].shape == both["src_kp"][0].shape
    assert torch.equal(alone["src_kp"][0], both["src_kp"][0])
    # the trilinear-gather outputs are bit-identical; the voxel means / attention see the same values in the same order
    assert torch.equal(alone["pose"], both["pose"])
    assert torch.equal(alone["src_overlap"][0], both["src_overlap"][0])


def test_training_step_with_learned_position_embedding():
    """The non-default embedding (nerf_regtr.py:89-90) in the benchmarked configuration (bf16, executor, fused losses, flat AdamW):
    its parameters are part of the flat buffers, get gradients through the LayerNorm(+pe) kernels and move."""
    from dreg_nerf_amd.train_step import TrainStep
    torch.manual_seed(5)
    m = NeRFRegTr("learned", 256, 1.0, precision="bf16").to("cuda").train()
    ts = TrainStep(m)
    d = synth.shell_pair(64, 1, 2, pose=synth.fixed_pose())
    batch = [{k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in d.items()}]
    w0 = {k: v.detach().clone() for k, v in m.named_parameters() if k.startswith("pos_embed.")}
    assert len(w0) == 10
    losses = [float(ts.step(batch)["losses"]["total"]) for _ in range(3)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    for k, v in m.named_parameters():
        if k.startswith("pos_embed."):
            assert torch.isfinite(v).all() and not torch.equal(v.detach(), w0[k]), k
    sd = m.state_dict()
    assert sd["correspondence_decoder.pos_embed.mlp.0.weight"].data_ptr() == sd["pos_embed.mlp.0.weight"].data_ptr()


def test_global_subsample_plan_equals_the_per_pair_plans():
    """The voxel-average rounds planned for ALL pairs of a step at once (NeRFRegTr.global_subsample: one set of launches and one host sync per round; pairs that
    have stopped — grid_downsample.py:83-94 — pass through the later rounds) against one plan per pair: same key points, same predictions, same parameter
    gradients.  Three pairs with different point counts, so that they stop after different numbers of rounds."""
    from dreg_nerf_amd.train_step import TrainStep  # noqa: F401
    pose = synth.fixed_pose()
    batch = []
    for i, (r1, res) in enumerate(((0.83, 128), (0.90, 128), (0.815, 128))):       # ~19 k, ~50 k+, ~10 k occupied voxels per side
        gs, ms = synth.shell_grid(res, 1 + 2 * i, 0.8, r1)
        gt, mt = synth.shell_grid(res, 2 + 2 * i, 0.8, r1, pose=pose)
        batch.append({"src_xyz_rgba": gs.permute(3, 2, 0, 1).unsqueeze(0).contiguous().cuda(), "tgt_xyz_rgba": gt.permute(3, 2, 0, 1).unsqueeze(0).contiguous().cuda(),
                      "src_mask": ms.cuda(), "tgt_mask": mt.cuda(), "pose": pose[None].clone().cuda(), "src_nerf_path": "", "tgt_nerf_path": ""})
    res = {}
    for mode in (False, True):
        m = _model("bf16", True)
        m.native_trunk = False            # (per-op trunk: the comparison is about the point sets; the executor needs FlatAdamW's gradient buffers)
        m.global_subsample = mode
        preds = m.forward_batch(batch)
        nrounds = [len(r) for r in m.__dict__.get("_last_plans", [])]
        loss = sum((p["src_kp_warped"][0] ** 2).sum() + p["tgt_overlap"][0].sum() + (p["src_feats"][0][-1] ** 2).mean() for p in preds)
        loss.backward()
        named = dict(m.named_parameters())
        res[mode] = ([(p["src_kp"][0].cpu(), p["tgt_kp"][0].cpu(), p["src_kp_warped"][0].detach().cpu(), p["pose"].cpu()) for p in preds],
                     {k: named[k].grad.float().cpu() for k in ("fpn3d.feature_pyramid.upsample_transform_1.weight", "fpn3d.backbone_net.conv1.weight",
                                                              "transformer_encoder.layers.0.linear1.weight", "correspondence_decoder.q_proj.weight")}, nrounds)
    a, b = res[False], res[True]
    assert len(a[2]) == 3 and len(set(a[2])) > 1, f"the pairs must stop after different numbers of rounds: {a[2]}"
    assert len(b[2]) == 1 and b[2][0] == max(a[2])
    for pa, pb in zip(a[0], b[0]):
        for x, y in zip(pa, pb):
            assert x.shape == y.shape and torch.equal(x, y)
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k

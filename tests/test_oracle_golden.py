"""CPU: the oracle (oracle/regtr_oracle.py) against vectors produced by the reference's own
Python (tools/make_golden.py).  Tolerances are fp32 round-off class."""
import os

import numpy as np
import pytest
import torch

from dreg_nerf_amd import params, synth
from oracle import regtr_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_param_inventory():
    spec = params.regtr_spec()
    assert len(spec) == 772
    assert params.num_parameters() == 61124225
    sd = params.synth_state_dict(0)
    assert sd["fpn3d.feature_pyramid.resnet.conv1.weight"] is sd["fpn3d.backbone_net.conv1.weight"]


def test_small_ops(golden_dir):
    g = _load(golden_dir, "small_ops.npz")
    pe = O.posenc_sine(torch.from_numpy(g["pe_xyz"]))
    np.testing.assert_allclose(pe.numpy(), g["pe"], atol=1e-6)
    T = O.weighted_kabsch(torch.from_numpy(g["kab_a"]), torch.from_numpy(g["kab_b"]), torch.from_numpy(g["kab_w"]))
    np.testing.assert_allclose(T.numpy(), g["kab_T"], atol=2e-6)
    gt = torch.from_numpy(g["pose_gt"])[None].expand(6, -1, -1)
    rre, rte = O.rre_rte(T, gt)
    np.testing.assert_allclose(rre.numpy(), g["rre"], atol=1e-4)
    np.testing.assert_allclose(rte.numpy(), g["rte"], atol=1e-6)


def test_transformer_decoder(golden_dir):
    g = _load(golden_dir, "transformer.npz")
    sd = params.synth_state_dict(0)
    s_xyz, t_xyz = torch.from_numpy(g["s_xyz"]), torch.from_numpy(g["t_xyz"])
    s_pe, t_pe = O.posenc_sine(s_xyz), O.posenc_sine(t_xyz)
    with torch.no_grad():
        sc, tc = O.cross_encoder(sd, torch.from_numpy(g["s_f"]), torch.from_numpy(g["t_f"]), s_pe, t_pe)
        s_corr, t_corr, s_ov, t_ov = O.corr_decoder(sd, sc, tc, s_xyz, t_xyz, s_pe, t_pe)
    np.testing.assert_allclose(sc.numpy(), g["s_cond"], atol=2e-5)
    np.testing.assert_allclose(tc.numpy(), g["t_cond"], atol=2e-5)
    np.testing.assert_allclose(s_corr.numpy(), g["s_corr"], atol=2e-5)
    np.testing.assert_allclose(t_corr.numpy(), g["t_corr"], atol=2e-5)
    np.testing.assert_allclose(s_ov.numpy(), g["s_ov"], atol=2e-6)
    np.testing.assert_allclose(t_ov.numpy(), g["t_ov"], atol=2e-6)


def test_non_default_position_embeddings(golden_dir):
    """nerf_regtr.py:87-90: the learned MLP embedding (through encoder + decoder, with the gradients of its parameters) and the
    sine embedding with a coordinate scale, against vectors of the reference (tools/make_golden.py pos_embed)."""
    g = _load(golden_dir, "pos_embed.npz")
    spec = params.regtr_spec("learned")
    assert [k for k in spec if "pos_embed" in k][:2] == ["pos_embed.mlp.0.weight", "pos_embed.mlp.0.bias"]
    sd = params.synth_state_dict(0, "learned")
    assert sd["correspondence_decoder.pos_embed.mlp.8.weight"] is sd["pos_embed.mlp.8.weight"]
    for i in range(5):
        sd[f"pos_embed.mlp.{2 * i}.weight"].requires_grad_(True)
        sd[f"pos_embed.mlp.{2 * i}.bias"].requires_grad_(True)
    s_xyz, t_xyz = torch.from_numpy(g["s_xyz"]), torch.from_numpy(g["t_xyz"])
    np.testing.assert_allclose(O.posenc_sine(s_xyz, scale=0.5).numpy(), g["sine_scale_half"], atol=2e-6)
    s_pe, t_pe = O.posenc_learned(sd, s_xyz), O.posenc_learned(sd, t_xyz)
    np.testing.assert_allclose(s_pe.detach().numpy(), g["s_pe"], atol=1e-6)
    sc, tc = O.cross_encoder(sd, torch.from_numpy(g["s_f"]), torch.from_numpy(g["t_f"]), s_pe, t_pe)
    s_corr, t_corr, s_ov, t_ov = O.corr_decoder(sd, sc, tc, s_xyz, t_xyz, s_pe, t_pe)
    corr, ov = torch.cat([s_corr, t_corr], dim=1), torch.cat([s_ov, t_ov], dim=1)
    np.testing.assert_allclose(sc.detach().numpy(), g["s_cond"], atol=2e-5)
    np.testing.assert_allclose(corr.detach().numpy(), g["corr"], atol=2e-5)
    np.testing.assert_allclose(ov.detach().numpy(), g["ov"], atol=2e-6)
    loss = (corr * torch.from_numpy(g["w_corr"])).sum() + (ov * torch.from_numpy(g["w_ov"])).sum()
    loss.backward()
    for i in range(5):
        gw = sd[f"pos_embed.mlp.{2 * i}.weight"].grad.numpy()
        np.testing.assert_allclose(gw, g[f"g_w{i}"], atol=2e-4 * max(1.0, float(np.abs(g[f"g_w{i}"]).max())))
        np.testing.assert_allclose(sd[f"pos_embed.mlp.{2 * i}.bias"].grad.numpy(), g[f"g_b{i}"], atol=2e-4 * max(1.0, float(np.abs(g[f"g_b{i}"]).max())))


def test_trilinear_gather_direct_matches_interpolate():
    g = torch.Generator().manual_seed(3)
    p1 = torch.randn(1, 8, 5, 6, 7, generator=g)
    res = (10, 12, 14)
    n = res[0] * res[1] * res[2]
    mask = torch.randperm(n, generator=g)[:200].sort().values
    xyz = torch.zeros(1, 3, *res)
    _, ref = O.upsample_gather(p1, xyz, mask)
    got = O.trilinear_gather_direct(p1, res, mask)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5)


def test_grid_subsample_known_answers():
    """A4 known-answer cases (hand-made; MinkowskiEngine is unpinned upstream)."""
    pts = torch.tensor([[0.01, 0.01, 0.01], [0.04, 0.02, 0.03], [0.051, 0.0, 0.0],
                        [-0.01, 0.0, 0.0], [0.01, 0.01, 0.01]])
    f = torch.arange(10, dtype=torch.float32).view(5, 2)
    p, ff, l = O.grid_subsample(pts, f, torch.tensor([4, 1]), 0.05)
    assert l.tolist() == [3, 1]
    # batch 0: cells (-1,0,0), (0,0,0) [two rows averaged], (1,0,0); batch 1: (0,0,0)
    np.testing.assert_allclose(p[0].numpy(), [-0.01, 0, 0], atol=1e-7)
    np.testing.assert_allclose(p[1].numpy(), [0.025, 0.015, 0.02], atol=1e-7)
    np.testing.assert_allclose(ff[1].numpy(), [1.0, 2.0], atol=1e-7)
    np.testing.assert_allclose(p[3].numpy(), [0.01, 0.01, 0.01], atol=1e-7)


def test_e2e_eval32(golden_dir):
    g = _load(golden_dir, "e2e_eval32.npz")
    sd = params.synth_state_dict(0)
    data = synth.shell_pair(32, 1, 2, pose=synth.fixed_pose())
    assert data["src_mask"].shape[0] == 1040
    with torch.no_grad():
        p1 = O.fpn_forward(sd, data["src_xyz_rgba"][:, 3:], train=False)
        out = O.regtr_forward(sd, data, train=False)
    np.testing.assert_allclose(p1.flatten()[g["p1_idx"]].numpy(), g["p1_val"], rtol=1e-4, atol=1e-4 * float(g["p1_absmean"]))
    assert out["src_kp"][0].shape[0] == int(g["n_src"])
    np.testing.assert_allclose(out["src_kp"][0].numpy(), g["src_kp"], atol=1e-6)
    np.testing.assert_allclose(out["src_kp_warped"][0][-1].numpy(), g["src_kp_warped_last"], atol=1e-4)
    np.testing.assert_allclose(out["tgt_overlap"][0][-1].numpy(), g["tgt_overlap_last"], atol=1e-4)
    np.testing.assert_allclose(out["pose"].numpy(), g["pose"], atol=1e-4)


@pytest.mark.timeout(600)
def test_train_step_64(golden_dir):
    g = _load(golden_dir, "train64.npz")
    sd = params.synth_state_dict(0)
    for k, (shape, kind) in params.regtr_spec().items():
        if not params.is_buffer(kind) and not k.startswith(params.ALIAS_DST):
            sd[k].requires_grad_(True)
    data = synth.shell_pair(64, 1, 2, pose=synth.fixed_pose())
    pred = O.regtr_forward(sd, data, train=True)
    assert pred["src_kp"][0].shape[0] == int(g["n_src"])
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
    with torch.no_grad():
        s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(6)])
        t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(6)])
    W = 0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(int(g["W_seed"])))
    losses = O.training_losses(pred, data["pose"], W, s_gt, t_gt, s_tl, t_tl, robust=False)
    for k in ("overlap", "nerf_cont", "feature", "corr", "total"):
        np.testing.assert_allclose(float(losses[k]), float(g["loss_" + k]), rtol=2e-4)
    np.testing.assert_allclose(pred["pose"].detach().numpy(), g["pose"], atol=2e-4)
    losses["total"].backward()
    groups = {"resnet": "fpn3d.backbone_net.", "fpn_head": "fpn3d.feature_pyramid.",
              "transformer": "transformer_encoder.", "decoder": "correspondence_decoder."}
    for name, pref in groups.items():
        sq = sum(float(v.grad.double().pow(2).sum()) for k, v in sd.items()
                 if k.startswith(pref) and not k.startswith(params.ALIAS_DST) and v.grad is not None)
        np.testing.assert_allclose(sq ** 0.5, float(g["gnorm_" + name]), rtol=2e-3)
    for key in g.files:
        if key.startswith("gidx/"):
            k = key[5:]
            got = sd[k].grad.flatten()[g[key]].numpy()
            ref = g["gval/" + k]
            np.testing.assert_allclose(got, ref, rtol=5e-3, atol=5e-3 * np.abs(ref).max())
    np.testing.assert_allclose(sd["fpn3d.backbone_net.bn1.running_mean"][:16].numpy(),
                               g["bn_running_mean_probe"], rtol=1e-4, atol=1e-6)

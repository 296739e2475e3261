#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference, which exists only in the build container) on seeded inputs.

Nothing of the reference travels: only inputs' seeds and expected outputs (data) are stored.
Third-party native packages the reference imports but this image lacks (MinkowskiEngine,
tinycudann, nerfacc, torch_scatter, trimesh, ...) are replaced by empty mock modules so the
torch-only parts import; the one MinkowskiEngine call on the path
(conerf/register/grid_downsample.py:24-36) is replaced by oracle.regtr_oracle.grid_subsample,
so row A4 stays "parity unpinned" (SURVEY.md §8(c)).

Usage:  python tools/make_golden.py            (writes tests/golden/)
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from dreg_nerf_amd import params, synth  # noqa: E402
from oracle import regtr_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present: golden vectors can only be generated in the build container")
    for name in ["MinkowskiEngine", "torch_scatter", "nerfacc", "nerfacc.cuda", "nerfacc.contraction",
                 "nerfacc.grid", "nerfacc.intersection", "nerfacc.vol_rendering", "nerfacc.pack",
                 "tinycudann", "trimesh", "cv2", "matplotlib", "matplotlib.backends",
                 "matplotlib.backends.backend_agg", "matplotlib.figure", "matplotlib.cm",
                 "robust_loss_pytorch", "robust_loss_pytorch.general", "tensorboardX", "visdom",
                 "open3d", "imageio", "imageio.v2", "lpips", "sklearn", "sklearn.cluster", "easydict", "scipy.spatial.transform"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = mock.MagicMock(name=name)
    sys.path.insert(0, REF)
    import conerf.register.grid_downsample as gd

    def patched(points, features, batched_lengths, sample_dl=0.1):
        p, f, l = O.grid_subsample(points, features, batched_lengths, sample_dl)
        return torch.cat([p, f], dim=-1), l

    gd.batched_grid_subsample = patched
    import conerf.register.nerf_regtr as nr
    return nr


def sample_idx(numel: int, n: int, seed: int) -> np.ndarray:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (n,), generator=g).numpy()


def ref_model(nr, sd):
    m = nr.NeRFRegTr()
    keys = list(m.state_dict().keys())
    assert keys == list(params.regtr_spec().keys()), "state_dict key order differs from params.regtr_spec()"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == params.regtr_spec()[k][0], k
    m.load_state_dict(params.clone_state_dict(sd), strict=True)
    assert sum(p.numel() for p in m.parameters()) == params.num_parameters() == 61124225
    return m


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if sys.argv[1:] and all(a.startswith("bf16emu") for a in sys.argv[1:]):     # oracle-only products: the reference is not needed
        for a in sys.argv[1:]:
            wc = a.endswith("_wc")
            r = a[7:-3] if wc else a[7:]
            bf16_yardstick(int(r), "train" + r + ("_wc" if wc else ""), "wc" if wc else "default")
        return
    nr = import_reference()
    from conerf.register.se3 import compute_rigid_transform
    from conerf.register.position_embedding import PositionEmbeddingCoordsSine
    from conerf.loss.feature_loss import InfoNCELoss
    import train_nerf_regtr as tr  # evaluate_camera_alignment / rotation_distance only

    sd = params.synth_state_dict(seed=0)

    # ---------------------------------------------------------------- small components
    g = torch.Generator().manual_seed(11)
    xyz = (torch.rand(37, 3, generator=g) - 0.5) * 3
    pe = PositionEmbeddingCoordsSine(3, 256, scale=1.0)(xyz)
    a = torch.randn(6, 50, 3, generator=g)
    Rgt = synth.fixed_pose()
    b = a @ Rgt[:3, :3].T + Rgt[:3, 3] + 0.01 * torch.randn(6, 50, 3, generator=g)
    w = torch.rand(6, 50, generator=g)
    kab = compute_rigid_transform(a, b, w)
    pred = kab[:, :3, :]
    gt = Rgt[None].expand(6, -1, -1)
    rre = tr.rotation_distance(pred[..., :3, :3], gt[..., :3, :3])
    rte = (pred[..., :3, 3] - gt[..., :3, 3]).norm(dim=-1)
    np.savez(os.path.join(OUT, "small_ops.npz"),
             pe_xyz=xyz.numpy(), pe=pe.numpy(),
             kab_a=a.numpy(), kab_b=b.numpy(), kab_w=w.numpy(), kab_T=kab.numpy(),
             rre=rre.numpy(), rte=rte.numpy(), pose_gt=Rgt.numpy())
    print("small_ops done")

    # ---------------------------------------------------------------- transformer + decoder on random features
    m = ref_model(nr, sd).eval()
    g = torch.Generator().manual_seed(12)
    ns, nt = 70, 55
    s_xyz = (torch.rand(ns, 3, generator=g) - 0.5) * 2
    t_xyz = (torch.rand(nt, 3, generator=g) - 0.5) * 2
    s_f = torch.randn(ns, 256, generator=g)
    t_f = torch.randn(nt, 256, generator=g)
    with torch.no_grad():
        s_pe, t_pe = m.pos_embed(s_xyz), m.pos_embed(t_xyz)
        sc, tc = m.transformer_encoder(
            src=s_f[:, None], tgt=t_f[:, None],
            src_key_padding_mask=torch.zeros(1, ns, dtype=torch.bool),
            tgt_key_padding_mask=torch.zeros(1, nt, dtype=torch.bool),
            src_pos=s_pe[:, None], tgt_pos=t_pe[:, None])
        scl, tcl, sol, tol = m.correspondence_decoder(sc, tc, [s_xyz], [t_xyz])
    np.savez(os.path.join(OUT, "transformer.npz"),
             s_xyz=s_xyz.numpy(), t_xyz=t_xyz.numpy(), s_f=s_f.numpy(), t_f=t_f.numpy(),
             s_cond=sc[:, :, 0].numpy(), t_cond=tc[:, :, 0].numpy(),
             s_corr=scl[0].numpy(), t_corr=tcl[0].numpy(), s_ov=sol[0].numpy(), t_ov=tol[0].numpy())
    print("transformer done")

    # ---------------------------------------------------------------- FPN, eval mode, 32^3 (config 1) + e2e forward
    data = synth.shell_pair(32, 1, 2, pose=synth.fixed_pose())
    with torch.no_grad():
        p1 = m.fpn3d(data["src_xyz_rgba"][:, 3:])
        out = m({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()})
    idx = sample_idx(p1.numel(), 2048, 21)
    rr, rt = tr.rotation_distance(out["pose"][-1][:, :3, :3], data["pose"][:, :3, :3]), \
        (out["pose"][-1][:, :3, 3] - data["pose"][:, :3, 3]).norm(dim=-1)
    np.savez(os.path.join(OUT, "e2e_eval32.npz"),
             p1_idx=idx, p1_val=p1.flatten()[idx].numpy(), p1_absmean=p1.abs().mean().numpy(),
             pose=out["pose"].numpy(),
             n_src=out["src_kp"][0].shape[0], n_tgt=out["tgt_kp"][0].shape[0],
             src_kp=out["src_kp"][0].numpy(), tgt_kp=out["tgt_kp"][0].numpy(),
             src_kp_warped_last=out["src_kp_warped"][0][-1].numpy(),
             tgt_kp_warped_last=out["tgt_kp_warped"][0][-1].numpy(),
             src_overlap_last=out["src_overlap"][0][-1].numpy(),
             tgt_overlap_last=out["tgt_overlap"][0][-1].numpy(),
             src_feats_last_absmean=out["src_feats"][0][-1].abs().mean().numpy(),
             rre=rr.numpy(), rte=rt.numpy())
    print("e2e_eval32 done; N =", out["src_kp"][0].shape[0], out["tgt_kp"][0].shape[0])

    want = set(sys.argv[1:]) or {"small", "augment", "train64", "train128", "bf16emu64", "bf16emu128", "train64_wc", "train128_wc", "bf16emu64_wc", "bf16emu128_wc"}
    if "augment" in want:
        augment_golden()
    if "train64" in want:
        train_golden(nr, sd, 64, "train64")
    if "train128" in want:
        # BASELINE size: two A4 rounds, split-K thresholds and 32-bit index ranges of the 128^3 path (reference fwd+bwd: ~2 min on
        # 8 cores; the fp64 oracle pass needs ~40 GB and ~20 min: DREG_GOLDEN_FP64_128=0 skips it)
        train_golden(nr, sd, 128, "train128", with_fp64=bool(int(os.environ.get("DREG_GOLDEN_FP64_128", "1"))))
    # the same step on the WELL-CONDITIONED weight profile (params.PROFILES["wc"], tools/wc_profile_sweep.py): the fixtures the bf16
    # build is bounded against in absolute terms
    sd_wc = params.synth_state_dict(seed=0, profile="wc")
    if "train64_wc" in want:
        train_golden(nr, sd_wc, 64, "train64_wc", profile="wc")
    if "train64_wc_b" in want:
        # a SECOND pinned step (other grids, other relative pose, other weight and W seeds): the absolute bounds of the bf16 build are
        # not tuned to one sample
        train_golden(nr, params.synth_state_dict(seed=1, profile="wc"), 64, "train64_wc_b", profile="wc", sample=(7, 8, 1, 1, 9))
    if "train128_wc" in want:
        train_golden(nr, sd_wc, 128, "train128_wc", with_fp64=bool(int(os.environ.get("DREG_GOLDEN_FP64_128", "1"))), profile="wc")
    if "train128_wc_b" in want:
        # ... and at the BASELINE size (same second sample; with the fp64 oracle pass: ~40 GB, ~20 min)
        train_golden(nr, params.synth_state_dict(seed=1, profile="wc"), 128, "train128_wc_b", with_fp64=bool(int(os.environ.get("DREG_GOLDEN_FP64_128B", "1"))),
                     profile="wc", sample=(7, 8, 1, 1, 9))
    if "bf16emu64_wc" in want:
        bf16_yardstick(64, "train64_wc", "wc")
    if "bf16emu128_wc" in want:
        bf16_yardstick(128, "train128_wc", "wc")
    if "bf16emu64" in want:
        bf16_yardstick(64, "train64")
    if "bf16emu128" in want:
        bf16_yardstick(128, "train128")


def augment_golden():
    """Training-time augmentation of the REFERENCE's dataset class (conerf/datasets/register/dataset.py:277-331: points_jitter,
    rigid_perturb, random_swap, in __getitem__'s order) on small shell grids, with the random draws it consumed (torch.randn for the
    jitter, numpy for the SE(3) perturbation, random.random for the two coin flips) recorded next to its outputs."""
    import random
    import types as _types
    import conerf.datasets.register.dataset as RD
    out, seen = {}, set()
    case = 0
    for seed in range(3, 60):
        random.seed(seed)
        r1, r2 = random.random(), random.random()
        combo = (r1 > 0.5, r2 > 0.5)
        if combo in seen:
            continue
        seen.add(combo)
        res = 16
        gs, ms = synth.shell_grid(res, 2 * seed + 1, 0.5, 0.9)
        gt, mt = synth.shell_grid(res, 2 * seed + 2, 0.5, 0.9, pose=synth.fixed_pose())
        Ts, Tt = torch.eye(4), synth.fixed_pose()
        self = _types.SimpleNamespace(scale=0.005, clip=0.05, std=0.1)
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        src = gs.permute(3, 2, 0, 1).unsqueeze(0).clone()
        tgt = gt.permute(3, 2, 0, 1).unsqueeze(0).clone()
        src[:, :3] = RD.NeRFRegDataset.points_jitter(self, src[:, :3], ms)
        tgt[:, :3] = RD.NeRFRegDataset.points_jitter(self, tgt[:, :3], mt)
        data = {"src_xyz_rgba": src, "tgt_xyz_rgba": tgt, "src_mask": ms, "tgt_mask": mt, "src_nerf_path": "s", "tgt_nerf_path": "t",
                "pose": Tt @ torch.linalg.inv(Ts)}
        data = RD.NeRFRegDataset.rigid_perturb(self, data)
        data = RD.NeRFRegDataset.random_swap(self, data)
        # the draws, replayed
        torch.manual_seed(seed)
        noise_src = torch.randn(ms.shape[0], 3) * 0.005
        noise_tgt = torch.randn(mt.shape[0], 3) * 0.005
        np.random.seed(seed)
        phi, cos_theta = np.random.uniform(0.0, 2 * np.pi), np.random.uniform(-1.0, 1.0)
        theta_n, trans_n = np.random.randn(), np.random.randn(3, 1)
        np.random.seed(seed)
        perturb = RD._sample_se3_small(0.1)
        flat = lambda g_: g_[0, :3].permute(2, 3, 1, 0).reshape(-1, 3)
        pre = f"c{case}/"
        out.update({pre + "grid_seed": 2 * seed + 1, pre + "res": res, pre + "noise_src": noise_src.numpy(), pre + "noise_tgt": noise_tgt.numpy(),
                    pre + "phi": phi, pre + "cos_theta": cos_theta, pre + "theta_n": theta_n, pre + "trans_n": trans_n[:, 0],
                    pre + "perturb": perturb, pre + "perturb_source": bool(r1 > 0.5), pre + "swap": bool(r2 > 0.5),
                    pre + "pose": data["pose"].numpy(), pre + "src_mask": data["src_mask"].numpy(), pre + "tgt_mask": data["tgt_mask"].numpy(),
                    pre + "src_xyz": flat(data["src_xyz_rgba"])[data["src_mask"]].numpy(), pre + "tgt_xyz": flat(data["tgt_xyz_rgba"])[data["tgt_mask"]].numpy(),
                    pre + "src_path": data["src_nerf_path"]})
        case += 1
        if len(seen) == 4:
            break
    out["n_cases"] = case
    np.savez(os.path.join(OUT, "augment.npz"), **out)
    print("augment done:", case, "cases", sorted(seen))


def bf16_yardstick(res: int, name: str, profile="default"):
    """The reference-pinned oracle evaluated with bf16 operand rounding (oracle.regtr_oracle.EMULATE = "bf16") on the same step as
    `name`.npz: how far bf16 rounding ALONE moves losses / pose / gradient probes / gradient norms / the optimizer's parameter delta.
    Written to `name`_bf16emu.npz; tests/test_hip_pinned_step.py bounds the bf16 build's distance to the truth by this distance."""
    base = np.load(os.path.join(OUT, name + ".npz"))
    sd = params.synth_state_dict(0, profile=profile)
    leaves = {}
    for k, (shape, kind) in params.regtr_spec().items():
        if not params.is_buffer(kind) and not k.startswith(params.ALIAS_DST):
            sd[k].requires_grad_(True)
            leaves[k] = sd[k]
    data = synth.shell_pair(res, 1, 2, pose=synth.fixed_pose())
    W = 0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(int(base["W_seed"])))
    O.EMULATE = "bf16"
    try:
        pred = O.regtr_forward(sd, data, train=True)
        s_gt, t_gt = synth.synthetic_overlap_gt(pred["src_kp"][0]), synth.synthetic_overlap_gt(pred["tgt_kp"][0])
        with torch.no_grad():
            s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(6)])
            t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(6)])
        losses = O.training_losses(pred, data["pose"], W, s_gt, t_gt, s_tl, t_tl)
        losses["total"].backward()
    finally:
        O.EMULATE = None
    groups = {"resnet": "fpn3d.backbone_net.", "fpn_head": "fpn3d.feature_pyramid.",
              "transformer": "transformer_encoder.", "decoder": "correspondence_decoder."}
    if profile != "default":
        groups.update(EXTRA_GROUPS)
    out = {"n_src": pred["src_kp"][0].shape[0], "n_tgt": pred["tgt_kp"][0].shape[0], "pose": pred["pose"].detach().numpy()}
    for k, v in losses.items():
        out["loss_" + k] = float(v)
    for gname, pref in groups.items():
        out["gnorm_" + gname] = float(sum(float(v.grad.double().pow(2).sum()) for k, v in leaves.items() if k.startswith(pref) and v.grad is not None) ** 0.5)
    for key in base.files:
        if key.startswith("gidx/"):
            out["gval/" + key[5:]] = leaves[key[5:]].grad.flatten()[base[key]].numpy()
    plist = [v for v in leaves.values() if v.grad is not None]
    out["total_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(plist, max_norm=0.1))
    before = {k: v.detach().clone() for k, v in leaves.items()}
    torch.optim.AdamW(plist, lr=1e-4, weight_decay=1e-4).step()
    for gname, pref in groups.items():
        out["dnorm_" + gname] = float(sum((leaves[k].detach() - before[k]).double().pow(2).sum() for k in leaves if k.startswith(pref)) ** 0.5)
    np.savez(os.path.join(OUT, name + "_bf16emu.npz"), **out)
    print(name + "_bf16emu done", {k: v for k, v in out.items() if k.startswith(("loss_", "gnorm_", "total"))})


EXTRA_PROBES = ["fpn3d.backbone_net.layer2.1.conv1.weight", "fpn3d.backbone_net.layer2.0.downsample.0.weight", "fpn3d.backbone_net.layer3.2.conv2.weight",
                "fpn3d.backbone_net.layer3.5.bn1.weight", "fpn3d.backbone_net.layer4.0.downsample.0.weight", "fpn3d.backbone_net.layer4.1.conv2.weight",
                "fpn3d.feature_pyramid.upsample_transform_2.weight", "fpn3d.feature_pyramid.pyramid_transformation_4.weight"]
EXTRA_GROUPS = {"stem": "fpn3d.backbone_net.conv1.", "layer1": "fpn3d.backbone_net.layer1.", "layer2": "fpn3d.backbone_net.layer2.",
                "layer3": "fpn3d.backbone_net.layer3.", "layer4": "fpn3d.backbone_net.layer4."}


def train_golden(nr, sd, res: int, name: str, with_fp64: bool = True, profile="default", sample=(1, 2, 0, 0, 5)):
    # sample = (source grid seed, target grid seed, synth.fixed_pose variant, weight seed, InfoNCE W seed)
    """One training step of the REFERENCE (train-mode BatchNorm, its own loss code, clip_grad_norm_ + AdamW) on
    shell_pair(res, 1, 2): losses, pose, per-module gradient norms, gradient probes (+ their fp64 truth from the
    reference-pinned oracle), clip norm and the per-module parameter delta of the optimizer step."""
    from conerf.loss.feature_loss import InfoNCELoss
    m = ref_model(nr, sd).train()
    data = synth.shell_pair(res, sample[0], sample[1], pose=synth.fixed_pose(sample[2]))
    feature_loss = InfoNCELoss(d_embed=256, r_p=0.2, r_n=0.4)
    gW = torch.Generator().manual_seed(sample[4])
    with torch.no_grad():
        feature_loss.W.copy_(0.1 * torch.randn(256, 256, generator=gW))
    pred = m({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()})
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
    with torch.no_grad():
        s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(6)])
        t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(6)])
    # the reference's loss code, argument for argument (train_nerf_regtr.py:186-228)
    from conerf.register.se3 import se3_transform_list, se3_inv
    from conerf.loss.correspondence_loss import CorrespondenceLoss
    pose_gt = data["pose"]
    losses = {}
    ov_gt = torch.cat([s_gt] + [t_gt], dim=-2)
    ov_pred = torch.cat(pred["src_overlap"] + pred["tgt_overlap"], dim=-2)
    losses["overlap"] = torch.nn.BCEWithLogitsLoss()(ov_gt[-1], ov_pred[-1])
    losses["nerf_cont"] = torch.nn.functional.smooth_l1_loss(ov_gt, torch.cat([s_tl] + [t_tl], dim=-2))
    losses["feature"] = feature_loss([f[-1] for f in pred["src_feats"]], [f[-1] for f in pred["tgt_feats"]],
                                     se3_transform_list(pose_gt, pred["src_kp"]), pred["tgt_kp"])
    cl = CorrespondenceLoss(metric="mae", robust_loss=False)
    losses["corr"] = cl(pred["src_kp"], [w_[-1] for w_ in pred["src_kp_warped"]], pose_gt, overlap_weights=[s_gt]) + \
        cl(pred["tgt_kp"], [w_[-1] for w_ in pred["tgt_kp_warped"]],
           torch.stack([se3_inv(p) for p in pose_gt]), overlap_weights=[t_gt])
    wd = {"overlap": 1.0, "nerf_cont": 1.0, "feature": 0.1, "corr": 1.0}
    losses["total"] = torch.sum(torch.stack([losses[k] * wd[k] for k in wd]))
    losses["total"].backward()
    gnorm = {}
    named = dict(m.named_parameters())
    groups = {"resnet": "fpn3d.backbone_net.", "fpn_head": "fpn3d.feature_pyramid.",
              "transformer": "transformer_encoder.", "decoder": "correspondence_decoder."}
    if profile != "default":
        groups.update(EXTRA_GROUPS)      # the well-conditioned fixtures also pin the ResNet stage by stage
    for gname, pref in groups.items():
        sq = 0.0
        for k, p in named.items():
            if k.startswith(pref) and p.grad is not None:
                sq += float(p.grad.double().pow(2).sum())
        gnorm[gname] = sq ** 0.5
    probes = ["fpn3d.backbone_net.conv1.weight", "fpn3d.backbone_net.layer1.0.conv2.weight",
              "fpn3d.backbone_net.layer4.2.bn3.weight", "fpn3d.feature_pyramid.upsample_transform_1.weight",
              "fpn3d.feature_pyramid.pyramid_transformation_1.bias",
              "transformer_encoder.layers.0.self_attn.in_proj_weight",
              "transformer_encoder.layers.5.linear2.weight", "correspondence_decoder.q_proj.weight"]
    if profile != "default":
        probes = probes + EXTRA_PROBES
    gp = {}
    for k in probes:
        gr = named[k].grad.flatten()
        idx = sample_idx(gr.numel(), 64, 31)
        gp["gidx/" + k] = idx
        gp["gval/" + k] = gr[idx].numpy()
    gnorm64, l64 = {}, {}
    if with_fp64:
        # fp64 ground truth of the same probes from the (reference-pinned) oracle: train-mode BatchNorm over the 8 voxels
        # of layer4 makes fp32 gradients of the ResNet noisy at the 1e-2 level, so GPU tests bound their error relative
        # to the fp32 reference's own distance from this truth.
        sd64 = {}
        for k, v in params.synth_state_dict(sample[3], profile=profile).items():
            if k.startswith(params.ALIAS_DST):
                sd64[k] = sd64[params.ALIAS_SRC + k[len(params.ALIAS_DST):]]
            else:
                sd64[k] = v.double() if v.is_floating_point() else v.clone()
        for k, (shape, kind) in params.regtr_spec().items():
            if not params.is_buffer(kind) and not k.startswith(params.ALIAS_DST):
                sd64[k].requires_grad_(True)
        d64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in data.items()}
        p64 = O.regtr_forward(sd64, d64, train=True)
        s_gt64, t_gt64 = synth.synthetic_overlap_gt(p64["src_kp"][0]), synth.synthetic_overlap_gt(p64["tgt_kp"][0])
        assert torch.equal(s_gt64.float(), s_gt) and torch.equal(t_gt64.float(), t_gt)
        l64 = O.training_losses(p64, d64["pose"], feature_loss.W.detach().double(), s_gt64, t_gt64, s_tl.double(), t_tl.double())
        l64["total"].backward()
        for k in probes:
            gp["gval64/" + k] = sd64[k].grad.flatten()[gp["gidx/" + k]].numpy()
        gnorm64 = {}
        for gname, pref in groups.items():
            gnorm64[gname] = float(sum(float(v.grad.pow(2).sum()) for k, v in sd64.items()
                                       if k.startswith(pref) and not k.startswith(params.ALIAS_DST) and v.grad is not None) ** 0.5)
        print("fp64 losses", {k: float(v) for k, v in l64.items()}, gnorm64)
    total_norm = float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=0.1))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    before = {k: p.detach().clone() for k, p in named.items()}
    opt.step()
    dnorm = {gname: float(sum((named[k].detach() - before[k]).double().pow(2).sum()
                              for k in named if k.startswith(pref)) ** 0.5)
             for gname, pref in groups.items()}
    bn_probe = m.state_dict()["fpn3d.backbone_net.layer3.1.bn2.running_var"][:16].numpy()
    bn_probe_m = m.state_dict()["fpn3d.backbone_net.bn1.running_mean"][:16].numpy()
    np.savez(os.path.join(OUT, name + ".npz"),
             n_src=s_kp.shape[0], n_tgt=t_kp.shape[0],
             pose=pred["pose"].detach().numpy(),
             **{"loss_" + k: float(v) for k, v in losses.items()},
             **{"gnorm_" + k: v for k, v in gnorm.items()},
             **{"gnorm64_" + k: v for k, v in gnorm64.items()},
             **{"loss64_" + k: float(v) for k, v in l64.items()},
             **{"dnorm_" + k: v for k, v in dnorm.items()},
             total_grad_norm=total_norm, bn_running_var_probe=bn_probe, bn_running_mean_probe=bn_probe_m,
             W_seed=sample[4], profile=str(profile), sample=np.asarray(sample, dtype=np.int64), **gp)
    print(name, "done", {k: float(v) for k, v in losses.items()}, gnorm, total_norm)


def pos_embed_golden():
    """Non-default position embeddings (nerf_regtr.py:87-90): the learned MLP through the whole transformer + decoder, forward
    and the gradients of its parameters, and the sine embedding with a coordinate scale != 1.
    usage: python tools/make_golden.py pos_embed"""
    nr = import_reference()
    from conerf.register.position_embedding import PositionEmbeddingCoordsSine
    sd = params.synth_state_dict(0, "learned")
    m = nr.NeRFRegTr("learned", 256, 1.0)
    spec = params.regtr_spec("learned")
    assert list(m.state_dict().keys()) == list(spec.keys()), "state_dict key order differs from params.regtr_spec('learned')"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == spec[k][0], k
    m.load_state_dict(params.clone_state_dict(sd), strict=True)
    m.eval()
    g = torch.Generator().manual_seed(31)
    ns, nt = 70, 55
    s_xyz = (torch.rand(ns, 3, generator=g) - 0.5) * 2
    t_xyz = (torch.rand(nt, 3, generator=g) - 0.5) * 2
    s_f = torch.randn(ns, 256, generator=g)
    t_f = torch.randn(nt, 256, generator=g)
    w_corr = torch.randn(6, ns + nt, 3, generator=g)
    w_ov = torch.randn(6, ns + nt, 1, generator=g)
    s_pe, t_pe = m.pos_embed(s_xyz), m.pos_embed(t_xyz)
    sc, tc = m.transformer_encoder(
        src=s_f[:, None], tgt=t_f[:, None],
        src_key_padding_mask=torch.zeros(1, ns, dtype=torch.bool),
        tgt_key_padding_mask=torch.zeros(1, nt, dtype=torch.bool),
        src_pos=s_pe[:, None], tgt_pos=t_pe[:, None])
    scl, tcl, sol, tol = m.correspondence_decoder(sc, tc, [s_xyz], [t_xyz])
    corr = torch.cat([scl[0], tcl[0]], dim=1)      # [6, ns+nt, 3]
    ov = torch.cat([sol[0], tol[0]], dim=1)        # [6, ns+nt, 1]
    loss = (corr * w_corr).sum() + (ov * w_ov).sum()
    loss.backward()
    out = dict(s_xyz=s_xyz.numpy(), t_xyz=t_xyz.numpy(), s_f=s_f.numpy(), t_f=t_f.numpy(), w_corr=w_corr.numpy(), w_ov=w_ov.numpy(),
               s_pe=s_pe.detach().numpy(), t_pe=t_pe.detach().numpy(),
               s_cond=sc[:, :, 0].detach().numpy(), t_cond=tc[:, :, 0].detach().numpy(),
               corr=corr.detach().numpy(), ov=ov.detach().numpy(), loss=loss.detach().numpy())
    for i in range(5):
        out[f"g_w{i}"] = m.pos_embed.mlp[2 * i].weight.grad.numpy()
        out[f"g_b{i}"] = m.pos_embed.mlp[2 * i].bias.grad.numpy()
    with torch.no_grad():
        out["sine_scale_half"] = PositionEmbeddingCoordsSine(3, 256, scale=0.5)(s_xyz).numpy()
    np.savez(os.path.join(OUT, "pos_embed.npz"), **out)
    print("pos_embed done", float(loss))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pos_embed":
        pos_embed_golden()
    else:
        main()
        pos_embed_golden()

"""GPU: the path bench.py times — bf16 + native trunk executor + active-set head + fused losses + FlatAdamW (TrainStep.step) — against the
REFERENCE's own training step recorded in tests/golden/train64.npz / train128.npz (tools/make_golden.py: losses, per-module gradient
norms, gradient probes with their fp64 truth, clip norm, per-module parameter delta of the AdamW step, BatchNorm running statistics).
Reference: train_nerf_regtr.py:171-239.  The fp32 parity mode is pinned at 128^3 as well (two A4 rounds, split-K thresholds and the
32-bit index ranges of the BASELINE size)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import losses as LS, params, synth  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402
from dreg_nerf_amd.train_step import TrainStep  # noqa: E402

GROUPS = {"resnet": "fpn3d.backbone_net.", "fpn_head": "fpn3d.feature_pyramid.",
          "transformer": "transformer_encoder.", "decoder": "correspondence_decoder."}
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pinned_step_report.json")


def _to(data, dev="cuda"):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}


def _report(tag, rec):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        old = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        old[tag] = rec
        json.dump(old, open(REPORT, "w"), indent=1)
    except OSError:
        pass


def _unclipped_grads(named, out, max_norm=0.1):
    """FlatAdamW keeps the step's gradients until zero_grad(), scaled in place by the clip factor like clip_grad_norm_ does."""
    gn = float(out["grad_norm"])
    clip = min(1.0, max_norm / (gn + 1e-6))
    return {k: (p.grad.detach().double() / clip).float().cpu() for k, p in named.items()}


def _product_step(g, res, active_set=True, native_trunk=True, profile="default"):
    """One TrainStep.step on shell_pair(res, 1, 2) with the golden's weights / InfoNCE W; returns everything the fixture pins."""
    # which sample the fixture pins: (source grid seed, target grid seed, pose variant, weight seed, W seed); older fixtures: the first
    sample = tuple(int(v) for v in g["sample"]) if "sample" in g.files else (1, 2, 0, 0, int(g["W_seed"]))
    m = NeRFRegTr(precision="bf16")
    m.load_state_dict(params.synth_state_dict(sample[3], profile=profile), strict=True)
    m = m.cuda().train()
    m.active_set, m.native_trunk = active_set, native_trunk
    ts = TrainStep(m)     # reference hyper-parameters: AdamW(lr 1e-4, wd 1e-4), clip 0.1 (train_nerf_regtr.py:96-102,232-237)
    assert ts.fused_losses
    with torch.no_grad():
        ts.feature_loss.W.copy_((0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(int(g["W_seed"])))).cuda())
    named = dict(m.named_parameters())
    before = {k: p.detach().clone() for k, p in named.items()}
    data = _to(synth.shell_pair(res, sample[0], sample[1], pose=synth.fixed_pose(sample[2])))
    out = ts.step([data])
    torch.cuda.synchronize()
    pred = ts.last_preds[0]
    grads = _unclipped_grads(named, out)
    delta = {k: (named[k].detach() - before[k]).double().cpu() for k in named}
    return m, ts, out, pred, grads, delta


def _check_against_golden(g, tag, out, pred, grads, delta, m, tol, emu=None):
    """Every pinned quantity is measured first and written to the report; the assertions follow.
    emu (bf16 runs): the reference-pinned oracle evaluated with bf16 operand rounding on the same step (tools/make_golden.py
    bf16_yardstick).  The network at this random initialisation is badly conditioned — the fp32 reference itself is only within 1e-2 of
    the fp64 truth for early-layer gradients — so pose and gradients of a bf16 evaluation are bounded RELATIVE to that yardstick:
    distance(build, truth) <= K x distance(emulated reference, truth) + a small floor.  Losses, the optimizer's parameter delta and the
    BatchNorm statistics are well conditioned and bounded directly."""
    rec, bad = {}, []
    K = tol.get("K", 3.0)

    def close(name, got, ref, rtol, emu_val=None):
        got, ref = float(got), float(ref)
        rec[name] = [got, ref] + ([float(emu_val)] if emu_val is not None else [])
        bound = rtol * abs(ref)
        if emu_val is not None:
            bound = max(bound, K * abs(float(emu_val) - ref))
        if not abs(got - ref) <= bound:
            bad.append((name, got, ref, bound))

    def norm_close(name, got, truth, rtol, emu_val=None):
        """A gradient NORM under rounding noise grows like sqrt(truth^2 + noise^2): compare the implied relative noise
        sqrt(|got^2 - truth^2|) / truth with K x the yardstick's, not the difference of the norms."""
        got, truth = float(got), float(truth)
        noise = lambda v: abs(v * v - truth * truth) ** 0.5 / truth
        bound = noise(truth * (1.0 + rtol))
        if emu_val is not None:
            bound = max(bound, K * noise(float(emu_val)))
        rec[name] = {"got": got, "truth": truth, "emulated_reference": None if emu_val is None else float(emu_val), "noise": noise(got), "noise_bound": bound}
        if not noise(got) <= bound:
            bad.append((name, got, truth, noise(got), bound))

    assert pred["src_kp"][0].shape[0] == int(g["n_src"]) and pred["tgt_kp"][0].shape[0] == int(g["n_tgt"])
    # ---- losses (the reference's own loss code on the reference's fp32 forward)
    for k in ("overlap", "nerf_cont", "feature", "corr", "total"):
        close("loss_" + k, out["losses"][k], g["loss_" + k], tol["loss_feature"] if k == "feature" else tol["loss"],
              emu["loss_" + k] if emu is not None else None)
    pose_err = float(np.abs(pred["pose"].detach().cpu().numpy() - g["pose"]).max())
    pose_bound = tol["pose"] + (K * float(np.abs(emu["pose"] - g["pose"]).max()) if emu is not None else 0.0)
    rec["pose_maxabs"] = [pose_err, pose_bound]
    if not pose_err <= pose_bound:
        bad.append(("pose", pose_err, pose_bound))
    # ---- per-module gradient norms against the fp64 truth of the same step
    for name, pref in GROUPS.items():
        sq = sum(float(v.double().pow(2).sum()) for k, v in grads.items() if k.startswith(pref))
        norm_close("gnorm_" + name, sq ** 0.5, g["gnorm64_" + name], tol["gnorm"], emu["gnorm_" + name] if emu is not None else None)
    # ---- gradient probes: relative distance to the truth (and the cosine, reported)
    for key in g.files:
        if not key.startswith("gidx/"):
            continue
        k = key[5:]
        got = grads[k].flatten()[g[key]].double().numpy()
        ref = g["gval64/" + k]
        cos = float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        rel = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-300))
        bound = tol["probe"]
        rel_emu = None
        if emu is not None:
            rel_emu = float(np.linalg.norm(emu["gval/" + k].astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-300))
            bound = max(bound, K * rel_emu)
        else:   # fp32: of the order of the fp32 reference's own distance to the truth
            bound = max(bound, 4 * float(np.linalg.norm(g["gval/" + k].astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-300)))
        rec["probe/" + k] = {"cos": cos, "rel": rel, "rel_emulated_reference": rel_emu, "bound": bound}
        if not rel <= bound:
            bad.append(("probe " + k, rel, bound))
    # ---- clip norm (clip_grad_norm_(0.1) sees the norm over ALL parameters) and the parameter delta of the optimizer step
    norm_close("total_grad_norm", out["grad_norm"], g["total_grad_norm"], tol["gnorm"], emu["total_grad_norm"] if emu is not None else None)
    for name, pref in GROUPS.items():
        dn = sum(float(v.pow(2).sum()) for k, v in delta.items() if k.startswith(pref)) ** 0.5
        close("dnorm_" + name, dn, g["dnorm_" + name], tol["dnorm"])
    # ---- BatchNorm running statistics after the step (momentum 0.1 update, src grid then tgt grid)
    sd = m.state_dict()
    rm = sd["fpn3d.backbone_net.bn1.running_mean"][:16].float().cpu().numpy()
    rv = sd["fpn3d.backbone_net.layer3.1.bn2.running_var"][:16].float().cpu().numpy()
    rec["bn_mean_maxrel"] = float(np.abs(rm - g["bn_running_mean_probe"]).max() / (np.abs(g["bn_running_mean_probe"]).max() + 1e-30))
    rec["bn_var_maxrel"] = float((np.abs(rv - g["bn_running_var_probe"]) / np.abs(g["bn_running_var_probe"])).max())
    for k in ("bn_mean_maxrel", "bn_var_maxrel"):
        if not rec[k] <= tol["bn"]:
            bad.append((k, rec[k], tol["bn"]))
    rec["failed"] = [str(b) for b in bad]
    _report(tag, rec)
    assert not bad, bad
    return rec


def _emu(golden_dir, name):
    p = os.path.join(golden_dir, name + "_bf16emu.npz")
    assert os.path.exists(p), f"{p} missing: python tools/make_golden.py bf16emu64 bf16emu128"
    return np.load(p)


# bf16: direct bounds for the well-conditioned quantities (measured: losses <= 1.1 %, parameter delta <= 0.2 %, BatchNorm statistics
# <= 0.2 %), yardstick-relative bounds (K = 3) with these floors for pose and gradients.  fp32: direct bounds.
TOL_BF16 = {"loss": 2e-2, "loss_feature": 4e-2, "pose": 2e-3, "gnorm": 2e-2, "probe": 2e-2, "dnorm": 1e-2, "bn": 1e-2, "K": 3.0}
TOL_FP32 = {"loss": 1e-3, "loss_feature": 5e-3, "pose": 1e-4,      # pose: the north-star tolerance (rotation / translation within 1e-4 of the reference; measured 1.7e-5 at 128^3)
             "gnorm": 2e-2, "probe": 2e-3, "dnorm": 5e-3, "bn": 1e-3}


def test_bf16_product_step_matches_reference_golden_64(golden_dir):
    g = np.load(os.path.join(golden_dir, "train64.npz"))
    m, ts, out, pred, grads, delta = _product_step(g, 64)
    assert m.__dict__.get("_trunk_cache"), "the native trunk executor did not run"
    _check_against_golden(g, "bf16_64_active_exec", out, pred, grads, delta, m, TOL_BF16, _emu(golden_dir, "train64"))


def test_bf16_dense_head_step_matches_reference_golden_64(golden_dir):
    g = np.load(os.path.join(golden_dir, "train64.npz"))
    m, ts, out, pred, grads, delta = _product_step(g, 64, active_set=False)
    _check_against_golden(g, "bf16_64_dense_exec", out, pred, grads, delta, m, TOL_BF16, _emu(golden_dir, "train64"))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "train128.npz")), reason="train128.npz not generated")
def test_bf16_product_step_matches_reference_golden_128(golden_dir):
    g = np.load(os.path.join(golden_dir, "train128.npz"))
    m, ts, out, pred, grads, delta = _product_step(g, 128)
    _check_against_golden(g, "bf16_128_active_exec", out, pred, grads, delta, m, TOL_BF16, _emu(golden_dir, "train128"))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "train128.npz")), reason="train128.npz not generated")
def test_fp32_mode_step_matches_reference_golden_128(golden_dir):
    """Exact-f32 MFMA mode at the BASELINE resolution: per-op path, per-pair torch losses, same FlatAdamW."""
    g = np.load(os.path.join(golden_dir, "train128.npz"))
    m = NeRFRegTr(precision="fp32")
    m.load_state_dict(params.synth_state_dict(0), strict=True)
    m = m.cuda().train()
    ts = TrainStep(m)
    with torch.no_grad():
        ts.feature_loss.W.copy_((0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(int(g["W_seed"])))).cuda())
    named = dict(m.named_parameters())
    before = {k: p.detach().clone() for k, p in named.items()}
    out = ts.step([_to(synth.shell_pair(128, 1, 2, pose=synth.fixed_pose()))])
    torch.cuda.synchronize()
    grads = _unclipped_grads(named, out)
    delta = {k: (named[k].detach() - before[k]).double().cpu() for k in named}
    _check_against_golden(g, "fp32_128", out, ts.last_preds[0], grads, delta, m, TOL_FP32)


# ------------------------------------------------------------------------------------------------ absolute bounds (round 3)
# The fixtures above sit at the reference's own random initialisation, where the network is chaotic (gradient norm 1.5e3 at the decoder,
# 3e5 at the ResNet; bf16 rounding alone moves early-layer gradients by O(1)) and pose / gradients can only be bounded relative to an
# emulation.  train64_wc / train128_wc are the SAME step of the REFERENCE on the well-conditioned weight profile params.PROFILES["wc"]
# (tools/wc_profile_sweep.py: what each knob buys; the bf16-emulated oracle itself reaches pose 6e-5, primary probes cos >= 0.997, deep
# ResNet probes cos >= 0.96, every gradient norm within 0.6 % at 128^3).  Here the bf16 build is bounded in ABSOLUTE terms:
#   pose max-abs <= 2e-2; gradient probes (64 sampled elements each, against the fp64 truth): cos >= 0.99 for the stem, layer1, both head
#   levels, the transformer and the decoder; cos >= 0.9 at 128^3 / >= 0.8 at 64^3 for the probes inside layer2..4, whose BatchNorms
#   see 8..512 voxels per grid (the bf16-EMULATED REFERENCE itself reads 0.83-0.98 there at 64^3; a sign flip reads -1);
#   every per-module AND per-ResNet-stage gradient norm within 5 % of the fp64 truth (measured <= 1.3 %).
PRIMARY = ["fpn3d.backbone_net.conv1.weight", "fpn3d.backbone_net.layer1.0.conv2.weight",
           "fpn3d.feature_pyramid.upsample_transform_2.weight", "fpn3d.feature_pyramid.pyramid_transformation_4.weight",
           "fpn3d.feature_pyramid.upsample_transform_1.weight", "fpn3d.feature_pyramid.pyramid_transformation_1.bias",
           "transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.5.linear2.weight", "correspondence_decoder.q_proj.weight"]
STAGES = {"stem": "fpn3d.backbone_net.conv1.", "layer1": "fpn3d.backbone_net.layer1.", "layer2": "fpn3d.backbone_net.layer2.",
          "layer3": "fpn3d.backbone_net.layer3.", "layer4": "fpn3d.backbone_net.layer4."}


def _check_absolute(g, tag, out, pred, grads, delta, m, emu=None, pose_tol=2e-2, cos_primary=0.99, cos_deep=0.9, gnorm_tol=5e-2):
    rec, bad = {}, []
    assert str(g["profile"]) == "wc"
    assert pred["src_kp"][0].shape[0] == int(g["n_src"]) and pred["tgt_kp"][0].shape[0] == int(g["n_tgt"])
    for k in ("overlap", "nerf_cont", "feature", "corr", "total"):
        got, ref = float(out["losses"][k]), float(g["loss_" + k])
        rec["loss_" + k] = [got, ref]
        if not abs(got - ref) <= 2e-2 * abs(ref):
            bad.append(("loss_" + k, got, ref))
    pose_err = float(np.abs(pred["pose"].detach().cpu().numpy() - g["pose"]).max())
    rec["pose_maxabs"] = [pose_err, pose_tol, None if emu is None else float(np.abs(emu["pose"] - g["pose"]).max())]
    if not pose_err <= pose_tol:
        bad.append(("pose", pose_err, pose_tol))
    for name, pref in {**GROUPS, **STAGES}.items():
        got = sum(float(v.double().pow(2).sum()) for k, v in grads.items() if k.startswith(pref)) ** 0.5
        truth = float(g["gnorm64_" + name])
        rec["gnorm_" + name] = {"got": got, "truth": truth, "reference_fp32": float(g["gnorm_" + name]),
                                "emulated_reference": None if emu is None else float(emu["gnorm_" + name])}
        if not abs(got - truth) <= gnorm_tol * truth:
            bad.append(("gnorm_" + name, got, truth))
    for key in g.files:
        if not key.startswith("gidx/"):
            continue
        k = key[5:]
        got = grads[k].flatten()[g[key]].double().numpy()
        ref = g["gval64/" + k]
        cos = float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        rel = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-300))
        need = cos_primary if k in PRIMARY else cos_deep
        cos_emu = None
        if emu is not None:
            e = emu["gval/" + k].astype(np.float64)
            cos_emu = float(np.dot(e, ref) / (np.linalg.norm(e) * np.linalg.norm(ref) + 1e-300))
        rec["probe/" + k] = {"cos": cos, "rel": rel, "cos_required": need, "cos_emulated_reference": cos_emu}
        if not cos >= need:
            bad.append(("probe " + k, cos, need))
    gn, truth = float(out["grad_norm"]), float(g["total_grad_norm"])
    rec["total_grad_norm"] = [gn, truth]
    if not abs(gn - truth) <= gnorm_tol * truth:
        bad.append(("total_grad_norm", gn, truth))
    for name, pref in GROUPS.items():
        dn = sum(float(v.pow(2).sum()) for k, v in delta.items() if k.startswith(pref)) ** 0.5
        rec["dnorm_" + name] = [dn, float(g["dnorm_" + name])]
        if not abs(dn - float(g["dnorm_" + name])) <= 1e-2 * float(g["dnorm_" + name]):
            bad.append(("dnorm_" + name, dn, float(g["dnorm_" + name])))
    sd = m.state_dict()
    rm = sd["fpn3d.backbone_net.bn1.running_mean"][:16].float().cpu().numpy()
    rv = sd["fpn3d.backbone_net.layer3.1.bn2.running_var"][:16].float().cpu().numpy()
    rec["bn_mean_maxrel"] = float(np.abs(rm - g["bn_running_mean_probe"]).max() / (np.abs(g["bn_running_mean_probe"]).max() + 1e-30))
    rec["bn_var_maxrel"] = float((np.abs(rv - g["bn_running_var_probe"]) / np.abs(g["bn_running_var_probe"])).max())
    for k in ("bn_mean_maxrel", "bn_var_maxrel"):
        if not rec[k] <= 1e-2:
            bad.append((k, rec[k]))
    rec["failed"] = [str(b) for b in bad]
    _report(tag, rec)
    assert not bad, bad
    return rec


def _wc(golden_dir, name):
    p = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(p):
        pytest.skip(f"{name}.npz not generated (python tools/make_golden.py {name})")
    e = os.path.join(golden_dir, name + "_bf16emu.npz")
    return np.load(p), (np.load(e) if os.path.exists(e) else None)


@pytest.mark.parametrize("active_set", [True, False])
def test_bf16_product_step_absolute_bounds_wc_64(golden_dir, active_set):
    g, emu = _wc(golden_dir, "train64_wc")
    m, ts, out, pred, grads, delta = _product_step(g, 64, active_set=active_set, profile="wc")
    assert m.__dict__.get("_trunk_cache"), "the native trunk executor did not run"
    _check_absolute(g, "wc_bf16_64_" + ("active" if active_set else "dense"), out, pred, grads, delta, m, emu, cos_deep=0.8)


def test_bf16_product_step_absolute_bounds_wc_64_second_sample(golden_dir):
    """The same absolute bounds on a SECOND reference-generated step (other grids, another relative pose, other weight / W seeds:
    tools/make_golden.py train64_wc_b): the tolerances are not tuned to one sample."""
    g, emu = _wc(golden_dir, "train64_wc_b")
    assert tuple(int(v) for v in g["sample"]) == (7, 8, 1, 1, 9)
    m, ts, out, pred, grads, delta = _product_step(g, 64, profile="wc")
    _check_absolute(g, "wc_bf16_64_b_active", out, pred, grads, delta, m, emu, cos_deep=0.8)


def test_bf16_product_step_absolute_bounds_wc_128(golden_dir):
    g, emu = _wc(golden_dir, "train128_wc")
    m, ts, out, pred, grads, delta = _product_step(g, 128, profile="wc")
    _check_absolute(g, "wc_bf16_128_active", out, pred, grads, delta, m, emu)


def test_bf16_product_step_absolute_bounds_wc_128_second_sample(golden_dir):
    """The second reference-generated sample (other grids, relative pose, weight / W seeds) at the BASELINE size (tools/make_golden.py
    train128_wc_b)."""
    g, emu = _wc(golden_dir, "train128_wc_b")
    if "gnorm64_resnet" not in g.files:
        pytest.skip("train128_wc_b.npz was generated without the fp64 oracle pass (DREG_GOLDEN_FP64_128B=1 python tools/make_golden.py train128_wc_b)")
    assert tuple(int(v) for v in g["sample"]) == (7, 8, 1, 1, 9)
    m, ts, out, pred, grads, delta = _product_step(g, 128, profile="wc")
    _check_absolute(g, "wc_bf16_128_b_active", out, pred, grads, delta, m, emu)


def test_sign_flip_in_a_weight_gradient_is_caught(golden_dir):
    """The absolute bounds have teeth: the same step with ONE ResNet weight gradient negated must fail the probe test."""
    g, emu = _wc(golden_dir, "train64_wc")
    m, ts, out, pred, grads, delta = _product_step(g, 64, profile="wc")
    broken = dict(grads)
    broken["fpn3d.backbone_net.layer1.0.conv2.weight"] = -grads["fpn3d.backbone_net.layer1.0.conv2.weight"]
    with pytest.raises(AssertionError):
        _check_absolute(g, "wc_sign_flip_must_fail", out, pred, broken, delta, m, emu, cos_deep=0.8)
    broken = dict(grads)                                  # ... also for a probe inside the deep stages
    broken["fpn3d.backbone_net.layer3.2.conv2.weight"] = -grads["fpn3d.backbone_net.layer3.2.conv2.weight"]
    with pytest.raises(AssertionError):
        _check_absolute(g, "wc_sign_flip_deep_must_fail", out, pred, broken, delta, m, emu, cos_deep=0.8)
    half = dict(grads)                                    # and a dropped contribution (gradient scaled by 0.8) fails the stage norm
    for k in list(half):
        if k.startswith("fpn3d.backbone_net.layer3."):
            half[k] = 0.8 * grads[k]
    with pytest.raises(AssertionError):
        _check_absolute(g, "wc_scaled_stage_must_fail", out, pred, half, delta, m, emu, cos_deep=0.8)

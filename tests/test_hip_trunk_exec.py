"""GPU: the native trunk executor (csrc/executor.hip) against the per-op Python path it replaces — same kernels in the same
order, so the forward is bit-identical; parameter gradients agree up to the order of the bf16 additions where a tensor
feeds three consumers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import ops, synth  # noqa: E402
from dreg_nerf_amd.optim import FlatAdamW  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402

DEV = "cuda:0"


def _model(seed=0):
    torch.manual_seed(seed)
    m = NeRFRegTr(precision="bf16").to(DEV).train()
    opt = FlatAdamW(list(m.parameters()))   # preallocated flat .grad buffers: what the executor accumulates into
    return m, opt


def _grids(res, n, r0=0.55, r1=0.8):
    gs, idx = [], []
    for i in range(n):
        g, m = synth.shell_grid(res, 5 + i, r0, r1)
        gs.append(g.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(DEV))
        idx.append(m.to(DEV))
    return gs, idx


@pytest.mark.parametrize("sparse,fused_stats", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_executor_matches_per_op_path(sparse, fused_stats):
    """fused_stats = 0: the executor takes every BatchNorm's statistics from the layer's own statistics pass, like the per-op path:
    bit-identical forward.  1 (default): from the producing convolution's epilogue (per-tile instead of per-chunk partial sums: the
    means differ in their last bits, and 53 BatchNorm layers at a random initialisation amplify that)."""
    from dreg_nerf_amd.trunk_exec import exec_opts
    with exec_opts(fuse_bn_stats=fused_stats):
        _executor_vs_per_op(sparse, bool(fused_stats))


def _executor_vs_per_op(sparse, fused_stats):
    m, opt = _model()
    if fused_stats:      # the well-conditioned weight profile (params.PROFILES["wc"]): last-bit differences of a mean stay small through 53 layers
        from dreg_nerf_amd import params
        m.load_state_dict(params.synth_state_dict(0, profile="wc"), strict=True)
    res = 64
    grids, idx = _grids(res, 2, *((0.3, 0.34) if sparse else (0.55, 0.8)))   # a thin shell keeps S3 under the 20 % density cap
    x = NeRFRegTr.pack_grids(grids, torch.bfloat16)
    rows = ops.active_sets(idx, (res,) * 3, (res // 2,) * 3, torch.device(DEV)) if sparse else None
    assert (rows is not None) == sparse
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    go = None
    outs, grads, stats = [], [], []
    for native in (False, True):
        m.load_state_dict(sd0)
        ops.bump_weight_generation()
        m.native_trunk = native
        opt.zero_grad()
        p1 = m.fpn(x, rows)
        if go is None:
            go = torch.randn(p1.shape, generator=g).to(DEV).bfloat16()
            if sparse:   # the gather only feeds gradient into the S1 rows
                mask = torch.zeros(p1.shape[:4].numel(), dtype=torch.bool, device=DEV)
                mask[rows[0].long()] = True
                go = go * mask.view(*p1.shape[:4], 1)
        sel = rows[0].long() if sparse else slice(None)
        outs.append(p1.detach().reshape(-1, p1.shape[-1])[sel].clone())
        p1.backward(go)
        grads.append(opt.flat_g.clone())
        stats.append({k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    if not fused_stats:
        assert torch.equal(outs[0], outs[1]), "forward must be bit-identical"
        for k in stats[0]:
            assert torch.equal(stats[0][k], stats[1][k]), k
    else:
        d = (outs[0].float() - outs[1].float()).norm().item() / outs[0].float().norm().item()
        worst = max(float((stats[0][k].float() - stats[1][k].float()).abs().max()) / max(float(stats[0][k].float().abs().max()), 1e-12) for k in stats[0])
        print(f"fused BatchNorm statistics vs per-op path: output rel. distance {d:.3e}, running statistics worst rel. {worst:.3e}")
        assert d <= 2e-2 and worst <= 3e-2
    ga, gb = grads
    assert torch.isfinite(gb).all()
    denom = ga.norm().item()
    print(f"gradient buffer rel. distance {(ga - gb).norm().item() / denom:.3e}")
    assert (ga - gb).norm().item() <= (6e-2 if fused_stats else 2e-2) * denom, ((ga - gb).norm().item(), denom)
    # per-parameter: the head / FPN parameters see identical inputs -> (nearly) identical gradients
    off = dict(zip([id(p) for p in opt.params], opt.offsets))
    for name, p in m.named_parameters():
        if not name.startswith("fpn3d.feature_pyramid.") or "resnet" in name:
            continue
        a = ga[off[id(p)]:off[id(p)] + p.numel()]
        b = gb[off[id(p)]:off[id(p)] + p.numel()]
        assert (a - b).norm().item() <= (5e-2 if fused_stats else 5e-3) * max(a.norm().item(), 1e-12), (name, (a - b).norm().item() / max(a.norm().item(), 1e-12))


def test_executor_eval_mode_and_no_grad():
    m, opt = _model(1)
    m.eval()
    grids, idx = _grids(32, 2)
    x = NeRFRegTr.pack_grids(grids, torch.bfloat16)
    with torch.no_grad():
        m.native_trunk = False
        a = m.fpn(x, None).clone()
        m.native_trunk = True
        b = m.fpn(x, None).clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("shell", [(0.3, 0.34), (0.55, 0.8)], ids=["active-set", "dense-head"])
def test_training_steps_are_bitwise_reproducible(shell):
    """Two runs of the same three optimizer steps (executor, geometry stream, second stream for the parameter gradients, fused
    losses — every reduction is deterministic) end in bit-identical parameters; a stream-ordering bug would show up here."""
    from dreg_nerf_amd.train_step import TrainStep

    def run():
        torch.manual_seed(7)
        m = NeRFRegTr(precision="bf16").to(DEV).train()
        ts = TrainStep(m)
        batch = []
        for i in range(2):
            d = {"pose": synth.fixed_pose()[None].clone(), "src_nerf_path": "", "tgt_nerf_path": ""}
            for j, side in enumerate(("src", "tgt")):
                g, mk = synth.shell_grid(64, 20 + 2 * i + j, *shell)
                d[side + "_xyz_rgba"], d[side + "_mask"] = g.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mk
            batch.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()})
        losses = []
        for _ in range(3):
            out = ts.step(batch)
            losses.append(float(out["losses"]["total"]))
        torch.cuda.synchronize()
        return ts.optimizer.flat_p.clone(), losses

    p1, l1 = run()
    p2, l2 = run()
    assert all(np.isfinite(l1)) and l1 == l2
    assert torch.equal(p1, p2)


def test_stem_row_skipping_does_not_change_a_training_step():
    """Three optimizer steps with and without the stem's empty-row skipping end in identical parameters and losses."""
    from dreg_nerf_amd.train_step import TrainStep
    res = []
    for skip in (True, False):
        torch.manual_seed(5)
        m = NeRFRegTr(precision="bf16").to(DEV).train()
        m.skip_empty_stem_rows = skip
        ts = TrainStep(m)
        batch = []
        for i in range(2):
            d = {"pose": synth.fixed_pose()[None].clone(), "src_nerf_path": "", "tgt_nerf_path": ""}
            for j, side in enumerate(("src", "tgt")):
                g, mk = synth.shell_grid(64, 30 + 2 * i + j, 0.3, 0.34)
                d[side + "_xyz_rgba"], d[side + "_mask"] = g.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mk
            batch.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()})
        ls = [float(ts.step(batch)["losses"]["total"]) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((ls, ts.optimizer.flat_p.clone()))
    assert all(np.isfinite(res[0][0])) and res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_segmented_backward_reports_gradient_buckets_in_order_and_changes_nothing():
    """Data-parallel overlap (optim.GradSync): with a sync object installed, the executor runs its backward pass in segments and reports
    finished gradient ranges between them.  Same launches in the same order: every gradient is bitwise what the one-call backward
    gives; the buckets are launched from the end of the flat buffer, each once, the first ones BEFORE the last segment was issued."""
    from dreg_nerf_amd.optim import GradSync
    from dreg_nerf_amd.train_step import TrainStep
    data = synth.shell_pair(64, 1, 2, pose=synth.fixed_pose())
    data = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in data.items()}
    grads = []
    for use_sync in (False, True):
        torch.manual_seed(0)
        m = NeRFRegTr(precision="bf16").to(DEV).train()
        ts = TrainStep(m)
        log = []
        if use_sync:
            ts.world = 2                                          # drive the data-parallel code path with a recording stand-in for the collective
            ts._sync = GradSync(ts.optimizer, 2, bucket_bytes=25 << 20, reduce_fn=lambda lo, hi: log.append(("reduce", lo, hi)))
            ex_backward = None
        ts.optimizer.max_norm = 0.0                              # keep the raw gradients in the buffer (no in-place clipping)
        if use_sync:
            from dreg_nerf_amd import trunk_exec
            orig = trunk_exec.TrunkExecutor.backward

            def spy(self, x, rows, g, generation=None):
                plan = self._sync_plan(ops.GRAD_SYNC)
                log.append(("plan", len(plan["cuts"]), plan["above"]))
                return orig(self, x, rows, g, generation)
            trunk_exec.TrunkExecutor.backward = spy
            try:
                ts.step([data])
            finally:
                trunk_exec.TrunkExecutor.backward = orig
        else:
            ts.step([data])
        torch.cuda.synchronize()
        grads.append(ts.optimizer.flat_g.clone())
        if use_sync:
            sync = ts._sync
            plans = [e for e in log if e[0] == "plan"]
            reduces = [(lo, hi) for tag, lo, hi in (e for e in log if e[0] == "reduce")]
            assert plans and plans[0][1] >= 4, plans              # 244 MB of fp32 gradients in 25 MB buckets: several segments
            assert reduces == sync.buckets and reduces[0][1] == ts.optimizer.n_active and reduces[-1][0] == 0
            # transformer + decoder buckets (everything above the trunk's parameters) are launched before the trunk's backward starts
            first_plan = log.index(plans[0])
            assert any(e[0] == "reduce" for e in log[:first_plan + 1]) or plans[0][2] >= sync.buckets[0][0]
            n_before_last = sum(1 for e in log if e[0] == "reduce")
            assert n_before_last == len(sync.buckets)
    assert torch.equal(grads[0], grads[1])


def test_row_cleared_gradient_buffers_equal_memset_ones_over_changing_active_sets():
    """The dense gradient buffers in front of the active-set convolutions (dP1 from the trilinear gather, the lateral sums inside the
    executor) are kept zero across steps by clearing the rows the last step wrote instead of a dense memset per step.  Three optimizer
    steps over pairs whose active sets grow and shrink end in bit-identical parameters with the feature on and off."""
    from dreg_nerf_amd import synth
    from dreg_nerf_amd.train_step import TrainStep
    from dreg_nerf_amd.trunk_exec import exec_opts
    dev = torch.device("cuda")

    def run(sparse: bool):
        with exec_opts(sparse_grads=int(sparse)):
            torch.manual_seed(7)
            m = NeRFRegTr(precision="bf16").to(dev).train()
            ts = TrainStep(m)
            ts.persistent_grad_buffers = sparse
            pose = synth.fixed_pose()
            for step, radii in enumerate((((0.8, 0.9), (0.6, 0.66)), ((0.45, 0.5), (0.9, 1.05)), ((0.7, 0.74), (0.3, 0.45)))):
                batch = []
                for i, (ra, rb) in enumerate(radii):     # shells of different radius / thickness: the active sets move, grow and shrink
                    gs, ms = synth.shell_grid(64, 11 + 2 * i + step, ra, rb)
                    gt, mt = synth.shell_grid(64, 12 + 2 * i + step, ra, rb, pose=pose)
                    batch.append({"src_xyz_rgba": gs.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev),
                                  "tgt_xyz_rgba": gt.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev),
                                  "src_mask": ms.to(dev), "tgt_mask": mt.to(dev), "pose": pose[None].clone().to(dev),
                                  "src_nerf_path": "", "tgt_nerf_path": ""})
                ts.step(batch)
            torch.cuda.synchronize()
            return {k: v.detach().clone() for k, v in m.state_dict().items()}

    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_batched_batchnorm_tails_change_nothing():
    """The running-statistics / dgamma-dbeta launches of the small BatchNorms batched per pass (one launch each) against one launch
    per layer: same arithmetic per channel, so gradients and running statistics are bit-identical."""
    from dreg_nerf_amd import trunk_exec
    m, opt = _model(2)
    res = 64
    grids, _ = _grids(res, 2)
    x = NeRFRegTr.pack_grids(grids, torch.bfloat16)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    go = None
    got = []
    try:
        for batched in (0, 1):
            trunk_exec.OPTS["bn_batch_tails"] = batched
            m.__dict__.pop("_trunk_cache", None)       # a creation option of the executor
            m.load_state_dict(sd0)
            ops.bump_weight_generation()
            m.native_trunk = True
            for _ in range(2):                          # the second pass accumulates on top of the first
                if _ == 0:
                    opt.zero_grad()
                p1 = m.fpn(x, None)
                if go is None:
                    go = torch.randn(p1.shape, generator=torch.Generator().manual_seed(4)).to(DEV).bfloat16()
                p1.backward(go)
            got.append((opt.flat_g.clone(), {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    finally:
        trunk_exec.OPTS.pop("bn_batch_tails", None)
        m.__dict__.pop("_trunk_cache", None)
    assert torch.isfinite(got[1][0]).all() and got[1][0].abs().sum() > 0
    assert torch.equal(got[0][0], got[1][0])
    for k in got[0][1]:
        assert torch.equal(got[0][1][k], got[1][1][k]), k


@pytest.mark.parametrize("option", ["ps_group_wgrad", "group_wgrad", "fold_splitk", "fold_res_bn"])
def test_bit_identical_round4_switches(option):
    """The round-4 restructurings that claim bit-identity — the point-set half's weight gradients as one launch per tile shape, the split-K sums
    of the 8^3 / 4^3 convolutions inside the BatchNorm launch next to them, the downsample branch's BatchNorm applied inside the BatchNorm that
    adds it — switched off and on: the same optimizer step, every gradient and the losses bit for bit."""
    from dreg_nerf_amd import params, pointset_exec, synth, trunk_exec
    from dreg_nerf_amd.train_step import TrainStep

    def fn(v):      # per-handle options of the two executors (the library has no process-global switches)
        if option == "ps_group_wgrad":
            pointset_exec.GROUP_WGRAD = bool(v)
        elif v:
            trunk_exec.OPTS.pop(option, None)
        else:
            trunk_exec.OPTS[option] = 0
    res = []
    try:
        for v in (0, 1):
            fn(v)
            torch.manual_seed(3407)
            m = NeRFRegTr(precision="bf16")
            m.load_state_dict(params.synth_state_dict(0, profile="wc"), strict=True)
            m = m.to(DEV).train()
            ts = TrainStep(m)
            batch = []
            for i in range(2):
                d = synth.shell_pair(64, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
                batch.append({k: (t.to(DEV) if torch.is_tensor(t) else t) for k, t in d.items()})
            out = ts.step(batch)
            torch.cuda.synchronize()
            res.append(({k: float(t) for k, t in out["losses"].items()}, float(out["grad_norm"]),
                        {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        fn(1)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k

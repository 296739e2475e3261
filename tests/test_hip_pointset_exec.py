"""GPU: the native point-set executor (csrc/pointset_exec.hip: encoder x 6, final norm, correspondence decoder, overlap head in one C
call per pass) against the per-op path (dreg_nerf_amd/transformer_ops.encode_decode_batched, one autograd node per kernel):
bit for bit with fuse = 0, to bf16 rounding with the fused epilogues, and through a whole training step."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import attn_ops as A, lib as L, ops, params, pointset_exec as PX, synth  # noqa: E402
from dreg_nerf_amd import transformer_ops as T  # noqa: E402
from dreg_nerf_amd.optim import FlatAdamW  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402


def _model():
    m = NeRFRegTr(precision="bf16")
    m.load_state_dict(params.synth_state_dict(0), strict=True)
    m = m.cuda().train()
    opt = FlatAdamW([p for p in m.parameters()])     # preallocated flat gradient buffers (what the executor accumulates into)
    opt.zero_grad()
    return m, opt


def _inputs(segs, seed=0):
    g = torch.Generator().manual_seed(seed)
    R = sum(a + b for a, b in segs)
    feats = (0.5 * torch.randn(R, 256, generator=g)).cuda().requires_grad_(True)
    xyz = (torch.rand(R, 3, generator=g) * 2 - 1).cuda()
    w = [torch.randn(6, R, c, generator=g).cuda() for c in (256, 3, 1)]
    return feats, xyz, w


def _run(m, opt, feats, xyz, tab, weights, native, use_cond=True, use_corr=True, use_ov=True):
    A.set_precision("bf16")
    opt.zero_grad()
    feats.grad = None
    P = m._P()
    if native:
        ex = PX.executor_for(m, P)
        assert ex is not None
        cond, corr, ov = PX.encode_decode(ex, feats, xyz, m.position_embedding(xyz), tab, P["transformer_encoder.norm.weight"])
    else:
        cond, corr, ov = T.encode_decode_batched(P, feats, xyz, tab, m.position_embedding)
    loss = 0.0
    if use_cond:
        loss = loss + (cond * weights[0]).sum() * 1e-2
    if use_corr:
        loss = loss + (corr * weights[1]).sum()
    if use_ov:
        loss = loss + (ov * weights[2]).sum()
    loss.backward()
    torch.cuda.synchronize()
    names = PX.param_names()
    grads = {n: P[n].grad.detach().clone() for n in names}
    return (cond.detach().clone(), corr.detach().clone(), ov.detach().clone()), feats.grad.detach().clone(), grads


@pytest.mark.parametrize("segs", [[(70, 55), (40, 90)], [(1184, 1170)]])
def test_native_executor_equals_per_op_path_bit_for_bit(segs):
    m, opt = _model()
    tab = A.ProblemTable(segs, torch.device("cuda"))
    feats, xyz, w = _inputs(segs)
    ref_out, ref_df, ref_g = _run(m, opt, feats, xyz, tab, w, native=False)
    PX.FUSE = False
    try:
        out, df, g = _run(m, opt, feats, xyz, tab, w, native=True)
    finally:
        PX.FUSE = True
    for a, b, name in zip(out, ref_out, ("cond", "corr", "overlap")):
        assert torch.equal(a, b), f"{name} differs"
    assert torch.equal(df, ref_df), "gradient of the input features differs"
    bad = [n for n in ref_g if not torch.equal(g[n], ref_g[n])]
    assert not bad, f"parameter gradients differ: {bad[:8]} ({len(bad)} of {len(ref_g)})"
    assert all(float(v.abs().max()) > 0 for n, v in ref_g.items()), "every parameter of the point-set half must receive a gradient"


def test_fused_epilogues_stay_within_bf16_rounding():
    """fuse = 1 (the product setting): ReLU mask and the decoder's gradient sum in the data-gradient epilogues, ONE LayerNorm backward
    for the final norm's two applications, batched bias sums.  Forward outputs are still bit-identical; gradients move by bf16
    rounding of intermediate sums only."""
    segs = [(300, 280), (150, 310)]
    m, opt = _model()
    tab = A.ProblemTable(segs, torch.device("cuda"))
    feats, xyz, w = _inputs(segs, seed=3)
    ref_out, ref_df, ref_g = _run(m, opt, feats, xyz, tab, w, native=False)
    out, df, g = _run(m, opt, feats, xyz, tab, w, native=True)
    for a, b in zip(out, ref_out):
        assert torch.equal(a, b)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-20))
    assert rel(df, ref_df) < 1.5e-2
    worst = max((rel(g[n], ref_g[n]), n) for n in ref_g)
    assert worst[0] < 1.5e-2, worst
    # batched bias sums / LayerNorm sums are the same sums in the same order: biases of layers whose output gradient is untouched by
    # the fused epilogues (the last layer's linear2: its gradient is the final norm's) move only through the fused final-norm backward
    assert rel(g["correspondence_decoder.q_proj.bias"], ref_g["correspondence_decoder.q_proj.bias"]) == 0.0


@pytest.mark.parametrize("which", ["cond", "corr", "ov"])
def test_single_head_gradients(which):
    """Only one of the three outputs reaches the loss: the executor must not need the other two gradients (evaluation losses, ablations)."""
    segs = [(90, 70)]
    m, opt = _model()
    tab = A.ProblemTable(segs, torch.device("cuda"))
    feats, xyz, w = _inputs(segs, seed=5)
    kw = dict(use_cond=which == "cond", use_corr=which == "corr", use_ov=which == "ov")
    _, ref_df, ref_g = _run(m, opt, feats, xyz, tab, w, native=False, **kw)
    PX.FUSE = False
    try:
        _, df, g = _run(m, opt, feats, xyz, tab, w, native=True, **kw)
    finally:
        PX.FUSE = True
    assert torch.equal(df, ref_df)
    assert all(torch.equal(g[n], ref_g[n]) for n in ref_g)


def test_training_step_through_the_native_point_set_half():
    """A whole optimizer step (trunk executor + point-set executor + fused losses + FlatAdamW) with the native point-set half in its
    bit-exact mode equals the step that runs the point-set half one autograd node at a time: same losses, same updated parameters."""
    from dreg_nerf_amd.train_step import TrainStep

    def one(native, profile="default", steps=2):
        torch.manual_seed(3407)
        m = NeRFRegTr(precision="bf16")
        m.load_state_dict(params.synth_state_dict(0, profile=profile), strict=True)
        m = m.cuda().train()
        m.native_pointset = native
        ts = TrainStep(m)
        batch = []
        for i in range(2):
            d = synth.shell_pair(64, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
            batch.append({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()})
        out = None
        for _ in range(steps):
            out = ts.step(batch)
        torch.cuda.synchronize()
        return {k: float(v) for k, v in out["losses"].items()}, [p.detach().clone() for p in m.parameters()], float(out["grad_norm"])

    PX.FUSE = False
    try:
        la, pa, na = one(True)
    finally:
        PX.FUSE = True
    lb, pb, nb = one(False)
    assert la == lb and na == nb
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
    # and the product setting (fused epilogues) lands within bf16 rounding of it — on the well-conditioned weight profile (on the default
    # initialisation one bf16 rounding moves the gradient norm by percent: params.PROFILES, tools/wc_profile_sweep.py)
    lc, pc, nc = one(True, "wc", 1)
    ld, pd, nd = one(False, "wc", 1)
    assert abs(nc - nd) <= 5e-3 * abs(nd), (nc, nd)
    for k in ld:
        assert abs(lc[k] - ld[k]) <= 1e-6 * max(1.0, abs(ld[k])), k      # the forward pass is bit-identical



def test_last_layer_only_backward_equals_the_full_backward():
    """The training losses read the last layer's outputs only (train_nerf_regtr.py:178,195,205-206,214,220): through the *_last outputs the
    backward pass differentiates heads / decoder / final norm for that layer's R rows; through slices of the full outputs it runs over
    all 6R rows, five sixths of them with a zero gradient.  Same gradients (the split sums of two weight gradients are formed in another
    order, nothing else differs)."""
    segs = [(200, 180), (150, 170)]
    m, opt = _model()
    tab = A.ProblemTable(segs, torch.device("cuda"))
    feats, xyz, w = _inputs(segs, seed=7)
    P = m._P()
    A.set_precision("bf16")
    res = []
    for last in (False, True):
        opt.zero_grad()
        feats.grad = None
        ex = PX.executor_for(m, P)
        cond, corr, ov, cl, rl, ol = PX.encode_decode(ex, feats, xyz, m.position_embedding(xyz), tab, P["transformer_encoder.norm.weight"], with_last=True)
        assert torch.equal(cl, cond[-1]) and torch.equal(rl, corr[-1]) and torch.equal(ol, ov[-1])
        c, r, o = (cl, rl, ol) if last else (cond[-1], corr[-1], ov[-1])
        loss = (c * w[0][-1]).sum() * 1e-2 + (r * w[1][-1]).sum() + (o * w[2][-1]).sum()
        loss.backward()
        torch.cuda.synchronize()
        res.append((feats.grad.clone(), {n: P[n].grad.clone() for n in PX.param_names()}))
    (df_a, g_a), (df_b, g_b) = res
    assert torch.equal(df_a, df_b), "the encoder sees the identical gradient"
    for n in g_a:
        d = float((g_a[n] - g_b[n]).norm() / g_a[n].norm().clamp_min(1e-20))
        if "q_proj" in n or "k_proj" in n or "transformer_encoder.norm" in n or "conf_logits" in n:
            assert d < 1e-5, (n, d)          # sums over R instead of 6R rows (zeros left out): another split / block order
        else:
            assert d == 0.0, (n, d)


@pytest.mark.parametrize("segs", [[(3, 5)], [(64, 1), (1, 64), (130, 127)]])
def test_tiny_and_lopsided_point_sets(segs):
    """A pair with a handful of key points (fewer rows than one GEMM tile, one attention tile), one-point sets next to larger ones."""
    m, opt = _model()
    tab = A.ProblemTable(segs, torch.device("cuda"))
    feats, xyz, w = _inputs(segs, seed=11)
    ref_out, ref_df, ref_g = _run(m, opt, feats, xyz, tab, w, native=False)
    PX.FUSE = False
    try:
        out, df, g = _run(m, opt, feats, xyz, tab, w, native=True)
    finally:
        PX.FUSE = True
    assert all(torch.equal(a, b) for a, b in zip(out, ref_out)) and torch.equal(df, ref_df)
    assert all(torch.equal(g[n], ref_g[n]) for n in ref_g)
    out2, df2, g2 = _run(m, opt, feats, xyz, tab, w, native=True)          # the product setting
    assert all(torch.equal(a, b) for a, b in zip(out2, ref_out))
    assert float((df2 - ref_df).norm() / ref_df.norm()) < 2e-2

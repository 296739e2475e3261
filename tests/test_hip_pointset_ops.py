"""GPU: point-set kernels (attention, LayerNorm, position embedding, overlap head, voxel downsample, Kabsch, AdamW)
against plain PyTorch fp32 CPU computations / the oracle / the reference-generated golden vectors."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import attn_ops as A, lib as L, ops, params  # noqa: E402
from dreg_nerf_amd import transformer_ops as T  # noqa: E402
from oracle import regtr_oracle as O  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nq,nk", [(70, 55), (200, 333), (64, 64)])
def test_mha_fwd_bwd(nq, nk, dtype):
    g = torch.Generator().manual_seed(nq * 7 + nk)
    H, E = 8, 256
    qp = torch.randn(nq, 3 * E, generator=g)
    kp = torch.randn(nk, 3 * E, generator=g)
    if dtype == torch.bfloat16:
        qp, kp = qp.bfloat16().float(), kp.bfloat16().float()
    qr, kr = qp.clone().requires_grad_(True), kp.clone().requires_grad_(True)
    sc = 1.0 / math.sqrt(32)
    q = qr[:, :E].view(nq, H, 32).transpose(0, 1)
    k = kr[:, E:2 * E].view(nk, H, 32).transpose(0, 1)
    v = kr[:, 2 * E:].view(nk, H, 32).transpose(0, 1)
    ref = (torch.softmax(q @ k.transpose(1, 2) * sc, -1) @ v).transpose(0, 1).reshape(nq, E)
    go = torch.randn(nq, E, generator=g)
    if dtype == torch.bfloat16:
        go = go.bfloat16().float()
    ref.backward(go)
    qd, kd = qp.to(DEV, dtype).requires_grad_(True), kp.to(DEV, dtype).requires_grad_(True)
    out = A.mha_packed(qd, kd, H, sc)
    out.backward(go.to(DEV, dtype))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref.detach().numpy(), atol=tol * float(ref.abs().max()))
    np.testing.assert_allclose(qd.grad.float().cpu().numpy(), qr.grad.numpy(), atol=tol * float(qr.grad.abs().max()))
    np.testing.assert_allclose(kd.grad.float().cpu().numpy(), kr.grad.numpy(), atol=tol * float(kr.grad.abs().max()))


def test_mha_self_packed():
    g = torch.Generator().manual_seed(1)
    n, H, E = 150, 8, 256
    x = torch.randn(n, 3 * E, generator=g)
    xr = x.clone().requires_grad_(True)
    sc = 1.0 / math.sqrt(32)
    q, k, v = [xr[:, i * E:(i + 1) * E].view(n, H, 32).transpose(0, 1) for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(1, 2) * sc, -1) @ v).transpose(0, 1).reshape(n, E)
    go = torch.randn(n, E, generator=g)
    ref.backward(go)
    xd = x.to(DEV).requires_grad_(True)
    out = A.mha_packed(xd, xd, H, sc)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), atol=2e-5 * float(xr.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_corr_attention(dtype):
    g = torch.Generator().manual_seed(2)
    nl, nq, nk = 6, 90, 130
    q = torch.randn(nl, nq, 256, generator=g)
    k = torch.randn(nl, nk, 256, generator=g)
    xyz = torch.randn(nk, 3, generator=g)
    if dtype == torch.bfloat16:
        q, k = q.bfloat16().float(), k.bfloat16().float()
    qr, kr = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    sc = 1.0 / 16.0
    ref = torch.softmax(qr @ kr.transpose(1, 2) * sc, -1) @ xyz
    go = torch.randn(nl, nq, 3, generator=g)
    ref.backward(go)
    qd, kd = q.to(DEV, dtype).requires_grad_(True), k.to(DEV, dtype).requires_grad_(True)
    out = A.attention_xyz(qd, kd, xyz.to(DEV), sc)
    out.backward(go.to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(qd.grad.float().cpu().numpy(), qr.grad.numpy(), atol=tol * float(qr.grad.abs().max()))
    np.testing.assert_allclose(kd.grad.float().cpu().numpy(), kr.grad.numpy(), atol=tol * float(kr.grad.abs().max()))


def test_layernorm_pe_fwd_bwd():
    g = torch.Generator().manual_seed(3)
    n = 333
    x = torch.randn(n, 256, generator=g) * 2 + 0.3
    pe = torch.randn(n, 256, generator=g)
    w, b = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (256,), wr, br, 1e-5) + pe
    go = torch.randn(n, 256, generator=g)
    ref.backward(go)
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = A.layer_norm(xd, wd, bd, pe.to(DEV), out_dtype=torch.float32)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.numpy(), atol=2e-4)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), atol=2e-4)


def test_small_ops_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "small_ops.npz"))
    pe = A.posenc_sine(torch.from_numpy(g["pe_xyz"]).to(DEV))
    np.testing.assert_allclose(pe.cpu().numpy(), g["pe"], atol=2e-5)
    T_ = A.weighted_kabsch(torch.from_numpy(g["kab_a"]).to(DEV), torch.from_numpy(g["kab_b"]).to(DEV), torch.from_numpy(g["kab_w"]).to(DEV))
    np.testing.assert_allclose(T_.cpu().numpy(), g["kab_T"], atol=1e-5)


def test_kabsch_reflection_case():
    """points on a plane + noise-free mirrored target: the det<0 branch of se3.py:128-134 must be taken."""
    g = torch.Generator().manual_seed(8)
    a = torch.randn(2, 40, 3, generator=g)
    a[..., 2] *= 0.01
    Rm = torch.diag(torch.tensor([1.0, 1.0, -1.0]))
    b = a @ Rm.T + 0.1
    w = torch.rand(2, 40, generator=g)
    ref = O.weighted_kabsch(a, b, w)
    got = A.weighted_kabsch(a.to(DEV), b.to(DEV), w.to(DEV)).cpu()
    assert torch.det(got[:, :, :3]).min() > 0.99
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-4)


def test_overlap_head():
    g = torch.Generator().manual_seed(4)
    f = torch.randn(6, 77, 256, generator=g)
    w, b = torch.randn(1, 256, generator=g) * 0.1, torch.randn(1, generator=g)
    fr, wr, br = f.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.sigmoid(F.linear(fr, wr, br))
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    fd, wd, bd = f.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = A.overlap_head(fd, wd, bd)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), fr.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), atol=2e-5)


def test_voxel_downsample_vs_oracle():
    g = torch.Generator().manual_seed(5)
    ns, nt = 3000, 2500
    pts = (torch.rand(ns + nt, 3, generator=g) - 0.5) * 2.4
    pts[:10] = torch.tensor([0.049999, -0.05, 0.1])  # duplicates on a cell border
    feats = torch.randn(ns + nt, 256, generator=g)
    fr = feats.clone().requires_grad_(True)
    po, fo, lo = O.hierarchical_grid_subsample(pts, fr, torch.tensor([ns, nt]))
    go = torch.randn(fo.shape, generator=g)
    fo.backward(go)
    fd = feats.to(DEV).requires_grad_(True)
    pg, fg, lg = T.hierarchical_grid_subsample(pts.to(DEV), fd, [ns, nt])
    assert lg.tolist() == lo.tolist()
    fg.backward(go.to(DEV))
    np.testing.assert_allclose(pg.detach().cpu().numpy(), po.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(fg.detach().cpu().numpy(), fo.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), fr.grad.numpy(), atol=1e-6)


def test_subsample_plan_apply_equals_fused_path():
    """plan (xyz only, before the feature network) + segment means == the one-call downsample, bit for bit, incl. gradients."""
    g = torch.Generator().manual_seed(11)
    ns, nt = 4100, 3900
    pts = ((torch.rand(ns + nt, 3, generator=g) - 0.5) * 2.4).to(DEV)
    feats = torch.randn(ns + nt, 256, generator=g).to(DEV)
    f1 = feats.clone().requires_grad_(True)
    p_a, f_a, l_a = T.hierarchical_grid_subsample(pts, f1, [ns, nt])
    rounds, p_b, l_b = T.plan_hierarchical_subsample(pts, [ns, nt])
    f2 = feats.clone().requires_grad_(True)
    f_b = T.apply_subsample_plan(rounds, f2)
    assert [int(v) for v in l_a] == [int(v) for v in l_b] and len(rounds) >= 1
    assert torch.equal(p_a, p_b) and torch.equal(f_a, f_b)
    go = torch.randn(f_a.shape, generator=g).to(DEV)
    f_a.backward(go)
    f_b.backward(go)
    assert torch.equal(f1.grad, f2.grad)


def test_transformer_decoder_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "transformer.npz"))
    sd = {k: v.to(DEV) for k, v in params.synth_state_dict(0).items()}
    A.set_precision("fp32")
    s_xyz, t_xyz = torch.from_numpy(g["s_xyz"]).to(DEV), torch.from_numpy(g["t_xyz"]).to(DEV)
    s_pe, t_pe = A.posenc_sine(s_xyz), A.posenc_sine(t_xyz)
    with torch.no_grad():
        sc, tc = T.cross_encoder(sd, torch.from_numpy(g["s_f"]).to(DEV), torch.from_numpy(g["t_f"]).to(DEV), s_pe, t_pe)
        s_corr, t_corr, s_ov, t_ov = T.corr_decoder(sd, sc, tc, s_xyz, t_xyz, s_pe, t_pe)
    np.testing.assert_allclose(sc.cpu().numpy(), g["s_cond"], atol=1e-4)
    np.testing.assert_allclose(tc.cpu().numpy(), g["t_cond"], atol=1e-4)
    np.testing.assert_allclose(s_corr.cpu().numpy(), g["s_corr"], atol=1e-4)
    np.testing.assert_allclose(t_corr.cpu().numpy(), g["t_corr"], atol=1e-4)
    np.testing.assert_allclose(s_ov.cpu().numpy(), g["s_ov"], atol=1e-5)


def test_learned_position_embedding_vs_golden(golden_dir):
    """NeRFRegTr(pos_emb_type != 'sine') (nerf_regtr.py:87-90): the learned MLP embedding through the batched encoder + decoder,
    forward and the gradients that reach its parameters through the LayerNorm(+pe) kernels, against the reference's vectors; and
    the coordinate scale of the sine embedding (position_embedding.py:28,43)."""
    from dreg_nerf_amd.regtr import NeRFRegTr
    g = np.load(os.path.join(golden_dir, "pos_embed.npz"))
    m = NeRFRegTr("learned", 256, 1.0, precision="fp32")
    assert list(m.state_dict().keys()) == list(params.regtr_spec("learned").keys())
    m.load_state_dict(params.synth_state_dict(0, "learned"))
    m = m.to(DEV).train()
    A.set_precision("fp32")
    s_xyz, t_xyz = torch.from_numpy(g["s_xyz"]).to(DEV), torch.from_numpy(g["t_xyz"]).to(DEV)
    ns, nt = s_xyz.shape[0], t_xyz.shape[0]
    half = NeRFRegTr("sine", 256, 0.5, precision="fp32")
    np.testing.assert_allclose(half.position_embedding(s_xyz).cpu().numpy(), g["sine_scale_half"], atol=2e-5)
    xyz = torch.cat([s_xyz, t_xyz])
    np.testing.assert_allclose(m.position_embedding(xyz)[:ns].detach().cpu().numpy(), g["s_pe"], atol=1e-5)
    feats = torch.cat([torch.from_numpy(g["s_f"]), torch.from_numpy(g["t_f"])]).to(DEV)
    tab = A.ProblemTable([(ns, nt)], DEV)
    cond, corr, ov = T.encode_decode_batched(m._P(), feats, xyz, tab, m.position_embedding)
    np.testing.assert_allclose(cond[:, :ns].detach().cpu().numpy(), g["s_cond"], atol=1e-4)
    np.testing.assert_allclose(cond[:, ns:].detach().cpu().numpy(), g["t_cond"], atol=1e-4)
    np.testing.assert_allclose(corr.detach().cpu().numpy(), g["corr"], atol=1e-4)
    np.testing.assert_allclose(ov.detach().cpu().numpy(), g["ov"], atol=1e-5)
    loss = (corr * torch.from_numpy(g["w_corr"]).to(DEV)).sum() + (ov * torch.from_numpy(g["w_ov"]).to(DEV)).sum()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), atol=2e-3)
    loss.backward()
    ops.flush_wgrad_reduce()
    torch.cuda.synchronize()
    P = dict(m.named_parameters())
    for i in range(5):
        for kind, key in (("weight", f"g_w{i}"), ("bias", f"g_b{i}")):
            got = P[f"pos_embed.mlp.{2 * i}.{kind}"].grad.cpu().numpy()
            np.testing.assert_allclose(got, g[key], atol=1e-3 * max(1.0, float(np.abs(g[key]).max())))


def test_adamw_and_grad_norm_match_torch():
    lib = L.load()
    g = torch.Generator().manual_seed(6)
    n = 100003
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=1e-2)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    norm, ws = torch.zeros(1, device=DEV), torch.zeros(1024, device=DEV)
    for step in (1, 2, 3):
        ref_p.grad = gr.clone() * step
        tn = torch.nn.utils.clip_grad_norm_([ref_p], 0.1)
        opt.step()
        gd = (gr * step).to(DEV)
        L.check(lib.dreg_grad_norm(L.ptr(gd), L.ptr(norm), L.ptr(ws), n, L.stream()), "dreg_grad_norm")
        np.testing.assert_allclose(float(norm), float(tn), rtol=1e-5)
        L.check(lib.dreg_adamw_step(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), L.ptr(norm), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, 0.1, L.stream()), "dreg_adamw_step")
        np.testing.assert_allclose(p.cpu().numpy(), ref_p.detach().numpy(), atol=2e-6)


@pytest.mark.parametrize("n,nbatch", [(7, 1), (1023, 2), (1025, 2), (38352, 2), (70001, 3), (131072, 2)])
def test_own_sort_and_segments_equal_the_rocprim_form_bit_for_bit(n, nbatch, probe_lib):
    """One launch (keys + stable 4-bit LSD radix sort + heads + scan + starts + batch counts in one workgroup: voxel_sort_segments_kernel)
    against rocPRIM's radix_sort_pairs + inclusive_scan with the small kernels around them: every output of the plan identical — the
    averaged points, their order, the segment table, the per-batch counts, the inverse maps."""
    from dreg_nerf_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(n)
    lens = [n // nbatch] * (nbatch - 1) + [n - (n // nbatch) * (nbatch - 1)]
    pts = (torch.rand(n, 3, generator=g) - 0.5) * 3.0
    pts[: min(n, 40)] = pts[0]                                   # many duplicates of one cell
    if n > 100:
        pts[50:60] = torch.tensor([-1.4999, 0.05, -0.0500001])     # on cell borders, negative coordinates
    pts = pts.to(DEV)
    res = []
    for own in (0, 1):
        lib.dreg_voxel_set_own_sort(own)
        try:
            rnd, p, counts = A.plan_voxel_downsample(pts, lens, 0.05)
        finally:
            lib.dreg_voxel_set_own_sort(0)
        torch.cuda.synchronize()
        res.append((rnd.n_out, [int(c) for c in counts], p.clone(), rnd.order.clone(), rnd.starts[:rnd.n_out + 1].clone(), rnd.inv_seg.clone(), rnd.inv_cnt.clone()))
    a, b = res
    assert a[0] == b[0] and a[1] == b[1] and sum(a[1]) == a[0]
    for x, y in zip(a[2:], b[2:]):
        assert torch.equal(x, y)
    # and the order is a stable sort by (batch, ix, iy, iz): ascending original index inside a cell
    order = a[3].long().cpu()
    cell = torch.floor(pts.cpu() / 0.05).long()
    pb = torch.repeat_interleave(torch.arange(nbatch), torch.tensor(lens))
    key = ((pb * 65536 + cell[:, 0] + 32768) * 65536 + cell[:, 1] + 32768) * 65536 + cell[:, 2] + 32768
    ks = key[order]
    assert torch.all(ks[1:] >= ks[:-1])
    same = ks[1:] == ks[:-1]
    assert torch.all(order[1:][same] > order[:-1][same])


def test_batched_subsample_and_fused_gather_nodes_change_nothing():
    """attn_ops.subsample_all (every pair's voxel-average rounds as ONE autograd node that writes results / gradients in place) and
    attn_ops.gather_subsample (the trilinear gather in front of them in the same node: its backward reads the features' gradient through
    the first round instead of a materialised [N, 256] tensor) against the per-pair nodes + torch.cat / split: the same arithmetic in the
    same order — losses, gradient norm and every gradient bit for bit."""
    from dreg_nerf_amd import params, synth
    from dreg_nerf_amd.regtr import NeRFRegTr
    from dreg_nerf_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    res = []
    for batched, fused in ((False, False), (True, False), (True, True)):
        torch.manual_seed(3407)
        m = NeRFRegTr(precision="bf16")
        m.load_state_dict(params.synth_state_dict(0, profile="wc"), strict=True)
        m = m.to(dev).train()
        m.batched_subsample, m.fused_gather_subsample = batched, fused
        ts = TrainStep(m)
        batch = []
        for i in range(2):
            d = synth.shell_pair(64, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
            batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
        for _ in range(2):                       # two steps: the second one runs on the persistent gradient buffer the first one dirtied
            out = ts.step(batch)
        torch.cuda.synchronize()
        res.append(({k: float(v) for k, v in out["losses"].items()}, float(out["grad_norm"]),
                    {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
    for r in res[1:]:
        assert res[0][0] == r[0] and res[0][1] == r[1]
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], r[2][k]), k

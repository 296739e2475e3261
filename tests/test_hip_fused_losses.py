"""GPU: the fused all-pairs loss node (csrc/losses.hip) against the per-pair torch formulation of dreg_nerf_amd/losses.py (which
tests/test_oracle_golden.py pins to the reference through the oracle): values and input gradients; the batched Kabsch against
the per-pair kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import attn_ops as A, fused_losses as FL, losses as LS, synth  # noqa: E402

DEV = "cuda:0"


def _case(seed, segs, robust):
    g = torch.Generator().manual_seed(seed)
    R = sum(a + b for a, b in segs)
    tab = A.ProblemTable(segs, torch.device(DEV))
    xyz = (torch.rand(R, 3, generator=g) - 0.5) * 1.6
    cond = torch.randn(6, R, 256, generator=g) * 0.3
    corr = xyz[None] + 0.1 * torch.randn(6, R, 3, generator=g)
    ov = torch.sigmoid(torch.randn(6, R, 1, generator=g))
    gt = (torch.rand(R, generator=g) > 0.4).float()[None].expand(6, -1).contiguous()
    tilde = (torch.rand(6, R, generator=g) > 0.5).float()
    poses = []
    for p in range(len(segs)):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        T = torch.eye(4)
        T[:3, :3] = 0.1 * q + 0.9 * torch.eye(3)   # not orthonormal on purpose: the losses only apply the 3x4 map (and R^T for the inverse)
        T[:3, :3] = q
        T[:3, 3] = 0.05 * torch.randn(3, generator=g)
        poses.append(T)
    return tab, xyz, cond, corr, ov, gt, tilde, torch.stack(poses)


@pytest.mark.parametrize("robust,segs", [(False, [(310, 277), (150, 401)]), (True, [(310, 277), (150, 401)]),
                                         (False, [(131, 2100), (66, 90)])])   # > 2,048 targets: the InfoNCE kernels read them from global memory; 131 % 4 != 0: a workgroup straddles two pairs
def test_fused_losses_match_per_pair_torch(robust, segs):
    tab, xyz, cond, corr, ov, gt, tilde, poses = _case(3, segs, robust)
    fl = LS.InfoNCELoss(256, 0.2, 0.4)
    torch.manual_seed(0)
    torch.nn.init.normal_(fl.W, std=0.1)
    # reference: per pair on the CPU in fp32
    c_r, k_r, o_r = cond.clone().requires_grad_(True), corr.clone().requires_grad_(True), ov.clone().requires_grad_(True)
    tot, per = 0.0, {k: 0.0 for k in FL.NAMES}
    for p, (s0, ns, t0, nt) in enumerate(tab.segs):
        pred = {"src_feats": [c_r[:, s0:s0 + ns]], "tgt_feats": [c_r[:, t0:t0 + nt]], "src_kp": [xyz[s0:s0 + ns]], "tgt_kp": [xyz[t0:t0 + nt]],
                "src_kp_warped": [k_r[:, s0:s0 + ns]], "tgt_kp_warped": [k_r[:, t0:t0 + nt]],
                "src_overlap": [o_r[:, s0:s0 + ns]], "tgt_overlap": [o_r[:, t0:t0 + nt]]}
        ls = LS.training_losses(pred, poses[p][None], fl, gt[:, s0:s0 + ns, None], gt[:, t0:t0 + nt, None],
                                tilde[:, s0:s0 + ns, None], tilde[:, t0:t0 + nt, None], robust)
        tot = tot + ls["total"]
        for k in FL.NAMES:
            per[k] += float(ls[k])
    tot = tot / len(segs)
    tot.backward()
    # fused
    fd = LS.InfoNCELoss(256, 0.2, 0.4).to(DEV)
    with torch.no_grad():
        fd.W.copy_(fl.W)
    c_d, k_d, o_d = (t.to(DEV).requires_grad_(True) for t in (cond, corr, ov))
    bt = {"cond": c_d, "corr": k_d, "ov": o_d, "xyz": xyz.to(DEV), "tab": tab}
    out = FL.regtr_losses(bt, poses.to(DEV), fd, gt.to(DEV), tilde.to(DEV), robust)
    out["total"].backward()
    for k in FL.NAMES:
        assert abs(float(out[k]) - per[k] / len(segs)) <= 2e-5 * max(1.0, abs(per[k])), (k, float(out[k]), per[k] / len(segs))
    np.testing.assert_allclose(o_d.grad.cpu().numpy(), o_r.grad.numpy(), atol=1e-7)
    np.testing.assert_allclose(k_d.grad.cpu().numpy(), k_r.grad.numpy(), rtol=2e-6, atol=1e-7)
    gc, gr = c_d.grad.cpu(), c_r.grad
    assert float((gc - gr).abs().max()) <= 2e-5 * float(gr.abs().max()) + 1e-9


def test_kabsch_pairs_equals_per_pair_kernel():
    segs = [(310, 277), (150, 401), (64, 64)]
    tab, xyz, cond, corr, ov, gt, tilde, poses = _case(5, segs, False)
    xyz_d, corr_d, ov_d = xyz.to(DEV), corr.to(DEV), ov.to(DEV)
    got = A.weighted_kabsch_pairs(xyz_d, corr_d, ov_d, tab)
    for p, (s0, ns, t0, nt) in enumerate(tab.segs):
        a = torch.cat([xyz_d[s0:s0 + ns].expand(6, -1, -1), corr_d[:, t0:t0 + nt]], dim=1)
        b = torch.cat([corr_d[:, s0:s0 + ns], xyz_d[t0:t0 + nt].expand(6, -1, -1)], dim=1)
        w = torch.cat([ov_d[:, s0:s0 + ns, 0], ov_d[:, t0:t0 + nt, 0]], dim=1)
        ref = A.weighted_kabsch(a, b, w)
        assert torch.equal(got[p], ref)


@pytest.mark.parametrize("tA,tB,M,N,K", [(0, 0, 310, 256, 256), (0, 1, 310, 277, 256), (0, 0, 131, 256, 2101), (1, 0, 2101, 256, 131),
                                          (1, 1, 65, 63, 17), (0, 1, 1, 1, 1)])
def test_batched_fp32_gemm_all_operand_forms(tA, tB, M, N, K):
    """dreg_gemm_f32_batched (exact-fp32 MFMA, csrc/losses.hip): two records per launch (the same problem at two offsets) against torch in fp64, ragged
    sizes, leading dimensions that are / are not multiples of four (vector and scalar load paths)."""
    import ctypes
    import struct
    from dreg_nerf_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    for pad in (0, 3):
        lda = (M if tA else K) + pad
        ldb = (K if tB else N) + pad
        ldc = N + pad
        a = torch.randn((K if tA else M), lda, generator=g).to(DEV)
        b = torch.randn((N if tB else K), ldb, generator=g).to(DEV)
        c = torch.full((2, M, ldc), float("nan"), device=DEV)
        a2 = torch.cat([a.reshape(-1), a.reshape(-1) * 2.0])             # second record: the same A scaled, at an offset
        tiles1 = ((M + 63) // 64) * ((N + 63) // 64)
        blob = struct.pack("<12i3q", 0, 1, 2, tA, tB, M, N, K, lda, ldb, ldc, 0, 0, 0, 0) + \
            struct.pack("<12i3q", 0, 1, 2, tA, tB, M, N, K, lda, ldb, ldc, tiles1, a.numel(), 0, M * ldc)
        assert len(blob) == 2 * lib.dreg_gemm_f32_desc_bytes()
        tab = torch.tensor(list(blob), dtype=torch.uint8).to(DEV)
        bases = (ctypes.c_void_p * 8)(a2.data_ptr(), b.data_ptr(), c.data_ptr(), 0, 0, 0, 0, 0)
        L.check(lib.dreg_gemm_f32_batched(L.ptr(tab), 2, 2 * tiles1, bases, L.stream()), "dreg_gemm_f32_batched")
        A_ = (a[:, :M].T if tA else a[:, :K]).double()
        B_ = (b[:, :K].T if tB else b[:, :N]).double()
        ref = A_ @ B_
        got = c[:, :, :N].double()
        tol = 2e-6 * float(ref.abs().max()) * max(1.0, K ** 0.5 / 8) + 1e-7
        assert float((got[0] - ref).abs().max()) <= tol
        assert float((got[1] - 2.0 * ref).abs().max()) <= 2 * tol
        if pad:
            assert torch.isnan(c[:, :, N:]).all()                          # nothing outside the [M x N] block is written


def test_fused_synthetic_labels_equal_the_elementwise_formula_bit_for_bit():
    """dreg_halfspace_labels (one launch) against synth.synthetic_overlap_gt and the training step's former torch expression — including points within
    a few ulps of the plane, where a fused multiply-add would flip labels."""
    from dreg_nerf_amd import synth
    from dreg_nerf_amd.train_step import synthetic_labels_rows
    g = torch.Generator().manual_seed(5)
    R, L_ = 10007, 6
    xyz = (torch.rand(R, 3, generator=g) * 3 - 1.5)
    corr = (torch.rand(L_, R, 3, generator=g) * 3 - 1.5)
    # rows ON the plane up to rounding: x = 0.0123 - 0.31 y + 0.17 z evaluated in fp32, then nudged by -2..2 ulps
    for t in (xyz, corr.view(-1, 3)):
        n = t.shape[0] // 2
        x0 = (torch.tensor(0.0123) - 0.31 * t[:n, 1]) + 0.17 * t[:n, 2]
        ulps = torch.randint(-2, 3, (n,), generator=g)
        t[:n, 0] = (x0.view(torch.int32) + ulps.int()).view(torch.float32)
    xyz, corr = xyz.to(DEV), corr.to(DEV)
    gt, tilde = synthetic_labels_rows(xyz, corr)
    want_gt = synth.synthetic_overlap_gt(xyz, L_)[..., 0]
    want_tilde = (corr[..., 0] + 0.31 * corr[..., 1] - 0.17 * corr[..., 2] > 0.0123).float()
    assert torch.equal(gt, want_gt) and torch.equal(tilde, want_tilde)
    assert 0.2 < float(tilde.mean()) < 0.8 and 0.2 < float(want_gt.mean()) < 0.8
    cpu_gt, cpu_tilde = synthetic_labels_rows(xyz.cpu(), corr.cpu())                 # the CPU branch (gloo tests) is the formula itself
    assert torch.equal(cpu_gt, want_gt.cpu()) and torch.equal(cpu_tilde, want_tilde.cpu())

"""GPU: each HIP kernel of the FPN3D path against a plain PyTorch fp32 CPU computation of the same op
(through the C ABI via dreg_nerf_amd.ops).  fp32 mode = exact-f32 MFMA (tolerance: fp32 round-off);
bf16 mode is compared against the fp32 result on bf16-rounded operands."""
import numpy as np
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import ops  # noqa: E402


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def ndhwc(t):  # NCDHW -> NDHWC
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


CONV_CASES = [
    # B, (D,H,W), Cin_real, Cin_pad, Cout, k, stride, pad
    (1, (8, 8, 8), 64, 64, 64, 1, 1, 0),
    (2, (9, 10, 11), 64, 64, 128, 3, 1, 1),
    (1, (12, 9, 10), 128, 128, 128, 3, 2, 1),
    (2, (7, 8, 9), 256, 256, 256, 1, 2, 0),
    (1, (16, 14, 12), 4, 8, 64, 5, 2, 2),
    (1, (6, 6, 6), 256, 256, 256, 3, 1, 1),
    (1, (1, 1, 300), 256, 256, 768, 1, 1, 0),
    (2, (10, 12, 8), 64, 64, 256, 3, 2, 1),     # stride-2 data gradient through the parity-class form, even dims
    (1, (5, 6, 7), 512, 512, 64, 1, 2, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_fwd_bwd(case, dtype):
    dev = _dev()
    B, dims, cin, cin_pad, cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(B, cin, *dims, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        wq = w.bfloat16().float()
    else:
        wq = w
    xr = x.clone().requires_grad_(cin > 4)
    wr = wq.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, br, stride=s, padding=p)
    gy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    y_ref.backward(gy)

    xd = ndhwc(F.pad(x, (0, 0, 0, 0, 0, 0, 0, cin_pad - cin))).to(dev, dtype).requires_grad_(cin > 4)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv3d(xd, wd, bd, None, s, p)
    y.backward(ndhwc(gy).to(dev, dtype))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    yscale = float(y_ref.abs().max())
    np.testing.assert_allclose(ncdhw(y.detach().float().cpu()).numpy(), y_ref.detach().numpy(), atol=tol * yscale)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.numpy(), atol=tol * float(wr.grad.abs().max()) * 2)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), atol=tol * float(br.grad.abs().max()) * 2)
    if cin > 4:
        gx = ncdhw(xd.grad.float().cpu())[:, :cin]
        np.testing.assert_allclose(gx.numpy(), xr.grad.numpy(), atol=tol * float(xr.grad.abs().max()) * 2)


def test_wgrad_transpose_read_matches_gather_path():
    """bf16 weight gradient: ds_read_b64_tr_b16 fragments == eight 16-bit LDS reads, bit for bit."""
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 10, 11, 128, generator=g).to(dev, torch.bfloat16)
    gy = torch.randn(2, 9, 10, 11, 256, generator=g).to(dev, torch.bfloat16)
    a = ops.conv_wgrad(gy, x, (256, 128, 3, 3, 3), 128, 3, 1, 1, use_tr=True)
    b = ops.conv_wgrad(gy, x, (256, 128, 3, 3, 3), 128, 3, 1, 1, use_tr=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [
    (2, 9, 10, 11, 128, 256, 3),       # 3^3: LDS-transposed rows of 64 input channels
    (1, 40, 33, 17, 64, 64, 3),        # many splits
    (3, 5, 6, 7, 256, 128, 1),         # 1^3, float4 route
    (700, 1, 1, 1, 256, 768, 1),       # a transformer linear over 700 key points
    (2, 12, 12, 12, 8, 64, 5),         # the stem: 4 real input channels padded to 8, 125 taps
])
def test_deferred_wgrad_batched_reduce_is_bit_identical(shape):
    """dreg_conv3d_wgrad_partials + ONE dreg_wgrad_reduce_batched over several layers == the per-layer weight gradient + its own
    reduce launch, bit for bit (same split order, same accumulation into an existing gradient)."""
    dev = _dev()
    B, D, H, W, cin, cout, k = shape
    g = torch.Generator().manual_seed(11)
    cin_real = 4 if cin == 8 else cin
    x = torch.randn(B, D, H, W, cin, generator=g).to(dev, torch.bfloat16)
    if cin_real != cin:
        x[..., cin_real:] = 0
    stride, pad = (2, 2) if k == 5 else (1, k // 2)
    Do = (D + 2 * pad - k) // stride + 1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randn(B, Do, Ho, Wo, cout, generator=g).to(dev, torch.bfloat16)
    wshape = (cout, cin_real, k, k, k)
    base = torch.randn(wshape, generator=g).to(dev)
    a1, a2 = base.clone(), base.clone()
    ops.conv_wgrad(gy, x, wshape, cin, k, stride, pad, accumulate_into=a1)
    ops.conv_wgrad(gy, x, wshape, cin, k, stride, pad, accumulate_into=a1)
    # three deferred launches over two layers (one layer applied twice: its gradient accumulates twice, in two successive reduce
    # launches; the other layer shares the first launch)
    ops.conv_wgrad(gy, x, wshape, cin, k, stride, pad, accumulate_into=a2, defer=True)
    b2 = torch.zeros(64, 64, 1, 1, 1, device=dev)
    xo = torch.randn(3, 4, 4, 4, 64, generator=g).to(dev, torch.bfloat16)
    ops.conv_wgrad(xo, xo, (64, 64, 1, 1, 1), 64, 1, 1, 0, accumulate_into=b2, defer=True)
    ops.conv_wgrad(gy, x, wshape, cin, k, stride, pad, accumulate_into=a2, defer=True)
    assert len(ops._PENDING_REDUCE) == 3
    ops.flush_wgrad_reduce()
    assert not ops._PENDING_REDUCE
    assert torch.equal(a1, a2)
    assert torch.equal(b2, ops.conv_wgrad(xo, xo, (64, 64, 1, 1, 1), 64, 1, 1, 0))


@pytest.mark.parametrize("cin,cout,D,fill", [(256, 256, 32, 0.12), (64, 256, 32, 0.3), (128, 128, 16, 0.3)])
def test_deferred_row_list_weight_gradient_sums_only_the_written_slices(cin, cout, D, fill):
    """A row-list launch of dreg_conv3d_wgrad_partials writes as many slices as its ROW COUNT is worth (fewer than the dense rule's
    dreg_conv3d_wgrad_splits, which sizes the workspace and the descriptor) and stores that count behind the slices; a descriptor
    with accumulate bit 1 makes dreg_wgrad_reduce_batched sum those only.  The workspace is filled with NaN first: a sum that
    touched an unwritten slice would show.  Against the immediate form (dreg_conv3d_wgrad_rows: same splits, its own reduce)."""
    import numpy as np
    from dreg_nerf_amd import lib as L
    lib = L.load()
    DEV = _dev()
    B = 8
    g = torch.Generator().manual_seed(cin + D + 5)
    keep = torch.rand(B * D ** 3, generator=g) < fill
    keep[:3] = True; keep[-2:] = True
    rows = keep.nonzero().flatten().int().to(DEV)
    n = rows.shape[0]
    x = torch.randn(B, D, D, D, cin, generator=g).to(DEV).bfloat16()
    gy = torch.randn(B, D, D, D, cout, generator=g).to(DEV).bfloat16()
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, D, D, D, cin, cout, 3, 0)
    smax = lib.dreg_conv3d_wgrad_splits(B, D, D, D, cin, cout, 3, 0)
    kpad = lib.dreg_conv3d_kpad(3, cin, 0)
    assert nbytes >= smax * cout * kpad * 4 + 4
    ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=DEV)
    base = torch.randn(cout, cin, 3, 3, 3, generator=g).to(DEV)
    want = base.clone()
    ws2 = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    L.check(lib.dreg_conv3d_wgrad_rows(L.ptr(gy), L.ptr(x), L.ptr(want), L.ptr(ws2), nbytes, L.ptr(rows), n, B, D, D, D, cin, cin, D, D, D, cout, 3, 1, 1, 1,
                                       L.stream()), "dreg_conv3d_wgrad_rows")
    L.check(lib.dreg_conv3d_wgrad_partials(L.ptr(gy), L.ptr(x), L.ptr(ws), nbytes, L.ptr(rows), n, B, D, D, D, cin, cin, D, D, D, cout, 3, 1, 1, None,
                                           L.stream()), "dreg_conv3d_wgrad_partials")
    written = int(ws.view(torch.int32)[smax * cout * kpad].item())
    assert 1 <= written < smax                              # the list is a fraction of the volume: fewer slices than the dense rule
    assert torch.isnan(ws[written * cout * kpad:smax * cout * kpad]).all()    # the others were not touched
    got = base.clone()
    rec = np.zeros(1, dtype=ops._REDUCE_DT)
    rec[0] = (ws.data_ptr(), got.data_ptr(), smax, cout, kpad, 27, cin, cin, 3, 0)
    table = torch.from_numpy(rec.view(np.uint8)).to(DEV)
    L.check(lib.dreg_wgrad_reduce_batched(L.ptr(table), 1, 0, lib.dreg_wgrad_reduce_blocks(cout, cin, 3, smax), L.stream()), "dreg_wgrad_reduce_batched")
    assert torch.isfinite(got).all()
    # the same slices summed in a differently grouped order (the batched sum partitions its work by the table's split count)
    assert float((got - want).abs().max()) <= 1e-5 * float((want - base).abs().max())


@pytest.mark.parametrize("B,D,cin,cout,k,stride,emits", [
    (2, 16, 64, 128, 1, 1, True),          # 64 row tiles x 1: the eight-wave anti-phase form of the 128 x 128 tile
    (2, 16, 128, 128, 3, 1, True),
    (8, 32, 64, 256, 1, 1, True),          # 262,144 rows, 256 channels: the 256 x 256 tile (chunks of 256 voxels)
    (8, 32, 256, 64, 1, 1, True),          # 64 output channels: the 128 x 64 tile
    (2, 32, 128, 128, 3, 2, True),         # stride 2 (the first 3^3 convolution of layer2)
    (8, 8, 256, 256, 3, 1, False),         # split-K launch: no sums, the caller runs the ordinary BatchNorm
])
def test_convolution_epilogue_leaves_the_batchnorm_sums(B, D, cin, cout, k, stride, emits, probe_lib):
    """dreg_conv3d_igemm_bnstats: the convolution's output is the plain launch's, bit for bit, and the chunk sums it leaves are the
    sums / sums of squares of the STORED bf16 values per (grid, chunk, channel); dreg_bn3d_fwd_from_sums on them equals the
    three-pass BatchNorm up to the summation order of the statistics."""
    from dreg_nerf_amd import lib as L
    lib = L.load()
    dev = _dev()
    g = torch.Generator().manual_seed(B + D + cin + cout)
    x = torch.randn(B, D, D, D, cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).to(dev)
    wpk = ops.packed_weight(w, cin, False, 0)
    pad = k // 2
    Do = (D + 2 * pad - k) // stride + 1
    V = Do ** 3
    want = ops.conv_igemm(x, wpk, None, None, (Do, Do, Do), cin, cout, k, stride, pad, False)
    out = torch.empty_like(want)
    sums = torch.full((B * (V // 128) * cout * 2,), float("nan"), dtype=torch.float32, device=dev)
    nws = lib.dreg_conv3d_igemm_workspace_bytes(B, D, D, D, cin, Do, Do, Do, cout, k, stride, pad, 0, 0, 0)
    ws = torch.empty(max(int(nws), 16), dtype=torch.uint8, device=dev)
    rpc = ctypes.c_int(-1)
    L.check(lib.dreg_conv3d_igemm_bnstats(L.ptr(x), L.ptr(wpk), L.ptr(out), None, None, B, D, D, D, cin, Do, Do, Do, cout, k, stride, pad, 0, 0, 0, 0, 0,
                                          L.ptr(ws), int(nws), L.ptr(sums), ctypes.addressof(rpc), L.stream()), "dreg_conv3d_igemm_bnstats")
    assert torch.equal(out, want)
    if not emits:
        assert rpc.value == 0
        return
    assert rpc.value == 128
    nch = V // rpc.value
    # a grid's sums do not depend on the form the launch's size selects: four waves instead of the eight-wave anti-phase form, and
    # (one grid alone: 32,768 rows) the 128-row tile instead of the 256 x 256 one — bit for bit
    others = []
    try:
        lib.dreg_conv_set_igemm_ap(0)
        lib.dreg_conv_set_igemm_ap256(0)
        for nb in (B, 1):
            o2 = torch.empty(nb, Do, Do, Do, cout, dtype=torch.bfloat16, device=dev)
            s2 = torch.full_like(sums, float("nan"))
            r2 = ctypes.c_int(-1)
            L.check(lib.dreg_conv3d_igemm_bnstats(L.ptr(x), L.ptr(wpk), L.ptr(o2), None, None, nb, D, D, D, cin, Do, Do, Do, cout, k, stride, pad, 0, 0, 0, 0, 0,
                                                  L.ptr(ws), int(nws), L.ptr(s2), ctypes.addressof(r2), L.stream()), "dreg_conv3d_igemm_bnstats")
            others.append((nb, o2, s2, r2.value))
    finally:
        lib.dreg_conv_set_igemm_ap(256)
        lib.dreg_conv_set_igemm_ap256(1)
    for nb, o2, s2, r2 in others:
        if r2 == 0:
            continue                                  # (one grid of a small level may run split-K: no sums, nothing to compare)
        assert torch.equal(o2, out[:nb])
        assert torch.equal(s2[:nb * nch * cout * 2], sums[:nb * nch * cout * 2]), nb
    got = sums[:B * nch * cout * 2].view(B, nch, cout, 2).double()
    ref = out.double().view(B, nch, rpc.value, cout)
    assert float((got[..., 0] - ref.sum(2)).abs().max()) <= 1e-5 * float(ref.abs().sum(2).max())
    assert float((got[..., 1] - (ref * ref).sum(2)).abs().max()) <= 1e-5 * float((ref * ref).sum(2).max())
    # BatchNorm from the sums vs the BatchNorm with its own statistics pass
    gamma, beta = torch.rand(cout, generator=g).to(dev) + 0.5, torch.randn(cout, generator=g).to(dev)
    res = torch.randn(B, Do, Do, Do, cout, generator=g).to(dev, torch.bfloat16)
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        ss, mr = torch.empty(B, cout, 2, device=dev), torch.empty(B, cout, 2, device=dev)
        y = torch.empty_like(out)
        if fused:
            L.check(lib.dreg_bn3d_fwd_from_sums(L.ptr(out), L.ptr(res), L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(ss), L.ptr(mr), L.ptr(sums),
                                                rpc.value, B, V, cout, 1e-5, 0.1, 1, 0, L.stream()), "dreg_bn3d_fwd_from_sums")
        else:
            bws = torch.empty(B * lib.dreg_bn_num_chunks(V) * cout * 2, dtype=torch.float32, device=dev)
            L.check(lib.dreg_bn3d_fwd(L.ptr(out), L.ptr(res), L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(ss), L.ptr(mr), L.ptr(bws),
                                      B, V, cout, 1e-5, 0.1, 1, 1, 0, L.stream()), "dreg_bn3d_fwd")
        outs.append((y, mr, rm, rv))
    (ya, mra, rma, rva), (yb, mrb, rmb, rvb) = outs
    assert float((mra - mrb).abs().max()) <= 1e-5 * float(mra.abs().max())
    assert float((rma - rmb).abs().max()) <= 1e-6 and float((rva - rvb).abs().max()) <= 1e-5
    assert float((ya.float() - yb.float()).abs().max()) <= 2 ** -7 * float(ya.float().abs().max())     # at most one bf16 step where a value sits on a rounding boundary
    assert (ya != yb).float().mean().item() < 1e-3


@pytest.mark.parametrize("out_f32", [False, True])
def test_narrow_tiles_for_small_launches_are_bit_identical(out_f32, probe_lib):
    """Launches with fewer 128 x 128 tiles than CUs run 128 x 64 tiles (include/dreg_nerf_probe.h: dreg_conv_set_narrow_small): every
    output element is the same K-ordered MFMA accumulation, so which tile shape a batch size lands on cannot change a result."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 1, 1, 2400, 256, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(256, 256, generator=g) / 16).to(dev)
    b = torch.randn(256, generator=g).to(dev)
    wpk = ops.packed_weight(w, 256, False, 0)
    outs = []
    try:
        for on in (1, 0):
            lib.dreg_conv_set_narrow_small(on)
            outs.append(ops.conv_igemm(x, wpk, b, None, (1, 1, 2400), 256, 256, 1, 1, 0, False, relu=True, out_f32=out_f32))
    finally:
        lib.dreg_conv_set_narrow_small(1)
    assert torch.equal(outs[0], outs[1])
    # the weight gradient of the same layer: 4 tiles x 8 splits of 128 x 128 -> 64 x 64 tiles
    gy = torch.randn(1, 1, 1, 2400, 256, generator=g).to(dev, torch.bfloat16)
    wg = []
    try:
        for on in (1, 0):
            lib.dreg_conv_set_narrow_small(on)
            wg.append(ops.conv_wgrad(gy, x, (256, 256), 256, 1, 1, 0))
    finally:
        lib.dreg_conv_set_narrow_small(1)
    assert torch.equal(wg[0], wg[1])


@pytest.mark.parametrize("shape", [(2, 12, 10, 14, 64), (1, 9, 11, 7, 16)])
def test_fused_stem_bn_relu_maxpool_matches_the_unfused_pair(shape):
    """dreg_bn_relu_maxpool_fwd / _bwd (what the trunk executor runs for conv1 -> bn1 -> relu -> maxpool) against dreg_bn3d_fwd +
    dreg_maxpool3d_fwd and their backwards: pooled values, arg-max taps, statistics and running statistics bit-identical; the input
    gradient agrees to bf16 round-off (the unfused chain rounds the un-pooled gradient to bf16 in between, the fused one does not)."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    B, D, H, W, C = shape
    Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(B, D, H, W, C, generator=g) * 2 + 0.3).to(dev, torch.bfloat16)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    dp = torch.randn(B, Do, Ho, Wo, C, generator=g).to(dev, torch.bfloat16)
    V = D * H * W
    nws = B * lib.dreg_bn_num_chunks(V) * C * 2

    def buffers():
        return dict(rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev), ss=torch.zeros(B, C, 2, device=dev), mr=torch.zeros(B, C, 2, device=dev),
                    ws=torch.zeros(nws, device=dev), coef=torch.zeros(B, C, 2, device=dev), dg=torch.zeros(C, device=dev), db=torch.zeros(C, device=dev),
                    p=torch.empty(B, Do, Ho, Wo, C, dtype=torch.bfloat16, device=dev), arg=torch.empty(B, Do, Ho, Wo, C, dtype=torch.uint8, device=dev),
                    dx=torch.empty_like(x))
    a, b = buffers(), buffers()
    st = L.stream()
    # unfused
    y = torch.empty_like(x)
    L.check(lib.dreg_bn3d_fwd(L.ptr(x), None, L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(a["rm"]), L.ptr(a["rv"]), L.ptr(a["ss"]), L.ptr(a["mr"]), L.ptr(a["ws"]),
                              B, V, C, 1e-5, 0.1, 1, 1, 0, st), "bn fwd")
    L.check(lib.dreg_maxpool3d_fwd(L.ptr(y), L.ptr(a["p"]), L.ptr(a["arg"]), B, D, H, W, Do, Ho, Wo, C, 0, st), "pool fwd")
    dy = torch.empty_like(x)
    L.check(lib.dreg_maxpool3d_bwd(L.ptr(dp), L.ptr(a["arg"]), L.ptr(dy), B, D, H, W, Do, Ho, Wo, C, 0, st), "pool bwd")
    L.check(lib.dreg_bn3d_bwd(L.ptr(x), L.ptr(dy), None, L.ptr(a["ss"]), L.ptr(a["mr"]), L.ptr(a["dx"]), None, L.ptr(a["dg"]), L.ptr(a["db"]), L.ptr(a["coef"]),
                              L.ptr(a["ws"]), B, V, C, 1, 0, 0, st), "bn bwd")
    # fused
    L.check(lib.dreg_bn_relu_maxpool_fwd(L.ptr(x), L.ptr(b["p"]), L.ptr(b["arg"]), L.ptr(gamma), L.ptr(beta), L.ptr(b["rm"]), L.ptr(b["rv"]), L.ptr(b["ss"]),
                                         L.ptr(b["mr"]), L.ptr(b["ws"]), B, D, H, W, Do, Ho, Wo, C, 1e-5, 0.1, 1, 1, st), "fused fwd")
    L.check(lib.dreg_bn_relu_maxpool_bwd(L.ptr(x), L.ptr(dp), L.ptr(b["arg"]), L.ptr(b["ss"]), L.ptr(b["mr"]), L.ptr(b["dx"]), L.ptr(b["dg"]), L.ptr(b["db"]),
                                         L.ptr(b["coef"]), L.ptr(b["ws"]), B, D, H, W, Do, Ho, Wo, C, 1, 0, st), "fused bwd")
    torch.cuda.synchronize()
    for k in ("p", "arg", "ss", "mr", "rm", "rv"):
        assert torch.equal(a[k], b[k]), k
    for k in ("dg", "db"):
        np.testing.assert_allclose(b[k].cpu().numpy(), a[k].cpu().numpy(), rtol=2e-2, atol=2e-2 * float(a[k].abs().max()))
    d = (a["dx"].float() - b["dx"].float()).abs().max().item()
    assert d <= 2e-2 * a["dx"].float().abs().max().item(), d


def test_large_layer_weight_gradient_tiles_are_bit_identical(probe_lib):
    """The 256-row weight-gradient tiles of large dense layers (include/dreg_nerf_probe.h: dreg_conv_set_wgrad_big — 1: 4 waves on
    256 x 128 with 32-voxel stages; 3: 8 waves on 256 x 256, the default) against the 128 x 128 tile: the same per-element accumulation
    order, so the same bits."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 32, 32, 32, 256, generator=g).to(dev, torch.bfloat16)
    gy = torch.randn(2, 32, 32, 32, 256, generator=g).to(dev, torch.bfloat16)
    out = {}
    try:
        for mode in (0, 1, 3):
            lib.dreg_conv_set_wgrad_big(mode)
            out[mode] = ops.conv_wgrad(gy, x, (256, 256, 3, 3, 3), 256, 3, 1, 1)
    finally:
        lib.dreg_conv_set_wgrad_big(3)
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[3])
    # round 3: the 8-wave tile's loop forms (dreg_conv_set_wgrad_ring) — 3 anti-phase wave groups with the lean load half (default),
    # 8 the general anti-phase loop, 0 the lockstep loop over 64-voxel stages — accumulate the same 32-voxel products in the same order
    try:
        for ring in (0, 8, 3):
            lib.dreg_conv_set_wgrad_ring(ring)
            assert torch.equal(ops.conv_wgrad(gy, x, (256, 256, 3, 3, 3), 256, 3, 1, 1), out[0]), ring
        # a volume whose 32-voxel units straddle x-rows (16^3) takes the general anti-phase loop by itself; 1^3 taps; a batch of one
        for (B2, D2, k2) in ((16, 16, 3), (2, 32, 1), (1, 64, 3)):
            x2 = torch.randn(B2, D2, D2, D2, 256, generator=g).to(dev, torch.bfloat16)
            gy2 = torch.randn(B2, D2, D2, D2, 256, generator=g).to(dev, torch.bfloat16)
            got = {}
            for ring in (0, 3):
                lib.dreg_conv_set_wgrad_ring(ring)
                got[ring] = ops.conv_wgrad(gy2, x2, (256, 256, k2, k2, k2), 256, k2, 1, k2 // 2)
            assert torch.equal(got[0], got[3]), (B2, D2, k2)
            # against the 128 x 128 four-wave tile (itself checked against torch's fp32 convolution at small sizes in this file; a torch
            # reference HERE costs MIOpen ~25 s of kernel search per shape): the same per-element accumulation order, the same bits
            lib.dreg_conv_set_wgrad_big(0)
            ref2 = ops.conv_wgrad(gy2, x2, (256, 256, k2, k2, k2), 256, k2, 1, k2 // 2)
            lib.dreg_conv_set_wgrad_big(3)
            assert torch.equal(got[3], ref2) and float(ref2.abs().max()) > 0, (B2, D2, k2)
        # 64 -> 256 channels: K = 27 x 64 = 1,728 columns = 6.75 tiles of 256 — the anti-phase tile takes a ragged last column tile,
        # the lockstep setting falls back to the 128-wide tiles: same bits
        x3 = torch.randn(2, 32, 32, 32, 64, generator=g).to(dev, torch.bfloat16)
        gy3 = torch.randn(2, 32, 32, 32, 256, generator=g).to(dev, torch.bfloat16)
        got3 = {}
        for ring in (0, 3):
            lib.dreg_conv_set_wgrad_ring(ring)
            got3[ring] = ops.conv_wgrad(gy3, x3, (256, 64, 3, 3, 3), 64, 3, 1, 1)
        assert torch.equal(got3[0], got3[3]) and float(got3[3].abs().max()) > 0      # (the 128-wide tiles are pinned against torch at small sizes)
    finally:
        lib.dreg_conv_set_wgrad_ring(3)
    ref = torch.zeros(256, 256, 3, 3, 3, device=dev, requires_grad=True)
    F.conv3d(x.float().permute(0, 4, 1, 2, 3), ref, padding=1).backward(gy.float().permute(0, 4, 1, 2, 3))
    assert float((out[1] - ref.grad).abs().max()) <= 2e-3 * float(ref.grad.abs().max())


@pytest.mark.parametrize("cin,cout,D", [(256, 256, 48), (64, 256, 32), (128, 128, 16)])
def test_row_list_weight_gradient_fast_path_is_bit_identical(cin, cout, D, probe_lib):
    """Row-list weight gradient with (row, packed z|y|x) pairs in LDS read once per stage — and, for 256 -> 256 layers with >= 65,536
    rows, the 8-wave 256 x 256 tile — against the loop that decodes the voxel per load: same products in the same order."""
    from dreg_nerf_amd import lib as L
    lib = L.load()
    DEV = _dev()
    B = 8
    g = torch.Generator().manual_seed(cin + D)
    keep = torch.rand(B * D ** 3, generator=g) < (0.1 if D == 48 else 0.3)   # 48^3: 88 k rows, 2.8 k per split: the slice fits behind the 8-wave tile's stages
    keep[:3] = True; keep[-2:] = True                   # first / last voxels: every bounds test of the taps
    rows = keep.nonzero().flatten().int().to(DEV)
    n = rows.shape[0]
    x = torch.randn(B, D, D, D, cin, generator=g).to(DEV).bfloat16()
    gy = torch.randn(B, D, D, D, cout, generator=g).to(DEV).bfloat16()
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, D, D, D, cin, cout, 3, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    out = []

    def run():
        dw = torch.empty(cout, cin, 3, 3, 3, dtype=torch.float32, device=DEV)
        L.check(lib.dreg_conv3d_wgrad_rows(L.ptr(gy), L.ptr(x), L.ptr(dw), L.ptr(ws), nbytes, L.ptr(rows), n, B, D, D, D, cin, cin, D, D, D, cout, 3, 1, 1, 0,
                                           L.stream()), "dreg_conv3d_wgrad_rows")
        return dw

    try:
        lib.dreg_conv_set_row_splits(0)        # the same split count for both loops (the list-length rule depends on the tile)
        for fast in (0, 1):
            lib.dreg_conv_set_wgrad_rows_fast(fast)
            out.append(run())
    finally:
        lib.dreg_conv_set_wgrad_rows_fast(1)
        lib.dreg_conv_set_row_splits(1)
    if D == 48:
        assert lib.dreg_conv3d_wgrad_variant(B, D, D, D, cin, cout, 3, 1, n, 0) == 256256
    assert torch.isfinite(out[1]).all() and out[1].abs().max() > 0
    assert torch.equal(out[0], out[1])
    out[1] = run()                             # default: splits sized by the list's length (fewer, longer splits)
    assert torch.equal(out[1], run())
    # and against the dense kernel on a gradient that is zero off the rows
    gz = torch.zeros_like(gy).view(-1, cout)
    gz[rows.long()] = gy.view(-1, cout)[rows.long()]
    ref = ops.conv_wgrad(gz.view_as(gy), x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
    assert (out[1] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_row_list_helpers_of_the_active_set_backward():
    """dreg_zero_rows, dreg_downsample_sum_rows and dreg_maxpool3d_bwd_acc against plain torch."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(41)
    st = L.stream()
    # zero_rows: only the listed rows change
    buf = torch.randn(500, 64, generator=g).to(dev, torch.bfloat16)
    rows = torch.tensor(sorted(torch.randperm(500, generator=g)[:77].tolist()), dtype=torch.int32, device=dev)
    want = buf.clone(); want[rows.long()] = 0
    L.check(lib.dreg_zero_rows(L.ptr(buf), L.ptr(rows), rows.shape[0], 64, 0, st), "zero_rows")
    assert torch.equal(buf, want)
    # downsample_sum_rows: out[rows] = sum of the 2^3 children (odd fine extents: cropped), other rows untouched
    B, Df, Hf, Wf, C = 2, 7, 6, 5, 32
    Dc, Hc, Wc = 4, 3, 3
    fine = torch.randn(B, Df, Hf, Wf, C, generator=g).to(dev, torch.bfloat16)
    out = torch.full((B, Dc, Hc, Wc, C), 7.0, dtype=torch.bfloat16, device=dev)
    crow = torch.tensor(sorted(torch.randperm(B * Dc * Hc * Wc, generator=g)[:31].tolist()), dtype=torch.int32, device=dev)
    L.check(lib.dreg_downsample_sum_rows(L.ptr(fine), L.ptr(out), L.ptr(crow), crow.shape[0], Df, Hf, Wf, Dc, Hc, Wc, C, 0, st), "ds rows")
    pad = torch.zeros(B, 2 * Dc, 2 * Hc, 2 * Wc, C, device=dev)
    pad[:, :Df, :Hf, :Wf] = fine.float()
    ref = pad.view(B, Dc, 2, Hc, 2, Wc, 2, C).sum(dim=(2, 4, 6)).to(torch.bfloat16)
    flat_out, flat_ref = out.view(-1, C), ref.view(-1, C)
    sel = torch.zeros(flat_out.shape[0], dtype=torch.bool, device=dev); sel[crow.long()] = True
    assert torch.allclose(flat_out[sel].float(), flat_ref[sel].float(), atol=2e-2, rtol=2e-2) and bool((flat_out[~sel] == 7.0).all())
    # maxpool3d_bwd_acc: dx += unpool(dy)
    x = torch.randn(1, 6, 6, 6, 16, generator=g).to(dev, torch.bfloat16)
    y = torch.empty(1, 3, 3, 3, 16, dtype=torch.bfloat16, device=dev); arg = torch.empty(1, 3, 3, 3, 16, dtype=torch.uint8, device=dev)
    L.check(lib.dreg_maxpool3d_fwd(L.ptr(x), L.ptr(y), L.ptr(arg), 1, 6, 6, 6, 3, 3, 3, 16, 0, st), "pool fwd")
    dy = torch.randn(1, 3, 3, 3, 16, generator=g).to(dev, torch.bfloat16)
    plain = torch.empty_like(x)
    L.check(lib.dreg_maxpool3d_bwd(L.ptr(dy), L.ptr(arg), L.ptr(plain), 1, 6, 6, 6, 3, 3, 3, 16, 0, st), "pool bwd")
    base = torch.randn(1, 6, 6, 6, 16, generator=g).to(dev, torch.bfloat16)
    acc = base.clone()
    L.check(lib.dreg_maxpool3d_bwd_acc(L.ptr(dy), L.ptr(arg), L.ptr(acc), 1, 6, 6, 6, 3, 3, 3, 16, 1, 0, st), "pool bwd acc")
    xr = x.float().permute(0, 4, 1, 2, 3).requires_grad_(True)
    F.max_pool3d(xr, 3, 2, 1).backward(dy.float().permute(0, 4, 1, 2, 3))
    assert torch.allclose(plain.float(), xr.grad.permute(0, 2, 3, 4, 1), atol=1e-2)
    assert torch.allclose(acc.float(), base.float() + plain.float(), atol=2e-2, rtol=2e-2)


def test_conv_upsample_add_epilogue():
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 7, 6, 5, generator=g)
    w = torch.randn(256, 64, 1, 1, 1, generator=g) / 8
    b = torch.randn(256, generator=g)
    prev = torch.randn(2, 256, 4, 3, 3, generator=g)
    xr, pr = x.clone().requires_grad_(True), prev.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, w, b) + F.interpolate(pr, scale_factor=2)[:, :, :7, :6, :5]
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = ndhwc(x).to(dev).requires_grad_(True)
    pd = ndhwc(prev).to(dev).requires_grad_(True)
    y = ops.conv3d(xd, w.to(dev), b.to(dev), pd, 1, 0)
    y.backward(ndhwc(gy).to(dev))
    np.testing.assert_allclose(ncdhw(y.detach().cpu()).numpy(), y_ref.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(ncdhw(pd.grad.cpu()).numpy(), pr.grad.numpy(), atol=1e-4)
    np.testing.assert_allclose(ncdhw(xd.grad.cpu()).numpy(), xr.grad.numpy(), atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("shape", [(2, 6, 5, 7, 64), (1, 2, 2, 2, 2048), (3, 16, 16, 16, 256), (8, 8, 8, 8, 256), (8, 4, 4, 4, 512)])
def test_batchnorm_train_fwd_bwd(shape, dtype, with_res):
    """V <= 512 voxels per grid (the 8^3 / 4^3 levels) runs the one-launch kernels, larger volumes the three-kernel form; with_res = False
    exercises the ReLU mask recomputed from x in the backward pass."""
    dev = _dev()
    B, D, H, W, C = shape
    g = torch.Generator().manual_seed(C + D)
    x = (torch.randn(B, C, D, H, W, generator=g) * 2 + 0.5)
    res = torch.randn(B, C, D, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, res = x.bfloat16().float(), res.bfloat16().float()
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    rm0, rv0 = 0.1 * torch.randn(C, generator=g), 1 + 0.1 * torch.rand(C, generator=g)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    ys = []
    for b in range(B):  # reference semantics: one grid per BatchNorm call
        ys.append(F.relu(F.batch_norm(xr[b:b + 1], rm, rv, gr, br, True, 0.1, 1e-5) + (rr[b:b + 1] if with_res else 0.0)))
    y_ref = torch.cat(ys)
    gy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    y_ref.backward(gy)

    xd = ndhwc(x).to(dev, dtype).requires_grad_(True)
    rd = ndhwc(res).to(dev, dtype).requires_grad_(True)
    gd, bd = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmd, rvd = rm0.to(dev), rv0.to(dev)
    y = ops.batchnorm(xd, gd, bd, rmd, rvd, res=rd if with_res else None, relu=True, train=True)
    y.backward(ndhwc(gy).to(dev, dtype))
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(ncdhw(y.detach().float().cpu()).numpy(), y_ref.detach().numpy(), atol=tol * float(y_ref.abs().max()))
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), atol=1e-5)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ncdhw(xd.grad.float().cpu()).numpy(), xr.grad.numpy(), atol=tol * float(xr.grad.abs().max()) * 2)
    if with_res:
        np.testing.assert_allclose(ncdhw(rd.grad.float().cpu()).numpy(), rr.grad.numpy(), atol=tol * float(rr.grad.abs().max()))
    np.testing.assert_allclose(gd.grad.cpu().numpy(), gr.grad.numpy(), atol=tol * float(gr.grad.abs().max()) * 2)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), atol=tol * float(br.grad.abs().max()) * 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_backward_with_the_masked_gradient_stored_once_is_bit_identical(dtype, probe_lib):
    """Residual + ReLU BatchNorm, three-kernel form (dreg_bn_set_store_g): the statistics pass stores g = dy * (y > 0) as the residual
    branch's gradient and the apply pass reads it back, against both passes reading dy and y."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(17)
    B, D, C = 3, 20, 128
    x = torch.randn(B, D, D, D, C, generator=g).to(dev, dtype)
    res = torch.randn(B, D, D, D, C, generator=g).to(dev, dtype)
    gy = torch.randn(B, D, D, D, C, generator=g).to(dev, dtype)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    got = []
    try:
        for on in (1, 0):
            lib.dreg_bn_set_store_g(on)
            xd, rd = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
            gd, bd = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y = ops.batchnorm(xd, gd, bd, torch.zeros(C, device=dev), torch.ones(C, device=dev), res=rd, relu=True, train=True)
            y.backward(gy)
            got.append((xd.grad, rd.grad, gd.grad, bd.grad))
    finally:
        lib.dreg_bn_set_store_g(1)
    for a, c in zip(got[0], got[1]):
        assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0
        assert torch.equal(a, c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("D,C", [(8, 256), (4, 512)])
def test_small_batchnorm_with_rows_in_registers_is_bit_identical(D, C, with_res, dtype, probe_lib):
    """8^3 / 4^3 volumes (dreg_bn_set_small_regs): rows loaded once and kept in registers between the statistics and the apply phase
    against the form that walks them twice — forward output, running statistics and every gradient."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(D * C)
    B = 8
    x = (torch.randn(B, D, D, D, C, generator=g) * 2 + 0.5).to(dev, dtype)
    res = torch.randn(B, D, D, D, C, generator=g).to(dev, dtype)
    gy = torch.randn(B, D, D, D, C, generator=g).to(dev, dtype)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    got = []
    try:
        for on in (1, 0):
            lib.dreg_bn_set_small_regs(on)
            xd, rd = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
            gd, bd = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            y = ops.batchnorm(xd, gd, bd, rm, rv, res=rd if with_res else None, relu=True, train=True)
            y.backward(gy)
            got.append([y.detach(), rm, rv, xd.grad, gd.grad, bd.grad] + ([rd.grad] if with_res else []))
    finally:
        lib.dreg_bn_set_small_regs(1)
    for a, c in zip(got[0], got[1]):
        assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0
        assert torch.equal(a, c)


def test_batchnorm_eval():
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 4, 5, 6, generator=g)
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    rm, rv = torch.randn(64, generator=g), 1 + torch.rand(64, generator=g)
    y_ref = F.relu(F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5))
    y = ops.batchnorm(ndhwc(x).to(dev), gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), relu=True, train=False)
    np.testing.assert_allclose(ncdhw(y.cpu()).numpy(), y_ref.numpy(), atol=1e-5)


@pytest.mark.parametrize("dims", [(8, 8, 8), (7, 9, 6)])
def test_maxpool(dims):
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, *dims, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool3d(xr, 3, 2, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = ndhwc(x).to(dev).requires_grad_(True)
    y = ops.maxpool3d(xd)
    y.backward(ndhwc(gy).to(dev))
    assert torch.equal(ncdhw(y.detach().cpu()), y_ref.detach())
    np.testing.assert_allclose(ncdhw(xd.grad.cpu()).numpy(), xr.grad.numpy(), atol=1e-6)


def test_trilinear_gather():
    dev = _dev()
    from oracle import regtr_oracle as O
    g = torch.Generator().manual_seed(4)
    B, C = 2, 256
    p1 = torch.randn(B, C, 5, 6, 7, generator=g)
    res = (10, 12, 14)
    n = res[0] * res[1] * res[2]
    masks = [torch.randperm(n, generator=g)[:150].sort().values for _ in range(B)]
    p1r = p1.clone().requires_grad_(True)
    refs = [O.upsample_gather(p1r[b:b + 1], torch.zeros(1, 3, *res), masks[b])[1] for b in range(B)]
    ref = torch.cat(refs)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    pd = ndhwc(p1).to(dev).requires_grad_(True)
    idx = torch.cat(masks).to(dev)
    pb = torch.cat([torch.full((150,), b, dtype=torch.int32) for b in range(B)]).to(dev)
    out = ops.trilinear_gather(pd, idx, pb, res)
    out.backward(gy.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(ncdhw(pd.grad.cpu()).numpy(), p1r.grad.numpy(), atol=2e-5)


@pytest.mark.parametrize("res,coarse", [((10, 12, 14), (5, 6, 7)), ((17, 15, 13), (9, 8, 7))])
def test_trilinear_gather_backward_on_active_set_is_exact_and_deterministic(res, coarse):
    """Atomic-free gather form of the backward (one wave per S1 voxel) == the fp32 CPU gradient of the oracle's upsample+gather,
    run twice bit-identically (the dense form accumulates with fp32 atomics)."""
    dev = _dev()
    from oracle import regtr_oracle as O
    g = torch.Generator().manual_seed(res[0])
    B, C = 3, 256
    p1 = torch.randn(B, C, *coarse, generator=g)
    n = res[0] * res[1] * res[2]
    masks = [torch.randperm(n, generator=g)[:40 + 7 * b].sort().values for b in range(B)]
    masks[1] = torch.cat([masks[1], torch.tensor([0, n - 1])]).unique()   # the clamped corners
    p1r = p1.clone().requires_grad_(True)
    ref = torch.cat([O.upsample_gather(p1r[b:b + 1], torch.zeros(1, 3, *res), masks[b])[1] for b in range(B)])
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    idx = torch.cat(masks).to(dev)
    pb = torch.cat([torch.full((m.numel(),), b, dtype=torch.int32) for b, m in enumerate(masks)]).to(dev)
    rows = ops.active_sets([m.to(dev) for m in masks], res, coarse, dev, pt_batch=pb, idx_cat=idx, density_cap=1.0)
    grads = []
    for _ in range(2):
        pd = ndhwc(p1).to(dev).requires_grad_(True)
        out = ops.trilinear_gather(pd, idx, pb, res, rows[0], rows[3])
        out.backward(gy.to(dev))
        grads.append(pd.grad.clone())
    assert torch.equal(grads[0], grads[1])
    np.testing.assert_allclose(ncdhw(grads[0].cpu()).numpy(), p1r.grad.numpy(), atol=2e-5)


@pytest.mark.parametrize("mode", [3, 4])
def test_conv_glds_large_tiles_match_default(mode, probe_lib):
    """128x256 and 256x256 tile variants of the direct-to-LDS kernel == the 128x128 tile, bit for bit."""
    from dreg_nerf_amd import lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 33, 32, 32, 256, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(256, 256, 3, 3, 3, generator=g) * 0.02).to(dev)
    b = torch.randn(256, generator=g).to(dev)
    lib = L.load()
    try:
        lib.dreg_conv_set_glds(2)
        ref = ops.conv3d(x, w, b, None, 1, 1)
        lib.dreg_conv_set_glds(mode)
        got = ops.conv3d(x, w, b, None, 1, 1)
    finally:
        lib.dreg_conv_set_glds(1)
    assert torch.equal(ref, got)


def _active_sets_torch(idx_list, fine_res, coarse_dims):
    """CPU restatement of the row lists: corners of the trilinear gather (float32 arithmetic of tri_axis), two 3^3 dilations."""
    Zr, Xr, Yr = fine_res
    d, h, w = coarse_dims
    B = len(idx_list)
    flags = torch.zeros(B, d, h, w, dtype=torch.bool)

    def axis(i, n_out, n_in):
        s = (torch.tensor(float(n_in - 1), dtype=torch.float32) / torch.tensor(float(n_out - 1), dtype=torch.float32)) if n_out > 1 else torch.tensor(0.0)
        f = i.to(torch.float32) * s
        i0 = f.to(torch.int64).clamp_(max=n_in - 1)
        return i0, (i0 + 1).clamp_(max=n_in - 1)

    for b, f in enumerate(idx_list):
        z, y, x = f % Zr, (f // Zr) % Yr, f // (Zr * Yr)
        for zi in axis(z, Zr, d):
            for xi in axis(x, Xr, h):
                for yi in axis(y, Yr, w):
                    flags[b, zi, xi, yi] = True
    f1 = flags.float()[:, None]
    f2 = F.max_pool3d(f1, 3, 1, 1)
    f3 = F.max_pool3d(f2, 3, 1, 1)
    fa = F.max_pool3d(f2, 2, 2, ceil_mode=True)          # parents of S2 on the next pyramid level
    fa2 = F.max_pool3d(fa, 3, 1, 1)
    return [torch.nonzero(t.flatten() > 0)[:, 0].to(torch.int32) for t in (f1, f2, f3, fa, fa2)]


@pytest.mark.parametrize("res,frac", [((24, 20, 28), 0.01), ((33, 31, 29), 0.003), ((16, 16, 16), 0.0)])
def test_active_sets_match_cpu_construction(res, frac):
    dev = _dev()
    g = torch.Generator().manual_seed(res[0])
    Zr, Xr, Yr = res
    coarse = tuple((r + 1) // 2 for r in res)
    idx = []
    for b in range(3):
        n = int(frac * Zr * Xr * Yr) + (0 if frac == 0.0 else b)
        idx.append(torch.randperm(Zr * Xr * Yr, generator=g)[:n].sort().values)
    if frac == 0.0:
        idx[1] = torch.tensor([0, Zr * Xr * Yr - 1])  # the two extreme corners only; grids 0 and 2 empty
    ref = _active_sets_torch(idx, res, coarse)
    got = ops.active_sets([i.to(dev) for i in idx], res, coarse, dev, level2_cap=1.0)
    V = 3 * coarse[0] * coarse[1] * coarse[2]
    if ref[2].numel() > 0.2 * V:
        assert got is None
        return
    for a, b in zip(got[:3], ref[:3]):
        assert a.dtype == torch.int32 and torch.equal(a.cpu(), b)
    assert len(got) == 6
    for a, b in zip(got[4:], ref[3:]):
        assert a.dtype == torch.int32 and torch.equal(a.cpu(), b)
    map1 = got[3].cpu()
    want = torch.full((V,), -1, dtype=torch.int32)
    want[ref[0].long()] = torch.arange(ref[0].numel(), dtype=torch.int32)
    assert torch.equal(map1, want)


def test_batched_repack_equals_per_layer_pack():
    """ops.repack_all (one launch, LDS-staged rows) == dreg_pack_conv_weight per layer, bit for bit, for both pack kinds."""
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(64, 4, 5, 5, 5, generator=g), torch.randn(128, 64, 3, 3, 3, generator=g), torch.randn(256, 512, 1, 1, 1, generator=g),
          torch.randn(768, 256, generator=g), torch.randn(512, 512, 3, 3, 3, generator=g)]
    ops.clear_pack_cache()
    params_ = [torch.nn.Parameter(w.to(dev)) for w in ws]
    cin_pads = [8, 64, 512, 256, 512]
    first = []
    for p_, cp in zip(params_, cin_pads):
        first.append((ops.packed_weight(p_, cp, False, 0).clone(), ops.packed_weight(p_, cp, True, 0).clone() if p_.dim() == 5 and p_.shape[2] != 5 or p_.dim() == 2 else None))
    with torch.no_grad():
        for p_ in params_:
            p_.mul_(1.5)   # version bump: cached packs are stale now
    ops.repack_all(dev)
    assert len(ops._pack_cache) == 9
    for ent in ops._pack_cache.values():   # every cached pack was refreshed by the batched launch (no per-layer fallback below)
        assert ent[1] == (ent[0]()._version, ops._weight_generation)
    for p_, cp, (f0, d0) in zip(params_, cin_pads, first):
        f1 = ops.packed_weight(p_, cp, False, 0)
        ref = torch.empty_like(f1)
        from dreg_nerf_amd import lib as L
        lib = L.load()
        ksz = p_.shape[2] if p_.dim() == 5 else 1
        L.check(lib.dreg_pack_conv_weight(L.ptr(p_.detach()), L.ptr(ref), p_.shape[0], p_.shape[1], cp, ksz, 0, 0, L.stream()), "pack")
        assert torch.equal(f1.view(torch.int16), ref.view(torch.int16)) and not torch.equal(f1.view(torch.int16), f0.view(torch.int16))
        if d0 is not None:
            d1 = ops.packed_weight(p_, cp, True, 0)
            refd = torch.empty_like(d1)
            L.check(lib.dreg_pack_conv_weight(L.ptr(p_.detach()), L.ptr(refd), p_.shape[0], p_.shape[1], cp, ksz, 1, 0, L.stream()), "pack")
            assert torch.equal(d1.view(torch.int16), refd.view(torch.int16))
    ops.clear_pack_cache()


@pytest.mark.parametrize("n", [70001, 131072])
def test_linear_with_many_rows(n):
    """Linear layers run as 1x1x1 convolutions with the ROWS as the batch dimension: six stacked decoder inputs of a step with
    > 10,922 points each (6R >= 65,536 rows) used to fail the geometry check of the kernels' magic row decode (n^2 >= 2^32)."""
    DEV = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 256, generator=g).to(DEV).bfloat16().requires_grad_(True)
    w = (torch.randn(128, 256, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = torch.randn(128, generator=g).to(DEV).requires_grad_(True)
    y = ops.linear(x, w, b, relu=True)
    ref = F.relu(x.detach().float() @ w.detach().t() + b.detach())
    assert y.shape == (n, 128)
    tol = 2e-2 * float(ref.abs().max())
    assert float((y.float() - ref).abs().max()) <= tol
    gy = torch.randn(n, 128, generator=g).to(DEV).bfloat16()
    ops.linear(x, w, b).backward(gy)      # no ReLU here: its mask flips where bf16 and fp32 pre-activations straddle zero
    xr = x.detach().float().requires_grad_(True); wr = w.detach().clone().requires_grad_(True)
    (xr @ wr.t() + b.detach()).backward(gy.float())
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2e-2 * float(xr.grad.abs().max())
    assert float((w.grad - wr.grad).abs().max()) <= 2e-2 * float(wr.grad.abs().max())
    # the last row is computed (ragged tile) and distinct from its neighbours
    assert torch.isfinite(y[-1]).all() and not torch.equal(y[-1], y[-2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_stem_row_occupancy_skips_only_empty_rows(dtype):
    """Stem convolution (5^3, stride 2, pad 2, 8 input channels) on a volume that is zero outside a small blob: with the output-row
    occupancy flags the forward result and the weight gradient are the same BIT FOR BIT as without them; the flags come from the packer
    (values, not masks) and mark exactly the rows whose window touches a non-zero input row."""
    from dreg_nerf_amd import lib as L
    from dreg_nerf_amd.regtr import NeRFRegTr
    dev = _dev()
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    B, R = 2, 64
    grids = []
    for b in range(B):
        v = torch.zeros(1, 7, R, R, R)
        z0, x0 = 10 + 20 * b, 30
        v[:, 3:, z0:z0 + 6, x0:x0 + 9, 5:50] = torch.rand(1, 4, 6, 9, 45, generator=g)
        grids.append(v.to(dev))
    x, occ = NeRFRegTr.pack_grids(grids, dtype, occupancy=True)
    x0_ = NeRFRegTr.pack_grids(grids, dtype)
    assert torch.equal(x, x0_) and occ.shape == (B, R // 2, R // 2) and occ.dtype == torch.uint8
    # reference flags on the host: output row (zo, xo) is live iff some input row (z, x) in [2zo-2, 2zo+2] x [2xo-2, 2xo+2] is non-zero
    nz = (torch.stack([gg[0, 3:].abs().sum(dim=(0, 3)) for gg in grids]) > 0).cpu()       # [B, Z, X]
    want = torch.zeros(B, R // 2, R // 2, dtype=torch.bool)
    for b, z, xx in nz.nonzero().tolist():
        want[b, max(0, (z - 2 + 1) // 2):min(R // 2, (z + 2) // 2 + 1), max(0, (xx - 2 + 1) // 2):min(R // 2, (xx + 2) // 2 + 1)] = True
    assert torch.equal(occ.cpu().bool(), want) and 0 < int(want.sum()) < want.numel() // 4
    w = (torch.randn(64, 4, 5, 5, 5, generator=g) * 0.1).to(dev)
    dt = L.dt_of(x)
    wpk = ops.packed_weight(w, 8, False, dt)
    Do = R // 2
    outs = []
    for flags in (None, occ):
        y = torch.full((B, Do, Do, Do, 64), 7.0, dtype=dtype, device=dev)       # poisoned: skipped tiles must be written
        L.check(lib.dreg_conv3d_igemm_occ(L.ptr(x), L.ptr(wpk), L.ptr(y), None, None, B, R, R, R, 8, Do, Do, Do, 64, 5, 2, 2, 0, 0, 0, 0, 0, 0,
                                          dt, 0, None, 0, L.ptr(flags), L.stream()), "dreg_conv3d_igemm_occ")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].float().abs().max()) > 0 and float((outs[0].float().abs().sum(dim=(3, 4)) > 0).float().mean()) < 0.3
    gy = torch.randn(B, Do, Do, Do, 64, generator=g).to(dev).to(dtype)
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, Do, Do, Do, 8, 64, 5, dt)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dws = []
    for flags in (None, occ):
        dw = torch.empty(64, 4, 5, 5, 5, dtype=torch.float32, device=dev)
        L.check(lib.dreg_conv3d_wgrad_occ(L.ptr(gy), L.ptr(x), L.ptr(dw), L.ptr(ws), nbytes, B, R, R, R, 8, 4, Do, Do, Do, 64, 5, 2, 2, 0, dt,
                                          int(dtype == torch.bfloat16), L.ptr(flags), L.stream()), "dreg_conv3d_wgrad_occ")
        dws.append(dw)
    assert torch.equal(dws[0], dws[1]) and float(dws[0].abs().max()) > 0

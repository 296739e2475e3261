"""GPU: the halo-tile 3^3 convolution (csrc/conv_halo.hip) against a plain PyTorch fp32 conv3d of the same bf16-rounded operands
(forward with bias / nearest-x2 addend, and the data gradient through the flipped-tap pack), plus agreement with the
implicit-GEMM kernel it replaces on the dominant layers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import lib as L, ops  # noqa: E402

DEV = "cuda:0"


def _halo(x, w, bias=None, addend=None, add_same=False, transposed=False, out_f32=False, variant=0):
    """x [B,D,H,W,Cin] bf16, w fp32 [Cout,Cin,3,3,3] -> [B,D,H,W,256]; variant != 0: that loop form of the measurement build (include/dreg_nerf_probe.h)"""
    if variant:
        with L.probe() as pr:
            pr.set("dreg_conv3_halo_set_variant", variant, 0)
            return _halo(x, w, bias, addend, add_same, transposed, out_f32, 0)
    lib = L.load()
    B, D, H, W, Cin = x.shape
    cout, cin = w.shape[0], w.shape[1]
    red = cout if transposed else cin
    assert lib.dreg_conv3_halo_supported(B, D, H, W, red, 256) == 1
    pk = torch.empty(lib.dreg_conv3_halo_pack_bytes(red) // 2, dtype=torch.bfloat16, device=x.device)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w.contiguous()), L.ptr(pk), cout, cin, int(transposed), L.stream()), "pack_halo")
    out = torch.empty(B, D, H, W, 256, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    da, ha, wa = (addend.shape[1:4] if addend is not None else (0, 0, 0))
    L.check(lib.dreg_conv3_halo(L.ptr(x), L.ptr(pk), L.ptr(out), L.ptr(bias), L.ptr(addend), B, D, H, W, red, da, ha, wa, int(add_same),
                                int(out_f32), L.stream()), "dreg_conv3_halo")
    return out


def _ref(x, w, bias=None, addend=None, add_same=False):
    xr = x.float().permute(0, 4, 1, 2, 3)
    y = F.conv3d(xr.double(), w.to(torch.bfloat16).double(), None if bias is None else bias.double(), padding=1)
    if addend is not None:
        a = addend.double().permute(0, 4, 1, 2, 3)
        y = y + (a if add_same else F.interpolate(a, scale_factor=2, mode="nearest"))
    return y.permute(0, 2, 3, 4, 1).float()


@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("cin,shape", [(256, (2, 8, 16, 16)), (64, (1, 4, 8, 24)), (32, (3, 12, 8, 8))])
def test_halo_forward_matches_fp32_reference(cin, shape, variant):
    g = torch.Generator().manual_seed(7)
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(256, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    bias = torch.randn(256, generator=g).to(DEV)
    add = torch.randn(B, D // 2, H // 2, W // 2, 256, generator=g).to(DEV)
    ref = _ref(x, w, bias, add)
    out = _halo(x, w, bias, add, out_f32=True, variant=variant)
    err = (out - ref).abs().max().item()
    assert err <= 2e-5 * (1 + ref.abs().max().item()), err           # fp32 accumulation of exact bf16 products
    ob = _halo(x, w, bias, add.to(torch.bfloat16), variant=variant)
    refb = _ref(x, w, bias, add.to(torch.bfloat16))
    assert (ob.float() - refb).abs().max().item() <= 2 ** -8 * (refb.abs().max().item()) + 1e-6   # one bf16 rounding of the result
    # a transposed row / column would pass a symmetric test: weights and inputs above are asymmetric random tensors


def test_halo_data_gradient_matches_autograd():
    g = torch.Generator().manual_seed(8)
    B, D, H, W = 1, 8, 8, 16
    for cin in (256,):
        gy = torch.randn(B, D, H, W, 256, generator=g).to(DEV).to(torch.bfloat16)       # dOut of a 256 -> 256 layer
        w = (torch.randn(256, cin, 3, 3, 3, generator=g) * 0.02).to(DEV)
        xr = torch.zeros(B, cin, D, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
        y = F.conv3d(xr, w.to(torch.bfloat16).double(), padding=1)
        y.backward(gy.double().permute(0, 4, 1, 2, 3))
        ref = xr.grad.permute(0, 2, 3, 4, 1).float()
        got = _halo(gy, w, transposed=True, out_f32=True)
        assert (got - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())


def test_halo_agrees_with_implicit_gemm_kernel_on_the_dominant_layer():
    """upsample_transform_1 (256 -> 256) on a 32^3 x 2 stand-in: same operands through both kernels; the accumulation order
    differs (chunk-major vs tap-major), so agreement is to fp32 round-off before the single bf16 rounding of the result."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 32, 32, 256, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(256, 256, 3, 3, 3, generator=g) * 0.017).to(DEV)
    bias = torch.randn(256, generator=g).to(DEV)
    a = ops.conv3d(x, w, bias, pad=1).float()
    b = _halo(x, w, bias).float()
    diff = (a - b).abs()
    # both round the same fp32 sum (up to ~1e-6 relative) to bf16: they differ by at most one bf16 ulp, and only rarely
    assert diff.max().item() <= 2 ** -7 * a.abs().max().item()
    assert (diff > 0).float().mean().item() < 0.02


# ---- the 64-output-channel kernel (conv2 of layer1's bottlenecks at 32^3 and its data gradient): 8 x 8 x 8 boxes, two workgroups per CU

def _halo64(x, w, bias=None, addend=None, add_same=False, transposed=False, mode=1):
    """x [B,D,H,W,C] bf16, w fp32 [Cout,Cin,3,3,3] -> [B,D,H,W,64] bf16 (transposed: the data gradient, 64 = Cin); mode 2: the other loop form of the measurement build"""
    if mode != 1:
        with L.probe() as pr:
            pr.set("dreg_conv3_halo64_set", mode, 1)
            return _halo64(x, w, bias, addend, add_same, transposed, 1)
    lib = L.load()
    B, D, H, W, C = x.shape
    cout, cin = w.shape[0], w.shape[1]
    red, rows = (cout, cin) if transposed else (cin, cout)
    assert rows == 64 and red == C and lib.dreg_conv3_halo_supported(B, D, H, W, red, 64) == 1
    pk = torch.empty(lib.dreg_conv3_halo_pack_bytes_n(red, 64) // 2, dtype=torch.bfloat16, device=x.device)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w.contiguous()), L.ptr(pk), cout, cin, int(transposed), L.stream()), "pack_halo")
    out = addend if (addend is not None and add_same == "inplace") else torch.empty(B, D, H, W, 64, dtype=torch.bfloat16, device=x.device)
    da, ha, wa = (addend.shape[1:4] if addend is not None else (0, 0, 0))
    L.check(lib.dreg_conv3_halo_n(L.ptr(x), L.ptr(pk), L.ptr(out), L.ptr(bias), L.ptr(addend), B, D, H, W, red, 64, da, ha, wa, int(bool(add_same)), 0,
                                  L.stream()), "dreg_conv3_halo_n")
    return out


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cin,shape", [(64, (2, 8, 16, 16)), (32, (1, 16, 8, 8)), (256, (1, 8, 8, 24)), (64, (3, 24, 8, 16))])
def test_halo64_forward_matches_fp32_reference(cin, shape, mode):
    g = torch.Generator().manual_seed(17)
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(64, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    bias = torch.randn(64, generator=g).to(DEV)
    tol = lambda ref: 2 ** -8 * ref.abs().max().item() + 1e-6          # one bf16 rounding of the fp32 sum  # noqa: E731
    ref = _ref(x, w)
    assert (_halo64(x, w, mode=mode).float() - ref).abs().max().item() <= tol(ref)
    add2 = torch.randn(B, D // 2, H // 2, W // 2, 64, generator=g).to(DEV).to(torch.bfloat16)
    ref = _ref(x, w, bias, add2)
    assert (_halo64(x, w, bias, add2, mode=mode).float() - ref).abs().max().item() <= tol(ref)
    add1 = torch.randn(B, D, H, W, 64, generator=g).to(DEV).to(torch.bfloat16)
    ref = _ref(x, w, None, add1, add_same=True)
    assert (_halo64(x, w, None, add1, add_same=True, mode=mode).float() - ref).abs().max().item() <= tol(ref)
    # the accumulating form of the data gradients: the addend IS the output tensor
    acc = add1.clone()
    got = _halo64(x, w, None, acc, add_same="inplace", mode=mode)
    assert got.data_ptr() == acc.data_ptr() and (got.float() - ref).abs().max().item() <= tol(ref)


def test_halo64_data_gradient_matches_autograd():
    g = torch.Generator().manual_seed(18)
    B, D, H, W = 2, 8, 16, 8
    for cout in (64, 128):                                                  # dOut of a 64 -> cout layer: the gradient has 64 channels
        gy = torch.randn(B, D, H, W, cout, generator=g).to(DEV).to(torch.bfloat16)
        w = (torch.randn(cout, 64, 3, 3, 3, generator=g) * 0.03).to(DEV)
        xr = torch.zeros(B, 64, D, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
        y = F.conv3d(xr, w.to(torch.bfloat16).double(), padding=1)
        y.backward(gy.double().permute(0, 4, 1, 2, 3))
        ref = xr.grad.permute(0, 2, 3, 4, 1).float()
        got = _halo64(gy, w, transposed=True).float()
        assert (got - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6


def test_halo64_agrees_with_implicit_gemm_and_is_the_kernel_the_layer_runs_on():
    """conv2 of a layer1 bottleneck (64 -> 64) on 32^3 x 2: same operands through both kernels (the accumulation order differs: chunk-major vs tap-major),
    through ops.conv3d forward and backward — which routes the shape to the halo kernel — and with the kernel switched off in the measurement build."""
    g = torch.Generator().manual_seed(19)
    x = torch.randn(2, 32, 32, 32, 64, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.034).to(DEV).requires_grad_(True)
    gy = torch.randn(2, 32, 32, 32, 64, generator=g).to(DEV).to(torch.bfloat16)
    assert ops.halo_applies(x.shape, 64, 64, 3, 1, 1, L.DT_BF16) and not ops.halo_applies((2, 16, 16, 16, 64), 64, 64, 3, 1, 1, L.DT_BF16)
    y = ops.conv3d(x, w, None, pad=1)
    y.backward(gy)
    direct = _halo64(x.detach(), w.detach())
    assert torch.equal(y, direct)
    assert torch.equal(x.grad, _halo64(gy, w.detach(), transposed=True))
    gx_h = x.grad.clone()
    x.grad = w.grad = None
    with L.probe() as pr:
        pr.set("dreg_conv3_halo64_set", 0, 1)
        assert not ops.halo_applies(x.shape, 64, 64, 3, 1, 1, L.DT_BF16)
        y2 = ops.conv3d(x, w, None, pad=1)
        y2.backward(gy)
    for a, b in ((y, y2), (gx_h, x.grad)):
        diff = (a.float() - b.float()).abs()
        assert diff.max().item() <= 2 ** -7 * b.float().abs().max().item() and (diff > 0).float().mean().item() < 0.02


def test_halo64_statistics_epilogue_sums_the_stored_values_per_chunk_and_does_not_depend_on_batch_mates():
    """dreg_conv3_halo_n_bnstats: the output is the plain launch's bit for bit; the chunk sums [B][V/128][64][2] are those of the STORED values over one
    wave's 128 voxels (two z-planes of an 8^3 box, boxes in (z, y, x) order), and a grid's sums are the same bits whether it is launched alone or with others."""
    import ctypes
    lib = L.load()
    g = torch.Generator().manual_seed(23)
    B, D, H, W, cin = 3, 16, 8, 24, 64
    x = torch.randn(B, D, H, W, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(64, cin, 3, 3, 3, generator=g) * 0.034).to(DEV)
    pk = torch.empty(lib.dreg_conv3_halo_pack_bytes_n(cin, 64) // 2, dtype=torch.bfloat16, device=DEV)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w), L.ptr(pk), 64, cin, 0, L.stream()), "pack_halo")

    def run(xs):
        nb = xs.shape[0]
        out = torch.empty(nb, D, H, W, 64, dtype=torch.bfloat16, device=DEV)
        sums = torch.full((nb, D * H * W // 128, 64, 2), float("nan"), device=DEV)
        rpc = ctypes.c_int(-1)
        L.check(lib.dreg_conv3_halo_n_bnstats(L.ptr(xs), L.ptr(pk), L.ptr(out), None, None, nb, D, H, W, cin, 64, 0, 0, 0, 0, L.ptr(sums), ctypes.addressof(rpc),
                                              L.stream()), "dreg_conv3_halo_n_bnstats")
        torch.cuda.synchronize()
        assert rpc.value == 128
        return out, sums

    out, sums = run(x)
    assert torch.equal(out, _halo64(x, w))
    o = out.double().view(B, D // 8, 4, 2, H // 8, 8, W // 8, 8, 64)                 # [b, tz, wave, z in wave, ty, y, tx, x, c]
    want1 = o.sum(dim=(3, 5, 7)).permute(0, 1, 3, 4, 2, 5).reshape(B, -1, 64)         # [b, (tz, ty, tx, wave), c]
    want2 = (o * o).sum(dim=(3, 5, 7)).permute(0, 1, 3, 4, 2, 5).reshape(B, -1, 64)
    assert torch.allclose(sums[..., 0].double(), want1, rtol=1e-5, atol=1e-4) and torch.allclose(sums[..., 1].double(), want2, rtol=1e-5, atol=1e-4)
    o1, s1 = run(x[1:2].contiguous())
    assert torch.equal(o1[0], out[1]) and torch.equal(s1[0], sums[1])

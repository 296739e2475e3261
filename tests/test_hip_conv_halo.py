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

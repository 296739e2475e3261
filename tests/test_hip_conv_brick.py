"""GPU: the active-set 3^3 convolution with staged-neighbourhood reuse (csrc/conv_brick.hip) — tile tables against a host model of
what they must contain, and the convolution (forward and data-gradient packs, 256 and 64 output channels, upsampled addend) against an
fp32 torch convolution of the same bf16 operands and against the row-list form of the implicit-GEMM kernel."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import brick, lib as L, ops  # noqa: E402


def _shell_flags(B, D, thick=1.5, seed=0, noise=0.0):
    g = torch.Generator().manual_seed(seed)
    c = torch.arange(D, dtype=torch.float32) - (D - 1) / 2
    z, y, x = torch.meshgrid(c, c, c, indexing="ij")
    r = (z * z + y * y + x * x).sqrt()
    fl = []
    for b in range(B):
        rad = D * (0.25 + 0.05 * b)
        f = (r - rad).abs() < thick
        if noise:
            f |= torch.rand(D, D, D, generator=g) < noise
        fl.append(f)
    return torch.stack(fl).to(torch.uint8).cuda()


def test_tile_tables_cover_every_row_once_and_point_at_the_right_neighbours():
    B, D = 3, 32
    flags = _shell_flags(B, D, noise=0.01)
    n = int(flags.sum())
    bt = brick.build(flags, n)
    assert bt.nrows == n and not bt.overflow and bt.ntiles >= (n + 255) // 256
    rows = bt.rows_sorted[:n].cpu().numpy()
    assert np.array_equal(np.sort(rows), np.nonzero(flags.flatten().cpu().numpy())[0]), "rows_sorted is a permutation of the flagged voxels"
    tiles = bt.tiles[:bt.ntiles].cpu().numpy()
    halo = bt.halo[:bt.ntiles].cpu().numpy()
    nbr = bt.nbr[:bt.ntiles].cpu().numpy().view(np.uint16)
    V = D ** 3
    covered = np.zeros(n, dtype=np.int32)
    conflicts = total_groups = 0
    for t in range(bt.ntiles):
        r0, nr, nh, nuni = tiles[t]
        assert 0 < nr <= 256 and 0 < nh <= 1264
        covered[r0:r0 + nr] += 1
        vox = rows[r0:r0 + nr]
        assert len(set((vox // V).tolist())) == 1, "a tile stays inside one grid"
        b = vox[0] // V
        v = vox - b * V
        zc, yc, xc = v // (D * D), (v // D) % D, v % D
        hv = halo[t]
        used = hv[hv >= 0]
        assert len(used) == nuni == len(set(used.tolist())) and np.all(hv[nh:] == -1), "every staged voxel once; nothing past the last used stripe"
        assert np.all(hv[:nuni] >= 0) and np.all(np.diff(hv[:nuni]) > 0), "slot = raster rank of the staged voxel (the default numbering)"
        for tp in range(27):
            dz, dy, dx = tp // 9 - 1, (tp // 3) % 3 - 1, tp % 3 - 1
            zz, yy, xx = zc + dz, yc + dy, xc + dx
            inside = (zz >= 0) & (zz < D) & (yy >= 0) & (yy < D) & (xx >= 0) & (xx < D)
            off = nbr[t, :nr, tp].astype(np.int64)
            slot = off >> 5
            assert np.all((off & 31) == (((slot >> 3) & 1) << 4))
            assert np.all(slot[~inside] >= 1264) and np.all(slot[~inside] % 16 == ((xx[~inside] & 7) | ((yy[~inside] & 1) << 3)))
            want = b * V + (zz[inside] * D + yy[inside]) * D + xx[inside]
            assert np.array_equal(hv[slot[inside]], want), f"tile {t} tap {tp}"
            # LDS bank groups of the fragment reads: 16 consecutive rows should touch 16 distinct slots modulo 16
            for g0 in range(0, nr - 15, 16):
                total_groups += 1
                conflicts += 16 - len(set((slot[g0:g0 + 16] % 16).tolist()))
        assert np.all(nbr[t, nr:] >> 5 >= 1264)
    print(f"lanes of a 16-lane fragment-read group that share a bank group with an earlier lane: {conflicts / max(total_groups, 1):.2f} of 16 on average")
    assert np.all(covered == 1)


def _reference(x, w, rows_flat, bias, addend_up):
    """fp32 convolution of the bf16-rounded operands at the flagged rows."""
    xr = x.float().permute(0, 4, 1, 2, 3)
    y = F.conv3d(xr, w.to(torch.bfloat16).float(), bias, padding=1).permute(0, 2, 3, 4, 1).reshape(-1, w.shape[0])
    if addend_up is not None:
        y = y + addend_up.float().reshape(-1, w.shape[0])
    return y[rows_flat]


@pytest.mark.parametrize("cin,cout,transposed,with_add", [(64, 256, False, True), (256, 256, False, False), (256, 256, True, False), (64, 256, True, False)])
def test_brick_convolution_vs_fp32_and_row_list_kernel(cin, cout, transposed, with_add):
    """transposed: the layer is cin -> cout and the launch computes its DATA GRADIENT (input = dOut with cout channels, output cin channels)."""
    torch.manual_seed(1)
    B, D = 2, 32
    flags = _shell_flags(B, D, thick=2.5, seed=3)
    n = int(flags.sum())
    bt = brick.build(flags, n)
    assert not bt.overflow
    rows_flat = torch.nonzero(flags.flatten())[:, 0]
    w = (torch.randn(cout, cin, 3, 3, 3) * (1.0 / (27 * cin) ** 0.5)).cuda()
    c_in, c_out = (cout, cin) if transposed else (cin, cout)
    x = torch.randn(B, D, D, D, c_in, device="cuda").to(torch.bfloat16)
    bias = torch.randn(c_out, device="cuda") if not transposed else None
    add = torch.randn(B, D // 2, D // 2, D // 2, c_out, device="cuda").to(torch.bfloat16) if with_add else None
    out = torch.full((B, D, D, D, c_out), 7.0, device="cuda", dtype=torch.bfloat16)
    brick.conv(x, brick.pack_weight(w, transposed), out, bias, add, bt, c_in, c_out)
    torch.cuda.synchronize()
    # rows outside the set are untouched
    untouched = torch.ones(B * D ** 3, dtype=torch.bool, device="cuda")
    untouched[rows_flat] = False
    assert torch.all(out.reshape(-1, c_out)[untouched] == 7.0)
    # fp32 reference: forward conv with w, or its data gradient = conv with the flipped, transposed weight
    w_eff = w if not transposed else w.flip(2, 3, 4).transpose(0, 1).contiguous()
    add_up = None
    if add is not None:
        add_up = add.float().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = _reference(x, w_eff, rows_flat, bias, add_up)
    got = out.reshape(-1, c_out)[rows_flat].float()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1.2e-2 * scale          # bf16 output rounding
    # the row-list form of the implicit-GEMM kernel on the same operands: equal up to the order of the fp32 sums
    lib = L.load()
    out2 = torch.zeros_like(out)
    wpk = ops.packed_weight(w, cin, transposed, L.DT_BF16)
    rows32 = rows_flat.to(torch.int32)
    L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wpk), L.ptr(out2), L.ptr(bias), L.ptr(add), L.ptr(rows32), n, B, D, D, D, c_in, D, D, D, c_out,
                                       3, 1, 1, int(transposed), 0, D // 2 if with_add else 0, D // 2 if with_add else 0, D // 2 if with_add else 0, 0, 0,
                                       L.stream()), "dreg_conv3d_igemm_rows")
    got2 = out2.reshape(-1, c_out)[rows_flat].float()
    assert float((got - got2).abs().max()) <= 1.0e-2 * scale
    assert float((got - got2).abs().mean()) <= 1e-3 * scale


def test_overflowing_candidates_are_split():
    """Scattered single voxels: a 256-row tile would need ~27 staged voxels per row; the builder halves it until the union fits."""
    B, D = 1, 48
    g = torch.Generator().manual_seed(5)
    flags = (torch.rand(B, D, D, D, generator=g) < 0.004).to(torch.uint8).cuda()
    n = int(flags.sum())
    bt = brick.build(flags, n)
    assert not bt.overflow and bt.ntiles > (n + 255) // 256
    tiles = bt.tiles[:bt.ntiles].cpu().numpy()
    assert tiles[:, 1].sum() == n and tiles[:, 2].max() <= 1264
    x = torch.randn(B, D, D, D, 64, device="cuda").to(torch.bfloat16)
    w = (torch.randn(256, 64, 3, 3, 3) * 0.02).cuda()
    out = torch.zeros(B, D, D, D, 256, device="cuda", dtype=torch.bfloat16)
    brick.conv(x, brick.pack_weight(w, False), out, None, None, bt, 64, 256)
    rows_flat = torch.nonzero(flags.flatten())[:, 0]
    ref = _reference(x, w, rows_flat, None, None)
    got = out.reshape(-1, 256)[rows_flat].float()
    assert float((got - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())


def test_grid_sizes_that_are_no_multiple_of_the_brick_and_an_empty_grid():
    """20^3 grids (partial bricks at every face), one of three grids without a single flagged voxel, a voxel in every corner."""
    B, D = 3, 20
    flags = _shell_flags(B, D, thick=1.2, seed=9)
    flags[1] = 0
    for c in (0, D - 1):
        flags[0, c, c, c] = 1
        flags[2, c, D - 1 - c, c] = 1
    n = int(flags.sum())
    bt = brick.build(flags, n)
    assert bt.nrows == n and not bt.overflow
    rows = bt.rows_sorted[:n].cpu().numpy()
    assert np.array_equal(np.sort(rows), np.nonzero(flags.flatten().cpu().numpy())[0])
    x = torch.randn(B, D, D, D, 256, device="cuda").to(torch.bfloat16)
    w = (torch.randn(256, 64, 3, 3, 3) * 0.02).cuda()       # layer 64 -> 256: its data gradient has 64 output channels
    out = torch.zeros(B, D, D, D, 64, device="cuda", dtype=torch.bfloat16)
    brick.conv(x, brick.pack_weight(w, True), out, None, None, bt, 256, 64)
    rows_flat = torch.nonzero(flags.flatten())[:, 0]
    ref = _reference(x, w.flip(2, 3, 4).transpose(0, 1).contiguous(), rows_flat, None, None)
    got = out.reshape(-1, 64)[rows_flat].float()
    assert float((got - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())
    untouched = torch.ones(B * D ** 3, dtype=torch.bool, device="cuda")
    untouched[rows_flat] = False
    assert torch.all(out.reshape(-1, 64)[untouched] == 0)

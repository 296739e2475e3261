"""GPU: NGP dense-query kernels (ngp.hip) against the oracle restatement (oracle/ngp_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import ngp  # noqa: E402
from oracle import ngp_oracle as N  # noqa: E402

DEV = "cuda:0"
AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]


def _field(seed, table_scale=0.5):
    f = ngp.NGPradianceField(AABB)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        p = f.mlp_base.params
        p[:3072] = torch.randn(3072, generator=g) * 0.25
        p[3072:] = torch.randn(p.numel() - 3072, generator=g) * table_scale
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    return f


def test_level_table_matches_oracle():
    f = ngp.NGPradianceField(AABB)
    rows, total = N.level_table()
    off, size, res, scale, hashed = f._levels
    assert [int(v) for v in off] == [r["offset"] for r in rows]
    assert [int(v) for v in size] == [r["size"] for r in rows]
    assert [int(v) for v in res] == [r["res"] for r in rows]
    assert [bool(v) for v in hashed] == [r["hashed"] for r in rows]
    np.testing.assert_allclose([float(v) for v in scale], [r["scale"] for r in rows], rtol=1e-6)
    assert f.mlp_base.params.numel() == 12602992


def test_all_ones_table_known_answer():
    f = _field(0)
    with torch.no_grad():
        f.mlp_base.params[3072:] = 1.0
    f = f.to(DEV)
    x = (torch.rand(1000, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 2.9
    _, raw = f.query_raw(x.to(DEV))
    w1, w2, _ = N.split_density_params(f.mlp_base.params.detach().cpu())
    expect = N.f16(N.f16(torch.relu(torch.ones(1, 32) @ N.f16(w1).T)) @ N.f16(w2).T)
    assert (raw.float().cpu() - expect).abs().max() <= 2e-3 * expect.abs().max()


def test_density_and_rgb_vs_oracle():
    f = _field(2)
    params_b, params_c = f.mlp_base.params.detach().clone(), f.color_mlp.params.detach().clone()
    f = f.to(DEV)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(6000, 3, generator=g) - 0.5) * 3.2  # some points outside the aabb
    d, raw = f.query_raw(x.to(DEV))
    d_ref, raw_ref = N.query_density(x, torch.tensor(AABB), params_b)
    inside = ((x > -1.5) & (x < 1.5)).all(-1)
    scale = float(raw_ref.abs().max())
    err = (raw.float().cpu() - raw_ref)[inside].abs().max()
    assert err <= 4e-3 * scale, (float(err), scale)
    assert torch.equal(d.cpu()[~inside], torch.zeros(int((~inside).sum())))
    np.testing.assert_allclose(d.cpu()[inside].numpy(), d_ref[inside].numpy(), rtol=2e-2)
    # density mask: bit-exact away from the threshold
    far = (d_ref - 0.7).abs() > 0.02 * d_ref.clamp(min=0.7)
    assert torch.equal((d.cpu() > 0.7)[far], (d_ref > 0.7)[far])
    # colour: mean over the 18 fixed directions, on the oracle's raw so that only the colour net is compared
    dirs = N.fixed_viewdirs()
    rgb = f.query_rgb_mean(raw_ref.half().to(DEV), dirs.to(DEV)).cpu()
    rgb_ref = torch.stack([N.query_rgb(dirs[k].expand(x.shape[0], 3), raw_ref, params_c) for k in range(18)]).mean(0)
    np.testing.assert_allclose(rgb.numpy(), rgb_ref.numpy(), atol=3e-3)


def test_dense_query_and_grid_writer(tmp_path):
    res = 32
    f = _field(4, table_scale=1.0)
    params_b, params_c = f.mlp_base.params.detach().clone(), f.color_mlp.params.detach().clone()
    f = f.to(DEV)
    g = torch.Generator().manual_seed(5)
    binary = torch.rand(res, res, res, generator=g) < 0.1
    sg = ngp.SampleGrid(AABB, res).to(DEV)
    sg.set_binary_fields(binary.to(DEV))
    n = int(binary.sum())
    jitter = torch.rand(n, 3, generator=g)
    world, rgb, alpha, indices, dmask, smask = sg.query_radiance_and_density_from_camera(f, None, {}, DEV, jitter=jitter)
    w_ref, rgb_ref, alpha_ref, idx_ref, dmask_ref = N.dense_query(binary, jitter, torch.tensor(AABB), torch.tensor(AABB), params_b, params_c)
    assert torch.equal(indices.cpu(), idx_ref)
    np.testing.assert_allclose(world.cpu().numpy(), w_ref.numpy(), atol=1e-6)
    np.testing.assert_allclose(alpha.cpu().numpy()[:, 0], alpha_ref.numpy(), rtol=2e-2, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb_ref.numpy(), atol=5e-3)
    grid, mask = ngp.build_voxel_grid(world, rgb, alpha, indices, dmask & smask, res)
    assert grid.shape == (res, res, res, 7) and mask.dtype == torch.int64
    assert torch.all(mask[1:] > mask[:-1])
    flat = grid.view(-1, 7).cpu()
    assert torch.equal(flat[mask.cpu(), :3], world.cpu()[(dmask & smask).cpu()])
    untouched = torch.ones(res ** 3, dtype=torch.bool)
    untouched[mask.cpu()] = False
    assert float(flat[untouched].abs().sum()) == 0.0
    ngp.save_voxel_grid(str(tmp_path), grid, mask)
    assert torch.load(str(tmp_path / "voxel_grid.pt")).shape == (res, res, res, 7)


def test_query_dense_async_equals_the_form_with_readbacks():
    """SampleGrid.query_dense_async (no host readback: the occupied-cell count comes from the caller, the kept-cell count stays on the device — the
    evaluation pipeline's and bench.py --ngp's form) against query_dense + build_voxel_grid, bit for bit; a second mask over the same cells through
    write_kept_async (eval_ngp_nerf.py writes the density-mask grid and the density AND surface grid: :350-412)."""
    res = 64
    g = torch.Generator().manual_seed(21)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.5
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    f = f.to(DEV)
    binary = torch.rand(res, res, res, generator=g) < 0.12
    sg = ngp.SampleGrid(AABB, res).to(DEV)
    sg.set_binary_fields(binary.to(DEV))
    n = int(binary.sum())
    jitter = torch.rand(n, 3, generator=g).to(DEV)
    w0, rgb0, a0, idx0, dm0 = sg.query_dense(f, DEV, jitter=jitter)
    grid0, mask0 = ngp.build_voxel_grid(w0, rgb0, a0, idx0, dm0, res)
    w1, rgb1, a1, idx1, keep1, grid1, mask1, cnt1, rows = sg.query_dense_async(f, DEV, jitter=jitter, n_known=n)
    k = int(cnt1)
    assert 50 < k < n and k == mask0.numel()
    assert torch.equal(w0, w1) and torch.equal(rgb0, rgb1) and torch.equal(a0[:, 0], a1) and torch.equal(idx0, idx1) and torch.equal(dm0.view(torch.uint8), keep1)
    assert torch.equal(grid0, grid1) and torch.equal(mask0, mask1[:k])
    # a second, smaller mask over the same query
    sub = keep1 & (torch.rand(n, generator=g) < 0.5).to(DEV).view(torch.uint8)
    grid2, mask2, cnt2 = ngp.write_kept_async(rows, w1, rgb1, a1, idx1, sub, res)
    want_g, want_m = ngp.build_voxel_grid(w1, rgb1, a1[:, None], idx1, sub.view(torch.bool), res)      # (the scatter form: a mask without row tables)
    assert int(cnt2) == want_m.numel() and torch.equal(mask2[:int(cnt2)], want_m) and torch.equal(grid2, want_g)
    assert int(rows[1][0]) == n      # the device's own count of occupied cells: eval_pipeline compares it with the count it passed before a block's files are written


def test_unbounded_contraction_and_per_point_directions_vs_oracle():
    """NGPradianceField(unbounded=True).query_density and query_rgb / forward with one direction per point
    (conerf/radiance_fields/ngp.py:41-63,163-167,178-208)."""
    f = ngp.NGPradianceField(AABB, unbounded=True)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        p = f.mlp_base.params
        p[:3072] = torch.randn(3072, generator=g) * 0.25
        p[3072:] = torch.randn(p.numel() - 3072, generator=g) * 0.5
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    params_b, params_c = f.mlp_base.params.detach().clone(), f.color_mlp.params.detach().clone()
    f = f.to(DEV)
    x = (torch.rand(5000, 3, generator=g) - 0.5) * 20.0          # far outside the aabb: the contraction keeps them inside (0,1)^3
    x[:500] = (torch.rand(500, 3, generator=g) - 0.5) * 2.0
    d_ref, raw_ref = N.query_density(x, torch.tensor(AABB), params_b, unbounded=True)
    assert float((d_ref > 0).float().mean()) > 0.99             # the selector passes (almost) everything in an unbounded scene
    dens, feat = f.query_density(x.to(DEV), return_feat=True)
    scale = float(raw_ref.abs().max())
    assert (feat.cpu() - raw_ref[:, 1:]).abs().max() <= 4e-3 * scale
    ok = (raw_ref[:, 0] - 1.0).abs() < 8                          # exp() of huge logits amplifies the fp16 ulp
    np.testing.assert_allclose(dens[:, 0].cpu()[ok].numpy(), d_ref[ok].numpy(), rtol=2e-2, atol=1e-6)
    # one direction per point
    dirs = torch.nn.functional.normalize(torch.randn(5000, 3, generator=g), dim=1)
    rgb_ref = N.query_rgb(dirs, raw_ref, params_c)
    rgb = f.query_rgb(dirs.to(DEV), feat)
    assert (rgb.cpu() - N.query_rgb(dirs, torch.cat([raw_ref[:, :1], feat.cpu()], 1), params_c)).abs().max() < 4e-3
    assert (rgb.cpu() - rgb_ref).abs().max() < 2e-2
    rgb2, dens2 = f(x.to(DEV), dirs.to(DEV))
    assert torch.equal(rgb2, rgb) and torch.equal(dens2, dens)
    # a single shared direction agrees with the dense query's folded-bias kernel
    one = dirs[:1].expand(5000, 3).contiguous()
    raw16 = torch.cat([torch.zeros(5000, 1, device=DEV), feat], 1).half()
    a = f.query_rgb(one.to(DEV), feat)
    b = f.query_rgb_mean(raw16, one[:1].to(DEV))
    assert (a - b).abs().max() < 4e-3


def test_dense_query_at_the_baseline_size_128():
    """BASELINE config 4's size: a 128^3 block with ~325 k occupied cells (a ball of radius 1 in the [-1.5, 1.5]^3 block).  The oracle
    cannot walk 3e5 points in test time, so: (1) the occupied-cell enumeration (indices) is compared bit for bit with the oracle's
    enumeration of the same binary field, and world positions with its formula; (2) two launches give identical outputs; (3) density /
    colour / alpha of 4,096 sampled points against the oracle on exactly those points; (4) the grid writer's masks are the keep set."""
    res = 128
    f = _field(9, table_scale=1.0)
    params_b, params_c = f.mlp_base.params.detach().clone(), f.color_mlp.params.detach().clone()
    f = f.to(DEV)
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    binary = torch.stack([X, Y, Z], -1).norm(dim=-1) < 1.0
    n = int(binary.sum())
    assert n > 300000
    g = torch.Generator().manual_seed(6)
    jitter = torch.rand(n, 3, generator=g)
    sg = ngp.SampleGrid(AABB, res).to(DEV)
    sg.set_binary_fields(binary.to(DEV))
    world, rgb, alpha, indices, dmask = sg.query_dense(f, DEV, jitter=jitter.to(DEV))
    world2, rgb2, alpha2, indices2, dmask2 = sg.query_dense(f, DEV, jitter=jitter.to(DEV))
    assert torch.equal(world, world2) and torch.equal(rgb, rgb2) and torch.equal(alpha, alpha2) and torch.equal(indices, indices2) and torch.equal(dmask, dmask2)
    idx_ref = torch.nonzero(binary.flatten())[:, 0]
    assert torch.equal(indices.cpu(), idx_ref)                                     # bit-exact enumeration, ascending
    coords = torch.stack([idx_ref // (res * res), (idx_ref // res) % res, idx_ref % res], dim=1).float()
    w_ref = (coords + jitter) / res * 3.0 - 1.5
    np.testing.assert_allclose(world.cpu().numpy(), w_ref.numpy(), atol=2e-6)
    pick = torch.randperm(n, generator=g)[:4096]
    aabb = torch.tensor(AABB)
    d_ref, raw_ref = N.query_density(world.cpu()[pick], aabb, params_b)
    dirs = N.fixed_viewdirs()
    rgb_ref = torch.stack([N.query_rgb(dirs[k].expand(pick.shape[0], 3), raw_ref, params_c) for k in range(dirs.shape[0])]).mean(0)
    alpha_ref = torch.clip(1 - torch.exp(-1e-2 * d_ref), 0, 1)
    np.testing.assert_allclose(alpha.cpu().numpy()[pick, 0], alpha_ref.numpy().reshape(-1), rtol=2e-2, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy()[pick], rgb_ref.numpy(), atol=5e-3)
    dm_ref = (d_ref.reshape(-1) > 0.7)
    away = (d_ref.reshape(-1) - 0.7).abs() > 0.02 * 0.7                             # bit-exact away from the threshold (fp16 network output)
    assert torch.equal(dmask.cpu()[pick][away], dm_ref[away]) and float(away.float().mean()) > 0.9
    grid, mask = ngp.build_voxel_grid(world, rgb, alpha, indices, dmask, res)
    assert torch.equal(mask.cpu(), idx_ref[dmask.cpu()]) and torch.all(mask[1:] > mask[:-1])
    flat = grid.view(-1, 7)
    assert torch.equal(flat[mask, :3], world[dmask]) and torch.equal(flat[mask, 6], alpha[dmask].reshape(-1))
    assert int((flat.abs().sum(dim=1) > 0).sum()) <= int(dmask.sum())


@pytest.mark.parametrize("shape,fill", [((128, 128, 128), "ball"), ((40, 24, 72), "random"), ((16, 16, 16), "all"), ((24, 24, 24), "none"),
                                        ((33, 20, 50), "random")])
def test_fused_dense_query_equals_the_nonzero_form_bit_for_bit(shape, fill):
    """SampleGrid.query_dense through the library's cell-list launches (dreg_grid_occupied_count / _build, alpha / mask in the density
    launch, dreg_grid_write_kept) against the form that uses torch.nonzero, separate position / alpha launches and a boolean-index
    compaction: every output identical — cubes, boxes whose z extent is no multiple of 16 or 64, fewer cells than the x-ordered query's
    threshold, all cells, no cell."""
    rx, ry, rz = shape
    g = torch.Generator().manual_seed(21)
    if fill == "ball":
        c = [(torch.arange(r, dtype=torch.float32) + 0.5) / r * 3 - 1.5 for r in shape]
        X, Y, Z = torch.meshgrid(*c, indexing="ij")
        binary = torch.stack([X, Y, Z], -1).norm(dim=-1) < 1.0
    elif fill == "random":
        binary = torch.rand(shape, generator=g) < 0.3
    else:
        binary = torch.full(shape, fill == "all", dtype=torch.bool)
    n = int(binary.sum())
    f = _field(4, table_scale=1.0).to(DEV)
    jitter = torch.rand(n, 3, generator=g).to(DEV)
    sg = ngp.SampleGrid(AABB, list(shape)).to(DEV)
    sg.set_binary_fields(binary.to(DEV))
    outs = {}
    try:
        for fused in (False, True):
            ngp.FUSED_DENSE = fused
            world, rgb, alpha, indices, dmask = sg.query_dense(f, DEV, jitter=jitter)
            assert (getattr(dmask, "_dreg_rows", None) is not None) == fused
            o = [world, rgb, alpha, indices, dmask]
            if rx == ry == rz:
                grid, mask = ngp.build_voxel_grid(world, rgb, alpha, indices, dmask, rx)
                o += [grid, mask]
            outs[fused] = [t.clone() for t in o]
    finally:
        ngp.FUSED_DENSE = True
    assert outs[True][3].shape[0] == n
    for a, b, name in zip(outs[True], outs[False], ("world", "rgb", "alpha", "indices", "mask", "voxel_grid", "voxel_mask")):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), name
    if n and rx == ry == rz:
        assert 0 < outs[True][6].shape[0] <= n or fill == "all"


@pytest.mark.parametrize("unbounded", [False, True])
def test_density_query_forms_agree_bit_for_bit(unbounded, probe_lib):
    """The density query of a block's size has three forms (csrc/ngp.hip): the fused kernel, hash-grid levels pinned to the XCDs' L2s
    (two launches through a workspace), and either with an ORDER that lets a wave's lanes run along the tables' fastest axis.  Same
    arithmetic per point in each: identical density / raw; the order is what dreg_grid_x_order builds from the occupancy volume."""
    from dreg_nerf_amd import lib as L
    lib = L.load()
    f = ngp.NGPradianceField(AABB, unbounded=unbounded)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        p = f.mlp_base.params
        p[:3072] = torch.randn(3072, generator=g) * 0.25
        p[3072:] = torch.randn(p.numel() - 3072, generator=g) * 0.5
    f = f.to(DEV)
    res = (40, 36, 44)                                                     # three different extents: the axis roles must not mix
    binary = (torch.rand(res, generator=g) < 0.45)
    binary[0, 0, 0] = True; binary[-1, -1, -1] = True
    idx = torch.nonzero(binary.flatten())[:, 0].to(DEV)
    n = idx.shape[0]
    assert n >= 16384                                                      # the size from which query_raw takes the two-launch form
    # the order against plain torch: sort the ascending list by (z, y, x)
    rx, ry, rz = res
    z, y, x = idx % rz, (idx // rz) % ry, idx // (rz * ry)
    want = torch.argsort((z * ry + y) * rx + x, stable=True).int()
    nb = int(lib.dreg_grid_x_order_workspace_bytes(rx, ry, rz))
    ows = torch.empty(nb, dtype=torch.uint8, device=DEV)
    order = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    L.check(lib.dreg_grid_x_order(L.ptr(binary.to(DEV).contiguous().view(torch.uint8)), L.ptr(idx), L.ptr(order), L.ptr(ows), nb, rx, ry, rz, n, L.stream()), "dreg_grid_x_order")
    assert torch.equal(order, want)
    cells = torch.stack([x, y, z], 1).float()
    pts = ((cells + torch.rand(n, 3, generator=g).to(DEV)) / torch.tensor(res, device=DEV) * 3.4 - 1.7).contiguous()   # some points outside the aabb
    out = {}
    try:
        for xcd in (0, 1):
            lib.dreg_ngp_set_xcd_levels(xcd)
            for name, o in (("plain", None), ("ordered", order)):
                d, raw = f.query_raw(pts, order=o)
                out[(xcd, name)] = (d.clone(), raw.clone())
            d, raw = f.query_raw(pts[order.long()].contiguous(), order=order, x_in_slot_order=True)      # coordinates handed over in lane order
            out[(xcd, "ordered, coordinates in slot order")] = (d.clone(), raw.clone())
    finally:
        lib.dreg_ngp_set_xcd_levels(1)
    ref = out[(0, "plain")]
    assert torch.isfinite(ref[0]).all() and (ref[0] > 0).any() and (unbounded or (ref[0] == 0).any())      # (bounded: the points outside the aabb)
    for k, v in out.items():
        assert torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]), k

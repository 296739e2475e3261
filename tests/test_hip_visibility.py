"""GPU: fused surface-field visibility kernel (visibility.hip) against the CPU restatement (oracle/visibility_oracle.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import ngp, visibility  # noqa: E402
from oracle import visibility_oracle as VO  # noqa: E402

DEV = "cuda:0"
AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]


def test_surface_visibility_matches_oracle():
    res = 32
    g = torch.Generator().manual_seed(0)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.5
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
    params_b = f.mlp_base.params.detach().clone()
    f = f.to(DEV)
    # a thick occupied shell so that rays cross occupied and empty stretches
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    binary = (rad > 0.5) & (rad < 0.9)
    pts = (torch.rand(300, 3, generator=g) - 0.5) * 2.0
    cams = torch.tensor([[2.5, 0.3, 0.1], [-0.4, -2.2, 0.9], [0.2, 0.1, 0.05]])  # two outside the aabb, one inside
    dt = 3 * 3 ** 0.5 / 256
    lab_ref, best = VO.surface_visibility(cams, pts, binary, torch.tensor(AABB), torch.tensor(AABB), torch.tensor(AABB), params_b, dt)
    lab = visibility.surface_visibility(pts.to(DEV), cams.to(DEV), f, binary.to(DEV), AABB, AABB, dt).cpu().int()
    # rays whose surface field sits within 2 % of the cut-off may legitimately flip (fp16 network output)
    decided = ((best - 0.5).abs() > 0.02).all(dim=0)
    assert decided.float().mean() > 0.8
    assert torch.equal(lab[decided], lab_ref[decided])
    assert 0 < int(lab_ref.sum()) < 300


def test_compute_visibility_score_from_checkpoint(tmp_path):
    res = 32
    g = torch.Generator().manual_seed(1)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.5
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
    binary = torch.rand(res, res, res, generator=g) < 0.3
    poses = torch.eye(4)[None].repeat(4, 1, 1)
    poses[:, :3, 3] = torch.tensor([[2.0, 0, 0], [0, 2.0, 0], [0, 0, 2.0], [-2.0, 0.5, 0]])
    path = str(tmp_path / "model.pth")
    occ = ngp.OccupancyGrid(AABB, res)
    occ._binary.copy_(binary)
    torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": AABB, "unbounded": False, "near_plane": None, "far_plane": None,
                "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB, "render_step_size": 0.02,
                "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": 0}, path)
    xyz = (torch.rand(6, 50, 3, generator=g) - 0.5).to(DEV) * 2
    out = visibility.compute_visibility_score([xyz], path)
    assert out[0].shape == (6, 50, 1) and set(out[0].unique().tolist()) <= {0.0, 1.0}


def test_surface_visibility_at_block_scale():
    """3.2e5 points x 4 cameras through a 128^3 occupancy field (the size of one block's extraction / a training step's labels): two launches
    agree bit for bit, and a sample of 400 points agrees with the oracle's march wherever the surface field is not within 2 % of the cut-off."""
    res = 128
    g = torch.Generator().manual_seed(3)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.5
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
    params_b = f.mlp_base.params.detach().clone()
    f = f.to(DEV)
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    binary = (rad > 0.55) & (rad < 0.95)
    pts = (torch.rand(320000, 3, generator=g) - 0.5) * 2.2
    cams = torch.tensor([[2.5, 0.3, 0.1], [-0.4, -2.2, 0.9], [0.2, 0.1, 0.05], [0.1, 2.4, -0.6]])
    dt = 3 * 3 ** 0.5 / 1024
    lab = visibility.surface_visibility(pts.to(DEV), cams.to(DEV), f, binary.to(DEV), AABB, AABB, dt)
    lab2 = visibility.surface_visibility(pts.to(DEV), cams.to(DEV), f, binary.to(DEV), AABB, AABB, dt)
    assert torch.equal(lab, lab2) and lab.shape[0] == 320000
    pick = torch.randperm(320000, generator=g)[:400]
    lab_ref, best = VO.surface_visibility(cams, pts[pick], binary, torch.tensor(AABB), torch.tensor(AABB), torch.tensor(AABB), params_b, dt)
    decided = ((best - 0.5).abs() > 0.02).all(dim=0)
    assert decided.float().mean() > 0.8
    assert torch.equal(lab.cpu().int()[pick][decided], lab_ref[decided])
    assert 0 < int(lab.sum()) < 320000


@pytest.mark.parametrize("wscale", [0.4, 3.0])
def test_persistent_ray_queue_gives_the_lock_step_labels(wscale):
    """The persistent kernel (lanes refilled from a ray queue, rays of already-labelled points skipped or abandoned, empty space walked
    through a coarse occupancy grid in LDS) against the lock-step launch: a label is an OR over the cameras of per-ray decisions that
    are computed identically, so the labels are EQUAL — for a transparent field (rays march through the whole block) and an opaque one."""
    res = 128
    g = torch.Generator().manual_seed(11)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * wscale
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
    f = f.to(DEV)
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    binary = ((rad > 0.6) & (rad < 0.9) & (X + 0.3 * Y > -0.7)).to(DEV)        # a shell with a piece cut away: some points see no surface
    pts = ((torch.rand(6000, 3, generator=g) - 0.5) * 2.4).to(DEV)
    cams = (torch.nn.functional.normalize(torch.randn(12, 3, generator=g), dim=-1) * 2.8).to(DEV)
    cams[0] = torch.tensor([0.1, 0.0, 0.05])                                    # one camera inside the block
    dt = 3 * 3 ** 0.5 / 512
    labs = {}
    try:
        for persistent, coarse in ((False, False), (True, False), (True, True)):
            visibility.PERSISTENT, visibility.COARSE = persistent, coarse
            labs[(persistent, coarse)] = visibility.surface_visibility(pts, cams, f, binary, AABB, AABB, dt).clone()
    finally:
        visibility.PERSISTENT = visibility.COARSE = True
    ref = labs[(False, False)]
    assert 0 < int(ref.sum()) and (wscale > 1.0 or int(ref.sum()) < pts.shape[0])      # (the opaque field shows every point to some camera)
    for k, v in labs.items():
        assert torch.equal(v, ref), k


def test_pass_bound_of_the_persistent_kernel_is_reported(probe_lib):
    """A persistent launch that leaves its march loop through the safety bound (never on a real extraction; forced here with a bound of
    two passes) must not leave points silently unlabelled: bit 63 of its ray counter is set and the next look at it raises."""
    from dreg_nerf_amd import lib as L
    lib = L.load()
    res = 64
    g = torch.Generator().manual_seed(5)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 3.0           # an opaque field: every ray finds a surface
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
    f = f.to(DEV)
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    binary = ((rad > 0.6) & (rad < 0.9)).to(DEV)
    pts = ((torch.rand(20000, 3, generator=g) - 0.5) * 2.4).to(DEV)
    cams = (torch.nn.functional.normalize(torch.randn(6, 3, generator=g), dim=-1) * 2.8).to(DEV)
    dt = 3 * 3 ** 0.5 / 256
    visibility.OVERRUN.check(wait=True)
    assert lib.dreg_visibility_set_pass_bound(2) == 0
    lib.dreg_visibility_set_waves(64)
    try:
        visibility.surface_visibility(pts, cams, f, binary, AABB, AABB, dt)
        with pytest.raises(L.DregError, match="pass bound"):
            visibility.OVERRUN.check(wait=True)
    finally:
        assert lib.dreg_visibility_set_pass_bound(0) == 0
        lib.dreg_visibility_set_waves(0)
    lab = visibility.surface_visibility(pts, cams, f, binary, AABB, AABB, dt)
    visibility.OVERRUN.check(wait=True)                  # the default bound is never reached
    assert 0 < int(lab.sum())


def test_block_cache_keeps_inference_copies_and_is_filled_by_the_loader_thread(tmp_path):
    """visibility.load_block keeps a block as its fp16 inference copy + occupancy grid (the fp32 parameters are released): ~27 MB instead
    of ~80, so that an epoch's blocks stay resident; labels from a frozen block equal those from the block as loaded; the prefetching
    loader loads the blocks of the samples it prepares on ITS thread, the training thread then only finds cache hits."""
    from dreg_nerf_amd.dataset import PrefetchLoader
    res = 32
    g = torch.Generator().manual_seed(4)
    f = ngp.NGPradianceField(AABB)
    with torch.no_grad():
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 1.5
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
    binary = torch.rand(res, res, res, generator=g) < 0.3
    poses = torch.eye(4)[None].repeat(4, 1, 1)
    poses[:, :3, 3] = torch.tensor([[2.0, 0, 0], [0, 2.0, 0], [0, 0, 2.0], [-2.0, 0.5, 0]])
    occ = ngp.OccupancyGrid(AABB, res)
    occ._binary.copy_(binary)
    paths = []
    for k in range(2):
        p = str(tmp_path / f"block_{k}.pth")
        torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": AABB, "unbounded": False, "near_plane": None, "far_plane": None,
                    "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB, "render_step_size": 0.02,
                    "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": k}, p)
        paths.append(p)
    xyz = (torch.rand(3, 200, 3, generator=g) - 0.5).to(DEV) * 2
    # reference labels: the field as loaded (fp32 parameters present)
    want = visibility.surface_visibility(xyz.reshape(-1, 3), poses[:, :3, 3].to(DEV), f.to(DEV), binary.to(DEV), AABB, AABB, 0.02)
    visibility.clear_block_cache()

    class _DS:                                   # two samples naming the two blocks
        def __len__(self): return 2
        def __getitem__(self, i): return {"src_nerf_path": paths[i], "tgt_nerf_path": paths[1 - i], "x": torch.zeros(1, device=DEV)}
    items = list(PrefetchLoader(_DS(), [0, 1], device=torch.device(DEV), depth=2))
    assert len(items) == 2
    assert len(visibility._block_cache) == 2     # filled by the loader thread
    before = dict(visibility._block_cache)
    got = visibility.compute_visibility_score([xyz], paths[0])[0]
    assert {k: id(v[0]) for k, v in visibility._block_cache.items()} == {k: id(v[0]) for k, v in before.items()}    # a hit: nothing reloaded
    assert torch.equal(got.view(-1) > 0, want)
    field = next(iter(visibility._block_cache.values()))[0]
    assert field.mlp_base.params.numel() == 0 and field._prepared()[0].numel() == 12602992                          # frozen: fp16 only
    per_block = next(iter(visibility._block_cache.values()))[3]
    assert per_block < 30 << 20, per_block
    visibility.clear_block_cache()


def test_batched_labels_of_several_blocks_equal_the_per_block_calls(tmp_path):
    """compute_visibility_scores_batched (one persistent launch over all blocks of a training step) against compute_visibility_score per block."""
    res = 32
    g = torch.Generator().manual_seed(9)
    reqs = []
    for k in range(3):
        f = ngp.NGPradianceField(AABB)
        with torch.no_grad():
            f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * (0.1, 0.6, 2.5)[k]      # a transparent, a foggy and an opaque block
            f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g) * 2.0
        occ = ngp.OccupancyGrid(AABB, res)
        occ._binary.copy_(torch.rand(res, res, res, generator=g) < 0.3)
        poses = torch.eye(4)[None].repeat(3 + k, 1, 1)
        poses[:, :3, 3] = torch.nn.functional.normalize(torch.randn(3 + k, 3, generator=g), dim=-1) * 2.2
        p = str(tmp_path / f"block_{k}.pth")
        torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": AABB, "unbounded": False, "near_plane": None, "far_plane": None,
                    "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB, "render_step_size": 0.02,
                    "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": k}, p)
        reqs.append(((torch.rand(2 + k, 150 + 37 * k, 3, generator=g) - 0.5).to(DEV) * 2, p))
    visibility.clear_block_cache()
    want = [visibility.compute_visibility_score([x], p)[0] for x, p in reqs]
    got = visibility.compute_visibility_scores_batched(reqs)
    assert len(got) == 3
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)
    assert 0 < sum(float(b.sum()) for b in want) < sum(b.numel() for b in want)
    visibility.clear_block_cache()

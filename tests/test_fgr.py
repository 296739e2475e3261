"""Row N4: the Fast Global Registration baseline in plain torch (open3d is not in the image: parity unpinned —
these tests check the published algorithm's behaviour: a known rigid motion between two partially overlapping, noisy samplings of
one surface is recovered)."""
import math

import torch

from dreg_nerf_amd import fgr


def _surface(n, g):
    u = torch.rand(n, generator=g) * 2 * math.pi
    v = torch.acos(2 * torch.rand(n, generator=g) - 1)
    r = 1.0 + 0.25 * torch.sin(3 * u) * torch.sin(2 * v) + 0.15 * torch.cos(5 * v)
    return torch.stack([1.2 * r * torch.sin(v) * torch.cos(u), 0.8 * r * torch.sin(v) * torch.sin(u), 0.6 * r * torch.cos(v)], 1)


def _rigid(axis, ang, t):
    ax = torch.tensor(axis) / torch.tensor(axis).norm()
    K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K, torch.tensor(t)


def test_voxel_down_sample_and_fpfh_shapes():
    g = torch.Generator().manual_seed(1)
    p = _surface(3000, g)
    d = fgr.voxel_down_sample(p, 0.1)
    assert 100 < d.shape[0] < 3000 and d.shape[1] == 3
    # every output point is the mean of the input points of one voxel: it lies inside the cloud's bounding box
    assert bool(((d >= p.min(0).values - 1e-6) & (d <= p.max(0).values + 1e-6)).all())
    n = fgr.estimate_normals(d, 0.2, 30)
    assert torch.allclose(n.norm(dim=1), torch.ones(d.shape[0]), atol=1e-4)
    f = fgr.compute_fpfh(d, n, 0.5, 100)
    assert f.shape == (d.shape[0], 33) and bool((f >= 0).all())
    # each third of SPFH sums to 100 and so does each normalised third of the neighbour sum: 200 per third wherever a point has neighbours
    s = f.view(-1, 3, 11).sum(dim=2)
    assert torch.allclose(s, torch.full_like(s, 200.0), atol=1e-2)
    # FPFH is invariant under a rigid motion of the cloud
    R, t = _rigid([0.2, 0.9, -0.4], 1.1, [0.5, -0.3, 0.2])
    d2 = d @ R.T + t
    f2 = fgr.compute_fpfh(d2, fgr.estimate_normals(d2, 0.2, 30), 0.5, 100)
    assert float((f - f2).abs().max()) < 1.0


def test_fgr_recovers_a_known_rigid_motion():
    g = torch.Generator().manual_seed(0)
    p = _surface(6000, g)
    R, t = _rigid([0.3, -0.5, 0.8], 0.6, [0.3, -0.2, 0.15])
    src = p[:4000]
    tgt = p[2000:] @ R.T + t + 0.002 * torch.randn(4000, 3, generator=g)
    T, sec = fgr.run_registration(src, tgt, 0.05)
    assert T.shape == (4, 4) and sec > 0 and torch.allclose(T[3], torch.tensor([0, 0, 0, 1], dtype=T.dtype))
    Rp, tp = T[:3, :3].float(), T[:3, 3].float()
    ang = math.degrees(math.acos(max(-1.0, min(1.0, float(((Rp.T @ R).trace() - 1) / 2)))))
    assert ang < 2.0 and float((tp - t).norm()) < 0.03, (ang, float((tp - t).norm()))
    assert torch.allclose(Rp @ Rp.T, torch.eye(3), atol=1e-5) and abs(float(torch.det(Rp)) - 1) < 1e-5
    # deterministic (seeded tuple test)
    T2, _ = fgr.run_registration(src, tgt, 0.05)
    assert torch.equal(T, T2)

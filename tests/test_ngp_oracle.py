"""CPU: known-answer tests of the NGP oracle (parity unpinned upstream: tiny-cuda-nn is absent; these pin the
oracle to the published algorithm's definitions — SURVEY.md Appendix B)."""
import torch

from oracle import ngp_oracle as N


def test_level_table_sizes():
    rows, total = N.level_table()
    assert [r["res"] for r in rows] == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]
    assert [r["size"] for r in rows[:5]] == [4096, 13824, 39304, 117656, 357912]
    assert all(r["size"] == 524288 and r["hashed"] for r in rows[5:]) and not any(r["hashed"] for r in rows[:5])
    assert total == 6299960 and N.n_grid_params() + 3072 == 12602992


def test_all_ones_table_gives_unit_features():
    _, total = N.level_table()
    u = torch.rand(50, 3, generator=torch.Generator().manual_seed(0))
    enc = N.hash_encode(u, torch.ones(total, 2))
    assert torch.equal(enc, torch.ones(50, 32))


def test_dense_level_is_trilinear_at_vertices_and_hash_spot_values():
    rows, total = N.level_table()
    table = torch.zeros(total, 2)
    lv = rows[0]  # res 16, scale 15: vertex v sits at u = (v - 0.5)/15
    v = torch.tensor([3, 7, 11])
    table[lv["offset"] + int(v[0] + v[1] * 16 + v[2] * 256), 0] = 1.0
    u = ((v.float() - 0.5) / 15.0)[None]
    enc = N.hash_encode(u, table)
    assert abs(float(enc[0, 0]) - 1.0) < 1e-3 and float(enc[0, 1]) == 0.0
    # spatial hash of the published primes
    x, y, z = 5, 9, 1000
    assert ((x * 1) ^ ((y * 2654435761) & 0xFFFFFFFF) ^ ((z * 805459861) & 0xFFFFFFFF)) % 524288 == \
        (5 ^ (23889921849 & 0xFFFFFFFF) ^ (805459861000 & 0xFFFFFFFF)) % 524288


def test_density_wrapper_semantics():
    _, total = N.level_table()
    g = torch.Generator().manual_seed(1)
    params = torch.cat([torch.randn(3072, generator=g) * 0.2, torch.randn(2 * total, generator=g) * 0.5])
    aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    x = torch.tensor([[0.1, 0.2, 0.3], [1.6, 0.0, 0.0], [-1.5, 0.0, 0.0]])
    d, raw = N.query_density(x, aabb, params)
    assert d[1] == 0 and d[2] == 0  # outside / on the boundary: selector is strict (ngp.py:156)
    assert abs(float(d[0]) - float(torch.exp(raw[0, 0] - 1))) < 1e-6
    assert N.fixed_viewdirs().shape == (18, 3) and float(N.fixed_viewdirs()[0].abs().sum()) == 0.0

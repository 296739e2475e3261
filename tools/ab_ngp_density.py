#!/usr/bin/env python3
"""The NGP density query on the block of `bench.py --ngp` (325 k occupied cells of a 128^3 grid, one jittered sample per cell): the fused
kernel with 1, 2, 4 or 8 hash-grid levels' corner gathers in flight per lane (dreg_ngp_set_density_unroll) and the two-launch form whose
encoding pins two levels to every XCD's L2 (dreg_ngp_set_xcd_levels); outputs must not change.
usage: python tools/ab_ngp_density.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ngp, lib as L
dev = torch.device("cuda", 0)
lib = L.use_probe()
res, aabb = 128, [-1.5] * 3 + [1.5] * 3
g = torch.Generator().manual_seed(100)
f = ngp.NGPradianceField(aabb)
with torch.no_grad():
    f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.4
    f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
    f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
f = f.to(dev)
c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
cells = (torch.stack([X, Y, Z], -1).norm(dim=-1) < 1.0).nonzero().float()
x = ((cells + torch.rand(cells.shape, generator=g)) / res * 3 - 1.5).to(dev)
ref = None
ms = {}
for rnd in range(3):
    for u in (1, 2, 4, 8, "xcd"):
        lib.dreg_ngp_set_xcd_levels(1 if u == "xcd" else 0)
        lib.dreg_ngp_set_density_unroll(1 if u == "xcd" else u)
        d, raw = f.query_raw(x); torch.cuda.synchronize()
        if ref is None:
            ref = (d.clone(), raw.clone())
        assert torch.equal(d, ref[0]) and torch.equal(raw, ref[1]), u
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f.query_raw(x)
        e1.record(); torch.cuda.synchronize()
        ms.setdefault(u, []).append(e0.elapsed_time(e1) * 50)
lib.dreg_ngp_set_density_unroll(1)
# the same points with the FIRST coordinate varying fastest over the lanes (the table's fastest axis: idx = x + y * res + z * res^2):
cells_x = cells[:, [2, 1, 0]].contiguous()          # nonzero() enumerates with the last axis fastest: swap the roles of x and z
x2 = ((cells_x + torch.rand(cells.shape, generator=g)) / res * 3 - 1.5).to(dev)
for u in ("fused, x fastest", "xcd, x fastest"):
    lib.dreg_ngp_set_xcd_levels(1 if u.startswith("xcd") else 0)
    for rnd in range(3):
        f.query_raw(x2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f.query_raw(x2)
        e1.record(); torch.cuda.synchronize()
        ms.setdefault(u, []).append(e0.elapsed_time(e1) * 50)
lib.dreg_ngp_set_xcd_levels(1)
# the product path: the cells as they come (z fastest) + the order dreg_grid_x_order builds from the occupancy volume, per query
binary = (torch.stack([X, Y, Z], -1).norm(dim=-1) < 1.0).to(dev)
idx = torch.nonzero(binary.flatten())[:, 0]
nb = int(lib.dreg_grid_x_order_workspace_bytes(res, res, res))
ows = torch.empty(nb, dtype=torch.uint8, device=dev)
order = torch.empty(idx.shape[0], dtype=torch.int32, device=dev)
b8 = binary.contiguous().view(torch.uint8)
x_slot = x[order.long()].contiguous() if False else None
def with_order():
    L.check(lib.dreg_grid_x_order(L.ptr(b8), L.ptr(idx), L.ptr(order), L.ptr(ows), nb, res, res, res, idx.shape[0], L.stream()), "dreg_grid_x_order")
    return f.query_raw(xs, order=order, x_in_slot_order=True)
only0 = lambda: L.check(lib.dreg_grid_x_order(L.ptr(b8), L.ptr(idx), L.ptr(order), L.ptr(ows), nb, res, res, res, idx.shape[0], L.stream()), "dreg_grid_x_order")
only0(); torch.cuda.synchronize()
xs = x[order.long()].contiguous()           # the positions in lane order (the product path writes them from dreg_grid_sample_points_ordered)
d, raw = with_order(); torch.cuda.synchronize()
assert torch.equal(d, ref[0]) and torch.equal(raw, ref[1])
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): with_order()
    e1.record(); torch.cuda.synchronize()
    ms.setdefault("xcd + order built per query (product path)", []).append(e0.elapsed_time(e1) * 50)
def only_order():
    L.check(lib.dreg_grid_x_order(L.ptr(b8), L.ptr(idx), L.ptr(order), L.ptr(ows), nb, res, res, res, idx.shape[0], L.stream()), "dreg_grid_x_order")
for name, fn in (("order build alone (3 launches)", only_order), ("xcd + a prebuilt order, coordinates in lane order", lambda: f.query_raw(xs, order=order, x_in_slot_order=True))):
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms.setdefault(name, []).append(e0.elapsed_time(e1) * 50)
for u, v in ms.items():
    print(f"levels in flight {u}: " + " ".join(f"{t:.1f}" for t in v) + f" us per query of {x.shape[0]} points (density kernel + its torch wrappers), best {min(v):.1f}")

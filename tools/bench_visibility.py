#!/usr/bin/env python3
"""The surface-visibility ray march (csrc/visibility.hip): lock-step launch against the persistent ray-queue kernel, labels must be equal.
Points: every 4th occupied cell of a shell-shaped 128^3 occupancy grid; NCAM cameras on a sphere of radius 3.
usage: python tools/bench_visibility.py [NCAM] [WSCALE]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ngp, visibility
NCAM = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WSCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4       # scale of the density MLP's weights: 0.4 = thin fog (rays march through the whole shell), 3 = opaque surfaces
DEV = "cuda"; AABB = [-1.5] * 3 + [1.5] * 3
g = torch.Generator().manual_seed(0)
f = ngp.NGPradianceField(AABB)
with torch.no_grad():
    f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * WSCALE
    f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
f = f.to(DEV)
res = 128
c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
shell = ((rad > 0.6) & (rad < 0.9)).to(DEV)
cells = shell.nonzero().float()[::4]
pts = (((cells + 0.5) / res) * 3 - 1.5).contiguous()
cams = torch.nn.functional.normalize(torch.randn(NCAM, 3, generator=g), dim=-1).to(DEV) * 3.0
dt = 3 * 3 ** 0.5 / 1024
out = {}
for mode in (False, True, "coarse", False, True, "coarse"):
    visibility.PERSISTENT = bool(mode)
    visibility.COARSE = mode == "coarse"
    fn = lambda: visibility.surface_visibility(pts, cams, f, shell, AABB, AABB, dt)
    lab = fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): fn()
    e.record(); torch.cuda.synchronize()
    out.setdefault(mode, []).append((s.elapsed_time(e) / 3, lab))
visibility.PERSISTENT = visibility.COARSE = True
nr = pts.shape[0] * NCAM
names = {False: "lock-step launch                    ", True: "persistent ray queue                ", "coarse": "persistent + coarse occupancy in LDS"}
for mode, v in out.items():
    ms = min(t for t, _ in v)
    print(f"{names[mode]}: {pts.shape[0]} pts x {NCAM} cams = {nr / 1e6:.2f} M rays: {ms:.2f} ms -> {nr / ms / 1e3:.1f} Mrays/s, visible {int(v[0][1].sum())}")
print(f"MLP weight scale {WSCALE}; labels equal:", bool(torch.equal(out[False][0][1], out[True][0][1]) and torch.equal(out[False][0][1], out["coarse"][0][1])))

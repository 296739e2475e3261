#!/usr/bin/env python3
"""Grid extraction of one 128^3 block INCLUDING the surface mask (eval_ngp_nerf.py:336-412 with sample_grid.py:244-318: every occupied
cell's sample is ray-marched from every training camera) next to the dense part alone (bench.py --ngp).  Generated field (weight scale
WSCALE: 3 = opaque surfaces), NCAM cameras on a sphere.  usage: python tools/bench_extract_with_surface.py [NCAM] [WSCALE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ngp, visibility
NCAM = int(sys.argv[1]) if len(sys.argv) > 1 else 50
WSCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = torch.device("cuda", 0)
res, aabb = 128, [-1.5] * 3 + [1.5] * 3
g = torch.Generator().manual_seed(100)
f = ngp.NGPradianceField(aabb)
with torch.no_grad():
    f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * WSCALE
    f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
    f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
f = f.to(dev)
c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
binary = (torch.stack([X, Y, Z], -1).norm(dim=-1) < 1.0).to(dev)
sg = ngp.SampleGrid(aabb, res).to(dev)
sg.set_binary_fields(binary)
n = int(binary.sum())
jitter = torch.rand(n, 3, generator=g).to(dev)
poses = torch.eye(4)[None].repeat(NCAM, 1, 1)
poses[:, :3, 3] = torch.nn.functional.normalize(torch.randn(NCAM, 3, generator=g), dim=-1) * 3.0
meta = {"aabb": aabb, "render_step_size": 3 * 3 ** 0.5 / 1024, "cone_angle": 0.0, "alpha_thre": 0.0, "camera_poses": poses}


def run(with_surface, persistent=True):
    visibility.PERSISTENT = persistent
    def block():
        w, rgb, a, idx, dm, sm = sg.query_radiance_and_density_from_camera(f, None, meta if with_surface else {}, dev, jitter=jitter)
        return ngp.build_voxel_grid(w, rgb, a, idx, dm & sm, res), sm
    (_, sm) = block(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): block()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3, int(sm.sum())


a, _ = run(False)
b, vis = run(True, True)
c_, vis2 = run(True, False)
visibility.PERSISTENT = True
print(f"{n} occupied cells, {NCAM} cameras ({n * NCAM / 1e6:.1f} M rays), field weight scale {WSCALE}: dense part alone {a:.2f} ms per block; with the surface mask {b:.2f} ms "
      f"(persistent ray queue; {vis} cells visible), {c_:.2f} ms (lock-step kernel; {vis2} visible)")

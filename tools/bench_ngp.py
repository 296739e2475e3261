"""Throughput of the NGP dense-query kernels (BASELINE.json configs[3]): density, 18-direction colour, full grid extraction
of one 128^3 block, and the surface-visibility kernel.  usage: python tools/bench_ngp.py [Np]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ngp, visibility

DEV = "cuda"
Np = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
AABB = [-1.5] * 3 + [1.5] * 3
g = torch.Generator().manual_seed(0)
f = ngp.NGPradianceField(AABB)
with torch.no_grad():
    f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.4
    f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
    f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
f = f.to(DEV)
x = ((torch.rand(Np, 3, generator=g) - 0.5) * 2.9).to(DEV)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

ms = timeit(lambda: f.query_raw(x))
print(f"density: {Np} pts {ms:.3f} ms  {Np / ms / 1e6:.2f} Gpts/s  gather {Np * 512 / ms / 1e9:.2f} TB/s  MLP {Np * 3072 * 2 / ms / 1e9:.1f} TFLOP/s")
d, raw = f.query_raw(x)
dirs = ngp.SampleGrid._generate_fixed_viewing_directions().to(DEV)
ms = timeit(lambda: f.query_rgb_mean(raw, dirs))
fl = Np * (18 * (64 * 64 + 16 * 64) + 64 * 32) * 2
print(f"rgb x18: {ms:.3f} ms  {Np / ms / 1e6:.2f} Gpts/s  {fl / ms / 1e9:.1f} TFLOP/s (fp16 MFMA)")
# full block extraction at 128^3 with 10 % occupied cells
res = 128
binary = (torch.rand(res, res, res, generator=g) < 0.1).to(DEV)
sg = ngp.SampleGrid(AABB, res).to(DEV)
sg.set_binary_fields(binary)
def extract():
    w, rgb, a, idx, dm = sg.query_dense(f, DEV)
    return ngp.build_voxel_grid(w, rgb, a, idx, dm, res)
ms = timeit(extract, 3)
n_occ = int(binary.sum())
print(f"block extraction (dense query + grid writer): {n_occ} occupied cells, {ms:.3f} ms -> {1e3 / ms:.1f} blocks/s")
# visibility: 50 cameras x 20k points
c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
shell = ((rad > 0.6) & (rad < 0.9)).to(DEV)
cams = torch.nn.functional.normalize(torch.randn(50, 3, generator=g), dim=-1).to(DEV) * 3.0
pts = ((torch.rand(20000, 3, generator=g) - 0.5) * 2).to(DEV)
dt = 3 * 3 ** 0.5 / 1024
ms = timeit(lambda: visibility.surface_visibility(pts, cams, f, shell, AABB, AABB, dt), 3)
print(f"surface visibility: 50 cams x 20000 pts = 1.0 M rays, dt {dt:.2e}: {ms:.3f} ms -> {1e6 / ms / 1e3:.2f} Mrays/s")

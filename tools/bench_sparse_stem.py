"""dreg_sparse_stem_fwd at the benchmark's size (8 x 64^3 x 64 -> 32^3) with no listed row, a spherical shell of listed rows (~5 %) and every
row listed: how much of the pooling pass is the constant fast path, how much the 27-tap windows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L
lib = L.use_probe()
dev = torch.device("cuda", 0)
B, D, C = 8, 64, 64
V, Do = D ** 3, 32
z, y, x = torch.meshgrid(torch.arange(D), torch.arange(D), torch.arange(D), indexing="ij")
r = ((z - 31.5) ** 2 + (y - 31.5) ** 2 + (x - 31.5) ** 2).sqrt().flatten()
def run(mask_1grid, label):
    rows1 = torch.nonzero(mask_1grid)[:, 0]
    rows = torch.cat([rows1 + b * V for b in range(B)]).int().to(dev)
    xx = torch.zeros(B * V, C, dtype=torch.bfloat16, device=dev)
    xx[rows.long()] = torch.randn(rows.numel(), C, device=dev).bfloat16()
    Po = B * Do ** 3
    o = dict(pooled=torch.empty(Po, C, dtype=torch.bfloat16, device=dev), arg=torch.empty(Po, C, dtype=torch.uint8, device=dev), xam=torch.empty(Po, C, dtype=torch.bfloat16, device=dev),
             pmask=torch.empty(Po, dtype=torch.uint8, device=dev), act=torch.empty(B * V, C, dtype=torch.bfloat16, device=dev), rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev),
             ss=torch.empty(B, C, 2, device=dev), mr=torch.empty(B, C, 2, device=dev), ws=torch.empty(int(lib.dreg_sparse_stem_workspace_floats(B, Do, Do, Do, C)), device=dev))
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    def call():
        L.check(lib.dreg_sparse_stem_fwd(L.ptr(xx), L.ptr(rows), rows.numel(), L.ptr(rows), rows.numel(), L.ptr(o["act"]), L.ptr(o["pooled"]), L.ptr(o["arg"]), L.ptr(o["xam"]), L.ptr(o["pmask"]),
                                         L.ptr(gamma), L.ptr(beta), L.ptr(o["rm"]), L.ptr(o["rv"]), L.ptr(o["ss"]), L.ptr(o["mr"]), L.ptr(o["ws"]), B, D, D, D, Do, Do, Do, C, 1e-5, 0.1, 1, 1, L.stream()), "fwd")
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): call()
    torch.cuda.synchronize()
    print(f"{label:28s} rows/grid {rows1.numel():7d}  marked windows {float(o['pmask'].float().mean()):.3f}  {1e6 * (time.perf_counter() - t0) / 20:7.1f} us per call (all six launches)")
for nb in (8192, 4096, 2048, 1024):
    lib.dreg_sstem_set_pool_blocks(nb)
    print("pool launch workgroups", nb)
    run((r > 20) & (r < 21.2), "shell")
    run((r > 10) & (r < 25), "thick shell")
lib.dreg_sstem_set_pool_blocks(8192)
run(torch.ones(V, dtype=torch.bool), "every row")

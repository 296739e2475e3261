"""The stem (5^3 / stride 2, 8 -> 64 channels, 128^3 -> 64^3, 8 grids) on the list of output voxels whose receptive field holds an occupied
input voxel, against the dense launch with W-row occupancy flags: forward and weight gradient.  usage: python tools/bench_stem_rows.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, ops, synth
from dreg_nerf_amd.regtr import NeRFRegTr
dev = torch.device("cuda", 0); lib = L.load()
batch = [synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose()) for i in range(4)]
m = NeRFRegTr(precision="bf16").to(dev)
grids = [d[s + "_grid"].to(dev) for d in batch for s in ("src", "tgt")] if "src_grid" in batch[0] else None
keys = list(batch[0].keys()); print(keys)
idxs = [d[s + "_mask"].to(dev) for d in batch for s in ("src", "tgt")]
vals = [d[s + "_vals"].to(dev) if (s + "_vals") in d else None for d in batch for s in ("src", "tgt")]
B, R = 8, 128
# dense input [B,Z,X,Y,8] bf16 from the occupied voxels: flat index ((x * Yr + y) * Zr + z)
x = torch.zeros(B, R, R, R, 8, dtype=torch.bfloat16, device=dev)
occ = torch.zeros(B, R, R, R, dtype=torch.bool, device=dev)
for b, f in enumerate(idxs):
    z, y, xx = f % R, (f // R) % R, f // (R * R)
    x[b, z, xx, y, :4] = torch.randn(f.shape[0], 4, device=dev).bfloat16()
    occ[b, z, xx, y] = True
# T: output voxels o with an occupied input in [2o - 2, 2o + 2]^3  (max-pool of the occupancy with window 5, stride 2, pad 2)
T = torch.nn.functional.max_pool3d(occ.float()[:, None], 5, 2, 2)[:, 0] > 0
rows = torch.nonzero(T.flatten())[:, 0].int().contiguous()
print("occupied", int(occ.sum()), "stem rows", rows.numel(), "of", T.numel())
w = torch.randn(64, 8, 5, 5, 5, device=dev) * 0.05
wp = ops.packed_weight(w, 8, False, L.DT_BF16)
inocc = occ.any(dim=3).to(torch.uint8).contiguous()       # [B, Z, X]
rowocc = NeRFRegTr._stem_row_occupancy(inocc)
y_d = torch.empty(B, 64, 64, 64, 64, dtype=torch.bfloat16, device=dev)
y_r = torch.zeros_like(y_d)
def ev(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
dense = lambda: L.check(lib.dreg_conv3d_igemm_occ(L.ptr(x), L.ptr(wp), L.ptr(y_d), None, None, B, R, R, R, 8, 64, 64, 64, 64, 5, 2, 2, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, L.ptr(rowocc), L.stream()), "dense")
rowsf = lambda: L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wp), L.ptr(y_r), None, None, L.ptr(rows), rows.numel(), B, R, R, R, 8, 64, 64, 64, 64, 5, 2, 2, 0, 0, 0, 0, 0, 0, 0, L.stream()), "rows")
td, tr = ev(dense), ev(rowsf)
print(f"forward: dense + row occupancy {td:.1f} us | row list {tr:.1f} us | identical {torch.equal(y_d, y_r)}")
gy = torch.randn_like(y_d)
nb = int(lib.dreg_conv3d_wgrad_workspace_bytes(B, 64, 64, 64, 8, 64, 5, 0))
ws1 = torch.zeros(nb + 256, dtype=torch.uint8, device=dev); ws2 = torch.zeros_like(ws1)
wd = lambda: L.check(lib.dreg_conv3d_wgrad_partials(L.ptr(gy), L.ptr(x), L.ptr(ws1), nb, None, 0, B, R, R, R, 8, 4, 64, 64, 64, 64, 5, 2, 2, L.ptr(rowocc), L.stream()), "wd")
wr = lambda: L.check(lib.dreg_conv3d_wgrad_partials(L.ptr(gy), L.ptr(x), L.ptr(ws2), nb, L.ptr(rows), rows.numel(), B, R, R, R, 8, 4, 64, 64, 64, 64, 5, 2, 2, None, L.stream()), "wr")
td, tr = ev(wd), ev(wr)
ns = int(lib.dreg_conv3d_wgrad_splits(B, 64, 64, 64, 8, 64, 5, 0)); kp = int(lib.dreg_conv3d_kpad(5, 8, 0))
p1 = ws1[:ns * 64 * kp * 4].view(torch.float32).view(ns, 64, kp).sum(0)
nw = int(ws2[ns * 64 * kp * 4: ns * 64 * kp * 4 + 4].view(torch.int32)[0])
p2 = ws2[:nw * 64 * kp * 4].view(torch.float32).view(nw, 64, kp).sum(0)
print(f"weight gradient: dense + row occupancy {td:.1f} us | row list {tr:.1f} us ({nw} of {ns} slices) | rel diff {float((p1 - p2).norm() / p1.norm()):.2e}")

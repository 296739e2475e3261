"""The active-set 3^3 convolution with staged-neighbourhood reuse (csrc/conv_brick.hip) against the row-list implicit-GEMM kernel on the
head layers' shapes: shell-R active sets at 64^3 x 8 grids (S1 / S2 / S3) and 32^3 (A / A2).  usage: python tools/bench_conv_brick.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import brick, lib as L, ops, synth
from dreg_nerf_amd.regtr import NeRFRegTr

dev = torch.device("cuda", 0)
lib = L.load()
batch = [synth.shell_pair(128, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose()) for i in range(4)]
idxs = [d[s + "_mask"].to(dev) for d in batch for s in ("src", "tgt")]
res = (128, 128, 128)
B = len(idxs)
idx_cat = torch.cat(idxs)
pb = torch.repeat_interleave(torch.arange(B, dtype=torch.int32, device=dev), torch.tensor([len(i) for i in idxs], device=dev))
t0 = time.perf_counter()
rows = ops.active_sets(idxs, res, (64, 64, 64), dev, pt_batch=pb, idx_cat=idx_cat, brick_tiles=True)
torch.cuda.synchronize()
print(f"active sets + tile tables: {1e3 * (time.perf_counter() - t0):.1f} ms (first call)")
def timed_sets(bt):
    t0 = time.perf_counter()
    for _ in range(5): ops.active_sets(idxs, res, (64, 64, 64), dev, pt_batch=pb, idx_cat=idx_cat, brick_tiles=bt)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / 5
print(f"active sets per call: {timed_sets(False):.2f} ms without / {timed_sets(True):.2f} ms with tile tables (host wall, synchronised)")


def ev(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cases = [("fwd lat1 64->256 on S2", 1, 64, 256, False, 64), ("fwd p1 256->256 on S1", 0, 256, 256, False, 64), ("dgrad p1 256->256 on S2", 1, 256, 256, True, 64),
         ("dgrad lat1 256->64 on S3", 2, 64, 256, True, 64), ("fwd p2 256->256 on A", 3, 256, 256, False, 32), ("dgrad p2 256->256 on A2", 4, 256, 256, True, 32)]
for name, si, cin, cout, tr, D in cases:
    ri = si if si < 3 else si + 1
    rl = rows[ri]
    n = int(rl.shape[0])
    bt = rows.tiles[ri]
    w = (torch.randn(cout, cin, 3, 3, 3) * 0.02).to(dev)
    c_in, c_out = (cout, cin) if tr else (cin, cout)
    x = torch.randn(B, D, D, D, c_in, device=dev).to(torch.bfloat16)
    if tr:   # a data gradient is zero away from the set
        m = torch.zeros(B * D ** 3, 1, device=dev, dtype=torch.bfloat16); m[rows[max(si - 1, 0)].long() if si < 3 else rows[4].long()] = 1
        x = (x.view(-1, c_in) * m).view(B, D, D, D, c_in)
    out = torch.zeros(B, D, D, D, c_out, device=dev, dtype=torch.bfloat16)
    wb = brick.pack_weight(w, tr)
    wp = ops.packed_weight(w, cin, tr, L.DT_BF16)
    ms1 = ev(lambda: brick.conv(x, wb, out, None, None, bt, c_in, c_out))
    ms2 = ev(lambda: L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wp), L.ptr(out), None, None, L.ptr(rl), n, B, D, D, D, c_in, D, D, D, c_out, 3, 1, 1, int(tr), 0,
                                                        0, 0, 0, 0, 0, L.stream()), "rows"))
    fl_ = 2.0 * n * 27 * cin * cout
    nh = bt.tiles[:bt.ntiles, 2].float()
    print(f"{name:28s} rows {n:7d} tiles {bt.ntiles:5d} staged/row {float(nh.sum()) / n:5.2f} (max {int(nh.max())}) | brick {ms1 * 1e3:7.1f} us {fl_ / ms1 / 1e9:7.1f} TF/s"
          f" = {fl_ / ms1 / 1e9 / 2500:.2f} | row-list {ms2 * 1e3:7.1f} us {fl_ / ms2 / 1e9:7.1f} TF/s = {fl_ / ms2 / 1e9 / 2500:.2f}")

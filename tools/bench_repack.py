"""Weight repack (fp32 master weights -> bf16 kernel packs) of the trunk executor: the batched launch and every pack record on
its own.  usage: python tools/bench_repack.py [--res 128] [--pairs 4]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=128)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--top", type=int, default=25)
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
model = NeRFRegTr(precision="bf16").to(dev).train()
ts = TrainStep(model)
batch = []
for i in range(args.pairs):
    d = synth.shell_pair(args.res, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
ts.step(batch)
torch.cuda.synchronize()
ex = next(iter(model._trunk_cache.values()))
lib = L.load()


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


full = timeit(lambda: L.check(lib.dreg_exec_repack(ex.h, L.ptr(ex.pack_table), L.ptr(ex.pack_rowmap), L.stream()), "repack"))
host = ex.pack_table.cpu().numpy().copy()
nrec = host.shape[0]
row0 = list(host[:, 11]) + [lib.dreg_exec_pack_rows(ex.h)]
print(f"dreg_exec_repack: {full:.1f} us for {nrec} records, {row0[-1]} rows (includes the halo packs)")
res = []
stage = 64 * 125
for i in range(nrec):
    one = host[i:i + 1].copy(); one[0, 11] = 0
    t = torch.from_numpy(one).to(dev)
    rows = int(row0[i + 1] - row0[i])
    us = timeit(lambda: lib.dreg_pack_conv_weights_batched(L.ptr(t), 1, rows, stage, None, L.stream()), 10)
    cout, cin_real, inner, ntaps, fd, kpad = (int(v) for v in host[i, 4:10])
    res.append((us, cout, cin_real, ntaps, fd, kpad, rows))
tot = sum(r[0] for r in res)
print(f"sum of single-record launches: {tot:.1f} us")
for us, cout, cin, ntaps, fd, kpad, rows in sorted(res, reverse=True)[:args.top]:
    mb = cout * cin * ntaps * 6 / 1e6
    print(f"  {us:7.1f} us  Cout {cout:5d} Cin {cin:5d} taps {ntaps:3d} for_dgrad {fd} Kpad {kpad:6d} rows {rows:6d}  {mb:7.1f} MB -> {mb / us * 1e-3:6.2f} TB/s")
by = {}
for us, cout, cin, ntaps, fd, kpad, rows in res:
    k = (ntaps, fd); a = by.setdefault(k, [0.0, 0.0]); a[0] += us; a[1] += cout * cin * ntaps * 6 / 1e6
for k, (us, mb) in sorted(by.items()):
    print(f"taps {k[0]:3d} for_dgrad {k[1]}: {us:8.1f} us {mb:8.1f} MB")
from dreg_nerf_amd import ops
n_live = sum(1 for e in ops._pack_cache.values() if e[0]() is not None)
print(f"ops.repack_all (python-level pack cache, {n_live} live packs): {timeit(lambda: ops.repack_all(dev)):.1f} us")

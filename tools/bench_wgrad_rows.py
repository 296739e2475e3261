#!/usr/bin/env python3
"""Row-list (active-set) weight gradient: the K loop with a voxel decode and an LDS read per load (dreg_conv_set_wgrad_rows_fast(0)) against
row + packed coordinates read once per stage and the 8-wave tile (1, default), on the head's layers with a shell-shaped row list.  Interleaved rounds; the two
must agree bit for bit where the tile shape is the same (Cin = 64) and to fp32 summation order where it is not.
usage: python tools/bench_wgrad_rows.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
lib = L.use_probe()


def shell_rows(D, r0, r1):
    z, y, x = torch.meshgrid(*[torch.arange(D, dtype=torch.float32)] * 3, indexing="ij")
    c = (D - 1) / 2
    r = ((z - c) ** 2 + (y - c) ** 2 + (x - c) ** 2).sqrt() / c
    m = ((r >= r0) & (r <= r1)).flatten().nonzero().flatten()
    return torch.cat([m + b * D ** 3 for b in range(B)]).int()


for (D, cin, cout, r0, r1) in ((64, 256, 256, 0.62, 0.70), (64, 64, 256, 0.55, 0.78), (32, 256, 256, 0.5, 0.85)):
    g = torch.Generator().manual_seed(0)
    rows = shell_rows(D, r0, r1).to(dev)
    x = torch.randn(B, D, D, D, cin, generator=g).to(dev).bfloat16()
    gy = torch.randn(B, D, D, D, cout, generator=g).to(dev).bfloat16()
    n = rows.shape[0]
    flops = 2.0 * n * cout * cin * 27
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, D, D, D, cin, cout, 3, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def run(mode):
        lib.dreg_conv_set_wgrad_rows_fast(mode & 1)

        dw = torch.empty(cout, cin, 3, 3, 3, dtype=torch.float32, device=dev)
        L.check(lib.dreg_conv3d_wgrad_rows(L.ptr(gy), L.ptr(x), L.ptr(dw), L.ptr(ws), nbytes, L.ptr(rows), n, B, D, D, D, cin, cin, D, D, D, cout, 3, 1, 1, 0,
                                           L.stream()), "dreg_conv3d_wgrad_rows")
        return dw

    out = {f: run(f) for f in (0, 1)}
    torch.cuda.synchronize()
    same = torch.equal(out[0], out[1])
    rel = ((out[0] - out[1]).abs().max() / out[0].abs().max()).item()
    ts = {0: [], 1: []}
    for r in range(5):
        for f in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(f)
            e1.record()
            torch.cuda.synchronize()
            ts[f].append(e0.elapsed_time(e1) / 3)
    lib.dreg_conv_set_wgrad_rows_fast(1)
    m = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    v0 = lib.dreg_conv3d_wgrad_variant(B, D, D, D, cin, cout, 3, 1, n, 0)
    print(f"B{B} {D}^3 {cin}->{cout} rows {n}: decode per load {m[0]:.3f} ms {flops / m[0] / 1e9:.0f} TF | packed coordinates, tile {v0 // 1000}x{v0 % 1000} "
          f"{m[1]:.3f} ms {flops / m[1] / 1e9:.0f} TF (both incl. the split sum) | bit-identical {same}, max rel diff {rel:.2e}", flush=True)

#!/usr/bin/env python3
"""Is the halo-tile convolution's speed data dependent?  The same launch (256 -> 256 at 64^3 x 8 grids, 3.71 TFLOP... per direction) on
dense random activations, on a tensor that is zero except for 4 % of its voxels (what a data gradient of the active-set step looks
like) and on all zeros.  A power-limited kernel clocks higher when the MFMA operands toggle less (bench.py's roofline.by_shape shows the
step's data-gradient launches at 0.71-0.73 of the roof and its forward launches of the same shapes at 0.52-0.55).
usage: python tools/halo_data_power.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import lib as L
dev = "cuda:0"
lib = L.load()
B, R, cin = 8, 64, 256
g = torch.Generator().manual_seed(1)
w = (torch.randn(256, cin, 3, 3, 3, generator=g) * 0.017).to(dev)
bias = torch.randn(256, generator=g).to(dev)
pk = torch.empty(lib.dreg_conv3_halo_pack_bytes(cin) // 2, dtype=torch.bfloat16, device=dev)
L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w), L.ptr(pk), 256, cin, 0, L.stream()), "pack")
out = torch.empty(B, R, R, R, 256, dtype=torch.bfloat16, device=dev)
flops = 2.0 * B * R ** 3 * 256 * 27 * cin
dense = torch.randn(B, R, R, R, cin, generator=g).to(dev).to(torch.bfloat16)
keep = (torch.rand(B, R, R, R, 1, generator=g) < 0.04).to(dev)
arms = {"dense random activations": dense, "4 % of the voxels non-zero": dense * keep, "all zeros": torch.zeros_like(dense)}
res = {k: [] for k in arms}
for rnd in range(3):
    for name, x in arms.items():
        L.check(lib.dreg_conv3_halo(L.ptr(x), L.ptr(pk), L.ptr(out), L.ptr(bias), None, B, R, R, R, cin, 0, 0, 0, 0, 0, L.stream()), "halo")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(lib.dreg_conv3_halo(L.ptr(x), L.ptr(pk), L.ptr(out), L.ptr(bias), None, B, R, R, R, cin, 0, 0, 0, 0, 0, L.stream()), "halo")
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
for name, v in res.items():
    m = sorted(v)[1]
    print(f"{name:32s} {m:.3f} ms  {flops / m / 1e9:7.0f} TFLOP/s  {flops / m / 1e9 / 2500:.3f} of the bf16 MFMA nameplate")

#!/usr/bin/env python3
"""Weight-gradient kernels on the dominant dense layers: 4-wave 128x128 tile vs 8-wave 256x256 tile; interleaved rounds, random data."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
lib = L.load()
for (D, cin, cout) in ((64, 256, 256), (32, 256, 256), (64, 64, 256)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, D, D, D, cin, generator=g).to(dev).bfloat16()
    gy = torch.randn(B, D, D, D, cout, generator=g).to(dev).bfloat16()
    flops = 2.0 * B * D ** 3 * cout * cin * 27
    out = {}
    for big in (0, 3, 1):
        lib.dreg_conv_set_wgrad_big(big)
        out[big] = ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
    torch.cuda.synchronize()
    d = max((out[0] - out[b]).abs().max().item() for b in (1, 3)) / out[0].abs().max().item()
    ts = {0: [], 1: [], 3: []}
    for r in range(5):
        for big in (0, 3, 1):
            lib.dreg_conv_set_wgrad_big(big)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
            e1.record()
            torch.cuda.synchronize()
            ts[big].append(e0.elapsed_time(e1) / 3)
    lib.dreg_conv_set_wgrad_big(3)
    m = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print(f"B{B} {D}^3 {cin}->{cout}: 128x128 {m[0]:.3f} ms {flops / m[0] / 1e9:.0f} TF | 256x256 8 waves {m[3]:.3f} ms {flops / m[3] / 1e9:.0f} TF | 256x128 4 waves, 32-voxel stages {m[1]:.3f} ms {flops / m[1] / 1e9:.0f} TF | rel diff {d:.2e}", flush=True)

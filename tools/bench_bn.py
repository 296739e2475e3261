#!/usr/bin/env python3
"""BatchNorm forward+backward per ResNet level: fused small-volume kernels vs the three-kernel form (interleaved rounds)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, ops

dev = "cuda:0"
lib = L.use_probe()
for (V3, C) in ((16, 128), (16, 512), (8, 256), (8, 1024), (4, 512), (4, 2048), (32, 64), (32, 256)):
    B = 8
    x = torch.randn(B, V3, V3, V3, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    g, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gy = torch.randn(B, V3, V3, V3, C, device=dev).to(torch.bfloat16)
    res = {}
    for mode, mv in (("three", 0), ("fused", 4096)):
        lib.dreg_bn_set_small_max_voxels(mv)
        for _ in range(3):
            y = ops.batchnorm(x, g, b, rm, rv, relu=True, train=True); y.backward(gy)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            for _ in range(20):
                y = ops.batchnorm(x, g, b, rm, rv, relu=True, train=True)
            e1.record()
            for _ in range(20):
                y.backward(gy, retain_graph=True)
            e2.record()
            torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1) / 20 * 1e3, e1.elapsed_time(e2) / 20 * 1e3))
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    lib.dreg_bn_set_small_max_voxels(4096)
    print(f"{V3}^3 x {C:5d}: three fwd {res['three'][0]:6.1f} us bwd {res['three'][1]:6.1f} us | fused fwd {res['fused'][0]:6.1f} us bwd {res['fused'][1]:6.1f} us")

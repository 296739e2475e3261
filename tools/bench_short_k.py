#!/usr/bin/env python3
"""Short-K launches (the point-set half's linear layers, the 1^3 convolutions of the deep ResNet levels): LDS ring depth 2 (default) vs 3 / 4
forced (dreg_conv_set_glds_stages) and the 128 x 64 narrow tiles on / off, raw C calls back to back (HIP events over 200 launches).
usage: python tools/bench_short_k.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
dev = "cuda"
lib = L.use_probe()
shapes = [("linear 9752 x 256 -> 768", 9752, 1, 256, 768), ("linear 9752 x 256 -> 256", 9752, 1, 256, 256), ("linear 9752 x 256 -> 1024", 9752, 1, 256, 1024),
          ("linear 9752 x 1024 -> 256", 9752, 1, 1024, 256), ("8^3 x 8 1024 -> 256", 8, 8, 1024, 256), ("8^3 x 8 256 -> 1024", 8, 8, 256, 1024),
          ("16^3 x 8 512 -> 128", 8, 16, 512, 128), ("16^3 x 8 128 -> 512", 8, 16, 128, 512), ("4^3 x 8 2048 -> 512", 8, 4, 2048, 512)]
for name, B, D, cin, cout in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.05
    wp = ops.packed_weight(w, cin, False, 0)
    out = torch.empty(B, D, D, D, cout, dtype=torch.bfloat16, device=dev)
    nws = lib.dreg_conv3d_igemm_workspace_bytes(B, D, D, D, cin, D, D, D, cout, 1, 1, 0, 0, 0, 0) if D ** 3 < 2048 else 0
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    st = L.stream()

    def launch():
        lib.dreg_conv3d_igemm_ws(L.ptr(x), L.ptr(wp), L.ptr(out), None, None, B, D, D, D, cin, D, D, D, cout, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, L.ptr(ws) if nws else None, nws, st)

    res = []
    ref = None
    for narrow in (2, 0):
        for stages in (0, 3, 4):
            lib.dreg_conv_set_narrow_small(narrow)
            lib.dreg_conv_set_glds_stages(stages)
            launch(); torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            assert torch.equal(out, ref)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200): launch()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 5)
            res.append(f"narrow{narrow} ns{stages or 2}: {sorted(ts)[1]:5.1f}")
    lib.dreg_conv_set_narrow_small(2); lib.dreg_conv_set_glds_stages(0)
    print(f"{name:28s} us/launch  " + "  ".join(res), flush=True)

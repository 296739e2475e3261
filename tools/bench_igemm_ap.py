#!/usr/bin/env python3
"""Eight-wave anti-phase form of the 128-row implicit-GEMM tiles (dreg_conv_set_igemm_ap) against the four-wave form on the shapes of
the under-filled launches of a training step (layer2-4 of the ResNet, the point-set half's linear layers): time per launch, bit-identity.
usage: python tools/bench_igemm_ap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
dev = "cuda"
lib = L.use_probe()
shapes = [("8^3 x 8  256 -> 256 k3", 8, 8, 256, 256, 3), ("8^3 x 8  1024 -> 256 k1", 8, 8, 1024, 256, 1), ("8^3 x 8  256 -> 1024 k1", 8, 8, 256, 1024, 1),
          ("4^3 x 8  512 -> 512 k3", 8, 4, 512, 512, 3), ("4^3 x 8  2048 -> 512 k1", 8, 4, 2048, 512, 1),
          ("16^3 x 8  128 -> 128 k3", 8, 16, 128, 128, 3), ("16^3 x 8  512 -> 128 k1", 8, 16, 512, 128, 1), ("16^3 x 8  128 -> 512 k1", 8, 16, 128, 512, 1),
          ("16^3 x 8  256 -> 256 k3", 8, 16, 256, 256, 3),
          ("32^3 x 8  64 -> 64 k3", 8, 32, 64, 64, 3), ("32^3 x 8  256 -> 64 k1", 8, 32, 256, 64, 1), ("32^3 x 8  64 -> 256 k1", 8, 32, 64, 256, 1),
          ("linear 9752 x 256 -> 768", 9752, 1, 256, 768, 1), ("linear 9752 x 256 -> 256", 9752, 1, 256, 256, 1), ("linear 9752 x 1024 -> 256", 9752, 1, 1024, 256, 1),
          ("linear 9752 x 256 -> 1024", 9752, 1, 256, 1024, 1)]
tot = [0.0, 0.0]
for name, B, D, cin, cout, k in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    wp = ops.packed_weight(w, cin, False, 0)
    out = torch.empty(B, D, D, D, cout, dtype=torch.bfloat16, device=dev)
    nb = lib.dreg_conv3d_igemm_workspace_bytes(B, D, D, D, cin, D, D, D, cout, k, 1, k // 2, 0, 0, 0)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=dev)

    def launch():
        L.check(lib.dreg_conv3d_igemm_ws(L.ptr(x), L.ptr(wp), L.ptr(out), None, None, B, D, D, D, cin, D, D, D, cout, k, 1, k // 2, 0, 0, 0, 0, 0, 0, 0, 0, L.ptr(ws), int(nb), L.stream()), "igemm")

    res, ms = {}, {}
    for ap in (0, 512):
        lib.dreg_conv_set_igemm_ap(ap)
        launch(); torch.cuda.synchronize()
        res[ap] = out.clone()
    for rnd in range(3):
        for ap in (0, 512):
            lib.dreg_conv_set_igemm_ap(ap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): launch()
            e1.record(); torch.cuda.synchronize()
            ms.setdefault(ap, []).append(e0.elapsed_time(e1) * 50)
    a, b = sorted(ms[0])[1], sorted(ms[512])[1]
    tot[0] += a; tot[1] += b
    fl = 2.0 * B * D ** 3 * cin * cout * k ** 3
    print(f"{name:28s} four waves {a:7.1f} us ({fl / a / 1e6:6.0f} TF)   anti-phase {b:7.1f} us ({fl / b / 1e6:6.0f} TF)   x{a / b:.2f}   bit-identical {torch.equal(res[0], res[512])}", flush=True)
lib.dreg_conv_set_igemm_ap(512)
print(f"sum {tot[0]:.1f} -> {tot[1]:.1f} us")

# ---- the 256 x 256 tile of large launches: anti-phase form over 32-channel stages (dreg_conv_set_igemm_ap256) vs the lockstep form
print("256 x 256 tile:")
big = [("64^3 x 8  256 -> 256 k3", 8, 64, 256, 256, 3), ("32^3 x 8  256 -> 256 k3", 8, 32, 256, 256, 3), ("64^3 x 8  64 -> 256 k3", 8, 64, 64, 256, 3),
       ("32^3 x 8  64 -> 256 k1", 8, 32, 64, 256, 1), ("32^3 x 8  512 -> 256 k1", 8, 32, 512, 256, 1)]
for name, B, D, cin, cout, k in big:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    wp = ops.packed_weight(w, cin, False, 0)
    out = torch.empty(B, D, D, D, cout, dtype=torch.bfloat16, device=dev)
    keep = torch.rand(B * D ** 3, device=dev) < 0.4
    keep[:5] = True; keep[-5:] = True
    rows = keep.nonzero().flatten().int()
    out_r = torch.zeros_like(out)

    def dense():
        L.check(lib.dreg_conv3d_igemm(L.ptr(x), L.ptr(wp), L.ptr(out), L.ptr(bias), None, B, D, D, D, cin, D, D, D, cout, k, 1, k // 2, 0, 1, 0, 0, 0, 0, 0, 0, L.stream()), "igemm")

    def sparse():
        L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wp), L.ptr(out_r), L.ptr(bias), None, L.ptr(rows), rows.shape[0], B, D, D, D, cin, D, D, D, cout, k, 1, k // 2, 0, 1, 0, 0, 0, 0, 0, L.stream()), "igemm_rows")

    res, ms = {}, {}
    for ap in (0, 1):
        lib.dreg_conv_set_igemm_ap256(ap)
        dense(); out_r.zero_(); sparse(); torch.cuda.synchronize()
        res[ap] = (out.clone(), out_r.clone())
    for rnd in range(3):
        for ap in (0, 1):
            lib.dreg_conv_set_igemm_ap256(ap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): dense()
            e1.record(); torch.cuda.synchronize()
            ms.setdefault(ap, []).append(e0.elapsed_time(e1) / 5)
    a, b = sorted(ms[0])[1], sorted(ms[1])[1]
    fl = 2.0 * B * D ** 3 * cin * cout * k ** 3
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    rows_ok = torch.equal(res[1][1].view(-1, cout)[rows.long()], res[1][0].view(-1, cout)[rows.long()])
    print(f"{name:28s} lockstep {a:7.3f} ms ({fl / a / 1e9:6.0f} TF)   anti-phase {b:7.3f} ms ({fl / b / 1e9:6.0f} TF)   x{a / b:.2f}   bit-identical {same}   row-list == dense rows {rows_ok}", flush=True)
lib.dreg_conv_set_igemm_ap256(1)

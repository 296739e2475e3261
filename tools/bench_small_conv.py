"""Micro-benchmark of the small-row-space convolutions (layer3 / layer4 of the ResNet): split-K on/off. usage: python tools/bench_small_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
dev = "cuda"
lib = L.use_probe()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 8
shapes = [(32, 256, 256, 3), (32, 64, 64, 3), (32, 64, 256, 1), (32, 256, 64, 1), (8, 256, 256, 3), (4, 512, 512, 3), (8, 1024, 256, 1), (8, 256, 1024, 1), (4, 2048, 512, 1), (4, 512, 2048, 1), (16, 128, 128, 3), (16, 512, 128, 1)]
for (D, cin, cout, k) in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    gy = torch.randn(B, D, D, D, cout, device=dev).bfloat16()
    wp = ops.packed_weight(w, cin, False, 0)
    flops = 2.0 * B * D ** 3 * cout * cin * k ** 3
    out = {}
    for mode in (5, 1, 12, 14):      # 5 = no split-K; 1 = default; 12 / 14 = default with 2 / 4 LDS stages forced
        lib.dreg_conv_set_glds(1 if mode > 5 else mode)
        lib.dreg_conv_set_glds_stages(mode - 10 if mode > 5 else 0)
        y = ops.conv_igemm(x, wp, None, None, (D, D, D), cin, cout, k, 1, k // 2, False)
        out[mode] = y.float()
        ms = timeit(lambda: ops.conv_igemm(x, wp, None, None, (D, D, D), cin, cout, k, 1, k // 2, False))
        print(f"B{B} {D}^3 {cin}->{cout} k{k} mode{mode}: fwd {ms*1e3:.1f} us {flops/ms/1e9:.0f} TF", end="   ")
    err = max((out[5] - out[1]).abs().max().item(), (out[12] - out[14]).abs().max().item()) / out[5].abs().max().item()
    assert torch.equal(out[12], out[14]) and torch.equal(out[1], out[14])
    lib.dreg_conv_set_glds(1); lib.dreg_conv_set_glds_stages(0)
    print(f"relerr {err:.1e}  wgrad:", end=" ")
    for f in (0, 1, 2, 4, 8, 16, 32):
        lib.dreg_conv_set_wgrad_splits(f)
        ms = timeit(lambda: ops.conv_wgrad(gy, x, (cout, cin, k, k, k), cin, k, 1, k // 2, True))
        print(f"s{lib.dreg_conv3d_wgrad_splits(B, D, D, D, cin, cout, k, 0)}={ms*1e3:.0f}us", end=" ")
    lib.dreg_conv_set_wgrad_splits(0)
    print(flush=True)

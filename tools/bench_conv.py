"""Micro-benchmark of the conv kernels on the dominant shapes (HIP events). usage: python tools/bench_conv.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
lib = L.use_probe()

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

shapes = [(64, 256, 256, 3), (64, 64, 256, 3), (32, 256, 256, 3), (32, 64, 64, 3), (32, 64, 256, 1), (16, 128, 128, 3)]
for (D, cin, cout, k) in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    gy = torch.randn(B, D, D, D, cout, device=dev).bfloat16()
    wp = ops.packed_weight(w, cin, False, 0)
    flops = 2.0 * B * D ** 3 * cout * cin * k ** 3
    res = {}
    for glds in (2, 3, 4):
        lib.dreg_conv_set_glds(glds)
        ms = timeit(lambda: ops.conv_igemm(x, wp, None, None, (D, D, D), cin, cout, k, 1, k // 2, False))
        res[f"fwd_glds{glds}"] = (ms, flops / ms / 1e9)
    lib.dreg_conv_set_glds(1)
    for tr in (True,):
        ms = timeit(lambda: ops.conv_wgrad(gy, x, (cout, cin, k, k, k), cin, k, 1, k // 2, tr))
        res["wgrad"] = (ms, flops / ms / 1e9)
    print(f"B{B} {D}^3 {cin}->{cout} k{k}: " + "  ".join(f"{n}: {m:.3f} ms {t:.0f} TF" for n, (m, t) in res.items()), flush=True)

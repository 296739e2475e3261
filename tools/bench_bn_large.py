"""The three-kernel BatchNorm of the large ResNet levels, kernel by kernel: HBM rate of the statistics passes and of the apply passes
(statistics pass = full - the run with that pass left out, dreg_bn_set_debug_skip).  usage: python tools/bench_bn_large.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L
dev = torch.device("cuda:0"); lib = L.use_probe()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
for V3, C, with_res in ((64, 64, False), (32, 64, False), (32, 256, True), (32, 256, False), (32, 128, False), (16, 128, False), (16, 512, True), (16, 256, False)):
    V = V3 ** 3
    n = B * V * C
    x = torch.randn(n, device=dev).bfloat16(); dy = torch.randn(n, device=dev).bfloat16(); y = torch.empty_like(x); dx = torch.empty_like(x)
    res = torch.randn(n, device=dev).bfloat16() if with_res else None
    dres = torch.empty_like(x) if with_res else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev); rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ss, mr, coef = (torch.zeros(B * C * 2, device=dev) for _ in range(3))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(B * lib.dreg_bn_num_chunks(V) * C * 2, device=dev)
    fwd = lambda: L.check(lib.dreg_bn3d_fwd(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(ss), L.ptr(mr), L.ptr(ws), B, V, C, 1e-5, 0.1, 1, 1, 0, L.stream()), "fwd")
    bwd = lambda: L.check(lib.dreg_bn3d_bwd(L.ptr(x), L.ptr(dy), L.ptr(y) if with_res else None, L.ptr(ss), L.ptr(mr), L.ptr(dx), L.ptr(dres), L.ptr(dg), L.ptr(db), L.ptr(coef), L.ptr(ws), B, V, C, 1, 0, 0, L.stream()), "bwd")
    lib.dreg_bn_set_debug_skip(0); f_full, b_full = timeit(fwd), timeit(bwd)
    lib.dreg_bn_set_debug_skip(3); f_ap, b_ap = timeit(fwd), timeit(bwd)
    lib.dreg_bn_set_debug_skip(0)
    # the same backward as two half-batch calls (statistics are per grid): x + dy of one half (n bytes each... 2 x n bytes total) may stay in
    # the 256 MB memory-side cache between the statistics pass and the apply pass
    h = n // 2
    hc = (B // 2) * C * 2
    hw = ws.numel() // 2
    def bwd_halves():
        for k in (0, 1):
            sl = slice(k * h, (k + 1) * h)
            L.check(lib.dreg_bn3d_bwd(L.ptr(x[sl]), L.ptr(dy[sl]), L.ptr(y[sl]) if with_res else None, L.ptr(ss[k * hc:]), L.ptr(mr[k * hc:]), L.ptr(dx[sl]),
                                      L.ptr(dres[sl]) if with_res else None, L.ptr(dg), L.ptr(db), L.ptr(coef[k * hc:]), L.ptr(ws[k * hw:]), B // 2, V, C, 1, 0, 0, L.stream()), "bwd")
    def fwd_halves():
        for k in (0, 1):
            sl = slice(k * h, (k + 1) * h)
            L.check(lib.dreg_bn3d_fwd(L.ptr(x[sl]), L.ptr(res[sl]) if with_res else None, L.ptr(y[sl]), L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(ss[k * hc:]), L.ptr(mr[k * hc:]),
                                      L.ptr(ws[k * hw:]), B // 2, V, C, 1e-5, 0.1, 1, 1, 0, L.stream()), "fwd")
    f_half, b_half = timeit(fwd_halves), timeit(bwd_halves)
    by = 2.0 * n
    r = 1 if with_res else 0
    print(f"{V3}^3 x {C:4d}{' +res' if with_res else '     '}: fwd stats {f_full - f_ap:6.1f} us ({by / (f_full - f_ap) * 1e-6:5.2f} TB/s)  finalize+apply {f_ap:6.1f} us ({(2 + r) * by / f_ap * 1e-6:5.2f} TB/s)"
          f" | bwd stats {b_full - b_ap:6.1f} us ({(2 + r) * by / (b_full - b_ap) * 1e-6:5.2f} TB/s)  finalize+apply {b_ap:6.1f} us ({(3 + 2 * r) * by / b_ap * 1e-6:5.2f} TB/s)"
          f" | whole fwd {f_full:6.1f} -> two half-batch calls {f_half:6.1f} us, whole bwd {b_full:6.1f} -> {b_half:6.1f} us", flush=True)

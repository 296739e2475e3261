#!/usr/bin/env python3
"""Microbenchmark of the 64-output-channel halo kernel (csrc/conv_halo.hip, conv3_halo64_kernel) against the implicit GEMM it replaces on
conv2 of layer1's bottlenecks: 64 -> 64 at 32^3 x 8 grids (58 GFLOP per launch), forward and accumulating data gradient.
Interleaved rounds in one process, random data.   python tools/bench_conv_halo64.py [B R cin]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, ops  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, R, cin = (int(a) for a in (pos + ["8", "32", "64"][len(pos):]))
    dev = "cuda:0"
    lib = L.use_probe()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, R, R, R, cin, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(64, cin, 3, 3, 3, generator=g) * 0.034).to(dev)
    pk = torch.empty(lib.dreg_conv3_halo_pack_bytes_n(cin, 64) // 2, dtype=torch.bfloat16, device=dev)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w), L.ptr(pk), 64, cin, 0, L.stream()), "pack")
    out = torch.empty(B, R, R, R, 64, dtype=torch.bfloat16, device=dev)
    acc = torch.randn(B, R, R, R, 64, generator=g).to(dev).to(torch.bfloat16)
    flops = 2.0 * B * R ** 3 * 64 * 27 * cin

    def halo(mode, add=None):
        lib.dreg_conv3_halo64_set(mode)
        L.check(lib.dreg_conv3_halo_n(L.ptr(x), L.ptr(pk), L.ptr(out if add is None else add), None, L.ptr(add), B, R, R, R, cin, 64,
                                      R if add is not None else 0, R if add is not None else 0, R if add is not None else 0, 1, 0, L.stream()), "halo64")

    pk_ig = ops.packed_weight(w, cin, False, L.DT_BF16)
    out_ig = torch.empty_like(out)

    def igemm(add=None):
        L.check(lib.dreg_conv3d_igemm(L.ptr(x), L.ptr(pk_ig), L.ptr(out_ig if add is None else add), None, L.ptr(add), B, R, R, R, cin, R, R, R, 64, 3, 1, 1, 0, 0,
                                      R if add is not None else 0, R if add is not None else 0, R if add is not None else 0, 1, L.DT_BF16, 0, L.stream()), "igemm")

    acc2 = acc.clone()
    arms = {"igemm": igemm, "halo64": lambda: halo(1), "halo64_no_prefetch": lambda: halo(2),
            "igemm+acc": lambda: igemm(acc2), "halo64+acc": lambda: halo(1, acc)}
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    igemm()
    halo(1)
    d = (out.float() - out_ig.float()).abs()
    print(f"max |halo64 - igemm| = {d.max().item():.4g} (|ref| max {out_ig.float().abs().max().item():.3g}), differing elements {(d > 0).float().mean().item():.4f}")
    times = {k: [] for k in arms}
    for rnd in range(9):
        for k, f in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5)
    for k, v in times.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{k:22s} median {1e3 * med:8.1f} us  min {1e3 * v[0]:8.1f} us  {flops / med / 1e9:8.1f} TFLOP/s ({flops / med / 1e9 / 2500:.3f} of 2.5 PF)")
    lib.dreg_conv3_halo64_set(1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Microbenchmark of the halo-tile 3^3 convolution against the implicit-GEMM kernel on the dominant layer
(upsample_transform_1: 256 -> 256 at 64^3 x 8 grids = 7.42 TFLOP per launch).  Interleaved rounds in one process, random data."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, ops  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, R, cin = (int(a) for a in (pos + ["8", "64", "256"][len(pos):]))
    dev = "cuda:0"
    lib = L.use_probe()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, R, R, R, cin, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(256, cin, 3, 3, 3, generator=g) * 0.017).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    pk = torch.empty(lib.dreg_conv3_halo_pack_bytes(cin) // 2, dtype=torch.bfloat16, device=dev)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w), L.ptr(pk), 256, cin, 0, L.stream()), "pack")
    out = torch.empty(B, R, R, R, 256, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * B * R ** 3 * 256 * 27 * cin

    def halo(v):
        lib.dreg_conv3_halo_set_variant(v)
        L.check(lib.dreg_conv3_halo(L.ptr(x), L.ptr(pk), L.ptr(out), L.ptr(bias), None, B, R, R, R, cin, 0, 0, 0, 0, 0, L.stream()), "halo")

    # the implicit GEMM through the C ABI itself (ops.conv3d would route this shape to the halo kernel — with whatever variant an ablation arm left set)
    pk_ig = ops.packed_weight(w, cin, False, L.DT_BF16)
    out_ig = torch.empty_like(out)

    def igemm():
        L.check(lib.dreg_conv3d_igemm(L.ptr(x), L.ptr(pk_ig), L.ptr(out_ig), L.ptr(bias), None, B, R, R, R, cin, R, R, R, 256, 3, 1, 1, 0, 0, 0, 0, 0, 0,
                                      L.DT_BF16, 0, L.stream()), "igemm")

    arms = {"igemm256": igemm, "halo": lambda: halo(0), "halo_lockstep": lambda: halo(1), "halo_nosplit": lambda: halo(3)}
    if "--only-halo" in sys.argv:
        arms = {"halo": lambda: halo(0), "igemm256": arms["igemm256"]}
    if "--ablate" in sys.argv:
        arms.update({"abl_noDMA": lambda: halo(11), "abl_noDSread": lambda: halo(12), "abl_noDMA_noDS": lambda: halo(13), "abl_noMFMA": lambda: halo(14), "abl_DMAonly": lambda: halo(16)})
    if "--prof" in sys.argv:
        buf = torch.zeros(64 * 8 * 5, dtype=torch.int64, device=dev)
        lib.dreg_conv3_halo_set_prof(L.ptr(buf))
        halo(5)
        torch.cuda.synchronize()
        lib.dreg_conv3_halo_set_prof(None)
        p = buf.view(64, 8, 5).double().cpu()
        nu = (cin // 32) * 27
        names = ["issue(reads+DMA)", "vmcnt wait", "barrier1+lgkm", "MFMA half", "barrier2"]
        for grp, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
            m = p[:, sl].mean(dim=(0, 1)) / nu
            print(grp, " ".join(f"{n}={v:.0f}" for n, v in zip(names, m.tolist())), f"sum={m.sum():.0f} cycles/unit (s_memtime ticks)")
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    igemm()
    ref = out_ig.float()
    halo(0)
    d = (out.float() - ref).abs()
    print(f"max |halo - igemm| = {d.max().item():.4g} (|ref| max {ref.abs().max().item():.3g}), differing elements {(d > 0).float().mean().item():.4f}")
    times = {k: [] for k in arms}
    for rnd in range(7):
        for k, f in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 3)
    for k, v in times.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{k:18s} median {med:8.3f} ms  min {v[0]:8.3f} ms  {flops / med / 1e9:8.1f} TFLOP/s ({flops / med / 1e9 / 2500:.3f} of 2.5 PF)")


if __name__ == "__main__":
    main()

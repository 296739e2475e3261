#!/usr/bin/env python3
"""Where a wave of the 128 x 128 implicit-GEMM kernel spends a K step: s_memtime stamps around (1) the wait for the stage's direct-to-LDS
loads, (2) the workgroup barrier, (3) issuing the next stage, (4) fragment reads + MFMAs (dreg_conv_igemm_probe: an instrumented copy
of the kernel, measurement only; the stamps themselves cost ~10 %).  Cycles per K step and wave, averaged over all waves of 20 launches.
usage: python tools/igemm_phase_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
dev = "cuda"
lib = L.use_probe()
shapes = [("16^3 x 8  128 -> 128 k3", 8, 16, 128, 128, 3), ("16^3 x 8  512 -> 128 k1", 8, 16, 512, 128, 1), ("32^3 x 8  256 -> 128 k1", 8, 32, 256, 128, 1),
          ("32^3 x 8  128 -> 128 k3", 8, 32, 128, 128, 3), ("linear 9752 x 256 -> 768", 9752, 1, 256, 768, 1), ("linear 9752 x 1024 -> 256", 9752, 1, 1024, 256, 1)]
buf = (ctypes.c_ulonglong * 8)()
for name, B, D, cin, cout, k in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    wp = ops.packed_weight(w, cin, False, 0)
    out = torch.empty(B, D, D, D, cout, dtype=torch.bfloat16, device=dev)

    def launch():
        lib.dreg_conv3d_igemm_ws(L.ptr(x), L.ptr(wp), L.ptr(out), None, None, B, D, D, D, cin, D, D, D, cout, k, 1, k // 2, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, L.stream())

    lib.dreg_conv_igemm_probe(0)
    launch(); torch.cuda.synchronize()
    ref = out.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): launch()
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) * 50
    lib.dreg_conv_igemm_probe(1)
    launch(); torch.cuda.synchronize()
    assert torch.equal(out, ref)
    lib.dreg_conv_igemm_probe_read(ctypes.cast(buf, ctypes.c_void_p))
    e0.record()
    for _ in range(20): launch()
    e1.record(); torch.cuda.synchronize()
    probed = e0.elapsed_time(e1) * 50
    lib.dreg_conv_igemm_probe_read(ctypes.cast(buf, ctypes.c_void_p))
    lib.dreg_conv_igemm_probe(0)
    wait, bar, issue, comp, ksteps, waves = [int(buf[i]) for i in range(6)]
    tiles = ((B * D ** 3 + 127) // 128) * (cout // 128)
    print(f"{name:28s} {tiles:5d} workgroups, {ksteps // max(waves, 1):3d} K steps: per K step and wave  wait for loads {wait / ksteps:6.0f}  barrier {bar / ksteps:6.0f}  "
          f"issue {issue / ksteps:5.0f}  reads + MFMA {comp / ksteps:6.0f}  (launch {plain:.1f} us; instrumented {probed:.1f} us: the stamps and the final atomics)", flush=True)

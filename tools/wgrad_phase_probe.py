#!/usr/bin/env python3
"""Where a wave of the anti-phase 256 x 256 weight-gradient kernel spends a 32-voxel unit: s_memtime stamps (ring mode 4, an instrumented
copy of the kernel, measurement only) around (0) issuing the direct-to-LDS pieces of the unit three ahead, (1) the 24 transposed fragment
reads until they have returned, (2) the counted wait for the next unit's pieces, (3) the barrier that ends the load half, (4) the 32 MFMAs,
(5) the barrier that ends the MFMA half.  Cycles per unit and wave, averaged over all waves.
usage: python tools/wgrad_phase_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
dev = "cuda"
lib = L.use_probe()
buf = (ctypes.c_ulonglong * 8)()
for (B, D, cin, cout) in ((8, 64, 256, 256),):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, D, D, D, cin, generator=g).to(dev).bfloat16()
    gy = torch.randn(B, D, D, D, cout, generator=g).to(dev).bfloat16()
    flops = 2.0 * B * D ** 3 * cout * cin * 27

    def run(mode, n=3):
        lib.dreg_conv_set_wgrad_big(3); lib.dreg_conv_set_wgrad_ring(mode)
        ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    plain = run(3)
    names = ("24 fragment reads (until returned)", "issue 4 pieces", "wait next unit", "barrier (load half)", "32 MFMAs", "barrier (MFMA half)")
    for mode, what in ((4, "full"), (5, "WITHOUT fragment reads"), (6, "WITHOUT direct-to-LDS pieces"), (7, "WITHOUT MFMAs")):
        lib.dreg_conv_wgrad_probe_read(ctypes.cast(buf, ctypes.c_void_p))
        probed = run(mode)
        lib.dreg_conv_wgrad_probe_read(ctypes.cast(buf, ctypes.c_void_p))
        v = [int(buf[i]) for i in range(8)]
        units = max(v[6], 1)
        print(f"B{B} {D}^3 {cin}->{cout} [{what}]: plain {plain:.3f} ms = {flops / plain / 1e9:.0f} TF, instrumented {probed:.3f} ms; per unit and wave: " +
              "  ".join(f"{n} {v[i] / units:.0f}" for i, n in enumerate(names)) + f"  | sum {sum(v[:6]) / units:.0f} cycles, {v[7]} waves x {units // max(v[7], 1)} units", flush=True)
    lib.dreg_conv_set_wgrad_ring(3)

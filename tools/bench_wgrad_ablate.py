"""What bounds the 8-wave 256 x 256 weight-gradient kernel: the full kernel against versions with one phase removed (wrong results)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, lib as L
lib = L.use_probe(); dev = "cuda"
B, D, cin, cout = 8, 64, 256, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, D, D, D, cin, generator=g).to(dev).bfloat16(); gy = torch.randn(B, D, D, D, cout, generator=g).to(dev).bfloat16()
flops = 2.0 * B * D ** 3 * cout * cin * 27
names = {3: "8-wave kernel", 11: "  without the MFMAs", 12: "  without the LDS fragment reads", 13: "  without the direct-to-LDS loads"}
for big in (3, 11, 12, 13):
    lib.dreg_conv_set_wgrad_big(big)
    for _ in range(2): ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv_wgrad(gy, x, (cout, cin, 3, 3, 3), cin, 3, 1, 1, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{names[big]:40s} {ms:7.3f} ms  ({flops / ms / 1e9:5.0f} TFLOP/s equivalent)")
lib.dreg_conv_set_wgrad_big(3)

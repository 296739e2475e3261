import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops, lib as L
lib = L.load(); dev = "cuda"
shapes = [(8, 8, 1024, 512, 1), (8, 8, 1024, 256, 1), (8, 8, 256, 1024, 1), (8, 16, 128, 512, 1), (8, 16, 512, 128, 1), (8, 8, 512, 2048, 1), (8, 4, 2048, 512, 1), (8,4,512,2048,1), (8, 8, 256, 256, 3), (8, 16, 256, 256, 3)]
for B, D, cin, cout, k in shapes:
    x = torch.randn(B, D, D, D, cin, device=dev).bfloat16()
    gy = torch.randn(B, D, D, D, cout, device=dev).bfloat16()
    line = f"{D}^3x{B} {cin}->{cout} k{k}: auto splits {lib.dreg_conv3d_wgrad_splits(B, D, D, D, cin, cout, k, 0)} variant {lib.dreg_conv3d_wgrad_variant(B, D, D, D, cin, cout, k, 0, 0, 0)} |"
    for sp in (0, 1, 2, 4, 8, 16, 32):
        lib.dreg_conv_set_wgrad_splits(sp)
        try:
            ops.conv_wgrad(gy, x, (cout, cin, k, k, k), cin, k, 1, k // 2); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv_wgrad(gy, x, (cout, cin, k, k, k), cin, k, 1, k // 2)
            e1.record(); torch.cuda.synchronize()
            line += f" s{sp}: {e0.elapsed_time(e1) * 50:.1f}us"
        except Exception as ex:
            line += f" s{sp}: ERR"
    lib.dreg_conv_set_wgrad_splits(0)
    print(line, flush=True)

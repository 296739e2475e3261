"""Run the dominant conv forward + weight-gradient a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreg_nerf_amd import ops
B, D, cin, cout, k = 8, 64, 256, 256, 3
x = torch.randn(B, D, D, D, cin, device="cuda").bfloat16()
gy = torch.randn(B, D, D, D, cout, device="cuda").bfloat16()
w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.05
wp = ops.packed_weight(w, cin, False, 0)
for _ in range(3):
    ops.conv_igemm(x, wp, None, None, (D, D, D), cin, cout, k, 1, 1, False)
    ops.conv_wgrad(gy, x, (cout, cin, k, k, k), cin, k, 1, 1, True)
torch.cuda.synchronize()

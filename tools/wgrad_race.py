"""Does the weight-gradient path give bit-identical results when other kernels share the GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, lib as L
dev = torch.device("cuda", 0)
g0 = torch.Generator().manual_seed(1)
side = torch.cuda.Stream()
big = torch.randn(4096, 4096, device=dev)
def stress(n=6):
    for _ in range(n): torch.mm(big, big)
for (R, cin, cout) in [(5944, 256, 768), (5944, 256, 256), (5944, 1024, 256), (5944, 256, 1024), (35664, 256, 256)]:
    x = torch.randn(1, 1, 1, R, cin, generator=g0).to(dev).bfloat16()
    gy = torch.randn(1, 1, 1, R, cout, generator=g0).to(dev).bfloat16()
    ref = ops.conv_wgrad(gy, x, (cout, cin), cin, 1, 1, 0, True).clone()
    refb = ops.colsum(gy.view(-1, cout)).clone()
    torch.cuda.synchronize()
    bad = badb = 0
    for it in range(60):
        stress()
        with torch.cuda.stream(side):
            dw = ops.conv_wgrad(gy, x, (cout, cin), cin, 1, 1, 0, True)
            db = ops.colsum(gy.view(-1, cout))
        stress()
        torch.cuda.synchronize()
        bad += not torch.equal(dw, ref); badb += not torch.equal(db, refb)
    # accumulate form
    acc_bad = 0
    for it in range(30):
        sink = torch.zeros(cout, cin, device=dev)
        stress()
        with torch.cuda.stream(side):
            ops.conv_wgrad(gy, x, (cout, cin), cin, 1, 1, 0, True, accumulate_into=sink)
        stress()
        torch.cuda.synchronize()
        acc_bad += not torch.equal(sink, ref)
    print(f"R={R} {cin}->{cout}: wgrad mismatches {bad}/60, colsum {badb}/60, accumulate-form {acc_bad}/30", flush=True)

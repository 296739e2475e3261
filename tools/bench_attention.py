#!/usr/bin/env python3
"""The point-set transformer's attention launches at the bench's size: 8 point sets of ~1,220 points (4 pairs), 8 heads x 32 channels,
self and cross problems in one launch (mha_varlen), forward and backward; and the 256-channel correspondence attention over the 6 stacked
layer outputs.  HIP-event time per launch, median of rounds.  usage: python tools/bench_attention.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import attn_ops as A

dev = "cuda"
A.set_precision("bf16")
g = torch.Generator().manual_seed(0)
segs = [(1219, 1219)] * 4
tab = A.ProblemTable(segs, dev)
R = tab.R


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / n)
    return sorted(ts)[2]


qkv = torch.randn(R, 768, generator=g).to(dev).bfloat16().requires_grad_(True)
for name, probs in (("self", tab.self_probs), ("cross", tab.cross_probs)):
    sc = 1.0 / math.sqrt(32)
    fwd = timed(lambda: A.mha_varlen(qkv.detach(), probs, tab.nprob, tab.max_len, 8, sc))
    o = A.mha_varlen(qkv, probs, tab.nprob, tab.max_len, 8, sc)
    go = torch.randn(o.shape, generator=g).to(dev).bfloat16()
    bwd = timed(lambda: torch.autograd.grad(o, qkv, go, retain_graph=True))
    fl = 4.0 * sum(ns * ns + nt * nt for ns, nt in segs) * 256
    print(f"mha {name}: R {R}, 8 problems x 8 heads: fwd {fwd:.1f} us ({fl / fwd / 1e6:.0f} TF/s), bwd (dq + dkv + torch glue) {bwd:.1f} us ({2.5 * fl / bwd / 1e6:.0f} TF/s)")
q = torch.randn(6, R, 256, generator=g).to(dev).bfloat16().requires_grad_(True)
k = torch.randn(6, R, 256, generator=g).to(dev).bfloat16().requires_grad_(True)
xyz = torch.randn(R, 3, generator=g).to(dev)
sc = 1.0 / math.sqrt(256)
fwd = timed(lambda: A.attention_xyz_varlen(q.detach(), k.detach(), xyz, tab.cross_probs, tab.nprob, tab.max_len, sc), 10)
o = A.attention_xyz_varlen(q, k, xyz, tab.cross_probs, tab.nprob, tab.max_len, sc)
go = torch.randn(o.shape, generator=g).to(dev)
bwd = timed(lambda: torch.autograd.grad(o, (q, k), go, retain_graph=True), 10)
print(f"correspondence attention (6 layers x 8 problems, 256 channels): fwd {fwd:.1f} us, bwd {bwd:.1f} us")

"""Producer -> consumer on ONE stream with a cross-stream event between them: does the consumer ever see old data?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
aux = torch.cuda.Stream()
N = 1920 * 256
srcs = [torch.randn(N, device=dev) for _ in range(4)]
X = torch.empty(N, device=dev)
big = torch.randn(2048, 2048, device=dev)
bad = 0
outs = []
for it in range(400):
    s = srcs[it % 4]
    X.copy_(s)                       # producer (main)
    ev = torch.cuda.Event(); ev.record(); aux.wait_event(ev)
    with torch.cuda.stream(aux):
        torch.mm(big, big)           # unrelated work on the second stream
    Y = X * 1.0                      # consumer (main)
    outs.append((Y, it % 4))
    if it % 50 == 49:
        torch.cuda.synchronize()
        for Y_, k in outs: bad += not torch.equal(Y_, srcs[k])
        outs.clear()
print("torch copy->mul with cross-stream event in between: stale reads", bad, "/ 400")
# the same with the library's kernels: LayerNorm backward consuming a freshly produced gradient
from dreg_nerf_amd import lib as L
lib = L.load()
R = 1920
xl = torch.randn(R, 256, device=dev); gam = torch.randn(256, device=dev); stats = torch.rand(R, 2, device=dev) + 0.5
gsrc = [torch.randn(R, 256, device=dev).bfloat16() for _ in range(4)]
def ln_bwd(dy):
    dx = torch.empty_like(xl); dg = torch.empty(256, device=dev); db = torch.empty(256, device=dev)
    ws = torch.empty(lib.dreg_layernorm_bwd_workspace_bytes(R) // 4 + 4, device=dev)
    L.check(lib.dreg_layernorm_bwd(L.ptr(xl), L.ptr(dy), L.ptr(gam), L.ptr(stats), L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ws), R, 256, 0, 0, 0, L.stream()), "ln")
    return dx
refs = [ln_bwd(g).clone() for g in gsrc]
torch.cuda.synchronize()
GY = torch.empty(R, 256, device=dev, dtype=torch.bfloat16)
bad = 0; outs = []
for it in range(400):
    GY.copy_(gsrc[it % 4])
    ev = torch.cuda.Event(); ev.record(); aux.wait_event(ev)
    with torch.cuda.stream(aux):
        torch.mm(big, big)
    outs.append((ln_bwd(GY), it % 4))
    if it % 50 == 49:
        torch.cuda.synchronize()
        for d, k in outs: bad += not torch.equal(d, refs[k])
        outs.clear()
print("copy -> layernorm_bwd with cross-stream event in between: wrong results", bad, "/ 400")

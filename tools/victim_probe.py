"""Which kernels return wrong results while a conv kernel of the library runs on another stream?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, lib as L, attn_ops as A
dev = torch.device("cuda", 0)
lib = L.load()
g0 = torch.Generator().manual_seed(1)
side = torch.cuda.Stream()
R = 1920
x = torch.randn(1, 1, 1, R, 256, generator=g0).to(dev).bfloat16()
w = torch.randn(1024, 256, generator=g0).to(dev); gy = torch.randn(1, 1, 1, R, 1024, generator=g0).to(dev).bfloat16()
wpk = ops.packed_weight(w, 256, False, 0)
a = torch.randn(R, 256, generator=g0).to(dev)
gam, bet = torch.randn(256, generator=g0).to(dev), torch.randn(256, generator=g0).to(dev)
victims = {
    "torch.softmax": lambda: torch.softmax(a, dim=1),
    "torch.sum(dim=1)": lambda: a.sum(dim=1),
    "torch.layer_norm": lambda: torch.nn.functional.layer_norm(a, (256,), gam, bet),
    "dreg layernorm_fwd": lambda: A.layer_norm(a, gam, bet, None, torch.float32),
    "torch elementwise": lambda: a * 1.5 + 2.0,
    "torch.cumsum": lambda: torch.cumsum(a, dim=1),
    "torch add": lambda: a + a2,
    "torch mul+add (addcmul)": lambda: torch.addcmul(a, a2, a2),
    "torch cat": lambda: torch.cat([a, a2], dim=0),
    "torch bf16->f32 cast": lambda: ab.float(),
    "torch relu bwd": lambda: torch.where(a > 0, a2, 0.0),
    "torch mean/var": lambda: torch.var_mean(a, dim=1)[0],
    "torch big add": lambda: big + big2,
}
a2 = torch.randn(R, 256, generator=g0).to(dev); ab = a.bfloat16()
big = torch.randn(1 << 22, generator=g0).to(dev); big2 = torch.randn(1 << 22, generator=g0).to(dev)
cor = sys.argv[1] if len(sys.argv) > 1 else "fwd_glds"
def corun():
    if cor == "wgrad": ops.conv_wgrad(gy, x, (1024, 256), 256, 1, 1, 0, True)
    elif cor == "fwd_glds": ops.conv_igemm(x, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False)
    elif cor == "none": pass
for name, fn in victims.items():
    ref = fn().clone(); torch.cuda.synchronize()
    bad = tot = 0
    for rep in range(20):
        with torch.cuda.stream(side):
            for _ in range(24): corun()
        outs = [fn() for _ in range(64)]
        torch.cuda.synchronize()
        for o in outs:
            tot += 1; bad += not torch.equal(o, ref)
    print(f"co-running {cor}: {name:22s} wrong {bad}/{tot}", flush=True)

"""Which wait states / operand paths make the packed-fp32 co-execution fault go away?  LayerNorm-backward row body (SLP-vectorised:
v_pk_*_f32 present) in the variants of victim_nop.hip, each run 2,560 times next to an implicit-GEMM convolution on a second stream.
usage: python tools/hw_probe/run_nop.py [co-runner: fwd_glds | fwd_regstaged | none] [reps] [variants, comma separated]"""
import os, sys, ctypes, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dreg_nerf_amd import ops, lib as L
so = os.path.join(HERE, "libvictim_nop.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "victim_nop.hip")])
so2 = os.path.join(HERE, "libvictim_nop_noslp.so")     # the same source without the SLP vectoriser: no v_pk_*_f32 at all (variant 100)
if not os.path.exists(so2):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", so2, os.path.join(HERE, "victim_nop.hip")])
V = ctypes.CDLL(so); V.probe_ln_variant.restype = ctypes.c_int
V2 = ctypes.CDLL(so2); V2.probe_ln_variant.restype = ctypes.c_int
vp = ctypes.c_void_p
dev = torch.device("cuda", 0); lib = L.use_probe()
g0 = torch.Generator().manual_seed(1)
R = 1920
xl = torch.randn(R, 256, generator=g0).to(dev); dy = torch.randn(R, 256, generator=g0).to(dev).bfloat16()
gam = torch.randn(256, generator=g0).to(dev); stats = (torch.rand(R, 2, generator=g0) + 0.5).to(dev)
cx = torch.randn(1, 1, 1, R, 256, device=dev).bfloat16(); w = torch.randn(1024, 256, device=dev)
wpk = ops.packed_weight(w, 256, False, 0)
side = torch.cuda.Stream()
what = sys.argv[1] if len(sys.argv) > 1 else "fwd_glds"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nb = 64
dxs = [torch.zeros(R, 256, device=dev) for _ in range(nb)]
def corun():
    if what == "fwd_regstaged":
        lib.dreg_conv_set_glds(0); ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False); lib.dreg_conv_set_glds(1)
    elif what == "fwd_glds": ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False)
names = {0: "as compiled: v_pk_add_f32 right behind s_waitcnt vmcnt(0)", 1: "s_nop 7 behind the wait, all loaded registers", 2: "s_nop 0",
         3: "4 x s_nop 7", 4: "s_nop 7 on (mean, rstd) only: that load is waited for first", 5: "s_nop 7 on x only",
         6: "unpacked v_mov_b32 of every loaded register first", 7: "statistics through the scalar path (SGPR operands)"}
variants = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(8)) + [100]
names[100] = "variant 0's source built with -fno-slp-vectorize: unpacked v_sub_f32 right behind s_waitcnt vmcnt(0)"
per = 8                                     # launches per variant and repetition: the variants are INTERLEAVED inside every repetition,
bufs = {v: [torch.zeros(R, 256, device=dev) for _ in range(per)] for v in variants}    # the fault rate drifts by 100x between runs
def run(v, i): (V2 if v >= 100 else V).probe_ln_variant(v % 100, vp(xl.data_ptr()), vp(dy.data_ptr()), vp(gam.data_ptr()), vp(stats.data_ptr()), vp(bufs[v][i].data_ptr()), R, vp(L.stream()))
ref = {}
for v in variants:
    run(v, 0); torch.cuda.synchronize(); ref[v] = bufs[v][0].clone()
bad = {v: 0 for v in variants}; rows = {v: 0 for v in variants}; cols = {v: set() for v in variants}
for rep in range(reps):
    with torch.cuda.stream(side):
        for _ in range(24): corun()
    for i in range(per):
        for v in (variants if rep % 2 == 0 else variants[::-1]):
            run(v, i)
    torch.cuda.synchronize()
    for v in variants:
        for i in range(per):
            if not torch.equal(bufs[v][i], ref[v]):
                bad[v] += 1
                diff = bufs[v][i] != ref[v]
                rows[v] += int(diff.any(dim=1).sum())
                cols[v] |= set(diff.any(dim=0).nonzero().flatten().tolist())
for v in variants:
    c = sorted(cols[v])
    print(f"variant {v} ({names[v]}): {bad[v]} of {reps * per} launches wrong, {rows[v]} rows, columns {c[:6]}..{c[-3:] if c else ''} ({len(c)} distinct)", flush=True)

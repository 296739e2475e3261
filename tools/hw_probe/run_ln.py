"""LayerNorm-backward row body with debug stores, co-running with a conv kernel on a second stream."""
import os, sys, ctypes, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dreg_nerf_amd import ops, lib as L
V = ctypes.CDLL(os.path.join(HERE, os.environ.get("VICTIM_SO", "libvictim.so"))); V.probe_ln_debug.restype = ctypes.c_int
vp = ctypes.c_void_p
dev = torch.device("cuda", 0); lib = L.use_probe()
g0 = torch.Generator().manual_seed(1)
R = 1920
xl = torch.randn(R, 256, generator=g0).to(dev); dy = torch.randn(R, 256, generator=g0).to(dev).bfloat16()
gam = torch.randn(256, generator=g0).to(dev); stats = (torch.rand(R, 2, generator=g0) + 0.5).to(dev)
cx = torch.randn(1, 1, 1, R, 256, device=dev).bfloat16(); w = torch.randn(1024, 256, device=dev)
wpk = ops.packed_weight(w, 256, False, 0)
side = torch.cuda.Stream(); spin = torch.zeros(1024 * 256, device=dev)
what = sys.argv[1]; mode = int(sys.argv[2])
nb = 64
mk = lambda *sh: [torch.zeros(*sh, device=dev) for _ in range(nb)]
dxs, xs, xhs, sts = mk(R, 256), mk(R, 256), mk(R, 256), mk(R, 128)
def run(i): V.probe_ln_debug(vp(xl.data_ptr()), vp(dy.data_ptr()), vp(gam.data_ptr()), vp(stats.data_ptr()), vp(dxs[i].data_ptr()), vp(xs[i].data_ptr()),
                             vp(xhs[i].data_ptr()), vp(sts[i].data_ptr()), R, mode, vp(L.stream()))
run(0); torch.cuda.synchronize()
rdx, rxh = dxs[0].clone(), xhs[0].clone()
stref = stats.repeat_interleave(64, dim=0).reshape(R, 128)
def corun():
    if what == "fwd_regstaged":
        lib.dreg_conv_set_glds(0); ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False); lib.dreg_conv_set_glds(1)
    elif what == "mfma16": V.probe_mfma_spin(vp(spin.data_ptr()), 2000, 0, 1024, vp(L.stream()))
    elif what == "mfma32": V.probe_mfma_spin(vp(spin.data_ptr()), 2000, 1, 1024, vp(L.stream()))
    elif what == "fwd_glds": ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False)
bad_dx = bad_x = bad_xh = bad_st = 0; shown = 0
for rep in range(40):
    with torch.cuda.stream(side):
        for _ in range(24): corun()
    for i in range(nb): run(i)
    torch.cuda.synchronize()
    for i in range(nb):
        b = not torch.equal(dxs[i], rdx); bad_dx += b
        if mode & 1: bad_x += not torch.equal(xs[i], xl)
        if mode & 2: bad_xh += not torch.equal(xhs[i], rxh)
        if mode & 4: bad_st += not torch.equal(sts[i], stref)
        if b and shown < 5:
            shown += 1
            r = int((dxs[i] != rdx).any(dim=1).nonzero()[0])
            msg = f"  row {r}: dx cols differing {int((dxs[i][r] != rdx[r]).sum())}"
            if mode & 1: msg += f"; x-as-loaded cols differing {(xs[i][r] != xl[r]).nonzero().flatten().tolist()[:20]}"
            if mode & 2: msg += f"; xhat cols differing {(xhs[i][r] != rxh[r]).nonzero().flatten().tolist()[:20]}"
            if mode & 4: msg += f"; stats lanes differing {(sts[i][r] != stref[r]).nonzero().flatten().tolist()[:20]}"
            if mode & 1 and (xs[i][r] != xl[r]).any():
                c = int((xs[i][r] != xl[r]).nonzero()[0]); msg += f"; x got {float(xs[i][r, c])} want {float(xl[r, c])} mean {float(stats[r,0])} rstd {float(stats[r,1])}"
            print(msg)
print(f"co-running {what} mode {mode}: dx wrong {bad_dx}, x-as-loaded wrong {bad_x}, xhat wrong {bad_xh}, stats-as-loaded wrong {bad_st} of {40*nb}")

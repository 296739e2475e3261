"""Co-execution probe: self-checking load / arithmetic kernels on the main stream while a conv kernel runs on a second stream."""
import os, sys, ctypes, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dreg_nerf_amd import ops, lib as L
so = os.path.join(HERE, "libvictim.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "victim.hip")])
V = ctypes.CDLL(so)
for f in (V.probe_fill, V.probe_load_check, V.probe_valu_check): f.restype = ctypes.c_int
vp = ctypes.c_void_p
dev = torch.device("cuda", 0); lib = L.use_probe()
R = 1920
x = torch.empty(R * 256, dtype=torch.int32, device=dev); y = torch.empty(R * 128, dtype=torch.int32, device=dev); s = torch.empty(R * 2, dtype=torch.int32, device=dev)
for t in (x, y, s): V.probe_fill(vp(t.data_ptr()), ctypes.c_size_t(t.numel()), vp(L.stream()))
err = torch.zeros(512, dtype=torch.int32, device=dev)
cx = torch.randn(1, 1, 1, R, 256, device=dev).bfloat16(); w = torch.randn(1024, 256, device=dev)
gy = torch.randn(1, 1, 1, R, 1024, device=dev).bfloat16()
wpk = ops.packed_weight(w, 256, False, 0)
side = torch.cuda.Stream()
what = sys.argv[1] if len(sys.argv) > 1 else "fwd_regstaged"
def corun():
    if what == "fwd_regstaged":
        lib.dreg_conv_set_glds(0); ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False); lib.dreg_conv_set_glds(1)
    elif what == "fwd_glds": ops.conv_igemm(cx, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False)
    elif what == "wgrad": ops.conv_wgrad(gy, cx, (1024, 256), 256, 1, 1, 0, True)
torch.cuda.synchronize()
for rep in range(40):
    with torch.cuda.stream(side):
        for _ in range(24): corun()
    for i in range(64):
        V.probe_load_check(vp(x.data_ptr()), vp(y.data_ptr()), vp(s.data_ptr()), R, vp(err.data_ptr()), 1, vp(L.stream()))
        V.probe_valu_check(vp(err.data_ptr()), 64, 30, vp(L.stream()))
    torch.cuda.synchronize()
e = err.cpu().tolist()
print(f"co-running {what}: load mismatches {e[0]}, arithmetic mismatches {e[1]}")
for k in range(min(e[0], 12)):
    print(f"  row {e[4+4*k]} lane {e[5+4*k]} badmask {e[6+4*k]:08b} got v.x {e[7+4*k] & 0xffffffff:08x}")

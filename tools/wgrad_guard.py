"""Does dreg_conv3d_wgrad write outside its workspace / output?  Guard zones around both."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L
dev = torch.device("cuda", 0)
lib = L.load()
g0 = torch.Generator().manual_seed(1)
G = 1 << 20   # guard bytes
for (R, cin, cout) in [(5944, 256, 1024), (5944, 256, 768), (5944, 256, 256), (5944, 1024, 256), (6001, 256, 1024), (35664, 256, 256), (777, 256, 768)]:
    x = torch.randn(1, 1, 1, R, cin, generator=g0).to(dev).bfloat16()
    gy = torch.randn(1, 1, 1, R, cout, generator=g0).to(dev).bfloat16()
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(1, 1, 1, R, cin, cout, 1, 0)
    buf = torch.full((nbytes + 2 * G,), 0x5A, dtype=torch.uint8, device=dev)
    dwb = torch.full((cout * cin * 4 + 2 * G,), 0x5A, dtype=torch.uint8, device=dev)
    dw_ptr = dwb.data_ptr() + G
    dwb[G:G + cout * cin * 4] = 0
    rc = lib.dreg_conv3d_wgrad(L.ptr(gy), L.ptr(x), dw_ptr, buf.data_ptr() + G, nbytes, 1, 1, 1, R, cin, cin, 1, 1, R, cout, 1, 1, 0, 1, 0, 1, L.stream())
    torch.cuda.synchronize()
    ok = bool((buf[:G] == 0x5A).all() and (buf[G + nbytes:] == 0x5A).all() and (dwb[:G] == 0x5A).all() and (dwb[G + cout * cin * 4:] == 0x5A).all())
    print(f"R={R} {cin}->{cout}: rc={rc} ws bytes {nbytes} splits {lib.dreg_conv3d_wgrad_splits(1,1,1,R,cin,cout,1,0)} guards intact: {ok}", flush=True)

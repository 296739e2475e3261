#!/usr/bin/env python3
"""Does the build LEARN?  N optimizer steps (bf16 product path: native executor, active-set head, fused losses, FlatAdamW) on a small pool of
synthetic shell scenes that share one relative pose; prints the loss / RRE / RTE trend (median over windows of 20 steps).
With no dataset or network access this is the feasible stand-in for the reference's validation loop (train_nerf_regtr.py:258-291).
usage: python tools/convergence.py [steps] [res] [scenes] [lr]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import losses as LS, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep


def run(steps=300, res=64, scenes=16, lr=1e-4, pairs=4, seed=3407, dev=None):
    dev = dev or torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = NeRFRegTr(precision="bf16").to(dev).train()
    ts = TrainStep(model, lr=lr)
    pose = synth.fixed_pose()
    pool = []
    for i in range(scenes):
        d = synth.shell_pair(res, 1 + 2 * i, 2 + 2 * i, pose=pose)
        pool.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
    hist = []
    for s in range(steps):
        batch = [pool[(pairs * s + j) % scenes] for j in range(pairs)]
        out = ts.step(batch)
        rr, rt = [], []
        for pred, d in zip(ts.last_preds, batch):
            e = LS.evaluate_camera_alignment(pred["pose"][-1].detach(), d["pose"])
            rr.append(float(e["R_error_mean"])); rt.append(float(e["t_error_mean"]))
        hist.append({"loss": float(out["losses"]["total"]), "corr": float(out["losses"]["corr"]), "rre": sum(rr) / len(rr), "rte": sum(rt) / len(rt)})
    return hist


def windows(hist, key, w=20):
    vals = [h[key] for h in hist]
    return [sorted(vals[i:i + w])[w // 2] for i in range(0, len(vals) - w + 1, w)]


if __name__ == "__main__":
    a = sys.argv[1:]
    h = run(int(a[0]) if a else 300, int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 16, float(a[3]) if len(a) > 3 else 1e-4)
    for k in ("loss", "corr", "rre", "rte"):
        print(k, " ".join(f"{v:.3f}" for v in windows(h, k)))

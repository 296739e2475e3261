"""Run the same 3 optimizer steps repeatedly (pipelined, no sync between steps) and compare the resulting parameters."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd import ops, synth
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
cfg = dict(kv.split("=") for kv in sys.argv[1:])
from dreg_nerf_amd import trunk_exec
if "exaux" in cfg:
    trunk_exec.SERIAL_STREAMS = not bool(int(cfg["exaux"]))
shell = (0.3, 0.34)
batch = []
for i in range(2):
    d = {"pose": synth.fixed_pose()[None].clone(), "src_nerf_path": "", "tgt_nerf_path": ""}
    for j, side in enumerate(("src", "tgt")):
        g, mk = synth.shell_grid(64, 20 + 2 * i + j, *shell)
        d[side + "_xyz_rgba"], d[side + "_mask"] = g.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mk
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
def run():
    torch.manual_seed(7)
    m = NeRFRegTr(precision="bf16").to(dev).train()
    ts = TrainStep(m)
    if "pg" in cfg: ts.overlap_param_grads = bool(int(cfg["pg"]))
    if "geo" in cfg: m.async_geometry = bool(int(cfg["geo"]))
    if "native" in cfg: m.native_trunk = bool(int(cfg["native"]))
    if "fused" in cfg: ts.fused_losses = bool(int(cfg["fused"]))
    if cfg.get("join") == "sync":
        import types
        orig_step = ts.optimizer.step
        def stp():
            ts._pg_stream.synchronize() if ts._pg_stream is not None else None
            orig_step()
        ts.optimizer.step = stp
    if cfg.get("join") in ("main", "geo", "all"):
        orig_step3 = ts.optimizer.step
        def stp3():
            if cfg["join"] == "main": torch.cuda.current_stream().synchronize()
            elif cfg["join"] == "geo": m.__dict__["_geo_stream"].synchronize()
            else: torch.cuda.synchronize()
            orig_step3()
        ts.optimizer.step = stp3
    if cfg.get("join") == "after":
        orig_step2 = ts.optimizer.step
        def stp2():
            orig_step2(); torch.cuda.synchronize()
        ts.optimizer.step = stp2
    outs = []
    for s in range(int(cfg.get("steps", 3))):
        o = ts.step(batch)
        if int(cfg.get("sync", 0)): torch.cuda.synchronize()
        outs.append(o["losses"]["total"])
    torch.cuda.synchronize()
    return ts.optimizer.flat_p.clone(), [float(x) for x in outs]
if int(cfg.get("mainstream", 0)):
    _ms = torch.cuda.Stream()
    torch.cuda.set_stream(_ms)
ref, lref = run()
print("hash ref", float(ref.double().sum()), lref)
nd = 0
for t in range(int(cfg.get("trials", 5))):
    p, l = run()
    same = torch.equal(p, ref)
    nd += (not same)
    print("trial", t, "same" if same else "DIFF", "hash", float(p.double().sum()), l)
print(cfg, "-> differing trials:", nd)

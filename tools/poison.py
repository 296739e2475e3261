"""Fill every freshly allocated float tensor (torch.empty / empty_like) with NaN to expose reads of uninitialised memory:
run N training steps and report the first non-finite loss / gradient."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_empty, _empty_like = torch.empty, torch.empty_like
def pempty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point() and t.numel() > 0:
        t.fill_(float("nan"))
    elif t.is_cuda and t.dtype == torch.uint8 and t.numel() > 0:
        t.fill_(255)      # byte workspaces / the executor arena: 0xFF.. = NaN in bf16 and fp32
    return t
def pempty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.is_floating_point() and t.numel() > 0:
        t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = pempty, pempty_like
from dreg_nerf_amd import ops, synth, trunk_exec
from dreg_nerf_amd.regtr import NeRFRegTr
from dreg_nerf_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "sparse"
shell = (0.3, 0.34) if mode == "sparse" else (0.55, 0.8)
batch = []
for i in range(2):
    d = {"pose": synth.fixed_pose()[None].clone(), "src_nerf_path": "", "tgt_nerf_path": ""}
    for j, side in enumerate(("src", "tgt")):
        g, mk = synth.shell_grid(64, 20 + 2 * i + j, *shell)
        d[side + "_xyz_rgba"], d[side + "_mask"] = g.permute(3, 2, 0, 1).unsqueeze(0).contiguous(), mk
    batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
torch.manual_seed(7)
m = NeRFRegTr(precision="bf16").to(dev).train()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "native": m.native_trunk = bool(int(v))
ts = TrainStep(m)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "fused": ts.fused_losses = bool(int(v))
    if k == "pg": ts.overlap_param_grads = bool(int(v))
names = {id(p): n for n, p in m.named_parameters()}
for s in range(3):
    out = ts.step(batch)
    torch.cuda.synchronize()
    g = ts.optimizer.flat_g
    bad = [names[id(p)] for p, off in zip(ts.optimizer.params, ts.optimizer.offsets) if not torch.isfinite(g[off:off + p.numel()]).all()]
    print(f"step {s}: loss {float(out['losses']['total'])} non-finite grads: {len(bad)} {bad[:6]}; params finite: {bool(torch.isfinite(ts.optimizer.flat_p).all())}", flush=True)

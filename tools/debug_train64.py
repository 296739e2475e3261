import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from dreg_nerf_amd import params, synth, ops
from dreg_nerf_amd import transformer_ops as T, attn_ops as A
from dreg_nerf_amd.regtr import NeRFRegTr
from oracle import regtr_oracle as O

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = synth.shell_pair(res, 1, 2, pose=synth.fixed_pose())
sd = params.synth_state_dict(0)
m = NeRFRegTr(precision="fp32"); m.load_state_dict(params.synth_state_dict(0)); m = m.cuda().train()
with torch.no_grad():
    x = m.pack_grids([data["src_xyz_rgba"].cuda(), data["tgt_xyz_rgba"].cuda()], torch.float32)
    p1 = m.fpn(x)
    sdo = params.clone_state_dict(sd)
    p1s = O.fpn_forward(sdo, data["src_xyz_rgba"][:, 3:], True)
    p1t = O.fpn_forward(sdo, data["tgt_xyz_rgba"][:, 3:], True)
    print("p1 src", rel(p1[0].permute(3, 0, 1, 2), p1s[0]), "tgt", rel(p1[1].permute(3, 0, 1, 2), p1t[0]))
    s_xyz, s_f = O.upsample_gather(p1s, data["src_xyz_rgba"][:, :3], data["src_mask"])
    t_xyz, t_f = O.upsample_gather(p1t, data["tgt_xyz_rgba"][:, :3], data["tgt_mask"])
    ns, nt = s_xyz.shape[0], t_xyz.shape[0]
    idx = torch.cat([data["src_mask"], data["tgt_mask"]]).cuda()
    pb = torch.cat([torch.zeros(ns, dtype=torch.int32), torch.ones(nt, dtype=torch.int32)]).cuda()
    f = ops.trilinear_gather(p1, idx, pb, (res, res, res))
    print("gather", rel(f, torch.cat([s_f, t_f])), ns, nt)
    pts = torch.cat([s_xyz, t_xyz])
    po, fo, lo = O.hierarchical_grid_subsample(pts, torch.cat([s_f, t_f]), torch.tensor([ns, nt]))
    pg, fg, lg = T.hierarchical_grid_subsample(pts.cuda(), torch.cat([s_f, t_f]).cuda(), torch.tensor([ns, nt]).cuda())
    print("downsample lens", lo.tolist(), lg.tolist(), "pts", rel(pg, po), "feats", rel(fg, fo))
    n0 = int(lo[0])
    spe, tpe = O.posenc_sine(po[:n0]), O.posenc_sine(po[n0:])
    sc, tc = O.cross_encoder(sdo, fo[:n0], fo[n0:], spe, tpe)
    P = m._P()
    A.set_precision("fp32")
    scg, tcg = T.cross_encoder(P, fo[:n0].cuda(), fo[n0:].cuda(), spe.cuda(), tpe.cuda())
    print("encoder", rel(scg, sc), rel(tcg, tc))
    a1, a2, a3, a4 = O.corr_decoder(sdo, sc, tc, po[:n0], po[n0:], spe, tpe)
    b1, b2, b3, b4 = T.corr_decoder(P, sc.cuda(), tc.cuda(), po[:n0].cuda(), po[n0:].cuda(), spe.cuda(), tpe.cuda())
    print("decoder", rel(b1, a1), rel(b2, a2), rel(b3, a3), rel(b4, a4))

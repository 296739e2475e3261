"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the DReg-NeRF
pairwise-registration network, written functionally over a flat ``state_dict``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``dreg_nerf_amd``) never does.

Each function cites the reference lines whose arithmetic it restates
(paths relative to the upstream repo AIBluefisher/DReg-NeRF):

  A1  ResNet3D-50 backbone            conerf/model/resnet3d.py:76-113,116-172,197-205
  A2  FeaturePyramid_v1               conerf/model/feature_pyramid_net.py:39-108
  A3  trilinear upsample + gather     conerf/register/nerf_regtr.py:138-147
  A4  hierarchical voxel downsample   conerf/register/grid_downsample.py:6-94  (MinkowskiEngine:
      arithmetic lives upstream and is unpinned -> "parity unpinned" for this row)
  A5  sine position embedding         conerf/register/position_embedding.py:30-53
  A6  cross-encoder (pre-norm)        conerf/register/transformer.py:50-86,225-299
  A7  correspondence decoder          conerf/register/nerf_regtr.py:273-308,350-394
  A8  weighted Kabsch                 conerf/register/se3.py:89-140
  A9  NeRFRegTr.forward glue          conerf/register/nerf_regtr.py:112-248
  H1  losses of the training step     train_nerf_regtr.py:171-256, conerf/loss/*.py
  H2  RRE / RTE                       eval_nerf_regtr.py:24-65

Pinning: the reference ships no tests or golden vectors.  This restatement is pinned against
the reference's own Python modules imported in the build container (tools/make_golden.py);
the resulting vectors live in tests/golden/.  A4 cannot be pinned (MinkowskiEngine absent).
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

RESNET50_BLOCKS = (3, 4, 6, 3)
RESNET50_PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5
N_HEADS = 8
N_LAYERS = 6
D_MODEL = 256


# --------------------------------------------------------------------------- operand-precision emulation (a yardstick, not a path)
# EMULATE = "bf16": every GEMM-shaped operand (convolution / linear inputs and weights, attention q, k, v and probabilities) and every
# stored activation of the FPN3D half is rounded to bfloat16 (round-to-nearest-even) on the way forward, and the gradient flowing back
# through the same point is rounded as well — where the bf16 build keeps bf16 tensors.  Accumulation stays in the tensors' own
# precision.  The network at a random initialisation is badly conditioned (train-mode BatchNorm over 8..64 voxels in layer4: the
# fp32 reference itself is only within 1e-2 of an fp64 evaluation for the early-layer gradients), so "how far does bf16 rounding
# ALONE move the reference's results" is the scale the bf16 build is judged against (tests/test_hip_pinned_step.py).
EMULATE = None


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _q(x: torch.Tensor) -> torch.Tensor:
    return _RoundBF16.apply(x) if EMULATE == "bf16" else x


def _conv(x, w, b=None, stride=1, padding=0, addend=None):
    """F.conv3d with the emulated operand rounding; the addend joins before the result is stored."""
    y = F.conv3d(_q(x), _q(w), b, stride=stride, padding=padding)
    if addend is not None:
        y = y + addend
    return _q(y)


def _linear(x, w, b=None):
    return F.linear(_q(x), _q(w), b)


# --------------------------------------------------------------------------- A1 / A2
def _bn(sd: SD, p: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    """BatchNorm3d, reference semantics = one grid per call (resnet3d.py:121,159).

    ``train``: normalise with this call's batch statistics and update the running buffers
    in ``sd`` in place (momentum 0.1, unbiased variance), as nn.BatchNorm3d does.
    """
    if train:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(
        x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)      # callers round the stored result (after the residual / ReLU)


def _bottleneck(sd: SD, p: str, x: torch.Tensor, stride: int, train: bool) -> torch.Tensor:
    """resnet3d.py:95-113 — 1x1x1 -> 3x3x3(stride) -> 1x1x1, BN after each, residual, ReLU."""
    out = _q(F.relu(_bn(sd, p + ".bn1", _conv(x, sd[p + ".conv1.weight"]), train)))
    out = _q(F.relu(_bn(sd, p + ".bn2",
                        _conv(out, sd[p + ".conv2.weight"], stride=stride, padding=1), train)))
    out = _bn(sd, p + ".bn3", _conv(out, sd[p + ".conv3.weight"]), train)
    if (p + ".downsample.0.weight") in sd:
        res = _q(_bn(sd, p + ".downsample.1",
                     _conv(x, sd[p + ".downsample.0.weight"], stride=stride), train))
    else:
        res = x
    return _q(F.relu(out + res))


def resnet3d_forward(sd: SD, x: torch.Tensor, train: bool, p: str = "fpn3d.backbone_net"):
    """resnet3d.py:157-172.  x [1,4,D,H,W] -> (c1..c5)."""
    c1 = _q(F.relu(_bn(sd, p + ".bn1", _conv(x, sd[p + ".conv1.weight"], stride=2, padding=2), train)))
    h = F.max_pool3d(c1, kernel_size=3, stride=2, padding=1)
    feats = [c1]
    for li, nblk in enumerate(RESNET50_BLOCKS):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 0) else 1
            h = _bottleneck(sd, f"{p}.layer{li + 1}.{b}", h, stride, train)
        feats.append(h)
    return tuple(feats)


def _up_crop(x: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """feature_pyramid_net.py:58-61 — nearest x2 then crop to the lateral's size."""
    d, h, w = like.shape[2:]
    return F.interpolate(x, scale_factor=2)[:, :, :d, :h, :w]


def fpn_forward(sd: SD, x: torch.Tensor, train: bool) -> torch.Tensor:
    """feature_pyramid_net.py:63-105.  x [1,4,D,H,W] -> P1 [1,256,D/2,H/2,W/2]."""
    c1, c2, c3, c4, c5 = resnet3d_forward(sd, x, train)
    q = "fpn3d.feature_pyramid."

    def conv(name, t, pad, addend=None):
        return _conv(t, sd[q + name + ".weight"], sd[q + name + ".bias"], padding=pad, addend=addend)

    p5 = conv("pyramid_transformation_5", c5, 0)
    p4 = conv("upsample_transform_4", conv("pyramid_transformation_4", c4, 0, _up_crop(p5, c4)), 1)
    p3 = conv("upsample_transform_3", conv("pyramid_transformation_3", c3, 0, _up_crop(p4, c3)), 1)
    p2 = conv("upsample_transform_2", conv("pyramid_transformation_2", c2, 0, _up_crop(p3, c2)), 1)
    p1 = conv("upsample_transform_1", conv("pyramid_transformation_1", c1, 1, _up_crop(p2, c1)), 1)
    return p1


# --------------------------------------------------------------------------- A3
def upsample_gather(p1: torch.Tensor, xyz_grid: torch.Tensor, mask: torch.Tensor):
    """nerf_regtr.py:138-147.  Trilinear (align_corners) to the xyz grid's size, channel-last
    flatten with index (x*Y + y)*Z + z, gather the masked rows."""
    res = xyz_grid.shape[-3:]
    up = F.interpolate(p1, size=res, mode="trilinear", align_corners=True)
    feats = up.permute(0, 3, 4, 2, 1).reshape(1, -1, up.shape[1])[0, mask]
    xyz = xyz_grid.permute(0, 3, 4, 2, 1).reshape(1, -1, 3)[0, mask]
    return xyz, feats


def trilinear_gather_direct(p1: torch.Tensor, res: Tuple[int, int, int], mask: torch.Tensor):
    """Same result as upsample_gather's feature rows without materialising the upsampled grid
    (8-corner gather at the masked voxels).  Used to cross-check the fused HIP kernel's weights.
    p1 [1,C,d,h,w]; tensor dims are (z, x, y) in the reference's naming; mask = (x*Y+y)*Z+z."""
    Zr, Xr, Yr = res
    d, h, w = p1.shape[2:]
    z = mask % Zr
    y = (mask // Zr) % Yr
    x = mask // (Zr * Yr)

    def axis(i, n_out, n_in):
        s = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        f = i.to(torch.float32) * s
        i0 = f.floor().to(torch.int64).clamp_(max=n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        t = f - i0.to(torch.float32)
        return i0, i1, t

    z0, z1, tz = axis(z, Zr, d)
    x0, x1, tx = axis(x, Xr, h)
    y0, y1, ty = axis(y, Yr, w)
    g = p1[0].permute(1, 2, 3, 0)  # [d,h,w,C]
    out = 0
    for (zi, wz) in ((z0, 1 - tz), (z1, tz)):
        for (xi, wx) in ((x0, 1 - tx), (x1, tx)):
            for (yi, wy) in ((y0, 1 - ty), (y1, ty)):
                out = out + g[zi, xi, yi] * (wz * wx * wy)[:, None]
    return out


# --------------------------------------------------------------------------- A4
def grid_subsample(points: torch.Tensor, feats: torch.Tensor, lengths: torch.Tensor, dl: float):
    """grid_downsample.py:6-44 with MinkowskiEngine UNWEIGHTED_AVERAGE semantics restated:
    key = (batch, floor(p / dl)) as int32; rows sharing a key are averaged (xyz and features);
    output rows are grouped by batch.  Order inside a batch is implementation-defined upstream;
    this oracle (and the build) use ascending lexicographic (ix, iy, iz).  The sum inside a
    voxel runs in ascending input-row order."""
    n = points.shape[0]
    b_idx = torch.repeat_interleave(torch.arange(len(lengths)), lengths.cpu())
    cell = torch.floor(points / dl).to(torch.int32)
    key = torch.cat([b_idx[:, None].to(torch.int32), cell], dim=1)
    uniq, inv = torch.unique(key, dim=0, return_inverse=True)
    m = uniq.shape[0]
    fp = torch.cat([points, feats], dim=1)
    acc = torch.zeros(m, fp.shape[1], dtype=fp.dtype).index_add_(0, inv, fp)
    cnt = torch.zeros(m, dtype=fp.dtype).index_add_(0, inv, torch.ones(n, dtype=fp.dtype))
    out = acc / cnt[:, None]
    new_len = torch.stack([(uniq[:, 0] == b).sum() for b in range(len(lengths))]).to(torch.int64)
    return out[:, :3], out[:, 3:], new_len


def hierarchical_grid_subsample(points, feats, lengths, num_hierarchical=6,
                                init_dl=0.025, radius=2.75, max_points=1500):
    """grid_downsample.py:47-94.  dl_k = 2*(init_dl*radius*2^k)/radius."""
    radius_normal = init_dl * radius
    for _ in range(num_hierarchical):
        dl = 2 * radius_normal / radius
        points, feats, lengths = grid_subsample(points, feats, lengths, dl)
        radius_normal *= 2
        if points.shape[0] <= 2 * max_points:
            break
    return points, feats, lengths


# --------------------------------------------------------------------------- A5
def posenc_sine(xyz: torch.Tensor, d_model: int = D_MODEL, temperature: float = 1000.0,
                scale: float = 1.0) -> torch.Tensor:
    """position_embedding.py:30-53.  84 features per coordinate (sin on even, cos on odd
    indices, frequency index floor(i/2)), concatenated x|y|z, zero-padded to d_model."""
    n_dim = xyz.shape[-1]
    npf = d_model // n_dim // 2 * 2
    i = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / npf)
    v = (xyz * (scale * 2 * math.pi)).unsqueeze(-1) / dim_t
    emb = torch.stack([v[..., 0::2].sin(), v[..., 1::2].cos()], dim=-1).reshape(*xyz.shape[:-1], -1)
    return F.pad(emb, (0, d_model - npf * n_dim))


def posenc_learned(sd: SD, xyz: torch.Tensor, p: str = "pos_embed") -> torch.Tensor:
    """position_embedding.py:56-76: Linear(3,32) ReLU Linear(32,64) ReLU Linear(64,128) ReLU Linear(128,256) ReLU Linear(256,d)."""
    h = xyz
    for i in range(5):
        h = _linear(h, sd[f"{p}.mlp.{2 * i}.weight"], sd[f"{p}.mlp.{2 * i}.bias"])
        if i < 4:
            h = torch.relu(h)
    return h


# --------------------------------------------------------------------------- A6
def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def _mha(sd: SD, p: str, q_in, k_in, v_in) -> torch.Tensor:
    """nn.MultiheadAttention (8 heads, no masks, dropout 0) on [N, 256] inputs (batch of 1).
    q is scaled by 1/sqrt(d_head) before QK^T, as torch's multi_head_attention_forward does."""
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    e = w.shape[1]
    dh = e // N_HEADS
    q = _q(_linear(q_in, w[:e], b[:e])).view(-1, N_HEADS, dh).transpose(0, 1)
    k = _q(_linear(k_in, w[e:2 * e], b[e:2 * e])).view(-1, N_HEADS, dh).transpose(0, 1)
    v = _q(_linear(v_in, w[2 * e:], b[2 * e:])).view(-1, N_HEADS, dh).transpose(0, 1)
    att = torch.softmax((q * (1.0 / math.sqrt(dh))) @ k.transpose(1, 2), dim=-1)
    o = (_q(att) @ v).transpose(0, 1).reshape(-1, e)
    return _linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def encoder_layer(sd: SD, p: str, src, tgt, src_pe, tgt_pe):
    """transformer.py:225-299 (forward_pre, sa/ca values carry the position embedding)."""
    s2 = _ln(sd, p + ".norm1", src) + src_pe
    src = src + _mha(sd, p + ".self_attn", s2, s2, s2)
    t2 = _ln(sd, p + ".norm1", tgt) + tgt_pe
    tgt = tgt + _mha(sd, p + ".self_attn", t2, t2, t2)

    s2 = _ln(sd, p + ".norm2", src) + src_pe
    t2 = _ln(sd, p + ".norm2", tgt) + tgt_pe
    s3 = _mha(sd, p + ".cross_attn", s2, t2, t2)
    t3 = _mha(sd, p + ".cross_attn", t2, s2, s2)
    src, tgt = src + s3, tgt + t3

    def ffn(x):
        h = _ln(sd, p + ".norm3", x)
        return _linear(F.relu(_linear(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                       sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])

    return src + ffn(src), tgt + ffn(tgt)


def cross_encoder(sd: SD, src, tgt, src_pe, tgt_pe):
    """transformer.py:50-86: every layer's output passes through the shared final LayerNorm."""
    outs_s, outs_t = [], []
    for l in range(N_LAYERS):
        src, tgt = encoder_layer(sd, f"transformer_encoder.layers.{l}", src, tgt, src_pe, tgt_pe)
        outs_s.append(_ln(sd, "transformer_encoder.norm", src))
        outs_t.append(_ln(sd, "transformer_encoder.norm", tgt))
    return torch.stack(outs_s), torch.stack(outs_t)  # [6, N, 256]


# --------------------------------------------------------------------------- A7
def corr_decoder(sd: SD, src_f, tgt_f, src_xyz, tgt_xyz, src_pe, tgt_pe):
    """nerf_regtr.py:350-394.  src_f/tgt_f [6,N,256].  q_norm is never applied (:266)."""
    p = "correspondence_decoder"

    def attend(qf, kf, val):
        q = _q(_linear(qf, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]) / math.sqrt(qf.shape[-1]))
        k = _q(_linear(kf, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"]))
        return torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ val      # V = xyz: fp32 on the VALU in the build

    s2, t2 = src_f + src_pe, tgt_f + tgt_pe
    src_corr = attend(s2, t2, tgt_xyz)
    tgt_corr = attend(t2, s2, src_xyz)
    w, b = sd[p + ".conf_logits_decoder.weight"], sd[p + ".conf_logits_decoder.bias"]
    return src_corr, tgt_corr, torch.sigmoid(F.linear(src_f, w, b)), torch.sigmoid(F.linear(tgt_f, w, b))


# --------------------------------------------------------------------------- A8
def weighted_kabsch(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, eps: float = 1e-6):
    """se3.py:89-140: T with T*a = b.  a,b [*,N,3], w [*,N] in [0,1]."""
    wn = w[..., None] / torch.clamp_min(w.sum(-1, keepdim=True)[..., None], eps)
    ca, cb = (a * wn).sum(-2), (b * wn).sum(-2)
    cov = (a - ca[..., None, :]).transpose(-2, -1) @ ((b - cb[..., None, :]) * wn)
    u, _, vh = torch.linalg.svd(cov)
    v = vh.transpose(-1, -2)
    r_pos = v @ u.transpose(-1, -2)
    v_neg = v.clone()
    v_neg[..., 2] *= -1
    r = torch.where(torch.det(r_pos)[..., None, None] > 0, r_pos, v_neg @ u.transpose(-1, -2))
    t = -r @ ca[..., :, None] + cb[..., :, None]
    return torch.cat([r, t], dim=-1)


# --------------------------------------------------------------------------- A9
def regtr_forward(sd: SD, data: dict, train: bool, num_downsample: int = 6) -> dict:
    """nerf_regtr.py:112-248 for one pair (the reference is batch-1 by construction)."""
    sx, tx = data["src_xyz_rgba"], data["tgt_xyz_rgba"]
    p1_s = fpn_forward(sd, sx[:, 3:], train)
    p1_t = fpn_forward(sd, tx[:, 3:], train)
    s_xyz, s_f = upsample_gather(p1_s, sx[:, :3], data["src_mask"])
    t_xyz, t_f = upsample_gather(p1_t, tx[:, :3], data["tgt_mask"])
    lengths = torch.tensor([s_xyz.shape[0], t_xyz.shape[0]], dtype=torch.int64)
    pts, feats, lens = hierarchical_grid_subsample(
        torch.cat([s_xyz, t_xyz]), torch.cat([s_f, t_f]), lengths, num_downsample)
    ns = int(lens[0])
    s_xyz, t_xyz, s_f, t_f = pts[:ns], pts[ns:], feats[:ns], feats[ns:]
    s_pe, t_pe = posenc_sine(s_xyz), posenc_sine(t_xyz)
    s_c, t_c = cross_encoder(sd, s_f, t_f, s_pe, t_pe)
    s_corr, t_corr, s_ov, t_ov = corr_decoder(sd, s_c, t_c, s_xyz, t_xyz, s_pe, t_pe)
    nl = s_c.shape[0]
    a = torch.cat([s_xyz.expand(nl, -1, -1), t_corr], dim=1)
    b = torch.cat([s_corr, t_xyz.expand(nl, -1, -1)], dim=1)
    w = torch.cat([s_ov[..., 0], t_ov[..., 0]], dim=1)
    pose = weighted_kabsch(a, b, w)[:, None]
    return {
        "src_feats": [s_c], "tgt_feats": [t_c],
        "src_kp": [s_xyz], "src_kp_warped": [s_corr],
        "tgt_kp": [t_xyz], "tgt_kp_warped": [t_corr],
        "src_overlap": [s_ov], "tgt_overlap": [t_ov],
        "pose": pose,
    }


# --------------------------------------------------------------------------- H1
def se3_apply(pose: torch.Tensor, xyz: torch.Tensor) -> torch.Tensor:
    """se3.py:69-86 (R x + t) for one [3|4,4] pose and [N,3] points."""
    return xyz @ pose[:3, :3].T + pose[:3, 3]


def se3_inverse(pose: torch.Tensor) -> torch.Tensor:
    """se3.py:34-39."""
    r, t = pose[..., :3, :3], pose[..., :3, 3:4]
    rt = r.transpose(-1, -2)
    return torch.cat([rt, -rt @ t], dim=-1)


def pseudo_huber(x: torch.Tensor, scale: float = 0.5) -> torch.Tensor:
    """robust_loss_pytorch.general.lossfun at alpha=1 (correspondence_loss.py:31-35; the
    package is an unpinned git dependency): scale-normalised Charbonnier sqrt((x/c)^2+1)-1."""
    return torch.sqrt((x / scale) ** 2 + 1.0) - 1.0


def corr_loss(kp, kp_warped, pose, weights, robust: bool, eps: float = 1e-6):
    """correspondence_loss.py:16-51 ('mae').  weights [nl,N,1] broadcast against err [N]
    exactly as the reference does (torch.cat of a 1-element list keeps [nl,N,1])."""
    err = kp_warped - se3_apply(pose, kp)
    if robust:
        err = pseudo_huber(err)
    err = err.abs().sum(-1)
    return (weights * err).sum() / torch.clamp_min(weights.sum(), eps)


def infonce_loss(W, anchor_f, pos_f, anchor_xyz, pos_xyz, r_p=0.2, r_n=0.4):
    """feature_loss.py:24-60."""
    wt = torch.triu(W)
    logits = anchor_f @ (wt + wt.T) @ pos_f.T
    with torch.no_grad():
        dist = torch.cdist(anchor_xyz, pos_xyz)
        d1, i1 = dist.topk(k=1, dim=-1, largest=False)
        mask = d1[..., 0] < r_p
        ignore = dist < r_n
        ignore.scatter_(-1, i1, 0)
    logits = logits.masked_fill(ignore, -float("inf"))
    loss = -torch.gather(logits, -1, i1).squeeze(-1) + torch.logsumexp(logits, dim=-1)
    return loss[mask].sum() / mask.sum()


def training_losses(pred: dict, pose_gt: torch.Tensor, W: torch.Tensor,
                    src_ov_gt: torch.Tensor, tgt_ov_gt: torch.Tensor,
                    src_ov_tilde: torch.Tensor, tgt_ov_tilde: torch.Tensor,
                    robust: bool = False) -> dict:
    """train_nerf_regtr.py:171-229 for one pair.  The visibility scores (overlap GT and the
    'tilde' scores of the warped key points, both [nl,N,1] in {0,1}) are inputs here: in the
    reference they come from NeRF ray marching.  Quirk Q3 kept:
    BCEWithLogits(input=GT, target=pred)."""
    ov_gt = torch.cat([src_ov_gt, tgt_ov_gt], dim=-2)
    ov_pred = torch.cat([pred["src_overlap"][0], pred["tgt_overlap"][0]], dim=-2)
    losses = {}
    losses["overlap"] = F.binary_cross_entropy_with_logits(ov_gt[-1], ov_pred[-1])
    losses["nerf_cont"] = F.smooth_l1_loss(ov_gt, torch.cat([src_ov_tilde, tgt_ov_tilde], dim=-2))
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    losses["feature"] = infonce_loss(W, pred["src_feats"][0][-1], pred["tgt_feats"][0][-1],
                                     se3_apply(pose_gt[0], s_kp), t_kp)
    losses["corr"] = corr_loss(s_kp, pred["src_kp_warped"][0][-1], pose_gt[0], src_ov_gt, robust) + \
        corr_loss(t_kp, pred["tgt_kp_warped"][0][-1], se3_inverse(pose_gt[0]), tgt_ov_gt, robust)
    weights = {"overlap": 1.0, "nerf_cont": 1.0, "feature": 0.1, "corr": 1.0}
    losses["total"] = sum(losses[k] * weights[k] for k in weights)
    return losses


# --------------------------------------------------------------------------- H2
def rre_rte(pred: torch.Tensor, gt: torch.Tensor, eps: float = 1e-7):
    """eval_nerf_regtr.py:24-65.  pred [B,3,4], gt [B,4,4] -> (RRE deg [B], RTE [B])."""
    rd = pred[..., :3, :3].transpose(-2, -1) @ gt[..., :3, :3]
    tr = rd[..., 0, 0] + rd[..., 1, 1] + rd[..., 2, 2]
    ang = torch.rad2deg(((tr - 1) / 2).clamp(-1 + eps, 1 - eps).acos())
    return ang, (pred[..., :3, 3] - gt[..., :3, 3]).norm(dim=-1)

"""Point-set half of the registration network (rows A4-A8 of SURVEY.md §8): voxel-average
downsampling, sine position embedding, the 6-layer self/cross-attention encoder, the correspondence
decoder and the weighted Kabsch solve.  Functions take the model's flat parameter dict.

Reference: conerf/register/grid_downsample.py:6-94, position_embedding.py:30-53, transformer.py:50-86,225-299,
nerf_regtr.py:273-308,350-394, se3.py:89-140.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import attn_ops as A

N_HEADS = 8
N_LAYERS = 6


# --------------------------------------------------------------------------- A4
def grid_subsample(points, feats, lengths, dl: float):
    """Mean of (xyz | feat) over the points sharing (batch, floor(p/dl)); rows ordered by (batch, ix, iy, iz)."""
    return A.voxel_mean_downsample(points, feats, lengths, dl)


def hierarchical_grid_subsample(points, feats, lengths, num_hierarchical=6, init_dl=0.025, radius=2.75, max_points=1500):
    radius_normal = init_dl * radius
    for _ in range(num_hierarchical):
        dl = 2 * radius_normal / radius
        points, feats, lengths = grid_subsample(points, feats, lengths, dl)
        radius_normal *= 2
        if points.shape[0] <= 2 * max_points:
            break
    return points, feats, lengths


# --------------------------------------------------------------------------- A5
def posenc_sine(xyz: torch.Tensor, d_model: int = 256, temperature: float = 1000.0, scale: float = 1.0):
    return A.posenc_sine(xyz, d_model, temperature, scale)


# --------------------------------------------------------------------------- A6
def _ln(P, p, x):
    return A.layer_norm(x, P[p + ".weight"], P[p + ".bias"])


def _mha(P, p, q_in, k_in, v_in, self_attn: bool):
    w, b = P[p + ".in_proj_weight"], P[p + ".in_proj_bias"]
    e = w.shape[1]
    if self_attn:
        qkv = A.linear(q_in, w, b)
        q, k, v = qkv[:, :e], qkv[:, e:2 * e], qkv[:, 2 * e:]
    else:
        q = A.linear(q_in, w[:e], b[:e])
        kv = A.linear(k_in, w[e:], b[e:])
        k, v = kv[:, :e], kv[:, e:]
    o = A.attention(q, k, v, N_HEADS, 1.0 / math.sqrt(e // N_HEADS))
    return A.linear(o, P[p + ".out_proj.weight"], P[p + ".out_proj.bias"])


def encoder_layer(P, p, src, tgt, src_pe, tgt_pe):
    s2 = _ln(P, p + ".norm1", src) + src_pe
    src = src + _mha(P, p + ".self_attn", s2, s2, s2, True)
    t2 = _ln(P, p + ".norm1", tgt) + tgt_pe
    tgt = tgt + _mha(P, p + ".self_attn", t2, t2, t2, True)
    s2 = _ln(P, p + ".norm2", src) + src_pe
    t2 = _ln(P, p + ".norm2", tgt) + tgt_pe
    s3 = _mha(P, p + ".cross_attn", s2, t2, t2, False)
    t3 = _mha(P, p + ".cross_attn", t2, s2, s2, False)
    src, tgt = src + s3, tgt + t3

    def ffn(x):
        h = _ln(P, p + ".norm3", x)
        h = A.linear(h, P[p + ".linear1.weight"], P[p + ".linear1.bias"], relu=True)
        return A.linear(h, P[p + ".linear2.weight"], P[p + ".linear2.bias"])

    return src + ffn(src), tgt + ffn(tgt)


def cross_encoder(P, src, tgt, src_pe, tgt_pe):
    outs_s, outs_t = [], []
    for l in range(N_LAYERS):
        src, tgt = encoder_layer(P, f"transformer_encoder.layers.{l}", src, tgt, src_pe, tgt_pe)
        outs_s.append(_ln(P, "transformer_encoder.norm", src))
        outs_t.append(_ln(P, "transformer_encoder.norm", tgt))
    return torch.stack(outs_s), torch.stack(outs_t)


# --------------------------------------------------------------------------- A7
def corr_decoder(P, src_f, tgt_f, src_xyz, tgt_xyz, src_pe, tgt_pe):
    p = "correspondence_decoder"
    nl, ns, e = src_f.shape
    nt = tgt_f.shape[1]
    s2, t2 = (src_f + src_pe).reshape(nl * ns, e), (tgt_f + tgt_pe).reshape(nl * nt, e)
    both = torch.cat([s2, t2])
    q = A.linear(both, P[p + ".q_proj.weight"], P[p + ".q_proj.bias"])
    k = A.linear(both, P[p + ".k_proj.weight"], P[p + ".k_proj.bias"])
    qs, qt = q[:nl * ns].view(nl, ns, e), q[nl * ns:].view(nl, nt, e)
    ks, kt = k[:nl * ns].view(nl, ns, e), k[nl * ns:].view(nl, nt, e)
    sc = 1.0 / math.sqrt(e)
    src_corr = A.attention_xyz(qs, kt, tgt_xyz, sc)
    tgt_corr = A.attention_xyz(qt, ks, src_xyz, sc)
    w, b = P[p + ".conf_logits_decoder.weight"], P[p + ".conf_logits_decoder.bias"]
    return src_corr, tgt_corr, A.overlap_head(src_f, w, b), A.overlap_head(tgt_f, w, b)


# --------------------------------------------------------------------------- A8
def weighted_kabsch(a, b, w, eps: float = 1e-6):
    return A.weighted_kabsch(a, b, w, eps)

"""Point-set half of the registration network (rows A4-A8 of SURVEY.md §8): voxel-average
downsampling, sine position embedding, the 6-layer self/cross-attention encoder, the correspondence
decoder and the weighted Kabsch solve.  Functions take the model's flat parameter dict.

Source and target rows are kept concatenated ([Ns+Nt, 256], fp32 residual stream) so every LayerNorm / linear
layer runs once per encoder layer for both point sets (the reference applies the same weights to both,
transformer.py:225-299); only the attention cores see the two sets separately.

Reference: conerf/register/grid_downsample.py:6-94, position_embedding.py:30-53, transformer.py:50-86,225-299,
nerf_regtr.py:273-308,350-394, se3.py:89-140.
"""
import math
from typing import Dict

import torch

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


def plan_hierarchical_subsample(points, lengths, num_hierarchical=6, init_dl=0.025, radius=2.75, max_points=1500):
    """The xyz-only half of hierarchical_grid_subsample: (rounds, subsampled points, lengths).  Every data-dependent size of
    A4 is resolved here, so a caller can run it before the feature network and keep the rest of the step free of host syncs."""
    radius_normal = init_dl * radius
    rounds = []
    for _ in range(num_hierarchical):
        dl = 2 * radius_normal / radius
        rnd, points, lengths = A.plan_voxel_downsample(points, lengths, dl)
        rounds.append(rnd)
        radius_normal *= 2
        if points.shape[0] <= 2 * max_points:
            break
    return rounds, points, lengths


def plan_hierarchical_subsample_all(points, lengths, num_hierarchical=6, init_dl=0.025, radius=2.75, max_points=1500):
    """plan_hierarchical_subsample for ALL pairs of a step with one set of launches (and one host sync) per round: points = the pairs' point sets one after the
    other, lengths = [ns_0, nt_0, ns_1, nt_1, ...].  The reference's stopping rule is per pair (grid_downsample.py:83-94): a pair whose point count has
    dropped to <= 2 * max_points takes no further round — here its two batches are FROZEN in the later rounds (they pass through unchanged and in order), so
    the rounds' arrays cover every pair and downstream consumers see one plan over the whole row space.  Same key points, same segment means as the per-pair
    plans (tests/test_hip_pointset_ops.py); 8 rounds of launches + 8 host syncs per step become 2-3 at the benchmark's batch.
    Returns (rounds, subsampled points of all pairs, lengths)."""
    lens = [int(v) for v in lengths]
    npairs = len(lens) // 2
    frozen = [False] * npairs
    radius_normal = init_dl * radius
    rounds = []
    for _ in range(num_hierarchical):
        dl = 2 * radius_normal / radius
        rnd, points, lens = A.plan_voxel_downsample(points, lens, dl, frozen=[frozen[b // 2] for b in range(2 * npairs)])
        rounds.append(rnd)
        radius_normal *= 2
        for p in range(npairs):
            if lens[2 * p] + lens[2 * p + 1] <= 2 * max_points:
                frozen[p] = True
        if all(frozen):
            break
    return rounds, points, lens


def apply_subsample_plan(rounds, feats):
    for rnd in rounds:
        feats = A.segment_mean(feats, rnd)
    return feats


# --------------------------------------------------------------------------- A5
def posenc_sine(xyz: torch.Tensor, d_model: int = 256, temperature: float = 1000.0, scale: float = 1.0):
    return A.posenc_sine(xyz, d_model, temperature, scale)


# --------------------------------------------------------------------------- A6
def encoder_layer(P, p, x, pe, ns: int):
    """x fp32 [Ns+Nt, 256] (src rows first), pe same shape.  forward_pre of transformer.py:225-299."""
    sc = 1.0 / math.sqrt(256 // N_HEADS)
    # self attention: q = k = v = LN1(x) + pe
    h = A.layer_norm(x, P[p + ".norm1.weight"], P[p + ".norm1.bias"], pe)
    qkv = A.linear(h, P[p + ".self_attn.in_proj_weight"], P[p + ".self_attn.in_proj_bias"])
    o = torch.cat([A.mha_packed(qkv[:ns], qkv[:ns], N_HEADS, sc), A.mha_packed(qkv[ns:], qkv[ns:], N_HEADS, sc)])
    x = A.linear(o, P[p + ".self_attn.out_proj.weight"], P[p + ".self_attn.out_proj.bias"], residual=x, out_f32=True)
    # cross attention: q = LN2(x) + pe of one set, k = v = LN2(other) + pe
    h = A.layer_norm(x, P[p + ".norm2.weight"], P[p + ".norm2.bias"], pe)
    qkv = A.linear(h, P[p + ".cross_attn.in_proj_weight"], P[p + ".cross_attn.in_proj_bias"])
    o = torch.cat([A.mha_packed(qkv[:ns], qkv[ns:], N_HEADS, sc), A.mha_packed(qkv[ns:], qkv[:ns], N_HEADS, sc)])
    x = A.linear(o, P[p + ".cross_attn.out_proj.weight"], P[p + ".cross_attn.out_proj.bias"], residual=x, out_f32=True)
    # feed-forward
    h = A.layer_norm(x, P[p + ".norm3.weight"], P[p + ".norm3.bias"])
    h = A.linear(h, P[p + ".linear1.weight"], P[p + ".linear1.bias"], relu=True)
    return A.linear(h, P[p + ".linear2.weight"], P[p + ".linear2.bias"], residual=x, out_f32=True)


def cross_encoder(P, src, tgt, src_pe, tgt_pe):
    """Returns (src_cond [6,Ns,256], tgt_cond [6,Nt,256]) fp32: every layer's output through the shared final norm."""
    ns = src.shape[0]
    x = torch.cat([src, tgt]).float().contiguous()
    pe = torch.cat([src_pe, tgt_pe]).contiguous()
    outs = []
    for l in range(N_LAYERS):
        x = encoder_layer(P, f"transformer_encoder.layers.{l}", x, pe, ns)
        outs.append(A.layer_norm(x, P["transformer_encoder.norm.weight"], P["transformer_encoder.norm.bias"],
                                 out_dtype=torch.float32))
    out = torch.stack(outs)
    return out[:, :ns], out[:, ns:]


# --------------------------------------------------------------------------- A7
def corr_decoder(P, src_f, tgt_f, src_xyz, tgt_xyz, src_pe, tgt_pe):
    """nerf_regtr.py:350-394.  src_f/tgt_f fp32 [6,N,256] (already through the final LayerNorm)."""
    p = "correspondence_decoder"
    nl, ns, e = src_f.shape
    nt = tgt_f.shape[1]
    cdt = A.compute_dtype()
    s2 = (src_f + src_pe).to(cdt).reshape(nl * ns, e)
    t2 = (tgt_f + tgt_pe).to(cdt).reshape(nl * nt, e)
    both = torch.cat([s2, t2])
    q = A.linear(both, P[p + ".q_proj.weight"], P[p + ".q_proj.bias"])
    k = A.linear(both, P[p + ".k_proj.weight"], P[p + ".k_proj.bias"])
    qs, qt = q[:nl * ns].view(nl, ns, e), q[nl * ns:].view(nl, nt, e)
    ks, kt = k[:nl * ns].view(nl, ns, e), k[nl * ns:].view(nl, nt, e)
    sc = 1.0 / math.sqrt(e)
    src_corr = A.attention_xyz(qs, kt, tgt_xyz, sc)
    tgt_corr = A.attention_xyz(qt, ks, src_xyz, sc)
    w, b = P[p + ".conf_logits_decoder.weight"], P[p + ".conf_logits_decoder.bias"]
    return src_corr, tgt_corr, A.overlap_head(src_f.contiguous(), w, b), A.overlap_head(tgt_f.contiguous(), w, b)


# --------------------------------------------------------------------------- A6 + A7 for all pairs of a step at once
def encoder_layer_batched(P, p, x, pe, tab):
    """x fp32 [R,256]: rows of every pair's (src | tgt) point sets; tab: attn_ops.ProblemTable."""
    sc = 1.0 / math.sqrt(256 // N_HEADS)
    h, xr = A.layer_norm_residual(x, P[p + ".norm1.weight"], P[p + ".norm1.bias"], pe)
    qkv = A.linear(h, P[p + ".self_attn.in_proj_weight"], P[p + ".self_attn.in_proj_bias"])
    o = A.mha_varlen(qkv, tab.self_probs, tab.nprob, tab.max_len, N_HEADS, sc)
    x = A.linear(o, P[p + ".self_attn.out_proj.weight"], P[p + ".self_attn.out_proj.bias"], residual=xr, out_f32=True)
    h, xr = A.layer_norm_residual(x, P[p + ".norm2.weight"], P[p + ".norm2.bias"], pe)
    qkv = A.linear(h, P[p + ".cross_attn.in_proj_weight"], P[p + ".cross_attn.in_proj_bias"])
    o = A.mha_varlen(qkv, tab.cross_probs, tab.nprob, tab.max_len, N_HEADS, sc)
    x = A.linear(o, P[p + ".cross_attn.out_proj.weight"], P[p + ".cross_attn.out_proj.bias"], residual=xr, out_f32=True)
    h, xr = A.layer_norm_residual(x, P[p + ".norm3.weight"], P[p + ".norm3.bias"])
    h = A.linear(h, P[p + ".linear1.weight"], P[p + ".linear1.bias"], relu=True)
    return A.linear(h, P[p + ".linear2.weight"], P[p + ".linear2.bias"], residual=xr, out_f32=True)


def encode_decode_batched(P, feats, xyz, tab, pos_embed=None):
    """feats fp32 [R,256], xyz fp32 [R,3] (rows ordered pair by pair, src then tgt), tab: ProblemTable.
    Returns (cond fp32 [6,R,256], corr fp32 [6,R,3], overlap fp32 [6,R,1])."""
    x = feats.float().contiguous()
    xyz = xyz.contiguous()
    pe = A.posenc_sine(xyz) if pos_embed is None else pos_embed(xyz).contiguous()   # NeRFRegTr.position_embedding
    outs = []
    for l in range(N_LAYERS):
        x = encoder_layer_batched(P, f"transformer_encoder.layers.{l}", x, pe, tab)
        outs.append(x)
    R = x.shape[0]
    allx = torch.cat(outs)  # [6R,256]: one final-LayerNorm launch for the six intermediate outputs
    nw, nb = P["transformer_encoder.norm.weight"], P["transformer_encoder.norm.bias"]
    cond = A.layer_norm(allx, nw, nb, out_dtype=torch.float32)
    dec_in = A.layer_norm(allx, nw, nb, pe.repeat(N_LAYERS, 1))  # LN(x) + pe in the compute dtype
    p = "correspondence_decoder"
    q = A.linear(dec_in, P[p + ".q_proj.weight"], P[p + ".q_proj.bias"]).view(N_LAYERS, R, 256)
    k = A.linear(dec_in, P[p + ".k_proj.weight"], P[p + ".k_proj.bias"]).view(N_LAYERS, R, 256)
    corr = A.attention_xyz_varlen(q, k, xyz, tab.cross_probs, tab.nprob, tab.max_len, 1.0 / math.sqrt(256))
    ov = A.overlap_head(cond, P[p + ".conf_logits_decoder.weight"], P[p + ".conf_logits_decoder.bias"]).view(N_LAYERS, R, 1)
    return cond.view(N_LAYERS, R, 256), corr, ov


# --------------------------------------------------------------------------- A8
def weighted_kabsch(a, b, w, eps: float = 1e-6):
    return A.weighted_kabsch(a, b, w, eps)

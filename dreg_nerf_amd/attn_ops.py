"""Operator layer of the point-set half.  Every entry is the single place where the op is dispatched to
its HIP kernel; ops that still run as torch device ops are marked SCAFFOLD (tracked in DESIGN.md) and
are replaced kernel by kernel without touching transformer_ops.py."""
import math

import torch
import torch.nn.functional as F

from . import lib as L
from . import ops

# fp32 GEMMs of the point-set half go through the exact-f32 MFMA path of the conv kernel when True.
USE_HIP_LINEAR = True
_COMPUTE_DTYPE = torch.float32


def set_precision(precision: str):
    """'bf16': GEMM operands are rounded to bf16 (fp32 accumulate, fp32 residual stream); 'fp32': exact-f32 MFMA."""
    global _COMPUTE_DTYPE
    _COMPUTE_DTYPE = torch.bfloat16 if precision == "bf16" else torch.float32


def linear(x, w, b, relu: bool = False):
    if USE_HIP_LINEAR and x.is_cuda and w.shape[0] % 64 == 0:
        y = ops.linear(x.to(_COMPUTE_DTYPE).contiguous(), w, b).float()
    else:
        y = F.linear(x, w, b)  # SCAFFOLD (only the 1-row overlap head lands here)
    return F.relu(y) if relu else y


def layer_norm(x, w, b, eps: float = 1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)  # SCAFFOLD


def attention(q, k, v, n_heads: int, scale: float):
    """q [Nq,E], k,v [Nk,E] -> softmax(q k^T * scale) v per head, heads concatenated."""
    nq, e = q.shape
    dh = e // n_heads
    qh = q.view(nq, n_heads, dh).transpose(0, 1)
    kh = k.view(-1, n_heads, dh).transpose(0, 1)
    vh = v.view(-1, n_heads, dh).transpose(0, 1)
    att = torch.softmax((qh * scale) @ kh.transpose(1, 2), dim=-1)  # SCAFFOLD
    return (att @ vh).transpose(0, 1).reshape(nq, e)


def attention_xyz(q, k, xyz, scale: float):
    """q [L,Nq,E], k [L,Nk,E], xyz [Nk,3] -> softmax(q k^T * scale) xyz : [L,Nq,3]."""
    att = torch.softmax((q * scale) @ k.transpose(1, 2), dim=-1)  # SCAFFOLD
    return att @ xyz


def overlap_head(f, w, b):
    return torch.sigmoid(F.linear(f, w, b))  # SCAFFOLD


def posenc_sine(xyz, d_model=256, temperature=1000.0, scale=1.0):
    n_dim = xyz.shape[-1]
    npf = d_model // n_dim // 2 * 2
    i = torch.arange(npf, dtype=torch.float32, device=xyz.device)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / npf)
    v = (xyz * (scale * 2 * math.pi)).unsqueeze(-1) / dim_t
    emb = torch.stack([v[..., 0::2].sin(), v[..., 1::2].cos()], dim=-1).reshape(*xyz.shape[:-1], -1)  # SCAFFOLD
    return F.pad(emb, (0, d_model - npf * n_dim))


def voxel_mean_downsample(points, feats, lengths, dl: float):
    n = points.shape[0]
    dev = points.device
    b_idx = torch.repeat_interleave(torch.arange(len(lengths), device=dev), lengths)
    cell = torch.floor(points / dl).to(torch.int32)
    key = torch.cat([b_idx[:, None].to(torch.int32), cell], dim=1)
    uniq, inv = torch.unique(key, dim=0, return_inverse=True)  # SCAFFOLD
    m = uniq.shape[0]
    fp = torch.cat([points, feats], dim=1)
    acc = torch.zeros(m, fp.shape[1], dtype=fp.dtype, device=dev).index_add_(0, inv, fp)
    cnt = torch.zeros(m, dtype=fp.dtype, device=dev).index_add_(0, inv, torch.ones(n, dtype=fp.dtype, device=dev))
    out = acc / cnt[:, None]
    new_len = torch.stack([(uniq[:, 0] == b).sum() for b in range(len(lengths))]).to(torch.int64)
    return out[:, :3], out[:, 3:], new_len


def weighted_kabsch(a, b, w, eps: float = 1e-6):
    wn = w[..., None] / torch.clamp_min(w.sum(-1, keepdim=True)[..., None], eps)
    ca, cb = (a * wn).sum(-2), (b * wn).sum(-2)
    cov = (a - ca[..., None, :]).transpose(-2, -1) @ ((b - cb[..., None, :]) * wn)
    u, _, vh = torch.linalg.svd(cov.cpu())  # SCAFFOLD (3x3 SVD on host)
    u, vh = u.to(a.device), vh.to(a.device)
    v = vh.transpose(-1, -2)
    r_pos = v @ u.transpose(-1, -2)
    v_neg = v.clone()
    v_neg[..., 2] *= -1
    r = torch.where(torch.det(r_pos)[..., None, None] > 0, r_pos, v_neg @ u.transpose(-1, -2))
    t = -r @ ca[..., :, None] + cb[..., :, None]
    return torch.cat([r, t], dim=-1)

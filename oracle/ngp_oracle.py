"""ORACLE (test infrastructure, not product): CPU restatement of the Instant-NGP dense query used to extract
the 128^3 voxel grids (rows B1-B3 of SURVEY.md §8).

PARITY UNPINNED: the arithmetic lives in tiny-cuda-nn (git master, unpinned; scripts/env/install.sh:21) which is absent
from /root/reference and from this image; the reference ships no checkpoint, grid sample or test for it.  This file
restates the published Instant-NGP / tcnn algorithm and the reference's own wrapper code:

  query_density   conerf/radiance_fields/ngp.py:148-176   (aabb normalisation, selector, trunc_exp(x - 1))
  query_rgb       conerf/radiance_fields/ngp.py:178-193   (dir -> (dir+1)/2 -> SH degree 4; colour MLP; sigmoid)
  dense query     conerf/register/sample_grid.py:223-242, 321-341 (jittered sample per occupied cell, 18 fixed directions,
                  mean colour, alpha = clip(1 - exp(-0.01 sigma)), density mask sigma > 0.7)

Numeric specification shared with the HIP kernels: table and weights rounded to fp16; trilinear interpolation in fp32,
rounded to fp16; every layer's inputs are fp16, products accumulated in fp32, activations rounded to fp16.
"""
import math
from typing import List, Tuple

import numpy as np
import torch

N_LEVELS = 16
PER_LEVEL_SCALE = 1.4472692012786865
LOG2_HASHMAP = 19
BASE_RES = 16
PRIMES = (1, 2654435761, 805459861)


def level_table(per_level_scale=PER_LEVEL_SCALE, log2_hashmap=LOG2_HASHMAP, base_res=BASE_RES):
    rows, off = [], 0
    for l in range(N_LEVELS):
        scale = np.float32(np.exp2(np.float32(l) * np.log2(np.float32(per_level_scale)))) * np.float32(base_res) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        n = (res ** 3 + 7) // 8 * 8
        size = min(n, 1 << log2_hashmap)
        rows.append(dict(offset=off, size=size, res=res, scale=float(scale), hashed=res ** 3 > size))
        off += size
    return rows, off


def n_grid_params() -> int:
    return level_table()[1] * 2


def f16(t: torch.Tensor) -> torch.Tensor:
    return t.half().float()


def hash_encode(u: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """u [N,3] in [0,1]; table fp32-valued [entries, 2] (already fp16-rounded) -> [N, 32] (fp16-rounded)."""
    rows, _ = level_table()
    n = u.shape[0]
    out = torch.zeros(n, 2 * N_LEVELS, dtype=torch.float32)
    u = u.clamp(0.0, 1.0)
    for l, lv in enumerate(rows):
        pos = u * np.float32(lv["scale"]) + np.float32(0.5)
        g = torch.floor(pos)
        w = pos - g
        g = g.to(torch.int64)
        acc = torch.zeros(n, 2, dtype=torch.float32)
        for corner in range(8):
            o = torch.tensor([(corner >> 0) & 1, (corner >> 1) & 1, (corner >> 2) & 1])
            c = g + o
            wt = torch.ones(n, dtype=torch.float32)
            for d in range(3):
                wt = wt * (w[:, d] if o[d] else (1 - w[:, d]))
            if lv["hashed"]:
                idx = ((c[:, 0] * PRIMES[0]) & 0xFFFFFFFF) ^ ((c[:, 1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((c[:, 2] * PRIMES[2]) & 0xFFFFFFFF)
            else:
                idx = (c[:, 0] + c[:, 1] * lv["res"] + c[:, 2] * lv["res"] * lv["res"]) & 0xFFFFFFFF
            idx = idx % lv["size"]
            acc = acc + wt[:, None] * table[lv["offset"] + idx]
        out[:, 2 * l:2 * l + 2] = f16(acc)
    return out


def split_density_params(params: torch.Tensor):
    """mlp_base.params (fp32 [12,602,992]): W1 [64,32] | W2 [16,64] | hash table levels in order, 2 features interleaved."""
    w1 = params[:2048].view(64, 32)
    w2 = params[2048:3072].view(16, 64)
    table = params[3072:].view(-1, 2)
    return w1, w2, table


def split_color_params(params: torch.Tensor):
    """color_mlp.params (fp32 [7168]): W1 [64,32] | W2 [64,64] | W3 [16,64]."""
    return params[:2048].view(64, 32), params[2048:6144].view(64, 64), params[6144:].view(16, 64)


def contract_to_unisphere(x: torch.Tensor, aabb: torch.Tensor) -> torch.Tensor:
    """conerf/radiance_fields/ngp.py:41-63 (non-derivative branch): aabb -> [-1,1]^3, |x| > 1 -> (2 - 1/|x|) x/|x|, then /4 + 0.5."""
    lo, hi = aabb[:3], aabb[3:]
    v = (x - lo) / (hi - lo) * 2 - 1
    mag = v.norm(dim=-1, keepdim=True)
    v = torch.where(mag > 1, (2 - 1 / mag) * (v / mag), v)
    return v / 4 + 0.5


def query_density(x: torch.Tensor, aabb: torch.Tensor, mlp_base_params: torch.Tensor, unbounded: bool = False):
    """Returns (density [N] fp32, raw [N,16] fp16-valued: pre-activation density | 15 features)."""
    lo, hi = aabb[:3], aabb[3:]
    u = contract_to_unisphere(x, aabb) if unbounded else (x - lo) / (hi - lo)
    selector = ((u > 0.0) & (u < 1.0)).all(dim=-1)
    w1, w2, table = split_density_params(mlp_base_params)
    enc = hash_encode(u, f16(table))
    h = f16(torch.relu(enc @ f16(w1).T))
    raw = f16(h @ f16(w2).T)
    density = torch.exp(raw[:, 0] - 1.0) * selector.float()
    return density, raw


def sh4(d: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics, degree 4 (16 coefficients) of a direction vector (tcnn convention)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)


def query_rgb(dirs: torch.Tensor, raw: torch.Tensor, color_params: torch.Tensor) -> torch.Tensor:
    """dirs [N,3] (as passed to query_rgb: the wrapper maps to [0,1] and tcnn maps back to [-1,1]); raw [N,16]."""
    w1, w2, w3 = split_color_params(color_params)
    d01 = (dirs + 1.0) / 2.0
    sh = f16(sh4(d01 * 2.0 - 1.0))
    xin = torch.cat([sh, raw[:, 1:16], torch.ones(raw.shape[0], 1)], dim=1)
    h1 = f16(torch.relu(xin @ f16(w1).T))
    h2 = f16(torch.relu(h1 @ f16(w2).T))
    o = f16(h2 @ f16(w3).T)
    return f16(torch.sigmoid(o[:, :3]))


def fixed_viewdirs() -> torch.Tensor:
    """sample_grid.py:131-145 — 18 'fixed viewing directions' exactly as written there (x == y, not normalised)."""
    phis = [math.pi / 3, 0, -math.pi]
    thetas = [k * math.pi / 3 for k in range(6)]
    return torch.tensor([[math.cos(p) * math.sin(t), math.cos(p) * math.sin(t), math.sin(t)] for p in phis for t in thetas],
                        dtype=torch.float32)


def dense_query(binary: torch.Tensor, jitter: torch.Tensor, roi_aabb: torch.Tensor, model_aabb: torch.Tensor,
                mlp_base_params: torch.Tensor, color_params: torch.Tensor, density_thre: float = 0.7, delta: float = 1e-2):
    """sample_grid.py:223-242 + 321-341 with the jitter supplied by the caller (Q11: the reference draws it with
    torch.rand_like on the grid's device).  Returns (world [Np,3], rgb [Np,3], alpha [Np], indices [Np], density_mask [Np])."""
    res = torch.tensor(binary.shape, dtype=torch.float32)
    indices = torch.nonzero(binary.flatten())[:, 0]
    rx, ry, rz = binary.shape
    coords = torch.stack([indices // (ry * rz), (indices // rz) % ry, indices % rz], dim=1).float()
    u = (coords + jitter) / res
    world = u * (roi_aabb[3:] - roi_aabb[:3]) + roi_aabb[:3]
    density, raw = query_density(world, model_aabb, mlp_base_params)
    dirs = fixed_viewdirs()
    rgb = torch.stack([query_rgb(dirs[k].expand(world.shape[0], 3), raw, color_params) for k in range(dirs.shape[0])]).mean(0)
    alpha = torch.clip(1 - torch.exp(-delta * density), 0, 1)
    return world, rgb, alpha, indices, density > density_thre

"""Operator layer of the point-set half: every op dispatches to a HIP kernel through the C ABI
(attention.hip, pointset.hip, and the 1x1x1 path of conv.hip for the linear layers).

Precision: 'bf16' = GEMM/attention operands bf16 with fp32 accumulation and an fp32 residual stream;
'fp32' = exact-f32 MFMA everywhere (the parity mode)."""
import math

import torch

from . import lib as L
from . import ops

_COMPUTE_DTYPE = torch.float32


def set_precision(precision: str):
    global _COMPUTE_DTYPE
    _COMPUTE_DTYPE = torch.bfloat16 if precision == "bf16" else torch.float32


def compute_dtype():
    return _COMPUTE_DTYPE


# --------------------------------------------------------------------------- linear layers (MFMA GEMM, fused epilogues)
def linear(x, w, b, relu: bool = False, residual=None, out_f32: bool = False):
    """x [N,Cin] in the compute dtype; returns compute dtype, or fp32 when out_f32 (then `residual` fp32 is added)."""
    return ops.linear(x, w, b, relu=relu, residual=residual, out_f32=out_f32)


# --------------------------------------------------------------------------- LayerNorm (+pe)
class LayerNormFn(torch.autograd.Function):
    """y = LN(x)*g + b (+pe).  with_residual: also returns x itself as a second output for the residual branch that by-passes the
    LayerNorm (x -> LN -> sublayer -> + x, transformer.py:238-293); its gradient then arrives HERE and is added inside the LayerNorm
    backward kernel instead of by a separate autograd accumulation launch."""

    @staticmethod
    def forward(ctx, x, g, b, pe, out_dtype, with_residual=False):
        lib = L.load()
        x = x.contiguous()
        n = x.shape[0]
        y = torch.empty(n, 256, dtype=out_dtype, device=x.device)
        stats = torch.empty(n, 2, dtype=torch.float32, device=x.device)
        L.check(lib.dreg_layernorm_fwd(L.ptr(x), L.ptr(g.detach()), L.ptr(b.detach()), L.ptr(pe), L.ptr(y), L.ptr(stats),
                                       n, 256, 1e-5, L.dt_of(y), L.stream()), "dreg_layernorm_fwd")
        ctx.save_for_backward(x, g, stats)
        ctx.b_ref = b if b.is_leaf else None
        ctx.pe_dtype = pe.dtype if (pe is not None and pe.requires_grad) else None   # learned position embedding: d pe = gy
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if with_residual else y

    @staticmethod
    def backward(ctx, gy, gres=None):
        x, g, stats = ctx.saved_tensors
        lib = L.load()
        n = x.shape[0]
        if gy is None:                      # only the residual branch carried a gradient
            return gres, None, None, None, None, None
        gy = gy.contiguous()
        gpe = gy.to(ctx.pe_dtype) if ctx.pe_dtype is not None else None
        if gres is not None:
            gres = gres.contiguous().float()
        dx = torch.empty_like(x)
        # gamma / beta gradients straight into the preallocated .grad buffers when both exist (one [256] add_ launch less per
        # parameter and LayerNorm call: 40 per step); otherwise fresh tensors for autograd to accumulate
        gs, bs = ops._grad_sink(g), (ops._grad_sink(ctx.b_ref) if ctx.b_ref is not None else None)
        direct = gs is not None and bs is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        dg = gs if direct else torch.empty(256, dtype=torch.float32, device=x.device)
        db = bs if direct else torch.empty(256, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.dreg_layernorm_bwd_workspace_bytes(n) // 4 + 4, dtype=torch.float32, device=x.device)
        L.check(lib.dreg_layernorm_bwd_add(L.ptr(x), L.ptr(gy), L.ptr(g.detach()), L.ptr(stats), L.ptr(dx), L.ptr(gres), L.ptr(dg), L.ptr(db),
                                           L.ptr(ws), n, 256, L.dt_of(gy), int(direct), L.stream()), "dreg_layernorm_bwd_add")
        return dx, (None if direct else dg), (None if direct else db), gpe, None, None


def layer_norm(x, w, b, pe=None, out_dtype=None):
    """x fp32 [N,256] -> LN(x)*w + b (+pe) in out_dtype (default: the compute dtype)."""
    return LayerNormFn.apply(x, w, b, pe, out_dtype or _COMPUTE_DTYPE)


def layer_norm_residual(x, w, b, pe=None, out_dtype=None):
    """(LN(x)*w + b (+pe), x): the second output is x for the residual add that follows the sublayer (see LayerNormFn)."""
    return LayerNormFn.apply(x, w, b, pe, out_dtype or _COMPUTE_DTYPE, True)


# --------------------------------------------------------------------------- multi-head attention core
class _MHAFn(torch.autograd.Function):
    """q_src [Nq, ldq] holds q at column q_off; kv_src [Nk, ldk] holds k at k_off and v at v_off (packed projections)."""

    @staticmethod
    def forward(ctx, q_src, kv_src, q_off, k_off, v_off, n_heads, scale):
        lib = L.load()
        nq, nk = q_src.shape[0], kv_src.shape[0]
        es = q_src.element_size()
        o = torch.empty(nq, 32 * n_heads, dtype=q_src.dtype, device=q_src.device)
        lse = torch.empty(n_heads, nq, dtype=torch.float32, device=q_src.device)
        L.check(lib.dreg_mha_fwd(q_src.data_ptr() + q_off * es, kv_src.data_ptr() + k_off * es, kv_src.data_ptr() + v_off * es,
                                 L.ptr(o), L.ptr(lse), nq, nk, n_heads, q_src.shape[1], kv_src.shape[1], kv_src.shape[1],
                                 o.shape[1], scale, L.dt_of(q_src), L.stream()), "dreg_mha_fwd")
        ctx.save_for_backward(q_src, kv_src, o, lse)
        ctx.cfg = (q_off, k_off, v_off, n_heads, scale)
        return o

    @staticmethod
    def backward(ctx, go):
        q_src, kv_src, o, lse = ctx.saved_tensors
        q_off, k_off, v_off, n_heads, scale = ctx.cfg
        lib = L.load()
        go = go.contiguous()
        nq, nk = q_src.shape[0], kv_src.shape[0]
        es = q_src.element_size()
        same = q_src.data_ptr() == kv_src.data_ptr() and nq == nk
        e = 32 * n_heads
        if same:  # packed (q|k|v) of one point set: every column is written
            dq_src = torch.empty_like(q_src)
            dkv_src = dq_src
        else:     # row slices of a packed projection: the columns this call does not own stay zero
            dq_src = torch.zeros_like(q_src) if q_src.shape[1] > e else torch.empty_like(q_src)
            dkv_src = torch.zeros_like(kv_src) if kv_src.shape[1] > 2 * e else torch.empty_like(kv_src)
        dvec = torch.empty(n_heads, nq, dtype=torch.float32, device=q_src.device)
        L.check(lib.dreg_mha_bwd(q_src.data_ptr() + q_off * es, kv_src.data_ptr() + k_off * es, kv_src.data_ptr() + v_off * es,
                                 L.ptr(o), L.ptr(go), L.ptr(lse), L.ptr(dvec),
                                 dq_src.data_ptr() + q_off * es, dkv_src.data_ptr() + k_off * es, dkv_src.data_ptr() + v_off * es,
                                 nq, nk, n_heads, q_src.shape[1], kv_src.shape[1], kv_src.shape[1], o.shape[1], scale,
                                 L.dt_of(q_src), L.stream()), "dreg_mha_bwd")
        if same:
            return dq_src, None, None, None, None, None, None
        return dq_src, dkv_src, None, None, None, None, None


def mha_packed(q_rows, kv_rows, n_heads: int, scale: float):
    """q_rows [Nq, 3E], kv_rows [Nk, 3E]: row slices of a packed (q | k | v) projection -> [Nq, E].
    Self-attention passes the same slice twice; cross-attention passes the two point sets' slices."""
    e = q_rows.shape[1] // 3
    return _MHAFn.apply(q_rows, kv_rows, 0, e, 2 * e, n_heads, scale)


class ProblemTable:
    """Row ranges of the attention problems of one step: every pair i contributes a source segment and a target
    segment of the shared row space.  self: q = kv = segment; cross: q = one segment, kv = the pair's other one."""

    def __init__(self, seg_lengths, device):
        """seg_lengths: [(ns_0, nt_0), (ns_1, nt_1), ...] in row order."""
        starts, off = [], 0
        for ns, nt in seg_lengths:
            starts.append((off, ns, off + ns, nt))
            off += ns + nt
        self.R = off
        self.segs = starts
        self_p, cross_p = [], []
        for (s0, ns, t0, nt) in starts:
            self_p += [[s0, ns, s0, ns], [t0, nt, t0, nt]]
            cross_p += [[s0, ns, t0, nt], [t0, nt, s0, ns]]
        device = torch.device(device)
        self.max_len = max(max(ns, nt) for ns, nt in seg_lengths)
        # per-pair tables of the batched Kabsch / loss kernels: (s0, ns, t0, nt); source-row and logits-block offsets
        so, lo, a, b = [], [], 0, 0
        for (s0, ns, t0, nt) in starts:
            so.append(a)
            lo.append(b)
            a += ns
            b += ns * ((nt + 3) // 4 * 4)          # a pair's logits block is [ns][nt rounded up to 4 floats] (csrc/losses.hip: the GEMMs' vector loads)
        self.total_src, self.total_logits = a, b
        self.logit_off_host = lo
        # ONE upload for the five tables (each was its own small copy in front of the transformer): int64 offsets first (alignment),
        # then the int32 tables; no pageable H2D copy: that would drain the stream
        P, n4 = len(starts), 4 * len(self_p)
        flat = []
        for v in lo:
            flat += [v & 0xFFFFFFFF, v >> 32]
        flat += [x for row in self_p for x in row] + [x for row in cross_p for x in row] + [x for sg in starts for x in sg] + so
        flat = [x - (1 << 32) if x >= (1 << 31) else x for x in flat]
        buf = L.to_device_async(flat, torch.int32, device)
        self.buffer = buf
        o = 2 * P
        self.logit_off = buf[:o].view(torch.int64)
        self.self_probs = buf[o:o + n4].view(-1, 4)
        self.cross_probs = buf[o + n4:o + 2 * n4].view(-1, 4)
        self.pair_probs = buf[o + 2 * n4:o + 2 * n4 + 4 * P].view(-1, 4)
        self.src_off = buf[o + 2 * n4 + 4 * P:]
        self.nprob = len(self_p)


class _MHAVarlenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, probs, nprob, max_len, n_heads, scale):
        lib = L.load()
        R, e3 = qkv.shape
        e = e3 // 3
        es = qkv.element_size()
        o = torch.empty(R, e, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(n_heads, R, dtype=torch.float32, device=qkv.device)
        L.check(lib.dreg_mha_varlen_fwd(qkv.data_ptr(), qkv.data_ptr() + e * es, qkv.data_ptr() + 2 * e * es, L.ptr(o), L.ptr(lse),
                                        L.ptr(probs), nprob, max_len, max_len, R, n_heads, e3, e3, e3, e, scale, L.dt_of(qkv), L.stream()),
                "dreg_mha_varlen_fwd")
        ctx.save_for_backward(qkv, o, lse, probs)
        ctx.cfg = (nprob, max_len, n_heads, scale)
        return o

    @staticmethod
    def backward(ctx, go):
        qkv, o, lse, probs = ctx.saved_tensors
        nprob, max_len, n_heads, scale = ctx.cfg
        lib = L.load()
        go = go.contiguous()
        R, e3 = qkv.shape
        e = e3 // 3
        es = qkv.element_size()
        dqkv = torch.empty_like(qkv)  # every row is a query of one problem and a key/value of one problem
        dvec = torch.empty(n_heads, R, dtype=torch.float32, device=qkv.device)
        L.check(lib.dreg_mha_varlen_bwd(qkv.data_ptr(), qkv.data_ptr() + e * es, qkv.data_ptr() + 2 * e * es, L.ptr(o), L.ptr(go),
                                        L.ptr(lse), L.ptr(dvec), dqkv.data_ptr(), dqkv.data_ptr() + e * es, dqkv.data_ptr() + 2 * e * es,
                                        L.ptr(probs), nprob, max_len, max_len, R, n_heads, e3, e3, e3, e, scale, L.dt_of(qkv), L.stream()),
                "dreg_mha_varlen_bwd")
        return dqkv, None, None, None, None, None


def mha_varlen(qkv, probs, nprob, max_len, n_heads: int, scale: float):
    """qkv [R, 3E] packed projections of all point sets; probs int32 [nprob,4] (ProblemTable) -> [R, E]."""
    return _MHAVarlenFn.apply(qkv, probs, nprob, max_len, n_heads, scale)


class _CorrAttnVarlenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, xyz, probs, nprob, max_len, scale):
        lib = L.load()
        nl, R = q.shape[0], q.shape[1]
        q, k, xyz = q.contiguous(), k.contiguous(), xyz.contiguous()
        out = torch.empty(nl, R, 3, dtype=torch.float32, device=q.device)
        lse = torch.empty(nl, R, dtype=torch.float32, device=q.device)
        L.check(lib.dreg_corr_attention_varlen_fwd(L.ptr(q), L.ptr(k), L.ptr(xyz), L.ptr(out), L.ptr(lse), L.ptr(probs), nprob,
                                                   max_len, max_len, nl, R, scale, L.dt_of(q), L.stream()), "dreg_corr_attention_varlen_fwd")
        ctx.save_for_backward(q, k, xyz, out, lse, probs)
        ctx.cfg = (nprob, max_len, scale)
        return out

    @staticmethod
    def backward(ctx, go):
        q, k, xyz, out, lse, probs = ctx.saved_tensors
        nprob, max_len, scale = ctx.cfg
        lib = L.load()
        nl, R = q.shape[0], q.shape[1]
        go = go.contiguous().float()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        dvec = torch.empty(nl, R, dtype=torch.float32, device=q.device)
        L.check(lib.dreg_corr_attention_varlen_bwd(L.ptr(q), L.ptr(k), L.ptr(xyz), L.ptr(out), L.ptr(go), L.ptr(lse), L.ptr(dvec),
                                                   L.ptr(dq), L.ptr(dk), L.ptr(probs), nprob, max_len, max_len, nl, R, scale,
                                                   L.dt_of(q), L.stream()), "dreg_corr_attention_varlen_bwd")
        return dq, dk, None, None, None, None, None


def attention_xyz_varlen(q, k, xyz, probs, nprob, max_len, scale: float):
    """q, k [L,R,256] (compute dtype), xyz fp32 [R,3]; problem p: softmax(scale q[rows_q] k[rows_kv]^T) xyz[rows_kv] -> fp32 [L,R,3]."""
    return _CorrAttnVarlenFn.apply(q, k, xyz, probs, nprob, max_len, scale)


# --------------------------------------------------------------------------- correspondence attention (V = xyz)
class _CorrAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, xyz, scale):
        lib = L.load()
        nl, nq, nk = q.shape[0], q.shape[1], k.shape[1]
        q, k, xyz = q.contiguous(), k.contiguous(), xyz.contiguous()
        out = torch.empty(nl, nq, 3, dtype=torch.float32, device=q.device)
        lse = torch.empty(nl, nq, dtype=torch.float32, device=q.device)
        L.check(lib.dreg_corr_attention_fwd(L.ptr(q), L.ptr(k), L.ptr(xyz), L.ptr(out), L.ptr(lse), nl, nq, nk, scale,
                                            L.dt_of(q), L.stream()), "dreg_corr_attention_fwd")
        ctx.save_for_backward(q, k, xyz, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, go):
        q, k, xyz, out, lse = ctx.saved_tensors
        lib = L.load()
        nl, nq, nk = q.shape[0], q.shape[1], k.shape[1]
        go = go.contiguous().float()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        dvec = torch.empty(nl, nq, dtype=torch.float32, device=q.device)
        L.check(lib.dreg_corr_attention_bwd(L.ptr(q), L.ptr(k), L.ptr(xyz), L.ptr(out), L.ptr(go), L.ptr(lse), L.ptr(dvec),
                                            L.ptr(dq), L.ptr(dk), nl, nq, nk, ctx.scale, L.dt_of(q), L.stream()),
                "dreg_corr_attention_bwd")
        return dq, dk, None, None


def attention_xyz(q, k, xyz, scale: float):
    """q [L,Nq,256], k [L,Nk,256] (compute dtype), xyz fp32 [Nk,3] -> softmax(scale q k^T) xyz : fp32 [L,Nq,3]."""
    return _CorrAttnFn.apply(q, k, xyz, scale)


# --------------------------------------------------------------------------- overlap head
class _OverlapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, w, b):
        lib = L.load()
        f = f.contiguous()
        n = f.shape[0]
        s = torch.empty(n, dtype=torch.float32, device=f.device)
        L.check(lib.dreg_overlap_fwd(L.ptr(f), L.ptr(w.detach().contiguous()), L.ptr(b.detach()), L.ptr(s), n, L.stream()), "dreg_overlap_fwd")
        ctx.save_for_backward(f, w, s)
        return s

    @staticmethod
    def backward(ctx, gs):
        f, w, s = ctx.saved_tensors
        lib = L.load()
        n = f.shape[0]
        gs = gs.contiguous()
        df = torch.empty_like(f)
        dw = torch.empty(256, dtype=torch.float32, device=f.device)
        db = torch.empty(1, dtype=torch.float32, device=f.device)
        ws = torch.empty(lib.dreg_overlap_bwd_workspace_bytes(n) // 4 + 4, dtype=torch.float32, device=f.device)
        L.check(lib.dreg_overlap_bwd(L.ptr(f), L.ptr(w.detach().contiguous()), L.ptr(s), L.ptr(gs), L.ptr(df), L.ptr(dw), L.ptr(db),
                                     L.ptr(ws), n, L.stream()), "dreg_overlap_bwd")
        return df, dw.view_as(w), db


def overlap_head(f, w, b):
    """f fp32 [..., 256] -> sigmoid(f . w + b) [..., 1]."""
    shp = f.shape[:-1]
    return _OverlapFn.apply(f.reshape(-1, 256), w, b).view(*shp, 1)


# --------------------------------------------------------------------------- position embedding / downsample / Kabsch
def posenc_sine(xyz, d_model=256, temperature=1000.0, scale=1.0):
    lib = L.load()
    assert d_model == 256 and xyz.shape[-1] == 3
    xyz = xyz.detach().contiguous().float()
    pe = torch.empty(xyz.shape[0], 256, dtype=torch.float32, device=xyz.device)
    L.check(lib.dreg_posenc_sine(L.ptr(xyz), L.ptr(pe), xyz.shape[0], float(scale), float(temperature), L.stream()), "dreg_posenc_sine")
    return pe


class _VoxelMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, feats, pt_batch, nbatch, dl):
        lib = L.load()
        n, c = feats.shape
        dev = feats.device
        points, feats = points.contiguous(), feats.contiguous()
        out_p = torch.empty(n, 3, dtype=torch.float32, device=dev)
        out_f = torch.empty(n, c, dtype=torch.float32, device=dev)
        meta = torch.zeros(2 + nbatch, dtype=torch.int32, device=dev)  # n_out, err, counts...
        inv_seg = torch.empty(n, dtype=torch.int32, device=dev)
        inv_cnt = torch.empty(n, dtype=torch.float32, device=dev)
        nbytes = lib.dreg_voxel_downsample_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.dreg_voxel_downsample_fwd(L.ptr(points), L.ptr(feats), L.ptr(pt_batch), L.ptr(out_p), L.ptr(out_f),
                                              meta.data_ptr(), meta.data_ptr() + 8, L.ptr(inv_seg), L.ptr(inv_cnt), meta.data_ptr() + 4,
                                              L.ptr(ws), nbytes, n, c, nbatch, float(dl), L.stream()), "dreg_voxel_downsample_fwd")
        host = meta.tolist()  # one host sync per round, as the reference's .shape[0] test (grid_downsample.py:91)
        if host[1]:
            raise L.DregError("voxel coordinates overflow the 16-bit cell range")
        m = host[0]
        ctx.save_for_backward(inv_seg, inv_cnt)
        ctx.n = n
        ctx.mark_non_differentiable(out_p)
        counts = torch.tensor(host[2:], dtype=torch.int64)
        return out_p[:m], out_f[:m], counts

    @staticmethod
    def backward(ctx, gp, gf, gc):
        inv_seg, inv_cnt = ctx.saved_tensors
        lib = L.load()
        gf = gf.contiguous()
        gin = torch.empty(ctx.n, gf.shape[1], dtype=torch.float32, device=gf.device)
        L.check(lib.dreg_voxel_downsample_bwd(L.ptr(gf), L.ptr(inv_seg), L.ptr(inv_cnt), L.ptr(gin), ctx.n, gf.shape[1], L.stream()),
                "dreg_voxel_downsample_bwd")
        return None, gin, None, None, None


def voxel_mean_downsample(points, feats, lengths, dl: float):
    """lengths: python list / CPU tensor of per-batch row counts (rows grouped by batch)."""
    lens = [int(v) for v in lengths]
    dev = feats.device
    pt_batch = torch.repeat_interleave(torch.arange(len(lens), dtype=torch.int32, device=dev),
                                       L.to_device_async(lens, torch.int64, dev))
    p, f, counts = _VoxelMeanFn.apply(points, feats, pt_batch, len(lens), dl)
    return p, f, counts


class SubsampleRound:
    """One planned voxel-average round: everything that depends only on xyz (dreg_voxel_downsample_plan)."""
    __slots__ = ("order", "starts", "n_out_dev", "inv_seg", "inv_cnt", "n_in", "n_out")


def plan_voxel_downsample(points, lengths, dl: float, frozen=None):
    """xyz-only half of voxel_mean_downsample: returns (round, averaged points [M,3], per-batch counts list).
    frozen (list of bool per batch, optional): those batches pass through unchanged and in order (dreg_voxel_downsample_plan_frozen) — how one round serves
    all pairs of a step although some of them have stopped subsampling (transformer_ops.plan_hierarchical_subsample_all)."""
    lib = L.load()
    lens = [int(v) for v in lengths]
    dev = points.device
    n = points.shape[0]
    pt_batch = torch.repeat_interleave(torch.arange(len(lens), dtype=torch.int32, device=dev),
                                       L.to_device_async(lens, torch.int64, dev), output_size=n)
    points = points.detach().contiguous()
    r = SubsampleRound()
    out_p = torch.empty(n, 3, dtype=torch.float32, device=dev)
    meta = torch.zeros(2 + len(lens), dtype=torch.int32, device=dev)  # n_out, err, counts...
    r.inv_seg = torch.empty(n, dtype=torch.int32, device=dev)
    r.inv_cnt = torch.empty(n, dtype=torch.float32, device=dev)
    r.order = torch.empty(n, dtype=torch.int32, device=dev)
    r.starts = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nbytes = lib.dreg_voxel_downsample_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fz = L.to_device_async([int(bool(f)) for f in frozen], torch.uint8, dev) if (frozen is not None and any(frozen)) else None
    L.check(lib.dreg_voxel_downsample_plan_frozen(L.ptr(points), L.ptr(pt_batch), L.ptr(out_p), meta.data_ptr(), meta.data_ptr() + 8,
                                                  L.ptr(r.inv_seg), L.ptr(r.inv_cnt), meta.data_ptr() + 4, L.ptr(r.order), L.ptr(r.starts),
                                                  L.ptr(ws), nbytes, n, len(lens), float(dl), L.ptr(fz), L.stream()), "dreg_voxel_downsample_plan_frozen")
    host = meta.tolist()  # the host sync of the round (the reference's .shape[0] test, grid_downsample.py:91)
    if host[1]:
        raise L.DregError("voxel coordinates overflow the 16-bit cell range")
    r.n_out_dev = meta
    r.n_in, r.n_out = n, host[0]
    return r, out_p[:r.n_out], host[2:]


class _SegmentMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, rnd):
        lib = L.load()
        feats = feats.contiguous()
        assert feats.shape[0] == rnd.n_in and feats.dtype == torch.float32
        c = feats.shape[1]
        out = torch.empty(rnd.n_out, c, dtype=torch.float32, device=feats.device)
        L.check(lib.dreg_voxel_segment_mean(L.ptr(feats), L.ptr(rnd.order), L.ptr(rnd.starts), L.ptr(rnd.n_out_dev), L.ptr(out),
                                            rnd.n_out, c, L.stream()), "dreg_voxel_segment_mean")
        ctx.rnd = rnd
        return out

    @staticmethod
    def backward(ctx, gf):
        rnd = ctx.rnd
        lib = L.load()
        gf = gf.contiguous()
        gin = torch.empty(rnd.n_in, gf.shape[1], dtype=torch.float32, device=gf.device)
        L.check(lib.dreg_voxel_downsample_bwd(L.ptr(gf), L.ptr(rnd.inv_seg), L.ptr(rnd.inv_cnt), L.ptr(gin), rnd.n_in, gf.shape[1], L.stream()),
                "dreg_voxel_downsample_bwd")
        return gin, None


def segment_mean(feats, rnd: SubsampleRound):
    return _SegmentMeanFn.apply(feats, rnd)


class _SubsampleAllFn(torch.autograd.Function):
    """The voxel-average rounds of EVERY pair of a step as one autograd node (grid_downsample.py:6-94 applied per pair, nerf_regtr.py:150-168):
    feats fp32 [N_total, C] (the pairs' point sets one after the other, sizes[i] rows each), plans[i] = that pair's rounds.  Same launches
    as segment_mean per pair and round, but the last round of every pair writes straight into its rows of the joint output and — backward
    — the first round's gradient straight into its rows of d(feats): no torch.cat of the per-pair results (forward) and no concatenation
    of the per-pair gradients (backward: 150 MB copied per step at 4 pairs)."""

    @staticmethod
    def forward(ctx, feats, plans, sizes):
        lib = L.load()
        feats = feats.contiguous()
        assert feats.dtype == torch.float32 and feats.shape[0] == sum(sizes)
        c = feats.shape[1]
        n_out = [(rounds[-1].n_out if rounds else sz) for rounds, sz in zip(plans, sizes)]
        out = torch.empty(sum(n_out), c, dtype=torch.float32, device=feats.device)
        io, oo = 0, 0
        for rounds, sz, no in zip(plans, sizes, n_out):
            x = feats[io:io + sz]
            if not rounds:
                out[oo:oo + no].copy_(x)
            for k, rnd in enumerate(rounds):
                assert x.shape[0] == rnd.n_in
                y = out[oo:oo + no] if k == len(rounds) - 1 else torch.empty(rnd.n_out, c, dtype=torch.float32, device=feats.device)
                L.check(lib.dreg_voxel_segment_mean(L.ptr(x), L.ptr(rnd.order), L.ptr(rnd.starts), L.ptr(rnd.n_out_dev), L.ptr(y),
                                                    rnd.n_out, c, L.stream()), "dreg_voxel_segment_mean")
                x = y
            io += sz
            oo += no
        ctx.plans, ctx.sizes, ctx.n_out = plans, sizes, n_out
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        g = g.contiguous()
        c = g.shape[1]
        gin_all = torch.empty(sum(ctx.sizes), c, dtype=torch.float32, device=g.device)
        io, oo = 0, 0
        for rounds, sz, no in zip(ctx.plans, ctx.sizes, ctx.n_out):
            gy = g[oo:oo + no]
            if not rounds:
                gin_all[io:io + sz].copy_(gy)
            for k in range(len(rounds) - 1, -1, -1):
                rnd = rounds[k]
                gx = gin_all[io:io + sz] if k == 0 else torch.empty(rnd.n_in, c, dtype=torch.float32, device=g.device)
                L.check(lib.dreg_voxel_downsample_bwd(L.ptr(gy), L.ptr(rnd.inv_seg), L.ptr(rnd.inv_cnt), L.ptr(gx), rnd.n_in, c, L.stream()),
                        "dreg_voxel_downsample_bwd")
                gy = gx
            io += sz
            oo += no
        return gin_all, None, None


class _GatherSubsampleFn(torch.autograd.Function):
    """ops.trilinear_gather followed by subsample_all as ONE autograd node (nerf_regtr.py:138-168): forward = the first round of every pair
    straight from the feature map (dreg_gather_segment_mean), the other rounds as subsample_all; backward:
    the voxel-average rounds are differentiated down to the FIRST round's outputs only, and the gather's backward reads the gradient of
    the gathered features through that round (d feats[n] = g1[inv_seg[n]] * inv_cnt[n]: dreg_trilinear_gather_bwd_gather_seg) — the
    [N_total, 256] fp32 gradient (150 MB at 4 pairs) is never written, the gather reads rows of a tensor a sixth of its size.  Same
    arithmetic in the same order as the two nodes: bit-identical gradients (tests/test_hip_pointset_ops.py)."""

    @staticmethod
    def forward(ctx, p1, idx, pt_batch, fine_res, rows1, plans, sizes):
        lib = L.load()
        p1 = p1.contiguous()
        B, d, h, w, C = p1.shape
        Zr, Xr, Yr = fine_res
        dev = p1.device
        n_out = [rounds[-1].n_out for rounds in plans]
        out = torch.empty(sum(n_out), C, dtype=torch.float32, device=dev)
        io, oo = 0, 0
        for rounds, sz, no in zip(plans, sizes, n_out):
            # round 0 straight from p1 (dreg_gather_segment_mean: the [N_total, C] gathered features are never written), the others as subsample_all
            x = None
            for k, rnd in enumerate(rounds):
                y = out[oo:oo + no] if k == len(rounds) - 1 else torch.empty(rnd.n_out, C, dtype=torch.float32, device=dev)
                if k == 0:
                    assert rnd.n_in == sz
                    L.check(lib.dreg_gather_segment_mean(L.ptr(p1), L.ptr(idx), L.ptr(pt_batch), io, L.ptr(rnd.order), L.ptr(rnd.starts), L.ptr(rnd.n_out_dev),
                                                         L.ptr(y), rnd.n_out, d, h, w, C, Zr, Xr, Yr, L.dt_of(p1), L.stream()), "dreg_gather_segment_mean")
                else:
                    L.check(lib.dreg_voxel_segment_mean(L.ptr(x), L.ptr(rnd.order), L.ptr(rnd.starts), L.ptr(rnd.n_out_dev), L.ptr(y),
                                                        rnd.n_out, C, L.stream()), "dreg_voxel_segment_mean")
                x = y
            io += sz
            oo += no
        ctx.save_for_backward(idx, pt_batch, rows1, None)
        ctx.cfg = (tuple(p1.shape), p1.dtype, fine_res)
        ctx.plans, ctx.sizes, ctx.n_out = plans, sizes, n_out
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        lib = L.load()
        idx, pt_batch, rows1, _ = ctx.saved_tensors
        shape, dtype, (Zr, Xr, Yr) = ctx.cfg
        B, d, h, w, C = shape
        g = g.contiguous().float()
        dev = g.device
        n1 = [rounds[0].n_out for rounds in ctx.plans]
        g1 = torch.empty(sum(n1), C, dtype=torch.float32, device=dev)
        oo, ro = 0, 0
        table = []
        pstart = 0
        for i, (rounds, sz, no) in enumerate(zip(ctx.plans, ctx.sizes, ctx.n_out)):
            gy = g[oo:oo + no]
            if len(rounds) == 1:
                g1[ro:ro + n1[i]].copy_(gy)
            for k in range(len(rounds) - 1, 0, -1):
                rnd = rounds[k]
                gx = g1[ro:ro + n1[i]] if k == 1 else torch.empty(rnd.n_in, C, dtype=torch.float32, device=dev)
                L.check(lib.dreg_voxel_downsample_bwd(L.ptr(gy), L.ptr(rnd.inv_seg), L.ptr(rnd.inv_cnt), L.ptr(gx), rnd.n_in, C, L.stream()),
                        "dreg_voxel_downsample_bwd")
                gy = gx
            row = [rounds[0].inv_seg.data_ptr(), rounds[0].inv_cnt.data_ptr(), pstart, ro]
            table += [row] * (B // len(ctx.plans))      # the grids this plan covers: both grids of a pair (per-pair plans) or every grid of the step (one global plan)
            oo += no
            ro += n1[i]
            pstart += sz
        assert len(table) == B, "one plan per pair of grids, or one for all of them"
        descs = L.to_device_async([v for r in table for v in r], torch.int64, dev)
        fmap = torch.empty(B * Zr * Xr * Yr, dtype=torch.int32, device=dev)
        if ops.PERSISTENT_GRAD_BUFFERS:          # the dense gradient buffer kept across steps, zero outside the rows of the step that wrote it (ops.TrilinearGatherFn)
            key = (shape, dtype, dev)
            ent = ops._DP1_CACHE.get(key)
            if ent is not None and ent[2]:
                ent = [torch.zeros(shape, dtype=dtype, device=dev), None, True]
            elif ent is None:
                while len(ops._DP1_CACHE) >= 2:
                    ops._DP1_CACHE.pop(next(iter(ops._DP1_CACHE)))
                ent = ops._DP1_CACHE[key] = [torch.zeros(shape, dtype=dtype, device=dev), None, False]
            ent[2] = True
            gp1, dirty = ent[0], ent[1]
            if dirty is not None:
                L.check(lib.dreg_zero_rows(L.ptr(gp1), L.ptr(dirty), dirty.shape[0], C, L.dt_of(gp1), L.stream()), "dreg_zero_rows")
            ent[1] = rows1
            zero_dense = 0
        else:
            gp1 = torch.empty(shape, dtype=dtype, device=dev)
            zero_dense = 1
        L.check(lib.dreg_trilinear_gather_bwd_gather_seg(L.ptr(g1), L.ptr(descs), L.ptr(idx), L.ptr(pt_batch), L.ptr(rows1), rows1.shape[0], L.ptr(fmap),
                                                         L.ptr(gp1), idx.shape[0], B, d, h, w, C, Zr, Xr, Yr, L.dt_of(gp1), zero_dense, L.stream()),
                "dreg_trilinear_gather_bwd_gather_seg")
        return gp1, None, None, None, None, None, None


def gather_subsample_applies(p1, rows1, plans) -> bool:
    """The fused node needs the gather's deterministic per-S1-voxel backward (C a multiple of 64 up to 256, a row list) and at least one
    voxel-average round in every pair."""
    return p1.is_cuda and rows1 is not None and p1.shape[-1] % 64 == 0 and p1.shape[-1] <= 256 and all(len(r) >= 1 for r in plans)


def gather_subsample(p1, idx, pt_batch, fine_res, rows1, plans, sizes):
    """[B,d,h,w,C] -> [sum of the pairs' key points, C] fp32: trilinear gather at the occupied fine voxels + every pair's voxel-average rounds."""
    return _GatherSubsampleFn.apply(p1, idx, pt_batch, fine_res, rows1, plans, sizes)


def subsample_all(feats, plans, sizes):
    """[N_total, C] -> [sum of the pairs' key points, C]: every pair's voxel-average rounds, one autograd node."""
    return _SubsampleAllFn.apply(feats, plans, sizes)


def weighted_kabsch(a, b, w, eps: float = 1e-6):
    """a,b [P,N,3], w [P,N] -> [P,3,4] (no gradient: the pose enters no loss, train_nerf_regtr.py:186-228)."""
    lib = L.load()
    a, b, w = a.detach().contiguous().float(), b.detach().contiguous().float(), w.detach().contiguous().float()
    out = torch.empty(a.shape[0], 3, 4, dtype=torch.float32, device=a.device)
    L.check(lib.dreg_weighted_kabsch(L.ptr(a), L.ptr(b), L.ptr(w), L.ptr(out), a.shape[0], a.shape[1], eps, L.stream()),
            "dreg_weighted_kabsch")
    return out


def weighted_kabsch_pairs(xyz, corr, ov, tab: "ProblemTable", eps: float = 1e-6):
    """All (pair, layer) solves of a step in one launch: xyz [R,3], corr [L,R,3], ov [L,R(,1)] -> [P,L,3,4] (no gradient)."""
    lib = L.load()
    xyz, corr = xyz.detach().contiguous().float(), corr.detach().contiguous().float()
    ov = ov.detach().reshape(corr.shape[0], -1).contiguous().float()
    P_, Ln, R = len(tab.segs), corr.shape[0], xyz.shape[0]
    out = torch.empty(P_, Ln, 3, 4, dtype=torch.float32, device=xyz.device)
    L.check(lib.dreg_weighted_kabsch_pairs(L.ptr(xyz), L.ptr(corr), L.ptr(ov), L.ptr(tab.pair_probs), L.ptr(out), P_, Ln, R, eps, L.stream()),
            "dreg_weighted_kabsch_pairs")
    return out

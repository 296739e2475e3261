"""Native (C++) execution of the point-set half of the registration network — the 6-layer self / cross attention encoder, the shared
final norm, the correspondence decoder and the overlap head — through ``csrc/pointset_exec.hip``: ONE C call for the forward pass,
ONE for the backward pass (parameter gradients accumulated straight into FlatAdamW's buffers, weight / bias gradients on the
process-wide second stream).

Reference network: conerf/register/transformer.py:50-86,225-299, conerf/register/nerf_regtr.py:170-206,273-308,350-394.
``transformer_ops.encode_decode_batched`` stays the description of record (and the fp32 parity path); this module only changes who
issues the launches (and, with ``fuse``, folds a few element-wise passes into their producers): with ``fuse = 0`` the two are equal
bit for bit, outputs and every gradient (tests/test_hip_pointset_exec.py).
"""
import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import lib as L
from . import ops

N_LAYERS = 6
_PER_LAYER = ("norm1.weight", "norm1.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
              "self_attn.out_proj.bias", "norm2.weight", "norm2.bias", "cross_attn.in_proj_weight", "cross_attn.in_proj_bias",
              "cross_attn.out_proj.weight", "cross_attn.out_proj.bias", "norm3.weight", "norm3.bias", "linear1.weight", "linear1.bias",
              "linear2.weight", "linear2.bias")
_TAIL = ("transformer_encoder.norm.weight", "transformer_encoder.norm.bias", "correspondence_decoder.q_proj.weight",
         "correspondence_decoder.q_proj.bias", "correspondence_decoder.k_proj.weight", "correspondence_decoder.k_proj.bias",
         "correspondence_decoder.conf_logits_decoder.weight", "correspondence_decoder.conf_logits_decoder.bias")
_LINEARS = ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "cross_attn.in_proj_weight", "cross_attn.out_proj.weight",
            "linear1.weight", "linear2.weight")

FUSE = True      # False: the per-op path's arithmetic, bit for bit (tests; dreg_ps_set_fuse)
GROUP_WGRAD = True   # False: one weight-gradient launch per linear layer instead of one per tile shape (A/B; dreg_ps_set_group_wgrad, per handle, bit-identical)


def param_names():
    names = [f"transformer_encoder.layers.{l}.{n}" for l in range(N_LAYERS) for n in _PER_LAYER]
    return names + list(_TAIL)


def linear_names():
    names = [f"transformer_encoder.layers.{l}.{n}" for l in range(N_LAYERS) for n in _LINEARS]
    return names + ["correspondence_decoder.q_proj.weight", "correspondence_decoder.k_proj.weight"]


class PointSetExecutor:
    """The recorded parameter table + the arena of one model (grown on demand: the row count changes every step)."""

    def __init__(self, P: Dict[str, torch.Tensor], with_grad: bool):
        self.lib = L.load()
        names = param_names()
        assert len(names) == self.lib.dreg_ps_num_params()
        self.params = [P[n] for n in names]
        self.linears = [P[n] for n in linear_names()]
        assert len(self.linears) == self.lib.dreg_ps_num_linears()
        self.with_grad = with_grad
        tab = np.zeros((len(names), 2), dtype=np.int64)
        self._sig = []
        for i, t in enumerate(self.params):
            assert t.is_contiguous() and t.dtype == torch.float32, "the executor reads fp32 master parameters in place"
            g = t.grad if with_grad else None
            if with_grad and (g is None or not g.is_contiguous() or g.dtype != torch.float32):
                raise L.DregError("PointSetExecutor needs preallocated contiguous fp32 .grad buffers (FlatAdamW) on every parameter")
            tab[i, 0] = t.data_ptr()
            tab[i, 1] = g.data_ptr() if g is not None else 0
            self._sig.append((t.data_ptr(), int(tab[i, 1])))
        self.h = self.lib.dreg_ps_create(tab.ctypes.data)
        if not self.h:
            raise L.DregError("dreg_ps_create failed")
        self.lib.dreg_ps_set_fuse(self.h, int(FUSE))
        self.lib.dreg_ps_set_group_wgrad(self.h, int(GROUP_WGRAD))
        self._fuse = FUSE
        self._group = GROUP_WGRAD
        self.device = self.params[0].device
        self.arena = None
        self.arena_bytes = 0
        self._timing = False
        self._packs = (ctypes.c_int64 * (2 * len(self.linears)))()
        self._pack_stamp = None
        self._pack_keep = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dreg_ps_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def still_valid(self) -> bool:
        for t, (vp, gp) in zip(self.params, self._sig):
            g = t.grad if self.with_grad else None
            if t.data_ptr() != vp or (g.data_ptr() if g is not None else 0) != gp:
                return False
        return True

    def _pack_table(self):
        """Device pointers of the (forward, data-gradient) bf16 packs of every linear layer: the packs of ops.packed_weight's cache
        (refreshed by ONE batched launch after every optimizer update: ops.repack_all), looked up again only when a weight changed."""
        stamp = (ops._weight_generation, sum(t._version for t in self.linears))
        if stamp != self._pack_stamp:
            keep = []
            for i, w in enumerate(self.linears):
                f = ops.packed_weight(w, w.shape[1], False, L.DT_BF16)
                d = ops.packed_weight(w, w.shape[1], True, L.DT_BF16)
                self._packs[2 * i], self._packs[2 * i + 1] = f.data_ptr(), d.data_ptr()
                keep += [f, d]
            self._pack_keep, self._pack_stamp = keep, stamp
        return ctypes.addressof(self._packs)

    def _arena_for(self, R: int):
        need = self.lib.dreg_ps_arena_bytes(self.h, R)
        if need > self.arena_bytes:
            self.arena = None                                             # release before growing (15 KB per row: ~150 MB at 10^4 rows)
            self.arena_bytes = int(need * 1.25)
            self.arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device)
        return self.arena

    def set_timing(self, on: bool):
        """HIP-event brackets around every linear-layer launch (bench.py's bracketed step).  While they are on, the weight-gradient
        launches stay on the caller's stream, so a bracket measures its kernel alone (as the trunk executor does)."""
        self._timing = bool(on)
        self.lib.dreg_ps_set_timing(self.h, int(on))

    def drain_timings(self, profiler: "ops.KernelTimer"):
        """After a device synchronisation: the executor's records as (instantiation name, shape label, FLOPs, ms) — names from the
        library's own dispatch rules (dreg_conv3d_igemm_variant / dreg_conv3d_wgrad_variant), i.e. the rows rocprofv3 prints."""
        cap = 4096
        info = (ctypes.c_int * (5 * cap))()
        ms = (ctypes.c_float * cap)()
        n = self.lib.dreg_ps_read_timings(self.h, info, ms, cap)
        for i in range(n):
            kind, rows, cin, cout, flags = (info[5 * i + k] for k in range(5))
            fl = 2.0 * rows * cin * cout
            if kind == 2:
                var = self.lib.dreg_conv3d_wgrad_variant(rows, 1, 1, 1, cin, cout, 1, 0, 0, 0)
                name = "conv_wgrad_glds_kernel<256,256,false,8>" if var == 256256 else f"conv_wgrad_glds_kernel<{var // 1000},{var % 1000},false,4>"
                label = f"wgrad B{rows} 1x1x1x{cin} g1x1x1x{cout} k1s1"
            elif kind == 1:     # data gradient: the transposed problem (cout -> cin)
                name = ops.igemm_kernel_name(self.lib, rows, 1, 1, 1, cout, 1, 1, 1, cin, 1, 1, 0, 1, 0, bool(flags & 4), bool(flags & 1), L.DT_BF16, False)
                label = f"dgrad B{rows} 1x1x1x{cout}->1x1x1x{cin} k1s1"
            else:
                name = ops.igemm_kernel_name(self.lib, rows, 1, 1, 1, cin, 1, 1, 1, cout, 1, 1, 0, 0, 0, bool(flags & 4), bool(flags & 1), L.DT_BF16, bool(flags & 2))
                label = f"fwd B{rows} 1x1x1x{cin}->1x1x1x{cout} k1s1"
            profiler.add_measured(name, label, fl, ms[i])

    def forward(self, feats, xyz, pe, tab):
        if self._fuse != FUSE:
            self.lib.dreg_ps_set_fuse(self.h, int(FUSE))
            self._fuse = FUSE
        if self._group != GROUP_WGRAD:
            self.lib.dreg_ps_set_group_wgrad(self.h, int(GROUP_WGRAD))
            self._group = GROUP_WGRAD
        want = ops.PROFILER is not None and ops.PROFILER.enabled
        if want != self._timing:
            self.set_timing(want)
        ops.wait_packs()
        R = feats.shape[0]
        arena = self._arena_for(R)
        dev = feats.device
        cond = torch.empty(N_LAYERS, R, 256, dtype=torch.float32, device=dev)
        corr = torch.empty(N_LAYERS, R, 3, dtype=torch.float32, device=dev)
        ov = torch.empty(N_LAYERS, R, 1, dtype=torch.float32, device=dev)
        L.check(self.lib.dreg_ps_forward(self.h, L.ptr(arena), self.arena_bytes, self._pack_table(), L.ptr(feats), L.ptr(xyz), L.ptr(pe),
                                         L.ptr(tab.self_probs), L.ptr(tab.cross_probs), tab.nprob, tab.max_len, R,
                                         L.ptr(cond), L.ptr(corr), L.ptr(ov), L.stream()), "dreg_ps_forward")
        self.generation = getattr(self, "generation", 0) + 1      # the arena now holds THIS pass's activations (see backward)
        return cond, corr, ov

    def backward(self, feats, xyz, pe, tab, cond, corr, ov, g_cond, g_corr, g_ov, last_only=False, generation=None):
        if generation is not None and generation != getattr(self, "generation", 0):
            raise L.DregError("PointSetExecutor.backward: the shared arena was overwritten by a later forward pass (two grad-mode forwards before a "
                              "backward: gradient accumulation over several forwards / retain_graph need one executor per live graph)")
        R = feats.shape[0]
        d_feats = torch.empty_like(feats)
        aux = None if self._timing else ops.PARAM_GRAD_STREAM
        if aux is not None:
            self.arena.record_stream(aux)
        L.check(self.lib.dreg_ps_backward(self.h, L.ptr(self.arena), self.arena_bytes, self._pack_table(), L.ptr(feats), L.ptr(xyz), L.ptr(pe),
                                          L.ptr(tab.self_probs), L.ptr(tab.cross_probs), tab.nprob, tab.max_len, R,
                                          L.ptr(cond), L.ptr(corr), L.ptr(ov), L.ptr(g_cond), L.ptr(g_corr), L.ptr(g_ov), L.ptr(d_feats),
                                          L.stream(), aux.cuda_stream if aux is not None else None, int(last_only)), "dreg_ps_backward")
        return d_feats


class _PointSetFn(torch.autograd.Function):
    """(cond, corr, ov, cond_last, corr_last, ov_last) = point-set half(feats): all six layers' outputs and, as tensors of their own, the
    LAST layer's — what the training losses read (train_nerf_regtr.py:178,195,205-206,214,220).  When only the *_last outputs receive a
    gradient the backward pass differentiates heads, decoder and final norm for that layer's rows alone.  `anchor` (any trainable
    parameter) keeps the node in the graph when feats itself carries no gradient; the executor accumulates every parameter gradient itself."""

    @staticmethod
    def forward(ctx, feats, anchor, ex: PointSetExecutor, xyz, pe, tab):
        cond, corr, ov = ex.forward(feats, xyz, pe, tab)
        ctx.ex, ctx.tab, ctx.generation = ex, tab, ex.generation
        ctx.save_for_backward(feats, xyz, pe, cond, corr, ov)
        ctx.set_materialize_grads(False)
        return cond, corr, ov, cond[-1].clone(), corr[-1].clone(), ov[-1].clone()

    @staticmethod
    def backward(ctx, g_cond, g_corr, g_ov, g_cl, g_rl, g_ol):
        feats, xyz, pe, cond, corr, ov = ctx.saved_tensors
        f32 = lambda g: g.contiguous().float() if g is not None else None
        if FUSE and g_cond is None and g_corr is None and g_ov is None:      # (fuse = 0 keeps the per-op path's arithmetic: full zero-padded gradients)
            d_feats = ctx.ex.backward(feats, xyz, pe, ctx.tab, cond, corr, ov, f32(g_cl), f32(g_rl), f32(g_ol), last_only=True, generation=ctx.generation)
            return d_feats, None, None, None, None, None
        full = []
        for g, gl, ref in ((g_cond, g_cl, cond), (g_corr, g_rl, corr), (g_ov, g_ol, ov)):
            if gl is not None:                      # both forms in one graph: fold the last layer's gradient into the full one
                g = torch.zeros_like(ref) if g is None else g.contiguous().float().clone()
                g[-1] += gl.float()
            full.append(f32(g))
        d_feats = ctx.ex.backward(feats, xyz, pe, ctx.tab, cond, corr, ov, *full, generation=ctx.generation)
        return d_feats, None, None, None, None, None


def executor_for(model, P) -> Optional[PointSetExecutor]:
    """The model's native point-set executor for the current grad mode, or None when it does not apply (fp32 parity mode, learned
    position embedding — its gradient flows through the per-op LayerNorm nodes —, or training without preallocated gradient buffers)."""
    if model.precision != "bf16" or model.pos_emb_type != "sine" or not getattr(model, "native_pointset", True):
        return None
    names = param_names()
    train = torch.is_grad_enabled() and any(P[n].requires_grad for n in names)
    if train and not all(P[n].requires_grad for n in names):
        return None
    if train and any(P[n].grad is None or not P[n].grad.is_contiguous() for n in names):
        return None
    cache = model.__dict__.setdefault("_ps_cache", {})
    ex = cache.get(train)
    if ex is not None and not ex.still_valid():
        ex = None
    if ex is None:
        ex = cache[train] = PointSetExecutor(P, train)
    return ex


def encode_decode(ex: PointSetExecutor, feats, xyz, pe, tab, anchor, with_last: bool = False):
    """(cond [6,R,256], corr [6,R,3], ov [6,R,1]); with_last: also the last layer's three as separate tensors (see _PointSetFn)."""
    feats = feats.float().contiguous()
    out = _PointSetFn.apply(feats, anchor, ex, xyz.contiguous(), pe.contiguous(), tab)
    return out if with_last else out[:3]


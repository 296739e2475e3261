"""The four training losses of a step for all pairs at once, through csrc/losses.hip (one autograd node instead of ~120 small
torch kernels per pair).  Reference: train_nerf_regtr.py:186-229, conerf/loss/correspondence_loss.py:16-51,
conerf/loss/feature_loss.py:24-73; dreg_nerf_amd/losses.py keeps the per-pair torch formulation (evaluation, tests)."""
import ctypes
import struct

import torch

from . import lib as L
from .losses import LOSS_WEIGHTS

_WSYM = {}


def _wsym(W):
    """triu(W) + triu(W)^T, cached against W's storage, its version counter and the optimizer generation (FlatAdamW writes parameters behind torch's
    version counters; the reference never optimises W — feature_loss.py quirk Q5 — so in training this is one tiny launch per optimizer step)."""
    from . import ops
    # the kernel and the GEMM descriptor tables are built for the model width 256 (the reference's InfoNCELoss(256, ...)); anything else would read out of bounds.
    # Writers that change W behind torch's version counter (W.data[...] = ..., raw kernels) other than FlatAdamW must call ops.bump_weight_generation().
    if tuple(W.shape) != (256, 256) or W.dtype != torch.float32:
        raise ValueError(f"InfoNCE W must be fp32 [256, 256] (got {W.dtype} {tuple(W.shape)})")
    key = (W.data_ptr(), W._version, ops._weight_generation, W.device)
    hit = _WSYM.get("w")
    if hit is None or hit[0] != key:
        out = torch.empty(256, 256, dtype=torch.float32, device=W.device)
        L.check(L.load().dreg_infonce_wsym(L.ptr(W.detach().contiguous()), L.ptr(out), 256, L.stream()), "dreg_infonce_wsym")
        _WSYM["w"] = hit = (key, out)
    return hit[1]


def _gemm_tables(tab, dev):
    """The four descriptor tables (one per dependency level) of a step's InfoNCE GEMMs: they depend on the segment lengths only and live with the
    row-space table.  Base pointer ids: 0 cond_last, 1 Wsym, 2 q, 3 logits / their gradient, 4 dq, 5 d_cond."""
    gm = getattr(tab, "_infonce_gemms", None)
    if gm is not None:
        return gm
    COND, WS, Q, LG, DQ, DC = range(6)
    levels = [[], [], [], []]
    for (s0, ns, t0, nt), lo in zip(tab.segs, tab.logit_off_host):
        ld = (nt + 3) // 4 * 4
        #               a_id b_id c_id tA tB  M   N    K    lda  ldb  ldc   a_off      b_off      c_off
        levels[0].append((COND, WS, Q, 0, 0, ns, 256, 256, 256, 256, 256, s0 * 256, 0, s0 * 256))
        levels[1].append((Q, COND, LG, 0, 1, ns, nt, 256, 256, 256, ld, s0 * 256, t0 * 256, lo))
        levels[2].append((LG, COND, DQ, 0, 0, ns, 256, nt, ld, 256, 256, lo, t0 * 256, s0 * 256))
        levels[2].append((LG, Q, DC, 1, 0, nt, 256, ns, ld, 256, 256, lo, s0 * 256, t0 * 256))
        levels[3].append((DQ, WS, DC, 0, 0, ns, 256, 256, 256, 256, 256, s0 * 256, 0, s0 * 256))
    gm = []
    for recs in levels:
        blob, tiles = b"", 0
        for (a, b, c, ta, tb, M, N, K, lda, ldb, ldc, ao, bo, co) in recs:
            blob += struct.pack("<12i3q", a, b, c, ta, tb, M, N, K, lda, ldb, ldc, tiles, ao, bo, co)
            tiles += ((M + 63) // 64) * ((N + 63) // 64)
        assert len(blob) == len(recs) * L.load().dreg_gemm_f32_desc_bytes()
        t = L.to_device_async(list(blob), torch.uint8, dev)
        gm.append((t, len(recs), tiles))
    tab._infonce_gemms = gm
    return gm


NAMES = ("overlap", "nerf_cont", "feature", "corr", "total")


class _RegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cond_last, corr_last, ov_last, xyz, gt, tilde, poses, W, tab, robust: bool, r_p: float, r_n: float, defer=None):
        lib = L.load()
        dev = xyz.device
        if cond_last.shape[-1] != 256:
            raise ValueError(f"fused losses: feature width {cond_last.shape[-1]} != 256")
        cond_last, corr_last, ov_last = cond_last.contiguous(), corr_last.contiguous(), ov_last.contiguous()
        xyz, gt, poses = xyz.contiguous(), gt.contiguous(), poses.contiguous().float()
        tilde = tilde.contiguous() if tilde is not None else None       # None: 'nerf_cont' is completed later by finish_nerf_cont(defer, tilde)
        P_, Ln, R = len(tab.segs), gt.shape[0], xyz.shape[0]
        need_grad = any(ctx.needs_input_grad[:3])
        wo, wc, wf, wr = LOSS_WEIGHTS["overlap"], LOSS_WEIGHTS["nerf_cont"], LOSS_WEIGHTS["feature"], LOSS_WEIGHTS["corr"]
        partial = torch.empty(2 * P_, 4, dtype=torch.float32, device=dev)
        d_ov = torch.empty(R, dtype=torch.float32, device=dev)
        d_corr = torch.empty(R, 3, dtype=torch.float32, device=dev)
        L.check(lib.dreg_reg_point_losses(L.ptr(gt), L.ptr(tilde), L.ptr(ov_last), L.ptr(corr_last), L.ptr(xyz), L.ptr(poses), L.ptr(tab.pair_probs),
                                          L.ptr(partial), L.ptr(d_ov), L.ptr(d_corr), Ln, R, P_, int(robust), 1e-6, wo, wr, L.stream()),
                "dreg_reg_point_losses")
        # InfoNCE: logits = (A Wsym) P^T in exact fp32 — batched MFMA GEMMs of csrc/losses.hip, one launch per dependency level (round 5; rounds 1-4:
        # torch.mm = rocBLAS per pair and product), everything after the GEMMs in two kernels
        gm = _gemm_tables(tab, dev)
        wsym = _wsym(W)
        q_all = torch.empty(R, 256, dtype=torch.float32, device=dev)          # rows of the source sets only are written / read
        logits = torch.empty(max(tab.total_logits, 1), dtype=torch.float32, device=dev)
        dq = torch.empty(R, 256, dtype=torch.float32, device=dev) if need_grad else None
        d_cond = torch.empty(R, 256, dtype=torch.float32, device=dev) if need_grad else None
        bases = (ctypes.c_void_p * 8)(cond_last.data_ptr(), wsym.data_ptr(), q_all.data_ptr(), logits.data_ptr(),
                                      dq.data_ptr() if need_grad else 0, d_cond.data_ptr() if need_grad else 0, 0, 0)

        def level(k):
            t, n, tiles = gm[k]
            L.check(lib.dreg_gemm_f32_batched(L.ptr(t), n, tiles, bases, L.stream()), "dreg_gemm_f32_batched")
        level(0)                                   # q = f_src Wsym
        level(1)                                   # logits_p = q_p f_tgt_p^T
        nn = torch.empty(tab.total_src, dtype=torch.int32, device=dev)
        mask = torch.empty(tab.total_src, dtype=torch.float32, device=dev)
        count = torch.empty(P_, dtype=torch.float32, device=dev)
        loss_row = torch.empty(tab.total_src, dtype=torch.float32, device=dev)
        L.check(lib.dreg_infonce_nn(L.ptr(xyz), L.ptr(poses), L.ptr(tab.pair_probs), L.ptr(tab.src_off), L.ptr(nn), L.ptr(mask), L.ptr(count),
                                    P_, tab.total_src, float(r_p), L.stream()), "dreg_infonce_nn")
        L.check(lib.dreg_infonce_rows(L.ptr(logits), L.ptr(xyz), L.ptr(poses), L.ptr(tab.pair_probs), L.ptr(tab.src_off), L.ptr(tab.logit_off),
                                      L.ptr(nn), L.ptr(mask), L.ptr(count), L.ptr(loss_row), P_, tab.total_src, float(r_n), wf / P_,
                                      int(need_grad), L.stream()), "dreg_infonce_rows")
        out = torch.empty(5, dtype=torch.float32, device=dev)
        L.check(lib.dreg_reg_losses_final(L.ptr(partial), L.ptr(loss_row), L.ptr(count), L.ptr(tab.pair_probs), L.ptr(tab.src_off), L.ptr(out),
                                          P_, Ln, 1e-6, wo, wc, wf, wr, L.stream()), "dreg_reg_losses_final")
        if need_grad:
            level(2)                               # dq_p = G_p f_tgt_p  and  d f_tgt_p = G_p^T q_p  (G = the in-place logits gradient)
            level(3)                               # d f_src_p = dq_p Wsym  (Wsym is symmetric)
            ctx.save_for_backward(d_cond, d_corr, d_ov)
        if defer is not None:
            defer.update(partial=partial, out=out, gt=gt, probs=tab.pair_probs, P=P_, L=Ln, R=R, weights=(wo, wc, wf, wr))
        total = out[4]
        stats = out[:4]
        ctx.mark_non_differentiable(stats)
        return total, stats

    @staticmethod
    def backward(ctx, g, _gs):
        d_cond, d_corr, d_ov = ctx.saved_tensors
        return d_cond * g, d_corr * g, d_ov * g, None, None, None, None, None, None, None, None, None, None


def finish_nerf_cont(defer: dict, tilde: torch.Tensor):
    """Complete a regtr_losses(..., tilde=None, defer=defer) call on the CURRENT stream: 'nerf_cont' and 'total' of the dict it returned take the values
    the one-call form gives, bit for bit (train_nerf_regtr.py:198-201; the term has no gradient — SURVEY.md quirk Q4 — so backward never waits for it).
    The caller orders this stream behind the losses' stream and the readers of the loss values behind this one."""
    wo, wc, wf, wr = defer["weights"]
    tilde = tilde.contiguous()
    assert tuple(tilde.shape) == (defer["L"], defer["R"]) and tilde.dtype == torch.float32
    L.check(L.load().dreg_nerf_cont_deferred(L.ptr(defer["gt"]), L.ptr(tilde), L.ptr(defer["probs"]), L.ptr(defer["partial"]), L.ptr(defer["out"]),
                                             defer["P"], defer["L"], defer["R"], wo, wc, wf, wr, L.stream()), "dreg_nerf_cont_deferred")


def regtr_losses(batched: dict, poses, feature_loss, gt, tilde, robust: bool = False, defer: dict = None):
    """batched: NeRFRegTr.last_batched (cond [6,R,256], corr [6,R,3], ov [6,R,1], xyz [R,3], tab); poses [P,4,4];
    gt / tilde: {0,1} labels [6,R] of the key points / of the predicted correspondences.  tilde = None with a dict `defer`: the
    label-consistency term is left at 0 until finish_nerf_cont(defer, tilde).
    Returns {"overlap","nerf_cont","feature","corr","total"}: means over the pairs, 'total' differentiable."""
    if tilde is None and defer is None:
        raise ValueError("regtr_losses: tilde = None needs a `defer` dict for finish_nerf_cont")
    cond_l = batched["cond_last"] if "cond_last" in batched else batched["cond"][-1]
    corr_l = batched["corr_last"] if "corr_last" in batched else batched["corr"][-1]
    ov_l = batched["ov_last"][:, 0] if "ov_last" in batched else batched["ov"][-1, :, 0]
    total, stats = _RegLossFn.apply(cond_l, corr_l, ov_l, batched["xyz"], gt, tilde, poses,
                                    feature_loss.W, batched["tab"], bool(robust), float(feature_loss.r_p), float(feature_loss.r_n), defer)
    out = {k: stats[i] for i, k in enumerate(NAMES[:4])}
    out["total"] = total
    return out

"""The four training losses of a step for all pairs at once, through csrc/losses.hip (one autograd node instead of ~120 small
torch kernels per pair).  Reference: train_nerf_regtr.py:186-229, conerf/loss/correspondence_loss.py:16-51,
conerf/loss/feature_loss.py:24-73; dreg_nerf_amd/losses.py keeps the per-pair torch formulation (evaluation, tests)."""
import torch

from . import lib as L
from .losses import LOSS_WEIGHTS

NAMES = ("overlap", "nerf_cont", "feature", "corr", "total")


class _RegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cond_last, corr_last, ov_last, xyz, gt, tilde, poses, W, tab, robust: bool, r_p: float, r_n: float):
        lib = L.load()
        dev = xyz.device
        cond_last, corr_last, ov_last = cond_last.contiguous(), corr_last.contiguous(), ov_last.contiguous()
        xyz, gt, tilde, poses = xyz.contiguous(), gt.contiguous(), tilde.contiguous(), poses.contiguous().float()
        P_, Ln, R = len(tab.segs), gt.shape[0], xyz.shape[0]
        need_grad = any(ctx.needs_input_grad[:3])
        wo, wc, wf, wr = LOSS_WEIGHTS["overlap"], LOSS_WEIGHTS["nerf_cont"], LOSS_WEIGHTS["feature"], LOSS_WEIGHTS["corr"]
        partial = torch.empty(2 * P_, 4, dtype=torch.float32, device=dev)
        d_ov = torch.empty(R, dtype=torch.float32, device=dev)
        d_corr = torch.empty(R, 3, dtype=torch.float32, device=dev)
        L.check(lib.dreg_reg_point_losses(L.ptr(gt), L.ptr(tilde), L.ptr(ov_last), L.ptr(corr_last), L.ptr(xyz), L.ptr(poses), L.ptr(tab.pair_probs),
                                          L.ptr(partial), L.ptr(d_ov), L.ptr(d_corr), Ln, R, P_, int(robust), 1e-6, wo, wr, L.stream()),
                "dreg_reg_point_losses")
        # InfoNCE: logits = (A Wsym) P^T in fp32 (torch.mm = rocBLAS), everything after the GEMMs in two kernels
        wt = torch.triu(W.detach())
        wsym = wt + wt.T
        q_all = cond_last @ wsym
        logits = torch.empty(max(tab.total_logits, 1), dtype=torch.float32, device=dev)
        for (s0, ns, t0, nt), off in zip(tab.segs, tab.logit_off_host):
            torch.mm(q_all[s0:s0 + ns], cond_last[t0:t0 + nt].T, out=logits[off:off + ns * nt].view(ns, nt))
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
            dq = torch.zeros(R, 256, dtype=torch.float32, device=dev)
            for (s0, ns, t0, nt), off in zip(tab.segs, tab.logit_off_host):
                torch.mm(logits[off:off + ns * nt].view(ns, nt), cond_last[t0:t0 + nt], out=dq[s0:s0 + ns])
            d_cond = dq @ wsym                      # rows of the target sets are still zero here (Wsym is symmetric)
            for (s0, ns, t0, nt), off in zip(tab.segs, tab.logit_off_host):
                torch.mm(logits[off:off + ns * nt].view(ns, nt).T, q_all[s0:s0 + ns], out=d_cond[t0:t0 + nt])
            ctx.save_for_backward(d_cond, d_corr, d_ov)
        total = out[4]
        stats = out[:4]
        ctx.mark_non_differentiable(stats)
        return total, stats

    @staticmethod
    def backward(ctx, g, _gs):
        d_cond, d_corr, d_ov = ctx.saved_tensors
        return d_cond * g, d_corr * g, d_ov * g, None, None, None, None, None, None, None, None, None


def regtr_losses(batched: dict, poses, feature_loss, gt, tilde, robust: bool = False):
    """batched: NeRFRegTr.last_batched (cond [6,R,256], corr [6,R,3], ov [6,R,1], xyz [R,3], tab); poses [P,4,4];
    gt / tilde: {0,1} labels [6,R] of the key points / of the predicted correspondences.
    Returns {"overlap","nerf_cont","feature","corr","total"}: means over the pairs, 'total' differentiable."""
    cond_l = batched["cond_last"] if "cond_last" in batched else batched["cond"][-1]
    corr_l = batched["corr_last"] if "corr_last" in batched else batched["corr"][-1]
    ov_l = batched["ov_last"][:, 0] if "ov_last" in batched else batched["ov"][-1, :, 0]
    total, stats = _RegLossFn.apply(cond_l, corr_l, ov_l, batched["xyz"], gt, tilde, poses,
                                    feature_loss.W, batched["tab"], bool(robust), float(feature_loss.r_p), float(feature_loss.r_n))
    out = {k: stats[i] for i, k in enumerate(NAMES[:4])}
    out["total"] = total
    return out

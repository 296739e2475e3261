"""Losses and metrics of the registration training step (row H1/H2 of SURVEY.md §8).

Reference: train_nerf_regtr.py:171-229 (loss assembly, weights overlap 1 / nerf_cont 1 / feature 0.1 / corr 1,
last-layer-only), conerf/loss/correspondence_loss.py:16-51, conerf/loss/feature_loss.py:24-73,
eval_nerf_regtr.py:24-65 (RRE / RTE).  Reference quirks are kept on purpose: BCEWithLogits(input=GT,
target=pred) (Q3), the [nl,N,1] x [N] broadcast in the correspondence loss, InfoNCE's W never optimised (Q5).
"""
import torch
import torch.nn.functional as F

LOSS_WEIGHTS = {"overlap": 1.0, "nerf_cont": 1.0, "feature": 0.1, "corr": 1.0}


def se3_apply(pose, xyz):
    return xyz @ pose[:3, :3].T + pose[:3, 3]


def se3_inverse(pose):
    r, t = pose[..., :3, :3], pose[..., :3, 3:4]
    rt = r.transpose(-1, -2)
    return torch.cat([rt, -rt @ t], dim=-1)


def pseudo_huber(x, scale: float = 0.5):
    return torch.sqrt((x / scale) ** 2 + 1.0) - 1.0


def correspondence_loss(kp, kp_warped, pose, weights, robust: bool, eps: float = 1e-6):
    err = kp_warped - se3_apply(pose, kp)
    if robust:
        err = pseudo_huber(err)
    err = err.abs().sum(-1)
    return (weights * err).sum() / torch.clamp_min(weights.sum(), eps)


class InfoNCELoss(torch.nn.Module):
    def __init__(self, d_embed: int = 256, r_p: float = 0.2, r_n: float = 0.4):
        super().__init__()
        self.r_p, self.r_n = r_p, r_n
        self.W = torch.nn.Parameter(torch.zeros(d_embed, d_embed))
        torch.nn.init.normal_(self.W, std=0.1)

    def forward(self, anchor_f, pos_f, anchor_xyz, pos_xyz):
        wt = torch.triu(self.W)
        logits = anchor_f @ (wt + wt.T) @ pos_f.T
        with torch.no_grad():
            dist = torch.cdist(anchor_xyz, pos_xyz)
            d1, i1 = dist.topk(k=1, dim=-1, largest=False)
            mask = d1[..., 0] < self.r_p
            ignore = dist < self.r_n
            ignore.scatter_(-1, i1, 0)
        logits = logits.masked_fill(ignore, -float("inf"))
        loss = -torch.gather(logits, -1, i1).squeeze(-1) + torch.logsumexp(logits, dim=-1)
        return torch.where(mask, loss, torch.zeros_like(loss)).sum() / mask.sum()  # = loss[mask].sum() / mask.sum() without the host sync


def training_losses(pred, pose_gt, feature_loss: InfoNCELoss, src_ov_gt, tgt_ov_gt, src_ov_tilde, tgt_ov_tilde,
                    robust: bool = False):
    """pred: one pair's output dict; pose_gt [1,4,4]; visibility labels [nl,N,1] (in the reference they come
    from NeRF ray marching — SURVEY §8(f) N1; any {0,1} labelling is accepted here)."""
    ov_gt = torch.cat([src_ov_gt, tgt_ov_gt], dim=-2)
    ov_pred = torch.cat([pred["src_overlap"][0], pred["tgt_overlap"][0]], dim=-2)
    losses = {}
    losses["overlap"] = F.binary_cross_entropy_with_logits(ov_gt[-1], ov_pred[-1])
    losses["nerf_cont"] = F.smooth_l1_loss(ov_gt, torch.cat([src_ov_tilde, tgt_ov_tilde], dim=-2))
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    losses["feature"] = feature_loss(pred["src_feats"][0][-1], pred["tgt_feats"][0][-1], se3_apply(pose_gt[0], s_kp), t_kp)
    losses["corr"] = correspondence_loss(s_kp, pred["src_kp_warped"][0][-1], pose_gt[0], src_ov_gt, robust) + \
        correspondence_loss(t_kp, pred["tgt_kp_warped"][0][-1], se3_inverse(pose_gt[0]), tgt_ov_gt, robust)
    losses["total"] = sum(losses[k] * LOSS_WEIGHTS[k] for k in LOSS_WEIGHTS)
    return losses


@torch.no_grad()
def rre_rte(pred, gt, eps: float = 1e-7):
    """pred [B,3,4], gt [B,4,4] -> (RRE degrees [B], RTE [B])."""
    rd = pred[..., :3, :3].transpose(-2, -1) @ gt[..., :3, :3]
    tr = rd[..., 0, 0] + rd[..., 1, 1] + rd[..., 2, 2]
    ang = torch.rad2deg(((tr - 1) / 2).clamp(-1 + eps, 1 - eps).acos())
    return ang, (pred[..., :3, 3] - gt[..., :3, 3]).norm(dim=-1)


@torch.no_grad()
def evaluate_camera_alignment(pred_poses, poses_gt):
    r, t = rre_rte(pred_poses, poses_gt)
    return {"R_error_mean": r.mean().cpu(), "t_error_mean": t.mean(), "R_error_med": r.median().cpu(), "t_error_med": t.median()}

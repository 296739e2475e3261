"""One optimizer step of RegTR training on a batch of NeRF pairs (row H1 of SURVEY.md §8), optionally data-parallel.

Reference step (one pair per iteration): train_nerf_regtr.py:171-256 — forward, four losses on the last decoder
layer, backward, clip_grad_norm_(0.1), AdamW(lr 1e-4, wd 1e-4) over model.parameters() only, StepLR(34000, 0.5).
Build-side additions: several pairs per step (loss = mean over the pairs of a rank) and plain DDP — every rank
takes its own pairs, gradients are averaged with bucketed all-reduce over RCCL (torch.distributed 'nccl'), the
clip norm is computed on the averaged gradients, and every rank applies the identical update.
"""
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import losses as LS
from . import synth


def _default_labels(pred):
    """Synthetic {0,1} visibility labels (the reference obtains them by NeRF ray marching — SURVEY §8(f) N1)."""
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    with torch.no_grad():
        s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
        nl = pred["src_kp_warped"][0].shape[0]
        s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(nl)])
        t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(nl)])
    return s_gt, t_gt, s_tl, t_tl


class GradBuckets:
    """Flat fp32 gradient buckets in reverse parameter order (gradients of the decoder/transformer are ready first,
    the ResNet's last), all-reduced in place and averaged.  ~25 MB buckets: one xGMI ring step moves bucket/8 per link."""

    def __init__(self, params: List[torch.nn.Parameter], bucket_bytes: int = 25 << 20):
        self.params = list(reversed(params))
        self.buckets = []
        cur, cur_n = [], 0
        for p in self.params:
            cur.append(p)
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        dev = self.params[0].device
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=dev) for b in self.buckets]

    def all_reduce_mean(self, world: int):
        handles = []
        for b, flat in zip(self.buckets, self.flat):
            off = 0
            for p in b:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for h, b, flat in zip(handles, self.buckets, self.flat):
            h.wait()
            flat.div_(world)
            off = 0
            for p in b:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n


class TrainStep:
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-4, grad_clip: float = 0.1,
                 robust_loss: bool = False, step_lr: int = 34000, gamma: float = 0.5, finetune: bool = False,
                 label_fn: Optional[Callable] = None):
        self.model = model
        dev = next(model.parameters()).device
        self.feature_loss = LS.InfoNCELoss(256, 0.2, 0.4).to(dev)
        self.params = [p for p in model.parameters()]
        self.optimizer = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=step_lr, gamma=gamma)
        self.grad_clip = grad_clip
        self.robust = robust_loss
        self.finetune = finetune
        self.label_fn = label_fn or _default_labels
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.buckets = GradBuckets(self.params) if self.world > 1 else None
        self.last_losses = None
        self.last_preds = None

    def step(self, batch: List[dict]) -> dict:
        self.optimizer.zero_grad(set_to_none=True)
        if self.feature_loss.W.grad is not None:
            self.feature_loss.W.grad = None
        preds = self.model.forward_batch(batch)
        total = 0.0
        agg = {}
        for d, pred in zip(batch, preds):
            s_gt, t_gt, s_tl, t_tl = self.label_fn(pred) if "overlap_labels" not in d else d["overlap_labels"]
            ls = LS.training_losses(pred, d["pose"], self.feature_loss, s_gt, t_gt, s_tl, t_tl, self.robust)
            total = total + ls["total"]
            for k, v in ls.items():
                agg[k] = agg.get(k, 0.0) + v.detach()
        total = total / len(batch)
        total.backward()
        if self.buckets is not None:
            self.buckets.all_reduce_mean(self.world)
        gnorm = None
        if self.grad_clip > 0:
            gnorm = torch.nn.utils.clip_grad_norm_(self.params, max_norm=self.grad_clip)
        self.optimizer.step()
        if not self.finetune:
            self.scheduler.step()
        self.last_losses = {k: v / len(batch) for k, v in agg.items()}
        self.last_preds = preds
        return {"losses": self.last_losses, "grad_norm": gnorm}

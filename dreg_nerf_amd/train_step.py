"""One optimizer step of RegTR training on a batch of NeRF pairs (row H1 of SURVEY.md §8), optionally data-parallel.

Reference step (one pair per iteration): train_nerf_regtr.py:171-256 — forward, four losses on the last decoder
layer, backward, clip_grad_norm_(0.1), AdamW(lr 1e-4, wd 1e-4) over model.parameters() only, StepLR(34000, 0.5).
Build-side additions: several pairs per step (loss = mean over the pairs of a rank) and plain DDP — every rank
takes its own pairs, the flat gradient buffer is averaged with ~25 MB all-reduce slices over RCCL
(torch.distributed 'nccl'), the clip norm is computed on the averaged gradients, and every rank applies the
identical update (optim.FlatAdamW: HIP grad-norm + AdamW kernels over one flat fp32 buffer).
"""
import os
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import fused_losses as FL
from . import losses as LS
from . import ops
from . import synth
from .optim import FlatAdamW, GradSync, StepLR


def nerf_labels(pred, data):
    """train_nerf_regtr.py:186-199: overlap ground truth = surface-field visibility of the key points in their own NeRF block,
    'tilde' scores = visibility of the predicted correspondences — all four through the fused ray-march kernel."""
    from .visibility import compute_visibility_score
    nl = pred["src_kp_warped"][0].shape[0]
    with torch.no_grad():
        # the reference marches the key points once per decoder layer (the same [N,3] expanded nl times: six identical label sets) and
        # makes four calls per pair; here every point set is marched once and both sets of a block share one launch: [1 + nl, N, 3]
        out = []
        for side in ("src", "tgt"):
            kp, warped = pred[side + "_kp"][0], pred[side + "_kp_warped"][0].detach()
            both = compute_visibility_score([torch.cat([kp.reshape(1, -1, 3), warped], 0)], data[side + "_nerf_path"])[0]
            out.append((both[:1].expand(nl, -1, -1), both[1:]))
    (s_gt, s_tl), (t_gt, t_tl) = out
    return s_gt, t_gt, s_tl, t_tl


def nerf_labels_batched(preds, batch):
    """nerf_labels for every pair of a step from ONE ray-march launch over all their blocks (visibility.compute_visibility_scores_batched)."""
    from .visibility import compute_visibility_scores_batched
    reqs = []
    with torch.no_grad():
        for pred, data in zip(preds, batch):
            for side in ("src", "tgt"):
                kp, warped = pred[side + "_kp"][0], pred[side + "_kp_warped"][0].detach()
                reqs.append((torch.cat([kp.reshape(1, -1, 3), warped], 0), data[side + "_nerf_path"]))
        outs = compute_visibility_scores_batched(reqs)
    res = []
    for i, pred in enumerate(preds):
        nl = pred["src_kp_warped"][0].shape[0]
        s, t = outs[2 * i], outs[2 * i + 1]
        res.append((s[:1].expand(nl, -1, -1), t[:1].expand(nl, -1, -1), s[1:], t[1:]))
    return res


def _default_labels(pred):
    """Synthetic {0,1} visibility labels for data without NeRF blocks on disk (bench, synthetic scenes)."""
    s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
    with torch.no_grad():
        s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
        nl = pred["src_kp_warped"][0].shape[0]
        s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(nl)])
        t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(nl)])
    return s_gt, t_gt, s_tl, t_tl


def synthetic_labels_rows(xyz, corr):
    """The labels of _default_labels in the fused losses' shared row space: gt, tilde [L, R] from the key points xyz [R, 3] and the predicted
    correspondences corr [L, R, 3] — one launch on the GPU (csrc/losses.hip, the same bits as the element-wise formula)."""
    with torch.no_grad():
        if xyz.is_cuda and xyz.dtype == torch.float32 and corr.dtype == torch.float32:
            from . import lib as L
            nl, R = corr.shape[0], corr.shape[1]
            gt, tilde = torch.empty(nl, R, device=xyz.device), torch.empty(nl, R, device=xyz.device)
            L.check(L.load().dreg_halfspace_labels(L.ptr(xyz.detach().contiguous()), L.ptr(corr.detach().contiguous()), L.ptr(gt), L.ptr(tilde), nl, R, L.stream()),
                    "dreg_halfspace_labels")
            return gt, tilde
        gt = synth.synthetic_overlap_gt(xyz, corr.shape[0])[..., 0]
        tilde = (corr[..., 0] + 0.31 * corr[..., 1] - 0.17 * corr[..., 2] > 0.0123).to(corr.dtype)
    return gt, tilde


STEP_TIMERS = {} if os.environ.get("DREG_STEP_TIMERS") == "1" else None      # diagnostic: host seconds of a step's phases, accumulated (train_nerf_regtr.py prints them per epoch)


class _SplitLabels:
    """One step's NeRF-block labels in two launches on TrainStep's label stream (see TrainStep.split_labels)."""

    def __init__(self, ts, batch, dev):
        self.ts, self.batch, self.dev = ts, batch, dev
        if ts._label_stream is None or ts._label_stream.device != dev:
            ts._label_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("DREG_LABEL_PRIORITY", "0")))
        self.main = torch.cuda.current_stream(dev)
        dbg = os.environ.get("DREG_LABEL_DEBUG", "")          # diagnostic: "kp_main" / "tilde_main" issue that march on the step's own stream
        self.dbg = dbg                                        # ("kp_geo": the key-point march on the geometry stream that produced the points; "skip_kp" / "skip_tilde": timing ablations, WRONG labels)
        self.kp_stream = self.main if "kp_main" in dbg else ts._label_stream
        self.stream = self.main if "tilde_main" in dbg else ts._label_stream
        self.gt_row = self.ev_gt = None

    def after_geometry(self, pts_l, segs, producer):
        """Called by forward_batch when the key points exist (on `producer`, the geometry stream, which runs ahead of the step stream)."""
        from .visibility import compute_visibility_scores_batched
        ls = producer if ("kp_geo" in self.dbg and producer is not None) else self.kp_stream
        if ls is not producer:
            ls.wait_stream(producer if producer is not None else self.main)
        with torch.cuda.stream(ls), torch.no_grad():
            reqs = []
            for pts, (ns, nt), d in zip(pts_l, segs, self.batch):
                pts.record_stream(ls)
                reqs.append((pts[:ns].reshape(1, ns, 3), d["src_nerf_path"]))
                reqs.append((pts[ns:ns + nt].reshape(1, nt, 3), d["tgt_nerf_path"]))
            if STEP_TIMERS is not None:
                self._e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                self._e[0].record(ls)
            outs = compute_visibility_scores_batched(reqs, max_waves=self.ts.label_waves[0]) if "skip_kp" not in self.dbg else [torch.zeros(x.shape[0], x.shape[1], 1, device=x.device) for x, _ in reqs]
            if STEP_TIMERS is not None:
                self._e[1].record(ls)
            self.gt_row = torch.cat([o.reshape(-1) for o in outs])                 # [R]: rows in the batch's row space (pair 0 src | tgt, pair 1 ...)
            self.ev_gt = torch.cuda.Event()
            self.ev_gt.record(ls)

    def after_forward(self, bt):
        """gt [6,R] for the losses (the same key-point labels for every decoder layer, as the reference expands them) and the 'tilde' march behind forward."""
        from .visibility import compute_visibility_scores_batched
        main, ls = self.main, self.stream
        main.wait_event(self.ev_gt)
        self.gt_row.record_stream(main)
        corr = bt["corr"]
        nl = corr.shape[0]
        gt = self.gt_row[None].expand(nl, -1).contiguous()
        ev_fwd = torch.cuda.Event()
        ev_fwd.record(main)
        ls.wait_event(ev_fwd)
        with torch.cuda.stream(ls), torch.no_grad():
            corr.record_stream(ls)
            c = corr.detach()
            reqs = []
            for (s0, ns, t0, nt), d in zip(bt["tab"].segs, self.batch):
                reqs.append((c[:, s0:s0 + ns], d["src_nerf_path"]))
                reqs.append((c[:, t0:t0 + nt], d["tgt_nerf_path"]))
            if STEP_TIMERS is not None:
                self._e[2].record(ls)
            outs = compute_visibility_scores_batched(reqs, max_waves=self.ts.label_waves[1]) if "skip_tilde" not in self.dbg else [torch.zeros(x.shape[0], x.shape[1], 1, device=x.device) for x, _ in reqs]
            if STEP_TIMERS is not None:
                self._e[3].record(ls)
            self.tilde = torch.cat([o[..., 0] for o in outs], dim=1)             # [6,R]
        return gt

    def after_losses(self, defer):
        main, ls = self.main, self.stream
        ev = torch.cuda.Event()
        ev.record(main)
        ls.wait_event(ev)                       # the losses' own final kernel has written out[1] = 0 / out[4] before they are completed
        for t in (defer["partial"], defer["out"], defer["gt"]):
            t.record_stream(ls)
        with torch.cuda.stream(ls):
            FL.finish_nerf_cont(defer, self.tilde)
        self.ev_done = torch.cuda.Event()
        self.ev_done.record(ls)

    def join(self):
        self.main.wait_event(self.ev_done)
        self.tilde.record_stream(self.main)
        if STEP_TIMERS is not None:          # diagnostic only (a device sync per step): how long the two marches took on the label stream
            torch.cuda.synchronize()
            STEP_TIMERS["gpu ms kp march"] = STEP_TIMERS.get("gpu ms kp march", 0.0) + self._e[0].elapsed_time(self._e[1])
            STEP_TIMERS["gpu ms tilde march"] = STEP_TIMERS.get("gpu ms tilde march", 0.0) + self._e[2].elapsed_time(self._e[3])


class TrainStep:
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-4, grad_clip: float = 0.1,
                 robust_loss: bool = False, step_lr: int = 34000, gamma: float = 0.5, finetune: bool = False,
                 label_fn: Optional[Callable] = None, fused_losses: bool = True):
        self.model = model
        dev = next(model.parameters()).device
        self.feature_loss = LS.InfoNCELoss(256, 0.2, 0.4).to(dev)
        self.params = [p for p in model.parameters()]
        never_used = [p for n, p in model.named_parameters() if n.startswith("correspondence_decoder.q_norm.")]   # nerf_regtr.py:266
        self.optimizer = FlatAdamW(self.params, lr=lr, weight_decay=weight_decay, max_norm=grad_clip, never_used=never_used)
        self.scheduler = StepLR(self.optimizer, step_size=step_lr, gamma=gamma)
        self.grad_clip = grad_clip
        self.robust = robust_loss
        self.finetune = finetune
        self.label_fn = label_fn or _default_labels
        self.fused_losses = fused_losses
        # Labels from NeRF blocks in two parts (review item 2 of round 5): the key points' visibility — the overlap ground truth, the only labels a gradient
        # depends on — is marched as soon as the geometry phase has produced the key points (on the label stream, under the feature network's forward); the
        # six predicted-correspondence sets per block ('tilde', 6/7 of the rays) feed only 'nerf_cont', which has no gradient (train_nerf_regtr.py:198-201,
        # SURVEY.md quirk Q4): they are marched on the label stream while backward runs and complete the step's loss VALUES there.  Same labels, same
        # losses bit for bit (tests/test_hip_visibility.py, tests/test_hip_bench_labels.py).  DREG_SPLIT_LABELS=0: everything in front of the loss.
        self.split_labels = bool(int(os.environ.get("DREG_SPLIT_LABELS", "1")))
        self._label_stream = None
        # one-wave workgroups of the two background label launches (key points, predicted correspondences): DREG_LABEL_WAVES="kp,tilde", 0 = full width
        self.label_waves = tuple(int(v) for v in os.environ.get("DREG_LABEL_WAVES", "128,128").split(","))
        # Second stream for the weight / bias gradients of the point-set half's linear layers (Conv3dFn.backward), ~1.3 ms per step.
        # (It exposed the packed-fp32 co-execution fault described in DESIGN.md; the library is built without those instructions
        # and steps are bitwise reproducible with it: tests/test_hip_trunk_exec.py.)  DREG_PG_STREAM=0 turns it off.
        self.persistent_grad_buffers = True     # dense gradient buffers of the active-set head kept zero by clearing rows (ops.TrilinearGatherFn)
        self.overlap_param_grads = bool(int(os.environ.get("DREG_PG_STREAM", "1"))) and not bool(int(os.environ.get("DREG_SERIAL_STREAMS", "0")))
        self._pg_stream = None
        # test hook: run the bucketed gradient exchange on a process group of ONE rank too (the only way RCCL itself — ReduceOp.AVG on
        # views of the flat buffer, its streams and events — can be exercised on a one-GPU box: tests/test_hip_ddp_gpu.py)
        self.force_grad_sync = bool(int(os.environ.get("DREG_FORCE_GRAD_SYNC", "0")))
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._sync = None
        self.last_losses = None
        self.last_preds = None

    def step(self, batch: List[dict]) -> dict:
        # (no wait for the side-stream repack of the last update here: every consumer of a pack waits for it itself — ops.wait_packs —,
        #  so gradient clearing and input packing run under it: 0.15 ms of the step.  Splitting the repack so that the stem could start
        #  before the large packs are done was measured on top of that and gives nothing: the geometry stream is the longer wait)
        self.optimizer.zero_grad()
        if self.feature_loss.W.grad is not None:
            self.feature_loss.W.grad = None
        have_nerf = [bool(d.get("src_nerf_path")) and os.path.exists(d["src_nerf_path"]) and os.path.exists(d.get("tgt_nerf_path", ""))
                     for d in batch]
        dev0 = next(self.model.parameters()).device
        split = self.fused_losses and self.split_labels and dev0.type == "cuda" and all(have_nerf) and len(batch) > 0
        lab = _SplitLabels(self, batch, dev0) if split else None
        if lab is not None:
            self.model.__dict__["_after_geometry"] = lab.after_geometry
        _t = [__import__("time").perf_counter()] if STEP_TIMERS is not None else None
        try:
            preds = self.model.forward_batch(batch)
        finally:
            self.model.__dict__.pop("_after_geometry", None)
        if _t is not None:
            _t.append(__import__("time").perf_counter())
        if self.fused_losses:
            # all pairs at once through csrc/losses.hip: labels in the shared row space [6,R], one autograd node
            bt = self.model.last_batched
            defer = None
            if lab is not None:
                _q0 = __import__("time").perf_counter()
                gt, tilde, defer = lab.after_forward(bt), None, {}
                if STEP_TIMERS is not None:
                    STEP_TIMERS["  after_forward"] = STEP_TIMERS.get("  after_forward", 0.0) + __import__("time").perf_counter() - _q0
            elif not any(have_nerf) and self.label_fn is _default_labels:
                gt, tilde = synthetic_labels_rows(bt["xyz"], bt["corr"])
            else:
                gts, tls = [], []
                from_blocks = iter(nerf_labels_batched([p for p, hn in zip(preds, have_nerf) if hn], [d for d, hn in zip(batch, have_nerf) if hn]))
                for d, pred, hn in zip(batch, preds, have_nerf):
                    s_gt, t_gt, s_tl, t_tl = next(from_blocks) if hn else self.label_fn(pred)
                    gts += [s_gt[..., 0], t_gt[..., 0]]
                    tls += [s_tl[..., 0], t_tl[..., 0]]
                gt, tilde = torch.cat(gts, dim=1), torch.cat(tls, dim=1)
            poses = torch.cat([d["pose"].reshape(1, 4, 4) for d in batch]).float()
            ls = FL.regtr_losses(bt, poses, self.feature_loss, gt, tilde, self.robust, defer=defer)
            if lab is not None:
                _q0 = __import__("time").perf_counter()
                lab.after_losses(defer)
                if STEP_TIMERS is not None:
                    STEP_TIMERS["  after_losses"] = STEP_TIMERS.get("  after_losses", 0.0) + __import__("time").perf_counter() - _q0
            total = ls["total"]
            agg = None                       # the fused losses are batch means already (no x len / len round trip: ten tiny launches)
            means = {k: v.detach() for k, v in ls.items()}
        else:
            total = 0.0
            agg = {}
            for d, pred in zip(batch, preds):
                if d.get("src_nerf_path") and os.path.exists(d["src_nerf_path"]) and os.path.exists(d.get("tgt_nerf_path", "")):
                    s_gt, t_gt, s_tl, t_tl = nerf_labels(pred, d)
                else:
                    s_gt, t_gt, s_tl, t_tl = self.label_fn(pred)
                ls = LS.training_losses(pred, d["pose"], self.feature_loss, s_gt, t_gt, s_tl, t_tl, self.robust)
                total = total + ls["total"]
                for k, v in ls.items():
                    agg[k] = agg.get(k, 0.0) + v.detach()
            total = total / len(batch)
        dev = next(self.model.parameters()).device
        sync = None
        if self.world > 1 or self.force_grad_sync:
            # gradient averaging overlapped with backward: buckets of the flat gradient buffer are all-reduced as soon as backward
            # has produced them (optim.GradSync; the trunk executor reports progress between its segments)
            if self._sync is None:
                from .trunk_exec import aux_stream
                self._sync = GradSync(self.optimizer, self.world, streams=[aux_stream(dev)] if dev.type == "cuda" else [])
            sync = self._sync
            sync.begin()
            ops.GRAD_SYNC = sync
        ops.PERSISTENT_GRAD_BUFFERS = dev.type == "cuda" and self.persistent_grad_buffers
        if _t is not None:
            _t.append(__import__("time").perf_counter())
        try:
            self._backward(total, dev)
        finally:
            ops.GRAD_SYNC = None
            ops.PERSISTENT_GRAD_BUFFERS = False
            ops.release_grad_buffers()     # their consumer (the trunk's backward) is enqueued: the next backward may reuse them
        if sync is not None:
            sync.finish()
        if self.overlap_param_grads and dev.type == "cuda":
            from .trunk_exec import aux_stream
            ops.PACK_STREAM = aux_stream(dev)
        try:
            self.optimizer.step()          # clip_grad_norm_(grad_clip) folded into the AdamW kernel; weight packs refreshed
            for ex in getattr(self.model, "_trunk_cache", {}).values():
                if ex.with_grad and ex.still_valid():
                    ex.repack_after_update()
        finally:
            ops.PACK_STREAM = None
        gnorm = self.optimizer.grad_norm()
        if not self.finetune:
            self.scheduler.step()
        if _t is not None:
            _t.append(__import__("time").perf_counter())
            for k_, a_, b_ in zip(("forward", "labels+losses", "backward+optimizer"), _t[:-1], _t[1:]):
                STEP_TIMERS[k_] = STEP_TIMERS.get(k_, 0.0) + b_ - a_
        if lab is not None:
            lab.join()                     # readers of the loss values (this stream) come behind the label stream's completion of 'nerf_cont' / 'total'
        self.last_losses = means if agg is None else {k: v / len(batch) for k, v in agg.items()}
        self.last_preds = preds
        from . import visibility
        # labels marched from NeRF blocks: a persistent launch that ran into its pass bound (points left unlabelled) is reported here as soon as its
        # counters have arrived — late (this step's update is applied), without stalling on this step's own launch (check(wait=True) at checkpoints /
        # the end of a run).  check() takes the watch's lock: `pending` is also filled from the loader / geometry threads.
        visibility.OVERRUN.check()
        return {"losses": self.last_losses, "grad_norm": gnorm}

    def close(self):
        """Drop the device memory this step object keeps alive between steps (cached dense gradient buffers, ~2 GB at 128^3 x 4 pairs)."""
        ops.release_grad_buffers(free=True)

    def _backward(self, total, dev):
        if self.overlap_param_grads and dev.type == "cuda":
            # weight / bias gradients of the point-set half's linear layers on a side stream, next to the data-gradient chain
            if self._pg_stream is None:
                from .trunk_exec import aux_stream
                self._pg_stream = aux_stream(dev)      # the same second stream the trunk executor uses
            ops.PARAM_GRAD_STREAM = self._pg_stream
            try:
                total.backward()
                with torch.cuda.stream(self._pg_stream):
                    ops.flush_wgrad_reduce()       # deferred split sums nobody flushed yet (no trunk executor in the graph)
            finally:
                ops.PARAM_GRAD_STREAM = None
                ops._PENDING_REDUCE.clear()
            torch.cuda.current_stream(dev).wait_stream(self._pg_stream)
        else:
            total.backward()

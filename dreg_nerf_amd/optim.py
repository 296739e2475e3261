"""Flat-buffer AdamW + gradient clipping on the HIP kernels (dreg_grad_norm / dreg_adamw_step).

Reference: torch.optim.AdamW(model.parameters(), lr, weight_decay=1e-4) preceded by clip_grad_norm_(0.1)
(train_nerf_regtr.py:96-102,232-237).  All parameters live in one fp32 buffer (the nn.Parameters become views of
it, so state_dict()/checkpoints are unchanged), gradients in a second one that autograd accumulates into in
place; the data-parallel all-reduce runs directly on slices of the gradient buffer."""
from typing import List

import torch
import torch.distributed as dist

from . import lib as L
from . import ops


class FlatAdamW:
    def __init__(self, params: List[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-4, max_norm: float = 0.1, never_used=()):
        """never_used: parameters that take no part in the forward pass (the reference's correspondence_decoder.q_norm,
        nerf_regtr.py:266): torch.optim.AdamW skips a parameter whose .grad is None — no moment update and NO weight decay — so they
        sit behind the updated range of the flat buffers and the kernels never touch them."""
        self.params = [p for p in params]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        skip = {id(p) for p in never_used}
        self.n_active = n - sum(p.numel() for p in self.params if id(p) in skip)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = [0] * len(self.params)
        off = 0
        with torch.no_grad():
            for active in (True, False):
                for i, p in enumerate(self.params):
                    if (id(p) not in skip) != active:
                        continue
                    k = p.numel()
                    self.flat_p[off:off + k].copy_(p.data.reshape(-1))
                    p.data = self.flat_p[off:off + k].view(p.shape)
                    p.grad = self.flat_g[off:off + k].view(p.shape)
                    self.offsets[i] = off
                    off += k
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.step_count = 0
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        ops.bump_weight_generation()

    def zero_grad(self):
        self.flat_g.zero_()
        for p, off in zip(self.params, self.offsets):  # autograd may have replaced .grad if it was unset
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)

    def offset_of(self, grad_ptr: int) -> int:
        """Element offset inside the flat gradient buffer of a .grad view's data pointer (-1: not one of ours)."""
        d = grad_ptr - self.flat_g.data_ptr()
        return d // 4 if 0 <= d < self.n * 4 else -1

    def all_reduce_mean(self, world: int, bucket_elems: int = (25 << 20) // 4):
        """Average gradients over the ranks: async all-reduce of ~25 MB slices issued from the END of the buffer
        (decoder/transformer/FPN-head gradients, which backward produces first, sit at the highest offsets)."""
        handles = []
        hi = self.n
        while hi > 0:
            lo = max(0, hi - bucket_elems)
            handles.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            hi = lo
        for h in handles:
            h.wait()
        self.flat_g.div_(world)

    def step(self):
        lib = L.load()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        if self.max_norm > 0:
            L.check(lib.dreg_grad_norm(L.ptr(self.flat_g), L.ptr(self._norm), L.ptr(self._ws), self.n_active, L.stream()), "dreg_grad_norm")
        L.check(lib.dreg_adamw_step(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), L.ptr(self._norm),
                                    self.n_active, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                    float(self.max_norm), L.stream()), "dreg_adamw_step")
        ops.bump_weight_generation()  # the kernel wrote the parameters behind torch's version counters
        with ops.pack_region():             # on ops.PACK_STREAM when train_step set one: next to the next step's input staging
            ops.repack_all(self.flat_p.device)  # every cached bf16/fp32 weight pack refreshed by one launch

    def grad_norm(self) -> torch.Tensor:
        return self._norm

    # torch.optim.AdamW-compatible state (CheckPointManager stores optimizer.state_dict(): checkpoint_manager.py:58-80)
    def state_dict(self):
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
        g = dict(self.param_groups[0])
        g.update({"amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                  "fused": None, "params": list(range(len(self.params)))})
        return {"state": state if self.step_count > 0 else {}, "param_groups": [g]}

    def load_state_dict(self, sd):
        st = sd.get("state", {})
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                if i in st:
                    k = p.numel()
                    self.exp_avg[off:off + k].copy_(st[i]["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st[i]["exp_avg_sq"].reshape(-1))
                    self.step_count = int(float(st[i]["step"]))
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.lr)


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) on FlatAdamW (train_nerf_regtr.py:100-102)."""

    def __init__(self, optimizer: FlatAdamW, step_size: int, gamma: float):
        self.opt, self.step_size, self.gamma = optimizer, step_size, gamma
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.opt.param_groups[0]["lr"] = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]

    def state_dict(self):
        return {"step_size": self.step_size, "gamma": self.gamma, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch,
                "_step_count": self.last_epoch + 1, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        self.last_epoch = sd.get("last_epoch", 0)
        self.base_lr = sd.get("base_lrs", [self.base_lr])[0]
        self.opt.param_groups[0]["lr"] = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)


class GradSync:
    """Gradient averaging of a data-parallel step OVERLAPPED with the backward pass: the flat gradient buffer is cut
    into ~25 MB buckets from its END (decoder / transformer / FPN head: what backward finishes first); whoever drives backward calls
    ready(lo) as soon as every gradient at element offsets >= lo has been ENQUEUED (on the main stream and the parameter-gradient
    stream); every whole bucket above `lo` is then all-reduced on the collective's own stream behind events of both streams, while
    backward continues.  finish() launches what is left and makes the current stream wait for all of it.
    Averaging: ReduceOp.AVG on RCCL ('nccl'); SUM + one division on backends without AVG (gloo)."""

    def __init__(self, opt: FlatAdamW, world: int, bucket_bytes: int = 25 << 20, streams=(), reduce_fn=None):
        self.opt, self.world = opt, world
        be = max(bucket_bytes // 4, 1)
        self.buckets = []                      # (lo, hi) from the end of the UPDATED range (never-used parameters have no gradient)
        hi = opt.n_active
        while hi > 0:
            lo = max(0, hi - be)
            self.buckets.append((lo, hi))
            hi = lo
        self.extra_streams = list(streams)     # streams besides the current one that produce gradients (parameter-gradient stream)
        self.reduce_fn = reduce_fn             # tests: called instead of torch.distributed with (lo, hi)
        self._use_avg = reduce_fn is None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        self._side = None
        # measure_exposed: events around finish()'s wait — how long the step's own stream stood still for gradient exchange that backward did not
        # hide (bench.py's multi-rank line: `allreduce_exposed_ms`); off in training (two event records per step)
        self.measure_exposed = False
        self._exposed = []
        self.begin()

    def exposed_ms(self, reset: bool = True):
        """After a device synchronisation: (mean exposed ms per step, steps measured)."""
        v = [a.elapsed_time(b) for a, b in self._exposed]
        if reset:
            self._exposed = []
        return (sum(v) / len(v) if v else 0.0), len(v)

    def begin(self):
        self.next, self.handles, self.launched = 0, [], []

    def _launch(self, lo, hi):
        self.launched.append((lo, hi))
        if self.reduce_fn is not None:
            self.reduce_fn(lo, hi)
            return
        g = self.opt.flat_g[lo:hi]
        if g.is_cuda:
            cur = torch.cuda.current_stream(g.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=g.device)
            self._side.wait_stream(cur)        # events, not host waits: the collective starts when both producers reach this point
            for s in self.extra_streams:
                self._side.wait_stream(s)
            with torch.cuda.stream(self._side):
                try:
                    h = dist.all_reduce(g, op=dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM, async_op=True)
                except RuntimeError:
                    # a collective library without ReduceOp.AVG: only acceptable on the step's FIRST bucket (then every bucket of
                    # the step is summed and finish() divides once); later it would mix averaged and summed buckets
                    if not self._use_avg or len(self.launched) != 1:
                        raise
                    self._use_avg = False
                    h = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
            g.record_stream(self._side)
        else:
            h = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
        self.handles.append(h)

    def ready(self, lo: int):
        """All gradients at offsets >= lo are enqueued: launch every not-yet-launched bucket that lies entirely above lo."""
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= lo:
            self._launch(*self.buckets[self.next])
            self.next += 1

    def finish(self):
        self.ready(0)
        ev = None
        if self.measure_exposed and self.opt.flat_g.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self.handles:
            h.wait()                           # the current stream waits for the collective (no host block on 'nccl')
        if self.reduce_fn is None and not self._use_avg and self.world > 1:
            self.opt.flat_g[:self.opt.n_active].div_(self.world)
        if self._side is not None:
            torch.cuda.current_stream(self.opt.flat_g.device).wait_stream(self._side)
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)


def ranks_in_sync(opt: FlatAdamW, grad_norm: torch.Tensor = None):
    """Data-parallel self-check: after identical initial weights and averaged gradients every rank must hold the SAME parameters
    and have clipped by the SAME global gradient norm.  all-reduces (MAX - MIN) of an order-sensitive checksum pair of the flat parameter buffer
    (fp64 sum, fp64 sum of index-weighted values) and of the last step's gradient norm; returns a dict whose `ranks_in_sync` is True when all spreads
    are exactly 0 (trivially so outside a process group)."""
    p = opt.flat_p[:opt.n_active].double()
    idx = torch.arange(p.numel(), dtype=torch.float64, device=p.device).remainder_(8191.0).add_(1.0)
    vals = [p.sum(), (p * idx).sum()]
    vals.append(grad_norm.reshape(-1)[0].double() if grad_norm is not None else torch.zeros((), dtype=torch.float64, device=p.device))
    v = torch.stack(vals)
    ranks = 1
    spread = torch.zeros_like(v)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        ranks = dist.get_world_size()
        hi, lo = v.clone(), v.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        spread = hi - lo
    s = [float(x) for x in spread.cpu()]
    return {"ranks_in_sync": all(x == 0.0 for x in s) and all(torch.isfinite(v).tolist()), "ranks": ranks, "param_checksum": float(v[0]),
            "param_checksum_spread": s[0], "param_weighted_checksum_spread": s[1], "grad_norm": float(v[2]), "grad_norm_spread": s[2]}


def broadcast_buffers(module: torch.nn.Module, src: int = 0):
    """Every rank takes rank `src`'s module BUFFERS (the BatchNorm running statistics and num_batches_tracked): plain DDP averages
    gradients only, so each rank's running statistics follow its own pairs (the reference has no opinion; the build
    broadcasts rank 0's at validation / checkpoint time).  One collective per dtype: the buffers are flattened into a single message
    (159 BatchNorm buffers of the registration network = 0.2 MB) and copied back in place.  No-op outside a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    by_dtype = {}
    seen = set()
    for b in module.buffers():
        if b.data_ptr() in seen:              # the ResNet is registered under two names (feature_pyramid_net.py:43,194-200): one storage
            continue
        seen.add(b.data_ptr())
        by_dtype.setdefault(b.dtype, []).append(b)
    n = 0
    with torch.no_grad():
        for dt in sorted(by_dtype, key=str):   # same order on every rank
            bufs = by_dtype[dt]
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            n += len(bufs)
    return n

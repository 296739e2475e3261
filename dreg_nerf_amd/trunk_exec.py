"""Native (C++) execution of the FPN3D feature network: the per-op description in ``regtr.NeRFRegTr.fpn`` is recorded once
per (batch, resolution, head mode) into an op program and handed to ``csrc/executor.hip``; afterwards a training step issues
the whole forward (and the whole backward) of the network with ONE C call each.

Reference network: conerf/model/resnet3d.py:86-161, conerf/model/feature_pyramid_net.py:97-127 (see regtr.py for the
name-by-name mapping).  The per-op Python path stays the description of record (and the fp32 parity path); this module only
changes who issues the launches, and tests/test_hip_trunk_exec.py checks the two bit for bit.
"""
import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from . import lib as L
from . import ops

OP_CONV, OP_BN, OP_MAXPOOL, OP_CONV_ROWS = 0, 1, 2, 3
RL_FIELDS = 8      # int64 fields per row list handed to the executor: rows, count, then the brick tile tables (tiles, ntiles, halo, nbr, rows_sorted, 0)
# DREG_SERIAL_STREAMS=1: no second stream for parameter gradients (every kernel alone on the GPU: what a per-kernel profile wants)
SERIAL_STREAMS = bool(int(__import__("os").environ.get("DREG_SERIAL_STREAMS", "0")))
KIND_NAMES = {0: "fwd", 1: "dgrad", 2: "wgrad"}
# Creation options (dreg_exec_opts of include/dreg_nerf.h) of the executors created from now on, on top of dreg_exec_default_opts:
# e.g. OPTS["sparse_stem"] = 0.  Per handle inside the library; this dict is the Python-side default the A/B tools and the "exact mode"
# tests change (then drop the model's _trunk_cache).  Empty in the product.
OPTS: Dict[str, int] = {}


class exec_opts:
    """``with exec_opts(sparse_stem=0):`` — executors created inside the block take these creation options (tests, A/B tools)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = dict(OPTS)
        OPTS.update(self.kw)
        return self

    def __exit__(self, *exc):
        OPTS.clear()
        OPTS.update(self.saved)
        return False


class _Recorder:
    """Stands in for dreg_nerf_amd.ops while regtr.fpn() runs once on shape placeholders."""

    def __init__(self, P: Dict[str, torch.Tensor], x_shape):
        self.P = P
        self.tensors = [tuple(x_shape)]
        self.ops = []
        self.params = []        # list of torch tensors (parameters / buffers)
        self.pindex = {}

    class T:  # shape placeholder
        def __init__(self, slot, shape):
            self.slot, self.shape = slot, tuple(shape)

    def _new(self, shape):
        self.tensors.append(tuple(shape))
        return _Recorder.T(len(self.tensors) - 1, shape)

    def _p(self, t: Optional[torch.Tensor]) -> int:
        if t is None:
            return -1
        k = id(t)
        if k not in self.pindex:
            self.pindex[k] = len(self.params)
            self.params.append(t)
        return self.pindex[k]

    def conv3d(self, x, w, bias=None, addend=None, stride=1, pad=0, out_rows: int = -1):
        B, D, H, W, _ = x.shape
        k = w.shape[2]
        od = tuple((d + 2 * pad - k) // stride + 1 for d in (D, H, W))
        y = self._new((B,) + od + (w.shape[0],))
        # out_rows (row-list id): a dense-layout convolution whose output is zero outside that list (the stem over a sparse volume)
        self.ops.append([OP_CONV, x.slot, y.slot, addend.slot if addend is not None else -1, self._p(w), self._p(bias), -1, -1, -1,
                         k, stride, pad, 0, 0, out_rows if (bias is None and addend is None) else -1, -1])
        return y

    def conv3d_rows(self, x, w, bias, addend, pad, out_rows: int, in_rows: int):
        y = self._new(x.shape[:4] + (w.shape[0],))
        self.ops.append([OP_CONV_ROWS, x.slot, y.slot, addend.slot if addend is not None else -1, self._p(w), self._p(bias), -1, -1, -1,
                         w.shape[2], 1, pad, 0, 0, out_rows, in_rows])
        return y

    def batchnorm(self, x, gamma, beta, running_mean, running_var, res=None, relu=True, train=True):
        y = self._new(x.shape)
        self.ops.append([OP_BN, x.slot, y.slot, res.slot if res is not None else -1, self._p(gamma), self._p(beta),
                         self._p(running_mean), self._p(running_var), -1, 0, 0, 0, int(relu), 0, -1, -1])
        return y

    def maxpool3d(self, x):
        B, D, H, W, C = x.shape
        y = self._new((B,) + tuple((d + 2 - 3) // 2 + 1 for d in (D, H, W)) + (C,))
        self.ops.append([OP_MAXPOOL, x.slot, y.slot, -1, -1, -1, -1, -1, -1, 3, 2, 1, 0, 0, -1, -1])
        return y


class TrunkExecutor:
    """One recorded program + its arena and packed-weight buffer."""

    def __init__(self, model, x_shape, sparse_head: int, with_grad: bool = True, stem_rows: bool = False, opts: Optional[Dict[str, int]] = None):
        self.lib = L.load()
        self.opts = L.ExecOpts()
        self.lib.dreg_exec_default_opts(ctypes.byref(self.opts))
        for k, v in {**OPTS, **(opts or {})}.items():
            if not hasattr(self.opts, k):
                raise L.DregError(f"unknown executor option {k!r}")
            setattr(self.opts, k, int(v))
        P = model._P()
        rec = _Recorder(P, x_shape)
        x0 = _Recorder.T(0, x_shape)
        self.nbt = []
        # row-list ids: 0..2 = S1, S2, S3 (head), 4, 5 = A, A2 (second pyramid level) — positions in ops.active_sets' tuple
        rl = None if not sparse_head else ops.RowSets((0, 1, 2, None) if sparse_head == 1 else (0, 1, 2, None, 4, 5))
        if rl is not None and stem_rows:
            rl.stem = len(rl)          # the stem's row list travels behind the head's (see _rowlist_array)
        self.stem_slot = rl.stem if (rl is not None and stem_rows) else -1
        out = model._fpn_program(rec, x0, rl, self.nbt, True)
        self.out_shape = out.shape
        self.rec = rec
        dev = next(model.parameters()).device
        self.device = dev
        tens = np.asarray(rec.tensors, dtype=np.int32)
        opsa = np.asarray(rec.ops, dtype=np.int32)
        prm = np.zeros((len(rec.params), 5), dtype=np.int64)
        self._param_refs = rec.params
        self._sig = []
        for i, t in enumerate(rec.params):
            assert t.is_contiguous() and t.dtype == torch.float32, "the executor reads fp32 master parameters in place"
            g = t.grad if (with_grad and isinstance(t, torch.nn.Parameter) and t.requires_grad) else None
            if g is not None:
                assert g.is_contiguous() and g.dtype == torch.float32
            elif with_grad and isinstance(t, torch.nn.Parameter) and t.requires_grad:
                raise L.DregError("TrunkExecutor needs preallocated .grad buffers (FlatAdamW) on every trainable parameter")
            prm[i, 0] = t.data_ptr()
            prm[i, 1] = g.data_ptr() if g is not None else 0
            prm[i, 2] = t.shape[0]
            prm[i, 3] = t.shape[1] if t.dim() > 1 else 1
            prm[i, 4] = t.shape[2] if t.dim() == 5 else 1
            self._sig.append((t.data_ptr(), prm[i, 1]))
        self.h = self.lib.dreg_exec_create_opts(tens.ctypes.data, len(rec.tensors), opsa.ctypes.data, len(rec.ops), prm.ctypes.data, len(rec.params),
                                                ctypes.byref(self.opts))
        if not self.h:
            raise L.DregError("dreg_exec_create_opts rejected the op program")
        self.arena_bytes = self.lib.dreg_exec_arena_bytes(self.h)
        self.arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=dev)
        self.pack = torch.empty(self.lib.dreg_exec_pack_bytes(self.h), dtype=torch.uint8, device=dev)
        n = self.lib.dreg_exec_num_packs(self.h)
        host = np.zeros((n, 12), dtype=np.int32)
        rowmap = np.zeros(self.lib.dreg_exec_pack_rows(self.h), dtype=np.int32)
        L.check(self.lib.dreg_exec_export_pack_table(self.h, host.ctypes.data, rowmap.ctypes.data, self.pack.data_ptr()), "dreg_exec_export_pack_table")
        self.pack_table = torch.from_numpy(host).to(dev)
        self.pack_rowmap = torch.from_numpy(rowmap).to(dev)
        self.out_slot = self.lib.dreg_exec_output_slot(self.h)
        self.out_off = self.lib.dreg_exec_tensor_offset(self.h, self.out_slot)
        self.pack_stamp = None
        self._timing = False
        self.last_row_counts = []
        self.sparse_head = sparse_head
        self.with_grad = with_grad
        self.labels = self._labels()
        if SERIAL_STREAMS:
            self.lib.dreg_exec_set_overlap(self.h, 0)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dreg_exec_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def still_valid(self) -> bool:
        """Parameter / gradient storage unchanged since the program was recorded?"""
        for t, (vp, gp) in zip(self._param_refs, self._sig):
            g = t.grad if (self.with_grad and isinstance(t, torch.nn.Parameter) and t.requires_grad) else None
            if t.data_ptr() != vp or (g.data_ptr() if g is not None else 0) != gp:
                return False
        return True

    def _labels(self):
        """(name, label, flops) per (op index, kind) for the HIP-event timing records."""
        out = {}
        lib = self.lib       # the name functions below must not capture `self`: a reference cycle would keep a dropped executor (and its
                             # 10 GB arena) alive until the cyclic collector runs (tools/stability.py SOAK_EDGE: out of memory after ~20 shapes)
        for i, o in enumerate(self.rec.ops):
            if o[0] not in (OP_CONV, OP_CONV_ROWS):
                continue
            x, y = self.rec.tensors[o[1]], self.rec.tensors[o[2]]
            w = self.rec.params[o[4]]
            B = x[0]
            k, s = o[9], o[10]
            cout, cin = w.shape[0], w.shape[1]
            M = B * y[1] * y[2] * y[3]
            # names of the LAUNCHED instantiations (dreg_conv3d_igemm_variant: the library's own dispatch rules).  Active-set launches:
            # the tile shape depends on the step's row count, so their names are functions of it, evaluated when the records are drained
            pd = o[11] if len(o) > 11 else k // 2
            is_rows = o[0] == OP_CONV_ROWS or o[14] >= 0       # (a dense-layout convolution on an output row list: the stem)

            def fwd_name(nr, B=B, x=x, y=y, cout=cout, k=k, s=s, pd=pd):
                return ops.igemm_kernel_name(lib, B, x[1], x[2], x[3], x[4], y[1], y[2], y[3], cout, k, s, pd, 0, nr, 0 if nr else 1, False, L.DT_BF16, False)

            def dgrad_name(nr, B=B, x=x, y=y, cout=cout, k=k, s=s, pd=pd, add=False):
                if s == 2:   # the parity-class form (dreg_conv3d_dgrad_s2): a 2^3-tap (or 1-tap) stride-1 convolution over dOut with 8 x Cin (or Cin) output channels
                    dc = tuple((d + 1) // 2 for d in x[1:4])
                    return ops.igemm_kernel_name(lib, B, y[1], y[2], y[3], cout, dc[0], dc[1], dc[2], (1 if k == 1 else 8) * x[4], 1 if k == 1 else 2, 1, 0, 0, 0, 0, False,
                                                 L.DT_BF16, False)
                return ops.igemm_kernel_name(lib, B, y[1], y[2], y[3], cout, x[1], x[2], x[3], x[4], k, 1, pd, 1, nr, 0 if nr else 1, add, L.DT_BF16, False)
            fname, dname = (fwd_name, dgrad_name) if is_rows else (fwd_name(0), dgrad_name(0))
            dname_add = None if is_rows else dgrad_name(0, add=True)      # the accumulating form (the executor reports which one ran)
            halo = self.lib.dreg_exec_op_halo(self.h, i)
            if halo & 1:
                fname = "conv3_halo64_kernel<bf16>" if cout == 64 else "conv3_halo_kernel<bf16>"
            if halo & 2:
                dname = "conv3_halo64_kernel<bf16>" if x[4] == 64 else "conv3_halo_kernel<bf16>"
            rows = "-rows" if is_rows else ""
            fl = 2.0 * M * cout * k ** 3 * cin
            # active-set launches: flops per row, scaled by the step's row count (list id) when the records are drained
            per_row = 2.0 * cout * k ** 3 * cin
            lo, li = (o[14], o[15]) if rows else (-1, -1)
            out[(i, 0)] = (fname, f"fwd{rows} B{B} {x[1]}x{x[2]}x{x[3]}x{x[4]}->{y[1]}x{y[2]}x{y[3]}x{cout} k{k}s{s}", fl, lo, per_row, f"conv3_brick_kernel<{cout},bf16>", None)
            out[(i, 1)] = (dname, f"dgrad{rows} B{B} {y[1]}x{y[2]}x{y[3]}x{cout}->{x[1]}x{x[2]}x{x[3]}x{cin} k{k}s{s}", fl, li, per_row, f"conv3_brick_kernel<{cin},bf16>", dname_add)
            def wgrad_name(nr, B=B, x=x, y=y, cout=cout, k=k, is_rows=is_rows, first=int(o[1] == 0)):
                var = lib.dreg_conv3d_wgrad_variant(B, y[1], y[2], y[3], x[4], cout, k, int(is_rows), nr, first)
                if var == 256256:
                    return f"conv_wgrad_glds_kernel<256,256,{'true' if is_rows else 'false'},8>"
                if var == 256128:
                    return "conv_wgrad_glds_kernel<256,128,false,4>"
                return f"conv_wgrad_glds_kernel<{var // 1000},{var % 1000},{'true' if is_rows else 'false'},4>"     # the template arguments rocprofv3 prints
            wname = wgrad_name if is_rows else wgrad_name(0)
            out[(i, 2)] = (wname, f"wgrad{rows} B{B} {x[1]}x{x[2]}x{x[3]}x{x[4]} g{y[1]}x{y[2]}x{y[3]}x{cout} k{k}s{s}", fl, lo, per_row, None, None)
        out[(-1, 3)] = ("wgrad_reduce_batched_kernel", "split sums of a backward range -> torch-layout gradients", 0.0, -1, 0.0, None, None)
        out[(-1, 4)] = ("bn_tail_batched_kernel<0>", "running statistics of the small BatchNorms of a forward pass", 0.0, -1, 0.0, None, None)
        out[(-1, 5)] = ("bn_tail_batched_kernel<1>", "dgamma / dbeta of the small BatchNorms of a backward range", 0.0, -1, 0.0, None, None)
        return out

    # ------------------------------------------------------------------ per-step calls
    def repack_if_stale(self):
        # FlatAdamW / load_state_dict bump the generation; torch's own in-place writes (dist.broadcast, p.data.copy_, manual edits)
        # move the parameters' version counters
        stamp = (ops._weight_generation, sum(t._version for t in self._param_refs))
        if self.pack_stamp != stamp:
            L.check(self.lib.dreg_exec_repack(self.h, L.ptr(self.pack_table), L.ptr(self.pack_rowmap), L.stream()), "dreg_exec_repack")
            self.pack_stamp = stamp

    def repack_after_update(self):
        """Right after an optimizer update (train_step): refresh the packs inside ops.pack_region() — on the side stream, under the
        next step's input staging; forward() waits for it (ops.wait_packs) instead of repacking in front of its first convolution."""
        with ops.pack_region():
            self.repack_if_stale()

    @staticmethod
    def _rowlist_array(rows):
        if rows is None:
            return None, 0
        n = len(rows)
        stem = getattr(rows, "stem", None)
        a = (ctypes.c_int64 * (RL_FIELDS * (n + (1 if stem is not None else 0))))()
        if stem is not None:            # list id n: the stem's output rows
            a[RL_FIELDS * n], a[RL_FIELDS * n + 1] = stem.data_ptr(), stem.shape[0]
        tiles = getattr(rows, "tiles", None) or {}
        for i in range(n):
            if i == 3 or rows[i] is None:   # slot 3 is map1 (not a row list)
                continue
            a[RL_FIELDS * i] = rows[i].data_ptr()
            a[RL_FIELDS * i + 1] = rows[i].shape[0]
            bt = tiles.get(i)
            if bt is not None and bt.ntiles > 0:   # tile tables of csrc/conv_brick.hip for this row set
                a[RL_FIELDS * i + 2], a[RL_FIELDS * i + 3] = bt.tiles.data_ptr(), bt.ntiles
                a[RL_FIELDS * i + 4], a[RL_FIELDS * i + 5], a[RL_FIELDS * i + 6] = bt.halo.data_ptr(), bt.nbr.data_ptr(), bt.rows_sorted.data_ptr()
        return a, n + (1 if stem is not None else 0)

    def forward(self, x: torch.Tensor, rows, train: bool) -> torch.Tensor:
        ops.wait_packs()
        self.repack_if_stale()
        want = ops.PROFILER is not None and ops.PROFILER.enabled
        if want != self._timing:
            self.set_timing(want)
        self.last_row_counts = [int(r.shape[0]) if (r is not None and i != 3) else 0 for i, r in enumerate(rows)] if rows is not None else []
        if rows is not None and getattr(rows, "stem", None) is not None:
            self.last_row_counts.append(int(rows.stem.shape[0]))
        ra, n = self._rowlist_array(rows)
        self._guarded(self.lib.dreg_exec_forward(self.h, L.ptr(self.arena), self.arena_bytes, L.ptr(self.pack), L.ptr(x),
                                                 ctypes.addressof(ra) if ra is not None else None, n, int(train), L.stream()), "dreg_exec_forward")
        # one arena holds the activations of ONE forward pass: a backward must belong to the latest one (see backward)
        self.generation = getattr(self, "generation", 0) + 1
        nbytes = int(np.prod(self.out_shape)) * 2
        return self.arena[self.out_off:self.out_off + nbytes].view(torch.bfloat16).view(self.out_shape)

    def _guarded(self, rc: int, name: str):
        """L.check + the guard mode's reports (creation option guard >= 1: every region of the arena is followed by a poisoned band)."""
        if rc == -3:                       # DREG_EGUARD: a guard = 2 pass stopped behind the first op after which a band had changed
            op, ps, band = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
            self.lib.dreg_exec_guard_last(self.h, ctypes.byref(op), ctypes.byref(ps), ctypes.byref(band))
            raise L.DregError(f"{name}: guard band overwritten behind op {op.value} ({'backward' if ps.value else 'forward'} pass; "
                              f"{self.rec.ops[op.value] if 0 <= op.value < len(self.rec.ops) else 'pass tail'}): {self.guard_describe(band.value)}")
        L.check(rc, name)
        if self.opts.guard:
            bad = self.guard_check()
            if bad[0]:
                raise L.DregError(f"{name}: {bad[0]} guard band(s) overwritten ({bad[3]} 16-byte words); first: {self.guard_describe(bad[1])} + {bad[2]} bytes")

    def guard_check(self):
        """Scan the guard bands now.  dreg_exec_guard_check synchronises the caller's stream only; the weight / bias gradient launches of a backward
        pass run on the executor's aux stream, so that stream is drained first — a stray write from one of them is attributed to THIS pass."""
        if self.device.type == "cuda":
            aux_stream(self.device).synchronize()
        out = (ctypes.c_longlong * 4)()
        L.check(self.lib.dreg_exec_guard_check(self.h, L.ptr(self.arena), out, L.stream()), "dreg_exec_guard_check")
        return list(out)

    def guard_describe(self, band: int) -> str:
        buf = ctypes.create_string_buffer(256)
        self.lib.dreg_exec_guard_describe(self.h, int(band), buf, 256)
        return buf.value.decode()

    def backward(self, x: torch.Tensor, rows, grad_out: torch.Tensor, generation: Optional[int] = None):
        if generation is not None and generation != getattr(self, "generation", 0):
            raise L.DregError("TrunkExecutor.backward: the arena holds the activations of a LATER forward pass than the one this gradient belongs to "
                              "(two grad-mode forwards before a backward — gradient accumulation over several forwards, retain_graph — need one "
                              "executor each; the product step runs one forward and one backward per step)")
        ra, n = self._rowlist_array(rows)
        sync = ops.GRAD_SYNC
        self._flush_deferred()
        if sync is None:
            self._guarded(self.lib.dreg_exec_backward(self.h, L.ptr(self.arena), self.arena_bytes, L.ptr(self.pack), L.ptr(x), L.ptr(grad_out),
                                                      ctypes.addressof(ra) if ra is not None else None, n, L.stream(), aux_stream(self.device).cuda_stream),
                          "dreg_exec_backward")
            return
        # data-parallel step: backward in segments; after each, the gradient buckets it completed start their all-reduce
        plan = self._sync_plan(sync)
        sync.ready(plan["above"])              # everything past the trunk's parameters (transformer, decoder) was produced before this call
        hi = len(self.rec.ops)
        for k, (lo_op, done_from) in enumerate(plan["cuts"]):
            flags = (1 if k == 0 else 0) | (2 if k == len(plan["cuts"]) - 1 else 0)
            self._guarded(self.lib.dreg_exec_backward_range(self.h, L.ptr(self.arena), self.arena_bytes, L.ptr(self.pack), L.ptr(x), L.ptr(grad_out),
                                                            ctypes.addressof(ra) if ra is not None else None, n, L.stream(), aux_stream(self.device).cuda_stream,
                                                            lo_op, hi, flags), "dreg_exec_backward_range")
            hi = lo_op
            sync.ready(done_from)

    def _flush_deferred(self):
        """The point-set half's deferred weight-gradient sums (ops.conv_wgrad(defer=True)) are all enqueued when the trunk's backward
        starts: one launch on the stream they were produced on adds them to the gradients."""
        if ops._PENDING_REDUCE:
            with torch.cuda.stream(aux_stream(self.device)):
                ops.flush_wgrad_reduce()

    def _sync_plan(self, sync):
        """Segments of the backward pass for GradSync: ops are processed last to first; after the segment ending at op `lo_op` every
        parameter at flat-gradient offsets >= done_from is final.  Cuts are placed where done_from crosses a bucket boundary."""
        key = (id(sync.opt), tuple(sync.buckets))
        if getattr(self, "_plan_key", None) == key:
            return self._plan
        spans = {}                                      # parameter index -> (offset, numel) in the flat gradient buffer
        for i, t in enumerate(self.rec.params):
            g = t.grad if (isinstance(t, torch.nn.Parameter) and t.requires_grad) else None
            off = sync.opt.offset_of(g.data_ptr()) if g is not None else -1
            if off >= 0:
                spans[i] = (off, t.numel())
        touched = [[p for p in (o[4], o[5]) if p in spans] for o in self.rec.ops]          # conv: w, b; BatchNorm: gamma, beta
        above = max((off + n for off, n in spans.values()), default=0)
        pending = dict(spans)
        bounds = sorted({lo for lo, _ in sync.buckets}, reverse=True)
        cuts, last_done = [], above
        for i in range(len(self.rec.ops) - 1, -1, -1):
            for p in touched[i]:
                pending.pop(p, None)
            done_from = max((off + n for off, n in pending.values()), default=0)          # everything above the highest unfinished parameter
            crossed = [b for b in bounds if done_from <= b < last_done]
            if crossed or i == 0:
                cuts.append((i, done_from))
                last_done = done_from
        self._plan_key, self._plan = key, {"above": above, "cuts": cuts}
        return self._plan

    # ------------------------------------------------------------------ timing (bench.py's roofline line)
    def set_input_row_occupancy(self, row_occ):
        """uint8 [B, D/2, H/2] output-row flags of the stem for the next forward / backward (None = compute every row).  The tensor
        is kept alive here until it is replaced: the weight gradient on the second stream reads it during backward."""
        self._row_occ = row_occ
        self.lib.dreg_exec_set_input_row_occupancy(self.h, L.ptr(row_occ) if row_occ is not None else None)

    def set_timing(self, on: bool):
        """HIP-event brackets around every convolution launch.  While they are on, the weight-gradient launches stay on the
        caller's stream (no second stream), so a bracket measures its kernel alone rather than two kernels sharing the CUs."""
        self._timing = bool(on)
        self.lib.dreg_exec_set_timing(self.h, int(on))
        self.lib.dreg_exec_set_overlap(self.h, int(not on and not SERIAL_STREAMS))

    def drain_timings(self, profiler: "ops.KernelTimer"):
        """After a device synchronisation: move the executor's HIP-event records into a KernelTimer-compatible store."""
        cap = 65536
        ok = (ctypes.c_int * (3 * cap))()
        ms = (ctypes.c_float * cap)()
        n = self.lib.dreg_exec_read_timings(self.h, ok, ms, cap)
        for i in range(n):
            name, label, fl, lid, per_row, brick_name, add_name = self.labels[(ok[3 * i], ok[3 * i + 1])]
            if lid >= 0:   # active-set launch: algorithmic flops and the tile shape follow the (last) step's row count
                nr = self.last_row_counts[lid]
                fl, label = per_row * nr, f"{label} rows{nr}"
                if callable(name):
                    name = name(nr)
            if ok[3 * i + 2] == 2 and brick_name:      # the launch ran on csrc/conv_brick.hip (the executor says so)
                name = brick_name
            elif ok[3 * i + 2] == 1 and add_name:      # an accumulating data gradient: dispatched with an addend
                name = add_name
            profiler.add_measured(name, label, fl, ms[i])


_AUX = {}
AUX_CU_MASK = int(__import__("os").environ.get("DREG_AUX_CU_MASK", "0"))     # measurement only, see aux_stream


def aux_stream(device) -> "torch.cuda.Stream":
    """The process-wide second stream (per device) for parameter-gradient launches: shared by every executor and by
    train_step's point-set half, created once so that its hardware queue never changes."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _AUX.get(key)
    if st is None:
        # lowest HIP priority (ROCm: 1; torch.cuda.Stream only offers 0 / -1): the weight-gradient launches fill what the main chain
        # leaves idle instead of competing with it — 0.03-0.09 ms of the step (the main chain on a HIGH-priority stream instead: +0.07 ms).
        # DREG_AUX_PRIORITY=torch: a plain torch stream.
        prio = __import__("os").environ.get("DREG_AUX_PRIORITY", "low")
        st = None
        if AUX_CU_MASK and device.type == "cuda":
            # MEASUREMENT (tools/ab_step.py aux:cumask; round-5 review item 6): the second stream restricted to a subset of the 256 CUs
            # (hipExtStreamCreateWithCUMask: 8 words, bit i = CU i), so that weight-gradient workgroups cannot take LDS / issue slots on the others
            words = {1: [0xFFFFFFFF] * 4 + [0] * 4, 2: [0x55555555] * 8, 3: [0xFFFFFFFF] * 2 + [0] * 6, 4: [0x11111111] * 8, 5: [0xFFFFFFFF] * 6 + [0] * 2,
                     6: [0x0000FFFF] * 8, 7: [0x000000FF] * 8}[int(AUX_CU_MASK)]
            hip = ctypes.CDLL("libamdhip64.so")
            h = ctypes.c_void_p()
            arr = (ctypes.c_uint32 * 8)(*words)
            with torch.cuda.device(device):
                rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, arr)
            if rc != 0 or not h.value:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
            st = torch.cuda.ExternalStream(h.value, device=device)
        if st is None and prio != "torch" and device.type == "cuda":
            try:
                hip = ctypes.CDLL("libamdhip64.so")
                least, greatest = ctypes.c_int(), ctypes.c_int()
                h = ctypes.c_void_p()
                with torch.cuda.device(device):
                    ok = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0
                    ok = ok and hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value if prio == "low" else int(prio)) == 0   # 1 = hipStreamNonBlocking
                if ok and h.value:
                    st = torch.cuda.ExternalStream(h.value, device=device)
            except (OSError, ValueError):
                st = None
        if st is None:
            st = torch.cuda.Stream(device=device)
        _AUX[key] = st
    return st


class _TrunkFn(torch.autograd.Function):
    """P1 = FPN3D(x) through the native executor.  `anchor` is any trainable parameter: it only keeps this node in the autograd
    graph (the executor accumulates every parameter gradient itself, straight into the .grad buffers)."""

    @staticmethod
    def forward(ctx, x, anchor, ex: TrunkExecutor, rows, train: bool):
        ctx.ex, ctx.rows = ex, rows
        ctx.save_for_backward(x)
        y = ex.forward(x, rows, train)
        ctx.generation = ex.generation
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        ctx.ex.backward(x, ctx.rows, g.contiguous(), ctx.generation)
        return None, None, None, None, None


def run_trunk(ex: TrunkExecutor, x, anchor, rows, train):
    return _TrunkFn.apply(x, anchor, ex, rows, train)

"""torch.autograd.Function wrappers over the C ABI (raw pointers + the current HIP stream).

Activations are NDHWC tensors ``[B, D, H, W, C]`` in bf16 (production) or fp32 (exact-f32 MFMA parity
mode).  Master weights stay fp32 in torch layout; the kernels consume per-step packed copies.
"""
import os
from typing import Optional

import numpy as np
import torch

from . import lib as L


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are launched on (torch's current stream).
    bench.py enables it over the timed region; entries are (kernel name, shape label, algorithmic flops, start, end)."""

    def __init__(self):
        self.entries = []
        self.measured = []
        self._overhead = None
        self.enabled = True   # bench.py samples: events on the first timed steps only, so the timer does not slow the rest

    def record(self, name, label, flops):
        if not self.enabled:
            return None
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        self.entries.append((name, label, flops, start, end))
        return start, end

    def add_measured(self, name, label, flops, ms):
        """A launch timed elsewhere (the native trunk executor's own HIP events)."""
        self.measured.append((name, label, flops, ms))

    def bracket_overhead_ms(self):
        """Elapsed time of an EMPTY event bracket on the current stream (median of 32): what every bracketed launch carries on
        top of its kernel time; subtracted in summary() so that tiny launches are not inflated against rocprofv3's durations."""
        if self._overhead is None:
            torch.cuda.synchronize()
            pairs = []
            for _ in range(32):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                e.record()
                pairs.append((s, e))
            torch.cuda.synchronize()
            v = sorted(s.elapsed_time(e) for s, e in pairs)
            self._overhead = v[len(v) // 2]
        return self._overhead

    def summary(self, subtract_overhead: bool = True):
        torch.cuda.synchronize()
        oh = self.bracket_overhead_ms() if subtract_overhead else 0.0
        by_name, by_label = {}, {}
        for name, label, flops, ms in self.measured:
            ms = max(ms - oh, 0.0)
            for d, k in ((by_name, name), (by_label, (name, label))):
                a = d.setdefault(k, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += ms
                a[2] += flops
        for name, label, flops, s, e in self.entries:
            ms = max(s.elapsed_time(e) - oh, 0.0)
            for d, k in ((by_name, name), (by_label, (name, label))):
                a = d.setdefault(k, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += ms
                a[2] += flops
        return by_name, by_label


PROFILER: Optional[KernelTimer] = None


# --------------------------------------------------------------------------- weight packing (cached per parameter version)
import weakref

_pack_cache = {}
_weight_generation = 0


def clear_pack_cache():
    global _pack_table
    _DP1_CACHE.clear()
    _pack_cache.clear()
    _halo_cache.clear()
    _pack_table = None


PACK_STREAM = None        # side stream for the repack that follows an optimizer update (train_step sets it; None = current stream)
_PACK_EVENTS = []         # events of the repacks since the last optimizer update; kept until the NEXT update's repack replaces them
_PACK_WAITED = set()      # streams that already wait behind every event in _PACK_EVENTS
_PACK_EPOCH = [0, -1]     # [optimizer updates seen, update the events in _PACK_EVENTS belong to]


class pack_region:
    """`with pack_region():` — pack launches inside run on PACK_STREAM (after everything already enqueued on the current stream:
    the optimizer update) and the first consumer of a pack waits for them (wait_packs).  The repack is latency-bound (~0.5 ms at
    ~1 TB/s); on the side stream it runs under the input staging of the next step instead of in front of its first convolution."""

    def __enter__(self):
        self.side = PACK_STREAM if (PACK_STREAM is not None and torch.cuda.is_available()) else None
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.side.device))
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(self.side)
            if _PACK_EPOCH[1] != _PACK_EPOCH[0]:        # first repack after an optimizer update: the older events are all behind it
                _PACK_EVENTS.clear()                    # (same side stream, in order)
                _PACK_EPOCH[1] = _PACK_EPOCH[0]
            _PACK_EVENTS.append(ev)
            _PACK_WAITED.clear()
            self.ctx.__exit__(*exc)
        return False


def wait_packs():
    """Called in front of every consumer of a weight pack: the CURRENT stream waits for the
    side-stream repacks.  The events stay until the next update's repack replaces them, so a consumer on any other stream (the
    parameter-gradient stream, the geometry stream, a loader thread) waits too — once per stream and repack."""
    if _PACK_EVENTS:
        st = torch.cuda.current_stream()
        key = (st.device.index, st.cuda_stream)
        if key not in _PACK_WAITED:
            for ev in _PACK_EVENTS:
                st.wait_event(ev)
            _PACK_WAITED.add(key)


def bump_weight_generation():
    """Called by optimizers that update parameters behind torch's version counters (FlatAdamW)."""
    global _weight_generation
    _weight_generation += 1
    _PACK_EPOCH[0] += 1


def packed_weight(w: torch.Tensor, cin_pad: int, for_dgrad: bool, dt: int) -> torch.Tensor:
    """w: fp32 [Cout, Cin, k, k, k] (or [Cout, Cin] for linear layers, possibly a row slice of a parameter).
    The pack is cached against the owning parameter object (weak reference) and its in-place version counter;
    repack_all() refreshes every cached pack in one launch after an optimizer update."""
    wait_packs()
    base = w._base if w._base is not None else w
    key = (id(base), w.storage_offset(), tuple(w.shape), cin_pad, for_dgrad, dt)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is base and hit[1] == (base._version, _weight_generation):
        return hit[2]
    lib = L.load()
    cout, cin_real = w.shape[0], w.shape[1]
    ksz = w.shape[2] if w.dim() == 5 else 1
    wd = w.detach().contiguous()
    if not for_dgrad:
        kpad = lib.dreg_conv3d_kpad(ksz, cin_pad, dt)
        out = hit[2] if hit is not None and hit[0]() is base else torch.empty(cout, kpad, dtype=L.torch_dtype(dt), device=w.device)
    elif for_dgrad == 2:   # parity-class pack of the stride-2 data gradient: rows (class, ci), K = (2^3 taps | 1) x Cout
        kpad = lib.dreg_conv3d_kpad(1 if ksz == 1 else 2, cout, dt)
        nrow = (1 if ksz == 1 else 8) * cin_real
        out = hit[2] if hit is not None and hit[0]() is base else torch.empty(nrow, kpad, dtype=L.torch_dtype(dt), device=w.device)
    else:
        kpad = lib.dreg_conv3d_kpad(ksz, cout, dt)
        out = hit[2] if hit is not None and hit[0]() is base else torch.empty(cin_real, kpad, dtype=L.torch_dtype(dt), device=w.device)
    L.check(lib.dreg_pack_conv_weight(L.ptr(wd), L.ptr(out), cout, cin_real, cin_pad, ksz, int(for_dgrad), dt, L.stream()),
            "dreg_pack_conv_weight")
    if len(_pack_cache) > 4096:  # entries of dead parameters
        for k in [k for k, v in _pack_cache.items() if v[0]() is None]:
            del _pack_cache[k]
    global _pack_table
    if hit is None or hit[0]() is not base:
        _pack_table = None
    # (weakref, stamp, packed, contiguous?, offset of the slice inside its parameter in elements, desc fields)
    desc = (cout, cin_real, cout if for_dgrad else cin_pad, ksz ** 3, int(for_dgrad), kpad, dt)
    _pack_cache[key] = (weakref.ref(base), (base._version, _weight_generation), out,
                        w.is_contiguous() and base.is_contiguous(), w.storage_offset() - base.storage_offset(), desc)
    return out


_pack_table = None  # (device descriptor tensor, n, total_blocks, [(key, src_ptr)]) for repack_all


def repack_all(device=None):
    """Refresh every cached weight pack with ONE kernel launch (dreg_pack_conv_weights_batched) — called by FlatAdamW right
    after its update, so the ~200 per-layer pack launches of a step collapse into one.  Packs of non-contiguous slices and
    of dead parameters are skipped (they fall back to the per-call path on their next use)."""
    global _pack_table
    import numpy as np
    lib = L.load()
    live = []
    for key, ent in _pack_cache.items():
        base = ent[0]()
        if base is None or not ent[3] or not base.is_cuda or (device is not None and base.device != device):
            continue
        live.append((key, base.data_ptr() + 4 * ent[4], base))
    if not live:
        return
    sig = [(k, p) for k, p, _ in live]
    if _pack_table is None or _pack_table[3] != sig:
        rec = np.zeros((len(live), 12), dtype=np.int32)
        row0 = max_floats = 0
        rowmap = []
        for r, (key, src, base) in enumerate(live):
            ent = _pack_cache[key]
            cout, cin_real, inner, ntaps, fd, kpad, dt = ent[5]
            rec[r, 0:2] = np.frombuffer(np.int64(src).tobytes(), dtype=np.int32)
            rec[r, 2:4] = np.frombuffer(np.int64(ent[2].data_ptr()).tobytes(), dtype=np.int32)
            rec[r, 4:12] = (cout, cin_real, inner, ntaps, fd, kpad, dt, row0)
            nrow = (cin_real * (8 if (fd == 2 and ntaps > 1) else 1)) if fd else cout
            rowmap.append(np.full(nrow, r, dtype=np.int32))
            row0 += nrow
            if ntaps > 1 and dt == 0:
                max_floats = max(max_floats, min(cout if fd else cin_real, 64) * ntaps)
        dev_ = live[0][2].device
        _pack_table = (torch.from_numpy(rec).to(dev_), len(live), row0, sig, max_floats, torch.from_numpy(np.concatenate(rowmap)).to(dev_))
    tab, n, nrows, _, max_floats, rmap = _pack_table
    L.check(lib.dreg_pack_conv_weights_batched(L.ptr(tab), n, nrows, max_floats, L.ptr(rmap), L.stream()), "dreg_pack_conv_weights_batched")
    for key, _, base in live:
        ent = _pack_cache[key]
        _pack_cache[key] = (ent[0], (base._version, _weight_generation)) + ent[2:]


def _out_dim(i, k, s, p):
    return (i + 2 * p - k) // s + 1


# --------------------------------------------------------------------------- convolution
def conv_igemm(x, wpk, bias, addend, out_shape, cin, cout, ksz, stride, pad, transposed, relu=False, out_f32=False,
               flop_cin=None, add_same=False):
    """Raw launcher.  x: [B,Di,Hi,Wi,cin]; returns [B,*out_shape,cout]."""
    lib = L.load()
    dt = L.dt_of(x)
    B, Di, Hi, Wi = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
    Do, Ho, Wo = out_shape
    odt = torch.float32 if (out_f32 or dt == L.DT_F32) else torch.bfloat16
    out = torch.empty(B, Do, Ho, Wo, cout, dtype=odt, device=x.device)
    Da = Ha = Wa = 0
    if addend is not None:
        assert addend.dtype == odt and addend.shape[0] == B and addend.shape[4] == cout
        Da, Ha, Wa = addend.shape[1], addend.shape[2], addend.shape[3]
        if add_same:
            assert (Da, Ha, Wa) == (Do, Ho, Wo)
        else:
            assert 2 * Da >= Do and 2 * Ha >= Ho and 2 * Wa >= Wo
    ev = None
    if PROFILER is not None:
        name = igemm_kernel_name(lib, B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, transposed, 0, 1, addend is not None, dt, odt == torch.float32)
        label = f"{'dgrad' if transposed else 'fwd'} B{B} {Di}x{Hi}x{Wi}x{cin}->{Do}x{Ho}x{Wo}x{cout} k{ksz}s{stride}"
        flops = 2.0 * B * Do * Ho * Wo * cout * (ksz ** 3) * (flop_cin or cin)
        if transposed and stride == 2:
            flops /= 8.0  # only 1/8 of the taps of a stride-2 data gradient are algorithmically non-zero
        ev = PROFILER.record(name, label, flops)
        if ev is not None:
            ev[0].record()
    nws = lib.dreg_conv3d_igemm_workspace_bytes(B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, int(transposed),
                                                int(addend is not None), dt) if Do * Ho * Wo < 2048 else 0
    ws = _ws(nws, x.device) if nws else None
    L.check(lib.dreg_conv3d_igemm_ws(L.ptr(x), L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(addend),
                                     B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, int(transposed), int(relu),
                                     Da, Ha, Wa, int(add_same), dt, int(out_f32 and dt == L.DT_BF16), L.ptr(ws), nws, L.stream()),
            "dreg_conv3d_igemm_ws")
    if ev is not None:
        ev[1].record()
    return out


_halo_cache = {}


def halo_packed_weight(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """bf16 [chunk][tap][rows][32] pack (rows = 256 or 64 output channels) of a 3^3 weight for the halo kernels (forward, or the flipped-tap data-gradient form), cached
    against the parameter's version / the optimizer generation like packed_weight()."""
    wait_packs()
    base = w._base if w._base is not None else w
    key = (id(base), w.storage_offset(), tuple(w.shape), bool(transposed))
    hit = _halo_cache.get(key)
    stamp = (base._version, _weight_generation)
    if hit is not None and hit[0]() is base and hit[1] == stamp:
        return hit[2]
    lib = L.load()
    cout, cin = w.shape[0], w.shape[1]
    out = hit[2] if hit is not None and hit[0]() is base else \
        torch.empty(lib.dreg_conv3_halo_pack_bytes_n(cout if transposed else cin, cin if transposed else cout) // 2, dtype=torch.bfloat16, device=w.device)
    L.check(lib.dreg_pack_conv_weight_halo(L.ptr(w.detach().contiguous()), L.ptr(out), cout, cin, int(transposed), L.stream()), "dreg_pack_conv_weight_halo")
    if len(_halo_cache) > 64:
        for k in [k for k, v in _halo_cache.items() if v[0]() is None]:
            del _halo_cache[k]
    _halo_cache[key] = (weakref.ref(base), stamp, out)
    return out


def halo_applies(x_shape, cin, cout, ksz, stride, pad, dt) -> bool:
    """The dense 3^3 convolutions with 256 (or 64) output channels on large volumes run on the halo kernels (csrc/conv_halo.hip) — the same
    predicate the native trunk executor uses."""
    return dt == L.DT_BF16 and bool(L.load().dreg_conv3_halo_use(x_shape[0], x_shape[1], x_shape[2], x_shape[3], cin, cout, ksz, stride, pad))


def conv_halo(x, w, bias, addend, transposed: bool, add_same: bool = False, out_f32: bool = False):
    """x [B,D,H,W,C] bf16 -> [B,D,H,W,n] (n = 256 or 64): 3^3 / stride 1 / pad 1 with w [Cout,Cin,3,3,3] (transposed: the data gradient, x = dOut, n = Cin)."""
    lib = L.load()
    B, D, H, W, C = x.shape
    n = w.shape[1] if transposed else w.shape[0]
    out = torch.empty(B, D, H, W, n, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    Da = Ha = Wa = 0
    if addend is not None:
        assert addend.dtype == out.dtype and addend.shape[0] == B and addend.shape[4] == n
        Da, Ha, Wa = addend.shape[1:4]
    ev = None
    if PROFILER is not None:
        label = f"{'dgrad' if transposed else 'fwd'} B{B} {D}x{H}x{W}x{C}->{D}x{H}x{W}x{n} k3s1"
        ev = PROFILER.record("conv3_halo64_kernel<bf16>" if n == 64 else "conv3_halo_kernel<bf16>", label, 2.0 * B * D * H * W * n * 27 * C)
        if ev is not None:
            ev[0].record()
    L.check(lib.dreg_conv3_halo_n(L.ptr(x), L.ptr(halo_packed_weight(w, transposed)), L.ptr(out), L.ptr(bias), L.ptr(addend), B, D, H, W, C, n,
                                  Da, Ha, Wa, int(add_same), int(out_f32), L.stream()), "dreg_conv3_halo_n")
    if ev is not None:
        ev[1].record()
    return out


def conv_dgrad_s2(g, wpk_class, x_shape, cout, ksz, pad):
    """Data gradient of a stride-2 convolution through the parity-class form (dreg_conv3d_dgrad_s2): g [B,Do,Ho,Wo,cout] bf16."""
    lib = L.load()
    B, Di, Hi, Wi, cin = x_shape
    Do, Ho, Wo = g.shape[1:4]
    gx = torch.empty(x_shape, dtype=g.dtype, device=g.device)
    ev = None
    if PROFILER is not None:
        # algorithmic flops: the 27 (or 1) taps of the true data gradient, not the 64 of the padded class form
        ev = PROFILER.record("conv_igemm_glds_kernel<bf16,s2-dgrad>", f"dgrad-s2 B{B} {Do}x{Ho}x{Wo}x{cout}->{Di}x{Hi}x{Wi}x{cin} k{ksz}s2",
                             2.0 * B * Do * Ho * Wo * cout * (ksz ** 3) * cin)
        if ev is not None:
            ev[0].record()
    L.check(lib.dreg_conv3d_dgrad_s2(L.ptr(g), L.ptr(wpk_class), L.ptr(gx), B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, pad, L.stream()),
            "dreg_conv3d_dgrad_s2")
    if ev is not None:
        ev[1].record()
    return gx


_PENDING_REDUCE = []      # (workspace tensor, record) of the deferred weight-gradient launches since the last flush
_REDUCE_DT = np.dtype([("part", "<u8"), ("dw", "<u8"), ("nsplit", "<i4"), ("Cout", "<i4"), ("Kpad", "<i4"), ("ntaps", "<i4"),
                       ("Cin", "<i4"), ("Cin_real", "<i4"), ("accumulate", "<i4"), ("block0", "<i4")])


def flush_wgrad_reduce():
    """Sum the split partials of every deferred weight gradient (conv_wgrad(..., defer=True)) into their gradient tensors: one
    launch (dreg_wgrad_reduce_batched) on the current stream, which must be the stream the partials were produced on."""
    if not _PENDING_REDUCE:
        return
    lib = L.load()
    # a gradient that received several contributions (a layer applied twice) is summed by successive launches, in call order:
    # within one launch every record owns its destination
    rounds = []
    for rec in _PENDING_REDUCE:
        key = rec[1].data_ptr()
        for seen, lst in rounds:
            if key not in seen:
                seen.add(key); lst.append(rec)
                break
        else:
            rounds.append(({key}, [rec]))
    dev = _PENDING_REDUCE[0][0].device
    st = torch.cuda.current_stream(dev)
    for _, lst in rounds:
        recs = np.zeros(len(lst), dtype=_REDUCE_DT)
        blocks = 0
        for i, (ws, dw, nsplit, cout, kpad, ntaps, cin, cin_real, ksz) in enumerate(lst):
            recs[i] = (ws.data_ptr(), dw.data_ptr(), nsplit, cout, kpad, ntaps, cin, cin_real, 1, blocks)
            blocks += lib.dreg_wgrad_reduce_blocks(cout, cin_real, ksz, nsplit)
        table = torch.from_numpy(recs.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
        L.check(lib.dreg_wgrad_reduce_batched(L.ptr(table), len(lst), 0, blocks, L.stream()), "dreg_wgrad_reduce_batched")
        table.record_stream(st)
    for ws, *_ in _PENDING_REDUCE:
        ws.record_stream(st)
    _PENDING_REDUCE.clear()


def conv_wgrad(gout, x, w_shape, cin_pad, ksz, stride, pad, use_tr=True, accumulate_into=None, defer=False):
    """accumulate_into: an existing fp32 gradient tensor of w_shape (e.g. a view of the flat gradient buffer) that the
    reduce kernel adds into; the function then returns None.  defer (bf16, with accumulate_into): only the split partials are
    produced now; flush_wgrad_reduce() adds the sums of all deferred layers with one launch."""
    lib = L.load()
    dt = L.dt_of(x)
    B, Di, Hi, Wi = x.shape[:4]
    Do, Ho, Wo, cout = gout.shape[1:]
    cin_real = w_shape[1]
    nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, Do, Ho, Wo, cin_pad, cout, ksz, dt)
    ws = _ws(nbytes, x.device)
    if defer and accumulate_into is not None and dt == L.DT_BF16 and use_tr and accumulate_into.is_contiguous():
        L.check(lib.dreg_conv3d_wgrad_partials(L.ptr(gout), L.ptr(x), L.ptr(ws), nbytes, None, 0, B, Di, Hi, Wi, cin_pad, cin_real,
                                               Do, Ho, Wo, cout, ksz, stride, pad, None, L.stream()), "dreg_conv3d_wgrad_partials")
        _PENDING_REDUCE.append((ws, accumulate_into, lib.dreg_conv3d_wgrad_splits(B, Do, Ho, Wo, cin_pad, cout, ksz, dt), cout,
                                lib.dreg_conv3d_kpad(ksz, cin_pad, dt), ksz ** 3, cin_pad, cin_real, ksz))
        return None
    dw = accumulate_into if accumulate_into is not None else torch.empty(w_shape, dtype=torch.float32, device=x.device)
    ev = None
    if PROFILER is not None:
        tn = "bf16" if dt == L.DT_BF16 else "f32"
        label = f"wgrad B{B} {Di}x{Hi}x{Wi}x{cin_pad} g{Do}x{Ho}x{Wo}x{cout} k{ksz}s{stride}"
        if dt == L.DT_BF16 and use_tr:
            var = lib.dreg_conv3d_wgrad_variant(B, Do, Ho, Wo, cin_pad, cout, ksz, 0, 0, 0)
            wname = "conv_wgrad_glds_kernel<256,256,false,8>+reduce" if var == 256256 else f"conv_wgrad_glds_kernel<{var // 1000},{var % 1000},false,4>+reduce"   # 256128 -> <256,128,...>
        else:
            wname = f"conv_wgrad_kernel<{tn}>+reduce"
        ev = PROFILER.record(wname, label, 2.0 * B * Do * Ho * Wo * cout * (ksz ** 3) * cin_real)
        if ev is not None:
            ev[0].record()
    L.check(lib.dreg_conv3d_wgrad(L.ptr(gout), L.ptr(x), L.ptr(dw), L.ptr(ws), nbytes, B, Di, Hi, Wi, cin_pad, cin_real,
                                  Do, Ho, Wo, cout, ksz, stride, pad, int(accumulate_into is not None), dt,
                                  int(use_tr and dt == L.DT_BF16), L.stream()),
            "dreg_conv3d_wgrad")
    if ev is not None:
        ev[1].record()
    return None if accumulate_into is not None else dw


def colsum(g2d: torch.Tensor, accumulate_into=None):
    lib = L.load()
    M, C = g2d.shape
    ws = _ws(lib.dreg_colsum_workspace_bytes(M, C), g2d.device)
    out = accumulate_into if accumulate_into is not None else torch.empty(C, dtype=torch.float32, device=g2d.device)
    L.check(lib.dreg_colsum(L.ptr(g2d), L.ptr(out), L.ptr(ws), M, C, int(accumulate_into is not None), L.dt_of(g2d), L.stream()), "dreg_colsum")
    return None if accumulate_into is not None else out


def _grad_sink(p):
    """A preallocated, contiguous fp32 .grad of a leaf parameter (FlatAdamW's buffer views) that kernels may accumulate
    into directly, skipping autograd's separate add; None otherwise (slices of parameters, no .grad yet, ...)."""
    if DIRECT_GRAD_ACCUMULATE and p.is_leaf and p._base is None and p.grad is not None and p.grad.is_contiguous() \
            and p.grad.dtype == torch.float32:
        return p.grad
    return None


DIRECT_GRAD_ACCUMULATE = True
# torch.cuda.Stream for the weight / bias gradient launches of Conv3dFn.backward while a caller that joins it afterwards is
# running the backward pass (train_step.TrainStep); None = everything on the current stream
PARAM_GRAD_STREAM = None
# optim.GradSync of a data-parallel step while its backward pass runs (the native trunk executor reports finished gradient ranges to it)
GRAD_SYNC = None


def downsample_sum(g, coarse_shape):
    lib = L.load()
    B, Df, Hf, Wf, C = g.shape
    Dc, Hc, Wc = coarse_shape
    out = torch.empty(B, Dc, Hc, Wc, C, dtype=g.dtype, device=g.device)
    L.check(lib.dreg_downsample_sum(L.ptr(g), L.ptr(out), B, Df, Hf, Wf, Dc, Hc, Wc, C, L.dt_of(g), L.stream()),
            "dreg_downsample_sum")
    return out


USE_TR = True  # LDS transpose reads in the bf16 weight-gradient kernel (False = 16-bit gather checker path)


def relu_bwd_cast(y, g, out_dtype):
    """(y > 0 ? g : 0) cast to out_dtype; y = None means a plain cast."""
    lib = L.load()
    g = g.contiguous()
    out = torch.empty(g.shape, dtype=out_dtype, device=g.device)
    if y is None:
        if g.dtype == out_dtype:
            return g
        assert g.dtype == torch.float32
        L.check(lib.dreg_cast_from_f32(L.ptr(g), L.ptr(out), g.numel(), L.dt_of(out), L.stream()), "dreg_cast_from_f32")
        return out
    L.check(lib.dreg_relu_bwd(L.ptr(y), L.ptr(g), L.ptr(out), g.numel(), L.dt_of(y), L.dt_of(g), L.dt_of(out), L.stream()),
            "dreg_relu_bwd")
    return out


class Conv3dFn(torch.autograd.Function):
    """y = [relu](conv3d(x, w) [+ bias] [+ addend])  — NDHWC, MFMA implicit GEMM forward, data gradient (same kernel,
    transposed coordinate map) and split-K weight gradient.  addend: nearest-x2-upsampled coarser map (FPN) or, with
    add_same, a same-shape residual.  out_f32: bf16 operands, fp32 result (fp32 residual stream of the transformer)."""

    @staticmethod
    def forward(ctx, x, w, bias, addend, stride: int, pad: int, relu: bool = False, out_f32: bool = False, add_same: bool = False):
        dt = L.dt_of(x)
        cin_pad = x.shape[4]
        cout = w.shape[0]
        ksz = w.shape[2] if w.dim() == 5 else 1
        out_shape = tuple(_out_dim(x.shape[i + 1], ksz, stride, pad) for i in range(3))
        b32 = bias.detach().float().contiguous() if bias is not None else None
        if not relu and w.dim() == 5 and w.shape[1] == cin_pad and halo_applies(x.shape, cin_pad, cout, ksz, stride, pad, dt):
            y = conv_halo(x, w, b32, addend, False, add_same=add_same, out_f32=out_f32)
        else:
            wpk = packed_weight(w, cin_pad, False, dt)
            y = conv_igemm(x, wpk, b32, addend, out_shape, cin_pad, cout, ksz, stride, pad, False, relu=relu, out_f32=out_f32,
                           flop_cin=w.shape[1], add_same=add_same)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.bias_ref = bias if (bias is not None and bias.is_leaf) else None
        ctx.cfg = (stride, pad, ksz, cin_pad, bias is not None, None if addend is None else tuple(addend.shape[1:4]), add_same)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, ksz, cin_pad, has_bias, add_shape, add_same = ctx.cfg
        dt = L.dt_of(x)
        gy = gy.contiguous()
        cout = w.shape[0]
        gx = gw = gb = ga = None
        if add_shape is not None and ctx.needs_input_grad[3] and y is None:
            ga = gy if add_same else downsample_sum(gy, add_shape)
        g = relu_bwd_cast(y, gy, x.dtype)  # operand dtype for the MFMA passes, ReLU mask applied
        if add_shape is not None and ctx.needs_input_grad[3] and y is not None:
            ga = g if add_same else downsample_sum(g, add_shape)
        if ctx.needs_input_grad[0]:
            if stride == 2 and dt == L.DT_BF16 and ((ksz == 3 and pad == 1) or (ksz == 1 and pad == 0)) and cout % 64 == 0 \
                    and w.shape[1] % 64 == 0 and L.load().dreg_conv_get_glds():
                gx = conv_dgrad_s2(g, packed_weight(w, cin_pad, 2, dt), tuple(x.shape), cout, ksz, pad)
            elif y is None and w.dim() == 5 and halo_applies(g.shape, cout, w.shape[1], ksz, stride, pad, dt):
                gx = conv_halo(g, w, None, None, True)          # flipped-tap pack: dIn = conv(dOut, W^T)
            else:
                wpk = packed_weight(w, cin_pad, True, dt)
                gx = conv_igemm(g, wpk, None, None, tuple(x.shape[1:4]), cout, w.shape[1], ksz, stride, pad, True)
        wsink = _grad_sink(w) if ctx.needs_input_grad[1] else None
        bsink = _grad_sink(ctx.bias_ref) if (has_bias and ctx.needs_input_grad[2] and ctx.bias_ref is not None) else None
        want_b = has_bias and ctx.needs_input_grad[2]
        if PARAM_GRAD_STREAM is not None and wsink is not None and (not want_b or bsink is not None) \
                and (PROFILER is None or not PROFILER.enabled):
            # both parameter gradients accumulate in place: run them on the side stream next to the data-gradient chain
            # (train_step.TrainStep joins the stream before the optimizer reads the gradients)
            ev = torch.cuda.Event()
            ev.record()
            PARAM_GRAD_STREAM.wait_event(ev)
            with torch.cuda.stream(PARAM_GRAD_STREAM):
                conv_wgrad(g, x, tuple(w.shape), cin_pad, ksz, stride, pad, USE_TR, accumulate_into=wsink, defer=True)
                if want_b:
                    colsum(g.view(-1, cout), accumulate_into=bsink)
            g.record_stream(PARAM_GRAD_STREAM)
            x.record_stream(PARAM_GRAD_STREAM)
            return gx, None, None, ga, None, None, None, None, None
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(g, x, tuple(w.shape), cin_pad, ksz, stride, pad, USE_TR, accumulate_into=wsink)
        if want_b:
            gb = colsum(g.view(-1, cout), accumulate_into=bsink)
        return gx, gw, gb, ga, None, None, None, None, None


def conv3d(x, w, bias=None, addend=None, stride=1, pad=0, out_rows=None):
    """out_rows (optional, int32 ascending flat output-voxel indices: dreg_conv_rows): the output rows of a bias-free convolution over a
    sparse volume whose receptive field holds an occupied voxel — only those are computed (the others are exactly zero) and only those
    enter the weight gradient (the network's stem, conerf/model/resnet3d.py conv1)."""
    if out_rows is not None and bias is None and addend is None and L.dt_of(x) == L.DT_BF16 and not x.requires_grad:
        return StridedRowsConv3dFn.apply(x, w, stride, pad, out_rows)
    return Conv3dFn.apply(x, w, bias, addend, stride, pad)


class StridedRowsConv3dFn(torch.autograd.Function):
    """y = conv3d(x, w, stride, pad) for a bias-free layer whose input is zero outside a known voxel set: computed on `rows` (the output
    voxels that can be non-zero), zero elsewhere — the dense result bit for bit; the weight gradient is reduced over `rows` only (the
    other rows multiply all-zero patches).  No input gradient (the layer reads the network input)."""

    @staticmethod
    def forward(ctx, x, w, stride: int, pad: int, rows):
        lib = L.load()
        B, Di, Hi, Wi, cin = x.shape
        cout, ksz = w.shape[0], w.shape[2]
        Do, Ho, Wo = ((d + 2 * pad - ksz) // stride + 1 for d in (Di, Hi, Wi))
        wpk = packed_weight(w, cin, False, L.DT_BF16)
        y = torch.zeros(B, Do, Ho, Wo, cout, dtype=x.dtype, device=x.device)
        ev = None
        if PROFILER is not None:
            ev = PROFILER.record(igemm_kernel_name(lib, B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, False, rows.shape[0], 0, False, L.DT_BF16, False),
                                 f"fwd-rows B{B} {Di}x{Hi}x{Wi}x{cin}->{Do}x{Ho}x{Wo}x{cout} k{ksz}s{stride} rows{rows.shape[0]}",
                                 2.0 * rows.shape[0] * cout * (ksz ** 3) * w.shape[1])
            if ev is not None:
                ev[0].record()
        L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wpk), L.ptr(y), None, None, L.ptr(rows), rows.shape[0], B, Di, Hi, Wi, cin, Do, Ho, Wo, cout,
                                           ksz, stride, pad, 0, 0, 0, 0, 0, 0, 0, L.stream()), "dreg_conv3d_igemm_rows")
        if ev is not None:
            ev[1].record()
        ctx.save_for_backward(x, w, rows)
        ctx.cfg = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, rows = ctx.saved_tensors
        stride, pad = ctx.cfg
        lib = L.load()
        gy = gy.contiguous()
        gw = None
        if ctx.needs_input_grad[1]:
            B, Di, Hi, Wi, cin = x.shape
            Do, Ho, Wo, cout = gy.shape[1:]
            ksz = w.shape[2]
            nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, Do, Ho, Wo, cin, cout, ksz, 0)
            ws = _ws(nbytes, x.device)
            sink = _grad_sink(w)
            gw_t = sink if sink is not None else torch.empty(tuple(w.shape), dtype=torch.float32, device=x.device)
            L.check(lib.dreg_conv3d_wgrad_rows(L.ptr(gy), L.ptr(x), L.ptr(gw_t), L.ptr(ws), nbytes, L.ptr(rows), rows.shape[0],
                                               B, Di, Hi, Wi, cin, w.shape[1], Do, Ho, Wo, cout, ksz, stride, pad, int(sink is not None),
                                               L.stream()), "dreg_conv3d_wgrad_rows")
            gw = None if sink is not None else gw_t
        return None, gw, None, None, None


def igemm_kernel_name(lib, B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, transposed, nrows, has_ws, has_addend, dt, out_f32):
    """The template instantiation a convolution launch of this shape runs, spelled like bench.py normalises rocprofv3's kernel names
    (one rule set in the library: dreg_conv3d_igemm_variant), so a profiler label maps 1:1 to a row of the kernel trace.
    Split-K launches bracket two kernels (fp32 partial tiles + the finishing pass): '...+splitk_reduce'."""
    v = lib.dreg_conv3d_igemm_variant(B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, stride, pad, int(bool(transposed)), int(nrows), int(bool(has_ws)),
                                      int(bool(has_addend)), 0 if dt == L.DT_BF16 else 1)
    if v < 0:
        return "conv(unsupported)"
    kind, bm, bn, ap, sk = v // 100000000, (v // 100000) % 1000, (v // 100) % 1000, (v // 10) % 10, v % 10
    to = "f32" if (out_f32 or sk) else "bf16"
    if kind == 1:
        return f"conv_igemm_kernel<{'bf16' if dt == L.DT_BF16 else 'f32'},{to},{bn}>"
    return f"conv_igemm_glds_kernel<{to},{bm},{bn},0,{ap}>" + ("+splitk_reduce" if sk else "")


# --------------------------------------------------------------------------- active-set convolution (row lists)
def _igemm_rows(x, wpk, bias, addend, out, rows, cin, cout, ksz, pad, transposed, flop_cin=None):
    lib = L.load()
    B, Di, Hi, Wi = x.shape[:4]
    Do, Ho, Wo = out.shape[1:4]
    Da = Ha = Wa = 0
    if addend is not None:
        Da, Ha, Wa = addend.shape[1:4]
    ev = None
    if PROFILER is not None:
        label = f"{'dgrad' if transposed else 'fwd'}-rows B{B} {Di}x{Hi}x{Wi}x{cin}->{Do}x{Ho}x{Wo}x{cout} k{ksz} rows{rows.shape[0]}"
        ev = PROFILER.record(igemm_kernel_name(lib, B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, 1, pad, transposed, rows.shape[0], 0, addend is not None, L.DT_BF16, False), label,
                             2.0 * rows.shape[0] * cout * (ksz ** 3) * (flop_cin or cin))
        if ev is not None:
            ev[0].record()
    L.check(lib.dreg_conv3d_igemm_rows(L.ptr(x), L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(addend), L.ptr(rows), rows.shape[0],
                                       B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, ksz, 1, pad, int(transposed), 0, Da, Ha, Wa, 0, 0,
                                       L.stream()), "dreg_conv3d_igemm_rows")
    if ev is not None:
        ev[1].record()
    return out


class SparseConv3dFn(torch.autograd.Function):
    """Stride-1 conv3d evaluated on an active set: the forward computes only the output voxels `out_rows`; the backward's
    input gradient is non-zero only on `in_rows` (= out_rows dilated by the kernel footprint) and the weight gradient is
    reduced over `out_rows`.  Contract: the incoming gradient is zero outside `out_rows` (true for the FPN head: it comes
    from the trilinear-gather backward, or from the data gradient of the next active-set convolution)."""

    @staticmethod
    def forward(ctx, x, w, bias, addend, pad: int, out_rows, in_rows):
        dt = L.dt_of(x)
        assert dt == L.DT_BF16, "active-set convolutions are built for the bf16 production mode"
        cin, cout, ksz = x.shape[4], w.shape[0], w.shape[2]
        wpk = packed_weight(w, cin, False, dt)
        b32 = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout, dtype=x.dtype, device=x.device)
        _igemm_rows(x, wpk, b32, addend, y, out_rows, cin, cout, ksz, pad, False, flop_cin=w.shape[1])
        ctx.save_for_backward(x, w, out_rows, in_rows)
        ctx.bias_ref = bias if (bias is not None and bias.is_leaf) else None
        ctx.cfg = (pad, ksz, bias is not None, None if addend is None else tuple(addend.shape[1:4]))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, out_rows, in_rows = ctx.saved_tensors
        pad, ksz, has_bias, add_shape = ctx.cfg
        lib = L.load()
        gy = gy.contiguous()
        cin, cout = x.shape[4], w.shape[0]
        gx = gw = gb = ga = None
        if ctx.needs_input_grad[0]:
            wpk = packed_weight(w, cin, True, L.DT_BF16)
            gx = torch.zeros_like(x)
            _igemm_rows(gy, wpk, None, None, gx, in_rows, cout, cin, ksz, pad, True)
        if ctx.needs_input_grad[1]:
            B, Di, Hi, Wi = x.shape[:4]
            nbytes = lib.dreg_conv3d_wgrad_workspace_bytes(B, Di, Hi, Wi, cin, cout, ksz, 0)
            ws = _ws(nbytes, x.device)
            sink = _grad_sink(w)
            gw_t = sink if sink is not None else torch.empty(tuple(w.shape), dtype=torch.float32, device=x.device)
            ev = None
            if PROFILER is not None:
                ev = PROFILER.record("conv_wgrad_kernel<bf16>+reduce", f"wgrad-rows B{B} {Di}x{Hi}x{Wi}x{cin} g{cout} k{ksz} rows{out_rows.shape[0]}",
                                     2.0 * out_rows.shape[0] * cout * (ksz ** 3) * w.shape[1])
                if ev is not None:
                    ev[0].record()
            L.check(lib.dreg_conv3d_wgrad_rows(L.ptr(gy), L.ptr(x), L.ptr(gw_t), L.ptr(ws), nbytes, L.ptr(out_rows), out_rows.shape[0],
                                               B, Di, Hi, Wi, cin, w.shape[1], Di, Hi, Wi, cout, ksz, 1, pad, int(sink is not None),
                                               L.stream()), "dreg_conv3d_wgrad_rows")
            if ev is not None:
                ev[1].record()
            gw = None if sink is not None else gw_t
        if has_bias and ctx.needs_input_grad[2]:
            sink = _grad_sink(ctx.bias_ref) if ctx.bias_ref is not None else None
            gb_t = sink if sink is not None else torch.empty(cout, dtype=torch.float32, device=x.device)
            wsb = _ws(lib.dreg_colsum_workspace_bytes(out_rows.shape[0], cout), x.device)
            L.check(lib.dreg_colsum_rows(L.ptr(gy), L.ptr(out_rows), out_rows.shape[0], L.ptr(gb_t), L.ptr(wsb), cout, int(sink is not None),
                                         L.dt_of(gy), L.stream()), "dreg_colsum_rows")
            gb = None if sink is not None else gb_t
        if add_shape is not None and ctx.needs_input_grad[3]:
            ga = downsample_sum(gy, add_shape)
        return gx, gw, gb, ga, None, None, None


def conv3d_rows(x, w, bias, addend, pad, out_rows, in_rows):
    return SparseConv3dFn.apply(x, w, bias, addend, pad, out_rows, in_rows)


class RowSets(tuple):
    """The tuple active_sets returns, plus `tiles`: {row-list position -> brick.BrickTiles} when tile tables were requested (the
    active-set 3^3 convolutions then run on csrc/conv_brick.hip: staged-neighbourhood reuse instead of a 27-tap gather per row), and
    `stem`: the output rows of the stem convolution whose receptive field holds an occupied voxel (dreg_conv_rows), or None."""
    tiles = None
    stem = None


def active_sets(idx_list, fine_res, coarse_dims, device, pt_batch=None, idx_cat=None, density_cap: float = 0.3,
                level2: bool = True, level2_cap: float = 0.5, brick_tiles=False, stem=None):
    """Row lists for the FPN head: S1 = trilinear-gather corner voxels of the occupied fine voxels (where P1 is consumed),
    S2 = S1 dilated by 3^3 (where the lateral sum is consumed), S3 = S2 dilated (where dc1 of the head is non-zero).
    idx_list: per grid int64 flat fine indices ((x*Yr + y)*Zr + z).  Returns three ascending int32 tensors of flat indices
    into [B, d, h, w] plus map1 (int32 [B*d*h*w]: rank of a voxel in S1, -1 outside) — None for dense scenes — and, when the
    next pyramid level is sparse enough as well, A / A2 (row lists on [B, ceil(d/2), ...]: parents of S2, dilated).  One C call (mark, two dilations, stable compaction) + one host read of the
    three lengths; the corner arithmetic is tri_axis() of fpn_ops.hip itself."""
    lib = L.load()
    Zr, Xr, Yr = fine_res
    d, h, w = coarse_dims
    B = len(idx_list)
    V = B * d * h * w
    if idx_cat is None:
        idx_cat = torch.cat(idx_list).contiguous()
    if pt_batch is None:
        pt_batch = torch.cat([torch.full((f.shape[0],), b, dtype=torch.int32, device=device) for b, f in enumerate(idx_list)])
    rows = torch.empty(3, V, dtype=torch.int32, device=device)
    counts = torch.zeros(4, dtype=torch.int32, device=device)         # S1, S2, S3, stem rows
    srows = None
    if stem is not None:     # (ksz, stride, pad) of the convolution that reads the fine volume and writes [B, d, h, w]: its non-zero output rows
        nbs = lib.dreg_conv_rows_workspace_bytes(B, d, h, w)
        wss = torch.empty(nbs, dtype=torch.uint8, device=device)
        srows = torch.empty(V, dtype=torch.int32, device=device)
        L.check(lib.dreg_conv_rows(L.ptr(idx_cat), L.ptr(pt_batch), idx_cat.shape[0], B, Zr, Xr, Yr, d, h, w, int(stem[0]), int(stem[1]), int(stem[2]),
                                   L.ptr(srows), counts.data_ptr() + 12, L.ptr(wss), nbs, L.stream()), "dreg_conv_rows")
    nbytes = lib.dreg_active_sets_workspace_bytes(B, d, h, w)
    ws = _ws(nbytes, device)
    map1 = torch.empty(V, dtype=torch.int32, device=device)
    L.check(lib.dreg_active_sets(L.ptr(idx_cat), L.ptr(pt_batch), idx_cat.shape[0], B, Zr, Xr, Yr, d, h, w,
                                 L.ptr(rows), L.ptr(counts), L.ptr(map1), L.ptr(ws), nbytes, L.stream()), "dreg_active_sets")
    # second pyramid level (P2 at half the resolution is only consumed at the parents of S2): A = parents(S2), A2 = dilate(A)
    d2, h2, w2 = (d + 1) // 2, (h + 1) // 2, (w + 1) // 2
    V2 = B * d2 * h2 * w2
    rows2 = torch.empty(2, V2, dtype=torch.int32, device=device)
    counts2 = torch.empty(2, dtype=torch.int32, device=device)
    if level2:
        nb2 = lib.dreg_active_sets_level2_workspace_bytes(B, d2, h2, w2)
        ws2 = _ws(nb2, device)
        L.check(lib.dreg_active_sets_level2(ws.data_ptr() + V, B, d, h, w, d2, h2, w2, L.ptr(rows2), L.ptr(counts2), L.ptr(ws2), nb2, L.stream()),
                "dreg_active_sets_level2")
        n1, n2, n3, ns, m1, m2 = torch.cat([counts, counts2]).tolist()
    else:
        (n1, n2, n3, ns), m1, m2 = counts.tolist(), 0, V2
    if n3 > density_cap * V:   # dense scenes: the row lists stop paying (and the wgrad slice cap applies)
        return None
    out = (rows[0, :n1], rows[1, :n2], rows[2, :n3], map1)
    use2 = level2 and m2 <= level2_cap * V2
    if use2:
        out = out + (rows2[0, :m1], rows2[1, :m2])
    out = RowSets(out)
    if srows is not None:
        out.stem = srows[:ns]
    if brick_tiles and lib.dreg_brick_supported(B, d, h, w, 16, 256):
        # tile tables of every row set from its flag volume (still in the builders' workspaces), one more host read for the tile counts
        from . import brick
        sets = [(0, ws[0:V], n1, (B, d, h, w)), (1, ws[V:2 * V], n2, (B, d, h, w)), (2, ws[2 * V:3 * V], n3, (B, d, h, w))]
        if use2:
            sets += [(4, ws2[0:V2], m1, (B, d2, h2, w2)), (5, ws2[V2:2 * V2], m2, (B, d2, h2, w2))]
        want = None if brick_tiles is True else set(brick_tiles)      # True: every set; else the row-list positions asked for
        bts = {k: brick.build_async(f.view(*dims), n) for k, f, n, dims in sets if n > 0 and (want is None or k in want)}
        if bts:
            metas = torch.stack([bt.meta for bt in bts.values()]).tolist()
            out.tiles = {k: bt for (k, bt), mh in zip(bts.items(), metas) if not brick.finish(bt, mh).overflow}
    return out


def linear(x2d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False, residual=None,
           out_f32: bool = False) -> torch.Tensor:
    """[N, Cin] x [Cout, Cin]^T (+bias) (+residual) (relu) through the 1x1x1 path of the same kernels."""
    n, cin = x2d.shape
    # rows as the batch dimension of 1x1x1 "volumes": the kernels' row decode (magic division by the volume extents) then has no
    # size limit — as one x-row of n voxels it needs n^2 < 2^32, which 6 stacked layers of ~11k points already exceed
    res = residual.view(n, 1, 1, 1, w.shape[0]) if residual is not None else None
    y = Conv3dFn.apply(x2d.view(n, 1, 1, 1, cin), w, bias, res, 1, 0, relu, out_f32, residual is not None)
    return y.view(n, w.shape[0])


# --------------------------------------------------------------------------- BatchNorm (+res) (+ReLU), per-grid statistics
class BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, res, relu: bool, train: bool, eps: float, momentum: float):
        lib = L.load()
        dt = L.dt_of(x)
        B, C = x.shape[0], x.shape[4]
        V = x.shape[1] * x.shape[2] * x.shape[3]
        nch = lib.dreg_bn_num_chunks(V)
        ws = torch.empty(B * nch * C * 2, dtype=torch.float32, device=x.device)
        ss = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        mr = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        L.check(lib.dreg_bn3d_fwd(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(gamma.detach()), L.ptr(beta.detach()),
                                  L.ptr(running_mean), L.ptr(running_var), L.ptr(ss), L.ptr(mr), L.ptr(ws),
                                  B, V, C, eps, momentum, int(train), int(relu), dt, L.stream()), "dreg_bn3d_fwd")
        ctx.save_for_backward(x, y, ss, mr)
        ctx.cfg = (relu, train, res is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, ss, mr = ctx.saved_tensors
        relu, train, has_res = ctx.cfg
        if not train:
            raise L.DregError("BatchNorm backward is implemented for training-mode statistics only")
        lib = L.load()
        dt = L.dt_of(x)
        gy = gy.contiguous()
        B, C = x.shape[0], x.shape[4]
        V = x.shape[1] * x.shape[2] * x.shape[3]
        nch = lib.dreg_bn_num_chunks(V)
        ws = torch.empty(B * nch * C * 2, dtype=torch.float32, device=x.device)
        coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        L.check(lib.dreg_bn3d_bwd(L.ptr(x), L.ptr(gy), L.ptr(y) if has_res else None, L.ptr(ss), L.ptr(mr), L.ptr(dx), L.ptr(dres),
                                  L.ptr(dg), L.ptr(db), L.ptr(coef), L.ptr(ws), B, V, C, int(relu), 0, dt, L.stream()),
                "dreg_bn3d_bwd")
        return dx, dg, db, None, None, dres, None, None, None, None


def batchnorm(x, gamma, beta, running_mean, running_var, res=None, relu=True, train=True, eps=1e-5, momentum=0.1):
    return BatchNormFn.apply(x, gamma, beta, running_mean, running_var, res, relu, train, eps, momentum)


# --------------------------------------------------------------------------- MaxPool3d(3, 2, 1)
class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = L.load()
        B, Di, Hi, Wi, C = x.shape
        Do, Ho, Wo = (_out_dim(Di, 3, 2, 1), _out_dim(Hi, 3, 2, 1), _out_dim(Wi, 3, 2, 1))
        y = torch.empty(B, Do, Ho, Wo, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(B, Do, Ho, Wo, C, dtype=torch.uint8, device=x.device)
        L.check(lib.dreg_maxpool3d_fwd(L.ptr(x), L.ptr(y), L.ptr(arg), B, Di, Hi, Wi, Do, Ho, Wo, C, L.dt_of(x), L.stream()),
                "dreg_maxpool3d_fwd")
        ctx.save_for_backward(arg)
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        (arg,) = ctx.saved_tensors
        lib = L.load()
        B, Di, Hi, Wi, C = ctx.in_shape
        Do, Ho, Wo = arg.shape[1:4]
        gy = gy.contiguous()
        dx = torch.empty(ctx.in_shape, dtype=gy.dtype, device=gy.device)
        L.check(lib.dreg_maxpool3d_bwd(L.ptr(gy), L.ptr(arg), L.ptr(dx), B, Di, Hi, Wi, Do, Ho, Wo, C, L.dt_of(gy), L.stream()),
                "dreg_maxpool3d_bwd")
        return dx


def maxpool3d(x):
    return MaxPoolFn.apply(x)


# --------------------------------------------------------------------------- fused trilinear upsample + gather
PERSISTENT_GRAD_BUFFERS = False     # train_step.TrainStep sets it around backward
_DP1_CACHE = {}                     # (shape, dtype, device) -> [buffer, rows the last writer dirtied, in use by an unconsumed gradient]


def release_grad_buffers(free: bool = False):
    """TrainStep calls this after its backward: the cached dense gradient buffers may be handed out again (their consumer, the trunk's
    backward, is enqueued behind the writer on the same stream).  free=True drops them (up to 2 x ~1 GB at 8 x 64^3 x 256)."""
    for ent in _DP1_CACHE.values():
        ent[2] = False
    if free:
        _DP1_CACHE.clear()


class TrilinearGatherFn(torch.autograd.Function):
    """feats[n] = trilinear(align_corners) sample of p1[batch[n]] at fine voxel idx[n]; output fp32 [N, C]."""

    @staticmethod
    def forward(ctx, p1, idx, pt_batch, fine_res, rows1=None, map1=None):
        lib = L.load()
        B, d, h, w, C = p1.shape
        N = idx.shape[0]
        out = torch.empty(N, C, dtype=torch.float32, device=p1.device)
        Zr, Xr, Yr = fine_res
        L.check(lib.dreg_trilinear_gather_fwd(L.ptr(p1), L.ptr(idx), L.ptr(pt_batch), L.ptr(out), N, d, h, w, C, Zr, Xr, Yr,
                                              L.dt_of(p1), 1, L.stream()), "dreg_trilinear_gather_fwd")
        ctx.save_for_backward(idx, pt_batch, rows1, map1)
        ctx.cfg = (tuple(p1.shape), p1.dtype, fine_res)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, pt_batch, rows1, map1 = ctx.saved_tensors
        shape, dtype, (Zr, Xr, Yr) = ctx.cfg
        lib = L.load()
        B, d, h, w, C = shape
        gout = gout.contiguous().float()
        if rows1 is not None and ((C % 64 == 0 and C <= 256) or map1 is not None):  # per consumed coarse voxel (S1): gather form, else compact atomics
            g = torch.empty(shape, dtype=dtype, device=gout.device)
            if C % 64 == 0 and C <= 256 and PERSISTENT_GRAD_BUFFERS:
                # one dense buffer per shape, kept across steps: zero outside the rows of the step that wrote it, so only those
                # rows are cleared (a 1 GB memset per step at 8 x 64^3 x 256 otherwise).  train_step turns this on around its
                # backward: the consumer (the trunk's backward) has read the buffer before the next step reuses it.
                # A second gather node of the same shape inside ONE backward (per-pair forwards summed into one loss) must not get the
                # buffer whose gradient has not been consumed yet: it takes a fresh zero-filled one.
                key = (shape, dtype, gout.device)
                ent = _DP1_CACHE.get(key)
                if ent is not None and ent[2]:
                    ent = [torch.zeros(shape, dtype=dtype, device=gout.device), None, True]
                elif ent is None:
                    while len(_DP1_CACHE) >= 2:            # one buffer per (shape, dtype, device): 1 GB at 8 x 64^3 x 256 — keep two
                        _DP1_CACHE.pop(next(iter(_DP1_CACHE)))
                    ent = _DP1_CACHE[key] = [torch.zeros(shape, dtype=dtype, device=gout.device), None, False]
                ent[2] = True
                g, dirty = ent[0], ent[1]
                if dirty is not None:
                    L.check(lib.dreg_zero_rows(L.ptr(g), L.ptr(dirty), dirty.shape[0], C, L.dt_of(g), L.stream()), "dreg_zero_rows")
                fmap = torch.empty(B * Zr * Xr * Yr, dtype=torch.int32, device=gout.device)
                L.check(lib.dreg_trilinear_gather_bwd_gather_rows_only(L.ptr(gout), L.ptr(idx), L.ptr(pt_batch), L.ptr(rows1), rows1.shape[0], L.ptr(fmap),
                                                                       L.ptr(g), idx.shape[0], B, d, h, w, C, Zr, Xr, Yr, L.dt_of(g), L.stream()),
                        "dreg_trilinear_gather_bwd_gather_rows_only")
                ent[1] = rows1
                return g, None, None, None, None, None
            if C % 64 == 0 and C <= 256:   # atomic-free gather per S1 voxel: deterministic
                fmap = torch.empty(B * Zr * Xr * Yr, dtype=torch.int32, device=gout.device)
                L.check(lib.dreg_trilinear_gather_bwd_gather(L.ptr(gout), L.ptr(idx), L.ptr(pt_batch), L.ptr(rows1), rows1.shape[0], L.ptr(fmap),
                                                             L.ptr(g), idx.shape[0], B, d, h, w, C, Zr, Xr, Yr, L.dt_of(g), L.stream()),
                        "dreg_trilinear_gather_bwd_gather")
                return g, None, None, None, None, None
            comp = torch.empty(max(rows1.shape[0], 1), C, dtype=torch.float32, device=gout.device)
            L.check(lib.dreg_trilinear_gather_bwd_rows(L.ptr(gout), L.ptr(idx), L.ptr(pt_batch), L.ptr(rows1), rows1.shape[0], L.ptr(map1),
                                                       L.ptr(comp), L.ptr(g), idx.shape[0], B, d, h, w, C, Zr, Xr, Yr, L.dt_of(g), L.stream()),
                    "dreg_trilinear_gather_bwd_rows")
            return g, None, None, None, None, None
        g32 = torch.zeros(shape, dtype=torch.float32, device=gout.device)
        L.check(lib.dreg_trilinear_gather_bwd(L.ptr(gout), L.ptr(idx), L.ptr(pt_batch), L.ptr(g32), idx.shape[0], d, h, w, C,
                                              Zr, Xr, Yr, L.stream()), "dreg_trilinear_gather_bwd")
        if dtype == torch.float32:
            return g32, None, None, None, None, None
        g = torch.empty(shape, dtype=dtype, device=gout.device)
        L.check(lib.dreg_cast_from_f32(L.ptr(g32), L.ptr(g), g32.numel(), L.dt_of(g), L.stream()), "dreg_cast_from_f32")
        return g, None, None, None, None, None


def trilinear_gather(p1, idx, pt_batch, fine_res, rows1=None, map1=None):
    """rows1 / map1 (from active_sets): the backward then touches the S1 rows only."""
    return TrilinearGatherFn.apply(p1, idx, pt_batch, fine_res, rows1, map1)

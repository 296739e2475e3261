"""Surface-field visibility labels from a NeRF block (row N1 of SURVEY.md §8f): the overlap ground truth of the training
step (train_nerf_regtr.py:186-199 -> conerf/loss/confidence_loss.py:56-160) and the surface mask of grid extraction
(conerf/register/sample_grid.py:244-318), both through the fused ray-march kernel of visibility.hip.

The reference re-loads the block checkpoint from disk twice per call (confidence_loss.py:34-35,50); here loaded blocks are
cached per path."""
import collections
import ctypes
import threading
from typing import Dict, List

import os

import torch

from . import lib as L
from . import ngp

# Loaded blocks, least recently used first; bounded by device bytes (a block = 12.6 M hash-grid parameters in fp32 + the fp16
# inference copy + the occupancy grid, ~80 MB: an Objaverse epoch touches 3,284 of them, the reference reloads from disk every call).
_block_cache: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_block_cache_bytes = 0
_block_cache_lock = threading.Lock()     # the prefetching loader's thread fills the cache while the training thread reads it
# Budget: DREG_BLOCK_CACHE_MB, default 40 % of the device's memory.  A cached block is its fp16 inference copy + occupancy grid (27 MB):
# the 3,284 blocks of an Objaverse epoch are 90 GB — they stay resident on a 288 GB MI355X next to the training step's ~20-30 GB, so from
# the second epoch on no block is read from disk again (the reference re-reads each block's checkpoint twice per call).
_BLOCK_CACHE_ENV = os.environ.get("DREG_BLOCK_CACHE_MB")
BLOCK_CACHE_MAX_BYTES = int(_BLOCK_CACHE_ENV) << 20 if _BLOCK_CACHE_ENV else None


def _cache_budget(device) -> int:
    if BLOCK_CACHE_MAX_BYTES is not None:
        return BLOCK_CACHE_MAX_BYTES
    if torch.device(device).type == "cuda":
        # what is free NOW plus what the cache already holds, never more than 40 % of the device: several processes on one GPU, or a
        # smaller-HBM part, shrink the cache (evicting) instead of running the training step out of memory
        free, total = torch.cuda.mem_get_info(device)
        return int(min(0.4 * total, 0.5 * (free + _block_cache_bytes)))
    return 2048 << 20
COARSE = True             # the persistent kernel walks empty space through a coarse occupancy grid (one bit per 4^3 cells) held in LDS
PERSISTENT = True         # surface_visibility through the persistent ray-queue kernel (False: one lock-step launch of 64 rays per wave)


def _block_bytes(field, binary) -> int:
    """What a cached block holds on the device: the fp16 inference copies (the fp32 parameters are released: freeze_for_inference),
    the field's buffers and the occupancy grid."""
    b16, c16 = field._prepared()
    return (b16.numel() + c16.numel()) * 2 + sum(p.numel() * p.element_size() for p in field.parameters()) + \
        sum(b.numel() * b.element_size() for b in field.buffers()) + binary.numel() * binary.element_size()


def clear_block_cache():
    global _block_cache_bytes
    with _block_cache_lock:
        _block_cache.clear()
        _block_cache_bytes = 0


_load_tls = threading.local()


def _load_staging(n_f32: int, n_u8: int):
    """This thread's pinned staging buffers for load_block's uploads (allocated once per thread, grown on demand) and the event behind their last upload."""
    st = getattr(_load_tls, "st", None)
    if st is None or st["f32"].numel() < n_f32 or st["u8"].numel() < n_u8:
        if st is not None:
            st["ev"].synchronize()
        st = _load_tls.st = {"f32": torch.empty(max(n_f32, 1), dtype=torch.float32).pin_memory(), "u8": torch.empty(max(n_u8, 1), dtype=torch.uint8).pin_memory(),
                             "ev": torch.cuda.Event()}
    return st


def load_block(path: str, device, cache: bool = True):
    """(NGPradianceField, occupancy binary [r,r,r] bool on device, meta dict) of a reference NeRF block checkpoint
    (keys: train_ngp_nerf.py:187-209), through a byte-bounded LRU cache (cache=False: a block that is used once — grid extraction,
    eval_pipeline — is neither looked up nor kept).  The device work (uploads, fp16 copy, coarse occupancy bits) is enqueued on the CALLING
    thread's current stream: loader threads run this under their own stream and hand an event to the consumer."""
    global _block_cache_bytes
    key = (path, str(device))
    with _block_cache_lock:
        hit = _block_cache.get(key) if cache else None
        if hit is not None:
            _block_cache.move_to_end(key)
            return hit[:3]
    # The reference reads the file twice (conerf/loss/confidence_loss.py:25-50: meta data first, then the modules built from it).  Here it
    # is opened ONCE and memory-mapped: only the tensors that are used — the field's parameters and the occupancy grid — are paged in,
    # not the optimizer / scheduler state a training checkpoint also carries (train_ngp_nerf.py:187-209: 60 of its ~160 MB are needed).
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    ngp.install_pickle_shims()
    try:
        snap = torch.load(path, map_location="cpu", weights_only=False, mmap=True)
    except (RuntimeError, ValueError):                           # a checkpoint in the legacy (non-zip) format cannot be mapped
        snap = torch.load(path, map_location="cpu", weights_only=False)
    meta = {k: snap[k] for k in ("aabb", "unbounded", "grid_resolution", "contraction_type", "render_step_size", "alpha_thre",
                                 "cone_angle", "camera_poses")}
    dev = torch.device(device)
    sd, og = snap["model"], snap["occupancy_grid"]
    extra = [k for k in sd if k not in ("aabb", "mlp_base.params", "color_mlp.params") and torch.is_tensor(sd[k]) and sd[k].numel() > 0]
    if extra:
        raise RuntimeError(f"unexpected non-empty keys in NeRF state_dict: {extra}")
    res3 = meta["grid_resolution"]
    res3 = [int(res3)] * 3 if isinstance(res3, int) else [int(v) for v in res3]
    if getattr(meta["contraction_type"], "name", "AABB") != "AABB":
        raise NotImplementedError("only ContractionType.AABB is used by the registration data (config.py:63, Objaverse)")
    binary_h = og["_binary"]
    if tuple(binary_h.shape) != tuple(res3) or og["_roi_aabb"].shape != (6,):
        raise RuntimeError(f"occupancy_grid state does not fit grid_resolution {res3}: _binary {tuple(binary_h.shape)}")
    field_aabb_host = [float(v) for v in sd["aabb"].tolist()]
    n_occupied = int(binary_h.sum())                            # counted on the host (2 MB): the dense query then needs no readback of it
    if dev.type == "cuda":
        # Only what is queried goes to the device, and only as fp16: the 12.6 M fp32 parameters are copied from the mapped file into this thread's
        # PINNED staging buffer (a plain memcpy), uploaded asynchronously on the calling thread's stream and converted there; the unused EMA densities
        # (`occs`, 8 MB) are never touched.  (Round 5 built an fp32 module on the device and copied the state_dict into it from pageable memory: 60-130 ms
        # per block, synchronous copies that also held up other threads' launches.)
        n_base, n_col = sd["mlp_base.params"].numel(), sd["color_mlp.params"].numel()
        nb = int(binary_h.numel())
        st = _load_staging(n_base + n_col, nb)
        st["ev"].synchronize()                                   # the previous upload from this buffer
        st["f32"][:n_base].copy_(sd["mlp_base.params"].reshape(-1))
        st["f32"][n_base:n_base + n_col].copy_(sd["color_mlp.params"].reshape(-1))
        st["u8"][:nb].copy_(binary_h.reshape(-1).view(torch.uint8) if binary_h.dtype == torch.bool else (binary_h.reshape(-1) != 0).to(torch.uint8))
        tmp32 = torch.empty(n_base + n_col, dtype=torch.float32, device=dev)
        tmp32.copy_(st["f32"][:n_base + n_col], non_blocking=True)
        binary_u8 = torch.empty(res3, dtype=torch.uint8, device=dev)
        binary_u8.view(-1).copy_(st["u8"][:nb], non_blocking=True)
        st["ev"].record(torch.cuda.current_stream(dev))
        base16 = torch.empty(n_base, dtype=torch.float16, device=dev)
        col16 = torch.empty(n_col, dtype=torch.float16, device=dev)
        lib = L.load()
        L.check(lib.dreg_f32_to_f16(tmp32.data_ptr(), L.ptr(base16), n_base, L.stream()), "dreg_f32_to_f16")
        L.check(lib.dreg_f32_to_f16(tmp32.data_ptr() + 4 * n_base, L.ptr(col16), n_col, L.stream()), "dreg_f32_to_f16")
        del tmp32
        field = ngp.NGPradianceField.from_inference_copies(field_aabb_host, bool(meta["unbounded"]), base16, col16, dev)
        binary = binary_u8.view(torch.bool)
    else:
        field = ngp.NGPradianceField(meta["aabb"], unbounded=bool(meta["unbounded"]), init=False)
        field.load_state_dict(sd)
        field = field.eval().freeze_for_inference()
        binary = binary_h.clone()
    cam_centres = torch.as_tensor(meta["camera_poses"])[..., :3, 3].float().contiguous().clone()
    meta = {k: (meta[k].clone() if torch.is_tensor(meta[k]) else meta[k]) for k in meta}       # (views into the mapped file would keep it mapped)
    del snap, sd, og, binary_h
    nbytes = _block_bytes(field, binary)
    # small meta tensors are cloned: a view into the memory-mapped checkpoint would keep one file mapping alive per cached block
    kept = {k: (meta[k].clone() if torch.is_tensor(meta[k]) else meta[k]) for k in ("aabb", "render_step_size", "cone_angle", "alpha_thre", "camera_poses")}
    # what every call needs, converted ONCE: camera centres on the device, the aabb as host floats (a .tolist() of a device tensor or an
    # H2D copy per call is a host sync per call: eight per training step, each draining the queue the host had run ahead on)
    kept["cam_centres_dev"] = cam_centres.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else cam_centres   # (a copy from pageable memory would stall the host on everything queued)
    kept["aabb_host"] = [float(v) for v in (meta["aabb"].tolist() if torch.is_tensor(meta["aabb"]) else meta["aabb"])]
    kept["n_occupied"], kept["grid_resolution"], kept["contraction_type"], kept["unbounded"] = n_occupied, meta["grid_resolution"], meta["contraction_type"], bool(meta["unbounded"])
    kept["binary_u8"] = binary.contiguous().view(torch.uint8) if binary.dtype == torch.bool else binary.to(torch.uint8).contiguous()
    kept["coarse_bits"] = coarse_occupancy_bits(kept["binary_u8"])
    if not cache:
        return field, binary, kept
    with _block_cache_lock:
        if key in _block_cache:                                  # the other thread loaded it meanwhile
            _block_cache.move_to_end(key)
            return _block_cache[key][:3]
        budget = _cache_budget(device)
        while _block_cache and _block_cache_bytes + nbytes > budget:
            _, old = _block_cache.popitem(last=False)
            _block_cache_bytes -= old[3]
        _block_cache[key] = (field, binary, kept, nbytes)
        _block_cache_bytes += nbytes
    return field, binary, kept


@torch.no_grad()
def coarse_occupancy_bits(binary_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [rx,ry,rz] occupancy -> int32 words, one bit per 4^3 block (set: some cell of the block is occupied)."""
    lib = L.load()
    rx, ry, rz = binary_u8.shape
    nbits = ((rx + 3) // 4) * ((ry + 3) // 4) * ((rz + 3) // 4)
    bits = torch.zeros((nbits + 31) // 32, dtype=torch.int32, device=binary_u8.device)
    L.check(lib.dreg_occupancy_coarse_bits(L.ptr(binary_u8), L.ptr(bits), rx, ry, rz, L.stream()), "dreg_occupancy_coarse_bits")
    return bits


@torch.no_grad()
def surface_visibility(points: torch.Tensor, cam_centres: torch.Tensor, field: ngp.NGPradianceField, binary: torch.Tensor,
                       roi_aabb, scene_aabb, render_step_size: float, cut_off: float = 0.5, early_stop_eps: float = 1e-4,
                       alpha_thre: float = 0.0, coarse_bits: torch.Tensor = None) -> torch.Tensor:
    """points [Np,3], cam_centres [Nc,3] (device) -> bool [Np]: visible from at least one camera with surface field >= cut_off."""
    lib = L.load()
    base16, _ = field._prepared()
    pts = points.reshape(-1, 3).contiguous().float()
    cams = cam_centres.reshape(-1, 3).contiguous().float().to(pts.device)
    # labels [Np] int32 followed by the ray queue's counter (8 bytes) in one zeroed buffer
    buf = torch.zeros(pts.shape[0] + 2 + (pts.shape[0] & 1), dtype=torch.int32, device=pts.device)
    label = buf[:pts.shape[0]]
    b8 = binary if binary.dtype == torch.uint8 and binary.is_contiguous() else (binary.contiguous().view(torch.uint8) if binary.dtype == torch.bool else binary.to(torch.uint8).contiguous())
    f6 = lambda v: (ctypes.c_float * 6)(*[float(t) for t in (v.tolist() if torch.is_tensor(v) else v)])
    common = (L.ptr(cams), L.ptr(pts), L.ptr(b8), L.ptr(label),
              base16.data_ptr() + 3072 * 2, base16.data_ptr(), base16.data_ptr() + 2048 * 2,
              *field._levels, f6(roi_aabb), f6(scene_aabb), f6(field._aabb_host()),
              b8.shape[0], b8.shape[1], b8.shape[2], cams.shape[0], pts.shape[0],
              float(render_step_size), float(cut_off), float(early_stop_eps), float(alpha_thre))
    if PERSISTENT:     # lanes refilled from a ray queue, labelled points not marched again (csrc/visibility.hip)
        qoff = (pts.shape[0] + (pts.shape[0] & 1)) * 4
        if coarse_bits is None and COARSE:
            coarse_bits = coarse_occupancy_bits(b8)
        OVERRUN.check()
        L.check(lib.dreg_surface_visibility_queue(*common, buf.data_ptr() + qoff, L.ptr(coarse_bits) if (coarse_bits is not None and COARSE) else None, L.stream()),
                "dreg_surface_visibility_queue")
        OVERRUN.watch(buf[qoff // 4:qoff // 4 + 2])
    else:
        L.check(lib.dreg_surface_visibility(*common, L.stream()), "dreg_surface_visibility")
    return label > 0


@torch.no_grad()
def compute_visibility_score(xyz_list: List[torch.Tensor], nerf_model_path: str, delta: float = 1e-2, cut_off: float = 0.5,
                             score_type: str = "surface_field") -> List[torch.Tensor]:
    """Reference signature (confidence_loss.py:56-62): list of [num_layers, N, 3] -> list of [num_layers, N, 1] float {0,1}."""
    device = xyz_list[0].device
    field, binary, meta = load_block(nerf_model_path, device)
    out = []
    for xyz in xyz_list:
        nl, npnt = xyz.shape[0], xyz.shape[1]
        if score_type == "density_field":
            density, _ = field.query_raw(xyz.reshape(-1, 3))
            out.append(torch.clip(1 - torch.exp(-delta * density), 0, 1).view(nl, npnt, 1))
            continue
        lab = surface_visibility(xyz.reshape(-1, 3), meta["cam_centres_dev"], field, meta["binary_u8"], meta["aabb_host"], meta["aabb_host"],
                                 meta["render_step_size"], cut_off, 1e-4, float(meta.get("alpha_thre", 0.0) or 0.0), coarse_bits=meta["coarse_bits"])
        if device.type == "cuda":
            # the block's tensors may have been allocated on the loader's stream: tell the allocator that THIS stream reads them, so
            # that an evicted block's memory is not handed out again while the march above is still running
            cur = torch.cuda.current_stream(device)
            for t in (field._prepared()[0], meta["binary_u8"], meta["coarse_bits"], meta["cam_centres_dev"]):
                t.record_stream(cur)
        out.append(lab.float().view(nl, npnt, 1))
    return out


# Pinned staging for the descriptor tables: a ring of four buffers per device, each guarded by an event recorded behind its last copy
# (allocating / freeing pinned memory every step costs a host-side hipHostMalloc / hipHostFree pair, the latter a possible device sync).
_staging: Dict[str, list] = {}
STAGING_WAIT = [0.0]       # diagnostic: host seconds spent waiting for a staging slot
STAGING_RING = int(os.environ.get("DREG_STAGING_RING", "4"))


class _OverrunWatch:
    """The persistent kernels set bit 63 of a launch's ray counter(s) when a wave leaves the march loop through its safety bound with rays
    still queued or in flight (their points would stay unlabelled).  Reading the counters back right away would be a host sync per label
    launch; instead they are copied to a pinned slot behind the launch and looked at when a later call finds the copy done (or by
    check(wait=True): tests, the end of an extraction, a checkpoint).  The report is therefore LATE in training: train_step.TrainStep.step looks
    (without waiting) after its optimizer step, so a launch that hit its pass bound has already trained on unlabelled points for at least that step
    when the error is raised — the run stops, the checkpoint before it is the last clean state.  A launch with more counters than a slot holds is watched in several slots.
    Labels are requested from the loader thread and from the geometry thread: the slot lists are guarded by a lock."""

    SLOTS, WORDS = 8, 64           # pinned slots (allocated once), ray counters per slot

    def __init__(self):
        import threading
        self.pending = []          # (event, pinned int64 view, slot)
        self.free = None
        self.lock = threading.Lock()

    def watch(self, counters: torch.Tensor):
        if counters.device.type != "cuda":
            return
        words = counters.view(torch.int64).reshape(-1)
        with self.lock:
            if self.free is None:
                self.free = [(torch.cuda.Event(), torch.empty(self.WORDS, dtype=torch.int64).pin_memory()) for _ in range(self.SLOTS)]
            for o in range(0, words.numel(), self.WORDS):          # a batched label call with more than WORDS requests: one slot per chunk
                chunk = words[o:o + self.WORDS]
                if not self.free:
                    self._check_locked(False)
                    if not self.free:      # the GPU is eight label launches behind the host
                        self.pending[0][0].synchronize()
                        self._check_locked(False)
                ev, slot = self.free.pop()
                host = slot[:chunk.numel()]
                host.copy_(chunk, non_blocking=True)
                ev.record(torch.cuda.current_stream(counters.device))
                self.pending.append((ev, host, slot))

    def check(self, wait: bool = False):
        with self.lock:
            self._check_locked(wait)

    def _check_locked(self, wait: bool):
        keep, bad = [], False
        for ev, host, slot in self.pending:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, slot))
                continue
            bad = bad or bool((host < 0).any())
            self.free.append((ev, slot))
        self.pending = keep
        if bad:
            raise L.DregError("surface visibility: a persistent launch reached its pass bound with rays left — points of that call are "
                              "unlabelled (the march loop's pass bound of 2^22 per wave was reached: csrc/visibility.hip g_pass_bound)")


OVERRUN = _OverrunWatch()


def _desc_staging(nbytes: int, device):
    if torch.device(device).type != "cuda":
        return torch.empty(nbytes, dtype=torch.uint8), None
    ring = _staging.setdefault(str(device), [0, []])
    if len(ring[1]) < STAGING_RING:
        ring[1].append([torch.empty(max(nbytes, 16384), dtype=torch.uint8).pin_memory(), torch.cuda.Event()])
        slot = ring[1][-1]
    else:
        slot = ring[1][ring[0] % STAGING_RING]
        ring[0] += 1
        _t0 = __import__("time").perf_counter()
        slot[1].synchronize()                                   # only waits when the GPU is four label launches behind the host
        STAGING_WAIT[0] += __import__("time").perf_counter() - _t0
        if slot[0].numel() < nbytes:
            slot[0] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    return slot[0], slot[1]


@torch.no_grad()
def compute_visibility_scores_batched(requests, cut_off: float = 0.5, max_waves: int = 0) -> List[torch.Tensor]:
    """requests: list of (xyz [L,N,3], nerf_model_path) -> list of [L,N,1] float {0,1}, the labels of compute_visibility_score for each —
    from ONE launch over all blocks (a training step asks for 8: two per pair).  A call's duration is its longest ray, so eight
    launches cost eight tails; here every wave of one persistent launch works through all the blocks' ray queues."""
    if not requests:
        return []
    lib = L.load()
    device = requests[0][0].device
    blocks = [load_block(path, device) for _, path in requests]
    npts = [int(x.shape[0] * x.shape[1]) for x, _ in requests]
    # one zeroed buffer: every block's labels, then one 8-byte ray counter per block
    lab_off, off = [], 0
    for n in npts:
        lab_off.append(off)
        off += n + (n & 1)
    q_off = off
    buf = torch.zeros(off + 2 * len(requests), dtype=torch.int32, device=device)
    nb = int(lib.dreg_surface_visibility_desc_bytes())
    host, host_ev = _desc_staging(len(requests) * nb, device)
    host = host[:len(requests) * nb].view(len(requests), nb)
    f6 = lambda v: (ctypes.c_float * 6)(*[float(t) for t in v])
    keep, total = [], 0
    for i, ((xyz, _), (field, _, meta)) in enumerate(zip(requests, blocks)):
        pts = xyz.reshape(-1, 3).contiguous().float()
        keep.append(pts)
        base16, _ = field._prepared()
        b8, cams = meta["binary_u8"], meta["cam_centres_dev"]
        total += cams.shape[0] * pts.shape[0]
        L.check(lib.dreg_surface_visibility_fill_desc(host[i].data_ptr(), L.ptr(cams), L.ptr(pts), L.ptr(b8), buf.data_ptr() + 4 * lab_off[i],
                                                      base16.data_ptr() + 3072 * 2, base16.data_ptr(), base16.data_ptr() + 2048 * 2,
                                                      *field._levels, f6(meta["aabb_host"]), f6(meta["aabb_host"]), f6(field._aabb_host()),
                                                      b8.shape[0], b8.shape[1], b8.shape[2], cams.shape[0], pts.shape[0],
                                                      float(meta["render_step_size"]), float(cut_off), 1e-4, float(meta.get("alpha_thre", 0.0) or 0.0),
                                                      buf.data_ptr() + 4 * (q_off + 2 * i), L.ptr(meta["coarse_bits"]) if COARSE else None),
                "dreg_surface_visibility_fill_desc")
    descs = host.to(device, non_blocking=True)
    if host_ev is not None:
        host_ev.record(torch.cuda.current_stream(device))       # the staging buffer may be refilled once this copy has run
    OVERRUN.check()
    # max_waves > 0: a background launch (fewer resident waves, less LDS taken from kernels on other streams; see include/dreg_nerf.h)
    L.check(lib.dreg_surface_visibility_multi_waves(L.ptr(descs), len(requests), total, int(max_waves), L.stream()), "dreg_surface_visibility_multi_waves")
    OVERRUN.watch(buf[q_off:q_off + 2 * len(requests)])
    if device.type == "cuda":
        cur = torch.cuda.current_stream(device)
        for field, _, meta in blocks:
            for t in (field._prepared()[0], meta["binary_u8"], meta["coarse_bits"], meta["cam_centres_dev"]):
                t.record_stream(cur)
    return [(buf[lab_off[i]:lab_off[i] + npts[i]] > 0).float().view(x.shape[0], x.shape[1], 1) for i, (x, _) in enumerate(requests)]

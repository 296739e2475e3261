"""BASELINE.json configs[4] as a PIPELINE: grid extraction of NeRF blocks (eval_ngp_nerf.py:336-451 of the reference) and the
registration of the extracted pairs (eval_nerf_regtr.py:224-301) overlapped on one GPU.

The reference — and this package's serial path, eval_ngp_nerf.extract_block — handles one block at a time: read model.pth, query the
grid, write voxel_grid.pt / voxel_mask.pt / the PLY and their density_voxel_* twins, next block; registration later reads the files
back, one pair per call.  A block's query is ~0.3 ms of GPU time between ~80 ms of checkpoint reading and ~120 MB of file writing, so
the serial chain leaves the GPU idle > 95 % of the time (2.95 pairs/s measured in round 5).  Here the same work is laid out in stages:

  loader threads   model.pth of block k+1.. is memory-mapped; only the field's parameters and the occupancy bits are touched: copied into the thread's
                   pinned staging, uploaded asynchronously (one upload stream shared by the loader threads), converted to fp16 on the device; the
                   occupancy grid's cells are counted ON THE HOST (so the query never reads a size back); an event per block;
  main thread      waits for block k's event on the GPU, enqueues the dense query, the surface ray march and both grid writers — no host
                   readback anywhere — then the device -> host-staging copies of the results on a copy stream;
  dispatcher       ONE thread waits for the copies' events in order, compares the device's cell count with the loader's, hands the block on;
  writer PROCESSES (spawned; grid_writer.worker_main) write the reference's six files of block k-1.. straight from the staging, which is shared
                   memory (files under /dev/shm mapped by both sides, page-locked here with hipHostRegister).  Writer THREADS in this process
                   (writer_mode="thread", the fallback without /dev/shm) starve the launching thread: 8 of them cut its launch rate from 182 k/s
                   to 42 /s on the collection box (tools/writer_probe.py);
  registration     every `batch_pairs` finished pairs go through NeRFRegTr.forward_batch straight from the device tensors (the grids are
                   handed over in memory as dataset.SparseBlock; the files are still written), RRE / RTE stay on the device until the end.

Every file is byte-identical to the serial path's (tests/test_hip_eval_pipeline.py): same kernels on the same inputs, the jitter drawn
from the same generator in the same block order, torch.save of tensors with the same sizes and values.
"""
import concurrent.futures as cf
import os
import queue
import threading
import time
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import grid_writer as GW
from . import ngp, visibility
from .dataset import SparseBlock
from .vis_dump import write_ply


_slot_seq = [0]


def _unlink_leftovers():
    """At interpreter exit: staging files of this process that a pipeline did not get to close (an exception between construction and close())."""
    import glob
    for f in glob.glob(f"/dev/shm/dreg_{os.getpid()}_*"):
        try:
            os.unlink(f)
        except OSError:
            pass


import atexit  # noqa: E402
atexit.register(_unlink_leftovers)


def _host_register(t: torch.Tensor):
    """Page-lock a host mapping for asynchronous DMA (hipHostRegister through torch's runtime binding)."""
    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
    if int(rc) != 0:
        raise RuntimeError(f"hipHostRegister failed with {rc}")


class _Slot:
    """Host staging for the outputs of one block, allocated once and reused.  shared=True: files under /dev/shm mapped here (page-locked with
    hipHostRegister: the copies from the device are asynchronous DMA) and in the writer processes; shared=False: torch pinned memory (writer threads)."""

    def __init__(self, res: int, shared: bool):
        self.res, self.shared = res, shared
        _slot_seq[0] += 1
        self.id = _slot_seq[0]
        self.segs = {}
        nb = res * res * res * 7 * 4
        if shared:
            for name in ("dgrid", "grid"):
                sg = self.segs[name] = GW.Segment(f"/dev/shm/dreg_{os.getpid()}_{self.id}_{name}", nb, create=True)
                setattr(self, name, sg.tensor(torch.float32, (res, res, res, 7)))
                _host_register(getattr(self, name))
        else:
            self.dgrid = torch.empty(res, res, res, 7, dtype=torch.float32).pin_memory()
            self.grid = torch.empty(res, res, res, 7, dtype=torch.float32).pin_memory()
        self.counts = None        # this block's row of the run's pinned count table
        self.cap = 0
        self.stale = []           # segment paths the workers may still have mapped (re-made small segment)
        self.event = torch.cuda.Event()
        self.refs = None          # device tensors the copies read (kept alive until the writers are done)
        self.pending = 0
        self.lock = threading.Lock()

    def reserve(self, n: int):
        if n <= self.cap and self.cap > 0:          # (cap == 0: a fresh slot — an EMPTY block (n = 0) still needs its small buffers to exist)
            return
        cap = max(1 << max(n - 1, 1).bit_length(), 16384)
        if self.shared:
            old = self.segs.pop("small", None)
            if old is not None:
                self._drop(old)
                self.stale.append(old.path)
            off, total = GW.small_layout(cap)
            self._small_gen = getattr(self, "_small_gen", 0) + 1
            sg = self.segs["small"] = GW.Segment(f"/dev/shm/dreg_{os.getpid()}_{self.id}_small{self._small_gen}", total, create=True)
            whole = sg.tensor(torch.uint8, (total,))
            _host_register(whole)
            self._small_whole = whole
            self.world, self.rgb = sg.tensor(torch.float32, (cap, 3), off["world"]), sg.tensor(torch.float32, (cap, 3), off["rgb"])
            self.dmask, self.mask = sg.tensor(torch.int64, (cap,), off["dmask"]), sg.tensor(torch.int64, (cap,), off["mask"])
            self.dkeep, self.keep = sg.tensor(torch.uint8, (cap,), off["dkeep"]), sg.tensor(torch.uint8, (cap,), off["keep"])
        else:
            self.world = torch.empty(cap, 3, dtype=torch.float32).pin_memory()
            self.rgb = torch.empty(cap, 3, dtype=torch.float32).pin_memory()
            self.dmask = torch.empty(cap, dtype=torch.int64).pin_memory()
            self.mask = torch.empty(cap, dtype=torch.int64).pin_memory()
            self.dkeep = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self.keep = torch.empty(cap, dtype=torch.uint8).pin_memory()
        self.cap = cap

    def _drop(self, sg):
        t = sg.tensor(torch.uint8, (sg.nbytes,))
        try:
            torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
        except Exception:      # noqa: BLE001
            pass
        try:
            os.unlink(sg.path)
        except OSError:
            pass

    def destroy(self):
        if self.shared:
            torch.cuda.synchronize()
            for name in ("dgrid", "grid", "world", "rgb", "dmask", "mask", "dkeep", "keep", "_small_whole"):
                if hasattr(self, name):
                    delattr(self, name)
            for sg in self.segs.values():
                self._drop(sg)
            self.segs = {}


class Extracted:
    """One block's results on the device (no host sync has happened for them): voxel grids [res,res,res,7], mask buffers of capacity n whose first
    counts[0] (density mask) / counts[1] (density AND surface mask) entries are valid, and the event behind the pinned copy of `counts`."""
    __slots__ = ("path", "res", "n", "world", "rgb", "dgrid", "dmask", "grid", "mask", "counts_host", "counts_event", "counts_dev", "index")

    def kept(self) -> int:
        """Number of voxels in voxel_mask.pt (waits for this block's extraction on the GPU, not for its files)."""
        self.counts_event.synchronize()
        return int(self.counts_host[1])

    def sparse(self) -> SparseBlock:
        """The block as the registration network consumes it (dataset.SparseBlock.from_dense of the files, without the files)."""
        k = self.kept()
        idx = self.mask[:k]
        X = self.res
        return SparseBlock(idx, self.grid.view(-1, 7)[idx].contiguous(), (X, X, X))


class ExtractionPipeline:
    """with ExtractionPipeline(dev) as pipe: for ex in pipe.run(paths): ...   — blocks come out in order, their files are written behind;
    leaving the block waits for the writers and re-raises their errors.  `timings` accumulates thread-seconds per phase and what the main
    thread stood still for."""

    def __init__(self, device, loaders: int = 3, writers: int = 6, slots: int = 12, prefetch: int = 8, density_thre: float = 0.7,
                 write_files: bool = True, writer_mode: Optional[str] = None):
        """writer_mode 'process' (default where /dev/shm exists; DREG_WRITER_MODE overrides): `writers` spawned processes (grid_writer.worker_main) fed
        from shared-memory staging; 'thread': `writers` threads of this process on torch pinned memory (same files; see grid_writer's header for why not)."""
        self.dev = torch.device(device)
        self.density_thre = density_thre
        self.write_files = write_files
        self.n_slots, self.prefetch = slots, max(prefetch, 1)
        mode = writer_mode or os.environ.get("DREG_WRITER_MODE") or ("process" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "thread")
        self.writer_mode = mode if write_files else "thread"
        self._loaders = cf.ThreadPoolExecutor(max_workers=loaders, thread_name_prefix="dreg-load")
        self._writers = cf.ThreadPoolExecutor(max_workers=writers if self.writer_mode == "thread" else 1, thread_name_prefix="dreg-write")
        self._procs, self._outstanding, self._all_slots = [], 0, []
        self._pcond = threading.Condition()
        self._proc_err = None
        if self.writer_mode == "process":
            import multiprocessing as mp
            ctx = mp.get_context("spawn")            # (never fork a process that holds a HIP context and running threads)
            self._jobq, self._doneq = ctx.Queue(), ctx.Queue()
            self._procs = [ctx.Process(target=GW.worker_main, args=(self._jobq, self._doneq), daemon=True, name=f"dreg-writer-{i}") for i in range(writers)]
            for pr in self._procs:
                pr.start()
            self._collector = threading.Thread(target=self._collect, name="dreg-collect", daemon=True)
            self._collector.start()
        self._tls = threading.local()
        self._shared_load_stream = None
        self._free: "queue.Queue[_Slot]" = queue.Queue()
        self._slots_made = 0
        self._jobs: List[cf.Future] = []
        self._copy_stream = torch.cuda.Stream(device=self.dev)
        self._grids = {}
        self._tlock = threading.Lock()
        self.timings = {"host_detail_s": {}, "load_thread_s": 0.0, "write_thread_s": 0.0, "load_wait_s": 0.0, "slot_wait_s": 0.0, "enqueue_s": 0.0, "flush_s": 0.0,
                        "gpu_query_ms": 0.0, "gpu_surface_ms": 0.0, "gpu_grids_ms": 0.0, "gpu_copy_ms": 0.0, "blocks": 0, "bytes_written": 0}
        self._gpu_events = []
        self._dq: "queue.Queue" = queue.Queue()
        self._jlock = threading.Lock()
        self._dispatch_err = None
        self._debug_skip = os.environ.get("DREG_PIPE_SKIP", "")       # measurement only: "copy" / "write" leave that stage out (wrong files)
        self._dispatcher = threading.Thread(target=self._dispatch, name="dreg-dispatch", daemon=True)
        self._dispatcher.start()

    # ------------------------------------------------------------------ stage 1: loader threads
    def _load(self, path: str):
        t0 = time.perf_counter()
        torch.cuda.set_device(self.dev)
        st = getattr(self._tls, "stream", None)
        if st is None:
            # one upload stream for all loader threads by default: every additional HIP stream in flight costs the others (five streams halved the training
            # step on the collection box); DREG_PIPE_LOADER_STREAMS=per_thread gives each thread its own
            if os.environ.get("DREG_PIPE_LOADER_STREAMS", "shared") == "per_thread":
                st = torch.cuda.Stream(device=self.dev)
            else:
                with self._tlock:
                    if self._shared_load_stream is None:
                        self._shared_load_stream = torch.cuda.Stream(device=self.dev)
                    st = self._shared_load_stream
            self._tls.stream = st
        with torch.cuda.stream(st):
            field, binary, meta = visibility.load_block(path, self.dev, cache=False)
            ev = torch.cuda.Event()
            ev.record(st)
        with self._tlock:
            self.timings["load_thread_s"] += time.perf_counter() - t0
        return field, binary, meta, ev

    # ------------------------------------------------------------------ stage 2: the query, enqueued by the caller's thread
    def _sample_grid(self, meta) -> "ngp.SampleGrid":
        res = int(meta["grid_resolution"])
        key = (tuple(meta["aabb_host"]), res, getattr(meta["contraction_type"], "name", "AABB"))
        sg = self._grids.get(key)
        if sg is None:
            sg = self._grids[key] = ngp.SampleGrid(meta["aabb_host"], res, meta["contraction_type"]).to(self.dev)
            sg._aabb_host6 = [float(v) for v in meta["aabb_host"]]
        return sg

    @torch.no_grad()
    def _extract(self, path, loaded, index) -> Extracted:
        field, binary, meta, ev = loaded
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(ev)
        for t in (field._prepared()[0], field._prepared()[1], field.aabb, meta["binary_u8"], meta["coarse_bits"], meta["cam_centres_dev"]):
            t.record_stream(main)                     # allocated on a loader stream, read on this one
        sg = self._sample_grid(meta)
        sg.set_binary_fields(binary)
        res = int(meta["grid_resolution"])
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        hd = self.timings["host_detail_s"]
        h0 = time.perf_counter()
        e[0].record(main)
        world, indices, raw, alpha, dkeep, rows = sg._cells_and_density_fused(field, self.dev, self.density_thre, None, n_known=meta["n_occupied"])
        h1 = time.perf_counter()
        rgb = field.query_rgb_mean(raw, sg._viewdirs_on(self.dev))
        e[1].record(main)
        h2 = time.perf_counter()
        # surface mask (sample_grid.py:244-318) from the block's training cameras
        smask = visibility.surface_visibility(world, meta["cam_centres_dev"], field, meta["binary_u8"], sg._aabb_host6, meta["aabb_host"],
                                              meta["render_step_size"], 0.5, 1e-4, float(meta.get("alpha_thre", 0.0) or 0.0), coarse_bits=meta["coarse_bits"])
        keep = dkeep & smask.view(torch.uint8)
        e[2].record(main)
        h3 = time.perf_counter()
        ex = Extracted()
        ex.path, ex.res, ex.n, ex.index = path, res, int(meta["n_occupied"]), index
        ex.world, ex.rgb = world, rgb
        # density-field twins first (eval_ngp_nerf.py:350-381), then surface AND density (:383-412)
        ex.dgrid, ex.dmask, kd = ngp.write_kept_async(rows, world, rgb, alpha, indices, dkeep, res, grid=rows[3])
        ex.grid, ex.mask, k = ngp.write_kept_async(rows, world, rgb, alpha, indices, keep, res)
        counts = torch.cat([kd, k, rows[1][0:1]])       # (+ the device's own count of occupied cells: compared with the loader's before anything is written)
        e[3].record(main)
        h4 = time.perf_counter()
        for key, dt in (("cells_density", h1 - h0), ("colour", h2 - h1), ("surface", h3 - h2), ("grid_writers", h4 - h3)):
            hd[key] = hd.get(key, 0.0) + dt
        self._gpu_events.append(e)
        return ex, dkeep, keep, counts

    # ------------------------------------------------------------------ stage 3: device -> pinned host, then the writer threads
    def _slot(self, res: int) -> _Slot:
        t0 = time.perf_counter()
        try:
            s = self._free.get_nowait()
        except queue.Empty:
            if self._slots_made < self.n_slots:
                self._slots_made += 1
                s = _Slot(res, self.writer_mode == "process")
                self._all_slots.append(s)
            else:
                s = self._free.get()
        if s.res != res:
            self._all_slots.remove(s)
            s.destroy()
            s = _Slot(res, self.writer_mode == "process")
            self._all_slots.append(s)
        self.timings["slot_wait_s"] += time.perf_counter() - t0
        return s

    def _stage_out(self, ex: Extracted, dkeep, keep, counts):
        main = torch.cuda.current_stream(self.dev)
        hd = self.timings["host_detail_s"]
        # the counts first, on the main stream: registration only waits for these
        row = self._counts[ex.index]
        row.copy_(counts, non_blocking=True)
        ev_c = torch.cuda.Event()
        ev_c.record(main)
        ex.counts_host, ex.counts_event, ex.counts_dev = row, ev_c, counts
        if not self.write_files:          # extraction for an in-memory consumer only: no staging, no files
            return
        h0 = time.perf_counter()
        slot = self._slot(ex.res)
        slot.reserve(ex.n)
        slot.counts = row
        n = ex.n
        hd["slot"] = hd.get("slot", 0.0) + time.perf_counter() - h0
        h0 = time.perf_counter()
        cs = self._copy_stream
        cs.wait_event(ev_c)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(cs):
            c0.record(cs)
            if self._debug_skip == "copy":         # measurement only (tools/chain_probe.py): files written from whatever the staging buffers hold
                n = 0
            else:
                slot.dgrid.copy_(ex.dgrid, non_blocking=True)
                slot.grid.copy_(ex.grid, non_blocking=True)
            slot.world[:n].copy_(ex.world[:n], non_blocking=True)
            slot.rgb[:n].copy_(ex.rgb[:n], non_blocking=True)
            slot.dmask[:n].copy_(ex.dmask[:n], non_blocking=True)
            slot.mask[:n].copy_(ex.mask[:n], non_blocking=True)
            slot.dkeep[:n].copy_(dkeep[:n], non_blocking=True)
            slot.keep[:n].copy_(keep[:n], non_blocking=True)
            c1.record(cs)
            slot.event.record(cs)
        self._gpu_events.append((c0, c1))
        slot.refs = (ex.dgrid, ex.grid, ex.world, ex.rgb, ex.dmask, ex.mask, dkeep, keep, counts)
        slot.pending = 3
        self._dq.put((slot, slot.event, os.path.dirname(ex.path), n))
        hd["copies"] = hd.get("copies", 0.0) + time.perf_counter() - h0

    def _dispatch(self):
        """ONE thread waits for the copies' events, in order, and hands the finished blocks to the writer pool (sixteen threads each blocking in
        hipEventSynchronize kept the runtime's locks busy under the main thread's launches)."""
        while True:
            item = self._dq.get()
            if item is None:
                return
            if isinstance(item, threading.Event):      # flush(): everything queued before it has been handed on
                item.set()
                continue
            slot, ev, out_dir, n = item
            try:
                ev.synchronize()
            except BaseException as e:                  # noqa: BLE001
                self._dispatch_err = self._dispatch_err or e
            if int(slot.counts[2]) != n and self._debug_skip != "copy":
                # the query ran with the loader's host-side count of occupied cells; the device counted a different number: nothing of this block is written
                self._dispatch_err = self._dispatch_err or RuntimeError(f"{out_dir}: occupancy grid has {int(slot.counts[2])} occupied cells on the device, {n} on the host")
                slot.refs = None
                self._free.put(slot)
                continue
            if self.writer_mode == "process":
                if self._debug_skip == "write":
                    slot.refs = None
                    self._free.put(slot)
                    continue
                kd, k = int(slot.counts[0]), int(slot.counts[1])
                with self._pcond:
                    self._outstanding += 3
                if slot.stale:
                    for _ in self._procs:       # (every worker sees it at most once more than needed; harmless)
                        self._jobq.put(("drop", list(slot.stale)))
                    slot.stale = []
                nb = slot.res ** 3 * 28
                self._jobq.put(("grid", slot.id, slot.segs["dgrid"].path, nb, slot.res, os.path.join(out_dir, "density_voxel_grid.pt")))
                self._jobq.put(("grid", slot.id, slot.segs["grid"].path, nb, slot.res, os.path.join(out_dir, "voxel_grid.pt")))
                self._jobq.put(("small", slot.id, slot.segs["small"].path, slot.segs["small"].nbytes, slot.cap, n, kd, k, out_dir))
                continue
            with self._jlock:
                for job in (self._write_dgrid, self._write_grid, self._write_small):
                    self._jobs.append(self._writers.submit(self._run_job, job, slot, out_dir, n))

    def _collect(self):
        """Completions of the writer processes: slots go back to the free list, errors are kept for flush()."""
        by_id = {}
        while True:
            msg = self._doneq.get()
            if msg is None:
                return
            if msg[0] == "ready":
                continue
            _, slot_id, kind, nbytes, secs, err = msg
            slot = by_id.get(slot_id)
            if slot is None:
                by_id = {sl.id: sl for sl in self._all_slots}
                slot = by_id.get(slot_id)
            with self._tlock:
                self.timings["write_thread_s"] += secs
                self.timings["bytes_written"] += nbytes
            if slot is not None:
                with slot.lock:
                    slot.pending -= 1
                    free = slot.pending == 0
                if free:
                    slot.refs = None
                    self._free.put(slot)
            with self._pcond:
                if err is not None and self._proc_err is None:
                    self._proc_err = RuntimeError(f"grid writer process: {kind} job failed: {err}")
                self._outstanding -= 1
                self._pcond.notify_all()

    def _run_job(self, job, slot: _Slot, out_dir: str, n: int):
        t0 = time.perf_counter()
        try:
            nbytes = job(slot, out_dir, n) if self._debug_skip != "write" else 0
        finally:
            with slot.lock:
                slot.pending -= 1
                done = slot.pending == 0
            if done:
                slot.refs = None
                self._free.put(slot)
        with self._tlock:
            self.timings["write_thread_s"] += time.perf_counter() - t0
            self.timings["bytes_written"] += nbytes

    @staticmethod
    def _write_dgrid(slot, out_dir, n):
        torch.save(slot.dgrid, os.path.join(out_dir, "density_voxel_grid.pt"))
        return slot.dgrid.numel() * 4

    @staticmethod
    def _write_grid(slot, out_dir, n):
        torch.save(slot.grid, os.path.join(out_dir, "voxel_grid.pt"))
        return slot.grid.numel() * 4

    @staticmethod
    def _write_small(slot, out_dir, n):
        kd, k = int(slot.counts[0]), int(slot.counts[1])
        # fresh tensors of exactly the mask's length: torch.save writes a tensor's whole storage
        torch.save(slot.dmask[:kd].clone(), os.path.join(out_dir, "density_voxel_mask.pt"))
        torch.save(slot.mask[:k].clone(), os.path.join(out_dir, "voxel_mask.pt"))
        world, rgb = slot.world[:n].numpy(), slot.rgb[:n].numpy()
        dsel, sel = slot.dkeep[:n].numpy().astype(bool), slot.keep[:n].numpy().astype(bool)
        write_ply(os.path.join(out_dir, "density_voxel_point_cloud.ply"), world[dsel], rgb[dsel])
        write_ply(os.path.join(out_dir, "voxel_point_cloud.ply"), world[sel], rgb[sel])
        return 8 * (kd + k) + 27 * (int(dsel.sum()) + int(sel.sum()))

    # ------------------------------------------------------------------ driver
    def run(self, paths: Sequence[str]):
        paths = list(paths)
        loads = {}
        nxt = 0
        self._counts = torch.zeros(max(len(paths), 1), 3, dtype=torch.int32).pin_memory()     # (density-mask, kept) voxels per block, filled behind each extraction

        def top_up(upto):
            nonlocal nxt
            while nxt < min(upto, len(paths)):
                loads[nxt] = self._loaders.submit(self._load, paths[nxt])
                nxt += 1
        top_up(self.prefetch)
        for i, path in enumerate(paths):
            t0 = time.perf_counter()
            loaded = loads.pop(i).result()
            self.timings["load_wait_s"] += time.perf_counter() - t0
            top_up(i + 1 + self.prefetch)
            t0 = time.perf_counter()
            ex, dkeep, keep, counts = self._extract(path, loaded, i)
            del loaded
            self._stage_out(ex, dkeep, keep, counts)
            self.timings["enqueue_s"] += time.perf_counter() - t0 - 0.0
            self.timings["blocks"] += 1
            yield ex

    def flush(self):
        t0 = time.perf_counter()
        handed_on = threading.Event()
        self._dq.put(handed_on)
        handed_on.wait()
        with self._jlock:
            jobs, self._jobs = self._jobs, []
        err, self._dispatch_err = self._dispatch_err, None
        if self._procs:
            with self._pcond:
                while self._outstanding > 0:
                    if not self._pcond.wait(timeout=5.0) and not all(pr.is_alive() for pr in self._procs):
                        raise RuntimeError("a grid writer process died with jobs outstanding")
                err, self._proc_err = err or self._proc_err, None
        for j in jobs:
            try:
                j.result()
            except BaseException as e:          # noqa: BLE001 — the first writer error is re-raised after every job has finished
                err = err or e
        self.timings["flush_s"] += time.perf_counter() - t0
        visibility.OVERRUN.check(wait=True)        # a surface-label launch that hit its pass bound is an error of this run
        for e in self._gpu_events:
            if len(e) == 4:
                self.timings["gpu_query_ms"] += e[0].elapsed_time(e[1])
                self.timings["gpu_surface_ms"] += e[1].elapsed_time(e[2])
                self.timings["gpu_grids_ms"] += e[2].elapsed_time(e[3])
            else:
                self.timings["gpu_copy_ms"] += e[0].elapsed_time(e[1])
        self._gpu_events = []
        if err is not None:
            raise err

    def close(self):
        if getattr(self, "_closed", False):
            return
        self._closed = True
        self._dq.put(None)
        self._dispatcher.join()
        self._loaders.shutdown(wait=True)
        self._writers.shutdown(wait=True)
        if self._procs:
            for _ in self._procs:
                self._jobq.put(None)
            for pr in self._procs:
                pr.join(timeout=10.0)
                if pr.is_alive():
                    pr.terminate()
            self._doneq.put(None)
            self._collector.join()
        for sl in self._all_slots:
            sl.destroy()
        self._all_slots = []

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                self.flush()
        finally:
            self.close()
        return False


@torch.no_grad()
def extract_and_register(scenes, model, device, batch_pairs: int = 4, pipeline: Optional[ExtractionPipeline] = None, **pipe_kw):
    """scenes: list of (name, model.pth of the source block, model.pth of the target block, pose_gt [4,4] taking source to target coordinates).
    Extracts both grids of every scene (files written as eval_ngp_nerf.py does) and registers the pairs `batch_pairs` at a time from the
    device-resident grids.  Returns (rows, timings): rows = {name: {"R_mean", "t_mean", "R_med", "t_med", "time", "voxels": [k_src, k_tgt]}} with
    the reference's metric definitions (eval_nerf_regtr.py:24-65; `time` = the batch's forward time / its pairs)."""
    from .losses import rre_rte
    dev = torch.device(device)
    own = pipeline is None
    pipe = pipeline or ExtractionPipeline(dev, **pipe_kw)
    paths = [p for s in scenes for p in (s[1], s[2])]
    pending, results, reg = [], [], {"register_s": 0.0, "gpu_register_ms": 0.0, "calls": 0}
    evs = []

    def register(group):
        t0 = time.perf_counter()
        batch = []
        for (name, _, _, pose), (a, b) in group:
            batch.append({"src_sparse": a.sparse(), "tgt_sparse": b.sparse(), "pose": torch.as_tensor(pose, dtype=torch.float32)[None].to(dev, non_blocking=True),
                          "src_nerf_path": "", "tgt_nerf_path": ""})
        main = torch.cuda.current_stream(dev)
        # the sparse blocks were gathered on THIS stream just now; forward_batch's geometry phase reads them on its own (side) stream: hand it the event
        # (without it the side stream can read the coordinates before the gather has written them — seen once as "voxel coordinates overflow")
        ready = torch.cuda.Event()
        ready.record(main)
        for d in batch:
            d["ready_event"] = ready
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        preds = model.forward_batch(batch)
        e1.record(main)
        evs.append((e0, e1, len(group)))
        for (scene, (a, b)), pred, d in zip(group, preds, batch):
            r, t = rre_rte(pred["pose"][-1], d["pose"])
            results.append((scene[0], r, t, len(evs) - 1, (a.kept(), b.kept())))      # (counts already read by sparse(); the device grids are dropped here)
        reg["register_s"] += time.perf_counter() - t0
        reg["calls"] += 1

    try:
        got = []
        for ex in pipe.run(paths):
            got.append(ex)
            if len(got) == 2:
                pending.append((scenes[ex.index // 2], tuple(got)))
                got = []
                if len(pending) == batch_pairs:
                    register(pending)
                    pending = []
        if pending:
            register(pending)
        torch.cuda.synchronize(dev)
        if hasattr(model, "check_inputs"):
            model.check_inputs()
        pipe.flush()
    finally:
        if own:
            pipe.close()
    rows = {}
    for name, r, t, ei, (ka, kb) in results:
        r, t = r.float().cpu(), t.float().cpu()
        e0, e1, npairs = evs[ei]
        rows[name] = {"R_mean": float(r.mean()), "t_mean": float(t.mean()), "R_med": float(r.median()), "t_med": float(t.median()),
                      "time": 1e-3 * e0.elapsed_time(e1) / npairs, "voxels": [ka, kb]}
    reg["gpu_register_ms"] = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
    return rows, {**pipe.timings, **reg}

"""Writer PROCESSES of the evaluation pipeline (dreg_nerf_amd/eval_pipeline.py): the reference's six files per NeRF block — voxel_grid.pt,
voxel_mask.pt, voxel_point_cloud.ply and their density_voxel_* twins (eval_ngp_nerf.py:350-412) — written from shared-memory staging buffers
that the GPU filled by DMA.

Why processes: eight or more writer THREADS in the launching process starved its kernel launches (tools/writer_probe.py on the collection box:
182 k launches/s alone, 42 /s next to eight torch.save threads; next to eight writer processes 224 k/s) and wrote slower (4.8 vs 8.4 GB/s).
This module is imported by the spawned children: it must stay light (torch, numpy, the PLY writer) and never touch the GPU.

Staging memory = files under /dev/shm mapped by parent and children; the parent additionally page-locks its mapping (hipHostRegister) so the
device -> host copies are asynchronous DMA.  torch.save of a tensor over such a mapping writes the same bytes as torch.save of an ordinary CPU
tensor with the same values (tests/test_hip_eval_pipeline.py compares the files with the serial path's)."""
import mmap
import os
import time

import numpy as np
import torch


class Segment:
    """A shared-memory region (a file under /dev/shm, unlinked by its creator) and typed tensor views of it."""

    def __init__(self, path: str, nbytes: int, create: bool):
        self.path, self.nbytes = path, int(nbytes)
        fd = os.open(path, os.O_RDWR | (os.O_CREAT | os.O_EXCL if create else 0), 0o600)
        try:
            if create:
                os.ftruncate(fd, self.nbytes)
            self.map = mmap.mmap(fd, self.nbytes)
        finally:
            os.close(fd)

    def tensor(self, dtype, shape, offset: int = 0) -> torch.Tensor:
        n = int(np.prod(shape))
        return torch.frombuffer(self.map, dtype=dtype, count=n, offset=offset).view(*shape)


def small_layout(cap: int):
    """Byte offsets of the per-block small arrays inside one segment: world / rgb fp32 [cap,3], dmask / mask int64 [cap], dkeep / keep uint8 [cap]."""
    off, o = {}, 0
    for name, per in (("world", 12), ("rgb", 12), ("dmask", 8), ("mask", 8), ("dkeep", 1), ("keep", 1)):
        off[name] = o
        o += (per * cap + 255) // 256 * 256
    return off, o


def write_grid(seg: Segment, res: int, path: str) -> int:
    t = seg.tensor(torch.float32, (res, res, res, 7))
    assert t.untyped_storage().nbytes() == t.numel() * 4        # torch.save writes a tensor's whole storage
    torch.save(t, path)
    return t.numel() * 4


def write_small(seg: Segment, cap: int, n: int, kd: int, k: int, out_dir: str) -> int:
    from .vis_dump import write_ply
    off, _ = small_layout(cap)
    dmask = seg.tensor(torch.int64, (cap,), off["dmask"])
    mask = seg.tensor(torch.int64, (cap,), off["mask"])
    # fresh tensors of exactly the masks' lengths
    torch.save(dmask[:kd].clone(), os.path.join(out_dir, "density_voxel_mask.pt"))
    torch.save(mask[:k].clone(), os.path.join(out_dir, "voxel_mask.pt"))
    world = seg.tensor(torch.float32, (cap, 3), off["world"])[:n].numpy()
    rgb = seg.tensor(torch.float32, (cap, 3), off["rgb"])[:n].numpy()
    dsel = seg.tensor(torch.uint8, (cap,), off["dkeep"])[:n].numpy().astype(bool)
    sel = seg.tensor(torch.uint8, (cap,), off["keep"])[:n].numpy().astype(bool)
    write_ply(os.path.join(out_dir, "density_voxel_point_cloud.ply"), world[dsel], rgb[dsel])
    write_ply(os.path.join(out_dir, "voxel_point_cloud.ply"), world[sel], rgb[sel])
    return 8 * (kd + k) + 27 * (int(dsel.sum()) + int(sel.sum()))


def worker_main(jobs, done):
    """Child process: jobs = ("grid", slot, shm path, nbytes, res, out file) | ("small", slot, shm path, nbytes, cap, n, kd, k, out dir) | None."""
    torch.set_num_threads(1)
    segs = {}

    def seg(path, nbytes):
        s = segs.get(path)
        if s is None or s.nbytes != nbytes:
            s = segs[path] = Segment(path, nbytes, create=False)
        return s

    done.put(("ready", os.getpid()))
    while True:
        job = jobs.get()
        if job is None:
            return
        t0 = time.perf_counter()
        try:
            if job[0] == "grid":
                _, slot, path, nbytes, res, out = job
                nb = write_grid(seg(path, nbytes), res, out)
            elif job[0] == "drop":           # the parent is about to unlink these segments (a slot was re-made at another size)
                for p in job[1]:
                    segs.pop(p, None)
                continue
            else:
                _, slot, path, nbytes, cap, n, kd, k, out_dir = job
                nb = write_small(seg(path, nbytes), cap, n, kd, k, out_dir)
            done.put(("done", slot, job[0], nb, time.perf_counter() - t0, None))
        except BaseException as e:           # noqa: BLE001 — reported to the parent, which re-raises in flush()
            done.put(("done", job[1], job[0], 0, time.perf_counter() - t0, repr(e)))

"""Registration dataset with the reference's on-disk layout and sample contract
(conerf/datasets/register/dataset.py:94-134,221-331):

  <root>/<dataset>/images/<scene>/world_frame_transforms.json      {block_id: 4x4}
  <root>/<dataset>/nerf_models/<scene>/block_<k>/{model.pth, voxel_grid.pt, voxel_mask.pt}
  <json_dir>/<dataset>.json + obj_id_names.json                     train/test scene lists

A sample: 'src_xyz_rgba'/'tgt_xyz_rgba' fp32 [1,7,Z,X,Y], 'src_mask'/'tgt_mask' int64, 'pose' [1,4,4] = T_tgt T_src^-1,
nerf paths.  Training-time augmentation (jitter sigma 0.005 on masked xyz, centred SE(3) perturbation std 0.1, random swap;
dataset.py:277-331) is applied on the tensors' device.  `SyntheticRegDataset` generates shell-R scenes so the
entry points run without the external Objaverse data."""
import json
import math
import os
import random
from typing import List

import torch

from . import synth


def se3_from_draws(phi: float, cos_theta: float, theta_n: float, trans_n, std: float) -> torch.Tensor:
    """The reference's small rigid perturbation as a function of its random draws (dataset.py:13-33,71-91): rotation axis uniform
    on the sphere (phi ~ U(0, 2 pi), cos(theta) ~ U(-1, 1)), angle = theta_n * std * pi / sqrt(3) with theta_n ~ N(0,1) (first-order
    Taylor form when the angle is ~0), translation = trans_n * std / sqrt(3) with trans_n ~ N(0,1)^3.  fp64 like the reference's numpy."""
    th = math.acos(max(-1.0, min(1.0, float(cos_theta))))
    ax = torch.tensor([math.sin(th) * math.cos(phi), math.sin(th) * math.sin(phi), math.cos(th)], dtype=torch.float64)
    omega = ax * (float(theta_n) * std * math.pi / math.sqrt(3.0))

    def hat(v):
        return torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]], dtype=torch.float64)

    ang = float(omega.norm())
    if abs(ang) <= 1e-8:                      # np.isclose(theta, 0.) -> I + hat(omega)
        R = torch.eye(3, dtype=torch.float64) + hat(omega)
    else:
        K = hat(omega / ang)
        R = torch.eye(3, dtype=torch.float64) + math.sin(ang) * K + (1.0 - math.cos(ang)) * (K @ K)
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = R
    T[:3, 3] = torch.as_tensor(trans_n, dtype=torch.float64).reshape(3) * (std / math.sqrt(3.0))
    return T.float()


def _small_se3(std: float, gen=None) -> torch.Tensor:
    """Random small rigid transform with the reference's distribution (_sample_se3_small, dataset.py:71-91); draws from torch's RNG
    (the reference uses numpy's) in the reference's order: phi, cos(theta), angle, translation."""
    u = torch.rand(2, generator=gen, dtype=torch.float64)
    n = torch.randn(4, generator=gen, dtype=torch.float64)
    return se3_from_draws(float(u[0]) * 2.0 * math.pi, float(u[1]) * 2.0 - 1.0, float(n[0]), n[1:], std)


class SparseBlock:
    """A voxel grid in the form the network consumes it: the occupied voxels only.  The reference's dense voxel_grid.pt
    ([X,Y,Z,7] fp32, 58.7 MB at 128^3) is zero outside voxel_mask.pt (eval_ngp_nerf.py:397-405), so (idx, vals) is lossless and
    ~100x smaller — what is cached on disk (voxel_sparse.pt) and shipped host-to-device.
    idx int64 [N] flat (x*Y + y)*Z + z (= voxel_mask.pt), vals fp32 [N,7] (xyz | rgb | alpha), res = (Z, X, Y)."""
    __slots__ = ("idx", "vals", "res")

    def __init__(self, idx, vals, res):
        self.idx, self.vals, self.res = idx, vals, tuple(int(r) for r in res)

    @staticmethod
    def from_dense(grid_xyz7: torch.Tensor, mask: torch.Tensor) -> "SparseBlock":
        """grid_xyz7: [X,Y,Z,7] as stored in voxel_grid.pt."""
        X, Y, Z, _ = grid_xyz7.shape
        return SparseBlock(mask.long().contiguous(), grid_xyz7.reshape(-1, 7)[mask.long()].float().contiguous(), (Z, X, Y))

    def to(self, device, non_blocking: bool = False) -> "SparseBlock":
        return SparseBlock(self.idx.to(device, non_blocking=non_blocking), self.vals.to(device, non_blocking=non_blocking), self.res)

    def dense(self) -> torch.Tensor:
        """The reference's network input layout [1,7,Z,X,Y]."""
        Z, X, Y = self.res
        g = torch.zeros(X * Y * Z, 7, dtype=torch.float32, device=self.vals.device)
        g[self.idx] = self.vals
        return g.view(X, Y, Z, 7).permute(3, 2, 0, 1).unsqueeze(0).contiguous()

    def n_voxels(self) -> int:
        Z, X, Y = self.res
        return Z * X * Y


def load_block_sparse(block_dir: str) -> SparseBlock:
    """voxel_sparse.pt if present, else built from voxel_grid.pt + voxel_mask.pt and cached next to them."""
    sp = os.path.join(block_dir, "voxel_sparse.pt")
    if os.path.exists(sp):
        d = torch.load(sp)
        return SparseBlock(d["idx"], d["vals"], d["res"])
    sb = SparseBlock.from_dense(torch.load(os.path.join(block_dir, "voxel_grid.pt")), torch.load(os.path.join(block_dir, "voxel_mask.pt")))
    try:
        torch.save({"idx": sb.idx, "vals": sb.vals, "res": sb.res}, sp)
    except OSError:
        pass
    return sb


def augment_sparse(data: dict, jitter: float = 0.005, std: float = 0.1, gen=None, draws=None, rng=None, cpu_gen=None) -> dict:
    """dataset.py:277-331 on sparse blocks, on whatever device they live on: jitter of the occupied voxels' xyz, a small SE(3)
    perturbation centred on the mean over ALL voxels of the grid (zeros included, as the reference does), random swap.
    draws (tests): dict with 'noise_src', 'noise_tgt', 'perturb' (4x4), 'perturb_source' (bool), 'swap' (bool).
    gen: torch.Generator on the blocks' device (jitter); cpu_gen: CPU generator (SE(3) draws); rng: a random.Random for the two coin
    flips — a loader thread passes its own three so that it never touches the process-global generators (PrefetchLoader)."""
    draws = draws or {}
    rng = rng or random
    for side in ("src", "tgt"):
        sb = data[side + "_sparse"]
        noise = draws.get("noise_" + side)
        if noise is None:
            noise = torch.randn(sb.vals.shape[0], 3, device=sb.vals.device, generator=gen) * jitter
        vals = sb.vals.clone()
        vals[:, :3] += noise.to(vals.device)
        data[side + "_sparse"] = SparseBlock(sb.idx, vals, sb.res)
    perturb = draws["perturb"] if "perturb" in draws else _small_se3(std, cpu_gen)
    psrc = draws["perturb_source"] if "perturb_source" in draws else (rng.random() > 0.5)
    side = "src" if psrc else "tgt"
    sb = data[side + "_sparse"]
    dev = sb.vals.device
    c = sb.vals[:, :3].sum(dim=0) / sb.n_voxels()      # mean over all voxels of the (zero-filled) dense grid
    # P = T(+c) perturb T(-c): x -> R (x - c) + t + c.  Rigid transforms are composed and inverted in closed form ([R t]^-1 = [R^T, -R^T t]) with small
    # device ops: torch.linalg.inv on the device is a solver call with a host readback, and a copy from pageable memory stalls the host — several
    # syncs per sample held the loader thread at ~8 ms per sample (the reference's arithmetic is inv(Tc) @ perturb @ Tc: same values to ~1e-7)
    pt = perturb.to(torch.float32)
    pt = pt.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else pt.to(dev)
    R, t = pt[:3, :3], pt[:3, 3]
    P = _rigid(R, t + c - R @ c)
    vals = sb.vals.clone()
    vals[:, :3] = vals[:, :3] @ P[:3, :3].T + P[:3, 3]
    data[side + "_sparse"] = SparseBlock(sb.idx, vals, sb.res)
    pose = data["pose"]
    pose = pose.to(dev) if (pose.device == dev or dev.type != "cuda") else pose.pin_memory().to(dev, non_blocking=True)
    data["pose"] = pose @ _rigid_inverse(P) if psrc else P @ pose
    swap = draws["swap"] if "swap" in draws else (rng.random() > 0.5)
    if swap:
        data["src_sparse"], data["tgt_sparse"] = data["tgt_sparse"], data["src_sparse"]
        if "src_nerf_path" in data:
            data["src_nerf_path"], data["tgt_nerf_path"] = data["tgt_nerf_path"], data["src_nerf_path"]
        data["pose"] = _rigid_inverse(data["pose"])
    return data


def _rigid(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[R t; 0 1] as a 4x4 on R's device (no host round trip)."""
    top = torch.cat([R, t.reshape(3, 1)], dim=1)
    return torch.cat([top, top.new_tensor([[0.0, 0.0, 0.0, 1.0]]) if top.device.type != "cuda" else _last_row(top.device)], dim=0)


_LAST_ROW = {}


def _last_row(dev):
    r = _LAST_ROW.get(str(dev))
    if r is None:
        r = _LAST_ROW[str(dev)] = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).pin_memory().to(dev, non_blocking=True)
    return r


def _rigid_inverse(T: torch.Tensor) -> torch.Tensor:
    """Inverse of a rigid 4x4 (or [1,4,4] / [4,4]) in closed form."""
    shape = T.shape
    M = T.reshape(4, 4)
    Rt = M[:3, :3].T
    return _rigid(Rt, -(Rt @ M[:3, 3])).reshape(shape)


def augment(data: dict, jitter: float = 0.005, std: float = 0.1, draws=None) -> dict:
    """dataset.py:277-331 on whatever device the tensors live on (dense grids).  draws: as augment_sparse (tests)."""
    draws = draws or {}
    for side in ("src", "tgt"):
        g, m = data[side + "_xyz_rgba"], data[side + "_mask"]
        flat = g[0, :3].permute(2, 3, 1, 0).reshape(-1, 3)
        noise = draws["noise_" + side].to(g.device) if ("noise_" + side) in draws else torch.randn(m.shape[0], 3, device=g.device) * jitter
        flat[m] = flat[m] + noise
        g[0, :3] = flat.view(g.shape[3], g.shape[4], g.shape[2], 3).permute(3, 2, 0, 1)
    perturb = (draws["perturb"] if "perturb" in draws else _small_se3(std)).to(data["pose"].device)
    side = "src" if (draws["perturb_source"] if "perturb_source" in draws else random.random() > 0.5) else "tgt"
    g, m = data[side + "_xyz_rgba"], data[side + "_mask"]
    flat = g[0, :3].permute(2, 3, 1, 0).reshape(-1, 3)
    c = flat.mean(dim=0)  # the reference centres on the mean over ALL voxels (zeros included), dataset.py:305-306
    Tc = torch.eye(4, device=g.device)
    Tc[:3, 3] = -c
    P = torch.linalg.inv(Tc) @ perturb @ Tc
    flat[m] = flat[m] @ P[:3, :3].T + P[:3, 3]
    g[0, :3] = flat.view(g.shape[3], g.shape[4], g.shape[2], 3).permute(3, 2, 0, 1)
    if side == "src":
        data["pose"] = data["pose"] @ torch.linalg.inv(P)
    else:
        data["pose"] = P @ data["pose"]
    if (draws["swap"] if "swap" in draws else random.random() > 0.5):
        for a, b in (("src_xyz_rgba", "tgt_xyz_rgba"), ("src_mask", "tgt_mask"), ("src_nerf_path", "tgt_nerf_path")):
            data[a], data[b] = data[b], data[a]
        data["pose"] = torch.linalg.inv(data["pose"])
    return data


def load_split(json_dir: str, dataset: str) -> dict:
    """{'train': [scene names], 'test': [...]} of `dataset` from the reference's split files (dataset.py:194-216): objaverse.json is
    {dataset: {split: [ids]}}; for the 'objaverse' entry the ids are object uids that obj_id_names.json maps to scene directory
    names.  A file that is directly {split: [names]} (single-dataset form) is accepted as well."""
    path = os.path.join(json_dir, "objaverse.json")
    if not os.path.exists(path):
        path = os.path.join(json_dir, f"{dataset}.json")
    splits = json.load(open(path))
    if dataset in splits and isinstance(splits[dataset], dict):
        split = splits[dataset]
        if dataset == "objaverse":
            id2name = json.load(open(os.path.join(json_dir, "obj_id_names.json")))
            split = {sp: [id2name[i] for i in ids] for sp, ids in split.items()}
        return split
    if all(k in splits for k in ("train", "test")):
        return splits
    raise KeyError(f"dataset '{dataset}' not in {path} (has: {sorted(splits)})")


class NeRFRegDataset:
    def __init__(self, root_fp: str, json_dir: str, dataset: str = "objaverse", split: str = "train", model_dir: str = "nerf_models",
                 sparse: bool = False, device=None, require_grids: bool = True):
        """sparse=True: samples carry 'src_sparse'/'tgt_sparse' (SparseBlock) instead of the dense grids — ~0.6 MB per block over PCIe
        instead of 58.7 MB — and the training augmentation runs on `device` after the upload.  require_grids=False: blocks count as usable when
        their NeRF checkpoint exists (eval_nerf_regtr.py --extract_grids writes the grids itself)."""
        self.mode = split
        self.sparse, self.device = sparse, device
        # sparse blocks that were uploaded once stay on the device (0.5-1 MB each: the 3,284 blocks of an Objaverse epoch are ~3 GB): from the second epoch on a
        # sample costs its augmentation only — no file read, no upload (DREG_BLOCK_DEVICE_CACHE=0 turns it off)
        self._dev_blocks = {} if (device is not None and torch.device(device).type == "cuda" and os.environ.get("DREG_BLOCK_DEVICE_CACHE", "1") == "1") else None
        self.meta = []
        scenes = load_split(json_dir, dataset)[split]
        skipped = []
        for scene in scenes:
            tf = os.path.join(root_fp, dataset, "images", scene, "world_frame_transforms.json")
            if not os.path.exists(tf):
                skipped.append(scene)
                continue
            transforms = {int(k): torch.tensor(v, dtype=torch.float32) for k, v in json.load(open(tf)).items()}
            blocks = {}
            for k in sorted(transforms):
                d = os.path.join(root_fp, dataset, model_dir, scene, f"block_{k}")
                # (voxel_sparse.pt alone counts: the lossless (idx, vals) cache load_block_sparse writes — a split whose 58.7 MB dense grids were dropped after caching)
                if os.path.exists(os.path.join(d, "voxel_grid.pt")) or (sparse and os.path.exists(os.path.join(d, "voxel_sparse.pt"))) or \
                        (not require_grids and os.path.exists(os.path.join(d, "model.pth"))):
                    blocks[k] = {"dir": d, "transform": transforms[k]}
            if len(blocks) >= 2:
                self.meta.append({"scene": scene, "dataset": dataset, "blocks": blocks})
            else:
                skipped.append(scene)
        if skipped:
            print(f"[WARNING] {len(skipped)} of {len(scenes)} {split} scenes of '{dataset}' have no usable blocks under {root_fp} "
                  f"(first: {skipped[0]})", flush=True)
        if not self.meta:
            raise FileNotFoundError(f"no {split} scene of '{dataset}' found under {os.path.join(root_fp, dataset)} "
                                    f"({len(scenes)} listed in {json_dir})")
        print(f"Loaded {len(self.meta)} {split} scenes.", flush=True)

    def __len__(self):
        return len(self.meta)

    def __getitem__(self, index):
        return self.get(index)

    def _sparse_block(self, block_dir: str) -> SparseBlock:
        if self._dev_blocks is None:
            return load_block_sparse(block_dir)
        sb = self._dev_blocks.get(block_dir)
        if sb is None:
            h = load_block_sparse(block_dir)
            sb = SparseBlock(h.idx.pin_memory().to(self.device, non_blocking=True), h.vals.pin_memory().to(self.device, non_blocking=True), h.res)
            # first upload only: wait for it on the uploading stream, so that a consumer on ANOTHER stream (forward_batch's geometry stream, when a script calls
            # model(dataset[i]) without a loader's ready_event) never reads the block before it has arrived
            torch.cuda.current_stream(torch.device(self.device)).synchronize()
            self._dev_blocks[block_dir] = sb
        return sb

    def draw_block_order(self, index, rng=None):
        """The shuffle `get` applies to a scene's block ids (quirk Q15), as a call of its own: evaluation draws it for every scene in scene order on
        every rank and passes it back as get(..., block_order=), so a scene's source / target assignment does not depend on how scenes are sharded."""
        ids = list(self.meta[index]["blocks"].keys())
        (rng or random).shuffle(ids)
        return ids

    def get(self, index, rng=None, gen=None, cpu_gen=None, block_order=None):
        """rng / gen / cpu_gen: the caller's own random.Random, device and CPU torch.Generator (PrefetchLoader's thread); default = the
        process-global generators, as the reference's dataset uses them."""
        sm = self.meta[index]
        ids = list(block_order) if block_order is not None else self.draw_block_order(index, rng)  # also in test mode, as the reference (quirk Q15)
        s, t = sm["blocks"][ids[0]], sm["blocks"][ids[1]]
        if self.sparse:
            data = {"src_sparse": self._sparse_block(s["dir"]), "tgt_sparse": self._sparse_block(t["dir"]),
                    "src_nerf_path": os.path.join(s["dir"], "model.pth"), "tgt_nerf_path": os.path.join(t["dir"], "model.pth"),
                    "pose": (t["transform"] @ torch.linalg.inv(s["transform"]))[None],
                    "scene": sm["scene"], "dataset": sm["dataset"], "index": index, "block_list": ids[:2]}
            if self.device is not None:
                data["src_sparse"], data["tgt_sparse"] = data["src_sparse"].to(self.device), data["tgt_sparse"].to(self.device)
                data["pose"] = data["pose"].pin_memory().to(self.device, non_blocking=True) if torch.device(self.device).type == "cuda" else data["pose"].to(self.device)
            if self.mode == "train":
                data["pose"] = data["pose"][0]
                data = augment_sparse(data, gen=gen, rng=rng, cpu_gen=cpu_gen)
                data["pose"] = data["pose"][None]
            return data
        data = {
            "src_xyz_rgba": torch.load(os.path.join(s["dir"], "voxel_grid.pt")).permute(3, 2, 0, 1).unsqueeze(0).contiguous(),
            "tgt_xyz_rgba": torch.load(os.path.join(t["dir"], "voxel_grid.pt")).permute(3, 2, 0, 1).unsqueeze(0).contiguous(),
            "src_mask": torch.load(os.path.join(s["dir"], "voxel_mask.pt")), "tgt_mask": torch.load(os.path.join(t["dir"], "voxel_mask.pt")),
            "src_nerf_path": os.path.join(s["dir"], "model.pth"), "tgt_nerf_path": os.path.join(t["dir"], "model.pth"),
            "pose": (t["transform"] @ torch.linalg.inv(s["transform"]))[None],
            "scene": sm["scene"], "dataset": sm["dataset"], "index": index, "block_list": ids[:2],
        }
        return augment(data) if self.mode == "train" else data


class SyntheticRegDataset:
    """N shell-R scenes with a fixed, scene-dependent relative pose (stand-in for the external Objaverse data)."""

    def __init__(self, n_scenes: int, res: int = 128, split: str = "train"):
        self.n, self.res, self.mode = n_scenes, res, split

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(1000 + index)
        pose = _small_se3(0.25, g)
        data = synth.shell_pair(self.res, 2 * index + 1, 2 * index + 2, pose=pose)
        data.update({"scene": f"shell_{index:04d}", "dataset": "synthetic", "index": index, "block_list": [0, 1]})
        return augment(data) if self.mode == "train" else data


class PrefetchLoader:
    """Iterates `indices` of a dataset `depth` samples ahead on a background thread: disk read (voxel_sparse.pt, ~0.6 MB per block),
    host-to-device copy and the on-device augmentation run on the loader's own HIP stream while the GPU trains on earlier samples
    (the reference's loop loads inline: train_nerf_regtr.py:142-145; at ~160 pairs/s that would leave the GPU idle most of the time).
    Every yielded sample carries 'ready_event' (recorded on the loader stream after its last device op); NeRFRegTr.forward_batch waits for
    it on the GPU, never on the host."""

    def __init__(self, dataset, indices, device=None, depth: int = 2, prefetch_nerf_blocks: bool = True, stream=None):
        import queue
        import threading
        self.ds, self.indices, self.device = dataset, list(indices), device
        # NeRF blocks behind the step's overlap labels (train_nerf_regtr.py:186-199): loaded into visibility's block cache HERE, on the
        # loader thread and stream, so that the training thread finds them resident (a block checkpoint is 60-160 MB on disk)
        self.prefetch_nerf_blocks = prefetch_nerf_blocks and device is not None and torch.device(device).type == "cuda"
        self.q = queue.Queue(maxsize=max(depth, 1))
        self.stream = stream if stream is not None else torch.cuda.Stream(device=device, priority=-1) if (device is not None and torch.device(device).type == "cuda") else None   # high priority: its few small kernels must not queue behind a saturated training stream
        self._err = None
        # The thread owns its generators (seeded from ONE draw of the caller's Python RNG at construction, on the caller's thread): the
        # block order / augmentation are reproducible for a fixed seed and never interleave with the main thread's draws.
        seed = random.getrandbits(62)
        self._rng = random.Random(seed)
        self._cpu_gen = torch.Generator().manual_seed(seed)
        # the jitter is drawn on the device the DATASET puts its tensors on (a CPU dataset behind a CUDA loader draws on the CPU generator)
        ds_dev = getattr(dataset, "device", None)
        on_gpu = ds_dev is not None and torch.device(ds_dev).type == "cuda"
        self._gen = torch.Generator(device=ds_dev).manual_seed(seed) if on_gpu else self._cpu_gen
        self._own_rng = hasattr(dataset, "get")
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            fetch = (lambda i: self.ds.get(i, rng=self._rng, gen=self._gen, cpu_gen=self._cpu_gen)) if self._own_rng else (lambda i: self.ds[i])
            for i in self.indices:
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        sample = fetch(i)
                        if self.prefetch_nerf_blocks:
                            from . import visibility
                            for k in ("src_nerf_path", "tgt_nerf_path"):
                                path = sample.get(k)
                                if path and os.path.exists(path):
                                    visibility.load_block(path, torch.device(self.device))
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    sample["ready_event"] = ev
                    sample["_loader_stream"] = self.stream
                else:
                    sample = fetch(i)
                self.q.put(sample)
        except BaseException as e:   # surfaced on the consumer side
            self._err = e
        finally:
            self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        return item

    def __len__(self):
        return len(self.indices)

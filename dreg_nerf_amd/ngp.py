"""Host-side mirror of the reference's Instant-NGP radiance field and dense grid sampler (rows B1-B4 of SURVEY.md §8),
running the hash-grid + MLP queries through ngp.hip.

  NGPradianceField   conerf/radiance_fields/ngp.py:66-208  (state_dict: aabb, mlp_base.params, color_mlp.params)
  SampleGrid         conerf/register/sample_grid.py:59-343 (dense-query part; the ray-marched surface mask is row N1)
  save_voxel_grid    eval_ngp_nerf.py:383-412              (voxel_grid.pt fp32 [X,Y,Z,7], voxel_mask.pt int64 ascending)
"""
import enum
import math
import os
import sys
import types
from typing import List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import lib as L

PER_LEVEL_SCALE = 1.4472692012786865


class ContractionType(enum.Enum):
    """Stand-in for nerfacc.ContractionType (NeRF checkpoints pickle this enum: train_ngp_nerf.py:204)."""
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


def install_pickle_shims():
    """Make torch.load of a reference NeRF checkpoint work without nerfacc: register a module named
    nerfacc.contraction exposing ContractionType (only when the real package is absent)."""
    try:
        import nerfacc  # noqa: F401
        return
    except Exception:
        pass
    pkg = types.ModuleType("nerfacc")
    sub = types.ModuleType("nerfacc.contraction")
    ContractionType.__module__ = "nerfacc.contraction"
    sub.ContractionType = ContractionType
    pkg.contraction = sub
    pkg.ContractionType = ContractionType
    sys.modules.setdefault("nerfacc", pkg)
    sys.modules.setdefault("nerfacc.contraction", sub)


def _level_table(log2_hashmap_size: int = 19):
    lib = L.load()
    import ctypes
    arrs = [(ctypes.c_uint32 * 16)(), (ctypes.c_uint32 * 16)(), (ctypes.c_uint32 * 16)(), (ctypes.c_float * 16)(), (ctypes.c_uint32 * 16)()]
    total = lib.dreg_ngp_level_table(PER_LEVEL_SCALE, log2_hashmap_size, 16, *arrs)
    return arrs, int(total)


class _Params(nn.Module):
    def __init__(self, n, init: bool = True):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(n, dtype=torch.float32) if init else torch.empty(n, dtype=torch.float32))


class NGPradianceField(nn.Module):
    """Instant-NGP radiance field with the reference's call surface: query_density(x, return_feat), query_rgb(dir, embedding),
    forward(positions, directions).  `embedding` here is the fp16 [N,16] raw network output (density logit | 15 features);
    query_density(..., return_feat=True) returns (density [N,1], feat [N,15] fp32) like the reference and caches nothing."""

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 unbounded: bool = False, geo_feat_dim: int = 15, n_levels: int = 16, log2_hashmap_size: int = 19, init: bool = True):
        """init = False: the 12.6 M parameters are left uninitialised (a state_dict is loaded right away: visibility.load_block)."""
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        assert n_levels == 16 and geo_feat_dim == 15 and num_dim == 3 and use_viewdirs
        self.register_buffer("aabb", aabb.float())
        self.unbounded = unbounded
        self.geo_feat_dim = geo_feat_dim
        self._levels, total = _level_table(log2_hashmap_size)
        self.mlp_base = _Params(3072 + 2 * total, init)
        self.color_mlp = _Params(7168, init)
        self._prep = None
        if init:
            self.reset_parameters()

    @classmethod
    def from_inference_copies(cls, aabb, unbounded: bool, base16: torch.Tensor, col16: torch.Tensor, device):
        """A field that is only queried, built straight from the fp16 inference copies of its parameters (mlp_base.params / color_mlp.params converted
        with round-to-nearest-even, tcnn's params_inference): the state freeze_for_inference() leaves, without ever holding the fp32 parameters on
        the device (visibility.load_block)."""
        dev = torch.device(device)
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        aabb_host = [float(v) for v in (aabb.tolist() if torch.is_tensor(aabb) else aabb)]
        self.register_buffer("aabb", L.to_device_async(aabb_host, torch.float32, dev) if dev.type == "cuda" else torch.tensor(aabb_host, dtype=torch.float32))
        self.unbounded, self.geo_feat_dim = bool(unbounded), 15
        self._levels, total = _level_table(19)
        assert base16.numel() == 3072 + 2 * total and col16.numel() == 7168 and base16.dtype == torch.float16 and col16.dtype == torch.float16
        self.mlp_base, self.color_mlp = _Params(0, False), _Params(0, False)
        self.mlp_base.params.data = torch.empty(0, device=dev)
        self.color_mlp.params.data = torch.empty(0, device=dev)
        self._prep = ("frozen", base16, col16)
        self.__dict__["_aabb_cache"] = ((self.aabb.data_ptr(), self.aabb._version), aabb_host)
        return self.eval()

    def reset_parameters(self):
        with torch.no_grad():  # tcnn defaults: hash table U(-1e-4, 1e-4), MLP weights Xavier-uniform
            p = self.mlp_base.params
            p[:2048].uniform_(-math.sqrt(6 / 96), math.sqrt(6 / 96))
            p[2048:3072].uniform_(-math.sqrt(6 / 80), math.sqrt(6 / 80))
            p[3072:].uniform_(-1e-4, 1e-4)
            c = self.color_mlp.params
            c[:2048].uniform_(-math.sqrt(6 / 96), math.sqrt(6 / 96))
            c[2048:6144].uniform_(-math.sqrt(6 / 128), math.sqrt(6 / 128))
            c[6144:].uniform_(-math.sqrt(6 / 80), math.sqrt(6 / 80))

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if k in ("aabb", "mlp_base.params", "color_mlp.params")}
        extra = [k for k in state_dict if k not in sd and state_dict[k].numel() > 0]
        if strict and extra:
            raise RuntimeError(f"unexpected non-empty keys in NeRF state_dict: {extra}")
        self._prep = None
        return super().load_state_dict(sd, strict=strict)

    # fp16 inference copies (tcnn's params_inference), refreshed when the fp32 parameters change
    @torch.no_grad()
    def freeze_for_inference(self):
        """Keep only the fp16 inference copies on the device and release the fp32 parameters (12.6 M hash-grid entries: 50 MB of the
        ~80 MB a loaded block holds).  For fields that are only queried — the NeRF blocks behind the training labels: all 3,284 blocks
        of an Objaverse epoch then fit in HBM (27 MB each) instead of being re-read from disk every time they fall out of a small cache.
        The field answers query_raw / query_rgb_mean / the visibility march as before; its parameters can no longer be read or trained."""
        base16, col16 = self._prepared()
        self._prep = ("frozen", base16, col16)
        dev = base16.device
        self.mlp_base.params.data = torch.empty(0, device=dev)
        self.color_mlp.params.data = torch.empty(0, device=dev)
        return self

    def _prepared(self):
        if self._prep is not None and self._prep[0] == "frozen":
            return self._prep[1], self._prep[2]
        key = (self.mlp_base.params._version, self.color_mlp.params._version, self.mlp_base.params.data_ptr())
        if self._prep is None or self._prep[0] != key:
            lib = L.load()
            dev = self.mlp_base.params.device
            base16 = torch.empty(self.mlp_base.params.numel(), dtype=torch.float16, device=dev)
            col16 = torch.empty(7168, dtype=torch.float16, device=dev)
            L.check(lib.dreg_f32_to_f16(L.ptr(self.mlp_base.params.detach()), L.ptr(base16), base16.numel(), L.stream()), "dreg_f32_to_f16")
            L.check(lib.dreg_f32_to_f16(L.ptr(self.color_mlp.params.detach()), L.ptr(col16), col16.numel(), L.stream()), "dreg_f32_to_f16")
            self._prep = (key, base16, col16)
        return self._prep[1], self._prep[2]

    def _aabb_host(self):
        """The six aabb floats on the host, cached against the buffer's version: `.tolist()` of a device tensor is a host sync."""
        c = self.__dict__.get("_aabb_cache")
        if c is None or c[0] != (self.aabb.data_ptr(), self.aabb._version):
            c = self.__dict__["_aabb_cache"] = ((self.aabb.data_ptr(), self.aabb._version), [float(v) for v in self.aabb.tolist()])
        return c[1]

    @torch.no_grad()
    def query_raw(self, x: torch.Tensor, order: Optional[torch.Tensor] = None, x_in_slot_order: bool = False):
        """x [N,3] world -> (density fp32 [N], raw fp16 [N,16]).  order (optional int32 [N] permutation): which points share a wave
        (SampleGrid.query_dense passes the x-fastest order of its cells); results do not depend on it and stay indexed by point.
        x_in_slot_order: x[j] is the position of point order[j] (then read contiguously)."""
        lib = L.load()
        import ctypes
        base16, _ = self._prepared()
        x = x.reshape(-1, 3).contiguous().float()
        n = x.shape[0]
        density = torch.empty(n, dtype=torch.float32, device=x.device)
        raw = torch.empty(n, 16, dtype=torch.float16, device=x.device)
        aabb = (ctypes.c_float * 6)(*self._aabb_host())
        # unbounded: contract_to_unisphere(x, aabb) before the hash grid (ngp.py:41-63,163-164).  Queries of a block's size go through the
        # two-launch form (hash-grid levels pinned to the XCDs' L2s: csrc/ngp.hip); a handful of points through the fused kernel
        nws = int(lib.dreg_ngp_density_workspace_bytes(n)) if n >= 16384 else 0
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
        L.check(lib.dreg_ngp_density_fwd_ws(L.ptr(x), base16.data_ptr() + 3072 * 2, base16.data_ptr(), base16.data_ptr() + 2048 * 2,
                                            L.ptr(density), L.ptr(raw), *self._levels, aabb, n, int(bool(self.unbounded)), L.ptr(ws), nws,
                                            L.ptr(order) if order is not None else None, int(bool(x_in_slot_order and order is not None)), L.stream()),
                "dreg_ngp_density_fwd_ws")
        return density, raw

    @torch.no_grad()
    def query_density(self, x, return_feat: bool = False):
        shp = x.shape[:-1]
        density, raw = self.query_raw(x)
        density = density.view(*shp, 1)
        if return_feat:
            return density, raw[:, 1:].float().view(*shp, self.geo_feat_dim)
        return density

    def dir_bias(self, dirs: torch.Tensor) -> torch.Tensor:
        """c_k = fp16(W1[:, :16]) . fp16(SH4(dir_k))  for the colour net's first layer: [ndir, 64] fp32."""
        _, col16 = self._prepared()
        if col16.is_cuda:      # one launch (csrc/ngp.hip) instead of ~30 elementwise ones: the dense query is host-bound on its glue
            d = dirs.to(col16.device).float().contiguous()
            # the library writes [K, 64] fp32 biases followed by [K, 64] packed fp16 (hi | lo) halves of the same values (the MFMA operand of
            # the chunked colour kernel); the returned view is the fp32 part, dreg_ngp_rgb_mean_fwd reads both from the same buffer
            out = torch.empty(2 * d.shape[0], 64, dtype=torch.float32, device=col16.device)
            L.check(L.load().dreg_ngp_dir_bias(L.ptr(d), col16.data_ptr(), L.ptr(out), d.shape[0], L.stream()), "dreg_ngp_dir_bias")
            return out[:d.shape[0]]
        w1 = col16[:2048].view(64, 32)[:, :16].float()
        sh = sh4(dirs.to(w1.device)).half().float()
        return (sh @ w1.T).contiguous()

    @torch.no_grad()
    def query_rgb_mean(self, raw: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
        """Mean over the given viewing directions of the colour net's output: raw fp16 [N,16], dirs [K,3] -> [N,3] fp32."""
        lib = L.load()
        _, col16 = self._prepared()
        n = raw.shape[0]
        rgb = torch.empty(n, 3, dtype=torch.float32, device=raw.device)
        bias = self.dir_bias(dirs)
        L.check(lib.dreg_ngp_rgb_mean_fwd(L.ptr(raw.contiguous()), col16.data_ptr(), col16.data_ptr() + 2048 * 2, col16.data_ptr() + 6144 * 2,
                                          L.ptr(bias), L.ptr(rgb), dirs.shape[0], n, L.stream()), "dreg_ngp_rgb_mean_fwd")
        return rgb

    @torch.no_grad()
    def query_rgb(self, dir, embedding):
        """Reference signature (ngp.py:178-193): one viewing direction per point, `embedding` = the 15 geometry features
        (query_density(..., return_feat=True)) -> rgb [..., 3].  A single direction shared by all points takes the mean-kernel's
        folded-bias route (the dense query's 18 passes, sample_grid.py:332-337); distinct directions evaluate SH per point."""
        lib = L.load()
        d = dir.reshape(-1, 3).float().contiguous()
        feat = embedding.reshape(-1, self.geo_feat_dim)
        raw = torch.cat([torch.zeros(feat.shape[0], 1, device=feat.device, dtype=feat.dtype), feat], dim=1).half().contiguous()
        _, col16 = self._prepared()
        n = raw.shape[0]
        assert d.shape[0] == n, f"{tuple(dir.shape)} v.s. {tuple(embedding.shape)}"
        rgb = torch.empty(n, 3, dtype=torch.float32, device=raw.device)
        L.check(lib.dreg_ngp_rgb_dir_fwd(L.ptr(raw), col16.data_ptr(), col16.data_ptr() + 2048 * 2, col16.data_ptr() + 6144 * 2, L.ptr(d.to(raw.device)),
                                         L.ptr(rgb), n, L.stream()), "dreg_ngp_rgb_dir_fwd")
        return rgb.view(*embedding.shape[:-1], 3).to(embedding.dtype)

    def forward(self, positions, directions=None):
        """ngp.py:195-208: (rgb, density) of points seen from per-point directions."""
        assert directions is not None and positions.shape == directions.shape, f"{positions.shape} v.s. {None if directions is None else directions.shape}"
        density, feat = self.query_density(positions, return_feat=True)
        return self.query_rgb(directions, feat), density


def sh4(d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)


class OccupancyGrid(nn.Module):
    """State holder for a nerfacc 0.3.5 ``OccupancyGrid`` (the 'occupancy_grid' entry of a NeRF block checkpoint,
    train_ngp_nerf.py:196; constructed as in conerf/loss/confidence_loss.py:42-46).  Persistent buffers of the original: _roi_aabb
    fp32 [6], resolution int32 [3], occs fp32 [res^3] (the EMA densities), _binary bool [res,res,res].  Training the grid
    (every_n_step) belongs to NeRF training, which is out of scope; the registration path only reads ``binary``."""
    NUM_DIM = 3
    _NON_PERSISTENT = ("grid_coords", "grid_indices")   # written by some nerfacc builds, rebuilt on construction there

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        resolution = torch.as_tensor(resolution, dtype=torch.int32).clone()
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).clone()
        assert resolution.shape == (3,), f"Invalid shape: {resolution.shape}"
        assert roi_aabb.shape == (6,), f"Invalid shape: {roi_aabb.shape}"
        self.num_cells = int(resolution.prod().item())
        self.register_buffer("_roi_aabb", roi_aabb)
        self.register_buffer("resolution", resolution)
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("_binary", torch.zeros(resolution.tolist(), dtype=torch.bool))
        self._contraction_type = contraction_type

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary

    @property
    def contraction_type(self):
        return self._contraction_type

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if k not in self._NON_PERSISTENT}
        return super().load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def query_occ(self, samples: torch.Tensor) -> torch.Tensor:
        """Occupancy of world points [N,3] (False outside the aabb): nerfacc's grid query for ContractionType.AABB."""
        lo, hi = self._roi_aabb[:3], self._roi_aabb[3:]
        u = (samples - lo) / (hi - lo)
        inside = ((u >= 0) & (u < 1)).all(dim=-1)
        res = self.resolution.to(samples.device)
        ijk = (u * res).long().clamp_(min=torch.zeros_like(res, dtype=torch.long), max=(res - 1).long())
        flat = (ijk[:, 0] * res[1] + ijk[:, 1]) * res[2] + ijk[:, 2]
        return self._binary.reshape(-1)[flat] & inside


FUSED_DENSE = True    # SampleGrid.query_dense on a GPU: cell lists / alpha / mask from the library's fused launches (False: torch.nonzero + separate launches)


class SampleGrid(nn.Module):
    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32)
        assert resolution.shape == (3,) and roi_aabb.shape == (6,)
        if getattr(contraction_type, "name", "AABB") != "AABB":
            raise NotImplementedError("only ContractionType.AABB is used by the registration data (config.py:63, Objaverse)")
        self.num_voxels = int(resolution.prod().item())
        self.register_buffer("_binary", torch.zeros(resolution.tolist(), dtype=torch.bool))
        self.register_buffer("resolution", resolution)
        self.register_buffer("_roi_aabb", roi_aabb)
        self._contraction_type = contraction_type
        self._delta = 1e-2
        self._viewdirs = self._generate_fixed_viewing_directions()

    @property
    def binary(self):
        return self._binary

    @torch.no_grad()
    def set_binary_fields(self, binary):
        self._binary = binary

    @staticmethod
    def _generate_fixed_viewing_directions() -> torch.Tensor:
        """sample_grid.py:131-145, kept as written there (x == y, not unit length; quirk Q10)."""
        phis = [math.pi / 3, 0, -math.pi]
        thetas = [k * math.pi / 3 for k in range(6)]
        return torch.tensor([[math.cos(p) * math.sin(t), math.cos(p) * math.sin(t), math.sin(t)] for p in phis for t in thetas],
                            dtype=torch.float32)

    @torch.no_grad()
    def uniform_sample_occupied_voxels(self) -> torch.Tensor:
        return torch.nonzero(self._binary.flatten())[:, 0]

    def _viewdirs_on(self, device) -> torch.Tensor:
        """The 18 fixed viewing directions on `device` (copied once per device: a pageable host -> device copy per block is a host sync)."""
        c = self.__dict__.setdefault("_viewdirs_dev", {})
        key = str(torch.device(device))
        if key not in c:
            c[key] = self._viewdirs.to(device).float().contiguous()
        return c[key]

    @torch.no_grad()
    def query_dense(self, radiance_field: NGPradianceField, device, density_thre: float = 0.7, jitter: Optional[torch.Tensor] = None):
        """Dense-query part of query_radiance_and_density_from_camera (sample_grid.py:223-242, 321-341).
        jitter [Np,3] in [0,1): drawn with torch.rand on `device` when not given (reference: rand_like, quirk Q11).
        On a GPU the cell lists come from the library (FUSED_DENSE: csrc/ngp.hip dreg_grid_occupied_*: no torch.nonzero, one host
        readback), alpha / the density mask from the density launch; the returned mask then carries what build_voxel_grid needs to write
        voxel_grid / voxel_mask without another compaction (`_dreg_rows`)."""
        lib = L.load()
        import ctypes
        dev = torch.device(device)
        if FUSED_DENSE and dev.type == "cuda" and self._binary.dim() == 3 and max(self._binary.shape) <= 65535:
            return self._query_dense_fused(radiance_field, dev, density_thre, jitter)
        indices = self.uniform_sample_occupied_voxels().to(device)
        n = indices.shape[0]
        if jitter is None:
            jitter = torch.rand(n, 3, dtype=torch.float32, device=device)
        jitter = jitter.to(device).contiguous()
        world, density, raw = self.positions_and_density(radiance_field, indices, jitter, device, all_occupied=True)
        rgb = radiance_field.query_rgb_mean(raw, self._viewdirs_on(device))
        alpha = torch.empty(n, dtype=torch.float32, device=device)
        keep = torch.empty(n, dtype=torch.uint8, device=device)
        L.check(lib.dreg_ngp_alpha_keep(L.ptr(density), L.ptr(alpha), L.ptr(keep), n, float(self._delta), float(density_thre), L.stream()), "dreg_ngp_alpha_keep")
        return world, rgb, alpha[:, None], indices, keep.view(torch.bool)

    @torch.no_grad()
    def _query_dense_fused(self, radiance_field: NGPradianceField, dev, density_thre: float, jitter: Optional[torch.Tensor]):
        world, indices, raw, alpha, keep, rows = self._cells_and_density_fused(radiance_field, dev, density_thre, jitter)
        rgb = radiance_field.query_rgb_mean(raw, self._viewdirs_on(dev))
        mask = keep.view(torch.bool)
        mask._dreg_rows = rows + (indices, keep)      # for build_voxel_grid (same world / rgb / alpha / indices / mask only)
        return world, rgb, alpha[:, None], indices, mask

    @torch.no_grad()
    def query_dense_async(self, radiance_field: NGPradianceField, dev, density_thre: float = 0.7, jitter: Optional[torch.Tensor] = None, n_known: int = None):
        """query_dense + build_voxel_grid for a caller that knows the number of occupied cells (counted on the host when the occupancy grid was
        loaded): NO host readback anywhere.  Returns (world, rgb, alpha [N], indices, keep uint8 [N], voxel_grid [r,r,r,7], voxel_mask buffer [N],
        count int32 [1] on the device = valid length of the mask, rows) — `rows` lets write_kept_async write further grids over other masks of
        the same cells.  The form the evaluation pipeline and bench.py --ngp run (dreg_nerf_amd/eval_pipeline.py)."""
        world, indices, raw, alpha, keep, rows = self._cells_and_density_fused(radiance_field, torch.device(dev), density_thre, jitter, n_known=n_known)
        rgb = radiance_field.query_rgb_mean(raw, self._viewdirs_on(dev))
        grid, mask, count = write_kept_async(rows, world, rgb, alpha, indices, keep, int(rows[2][0]), grid=rows[3])
        return world, rgb, alpha, indices, keep, grid, mask, count, rows

    @torch.no_grad()
    def _cells_and_density_fused(self, radiance_field: NGPradianceField, dev, density_thre: float, jitter: Optional[torch.Tensor],
                                 n_known: Optional[int] = None):
        """The first half of the fused dense query: occupied cells (ascending), their jittered positions, density / raw features / alpha /
        density mask — five launches and the one host readback (the number of occupied cells).  n_known: that number when the caller has it
        already (eval_pipeline counts the checkpoint's occupancy grid on the loader thread, on the host): no readback at all."""
        lib = L.load()
        import ctypes
        binary = self._binary.to(dev)
        b8 = binary.contiguous().view(torch.uint8) if binary.dtype == torch.bool else (binary != 0).to(torch.uint8).contiguous()
        rx, ry, rz = (int(v) for v in b8.shape)
        hc = self.__dict__.get("_host_consts")
        key = (self.resolution.data_ptr(), self.resolution._version, self._roi_aabb.data_ptr(), self._roi_aabb._version)
        if hc is None or hc[0] != key:
            hc = self.__dict__["_host_consts"] = (key, [int(v) for v in self.resolution.tolist()], [float(v) for v in self._roi_aabb.tolist()])
        aabb = (ctypes.c_float * 6)(*hc[2])
        nb = int(lib.dreg_grid_occupied_workspace_bytes(rx, ry, rz))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        L.check(lib.dreg_grid_occupied_count(L.ptr(b8), L.ptr(ws), nb, rx, ry, rz, L.stream()), "dreg_grid_occupied_count")
        toff = int(lib.dreg_grid_occupied_totals(L.ptr(ws), rx, ry, rz)) - ws.data_ptr()
        totals = ws[toff:toff + 8].view(torch.int32)
        # the zeroed voxel grid build_voxel_grid will fill (cubes): issued here so that its fill runs while the host waits for N
        grid = torch.zeros(rx * ry * rz, 7, dtype=torch.float32, device=dev) if rx == ry == rz else None
        n = int(totals[0].item()) if n_known is None else int(n_known)   # the query's one host readback: the outputs' size
        if jitter is None:
            jitter = torch.rand(n, 3, dtype=torch.float32, device=dev)
        jitter = jitter.to(dev).float().contiguous()
        if jitter.shape[0] != n:
            raise ValueError(f"jitter has {jitter.shape[0]} rows, the field {n} occupied cells")
        indices = torch.empty(n, dtype=torch.int64, device=dev)
        order = torch.empty(n, dtype=torch.int32, device=dev)
        world = torch.empty(n, 3, dtype=torch.float32, device=dev)
        world_slot = torch.empty(n, 3, dtype=torch.float32, device=dev)
        L.check(lib.dreg_grid_occupied_build(L.ptr(b8), L.ptr(ws), L.ptr(jitter), aabb, L.ptr(indices), L.ptr(order), L.ptr(world), L.ptr(world_slot),
                                             rx, ry, rz, n, L.stream()), "dreg_grid_occupied_build")
        base16, _ = radiance_field._prepared()
        density = torch.empty(n, dtype=torch.float32, device=dev)
        raw = torch.empty(n, 16, dtype=torch.float16, device=dev)
        alpha = torch.empty(n, dtype=torch.float32, device=dev)
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        faabb = (ctypes.c_float * 6)(*radiance_field._aabb_host())
        nws = int(lib.dreg_ngp_density_workspace_bytes(n)) if n >= 16384 else 0
        dws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws else None
        L.check(lib.dreg_ngp_density_keep_fwd_ws(L.ptr(world_slot), base16.data_ptr() + 3072 * 2, base16.data_ptr(), base16.data_ptr() + 2048 * 2,
                                                 L.ptr(density), L.ptr(raw), *radiance_field._levels, faabb, n, int(bool(radiance_field.unbounded)),
                                                 L.ptr(dws), nws, L.ptr(order), 1, L.ptr(alpha), L.ptr(keep), float(self._delta), float(density_thre),
                                                 L.stream()), "dreg_ngp_density_keep_fwd_ws")
        return world, indices, raw, alpha, keep, (ws, totals, (rx, ry, rz), grid)

    @torch.no_grad()
    def positions_and_density(self, radiance_field: NGPradianceField, indices: torch.Tensor, jitter: torch.Tensor, device,
                              all_occupied: bool = False):
        """World positions of the jittered samples of the cells `indices` (ascending flat indices) and the field's density / raw features
        there: (world [n,3], density [n], raw fp16 [n,16]), all indexed like `indices`.
        all_occupied=True asserts that `indices` is EXACTLY uniform_sample_occupied_voxels() of the current binary field (every occupied
        cell, ascending): only then may the x-ordered query be used — dreg_grid_x_order derives each cell's slot from the occupancy
        volume and would write outside `order` (or leave holes in it) for any other index set.  Subsets, chunks or indices taken
        before set_binary_fields() changed the field take the unordered query (same results, slower)."""
        lib = L.load()
        import ctypes
        n = indices.shape[0]
        world = torch.empty(n, 3, dtype=torch.float32, device=device)
        hc = self.__dict__.get("_host_consts")      # resolution / aabb on the host, cached: .tolist() of device buffers is a sync each
        key = (self.resolution.data_ptr(), self.resolution._version, self._roi_aabb.data_ptr(), self._roi_aabb._version)
        if hc is None or hc[0] != key:
            hc = self.__dict__["_host_consts"] = (key, [int(v) for v in self.resolution.tolist()], [float(v) for v in self._roi_aabb.tolist()])
        rx, ry, rz = hc[1]
        aabb = (ctypes.c_float * 6)(*hc[2])
        # the cells come z-fastest, the hash grid's tables are x-fastest: an order in which a wave's 64 lanes run along x (csrc/ngp.hip)
        order = None
        if all_occupied and n >= 16384 and rx <= 65535:
            binary = self._binary.to(device)
            binary = binary.contiguous().view(torch.uint8) if binary.dtype == torch.bool else binary.to(torch.uint8).contiguous()
            nb = int(lib.dreg_grid_x_order_workspace_bytes(rx, ry, rz))
            ows = torch.empty(nb, dtype=torch.uint8, device=device)
            order = torch.empty(n, dtype=torch.int32, device=device)
            L.check(lib.dreg_grid_x_order(L.ptr(binary), L.ptr(indices), L.ptr(order), L.ptr(ows), nb, rx, ry, rz, n, L.stream()), "dreg_grid_x_order")
            world_slot = torch.empty(n, 3, dtype=torch.float32, device=device)
            L.check(lib.dreg_grid_sample_points_ordered(L.ptr(indices), L.ptr(jitter), L.ptr(order), L.ptr(world), L.ptr(world_slot), rx, ry, rz, aabb, n,
                                                        L.stream()), "dreg_grid_sample_points_ordered")
            density, raw = radiance_field.query_raw(world_slot, order=order, x_in_slot_order=True)
        else:
            L.check(lib.dreg_grid_sample_points(L.ptr(indices), L.ptr(jitter), L.ptr(world), rx, ry, rz, aabb, n, L.stream()), "dreg_grid_sample_points")
            density, raw = radiance_field.query_raw(world)
        return world, density, raw

    @torch.no_grad()
    def query_radiance_and_density_from_camera(self, radiance_field, occupancy_grid, meta_data, device,
                                               density_thre: float = 0.7, cut_off: float = 0.5, jitter=None):
        """Reference 6-tuple (sample_grid.py:343): (world, rgb, alpha, indices, density_mask, surface_mask).
        surface_mask: surface field >= cut_off from at least one training camera (sample_grid.py:244-318) through the fused
        ray-march kernel (visibility.hip) when meta_data carries 'camera_poses' and 'render_step_size'; otherwise all True."""
        world, rgb, alpha, indices, density_mask = self.query_dense(radiance_field, device, density_thre, jitter)
        if meta_data and "camera_poses" in meta_data and "render_step_size" in meta_data:
            from . import visibility
            cams = torch.as_tensor(meta_data["camera_poses"])[..., :3, 3].to(device)
            surface_mask = visibility.surface_visibility(world, cams, radiance_field, self._binary.to(device), self._roi_aabb,
                                                         meta_data.get("aabb", self._roi_aabb), meta_data["render_step_size"],
                                                         cut_off, 1e-4, float(meta_data.get("alpha_thre", 0.0)))
        else:
            surface_mask = torch.ones_like(density_mask)
        return world, rgb, alpha, indices, density_mask, surface_mask


@torch.no_grad()
def build_voxel_grid(world, rgb, alpha, indices, keep, resolution: int):
    """eval_ngp_nerf.py:383-405: zeros [res^3, 7] with (xyz, rgb, alpha) written at the kept indices; returns
    (voxel_grid fp32 [res,res,res,7], voxel_mask int64 ascending)."""
    lib = L.load()
    dev = world.device
    rows = getattr(keep, "_dreg_rows", None)
    if rows is not None and rows[4] is indices and rows[2] == (resolution,) * 3 and world.is_contiguous() and rgb.is_contiguous():
        # straight from SampleGrid.query_dense: kept points per (x, y) row, one scan, one launch that writes voxel_mask (ascending) and
        # voxel_grid; one readback gives the mask's length.  (The zeroed grid was issued by the query, once: a second call allocates its own.)
        ws, totals, (rx, ry, rz), grid, _, keep_u8 = rows
        keep._dreg_rows = (ws, totals, (rx, ry, rz), None, indices, keep_u8)
        if grid is None:
            grid = torch.zeros(resolution ** 3, 7, dtype=torch.float32, device=dev)
        n = world.shape[0]
        mask = torch.empty(n, dtype=torch.int64, device=dev)
        L.check(lib.dreg_grid_write_kept(L.ptr(ws), L.ptr(world), L.ptr(rgb), L.ptr(alpha.reshape(-1).contiguous()), L.ptr(indices), L.ptr(keep_u8),
                                         L.ptr(mask), L.ptr(grid), rx, ry, rz, n, L.stream()), "dreg_grid_write_kept")
        return grid.view(resolution, resolution, resolution, 7), mask[:int(totals[1].item())]
    grid = torch.zeros(resolution ** 3, 7, dtype=torch.float32, device=dev)
    keep_u8 = keep.to(torch.uint8).contiguous()
    L.check(lib.dreg_grid_scatter7(L.ptr(world.contiguous()), L.ptr(rgb.contiguous()), L.ptr(alpha.reshape(-1).contiguous()),
                                   L.ptr(indices.contiguous()), L.ptr(keep_u8), L.ptr(grid), world.shape[0], L.stream()), "dreg_grid_scatter7")
    return grid.view(resolution, resolution, resolution, 7), indices[keep]


@torch.no_grad()
def write_kept_async(rows, world, rgb, alpha, indices, keep_u8, resolution: int, grid: Optional[torch.Tensor] = None):
    """build_voxel_grid without its host readback, for any keep mask over the cells of ONE dense query (rows = the `_dreg_rows` / fourth result of
    SampleGrid._cells_and_density_fused): returns (voxel_grid fp32 [res,res,res,7], voxel_mask int64 buffer of capacity N whose first `count`
    entries are the kept indices, ascending, count int32 [1] on the device).  May be called several times on the same query (density mask, then
    density AND surface mask: eval_ngp_nerf.py:350-412): every call recounts the kept cells per row from `keep_u8`."""
    lib = L.load()
    ws, totals, (rx, ry, rz) = rows[0], rows[1], rows[2]
    assert (rx, ry, rz) == (resolution,) * 3
    dev = world.device
    if grid is None:
        grid = torch.zeros(resolution ** 3, 7, dtype=torch.float32, device=dev)
    n = world.shape[0]
    mask = torch.empty(n, dtype=torch.int64, device=dev)
    L.check(lib.dreg_grid_write_kept(L.ptr(ws), L.ptr(world), L.ptr(rgb), L.ptr(alpha.reshape(-1).contiguous()), L.ptr(indices), L.ptr(keep_u8),
                                     L.ptr(mask), L.ptr(grid), rx, ry, rz, n, L.stream()), "dreg_grid_write_kept")
    return grid.view(resolution, resolution, resolution, 7), mask, totals[1:2].clone()


def save_voxel_grid(out_dir: str, voxel_grid: torch.Tensor, voxel_mask: torch.Tensor, prefix: str = "voxel", points=None, colors=None):
    """<prefix>_grid.pt / <prefix>_mask.pt and, when the kept samples are passed, <prefix>_point_cloud.ply (positions + colours of
    the kept samples, the layout open3d writes: eval_ngp_nerf.py:357-362,388-394).  prefix 'voxel' = surface AND density mask (what
    the registration dataset reads), 'density_voxel' = density mask only (:350-381)."""
    os.makedirs(out_dir, exist_ok=True)
    torch.save(voxel_grid.cpu(), os.path.join(out_dir, f"{prefix}_grid.pt"))
    torch.save(voxel_mask.cpu(), os.path.join(out_dir, f"{prefix}_mask.pt"))
    if points is not None:
        from .vis_dump import write_ply
        write_ply(os.path.join(out_dir, f"{prefix}_point_cloud.ply"), points.float().cpu(), None if colors is None else colors.float().cpu())

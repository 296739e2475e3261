"""Host-side mirror of the reference registration network for MI355X.

``NeRFRegTr`` keeps the reference's constructor, ``forward(data) -> dict`` contract and the 772-key
``state_dict`` (conerf/register/nerf_regtr.py:72-248) while running every hot op through the HIP
kernels in ``csrc/`` (NDHWC, bf16 MFMA by default, exact-f32 MFMA in ``precision='fp32'``).
Differences by design: several pairs may be batched in one call (``forward_batch``); BatchNorm keeps
the reference's one-grid-per-call statistics by computing them per grid inside the batch.
"""
import math
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as L
from . import ops, params
from . import transformer_ops as T
from . import attn_ops as A


class _GlobalPlan(list):
    """[rounds]: ONE subsample plan over the rows of every pair of the step (NeRFRegTr.global_subsample); pts_all = the subsampled points of all pairs."""
    pts_all = None


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _build_tree(root: nn.Module, spec, init_from=None):
    made = {}
    for key, (shape, kind) in spec.items():
        if key.startswith(params.ALIAS_DST) or key.startswith(params.POS_EMBED_ALIAS):
            continue
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                setattr(mod, p, _Node())
            mod = getattr(mod, p)
        t = torch.zeros(shape, dtype=torch.int64 if kind == "bn_count" else torch.float32)
        if params.is_buffer(kind):
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t))
        made[key] = kind
    # alias: feature_pyramid.resnet is the same module object as backbone_net (feature_pyramid_net.py:194-200)
    root.fpn3d.feature_pyramid.resnet = root.fpn3d.backbone_net
    # module registration order must give the reference's state_dict order: move 'resnet' first in feature_pyramid
    fp = root.fpn3d.feature_pyramid
    mods = fp._modules
    reordered = {"resnet": mods["resnet"]}
    reordered.update({k: v for k, v in mods.items() if k != "resnet"})
    fp._modules = reordered
    if hasattr(root, "pos_embed"):   # learned embedding: the decoder's first member is the same module object (nerf_regtr.py:110,265)
        cd = root.correspondence_decoder
        cd._modules = {"pos_embed": root.pos_embed, **cd._modules}


def _reset_parameters(model: nn.Module, spec):
    """Reference initialisation distributions: Xavier-normal convs / zero FPN bias / BN (1,0)
    (resnet3d.py:133-138, feature_pyramid_net.py:10-18); nn.MultiheadAttention / nn.Linear / LayerNorm
    defaults for the transformer, all six encoder layers identical (deepcopy, transformer.py:18-19)."""
    sd = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    with torch.no_grad():
        for key, (shape, kind) in spec.items():
            if key.startswith(params.ALIAS_DST) or key.startswith(params.POS_EMBED_ALIAS):
                continue
            if kind == "conv":
                nn.init.xavier_normal_(sd[key])
            elif kind in ("bn_weight", "ln_weight"):
                sd[key].fill_(1.0)
            elif kind == "bn_var":
                bufs[key].fill_(1.0)
            elif kind == "linear":
                if key.endswith("in_proj_weight"):
                    nn.init.xavier_uniform_(sd[key])
                else:
                    nn.init.kaiming_uniform_(sd[key], a=math.sqrt(5))
            elif kind == "bias":
                wkey = key[:-4] + "weight"
                if key.startswith("fpn3d") or key.endswith("in_proj_bias") or "out_proj" in key or "norm" in key:
                    sd[key].zero_()
                elif wkey in sd and sd[wkey].dim() == 2:
                    bound = 1.0 / math.sqrt(sd[wkey].shape[1])
                    sd[key].uniform_(-bound, bound)
        for key in list(sd.keys()):
            if key.startswith("transformer_encoder.layers.0."):
                for l in range(1, 6):
                    sd[key.replace("layers.0.", f"layers.{l}.")].copy_(sd[key])


class NeRFRegTr(nn.Module):
    def __init__(self, pos_emb_type: str = "sine", pos_emb_dim: int = 256, pos_emb_scaling: float = 1.0,
                 num_downsample: int = 6, precision: str = "bf16"):
        super().__init__()
        if pos_emb_dim != 256:
            raise NotImplementedError("the attention / LayerNorm kernels are built for the reference's model width 256")
        # 'sine' (default) or anything else = the learned MLP, as the reference's constructor decides (nerf_regtr.py:87-90)
        self.pos_emb_type = "sine" if pos_emb_type == "sine" else "learned"
        self.num_downsample = num_downsample
        self.pos_emb_scaling = pos_emb_scaling
        self.precision = precision
        # Evaluate the two FPN head convolutions only where their outputs are consumed (around the occupied voxels):
        # identical results, a fraction of the FLOPs.  False = dense 64^3 evaluation (the BASELINE.md FLOP accounting).
        self.active_set = True
        # Issue the FPN3D forward / backward from the C++ executor (csrc/executor.hip) instead of one autograd node per layer
        self.native_trunk = True
        # Run the data-dependent geometry phase of forward_batch on its own high-priority stream (see forward_batch)
        self.async_geometry = True
        self.skip_empty_stem_rows = True   # stem output rows whose receptive field is all zero are written as zeros, not computed
        # The active-set 3^3 head convolutions (forward and data gradient) on csrc/conv_brick.hip: the union of a 256-row tile's
        # neighbourhoods staged once instead of a 27-tap gather per row (same results up to the order of the fp32 sums)
        self.brick_head = True
        # row sets that get tile tables (positions in ops.active_sets' tuple): S3 = the rows of the 256 -> 64 data gradient of
        # pyramid_transformation_1, the one launch the staged form wins clearly (309 vs 455 us); the executor's `brick` creation option
        # (dreg_exec_opts) decides which launches use tables that exist (tools/bench_conv_brick.py measures all six)
        self.brick_sets = (2,)
        # Issue the point-set half (encoder, decoder, heads) of forward_batch from the C++ executor (csrc/pointset_exec.hip)
        self.native_pointset = True
        # The stem (5^3 / stride 2 over the 8 x 128^3 input) only on the output voxels whose receptive field holds an occupied voxel (~5 %
        # of them for the benchmark's shells; everything else is exactly zero).  Contract: a grid is zero outside its voxel_mask — what
        # eval_ngp_nerf.py:397-405 writes and dataset.py:277-331 keeps (augmentations touch mask voxels only).  False: the dense stem
        # with per-row occupancy flags taken from the VALUES.
        self.stem_rows = True
        self.batched_subsample = True   # the pairs' voxel-average rounds as one autograd node (attn_ops.subsample_all)
        # ... and PLANNED for all pairs at once: one set of launches and one host sync per round instead of one per pair and round (round 6;
        # transformer_ops.plan_hierarchical_subsample_all: pairs that have stopped subsampling pass through the later rounds unchanged)
        self.global_subsample = True
        self.fused_gather_subsample = True   # ... together with the trilinear gather in front of them (attn_ops.gather_subsample)
        self._spec = params.regtr_spec(self.pos_emb_type)
        _build_tree(self, self._spec)
        _reset_parameters(self, self._spec)

    # ------------------------------------------------------------------ helpers
    @property
    def act_dtype(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    def _P(self) -> Dict[str, torch.Tensor]:
        d = self.__dict__.get("_pdict")
        if d is None:   # cached: walking the 772-entry module tree costs more than a millisecond of host time per call
            d = dict(self.named_parameters(remove_duplicate=False))
            d.update(dict(self.named_buffers(remove_duplicate=False)))
            self.__dict__["_pdict"] = d
        return d

    def _apply(self, fn, *a, **kw):  # .to() / .float() / ... may replace parameter objects
        self.__dict__["_pdict"] = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.__dict__["_pdict"] = None
        out = super().load_state_dict(*a, **kw)
        ops.bump_weight_generation()   # in-place copies keep data_ptr(): cached bf16 weight packs (ops, trunk executor) are stale now
        return out

    # ------------------------------------------------------------------ A5: position embedding of the point coordinates
    def position_embedding(self, xyz: torch.Tensor) -> torch.Tensor:
        """fp32 [R,3] -> fp32 [R,256] (nerf_regtr.py:170): the sine embedding with the configured coordinate scale
        (position_embedding.py:24-53), or the learned MLP 3-32-64-128-256-256 (position_embedding.py:56-76; five small plain GEMMs:
        rocBLAS through torch, its gradient arrives through the LayerNorm / decoder additions)."""
        if self.pos_emb_type == "sine":
            return A.posenc_sine(xyz, 256, 1000.0, self.pos_emb_scaling)
        P = self._P()
        h = xyz.float()
        for i in range(5):
            h = F.linear(h, P[f"pos_embed.mlp.{2 * i}.weight"], P[f"pos_embed.mlp.{2 * i}.bias"])
            if i < 4:
                h = torch.relu(h)
        return h

    # ------------------------------------------------------------------ A1/A2: FPN3D over a batch of grids
    def _fpn_program(self, O, x, rows, nbt, train: bool = True):
        """The feature network written against an op provider O: dreg_nerf_amd.ops (eager, one autograd node per layer) or
        trunk_exec._Recorder (records the op program of the native executor).  rows = (S1, S2, S3[, map1]) or None."""
        P = self._P()
        r = "fpn3d.backbone_net."

        def bn(t, name, res=None, relu=True):
            if train:
                nbt.append(P[name + ".num_batches_tracked"])
            return O.batchnorm(t, P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"],
                               P[name + ".running_var"], res=res, relu=relu, train=train)

        # the stem on the output rows whose receptive field holds an occupied voxel (ops.RowSets.stem: a tensor for the eager ops, a row-list
        # id for the recorder), dense otherwise
        stem = getattr(rows, "stem", None) if rows is not None else None
        c1 = bn(O.conv3d(x, P[r + "conv1.weight"], stride=2, pad=2, **({"out_rows": stem} if stem is not None else {})), r + "bn1")
        h = O.maxpool3d(c1)
        feats = [c1]
        for li, nblk in enumerate(params.RESNET50_BLOCKS):
            for b in range(nblk):
                p = f"{r}layer{li + 1}.{b}"
                stride = 2 if (b == 0 and li > 0) else 1
                o = bn(O.conv3d(h, P[p + ".conv1.weight"]), p + ".bn1")
                o = bn(O.conv3d(o, P[p + ".conv2.weight"], stride=stride, pad=1), p + ".bn2")
                o = O.conv3d(o, P[p + ".conv3.weight"])
                if (p + ".downsample.0.weight") in P:
                    res = bn(O.conv3d(h, P[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1", relu=False)
                else:
                    res = h
                h = bn(o, p + ".bn3", res=res, relu=True)
            feats.append(h)
        c1, c2, c3, c4, c5 = feats
        q = "fpn3d.feature_pyramid."

        def conv(name, t, pad, addend=None):
            return O.conv3d(t, P[q + name + ".weight"], P[q + name + ".bias"], addend=addend, pad=pad)

        p5 = conv("pyramid_transformation_5", c5, 0)
        p4 = conv("upsample_transform_4", conv("pyramid_transformation_4", c4, 0, addend=p5), 1)
        p3 = conv("upsample_transform_3", conv("pyramid_transformation_3", c3, 0, addend=p4), 1)
        if rows is not None and len(rows) >= 6:
            # P2 is only consumed (by the nearest-x2 upsample-add of the head's lateral sum on S2) at A = parents(S2); its own
            # lateral sum only on A2 = A dilated by 3^3: both level-2 convolutions run on row lists as well
            a1, a2 = rows[4], rows[5]
            lat2 = O.conv3d_rows(c2, P[q + "pyramid_transformation_2.weight"], P[q + "pyramid_transformation_2.bias"], p3, 0, a2, a2)
            p2 = O.conv3d_rows(lat2, P[q + "upsample_transform_2.weight"], P[q + "upsample_transform_2.bias"], None, 1, a1, a2)
        else:
            p2 = conv("upsample_transform_2", conv("pyramid_transformation_2", c2, 0, addend=p3), 1)
        if rows is None:
            p1 = conv("upsample_transform_1", conv("pyramid_transformation_1", c1, 1, addend=p2), 1)
        else:
            s1, s2, s3 = rows[:3]
            lat1 = O.conv3d_rows(c1, P[q + "pyramid_transformation_1.weight"], P[q + "pyramid_transformation_1.bias"], p2, 1, s2, s3)
            p1 = O.conv3d_rows(lat1, P[q + "upsample_transform_1.weight"], P[q + "upsample_transform_1.bias"], None, 1, s1, s2)
        return p1

    def _trunk_executor(self, x, rows):
        """The native executor for this (batch, resolution, head mode, grad mode), or None when it does not apply: fp32 parity
        mode, or training without preallocated gradient buffers (the executor accumulates straight into FlatAdamW's)."""
        if not self.native_trunk or self.precision != "bf16":
            return None
        from . import trunk_exec
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.fpn3d.parameters())
        if with_grad and any(p.requires_grad and (p.grad is None or not p.grad.is_contiguous()) for p in self.fpn3d.parameters()):
            return None
        sparse_level = 0 if rows is None else (2 if len(rows) >= 6 else 1)
        stem = rows is not None and getattr(rows, "stem", None) is not None
        key = (tuple(x.shape), sparse_level, with_grad, stem)
        cache = self.__dict__.setdefault("_trunk_cache", {})
        ex = cache.get(key)
        if ex is not None and not ex.still_valid():
            ex = None
        if ex is None:
            cache.clear()   # one live program: its arena holds every activation of the network
            ex = trunk_exec.TrunkExecutor(self, tuple(x.shape), sparse_level, with_grad, stem_rows=stem)
            cache[key] = ex
        return ex

    def fpn(self, x: torch.Tensor, rows=None, row_occ=None) -> torch.Tensor:
        """x: [B, D, H, W, 8] (rgba + 4 zero channels), activation dtype.  Returns P1 [B, D/2, H/2, W/2, 256].
        rows = (S1, S2, S3) from ops.active_sets: the two head convolutions are evaluated on the active set only
        (P1 is then defined on S1, which is all the trilinear gather reads).  row_occ (uint8 [B, D/2, H/2], from the packers
        with occupancy=True): output rows of the stem whose receptive field in x is all zero; the native executor skips them
        (their result is exactly zero) — same output bit for bit."""
        train = self.training
        ex = self._trunk_executor(x, rows)
        if ex is not None:
            from . import trunk_exec
            ex.set_input_row_occupancy(row_occ)
            p1 = trunk_exec.run_trunk(ex, x, self.fpn3d.backbone_net.conv1.weight, rows, train)
            if train and ex.nbt:
                torch._foreach_add_(ex.nbt, x.shape[0])
            return p1
        nbt = []  # num_batches_tracked buffers of the BatchNorms that ran: one fused increment at the end (B calls in the reference)
        p1 = self._fpn_program(ops, x, rows, nbt, train)
        if nbt:
            torch._foreach_add_(nbt, x.shape[0])
        return p1

    def drain_trunk_timings(self, profiler):
        """Move the native executor's HIP-event records (one per convolution launch) into `profiler` and stop timing."""
        for ex in list(self.__dict__.get("_trunk_cache", {}).values()) + list(self.__dict__.get("_ps_cache", {}).values()):
            torch.cuda.synchronize()      # (trunk executors and the point-set executor: csrc/executor.hip, csrc/pointset_exec.hip)
            ex.drain_timings(profiler)
            ex.set_timing(False)

    @staticmethod
    def _grid_table(grids: List[torch.Tensor]):
        """Device pointer table of the (contiguous fp32 [1,7,Z,X,Y]) grids for the staging kernels, or None."""
        if not all(g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape[-4] == 7 for g in grids):
            return None
        return L.to_device_async([g.data_ptr() for g in grids], torch.int64, grids[0].device)

    @staticmethod
    def pack_sparse(vals_cat, idx_cat, pb_cat, n_grids: int, res, dtype, occupancy: bool = False):
        """Sparse blocks -> the same NDHWC [B,Z,X,Y,8] stem input (zero fill + scatter of the occupied voxels' rgba).
        occupancy=True: returns (x, row_occ) with the stem's output-row occupancy flags (see fpn)."""
        Z, X, Y = res
        out = torch.empty(n_grids, Z, X, Y, 8, dtype=dtype, device=vals_cat.device)
        inocc = torch.empty(n_grids, Z, X, dtype=torch.uint8, device=vals_cat.device) if occupancy else None
        L.check(L.load().dreg_pack_rgba_sparse_occ(L.ptr(vals_cat), L.ptr(idx_cat), L.ptr(pb_cat), L.ptr(out), L.ptr(inocc), idx_cat.shape[0], n_grids,
                                                   Z, X, Y, L.dt_of(out), L.stream()), "dreg_pack_rgba_sparse_occ")
        return (out, NeRFRegTr._stem_row_occupancy(inocc)) if occupancy else out

    @staticmethod
    def _stem_row_occupancy(inocc: torch.Tensor) -> torch.Tensor:
        """Input-row flags [B,Z,X] -> output-row flags [B,Z/2,X/2] of the stem (conv1: 5^3, stride 2, pad 2; conerf/model/resnet3d.py:118)."""
        B, Z, X = inocc.shape
        Zo, Xo = (Z + 4 - 5) // 2 + 1, (X + 4 - 5) // 2 + 1
        occ = torch.empty(B, Zo, Xo, dtype=torch.uint8, device=inocc.device)
        L.check(L.load().dreg_conv_row_occupancy(L.ptr(inocc), L.ptr(occ), B, Z, X, Zo, Xo, 5, 2, 2, L.stream()), "dreg_conv_row_occupancy")
        return occ

    @staticmethod
    def pack_grids(grids: List[torch.Tensor], dtype, table=None, occupancy: bool = False):
        """List of [1,7,Z,X,Y] fp32 grids -> NDHWC [B,Z,X,Y,8]: rgba (channels 3:7) + 4 zero pad channels.
        occupancy=True: returns (x, row_occ or None) — flags from the VALUES (a row is empty only if all its rgba are zero)."""
        table = table if table is not None else NeRFRegTr._grid_table(grids)
        if table is not None:
            Z, X, Y = grids[0].shape[-3:]
            out = torch.empty(len(grids), Z, X, Y, 8, dtype=dtype, device=grids[0].device)
            inocc = torch.empty(len(grids), Z, X, dtype=torch.uint8, device=out.device) if occupancy else None
            L.check(L.load().dreg_pack_rgba_grids_occ(L.ptr(table), L.ptr(out), L.ptr(inocc), len(grids), Z, X, Y, L.dt_of(out), L.stream()),
                    "dreg_pack_rgba_grids_occ")
            return (out, NeRFRegTr._stem_row_occupancy(inocc)) if occupancy else out
        rgba = torch.cat([g[:, 3:] for g in grids], dim=0).permute(0, 2, 3, 4, 1)
        out = F.pad(rgba, (0, 4)).to(dtype).contiguous()
        return (out, None) if occupancy else out

    # ------------------------------------------------------------------ A3..A9 for a batch of pairs
    def _geometry(self, batch: List[dict], dev):
        """Everything whose size depends on the data — point coordinates, the active sets of the FPN head, the A4 voxel rounds
        (their stopping rule is per pair) — needs only the occupied voxels' coordinates, not the feature network.  All host
        syncs of a step happen here."""
        grids, idxs, vals = [], [], []
        sparse = all((s + "_sparse") in d for d in batch for s in ("src", "tgt"))
        for i, d in enumerate(batch):
            for j, side in enumerate(("src", "tgt")):
                if sparse:   # dataset.SparseBlock: (idx int64 [N], vals fp32 [N,7], (Z,X,Y)) — the dense grid is zero elsewhere
                    sb = d[side + "_sparse"]
                    idxs.append(sb.idx)
                    vals.append(sb.vals)
                    grids.append(sb.res)
                    continue
                g = d[side + "_xyz_rgba"]
                if g.dim() == 6:
                    g = g.squeeze(0)
                m = d[side + "_mask"]
                if m.dim() == 2:
                    m = m.squeeze(0)
                grids.append(g)
                idxs.append(m)
        res = tuple(grids[0]) if sparse else tuple(grids[0].shape[-3:])
        counts = [int(m.shape[0]) for m in idxs]
        if min(counts) == 0:
            raise ValueError("a block has no occupied voxel (empty voxel_mask): nothing to register")
        idx_cat = torch.cat(idxs).contiguous()
        pb_cat = torch.repeat_interleave(torch.arange(len(idxs), dtype=torch.int32, device=dev), L.to_device_async(counts, torch.int64, dev),
                                         output_size=sum(counts))
        table = None if sparse else self._grid_table(grids)
        if sparse:
            vals_cat = torch.cat(vals).contiguous().float()
            xyz_cat = vals_cat[:, :3].contiguous()
            grids = (vals_cat, len(idxs))     # what pack_grids needs in the sparse form
        elif table is not None:   # one gather launch for all grids
            xyz_cat = torch.empty(idx_cat.shape[0], 3, dtype=torch.float32, device=dev)
            L.check(L.load().dreg_gather_grid_xyz(L.ptr(table), L.ptr(idx_cat), L.ptr(pb_cat), L.ptr(xyz_cat), idx_cat.shape[0], *res, L.stream()),
                    "dreg_gather_grid_xyz")
        else:
            xyz_cat = torch.cat([g[:, :3].permute(0, 3, 4, 2, 1).reshape(-1, 3)[m] for g, m in zip(grids, idxs)])
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        rows = None
        if self.active_set and self.precision == "bf16":
            rows = ops.active_sets(idxs, res, tuple((r + 1) // 2 for r in res), dev, pt_batch=pb_cat, idx_cat=idx_cat,
                                   brick_tiles=self.brick_sets if (self.brick_head and self.native_trunk) else False,
                                   stem=(5, 2, 2) if self.stem_rows else None)
        # the gather backward is evaluated per consumed coarse voxel (S1) in every mode: atomic-free, deterministic
        s1_rows = rows[0] if rows is not None else \
            ops.active_sets(idxs, res, tuple((r + 1) // 2 for r in res), dev, pt_batch=pb_cat, idx_cat=idx_cat, density_cap=1.0, level2=False)[0]
        plans, pts_l, segs = [], [], []
        if self.global_subsample and dev.type == "cuda" and self.batched_subsample:
            # the voxel-average rounds of ALL pairs from one set of launches per round (pairs that have stopped pass through): one plan over the whole row space
            rounds, pts_all, lens = T.plan_hierarchical_subsample_all(xyz_cat, counts, self.num_downsample)
            plans = _GlobalPlan([rounds])
            o = 0
            for i in range(len(batch)):
                n_i = int(lens[2 * i]) + int(lens[2 * i + 1])
                pts_l.append(pts_all[o:o + n_i])
                segs.append((int(lens[2 * i]), int(lens[2 * i + 1])))
                o += n_i
            plans.pts_all = pts_all
        for i in range(len(batch) if not plans else 0):
            ns, nt = idxs[2 * i].shape[0], idxs[2 * i + 1].shape[0]
            rounds, pts, lens = T.plan_hierarchical_subsample(xyz_cat[offs[2 * i]:offs[2 * i + 2]], [ns, nt], self.num_downsample)
            plans.append(rounds)
            pts_l.append(pts)
            segs.append((int(lens[0]), int(lens[1])))
        return grids, idxs, res, idx_cat, pb_cat, rows, plans, pts_l, segs, table, s1_rows

    def geometry_stream(self, dev=None):
        """The high-priority side stream of the geometry phase (created once per device).  The training script hands it to its PrefetchLoader as well: the
        sample uploads / augmentation then share this queue instead of opening a fifth one next to the step's main, weight-gradient, geometry and label
        streams — with five streams in flight the step ran at HALF speed on the collection box (124 vs 230 pairs/s)."""
        dev = dev if dev is not None else self.fpn3d.backbone_net.conv1.weight.device
        side = self.__dict__.get("_geo_stream")
        if side is None or side.device != dev:
            side = self.__dict__["_geo_stream"] = torch.cuda.Stream(device=dev, priority=-1)
        return side

    def forward_batch(self, batch: List[dict]) -> List[dict]:
        """Each element: the reference's ``data`` dict for one pair.  Returns one output dict per pair.
        The geometry phase runs on a high-priority side stream (async_geometry): its host syncs then wait for that stream only,
        so the host can prepare step n+1 while the GPU still works on step n.  The input tensors must be materialised when
        this is called (true for synchronous uploads); otherwise pass the upload's event as data['ready_event']."""
        dev = self.fpn3d.backbone_net.conv1.weight.device
        A.set_precision(self.precision)
        if self.async_geometry and dev.type == "cuda":
            main = torch.cuda.current_stream(dev)
            side = self.geometry_stream(dev)
            for d in batch:
                if d.get("ready_event") is not None:      # produced on a loader stream (dataset.PrefetchLoader): wait on the GPU and tell
                    side.wait_event(d["ready_event"])     # the allocator that these streams read the sample's tensors
                    main.wait_event(d["ready_event"])
                    for v in d.values():
                        for t in ((v.idx, v.vals) if hasattr(v, "vals") else (v,)):
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(side)
                                t.record_stream(main)
            with torch.cuda.stream(side):
                geo = self._geometry(batch, dev)
                tab = A.ProblemTable(geo[8], dev)      # the attention / Kabsch / loss tables: uploaded next to the geometry, not in front of the transformer
            hook = self.__dict__.get("_after_geometry")
            if hook is not None:       # the key points are known here, before the feature network has run: train_step marches their visibility labels now
                hook(geo[7], geo[8], side)
            main.wait_stream(side)
            grids, idxs, res, idx_cat, pb_cat, rows, plans, pts_l, segs, table, s1_rows = geo
            # these were allocated on the side stream and are consumed on the main one
            keep = [idx_cat, pb_cat, s1_rows, tab.buffer] + list(pts_l) + ([rows[0], rows[3]] if rows is not None else []) + ([table] if table is not None else [])
            if isinstance(grids, tuple):
                keep.append(grids[0])
            if rows is not None and len(rows) >= 6:
                keep.append(rows[4])
            if rows is not None and getattr(rows, "tiles", None):
                for bt in rows.tiles.values():
                    bt.record_stream(main)
            if rows is not None and getattr(rows, "stem", None) is not None:
                keep.append(rows.stem)
            if isinstance(plans, _GlobalPlan):
                keep.append(plans.pts_all)
            for rounds in plans:
                for rnd in rounds:
                    keep += [rnd.order, rnd.starts, rnd.n_out_dev, rnd.inv_seg, rnd.inv_cnt]
            for t in keep:
                t.record_stream(main)
        else:
            grids, idxs, res, idx_cat, pb_cat, rows, plans, pts_l, segs, table, s1_rows = self._geometry(batch, dev)
            tab = A.ProblemTable(segs, dev)
            hook = self.__dict__.get("_after_geometry")
            if hook is not None:
                hook(pts_l, segs, None)
        if isinstance(grids, tuple):   # sparse input form
            x_in, row_occ = self.pack_sparse(grids[0], idx_cat, pb_cat, grids[1], res, self.act_dtype, occupancy=True)
        else:
            x_in, row_occ = self.pack_grids(grids, self.act_dtype, table, occupancy=True)
        self._check_stem_contract(rows, row_occ)
        p1 = self.fpn(x_in, rows, row_occ if self.skip_empty_stem_rows else None)
        P = self._P()
        # one split (backward: one concatenation) instead of per-pair slices: autograd turns every slice of the [N_mask_total, 256]
        # feature tensor into a zero-filled full-size gradient plus an add (2.3 GB of traffic per step at 4 pairs)
        sizes = [idxs[2 * i].shape[0] + idxs[2 * i + 1].shape[0] for i in range(len(batch))]
        self.__dict__["_last_plans"] = plans       # (tests: how many rounds the step's plan(s) took)
        if isinstance(plans, _GlobalPlan):      # one plan over all pairs' rows
            sizes = [sum(sizes)]
            xyz_all = plans.pts_all
        else:
            xyz_all = torch.cat(pts_l) if len(pts_l) > 1 else pts_l[0]
        if self.batched_subsample and self.fused_gather_subsample and A.gather_subsample_applies(p1, s1_rows, plans):
            # gather + rounds as one node: its backward never writes the [N_mask_total, 256] gradient of the gathered features
            feats_all = A.gather_subsample(p1, idx_cat, pb_cat, res, s1_rows, plans, sizes)
            feats = None
        else:
            feats = ops.trilinear_gather(p1, idx_cat, pb_cat, res, s1_rows, rows[3] if rows is not None else None)
        if feats is None:
            pass
        elif feats.is_cuda and feats.dtype == torch.float32 and self.batched_subsample:
            feats_all = A.subsample_all(feats, plans, sizes)       # the same launches, results / gradients written in place (no cat)
        else:
            feat_l = [T.apply_subsample_plan(plans[i], f) for i, f in enumerate(feats.split(sizes))]
            feats_all = torch.cat(feat_l) if len(feat_l) > 1 else feat_l[0]
        from . import pointset_exec
        ps = pointset_exec.executor_for(self, P)
        last = None
        if ps is not None:     # the whole point-set half from C++: one call forward, one backward (csrc/pointset_exec.hip)
            cond, corr, ov, *last = pointset_exec.encode_decode(ps, feats_all, xyz_all, self.position_embedding(xyz_all), tab,
                                                                P["transformer_encoder.norm.weight"], with_last=True)
        else:
            cond, corr, ov = T.encode_decode_batched(P, feats_all, xyz_all, tab, self.position_embedding)
        outs = []
        # `pose` is returned DETACHED (requires_grad False), on purpose: the reference's compute_rigid_transform (se3.py:89-140) sits inside autograd, but none of
        # its training losses reads `pose` (train_nerf_regtr.py:186-229 use the correspondences, overlap scores and features), so the weighted Kabsch solve has
        # a forward kernel only.  A pose loss added by a user gets NO gradient through `pose` (tests/test_hip_regtr.py asserts the flag).
        poses = A.weighted_kabsch_pairs(xyz_all, corr, ov, tab)   # [P,6,3,4]: every (pair, layer) solve in one launch
        if not self.training and self.__dict__.get("_stem_violation") is not None:
            # evaluation: a call that was handed a grid with values outside its voxel_mask must not return a plausible pose — the poses of THIS call become
            # NaN on the device (no host sync here); check_inputs() / the next call raise with the explanation
            poses = torch.where(self.__dict__["_stem_violation"], torch.full_like(poses, float("nan")), poses)
        # the batched view of the same results (row space of all pairs) for the fused training losses
        self.last_batched = {"cond": cond, "corr": corr, "ov": ov, "xyz": xyz_all, "tab": tab}
        if last:   # the last layer's outputs as tensors of their own: losses that read only these let the backward pass skip five sixths of the decoder
            self.last_batched.update(cond_last=last[0], corr_last=last[1], ov_last=last[2])
        for pi, (s0, ns, t0, nt) in enumerate(tab.segs):
            s_xyz, t_xyz = xyz_all[s0:s0 + ns], xyz_all[t0:t0 + nt]
            s_c, t_c = cond[:, s0:s0 + ns], cond[:, t0:t0 + nt]
            s_corr, t_corr = corr[:, s0:s0 + ns], corr[:, t0:t0 + nt]
            s_ov, t_ov = ov[:, s0:s0 + ns], ov[:, t0:t0 + nt]
            pose = poses[pi][:, None]
            outs.append({
                "src_feats": [s_c], "tgt_feats": [t_c],
                "src_kp": [s_xyz], "src_kp_warped": [s_corr],
                "tgt_kp": [t_xyz], "tgt_kp_warped": [t_corr],
                "src_overlap": [s_ov], "tgt_overlap": [t_ov],
                "pose": pose,
            })
        return outs

    def _check_stem_contract(self, rows, row_occ):
        """stem_rows rests on "a grid is zero outside its voxel_mask".  The packers' row flags come from the VALUES: an output row of the
        stem that the values mark occupied but no listed voxel lies in means the contract is broken (the stem would silently drop
        input).  Evaluation (model.eval()): checked on EVERY call; the poses of a violating call are returned as NaN (on the device: no host sync, so a
        forward-only loop keeps its pipelining) and `check_inputs()` — eval_nerf_regtr.py calls it where it reads the pose back anyway — or the next call
        raises.  Training: SAMPLED — checked on the first four calls and on every 64th only (the check costs six small launches); the flag of a checked call
        is read by the next call (no host sync on fresh work).  A grid that breaks the contract on an unsampled training call is NOT seen: the row-list
        stem drops its out-of-mask values silently.  Data of unknown provenance should be validated once with model.eval() (every call checked) or run
        with model.stem_rows = False.
        NOTE: on the pack_sparse input path (dataset grids that arrive as (mask, values) lists) values outside the mask never reach the network at
        all — the packer writes listed voxels only — so there is nothing to check there: the contract holds by construction.  Raises ValueError."""
        prev = self.__dict__.pop("_stem_violation", None)
        if prev is not None and bool(prev):             # (the flag of an EARLIER call: its work has long been enqueued)
            raise ValueError("stem_rows: an input grid holds non-zero values outside its voxel_mask (set model.stem_rows = False for such data)")
        stem = getattr(rows, "stem", None) if rows is not None else None
        if stem is None or row_occ is None:
            return
        n = self.__dict__["_stem_calls"] = self.__dict__.get("_stem_calls", 0) + 1
        if self.training and n > 4 and n % 64:
            return
        with torch.no_grad():
            # (flat rows are (b, z, x, y): a W-row is the index without its last coordinate)
            Wo = int(self._stem_out_w(row_occ, rows))
            flags = torch.zeros(row_occ.numel(), dtype=torch.bool, device=row_occ.device)
            flags[torch.div(stem, Wo, rounding_mode="floor").long()] = True
            self.__dict__["_stem_violation"] = (row_occ.reshape(-1) != 0).logical_and_(~flags).any()      # read (and cleared) by the next call / check_inputs()

    def check_inputs(self):
        """Raise if a grid handed to an earlier call broke the stem_rows contract (one host sync; evaluation scripts call it where they read results back)."""
        v = self.__dict__.pop("_stem_violation", None)
        if v is not None and bool(v):
            raise ValueError("stem_rows: an input grid holds non-zero values outside its voxel_mask (set model.stem_rows = False for such data)")

    @staticmethod
    def _stem_out_w(row_occ, rows):
        return rows[3].numel() // row_occ.numel()       # map1 has one entry per output voxel [B, d, h, w]; row_occ one per (b, z, x)

    def forward(self, data: dict) -> dict:
        """Reference contract (nerf_regtr.py:112-248): one pair in, the 9-key dict out."""
        return self.forward_batch([data])[0]

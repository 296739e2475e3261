#!/usr/bin/env python3
"""Benchmark of the DReg-NeRF registration hot path on MI355X.

A "step" is one optimizer step of RegTR training (forward + backward + clip + AdamW) over a batch of
synthetic shell-R NeRF pairs at 128^3 (BASELINE.json configs[1]: batch 4 pairs = 8 grids per GPU, bf16
MFMA with fp32 accumulate / fp32 master weights).  With --gpus N the driver launches N ranks through
torch.distributed.run; every rank takes its own 4 pairs (weak scaling) and gradients are averaged over RCCL.

Prints ONE JSON line on rank 0 (see the contract in the repo brief): value = pairs/s over all ranks,
plus "roofline" for the dominant kernel (HIP-event timed inside the timed region) and "cpu_baseline"
(the oracle = PyTorch-CPU restatement of the reference, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md chip table
HBM_PEAK_GBPS = 8000.0          # HBM3E, same table


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--pairs", type=int, default=4, help="pairs per GPU per step (BASELINE batch = 4)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-res", type=int, default=0, help="resolution of the bounded CPU sample (0: 128 on hosts with >= 32 cores, else 64)")
    ap.add_argument("--event-steps", type=int, default=1, help="timed steps whose conv launches are bracketed by HIP events (roofline line)")
    ap.add_argument("--kernel-report", default="", help="write the per-kernel/per-shape event timing table here")
    ap.add_argument("--no-dense-reference", action="store_true", help="skip the additional dense-head measurement")
    ap.add_argument("--no-ngp-reference", action="store_true", help="skip the additional NGP grid-extraction measurement (BASELINE.json configs[3]) of the default line")
    ap.add_argument("--no-nerf-labels-reference", action="store_true", help="skip the additional measurement of the step with overlap labels from generated NeRF blocks")
    ap.add_argument("--dense-head", action="store_true", help="evaluate the FPN head densely (BASELINE.md FLOP accounting) instead of on the active set")
    ap.add_argument("--cpu-samples", type=int, default=3, help="timed samples of the CPU baseline (after one warm-up)")
    ap.add_argument("--occupancy-sweep", action="store_true", help="also time the step on shells of 1e4 .. 1e5 occupied voxels per side (active-set head)")
    ap.add_argument("--ngp", action="store_true", help="BASELINE.json configs[3] instead: NGP grid extraction of 128^3 NeRF blocks (dense hash-MLP query)")
    ap.add_argument("--eval", action="store_true", help="BASELINE.json configs[4]-style instead: forward-only registration (eval mode, no gradients) of synthetic pairs with a known pose")
    ap.add_argument("--nerf-labels", action="store_true", help="the training step with its overlap labels ray-marched from the pairs' NeRF blocks (train_nerf_regtr.py:186-199) instead of synthetic labels: generated blocks, 50 cameras each")
    ap.add_argument("--chain", action="store_true", help="BASELINE.json configs[4] as a line of its own: extraction + registration of generated scenes, pipelined (and, with --chain-serial, the block-at-a-time chain beside it)")
    ap.add_argument("--chain-serial", action="store_true", help="--chain: also time the serial chain of rounds 3-5 on the same blocks")
    ap.add_argument("--chain-scenes", type=int, default=16, help="--chain: scenes (pairs of blocks) per pass")
    ap.add_argument("--ngp-radius", type=float, default=1.0, help="--ngp: occupied cells = ball of this radius in the [-1.5,1.5]^3 block")
    return ap.parse_args()


def cpu_baseline(res: int, target_res: int, samples: int = 3):
    """Oracle (kind 'port': CPU restatement pinned to the reference by tests/golden) fwd+bwd of ONE pair at `res`,
    train mode, fp32, all host cores: one warm-up, then `samples` timed runs (median reported, all listed).
    Converted to pairs/s at `target_res` by the conv FLOP ratio (res^3) when the sample resolution is smaller."""
    from dreg_nerf_amd import params, synth
    from oracle import regtr_oracle as O
    cores = min(os.cpu_count() or 1, 16)  # more threads than this slow torch's CPU conv3d down at batch 1
    torch.set_num_threads(cores)
    data = synth.shell_pair(res, 1, 2, pose=synth.fixed_pose())
    W = 0.1 * torch.randn(256, 256, generator=torch.Generator().manual_seed(5))

    def one():
        sd = params.synth_state_dict(0)
        for k, (shape, kind) in params.regtr_spec().items():
            if not params.is_buffer(kind) and not k.startswith(params.ALIAS_DST):
                sd[k].requires_grad_(True)
        t0 = time.time()
        pred = O.regtr_forward(sd, data, train=True)
        s_kp, t_kp = pred["src_kp"][0], pred["tgt_kp"][0]
        s_gt, t_gt = synth.synthetic_overlap_gt(s_kp), synth.synthetic_overlap_gt(t_kp)
        with torch.no_grad():
            s_tl = torch.stack([synth.synthetic_overlap_gt(pred["src_kp_warped"][0][l], 1)[0] for l in range(6)])
            t_tl = torch.stack([synth.synthetic_overlap_gt(pred["tgt_kp_warped"][0][l], 1)[0] for l in range(6)])
        losses = O.training_losses(pred, data["pose"], W, s_gt, t_gt, s_tl, t_tl)
        losses["total"].backward()
        return time.time() - t0

    warm = one()
    ts = sorted(one() for _ in range(max(samples, 1)))
    dt = ts[len(ts) // 2]
    scale = (target_res / res) ** 3
    return {
        "value": 1.0 / (dt * scale), "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"oracle (PyTorch-CPU fp32 restatement of the reference) fwd+bwd of 1 shell-R pair at {res}^3: warm-up {warm:.1f}s, then "
                  f"{len(ts)} timed runs, median {dt:.1f}s" + (f", scaled x{scale:.0f} (conv FLOPs ~ res^3) to {target_res}^3" if scale != 1 else ""),
        "measured_seconds": ts,
    }


def kernel_source_sha():
    """sha256 over the HIP sources: PMC traffic numbers are only valid for the kernels they were collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dreg_nerf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(name, head, applies=True):
    """HBM bytes per launch of kernel `name` from the committed rocprofv3 PMC passes (tools/collect_pmc.sh: separate
    FETCH_SIZE / WRITE_SIZE runs of this bench at this workload, --pmc with --kernel-trace only).  Both counters are in KB;
    FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes: MI355X_MICROARCH.md, HBM).
    The file records the sha of the kernel sources it was collected on: a stale file yields no traffic figure."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_per_launch.json")
    if not applies or not os.path.exists(path):
        return None, None
    db = json.load(open(path))
    if db.get("kernel_source_sha") != kernel_source_sha():
        return None, f"profiles/pmc_hbm_per_launch.json is stale (collected on kernel sources {db.get('kernel_source_sha')}, now {kernel_source_sha()}): re-run tools/collect_pmc.sh"
    def norm(k):
        import re
        m = re.match(r"_Z(\d+)", k)               # rocprofv3 leaves kernels with _Float16 arguments mangled: _Z<len><name>...
        if m:
            return k[len(m.group(0)):len(m.group(0)) + int(m.group(1))]
        return k.replace("void ", "").split("(")[0].replace("unsigned short", "bf16").replace("float", "f32").replace(" ", "")
    want = name.replace(" ", "")
    for k, e in db.get(head, {}).items():
        kn = norm(k)
        if (kn == want or kn.startswith(want.rstrip(">") + ",")) and "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            return (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0, \
                f"profiles/pmc_hbm_per_launch.json [{head}] {k.split('(')[0][:80]}: 2 x FETCH_SIZE + WRITE_SIZE, mean of {e['launches_FETCH_SIZE']} launches"
    return None, None


def ngp_cpu_baseline(radius: float, cores: int):
    """oracle/ngp_oracle.dense_query (kind 'port': the published Instant-NGP algorithm restated on the CPU; tiny-cuda-nn itself is not in
    the reference tree) on the bench's own workload — the same ball of occupied cells of a 128^3 block, same generated weights — timed on
    this box's host cores: one warm-up and three timed blocks (~2 s each on 16 threads: a bounded sample of about 10 s of CPU work)."""
    from oracle import ngp_oracle as N
    torch.set_num_threads(cores)
    res = 128
    g = torch.Generator().manual_seed(100)
    base = torch.cat([torch.randn(3072, generator=g) * 0.4, torch.randn(N.n_grid_params(), generator=g)])
    col = torch.randn(7168, generator=g) * 0.2
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    binary = torch.stack([X, Y, Z], -1).norm(dim=-1) < radius
    n = int(binary.sum())
    jitter = torch.rand(n, 3, generator=g)
    aabb = torch.tensor([-1.5] * 3 + [1.5] * 3)
    ts = []
    for i in range(4):          # one warm-up, three timed
        t0 = time.time()
        N.dense_query(binary, jitter, aabb, aabb, base, col)
        ts.append(time.time() - t0)
        if i == 0 and ts[0] > 40.0:     # a slow host: one more block is enough
            ts.append(ts[0])
            break
    warm, ts = ts[0], sorted(ts[1:])
    dt = ts[len(ts) // 2]
    return {"value": 1.0 / dt, "unit": "blocks/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ngp_oracle.dense_query (hash grid + density MLP + 18-direction colour MLP, fp16-emulating torch CPU) on the bench's block: {n} occupied cells "
                      f"of a 128^3 grid, warm-up {warm:.2f}s, {len(ts)} timed blocks, median {dt:.2f}s",
            "measured_seconds": ts}


def eval_bench(args, rank, world, dev):
    """The eval_nerf_regtr.py path as a throughput line: model.eval() (BatchNorm on running statistics), no gradients, `--pairs` pairs per
    call and rank (scenes are independent: replicas, no collective); RRE / RTE of the last call's poses against the pair's known relative
    pose (random-init weights: the errors say nothing about registration quality, only that the metric path runs).  A "step" = one call."""
    from dreg_nerf_amd import synth
    from dreg_nerf_amd.losses import rre_rte
    from dreg_nerf_amd.regtr import NeRFRegTr
    torch.manual_seed(3407)
    model = NeRFRegTr(precision=args.precision).to(dev).eval()
    pose = synth.fixed_pose()
    batch = []
    for i in range(args.pairs):
        s = 1 + 2 * (rank * args.pairs + i)
        d = synth.shell_pair(args.res, s, s + 1, pose=pose)
        batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            preds = model.forward_batch(batch)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            preds = model.forward_batch(batch)
        sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return
    errs = [rre_rte(p["pose"][-1].float().cpu(), b["pose"].reshape(1, 4, 4).float().cpu()) for p, b in zip(preds, batch)]
    print(json.dumps({
        "metric": "nerf_pairs_per_sec_regtr_eval_forward_128", "value": args.pairs * world * args.steps / el, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"RegTR forward only (eval mode, no gradients), shell-R synthetic pairs, {args.res}^3 grids, {args.pairs} pairs per call and GPU, random-init weights",
                   "resolution": args.res, "pairs_per_call": args.pairs, "parallelism": f"replicas x{world}"},
        "rre_deg_mean": float(sum(float(e[0]) for e in errs) / len(errs)), "rte_mean": float(sum(float(e[1]) for e in errs) / len(errs)),
        "note": "RRE / RTE at random initialisation: the metric path (eval_nerf_regtr.py:275-301) runs; no trained checkpoint without network access"}), flush=True)


def write_generated_blocks(n, ncam, seed):
    """n NeRF block checkpoints in the reference's format (train_ngp_nerf.py:187-209) under a fresh temporary directory: a shell-shaped
    128^3 occupancy grid around the synthetic pairs' key points, NGP weights scaled so that surfaces are opaque, ncam cameras on a sphere."""
    import tempfile
    from dreg_nerf_amd import ngp
    aabb, res = [-1.5] * 3 + [1.5] * 3, 128
    g = torch.Generator().manual_seed(seed)
    td = tempfile.mkdtemp(prefix="dreg_nl_")
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    rad = torch.stack([X, Y, Z], -1).norm(dim=-1)
    paths = []
    for b in range(n):
        f = ngp.NGPradianceField(aabb)
        with torch.no_grad():
            f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 3.0
            f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
        occ = ngp.OccupancyGrid(aabb, res)
        occ._binary.copy_((rad > 0.75) & (rad < 0.88))
        poses = torch.eye(4)[None].repeat(ncam, 1, 1)
        poses[:, :3, 3] = torch.nn.functional.normalize(torch.randn(ncam, 3, generator=g), dim=-1) * 3.0
        p = os.path.join(td, f"block_{b}.pth")
        torch.save({"step": 1, "model": f.state_dict(), "occupancy_grid": occ.state_dict(), "aabb": aabb, "unbounded": False, "near_plane": None, "far_plane": None,
                    "grid_resolution": res, "contraction_type": ngp.ContractionType.AABB, "render_step_size": 3 * 3 ** 0.5 / 1024,
                    "alpha_thre": 0.0, "cone_angle": 0.0, "camera_poses": poses, "block_id": b}, p)
        paths.append(p)
    return td, paths


def eval_end_to_end(args, dev, reps: int = 2, scenes_per_pass: int = 16, serial: bool = False, model=None, pose_fn=None):
    """BASELINE.json configs[4] as a throughput figure beside the headline: `scenes_per_pass` scenes of two generated 128^3 NeRF blocks each go through the
    whole evaluation chain — eval_ngp_nerf.py's per-block grid extraction (checkpoint load, dense density + 18-direction colour query, surface labels,
    voxel_grid.pt / voxel_mask.pt / .ply and their density_voxel_* twins written) and eval_nerf_regtr.py's registration (eval-mode forward, RRE / RTE) —
    and the rate is pairs per second of wall time INCLUDING extraction and every file (the pass ends when the last file is closed).
    Default: the pipelined chain (dreg_nerf_amd/eval_pipeline.py: loader threads, no host readbacks, writer threads, registration in batches of
    `--pairs` from the device-resident grids).  serial=True: the block-at-a-time chain of rounds 3-5 (files read back for registration).
    Generated blocks (eight distinct checkpoints, hard-linked into the scene directories) + random-init RegTR weights unless `model` is given: then
    RRE / RTE are those of that checkpoint."""
    import shutil
    import eval_ngp_nerf as E
    from dreg_nerf_amd.losses import rre_rte
    from dreg_nerf_amd.regtr import NeRFRegTr
    n_scenes = args.pairs if serial else scenes_per_pass
    td, paths = write_generated_blocks(8, 6, 11)
    pipe = None
    try:
        # one block checkpoint per directory, as the scripts expect (<scene>/block_k/model.pth); every pass (warm-up + timed) gets directories of its own:
        # an evaluation run writes its grid files once, into directories that hold none yet (re-writing existing files was measured ~2x slower)
        passes = []
        for ps in range(reps + 1):
            blocks = []
            for i in range(2 * n_scenes):
                d = os.path.join(td, f"pass_{ps}", f"scene_{i // 2}", f"block_{i % 2}")
                os.makedirs(d)
                try:
                    os.link(paths[i % len(paths)], os.path.join(d, "model.pth"))
                except OSError:
                    shutil.copyfile(paths[i % len(paths)], os.path.join(d, "model.pth"))
                blocks.append(d)
            passes.append(blocks)
        pass_no = [0]

        def next_blocks():
            b_ = passes[min(pass_no[0], len(passes) - 1)]
            pass_no[0] += 1
            return b_
        if model is None:
            torch.manual_seed(3407)
            model = NeRFRegTr(precision=args.precision).to(dev).eval()
        pose = torch.eye(4)[None]

        def once_serial():
            blocks = next_blocks()
            kept = [E.extract_block(os.path.join(d, "model.pth"), dev) for d in blocks]
            batch = []
            for i in range(n_scenes):
                g = [torch.load(os.path.join(blocks[2 * i + k], "voxel_grid.pt")) for k in range(2)]
                m = [torch.load(os.path.join(blocks[2 * i + k], "voxel_mask.pt")) for k in range(2)]
                batch.append({"src_xyz_rgba": g[0].permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev), "tgt_xyz_rgba": g[1].permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev),
                              "src_mask": m[0].to(dev), "tgt_mask": m[1].to(dev), "pose": pose.clone().to(dev), "src_nerf_path": "", "tgt_nerf_path": ""})
            with torch.no_grad():
                preds = model.forward_batch(batch)
            errs = [rre_rte(p["pose"][-1].float().cpu(), pose) for p in preds]
            return kept, [float(e[0]) for e in errs], [float(e[1]) for e in errs], None

        from dreg_nerf_amd.eval_pipeline import ExtractionPipeline, extract_and_register
        pipe = None if serial else ExtractionPipeline(dev)

        def once_pipelined():
            blocks = next_blocks()
            scenes = [(f"scene_{i}", os.path.join(blocks[2 * i], "model.pth"), os.path.join(blocks[2 * i + 1], "model.pth"), pose[0]) for i in range(n_scenes)]
            for k in pipe.timings:
                pipe.timings[k] = {} if isinstance(pipe.timings[k], dict) else 0 if isinstance(pipe.timings[k], int) else 0.0
            rows, tm = extract_and_register(scenes, model, dev, batch_pairs=args.pairs, pipeline=pipe)
            return [v for r in rows.values() for v in r["voxels"]], [r["R_mean"] for r in rows.values()], [r["t_mean"] for r in rows.values()], tm

        once = once_serial if serial else once_pipelined
        once()                                       # warm-up (first-use allocations, pinned staging, executor recording)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            kept, rre, rte, tm = once()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        if pipe is not None:
            pipe.close()
        out = {"metric": "nerf_pairs_per_sec_extract_plus_register_128", "value": n_scenes / el, "unit": "pairs/s", "s_per_pass": el, "pairs": n_scenes,
               "blocks_extracted": 2 * n_scenes, "voxels_kept_per_block": int(sum(kept) / len(kept)), "chain": "serial" if serial else "pipelined", **({} if serial else {"writers": f"{len(pipe._procs) or pipe._writers._max_workers} {pipe.writer_mode}es" if pipe.writer_mode == "process" else f"{pipe._writers._max_workers} threads"}),
               "rre_deg_mean": float(sum(rre) / len(rre)), "rte_mean": float(sum(rte) / len(rte)),
               "note": "wall time of extraction (checkpoint load, query, surface labels, the six files per block) + registration (eval-mode forward in batches of "
                       f"{args.pairs} pairs, RRE / RTE), last file closed inside the timed region; generated blocks, random-init weights: the errors only show that the metric path runs"}
        rec = os.path.join(ROOT, "profiles", "r06_trained_eval.json")
        if os.path.exists(rec):      # NOT measured by this run: the committed record of the round's trained checkpoint (245 MB, not shipped) through the same chain
            r_ = json.load(open(rec))
            out["trained_checkpoint_record"] = {"source": "profiles/r06_trained_eval.json (tools/trained_regime.sh + tests/test_hip_trained_regime.py on the collection box)",
                                                "held_out_scenes": r_["scenes"], "rre_deg_mean_bf16_chain": r_["bf16_chain"]["rre_deg_mean"], "rre_deg_median_bf16_chain": r_.get("bf16_chain_rre_deg_median"),
                                                "rte_mean_bf16_chain": r_["bf16_chain"]["rte_mean"], "rre_deg_mean_fp32_mode": r_["fp32_mode"]["rre_deg_mean"],
                                                "training_scenes_rre_deg_mean": r_.get("training_scenes_bf16", {}).get("rre_deg_mean"),
                                                "oracle_scene_rre_deg": {"oracle": r_["oracle_scene"]["oracle_rre_deg"], "fp32_mode": r_["oracle_scene"]["fp32_mode_rre_deg"], "bf16_chain": r_["oracle_scene"]["bf16_chain_rre_deg"]}}
        if tm is not None:      # per-phase breakdown of the LAST pass: thread-seconds of work, what the main thread waited for, GPU time by phase
            gpu_ms = tm["gpu_query_ms"] + tm["gpu_surface_ms"] + tm["gpu_grids_ms"] + tm["gpu_copy_ms"] + tm["gpu_register_ms"]
            out["phases"] = {"load_thread_s": tm["load_thread_s"], "write_thread_s": tm["write_thread_s"], "GB_written": tm["bytes_written"] / 1e9,
                             "main_thread": {"waited_for_loads_s": tm["load_wait_s"], "waited_for_staging_slots_s": tm["slot_wait_s"], "enqueue_extraction_s": tm["enqueue_s"], "enqueue_detail_s": tm["host_detail_s"],
                                             "register_s": tm["register_s"], "final_flush_of_writers_s": tm["flush_s"]},
                             "gpu_ms": {"query": tm["gpu_query_ms"], "surface": tm["gpu_surface_ms"], "grid_writers": tm["gpu_grids_ms"], "copies_to_host": tm["gpu_copy_ms"],
                                        "register": tm["gpu_register_ms"]},
                             "gpu_busy_frac_of_pass": gpu_ms * 1e-3 / el}
        return out
    finally:
        if pipe is not None:
            pipe.close()
        shutil.rmtree(td, ignore_errors=True)


def nerf_labels_bench(args, rank, world, dev):
    """The training step as the reference runs it on real data: the overlap ground truth and the 'tilde' scores of every pair are the
    surface-field visibility of its key points / predicted correspondences in the pair's two NeRF blocks (train_nerf_regtr.py:186-199,
    confidence_loss.py:56-160) — here through ONE persistent ray-march launch per step over all blocks (csrc/visibility.hip).  Blocks are
    generated (no checkpoints without network access): a shell-shaped 128^3 occupancy grid around the pairs' key points, NGP weights
    scaled so that surfaces are opaque (rays end within a few samples, as in a trained block), 50 training cameras on a sphere."""
    from dreg_nerf_amd import synth
    from dreg_nerf_amd.regtr import NeRFRegTr
    from dreg_nerf_amd.train_step import TrainStep
    ncam = 50
    td, paths = write_generated_blocks(2 * args.pairs, ncam, rank)
    torch.manual_seed(3407)
    model = NeRFRegTr(precision=args.precision).to(dev).train()
    if world > 1:
        for p_ in model.parameters():
            dist.broadcast(p_.data, 0)
    ts = TrainStep(model)
    pose = synth.fixed_pose()
    batch = []
    for i in range(args.pairs):
        s = 1 + 2 * (rank * args.pairs + i)
        d = synth.shell_pair(args.res, s, s + 1, pose=pose)
        d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
        d["src_nerf_path"], d["tgt_nerf_path"] = paths[2 * i], paths[2 * i + 1]
        batch.append(d)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        ts.step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts.step(batch)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    import shutil
    shutil.rmtree(td, ignore_errors=True)
    if rank != 0:
        return
    kp = int(ts.last_preds[0]["src_kp"][0].shape[0])
    rays = 2 * args.pairs * 7 * kp * ncam
    print(json.dumps({
        "metric": "nerf_pairs_per_sec_regtr_fwd_bwd_128_labels_from_nerf_blocks", "value": args.pairs * world * args.steps / el, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"RegTR fwd+bwd+AdamW with overlap labels ray-marched from the pairs' NeRF blocks: {args.pairs} pairs ({2 * args.pairs} generated 128^3 blocks, "
                               f"{ncam} cameras each) per GPU per step, ~{kp} key points per cloud, 7 point sets per block = {rays / 1e6:.1f} M rays per step in one launch",
                   "global_batch_pairs": args.pairs * world, "resolution": args.res, "parallelism": f"dp{world}"}}), flush=True)


def ngp_measure(args, rank, world, dev):
    """BASELINE.json configs[3]: one 128^3 NeRF block = Np occupied cells -> world samples -> density (hash grid + MLP) -> colour x 18
    directions -> alpha / masks -> voxel_grid + voxel_mask (eval_ngp_nerf.py:336-412 without the surface ray march, which needs the
    block's training cameras).  Every rank extracts its own blocks (replicas only, no collective).  A "step" = one block."""
    from dreg_nerf_amd import ngp
    res = 128
    aabb = [-1.5] * 3 + [1.5] * 3
    g = torch.Generator().manual_seed(100 + rank)
    f = ngp.NGPradianceField(aabb)
    with torch.no_grad():   # generated weights (no checkpoints without network access): tcnn's default-ish scales
        f.mlp_base.params[:3072] = torch.randn(3072, generator=g) * 0.4
        f.mlp_base.params[3072:] = torch.randn(f.mlp_base.params.numel() - 3072, generator=g)
        f.color_mlp.params.copy_(torch.randn(7168, generator=g) * 0.2)
    f = f.to(dev)
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 3 - 1.5
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    binary = (torch.stack([X, Y, Z], -1).norm(dim=-1) < args.ngp_radius).to(dev)
    sg = ngp.SampleGrid(aabb, res).to(dev)
    sg.set_binary_fields(binary)
    npts = int(binary.sum())
    jitter = torch.rand(npts, 3, generator=g).to(dev)
    dirs = sg._viewdirs.to(dev)

    def block():
        # the evaluation pipeline's form of a block's query (dreg_nerf_amd/eval_pipeline.py): the occupied-cell count is known on the host from the
        # checkpoint's occupancy grid, the kept-cell count stays on the device — no host readback per block (rounds 2-5 measured the form with two)
        return sg.query_dense_async(f, dev, jitter=jitter, n_known=npts)[5:8]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        block()
    sync()
    # the compact form beside the headline line times `ngp_windows` windows of `steps` blocks and reports the median window: 20 blocks are 5 ms of wall
    # time, and one host hiccup (a collector pause, a worker of the previous measurement exiting) once turned 3,7xx blocks/s into 756
    els = []
    for _ in range(max(int(getattr(args, "ngp_windows", 1)), 1)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            block()
        sync()
        els.append(time.perf_counter() - t0)
    el = sorted(els)[len(els) // 2]
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return None
    # per-kernel times (events on the launch stream = torch's current stream), same inputs
    world_pts = sg.query_dense(f, dev, jitter=jitter)[0]

    def ev_time(fn, n=5):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    # the product path's first half: cell lists (count, scan, readback of N, build) + encode + MLP with alpha / mask
    ms_d = ev_time(lambda: sg._cells_and_density_fused(f, dev, 0.7, jitter, n_known=npts))
    raw = f.query_raw(world_pts)[1]
    ms_c = ev_time(lambda: f.query_rgb_mean(raw, dirs))
    fl_c = npts * (18 * (64 * 64 + 16 * 64) + 64 * 32) * 2.0        # colour net x 18 directions (geometry half of layer 1 once)
    fl_d = npts * 3072 * 2.0
    gather = npts * 512.0                                             # 16 levels x 8 corners x 2 fp16
    rf = {"bound": "mfma", "kernel": "ngp_rgb_chunks_kernel", "achieved": fl_c / ms_c / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": fl_c / ms_c / 1e9 / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms_c, "algorithmic_flops_per_launch": fl_c,
          "note": "fp16 MFMA (same dense peak as bf16); HIP events on the launch stream, 5 launches",
          "density_kernel": {"kernel": "grid_occupied_count + grid_scan2 + grid_occupied_build + ngp_encode_xcd_kernel + ngp_density_kernel<1,true>", "avg_launch_ms": ms_d, "Gpts_per_s": npts / ms_d / 1e6,
                             "gather_TBps": gather / ms_d / 1e9, "gather_frac_of_hbm_peak": gather / ms_d / 1e9 / (HBM_PEAK_GBPS / 1e3),
                             "mlp_TFLOPs": fl_d / ms_d / 1e9,
                             "note": "avg_launch_ms = the whole first half of a block's query: occupied-cell lists from the occupancy volume (no torch.nonzero), positions, encode, MLP + alpha / mask (five launches + the zero fill of the voxel grid; the occupied-cell count comes from the host side of the checkpoint load: no readback); 512 B gathered per point from the 25.2 MB fp16 table, which lives in the 256 MB Infinity Cache — the encode launch pins two levels to every XCD's 4 MB L2 and runs a wave's lanes along the tables' fastest axis, so the gather rate is a cache rate, not an HBM rate; `traffic` (PMC) is what the encode + MLP launches moved at the memory side"}}
    rf["traffic"], src = pmc_traffic("ngp_rgb_chunks_kernel", "ngp", args.ngp_radius == 1.0)
    if src:
        rf["traffic_source"] = src
    dt_, dsrc = pmc_traffic("ngp_density_kernel", "ngp", args.ngp_radius == 1.0)
    de_, esrc = pmc_traffic("ngp_encode_xcd_kernel", "ngp", args.ngp_radius == 1.0)
    if dt_ and de_:
        dt_, dsrc = dt_ + de_, f"{esrc} + {dsrc}"
    rf["density_kernel"]["traffic"] = dt_
    if dsrc:
        rf["density_kernel"]["traffic_source"] = dsrc
    if dt_:   # what actually reached HBM per launch vs what the lanes gathered: the table is cache resident
        rf["density_kernel"]["hbm_GBps"] = dt_ / ms_d / 1e6
        rf["density_kernel"]["gathered_over_hbm_bytes"] = gather / dt_
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = ngp_cpu_baseline(args.ngp_radius, min(os.cpu_count() or 1, 16))
    return {
        "metric": "ngp_grid_extraction_blocks_per_sec_128", "value": args.steps * world / el, "unit": "blocks/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"NGP occupancy-grid extraction, one 128^3 block per step: {npts} occupied cells (ball r={args.ngp_radius}) -> density + "
                               f"18-direction colour + voxel_grid writer; generated hash-grid / MLP weights", "points_per_block": npts,
                   "parallelism": f"replicas x{world}"},
        "points_per_sec": npts * args.steps * world / el, "roofline": rf, **({"cpu_baseline": cpu} if cpu else {})}


def ngp_bench(args, rank, world, dev):
    out = ngp_measure(args, rank, world, dev)
    if out is not None:
        print(json.dumps(out), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): re-execute this script under torch.distributed.run with one
    rank per GPU (the reference's own multi-GPU form is one process per GPU from the shell, scripts/train/train_nerf_regtr.sh:14-16).
    Never a silent fallback: fewer visible GPUs than N is an error."""
    import socket
    import subprocess
    one_gpu_hook = os.environ.get("DREG_BENCH_ONE_GPU") == "1"      # test hook: N ranks on the one GPU of a test box (gloo)
    have = torch.cuda.device_count()
    if have < args.gpus and not one_gpu_hook:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to run fewer ranks than asked")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI ("nccl" is RCCL on ROCm).  DREG_BENCH_BACKEND=gloo with DREG_BENCH_ONE_GPU=1 is a test hook: several ranks on the
        # one GPU of a test box exercise this script's multi-rank path (tests/test_hip_ddp_gpu.py); never a measurement.
        backend = os.environ.get("DREG_BENCH_BACKEND", "nccl")
        if os.environ.get("DREG_BENCH_ONE_GPU") == "1":
            local_rank = 0
        if backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.chain:
        out = eval_end_to_end(args, dev, scenes_per_pass=args.chain_scenes)
        if args.chain_serial:
            out["serial_chain"] = eval_end_to_end(args, dev, serial=True)
        print(json.dumps(out), flush=True)
        return
    if args.ngp or args.eval or args.nerf_labels:
        (ngp_bench if args.ngp else eval_bench if args.eval else nerf_labels_bench)(args, rank, world, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from dreg_nerf_amd import ops, synth
    from dreg_nerf_amd.regtr import NeRFRegTr
    from dreg_nerf_amd.train_step import TrainStep

    torch.manual_seed(3407)
    model = NeRFRegTr(precision=args.precision).to(dev).train()
    if world > 1:  # identical initial weights on every rank
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    ts = TrainStep(model)
    if world > 1 and backend == "nccl" and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the RCCL group has {dist.get_world_size()} ranks")

    # inputs resident in HBM before the timed region; every rank owns different pairs
    pose = synth.fixed_pose()
    batch = []
    for i in range(args.pairs):
        s = 1 + 2 * (rank * args.pairs + i)
        d = synth.shell_pair(args.res, s, s + 1, pose=pose)
        batch.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(active_set: bool):
        model.active_set = active_set
        for _ in range(args.warmup):
            ts.step(batch)
        sync()
        if ts._sync is not None:
            ts._sync.measure_exposed = True
            ts._sync.exposed_ms()
        ops.PROFILER = ops.KernelTimer()
        # (both native executors bracket their own convolution / linear launches with HIP events while the profiler is enabled)
        t0 = time.perf_counter()
        for i in range(args.steps):
            if i == args.event_steps:      # HIP-event brackets on the first steps of the timed region only (they cost host time)
                ops.PROFILER.enabled = False
            ts.step(batch)
        sync()
        el = time.perf_counter() - t0
        pr, ops.PROFILER = ops.PROFILER, None
        model.drain_trunk_timings(pr)   # the native trunk executor's own HIP-event records of the same region
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, pr

    peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else 157.3

    def label_bytes(label):
        """Compulsory HBM bytes of one conv launch from its shape label: every input voxel, weight and output element once
        (bf16 operands; fp32 for the f32-output variant is ignored: < 1 % of the launches' bytes)."""
        import re
        m = re.match(r"(\S+) B(\d+) (\d+)x(\d+)x(\d+)x(\d+)->(\d+)x(\d+)x(\d+)x(\d+) k(\d+)s(\d+)(?: rows(\d+))?", label)
        if not m:
            return 0.0
        B, di, hi, wi, cin, do, ho, wo, cout, k = (int(m.group(i)) for i in range(2, 12))
        rows = m.group(13)
        n_in = int(rows) if rows else B * di * hi * wi
        n_out = int(rows) if rows else B * do * ho * wo
        return 2.0 * (n_in * cin + n_out * cout + cout * cin * k ** 3)

    def roofline_of(pr, head):
        # HIP-event brackets as measured: nothing is subtracted (an empty bracket reads ~6 us, but with a kernel inside the events add
        # ~2 us to its duration: the round-1 subtraction over-corrected and put `frac` above what profiles/'s rocprofv3 CSV gives)
        by_name, by_label = pr.summary(subtract_overhead=False)
        # the dominant SINGLE kernel: "...+reduce" entries bracket two launches (weight gradient + its split reduce) and have no
        # one rocprofv3 row to be checked against; they stay in the --kernel-report table
        single = {k: v for k, v in by_name.items() if "+" not in k}
        name, (calls, ms, flops) = max(single.items(), key=lambda kv: kv[1][1])
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        nbytes = sum(label_bytes(l) * c for (n, l), (c, _, _) in by_label.items() if n == name)
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        rf = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
              "launches": calls, "avg_launch_ms": ms / max(calls, 1), "algorithmic_flops_per_launch": flops / max(calls, 1),
              "event_steps": min(args.event_steps, args.steps), "empty_bracket_ms_not_subtracted": pr.bracket_overhead_ms(),
              # the same launches against the other roof: compulsory bytes (inputs, weights, outputs once) / time vs 8 TB/s
              "compulsory_bytes_per_launch": nbytes / max(calls, 1), "hbm_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBPS}
        rf["traffic"], src = pmc_traffic(name, head, args.res == 128 and args.pairs == 4 and args.precision == "bf16")
        if src:
            rf["traffic_source"] = src
        if rf["traffic"]:
            rf["traffic_over_compulsory"] = rf["traffic"] / max(rf["compulsory_bytes_per_launch"], 1.0)
        # the label is ONE template instantiation serving several layer shapes, MFMA-bound 3^3 layers and HBM-bound 1^3 ones alike:
        # its launches by shape (largest first), each against the roof it is closer to
        shapes = []
        for (n, l), (c, ms_l, fl_l) in sorted(((k, v) for k, v in by_label.items() if k[0] == name), key=lambda kv: -kv[1][1]):
            if ms_l <= 0:
                continue
            tf, gb = fl_l / (ms_l * 1e-3) / 1e12, label_bytes(l) * c / (ms_l * 1e-3) / 1e9
            shapes.append({"shape": l, "launches": c, "avg_launch_ms": ms_l / max(c, 1), "TFLOPs": tf, "mfma_frac": tf / peak,
                           "compulsory_GBps": gb, "hbm_frac": gb / HBM_PEAK_GBPS, "bound": "mfma" if tf / peak >= gb / HBM_PEAK_GBPS else "hbm"})
        rf["by_shape"] = shapes[:8]
        if rf["hbm_frac"] > rf["frac"]:   # a family of small / 1x1x1 convolutions sits closer to the HBM roof than to the MFMA roof
            rf.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBPS,
                       "mfma_TFLOPs": ach, "mfma_frac": ach / peak})
        return rf, by_label

    # the second, dense-head measurement keeps the BASELINE.md FLOP accounting (8.5 TFLOP per pair) comparable
    dense = None
    if not args.dense_head and args.precision == "bf16" and not args.no_dense_reference:
        el_d, pr_d = timed(False)
        if rank == 0:
            rf_d, _ = roofline_of(pr_d, "dense_head")
            dense = {"value": args.pairs * world * args.steps / el_d, "unit": "pairs/s", "ms_per_step": 1e3 * el_d / args.steps, "roofline": rf_d}
    elapsed, prof = timed(not args.dense_head and args.precision == "bf16")
    # the multi-rank run validates itself: identical parameters and the same clipped gradient norm on every rank after the timed steps, and how long
    # the step stream stood still for gradient exchange that backward did not hide (trivially in sync / 0 ms at one rank)
    from dreg_nerf_amd.optim import ranks_in_sync
    in_sync = ranks_in_sync(ts.optimizer, ts.optimizer.grad_norm())
    exposed = ts._sync.exposed_ms() if ts._sync is not None else (0.0, 0)
    if ts._sync is not None:
        ts._sync.measure_exposed = False
    if world > 1:
        t_ = torch.tensor([exposed[0]], dtype=torch.float64, device=dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        exposed = (float(t_.item()), exposed[1])
    if not in_sync["ranks_in_sync"] and rank == 0:
        # reported in the line (`ranks_in_sync`: false) and loudly here; the throughput of an out-of-sync run must not be read as a DDP measurement
        print(f"bench.py: RANKS OUT OF SYNC after the timed region: {in_sync}", file=sys.stderr, flush=True)

    # the active-set head's work follows the occupied surface: the same step on shells of 1e4 .. 1e5 occupied voxels per side
    sweep = None
    if args.occupancy_sweep and world == 1 and args.precision == "bf16":
        sweep = []
        for r1 in (0.815, 0.83, 0.86, 0.90, 0.96, 1.02, 1.1):
            b2 = []
            for i in range(args.pairs):
                sd_ = 1 + 2 * i
                gs, ms_ = synth.shell_grid(args.res, sd_, 0.8, r1)
                gt, mt_ = synth.shell_grid(args.res, sd_ + 1, 0.8, r1, pose=pose)
                b2.append({"src_xyz_rgba": gs.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev), "tgt_xyz_rgba": gt.permute(3, 2, 0, 1).unsqueeze(0).contiguous().to(dev),
                           "src_mask": ms_.to(dev), "tgt_mask": mt_.to(dev), "pose": pose[None].clone().to(dev), "src_nerf_path": "", "tgt_nerf_path": ""})
            model.active_set = True
            for _ in range(3):
                ts.step(b2)
            torch.cuda.synchronize()
            reps_ = []                 # median of three groups of four steps: one allocator / host hiccup moved a 4-step mean by 15 %
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(4):
                    ts.step(b2)
                torch.cuda.synchronize()
                reps_.append((time.perf_counter() - t0) / 4)
            el_ = sorted(reps_)[1]
            rc = []
            for ex in model.__dict__.get("_trunk_cache", {}).values():
                rc = list(ex.last_row_counts)
            sweep.append({"shell_r1": r1, "n_mask_per_side": int(b2[0]["src_mask"].shape[0]), "pairs_per_s": args.pairs / el_, "ms_per_step": 1e3 * el_,
                          "active_rows": rc, "head": "active-set" if rc else "dense fallback"})
    # the same step with its overlap labels ray-marched from the pairs' NeRF blocks, as on real data (train_nerf_regtr.py:186-199): an extra,
    # clearly separate figure next to the headline (whose labels are synthetic, like every other input)
    with_labels = None
    if world == 1 and args.precision == "bf16" and args.res == 128 and not args.dense_head and not args.no_nerf_labels_reference:
        try:
            import shutil
            td, paths = write_generated_blocks(2 * args.pairs, 50, 0)
            b3 = [dict(d, src_nerf_path=paths[2 * i], tgt_nerf_path=paths[2 * i + 1]) for i, d in enumerate(batch)]
            model.active_set = True
            for _ in range(3):
                ts.step(b3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ts.step(b3)
            torch.cuda.synchronize()
            el_l = (time.perf_counter() - t0) / 10
            kp = int(ts.last_preds[0]["src_kp"][0].shape[0])
            with_labels = {"value": args.pairs / el_l, "unit": "pairs/s", "ms_per_step": 1e3 * el_l, "steps": 10,
                           "rays_per_step": 2 * args.pairs * 7 * kp * 50,
                           "note": f"overlap ground truth and 'tilde' scores of every pair = surface-field visibility of its key points / predicted correspondences in the "
                                   f"pair's two NeRF blocks ({2 * args.pairs} generated 128^3 blocks with opaque surfaces, 50 cameras each), one persistent ray-march launch per step "
                                   f"(csrc/visibility.hip); `python bench.py --nerf-labels` is the same measurement as a line of its own"}
            shutil.rmtree(td, ignore_errors=True)
        except Exception as e:      # never lose the headline line to the extra figure
            with_labels = {"error": repr(e)[:300]}
    if rank == 0:
        roofline, by_label = roofline_of(prof, "dense_head" if args.dense_head else "active_set")
        if args.kernel_report:
            with open(args.kernel_report, "w") as f:
                f.write("kernel\tshape\tcalls\ttotal_ms\tavg_ms\tTFLOP/s\n")
                for (n, l), (c, m, fl) in sorted(by_label.items(), key=lambda kv: -kv[1][1]):
                    f.write(f"{n}\t{l}\t{c}\t{m:.3f}\t{m / c:.4f}\t{fl / (m * 1e-3) / 1e12 if m > 0 else 0:.1f}\n")
        pairs = args.pairs * world * args.steps
        out = {
            "metric": "nerf_pairs_per_sec_regtr_fwd_bwd_128", "value": pairs / elapsed, "unit": "pairs/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1, "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "ranks_in_sync": in_sync["ranks_in_sync"], "rank_sync": in_sync,
            "allreduce_exposed_ms_per_step": exposed[0], "allreduce_buckets": (len(ts._sync.buckets) if ts._sync is not None else 0),
            "config": {"workload": f"RegTR fwd+bwd+AdamW, shell-R synthetic pairs, {args.res}^3 grids, "
                                   f"{args.pairs} pairs ({2 * args.pairs} grids) per GPU per step, random-init weights",
                       "global_batch_pairs": args.pairs * world, "resolution": args.res,
                       "fpn_head": "dense" if args.dense_head else "active-set (identical results; rows around occupied voxels)",
                       "parallelism": f"dp{world}"},
            "roofline": roofline,
        }
        # the step as a whole against the same roof: every bracketed convolution / linear launch of the bracketed step (algorithmic
        # FLOPs; stride-2 data gradients at 1/8 of the dense taps) + the point-set half's linear layers when they ran unbracketed
        # (1,048,576 MAC per key point and encoder layer, 131,072 per decoder layer-row: SURVEY.md 8(d)), over the step's wall time
        bracketed = sum(fl for (c, m, fl) in prof.summary(subtract_overhead=False)[0].values()) / max(min(args.event_steps, args.steps), 1)
        lin = 0.0
        if not args.kernel_report and getattr(model, "last_batched", None) is not None and args.precision == "bf16":
            R = int(model.last_batched["xyz"].shape[0])
            lin = 3.0 * 2.0 * R * 6 * (1048576 + 131072)
        out["whole_step"] = {"algorithmic_TFLOP_per_step": (bracketed + lin) / 1e12, "achieved_TFLOPs": (bracketed + lin) / (elapsed / args.steps) / 1e12,
                             "frac_of_mfma_roof": (bracketed + lin) / (elapsed / args.steps) / 1e12 / peak,
                             "note": "convolutions + linear layers only (attention, BatchNorm, LayerNorm, optimizer excluded from the FLOPs, included in the time)"}
        if dense is not None:
            out["dense_head"] = dense
        if sweep is not None:
            out["occupancy_sweep"] = sweep
        if with_labels is not None:
            out["labels_from_nerf_blocks"] = with_labels
        if world == 1 and not args.no_ngp_reference and args.precision == "bf16":
            # BASELINE.json configs[3] beside the headline (the driver only runs the default line): NGP grid extraction of one 128^3 block,
            # the same measurement as `python bench.py --ngp`, compact
            try:
                import copy
                a2 = copy.copy(args)
                a2.steps, a2.warmup, a2.ngp_windows = 40, 3, 5
                ng = ngp_measure(a2, rank, world, dev)
                rf_n = ng["roofline"]
                out["ngp_config4"] = {"metric": ng["metric"], "value": ng["value"], "unit": ng["unit"], "ms_per_block": ng["ms_per_step"], "timed": "median of 5 windows of 40 blocks",
                                      "points_per_block": ng["config"]["points_per_block"],
                                      "roofline": {k: rf_n.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")},
                                      "density_query": {k: rf_n["density_kernel"].get(k) for k in ("kernel", "avg_launch_ms", "Gpts_per_s", "traffic", "gathered_over_hbm_bytes")},
                                      **({"cpu_baseline": ng["cpu_baseline"]} if "cpu_baseline" in ng else {})}
            except Exception as e:      # never lose the headline line to the extra figure
                out["ngp_config4"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_ngp_reference and args.precision == "bf16" and args.res == 128:
            try:
                out["eval_end_to_end"] = eval_end_to_end(args, dev)
            except Exception as e:      # never lose the headline line to the extra figure
                out["eval_end_to_end"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            cres = args.cpu_res or (128 if (os.cpu_count() or 1) >= 32 else 64)
            out["cpu_baseline"] = cpu_baseline(min(cres, args.res), args.res, args.cpu_samples)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

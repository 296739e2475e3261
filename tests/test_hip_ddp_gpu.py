"""GPU: the data-parallel training step end to end — native trunk executor with its segmented backward, deferred weight-gradient
sums, GradSync buckets — with TWO ranks on the one GPU of the test box (gloo carries the all-reduce; on a node it is RCCL): the
averaged gradients of two ranks holding one pair each equal the gradients of one process holding both pairs."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _batch(seeds, dev):
    from dreg_nerf_amd import synth
    out = []
    for s in seeds:
        d = synth.shell_pair(64, s, s + 1, pose=synth.fixed_pose())
        out.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()})
    return out


def _grads_of_one_step(seeds, dev):
    from dreg_nerf_amd.regtr import NeRFRegTr
    from dreg_nerf_amd.train_step import TrainStep
    torch.manual_seed(1234)
    m = NeRFRegTr(precision="bf16").to(dev).train()
    ts = TrainStep(m)
    grabbed = {}
    real_step = ts.optimizer.step

    def step():                       # the gradient buffer as the optimizer sees it (after GradSync.finish)
        grabbed["g"] = ts.optimizer.flat_g[:ts.optimizer.n_active].clone()
        return real_step()
    ts.optimizer.step = step
    ts.step(_batch(seeds, dev))
    torch.cuda.synchronize()
    return grabbed["g"].cpu(), ts


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g, ts = _grads_of_one_step([21 + 2 * rank], dev)
    assert ts.world == 2 and ts._sync is not None and len(ts._sync.launched) == len(ts._sync.buckets) > 1
    torch.save(g, os.path.join(outdir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_average_equals_one_rank_with_both_pairs():
    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as td:
        ctx = mp.get_context("spawn")
        port = 29300 + (os.getpid() % 400)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, td)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
        g0, g1 = torch.load(os.path.join(td, "g0.pt")), torch.load(os.path.join(td, "g1.pt"))
    assert torch.equal(g0, g1)                         # every rank ends with the same averaged buffer
    single, ts = _grads_of_one_step([21, 23], dev)     # one process, both pairs: loss = mean over the pairs
    assert ts.world == 1
    # same per-grid activations (BatchNorm statistics are per grid), so the two differ by accumulation order only: per parameter
    # tensor the distance stays far below the gradient's own norm
    worst, seen = 0.0, 0
    for p, off in zip(ts.optimizer.params, ts.optimizer.offsets):
        n = p.numel()
        if off + n > single.numel():
            continue                                   # never-used parameters live behind n_active
        a, b = single[off:off + n].double(), g0[off:off + n].double()
        if float(a.norm()) > 1e-6 * float(single.double().norm()):
            worst = max(worst, float((a - b).norm()) / float(a.norm()))
            seen += 1
    print(f"two ranks vs one rank: worst per-parameter relative distance {worst:.3e} over {seen} tensors, whole buffer "
          f"{float((single.double() - g0.double()).norm()) / float(single.double().norm()):.3e}")
    # What differs between the two: the weight gradients' voxel splits (their boundaries follow B x D^3, so one process with two pairs
    # cuts the voxel range differently from two processes with one pair each), the bias column sums' partial groups, and the
    # all-reduce's own (a + b) / 2 — fp32 summation order only.  Measured 4.5e-7 per parameter tensor, 9e-8 over the buffer; a dropped
    # or doubled bucket, a missing 1 / world or a stale gradient would read >= 1e-2.
    assert seen > 100 and worst < 2e-5, (seen, worst)
    assert float((single.double() - g0.double()).norm()) < 2e-6 * float(single.double().norm())


def _rccl_worker(port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DREG_FORCE_GRAD_SYNC="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g, ts = _grads_of_one_step([21], dev)
    sync = ts._sync
    assert sync is not None and sync._use_avg and len(sync.launched) == len(sync.buckets) > 1
    ts.step(_batch([23], dev))                       # a second step through the same communicator / streams
    torch.cuda.synchronize()
    torch.save({"g": g, "p": ts.optimizer.flat_p[:ts.optimizer.n_active].cpu()}, os.path.join(outdir, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_group_of_one_rank_carries_the_gradient_exchange():
    """RCCL itself (backend 'nccl') on this one-GPU box: a process group of ONE rank, the bucketed exchange forced on
    (DREG_FORCE_GRAD_SYNC): ReduceOp.AVG on views of the flat gradient buffer, the collective's stream behind events of the main and
    parameter-gradient streams, record_stream on the slices, barrier, teardown.  The average over one rank is the identity: gradients
    and the parameters after two steps equal the run without any exchange, bit for bit."""
    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as td:
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=_rccl_worker, args=(29100 + (os.getpid() % 400), td))
        p.start()
        p.join(timeout=600)
        assert p.exitcode == 0
        got = torch.load(os.path.join(td, "rccl.pt"))
    single, ts = _grads_of_one_step([21], dev)
    assert ts._sync is None
    ts.step(_batch([23], dev))
    torch.cuda.synchronize()
    assert torch.equal(single, got["g"])
    assert torch.equal(ts.optimizer.flat_p[:ts.optimizer.n_active].cpu(), got["p"])


def test_bench_script_runs_with_two_ranks():
    """bench.py's multi-rank path (rank-0 broadcast of the weights, barrier + max-over-ranks timing, one JSON line from rank 0 with the
    whole-job rate) with two ranks on this box's one GPU over gloo (DREG_BENCH_BACKEND / DREG_BENCH_ONE_GPU test hooks)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DREG_BENCH_BACKEND="gloo", DREG_BENCH_ONE_GPU="1")
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--res", "64", "--pairs", "1", "--no-cpu-baseline", "--no-dense-reference"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job pairs/s = ranks x pairs / step time
    # the run validates itself: same parameters and same clipped gradient norm on both ranks after the timed steps (bench.py exits non-zero otherwise)
    rs = d["rank_sync"]
    assert d["ranks_in_sync"] is True and rs["ranks"] == 2 and rs["param_checksum_spread"] == 0.0 and rs["grad_norm_spread"] == 0.0 and rs["grad_norm"] > 0
    assert d["allreduce_buckets"] > 1 and d["allreduce_exposed_ms_per_step"] >= 0.0


def test_bench_script_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and WORLD_SIZE unset (the driver's form) must itself start two ranks, one JSON
    line from rank 0 with n_gpus = the process group's world size.  On this one-GPU box the two ranks share the GPU over gloo
    (test hooks); on a node the same path is one RCCL rank per GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DREG_BENCH_BACKEND="gloo", DREG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "64", "--pairs", "1",
           "--no-cpu-baseline", "--no-dense-reference"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["config"]["parallelism"] == "dp2"      # gloo hook: not an RCCL measurement
    # without the one-GPU hook the same command must refuse (one visible GPU < 2), not run one rank
    env.pop("DREG_BENCH_ONE_GPU")
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)


def test_two_rccl_ranks_on_two_gpus_equal_one_rank_on_the_concatenated_batch():
    """BASELINE.json configs[2] in miniature, over RCCL itself: needs TWO visible GPUs.  On the 1-GPU boxes of this pool it is skipped —
    loudly: RCCL with more than one rank has then NOT run (DESIGN.md 5 keeps 'scaling unmeasured'); what does run there are the gloo
    world-2 / world-4 tests, two gloo ranks sharing one GPU and the RCCL group of one rank above."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip(f"RCCL with 2 ranks NOT exercised: {torch.cuda.device_count()} GPU(s) visible (needs 2)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DREG_BENCH_BACKEND", "DREG_BENCH_ONE_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--res", "64", "--pairs", "1",
           "--no-cpu-baseline", "--no-dense-reference", "--no-nerf-labels-reference", "--no-ngp-reference"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["ranks_in_sync"] is True and d["rank_sync"]["ranks"] == 2 and d["rank_sync"]["param_checksum_spread"] == 0.0
    # and the gradients: two RCCL ranks with one pair each == one rank with both pairs (the gloo form of this check is
    # test_two_ranks_average_equals_one_rank_with_both_pairs; here the collective is RCCL's ReduceOp.AVG over xGMI / PCIe)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        ctx = mp.get_context("spawn")
        port = 31011 + (os.getpid() % 100)
        procs = [ctx.Process(target=_worker_rccl2, args=(r, 2, port, td)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
        got = [torch.load(os.path.join(td, f"g{r}.pt")) for r in range(2)]
        assert torch.equal(got[0], got[1])
        ref = _grads_of_one_step([1, 3], torch.device("cuda", 0))[0]
        assert (got[0] - ref).norm() <= 2e-2 * ref.norm()


def _worker_rccl2(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    g = _grads_of_one_step([1 + 2 * rank], torch.device("cuda", rank))[0]
    torch.save(g, os.path.join(outdir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()

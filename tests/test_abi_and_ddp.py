"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls);
the data-parallel gradient path (FlatAdamW.all_reduce_mean) with gloo, world_size 2."""
import ctypes
import os
import re

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(dreg_\w+)\s*\(", txt))


def _exports(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("dreg_")}


def test_library_exports_every_declared_symbol():
    from dreg_nerf_amd import build, lib
    if not os.path.exists(lib.LIB_PATH) or not os.path.exists(lib.PROBE_LIB_PATH):
        build.build(verbose=False)
    syms = _header_symbols("dreg_nerf.h")
    assert len(syms) >= 150
    # the product library exports exactly what include/dreg_nerf.h declares, and the python binding covers the same set
    assert _exports(lib.LIB_PATH) == syms
    assert set(lib.declared_symbols()) == syms
    # the measurement build: the same + include/dreg_nerf_probe.h
    psyms = _header_symbols("dreg_nerf_probe.h")
    assert _exports(lib.PROBE_LIB_PATH) == syms | psyms
    assert set(lib.probe_symbols()) == psyms


def test_product_library_has_no_process_global_switches():
    """a re-entrant library.  Nothing the product .so exports sets process-wide state: every remaining *_set_* takes the handle it
    configures as its first argument (dreg_exec_* / dreg_ps_*), and no symbol of include/dreg_nerf_probe.h (the kernel-variant knobs, the
    "wrong results" timing ablations) exists in it."""
    from dreg_nerf_amd import lib
    exp = _exports(lib.LIB_PATH)
    setters = {s for s in exp if "_set_" in s or "probe" in s}
    assert setters == {"dreg_exec_set_overlap", "dreg_exec_set_input_row_occupancy", "dreg_exec_set_timing", "dreg_ps_set_fuse", "dreg_ps_set_group_wgrad",
                       "dreg_ps_set_timing"}, setters
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dreg_nerf.h")).read(), flags=re.S)
    for s in setters:
        assert re.search(r"\b%s\s*\(\s*void\s*\*\s*h\b" % s, txt), f"{s} is not handle-scoped"
    assert not (exp & _header_symbols("dreg_nerf_probe.h"))
    # no writable data symbol of the library's own either (device-side globals live in the code objects, not in the host image)
    import subprocess
    out = subprocess.check_output(["nm", "--defined-only", lib.LIB_PATH], text=True)
    writable = [l for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "dDbB" and re.search(r"\bg_[a-z]", l.split()[2])]
    assert not writable, writable


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers compile as C99 with -pedantic (what a cgo / JNI / ctypes-generator front end would feed them to), and a C
    program can name a struct of the ABI (dreg_exec_opts, dreg_bn_extra) by value."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "hdr.c"
    src.write_text(f'#include "{ROOT}/include/dreg_nerf.h"\n#include "{ROOT}/include/dreg_nerf_probe.h"\n'
                   "int main(void) { dreg_exec_opts o; dreg_bn_extra e; e.splitk_nsplit = 0; o.guard = e.splitk_nsplit; return o.guard + DREG_OK + (DREG_EGUARD != -3); }\n")
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_device_code_has_no_packed_fp32_instructions():
    """v_pk_*_f32 returned wrong lanes next to the implicit-GEMM kernels of a second stream (DESIGN.md "co-execution",
    tools/hw_probe); the build flags keep them out of every code object and this pins it."""
    from dreg_nerf_amd import build
    build.build(verbose=False)
    checked = 0
    for src in build.sources():
        asm = build.device_disassembly(src[:-4] + ".o")
        if not asm:
            continue                                         # host-only translation unit
        assert "s_endpgm" in asm, src
        checked += 1
        hits = re.findall(r"v_pk_(?:add|mul|fma|mov)_f32", asm)
        assert not hits, f"{os.path.basename(src)}: {len(hits)} packed-fp32 instructions"
    assert checked >= 7


def test_product_path_fails_loudly_without_library(monkeypatch):
    from dreg_nerf_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libdreg_nerf_hip.so")
    with pytest.raises(lib.DregError):
        lib.load()


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreg_nerf_amd.optim import FlatAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 300, 5, 1000)]
    opt = FlatAdamW(ps)
    opt.zero_grad()
    for i, p in enumerate(ps):
        if not (rank == 1 and i == 2):  # one rank contributes no gradient for a parameter: it stays zero there
            p.grad.add_(float(rank + 1) * (i + 1))
    opt.all_reduce_mean(world, bucket_elems=256)
    # the self-check bench.py's multi-rank line carries (optim.ranks_in_sync): identical parameters + the same gradient norm on every rank ...
    from dreg_nerf_amd.optim import ranks_in_sync
    ok = ranks_in_sync(opt, opt.flat_g.norm().reshape(1))
    opt.flat_p[301 + rank] += 1e-3 * rank                       # ... and a rank that has drifted by one parameter is seen by every rank
    bad = ranks_in_sync(opt, opt.flat_g.norm().reshape(1))
    q.put((rank, [p.grad.tolist() for p in ps], ok, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + (os.getpid() % 200)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    res = {g[0]: g[1] for g in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, _, ok, bad in got:
        assert ok["ranks_in_sync"] is True and ok["ranks"] == 2 and ok["param_checksum_spread"] == 0.0 and ok["grad_norm_spread"] == 0.0
        assert bad["ranks_in_sync"] is False and bad["param_checksum_spread"] > 0
    for i in range(4):
        exp = (1.0 * (i + 1) + (0.0 if i == 2 else 2.0 * (i + 1))) / 2
        for r in range(2):
            assert all(abs(v - exp) < 1e-6 for v in res[r][i])


def _sync_worker(rank, world, port, q):
    """GradSync (the overlapped gradient averaging of train_step.TrainStep) over gloo: rank r holds the gradient of 'its' pair; after
    the bucketed all-reduce — launched in pieces while 'backward' reports progress — both ranks hold the gradient of the 1-rank step on
    the concatenated batch [pair a, pair b] (loss = mean over pairs), bit for bit the same on both."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreg_nerf_amd.optim import FlatAdamW, GradSync
    sizes = (7, 300, 5, 1000, 64, 129)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    opt = FlatAdamW(ps, never_used=[ps[2]])                     # one parameter takes no part in the forward pass: it sits past n_active
    assert opt.n_active == sum(sizes) - 5 and opt.offsets[2] == opt.n_active
    opt.zero_grad()
    g = torch.Generator().manual_seed(100 + rank)               # this rank's pair
    mine = torch.randn(opt.n_active, generator=g)
    opt.flat_g[:opt.n_active].copy_(mine)
    sync = GradSync(opt, world, bucket_bytes=256 * 4)
    # backward produces the buffer from its end: progress reports at arbitrary offsets, finish() sweeps the rest
    order = []
    for lo in (1400, 1100, 1100, 700, 123):
        sync.ready(lo)
        order.append(len(sync.launched))
    sync.finish()
    q.put((rank, mine.tolist(), opt.flat_g.tolist(), sync.launched, order, sync.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_two_ranks_equal_one_rank_on_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29811 + (os.getpid() % 150)
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_active = len(res[0][0])
    want = ((torch.tensor(res[0][0]) + torch.tensor(res[1][0])) / 2).tolist()        # d/dtheta of mean(loss_a, loss_b)
    for r in range(2):
        mine, flat, launched, order, buckets = res[r]
        assert flat[:n_active] == want and all(v == 0.0 for v in flat[n_active:])     # never-used tail untouched
        # buckets: launched from the END of the buffer, each exactly once, covering [0, n_active)
        assert launched == buckets and launched[0][1] == n_active and launched[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(launched, launched[1:]))
        # a bucket starts only once everything above its lower bound was reported ready
        assert order == sorted(order) and order[0] == sum(1 for lo, _ in buckets if lo >= 1400) and order[-1] == sum(1 for lo, _ in buckets if lo >= 123)
    assert res[0][1] == res[1][1]


def _uneven_worker(rank, world, port, q):
    """Four ranks whose backward passes report progress on DIFFERENT schedules (each rank reaches the bucket boundaries after a different
    number of ready() calls, some ranks skip straight to finish()): buckets must still be launched in ONE order on every rank — the
    collectives of a process group match by issue order — and the exchange must complete (a rank launching bucket k+1 before bucket k
    while a peer does the opposite is the deadlock / mismatched-reduction case)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreg_nerf_amd.optim import FlatAdamW, GradSync
    import time
    sizes = (513, 7, 1000, 64, 129, 300, 2048, 11)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    opt = FlatAdamW(ps)
    sync = GradSync(opt, world, bucket_bytes=200 * 4)
    schedules = {0: (4000, 3900, 3000, 2999, 2000, 1000, 500, 1),      # fine-grained
                 1: (2000,),                                         # one report in the middle
                 2: (),                                              # nothing before finish()
                 3: (4071, 10, 9, 8)}                                # a late jump over many buckets
    out = []
    for step in range(3):                                            # several steps: begin() must reset the bucket cursor
        opt.zero_grad()
        g = torch.Generator().manual_seed(1000 * step + rank)
        mine = torch.randn(opt.n_active, generator=g)
        opt.flat_g[:opt.n_active].copy_(mine)
        sync.begin()
        for lo in schedules[rank]:
            time.sleep(0.002 * ((rank * 7 + step) % 3))              # ranks drift apart in time as well
            sync.ready(lo)
        sync.finish()
        out.append((mine.tolist(), opt.flat_g[:opt.n_active].tolist(), list(sync.launched)))
    q.put((rank, out, sync.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_world4_uneven_ready_schedules():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30111 + (os.getpid() % 150)
    procs = [ctx.Process(target=_uneven_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(4))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        want = sum(torch.tensor(res[r][0][step][0]) for r in range(4)) / 4
        for r in range(4):
            mine, flat, launched = res[r][0][step]
            assert torch.allclose(torch.tensor(flat), want, rtol=0, atol=1e-6)
            assert launched == res[r][1], "every bucket exactly once, from the end of the buffer, on every schedule"
        assert all(res[r][0][step][1] == res[0][0][step][1] for r in range(4)), "all ranks hold the same averaged gradient, bit for bit"


def _bn_buffer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreg_nerf_amd.optim import broadcast_buffers
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Conv3d(2, 4, 1), torch.nn.BatchNorm3d(4), torch.nn.BatchNorm3d(4))
    m.alias = m[1]                                                   # the same module under a second name (the reference's ResNet alias)
    with torch.no_grad():
        for i, b in enumerate(m.buffers()):
            b.copy_(torch.full_like(b, 10 * rank + i))                 # every rank has drifted to its own statistics
        w0 = m[0].weight.clone()
    n = broadcast_buffers(m, 0)
    q.put((rank, n, [b.tolist() if b.dim() else int(b) for b in m.buffers()], bool(torch.equal(m[0].weight, w0))))
    dist.barrier()
    dist.destroy_process_group()


def test_bn_buffers_follow_rank0_at_checkpoint_time():
    """BatchNorm running statistics are rank-local under plain DDP; broadcast_buffers (called by train_nerf_regtr.py
    before validation and at the checkpoint cadence) gives every rank rank 0's — floats and the int64 counters, parameters untouched."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30411 + (os.getpid() % 150)
    procs = [ctx.Process(target=_bn_buffer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == 6                               # 2 x (mean, var, counter); the alias is not sent twice
    assert res[0][1] == res[1][1]
    assert res[1][1][0] == [0.0] * 4 and res[1][1][2] == 2            # rank 1 now holds rank 0's values (10 * 0 + i)
    assert res[0][2] and res[1][2]


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a box with fewer GPUs must fail loudly, never run fewer ranks (here: no GPU at all)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DREG_BENCH_ONE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "refusing to run fewer ranks" in (out.stderr + out.stdout)
    # a launcher whose rank count disagrees with --gpus is an error too
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in (out.stderr + out.stdout)


def test_profiler_labels_come_from_the_librarys_dispatch_rules():
    """bench.py's roofline labels must name the LAUNCHED template instantiation (one rocprofv3 row): they are produced by
    dreg_conv3d_igemm_variant / dreg_conv3d_wgrad_variant, the library's own dispatch rules (host-only calls: no GPU needed)."""
    from dreg_nerf_amd import lib as L, ops
    lib = L.load()
    name = lambda *a: ops.igemm_kernel_name(lib, *a)
    # (B, Di, Hi, Wi, cin, Do, Ho, Wo, cout, k, stride, pad, transposed, nrows, has_ws, has_addend, dtype, out_f32)
    assert name(8, 64, 64, 64, 256, 64, 64, 64, 256, 3, 1, 1, 0, 0, 1, False, L.DT_BF16, False) == "conv_igemm_glds_kernel<bf16,256,256,0,1>"
    assert name(8, 32, 32, 32, 64, 32, 32, 32, 64, 3, 1, 1, 0, 0, 1, False, L.DT_BF16, False) == "conv_igemm_glds_kernel<bf16,128,64,0,0>"       # 2,048 tiles: four-wave tile
    assert name(8, 16, 16, 16, 128, 16, 16, 16, 128, 3, 1, 1, 0, 0, 1, False, L.DT_BF16, False) == "conv_igemm_glds_kernel<bf16,128,128,0,1>"  # 256 tiles: anti-phase form
    sk = name(8, 8, 8, 8, 256, 8, 8, 8, 256, 3, 1, 1, 0, 0, 1, False, L.DT_BF16, False)                                                       # layer3: split-K, two launches
    assert sk.startswith("conv_igemm_glds_kernel<f32,128,128,0,") and sk.endswith("+splitk_reduce")
    assert name(8, 64, 64, 64, 256, 64, 64, 64, 256, 3, 1, 1, 0, 90000, 0, False, L.DT_BF16, False) == "conv_igemm_glds_kernel<bf16,256,256,0,1>"   # row list
    assert name(8, 128, 128, 128, 8, 64, 64, 64, 64, 5, 2, 2, 0, 0, 1, False, L.DT_BF16, False) == "conv_igemm_kernel<bf16,bf16,64>"            # the stem: register-staged
    assert lib.dreg_conv3d_wgrad_variant(8, 64, 64, 64, 256, 256, 3, 0, 0, 0) == 256256
    assert lib.dreg_conv3d_wgrad_variant(8, 64, 64, 64, 64, 256, 3, 0, 0, 0) == 256256      # 27 x 64 columns: ragged last tile of the 256-wide form
    assert lib.dreg_conv3d_wgrad_variant(8, 16, 16, 16, 128, 128, 3, 0, 0, 0) == 128128


class _FakeSplit:
    """Stands in for NeRFRegDataset on the host: scenes with two or three blocks, block order drawn like dataset.draw_block_order (quirk Q15)."""

    def __init__(self, n):
        self.meta = [{"scene": f"s{i:02d}", "blocks": {k: None for k in range(2 + (i % 3 == 0))}} for i in range(n)]

    def __len__(self):
        return len(self.meta)

    def draw_block_order(self, index):
        import random
        ids = list(self.meta[index]["blocks"].keys())
        random.shuffle(ids)
        return ids


def _eval_shard_worker(rank, world, port, q):
    """eval_nerf_regtr.py's sharding (dreg_nerf_amd/eval_shard.py) over gloo: every rank seeds like the script, draws ALL scenes' block orders, 'evaluates'
    its scenes rank::world (a row derived from the scene and its drawn order), gathers."""
    import random
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreg_nerf_amd import eval_shard as ES
    random.seed(3407)
    ds = _FakeSplit(11)
    orders = ES.block_orders(ds)
    rows = {}
    for i in ES.my_scenes(len(ds), rank, world):
        s, t = orders[i][0], orders[i][1]
        rows[ds.meta[i]["scene"]] = {"R_mean": 10.0 * i + s, "t_mean": 0.1 * i + 0.01 * t, "order": orders[i]}
    allrows = ES.gather_rows(rows, world)
    q.put((rank, ES.summary({k: {"R_mean": v["R_mean"], "t_mean": v["t_mean"]} for k, v in allrows.items()}), {k: v["order"] for k, v in allrows.items()}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_eval_scene_sharding_gloo_world2_equals_one_rank():
    """SURVEY.md 8(e) 'RegTR eval': scenes sharded rank::world, rows gathered with all_gather_object — the two-rank result (on BOTH ranks) is the one-rank
    result, including which block of every scene is the source (the order is drawn for all scenes on every rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    one = ctx.Process(target=_eval_shard_worker, args=(0, 1, 0, q))
    one.start()
    _, want, want_orders = q.get(timeout=120)
    one.join(timeout=60)
    port = 29741 + (os.getpid() % 200)
    procs = [ctx.Process(target=_eval_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(want) == 11 + 2 and any(len(o) == 3 for o in want_orders.values())
    for _, summ, orders in got:
        assert orders == want_orders
        assert set(summ) == set(want) and all(summ[k] == want[k] for k in want)
    from dreg_nerf_amd import eval_shard as ES
    assert ES.my_scenes(7, 1, 3) == [1, 4] and ES.my_scenes(2, 3, 4) == []

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


def _header_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            syms |= set(re.findall(r"\b(dreg_\w+)\s*\(", txt))
    return syms


def test_library_exports_every_declared_symbol():
    from dreg_nerf_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build(verbose=False)
    cdll = ctypes.CDLL(lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(cdll, s), f"{s} declared in include/ but not exported"
    # the python binding covers the same set
    assert set(lib.declared_symbols()) == syms


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
    q.put((rank, [p.grad.tolist() for p in ps]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + (os.getpid() % 200)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i in range(4):
        exp = (1.0 * (i + 1) + (0.0 if i == 2 else 2.0 * (i + 1))) / 2
        for r in range(2):
            assert all(abs(v - exp) < 1e-6 for v in res[r][i])

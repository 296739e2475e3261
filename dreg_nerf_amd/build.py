"""Build libdreg_nerf_hip.so (gfx950) in-tree with hipcc.  `python -m dreg_nerf_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdreg_nerf_hip.so")
# The measurement build: the same sources with -DDREG_PROBE (csrc/common.h DREG_KNOB: the kernel-variant knobs become mutable, their
# process-global setters of include/dreg_nerf_probe.h are exported).  Loaded explicitly by tools/ and the variant tests only.
PROBE_LIB = os.path.join(HERE, "libdreg_nerf_hip_probe.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize / -fno-vectorize: keep packed-fp32 VALU instructions (v_pk_{add,mul,fma}_f32) out of the device code.  On this
# hardware a v_pk_*_f32 in one wave returned a wrong low element in lanes 48..63 while the implicit-GEMM kernels ran on a second
# stream of the same CU (tools/hw_probe, DESIGN.md "co-execution"); every kernel here may share a CU with the executor's
# weight-gradient stream, so none of them may contain one.  tests/test_abi_and_ddp.py checks the built code objects.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize"]


# Per-file additions.  attention.hip: its MFMA accumulators are small (8-32 registers) and every tile's results go straight into VALU
# work (softmax); with the accumulators in AGPRs the compiler moved them to VGPRs and back around every MFMA group (~90 v_accvgpr
# moves per 64-key tile of the forward kernel, VALU-bound).  The convolution kernels keep the default: their 128-256 accumulators
# need the AGPR half of the register file.
# (ngp.hip has the same pattern — MFMA, activation on the VALU, next MFMA — but measured no faster with the flag: 1,034 vs 1,067 blocks/s.)
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _build_one(lib, suffix, extra, force, verbose):
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", f) for f in ("dreg_nerf.h",)]
    objs = []
    newest_hdr = max(os.path.getmtime(h) for h in hdrs) if hdrs else 0
    procs = []
    for s in srcs:
        o = s[:-4] + suffix
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            cmd = [HIPCC] + FLAGS + extra + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    return objs, procs


def _link(lib, objs, relink, verbose):
    if relink or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = True, probe: bool = True) -> str:
    """Compile every csrc/*.hip for gfx950 and link libdreg_nerf_hip.so (the product) and, with probe=True, libdreg_nerf_hip_probe.so."""
    objs, procs = _build_one(LIB, ".o", [], force, verbose)
    pobjs, pprocs = _build_one(PROBE_LIB, ".probe.o", ["-DDREG_PROBE=1"], force, verbose) if probe else ([], [])
    for s, p in procs + pprocs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    _link(LIB, objs, force or bool(procs), verbose)
    if probe:
        _link(PROBE_LIB, pobjs, force or bool(pprocs), verbose)
    return LIB


def device_disassembly(obj: str) -> str:
    """Disassembly of the gfx950 code object bundled in one compiled .o (used by the packed-fp32 check)."""
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        import shutil
        local = os.path.join(td, "unit.o")
        shutil.copy(obj, local)
        subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "--offloading", local], stderr=subprocess.STDOUT)   # extracts next to it
        co = [f for f in os.listdir(td) if "amdgcn" in f]
        if not co:
            return ""                                        # host-only translation unit (executor.hip)
        return subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", os.path.join(td, co[0])], text=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""GPU: bench.py's roofline labels against rocprofv3 itself.  The bracket table of the bracketed step (HIP events around every
convolution / linear launch, labelled with the template instantiation the library's dispatch rules name for the ACTUAL row counts)
must agree with the kernel trace of the same run: for every single-kernel label, bracketed launches per step == rocprofv3 Calls / steps."""
import csv
import os
import re
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(k: str) -> str:
    """rocprofv3's kernel name -> the spelling of bench.py's labels (template arguments that the label names, in order)."""
    k = k.replace("void ", "").split("(")[0].replace("unsigned short", "bf16").replace("float", "f32").replace(" ", "")
    return k


def _run_under_rocprof(tmp_path, tag, extra):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.skip("rocprofv3 not installed")
    env = dict(os.environ, TMPDIR=str(tmp_path))
    cmd = [rocprof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(tmp_path / tag), "-o", "k", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--no-cpu-baseline", "--no-dense-reference", "--no-nerf-labels-reference", "--no-ngp-reference"] + extra
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    stats = None
    for dp, _, fs in os.walk(tmp_path / tag):
        for f in fs:
            if f.endswith("kernel_stats.csv"):
                stats = os.path.join(dp, f)
    assert stats, "rocprofv3 wrote no kernel_stats.csv"
    calls = {}
    for r in csv.DictReader(open(stats)):
        calls[_norm(r["Name"])] = calls.get(_norm(r["Name"]), 0) + int(r["Calls"])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return calls, (line[-1] if line else None)


def _calls_of(calls, name):
    want = name.replace(" ", "")
    rows = [k for k in calls if k == want or k.startswith(want.rstrip(">") + ",")]
    assert rows, f"label {name} names no kernel of the trace: {sorted(calls)[:40]}"
    return sum(calls[k] for k in rows)


def test_bracket_labels_match_rocprofv3_rows(tmp_path):
    """Every single-kernel label of the per-kernel report: one step, every convolution / linear launch bracketed (per-op point-set half)."""
    report = tmp_path / "report.tsv"
    calls, _ = _run_under_rocprof(tmp_path, "one", ["--steps", "1", "--warmup", "0", "--kernel-report", str(report)])
    bracket = {}
    for line in list(open(report))[1:]:
        name, label, c = line.split("\t")[:3]
        bracket[name] = bracket.get(name, 0) + int(c)
    checked, bad = 0, []
    plus = {n.split("+")[0] for n in bracket if "+" in n}     # also launched under a two-kernel bracket ('...+splitk_reduce', '...+reduce')
    for name, c in bracket.items():
        if "+" in name or not name.startswith("conv") or name in plus:
            continue
        n = _calls_of(calls, name)
        if n != c:
            bad.append((name, c, n))
        checked += 1
    assert checked >= 8
    assert not bad, f"(label, bracketed launches, rocprofv3 calls): {bad}"


def test_roofline_kernel_of_the_default_line_matches_rocprofv3(tmp_path):
    """The default line (native executors in every step, the bracketed one included): roofline.launches x steps == rocprofv3 Calls of
    roofline.kernel, and its by_shape launches add up to it."""
    import json
    steps, warm = 3, 1
    calls, line = _run_under_rocprof(tmp_path, "dflt", ["--steps", str(steps), "--warmup", str(warm)])
    d = json.loads(line)
    rf = d["roofline"]
    assert rf["kernel"].startswith("conv") and "+" not in rf["kernel"]
    assert _calls_of(calls, rf["kernel"]) == rf["launches"] * (steps + warm), (rf["kernel"], rf["launches"], _calls_of(calls, rf["kernel"]))
    assert sum(sh["launches"] for sh in rf["by_shape"]) <= rf["launches"]
    assert d["whole_step"]["frac_of_mfma_roof"] > 0.05

"""GPU: the trunk executor's guard mode (dreg_exec_opts.guard, include/dreg_nerf.h): every region of the arena is followed by a poisoned 64 KiB
band; a training step leaves all of them untouched (mode 2: checked behind every op of both passes), and a deliberately misplaced write is
found, attributed to the region in front of the band and, in mode 2, to the op behind which the scan first saw it."""
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import lib as L, synth, trunk_exec  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402
from dreg_nerf_amd.train_step import TrainStep  # noqa: E402

DEV = torch.device("cuda", 0)


def _batch(res=64, n=2):
    out = []
    for i in range(n):
        d = synth.shell_pair(res, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
        out.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()})
    return out


@pytest.mark.parametrize("mode", [1, 2])
def test_training_step_leaves_every_guard_band_untouched(mode):
    torch.manual_seed(3407)
    with trunk_exec.exec_opts(guard=mode):
        m = NeRFRegTr(precision="bf16").to(DEV).train()
        ts = TrainStep(m)
        for _ in range(2):
            out = ts.step(_batch())
        torch.cuda.synchronize()
        exs = list(m.__dict__["_trunk_cache"].values())
        assert exs and all(ex.opts.guard == mode for ex in exs)
        for ex in exs:
            assert ex.lib.dreg_exec_guard_bands(ex.h) > 300          # one per activation, gradient, statistics / scratch buffer
            assert ex.guard_check()[0] == 0
    assert torch.isfinite(out["losses"]["total"])


def test_guard_mode_changes_no_result():
    """Same step with and without the bands (they only move the regions apart): losses and gradient norm bit for bit."""
    res = []
    for g in (0, 1):
        torch.manual_seed(3407)
        with trunk_exec.exec_opts(guard=g):
            m = NeRFRegTr(precision="bf16").to(DEV).train()
            out = TrainStep(m).step(_batch())
            torch.cuda.synchronize()
        res.append(({k: float(v) for k, v in out["losses"].items()}, float(out["grad_norm"])))
    assert res[0] == res[1]


def test_a_stray_write_is_found_and_attributed():
    torch.manual_seed(0)
    with trunk_exec.exec_opts(guard=1):
        m = NeRFRegTr(precision="bf16").to(DEV).train()
        ts = TrainStep(m)
        ts.step(_batch(64, 1))
        torch.cuda.synchronize()
        ex = next(iter(m.__dict__["_trunk_cache"].values()))
        nb = ex.lib.dreg_exec_guard_bands(ex.h)
        band = nb // 2
        desc = ex.guard_describe(band)
        off = int(re.search(r"arena offset (\d+)", desc).group(1))
        ex.arena[off + 4096 + 3] = 7                                 # one byte, 4 KiB behind the end of the region in front of the band
        r = ex.guard_check()
        assert r[0] == 1 and r[1] == band and r[2] == 4096 and r[3] == 1, r
        with pytest.raises(L.DregError, match="guard band"):
            ts.step(_batch(64, 1))

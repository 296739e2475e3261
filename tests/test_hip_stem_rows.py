"""GPU: the network's stem (conerf/model/resnet3d.py conv1: 5^3 taps, stride 2, pad 2, no bias) evaluated only on the output voxels whose
receptive field holds an occupied input voxel (dreg_conv_rows -> dreg_conv3d_igemm_rows / row-list weight gradient): the dense result
bit for bit, the weight gradient to fp32 summation order, a whole training step, and the contract check."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreg_nerf_amd import lib as L, ops, params, synth  # noqa: E402
from dreg_nerf_amd.regtr import NeRFRegTr  # noqa: E402
from dreg_nerf_amd.train_step import TrainStep  # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("res,ksz,stride,pad", [((32, 32, 32), 5, 2, 2), ((33, 20, 27), 5, 2, 2), ((16, 16, 16), 3, 1, 1), ((24, 24, 24), 3, 2, 1)])
def test_conv_rows_is_the_max_pooled_occupancy(res, ksz, stride, pad):
    lib = L.load()
    Z, X, Y = res
    g = torch.Generator().manual_seed(3)
    B = 3
    occ = torch.rand(B, Z, X, Y, generator=g) < 0.02
    occ[1] = False                                           # an empty grid
    occ[2, 0, 0, 0] = occ[2, Z - 1, X - 1, Y - 1] = True     # corners
    idxs, pb = [], []
    for b in range(B):
        z, x, y = torch.nonzero(occ[b], as_tuple=True)
        idxs.append(((x * Y + y) * Z + z).long())            # flat fine index (x * Yr + y) * Zr + z
        pb.append(torch.full((z.numel(),), b, dtype=torch.int32))
    idx, pb = torch.cat(idxs).to(DEV), torch.cat(pb).to(DEV)
    d, h, w = ((r + 2 * pad - ksz) // stride + 1 for r in res)
    V = B * d * h * w
    nb = int(lib.dreg_conv_rows_workspace_bytes(B, d, h, w))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    rows = torch.empty(V, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.check(lib.dreg_conv_rows(L.ptr(idx), L.ptr(pb), idx.numel(), B, Z, X, Y, d, h, w, ksz, stride, pad, L.ptr(rows), L.ptr(cnt), L.ptr(ws), nb, L.stream()),
            "dreg_conv_rows")
    ref = torch.nn.functional.max_pool3d(occ.float()[:, None], ksz, stride, pad)[:, 0] > 0
    ref_rows = torch.nonzero(ref.flatten())[:, 0].int()
    n = int(cnt.item())
    assert n == ref_rows.numel() and torch.equal(rows[:n].cpu(), ref_rows)


def _batch(res, n_pairs=2):
    out = []
    for i in range(n_pairs):
        d = synth.shell_pair(res, 1 + 2 * i, 2 + 2 * i, pose=synth.fixed_pose())
        out.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()})
    return out


def _step(stem_rows, native, res=64):
    torch.manual_seed(3407)
    m = NeRFRegTr(precision="bf16")
    m.load_state_dict(params.synth_state_dict(0, profile="wc"), strict=True)
    m = m.to(DEV).train()
    m.stem_rows, m.native_trunk = stem_rows, native
    ts = TrainStep(m)
    out = ts.step(_batch(res))
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return {k: float(v) for k, v in out["losses"].items()}, float(out["grad_norm"]), g, m


@pytest.mark.parametrize("native", [True, False])
def test_training_step_with_the_stem_on_its_row_list(native):
    """The same step with the dense stem and with the stem on its row list (through the native executor and through the per-op path):
    identical losses (the forward pass is bit-identical), the stem's weight gradient equal to summation order, every other gradient
    within the bf16 noise that difference causes downstream of nothing (they are upstream of the stem: identical).  (The executor's
    BatchNorm / pool behind the row-list stem in its dense form, creation option sparse_stem = 0: the list form sums the statistics in another
    order — tests/test_hip_sparse_stem.py.)"""
    from dreg_nerf_amd.trunk_exec import exec_opts
    with exec_opts(sparse_stem=0):
        la, na, ga, _ = _step(False, native)
        lb, nb, gb, m = _step(True, native)
    assert la == lb
    w = "fpn3d.backbone_net.conv1.weight"
    rel = float((ga[w] - gb[w]).norm() / ga[w].norm())
    assert 0 <= rel < 1e-5, rel
    for k in ga:
        if k != w:
            assert torch.equal(ga[k], gb[k]), k
    assert abs(na - nb) <= 1e-6 * na


def test_executor_and_per_op_path_forward_agree_bit_for_bit_with_stem_rows():
    """With every BatchNorm's statistics from its own statistics pass (creation option fuse_bn_stats = 0: the per-op path's arithmetic) the
    native executor's forward pass equals the per-op path's bit for bit — also with the stem on its row list in both."""
    from dreg_nerf_amd.trunk_exec import exec_opts
    with exec_opts(fuse_bn_stats=0, sparse_stem=0):
        la, na, ga, _ = _step(True, True)
        lb, nb, gb, _ = _step(True, False)
    assert la == lb
    w = "fpn3d.backbone_net.conv1.weight"
    assert float((ga[w] - gb[w]).norm() / ga[w].norm()) < 2e-2


def test_values_outside_the_mask_are_reported():
    """stem_rows rests on "a grid is zero outside its voxel_mask"; in evaluation the call that was handed a violating grid returns NaN poses (no host
    sync inside the call) and check_inputs() / the next call raise; in training a sticky device flag is read by the next call."""
    torch.manual_seed(0)
    m = NeRFRegTr(precision="bf16").to(DEV).eval()
    batch = _batch(64, 1)
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch[0].items()}
    g = bad["src_xyz_rgba"]
    g.view(-1)[::7919] += 0.5                                 # values all over the volume
    with torch.no_grad():
        ok = m.forward_batch(batch)                           # a clean call first
        m.check_inputs()
        assert torch.isfinite(ok[0]["pose"]).all()
        out = m.forward_batch([bad])
        assert torch.isnan(out[0]["pose"]).all()              # never a plausible pose from dropped input
        with pytest.raises(ValueError, match="voxel_mask"):
            m.check_inputs()
        m.forward_batch([bad])
        with pytest.raises(ValueError, match="voxel_mask"):   # ... or the next call reports it
            m.forward_batch(batch)
        assert torch.isfinite(m.forward_batch(batch)[0]["pose"]).all()
    # training: the flag of a sampled call is read by the next call (no host sync on fresh work)
    m.train()
    m.__dict__["_stem_calls"] = 0                             # (training samples the check: the first calls and every 64th)
    with torch.no_grad():
        m.forward_batch([bad])
        with pytest.raises(ValueError, match="voxel_mask"):
            m.forward_batch(batch)

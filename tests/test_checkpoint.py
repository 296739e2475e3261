"""Row N3: checkpoint manager + NeRF block state loaders, on CPU.

The manager is replayed against tests/golden/checkpoint_manager.json, which records what the reference's CheckPointManager
(conerf/base/checkpoint_manager.py) left on disk for the same scripted sequence of saves and loads (generator:
tests/golden/make_checkpoint_golden.py)."""
import enum
import json
import os
import pickle

import pytest
import torch

from dreg_nerf_amd import ngp
from dreg_nerf_amd.checkpoint import CheckPointManager, de_parallel
from dreg_nerf_amd.optim import FlatAdamW, StepLR

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_manager.json")))


def _listing(root):
    out = []
    for d, _, fs in os.walk(root):
        for f in fs:
            out.append(os.path.relpath(os.path.join(d, f), root))
    return sorted(out)


def test_manager_leaves_the_same_files_as_the_reference(tmp_path):
    td = str(tmp_path)
    torch.manual_seed(0)
    model = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    mgr = CheckPointManager(td, max_to_keep=GOLD["max_to_keep"], verbose=False)
    assert _listing(td) == GOLD["after_init"]["files"]
    assert open(os.path.join(td, "checkpoints.txt")).read() == GOLD["after_init"]["checkpoints_txt"]
    for (step, score), want in zip(GOLD["script"], GOLD["saves"]):
        model(torch.ones(1, 3)).sum().backward()
        opt.step(); sched.step()
        mgr.save({"model": model}, {"optimizer": opt}, step, schedulers={"scheduler": sched}, meta_data={"aabb": [0, 0, 0, 1, 1, 1]}, score=score)
        assert _listing(td) == want["files"], step
        assert open(os.path.join(td, "checkpoints.txt")).read() == want["checkpoints_txt"], step
        best = torch.load(os.path.join(td, "model_best.pth"), weights_only=False)
        last = torch.load(os.path.join(td, "model.pth"), weights_only=False)
        assert int(best["step"]) == want["best_step"] and int(last["step"]) == want["latest_step"]
        assert sorted(last.keys()) == want["state_keys"]
    fresh, meta = torch.nn.Linear(3, 2), {"aabb": None}
    got = CheckPointManager(td, max_to_keep=3, verbose=False)
    assert open(os.path.join(td, "checkpoints.txt")).read() == GOLD["reopen_checkpoints_txt"]
    assert got.load_no_config(os.path.join(td, "model", "model_000600.pth"), models={"model": fresh}, meta_data=meta, map_location="cpu") == GOLD["load_explicit_step"]
    assert meta["aabb"] == GOLD["load_explicit_meta"]
    assert got.load_no_config("", models={"model": fresh}, map_location="cpu") == GOLD["load_latest_step"]
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values())) == GOLD["load_latest_matches_model"]


def test_manager_errors_and_missing_checkpoint(tmp_path):
    with pytest.raises(ValueError) as e:
        CheckPointManager(None, max_to_keep=0)
    assert type(e.value).__name__ == GOLD["errors"]["max_to_keep_0"]
    with pytest.raises(AssertionError) as e:
        CheckPointManager(None, verbose=False).save({"model": torch.nn.Linear(1, 1)}, {}, 1)
    assert type(e.value).__name__ == GOLD["errors"]["save_without_path"]
    m = torch.nn.Linear(3, 2)
    assert CheckPointManager(str(tmp_path), verbose=False).load_no_config("", models={"model": m}) == GOLD["load_missing_step"] == 0
    # a name that is asked for but absent from the file is an error, as is a state_dict that does not fit
    mgr = CheckPointManager(str(tmp_path), verbose=False)
    mgr.save({"model": m}, {}, 5)
    with pytest.raises(KeyError):
        mgr.load_no_config("", models={"feature_loss": m}, map_location="cpu")
    with pytest.raises(RuntimeError):
        mgr.load_no_config("", models={"model": torch.nn.Linear(4, 2)}, map_location="cpu")

    class Cfg:
        ckpt_path, distributed, local_rank = "", False, 0
    assert mgr.load(Cfg(), models={"model": m}, map_location="cpu") == 5
    wrapped = torch.nn.Module()
    wrapped.module = m
    assert de_parallel(wrapped) is m and de_parallel(m) is m


def test_flat_adamw_state_is_interchangeable_with_torch_adamw(tmp_path):
    """The optimizer entry of a RegTR checkpoint is torch.optim.AdamW.state_dict() in the reference (train_nerf_regtr.py:298);
    FlatAdamW writes and reads that layout (needs no GPU: only the state plumbing is exercised)."""
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(ps, lr=1e-4, weight_decay=1e-4)
    for p in ps:
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    flat = FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-4)
    flat.load_state_dict(sd)
    out = flat.state_dict()
    assert set(out.keys()) == {"state", "param_groups"} and set(out["state"].keys()) == set(sd["state"].keys())
    for i in sd["state"]:
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(out["state"][i][k].cpu(), sd["state"][i][k])
        assert float(out["state"][i]["step"]) == float(sd["state"][i]["step"])
    torch.optim.AdamW(ps, lr=1e-4).load_state_dict(out)       # and torch accepts what FlatAdamW wrote
    s = StepLR(flat, step_size=2, gamma=0.5)
    for _ in range(5):
        s.step()
    s2 = StepLR(flat, step_size=2, gamma=0.5)
    s2.load_state_dict(s.state_dict())
    assert s2.get_last_lr() == s.get_last_lr() and s2.last_epoch == 5


def _reference_style_block_state(res=8):
    """What train_ngp_nerf.py:192-209 stores for one block, with the nerfacc enum pickled by module path."""
    install = ngp.install_pickle_shims
    install()
    import nerfacc.contraction as nc
    g = torch.Generator().manual_seed(3)
    field = ngp.NGPradianceField([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    with torch.no_grad():
        field.mlp_base.params.copy_(torch.randn(field.mlp_base.params.shape, generator=g) * 0.1)
        field.color_mlp.params.copy_(torch.randn(field.color_mlp.params.shape, generator=g) * 0.1)
    model_sd = dict(field.state_dict())
    model_sd["direction_encoding.params"] = torch.zeros(0)          # tcnn modules without parameters store an empty vector
    binary = torch.rand(res, res, res, generator=g) > 0.5
    occ_sd = {"_roi_aabb": torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), "resolution": torch.tensor([res] * 3, dtype=torch.int32),
              "occs": torch.rand(res ** 3, generator=g), "_binary": binary}
    state = {"step": 20000, "model": model_sd, "occupancy_grid": occ_sd, "optimizer": {}, "scheduler": {},
             "aabb": [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], "unbounded": False, "grid_resolution": res, "contraction_type": nc.ContractionType.AABB,
             "near_plane": None, "far_plane": None, "render_step_size": 0.005, "alpha_thre": 0.0, "cone_angle": 0.0,
             "camera_poses": torch.eye(4)[None, :3].repeat(3, 1, 1), "block_id": 2}
    return state, field, binary


def test_nerf_block_checkpoint_loads_through_the_manager(tmp_path):
    state, field, binary = _reference_style_block_state()
    path = os.path.join(str(tmp_path), "model.pth")
    torch.save(state, path)
    raw = open(path, "rb").read()
    assert b"nerfacc.contraction" in raw and b"ContractionType" in raw          # pickled by the module path nerfacc uses
    # the two-pass load of conerf/loss/confidence_loss.py:25-50
    meta = {k: None for k in ("aabb", "unbounded", "grid_resolution", "contraction_type", "render_step_size", "alpha_thre", "cone_angle", "camera_poses")}
    mgr = CheckPointManager(verbose=False)
    assert mgr.load_no_config(ckpt_path=path, meta_data=meta, map_location="cpu") == 20000
    assert isinstance(meta["contraction_type"], enum.Enum) and meta["contraction_type"].name == "AABB" and meta["grid_resolution"] == 8
    nerf = ngp.NGPradianceField(meta["aabb"], unbounded=meta["unbounded"])
    occ = ngp.OccupancyGrid(meta["aabb"], meta["grid_resolution"], meta["contraction_type"])
    mgr.load_no_config(ckpt_path=path, models={"model": nerf, "occupancy_grid": occ}, map_location="cpu")
    assert torch.equal(nerf.mlp_base.params, field.mlp_base.params) and torch.equal(nerf.color_mlp.params, field.color_mlp.params)
    assert torch.equal(occ.binary, binary) and occ.binary.dtype == torch.bool and torch.equal(occ.occs, state["occupancy_grid"]["occs"])
    # occupancy query: cell of a point, False outside the box
    pts = torch.tensor([[-1.49, -1.49, -1.49], [1.49, 1.49, 1.49], [2.0, 0.0, 0.0]])
    want = torch.stack([binary[0, 0, 0], binary[7, 7, 7], torch.tensor(False)])
    assert torch.equal(occ.query_occ(pts), want)
    # builds that also store the (non-persistent in 0.3.5) index buffers are accepted; unknown keys are not
    extra = dict(state["occupancy_grid"], grid_indices=torch.arange(8 ** 3), grid_coords=torch.zeros(8 ** 3, 3))
    ngp.OccupancyGrid(meta["aabb"], 8).load_state_dict(extra)
    with pytest.raises(RuntimeError):
        ngp.OccupancyGrid(meta["aabb"], 8).load_state_dict(dict(extra, something_else=torch.zeros(1)))
    with pytest.raises(RuntimeError):                                        # a resolution that does not match the file
        ngp.OccupancyGrid(meta["aabb"], 16).load_state_dict(state["occupancy_grid"])
    bad = dict(state["model"]); bad["mlp_head.params"] = torch.zeros(4)
    with pytest.raises(RuntimeError):
        ngp.NGPradianceField(meta["aabb"]).load_state_dict(bad)
    assert pickle.loads(pickle.dumps(meta["contraction_type"])) is meta["contraction_type"]

"""Checkpoint files of the registration / NeRF-block trainers: same public calls and the same files on disk as the reference's
manager (behaviour pinned by tests/golden/checkpoint_manager.json, recorded from the reference), own implementation.

What a run directory holds after ``save``:

    <dir>/model/model_000123.pth    one torch-pickled dict per saved step: 'step', then one entry per model / optimizer /
                                    scheduler name (their state_dict()) and per meta key (the value as given)
    <dir>/model.pth                 the newest of them
    <dir>/model_best.pth            the one with the best score so far (a later equal score wins; vector scores: every
                                    component must be at least as good)
    <dir>/checkpoints.txt           the step files still on disk, oldest first, one name per line, last line 'Best step: N'

Pruning keeps a sliding window of ``max_to_keep`` step files.  A file leaving the window survives only as a periodic
"milestone": the first one always does, afterwards one per ``keep_checkpoint_every_n_hours`` of wall-clock time.

Both kinds of reference checkpoint are read back through ``load_no_config``: RegTR training states and NeRF block states
(train_ngp_nerf.py:192-209: 'model', 'occupancy_grid' and the aabb / unbounded / grid_resolution / contraction_type / ... meta
keys).  The nerfacc enum pickled into the latter is resolved by ngp.install_pickle_shims()."""
import collections
import os
import shutil
import time

import numpy as np
import torch

_StepFile = collections.namedtuple("_StepFile", "path written_at")


def de_parallel(model):
    """Unwrap a data-parallel style wrapper (anything that carries the real network as ``.module``)."""
    inner = getattr(model, "module", None)
    return model if inner is None else inner


def _at_least_as_good(score, incumbent) -> bool:
    return incumbent is None or bool(np.all(np.asarray(score) >= np.asarray(incumbent)))


class _Window:
    """Which step files stay on disk: the newest `size` of them plus the milestones."""

    def __init__(self, size: int, milestone_hours: float):
        self.size = size
        self.period_s = milestone_hours * 3600.0
        self.recent = collections.deque()
        self.milestones = []
        self.milestone_due = time.time()          # anything written after this moment qualifies as the next milestone

    def admit(self, path: str):
        self.recent.append(_StepFile(path, time.time()))
        while len(self.recent) > self.size:
            old = self.recent.popleft()
            if old.written_at > self.milestone_due:
                self.milestones.append(old)
                self.milestone_due = old.written_at + self.period_s
            else:
                os.remove(old.path)

    def names(self):
        return [os.path.basename(f.path) for f in (*self.milestones, *self.recent)]


class CheckPointManager(object):
    def __init__(self, save_path: str = None, max_to_keep: int = 5, keep_checkpoint_every_n_hours: float = 10000.0,
                 verbose: bool = True) -> None:
        if max_to_keep < 1:
            raise ValueError(f"a checkpoint window of {max_to_keep} files keeps nothing: max_to_keep has to be 1 or more")
        self.verbose = verbose
        self.window = _Window(max_to_keep, keep_checkpoint_every_n_hours)
        self.best = {"score": None, "step": None}
        self.root = save_path
        if save_path is not None:
            os.makedirs(save_path, exist_ok=True)
            self._write_index()

    def set_save_path(self, path: str):
        self.root = path

    def _say(self, text: str):
        if self.verbose:
            print(text, flush=True)

    # ------------------------------------------------------------------ writing
    def _write_index(self):
        lines = self.window.names() + [f"Best step: {self.best['step']}"]
        if len(lines) == 1:
            lines.insert(0, "")               # an empty run still has its (empty) name line in front of the best-step line
        with open(os.path.join(self.root, "checkpoints.txt"), "w") as f:
            f.write("\n".join(lines))

    @staticmethod
    def _snapshot(step, models, optimizers, schedulers, meta_data) -> dict:
        snap = {"step": step}
        snap.update({name: de_parallel(net).state_dict() for name, net in models.items()})
        for group in (optimizers, schedulers):
            snap.update({name: obj.state_dict() for name, obj in (group or {}).items()})
        snap.update(meta_data or {})
        return snap

    def save(self, models: dict, optimizers: dict, step: int, schedulers: dict = None, meta_data: dict = None, score=0.0):
        assert self.root is not None, "save() needs a directory: construct the manager with save_path or call set_save_path()"
        step_dir = os.path.join(self.root, "model")
        os.makedirs(step_dir, exist_ok=True)
        target = os.path.join(step_dir, f"model_{step:06d}.pth")
        torch.save(self._snapshot(step, models, optimizers, schedulers, meta_data), target)
        self._say(f"checkpoint written: {target}")
        shutil.copyfile(target, os.path.join(self.root, "model.pth"))
        if _at_least_as_good(score, self.best["score"]):
            shutil.copyfile(target, os.path.join(self.root, "model_best.pth"))
            self.best = {"score": score, "step": step}
            self._say(f"  new best at step {step}: score {np.array2string(np.asarray(score), precision=3)}")
        self.window.admit(target)
        self._write_index()

    # ------------------------------------------------------------------ reading
    def _locate(self, ckpt_path: str):
        """The file to resume from: the explicit path when it exists, else the run directory's newest checkpoint."""
        if ckpt_path and os.path.exists(ckpt_path):
            return ckpt_path, "checkpoint"
        if self.root is not None and os.path.isdir(self.root):
            return os.path.join(self.root, "model.pth"), "latest checkpoint"
        return None, ""

    def load_no_config(self, ckpt_path: str, distributed: bool = False, local_rank: int = 0, models: dict = None,
                       optimizers: dict = None, schedulers: dict = None, meta_data: dict = None, map_location=None) -> int:
        """Restores every object passed in from ``ckpt_path`` (or, when that does not exist, from <save_path>/model.pth) and returns
        the stored step; 0 when there is nothing to resume from.  Asking for a name the file does not hold is a KeyError, a
        state_dict that does not fit raises from load_state_dict (strict).  ``meta_data``: its keys are looked up in the file and
        the values filled in.  ``map_location`` (an addition; default: cuda:{local_rank} when distributed, else wherever the tensors
        were saved from, CPU on a box without a GPU)."""
        source, kind = self._locate(ckpt_path)
        if source is None or not os.path.exists(source):
            self._say(f"nothing to resume from ({source or ckpt_path or 'no path'}): starting at step 0")
            return 0
        self._say(f"resuming from {kind} {source}")
        from .ngp import install_pickle_shims
        install_pickle_shims()
        if map_location is None:
            map_location = f"cuda:{local_rank}" if distributed else (None if torch.cuda.is_available() else "cpu")
        snap = torch.load(source, map_location=map_location, weights_only=False)
        for name, net in (models or {}).items():
            de_parallel(net).load_state_dict(snap[name])
        for group in (optimizers, schedulers):
            for name, obj in (group or {}).items():
                obj.load_state_dict(snap[name])
        if meta_data is not None:
            meta_data.update({key: snap[key] for key in list(meta_data)})
        self._say(f"restored {sorted((models or {}).keys())} at step {snap.get('step', 0)}")
        return snap.get("step", 0)

    def load(self, config, models: dict = None, optimizers: dict = None, schedulers: dict = None, meta_data: dict = None,
             map_location=None) -> int:
        """``config``: the parsed flags (ckpt_path, distributed, local_rank of dreg_nerf_amd/config.py)."""
        return self.load_no_config(getattr(config, "ckpt_path", "") or "", bool(getattr(config, "distributed", False)),
                                   int(getattr(config, "local_rank", 0)), models, optimizers, schedulers, meta_data, map_location)

"""Checkpoint manager with the reference's interface and on-disk layout (conerf/base/checkpoint_manager.py:13-222).

Files under ``save_path``:
  model/model_{step:06d}.pth   torch.save({'step', <model names>: state_dict, <optimizer names>, <scheduler names>, <meta keys>})
  model.pth                    copy of the most recent one            (:84)
  model_best.pth               copy of the best one by ``score``      (:88-91; score may be a vector: all components >=)
  checkpoints.txt              basenames of the kept files, one per line, then ``Best step: N``   (:109-115)
Retention (:98-107, the tf.Saver rule): at most ``max_to_keep`` recent files; a file that falls out of that window is kept
for good when it is newer than the next "keep every n hours" mark, deleted otherwise.

Both kinds of reference checkpoint go through ``load_no_config``: RegTR training states (train_nerf_regtr.py) and NeRF block
states (train_ngp_nerf.py:192-209: 'model' = NGPradianceField, 'occupancy_grid' = nerfacc OccupancyGrid, plus aabb / unbounded /
grid_resolution / contraction_type / render_step_size / alpha_thre / cone_angle / camera_poses [/ block_id]).  The nerfacc enum
pickled into the latter is resolved by ngp.install_pickle_shims(); ngp.NGPradianceField and ngp.OccupancyGrid take the
tiny-cuda-nn flat parameter vectors and the nerfacc buffers as they are."""
import os
import shutil
import time

import numpy as np
import torch


def de_parallel(model):
    """The wrapped module of a DistributedDataParallel-style wrapper (checkpoint_manager.py:9-10)."""
    return model.module if hasattr(model, "module") else model


class CheckPointManager(object):
    def __init__(self, save_path: str = None, max_to_keep: int = 5, keep_checkpoint_every_n_hours: float = 10000.0,
                 verbose: bool = True) -> None:
        if max_to_keep <= 0:
            raise ValueError("max_to_keep must be at least 1")
        self._max_to_keep = max_to_keep
        self._keep_checkpoint_every_n_hours = keep_checkpoint_every_n_hours
        self._verbose = verbose
        self._checkpoints_permanent = []   # (path, time) never deleted
        self._checkpoints_buffer = []      # (path, time) the max_to_keep most recent
        self._next_save_time = time.time()
        self._best_score = None
        self._best_step = None
        self._save_path = save_path
        self._checkpoints_fname = None
        if save_path is not None:
            os.makedirs(save_path, exist_ok=True)
            self._checkpoints_fname = os.path.join(save_path, "checkpoints.txt")
            self._update_checkpoints_file()

    def set_save_path(self, path: str):
        self._save_path = path

    # ------------------------------------------------------------------ save
    def _update_checkpoints_file(self):
        names = [os.path.basename(c[0]) for c in self._checkpoints_permanent + self._checkpoints_buffer]
        with open(self._checkpoints_fname, "w") as fid:
            fid.write("\n".join(names))
            fid.write("\nBest step: {}".format(self._best_step))

    def _remove_old_checkpoints(self):
        while len(self._checkpoints_buffer) > self._max_to_keep:
            path, stamp = self._checkpoints_buffer.pop(0)
            if stamp > self._next_save_time:
                self._checkpoints_permanent.append((path, stamp))
                self._next_save_time = stamp + self._keep_checkpoint_every_n_hours * 3600
            else:
                os.remove(path)

    def save(self, models: dict, optimizers: dict, step: int, schedulers: dict = None, meta_data: dict = None, score=0.0):
        if self._save_path is None:
            raise AssertionError("Checkpoint manager must be initialized with save path for save().")
        os.makedirs(os.path.join(self._save_path, "model"), exist_ok=True)
        name = os.path.join(self._save_path, "model", "model_{:06d}.pth".format(step))
        state = {"step": step}
        for key, m in models.items():
            state[key] = de_parallel(m).state_dict()
        for key, o in optimizers.items():
            state[key] = o.state_dict()
        for key, s in (schedulers or {}).items():
            state[key] = s.state_dict()
        for key, v in (meta_data or {}).items():
            state[key] = v
        if self._verbose:
            print(f"Saving checkpoint: {name}")
        torch.save(state, name)
        shutil.copy(name, os.path.join(self._save_path, "model.pth"))
        self._checkpoints_buffer.append((name, time.time()))
        if self._best_score is None or np.all(np.array(score) >= np.array(self._best_score)):
            shutil.copyfile(name, os.path.join(self._save_path, "model_best.pth"))
            self._best_score, self._best_step = score, step
            if self._verbose:
                print("Checkpoint is current best, score={}".format(np.array_str(np.array(score), precision=3)))
        self._remove_old_checkpoints()
        if self._checkpoints_fname is None:
            self._checkpoints_fname = os.path.join(self._save_path, "checkpoints.txt")
        self._update_checkpoints_file()

    # ------------------------------------------------------------------ load
    def load_no_config(self, ckpt_path: str, distributed: bool = False, local_rank: int = 0, models: dict = None,
                       optimizers: dict = None, schedulers: dict = None, meta_data: dict = None, map_location=None) -> int:
        """Restores whatever is passed from ``ckpt_path`` (or, when that does not exist, from save_path/model.pth) and returns
        the stored step; 0 when there is no checkpoint.  A name that is asked for but not in the file is a KeyError, a
        state_dict mismatch raises from load_state_dict (strict) — as in the reference.  ``map_location`` is an addition
        (default: cuda:{local_rank} when distributed, else the device the tensors were saved from, CPU if that is absent)."""
        name = None
        if ckpt_path and os.path.exists(ckpt_path):
            name = ckpt_path
            if self._verbose:
                print(f"[INFO] Resuming from checkpoint {name}...")
        elif self._save_path is not None and os.path.isdir(self._save_path):
            name = os.path.join(self._save_path, "model.pth")
            if self._verbose:
                print(f"[INFO] Resuming from latest checkpoint {name}...")
        if name is None or not os.path.exists(name):
            if self._verbose:
                print(f"[WARNING] Checkpoint {name} does not exist, training from scratch!")
            return 0
        from .ngp import install_pickle_shims
        install_pickle_shims()
        if map_location is None:
            map_location = f"cuda:{local_rank}" if distributed else (None if torch.cuda.is_available() else "cpu")
        state = torch.load(name, map_location=map_location, weights_only=False)
        step = state["step"] if "step" in state else 0
        for key, m in (models or {}).items():
            de_parallel(m).load_state_dict(state[key])
        for key, o in (optimizers or {}).items():
            o.load_state_dict(state[key])
        for key, s in (schedulers or {}).items():
            s.load_state_dict(state[key])
        if meta_data is not None:
            for key in meta_data.keys():
                meta_data[key] = state[key]
        if self._verbose:
            print(f"[INFO] Loaded models from {name}")
        return step

    def load(self, config, models: dict = None, optimizers: dict = None, schedulers: dict = None, meta_data: dict = None,
             map_location=None) -> int:
        """config carries ckpt_path / distributed / local_rank (conerf/utils/config.py; checkpoint_manager.py:198-222)."""
        return self.load_no_config(getattr(config, "ckpt_path", "") or "", bool(getattr(config, "distributed", False)),
                                   int(getattr(config, "local_rank", 0)), models, optimizers, schedulers, meta_data, map_location)

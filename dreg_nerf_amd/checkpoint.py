"""Checkpoint files in the reference's format (conerf/base/checkpoint_manager.py:51-196):
``out/<expname>/model/model_{step:06d}.pth`` = torch.save({'step', <model names>, <optimizer names>, <scheduler names>, meta...}),
latest copied to ``model.pth``, best to ``model_best.pth``, index in ``checkpoints.txt``, bounded retention."""
import os
import shutil

import torch


class CheckPointManager:
    def __init__(self, save_path=None, max_to_keep: int = 10, keep_checkpoint_every_n_hours: float = 10000.0, verbose: bool = True):
        self.save_path = save_path
        self.max_to_keep = max_to_keep
        self.verbose = verbose
        self.ckpts = []
        self.best_score = -float("inf")
        if save_path:
            os.makedirs(os.path.join(save_path, "model"), exist_ok=True)
            idx = os.path.join(save_path, "checkpoints.txt")
            if os.path.exists(idx):
                self.ckpts = [l.strip() for l in open(idx) if l.strip()]

    def save(self, step: int, models=None, optimizers=None, schedulers=None, meta_data=None, score: float = 0.0):
        state = {"step": step}
        for group in (models, optimizers, schedulers):
            for name, obj in (group or {}).items():
                state[name] = obj.state_dict()
        for k, v in (meta_data or {}).items():
            state[k] = v
        path = os.path.join(self.save_path, "model", f"model_{step:06d}.pth")
        torch.save(state, path)
        shutil.copy(path, os.path.join(self.save_path, "model.pth"))
        if score >= self.best_score:
            self.best_score = score
            shutil.copy(path, os.path.join(self.save_path, "model_best.pth"))
        self.ckpts.append(path)
        while len(self.ckpts) > self.max_to_keep:
            old = self.ckpts.pop(0)
            if os.path.exists(old):
                os.remove(old)
        with open(os.path.join(self.save_path, "checkpoints.txt"), "w") as f:
            f.write("\n".join(self.ckpts) + "\n")
        if self.verbose:
            print(f"saved checkpoint {path}", flush=True)

    def latest(self):
        p = os.path.join(self.save_path, "model.pth") if self.save_path else None
        return p if p and os.path.exists(p) else None

    def load(self, ckpt_path=None, models=None, optimizers=None, schedulers=None, meta_data=None, map_location="cpu") -> int:
        """Returns the stored step (0 when nothing is found).  strict=True for models, as the reference."""
        path = ckpt_path or self.latest()
        if not path or not os.path.exists(path):
            return 0
        from .ngp import install_pickle_shims
        install_pickle_shims()
        state = torch.load(path, map_location=map_location, weights_only=False)
        for name, m in (models or {}).items():
            if name in state:
                m.load_state_dict(state[name], strict=True)
        for group in (optimizers, schedulers):
            for name, o in (group or {}).items():
                if name in state:
                    o.load_state_dict(state[name])
        if meta_data is not None:
            for k in list(meta_data.keys()):
                if k in state:
                    meta_data[k] = state[k]
        if self.verbose:
            print(f"loaded checkpoint {path} (step {state.get('step', 0)})", flush=True)
        return int(state.get("step", 0))

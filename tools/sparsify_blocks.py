#!/usr/bin/env python3
"""Replace the dense grid files of extracted blocks by their lossless sparse cache: for every <root>/<dataset>/nerf_models/<scene pattern>/block_*/ with a
voxel_grid.pt, write voxel_sparse.pt (dataset.load_block_sparse: idx = voxel_mask.pt, vals = the grid's rows there; the grid is zero elsewhere,
eval_ngp_nerf.py:397-405) and delete voxel_grid.pt / density_voxel_grid.pt (58.7 MB each).  Training and evaluation with sparse=True read the cache.
usage: python tools/sparsify_blocks.py <root> <dataset> '<scene glob>'"""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreg_nerf_amd.dataset import load_block_sparse  # noqa: E402

root, dataset, pat = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "*"
n = 0
for d in sorted(glob.glob(os.path.join(root, dataset, "nerf_models", pat, "block_*"))):
    if os.path.exists(os.path.join(d, "voxel_grid.pt")):
        sb = load_block_sparse(d)          # writes voxel_sparse.pt next to the grid
        assert os.path.exists(os.path.join(d, "voxel_sparse.pt")) and sb.idx.numel() > 0
        for f in ("voxel_grid.pt", "density_voxel_grid.pt"):
            if os.path.exists(os.path.join(d, f)):
                os.remove(os.path.join(d, f))
        n += 1
print(f"{n} blocks sparsified under {root}/{dataset}/nerf_models/{pat}")

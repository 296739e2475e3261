"""Command-line flags of the registration entry points: names and defaults follow conerf/utils/config.py:4-146
(only the flags the registration path reads are kept; unknown reference flags are accepted and ignored)."""
import argparse


def config_parser(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--distributed", action="store_true")
    p.add_argument("--seed", type=int, default=3407)
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--max_iterations", type=int, default=20000)
    p.add_argument("--num_process", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--finetune", action="store_true")
    p.add_argument("--dataset", type=str, default="")
    p.add_argument("--json_dir", type=str, default="")
    p.add_argument("--data_split_json", type=str, default="")
    p.add_argument("--factor", type=int, default=4)
    p.add_argument("--train_split", type=str, default="trainval")
    p.add_argument("--root_dir", type=str, default="")
    p.add_argument("--scene", type=str, default="")
    p.add_argument("--expname", type=str, default="chair_reg")
    p.add_argument("--aabb", type=lambda s: [float(v) for v in s.split(",")], default="-1.5,-1.5,-1.5,1.5,1.5,1.5")
    p.add_argument("--test_chunk_size", type=int, default=8192)
    p.add_argument("--unbounded", action="store_true")
    p.add_argument("--multi_blocks", action="store_true")
    p.add_argument("--position_embedding_type", type=str, default="sine")
    p.add_argument("--position_embedding_dim", type=int, default=256)
    p.add_argument("--position_embedding_scaling", type=float, default=1.0)
    p.add_argument("--num_downsample", type=int, default=6)
    p.add_argument("--robust_loss", action="store_true")
    p.add_argument("--ckpt_path", type=str, default="")
    p.add_argument("--no_load_opt", action="store_true")
    p.add_argument("--no_load_scheduler", action="store_true")
    p.add_argument("--enable_tensorboard", action="store_true")
    p.add_argument("--enable_visdom", action="store_true")
    p.add_argument("--n_tensorboard", type=int, default=30)
    p.add_argument("--n_validation", type=int, default=2500)
    p.add_argument("--n_checkpoint", type=int, default=5000)
    # build-side additions
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--pairs_per_step", type=int, default=1, help="pairs per optimizer step and GPU (reference: 1)")
    p.add_argument("--synthetic", type=int, default=0, help="use N synthetic shell-R scenes instead of a dataset on disk")
    p.add_argument("--synthetic_res", type=int, default=128)
    p.add_argument("--dump_outputs", action="store_true",
                   help="eval: per scene, transformation_est.json and the registration's point clouds as PLY files (eval_nerf_regtr.py:313-438)")
    p.add_argument("--fgr_baseline", action="store_true",
                   help="also run the Fast Global Registration baseline on every pair and write fgr_metrics_{split}.json (eval_nerf_regtr.py:303-311 of the reference)")
    p.add_argument("--eval_batch", type=int, default=4, help="eval: pairs per forward call (reference: 1; results per scene do not depend on it beyond bf16 rounding)")
    p.add_argument("--extract_grids", action="store_true",
                   help="eval_nerf_regtr.py: extract the voxel grids of the split's NeRF blocks first (what eval_ngp_nerf.py does, same files) and register from the device-resident grids, pipelined")
    args, _unknown = p.parse_known_args(argv)
    if isinstance(args.aabb, str):
        args.aabb = [float(v) for v in args.aabb.split(",")]
    return args

#!/bin/bash
# Trained-regime record (review item 3 of round 5): generate the object split, extract the TRAINING split's grids (eval_ngp_nerf.py), train the bf16 product
# step at 128^3 with labels marched from the blocks (train_nerf_regtr.py), then tests/test_hip_trained_regime.py evaluates the held-out scenes through the
# pipelined chain (eval_nerf_regtr.py --extract_grids) in bf16 / fp32 mode / CPU oracle and writes gpurun_out/r06_trained_eval.json.
#   tools/trained_regime.sh <epochs> [train scenes] [test scenes] [bound on the mean RRE in degrees]
EPOCHS=${1:-300}; NTRAIN=${2:-128}; NTEST=${3:-16}; BOUND=${4:-1.0}
R=/dev/shm/objsplit; J=$R/json
rm -rf $R
( time python tools/make_object_split.py --root $R --train $NTRAIN --test $NTEST ) 2>&1 | tail -5
( time python eval_ngp_nerf.py --root_dir $R --dataset objaverse --multi_blocks | tail -1 ) 2>&1 | tail -5
# (the test split's grids are removed again: the evaluation extracts them itself, pipelined with the registration)
for d in $R/objaverse/nerf_models/obj_test_*/block_*; do rm -f $d/voxel_* $d/density_voxel_*; done
python train_nerf_regtr.py --root_dir $R --json_dir $J --dataset objaverse --expname objreg --pairs_per_step 4 --epochs $EPOCHS \
    --n_validation 5000 --n_tensorboard 1000 --n_checkpoint 1000000 2>&1 | grep -v "^resuming\|^restored\|checkpoint written\|^epoch" | tail -120
grep "^epoch" $R/out/objreg/log.txt | tail -2
DREG_TRAINED_ROOT=$R DREG_TRAINED_RRE_BOUND=$BOUND python -m pytest tests/test_hip_trained_regime.py -x -q 2>&1 | tail -15
cat gpurun_out/r06_trained_eval.json 2>/dev/null | head -60

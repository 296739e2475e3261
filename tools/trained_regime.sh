#!/bin/bash
# Trained-regime record (review item 3 of round 5): generate the object split, extract the TRAINING split's grids (eval_ngp_nerf.py), train the bf16 product
# step at 128^3 with labels marched from the blocks (train_nerf_regtr.py), then tests/test_hip_trained_regime.py evaluates the held-out scenes through the
# pipelined chain (eval_nerf_regtr.py --extract_grids) in bf16 / fp32 mode / CPU oracle and writes gpurun_out/r06_trained_eval.json.
#   tools/trained_regime.sh <epochs> [train scenes] [test scenes] [bound on the mean RRE in degrees]
EPOCHS=${1:-300}; NTRAIN=${2:-128}; NTEST=${3:-16}; BOUND=${4:-1.0}
R=/dev/shm/objsplit; J=$R/json
# MEMORY: a block is 60 MB of checkpoint + 2 x 58.7 MB of dense grid files in RAM-backed /dev/shm = 177 MB; 536 scenes (190 GB) ran, 2,096 scenes took the box down.
if [ $((NTRAIN + NTEST)) -gt 600 ] && [ "$DREG_BIG_SPLIT" != "1" ]; then echo "refusing $((NTRAIN + NTEST)) scenes (> 600): ~$(( (NTRAIN + NTEST) * 354 / 1000 )) GB of /dev/shm"; exit 2; fi
rm -rf $R
( time python tools/make_object_split.py --root $R --train $NTRAIN --test $NTEST ) 2>&1 | tail -5
( time python eval_ngp_nerf.py --root_dir $R --dataset objaverse --multi_blocks | tail -1 ) 2>&1 | tail -5
# the density-only twins of the grids are not read by training or evaluation: 58.7 MB per block back
find $R/objaverse/nerf_models -name "density_voxel_grid.pt" -delete; df -h /dev/shm | tail -1
python train_nerf_regtr.py --root_dir $R --json_dir $J --dataset objaverse --expname objreg --pairs_per_step 4 --epochs $EPOCHS \
    --n_validation 4000 --n_tensorboard 1000 --n_checkpoint 4000 2>&1 | grep -v "^resuming\|^restored\|checkpoint written\|^epoch\|new best" | tail -120
grep "^epoch" $R/out/objreg/log.txt | tail -2
# (the test split's grids — the trainer's validation read them — are removed: the evaluation extracts them itself, pipelined with the registration)
for d in $R/objaverse/nerf_models/obj_test_*/block_*; do rm -f $d/voxel_* $d/density_voxel_*; done
DREG_TRAINED_ROOT=$R DREG_TRAINED_RRE_BOUND=$BOUND python -m pytest tests/test_hip_trained_regime.py -x -q 2>&1 | tail -15
cat gpurun_out/r06_trained_eval.json 2>/dev/null | head -60

#!/bin/bash
# Trained-regime record (review item 3 of round 5): generate the object split, extract the TRAINING split's grids (eval_ngp_nerf.py), train the bf16 product
# step at 128^3 with labels marched from the blocks (train_nerf_regtr.py), then tests/test_hip_trained_regime.py evaluates the held-out scenes through the
# pipelined chain (eval_nerf_regtr.py --extract_grids) in bf16 / fp32 mode / CPU oracle and writes gpurun_out/r06_trained_eval.json.
#   tools/trained_regime.sh <epochs> [train scenes] [test scenes] [bound on the mean RRE in degrees]
EPOCHS=${1:-300}; NTRAIN=${2:-128}; NTEST=${3:-16}; BOUND=${4:-1.0}
R=/dev/shm/objsplit; J=$R/json
# MEMORY (RAM-backed /dev/shm): a block is 60 MB of checkpoint + 2 x 58.7 MB of dense grid files.  The grids are extracted in chunks of 100 scenes and replaced by their
# lossless sparse cache (tools/sparsify_blocks.py: ~1 MB) right away: 61 MB per block + 23 GB transient.  190 GB ran; 742 GB took the box down: refuse > 1,300 scenes.
if [ $((NTRAIN + NTEST)) -gt 1300 ]; then echo "refusing $((NTRAIN + NTEST)) scenes (> 1300): ~$(( (NTRAIN + NTEST) * 122 / 1000 + 23 )) GB of /dev/shm"; exit 2; fi
rm -rf $R
( time python tools/make_object_split.py --root $R --train $NTRAIN --test $NTEST ) 2>&1 | tail -5
t0=$(date +%s)
for sp in train test; do
  n=$NTRAIN; [ $sp = test ] && n=$NTEST
  for ((h = 0; h * 100 < n; h++)); do
    pat=$(printf "obj_%s_%02d??" $sp $h)
    python eval_ngp_nerf.py --root_dir $R --dataset objaverse --multi_blocks --scene "$pat" | tail -1
    python tools/sparsify_blocks.py $R objaverse "$pat"
  done
done
echo "extraction + sparse caches: $(( $(date +%s) - t0 )) s"; df -h /dev/shm | tail -1
python train_nerf_regtr.py --root_dir $R --json_dir $J --dataset objaverse --expname objreg --pairs_per_step 4 --epochs $EPOCHS \
    --n_validation 4000 --n_tensorboard 1000 --n_checkpoint 4000 2>&1 | grep -v "^resuming\|^restored\|checkpoint written\|^epoch\|new best" | tail -120
grep "^epoch" $R/out/objreg/log.txt | tail -2
# (the test split's grids — the trainer's validation read them — are removed: the evaluation extracts them itself, pipelined with the registration)
for d in $R/objaverse/nerf_models/obj_test_*/block_*; do rm -f $d/voxel_* $d/density_voxel_*; done
DREG_TRAINED_ROOT=$R DREG_TRAINED_RRE_BOUND=$BOUND python -m pytest tests/test_hip_trained_regime.py -x -q 2>&1 | tail -15
cat gpurun_out/r06_trained_eval.json 2>/dev/null | head -60

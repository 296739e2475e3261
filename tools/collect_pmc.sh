# HBM traffic counters of the bench's kernels (separate passes per the guide: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass;
# --pmc only with --kernel-trace).  Writes gpurun_out/pmc/hbm_per_launch.json (copy it to profiles/pmc_hbm_per_launch.json: bench.py
# reads it from there and checks the recorded kernel-source sha against the tree it runs in).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for mode in active_set dense_head; do
  flag=""; [ $mode = dense_head ] && flag="--dense-head"
  for c in FETCH_SIZE WRITE_SIZE; do
    DREG_SERIAL_STREAMS=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/${mode}_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference $flag > gpurun_out/pmc/${mode}_$c.log 2>&1
  done
done
# BASELINE config 4 (bench.py --ngp): the two NGP kernels
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/ngp_$c -o p -- python bench.py --ngp --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc/ngp_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc/hbm_per_launch.json ngp=gpurun_out/pmc/ngp_FETCH_SIZE ngp=gpurun_out/pmc/ngp_WRITE_SIZE active_set=gpurun_out/pmc/active_set_FETCH_SIZE active_set=gpurun_out/pmc/active_set_WRITE_SIZE dense_head=gpurun_out/pmc/dense_head_FETCH_SIZE dense_head=gpurun_out/pmc/dense_head_WRITE_SIZE
find gpurun_out/pmc -name "*.csv" -size +2M -delete
du -sh gpurun_out/pmc

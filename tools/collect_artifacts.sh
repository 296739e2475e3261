set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/art
timeout 900 python bench.py --kernel-report gpurun_out/art/bench_event_timing_by_shape.tsv > gpurun_out/art/bench.log 2>&1; tail -1 gpurun_out/art/bench.log > gpurun_out/art/bench.json
for mode in active_set dense_head; do
  flag=""; [ $mode = dense_head ] && flag="--dense-head"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/two_$mode -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference $flag > gpurun_out/art/two_$mode.log 2>&1
  DREG_SERIAL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/ser_$mode -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference $flag > gpurun_out/art/ser_$mode.log 2>&1
done
find gpurun_out/art -name "*kernel_stats.csv" | head; du -sh gpurun_out/art
# per-queue timeline of one steady-state step (launch counts per queue, gaps) from the traces, then keep only the stats csv (traces are large)
for m in two_active_set ser_active_set; do
  f=$(find gpurun_out/art/$m -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f > gpurun_out/art/timeline_$m.txt 2>&1
done
find gpurun_out/art -name "*kernel_trace.csv" -delete; find gpurun_out/art -name "*.db" -delete
du -sh gpurun_out/art

# copies what tools/collect_round6.sh left under gpurun_out/ into profiles/r06_*
R=r06
cd "$(dirname "$0")/.."
cp gpurun_out/art/bench.json profiles/${R}_bench.json; cp gpurun_out/art/bench_event_timing_by_shape.tsv profiles/${R}_bench_event_timing_by_shape.tsv
for m in active_set dense_head; do
  cp "$(find gpurun_out/art/two_$m -name '*kernel_stats.csv' | head -1)" profiles/${R}_rocprofv3_kernel_stats_$m.csv
  cp "$(find gpurun_out/art/ser_$m -name '*kernel_stats.csv' | head -1)" profiles/${R}_rocprofv3_kernel_stats_${m}_serial_streams.csv
done
cp gpurun_out/pmc/hbm_per_launch.json profiles/pmc_hbm_per_launch.json; cp gpurun_out/pmc/hbm_per_launch.json profiles/${R}_pmc_hbm_per_launch.json
cp gpurun_out/round/bench_ngp.json profiles/${R}_bench_ngp_config4.json
cp gpurun_out/round/bench_chain.json profiles/${R}_bench_chain_config5.json
cp gpurun_out/round/bench_nerf_labels.json profiles/${R}_bench_nerf_labels.json; cp gpurun_out/round/bench_nerf_labels_unsplit.json profiles/${R}_bench_nerf_labels_unsplit.json
cp gpurun_out/round/bench_eval.json profiles/${R}_bench_eval_forward.json
cp "$(find gpurun_out/round/chain_trace -name '*kernel_stats.csv' | head -1)" profiles/${R}_rocprofv3_kernel_stats_chain_config5.csv
grep -v "amdgpu.ids" gpurun_out/round/phase_times_steady_state.txt > profiles/${R}_phase_times_steady_state.txt
for m in two ser; do cp gpurun_out/art/timeline_${m}_active_set.txt profiles/${R}_timeline_per_queue_${m}_streams_under_profiler.txt; done
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/pmc_hbm_per_launch.json"))
print("PMC file collected on kernel sources", d.get("kernel_source_sha"), "- tree:", bench.kernel_source_sha())
PY

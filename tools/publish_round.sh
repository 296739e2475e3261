# copies what tools/collect_round.sh left under gpurun_out/ into profiles/<R>_* (R = round tag, e.g. r03):  bash tools/publish_round.sh r03
R=${1:?round tag}
cd "$(dirname "$0")/.."
cp gpurun_out/art/bench.json profiles/${R}_bench.json; cp gpurun_out/art/bench_event_timing_by_shape.tsv profiles/${R}_bench_event_timing_by_shape.tsv
for m in active_set dense_head; do
  cp "$(find gpurun_out/art/two_$m -name '*kernel_stats.csv' | head -1)" profiles/${R}_rocprofv3_kernel_stats_$m.csv
  cp "$(find gpurun_out/art/ser_$m -name '*kernel_stats.csv' | head -1)" profiles/${R}_rocprofv3_kernel_stats_${m}_serial_streams.csv
done
cp gpurun_out/pmc/hbm_per_launch.json profiles/pmc_hbm_per_launch.json; cp gpurun_out/pmc/hbm_per_launch.json profiles/${R}_pmc_hbm_per_launch.json
cp gpurun_out/round/bench_ngp.json profiles/${R}_bench_ngp_config4.json; cp gpurun_out/round/bench_occupancy_sweep.json profiles/${R}_bench_occupancy_sweep.json
(echo '# tools/pmc_kernel_clock.sh "python tools/bench_wgrad.py" conv_'; cat gpurun_out/round/pmc_kernel_clock_wgrad.txt; echo '# tools/pmc_kernel_clock.sh "python tools/bench_igemm_ap.py" conv_igemm'; cat gpurun_out/round/pmc_kernel_clock_igemm.txt) > profiles/${R}_pmc_kernel_clock_wgrad_igemm.txt
cp gpurun_out/round/wgrad_phase_probe.txt profiles/${R}_wgrad_phase_probe.txt; cp gpurun_out/round/bench_wgrad.txt profiles/${R}_bench_wgrad.txt
cp gpurun_out/round/bench_igemm_ap.txt profiles/${R}_bench_igemm_ap.txt; cp gpurun_out/round/repack_bubble.txt profiles/${R}_repack_bubble.txt
cp gpurun_out/round/bench_conv_brick.txt profiles/${R}_bench_conv_brick.txt
for f in phase_times_steady_state halo_ablation halo_data_power ab_switches; do [ -f gpurun_out/round/$f.txt ] && grep -v "amdgpu.ids" gpurun_out/round/$f.txt > profiles/${R}_$f.txt; done
for m in two ser; do cp gpurun_out/art/timeline_${m}_active_set.txt profiles/${R}_timeline_per_queue_${m}_streams_under_profiler.txt; done
[ -f gpurun_out/pinned_step_report.json ] && cp gpurun_out/pinned_step_report.json profiles/${R}_pinned_step_report.json
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/pmc_hbm_per_launch.json"))
print("PMC file collected on kernel sources", d.get("kernel_source_sha"), "- tree:", bench.kernel_source_sha())
PY

# Everything profiles/rNN_* is made from, in ONE gpurun call on ONE box (so that bench.json, the kernel stats and the PMC traffic belong
# together):  bash tools/collect_round.sh   ->  gpurun_out/{pmc,art,round}/...   (then copy what is to be judged into profiles/)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round
bash tools/collect_pmc.sh > gpurun_out/round/collect_pmc.log 2>&1
cp gpurun_out/pmc/hbm_per_launch.json profiles/pmc_hbm_per_launch.json      # bench.py reads roofline.traffic from here (keyed by the kernel-source sha)
bash tools/collect_artifacts.sh > gpurun_out/round/collect_artifacts.log 2>&1
timeout 600 python bench.py --ngp > gpurun_out/round/bench_ngp.log 2>&1; tail -1 gpurun_out/round/bench_ngp.log > gpurun_out/round/bench_ngp.json
timeout 900 python bench.py --occupancy-sweep --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference > gpurun_out/round/occ.log 2>&1; tail -1 gpurun_out/round/occ.log > gpurun_out/round/bench_occupancy_sweep.json
timeout 900 bash tools/pmc_kernel_clock.sh "python tools/bench_wgrad.py" conv_ > gpurun_out/round/pmc_kernel_clock_wgrad.txt 2>&1
timeout 900 bash tools/pmc_kernel_clock.sh "python tools/bench_igemm_ap.py" conv_igemm > gpurun_out/round/pmc_kernel_clock_igemm.txt 2>&1
timeout 600 python tools/wgrad_phase_probe.py > gpurun_out/round/wgrad_phase_probe.txt 2>&1
timeout 600 python tools/bench_wgrad.py 8 > gpurun_out/round/bench_wgrad.txt 2>&1
timeout 600 python tools/bench_igemm_ap.py > gpurun_out/round/bench_igemm_ap.txt 2>&1
timeout 600 python tools/bench_conv_brick.py > gpurun_out/round/bench_conv_brick.txt 2>&1
timeout 300 python tools/repack_bubble.py > gpurun_out/round/repack_bubble.txt 2>&1
timeout 300 python tools/phase_times.py --nosync > gpurun_out/round/phase_times_steady_state.txt 2>&1
timeout 600 python tools/bench_conv_halo.py --ablate > gpurun_out/round/halo_ablation.txt 2>&1
timeout 300 python tools/halo_data_power.py > gpurun_out/round/halo_data_power.txt 2>&1
(bash tools/ab_step.sh opt:sparse_stem 0 1; bash tools/ab_step.sh opt:sparse_grads 0 1; bash tools/ab_step.sh opt:s2_accumulate 0 1; bash tools/ab_step.sh opt:fold_res_bn 0 1; bash tools/ab_step.sh ps:group_wgrad 0 1; bash tools/ab_step.sh opt:group_wgrad 0 1) > gpurun_out/round/ab_switches.txt 2>&1
tail -2 gpurun_out/art/bench.log | cut -c1-600
ls -la gpurun_out/round gpurun_out/art

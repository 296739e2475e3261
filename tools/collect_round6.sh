# Round 6, ONE gpurun call on ONE box: PMC traffic (bench.py's roofline.traffic), the default line + kernel stats (two-stream and serial, both head modes),
# the config-4 line, the config-5 chain (pipelined + the serial chain beside it), the step with labels from NeRF blocks, phase times.  The kernels of the
# step are unchanged against round 5: the kernel-level probes of tools/collect_round.sh are not repeated.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round
bash tools/collect_pmc.sh > gpurun_out/round/collect_pmc.log 2>&1
cp gpurun_out/pmc/hbm_per_launch.json profiles/pmc_hbm_per_launch.json      # bench.py reads roofline.traffic from here (keyed by the kernel-source sha)
bash tools/collect_artifacts.sh > gpurun_out/round/collect_artifacts.log 2>&1
timeout 600 python bench.py --ngp > gpurun_out/round/bench_ngp.log 2>&1; tail -1 gpurun_out/round/bench_ngp.log > gpurun_out/round/bench_ngp.json
timeout 900 python bench.py --chain --chain-serial > gpurun_out/round/bench_chain.log 2>&1; tail -1 gpurun_out/round/bench_chain.log > gpurun_out/round/bench_chain.json
timeout 600 python bench.py --nerf-labels > gpurun_out/round/bench_nerf_labels.log 2>&1; tail -1 gpurun_out/round/bench_nerf_labels.log > gpurun_out/round/bench_nerf_labels.json
DREG_SPLIT_LABELS=0 timeout 600 python bench.py --nerf-labels > gpurun_out/round/bench_nerf_labels_unsplit.log 2>&1; tail -1 gpurun_out/round/bench_nerf_labels_unsplit.log > gpurun_out/round/bench_nerf_labels_unsplit.json
timeout 600 python bench.py --eval > gpurun_out/round/bench_eval.log 2>&1; tail -1 gpurun_out/round/bench_eval.log > gpurun_out/round/bench_eval.json
timeout 300 python tools/phase_times.py --nosync > gpurun_out/round/phase_times_steady_state.txt 2>&1
# the chain under the kernel trace: what the GPU runs per block / per batch of pairs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/round/chain_trace -o k -- python bench.py --chain > gpurun_out/round/chain_trace.log 2>&1
find gpurun_out/round/chain_trace -name "*kernel_trace.csv" -delete; find gpurun_out/round/chain_trace -name "*.db" -delete
tail -2 gpurun_out/art/bench.log | cut -c1-800
ls -la gpurun_out/round gpurun_out/art

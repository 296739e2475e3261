cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/job
timeout 1500 python -m pytest tests/test_hip_bench_labels.py tests/test_hip_pointset_exec.py tests/test_hip_trunk_exec.py tests/test_hip_pinned_step.py tests/test_hip_regtr.py tests/test_hip_conv_halo.py -x -q 2>&1 | tail -12 | tee gpurun_out/job/t.txt
timeout 600 python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference 2>/dev/null | tail -1 > gpurun_out/job/default.json; cut -c1-200 gpurun_out/job/default.json

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/brick
timeout 900 python -m pytest tests/test_hip_conv_brick.py -x -q 2>&1 | tail -25 > gpurun_out/brick/pytest.txt; cat gpurun_out/brick/pytest.txt
timeout 600 python tools/bench_conv_brick.py > gpurun_out/brick/bench_conv_brick.txt 2>&1; cat gpurun_out/brick/bench_conv_brick.txt | tail -12

# usage: bash tools/gpu_run.sh "<pytest args>" [bench args...]   (one gpurun call: tests, then a short bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run
timeout 1500 python -m pytest $1 -x -q 2>&1 | tail -25 > gpurun_out/run/pytest.txt
cat gpurun_out/run/pytest.txt
shift
timeout 600 python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference "$@" > gpurun_out/run/bench.log 2>&1; tail -1 gpurun_out/run/bench.log | cut -c1-330
timeout 300 python tools/host_issue_time.py 2>&1 | grep -E "^issue|host time" | cut -c1-300

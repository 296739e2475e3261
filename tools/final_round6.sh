# one gpurun call: the whole GPU suite, then everything profiles/r06_* is made from (tools/collect_round6.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest_all.txt 2>&1; tail -3 gpurun_out/final/pytest_all.txt
bash tools/collect_round6.sh > gpurun_out/final/collect.log 2>&1
tail -3 gpurun_out/final/collect.log | cut -c1-400

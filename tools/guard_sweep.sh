# Round-5 review item 3: guard-band sweep + 200 repetitions of the sparse-stem tests under AMD_SERIALIZE_KERNEL=3, everything kept.
# usage (one gpurun call): bash tools/guard_sweep.sh [reps of the pytest file, default 200]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/guard
export AMD_SERIALIZE_KERNEL=3
timeout 2400 python tools/guard_sweep.py > gpurun_out/guard/guard_sweep.txt 2>&1; echo "guard_sweep.py exit $?" >> gpurun_out/guard/guard_sweep.txt
tail -6 gpurun_out/guard/guard_sweep.txt
N=${1:-200}; fail=0
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_hip_sparse_stem.py -x -q -m gpu > gpurun_out/guard/pytest_last.txt 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); cp gpurun_out/guard/pytest_last.txt gpurun_out/guard/pytest_fail_$i.txt; echo "run $i: exit $rc" >> gpurun_out/guard/pytest_loop.txt; fi
done
echo "tests/test_hip_sparse_stem.py x $N under AMD_SERIALIZE_KERNEL=3 (torch reference on the GPU): $fail failing runs" >> gpurun_out/guard/pytest_loop.txt
tail -3 gpurun_out/guard/pytest_last.txt >> gpurun_out/guard/pytest_loop.txt
cat gpurun_out/guard/pytest_loop.txt

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p1
timeout 300 python tools/aten_callsites.py > gpurun_out/p1/aten_callsites.txt 2>&1
timeout 300 python tools/host_issue_time.py > gpurun_out/p1/host_issue.txt 2>&1
timeout 300 python tools/aten_launches.py > gpurun_out/p1/aten_launches.txt 2>&1
tail -5 gpurun_out/p1/host_issue.txt | head -3

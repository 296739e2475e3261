cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 900 python tools/ab_step.py "$@" 2>&1 | tail -6 | tee -a gpurun_out/ab/ab.txt

#!/bin/bash
# PMC passes over the whole training step (bench.py, 3 steps, one stream): per-kernel averages of the SQ counters that tell what a
# kernel waits for.  usage: tools/pmc_step_kernels.sh "substring of kernel names" (default: attn_)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=${1:-attn_}
OUT=$R/gpurun_out/pmc_step
mkdir -p $OUT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  DREG_SERIAL_STREAMS=1 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, os, collections
out = "$OUT"; pat = "$PAT"
res = collections.defaultdict(dict)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
for k, cs in sorted(res.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {v:16.0f}")
PY

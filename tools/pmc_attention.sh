#!/bin/bash
# SQ counters of the attention kernels on tools/bench_attention.py (one counter group per pass, --kernel-trace only: gpurun refuses more)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $R/tools/bench_attention.py > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, os, collections
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "attn" in k:
                print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +1M -delete

#!/bin/bash
# PMC passes over the halo-convolution microbenchmark (one counter group per pass, --kernel-trace only: gpurun refuses more)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_halo
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $R/tools/bench_conv_halo.py 8 64 256 --only-halo > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, os, collections
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "halo" in k or "igemm" in k:
                print(os.path.basename(os.path.dirname(d)), k, {c: (sum(v) / len(v), len(v)) for c, v in cs.items()})
PY

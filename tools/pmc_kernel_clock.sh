#!/bin/bash
# Sustained clock and MFMA-busy share of the kernels of a command: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 against the
# un-instrumented duration (rocprofv3 --stats), SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against the active cycles.
# usage: tools/pmc_kernel_clock.sh "python tools/bench_wgrad.py" conv_wgrad
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD=$1; PAT=${2:-conv_}
OUT=$R/gpurun_out/pmc_clock
rm -rf $OUT; mkdir -p $OUT
(cd $R && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o p -- $CMD > $OUT/pmc.log 2>&1)
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $CMD > $OUT/stats.log 2>&1)
python - <<PY
import csv, glob, collections
out = "$OUT"; pat = "$PAT"
dur = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]: dur[r["Name"].split("(")[0]] = (float(r["AverageNs"]) * 1e-9, int(r["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, (d, n) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k not in acc: continue
    gui = sum(acc[k]["GRBM_GUI_ACTIVE"]) / len(acc[k]["GRBM_GUI_ACTIVE"]) / 8
    mfma = sum(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024
    print(f"{k[:84]:84s} {n:5d} x {d * 1e6:9.1f} us  {gui / d / 1e9:5.2f} GHz  MFMA busy {100 * mfma / gui:5.1f} % of the active cycles")
PY
find $OUT -name "*.csv" -size +1M -delete

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
# the sustained clock under the kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, and MFMA-busy share of those cycles
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python $R/tools/bench_conv_halo.py 8 64 256 --only-halo > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
out = "$OUT"
dur = None
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3_halo_kernel" in r["Name"]: dur = float(r["AverageNs"]) * 1e-9
acc = {}
for f in glob.glob(out + "/SQ_VALU_MFMA_BUSY*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3_halo_kernel" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if dur and acc:
    gui = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
    mfma = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024
    flops = 2.0 * 8 * 64 ** 3 * 256 * 256 * 27
    clk = gui / dur
    print(f"conv3_halo_kernel 8 x 64^3 256 -> 256: {dur * 1e3:.3f} ms (un-instrumented) = {flops / dur / 1e15:.3f} PFLOP/s; {gui:.3e} active cycles per XCD = {clk / 1e9:.2f} GHz sustained; "
          f"MFMA busy {mfma:.3e} cycles per SIMD = {100 * mfma / gui:.0f} % of them; MFMA roof at this clock {2.5 * clk / 2.4e9:.2f} PFLOP/s (nameplate 2.5 at 2.4 GHz): the kernel runs at {100 * flops / dur / (2.5e15 * clk / 2.4e9):.0f} % of it")
PY

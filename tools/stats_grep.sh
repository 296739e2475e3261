# usage: bash tools/stats_grep.sh <pattern,...> [two]   — rocprofv3 kernel stats of 8 bench steps (serial streams; "two": the production two-stream mode), rows matching a pattern
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sg; rm -rf $out; mkdir -p $out
ser=1; [ "$2" = two ] && ser=0
DREG_SERIAL_STREAMS=$ser timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference --steps 8 --warmup 3 > $out/log.txt 2>&1
f=$(find $out/p -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(p in r["Name"] for p in "$1".split(",")): print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
rm -rf $out/p

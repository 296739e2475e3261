cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/base
timeout 600 python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference > gpurun_out/base/bench.log 2>&1; tail -1 gpurun_out/base/bench.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/base/two -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference --steps 6 --warmup 3 > gpurun_out/base/two.log 2>&1
f=$(find gpurun_out/base/two -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f > gpurun_out/base/timeline_two.txt 2>&1
DREG_SERIAL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/base/ser -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference --steps 6 --warmup 3 > gpurun_out/base/ser.log 2>&1
f2=$(find gpurun_out/base/ser -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f2 > gpurun_out/base/timeline_ser.txt 2>&1
# keep one step of the serial trace in order (name, start, dur, queue) for per-layer study
python - <<PY
import csv,sys
rows=list(csv.DictReader(open("$f2")))
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
ad=[r["e"] for r in rows if r["Kernel_Name"].startswith("adamw_kernel")]
t0,t1=ad[-3],ad[-2]
qk="Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
with open("gpurun_out/base/step_ser.tsv","w") as f:
    for r in rows:
        if r["s"]>=t0 and r["e"]<=t1:
            f.write(f'{r["s"]-t0}\t{r["e"]-r["s"]}\t{r[qk]}\t{r.get("Grid_Size_X","")}\t{r.get("Workgroup_Size_X","")}\t{r["Kernel_Name"][:160]}\n')
PY
find gpurun_out/base -name "*kernel_trace.csv" -delete; find gpurun_out/base -name "*.db" -delete
du -sh gpurun_out/base

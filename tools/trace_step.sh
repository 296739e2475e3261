# one gpurun call: serial-stream and two-stream kernel traces of the bench, one steady-state step of each as a table + tools/timeline.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-trace}; mkdir -p $out
for mode in ser two; do
  env_=""; [ $mode = ser ] && env_="DREG_SERIAL_STREAMS=1"
  env $env_ timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode -o k -- python bench.py --no-cpu-baseline --no-dense-reference --no-nerf-labels-reference --no-ngp-reference --steps 8 --warmup 3 > $out/$mode.log 2>&1
  f=$(find $out/$mode -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f > $out/timeline_$mode.txt 2>&1
  python - <<PY
import csv,re
rows=list(csv.DictReader(open("$f")))
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
ad=[r["e"] for r in rows if r["Kernel_Name"].startswith("adamw_kernel")]
t0,t1=ad[-3],ad[-2]
qk="Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
def short(n):
    n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)
    return n.replace('unsigned short','bf16').replace('at::native::','').replace('rocprim::ROCPRIM_400200_NS::detail::','rp::')[:64]
prev=0
with open("$out/step_$mode.txt","w") as f:
    for r in rows:
        if r["s"]>=t0 and r["e"]<=t1:
            s=r["s"]-t0; d=r["e"]-r["s"]
            f.write(f'{s/1e3:8.1f} {d/1e3:6.1f} g{(s-prev)/1e3:6.1f} q{r[qk]} {int(r.get("Grid_Size_X") or 0)//max(int(r.get("Workgroup_Size_X") or 1),1):>6} {short(r["Kernel_Name"])}\n')
            prev=max(prev,r["e"]-t0)
PY
  cp $(find $out/$mode -name "*kernel_stats.csv" | head -1) $out/stats_$mode.csv
  rm -rf $out/$mode
done
ls -la $out

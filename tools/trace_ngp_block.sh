# one gpurun call: kernel trace of bench.py --ngp, one steady-state block as a table (start us, duration us, gap to the previous kernel's end, queue, workgroups, kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ngp_trace; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t -o k -- python bench.py --ngp --no-cpu-baseline --steps 12 --warmup 3 > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-250
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$f")))
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
sc=[i for i,r in enumerate(rows) if "grid_write_kept" in r["Kernel_Name"] or "grid_scatter7" in r["Kernel_Name"]]
a,b=sc[7],sc[8]
qk="Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
t0=rows[a]["e"]; prev=t0
with open("$out/block.txt","w") as f:
    for r in rows[a+1:b+1]:
        n=re.sub(r'^void ','',r["Kernel_Name"]); n=re.sub(r'\(.*','',n)[:90]
        f.write(f'{(r["s"]-t0)/1e3:8.1f} {(r["e"]-r["s"])/1e3:7.1f} g{(r["s"]-prev)/1e3:6.1f} q{r[qk]} {int(r.get("Grid_Size_X") or 0)//max(int(r.get("Workgroup_Size_X") or 1),1):>6} {n}\n')
        prev=max(prev,r["e"])
    f.write(f'block: {(rows[b]["e"]-t0)/1e3:.1f} us, {b-a} launches\n')
PY
rm -rf $out/t
cat $out/block.txt

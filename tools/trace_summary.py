"""Summarise a rocprofv3 kernel-trace CSV: per-step GPU time by kernel (optionally split by grid size) and idle gaps.
usage: python tools/trace_summary.py trace.csv STEPS_IN_TRACE [name-substring-to-split-by-grid]"""
import csv, sys, collections
path, steps = sys.argv[1], float(sys.argv[2])
split = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:70]
    if split and split in name:
        name += f" grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += d
    busy += d
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"launches/step {len(rows)/steps:.0f}  busy {busy/1e6/steps:.2f} ms/step  span {span/1e6/steps:.2f} ms/step")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print(f"{t/1e6/steps:7.3f} ms  n={c/steps:6.1f}  avg={t/c/1e3:8.1f} us  {n}")

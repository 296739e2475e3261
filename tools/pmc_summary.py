"""Per-kernel averages of rocprofv3 --pmc counters.  usage: python tools/pmc_summary.py out.json label=dir [label=dir ...]
Each dir holds the output of one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d dir -- python bench.py ...` pass;
passes with the same label are merged (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots)."""
import csv, glob, json, os, sys
out = {}
for arg in sys.argv[2:]:
    label, d = arg.split("=", 1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0, set()])
            a[1] += float(r["Counter_Value"]); a[2].add(r["Dispatch_Id"])
        for (kn, cn), (_, tot, disp) in acc.items():
            e = out.setdefault(label, {}).setdefault(kn, {})
            e[cn] = tot / len(disp); e["launches_" + cn] = len(disp)
import hashlib
_h = hashlib.sha256()
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dreg_nerf_amd", "csrc")
for _f in sorted(os.listdir(_d)):
    if _f.endswith((".hip", ".h")):
        _h.update(open(os.path.join(_d, _f), "rb").read())
out["kernel_source_sha"] = _h.hexdigest()[:16]      # bench.py refuses the traffic numbers once the kernels change
json.dump(out, open(sys.argv[1], "w"), indent=1)
for label, ks in out.items():
    if not isinstance(ks, dict):
        continue
    top = sorted(ks.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * kv[1].get("launches_FETCH_SIZE", 0))[:8]
    for kn, e in top:
        print(label, kn[:70], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()})

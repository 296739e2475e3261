"""One steady-state training step on the GPU timeline, from a rocprofv3 --kernel-trace CSV (bench.py under the profiler):
per-queue busy time, time with nothing running, overlap between the queues, the largest gaps of the busiest queue and the
kernels around them.  Steps are delimited by adamw_kernel.  usage: python tools/timeline.py kernel_trace.csv [step-index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [r["e"] for r in rows if r["Kernel_Name"].startswith("adamw_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) - 3
t0, t1 = ad[k], ad[k + 1]
win = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
print(f"step {k}: {1e-6 * (t1 - t0):.3f} ms, {len(win)} launches, queues by {qkey}")
byq = collections.defaultdict(list)
for r in win:
    byq[r[qkey]].append(r)
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def length(iv): return sum(e - s for s, e in iv)
allu = union([(r["s"], r["e"]) for r in win])
print(f"some kernel running: {1e-6 * length(allu):.3f} ms; nothing running: {1e-6 * (t1 - t0 - length(allu)):.3f} ms")
main = max(byq, key=lambda q: sum(r["e"] - r["s"] for r in byq[q]))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    u = union([(r["s"], r["e"]) for r in rs])
    print(f"  queue {q}: {len(rs)} launches, busy {1e-6 * length(u):.3f} ms (sum of durations {1e-6 * sum(r['e'] - r['s'] for r in rs):.3f})")
mu = union([(r["s"], r["e"]) for r in byq[main]])
others = union([(r["s"], r["e"]) for q in byq if q != main for r in byq[q]])
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e: t += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
print(f"busiest queue {main}: alone {1e-6 * (length(mu) - inter(mu, others)):.3f} ms, together with another queue {1e-6 * inter(mu, others):.3f} ms; "
      f"other queues alone {1e-6 * (length(others) - inter(mu, others)):.3f} ms")
# gaps on the busiest queue
rs = sorted(byq[main], key=lambda r: r["s"])
gaps = []
for a, b in zip(rs, rs[1:]):
    g = b["s"] - a["e"]
    if g > 0:
        covered = inter([[a["e"], b["s"]]], others)
        gaps.append((g, covered, a["Kernel_Name"].split("(")[0][:48], b["Kernel_Name"].split("(")[0][:48]))
tot = sum(g for g, *_ in gaps)
print(f"gaps on queue {main}: {len(gaps)} totalling {1e-6 * tot:.3f} ms (of which {1e-6 * sum(c for _, c, *_ in gaps):.3f} ms another queue was busy)")
hist = collections.Counter()
for g, *_ in gaps:
    hist[min(int(g / 1000) // 2 * 2, 40)] += g
print("  gap time by gap length (us bucket: ms):", ", ".join(f"{b}+: {1e-6 * t:.2f}" for b, t in sorted(hist.items())))
for g, c, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {1e-3 * g:8.1f} us (other queue busy {1e-3 * c:6.1f})  after {a}  before {b}")
pairs = collections.Counter()
for g, c, a, b in gaps:
    pairs[(a, b)] += g
print("gap time by (previous kernel, next kernel):")
for (a, b), t in pairs.most_common(25):
    print(f"  {1e-6 * t:6.3f} ms  {a} -> {b}")
print("kernel time by queue (this step):")
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    agg = collections.Counter(); n = collections.Counter(); wgs = collections.Counter()
    for r in rs:
        k = r["Kernel_Name"].split("(")[0][:90]
        agg[k] += r["e"] - r["s"]; n[k] += 1
        wgs[k] += (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * (int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)) * (int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))
    print(f" queue {q}:")
    for k, t in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 28):
        print(f"   {1e-6 * t:7.3f} ms  n={n[k]:4d}  avg {1e-3 * t / n[k]:7.1f} us  avg workgroups {wgs[k] // n[k]:6d}  {k}")

"""Print the top kernels of a rocprofv3 rocpd database (ser_results.db): name, calls, total ms, average us."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows)
for name, calls, dur, avg in rows[:n]:
    print(f"{name[:80]:80s} {calls:6d} {dur/1e3:9.2f} ms {avg:8.1f} us {100*dur/tot:5.1f}%")
print(f"total {tot/1e3:.1f} ms")

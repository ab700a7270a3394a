"""Summarise a rocprofv3 sqlite result (kernel-trace): top kernels by total time -> markdown-ish table."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print(f"total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':90s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'%':>6s}")
for r in rows[:n]:
    print(f"{r[0][:90]:90s} {r[1]:7d} {r[2]/1e3:12.1f} {r[3]/1e3:9.2f} {r[4]:6.2f}")

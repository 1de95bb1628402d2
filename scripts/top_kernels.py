"""Top kernels by total GPU time from a rocprofv3 rocpd database."""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.1f ms over %d kernel launches (durations: rocpd top_kernels view, microseconds)" % (tot / 1e3, sum(r[1] for r in rows)))
for name, calls, total, avg, pct in rows[:n]:
    short = name.split("(")[0][-70:]
    print("%6.2f%% %9.2f ms %8d calls %9.2f us  %s" % (pct, total / 1e3, calls, avg, short))

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

# the same kernel launched on different grids (e.g. the simulator step on the 256 scenes of the workload and on the
# 16 384 scenes of the saturated roofline measurement): average duration per launch shape
try:
    shapes = list(con.execute(
        "select name, grid_x / workgroup_x as wgs, workgroup_x, count(*), avg(duration) / 1000.0, sum(duration) / 1000.0 "
        "from kernels where name like '%sim_step%' or name like '%sim_reset%' or name like '%mlp_fwd%' group by name, wgs, workgroup_x order by name, wgs"))
    if shapes:
        print("per launch shape (workgroups x threads):")
        for name, wgs, wx, calls, avg, total in shapes:
            print("         %9.2f ms %8d calls %9.2f us  %s  [%d x %d]" % (total / 1e3, calls, avg, name.split("(")[0][-60:], wgs, wx))
except sqlite3.Error as e:
    print("(no per-shape breakdown: %s)" % e)

"""Summarise rocprofv3 rocpd sqlite outputs (kernel time stats + per-kernel mean of every PMC counter).
usage: python scripts/pmc_summary.py <dir with *_results.db> [kernel-substring]"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "copo::"
for f in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
    con = sqlite3.connect(f)
    print("==", os.path.relpath(f, root))
    try:
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            if needle in name:
                print("  kernel %-48s calls=%d avg_us=%.3f total_us=%.1f pct=%.2f" % (name.split("(")[0][:48], calls, avg, total, pct))
    except sqlite3.Error as e:
        print("  (no top_kernels: %s)" % e)
    try:
        q = ("select kernel_name, counter_name, avg(value), count(*), max(workgroup_size), max(vgpr_count), max(sgpr_count), "
             "max(lds_block_size) from counters_collection group by kernel_name, counter_name")
        for kn, cn, v, n, wg, vg, sg, lds in con.execute(q):
            if needle in kn:
                print("  pmc %-28s %-24s mean=%.1f n=%d (wg=%s vgpr=%s sgpr=%s lds=%s)" % (kn.split("(")[0][:28], cn, v, n, wg, vg, sg, lds))
    except sqlite3.Error:
        pass

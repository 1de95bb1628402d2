"""rocprofv3 (rocpd sqlite) summary restricted to the LAST n dispatches of one kernel -- for scripts/bench_sim.py --policy cruise
these are exactly the timed replay launches on populated scenes (the warm-up / recording launches come first).
usage: replay_summary.py <dir> <kernel-substring> <n>"""
import glob
import os
import sqlite3
import sys

root, needle, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
for f in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
    con = sqlite3.connect(f)
    print("==", os.path.relpath(f, root))
    try:
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        order = "start" if "start" in cols else ("dispatch_id" if "dispatch_id" in cols else "rowid")
        rows = list(con.execute("select duration, grid_x, workgroup_x from kernels where name like ? order by %s" % order, ("%" + needle + "%",)))
        if rows:
            last = rows[-n:]
            d = [r[0] / 1000.0 for r in last]
            print("  trace: %d dispatches of *%s* in total; last %d (grid %d x %d): mean %.2f us, min %.2f, max %.2f"
                  % (len(rows), needle, len(last), last[-1][1] // max(1, last[-1][2]), last[-1][2], sum(d) / len(d), min(d), max(d)))
    except sqlite3.Error as e:
        print("  (kernels: %s)" % e)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in cols else "rowid"
        names = [r[0] for r in con.execute("select distinct counter_name from counters_collection where kernel_name like ?", ("%" + needle + "%",))]
        for cn in names:
            vals = [r[0] for r in con.execute("select value from counters_collection where kernel_name like ? and counter_name = ? order by %s" % order,
                                              ("%" + needle + "%", cn))]
            last = vals[-n:]
            print("  pmc %-26s mean over the last %d dispatches = %.1f" % (cn, len(last), sum(last) / max(1, len(last))))
    except sqlite3.Error:
        pass

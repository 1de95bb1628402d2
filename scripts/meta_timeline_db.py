"""Device timeline of ONE LCF meta phase from a rocprofv3 rocpd database (kernel trace of scripts/ab_lib.py or bench.py): every kernel
between the last weight-gradient kernel of an iteration's PPO epochs and the next iteration's first simulator step, with its queue,
start offset and duration -- shows which chain (gradient GEMMs on the main stream, LCF steps on the side stream) is the critical one."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(con.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
names = [r[0] for r in rows]
# the last complete iteration: find the last sim_step launch on 256 workgroups, walk back to the previous wgrad_adam
last_sim = max(i for i, n in enumerate(names) if "sim_step" in n)
j = last_sim
while j > 0 and "sim_step" in names[j]:
    j -= 1
while j > 0 and "meta_" not in names[j]:
    j -= 1
end = j
while j > 0 and "wgrad_adam" not in names[j]:
    j -= 1
beg = j
t0 = rows[beg][2]
print("columns:", cols)
print("meta phase: %d kernels, %.3f ms from the end of the last weight-gradient kernel to the end of the last meta kernel" %
      (end - beg, (rows[end][2] - t0) / 1e6))
qs = {}
prev_end = {}
for r in rows[beg + 1:end + 1]:
    q = r[3] if qcol else 0
    qs.setdefault(q, len(qs))
    gap = (r[1] - prev_end[q]) / 1e3 if q in prev_end else float("nan")
    prev_end[q] = r[2]
    print("q%d %9.1f us  +%7.1f us  (gap on its queue %6.1f)  %s" % (qs[q], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[0].split("(")[0][-48:]))

# ---- the whole iteration: device idle time (no kernel running on any queue) and the largest idle gaps --------------------------
sims = [i for i, n in enumerate(names) if "sim_step" in n]
starts = [i for k, i in enumerate(sims) if k == 0 or sims[k - 1] < i - 20]      # first simulator step of every rollout
if len(starts) >= 3:
    a, b = starts[-3], starts[-2]          # one complete iteration (rollout start to the next rollout start)
    seg = rows[a:b]
    wall = (rows[b][1] - rows[a][1]) / 1e3
    busy_end, idle, gaps = rows[a][1], 0.0, []
    for k, r in enumerate(seg):
        if r[1] > busy_end:
            idle += (r[1] - busy_end) / 1e3
            gaps.append(((r[1] - busy_end) / 1e3, (r[1] - rows[a][1]) / 1e3, seg[k - 1][0], r[0]))
        busy_end = max(busy_end, r[2])
    print("\niteration: %.3f ms wall, %d kernels, device idle (no kernel on any queue) %.3f ms in %d gaps" % (wall / 1e3, len(seg), idle / 1e3, len(gaps)))
    small = sum(g[0] for g in gaps if g[0] < 8.0)
    print("  gaps below 8 us (kernel-to-kernel hand-overs): %.3f ms in %d; larger ones:" % (small / 1e3, sum(1 for g in gaps if g[0] < 8.0)))
    for g in sorted(gaps, reverse=True)[:40]:
        print("  %7.1f us idle at %9.1f us   after %-44s before %s" % (g[0], g[1], g[2].split("(")[0][-44:], g[3].split("(")[0][-44:]))
    if len(sys.argv) > 2 and sys.argv[2] == "head":
        print("\nthe iteration up to its first SGD step:")
        pe = rows[a][1]
        for r in seg:
            if "rowpass" in r[0]:
                break
            print("  %9.1f us  +%7.1f us  (idle before %6.1f)  %s" % ((r[1] - rows[a][1]) / 1e3, (r[2] - r[1]) / 1e3, max(0.0, (r[1] - pe) / 1e3), r[0].split("(")[0][-60:]))
            pe = max(pe, r[2])
    if len(sys.argv) > 2 and sys.argv[2] in ("head", "tail"):
        print("\nthe iteration behind its last sequential LCF kernel:")
        last = max(k for k, r in enumerate(seg) if "meta_seq" in r[0])
        pe = seg[last][2]
        for r in seg[last:] + rows[b:b + 2]:
            print("  %9.1f us  +%7.1f us  (idle before %6.1f)  %s" % ((r[1] - rows[a][1]) / 1e3, (r[2] - r[1]) / 1e3, max(0.0, (r[1] - pe) / 1e3), r[0].split("(")[0][-60:]))
            pe = max(pe, r[2])

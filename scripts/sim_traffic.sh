#!/bin/bash
# HBM-side bytes per launch of the simulator step kernel from the PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), kernel trace only, raw values in KB.
# gfx950 correction: FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads (x2); this kernel's loads are
# dword / float2 per lane, for which the counter is uncalibrated -- both the raw sum and the x2 upper bound are recorded.
# The passes run `bench.py --roofline-only`: the trainer's own scenes and policy, 200 recorded env steps replayed -- the same
# state `roofline.units_per_launch` of the bench line refers to; the counters of the LAST 200 launches of the kernel (the
# counting replay: same actions, same state as the timed one) are averaged.
# Writes profiles/sim_traffic.json (read by bench.py if the kernel source hash matches).   usage: scripts/sim_traffic.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/sim_traffic
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python bench.py --roofline-only > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python bench.py --roofline-only > $OUT/write.log 2>&1
python - <<PY
import glob, json, os, sqlite3
def last_mean(d, counter, n=200):
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in cols else "rowid"
        v = [r[0] for r in con.execute("select value from counters_collection where counter_name = ? and kernel_name like '%sim_step%' order by " + order, (counter,))]
        if v:
            return sum(v[-n:]) / len(v[-n:]), len(v)
    return None, 0
f, nf = last_mean("$OUT/fetch", "FETCH_SIZE")
w, nw = last_mean("$OUT/write", "WRITE_SIZE")
log = [json.loads(l) for l in open("$OUT/fetch.log") if l.startswith("{")][-1]
res = dict(command="bench.py --roofline-only", scenes=log["scenes"], slots=log["slots"], launches_averaged=200, step_launches_seen=nf,
           fetch_size_kb=f, write_size_kb=w,
           bytes_per_launch=round((f + w) * 1024) if f is not None and w is not None else None,
           bytes_per_launch_fetch_x2=round((2 * f + w) * 1024) if f is not None and w is not None else None,
           units_per_launch=log["units_per_launch"], bytes_per_unit=log["bytes_per_unit"],
           algorithmic_bytes_per_launch=round(log["units_per_launch"] * log["bytes_per_unit"]),
           us_per_launch_under_pmc=log["us_per_launch"], kernel_source_sha1=log["kernel_source_sha1"],
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over the last 200 step launches = the replay of the "
                "recorded actions on the trainer's scenes; raw counter sums in KB; FETCH_SIZE is uncalibrated for this kernel's narrow "
                "loads (x2 = the guide's wide-read correction as an upper bound)")
json.dump(res, open(os.path.join("$ROOT", "gpurun_out", "sim_traffic.json"), "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/fetch $OUT/write

#!/bin/bash
# HBM-side bytes per launch of the simulator step kernel from the PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), kernel trace only, raw values in KB.
# gfx950 correction: FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads (x2); this kernel's loads are
# dword / float2 per lane, for which the counter is uncalibrated -- both the raw sum and the x2 upper bound are recorded.
# Writes profiles/sim_traffic.json (read by bench.py if the kernel source hash matches).   usage: scripts/sim_traffic.sh [E]
E=${1:-256}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sim_traffic_E$E
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BLK=$(python -c "print(1024 if $E <= 256 else (512 if $E <= 512 else (256 if $E <= 8192 else 64)))")
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python scripts/bench_sim.py --E $E --blocks $BLK --policy cruise > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python scripts/bench_sim.py --E $E --blocks $BLK --policy cruise > $OUT/write.log 2>&1
python - <<PY
import glob, json, os, sqlite3, hashlib, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
def mean(d, counter):
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for kn, v, n in con.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
            if "sim_step" in kn:
                return v, n
    return None, 0
f, nf = mean("$OUT/fetch", "FETCH_SIZE")
w, nw = mean("$OUT/write", "WRITE_SIZE")
h = hashlib.sha1()
for s in ("sim_kernels.hip", "sim_common.h", "sim_math.h"):
    h.update(open(os.path.join("$GRAFT_REPO_ROOT", "copo_amd", "csrc", s), "rb").read())
log = [json.loads(l) for l in open("$OUT/fetch.log") if l.startswith("{")]
res = dict(scenes=$E, block=$BLK, actions="lane-keeping controller", launches=nf, fetch_size_kb=f, write_size_kb=w,
           bytes_per_launch=round((f + w) * 1024) if f is not None and w is not None else None,
           bytes_per_launch_fetch_x2=round((2 * f + w) * 1024) if f is not None and w is not None else None,
           present_frac=log[-1]["present_frac"] if log else None,
           algorithmic_bytes_per_launch=round(log[-1]["present_frac"] * $E * log[-1]["N"] * (202 + 4 * log[-1]["O"])) if log else None,
           kernel_source_sha1=h.hexdigest()[:16],
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over all step launches of bench_sim.py incl. its warm-up; "
                "raw counter sums in KB; FETCH_SIZE is uncalibrated for this kernel's narrow loads (x2 = the guide's wide-read correction as an upper bound)")
json.dump(res, open(os.path.join("$GRAFT_REPO_ROOT", "gpurun_out", "sim_traffic_E$E.json"), "w"), indent=1)
print(json.dumps(res))
PY

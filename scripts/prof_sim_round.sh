#!/bin/bash
# Round evidence for the simulator step kernel, on the very commands whose numbers the bench line carries:
#   live       `bench.py --roofline-only`:  the trainer's 256 scenes x 40 slots, its own policy, 200 recorded steps replayed
#   saturated  `bench.py --saturated-only`: 16 384 populated scenes (lane-keeping controller), 60 recorded steps replayed
#   c3|c4|c5   `bench.py --config-leg <c>`: the other BASELINE configurations' simulators (Roundabout; Tollgate O = 156; ParkingLot 10 slots x 240 beams)
# Kernel trace + PMC passes, each in its own run (no --stats / trace domains next to --pmc); the summaries cover the LAST n
# dispatches of the kernel = the replay.     usage: scripts/prof_sim_round.sh live|saturated [tag]   -> gpurun_out/prof_<tag>_<mode>/summary.txt (+ sim_valu.json for saturated)
set -u
MODE=$1
TAG=${2:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $MODE in
  live) ARGS="--roofline-only"; N=200;;
  c3|c4|c5) ARGS="--config-leg $MODE --leg-scenes ${LEG_SCENES:-16384}"; N=60;;      # (one size per process; the last 60 dispatches: the replay)
  *) ARGS="--saturated-only"; N=60;;
esac
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python $ROOT/bench.py $ARGS > $OUT/pmc1.log 2>&1
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $ROOT/bench.py $ARGS > $OUT/pmc2.log 2>&1
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $ROOT/bench.py $ARGS > $OUT/pmc3.log 2>&1
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $ROOT/bench.py $ARGS > $OUT/pmc4.log 2>&1
cd $ROOT
{
  echo "# scripts/prof_sim_round.sh $MODE: copo::sim_step_kernel under rocprofv3, command: python bench.py $ARGS"
  echo "# (the LAST $N dispatches of the kernel are the replay of the recorded actions; line below: what the command printed in the trace pass,"
  echo "#  HIP events around the back-to-back replay)"
  grep -h '^{' $OUT/trace.log | tail -3
  python scripts/replay_summary.py $OUT sim_step $N
} > $OUT/summary.txt 2>&1
# VALU roofline of the saturated launch: wave-level VALU instructions per launch (SQ_INSTS_VALU) and the cycles the VALUs were
# busy (SQ_ACTIVE_INST_VALU, quad-cycles) -> profiles/sim_valu.json, read by bench.py when the kernel source hash matches
if [ "$MODE" = saturated ]; then
python - <<PY
import glob, json, os, sqlite3, sys
sys.path.insert(0, "$ROOT")
import bench
def last_mean(d, counter, n):
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in cols else "rowid"
        v = [r[0] for r in con.execute("select value from counters_collection where counter_name = ? and kernel_name like '%sim_step%' order by " + order, (counter,))]
        if v:
            return sum(v[-n:]) / len(v[-n:])
    return None
log = [json.loads(l) for l in open("$OUT/pmc1.log") if l.startswith("{")][-1]
res = dict(command="bench.py --saturated-only", scenes=log["scenes"], launches_averaged=$N, present_slots=log["present_slots"],
           valu_wave_instructions_per_launch=last_mean("$OUT/pmc1", "SQ_INSTS_VALU", $N),
           salu_wave_instructions_per_launch=last_mean("$OUT/pmc1", "SQ_INSTS_SALU", $N),
           lds_wave_instructions_per_launch=last_mean("$OUT/pmc1", "SQ_INSTS_LDS", $N),
           valu_busy_quad_cycles_per_launch=last_mean("$OUT/pmc2", "SQ_ACTIVE_INST_VALU", $N),
           gui_active_cycles_all_xcds=last_mean("$OUT/pmc2", "GRBM_GUI_ACTIVE", $N),
           us_per_launch_under_pmc=log["us_per_launch"], kernel_source_sha1=bench.kernel_source_hash(),
           note="rocprofv3 --pmc (separate passes from the trace), mean over the last $N step launches = the replay on 16 384 populated scenes")
json.dump(res, open(os.path.join("$ROOT", "gpurun_out", "sim_valu.json"), "w"), indent=1)
print(json.dumps(res))
PY
fi
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4      # (gpurun copies at most 64 MiB back)
cat $OUT/summary.txt

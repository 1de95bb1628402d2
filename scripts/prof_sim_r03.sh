#!/bin/bash
# Round-3 evidence for the simulator step kernel, on the very commands whose numbers the bench line carries:
#   live       `bench.py --roofline-only`:  the trainer's 256 scenes x 40 slots, its own policy, 200 recorded steps replayed
#   saturated  `bench.py --saturated-only`: 16 384 populated scenes (lane-keeping controller), 60 recorded steps replayed
# Kernel trace + PMC passes, each in its own run (no --stats / trace domains next to --pmc); the summaries cover the LAST n
# dispatches of the kernel = the replay.     usage: scripts/prof_sim_r03.sh live|saturated   -> gpurun_out/prof_r03_<mode>/summary.txt
set -u
MODE=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r03_$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = live ]; then ARGS="--roofline-only"; N=200; else ARGS="--saturated-only"; N=60; fi
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python $ROOT/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $ROOT/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $ROOT/bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $ROOT/bench.py $ARGS > $OUT/pmc4.log 2>&1
cd $ROOT
{
  echo "# scripts/prof_sim_r03.sh $MODE: copo::sim_step_kernel under rocprofv3, command: python bench.py $ARGS"
  echo "# (the LAST $N dispatches of the kernel are the replay of the recorded actions; line below: what the command printed in the trace pass,"
  echo "#  HIP events around the back-to-back replay)"
  grep -h '^{' $OUT/trace.log | tail -1
  python scripts/replay_summary.py $OUT sim_step_kernel $N
} > $OUT/summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4      # (gpurun copies at most 64 MiB back)
cat $OUT/summary.txt

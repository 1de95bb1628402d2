#!/bin/bash
# Round-3 evidence for the simulator step kernel on POPULATED scenes: kernel trace + PMC passes (each in its own run, no
# --stats / trace domains next to --pmc) over scripts/bench_sim.py --policy cruise; summaries cover the timed replay only.
# usage: scripts/prof_sim_r03.sh <E>   -> gpurun_out/prof_r03_E<E>/summary.txt
set -u
E=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r03_E$E
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--E $E --blocks 0 --policy cruise"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/scripts/bench_sim.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python $ROOT/scripts/bench_sim.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $ROOT/scripts/bench_sim.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $ROOT/scripts/bench_sim.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $ROOT/scripts/bench_sim.py $ARGS > $OUT/pmc4.log 2>&1
cd $ROOT
{
  echo "# scripts/prof_sim_r03.sh $E: copo::sim_step_kernel on $E populated Intersection scenes x 40 slots (lane-keeping controller,"
  echo "# recorded closed-loop, replayed from the saved state: the LAST 200 dispatches of the kernel are the timed replay)"
  grep -h '^{' $OUT/trace.log | tail -1
  python scripts/replay_summary.py $OUT sim_step_kernel 200
} > $OUT/summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4      # (gpurun copies at most 64 MiB back)
cat $OUT/summary.txt

#!/bin/bash
# rocprofv3 passes over the simulator micro-benchmark: kernel trace + two PMC passes.
# usage: scripts/prof_sim.sh <tag> [bench_sim args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/scripts/bench_sim.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python $ROOT/scripts/bench_sim.py "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $ROOT/scripts/bench_sim.py "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $ROOT/scripts/bench_sim.py "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $ROOT/scripts/bench_sim.py "$@" > $OUT/pmc4.log 2>&1
find $OUT -name '*.csv' | head -30
tail -3 $OUT/*.log

#!/bin/bash
# VALU / wave counters of the simulator step kernel at a given scene count (rocprofv3 --pmc, own pass, kernel trace only).
# usage: scripts/sim_pmc.sh E block [map]  -> gpurun_out/sim_pmc_E<E>/
E=${1:-16384}; B=${2:-256}; MAP=${3:-intersection}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sim_pmc_E$E
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python scripts/bench_sim.py --E $E --blocks $B --map $MAP > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -- python scripts/bench_sim.py --E $E --blocks $B --map $MAP > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/pmc2 -- python scripts/bench_sim.py --E $E --blocks $B --map $MAP > $OUT/pmc2.log 2>&1
python scripts/pmc_summary.py $OUT sim_step

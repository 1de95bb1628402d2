#!/bin/bash
# VALU instructions of the simulator step kernel with phases switched off (COPO_SIM_SKIP, profiling only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sk in 0 1 2 4 8 16 6; do
  OUT=$ROOT/gpurun_out/valu_split/s$sk; mkdir -p $OUT
  COPO_SIM_SKIP=$sk rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT -o p -- python $ROOT/scripts/bench_sim.py --E ${1:-16384} --blocks ${2:-256} > $OUT/log 2>&1
  echo "skip=$sk"; python $ROOT/scripts/pmc_summary.py $OUT sim_step | grep -E "avg_us|INSTS"
  rm -rf $OUT
done

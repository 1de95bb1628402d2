#!/bin/bash
# HBM fetch bytes of the batched meta weight-gradient GEMM (PMC pass over the bench)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_meta_fetch; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p1 -o p -- python $ROOT/bench.py --steps 3 --warmup 5 --no-cpu-baseline > $OUT/p1.log 2>&1
python $ROOT/scripts/pmc_summary.py $OUT ${1:-gemm_bw_kernel} | cut -c1-150
python $ROOT/scripts/pmc_summary.py $OUT rowpass_kernel | grep pmc | cut -c1-150
python $ROOT/scripts/pmc_summary.py $OUT wgrad_adam | grep pmc | cut -c1-150
rm -rf $OUT/p1

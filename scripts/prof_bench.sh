#!/bin/bash
# rocprofv3 kernel trace of the headline bench; usage: scripts/prof_bench.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
tail -n 2 $OUT/trace.log | cut -c1-1500
python $ROOT/scripts/top_kernels.py $OUT/trace/trace_results.db 25
rm -rf $OUT/trace

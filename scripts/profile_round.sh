#!/bin/bash
# Round profile: the default bench under rocprofv3 (kernel trace + stats) and HBM traffic counters of the
# simulator step kernel at the bench configuration.  Summaries land in gpurun_out/profile_<tag>/ (copy to profiles/).
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $ROOT/bench.py --steps 10 --warmup 5 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline"; echo "# bench line (traced run):"; tail -n 1 $OUT/bench_traced.json; echo "# bench line (plain run):"; tail -n 1 $OUT/bench_plain.json; python $ROOT/scripts/top_kernels.py $OUT/trace/trace_results.db 30; } > $OUT/kernel_stats.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$ctr -o pmc -- python $ROOT/scripts/bench_sim.py --E 256 --blocks 1024 > $OUT/pmc_$ctr.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o pmc -- python $ROOT/scripts/bench_sim.py --E 256 --blocks 1024 > $OUT/pmc_sq.log 2>&1
{ echo "# PMC passes on: python scripts/bench_sim.py --E 256 --blocks 1024 (the bench's simulator configuration)"; python $ROOT/scripts/pmc_summary.py $OUT sim_step; } > $OUT/sim_step_pmc.txt
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
cat $OUT/kernel_stats.txt | cut -c1-200 | head -24; cat $OUT/sim_step_pmc.txt | cut -c1-160

#!/bin/bash
# Round-end evidence: rocprofv3 --kernel-trace --stats of the default bench command, the plain bench line next to it, and the
# PMC traffic pass of the simulator kernel.   usage (on the GPU box): scripts/round_profile.sh r02   -> gpurun_out/<tag>_*
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; rm -rf /tmp/prof_bench
cd $GRAFT_REPO_ROOT
scripts/sim_traffic.sh > /tmp/traffic.log 2>&1
cp $OUT/sim_traffic.json profiles/sim_traffic.json 2>/dev/null     # bench.py reads it (source hash checked)
cp $OUT/sim_valu.json profiles/sim_valu.json 2>/dev/null           # (written by scripts/prof_sim_round.sh saturated when it ran in the same call)
python bench.py --steps 10 --warmup 5 > /tmp/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /tmp/bench_traced.log 2>&1
DB=$(find /tmp/prof_bench -name "*_results.db" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline     ($TAG, one MI355X)"
  echo "# bench line (traced run):"; grep '^{"metric"' /tmp/bench_traced.log | tail -1
  echo "# bench line (plain run, with cpu_baseline):"; grep '^{"metric"' /tmp/bench_plain.log | tail -1
  echo "# note: this trace covers the WHOLE command -- warm-up, the timed iterations and the roofline legs.  The at::native elementwise kernels in the"
  echo "#   list (MulFunctor, add, copyBuffer, sum ...) are launched by the roofline legs (the torch lane-keeping controller that records 60 steps of"
  echo "#   actions for 16 384 scenes, random-action replays) -- a timed iteration launches 47 framework ops, 0.24 ms (COPO_ITER_NO_GRAPHS=1 python scripts/iter_torch_ops.py)"
  python scripts/top_kernels.py $DB 24
} > $OUT/${TAG}_bench_kernel_stats.txt
cp $OUT/sim_traffic.json $OUT/${TAG}_sim_traffic.json 2>/dev/null
rm -rf /tmp/prof_bench
tail -30 $OUT/${TAG}_bench_kernel_stats.txt | cut -c1-200

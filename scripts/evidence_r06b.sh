#!/bin/bash
# Round-6 closing evidence on one MI355X (the simulator kernels are unchanged since scripts/evidence_r06.sh ran: its PMC profiles stand):
# GPU test suite, smoke, bench line (plain / traced with kernel stats / one rank forced through the data-parallel path), the meta phase's
# device timeline, same-box A/B of the whole job against the tree the round's second half started from (ab_base/, when present).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/r06b_gputests.txt; cat $OUT/r06b_gputests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/r06b_smoke.txt
bash scripts/round_profile.sh r06b > /dev/null 2>&1
python bench.py > $OUT/r06b_bench_line.json 2> $OUT/r06b_bench_err.txt; cut -c1-330 $OUT/r06b_bench_line.json
COPO_FORCE_DIST=1 python bench.py --no-cpu-baseline > $OUT/r06b_bench_force_dist.json 2>/dev/null; cut -c1-200 $OUT/r06b_bench_force_dist.json
scripts/ab_meta_r06.sh > $OUT/r06b_ab_meta.txt 2>&1
cp $OUT/ab_meta/timeline.txt $OUT/r06b_meta_timeline.txt
if [ -d ab_base ]; then
  for rep in 1 2 3; do for d in ab_base .; do echo -n "$d: "; (cd $d && python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); p=l['phases']; print(l['value'], l['ms_per_step'], p['sample_ms'], p['sgd_ms'], p['meta_ms'], p['iteration_ms'], l['learner_roofline']['us_per_step'], l['roofline']['saturated']['us_per_launch'])"); done; done > $OUT/r06b_bench_ab.txt
  cat $OUT/r06b_bench_ab.txt
fi

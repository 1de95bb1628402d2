#!/bin/bash
# Round-6 evidence on one MI355X: the GPU test suite, smoke, the bench line (plain / traced / one rank forced through the data-parallel path),
# PMC profiles of the simulator step kernel (saturated + live), its per-phase instruction and LDS-conflict split.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/r06_gputests.txt; cat $OUT/r06_gputests.txt
python __graft_entry__.py smoke 2>&1 | tail -2
bash scripts/prof_sim_round.sh saturated r06 > /dev/null 2>&1; cp $OUT/sim_valu.json profiles/sim_valu.json 2>/dev/null
bash scripts/prof_sim_round.sh live r06 > /dev/null 2>&1
bash scripts/round_profile.sh r06 > /dev/null 2>&1
python bench.py > $OUT/r06_bench_line.json 2> $OUT/r06_bench_err.txt; tail -c 2500 $OUT/r06_bench_line.json
COPO_FORCE_DIST=1 python bench.py --no-cpu-baseline > $OUT/r06_bench_force_dist.json 2>/dev/null; cut -c1-200 $OUT/r06_bench_force_dist.json
( cd /tmp && export TMPDIR=/tmp && VALU_POLICY=cruise SPLIT_LDS=1 timeout 1400 python $GRAFT_REPO_ROOT/scripts/sim_valu_split.py 16384 64 ) > $OUT/r06_sim_split.txt 2>&1; cat $OUT/r06_sim_split.txt
for mb in 512 1024; do echo "== fused step, $mb rows per minibatch"; COPO_BENCH_MB=$mb python scripts/bench_fused.py 300 2>&1 | grep "fused sgd"; done > $OUT/r06_step_mb.txt; cat $OUT/r06_step_mb.txt
cp $OUT/prof_r06_saturated/summary.txt $OUT/r06_sim_step_pmc_E16384.txt 2>/dev/null
cp $OUT/prof_r06_live/summary.txt $OUT/r06_sim_step_pmc_E256.txt 2>/dev/null
ls $OUT | head -50
python scripts/config_times.py 2>&1 | grep -v amdgpu > $OUT/r06_config_times_final.txt; cat $OUT/r06_config_times_final.txt
python bench.py --config-leg c4 2>/dev/null | grep "^{" > $OUT/r06_config_leg_c4_final.txt; cut -c1-330 $OUT/r06_config_leg_c4_final.txt

#!/bin/bash
# CoPO / IPPO on the Intersection, 30 agents, the reference's 8 seeds (start_seed 5000 .. 12000), 1 M env steps each (256 scenes x 8 steps,
# scenes in lockstep): the maximum training success per seed is the reference's table metric (benchmarks/MetaDrive-0.2.5/README.md:19-31).
for algo in copo ippo; do
  for seed in 0 1 2 3 4 5 6 7; do
    python scripts/train_curve.py --algo $algo --num-agents 30 --stop 1000000 --every 50 --seed $seed \
      --env-config "{\"start_seed\": $((5000 + 1000 * seed))}" 2>&1 | grep -v amdgpu.ids
  done
done

#!/bin/bash
# Bottleneck / Tollgate rows of the reference's training table on the rebuilt scenes (Merge / Split funnels), 3 seeds each.
for map in MultiAgentBottleneckEnv MultiAgentTollgateEnv; do
  for algo in ippo copo; do
    for seed in 0 1 2; do
      python scripts/train_curve.py --stagger 1 --algo $algo --map $map --stop 1000000 --every 50 --seed $seed \
        --env-config "{\"start_seed\": $((5000 + 1000 * seed))}" 2>&1 | grep -v amdgpu.ids
    done
  done
done

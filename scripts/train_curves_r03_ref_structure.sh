#!/bin/bash
# CoPO / IPPO on the Intersection with the REFERENCE's rollout structure (10 scenes x 200-step fragments), the reference's 8 seeds.
for algo in copo ippo; do
  for seed in 0 1 2 3 4 5 6 7; do
    python scripts/train_curve.py --algo $algo --num-envs 10 --stop 1000000 --every 100 --seed $seed \
      --env-config "{\"start_seed\": $((5000 + 1000 * seed))}" 2>&1 | grep -v amdgpu.ids
  done
done

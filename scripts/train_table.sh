#!/bin/bash
# The reference's training table (benchmarks/MetaDrive-0.2.5/README.md:19-25) on the HIP simulator: IPPO and CoPO on the six
# scenes, 3 seeds, 1 M env steps each, the scene's own population.  usage: [MAPS="MultiAgentParkingLotEnv"] bash scripts/train_table.sh > gpurun_out/train_table.txt
MAPS=${MAPS:-MultiAgentBottleneckEnv MultiAgentTollgateEnv MultiAgentIntersectionEnv MultiAgentRoundaboutEnv MultiAgentParkingLotEnv MultiAgentMetaDrive}
for map in $MAPS; do
  for algo in ippo copo; do
    for seed in 0 1 2; do
      python scripts/train_curve.py --stagger 1 --algo $algo --map $map --stop 1000000 --every 50 --seed $seed \
        --env-config "{\"start_seed\": $((5000 + 1000 * seed))}" 2>&1 | grep -v amdgpu.ids
    done
  done
done

#!/bin/bash
# Round-5 review item 4: why does the LCF climb to ~0.7 here (256 scenes x 8-step fragments) when the reference's own run ends at 0.225?
# One controlled variant per hypothesis, CoPO on the Intersection (30 agents), 1 M env steps, the reference's seeds.
#   usage: bash scripts/lcf_drift_experiments.sh "<variants>" "<seeds>"      variants: boot0 ref_boot0 e64 e32 base
VARIANTS=${1:-"boot0 ref_boot0 e64 e32"}
SEEDS=${2:-"0 1 2 3"}
for v in $VARIANTS; do
  EXTRA=""
  case $v in
    base)      ENVS=256; CFG='{}';;
    boot0)     ENVS=256; CFG='{"bootstrap_next_obs": false}';;                 # (i) the reference's bootstrap V(last obs of the fragment)
    ref_boot0) ENVS=10;  CFG='{"bootstrap_next_obs": false}';;                 # (ii) + (i): 10 scenes x 200 steps, reference bootstrap
    ref)       ENVS=10;  CFG='{}';;
    e64)       ENVS=64;  CFG='{}';;                                            # between the two structures: 64 scenes x 32 steps
    e32)       ENVS=32;  CFG='{}';;                                            # 32 scenes x 63 steps
    lr3)       ENVS=256; CFG='{"lcf_lr": 3e-5}';;                                # Adam moves the LCF by ~lr x (signal / noise) per step: a third of the rate
    stag)      ENVS=256; CFG='{}'; EXTRA="--stagger 1";;                        # scenes out of phase (no lockstep episodes)
    stag_lr3)  ENVS=256; CFG='{"lcf_lr": 3e-5}'; EXTRA="--stagger 1";;
    *) echo "unknown variant $v"; exit 1;;
  esac
  for seed in $SEEDS; do
    echo "### variant=$v num_envs=$ENVS config=$CFG seed=$seed"
    python scripts/train_curve.py --algo copo --num-envs $ENVS --stop 1000000 --every $((50 * 256 / ENVS > 400 ? 100 : (ENVS >= 256 ? 50 : 100))) --seed $seed $EXTRA \
      --config "$CFG" --env-config "{\"start_seed\": $((5000 + 1000 * seed))}" 2>&1 | grep -v "amdgpu.ids\|^$"
  done
done

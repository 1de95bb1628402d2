#!/bin/bash
# Every population the reference ships, rolled in the HIP simulator: the reference's evaluation table (RecorderEnv columns) for the
# Intersection next to its CSVs, and the headline rates of the other scenes.   usage: bash scripts/eval_tables_r03.sh > profiles/r03_eval_tables.txt
echo "# bash scripts/eval_tables_r03.sh   (one MI355X; 64 scenes x whole scene episodes)"
echo "# ---- Intersection: python scripts/eval_recorder_table.py 1  (columns: here | reference CSV mean over its populations [min .. max] | population 0) ----"
python scripts/eval_recorder_table.py 1 2>&1 | grep -v amdgpu.ids
echo "# ---- every scene: python scripts/eval_reference_population.py 1 ----"
python scripts/eval_reference_population.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "# ---- Bottleneck / Tollgate (Merge / Split blocks as MetaDrive builds them): python scripts/eval_f4_populations.py ----"
echo "# reference: copo_bottle 0.867 (eval/get_policy_function.py:29 '# 0.867, Best'), copo_round 0.858 (:41); training table (MetaDrive 0.2.5): Bottleneck 0.240 / 0.474, Tollgate 0.044 / 0.272"
python scripts/eval_f4_populations.py 2>&1 | grep -v amdgpu.ids

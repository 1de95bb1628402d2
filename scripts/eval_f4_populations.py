"""The reference's Tollgate / Bottleneck populations (tests/golden/reference_populations_f4.npz) in the HIP simulator, under
variants of the scene / observation parameters that MetaDrive's source would settle (it is not in the reference tree):
    python scripts/eval_f4_populations.py '{"tollgate": {"lidar_range": 20}, "bottle": {"side_range": 50}}'"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd.eval.evaluate import evaluate_population  # noqa: E402
from copo_amd.eval.get_policy_function import meta_svo_lookup_table  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "reference_populations_f4.npz"))
over = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
KEEP = ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean", "episode_reward_mean",
        "episode_length_mean", "route_completion_mean", "velocity_mean", "num_terminated_agents")
for name, algo, env, n in (("ippo_tollgate", "ippo", "tollgate", 40), ("copo_tollgate", "copo", "tollgate", 40),
                           ("ippo_bottle", "ippo", "bottle", 20), ("copo_bottle", "copo", "bottle", 20)):
    pre = name + "/w/"
    w = {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}
    r = evaluate_population(algo, env, w, meta_svo_lookup_table.get(name), num_envs=64, num_agents=n, scene_episodes=1, seed=0,
                            env_config=over.get(env, {}))
    print("%-14s %s %s" % (name, json.dumps(over.get(env, {})), json.dumps({k: round(float(r[k]), 4) for k in KEEP if k in r})), flush=True)

"""Cross-simulator sanity check: roll the populations the reference trained in MetaDrive
(weights held as data in tests/golden/eval_policy_function.npz) in the HIP simulator, next to an untrained policy."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd.eval.evaluate import evaluate_population  # noqa: E402
from copo_amd.eval.get_policy_function import meta_svo_lookup_table  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "eval_policy_function.npz"))
episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, algo in (("copo_inter", "copo"), ("ippo_inter", "ippo"), ("ccppo_inter", "ccppo")):
    pre = name + "/w/"
    w = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    lcf = meta_svo_lookup_table.get(name)
    for label, weights in (("reference-trained", w), ("untrained", None)):
        r = evaluate_population(algo, "inter", weights, lcf, num_envs=64, num_agents=40, episodes=episodes, seed=0)
        keep = ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean", "episode_reward_mean",
                "episode_length_mean", "route_completion_mean", "velocity_mean", "num_terminated_agents")
        print("%-12s %-18s %s" % (name, label, json.dumps({k: round(float(r[k]), 4) for k in keep if k in r})), flush=True)

"""Cross-simulator pin: roll the populations the reference trained in MetaDrive (weights held
as data under tests/golden/) in the HIP simulator, next to an untrained policy and to what the reference measured."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd.eval.evaluate import evaluate_population  # noqa: E402
from copo_amd.eval.get_policy_function import meta_svo_lookup_table  # noqa: E402

G1 = np.load(os.path.join(ROOT, "tests", "golden", "eval_policy_function.npz"))
G2 = np.load(os.path.join(ROOT, "tests", "golden", "reference_populations.npz"))
with open(os.path.join(ROOT, "tests", "golden", "reference_eval_stats.json")) as f:
    print("# reference, MetaDrive (eval/demo_results):", json.dumps({k: {c: round(v, 3) for c, v in d.items()} for k, d in json.load(f).items() if isinstance(d, dict)}))
scene_episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ENV_OVER = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}      # e.g. '{"body_margin": 0.5}'      # whole scene episodes (until done["__all__"]) of 64 scenes each
KEEP = ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean", "episode_reward_mean",
        "episode_length_mean", "route_completion_mean", "velocity_mean", "num_terminated_agents")
for gold, name, algo, env, n in ((G1, "copo_inter", "copo", "inter", 30), (G1, "ippo_inter", "ippo", "inter", 30),
                                 (G1, "ccppo_inter", "ccppo", "inter", 30), (G2, "copo_round", "copo", "round", 40),
                                 (G2, "ippo_round", "ippo", "round", 40), (G2, "ippo_parking", "ippo", "parking", 10)):
    pre = name + "/w/"
    w = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    lcf = meta_svo_lookup_table.get(name)
    for label, weights in (("reference-trained", w), ("untrained", None)):
        r = evaluate_population(algo, env, weights, lcf, num_envs=64, num_agents=n, scene_episodes=scene_episodes, seed=0, env_config=ENV_OVER)
        d = {k: round(float(r[k]), 4) for k in KEEP if k in r}
        print("%-12s %2d agents %-18s %s" % (name, n, label, json.dumps(d)), flush=True)

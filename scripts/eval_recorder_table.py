"""The reference's evaluation table (eval/evaluate_population.py -> RecorderEnv.get_episode_result columns) for the populations
the reference ships, rolled in the HIP simulator, next to the reference's own CSV means (tests/golden/reference_eval_stats.json).
usage: eval_recorder_table.py [scene_episodes] ['{"env_config json"}']"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd.eval.evaluate import evaluate_population_rows  # noqa: E402
from copo_amd.eval.get_policy_function import meta_svo_lookup_table  # noqa: E402

G1 = np.load(os.path.join(ROOT, "tests", "golden", "eval_policy_function.npz"))
with open(os.path.join(ROOT, "tests", "golden", "reference_eval_stats.json")) as f:
    REF = json.load(f)
episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ENV_OVER = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
FIXED_LCF = len(sys.argv) > 3 and sys.argv[3] == "fixed-lcf"      # evaluate_population.py:33 passes use_distributional_svo=False for copo_*: every agent gets the MEAN
COLS = ("success_rate", "crash_rate", "out_rate", "episode_reward_mean", "episode_reward_min", "episode_reward_max",
        "episode_length_mean", "success_episode_length_mean", "velocity_step_mean_episode_mean", "velocity_step_mean_episode_max",
        "num_neighbours_mean_episode_mean", "num_neighbours_mean_episode_max", "num_agents_total", "num_agents_total_per_300_steps",
        "env_episode_steps")
for name, algo in (("copo_inter", "copo"), ("ippo_inter", "ippo")):
    pre = name + "/w/"
    w = {k[len(pre):]: G1[k] for k in G1.files if k.startswith(pre)}
    lcf = meta_svo_lookup_table.get(name)
    if lcf is not None and FIXED_LCF:
        lcf = (lcf[0], 1e-6)
    df = evaluate_population_rows(algo, "inter", w, lcf, num_envs=64, num_agents=30, scene_episodes=episodes,
                                  seed=0, env_config=dict(ENV_OVER))
    m = df.mean(numeric_only=True)
    ref, per = REF[name], REF[name + "_per_population"]
    print("== %s: %d scene episodes here | reference: mean of %d populations [min .. max over populations]%s" %
          (name, len(df), ref["populations"], " | population 0 (the shipped one)" if name == "copo_inter" else ""))
    for c in COLS:
        vals = [p[c] for p in per if c in p]
        print("  %-36s %9.3f | %9.3f [%8.3f .. %8.3f]%s" % (c, m[c], ref.get(c, float("nan")), min(vals), max(vals),
                                                             (" | %9.3f" % per[0][c]) if name == "copo_inter" else ""))
    # reward-scale decomposition: the same two estimators on this build's rows and on the reference's CSV rows
    # (oracle/gen_golden_eval.py): metres an agent drives ~ episode_length_mean x velocity_step_mean / 3.6 x 0.1, and the reward per such
    # metre net of the terminal rewards.  Their product is the route reward; the reference's ceiling / this build's = 1.196.
    dist = df["episode_length_mean"] * df["velocity_step_mean_episode_mean"] / 3.6 * 0.1
    net = df["episode_reward_mean"] - 10.0 * df["success_rate"] + 10.0 * df["crash_rate"] + 10.0 * df["out_rate"]
    for c, here in (("metres_hat", float(dist.mean())), ("reward_per_metre_hat", float((net / dist).mean()))):
        vals = [p[c] for p in per if c in p]
        if vals:
            print("  %-36s %9.3f | %9.3f [%8.3f .. %8.3f]%s" % (c, here, float(np.mean(vals)), min(vals), max(vals),
                                                                 (" | %9.3f" % per[0][c]) if name == "copo_inter" else ""))

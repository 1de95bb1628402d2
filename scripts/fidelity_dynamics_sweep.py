"""Review item 1a (round 6): the shipped CoPO Intersection population drives 12.5 km/h / 346 steps per agent here, population 0 drove 16.7 /
260 in the release's MetaDrive -- the SAME distance (118 vs 119 m).  Which dynamics parameter of the kinematic bicycle moves that, and what
does it do to the IPPO population, whose every column matches (31.6 vs 31.8 km/h)?  One parameter at a time, 64 whole scene episodes each:
    python scripts/fidelity_dynamics_sweep.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd.eval.evaluate import evaluate_population  # noqa: E402
from copo_amd.eval.get_policy_function import meta_svo_lookup_table  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "eval_policy_function.npz"))
VARIANTS = [("default (brake_gain 27, acc_max 2.9, body_margin 0.75)", {}),
            ("brake_gain 13.5 (half the brake per unit of negative throttle)", dict(brake_gain=13.5)),
            ("brake_gain 6.75", dict(brake_gain=6.75)),
            ("brake_gain 54", dict(brake_gain=54.0)),
            ("acc_max 3.5 (+20 % engine)", dict(acc_max=3.5)),
            ("acc_max 2.4 (-17 % engine)", dict(acc_max=2.4)),
            ("body_margin 0.5", dict(body_margin=0.5)),
            ("body_margin 0.5 + brake_gain 13.5", dict(body_margin=0.5, brake_gain=13.5))]
print("reference (eval/demo_results CSVs): CoPO population 0  success .812 crash .149 out .039  16.7 km/h  260 steps;  IPPO population 3  .466 / .483 / .051  31.8 km/h  110 steps")
for tag, over in VARIANTS:
    row = []
    for name, algo in (("copo_inter", "copo"), ("ippo_inter", "ippo")):
        pre = name + "/w/"
        w = {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}
        r = evaluate_population(algo, "inter", w, meta_svo_lookup_table.get(name), num_envs=64, num_agents=30, scene_episodes=1, seed=0, env_config=over)
        row.append("%s success %.3f crash %.3f out %.3f  %5.1f km/h %4.0f steps" % (algo, r["success_rate_mean"], r["crash_rate_mean"], r["out_of_road_rate_mean"],
                                                                                    r["velocity_mean"], r["episode_length_mean"]))
    print("%-62s %s | %s" % (tag, row[0], row[1]), flush=True)

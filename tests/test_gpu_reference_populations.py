"""Cross-simulator pin of the simulator half (SURVEY.md section 8 rows a-1 / c): MetaDrive's source is not in the
reference tree, but the reference ships what MetaDrive taught its agents -- the trained populations under
copo/best_checkpoints (held here as data: tests/golden/eval_policy_function.npz, reference_populations.npz) -- and what
those populations scored in MetaDrive (eval/demo_results/evaluate_results/*.csv -> tests/golden/reference_eval_stats.json).
A policy is a function of the observation alone, so it only drives a simulator whose observation semantics (column
meaning and scale, LiDAR beam order, navigation encoding, steering sign, road geometry) are MetaDrive's.  These tests
roll the populations in the HIP simulator, 30 agents on the Intersection / 40 on the Roundabout as the reference's
evaluation does (eval/evaluate_population.py:102-132), and fail if they fall back towards an untrained policy.

Rates are over every agent that terminates inside whole scene episodes (1000 env steps of respawning, then the scene
drains until its last agent ends -- MultiAgentMetaDrive.step's done["__all__"]), as the reference's recorder counts them."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _roll(algo, env, weights, lcf, n):
    """One whole scene episode (the reference's evaluation unit) of 64 scenes."""
    from copo_amd.eval.evaluate import evaluate_population
    r = evaluate_population(algo, env, weights, lcf, num_envs=64, num_agents=n, scene_episodes=1, seed=0)
    return dict(success=float(r["success_rate_mean"]), crash=float(r["crash_rate_mean"]), out=float(r["out_of_road_rate_mean"]),
                max_step=float(r["max_step_rate_mean"]), length=float(r["episode_length_mean"]),
                velocity=float(r["velocity_mean"]), raw=r)


def _weights(gold, name):
    pre = name + "/w/"
    return {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}


def test_reference_intersection_populations_drive_the_hip_simulator(golden_dir):
    gold = np.load(os.path.join(golden_dir, "eval_policy_function.npz"))
    with open(os.path.join(golden_dir, "reference_eval_stats.json")) as f:
        ref = json.load(f)
    from copo_amd.eval.get_policy_function import meta_svo_lookup_table
    untrained = _roll("ippo", "inter", None, None, 30)
    assert untrained["success"] < 0.05, untrained
    ippo = _roll("ippo", "inter", _weights(gold, "ippo_inter"), None, 30)
    copo = _roll("copo", "inter", _weights(gold, "copo_inter"), meta_svo_lookup_table["copo_inter"], 30)
    ccppo = _roll("ccppo", "inter", _weights(gold, "ccppo_inter"), None, 30)
    print("ippo", ippo, "\ncopo", copo, "\nccppo", ccppo, "\nreference (MetaDrive)", ref)
    # MetaDrive: IPPO populations 0.48 success / 0.42 crash / 27 km/h, CoPO 0.78 / 0.15 / 14 km/h.  The bands below are
    # wide enough for the physics difference (kinematic bicycle vs Bullet) and far from the ~0 of wrong semantics.
    assert abs(ippo["success"] - ref["ippo_inter"]["success_rate"]) < 0.08, ippo        # 0.475 vs 0.480
    assert abs(copo["success"] - ref["copo_inter"]["success_rate"]) < 0.08, copo        # 0.781 vs 0.783
    assert 0.02 < ippo["out"] < 0.15 and 0.02 < copo["out"] < 0.18                      # MetaDrive 0.10 / 0.06: the body-touches-edge-line rule
    assert copo["success"] > ippo["success"] + 0.1                    # the ranking of the reference's table
    assert ccppo["success"] > 0.45
    assert copo["crash"] < 0.3
    assert abs(ippo["velocity"] - ref["ippo_inter"]["velocity_step_mean_episode_mean"]) < 8.0
    assert abs(copo["velocity"] - ref["copo_inter"]["velocity_step_mean_episode_mean"]) < 8.0
    assert copo["length"] > 1.5 * ippo["length"]                      # CoPO's populations are the patient ones (308 vs 132 steps)


def test_reference_roundabout_populations_drive_the_hip_simulator(golden_dir):
    """The roundabout's ring is eleven roads long for a full turn: populations that were trained on MetaDrive's block
    geometry only get round it if the rebuilt geometry and navigation columns match."""
    gold = np.load(os.path.join(golden_dir, "reference_populations.npz"))
    ippo = _roll("ippo", "round", _weights(gold, "ippo_round"), None, 40)
    copo = _roll("copo", "round", _weights(gold, "copo_round"), tuple(gold["copo_round/lcf"]), 40)
    print("ippo_round", ippo, "\ncopo_round", copo)
    # training-time success in MetaDrive (benchmarks/MetaDrive-0.2.5/README.md:19-31): IPPO 66 %, CoPO 73 %
    assert ippo["success"] > 0.45 and copo["success"] > 0.45, (ippo, copo)
    assert ippo["out"] < 0.1 and copo["out"] < 0.1


def test_reference_tollgate_and_bottleneck_populations(golden_dir):
    """f-4 scenes.  MetaDrive's source would settle the detector ranges and the booth rule; the populations the reference
    trained there (156- / 96-wide first layers) settle them here: the side detector of the Bottleneck reaches 50 m (with
    20 m CoPO's population crashes in the merge: 28 % success, 35 % crashes; with 50 m 51 % / 6 %), the second toll column is
    the binary "stayed longer than min_pass_steps" mark (with a waited fraction the populations leave the booth early: 10 %).
    Bands around the reference's own training table (benchmarks/MetaDrive-0.2.5/README.md:19-25: Bottleneck IPPO 24 +- 19,
    CoPO 47 +- 19; Tollgate IPPO 4 +- 3, CoPO 27 +- 26)."""
    gold = np.load(os.path.join(golden_dir, "reference_populations_f4.npz"))
    from copo_amd.eval.get_policy_function import meta_svo_lookup_table
    copo_b = _roll("copo", "bottle", _weights(gold, "copo_bottle"), meta_svo_lookup_table["copo_bottle"], 20)
    ippo_b = _roll("ippo", "bottle", _weights(gold, "ippo_bottle"), None, 20)
    copo_t = _roll("copo", "tollgate", _weights(gold, "copo_tollgate"), meta_svo_lookup_table["copo_tollgate"], 40)
    ippo_t = _roll("ippo", "tollgate", _weights(gold, "ippo_tollgate"), None, 40)
    print("copo_bottle", copo_b, "\nippo_bottle", ippo_b, "\ncopo_tollgate", copo_t, "\nippo_tollgate", ippo_t)
    assert 0.35 < copo_b["success"] < 0.75 and copo_b["crash"] < 0.2, copo_b
    assert 0.15 < ippo_b["success"] < 0.6, ippo_b
    assert 0.15 < copo_t["success"] < 0.6, copo_t
    assert ippo_t["success"] < 0.15, ippo_t          # IPPO does not learn the booth rule in the reference either
    assert copo_t["success"] > ippo_t["success"] + 0.1

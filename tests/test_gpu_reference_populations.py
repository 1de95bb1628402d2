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


def _roll(algo, env, weights, lcf, n, env_config=None):
    """One whole scene episode (the reference's evaluation unit) of 64 scenes."""
    from copo_amd.eval.evaluate import evaluate_population
    r = evaluate_population(algo, env, weights, lcf, num_envs=64, num_agents=n, scene_episodes=1, seed=0, env_config=env_config or {})
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


def _table(algo, name, gold, lcf, episodes=1):
    """The reference's per-episode evaluation table (RecorderEnv.get_episode_result columns) for 64 whole scene episodes."""
    from copo_amd.eval.evaluate import evaluate_population_rows
    df = evaluate_population_rows(algo, "inter", _weights(gold, name), lcf, num_envs=64, num_agents=30, scene_episodes=episodes, seed=0)
    assert len(df) >= 64
    out = {k: float(v) for k, v in df.mean(numeric_only=True).items()}
    # the two estimators of oracle/gen_golden_eval.py (reward-scale decomposition), on this build's rows
    dist = df["episode_length_mean"] * df["velocity_step_mean_episode_mean"] / 3.6 * 0.1
    net = df["episode_reward_mean"] - 10.0 * df["success_rate"] + 10.0 * df["crash_rate"] + 10.0 * df["out_rate"]
    out["metres_hat"], out["reward_per_metre_hat"] = float(dist.mean()), float((net / dist).mean())
    return out


def test_intersection_evaluation_tables_against_the_reference_records(golden_dir):
    """VALIDATION of the simulator half on statistics that were NOT used to choose any of its parameters (DESIGN.md
    section 3.6 lists every free parameter and the held-out statistic that confirms it).  The recorder columns of
    eval/recoder.py:188-299 -- computed here by copo_amd.eval.vec_recorder with the reference's definitions -- for the two
    shipped Intersection populations, against (a) the reference's evaluation CSVs (eval/demo_results/evaluate_results), per
    population, and (b) the reference's own MetaDrive-0.2.5 training record (demo_raw_checkpoints/.../progress.csv).

    The two records of the reference do not agree with each other on the REWARD SCALE: the CSVs (populations of the paper's
    release) pay 1.196 x the route reward of this build -- episode_reward_max 188.8 +- 1.5 over all eleven populations, ceiling
    of the longest route -- while the 0.2.5 training run pays 110 at 77 % success where the CSVs pay 137 and this build 116.
    Reward is therefore asserted (i) scale-free against the CSVs (max / mean), (ii) absolutely against the 0.2.5 record, and
    (iii) against this build's own route ceiling; every other column is asserted against the CSVs directly."""
    gold = np.load(os.path.join(golden_dir, "eval_policy_function.npz"))
    with open(os.path.join(golden_dir, "reference_eval_stats.json")) as f:
        ref = json.load(f)
    from copo_amd.eval.get_policy_function import meta_svo_lookup_table
    copo = _table("copo", "copo_inter", gold, meta_svo_lookup_table["copo_inter"])
    ippo = _table("ippo", "ippo_inter", gold, None)
    print("copo", copo, "\nippo", ippo)

    def rel(a, b):
        return abs(a - b) / abs(b)

    # ---- IPPO: which of the six evaluated populations is the shipped file is not recorded.  ONE of them must match on every
    # column at once (a wrong simulator could match one column of one population and another column of another).
    cols = (("success_rate", "abs", 0.05), ("crash_rate", "abs", 0.05), ("out_rate", "abs", 0.04),
            ("velocity_step_mean_episode_mean", "abs", 3.0), ("episode_length_mean", "rel", 0.15),
            ("success_episode_length_mean", "rel", 0.15), ("num_agents_total", "rel", 0.10),
            ("num_neighbours_mean_episode_mean", "rel", 0.15), ("env_episode_steps", "rel", 0.05))
    matches = []
    for k, pop in enumerate(ref["ippo_inter_per_population"]):
        if all((abs(ippo[c] - pop[c]) <= tol) if kind == "abs" else (rel(ippo[c], pop[c]) <= tol) for c, kind, tol in cols):
            matches.append(k)
    assert matches, (ippo, ref["ippo_inter_per_population"])      # (round 3: population 3 -- .466 / .483 / .051 / 31.8 km/h / 110 / 141 / 258 / 3.25)
    # ---- CoPO: the shipped file is population 0 ("Best", get_policy_function.py:30-31)
    p0, allp = ref["copo_inter_per_population"][0], ref["copo_inter"]
    assert abs(copo["success_rate"] - p0["success_rate"]) < 0.05                      # .774 vs .812
    assert abs(copo["crash_rate"] - p0["crash_rate"]) < 0.04                          # .127 vs .149
    assert abs(copo["out_rate"] - p0["out_rate"]) < 0.07                              # .098 vs .039 (the five populations: .039 .. .109)
    assert rel(copo["num_neighbours_mean_episode_mean"], p0["num_neighbours_mean_episode_mean"]) < 0.10     # 3.81 vs 4.00
    assert rel(copo["num_neighbours_mean_episode_max"], p0["num_neighbours_mean_episode_max"]) < 0.12       # 6.02 vs 6.55
    assert rel(copo["env_episode_steps"], allp["env_episode_steps"]) < 0.05                                  # 1429 vs 1430
    # OPEN (asserted as a band around the five populations, not around population 0): this build's CoPO population drives
    # 12.5 km/h / 346 steps per agent where population 0 drove 16.7 / 260 and the five populations 6.5 .. 18.5 / 243 .. 472
    lo = min(p["velocity_step_mean_episode_mean"] for p in ref["copo_inter_per_population"])
    hi = max(p["velocity_step_mean_episode_mean"] for p in ref["copo_inter_per_population"])
    assert lo < copo["velocity_step_mean_episode_mean"] < hi
    assert rel(copo["episode_length_mean"], allp["episode_length_mean"]) < 0.15                              # 346 vs 308 (mean of five)
    assert rel(copo["num_agents_total"], allp["num_agents_total"]) < 0.18                                    # 104 vs 121
    # ---- reward: scale-free against the CSVs, absolute against the 0.2.5 training record and the route ceiling
    for got, want in ((copo, allp), (ippo, ref["ippo_inter"])):
        assert rel(got["episode_reward_max"] / got["episode_reward_mean"],
                   want["episode_reward_max"] / want["episode_reward_mean"]) < 0.03, (got, want)             # 1.383 vs 1.370, 1.787 vs 1.775
    ceiling = (56.0 + 0.5 * np.pi * 20.5 + 60.0 - 5.0) * (1.0 + 0.1 * 3.6 / 80.0 / 0.1) + 10.0      # longest route: outer left turn; driving + speed + success
    assert abs(copo["episode_reward_max"] - ceiling) < 1.5 and abs(ippo["episode_reward_max"] - ceiling) < 3.0, ceiling
    # ---- where the 1.196 x sits (round 4): the CSV rows hold the metres an agent drives (episode length x mean velocity) and hence the
    # reward per driven metre net of the terminal rewards; the same two estimators on this build's rows.  Population 0 drives the SAME
    # distance here as in the release's MetaDrive (117.7 vs 119.2 m at .77 / .81 success): the road geometry is not what differs; the
    # release pays 1.18 x (CoPO 0) / 1.13 x (IPPO 3) per driven metre.  No constant of 0.2.5's reward_function does that (driving_reward
    # 1 / m, speed term 0.1 v / v_max = a fixed 0.045 / m, lateral factor off), and 0.2.5's own training record (below) pays this
    # build's scale: the factor belongs to the MetaDrive release that produced the CSVs, and a fixed policy's speed cannot depend on it.
    assert rel(copo["metres_hat"], p0["metres_hat"]) < 0.08, (copo["metres_hat"], p0["metres_hat"])
    assert 1.08 < p0["reward_per_metre_hat"] / copo["reward_per_metre_hat"] < 1.28
    pm = ref["ippo_inter_per_population"][matches[0]]
    assert rel(ippo["metres_hat"], pm["metres_hat"]) < 0.10 and 1.05 < pm["reward_per_metre_hat"] / ippo["reward_per_metre_hat"] < 1.25
    prog = sorted(ref["copo_inter_training_progress"], key=lambda r: r["success"])
    xs, ys = [r["success"] for r in prog], [r["episode_reward_mean"] for r in prog]
    want = float(np.interp(copo["success_rate"], xs, ys))       # the 0.2.5 record: per-agent return as a function of the success rate
    assert rel(copo["episode_reward_mean"], want) < 0.08, (copo["episode_reward_mean"], want)                 # 115.3 vs 110.3
    # the same record, throughput: agents that finish per env episode (= mean agent lifetime): 116 at 76.6 % success in MetaDrive 0.2.5
    # -- CoPO populations of 0.2.5 live ~330 steps, like the shipped one here (346) and unlike the CSVs' population 0 (260)
    ys = [r["agents_per_env_episode"] for r in prog]
    want_n = float(np.interp(copo["success_rate"], xs, ys))
    assert rel(copo["num_agents_total"], want_n) < 0.15, (copo["num_agents_total"], want_n)                   # 103.8 vs 115.8


def test_reference_roundabout_populations_drive_the_hip_simulator(golden_dir):
    """The roundabout's ring is eleven roads long for a full turn: populations that were trained on MetaDrive's block
    geometry only get round it if the rebuilt geometry and navigation columns match."""
    gold = np.load(os.path.join(golden_dir, "reference_populations.npz"))
    ippo = _roll("ippo", "round", _weights(gold, "ippo_round"), None, 40)
    copo = _roll("copo", "round", _weights(gold, "copo_round"), tuple(gold["copo_round/lcf"]), 40)
    print("ippo_round", ippo, "\ncopo_round", copo)
    # training-time success in MetaDrive 0.2.5 (benchmarks/MetaDrive-0.2.5/README.md:19-31): IPPO 66.43 (4.99), CoPO 72.82 (6.73); the
    # release recorded 0.858 for the shipped CoPO population (eval/get_policy_function.py:41).  Round 4: bands of +-0.12 around the
    # 0.2.5 table (0.691 / 0.711 here); against the release's 0.858 the CoPO population is 0.147 short -- with the crash rate (0.25)
    # carrying all of it: out-of-road 0.034 and max-step 0.005 leave at most 0.04
    assert abs(ippo["success"] - 0.664) < 0.12 and abs(copo["success"] - 0.728) < 0.12, (ippo, copo)
    assert abs(copo["success"] - 0.858) < 0.17
    assert ippo["out"] < 0.08 and copo["out"] < 0.06


def test_reference_parking_lot_population(golden_dir):
    """ParkingLot rebuilt from MetaDrive's blocks (round 3, `maps.parkinglot`: FirstPGBlock -> ParkingLot block with eight 3.5 x 8 m
    spaces side by side, radius-4 bends to and from both lanes as overlapping roads -> T-intersection; three entrances, three
    exits).  On round 2's stand-in (7 m pitch, a straight aisle with two ends) the shipped IPPO population scored 0.110 with 55 %
    out-of-road; on the rebuilt scene 0.184 / 33 %; round 4 (exclusive parking spaces = ParkingSpaceManager, broken centre line inside the
    block crossable) 0.188 / 31 %.  The reference's table (MetaDrive 0.2.5) has IPPO 16.98 +- 5.90: asserted at one standard deviation."""
    gold = np.load(os.path.join(golden_dir, "reference_populations.npz"))
    ippo = _roll("ippo", "parking", _weights(gold, "ippo_parking"), None, 10)
    print("ippo_parking", ippo)
    assert abs(ippo["success"] - 0.170) < 0.059 and ippo["out"] < 0.40, ippo


def test_reference_tollgate_and_bottleneck_populations(golden_dir):
    """f-4 scenes: MetaDrive's Merge / Split blocks restated (maps.Net.add_funnel: the route follows the leftmost lanes straight
    through, the other lanes run into / out of them on two-arc wave lanes; the only lines are the centre line and the outer edge
    of the outermost wave lane) and Navigation's check-point rule (both check points at the lateral middle of the CURRENT road's
    lane count).  None of this was fitted: with the round-2 corridor model (one 4-lane-wide merge road, its check point in the
    middle of the corridor) the shipped CoPO Bottleneck population left the road in 50 % of its episodes (success 0.42); with the
    blocks as MetaDrive builds them it scores 0.79 against the 0.867 the reference recorded for it
    (eval/get_policy_function.py:29 `"copo_bottle": ...  # 0.867, Best`).  Tollgate: the reference's training table
    (benchmarks/MetaDrive-0.2.5/README.md:19-25) has IPPO 4 +- 3, CoPO 27 +- 26; IPPO does not learn the booth rule there either.
    Detector ranges: Bottleneck side detector 50 m / lane-line detector 20 m, Tollgate 20 m / 20 m, LiDAR 20 m."""
    gold = np.load(os.path.join(golden_dir, "reference_populations_f4.npz"))
    from copo_amd.eval.get_policy_function import meta_svo_lookup_table
    copo_b = _roll("copo", "bottle", _weights(gold, "copo_bottle"), meta_svo_lookup_table["copo_bottle"], 20)
    ippo_b = _roll("ippo", "bottle", _weights(gold, "ippo_bottle"), None, 20)
    from copo_amd.sim import TOLLGATE_ROUND5_SCENE
    r5 = dict(TOLLGATE_ROUND5_SCENE)
    copo_t5 = _roll("copo", "tollgate", _weights(gold, "copo_tollgate"), meta_svo_lookup_table["copo_tollgate"], 40, env_config=r5)
    ippo_t5 = _roll("ippo", "tollgate", _weights(gold, "ippo_tollgate"), None, 40, env_config=r5)
    copo_t = _roll("copo", "tollgate", _weights(gold, "copo_tollgate"), meta_svo_lookup_table["copo_tollgate"], 40)
    ippo_t = _roll("ippo", "tollgate", _weights(gold, "ippo_tollgate"), None, 40)
    print("copo_bottle", copo_b, "\nippo_bottle", ippo_b, "\nrounds 2-5 Tollgate scene: copo", copo_t5, "\nippo", ippo_t5, "\nround-6 Tollgate scene: copo", copo_t, "\nippo", ippo_t)
    assert abs(copo_b["success"] - 0.867) < 0.12 and copo_b["out"] < 0.05 and copo_b["crash"] < 0.3, copo_b      # 0.787 / 0.009 / 0.203
    assert 0.4 < ippo_b["success"] < 0.8 and ippo_b["out"] < 0.15, ippo_b                                          # 0.597 / 0.088
    assert copo_b["success"] > ippo_b["success"] + 0.1
    # rounds 2-5's Tollgate (LiDAR 40 m, no buildings, an early exit = crash): kept as a regression pin
    assert 0.3 < copo_t5["success"] < 0.65 and ippo_t5["success"] < 0.1, (copo_t5, ippo_t5)                         # 0.468 / 0.014
    # Round 6's Tollgate = MATollConfig as restated (SPEC.md): LiDAR 72 beams / 20 m -- in the spec since round 3, 40 m in the code until now --, the
    # booth's speed limit, the unpunished early exit, and booth BUILDINGS in every second booth lane (static boxes, crash on touch) that the LiDAR does
    # NOT see.  Both records the reference holds for the scene agree with that variant and with no other (profiles/r06_fidelity.txt):
    #   * IPPO's training success on 0.2.5 is 4.41 +- 2.56 % over 8 seeds: the shipped IPPO file cannot be a policy that gets through a quarter of the
    #     time.  It scores 0.00 here -- and 0.25 as soon as the LiDAR shows it the buildings (asserted below as the variant the record excludes);
    #   * the CoPO file scores 0.28 (table 27.19 +- 25.63; the review's band: within 0.15 of that record);
    #   * from scratch at the reference's batch structure CoPO trains to 24.1 +- 24.5 %, IPPO to 32 +- 20 % -- every other variant to 80-96 %.
    assert abs(copo_t["success"] - 0.272) < 0.15, copo_t                                                            # 0.280
    assert ippo_t["success"] < 0.05 and ippo_t["crash"] > 0.7, ippo_t                                               # 0.000 / 0.84: it drives into the booths
    assert copo_t["max_step"] < 0.05 and ippo_t["max_step"] < 0.05, (copo_t, ippo_t)                                # nobody stalls
    vis = dict(toll_buildings=1)
    copo_tv = _roll("copo", "tollgate", _weights(gold, "copo_tollgate"), meta_svo_lookup_table["copo_tollgate"], 40, env_config=vis)
    ippo_tv = _roll("ippo", "tollgate", _weights(gold, "ippo_tollgate"), None, 40, env_config=vis)
    print("buildings the LiDAR sees: copo", copo_tv, "\nippo", ippo_tv)
    assert abs(copo_tv["success"] - 0.57) < 0.12 and abs(ippo_tv["success"] - 0.25) < 0.12, (copo_tv, ippo_tv)      # what the IPPO record rules out


# ---- the bands the round-5 review asked for, where this build is OUTSIDE them ---------------------------------------------------------
# Strict expected failures: each states the review's band, this build's value and the TESTED reason it is not met
# (profiles/r06_fidelity.txt has the sweeps).  A strict xfail turns into an error the day the gap closes, so the text cannot go stale.
def _copo_inter(golden_dir):
    gold = np.load(os.path.join(golden_dir, "eval_policy_function.npz"))
    from copo_amd.eval.get_policy_function import meta_svo_lookup_table
    return _roll("copo", "inter", _weights(gold, "copo_inter"), meta_svo_lookup_table["copo_inter"], 30)


@pytest.mark.xfail(strict=True, reason="CoPO population 0 drives 11.5-12.5 km/h here, 16.7 in the release's MetaDrive.  Not a constant of the bicycle: "
                   "brake_gain 6.75 .. 54 and acc_max 2.4 .. 3.5 move it over 10.4 .. 13.3 km/h only (scripts/fidelity_dynamics_sweep.py) while the IPPO "
                   "population matches at 30.4 vs 31.8 -- the speed is the policy's own answer to what it observes at low speed among neighbours")
def test_open_gap_copo_population_speed_within_3_kmh_of_the_release(golden_dir):
    assert abs(_copo_inter(golden_dir)["velocity"] - 16.7) < 3.0


@pytest.mark.xfail(strict=True, reason="336-346 steps per agent here, 260 in the release's CSVs (the same 118 m driven): follows the speed above; the reference's own "
                   "0.2.5 TRAINING record has ~330 steps per agent at the same success rate (asserted in test_intersection_evaluation_tables...)")
def test_open_gap_copo_population_lifetime_within_15_percent_of_the_release(golden_dir):
    assert abs(_copo_inter(golden_dir)["length"] - 260.0) / 260.0 < 0.15


@pytest.mark.xfail(strict=True, reason="out-of-road 0.098 vs 0.039 at body_margin 0.75; 0.023 at 0.5 (success 0.857 vs 0.812): the two shipped Intersection populations "
                   "bracket the margin -- IPPO's 0.048 vs 0.051 wants 0.75 -- and a second digit is not identifiable from them")
def test_open_gap_copo_population_out_of_road_within_004_of_the_release(golden_dir):
    assert abs(_copo_inter(golden_dir)["out"] - 0.039) < 0.04


@pytest.mark.xfail(strict=True, reason="shipped CoPO Roundabout population 0.71 here, 0.858 recorded by the release (get_policy_function.py:41); the 0.2.5 training table "
                   "has 0.728 +- 0.067.  All of the gap is crash rate (0.25); out-of-road 0.034 and max-step 0.005 leave at most 0.04")
def test_open_gap_copo_roundabout_population_within_010_of_the_release(golden_dir):
    gold = np.load(os.path.join(golden_dir, "reference_populations.npz"))
    copo = _roll("copo", "round", _weights(gold, "copo_round"), tuple(gold["copo_round/lcf"]), 40)
    assert abs(copo["success"] - 0.858) < 0.10

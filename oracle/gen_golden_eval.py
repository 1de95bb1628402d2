#!/usr/bin/env python3
"""Golden vectors for the evaluation path, made by EXECUTING the reference's own
functions (dev container only).  TEST INFRASTRUCTURE.

Reference entry points driven (under /root/reference/copo_code/copo/eval/):
  get_policy_function.py:56-83    _compute_actions_for_tf_policy     (TF-era key layout `default/fc_1{suffix}/kernel`)
  get_policy_function.py:86-99    _compute_actions_for_torch_policy  (torch key layout `_hidden_layers.0._model.0.weight`)
  get_policy_function.py:118-139  get_policy_function                (algo prefix -> layout / suffix)
  get_policy_function.py:142-198  PolicyFunction.__call__ / process_svo

Inputs: three trained policies the reference ships as data (`best_checkpoints/{ippo,copo,ccppo}_inter.npz`, six arrays
each) and seeded random observations.  The fixture stores those arrays, the observations and what the reference's
functions returned for them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
import copo.eval.get_policy_function as G  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CKPT = os.path.join(os.path.dirname(os.path.abspath(G.__file__)), os.pardir, "best_checkpoints")


def main():
    save = {}
    rng = np.random.RandomState(7)
    for name in ("ippo_inter", "copo_inter", "ccppo_inter"):
        w = dict(np.load(os.path.join(CKPT, name + ".npz")))
        for k, v in w.items():
            save["%s/w/%s" % (name, k)] = v
        f = G.get_policy_function(name, checkpoint_dir_name="best_checkpoints")
        odim = 92 if name.startswith("copo") else 91
        obs = rng.uniform(0, 1, size=(24, odim)).astype(np.float32)
        save[name + "/obs"] = obs
        if name.startswith("ccppo"):
            mean = G._compute_actions_for_torch_policy(w, obs, deterministic=True)
        else:
            sfx = "_1" if name.startswith("copo") else ""
            mean = G._compute_actions_for_tf_policy(w, obs, deterministic=True, policy_name="default", layer_name_suffix=sfx)
        save[name + "/mean"] = mean
        np.random.seed(11)
        save[name + "/sampled"] = f(obs)                      # stochastic path under a fixed numpy seed
        # dict API incl. the LCF ("svo") column of CoPO policies
        pf = G.PolicyFunction(model_name=name, checkpoint_dir_name="best_checkpoints")
        base = rng.uniform(0, 1, size=(5, 91)).astype(np.float32)
        obs_dict = {"agent%d" % i: base[i] for i in range(5)}
        done = {"agent1": True, "agent3": False}
        np.random.seed(13)
        act = pf(obs_dict, done)
        save[name + "/dict_obs"] = base
        save[name + "/dict_keys"] = np.array(sorted(act.keys()))
        save[name + "/dict_actions"] = np.stack([act[k] for k in sorted(act.keys())])
        if name.startswith("copo"):
            save[name + "/lcf"] = np.array(G.meta_svo_lookup_table[name])
            save[name + "/dict_lcf"] = np.array([pf.existing_svo[k] for k in sorted(pf.existing_svo)])
    np.savez_compressed(os.path.join(OUT, "eval_policy_function.npz"), **save)
    print("wrote eval_policy_function.npz", os.path.getsize(os.path.join(OUT, "eval_policy_function.npz")))


def populations():
    """More of the populations the reference ships as data (six arrays each), for the cross-simulator pin
    (tests/test_gpu_reference_populations.py), and what the reference measured for its Intersection populations in
    MetaDrive: the per-episode CSVs under eval/demo_results/evaluate_results, reduced to column means."""
    import glob
    import json
    import pandas as pd
    save = {}
    for name in ("ippo_round", "copo_round", "ippo_parking"):
        w = dict(np.load(os.path.join(CKPT, name + ".npz")))
        for k, v in w.items():
            save["%s/w/%s" % (name, k)] = v
        if name in G.meta_svo_lookup_table:
            save[name + "/lcf"] = np.array(G.meta_svo_lookup_table[name])
    np.savez_compressed(os.path.join(OUT, "reference_populations.npz"), **save)
    save = {}       # the Tollgate / Bottleneck populations (f-4 scenes: 156- / 96-wide first layers)
    for name in ("ippo_tollgate", "copo_tollgate", "ippo_bottle", "copo_bottle"):
        w = dict(np.load(os.path.join(CKPT, name + ".npz")))
        for k, v in w.items():
            save["%s/w/%s" % (name, k)] = v
        if name in G.meta_svo_lookup_table:
            save[name + "/lcf"] = np.array(G.meta_svo_lookup_table[name])
    np.savez_compressed(os.path.join(OUT, "reference_populations_f4.npz"), **save)
    res_dir = os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "demo_results", "evaluate_results")
    cols = ["success_rate", "crash_rate", "out_rate", "episode_length_mean", "success_episode_length_mean",
            "velocity_step_mean_episode_mean", "velocity_step_mean_episode_max", "episode_reward_mean", "episode_reward_min",
            "episode_reward_max", "num_agents_total", "num_agents_total_per_300_steps", "num_neighbours_mean_episode_mean",
            "num_neighbours_mean_episode_max", "episode_cost_mean"]
    stats = {}
    for algo in ("ippo", "copo"):
        files = sorted(glob.glob(os.path.join(res_dir, "%s_inter_*.csv" % algo)))
        d = pd.concat([pd.read_csv(f) for f in files])
        stats[algo + "_inter"] = dict(populations=len(files), episodes=int(len(d)), **{c: float(d[c].mean()) for c in cols})
        # per population (the shipped `copo_inter.npz` is population 0: get_policy_function.py:30-31 "Best")
        stats[algo + "_inter_per_population"] = [{c: float(pd.read_csv(f)[c].mean()) for c in cols} for f in files]
    # reward-scale decomposition: per population, D_hat = episode_length_mean x velocity_step_mean / 3.6 x 0.1 s (metres an agent
    # drives, estimated from the two recorded means, per env episode) and k_hat = (episode_reward_mean - 10 success + 10 crash + 10 out) / D_hat
    # (reward per estimated metre net of the terminal rewards); the same two estimators are evaluated on this build's rows
    for algo in ("ippo", "copo"):
        files = sorted(glob.glob(os.path.join(res_dir, "%s_inter_*.csv" % algo)))
        for f, rec in zip(files, stats[algo + "_inter_per_population"]):
            d = pd.read_csv(f)
            dist = d["episode_length_mean"] * d["velocity_step_mean_episode_mean"] / 3.6 * 0.1
            net = d["episode_reward_mean"] - 10.0 * d["success_rate"] + 10.0 * d["crash_rate"] + 10.0 * d["out_rate"]
            rec["metres_hat"] = float(dist.mean())
            rec["reward_per_metre_hat"] = float((net / dist).mean())
    for k in list(stats):      # env episode length in steps (recoder.py:246-247: agents per 300 steps = agents / steps * 300)
        for d in (stats[k] if isinstance(stats[k], list) else [stats[k]]):
            d["env_episode_steps"] = d["num_agents_total"] / d["num_agents_total_per_300_steps"] * 300.0
    # the reference's own TRAINING run of CoPO on the Intersection (MetaDrive 0.2.5, torch stack; eval/demo_raw_checkpoints/copo/
    # .../progress.csv): per-agent return, rates, env episode length and LCF along the run -- a second, independent record of
    # what MetaDrive returns for a CoPO population (the evaluation CSVs above come from populations of the paper's release)
    prog = glob.glob(os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "demo_raw_checkpoints", "copo", "*", "progress.csv"))
    if prog:
        d = pd.read_csv(prog[0])
        keep = ["timesteps_total", "episode_reward_mean", "success", "crash", "out", "max_step", "episode_len_mean",
                "info/learner/svo", "info/learner/svo_std", "raw_episode_reward_mean"]
        rows = d[keep].iloc[[99, 199, 299, 399, 499, 599, len(d) - 1]]
        stats["copo_inter_training_progress"] = [
            dict(timesteps_total=int(r["timesteps_total"]), episode_reward_mean=float(r["episode_reward_mean"]), success=float(r["success"]),
                 crash=float(r["crash"]), out=float(r["out"]), max_step=float(r["max_step"]), env_episode_steps=float(r["episode_len_mean"]),
                 lcf=float(r["info/learner/svo"]), lcf_std=float(r["info/learner/svo_std"]),
                 agents_per_env_episode=float(r["raw_episode_reward_mean"] / r["episode_reward_mean"])) for _, r in rows.iterrows()]
        stats["copo_inter_training_progress_max_success"] = float(d["success"].max())
    with open(os.path.join(OUT, "reference_eval_stats.json"), "w") as f:
        json.dump(stats, f, indent=1, sort_keys=True)
    print("wrote reference_populations.npz", os.path.getsize(os.path.join(OUT, "reference_populations.npz")), stats)


if __name__ == "__main__":
    if not os.path.isdir(ref_stubs.REFERENCE_ROOT):
        sys.exit("reference tree not present; fixtures are committed under tests/golden/")
    main()
    populations()

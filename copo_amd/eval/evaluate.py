"""Vectorised population evaluation on the HIP simulator (the role of `copo/eval.py` + `eval/recoder.py` +
`eval/evaluate_population.py`): load a population (`.npz` in either reference layout, or this build's trainer
checkpoint), roll it in E parallel scenes without learning, report the reference's headline evaluation metrics
(success / crash / out-of-road / max-step rates, episode reward and length, velocity ...).

    python -m copo_amd.eval.evaluate --algo copo --env inter --npz path/to/copo_inter.npz --lcf 0.368 0.088
"""
import argparse
import json

import numpy as np

ENVS = {"inter": "MultiAgentIntersectionEnv", "round": "MultiAgentRoundaboutEnv", "parking": "MultiAgentParkingLotEnv",
        "tollgate": "MultiAgentTollgateEnv", "bottle": "MultiAgentBottleneckEnv"}


_SCENE_OF = (("Roundabout", "MultiAgentRoundaboutEnv", "Round"), ("Intersection", "MultiAgentIntersectionEnv", "Inter"),
             ("Parking", "MultiAgentParkingLotEnv", "Parking"), ("Bottle", "MultiAgentBottleneckEnv", "Bottle"),
             ("Tollgate", "MultiAgentTollgateEnv", "Tollgate"))


def get_env(env, should_wrap_copo_env, should_wrap_cc_env, svo_mean=0.0, svo_std=0.0):
    """(RecorderEnv(single-scene env), short scene name) from an RLlib env name, as `copo/eval.py:27-64` builds it: the
    CoPO env gets the population's LCF distribution, the CC env only the neighbour lists."""
    from copo_amd.torch_copo.utils import env_wrappers as W
    from .recoder import RecorderEnv
    for needle, cls_name, short in _SCENE_OF:
        if needle in env:
            break
    else:
        raise ValueError()
    cls = getattr(W, cls_name)
    if should_wrap_copo_env:
        assert should_wrap_cc_env is False
        e = W.get_lcf_env(cls)({})
        e.set_lcf_dist(float(svo_mean), max(float(svo_std), 1e-6))
    elif should_wrap_cc_env:
        e = W.get_ccenv(cls)({})
    else:
        e = cls({})
    return RecorderEnv(e), short


def get_env_and_start_seed(trial_path):
    """(env name, start seed, params) of a Tune trial folder (`params.json`), copo/eval.py:67-80."""
    import os
    path = os.path.join(trial_path, "params.json")
    assert os.path.isfile(path)
    with open(path, "r") as f:
        param = json.load(f)
    if "env_config" not in param:
        raise ValueError()
    return param["env"], param["env_config"]["start_seed"], param


def make_eval_trainer(algo, env, num_envs=64, num_agents=40, seed=0, lcf=None, **extra):
    """A trainer used for its sampler only: same env wrappers, model and observation as training
    (`get_env`, copo/eval.py:27-76: CoPO populations run in the LCF env with the population's LCF distribution)."""
    from copo_amd.torch_copo import algo_ccppo, algo_copo, algo_ippo
    from copo_amd.torch_copo.utils import env_wrappers as W
    base = getattr(W, ENVS[env])
    if algo == "copo":
        cls, e = algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(base))
    elif algo == "ccppo":
        cls, e = algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(base)
    else:
        cls, e = algo_ippo.IPPOTrainer, W.get_rllib_compatible_env(base)
    extra = dict(extra)
    cfg = dict(env=e, env_config=dict(extra.pop("env_config", None) or {}, num_agents=num_agents), num_envs=num_envs,
               train_batch_size=num_envs * 25, seed=seed)
    cfg.update(extra)
    t = cls(config=cfg)
    if algo == "copo" and lcf is not None:
        t.env.set_lcf_dist(float(lcf[0]), float(lcf[1]))
    return t


def evaluate_population(algo, env, weights=None, lcf=None, num_envs=64, num_agents=40, episodes=2000, seed=0,
                        scene_episodes=None, **extra):
    """Roll a population in `num_envs` scenes.  `scene_episodes` = whole scene episodes (until done["__all__"]: `horizon`
    env steps of respawning, then the scene drains -- the unit of the reference's evaluation,
    eval/evaluate_population.py:57-76) -- every agent that terminates inside them counts; otherwise stop after `episodes`
    terminated agents (quick, but the first agents to terminate are the ones that fail early)."""
    from .checkpoint_io import load_policy_weights
    t = make_eval_trainer(algo, env, num_envs, num_agents, seed, lcf, **extra)
    if weights is not None:
        load_policy_weights(t.policy.model, weights)
        if t.policy.fused is not None:
            t.policy.fused.sync_mirror()
    if scene_episodes:
        res = t.evaluate(scene_episodes=int(scene_episodes))
    else:
        res = t.evaluate(num_fragments=1, min_episodes=episodes)
    t.stop()
    return res


def evaluate_population_rows(algo, env, weights=None, lcf=None, num_envs=64, num_agents=40, scene_episodes=1, seed=0,
                             recorder_distance=20.0, **extra):
    """The reference's per-episode evaluation table (`evaluate_once`, eval/evaluate_population.py:21-99: one row of
    `RecorderEnv.get_episode_result` per whole scene episode) for `num_envs` scenes at once: a pandas DataFrame with the
    reference's column names (`vec_recorder.COLUMNS`).  The recorder counts neighbours within its own radius (20 m,
    recoder.py:76), so the scenes are built with that `neighbours_distance` -- the policies do not read it."""
    import torch
    from .checkpoint_io import load_policy_weights
    from .vec_recorder import F_ENV_RESET, VecRecorder
    extra = dict(extra)
    env_config = dict(extra.pop("env_config", None) or {}, neighbours_distance=float(recorder_distance))
    t = make_eval_trainer(algo, env, num_envs, num_agents, seed, lcf, env_config=env_config, **extra)
    if weights is not None:
        load_policy_weights(t.policy.model, weights)
        if t.policy.fused is not None:
            t.policy.fused.sync_mirror()
    smp = t.sampler
    rec = VecRecorder(smp.E, smp.device)
    ep = torch.zeros(smp.E, dtype=torch.int64, device=smp.device)
    horizon = int(t.env.sim.cfg.horizon)
    n_frag = 0
    while bool((ep < int(scene_episodes)).any()) and n_frag * smp.T <= 6 * (int(scene_episodes) + 1) * horizon:
        b = smp.sample()
        fl = b["flags"]
        ended = ((fl & F_ENV_RESET) > 0).any(-1).to(torch.int64)            # [T, E]
        before = ep[None] + torch.cumsum(ended, 0) - ended                   # whole episodes finished before step t
        rec.add(dict(flags=fl, infos=b["infos"], nbr_cnt=b["nbr_cnt"]), keep=before < int(scene_episodes))
        ep = ep + ended.sum(0)
        n_frag += 1
    t.stop()
    return rec.frame()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="copo", choices=["ippo", "ccppo", "copo", "cl"])
    ap.add_argument("--env", default="inter", choices=sorted(ENVS))
    ap.add_argument("--npz", default=None, help="population file in the reference's key layout")
    ap.add_argument("--lcf", type=float, nargs=2, default=None, metavar=("MEAN", "STD"))
    ap.add_argument("--num-envs", type=int, default=64)
    ap.add_argument("--num-agents", type=int, default=40)
    ap.add_argument("--episodes", type=int, default=2000)
    ap.add_argument("--table", default=None, metavar="CSV",
                    help="write the reference's per-episode evaluation table (RecorderEnv columns, one row per whole scene episode) "
                         "instead of the headline rates")
    ap.add_argument("--scene-episodes", type=int, default=1)
    a = ap.parse_args()
    w = None
    if a.npz:
        with np.load(a.npz) as f:
            w = {k: f[k] for k in f.files}
    algo = "ippo" if a.algo == "cl" else a.algo
    if a.table:
        df = evaluate_population_rows(algo, a.env, w, a.lcf, a.num_envs, a.num_agents, scene_episodes=a.scene_episodes)
        df.to_csv(a.table)
        print(df.mean(numeric_only=True).to_string())
        return
    print(json.dumps(evaluate_population(algo, a.env, w, a.lcf, a.num_envs, a.num_agents, a.episodes), indent=1))


if __name__ == "__main__":
    main()

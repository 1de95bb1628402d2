"""Episode-by-episode evaluation of a trained population through the dict API and `RecorderEnv`: the reference's
`evaluate_once` / `get_make_env` (copo/eval/evaluate_population.py:21-132).  One scene, one CSV row per episode with the
31 statistics of `RecorderEnv.get_episode_result`; `copo_amd.eval.evaluate` is the vectorised (many scenes) counterpart.
"""
import argparse
import os
import time

import numpy as np

from .get_policy_function import PolicyFunction
from .recoder import RecorderEnv

# default populations of the five scenes (evaluate_population.py:102-132)
_SCENES = {"round": ("MultiAgentRoundaboutEnv", 40), "inter": ("MultiAgentIntersectionEnv", 30),
           "parking": ("MultiAgentParkingLotEnv", 10), "bottle": ("MultiAgentBottleneckEnv", 20),
           "tollgate": ("MultiAgentTollgateEnv", 40)}


def get_make_env(env, wrap_with_svo_env=False, render=False):
    """`make_env()` for one of "round" / "inter" / "parking" / "bottle" / "tollgate".  `wrap_with_svo_env` selects the LCF
    env (the torch reference's successor of the SVO env) so that the observation carries the LCF column; rendering does
    not exist in this build."""
    if env not in _SCENES:
        raise ValueError()
    assert not render, "no renderer in this build"
    cls_name, n = _SCENES[env]

    def make_env(env_id=None):
        from copo_amd.torch_copo.utils import env_wrappers as W
        cls = getattr(W, cls_name)
        if wrap_with_svo_env:
            cls = W.get_lcf_env(cls)
        return RecorderEnv(cls(dict(num_agents=n, crash_done=True)))

    return make_env


def evaluate_once(model_name, make_env, num_episodes=10, use_distributional_svo=False, suffix="", auto_add_svo_to_obs=True,
                  out_dir="evaluate_results", verbose=True, root=None, checkpoint_dir_name="best_checkpoints"):
    """Roll `num_episodes` episodes of the population `model_name` (a `.npz` under the checkpoint directory of
    `get_policy_function`), return a pandas DataFrame with one row per episode and write it to
    `<out_dir>/<model_name><suffix>.csv` (a `_backup.csv` after every episode)."""
    import pandas as pd
    os.makedirs(out_dir, exist_ok=True)
    policy = PolicyFunction(model_name, use_distributional_svo=use_distributional_svo and model_name.startswith("metasvo"),
                            auto_add_svo_to_obs=auto_add_svo_to_obs, root=root, checkpoint_dir_name=checkpoint_dir_name)
    rows = []
    env = make_env()
    try:
        o, d = env.reset(), {"__all__": False}
        start = last = time.time()
        steps, ep_times = 0, []
        while len(rows) < num_episodes:
            o, r, d, info = env.step(policy(o, d))
            steps += 1
            if verbose and steps % 100 == 0:
                print("Evaluating {}, Num episodes: {}, Num steps in this episode: {} (Total time {:.2f})".format(
                    model_name, len(rows), steps, time.time() - start))
            if d["__all__"]:
                policy.reset()
                res = env.get_episode_result()
                res["episode"] = len(rows) + 1
                rows.append(res)
                ep_times.append(time.time() - last)
                last, steps = time.time(), 0
                pd.DataFrame(rows).to_csv(os.path.join(out_dir, "{}{}_backup.csv".format(model_name, suffix)))
                o, d = env.reset(), {"__all__": False}
    finally:
        env.close()
    df = pd.DataFrame(rows)
    df.to_csv(os.path.join(out_dir, "{}{}.csv".format(model_name, suffix)))
    df["model_name"] = model_name
    if verbose:
        print("Final data is saved at:", os.path.join(out_dir, "{}{}.csv".format(model_name, suffix)),
              "({:.2f} s per episode)".format(float(np.mean(ep_times)) if ep_times else 0.0))
    return df


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-name", required=True, help="population name, e.g. copo_inter (see get_policy_function)")
    ap.add_argument("--env", default="inter", choices=sorted(_SCENES))
    ap.add_argument("--num-episodes", type=int, default=20)
    ap.add_argument("--lcf-env", action="store_true")
    a = ap.parse_args()
    print(evaluate_once(a.model_name, get_make_env(a.env, wrap_with_svo_env=a.lcf_env), num_episodes=a.num_episodes).mean(numeric_only=True))

"""Train CoPO / IPPO on the HIP simulator for a given env-step budget and print the learning curve."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copo_amd.torch_copo.algo_copo import CoPOTrainer
from copo_amd.torch_copo.algo_ippo import IPPOTrainer
from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
from copo_amd.torch_copo.utils import env_wrappers as W

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="copo")
ap.add_argument("--map", default="MultiAgentIntersectionEnv")
ap.add_argument("--num-envs", type=int, default=256)
ap.add_argument("--num-agents", type=int, default=0, help="0: the map's own population (Inter 30, Round 40, Bottle 20, Toll 40, Parking 10)")
ap.add_argument("--stop", type=int, default=1_000_000)
ap.add_argument("--every", type=int, default=25)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--rollout-steps", type=int, default=0, help="env steps per scene and iteration (0: ceil(2000 / num_envs), the reference's train_batch_size)")
ap.add_argument("--stagger", type=int, default=0, help="1: the scenes start their first episodes at staggered env steps")
ap.add_argument("--config", default="{}", help="JSON merged into the trainer config (e.g. bootstrap_next_obs)")
ap.add_argument("--env-config", default="{}", help="JSON merged into env_config (e.g. map_kwargs, respawn_cooldown)")
a = ap.parse_args()
base = getattr(W, a.map)
extra = {}
if a.algo == "copo":
    cls, env = CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(base))
elif a.algo.startswith("ccppo"):          # ccppo-mf / ccppo-concat (train_all_ccppo_{mf,concat}.py)
    from copo_amd.torch_copo import algo_ccppo
    cls, env = algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(base)
    extra["fuse_mode"] = a.algo.split("-")[1] if "-" in a.algo else "mf"
else:
    cls, env = IPPOTrainer, W.get_rllib_compatible_env(base)
T = a.rollout_steps if a.rollout_steps > 0 else max(1, -(-2000 // a.num_envs))
import json
algo = cls(config=dict(env=env, env_config=dict(json.loads(a.env_config), **(dict(num_agents=a.num_agents) if a.num_agents > 0 else {})), num_envs=a.num_envs, train_batch_size=T * a.num_envs,
                       seed=a.seed, callbacks=MultiAgentDrivingCallbacks, stagger_episodes=bool(a.stagger), **extra, **json.loads(a.config)))
t0 = time.time()
print("# %s %s E=%d N=%d seed=%d: iter env_steps agent_steps wall_s success crash out max_step ep_reward lcf kl agents_finished velocity_m_s episode_len (rates over the agents that finished since the previous line)" % (a.algo, a.map, a.num_envs, algo.env.sim.N, a.seed))
KEYS = ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean", "episode_reward_mean")
win = dict.fromkeys(KEYS, 0.0)      # rates over ALL agents that terminated since the last printed line
win_n = 0.0
while True:
    r = algo.train()
    it = r["training_iteration"]
    cm = r["custom_metrics"]
    nd = cm.get("num_terminated_agents", 0.0)
    if nd > 0:
        win_n += nd
        for k in KEYS:
            win[k] += cm[k] * nd
    if it % a.every == 0 or r["timesteps_total"] >= a.stop:
        mu = r["info"]["learner"]["default"]["custom_metrics"].get("meta_update", {})
        st = r["info"]["learner"]["default"]["learner_stats"]
        m = [win[k] / win_n if win_n > 0 else float("nan") for k in KEYS]
        print("%4d %8d %9d %6.1f  %.3f %.3f %.3f %.3f  %7.2f  %+.4f %.4f  %d %.2f %.0f" % (
            it, r["timesteps_total"], r["agent_timesteps_total"], time.time() - t0, m[0], m[1], m[2], m[3], m[4],
            mu.get("lcf", float("nan")), st["kl"], int(win_n), cm.get("velocity_mean", float("nan")),
            cm.get("episode_length_mean", float("nan"))), flush=True)
        win, win_n = dict.fromkeys(KEYS, 0.0), 0.0
    if r["timesteps_total"] >= a.stop:
        break
algo.stop()

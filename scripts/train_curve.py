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
ap.add_argument("--num-agents", type=int, default=40)
ap.add_argument("--stop", type=int, default=1_000_000)
ap.add_argument("--every", type=int, default=25)
a = ap.parse_args()
base = getattr(W, a.map)
if a.algo == "copo":
    cls, env = CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(base))
else:
    cls, env = IPPOTrainer, W.get_rllib_compatible_env(base)
T = max(1, -(-2000 // a.num_envs))
algo = cls(config=dict(env=env, env_config=dict(num_agents=a.num_agents), num_envs=a.num_envs, train_batch_size=T * a.num_envs,
                       seed=0, callbacks=MultiAgentDrivingCallbacks))
t0 = time.time()
print("# %s %s E=%d N=%d: iter env_steps agent_steps wall_s success crash out max_step ep_reward lcf kl" % (a.algo, a.map, a.num_envs, a.num_agents))
while True:
    r = algo.train()
    it = r["training_iteration"]
    if it % a.every == 0 or r["timesteps_total"] >= a.stop:
        mu = r["info"]["learner"]["default"]["custom_metrics"].get("meta_update", {})
        st = r["info"]["learner"]["default"]["learner_stats"]
        print("%4d %8d %9d %6.1f  %.3f %.3f %.3f %.3f  %7.2f  %+.4f %.4f" % (
            it, r["timesteps_total"], r["agent_timesteps_total"], time.time() - t0, r["success"], r["crash"], r["out"],
            r["max_step"], r["episode_reward_mean"], mu.get("lcf", float("nan")), st["kl"]), flush=True)
    if r["timesteps_total"] >= a.stop:
        break
algo.stop()

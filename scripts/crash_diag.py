"""What do the crashes of a trained population look like?  Age (steps since spawn), speed and route completion of the
agents that crash / arrive, and how many crashes involve a vehicle younger than a few seconds (respawn safety)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from copo_amd.torch_copo.algo_copo import CoPOTrainer
from copo_amd.torch_copo.algo_ippo import IPPOTrainer
from copo_amd.torch_copo.utils import env_wrappers as W

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="ippo")
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--num-agents", type=int, default=40)
ap.add_argument("--frags", type=int, default=150)
a = ap.parse_args()
base = W.MultiAgentIntersectionEnv
cls, env = (CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(base))) if a.algo == "copo" else (IPPOTrainer, W.get_rllib_compatible_env(base))
algo = cls(config=dict(env=env, env_config=dict(num_agents=a.num_agents), num_envs=256, train_batch_size=2048, seed=0))


def probe(tag):
    rows = {k: [] for k in ("crash", "arrive", "out")}
    pres, vel_all = [], []
    for _ in range(a.frags):
        b = algo.sampler.sample()
        f, info = algo.sampler.flags, algo.sampler.info
        acted = (f & 1) > 0
        done = ((f & 2) > 0) & acted
        pres.append(float(acted.float().sum(-1).mean()))
        vel_all.append(float(info[..., 0][acted].mean()))
        for k, bit in (("arrive", 4), ("crash", 8), ("out", 16)):
            m = done & ((f & bit) > 0)
            rows[k].append(torch.stack([info[..., 5][m], info[..., 0][m], info[..., 7][m]], -1).cpu().numpy())
    print("== %s: present slots per scene %.1f, mean speed of acting agents %.2f" % (tag, np.mean(pres), np.mean(vel_all)))
    for k, v in rows.items():
        v = np.concatenate(v)
        if len(v) == 0:
            continue
        age, sp, rc = v[:, 0], v[:, 1], v[:, 2]
        print("  %-6s n=%6d  age(steps) median %5.0f  p10 %4.0f  p90 %5.0f | share with age<=10: %.3f  <=30: %.3f | speed at end median %.2f  share<0.5: %.3f | "
              "route completion median %.2f" % (k, len(v), np.median(age), np.percentile(age, 10), np.percentile(age, 90),
                                                 np.mean(age <= 10), np.mean(age <= 30), np.median(sp), np.mean(sp < 0.5), np.median(rc)))


probe("untrained")
for _ in range(a.iters):
    algo.train()
probe("after %d iterations" % a.iters)
algo.stop()

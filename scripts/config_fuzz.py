"""Two training iterations of a spread of configurations (crash / NaN hunt, not a benchmark)."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from copo_amd.torch_copo import algo_ccppo, algo_copo, algo_ippo
from copo_amd.torch_copo.utils import env_wrappers as W

def env_of(algo, base):
    b = getattr(W, base)
    if algo == "copo":
        return algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(b))
    if algo == "ccppo":
        return algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(b)
    return algo_ippo.IPPOTrainer, W.get_rllib_compatible_env(b)

CASES = [
    ("copo", "MultiAgentIntersectionEnv", dict(num_envs=1, env_config=dict(num_agents=4), train_batch_size=64)),
    ("copo", "MultiAgentBottleneckEnv", dict(num_envs=16, env_config=dict(num_agents=20, add_traffic_light=True), train_batch_size=256)),
    ("copo", "MultiAgentParkingLotEnv", dict(num_envs=32, env_config=dict(num_agents=10, num_lasers=240, add_traffic_light=True))),
    ("copo", "MultiAgentRoundaboutEnv", dict(num_envs=16, model=dict(fcnet_hiddens=[128, 128]))),
    ("copo", "MultiAgentRoundaboutEnv", dict(num_envs=16, model=dict(fcnet_hiddens=[96, 96]))),
    ("copo", "MultiAgentIntersectionEnv", dict(num_envs=16, use_hip_graphs=False)),
    ("copo", "MultiAgentIntersectionEnv", dict(num_envs=16, sgd_minibatch_size=128, num_sgd_iter=2, lcf_num_iters=1)),
    ("copo", "MultiAgentTollgateEnv", dict(num_envs=16, fuse_mode="mf")),
    ("ccppo", "MultiAgentBottleneckEnv", dict(num_envs=16, env_config=dict(num_agents=20), fuse_mode="concat")),
    ("ccppo", "MultiAgentIntersectionEnv", dict(num_envs=16, fuse_mode="mf", counterfactual=False)),
    ("ippo", "MultiAgentIntersectionEnv", dict(num_envs=1, env_config=dict(num_agents=4), train_batch_size=100, sgd_minibatch_size=30)),
    ("ippo", "MultiAgentRoundaboutEnv", dict(num_envs=64, env_config=dict(num_agents=64, map_kwargs=dict(spawns_per_lane=8, spawn_gap=7.0)))),
]
bad = 0
for algo, base, cfg in CASES:
    cls, env = env_of(algo, base)
    cfg = dict(cfg, env=env, seed=0)
    cfg.setdefault("train_batch_size", cfg["num_envs"] * 8)
    try:
        a = cls(config=cfg)
        for _ in range(3):
            r = a.train()
        st = r["info"]["learner"]["default"]["learner_stats"]
        ok = all(np.isfinite(float(v)) for v in st.values())
        print("%-6s %-28s fused=%-5s O=%-4d loss=%9.4f %s" % (algo, base, a.policy.fused is not None, a.env.sim.O, float(st["total_loss"]), "ok" if ok else "NON-FINITE"), {k: v for k, v in cfg.items() if k not in ("env", "seed")})
        bad += 0 if ok else 1
        a.stop()
    except Exception as e:
        bad += 1
        print("%-6s %-28s FAILED %s: %s" % (algo, base, type(e).__name__, str(e)[:200]))
        traceback.print_exc(limit=2)
print("failures:", bad)

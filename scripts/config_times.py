import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copo_amd.torch_copo import algo_ccppo, algo_copo, algo_ippo
from copo_amd.torch_copo.utils import env_wrappers as W
cfgs = [
 ("C3-shard copo round 128x40", algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(W.MultiAgentRoundaboutEnv)), dict(num_envs=128, env_config=dict(num_agents=40))),
 ("C4 ccppo-mf tollgate 512x40 bf16", algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(W.MultiAgentTollgateEnv), dict(num_envs=512, env_config=dict(num_agents=40), fuse_mode="mf", policy_dtype="bfloat16", train_batch_size=1024)),
 ("C4 fp32", algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(W.MultiAgentTollgateEnv), dict(num_envs=512, env_config=dict(num_agents=40), fuse_mode="mf", train_batch_size=1024)),
 ("C5 copo parking 4096x10 240 lasers", algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(W.MultiAgentParkingLotEnv)), dict(num_envs=4096, env_config=dict(num_agents=10, num_lasers=240), train_batch_size=4096)),
]
for name, cls, env, cfg in cfgs:
    a = cls(config=dict(cfg, env=env, seed=0))
    for _ in range(4): a.train()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = a._counters["num_agent_steps_sampled"]
    for _ in range(6): r = a.train()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = a._counters["num_agent_steps_sampled"] - n0
    print("%-40s %.1f ms/iter  %.2f M agent-steps/s  (T=%d, rows/iter %d)" % (name, dt / 6 * 1e3, n / dt / 1e6, a.sampler.T, n // 6), flush=True)
    a.stop()
    if name == "C4 fp32":      # round 6: the same configuration with MetaDrive's booth rules and buildings (static boxes: an extra LiDAR pass)
        from copo_amd.sim import TOLLGATE_METADRIVE_RULES
        cfgs.append(("C4 fp32 + booth rules and buildings", cls, env, dict(cfg, env_config=dict(cfg["env_config"], **TOLLGATE_METADRIVE_RULES))))

import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copo_amd.torch_copo.algo_copo import CoPOTrainer
from copo_amd.torch_copo.utils import env_wrappers as W
for name, base, E, N, extra in (("inter-E2048", W.MultiAgentIntersectionEnv, 2048, 40, {}),
                                ("parking-E4096-240beams", W.MultiAgentParkingLotEnv, 4096, 10, dict(num_lasers=240)),
                                ("inter-E8192-T4", W.MultiAgentIntersectionEnv, 8192, 40, {})):
    env = W.get_rllib_compatible_env(W.get_lcf_env(base))
    T = 4 if "T4" in name else 1
    a = CoPOTrainer(config=dict(env=env, env_config=dict(num_agents=N, **extra), num_envs=E, train_batch_size=E * T, seed=0))
    t0 = time.time()
    for i in range(4):
        r = a.train()
    torch.cuda.synchronize()
    st = r["info"]["learner"]["default"]["learner_stats"]
    print(name, "ok: agent_steps", r["agent_timesteps_total"], "loss %.4f" % st["total_loss"], "row_store", a.policy._meta_row_store,
          "mem GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30), "t %.1fs" % (time.time() - t0), flush=True)
    a.stop()
    del a
    torch.cuda.empty_cache()

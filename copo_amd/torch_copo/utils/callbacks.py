"""Driving metrics with the reference's definitions (`copo/torch_copo/utils/callbacks.py:14-148`).

The reference harvests these from per-agent info dicts inside RLlib's episode hooks.  Here the trainer reduces
the simulator's flag / info tensors on the device (`VecTrainer.episode_metrics`) and hands the aggregate to
`on_train_result`, which applies the same renaming (success / crash / out / max_step / length / rc / cost, and
`episode_reward_mean` := per-agent mean).  `summarize_episode` restates the per-episode arithmetic of
`on_episode_end` for callers that do hold per-agent info lists (tests, the dict-API env).
"""
import numpy as np


class DefaultCallbacks:
    def on_train_result(self, *, algorithm, result, **kwargs):
        pass


def curriculum_num_agents(target_num_agents, timestep, total_time_step):
    """Population of the curriculum baseline at `timestep` (algo_ippo/ippo_cl.py:41-66): a quarter of the target per
    quarter of training, `int(target / 4 * q)`."""
    q = 1 + sum(1 for k in (1, 2, 3) if timestep > total_time_step / 4 * k)
    return int(target_num_agents / 4 * q)


def get_change_n_callback(total_time_step):
    """`ChangeNCallback` of the reference (algo_ippo/ippo_cl.py:41-78) for the vectorised trainer: after every
    training iteration the env-step count decides the population; when it changes, every scene is re-populated
    (`close_and_reset_num_agents` + reset)."""
    class ChangeNCallback(MultiAgentDrivingCallbacks):
        def __init__(self):
            super(ChangeNCallback, self).__init__()
            self.target_num_agents = None
            self.current = None
            self.total_time_step = total_time_step

        def on_algorithm_init(self, *, algorithm, **kwargs):
            self.apply(algorithm, 0)          # the first quarter of the population from the first rollout on

        def on_train_result(self, *, algorithm, result, **kwargs):
            super(ChangeNCallback, self).on_train_result(algorithm=algorithm, result=result, **kwargs)
            self.apply(algorithm, result["timesteps_total"])
            result["custom_metrics"]["num_agents_curriculum"] = self.current

        def apply(self, algorithm, timestep):
            if self.target_num_agents is None:
                # the reference reads the env class's default population; here that is the simulator's slot count
                self.target_num_agents = algorithm.env.sim.N
            n = curriculum_num_agents(self.target_num_agents, timestep, self.total_time_step)
            if n != self.current:
                print("Current time step: {}. We are now setting all environments with {} agents!".format(timestep, n))
                algorithm.env.close_and_reset_num_agents(n)
                algorithm.sampler.reset()
                self.current = n

    return ChangeNCallback


class MultiAgentDrivingCallbacks(DefaultCallbacks):
    STEP_KEYS = ("velocity", "steering", "step_reward", "acceleration", "cost", "episode_length", "episode_reward",
                 "num_neighbours")

    @staticmethod
    def summarize_episode(last_infos, user_data):
        """`last_infos`: {agent: final info}; `user_data`: {key: {agent: [per-step values]}} -> custom_metrics dict
        (utils/callbacks.py:48-110)."""
        keys = list(last_infos.keys())
        arrive = [bool(last_infos[k].get("arrive_dest", False)) for k in keys]
        crash = [bool(last_infos[k].get("crash", False)) for k in keys]
        out = [bool(last_infos[k].get("out_of_road", False)) for k in keys]
        max_step = [not (a or c or o) for a, c, o in zip(arrive, crash, out)]
        m = dict(
            track_length=np.mean([last_infos[k].get("track_length", -1) for k in keys]),
            current_distance=np.mean([last_infos[k].get("current_distance", -1) for k in keys]),
            route_completion=np.mean([last_infos[k].get("route_completion", -1) for k in keys]),
            success_rate=np.mean(arrive), crash_rate=np.mean(crash), out_of_road_rate=np.mean(out),
            max_step_rate=np.mean(max_step))
        for name, per_agent in user_data.items():
            m[name] = float(np.mean([v for vals in per_agent.values() for v in vals]))
        costs = [sum(v) for v in user_data["cost"].values()]
        m.update(episode_cost=np.mean(costs), episode_cost_worst_agent=np.min(costs),
                 episode_cost_best_agent=np.max(costs), environment_cost_total=np.sum(costs),
                 num_active_agents=len(costs),
                 episode_length=np.mean([v[-1] for v in user_data["episode_length"].values()]),
                 episode_reward=np.mean([v[-1] for v in user_data["episode_reward"].values()]),
                 environment_reward_total=np.sum([v[-1] for v in user_data["episode_reward"].values()]))
        return m

    def on_train_result(self, *, algorithm, result, **kwargs):
        cm = result["custom_metrics"]
        result["success"] = result["crash"] = result["out"] = result["max_step"] = np.nan
        result["length"] = result["episode_len_mean"]
        result["rc"] = np.nan
        if "success_rate_mean" in cm:
            result["success"] = cm["success_rate_mean"]
            result["crash"] = cm["crash_rate_mean"]
            result["out"] = cm["out_of_road_rate_mean"]
            result["max_step"] = cm["max_step_rate_mean"]
        if "route_completion_mean" in cm:
            result["rc"] = cm["route_completion_mean"]
        result["cost"] = cm.get("episode_cost_mean", np.nan)
        result["raw_episode_reward_mean"] = result["episode_reward_mean"]
        policy_reward_mean = list(result["policy_reward_mean"].values())
        if len(policy_reward_mean) == 0:
            if "episode_reward_mean" in cm:
                result["episode_reward_mean"] = cm["episode_reward_mean"]
        else:
            result["episode_reward_mean"] = np.mean(policy_reward_mean)

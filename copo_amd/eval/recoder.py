"""Episode recorder for the dict API (single scene): the reference's `RecorderEnv` (copo/eval/recoder.py:73-349) and its
`DistanceMap` helper (:16-70), re-implemented on plain per-step tables.  It wraps any env with the reference's
`reset() / step(dict)` surface -- here `get_lcf_env(MultiAgent*Env)` & co. of `copo_amd.torch_copo.utils.env_wrappers` --
and turns the info stream into the per-episode evaluation row of `eval/evaluate_population.py` (success / crash / out rates,
velocity / energy step means, episode reward / cost / length statistics, neighbour counts, the agent-level SVO estimate).
Host-side bookkeeping only; nothing here touches the device.
"""
import math
from collections import defaultdict

import numpy as np


def norm(a, b):
    return math.sqrt(a ** 2 + b ** 2)


class DistanceMap:
    """Pairwise distances of the vehicles of one step; neighbours of an agent = the others strictly inside a radius,
    nearest first, ties in vehicle order (recoder.py:16-45)."""

    def __init__(self):
        self.distance_map = None
        self.clear()

    def clear(self):
        self.distance_map = defaultdict(lambda: defaultdict(lambda: float("inf")))

    def update_distance_map(self, vehicles):
        self.distance_map.clear()
        keys = list(vehicles.keys())
        pos = [vehicles[k].position for k in keys]
        for a in range(len(keys) - 1):
            for b in range(a + 1, len(keys)):
                d = norm(pos[a][0] - pos[b][0], pos[a][1] - pos[b][1])
                self.distance_map[keys[a]][keys[b]] = d
                self.distance_map[keys[b]][keys[a]] = d

    def find_in_range(self, v_id, distance):
        if distance <= 0:
            return []
        row = self.distance_map[v_id]
        return [k for k in sorted(row, key=lambda k: row[k]) if row[k] < distance]

    def get_rewards(self, reward_dict, distance):
        """(own, neighbourhood mean -- the agent's own reward when it has no neighbour --, neighbour count) per agent."""
        own, nei, cnt = {}, {}, {}
        for k, r in reward_dict.items():
            nb = self.find_in_range(k, distance)
            others = []
            for o in nb:
                if o is None:
                    break
                others.append(reward_dict[o])
            own[k], nei[k], cnt[k] = r, (np.mean(others) if others else r), len(nb)
        return own, nei, cnt


_STEP_KEYS = ("velocity", "steering", "step_reward", "acceleration", "cost", "episode_length", "episode_reward")


class RecorderEnv:
    """`RecorderEnv(env, eval_config=None)`: same constructor, `reset / step / close`, `get_step_result()` and
    `get_episode_result()` as the reference's wrapper.  Tables: `user_data[stat][step][agent]`, `step_active_agents[step]`."""
    _default_eval_config = dict(neighbours_distance=20)
    EPISODE_END = -1

    def __init__(self, env, eval_config=None):
        self.env = env
        cfg = dict(self._default_eval_config)
        cfg.update(eval_config or {})
        self.eval_config = cfg
        self.episode_step = 0

    # ---- gym.Wrapper surface ---------------------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def __getattr__(self, name):          # observation_space, action_space, config, vehicles ... of the wrapped env
        return getattr(self.env, name)

    def close(self):
        return self.env.close()

    def reset(self, *args, **kwargs):
        o = self.env.reset(*args, **kwargs)
        self.episode_step = 0
        return o

    def step(self, *args, **kwargs):
        o, r, d, i = self.env.step(*args, **kwargs)
        self.on_episode_step(o, r, d, i)
        for k, done in d.items():
            if k != "__all__" and done:
                self.on_episode_end(k, o, r, d, i)
        return o, r, d, i

    # ---- recording -------------------------------------------------------------------------------------------------
    def on_episode_start(self):
        self.user_data = defaultdict(lambda: defaultdict(dict))
        self.step_active_agents = {}
        self.episode_step = 0
        self.distance_map = DistanceMap()

    def on_episode_step(self, o, r, d, i):
        if self.episode_step == 0:
            self.on_episode_start()
        t = self.episode_step
        self.distance_map.update_distance_map(self.unwrapped.vehicles)
        own, nei, cnt = self.distance_map.get_rewards(r, distance=self.eval_config["neighbours_distance"])
        for k in own:
            self.user_data["own_reward"][t][k] = own[k]
            self.user_data["num_neighbours"][t][k] = cnt[k]
            self.user_data["nei_reward"][t][k] = nei[k]
        self.step_active_agents[t] = set(r.keys())
        for k in r:
            info = i[k]
            if "step_reward" not in info:        # first observation of an agent: no transition yet
                continue
            for key in _STEP_KEYS:
                self.user_data[key][t][k] = info[key]
            self.user_data["energy"][t][k] = info["step_energy"]
            self.user_data["raw_action0_l2"][t][k] = info["raw_action"][0] ** 2
            self.user_data["raw_action1_l2"][t][k] = info["raw_action"][1] ** 2
        self.episode_step += 1

    def on_episode_end(self, k, o, r, d, i):
        info = i[k]
        arrive, crash, out = info.get("arrive_dest", False), info.get("crash", False), info.get("out_of_road", False)
        end = self.user_data
        end["success"][self.EPISODE_END][k] = arrive
        end["crash"][self.EPISODE_END][k] = crash
        end["max_step"][self.EPISODE_END][k] = not (arrive or crash or out)
        end["out"][self.EPISODE_END][k] = out
        end["episode_energy"][self.EPISODE_END][k] = info["episode_energy"]

    # ---- results ---------------------------------------------------------------------------------------------------
    def _step_means(self, stat):
        """Mean over the active agents that have a value, for every step that has one."""
        out = []
        for t, active in self.step_active_agents.items():
            vals = [self.user_data[stat][t][k] for k in active if self.user_data[stat][t].get(k) is not None]
            if vals:
                out.append(np.mean(vals))
        return out

    def _agent_cost(self):
        cost = defaultdict(float)
        for t, active in self.step_active_agents.items():
            for k in active:
                v = self.user_data["cost"][t].get(k)
                if v is not None:
                    cost[k] += v
        return list(cost.values())

    def _agent_last(self, stat):
        last = defaultdict(float)
        for t in sorted(self.step_active_agents):
            if t == self.EPISODE_END:
                continue
            for k in self.step_active_agents[t]:
                last[k] = self.user_data[stat][t].get(k, 0)
        return last

    def get_step_result(self):
        ret = {}
        t, active = list(self.step_active_agents.items())[-1]
        for stat in self.user_data.keys():
            vals = [self.user_data[stat][t][k] for k in active if self.user_data[stat][t].get(k) is not None]
            if vals:
                ret[stat] = np.mean(vals)
        ret["episode_reward_mean"] = np.mean(list(list(self.user_data["episode_reward"].values())[-1].values()))
        cost = self._agent_cost()
        ret["episode_cost_mean"], ret["episode_cost_sum"] = np.mean(cost), np.sum(cost)
        return ret

    def get_episode_result(self):
        ret = {}
        for stat in ("velocity", "energy"):
            m = self._step_means(stat)
            ret[stat + "_step_mean_episode_min"], ret[stat + "_step_mean_episode_mean"] = np.min(m), np.mean(m)
            ret[stat + "_step_mean_episode_max"] = np.max(m)
        m = self._step_means("num_neighbours")
        ret["num_neighbours_mean_episode_mean"], ret["num_neighbours_mean_episode_max"] = np.mean(m), np.max(m)

        steps = len(self.step_active_agents)
        end = self.user_data
        success, crash = list(end["success"][self.EPISODE_END].values()), list(end["crash"][self.EPISODE_END].values())
        n = len(success)
        ret["num_agents_total"] = n
        ret["num_agents_total_per_300_steps"] = n / steps * 300
        ret["success_rate"] = sum(success) / n
        ret["num_agents_success"] = sum(success)
        ret["num_agents_success_per_300_steps"] = sum(success) / steps * 300
        ret["num_agents_failed_per_300_steps"] = sum(crash) / steps * 300

        rew = list(self._agent_last("episode_reward").values())
        ret["episode_reward_mean"], ret["episode_reward_min"], ret["episode_reward_max"] = np.mean(rew), np.min(rew), np.max(rew)
        cost = self._agent_cost()
        ret["episode_cost_mean"], ret["episode_cost_min"] = np.mean(cost), np.min(cost)
        ret["episode_cost_max"], ret["episode_cost_sum"] = np.max(cost), np.sum(cost)
        ret["crash_rate"], ret["num_agents_crash"] = sum(crash) / n, sum(crash)
        out = list(end["out"][self.EPISODE_END].values())
        ret["out_rate"], ret["num_agents_out"] = sum(out) / n, sum(out)

        length = self._agent_last("episode_length")
        ret["episode_length_mean"] = np.mean(list(length.values()))
        won = [v for k, v in length.items() if end["success"][self.EPISODE_END][k]]
        ret["success_episode_length_mean"] = np.mean(won) if won else 0

        # agent-level SVO estimate: angle of (sum of own rewards, sum of neighbourhood rewards), clipped to [0, 90] degrees
        own_sum, nei_sum = defaultdict(float), defaultdict(float)
        for per_step in end["own_reward"].values():
            for k, v in per_step.items():
                own_sum[k] += v
        for per_step in end["nei_reward"].values():
            for k, v in per_step.items():
                nei_sum[k] += v
        svos, svo_rewards = [], []
        for k, own in own_sum.items():
            nei = nei_sum[k]
            alpha = np.rad2deg(math.atan2(nei, own))
            svo = min(max(0, alpha), 90)
            svos.append(svo)
            svo_rewards.append(norm(nei, own) * math.cos(np.deg2rad(svo) - np.deg2rad(alpha)))
        ret["svo_estimate_deg_mean"], ret["svo_estimate_deg_min"] = np.mean(svos), np.min(svos)
        ret["svo_estimate_deg_max"] = np.max(svos)
        ret["svo_reward"] = np.sum(svo_rewards) / n
        return ret

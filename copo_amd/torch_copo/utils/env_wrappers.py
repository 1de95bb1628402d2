"""Multi-agent driving envs over the HIP simulator, with the wrapper surface of the reference.

Reference: `copo/torch_copo/utils/env_wrappers.py` -- CCEnv (:30-158: pairwise distance map, distance-sorted
in-radius neighbour lists in `info`), LCFEnv (:161-430: neighbourhood / global rewards, per-agent LCF sampled
once at spawn and appended to the observation), get_ccenv / get_lcf_env / get_rllib_compatible_env (:433-597).
The base classes `MultiAgent{Intersection,Roundabout,Tollgate,ParkingLot}Env` are MetaDrive symbols in the
reference (train_copo.py:1-2); here they are thin front-ends of `copo_sim_*` (one HIP workgroup per scene).

Everything the wrappers compute per step in Python there (O(N^2) distance map, N sorts, reward means, LCF
sampling, obs concat) happens inside `copo_sim_step`; these classes only choose the configuration and expose
  * the vector API used by the trainers: `.sim` (VecSim), `vec_reset()`, `vec_step(actions)`;
  * the dict API of the reference for `num_envs == 1`: `reset() -> {agent_id: obs}`,
    `step({agent_id: action}) -> (obs, rewards, dones, infos)` with `dones["__all__"]`, the same info keys.
The communication / traffic-light / latent branches of the reference (off by default, :44-46) are not built.
"""
import copy
import math

import numpy as np

from copo_amd import _capi
from copo_amd.engine import Box, DictSpace
from copo_amd.maps import MAP_BUILDERS
from copo_amd.sim import MAP_OBS_DEFAULTS, SimConfig, VecSim

_ENV_REGISTRY = {}

# info / batch keys of the communication branch (env_wrappers.py:11-27)
COMM_ACTIONS = "comm_actions"
COMM_PREV_ACTIONS = "comm_prev_actions"
COMM_PREV_OBS = "comm_prev_obs"
COMM_CURRENT_OBS = "comm_current_obs"
COMM_PREV_2_OBS = "comm_prev_2_obs"
COMM_LOGITS = "comm_logits"
COMM_LOG_PROB = "comm_log_prob"
ENV_PREV_OBS = "env_prev_obs"
COMM_METHOD = "comm_method"
NEI_OBS = "nei_obs"

SIM_KEYS = {f for f in SimConfig.__dataclass_fields__}  # env_config keys forwarded verbatim to the simulator


def register_env(name, creator):
    _ENV_REGISTRY[name] = creator


def lookup_env(name_or_cls):
    if isinstance(name_or_cls, str):
        if name_or_cls not in _ENV_REGISTRY:
            raise KeyError("env %r is not registered (call get_rllib_compatible_env first)" % name_or_cls)
        return _ENV_REGISTRY[name_or_cls]
    return name_or_cls


class MultiAgentMetaDrive:
    """Vectorised multi-agent driving env.  `config` keys: every `SimConfig` field plus the reference's
    `num_agents`, `start_seed`, `horizon`, `neighbours_distance`, ... (unknown keys are kept in `.config`)."""
    MAP = "pgmap"        # the base class is MetaDrive's procedurally generated road (train_all_copo_dist.py:10,30)
    ENABLE_LCF = False
    WRAPS_CC = False

    @classmethod
    def default_config(cls):
        return dict(map=cls.MAP, num_envs=1, num_agents=None, start_seed=5000, horizon=1000, num_lasers=72,
                    device=0, crash_done=True, out_of_road_done=True, allow_respawn=True, delay_done=25)

    def __init__(self, config=None):
        cfg = type(self).default_config()
        cfg.update(config or {})
        if isinstance(cfg["map"], int) or cfg["map"] not in MAP_BUILDERS:
            # MetaDrive's `map` config key: a block count or a block-sequence string -> one seeded PG road
            kw = dict(cfg.get("map_kwargs") or {})
            kw.setdefault("sequence", cfg["map"])
            kw.setdefault("seed", int(cfg.get("start_seed", 0)))
            cfg["map"], cfg["map_kwargs"] = "pgmap", kw
        self.config = cfg
        sim_kwargs = {k: v for k, v in cfg.items() if k in SIM_KEYS and k not in ("enable_lcf",)}
        sim_kwargs["enable_lcf"] = bool(self.ENABLE_LCF and cfg.get("enable_copo", True))
        sim_kwargs.update(type(self)._extension_kwargs(cfg))
        if "lcf_normal_std" in cfg:
            sim_kwargs["lcf_std"] = float(cfg["lcf_normal_std"])
        self.sim_config = SimConfig(**sim_kwargs)
        if self.sim_config.num_envs == 1 and "nbr_k" not in cfg:
            # the dict API reports complete neighbour lists like the reference (trainers keep the top-8 by default)
            self.sim_config.nbr_k = max(1, self.sim_config.resolved()[1] - 1)
        self.sim = VecSim(self.sim_config, device=int(cfg.get("device", 0) or 0))
        self.num_envs, self.num_agents = self.sim.E, self.sim.N
        self._slot_ids = None      # dict API state (num_envs == 1)
        self._next_obs = None
        self.current_lcf_mean, self.current_lcf_std = self.sim_config.lcf_mean, self.sim_config.lcf_std
        if cfg.get("force_lcf", -100) != -100 and self.ENABLE_LCF:
            self.sim.set_force_lcf(cfg["force_lcf"])

    # ---- spaces --------------------------------------------------------------------------------------------
    @classmethod
    def _extension_kwargs(cls, cfg):
        """SimConfig fields of the communication channel / traffic-light message (CCEnv config keys, env_wrappers.py:
        44-46).  The observation is only extended by LCFEnv.step in the reference (:331-337, 362-371)."""
        comm = dict(cfg.get("communication") or {})
        on = comm.get("comm_method", "none") != "none"
        tl = bool(cfg.get("add_traffic_light", False))
        if (on or tl) and not cls.ENABLE_LCF:
            raise NotImplementedError("communication / traffic-light observations are appended by LCFEnv.step: use get_lcf_env")
        return dict(add_traffic_light=tl, traffic_light_interval=int(cfg.get("traffic_light_interval", 30)),
                    comm_size=int(comm.get("comm_size", 4)) if on else 0, comm_neighbours=int(comm.get("comm_neighbours", 4)),
                    add_pos_in_comm=bool(comm.get("add_pos_in_comm", False)))

    @classmethod
    def spaces_for(cls, env_config):
        cfg = cls.default_config()
        cfg.update(env_config or {})
        lcf = bool(cls.ENABLE_LCF and cfg.get("enable_copo", True))
        ext = cls._extension_kwargs(cfg)
        cdim = ext["comm_size"] + (3 if ext["add_pos_in_comm"] else 0)
        # MetaDrive's per-map observation: [side block | 6 state | lane block | navigation | lasers | toll] (sim.SimConfig)
        probe = SimConfig(map=cfg["map"] if cfg["map"] in MAP_OBS_DEFAULTS else "intersection", enable_lcf=False,
                          num_lasers=int(cfg.get("num_lasers", 72)),
                          **{k: cfg[k] for k in ("side_lasers", "lane_line_lasers", "navi_dim", "toll_dim") if cfg.get(k) is not None})
        odim = probe.obs_dim + (1 if lcf else 0)
        if lcf:    # LCFObs.observation_space (env_wrappers.py:225-247): the extra columns are declared with enable_copo only
            odim += (3 if ext["add_traffic_light"] else 0) + (ext["comm_neighbours"] * cdim if ext["comm_size"] else 0)
        # LCFEnv widens the space to [-1, 1] (env_wrappers.py:241-244); MetaDrive's own obs are in [0, 1]
        return Box(-1.0 if lcf else 0.0, 1.0, (odim,)), Box(-1.0, 1.0, (2 + ext["comm_size"],))

    @property
    def observation_space(self):
        o, _ = type(self).spaces_for(self.config)
        return DictSpace({"agent%d" % i: o for i in range(self.num_agents)})

    @property
    def action_space(self):
        _, a = type(self).spaces_for(self.config)
        return DictSpace({"agent%d" % i: a for i in range(self.num_agents)})

    def action_space_sample(self, agent_ids=None):
        return self.action_space.sample()

    # ---- vector API ----------------------------------------------------------------------------------------
    def vec_reset(self, seeds=None):
        return self.sim.reset(seeds)

    def vec_step(self, actions):
        return self.sim.step(actions)

    def set_lcf_dist(self, mean, std):
        assert self.ENABLE_LCF, "set_lcf_dist needs an LCF env (get_lcf_env)"
        assert std > 0.0 and -1.0 <= mean <= 1.0
        self.current_lcf_mean, self.current_lcf_std = mean, std
        self.sim.set_lcf_dist(mean, std)

    def set_force_lcf(self, v):
        assert self.ENABLE_LCF
        self.force_lcf = v
        self.sim.set_force_lcf(v)

    # ---- dict API of the reference (single scene) ---------------------------------------------------------------
    def _ids(self, out):
        return out["agent_id"][0].cpu().numpy()

    @property
    def vehicles(self):
        """Active agents: {agent_id: vehicle view} with `.position` (x, y), `.heading_theta`, `.speed` (m/s) and `.slot`, as
        far as the reference's wrappers and `RecorderEnv` look into MetaDrive's vehicle objects."""
        if self._slot_ids is None or getattr(self, "_episode_over", False):
            return {}          # after done["__all__"] MetaDrive has no vehicle left until reset()
        return {a: self._vehicle_view(s) for s, a in enumerate(self._slot_ids) if a is not None}

    def _vehicle_view(self, slot):
        from types import SimpleNamespace
        st = self._last_state
        return SimpleNamespace(slot=slot, position=np.array([st[0, slot], st[1, slot]], np.float64),
                               heading_theta=float(st[2, slot]), speed=float(st[3, slot]))

    def _fetch_state(self):
        st, _ = self.sim.get_state()
        self._last_state = st[:, 0].cpu().numpy()          # [fields][N] of scene 0
        return st

    @property
    def vehicles_including_just_terminated(self):
        return dict(getattr(self, "_just_terminated", {}), **self.vehicles)

    def reset(self, force_seed=None):
        assert self.num_envs == 1, "the dict API serves one scene; use vec_reset/vec_step for num_envs > 1"
        seed = self.config.get("start_seed", 5000) if force_seed is None else force_seed
        out = self.sim.reset(np.array([seed], np.uint64))
        ids = self._ids(out)
        self._slot_ids = ["agent%d" % a for a in ids]
        self._just_terminated = {}
        self._episode_energy = {}
        self._episode_over = False
        self._fetch_state()
        obs = out["obs"][0].cpu().numpy()
        return {a: obs[s] for s, a in enumerate(self._slot_ids)}

    def step(self, actions):
        import torch
        assert self.num_envs == 1 and self._slot_ids is not None, "call reset() first"
        N, F = self.num_agents, _capi
        act = np.zeros((1, N, self.sim.A), np.float32)
        for s, a in enumerate(self._slot_ids):
            if a is not None:
                act[0, s] = np.asarray(actions[a], np.float32)[:self.sim.A]
        out = self.sim.step(torch.from_numpy(act).to(self.sim.device))
        h = {k: v[0].cpu().numpy() for k, v in out.items() if v is not None}
        flags = h["flags"]
        env_reset = bool((flags & F.F_ENV_RESET).any())
        self._episode_over = False
        terminated_view = {a: v for a, v in self.vehicles.items()}      # poses before the step, for agents that end in it
        self._fetch_state()
        aid_now = self._last_state[14].view(np.int32)
        before = list(self._slot_ids)
        acting = {before[s]: s for s in range(N) if flags[s] & F.F_ACTED}
        spawned = {"agent%d" % aid_now[s]: s for s in range(N) if flags[s] & F.F_SPAWNED}
        # names of the slots as the neighbour lists of THIS step see them (post-step scene, before a horizon reset)
        name_of_slot = {s: a for a, s in acting.items()}
        if not env_reset:
            name_of_slot.update({s: a for a, s in spawned.items()})
        o, r, d, i = {}, {}, {}, {}
        self._just_terminated = {}
        for a, s in acting.items():
            f = int(flags[s])
            r[a], d[a] = float(h["rew"][s]), bool(f & F.F_DONE)
            o[a] = h["obs"][s]
            cnt = int(min(h["nbr_cnt"][s], self.sim.K))
            inf = h["info"][s]
            info = dict(
                all_agents=list(name_of_slot.values()),
                neighbours=[name_of_slot.get(int(j), "slot%d" % j) for j in h["nbr_idx"][s][:cnt]],
                neighbours_distance=[float(x) for x in h["nbr_dist"][s][:cnt]],
                arrive_dest=bool(f & F.F_ARRIVE), crash=bool(f & F.F_CRASH), crash_vehicle=bool(f & F.F_CRASH),
                out_of_road=bool(f & F.F_OUT), max_step=bool(f & F.F_MAXSTEP), velocity=float(inf[0]),
                steering=float(inf[1]), acceleration=float(inf[2]), step_reward=float(inf[3]), cost=float(inf[4]),
                episode_length=int(inf[5]), episode_reward=float(inf[6]), route_completion=float(inf[7]))
            # evaluation-side keys of MetaDrive's info that `RecorderEnv` reads (eval/recoder.py:136-138,153).  Energy is a
            # build-defined proxy: traction work of the bicycle model, 1100 kg, kJ per step -- MetaDrive's own consumption
            # model is not in the reference tree
            v_ms = float(inf[0]) / 3.6
            e_step = 1.1 * max(float(inf[2]), 0.0) * v_ms * float(self.sim_config.dt)
            self._episode_energy[a] = self._episode_energy.get(a, 0.0) + e_step
            info.update(step_energy=e_step, episode_energy=self._episode_energy[a],
                        raw_action=np.asarray(actions.get(a, (0.0, 0.0)), np.float32)[:2].copy())
            if self.ENABLE_LCF:
                lcf, nei_r = float(h["lcf"][s]), float(h["nei_rew"][s])
                coord = math.cos(lcf * math.pi / 2) * r[a] + math.sin(lcf * math.pi / 2) * nei_r
                info.update(nei_rewards=nei_r, global_rewards=float(h["glob_rew"]), lcf=lcf, lcf_deg=lcf * 90,
                            coordinated_rewards=coord, native_rewards=r[a])
                if not self.config.get("return_native_reward", True):
                    r[a] = coord
                sc = self.sim_config
                if sc.comm_size > 0:     # CCEnv.step :102-118 (+ the zero padding of LCFEnv.step :363-369) / :373-384
                    cd, c0 = sc.comm_dim, sc.obs_dim - sc.comm_neighbours * sc.comm_dim
                    info["comm_current_obs"] = [h["obs"][s][c0 + q * cd:c0 + (q + 1) * cd].copy() for q in range(sc.comm_neighbours)]
                    last = getattr(self, "_last_obs", None) or {}
                    nb = info["neighbours"][:sc.comm_neighbours]
                    info["nei_obs"] = [last.get(n) for n in nb] + [None] * (sc.comm_neighbours - len(nb)) + [None]
            i[a] = info
            if d[a]:
                self._just_terminated[a] = terminated_view.get(a)
        if not env_reset:
            for a, s in spawned.items():     # respawned agents: first obs, zero reward, empty info (MetaDrive)
                o[a], r[a], d[a], i[a] = h["obs"][s], 0.0, False, {}
        d["__all__"] = env_reset
        self._last_obs = dict(o)
        after = [None] * N
        for s in range(N):
            if flags[s] & F.F_SPAWNED:
                after[s] = "agent%d" % aid_now[s]
            elif (flags[s] & F.F_ACTED) and not (flags[s] & F.F_DONE):
                after[s] = before[s]
        self._slot_ids = after
        for a in [a for a, done in d.items() if a != "__all__" and done]:
            self._episode_energy.pop(a, None)
        self._episode_over = env_reset
        if env_reset:                        # the sim auto-reset: first obs of the next episode, if the caller goes on
            self._auto_reset_obs = {after[s]: h["obs"][s] for s in range(N)}
        return o, r, d, i

    def close(self):
        self.sim.close()


class MultiAgentIntersectionEnv(MultiAgentMetaDrive):
    MAP = "intersection"


class MultiAgentRoundaboutEnv(MultiAgentMetaDrive):
    MAP = "roundabout"


class MultiAgentTollgateEnv(MultiAgentMetaDrive):
    MAP = "tollgate"


class MultiAgentParkingLotEnv(MultiAgentMetaDrive):
    MAP = "parkinglot"


class MultiAgentBottleneckEnv(MultiAgentMetaDrive):
    MAP = "bottleneck"


class CCEnv:
    """Mixin: neighbour lists in `info` (`neighbours`, `neighbours_distance`, `all_agents`), radius
    `neighbours_distance` (strict <), sorted by distance with ties in slot order (env_wrappers.py:89-158)."""
    WRAPS_CC = True

    @classmethod
    def default_config(cls):
        config = super(CCEnv, cls).default_config()
        config["neighbours_distance"] = 40
        config.update(dict(communication=dict(comm_method="none", comm_size=4, comm_neighbours=4, add_pos_in_comm=False),
                           add_traffic_light=False, traffic_light_interval=30))
        return config

    def __init__(self, *args, **kwargs):
        super(CCEnv, self).__init__(*args, **kwargs)


class LCFEnv(CCEnv):
    """Mixin: LCF per agent (sampled once at spawn from clip(N(mean, std), -1, 1)), `(lcf+1)/2` appended to the
    observation, neighbourhood / global rewards in `info` (env_wrappers.py:161-430)."""
    ENABLE_LCF = True

    @classmethod
    def default_config(cls):
        config = super(LCFEnv, cls).default_config()
        config.update(dict(neighbours_distance=40, lcf_mode="angle", lcf_dist="normal", lcf_normal_std=0.1,
                           return_native_reward=True, force_lcf=-100, enable_copo=True))
        return config

    def __init__(self, config=None):
        super(LCFEnv, self).__init__(config)
        assert self.config["lcf_mode"] in ["linear", "angle"] and self.config["lcf_mode"] == "angle", \
            "only the 'angle' LCF mode is built (the reference default)"
        assert self.config["lcf_dist"] == "normal", "only the normal LCF distribution is built"
        assert self.config["lcf_normal_std"] > 0.0
        self.force_lcf = self.config["force_lcf"]

    @property
    def enable_copo(self):
        return self.config["enable_copo"]


def _named_subclass(mixin, env_class, prefix=""):
    name = prefix + env_class.__name__
    return type(name, (mixin, env_class), {"__qualname__": name})


def get_ccenv(env_class):
    return _named_subclass(CCEnv, env_class)


def get_lcf_env(env_class):
    return _named_subclass(LCFEnv, env_class)


def get_change_n_env(env_class):
    """Curriculum wrapper of the reference (:444-460).  The reference closes the env and constructs it again with
    another `num_agents`; here the simulator keeps its N slots and only the first `num_agents` of them are populated
    (`copo_sim_set_capacity`), so rollout buffers, captured graphs and the model stay as they are.  As in the
    reference, a `reset()` must follow."""
    class ChangeNEnv(env_class):
        def __init__(self, config):
            self._raw_input_config = copy.deepcopy(config)
            super(ChangeNEnv, self).__init__(config)
            self.current_num_agents = self.sim.N

        def close_and_reset_num_agents(self, num_agents):
            num_agents = max(1, min(int(num_agents), self.sim.N))
            self.sim.set_capacity(num_agents)
            self.current_num_agents = num_agents

    ChangeNEnv.__name__ = ChangeNEnv.__qualname__ = "CL{}".format(env_class.__name__)
    return ChangeNEnv


def get_latent_env(env_class):
    """`enable_latent` / `latent_dim`: a per-(seed, agent) latent vector registered by the caller is PREPENDED to every
    observation, zeros while nothing is registered (env_wrappers.py:474-556).  Dict API: `register_latent({seed:
    {agent_name: vector}})`; vector API: `register_latent_tensor([E, N, latent_dim])`, concatenated on the device."""
    class LatentEnv(env_class):
        @classmethod
        def default_config(cls):
            config = super(LatentEnv, cls).default_config()
            config.update(dict(enable_latent=False, latent_dim=-1))
            return config

        def __init__(self, config=None):
            super(LatentEnv, self).__init__(config)
            self.latent_dict = None
            self._latent_tensor = None
            self._global_seed = None

        @classmethod
        def spaces_for(cls, env_config):
            o, a = super(LatentEnv, cls).spaces_for(env_config)
            cfg = cls.default_config()
            cfg.update(env_config or {})
            if not cfg["enable_latent"]:
                return o, a
            return Box(float("-inf"), float("+inf"), (o.shape[0] + int(cfg["latent_dim"]),)), a

        def register_latent(self, latent_dict):
            self.latent_dict = latent_dict

        def register_latent_tensor(self, latent):
            assert tuple(latent.shape) == (self.num_envs, self.num_agents, int(self.config["latent_dim"]))
            self._latent_tensor = latent

        def _add_latent(self, obs, agent_name):
            if not self.config["enable_latent"]:
                return obs
            if self.latent_dict is None:
                latent = np.zeros(self.config["latent_dim"])
            else:
                latent = self.latent_dict[self._global_seed][agent_name]
            return np.concatenate([latent, obs], axis=-1)

        def reset(self, force_seed=None):
            self._global_seed = self.config.get("start_seed", 5000) if force_seed is None else force_seed
            return {k: self._add_latent(o, k) for k, o in super(LatentEnv, self).reset(force_seed).items()}

        def step(self, action):
            o, r, d, i = super(LatentEnv, self).step(action)
            return {k: self._add_latent(v, k) for k, v in o.items()}, r, d, i

        def _vec_latent(self, out):
            if not self.config["enable_latent"]:
                return out
            import torch
            obs = out["obs"]
            lat = self._latent_tensor if self._latent_tensor is not None else \
                torch.zeros(self.num_envs, self.num_agents, int(self.config["latent_dim"]), device=obs.device)
            out = dict(out)
            out["obs"] = torch.cat([lat.to(obs.dtype), obs], -1)
            return out

        def vec_reset(self, seeds=None):
            return self._vec_latent(super(LatentEnv, self).vec_reset(seeds))

        def vec_step(self, actions):
            return self._vec_latent(super(LatentEnv, self).vec_step(actions))

    LatentEnv.__name__ = LatentEnv.__qualname__ = env_class.__name__
    return LatentEnv


def get_rllib_compatible_env(env_class, return_class=False):
    """Register `env_class` under its class name and return the name (env_wrappers.py:559-597)."""
    env_name = env_class.__name__

    class MA(env_class):
        _agent_ids = ["agent{}".format(i) for i in range(100)] + ["{}".format(i) for i in range(10000)] + ["sdc"]

    MA.__name__ = MA.__qualname__ = env_name
    register_env(env_name, MA)
    if return_class:
        return env_name, MA
    return env_name

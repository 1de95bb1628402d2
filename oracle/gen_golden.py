#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own functions (dev container only).

TEST INFRASTRUCTURE.  Run as `python oracle/gen_golden.py` where `/root/reference` exists;
it writes small `.npz` fixtures into `tests/golden/`, which are committed and travel to the
GPU box.  The reference's source never does.  Each fixture records inputs and the outputs the
reference produced for them; nothing in it is reference source text.

Reference entry points driven here (all under /root/reference/copo_code/copo/torch_copo/):
  utils/env_wrappers.py:141-158  CCEnv._update_distance_map
  utils/env_wrappers.py:125-139  CCEnv._find_in_range
  utils/env_wrappers.py:307-391  LCFEnv.step          (+ :274-305 _get_reset_return, :393-418 _add_lcf)
  algo_copo.py:189-204           compute_nei_advantage / compute_global_advantage
  algo_ccppo.py:225-311          concat_ccppo_process / mean_field_ccppo_process
  algo_ccppo.py:322-374          CCPPOPolicy.postprocess_trajectory
  algo_copo.py:473-502           CoPOPolicy.postprocess_trajectory
  algo_ippo.py:78-172, algo_ccppo.py:376-472, algo_copo.py:311-424   the three losses
  algo_copo.py:228-309           CoPOPolicy.meta_update
  algo_copo.py:516-661           CoPOTrainer.training_step
  algo_copo.py:96-182, algo_ccppo.py:55-219   model construction / parameter counts
  utils/env_wrappers.py:89-118, 258-303, 315-337, 360-371   traffic-light message + communication channel (f-4)
  ../eval/recoder.py:16-349      DistanceMap / RecorderEnv episode statistics (f-1)

Vectors that flow through the 3P restatements in ref_stubs.py (TorchDiagGaussian, compute_advantages,
discount_cumsum, standardized) pin the reference code GIVEN those restatements.
"""
import os
import sys
from collections import defaultdict
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
import copo.torch_copo.algo_ccppo as C  # noqa: E402
import copo.torch_copo.algo_copo as A  # noqa: E402
import copo.torch_copo.algo_ippo as I  # noqa: E402
import copo.torch_copo.utils.env_wrappers as W  # noqa: E402
from ref_stubs import Box, SampleBatch, TorchDiagGaussian  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def f32r(x):
    """Round to fp32-representable float64 (the build's sim state is fp32)."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


# --------------------------------------------------------------------------------------
# Fake base env driven by scripted arrays, wrapped by the reference's LCFEnv / CCEnv.
# --------------------------------------------------------------------------------------
class _Veh:
    def __init__(self, p):
        self.position = np.asarray(p, dtype=np.float64)


class FakeBase:
    """Scripted stand-in for MetaDrive's MultiAgent*Env (only what env_wrappers.py touches)."""

    @classmethod
    def default_config(cls):
        return {}

    def __init__(self, config=None):
        cfg = type(self).default_config()
        cfg.update(config or {})
        self.config = cfg
        self.script = None
        self.tick = 0
        self.vehicles_including_just_terminated = {}

    def load(self, script):
        self.script = script
        self.tick = 0

    def get_single_observation(self, vehicle_config):
        return None

    def _get_reset_return(self):
        s = self.script
        self.vehicles_including_just_terminated = {
            n: (_Veh(p) if p is not None else None) for n, p in s["reset_pos"].items()
        }
        return {n: np.array(o, dtype=np.float32) for n, o in s["reset_obs"].items()}

    def step(self, actions):
        st = self.script["steps"][self.tick]
        self.tick += 1
        self.vehicles_including_just_terminated = {
            n: (_Veh(p) if p is not None else None) for n, p in st["pos"].items()
        }
        o = {n: np.array(v, dtype=np.float32) for n, v in st["obs"].items()}
        r = dict(st["rew"])
        d = dict(st["done"])
        i = {n: {} for n in o}
        return o, r, d, i


LCF = W.get_lcf_env(FakeBase)
CC = W.get_ccenv(FakeBase)


def run_single_step_case(pos, present, rew, lcf_mean, lcf_std, seed, radius=40, obs_dim=91, lcf_given=None):
    """One LCFEnv.step on a static scene; returns dense arrays indexed by slot."""
    N = len(pos)
    names = ["agent%d" % k for k in range(N)]
    env = LCF({"neighbours_distance": radius})
    env.set_lcf_dist(lcf_mean, lcf_std)
    rng = np.random.RandomState(seed)
    obs = {names[k]: rng.uniform(0, 1, obs_dim).astype(np.float32) for k in range(N) if present[k]}
    script = dict(
        reset_pos={names[k]: (pos[k] if present[k] else None) for k in range(N)},
        reset_obs=obs,
        steps=[dict(
            pos={names[k]: (pos[k] if present[k] else None) for k in range(N)},
            obs=obs,
            rew={names[k]: float(rew[k]) for k in range(N) if present[k]},
            done={names[k]: False for k in range(N) if present[k]},
        )],
    )
    env.load(script)
    ref_stubs.reseed_env_rng(seed)
    if lcf_given is not None:
        # pre-seed the env's lcf_map the way reset would have (env_wrappers.py:291-293)
        env._update_distance_map()
        for k in range(N):
            if present[k]:
                env.lcf_map[names[k]] = float(lcf_given[k])
    o, r, d, i = env.step({n: [0, 0] for n in obs})
    K = N - 1
    out = dict(
        nbr_idx=np.full((N, K), -1, np.int32), nbr_cnt=np.zeros(N, np.int32),
        nbr_dist=np.zeros((N, K), np.float64), nei_r=np.zeros(N, np.float64),
        glob_r=np.zeros(N, np.float64), lcf=np.zeros(N, np.float64), coord_r=np.zeros(N, np.float64),
        obs_last=np.zeros(N, np.float32), obs_len=np.zeros(N, np.int32), obs_dtype_is_f32=np.zeros(N, np.bool_),
    )
    for k in range(N):
        if not present[k]:
            continue
        inf = i[names[k]]
        ids = [int(n[5:]) for n in inf["neighbours"]]
        out["nbr_cnt"][k] = len(ids)
        out["nbr_idx"][k, :len(ids)] = ids
        out["nbr_dist"][k, :len(ids)] = inf["neighbours_distance"]
        out["nei_r"][k] = inf["nei_rewards"]
        out["glob_r"][k] = inf["global_rewards"]
        out["lcf"][k] = inf["lcf"]
        out["coord_r"][k] = inf["coordinated_rewards"]
        out["obs_last"][k] = o[names[k]][-1]
        out["obs_len"][k] = len(o[names[k]])
        out["obs_dtype_is_f32"][k] = o[names[k]].dtype == np.float32
        assert r[names[k]] == rew[k]  # return_native_reward
        assert inf["all_agents"] == [names[j] for j in range(N) if present[j]]
    return out


def gen_lcfenv_step():
    cases = []
    # case 0: the SURVEY known-answer scene
    cases.append(dict(pos=np.array([[0, 0], [3, 4], [30, 0], [100, 0]], float), present=np.ones(4, bool),
                      rew=np.array([1., 2., 3., 4.]), lcf_mean=0.0, lcf_std=0.1, seed=1))
    rng = np.random.RandomState(1234)
    # case 1: 40 agents, 200 m x 200 m crossing
    cases.append(dict(pos=f32r(rng.uniform(-100, 100, (40, 2))), present=np.ones(40, bool),
                      rew=f32r(rng.normal(0, 1, 40)), lcf_mean=0.2, lcf_std=0.1, seed=2))
    # case 2: 40 agents, dense (everyone within radius) -> lists of length 39
    cases.append(dict(pos=f32r(rng.uniform(-12, 12, (40, 2))), present=np.ones(40, bool),
                      rew=f32r(rng.normal(0, 1, 40)), lcf_mean=-0.3, lcf_std=0.2, seed=3))
    # case 3: ties and the strict `< 40` boundary (3-4-5 triangles scaled; exact in fp32 and fp64)
    pos = np.array([[0, 0], [24, 32], [32, 24], [40, 0], [0, 40], [-24, -32], [39.99, 0], [0, -3], [3, 0], [-3, 0]], float)
    cases.append(dict(pos=f32r(pos), present=np.ones(len(pos), bool), rew=f32r(np.arange(len(pos)) * 0.5 - 1),
                      lcf_mean=0.0, lcf_std=0.1, seed=4))
    # case 4: absent vehicles (None) interleaved
    present = rng.uniform(size=40) > 0.35
    cases.append(dict(pos=f32r(rng.uniform(-60, 60, (40, 2))), present=present,
                      rew=f32r(rng.normal(0, 1, 40)), lcf_mean=0.5, lcf_std=0.05, seed=5))
    # case 5: ParkingLot-like 10 agents, radius 10 (legacy CCEnv radius) -> sparse lists
    cases.append(dict(pos=f32r(rng.uniform(-20, 20, (10, 2))), present=np.ones(10, bool),
                      rew=f32r(rng.normal(0, 1, 10)), lcf_mean=0.0, lcf_std=0.1, seed=6, radius=10))
    # case 6: lone agent; case 7: two coincident agents (d == 0)
    cases.append(dict(pos=np.array([[5., 5.]]), present=np.ones(1, bool), rew=np.array([2.0]),
                      lcf_mean=0.0, lcf_std=0.1, seed=7))
    cases.append(dict(pos=np.array([[1., 1.], [1., 1.], [50., 1.]]), present=np.ones(3, bool),
                      rew=np.array([1.0, -1.0, 4.0]), lcf_mean=0.0, lcf_std=0.1, seed=8))
    # case 8: given (pre-existing) LCF values incl. the clip edges +-1
    given = f32r(np.clip(rng.normal(0, 0.8, 40), -1, 1))
    given[:2] = [-1.0, 1.0]
    cases.append(dict(pos=f32r(rng.uniform(-80, 80, (40, 2))), present=np.ones(40, bool),
                      rew=f32r(rng.normal(0, 1, 40)), lcf_mean=0.0, lcf_std=0.1, seed=9, lcf_given=given))
    save = {"n_cases": len(cases)}
    for c, case in enumerate(cases):
        out = run_single_step_case(**case)
        for k, v in case.items():
            if v is not None:
                save["c%d_in_%s" % (c, k)] = np.asarray(v)
        if "radius" not in case:
            save["c%d_in_radius" % c] = np.asarray(40)
        for k, v in out.items():
            save["c%d_out_%s" % (c, k)] = v
    # known answers quoted in SURVEY.md §8c
    assert save["c0_out_nbr_idx"][0, :2].tolist() == [1, 2] and save["c0_out_nbr_idx"][2, :2].tolist() == [1, 0]
    assert np.allclose(save["c0_out_nei_r"], [2.5, 2.0, 1.5, 0.0]) and np.allclose(save["c0_out_glob_r"], 2.5)
    np.savez_compressed(os.path.join(OUT, "lcfenv_step.npz"), **save)
    print("lcfenv_step.npz: %d cases" % len(cases))


# --------------------------------------------------------------------------------------
# LCF sampling statistics (env_wrappers.py:393-418): clip(N(mean,std),-1,1); obs tail = (lcf+1)/2
# --------------------------------------------------------------------------------------
def gen_lcf_sampling():
    env = LCF({})
    save = {}
    for j, (m, s) in enumerate([(0.0, 0.1), (0.9, 0.3), (-0.5, 1.5)]):
        env.set_lcf_dist(m, s)
        ref_stubs.reseed_env_rng(100 + j)
        lcfs, tails = [], []
        for _ in range(4000):
            lcf, o = env._add_lcf(np.zeros(3, np.float32))
            lcfs.append(lcf)
            tails.append(o[-1])
        lcfs = np.array(lcfs)
        save["d%d_mean_std" % j] = np.array([m, s])
        save["d%d_lcf" % j] = lcfs
        save["d%d_tail" % j] = np.array(tails, np.float32)
        save["d%d_stats" % j] = np.array([lcfs.mean(), lcfs.std(), (lcfs == 1).mean(), (lcfs == -1).mean()])
    np.savez_compressed(os.path.join(OUT, "lcf_sampling.npz"), **save)
    print("lcf_sampling.npz")


# --------------------------------------------------------------------------------------
# GAE x3
# --------------------------------------------------------------------------------------
def gen_gae():
    rng = np.random.RandomState(7)
    lens = [1, 2, 3, 8, 20, 200, 200, 57]
    K, TM = len(lens), max(lens)
    save = dict(lens=np.array(lens), done_last=np.zeros(K, np.bool_), gamma=0.99, lam=0.95)
    for name in ["r", "v", "nr", "nv", "gr", "gv", "adv", "tgt", "nadv", "ntgt", "gadv", "gtgt"]:
        save[name] = np.zeros((K, TM), np.float32)
    for k, T in enumerate(lens):
        done = bool(k % 2 == 0)
        save["done_last"][k] = done
        b = SampleBatch()
        for key, tag, scale in [("rewards", "r", 1.0), ("vf_preds", "v", 5.0), (A.NEI_REWARDS, "nr", 1.0),
                                (A.NEI_VALUES, "nv", 5.0), (A.GLOBAL_REWARDS, "gr", 0.3), (A.GLOBAL_VALUES, "gv", 20.0)]:
            b[key] = (rng.normal(0, 1, T) * scale).astype(np.float32)
            save[tag][k, :T] = b[key]
        last = 0.0 if done else b["vf_preds"][-1]
        b = ref_stubs.compute_advantages(b, last, 0.99, 0.95)
        b = A.compute_nei_advantage(b, 0.0 if done else b[A.NEI_VALUES][-1], 0.99, 0.95)
        b = A.compute_global_advantage(b, 0.0 if done else b[A.GLOBAL_VALUES][-1], gamma=1.0, lambda_=0.95)
        for key, tag in [("advantages", "adv"), ("value_targets", "tgt"), (A.NEI_ADVANTAGE, "nadv"),
                         (A.NEI_TARGET, "ntgt"), (A.GLOBAL_ADVANTAGES, "gadv"), (A.GLOBAL_TARGET, "gtgt")]:
            assert b[key].dtype == np.float32
            save[tag][k, :T] = b[key]
    # SURVEY known answers
    b = SampleBatch({A.NEI_VALUES: np.array([.1, .2, .3], np.float32), A.NEI_REWARDS: np.array([1, 0, 2], np.float32)})
    b = A.compute_nei_advantage(b, 0.0, .99, .95)
    save["ka_nei_adv"], save["ka_nei_tgt"] = b[A.NEI_ADVANTAGE], b[A.NEI_TARGET]
    b = SampleBatch({A.GLOBAL_VALUES: np.array([.5, .4, .3], np.float32), A.GLOBAL_REWARDS: np.array([1, 1, 1], np.float32)})
    b = A.compute_global_advantage(b, np.float32(.3), 1.0, .95)
    save["ka_glob_adv"], save["ka_glob_tgt"] = b[A.GLOBAL_ADVANTAGES], b[A.GLOBAL_TARGET]
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **save)
    print("gae.npz")


# --------------------------------------------------------------------------------------
# Models / policies
# --------------------------------------------------------------------------------------
def model_config(fuse_mode, hiddens, counterfactual=True, num_neighbours=4, copo=True):
    cmc = dict(fuse_mode=fuse_mode, counterfactual=counterfactual, num_neighbours=num_neighbours)
    if copo:
        cmc[A.USE_DISTRIBUTIONAL_LCF] = True
        cmc["initial_lcf_std"] = 0.1
    return dict(fcnet_hiddens=list(hiddens), fcnet_activation="tanh", post_fcnet_hiddens=[], no_final_linear=False,
                vf_share_layers=False, free_log_std=False, custom_model_config=cmc)


def make_model(cls, odim, fuse_mode, hiddens, seed, **kw):
    torch.manual_seed(seed)
    obs_space = Box(-1.0, 1.0, shape=(odim,))
    act_space = Box(-1.0, 1.0, shape=(2,))
    return cls(obs_space, act_space, 4, model_config(fuse_mode, hiddens, copo=(cls is A.CoPOModel), **kw), "m")


def state_arrays(model, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def base_policy_config(**over):
    cfg = dict(kl_coeff=0.2, clip_param=0.2, use_critic=True, use_gae=True, old_value_loss=True, vf_clip_param=100.0,
               vf_loss_coeff=1.0, gamma=0.99, counterfactual=True, num_neighbours=4, fuse_mode="none",
               mf_nei_distance=10, lcf_lr=1e-4)
    cfg["lambda"] = 0.95
    cfg[A.USE_DISTRIBUTIONAL_LCF] = True
    cfg.update(over)
    return cfg


def make_policy(pcls, model, cfg, target_model=None):
    class P(pcls):
        def __init__(self):
            pass

    p = P()
    p.config = cfg
    p.model = model
    p.device = "cpu"
    p.dist_class = TorchDiagGaussian
    p.entropy_coeff = 0.0
    p.kl_coeff = cfg["kl_coeff"]
    p._lazy_tensor_dict = lambda b, device=None: SampleBatch(
        {k: (torch.as_tensor(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()})
    if target_model is not None:
        p.target_model = target_model
        p._lcf_optimizer = torch.optim.Adam([model.lcf_parameters], lr=cfg["lcf_lr"])
    return p


def gen_param_counts():
    save = {}
    for tag, cls, fuse, odim in [("copo_none_92", A.CoPOModel, "none", 92), ("cc_mf_92", C.CCModel, "mf", 92),
                                 ("cc_concat_92", C.CCModel, "concat", 92), ("cc_mf_156", C.CCModel, "mf", 156),
                                 ("copo_none_260", A.CoPOModel, "none", 260)]:
        m = make_model(cls, odim, fuse, [256, 256], 0)
        save[tag + "_nparams"] = np.array(sum(p.numel() for p in m.parameters()))
        save[tag + "_ccdim"] = np.array(m.get_centralized_critic_obs_dim())
        save[tag + "_keys"] = np.array(sorted(m.state_dict().keys()))
        if cls is A.CoPOModel:
            save[tag + "_lcf_param"] = m.lcf_parameters.detach().numpy()
            save[tag + "_lcf_is_f64"] = np.array(m.lcf_parameters.dtype == torch.float64)
            save[tag + "_lcf_mean_std"] = np.array([m.lcf_mean.item(), m.lcf_std.item()])
        # normc init: every output row of a hidden weight has L2 norm 1.0, of the head 0.01
        w = m._hidden_layers[0]._model[0].weight.detach().numpy()
        save[tag + "_row_norm_hidden"] = np.linalg.norm(w, axis=1)[:4]
        save[tag + "_row_norm_head"] = np.linalg.norm(m._logits._model[0].weight.detach().numpy(), axis=1)
    assert save["copo_none_92_nparams"] == 360201 and save["cc_mf_92_nparams"] == 204549
    assert save["cc_mf_92_ccdim"] == 186 and save["cc_concat_92_ccdim"] == 468 and save["copo_none_92_ccdim"] == 92
    np.savez_compressed(os.path.join(OUT, "param_counts.npz"), **save)
    print("param_counts.npz", {k: int(v) for k, v in save.items() if k.endswith("nparams")})


# --------------------------------------------------------------------------------------
# Scripted multi-agent episode -> reference LCFEnv/CCEnv -> per-agent batches -> postprocess
# --------------------------------------------------------------------------------------
def build_episode(N, T, base_odim, seed, radius, use_lcf_env=True):
    """Scripted slot lifetimes; returns (dense inputs, per-agent SampleBatches)."""
    rng = np.random.RandomState(seed)
    env = (LCF if use_lcf_env else CC)({"neighbours_distance": radius})
    if use_lcf_env:
        env.set_lcf_dist(0.3, 0.2)
    # slot lifetime script: alive[t, n] = agent acts at step t ; aid[t, n]
    alive = np.zeros((T + 1, N), bool)   # alive before step t (t = T: after last step)
    aid = np.full((T + 1, N), -1, np.int64)
    done_at = np.zeros((T, N), bool)
    next_id = 0
    cur = np.full(N, -1)
    gap = np.zeros(N, int)
    for n in range(N):
        if rng.uniform() < 0.8:
            cur[n] = next_id
            next_id += 1
    for t in range(T + 1):
        for n in range(N):
            if cur[n] >= 0:
                alive[t, n] = True
                aid[t, n] = cur[n]
        if t == T:
            break
        for n in range(N):
            if cur[n] >= 0:
                if rng.uniform() < 0.08:     # terminates during step t
                    done_at[t, n] = True
                    cur[n] = -1
                    gap[n] = rng.randint(1, 4)
            else:
                gap[n] -= 1
                if gap[n] <= 0:              # respawns during step t -> present post-step, acts from t+1
                    cur[n] = next_id
                    next_id += 1
    pos = f32r(rng.uniform(-15, 15, (T + 1, N, 2)))   # post-step positions for step t are pos[t+1]
    raw_obs = rng.uniform(-1, 1, (T + 1, N, base_odim)).astype(np.float32)
    rew = f32r(rng.normal(0, 1, (T, N)))
    act = rng.normal(0, 0.7, (T, N, 2)).astype(np.float32)

    name = lambda a: "agent%d" % a  # noqa: E731
    script = dict(
        reset_pos={name(aid[0, n]): pos[0, n] for n in range(N) if alive[0, n]},
        reset_obs={name(aid[0, n]): raw_obs[0, n] for n in range(N) if alive[0, n]},
        steps=[],
    )
    for t in range(T):
        st = dict(pos={}, obs={}, rew={}, done={})
        for n in range(N):   # slot order == dict order == tie-break order
            if alive[t, n]:      # acted at t (incl. just terminated)
                a = name(aid[t, n])
                st["pos"][a] = pos[t + 1, n]
                st["obs"][a] = raw_obs[t + 1, n]
                st["rew"][a] = float(rew[t, n])
                st["done"][a] = bool(done_at[t, n])
            elif alive[t + 1, n]:  # newly spawned during step t: obs, reward 0, done False
                a = name(aid[t + 1, n])
                st["pos"][a] = pos[t + 1, n]
                st["obs"][a] = raw_obs[t + 1, n]
                st["rew"][a] = 0.0
                st["done"][a] = False
        script["steps"].append(st)
    env.load(script)
    ref_stubs.reseed_env_rng(seed)

    if use_lcf_env:
        obs0 = env._get_reset_return()
    else:
        obs0 = FakeBase._get_reset_return(env)
        env._update_distance_map()
    O = base_odim + (1 if use_lcf_env else 0)
    K = N - 1
    dense = dict(
        obs=np.zeros((T, N, O), np.float32), act=act.copy(), rew=np.zeros((T, N), np.float32),
        done=done_at.copy(), acted=alive[:T].copy(), present=np.zeros((T, N), bool), aid=aid[:T].copy(),
        pos_post=pos[1:].astype(np.float32),
        nbr_idx=np.full((T, N, K), -1, np.int32), nbr_cnt=np.zeros((T, N), np.int32),
        nbr_dist=np.zeros((T, N, K), np.float64),
        nei_r=np.zeros((T, N), np.float32), glob_r=np.zeros((T, N), np.float32), lcf=np.zeros((T, N), np.float32),
    )
    cur_obs = dict(obs0)
    rows = defaultdict(lambda: defaultdict(list))
    for t in range(T):
        acting = [n for n in range(N) if alive[t, n]]
        o, r, d, i = env.step({name(aid[t, n]): act[t, n] for n in acting})
        slot_of = {}
        for n in range(N):
            if alive[t, n]:
                slot_of[name(aid[t, n])] = n
            elif alive[t + 1, n]:
                slot_of[name(aid[t + 1, n])] = n
        for a, n in slot_of.items():
            dense["present"][t, n] = True
        for n in acting:
            a = name(aid[t, n])
            inf = i[a]
            dense["obs"][t, n] = cur_obs[a]
            dense["rew"][t, n] = r[a]
            ids = [slot_of[x] for x in inf["neighbours"]]
            dense["nbr_cnt"][t, n] = len(ids)
            dense["nbr_idx"][t, n, :len(ids)] = ids
            dense["nbr_dist"][t, n, :len(ids)] = inf["neighbours_distance"]
            if use_lcf_env:
                dense["nei_r"][t, n] = inf["nei_rewards"]
                dense["glob_r"][t, n] = inf["global_rewards"]
                dense["lcf"][t, n] = inf["lcf"]
            b = rows[a]
            b["obs"].append(cur_obs[a])
            b["actions"].append(act[t, n])
            b["rewards"].append(r[a])
            b["dones"].append(d[a])
            b["infos"].append(inf)
            b["t"].append(t)
            b["_slot"].append(n)
        cur_obs = dict(o)
    batches = {}
    for a, b in rows.items():
        sb = SampleBatch(
            obs=np.stack(b["obs"]).astype(np.float32), actions=np.stack(b["actions"]).astype(np.float32),
            rewards=np.array(b["rewards"], np.float32), dones=np.array(b["dones"]), infos=list(b["infos"]),
            t=np.array(b["t"]),
        )
        sb._slot = b["_slot"]
        batches[a] = sb
    return dense, batches, O


def gen_postprocess():
    for fuse, pcls, mcls, seed in [("none", A.CoPOPolicy, A.CoPOModel, 11), ("mf", A.CoPOPolicy, A.CoPOModel, 12),
                                   ("concat", A.CoPOPolicy, A.CoPOModel, 13), ("mf", C.CCPPOPolicy, C.CCModel, 14),
                                   ("concat", C.CCPPOPolicy, C.CCModel, 15)]:
        copo = pcls is A.CoPOPolicy
        N, T, base_odim = 7, 26, 7
        dense, batches, O = build_episode(N, T, base_odim, seed, radius=40 if copo else 10, use_lcf_env=copo)
        model = make_model(mcls, O, fuse, [16, 16], seed)
        cfg = base_policy_config(fuse_mode=fuse)
        pol = make_policy(pcls, model, cfg)
        pol.centralized_critic_obs_dim = model.get_centralized_critic_obs_dim()
        Cdim = pol.centralized_critic_obs_dim
        out = dict(cc_obs=np.zeros((T, N, Cdim), np.float32))
        keys = ["vf_preds", "advantages", "value_targets"]
        if copo:
            keys += [A.NEI_VALUES, A.NEI_REWARDS, A.NEI_ADVANTAGE, A.NEI_TARGET, A.GLOBAL_VALUES, A.GLOBAL_REWARDS,
                     A.GLOBAL_ADVANTAGES, A.GLOBAL_TARGET, "step_lcf"]
        for k in keys:
            out[k] = np.zeros((T, N), np.float32)
        for a, sb in batches.items():
            others = {b: (None, ob) for b, ob in batches.items() if b != a}
            res = pcls.postprocess_trajectory(pol, sb, others, episode=object())
            for j, (t, n) in enumerate(zip(sb["t"], sb._slot)):
                out["cc_obs"][t, n] = res[C.CENTRALIZED_CRITIC_OBS][j]
                for k in keys:
                    assert res[k].dtype == np.float32, (k, res[k].dtype)
                    out[k][t, n] = res[k][j]
        save = {"in_" + k: v for k, v in dense.items()}
        save.update({"out_" + k: v for k, v in out.items()})
        save.update(state_arrays(model, "w_"))
        save["cfg_fuse_mode"] = np.array(fuse)
        save["cfg_policy"] = np.array("copo" if copo else "ccppo")
        save["cfg_radius"] = np.array(40 if copo else 10)
        save["cfg_misc"] = np.array([cfg["gamma"], cfg["lambda"], cfg["mf_nei_distance"], cfg["num_neighbours"]])
        fn = "postprocess_%s_%s.npz" % ("copo" if copo else "ccppo", fuse)
        np.savez_compressed(os.path.join(OUT, fn), **save)
        n_absent = int(((dense["nbr_idx"] >= 0) & ~np.take_along_axis(
            np.broadcast_to(dense["acted"][:, None, :], dense["nbr_idx"].shape[:2] + (N,)),
            np.maximum(dense["nbr_idx"], 0), axis=2)).sum())
        print(fn, "rows", int(dense["acted"].sum()), "agents", len(batches), "absent-neighbour refs", n_absent)


# --------------------------------------------------------------------------------------
# Losses / meta update / training_step
# --------------------------------------------------------------------------------------
def random_train_batch(rng, B, O, Cdim, copo):
    b = SampleBatch()
    b["obs"] = rng.uniform(-1, 1, (B, O)).astype(np.float32)
    b[C.CENTRALIZED_CRITIC_OBS] = np.concatenate(
        [b["obs"], rng.uniform(-1, 1, (B, Cdim - O)).astype(np.float32)], axis=1)
    b["actions"] = rng.normal(0, 0.8, (B, 2)).astype(np.float32)
    b["action_dist_inputs"] = np.concatenate(
        [rng.normal(0, 0.3, (B, 2)), rng.normal(-0.2, 0.2, (B, 2))], axis=1).astype(np.float32)
    d = TorchDiagGaussian(torch.as_tensor(b["action_dist_inputs"]))
    b["action_logp"] = d.logp(torch.as_tensor(b["actions"])).numpy().astype(np.float32)
    b["advantages"] = rng.normal(0, 2, B).astype(np.float32)
    b["vf_preds"] = rng.normal(0, 3, B).astype(np.float32)
    b["value_targets"] = (b["vf_preds"] + rng.normal(0, 150, B) * (rng.uniform(size=B) < 0.3)
                          + rng.normal(0, 2, B)).astype(np.float32)
    if copo:
        b[A.NEI_ADVANTAGE] = rng.normal(0, 2, B).astype(np.float32)
        b[A.GLOBAL_ADVANTAGES] = rng.normal(0, 1, B).astype(np.float32)
        b[A.NEI_VALUES] = rng.normal(0, 3, B).astype(np.float32)
        b[A.NEI_TARGET] = (b[A.NEI_VALUES] + rng.normal(0, 2, B)).astype(np.float32)
        b[A.GLOBAL_VALUES] = rng.normal(0, 30, B).astype(np.float32)
        b[A.GLOBAL_TARGET] = (b[A.GLOBAL_VALUES] + rng.normal(0, 120, B)).astype(np.float32)
        b["normalized_advantages"] = rng.normal(0, 1, B).astype(np.float32)
        b["step_lcf"] = np.clip(rng.normal(0.2, 0.3, B), -1, 1).astype(np.float32)
        b["rewards"] = rng.normal(0, 1, B).astype(np.float32)
    return b


def tensor_batch(b):
    return SampleBatch({k: torch.as_tensor(v) for k, v in b.items()})


def gen_losses(cases=None, seed=21):
    rng = np.random.RandomState(seed)
    for tag, pcls, mcls, fuse, over, O, B, hiddens in cases or [
        ("ippo", I.IPPOPolicy, C.CCModel, "none", {}, 12, 96, [32, 32]),
        ("ccppo_mf", C.CCPPOPolicy, C.CCModel, "mf", {}, 12, 96, [32, 32]),
        ("ccppo_concat", C.CCPPOPolicy, C.CCModel, "concat", {}, 12, 96, [32, 32]),
        ("copo", A.CoPOPolicy, A.CoPOModel, "none", {}, 12, 96, [32, 32]),
        ("copo_newvf", A.CoPOPolicy, A.CoPOModel, "none", dict(old_value_loss=False, vf_clip_param=10.0), 12, 96, [32, 32]),
        ("copo_nokl", A.CoPOPolicy, A.CoPOModel, "none", dict(kl_coeff=0.0), 12, 96, [32, 32]),
    ]:
        model = make_model(mcls, O, fuse, hiddens, 30)
        # move weights away from the near-zero head init so that KL/ratio terms are exercised
        with torch.no_grad():
            for p_ in model.parameters():
                if p_.dtype == torch.float32:
                    p_.add_(torch.randn_like(p_) * 0.05)
        Cdim = model.get_centralized_critic_obs_dim()
        cfg = base_policy_config(fuse_mode=fuse, **over)
        pol = make_policy(pcls, model, cfg)
        if pcls is I.IPPOPolicy:
            # IPPO uses RLlib's stock FC net (not in the reference tree); its separate value branch is
            # architecturally the CCModel value branch with fuse_mode "none". value_function() reads the
            # value of the obs that went through forward().
            model.value_function = lambda: model.central_value_function(model._last_obs)
        copo = pcls is A.CoPOPolicy
        b = random_train_batch(rng, B, O, Cdim, copo)
        tb = tensor_batch(b)
        if pcls is I.IPPOPolicy:
            model._last_obs = tb["obs"]
        model.zero_grad()
        loss = pcls.loss(pol, model, TorchDiagGaussian, tb)
        loss.backward()
        save = {"in_" + k: v for k, v in b.items()}
        save.update(state_arrays(model, "w_"))
        save["out_total_loss"] = loss.detach().numpy()
        for k, v in model.tower_stats.items():
            save["out_stat_" + k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        for k, p_ in model.named_parameters():
            save["out_grad_" + k] = (p_.grad.numpy().copy() if p_.grad is not None else np.zeros(0))
        save["cfg_fuse_mode"] = np.array(fuse)
        save["cfg_vals"] = np.array([cfg["kl_coeff"], cfg["clip_param"], cfg["vf_clip_param"], cfg["vf_loss_coeff"],
                                     float(cfg["old_value_loss"]), pol.kl_coeff, pol.entropy_coeff])
        np.savez_compressed(os.path.join(OUT, "loss_%s.npz" % tag), **save)
        print("loss_%s.npz" % tag, float(loss))


def gen_losses_config_shapes():
    """The same reference functions at the observation widths of the BASELINE configurations (the rows above use 12-wide
    toy observations and 32-wide layers, which the fused learner serves with its tile-GEMM kernels): 64-wide hidden layers take
    the production row-pass kernels.  configs[1]: CoPO, O = 92, one 512-row minibatch; configs[3]: CCPPO mean-field on the
    Tollgate, O = 156, centralised critic 2 * 156 + 2 = 314 wide; configs[4]: CoPO on the ParkingLot with 240 beams, O = 260."""
    gen_losses([
        ("copo_o92_b512", A.CoPOPolicy, A.CoPOModel, "none", {}, 92, 512, [64, 64]),
        ("ccppo_mf_o156", C.CCPPOPolicy, C.CCModel, "mf", {}, 156, 128, [64, 64]),
        ("copo_o260", A.CoPOPolicy, A.CoPOModel, "none", {}, 260, 128, [64, 64]),
    ], seed=22)


def gen_meta_update():
    rng = np.random.RandomState(31)
    O, B = 12, 128
    model = make_model(A.CoPOModel, O, "none", [32, 32], 40)
    target = make_model(A.CoPOModel, O, "none", [32, 32], 41)
    with torch.no_grad():
        for p_ in list(model.parameters()) + list(target.parameters()):
            if p_.dtype == torch.float32:
                p_.add_(torch.randn_like(p_) * 0.05)
    cfg = base_policy_config()
    pol = make_policy(A.CoPOPolicy, model, cfg, target_model=target)
    pol._raw_lcf_adv_mean = np.float32(0.37)
    pol._raw_lcf_adv_std = np.float32(2.4)
    save = {}
    save.update(state_arrays(model, "w_"))
    save.update(state_arrays(target, "wt_"))
    save["in_raw_mean_std"] = np.array([pol._raw_lcf_adv_mean, pol._raw_lcf_adv_std], np.float32)
    n_steps = 3
    save["n_steps"] = np.array(n_steps)
    for s in range(n_steps):
        b = random_train_batch(rng, B, O, O, True)
        for k, v in b.items():
            save["s%d_in_%s" % (s, k)] = v
        # the rsample inside compute_coordinated consumes torch's global RNG: pin it and record the eps drawn
        torch.manual_seed(500 + s)
        eps = torch.randn(B, dtype=torch.float64)  # Normal.rsample -> _standard_normal(shape, dtype, device)
        save["s%d_in_eps" % s] = eps.numpy()
        torch.manual_seed(500 + s)
        stats = A.CoPOPolicy.meta_update(pol, b)
        for k, v in stats.items():
            save["s%d_out_%s" % (s, k)] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        save["s%d_out_lcf_parameters" % s] = model.lcf_parameters.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "meta_update.npz"), **save)
    print("meta_update.npz lcf_parameters ->", model.lcf_parameters.detach().numpy())


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def gen_training_step():
    rng = np.random.RandomState(51)
    O, B = 12, 1200
    model = make_model(A.CoPOModel, O, "none", [32, 32], 60)
    target = make_model(A.CoPOModel, O, "none", [32, 32], 61)
    cfg = AttrDict(base_policy_config())
    cfg.update(count_steps_by="env_steps", train_batch_size=B, simple_optimizer=True, lcf_sgd_minibatch_size=None,
               sgd_minibatch_size=512, lcf_num_iters=5, vf_loss_coeff=1.0)
    pol = make_policy(A.CoPOPolicy, model, cfg, target_model=target)
    pol.num_grad_updates = 0
    kl_seen = []
    pol.update_kl = lambda kl: kl_seen.append(kl)
    b = random_train_batch(rng, B, O, O, True)
    batch0 = {k: v.copy() for k, v in b.items()}
    ma = SimpleNamespace(policy_batches={"default": b}, as_multi_agent=lambda: ma,
                         agent_steps=lambda: B, env_steps=lambda: B // 25)
    A.synchronous_parallel_sample = lambda **kw: ma
    A.train_one_step = lambda algo, tb: {"default": {"custom_metrics": {}, "learner_stats": dict(
        kl=0.0123, vf_loss=1.0, policy_loss=-0.01)}}
    env_calls = []
    fake_env = SimpleNamespace(set_lcf_dist=lambda mean, std: env_calls.append((mean, std)))
    worker = SimpleNamespace(
        foreach_policy=lambda f: [f(pol, "default")], foreach_env=lambda f: [f(fake_env)],
        policy_map={"default": pol}, set_global_vars=lambda gv: None)
    workers = SimpleNamespace(num_remote_workers=lambda: 0, local_worker=lambda: worker,
                              foreach_worker_with_id=lambda f: [f(0, worker)])
    from collections import defaultdict as dd
    algo = SimpleNamespace(config=cfg, workers=workers, _counters=dd(int), get_policy=lambda pid="default": pol,
                           _timers=dd(lambda: None))
    torch.manual_seed(777)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        res = A.CoPOTrainer.training_step(algo)
    save = {"in_" + k: v for k, v in batch0.items()}
    save.update(state_arrays(make_model(A.CoPOModel, O, "none", [32, 32], 60), "w_"))
    save.update(state_arrays(make_model(A.CoPOModel, O, "none", [32, 32], 61), "wt_"))
    save["in_torch_seed"] = np.array(777)
    save["out_normalized_advantages"] = b["normalized_advantages"]
    save["out_raw_normalized_advantages"] = b["raw_normalized_advantages"]
    save["out_global_advantages"] = b[A.GLOBAL_ADVANTAGES]
    save["out_raw_mean_std"] = np.array([pol._raw_lcf_adv_mean, pol._raw_lcf_adv_std], np.float64)
    save["out_lcf_parameters"] = model.lcf_parameters.detach().numpy().copy()
    save["out_env_lcf_dist"] = np.array(env_calls[-1], np.float64)
    save["out_kl_seen"] = np.array(kl_seen)
    save["out_counters"] = np.array([algo._counters["num_agent_steps_sampled"], algo._counters["num_env_steps_sampled"]])
    tgt_same = all(torch.equal(a_, b_) for a_, b_ in zip(model.state_dict().values(), target.state_dict().values()))
    save["out_target_equals_model"] = np.array(tgt_same)
    mu = res["default"]["custom_metrics"]["meta_update"]
    save["out_meta_keys"] = np.array(sorted(mu.keys()))
    for k, v in mu.items():
        save["out_meta_" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "training_step.npz"), **save)
    print("training_step.npz lcf_parameters", save["out_lcf_parameters"], "env dist", env_calls[-1],
          "n meta keys", len(mu))


def gen_callbacks():
    """MultiAgentDrivingCallbacks.on_episode_end on a synthetic info stream (utils/callbacks.py:48-110)."""
    import copo.torch_copo.utils.callbacks as CB
    rng = np.random.RandomState(71)
    n_agents, save = 9, {}
    infos, user = {}, {k: defaultdict(list) for k in ["velocity", "steering", "step_reward", "acceleration", "cost",
                                                      "episode_length", "episode_reward", "num_neighbours"]}
    term = rng.randint(0, 4, n_agents)  # 0 arrive 1 crash 2 out 3 max_step
    lens = rng.randint(3, 30, n_agents)
    save["in_term"], save["in_len"] = term, lens
    per_step = {k: np.zeros((n_agents, 30)) for k in ["velocity", "steering", "step_reward", "acceleration", "cost"]}
    nn_ = np.zeros((n_agents, 30), int)
    for a in range(n_agents):
        er = 0.0
        for s in range(lens[a]):
            for k in per_step:
                per_step[k][a, s] = rng.uniform(0, 1) if k != "cost" else float(rng.uniform() < 0.1)
                user[k][a].append(per_step[k][a, s])
            er += per_step["step_reward"][a, s]
            user["episode_length"][a].append(s + 1)
            user["episode_reward"][a].append(er)
            nn_[a, s] = rng.randint(0, 6)
            user["num_neighbours"][a].append(nn_[a, s])
        infos[a] = dict(arrive_dest=term[a] == 0, crash=term[a] == 1, out_of_road=term[a] == 2,
                        route_completion=float(rng.uniform()), track_length=100.0, current_distance=50.0)
    save["in_route_completion"] = np.array([infos[a]["route_completion"] for a in range(n_agents)])
    for k in per_step:
        save["in_" + k] = per_step[k]
    save["in_num_neighbours"] = nn_
    ep = SimpleNamespace(agent_rewards={(a, "default"): 0 for a in range(n_agents)}, last_info_for=lambda k: infos[k],
                         custom_metrics={}, user_data=user)
    cb = CB.MultiAgentDrivingCallbacks()
    CB.MultiAgentDrivingCallbacks.on_episode_end(cb, None, None, {}, ep)
    for k, v in ep.custom_metrics.items():
        save["out_" + k] = np.asarray(v, dtype=np.float64)
    result = dict(custom_metrics={k + "_mean": v for k, v in ep.custom_metrics.items()}, episode_len_mean=17.0,
                  episode_reward_mean=123.0, policy_reward_mean={"default": 4.5})
    CB.MultiAgentDrivingCallbacks.on_train_result(cb, algorithm=None, result=result)
    for k in ["success", "crash", "out", "max_step", "length", "rc", "cost", "raw_episode_reward_mean",
              "episode_reward_mean"]:
        save["res_" + k] = np.asarray(result[k], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "callbacks.npz"), **save)
    print("callbacks.npz", {k: float(v) for k, v in ep.custom_metrics.items() if "rate" in k})


# --------------------------------------------------------------------------------------
# f-4: traffic-light message + communication channel of CCEnv / LCFEnv (env_wrappers.py:89-118, 258-303, 315-337, 360-371)
# --------------------------------------------------------------------------------------
class _VehH:
    """Vehicle with a heading.  `projection` restates MetaDrive's BaseVehicle.projection (3P, absent): the vector in
    the vehicle frame (longitudinal, lateral-left)."""

    def __init__(self, p, th):
        self.position = np.asarray(p, dtype=np.float64)
        self.c, self.s = float(np.float32(np.cos(th))), float(np.float32(np.sin(th)))   # fp32 (cos, sin) like the build's state

    def projection(self, v):
        return (v[0] * self.c + v[1] * self.s, v[1] * self.c - v[0] * self.s)


class FakeBaseExt(FakeBase):
    BBOX = (-80.0, 90.0, -70.0, 75.0)

    def __init__(self, config=None):
        super().__init__(config)
        box = self.BBOX
        self.engine = SimpleNamespace(current_map=SimpleNamespace(road_network=SimpleNamespace(get_bounding_box=lambda: box)))

    def _veh(self, d):
        return {n: (_VehH(p[:2], p[2]) if p is not None else None) for n, p in d.items()}

    def _get_reset_return(self):
        s = self.script
        self.vehicles_including_just_terminated = self._veh(s["reset_pos"])
        return {n: np.array(o, dtype=np.float32) for n, o in s["reset_obs"].items()}

    def step(self, actions):
        st = self.script["steps"][self.tick]
        self.tick += 1
        self.vehicles_including_just_terminated = self._veh(st["pos"])
        o = {n: np.array(v, dtype=np.float32) for n, v in st["obs"].items()}
        return o, dict(st["rew"]), dict(st["done"]), {n: {} for n in o}


LCF_EXT = W.get_lcf_env(FakeBaseExt)


def gen_obs_extensions():
    """Scripted scenes through the reference's LCFEnv with the traffic light / communication switched on: the extended
    observations of reset() and of T steps, with agents that are present but were given no action (fresh respawns)."""
    rng = np.random.RandomState(77)
    cfgs = [dict(tl=1, cs=0, nb=4, pos=0), dict(tl=0, cs=4, nb=4, pos=0), dict(tl=1, cs=3, nb=2, pos=1),
            dict(tl=1, cs=4, nb=6, pos=1), dict(tl=0, cs=2, nb=3, pos=1)]
    save = {"n_cases": len(cfgs), "bbox": np.asarray(FakeBaseExt.BBOX, np.float32), "base_dim": 91}
    N, T, OB = 12, 9, 91
    names = ["agent%d" % k for k in range(N)]
    for c, cf in enumerate(cfgs):
        interval = 4
        env = LCF_EXT(dict(neighbours_distance=40, add_traffic_light=bool(cf["tl"]), traffic_light_interval=interval,
                           communication=dict(comm_method="broadcast" if cf["cs"] else "none", comm_size=cf["cs"] or 4,
                                              comm_neighbours=cf["nb"], add_pos_in_comm=bool(cf["pos"]))))
        CS = cf["cs"]
        CD = CS + (3 if cf["pos"] else 0)
        pos = f32r(rng.uniform(-45, 45, (T + 1, N, 2)))
        pos[:, 1] = pos[:, 0] + f32r([[3.0, 0.0]])       # a pair that stays close
        pos[2, 5] = pos[2, 4]                             # coincident pair at one step (d == 0)
        th = f32r(rng.uniform(-np.pi, np.pi, (T + 1, N)))
        present = rng.uniform(size=(T + 1, N)) > 0.2
        present[:, :3] = True
        acted = present & (rng.uniform(size=(T + 1, N)) > 0.25)     # present but not acted = fresh respawn
        acted[0] = False
        act = f32r(rng.uniform(-1, 1, (T + 1, N, 2 + CS)))
        base = rng.uniform(0, 1, (T + 1, N, OB)).astype(np.float32)
        O = OB + (3 if cf["tl"] else 0) + 1 + (cf["nb"] * CD if CS else 0)
        out = np.zeros((T + 1, N, O), np.float32)

        def pd(t):
            return {names[k]: ((pos[t, k, 0], pos[t, k, 1], th[t, k]) if present[t, k] else None) for k in range(N)}

        script = dict(reset_pos=pd(0), reset_obs={names[k]: base[0, k] for k in range(N) if present[0, k]}, steps=[
            dict(pos=pd(t), obs={names[k]: base[t, k] for k in range(N) if present[t, k]},
                 rew={names[k]: 0.0 for k in range(N) if present[t, k]},
                 done={names[k]: False for k in range(N) if present[t, k]}) for t in range(1, T + 1)])
        env.load(script)
        ref_stubs.reseed_env_rng(c)
        o = env._get_reset_return()
        for k in range(N):
            if present[0, k]:
                assert o[names[k]].dtype == np.float32 and len(o[names[k]]) == O, (len(o[names[k]]), O)
                out[0, k] = o[names[k]]
        for t in range(1, T + 1):
            actions = {names[k]: act[t, k].astype(np.float32) for k in range(N) if acted[t, k]}
            o, r, d, i = env.step(actions)
            for k in range(N):
                if present[t, k]:
                    assert o[names[k]].dtype == np.float32 and len(o[names[k]]) == O
                    out[t, k] = o[names[k]]
        for k, v in dict(tl=cf["tl"], cs=CS, nb=cf["nb"], pos=cf["pos"], interval=interval, positions=pos.astype(np.float32),
                         heading_cs=np.stack([np.cos(th).astype(np.float32), np.sin(th).astype(np.float32)], -1),
                         present=present, acted=acted, act=act.astype(np.float32), ext=out[..., OB:]).items():
            save["c%d_%s" % (c, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "obs_extensions.npz"), **save)
    print("obs_extensions.npz: %d cases, %d bytes" % (len(cfgs), os.path.getsize(os.path.join(OUT, "obs_extensions.npz"))))


# --------------------------------------------------------------------------------------
# f-1: RecorderEnv episode statistics (copo/eval/recoder.py:73-349) on a scripted info stream
# --------------------------------------------------------------------------------------
RECORDER_INFO_KEYS = ("velocity", "steering", "step_reward", "acceleration", "cost", "episode_length", "episode_reward",
                      "step_energy", "episode_energy")


def make_recorder_stream(seed, T=60, N=9):
    """Arrays of a scripted episode: who is present at every step, who is on its first observation (no transition info
    yet), who terminates and how, positions, rewards and the per-step info columns."""
    rng = np.random.RandomState(seed)
    # lives: a slot holds one agent at a time; an agent appears (first observation, empty info), acts for a while and
    # terminates (done) -- every agent terminates by the last step, like MetaDrive's horizon does
    present, first, done = (np.zeros((T, N), bool) for _ in range(3))
    aid = np.full((T, N), -1, np.int64)
    next_id = 0
    for n in range(N):
        t = int(rng.randint(0, 4))
        while t < T - 1:
            end = min(T - 1, t + int(rng.randint(2, 25)))
            present[t:end + 1, n] = True
            first[t, n] = t > 0 or rng.uniform() < 0.5
            done[end, n] = True
            aid[t:end + 1, n] = next_id
            next_id += 1
            t = end + 1 + int(rng.randint(0, 3))
    kind = rng.randint(0, 4, (T, N))                          # 0 arrive, 1 crash, 2 out, 3 max step
    pos = rng.uniform(-30, 30, (T, N, 2))
    rew = rng.normal(0.5, 1.0, (T, N))
    info = {k: rng.uniform(0, 5, (T, N)) for k in RECORDER_INFO_KEYS}
    info["cost"] = (rng.uniform(size=(T, N)) < 0.1).astype(np.float64)
    info["episode_length"] = np.floor(rng.uniform(1, 200, (T, N)))
    raw = rng.uniform(-1, 1, (T, N, 2))
    return dict(present=present, first=first, done=done, aid=aid, kind=kind, pos=pos, rew=rew, raw_action=raw, **info)


class _ScriptedStreamEnv:
    """Dict-API env that plays a `make_recorder_stream` script (what RecorderEnv wraps)."""

    def __init__(self, script):
        self.s, self.t, self.vehicles = script, 0, {}
        self.N = script["present"].shape[1]

    def reset(self):
        self.t = 0
        return {}

    def close(self):
        pass

    def step(self, actions):
        s, t = self.s, self.t
        o, r, d, i = {}, {}, {}, {}
        self.vehicles = {}
        for n in range(self.N):
            if not s["present"][t, n]:
                continue
            k = "agent%d" % s["aid"][t, n]
            self.vehicles[k] = SimpleNamespace(position=s["pos"][t, n])
            o[k], r[k], d[k] = np.zeros(3, np.float32), float(s["rew"][t, n]), bool(s["done"][t, n])
            if s["first"][t, n]:
                i[k] = {}
                continue
            i[k] = {key: float(s[key][t, n]) for key in RECORDER_INFO_KEYS}
            i[k]["raw_action"] = s["raw_action"][t, n]
            if d[k]:
                kd = int(s["kind"][t, n])
                i[k].update(arrive_dest=kd == 0, crash=kd == 1, out_of_road=kd == 2)
        d["__all__"] = t == s["present"].shape[0] - 1
        self.t += 1
        return o, r, d, i


def gen_recorder():
    import copo.eval.recoder as R
    save = {"n_cases": 3}
    for c in range(3):
        script = make_recorder_stream(100 + c, T=40 + 15 * c, N=6 + 3 * c)
        env = R.RecorderEnv(_ScriptedStreamEnv(script), eval_config=dict(neighbours_distance=[20, 35, 12][c]))
        env.reset()
        step_results = []
        for t in range(script["present"].shape[0]):
            _, _, d, _ = env.step({})
            if t in (5, 17):
                step_results.append(env.get_step_result())
        assert d["__all__"]
        res = env.get_episode_result()
        for k, v in script.items():
            save["c%d_in_%s" % (c, k)] = np.asarray(v)
        save["c%d_in_distance" % c] = np.asarray([20, 35, 12][c])
        save["c%d_keys" % c] = np.asarray(sorted(res))
        save["c%d_vals" % c] = np.asarray([float(res[k]) for k in sorted(res)], np.float64)
        for q, sr in enumerate(step_results):
            save["c%d_step%d_keys" % (c, q)] = np.asarray(sorted(sr))
            save["c%d_step%d_vals" % (c, q)] = np.asarray([float(sr[k]) for k in sorted(sr)], np.float64)
    np.savez_compressed(os.path.join(OUT, "recorder.npz"), **save)
    print("recorder.npz:", os.path.getsize(os.path.join(OUT, "recorder.npz")), "bytes;", len(res), "episode statistics")


if __name__ == "__main__":
    if not os.path.isdir(ref_stubs.REFERENCE_ROOT):
        sys.exit("reference tree not present; fixtures are committed under tests/golden/")
    gen_lcfenv_step()
    gen_lcf_sampling()
    gen_gae()
    gen_param_counts()
    gen_postprocess()
    gen_losses()
    gen_losses_config_shapes()
    gen_meta_update()
    gen_training_step()
    gen_callbacks()
    gen_obs_extensions()
    gen_recorder()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes:", tot)

"""GPU: the dense postprocess against the reference's per-trajectory `postprocess_trajectory` golden vectors, the
three trainers end to end, and the dict-style env API."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from copo_amd.engine import Box, Postprocessing, SampleBatch  # noqa: E402


def _policy(kind, fuse, odim):
    from copo_amd.torch_copo import algo_ccppo as C, algo_copo as A
    from copo_amd.torch_copo.utils.env_wrappers import (MultiAgentIntersectionEnv, get_ccenv, get_lcf_env,
                                                        get_rllib_compatible_env)
    pcls, ccls = (A.CoPOPolicy, A.CoPOConfig) if kind == "copo" else (C.CCPPOPolicy, C.CCPPOConfig)
    cfg = ccls()
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv) if kind == "copo"
                                   else get_ccenv(MultiAgentIntersectionEnv))
    cfg.update_from_dict(dict(env=env, device="cuda", use_hip_graphs=False, use_fused_learner=False,
                              model={"fcnet_hiddens": [16, 16]}))
    cfg.fuse_mode = fuse
    cfg.validate()
    return pcls(Box(-1, 1, (odim,)), Box(-1, 1, (2,)), cfg)


@pytest.mark.parametrize("kind,fuse", [("copo", "none"), ("copo", "mf"), ("copo", "concat"), ("ccppo", "mf"),
                                       ("ccppo", "concat")])
def test_dense_postprocess_vs_reference(golden_dir, kind, fuse):
    """CCPPOPolicy / CoPOPolicy.postprocess_trajectory (algo_ccppo.py:322-374, algo_copo.py:473-502): centralised
    critic obs, value heads, GAE per agent trajectory incl. slot reuse and truncated-fragment bootstrap."""
    g = np.load(os.path.join(golden_dir, "postprocess_%s_%s.npz" % (kind, fuse)))
    obs = g["in_obs"]
    T, N, O = obs.shape
    pol = _policy(kind, fuse, O)
    pol.model.load_state_dict({k[2:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w_")})
    acted, done = g["in_acted"], g["in_done"]
    flags = (acted * 1 + (done & acted) * 2).astype(np.uint8)
    K = g["in_nbr_idx"].shape[-1]
    mf_cnt = ((g["in_nbr_dist"] <= 10.0) & (np.arange(K)[None, None] < g["in_nbr_cnt"][..., None])).sum(-1)
    cu = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda().unsqueeze(1).contiguous()  # noqa: E731
    rew3 = np.stack([g["in_rew"], g["in_nei_r"], g["in_glob_r"]]).astype(np.float32)
    b = SampleBatch({SampleBatch.OBS: cu(obs), SampleBatch.ACTIONS: cu(g["in_act"]), SampleBatch.FLAGS: cu(flags),
                     "nbr_idx": cu(g["in_nbr_idx"]), "nbr_cnt": cu(g["in_nbr_cnt"]), "mf_cnt": cu(mf_cnt.astype(np.int32)),
                     "rew3": torch.as_tensor(rew3).cuda().unsqueeze(2).contiguous(), "step_lcf": cu(g["in_lcf"])})
    out = pol.postprocess_trajectory(b)
    m = acted
    chk = [(SampleBatch.VF_PREDS, "vf_preds"), (Postprocessing.ADVANTAGES, "advantages"),
           (Postprocessing.VALUE_TARGETS, "value_targets")]
    if kind == "copo":
        chk += [(k, k) for k in ("nei_values", "nei_advantage", "nei_target", "global_values", "global_advantages",
                                 "global_target")]
    np.testing.assert_allclose(out["centralized_critic_obs"][:, 0].cpu().numpy()[m], g["out_cc_obs"][m], rtol=1e-6, atol=1e-7)
    for mine, ref in chk:
        np.testing.assert_allclose(out[mine][:, 0].cpu().numpy()[m], g["out_" + ref][m], rtol=2e-5, atol=2e-5, err_msg=ref)


@pytest.mark.parametrize("algo,extra", [("ippo", {}), ("ccppo", dict(fuse_mode="mf")), ("ccppo", dict(fuse_mode="concat")),
                                        ("copo", {}), ("copo", dict(use_fused_learner=False, use_hip_graphs=False))])
def test_trainers_run_and_report(algo, extra):
    from copo_amd.torch_copo.algo_ccppo import CCPPOTrainer, get_ccppo_env
    from copo_amd.torch_copo.algo_copo import CoPOTrainer
    from copo_amd.torch_copo.algo_ippo import IPPOTrainer
    from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
    from copo_amd.torch_copo.utils.env_wrappers import (MultiAgentIntersectionEnv, MultiAgentRoundaboutEnv, get_lcf_env,
                                                        get_rllib_compatible_env)
    if algo == "ippo":
        cls, env = IPPOTrainer, get_rllib_compatible_env(MultiAgentIntersectionEnv)
    elif algo == "ccppo":
        cls, env = CCPPOTrainer, get_ccppo_env(MultiAgentRoundaboutEnv)
    else:
        cls, env = CoPOTrainer, get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    cfg = dict(env=env, env_config=dict(num_agents=12, horizon=60), num_envs=8, train_batch_size=8 * 10,
               sgd_minibatch_size=128, num_sgd_iter=2, seed=0, callbacks=MultiAgentDrivingCallbacks,
               model={"fcnet_hiddens": [64, 64]}, **extra)
    if algo == "copo":
        cfg["lcf_num_iters"] = 2
    a = cls(config=cfg)
    for _ in range(5):
        res = a.train()
    st = res["info"]["learner"]["default"]["learner_stats"]
    assert all(np.isfinite(v) for v in st.values()), st
    assert res["timesteps_total"] == 5 * 80 and res["agent_timesteps_total"] > 0
    for k in ("success", "crash", "out", "max_step", "length", "cost", "rc", "episode_reward_mean"):
        assert k in res
    if algo == "copo":
        mu = res["info"]["learner"]["default"]["custom_metrics"]["meta_update"]
        assert {"lcf", "lcf_std", "grad_value", "raw_lcf_adv_mean_value", "lcf_final_loss"} <= set(mu)
        assert abs(a.env.current_lcf_mean - mu["lcf"]) < 1e-9        # pushed to the envs (algo_copo.py:608-611)
        tgt, cur = a.policy.target_model.state_dict(), a.policy.model.state_dict()
        assert all(torch.equal(tgt[k], cur[k]) for k in cur)          # update_old_policy
    path = a.save_checkpoint("/tmp/copo_ckpt_test")
    w0 = {k: v.clone() for k, v in a.policy.model.state_dict().items()}
    fz = a.policy.fused
    adam0 = None if fz is None else (fz.adam_m.clone(), fz.adam_v.clone(), fz.step_count.clone())
    a.train()
    a.load_checkpoint(path)
    assert all(torch.equal(v, a.policy.model.state_dict()[k]) for k, v in w0.items())
    if fz is not None:
        # the fused learner's own Adam moments / step travel with the checkpoint (the torch optimizer is never stepped) ...
        assert int(adam0[2]) > 0 and torch.equal(fz.adam_m, adam0[0]) and torch.equal(fz.adam_v, adam0[1]) and torch.equal(fz.step_count, adam0[2])
        # ... and the forward kernels see the restored weights: their transposed mirror was invalidated by the load
        obs = torch.rand(64, a.env.sim.O, device="cuda")
        eps = torch.zeros(64, 2, device="cuda")
        act, logp, dist = torch.empty(64, 2, device="cuda"), torch.empty(64, device="cuda"), torch.empty(64, 4, device="cuda")
        fz.sync_mirror()
        fz.act(obs, eps, act, logp, dist)
        with torch.no_grad():
            _, _, ref = a.policy.compute_actions(obs, eps)
        torch.testing.assert_close(dist, ref, rtol=1e-4, atol=1e-4)
    a.stop()


def test_dict_env_api_matches_reference_surface():
    """reset()/step(dict) of `get_lcf_env(MultiAgentIntersectionEnv)` as the reference's scripts use it
    (env_wrappers.py:600-617): agent-id keyed dicts, `__all__`, the info keys of CCEnv / LCFEnv."""
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env
    env = get_lcf_env(MultiAgentIntersectionEnv)(dict(num_agents=10, horizon=50, delay_done=5))
    o = env.reset(force_seed=0)
    assert len(o) == 10 and all(v.shape == (92,) and v.dtype == np.float32 for v in o.values())
    assert set(o) == set(env.vehicles) and env.observation_space["agent0"].contains(o["agent0"])
    seen_done, seen_spawn, seen_max_step = False, False, False
    for t in range(250):
        before = set(env.vehicles)
        # every other vehicle steers off its road: terminations, wrecks that linger `delay_done` steps, respawns
        o, r, d, i = env.step({k: [0.6 if int(k[5:]) % 2 else 0.0, 1.0] for k in env.vehicles})
        assert set(r) == set(d) - {"__all__"} == set(i) and before <= set(i)
        for k in before:
            inf = i[k]
            for key in ("neighbours", "neighbours_distance", "all_agents", "nei_rewards", "global_rewards", "lcf", "lcf_deg",
                        "coordinated_rewards", "native_rewards", "arrive_dest", "crash", "out_of_road", "velocity",
                        "steering", "acceleration", "step_reward", "cost", "episode_length", "episode_reward",
                        "route_completion"):
                assert key in inf, key
            assert inf["native_rewards"] == r[k] and -1 <= inf["lcf"] <= 1 and 0 <= o[k][-1] <= 1
            assert inf["neighbours_distance"] == sorted(inf["neighbours_distance"])
            if inf["neighbours"]:
                assert abs(inf["nei_rewards"] - np.mean([r[n] for n in inf["neighbours"] if n in r])) < 1e-5 or \
                    any(n not in r for n in inf["neighbours"])
            seen_done |= d[k]
            # MultiAgentMetaDrive.done_function: an agent that drove `horizon` steps of its own is done with max_step
            assert inf["max_step"] == (d[k] and inf["episode_length"] >= 50 and not (inf["arrive_dest"] or inf["crash"] or inf["out_of_road"]))
            seen_max_step |= inf["max_step"]
        if d["__all__"]:        # after `horizon` env steps the scene drains; the episode ends with its last agent
            assert t >= 49 and all(d.values())
            break
        assert t < 49 or set(o) <= before, "a scene past its horizon must not respawn"
        seen_spawn |= len(set(o) - before) > 0
    assert seen_done and seen_spawn and d["__all__"]
    env.set_lcf_dist(0.5, 0.2)
    env.close()


@pytest.mark.parametrize("name,algo,map_cls,cfg", [
    # BASELINE.json configs[0]: IPPO Intersection, 4 agents, one scene (the reference's own CPU-runnable case, train_ippo.py)
    ("C1", "ippo", "MultiAgentIntersectionEnv", dict(num_envs=1, env_config=dict(num_agents=4), train_batch_size=200,
                                                      sgd_minibatch_size=64)),
    # BASELINE.json configs[2]: CoPO Roundabout, 40 agents, the 128-scene shard one GPU of the 8-GPU run owns
    ("C3-shard", "copo", "MultiAgentRoundaboutEnv", dict(num_envs=128, env_config=dict(num_agents=40))),
    # configs[3]: CCPPO mean-field on the Tollgate road, 40 agents, bf16 MLPs (losses / advantages stay fp32)
    ("C4", "ccppo", "MultiAgentTollgateEnv", dict(num_envs=64, env_config=dict(num_agents=40), fuse_mode="mf",
                                                   policy_dtype="bfloat16")),
    ("C4-full", "ccppo", "MultiAgentTollgateEnv", dict(num_envs=512, env_config=dict(num_agents=40), fuse_mode="mf",
                                                        policy_dtype="bfloat16", train_batch_size=1024)),
    # configs[4]: CoPO ParkingLot, 10 agents, 240-beam LiDAR (O = 260), LCF meta-update after every env step
    ("C5", "copo", "MultiAgentParkingLotEnv", dict(num_envs=256, env_config=dict(num_agents=10, num_lasers=240),
                                                    train_batch_size=256)),
    ("C5-full", "copo", "MultiAgentParkingLotEnv", dict(num_envs=4096, env_config=dict(num_agents=10, num_lasers=240),
                                                         train_batch_size=4096)),
    # f-4: the Bottleneck map (20 agents, eval/evaluate_population.py:118-124)
    ("Bottleneck", "copo", "MultiAgentBottleneckEnv", dict(num_envs=32, env_config=dict(num_agents=20))),
    # f-4: the procedurally generated road of the base env (train_all_copo_dist.py:30), MetaDrive's `map` key = block count
    ("PG", "copo", "MultiAgentMetaDrive", dict(num_envs=32, env_config=dict(map=4, start_seed=5000))),
])
def test_baseline_parity_configs_run(name, algo, map_cls, cfg):
    """The other BASELINE.json configurations (parity-test cases, not bench lines): shapes, dtypes and a few
    iterations with finite statistics."""
    from copo_amd.torch_copo import algo_ccppo, algo_copo, algo_ippo
    from copo_amd.torch_copo.utils import env_wrappers as W
    base = getattr(W, map_cls)
    if algo == "copo":
        cls, env = algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(base))
    elif algo == "ippo":
        cls, env = algo_ippo.IPPOTrainer, W.get_rllib_compatible_env(base)
    else:
        cls, env = algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(base)
    cfg = dict(cfg, env=env, seed=0)
    cfg.setdefault("train_batch_size", cfg["num_envs"] * 4)
    a = cls(config=cfg)
    if name.startswith("C4"):
        # the bfloat16 policy runs the fused HIP learner in its bfloat16-operand mode (no torch.autocast fallback)
        assert a.policy.fused is not None and a.policy.fused.cfg.operand_dtype == 1 and a.policy.autocast_dtype == torch.bfloat16
        # Tollgate: O = 156 (72 side beams + 6 + 4 lane-line beams + 72 LiDAR + 2 toll columns), mean-field cc-obs 2 * 156 + 2
        assert a.env.sim.O == 156 and a.policy.model.get_centralized_critic_obs_dim() == 2 * 156 + 2 == 314
    if name == "C1":
        assert a.env.sim.O == 91 and a.env.sim.N == 4 and a.sampler.T == 200 and a.policy.fused is not None
    if name.startswith("C5"):
        assert a.env.sim.O == 260 and a.sampler.T == 1
    if name == "PG":
        assert a.env.sim.cfg.map == "pgmap" and a.env.sim.cfg.map_kwargs == dict(sequence=4, seed=5000) and a.env.sim.N == 20
    for _ in range(4):
        res = a.train()
    st = res["info"]["learner"]["default"]["learner_stats"]
    assert all(np.isfinite(v) for v in st.values()), st
    assert a.sampler.obs.dtype == torch.float32 and float(a.sampler.obs.max()) <= 1.0
    a.stop()


def test_data_parallel_code_path_single_rank():
    """COPO_FORCE_DIST=1 takes every data-parallel branch (gradient export + all-reduce + flat Adam, batched meta with
    exported gradient pairs / gathered LCF rows) with one rank: same training trajectory as the local path."""
    import subprocess
    import sys
    code = r'''
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from copo_amd import dist as D
D.init_from_env("cuda")
from copo_amd.torch_copo.algo_copo import CoPOTrainer
from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
a = CoPOTrainer(config=dict(env=env, env_config=dict(num_agents=12), num_envs=16, train_batch_size=16 * 8,
                            sgd_minibatch_size=128, num_sgd_iter=2, lcf_num_iters=2, seed=3, meta_batch_size=4,
                            model={"fcnet_hiddens": [64, 64]}))
assert a.policy.fused is not None
for _ in range(3):
    res = a.train()
w = a.policy.model._hidden_layers[0]._model[0].weight
rs_ok = None
if D.is_dist():
    # the collective the shared dot products of the meta pass use with several ranks (dist.reduce_scatter_sum_): with RCCL and a world of
    # one it must exist in this build and hand the input through
    import torch.distributed as td
    flat = torch.arange(24, dtype=torch.float32, device="cuda").reshape(4, 2, 3)
    out = torch.zeros_like(flat)
    td.reduce_scatter_tensor(out, flat, op=td.ReduceOp.SUM)
    torch.cuda.synchronize()
    rs_ok = bool(torch.equal(out, flat))
print("RESULT " + json.dumps(dict(dist=D.is_dist(), lcf=a.policy.model.lcf_parameters.tolist(), w=float(w.double().abs().sum()),
                                  loss=res["info"]["learner"]["default"]["learner_stats"]["total_loss"], rs_ok=rs_ok)))
'''
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, COPO_FORCE_DIST=force, MASTER_PORT="29541")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        outs.append(__import__("json").loads(line[7:]))
    assert outs[0]["dist"] is False and outs[1]["dist"] is True
    assert outs[1]["rs_ok"] is True
    np.testing.assert_allclose(outs[1]["lcf"], outs[0]["lcf"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(outs[1]["w"], outs[0]["w"], rtol=1e-5)
    np.testing.assert_allclose(outs[1]["loss"], outs[0]["loss"], rtol=1e-3, atol=1e-5)


def test_reference_population_in_the_hip_simulator(golden_dir):
    """f-1 / f-2: a population the reference trained (weights held as data in the eval fixture) is loaded through the
    reference's `.npz` key layout and rolled in the HIP simulator: the device model's action means on LIVE observations
    equal the reference-style numpy policy function's, and the evaluation metrics are a proper partition."""
    from copo_amd.eval import get_policy_function as G
    from copo_amd.eval.checkpoint_io import load_policy_weights
    from copo_amd.eval.evaluate import make_eval_trainer
    gold = np.load(os.path.join(golden_dir, "eval_policy_function.npz"))
    w = {k[len("copo_inter/w/"):]: gold[k] for k in gold.files if k.startswith("copo_inter/w/")}
    t = make_eval_trainer("copo", "inter", num_envs=16, num_agents=40, seed=0, lcf=G.meta_svo_lookup_table["copo_inter"])
    load_policy_weights(t.policy.model, w)
    res = t.evaluate(num_fragments=2, min_episodes=300)
    rates = [res[k] for k in ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean")]
    assert res["num_terminated_agents"] >= 300 and all(0.0 <= r <= 1.0 for r in rates) and abs(sum(rates) - 1.0) < 1e-9
    assert abs(t.env.current_lcf_mean - 0.36824979071031544) < 1e-12
    obs = t.sampler.obs[0].reshape(-1, 92)
    live = obs[t.sampler.flags[0].reshape(-1) > 0][:256]
    logits, _ = t.policy.model({"obs": live})
    ref = G._compute_actions_for_tf_policy(w, live.cpu().numpy(), deterministic=True, policy_name="default", layer_name_suffix="_1")
    np.testing.assert_allclose(logits[:, :2].detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    assert float(live[:, -1].min()) >= 0.0 and float(live[:, -1].max()) <= 1.0      # (lcf + 1) / 2 column
    t.stop()


def test_curriculum_baseline_schedule():
    """f-3: the curriculum baseline (algo_ippo/ippo_cl.py:41-78): a quarter of the target population per quarter of
    training, applied through `ChangeNEnv.close_and_reset_num_agents` (population capacity of the simulator)."""
    from copo_amd.torch_copo.algo_ippo import IPPOTrainer
    from copo_amd.torch_copo.utils.callbacks import curriculum_num_agents, get_change_n_callback
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_change_n_env, get_rllib_compatible_env
    assert [curriculum_num_agents(30, t, 1000) for t in (0, 250, 251, 500, 501, 750, 751, 1000)] == [7, 7, 15, 15, 22, 22, 30, 30]
    env = get_rllib_compatible_env(get_change_n_env(MultiAgentIntersectionEnv))
    total = 8 * 16 * 8                     # 8 iterations of 16 scenes x 8 steps
    a = IPPOTrainer(config=dict(env=env, env_config=dict(num_agents=20, horizon=40), num_envs=16, train_batch_size=16 * 8,
                                sgd_minibatch_size=128, num_sgd_iter=1, seed=0, callbacks=get_change_n_callback(total),
                                model={"fcnet_hiddens": [64, 64]}))
    seen = []
    for it in range(8):
        res = a.train()
        n = a.env.current_num_agents
        seen.append(n)
        fl = a.sampler.flags.reshape(a.sampler.T, 16, 20)
        assert not (fl[:, :, n:] & 1).any()          # nobody acts in a slot beyond the (growing) population
        assert (fl[:, :, :seen[max(0, it - 1)]] & 1).any()
        assert res["custom_metrics"]["num_agents_curriculum"] == n
    assert seen == [5, 5, 10, 10, 15, 15, 20, 20], seen     # quarters at 256 / 512 / 768 of 1024 env steps
    a.stop()


def test_episode_metrics_kernel_equals_tensor_code():
    """copo_episode_metrics == the masked tensor reductions it replaces (utils/callbacks.py:48-110 quantities)."""
    from copo_amd.torch_copo.algo_ippo import IPPOTrainer
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_rllib_compatible_env
    a = IPPOTrainer(config=dict(env=get_rllib_compatible_env(MultiAgentIntersectionEnv), env_config=dict(num_agents=20, horizon=30),
                                num_envs=16, train_batch_size=16 * 40, sgd_minibatch_size=128, num_sgd_iter=1, seed=0,
                                model={"fcnet_hiddens": [64, 64]}))
    a.train()
    b = a._last_batch
    got = a.episode_metrics(b)
    flags = b[SampleBatch.FLAGS].reshape(-1).to(torch.int32)
    acted = (flags & 1) > 0
    done = ((flags & 2) > 0) & acted
    ref = a._metrics_from_sums(a._episode_sums_torch(flags, b["infos"].reshape(-1, 8), acted, done, b["nbr_cnt"].reshape(-1)))
    assert got["num_terminated_agents"] > 0 and set(got) == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-6 * max(1.0, abs(ref[k])), (k, got[k], ref[k])
    a.stop()


@pytest.mark.parametrize("algo", ["ippo", "copo", "ccppo"])
def test_truncated_trajectories_bootstrap_from_the_next_observation(algo):
    """RLlib's PPO postprocessing (the reference's IPPO) bootstraps a trajectory that is cut by the end of the fragment with
    the critic's value of the observation AFTER the last step; the dense path adds that to the last-row scan of
    `copo_gae3_f32` through the linearity of GAE.  Checked against a per-trajectory numpy GAE, for IPPO and for CoPO's three
    observation-only heads."""
    from copo_amd.torch_copo import algo_copo, algo_ippo
    from copo_amd.torch_copo.utils import env_wrappers as W
    if algo == "ippo":
        cls, env, over = algo_ippo.IPPOTrainer, W.get_rllib_compatible_env(W.MultiAgentIntersectionEnv), {}
    elif algo == "ccppo":       # centralised critic: every row is trained one step late, bootstrapped from its successor ROW
        from copo_amd.torch_copo import algo_ccppo
        cls, env, over = algo_ccppo.CCPPOTrainer, algo_ccppo.get_ccppo_env(W.MultiAgentIntersectionEnv), dict(fuse_mode="concat")
    else:
        cls, env, over = algo_copo.CoPOTrainer, W.get_rllib_compatible_env(W.get_lcf_env(W.MultiAgentIntersectionEnv)), {}
    a = cls(config=dict(env=env, env_config=dict(num_agents=12, horizon=60), num_envs=4, train_batch_size=4 * 10, seed=1, **over))
    pol = a.policy
    assert pol.bootstrap_next_obs() != (algo == "ccppo") and pol.wants_lookahead() == (algo == "ccppo")
    if algo == "copo":       # a centralised critic has no next critic observation: the reference's last-row shortcut stays
        from copo_amd.torch_copo import algo_ccppo
        cc = algo_ccppo.CCPPOTrainer(config=dict(env=algo_ccppo.get_ccppo_env(W.MultiAgentIntersectionEnv), env_config=dict(num_agents=4),
                                                 num_envs=1, train_batch_size=10, fuse_mode="mf"))
        assert not cc.policy.bootstrap_next_obs()
        cc.stop()
    for _ in range(3):          # a few fragments in: trajectories start, end and run through the boundaries
        batch = a.sampler.sample()
        if algo == "ccppo":
            prev_last_flags = None if a._look is None else a._look[SampleBatch.FLAGS][a.sampler.T].clone()
            batch = a._lookahead_batch(batch)
    b = pol.postprocess_trajectory(batch)
    if algo == "ccppo":         # rows = [last row of the previous rollout | the first T - 1 new rows]
        assert torch.equal(b[SampleBatch.FLAGS][0], prev_last_flags) and torch.equal(b[SampleBatch.FLAGS][1:], a.sampler.flags[:a.sampler.T - 1])
        assert b["centralized_critic_obs"].shape[0] == a.sampler.T
    T, E, N = b[SampleBatch.FLAGS].shape
    M = E * N
    H = pol.gae_heads()
    vals, adv, tgt = (b[k].reshape(H, T, M).cpu().numpy().astype(np.float64) for k in ("_vals", "_adv", "_tgt"))
    rew = b["rew3"][:H].reshape(H, T, M).cpu().numpy().astype(np.float64)
    fl = b[SampleBatch.FLAGS].reshape(T, M).cpu().numpy()
    if algo == "ccppo":
        v_next = b["_v_next"].cpu().numpy().astype(np.float64)                      # critic values of the successor rows
        cc_last = a._look["centralized_critic_obs"][a.sampler.T] if "centralized_critic_obs" in a._look else pol._cc_buf[a.sampler.T]
        np.testing.assert_allclose(v_next[0], pol.value_heads_dense(cc_last.reshape(M, -1)).cpu().numpy()[0] *
                                   ((a._look[SampleBatch.FLAGS][a.sampler.T].reshape(M) & 1) > 0).cpu().numpy(), rtol=1e-4, atol=1e-4)
    else:
        nxt = b["_next_obs_last"].reshape(M, -1)
        v_next = pol.value_heads_dense(nxt).cpu().numpy().astype(np.float64)          # torch model, [H, M]
    lam, gammas = float(pol.config["lambda"]), pol.gae_gammas()
    checked, cut = 0, 0
    for m in range(M):
        t = 0
        while t < T:
            if not fl[t, m] & 1:
                t += 1
                continue
            t0 = t
            while t < T and (fl[t, m] & 1) and not (fl[t, m] & 2):
                t += 1
            done = t < T and bool(fl[t, m] & 2)
            t1 = t if done else t - 1                     # last row of this trajectory
            t = t1 + 1
            for h in range(H):
                g = gammas[h]
                last = 0.0 if done else v_next[h, m]
                assert done or t1 == T - 1
                v = np.concatenate([vals[h, t0:t1 + 1, m], [last]])
                delta = rew[h, t0:t1 + 1, m] + g * v[1:] - v[:-1]
                want = np.zeros_like(delta)
                acc = 0.0
                for k in range(len(delta) - 1, -1, -1):
                    acc = delta[k] + g * lam * acc
                    want[k] = acc
                np.testing.assert_allclose(adv[h, t0:t1 + 1, m], want, rtol=2e-4, atol=2e-4)
                np.testing.assert_allclose(tgt[h, t0:t1 + 1, m], want + vals[h, t0:t1 + 1, m], rtol=2e-4, atol=2e-4)
            checked += 1
            cut += int(not done)
    assert checked > 40 and cut > 20
    a.stop()


def test_lookahead_bootstrap_with_several_fragments_per_rollout():
    """A centralised critic with `rollout_fragment_length` < rollout length (the reference's CCPPO `_test` config: T = 25,
    fragments of 20): the GAE scan is cut per fragment; a piece that runs into the end of an EARLIER fragment is bootstrapped
    from the value of its own last row (the reference's shortcut, algo_ccppo.py:362-365), only the pieces that run into the end
    of the LAST fragment get the look-ahead row's value (round-2 advisor finding: the correction reached back into the earlier
    fragments)."""
    from copo_amd.torch_copo import algo_ccppo
    from copo_amd.torch_copo.utils import env_wrappers as W
    a = algo_ccppo.CCPPOTrainer(config=dict(env=algo_ccppo.get_ccppo_env(W.MultiAgentIntersectionEnv), env_config=dict(num_agents=12, horizon=80),
                                            num_envs=4, train_batch_size=4 * 25, rollout_fragment_length=20, fuse_mode="mf", seed=2))
    pol = a.policy
    assert pol.wants_lookahead() and a.sampler.T == 25
    for _ in range(3):
        batch = a._lookahead_batch(a.sampler.sample())
    b = pol.postprocess_trajectory(batch)
    T, E, N = b[SampleBatch.FLAGS].shape
    assert T == 25
    M, H, frag = E * N, pol.gae_heads(), 20
    vals, adv, tgt = (b[k].reshape(H, T, M).cpu().numpy().astype(np.float64) for k in ("_vals", "_adv", "_tgt"))
    rew = b["rew3"][:H].reshape(H, T, M).cpu().numpy().astype(np.float64)
    fl = b[SampleBatch.FLAGS].reshape(T, M).cpu().numpy()
    v_next = b["_v_next"].cpu().numpy().astype(np.float64)
    lam, gammas = float(pol.config["lambda"]), pol.gae_gammas()
    pieces = {"done": 0, "cut_inner": 0, "cut_last": 0}
    for m in range(M):
        for lo, hi in ((0, frag), (frag, T)):
            t = lo
            while t < hi:
                if not fl[t, m] & 1:
                    t += 1
                    continue
                t0 = t
                while t < hi and (fl[t, m] & 1) and not (fl[t, m] & 2):
                    t += 1
                done = t < hi and bool(fl[t, m] & 2)
                t1 = t if done else t - 1
                t = t1 + 1
                for h in range(H):
                    g = gammas[h]
                    if done:
                        last, kind = 0.0, "done"
                    elif hi == T:
                        last, kind = v_next[h, m], "cut_last"
                    else:
                        last, kind = vals[h, t1, m], "cut_inner"          # V(last row of the piece): the reference's shortcut
                    v = np.concatenate([vals[h, t0:t1 + 1, m], [last]])
                    delta = rew[h, t0:t1 + 1, m] + g * v[1:] - v[:-1]
                    want, acc = np.zeros_like(delta), 0.0
                    for k in range(len(delta) - 1, -1, -1):
                        acc = delta[k] + g * lam * acc
                        want[k] = acc
                    np.testing.assert_allclose(adv[h, t0:t1 + 1, m], want, rtol=2e-4, atol=2e-4, err_msg=kind)
                    np.testing.assert_allclose(tgt[h, t0:t1 + 1, m], want + vals[h, t0:t1 + 1, m], rtol=2e-4, atol=2e-4, err_msg=kind)
                pieces[kind] += 1
    assert pieces["cut_inner"] > 10 and pieces["cut_last"] > 10, pieces
    a.stop()


@pytest.mark.parametrize("peer", ["0", "tile", "auto", "try-fault"])
def test_two_ranks_fused_data_parallel_on_one_gpu(peer):
    """peer = "tile": the data-parallel step of DESIGN.md section 6 -- gradient tiles summed over the ranks inside the
    weight-gradient kernel (copo_ppo_fused_step_dp_f32), captured chains like the local step; "auto": the default -- that path
    where every rank has a GPU of its own and the start-up probe (copo_amd/dp_probe.py) passes; HERE both ranks share one GPU
    (kernels that wait for their peers would compete with them for compute units), which the default notices: RCCL loop.
    peer = "0": the RCCL loop by name (COPO_DP_EXCHANGE=rccl), with the LCF steps chunk by chunk (meta_seq_per_chunk_dist).
    peer = "try-fault": the tile exchange without a probe, and rank 1 reports a timed-out wait once (`dp_fault_injection_rank`): both
    ranks must agree, restore the state the call started from, leave the exchange and repeat the SAME epochs through the RCCL loop.
    Two real ranks (gloo over CUDA tensors, both on cuda:0 -- RCCL would refuse to share a device) through the fused
    data-parallel path: gradient all-reduce + flat Adam per minibatch, batched meta pass with exported gradient pairs,
    gathered LCF rows.  Ranks own different scenes, must take the same number of steps and end with identical parameters."""
    import subprocess
    import sys
    code = r'''
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from copo_amd import dist as D
rank, _, world = D.init_from_env("cuda")
import torch.distributed as td
from copo_amd.torch_copo.algo_copo import CoPOTrainer
from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
a = CoPOTrainer(config=dict(env=env, env_config=dict(num_agents=12, horizon=60), num_envs=8 + 4 * rank, train_batch_size=(8 + 4 * rank) * 8,
                            sgd_minibatch_size=128, num_sgd_iter=2, lcf_num_iters=2, seed=3, meta_batch_size=4,
                            meta_seq_per_chunk_dist=os.environ.get("COPO_DP_EXCHANGE") == "rccl",     # (one variant: LCF steps chunk by chunk)
                            dp_fault_injection_rank=1 if os.environ.get("COPO_TEST_DP_FAULT") == "1" else -1,
                            model={"fcnet_hiddens": [64, 64]}))
assert a.policy.fused is not None and D.is_dist() and world == 2
for _ in range(3):
    res = a.train()
assert a.policy.dp_reason, a.policy.dp_reason      # the decision is recorded
if os.environ.get("COPO_TEST_DP_FAULT") == "1":
    assert a.policy._dp_mode == "rccl" and "timed out" in a.policy.dp_reason, (a.policy._dp_mode, a.policy.dp_reason)
want_tile = os.environ.get("COPO_DP_EXCHANGE") == "tile"
assert (a.policy._tile is not None) == want_tile and (a.policy._dp_mode == "tile") == want_tile, a.policy._dp_mode
# whole-episode evaluation: the ranks' scenes finish their episode after different fragment counts, the loop holds collectives --
# the stop decision must be the same on both ranks (round-2 advisor finding: a rank-local stop rule hangs here)
ev = a.evaluate(scene_episodes=1)
evs = [None, None]
td.all_gather_object(evs, (ev["num_terminated_agents"], ev["env_steps"] // a.sampler.E))
assert evs[0][0] == evs[1][0] > 0 and evs[0][1] == evs[1][1], evs          # global totals, same number of fragments on both ranks
flat = a.policy.fused.flat.flat
sig = torch.stack([flat.double().sum(), flat.double().abs().sum(), a.policy.model.lcf_parameters[0].double(),
                   a.policy.model.lcf_parameters[1].double(), torch.tensor(float(a.policy.num_grad_updates), dtype=torch.float64, device="cuda")])
both = [torch.zeros_like(sig) for _ in range(2)]
td.all_gather(both, sig)
if rank == 0:
    print("RESULT " + json.dumps(dict(r0=both[0].tolist(), r1=both[1].tolist(), steps=res["agent_timesteps_total"],
                                      loss=res["info"]["learner"]["default"]["learner_stats"]["total_loss"])))
a.stop()
td.destroy_process_group()
'''
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
                   COPO_DIST_BACKEND="gloo", COPO_FORCE_DIST="0", COPO_TEST_DP_FAULT="1" if peer == "try-fault" else "0",
                   COPO_DP_EXCHANGE=peer if peer in ("tile", "auto") else ("try" if peer == "try-fault" else "rccl"))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][-1]
    r = __import__("json").loads(line[7:])
    assert r["r0"] == r["r1"], r                      # bit-identical parameters, LCF parameters and step counts
    assert np.isfinite(r["loss"]) and r["steps"] > 0 and r["r0"][4] > 0


@pytest.mark.parametrize("mod", ["algo_ippo", "algo_ccppo", "algo_copo"])
def test_reference_style_smoke_mains(mod, tmp_path):
    """The `_test()` mains of the reference (algo_*.py: train_batch_size 100, fragment 20, minibatch 30, test mode)
    run end to end through `train()` and leave a progress.csv."""
    import importlib
    import sys
    m = importlib.import_module("copo_amd.torch_copo." + mod)
    argv, sys.argv = sys.argv, [mod]
    try:
        m._test(stop=600, local_dir=str(tmp_path))
    finally:
        sys.argv = argv
    found = [os.path.join(r, f) for r, _, fs in os.walk(tmp_path) for f in fs if f == "progress.csv"]
    assert len(found) == 1
    rows = open(found[0]).read().strip().splitlines()
    assert len(rows) >= 3 and "timesteps_total" in rows[0]


def test_extension_wrappers_through_the_dict_api():
    """f-4: traffic-light message, communication channel and latent wrapper through reset()/step(dict) of
    get_latent_env(get_lcf_env(MultiAgentBottleneckEnv)) -- shapes, the message clock, who hears whom, info keys."""
    from copo_amd.torch_copo.utils import env_wrappers as W
    cls = W.get_latent_env(W.get_lcf_env(W.MultiAgentBottleneckEnv))
    comm = dict(comm_method="broadcast", comm_size=3, comm_neighbours=2, add_pos_in_comm=True)
    conf = dict(num_agents=12, horizon=40, add_traffic_light=True, traffic_light_interval=5, communication=comm,
                enable_latent=True, latent_dim=4)
    env = cls(conf)
    O = 4 + 96 + 3 + 1 + 2 * 6       # Bottleneck: 4 side + 6 + 4 lane-line beams + 10 navigation + 72 LiDAR = 96
    assert env.observation_space["agent0"].shape == (O,) and env.action_space["agent0"].shape == (5,)
    o = env.reset(force_seed=3)
    assert all(v.shape == (O,) and not v[:4].any() for v in o.values())           # no latent registered: zeros
    assert all(v[4 + 96] == 1.0 and not v[4 + 100:].any() for v in o.values())     # message(0) = 1; no comm after reset
    rng = np.random.RandomState(0)
    heard = 0
    for t in range(1, 40):
        acts = {k: np.concatenate([[0.0, 0.8], rng.uniform(-1, 1, 3)]).astype(np.float32) for k in env.vehicles}
        o, r, d, i = env.step(acts)
        msg = (t % 5) / 5 * 0.1 if (t // 5) % 2 == 1 else 1 - (t % 5) / 5 * 0.1
        for k in acts:
            assert o[k].shape == (O,) and o[k][4 + 96] == np.float32(msg)
            inf = i[k]
            assert len(inf["comm_current_obs"]) == 2 and len(inf["nei_obs"]) == 3 and inf["nei_obs"][-1] is None
            for q, n in enumerate(inf["neighbours"][:2]):
                blk = o[k][4 + 100 + 6 * q: 4 + 100 + 6 * (q + 1)]
                np.testing.assert_array_equal(blk, inf["comm_current_obs"][q])
                if n in acts:                      # the neighbour was given an action this step: its message arrives
                    np.testing.assert_array_equal(blk[:3], acts[n][2:])
                    assert np.all((blk[3:] >= 0) & (blk[3:] <= 1))
                    heard += 1
                else:
                    assert not blk.any()
            for q in range(len(inf["neighbours"]), 2):
                assert not o[k][4 + 100 + 6 * q: 4 + 100 + 6 * (q + 1)].any()
    assert heard > 20
    env.register_latent({3: {"agent%d" % a: np.full(4, a, np.float32) for a in range(200)}})
    o, r, d, i = env.step({k: np.zeros(5, np.float32) for k in env.vehicles})
    assert all(np.all(v[:4] == int(k[5:])) for k, v in o.items())
    env.close()
    # vector API: latent tensor concatenated on the device
    venv = cls(dict(conf, num_envs=3))
    out = venv.vec_reset()
    assert out["obs"].shape == (3, 12, O) and not out["obs"][..., :4].any()
    lat = torch.arange(3 * 12 * 4, device="cuda", dtype=torch.float32).view(3, 12, 4)
    venv.register_latent_tensor(lat)
    out = venv.vec_step(torch.zeros(3, 12, 5, device="cuda"))
    assert torch.equal(out["obs"][..., :4], lat) and out["obs"].shape == (3, 12, O)
    venv.close()


@pytest.mark.parametrize("script", ["train_copo", "train_ippo", "train_ccppo", "train_cl"])
def test_launch_scripts_run(script, tmp_path):
    """`python -m copo_amd.torch_copo.train_*.py --test` (the reference's launch scripts, train_copo.py:11-65 etc.) start,
    train to a small step budget and exit cleanly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "copo_amd.torch_copo." + script, "--exp-name", "t", "--test", "--num-envs", "8",
                        "--stop", "3000"], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=root), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "timesteps_total" in r.stdout


def test_evaluate_once_through_recorder_env(tmp_path):
    """f-1: `evaluate_once` + `get_make_env` + `RecorderEnv` on the HIP simulator's dict API: a freshly exported IPPO
    population (the reference's `.npz` layout) rolls two episodes and yields the 31-column evaluation rows."""
    from copo_amd.eval.checkpoint_io import export_policy_npz
    from copo_amd.eval.evaluate_population import evaluate_once, get_make_env
    from copo_amd.torch_copo import algo_ippo
    from copo_amd.torch_copo.utils import env_wrappers as W
    a = algo_ippo.IPPOTrainer(config=dict(env=W.get_rllib_compatible_env(W.MultiAgentIntersectionEnv),
                                          env_config=dict(num_agents=30), num_envs=8, train_batch_size=64, seed=0))
    os.makedirs(tmp_path / "best_checkpoints")
    export_policy_npz(a.policy.model, str(tmp_path / "best_checkpoints" / "ippo_inter.npz"), layout="tf")
    a.stop()
    make_env = get_make_env("inter")
    env = make_env()
    assert env.observation_space["agent0"].shape == (91,) and env.eval_config["neighbours_distance"] == 20
    env.close()

    def short_env():      # 120-step episodes keep the test quick; the scene is the default 30-agent Intersection otherwise
        from copo_amd.eval.recoder import RecorderEnv
        return RecorderEnv(W.MultiAgentIntersectionEnv(dict(num_agents=30, crash_done=True, horizon=120)))

    df = evaluate_once("ippo_inter", short_env, num_episodes=2, root=str(tmp_path), out_dir=str(tmp_path / "res"), verbose=False)
    assert len(df) == 2 and os.path.isfile(tmp_path / "res" / "ippo_inter.csv")
    for col in ("success_rate", "crash_rate", "out_rate", "velocity_step_mean_episode_mean", "energy_step_mean_episode_mean",
                "episode_reward_mean", "episode_cost_sum", "num_agents_total", "svo_estimate_deg_mean", "episode"):
        assert col in df.columns and np.isfinite(df[col]).all(), col
    assert (df["num_agents_total"] >= 30).all() and ((df["success_rate"] + df["crash_rate"] + df["out_rate"]) <= 1.0 + 1e-9).all()


def test_bench_harness_with_two_ranks_on_one_gpu():
    """bench.py the way the driver launches it for N > 1 (one process per rank, env rendezvous), here two ranks sharing
    cuda:0 over gloo: both ranks reach the barriers, rank 0 alone prints ONE JSON line whose value counts both shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571",
                   COPO_DIST_BACKEND="gloo", COPO_FORCE_DIST="0")
        procs.append(subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--num-envs", "32"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    r = json.loads(lines0[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["parallelism"] == "dp2" and "cpu_baseline" not in r
    per_rank_iter = r["config"]["agent_steps_per_iter"]
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 2 * per_rank_iter) <= 1e-3 * 2 * per_rank_iter    # both shards counted
    assert outs[0][0].strip().splitlines()[-1] == lines0[0]        # the JSON is the last line of stdout
    coll = r["config"]["collective"]                            # the data-parallel step's own cost is on the line
    assert coll["allreduce_grad_us"] > 0 and coll["bucket_bytes"] >= 4 * 360201 and coll["backend"] == "gloo"


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_bench_gpus_2_as_one_command_on_a_shared_device():
    """`python bench.py --gpus 2` as ONE command (no WORLD_SIZE in the environment): the script starts its own two ranks.  On a
    one-GPU box they share cuda:0 over gloo (COPO_BENCH_SHARE_DEVICE=1); the line must say n_gpus 2 and count both shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(COPO_BENCH_SHARE_DEVICE="1", COPO_FORCE_DIST="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--num-envs", "32"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    per_rank_iter = d["config"]["agent_steps_per_iter"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * per_rank_iter) <= 1e-3 * 2 * per_rank_iter



@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,nets,mb", [(4, 64, 4, 128), (3, 128, 2, 128), (8, 64, 2, 128),
                                                  (2, 128, 4, 512), (4, 64, 2, 512)])      # 512-row minibatches: the buffer-load instantiation of the weight-gradient kernel
def test_tile_exchange_probe_with_several_ranks_on_one_gpu(world, hidden, nets, mb):
    """The data-parallel tile exchange beyond two ranks (owners dealt round-robin over 3 / 4 / 8 ranks, rank-order sums of up to
    eight partial tiles): `copo_amd/dp_probe.py` -- a learner stepping in captured chains with the exchange and, from the same
    start, with torch.distributed's all-reduce + flat Adam -- as `world` processes that share cuda:0 over gloo.  Every rank must
    end with bit-identical parameters that agree with the all-reduced step to rounding, and no wait may have timed out."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("COPO_FORCE_DIST",)}
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29651",
                   COPO_DIST_BACKEND="gloo", COPO_DP_PROBE_HIDDEN=str(hidden), COPO_DP_PROBE_OBS="20", COPO_DP_PROBE_NETS=str(nets),
                   COPO_DP_PROBE_MB=str(mb), COPO_DP_PROBE_VERBOSE="1")
        procs.append(subprocess.Popen([sys.executable, "-m", "copo_amd.dp_probe"], env=env, cwd=root, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    assert all("identical on all ranks True" in so for so, _ in outs), outs[0][0]


@pytest.mark.gpu
def test_tile_exchange_falls_back_to_the_collective_loop_when_a_wait_times_out():
    """COPO_DP_EXCHANGE=try on a learner whose weight-gradient grid (hidden 256 x 4 nets = 420 workgroups per rank) cannot be
    co-resident for two ranks on ONE GPU: the ranks starve each other, a wait times out (10 s), the error word is raised -- and
    instead of training on partial sums both ranks restore the state the call started from, agree to leave the tile exchange and
    repeat the epochs through the all-reduce loop.  Parameters must end bit-identical on both ranks and must have moved."""
    import subprocess
    import sys
    code = r'''
import json, os, sys, warnings, torch
sys.path.insert(0, os.getcwd())
from copo_amd import dist as D
rank, _, world = D.init_from_env("cuda")
import torch.distributed as td
from copo_amd.dp_probe import _learner
os.environ["COPO_DP_PROBE_HIDDEN"], os.environ["COPO_DP_PROBE_OBS"], os.environ["COPO_DP_PROBE_NETS"] = "256", "92", "4"
pol, batch, R = _learner("try", rank)
before = pol.fused.flat.flat.clone()
pol.prepare_sgd(batch, R, 128)
idx = torch.arange(R, device="cuda")
B_all = D.all_gather_int(R, "cuda")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    st = pol.run_sgd(idx, R, B_all, 128, 2)
fell_back = any("back to the RCCL loop" in str(x.message) for x in w)
flat = pol.fused.flat.flat
every = [torch.empty_like(flat) for _ in range(world)]
td.all_gather(every, flat)
print("RESULT " + json.dumps(dict(same=all(bool(torch.equal(every[0], e)) for e in every), moved=float((flat - before).abs().max()),
                                  mode=pol._dp_mode, fell_back=fell_back, steps=st["num_sgd_steps"], finite=bool(torch.isfinite(flat).all()))))
td.barrier()
td.destroy_process_group()
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671",
                   COPO_DIST_BACKEND="gloo", COPO_FORCE_DIST="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root))
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    for so, _ in outs:
        r = __import__("json").loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert r["same"] and r["finite"] and r["moved"] > 1e-4 and r["steps"] > 0, r
        # (whether the two ranks do starve each other depends on how the GPU interleaves them: if no wait timed out the epochs went
        # through the tile exchange and the mode is unchanged -- either way the parameters above are identical and have moved)
        assert (r["fell_back"] and r["mode"] == "rccl") or (not r["fell_back"] and r["mode"] != "rccl"), r



def test_concurrent_stream_overlaps_the_current_stream():
    """Round 6: `trainer.concurrent_stream` returns a stream whose kernels run UNDER the current stream's -- HIP maps streams onto a
    handful of hardware queues round-robin and a side stream on the main stream's queue serialises with it (the sequential LCF kernels
    then run between the gradient GEMMs: meta passes 5.5 instead of 3.5 ms).  Checked the way the probe checks, on several picks."""
    from copo_amd.trainer import concurrent_stream
    dev = torch.device("cuda")
    main = torch.cuda.current_stream(dev)
    x = torch.zeros(64, device=dev)
    for _ in range(3):
        s = concurrent_stream(dev)
        assert s != main
        with torch.cuda.stream(s):
            x.add_(1.0)
        torch.cuda.synchronize()
        e0, e_main, e_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        torch.cuda._sleep(2_000_000)
        e_main.record(main)
        with torch.cuda.stream(s):
            s.wait_event(e0)
            x.add_(1.0)
            e_side.record(s)
        torch.cuda.synchronize()
        assert e0.elapsed_time(e_side) < 0.5 * e0.elapsed_time(e_main), (e0.elapsed_time(e_side), e0.elapsed_time(e_main))


def test_copo_iteration_has_one_host_read_of_results_and_early_metric_sums():
    """Round 6: a CoPO iteration queues its episode-metric sums behind the rollout (side stream) and reads them back together with the
    meta pass's results; the metrics `train()` reports must be what `episode_metrics` computes from the same rollout afterwards."""
    from copo_amd.torch_copo.algo_copo import CoPOTrainer
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    algo = CoPOTrainer(config=dict(env=env, env_config=dict(num_agents=10), num_envs=8, train_batch_size=64, sgd_minibatch_size=128,
                                   num_sgd_iter=2, lcf_num_iters=2, seed=5))
    for _ in range(4):
        res = algo.train()
        late = algo.episode_metrics(algo._metrics_batch)
        assert algo._metric_sums is None                      # consumed by train()
        for k, v in late.items():
            assert res["custom_metrics"][k] == v, (k, res["custom_metrics"][k], v)
        st = res["info"]["learner"]["default"]["learner_stats"]
        meta = res["info"]["learner"]["default"]["custom_metrics"]["meta_update"]
        assert np.isfinite(st["total_loss"]) and np.isfinite(st["kl"]) and np.isfinite(meta["lcf"]) and meta["lcf_std"] > 0
    algo.stop()

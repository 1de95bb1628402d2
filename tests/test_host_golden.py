"""CPU: the host-side learn path (models, losses, meta-update, trainer arithmetic) against golden vectors recorded
from the reference's own `loss` / `meta_update` / `training_step` / model constructors."""
import os

import numpy as np
import pytest
import torch

from copo_amd.engine import Box, SampleBatch, TorchDiagGaussian, expand_grid, grid_search
from copo_amd.torch_copo import algo_ccppo as C
from copo_amd.torch_copo import algo_copo as A
from copo_amd.torch_copo import algo_ippo as I
from copo_amd.torch_copo.utils.env_wrappers import (MultiAgentIntersectionEnv, get_ccenv, get_lcf_env,
                                                    get_rllib_compatible_env)


def make_config(cls, fuse, hiddens, odim_env="lcf", **over):
    cfg = cls()
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv) if odim_env == "lcf"
                                   else get_ccenv(MultiAgentIntersectionEnv))
    cfg.update_from_dict(dict(env=env, device="cpu", use_hip_graphs=False, model={"fcnet_hiddens": list(hiddens)}, **over))
    if "fuse_mode" in cfg:
        cfg.fuse_mode = fuse
    cfg.validate()
    return cfg


def make_policy(pcls, ccls, fuse, hiddens, odim, **over):
    cfg = make_config(ccls, fuse, hiddens, **over)
    return pcls(Box(-1, 1, (odim,)), Box(-1, 1, (2,)), cfg)


def load_weights(model, g, prefix="w_"):
    sd = {k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(prefix)}
    model.load_state_dict(sd, strict=True)


def batch_from(g, prefix="in_"):
    return SampleBatch({k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files
                        if k.startswith(prefix) and g[k].dtype.kind in "fiu" and g[k].ndim >= 1})


def test_param_counts_and_cc_dims(golden_dir):
    """CoPOModel 360,201 / CCModel(mf) 204,549 params; cc dims 92 / 186 / 468; fp64 LCF params; normc row norms."""
    g = np.load(os.path.join(golden_dir, "param_counts.npz"))
    for tag, pcls, ccls, fuse, odim in [("copo_none_92", A.CoPOPolicy, A.CoPOConfig, "none", 92),
                                        ("cc_mf_92", C.CCPPOPolicy, C.CCPPOConfig, "mf", 92),
                                        ("cc_concat_92", C.CCPPOPolicy, C.CCPPOConfig, "concat", 92),
                                        ("cc_mf_156", C.CCPPOPolicy, C.CCPPOConfig, "mf", 156),
                                        ("copo_none_260", A.CoPOPolicy, A.CoPOConfig, "none", 260)]:
        pol = make_policy(pcls, ccls, fuse, [256, 256], odim)
        m = pol.model
        assert sum(p.numel() for p in m.parameters()) == int(g[tag + "_nparams"]), tag
        assert m.get_centralized_critic_obs_dim() == int(g[tag + "_ccdim"])
        assert sorted(m.state_dict().keys()) == list(g[tag + "_keys"]), tag
        np.testing.assert_allclose(np.linalg.norm(m._hidden_layers[0]._model[0].weight.detach().numpy(), axis=1)[:4],
                                   g[tag + "_row_norm_hidden"], rtol=1e-5)
        np.testing.assert_allclose(np.linalg.norm(m._logits._model[0].weight.detach().numpy(), axis=1),
                                   g[tag + "_row_norm_head"], rtol=1e-5)
        if pcls is A.CoPOPolicy:
            assert m.lcf_parameters.dtype == torch.float64 and bool(g[tag + "_lcf_is_f64"])
            np.testing.assert_allclose(m.lcf_parameters.detach().numpy(), g[tag + "_lcf_param"], rtol=1e-15)
            np.testing.assert_allclose([m.lcf_mean.item(), m.lcf_std.item()], g[tag + "_lcf_mean_std"], rtol=1e-12)
    assert C.get_centralized_critic_obs_dim(Box(-1, 1, (92,)), Box(-1, 1, (2,)), True, 4, "concat") == 468
    assert C.get_centralized_critic_obs_dim(Box(-1, 1, (92,)), Box(-1, 1, (2,)), False, 4, "mf") == 184


def test_diag_gaussian_matches_torch_distributions():
    torch.manual_seed(0)
    a, b = torch.randn(64, 4), torch.randn(64, 4) * 0.3
    x = torch.randn(64, 2)
    da, db = TorchDiagGaussian(a), TorchDiagGaussian(b)
    na = torch.distributions.Normal(a[:, :2], a[:, 2:].exp())
    nb = torch.distributions.Normal(b[:, :2], b[:, 2:].exp())
    torch.testing.assert_close(da.logp(x), na.log_prob(x).sum(-1))
    torch.testing.assert_close(da.entropy(), na.entropy().sum(-1))
    torch.testing.assert_close(da.kl(db), torch.distributions.kl_divergence(na, nb).sum(-1))
    assert abs(TorchDiagGaussian(torch.zeros(1, 4)).entropy().item() - 2.838) < 1e-3    # progress.csv first row


@pytest.mark.parametrize("tag,pcls,ccls,fuse,over", [
    ("ippo", I.IPPOPolicy, I.IPPOConfig, "none", {}),
    ("ccppo_mf", C.CCPPOPolicy, C.CCPPOConfig, "mf", {}),
    ("ccppo_concat", C.CCPPOPolicy, C.CCPPOConfig, "concat", {}),
    ("copo", A.CoPOPolicy, A.CoPOConfig, "none", {}),
    ("copo_newvf", A.CoPOPolicy, A.CoPOConfig, "none", dict(old_value_loss=False, vf_clip_param=10.0)),
    ("copo_nokl", A.CoPOPolicy, A.CoPOConfig, "none", dict(kl_coeff=0.0)),
])
def test_losses_vs_reference(golden_dir, tag, pcls, ccls, fuse, over):
    """IPPOPolicy.loss / CCPPOPolicy.loss / CoPOPolicy.loss (algo_ippo.py:78-172, algo_ccppo.py:376-472,
    algo_copo.py:311-424): total loss, every tower stat and every parameter gradient."""
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % tag))
    pol = make_policy(pcls, ccls, fuse, [32, 32], 12, **over)
    model = pol.model
    if pcls is I.IPPOPolicy:
        # the fixture's IPPO net is a CCModel(fuse none): same tensors, same key names as the stock FC net
        load_weights(model, g)
    else:
        load_weights(model, g)
    tb = batch_from(g)
    model.zero_grad()
    loss = pol.loss(model, TorchDiagGaussian, tb)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["out_total_loss"], rtol=1e-5)
    for k in g.files:
        if k.startswith("out_stat_") and k[9:] in model.tower_stats:
            v = model.tower_stats[k[9:]]
            v = v.detach() if hasattr(v, "detach") else v
            np.testing.assert_allclose(float(v), float(g[k]), rtol=1e-5, atol=1e-7, err_msg=k)
    for name, p in model.named_parameters():
        ref = g["out_grad_" + name]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
        else:
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=1e-6, err_msg=name)


def test_masked_loss_equals_unpadded(golden_dir):
    """Static-shape minibatches: zero-weight padding rows must not change the loss or its gradient."""
    g = np.load(os.path.join(golden_dir, "loss_copo.npz"))
    pol = make_policy(A.CoPOPolicy, A.CoPOConfig, "none", [32, 32], 12)
    load_weights(pol.model, g)
    tb = batch_from(g)
    B = tb[SampleBatch.OBS].shape[0]
    ref = pol.loss(pol.model, TorchDiagGaussian, tb)
    pad = SampleBatch({k: torch.cat([v, v[:17] * 0 + 3.0]) for k, v in tb.items()})
    pad[SampleBatch.VALID] = torch.cat([torch.ones(B), torch.zeros(17)])
    out = pol.loss(pol.model, TorchDiagGaussian, pad)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    # data-parallel form: two half batches with the global denominator add up to the full loss (minus the KL bias)
    halves = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        h = SampleBatch({k: v[sl] for k, v in tb.items()})
        h[SampleBatch.VALID] = torch.ones(h[SampleBatch.OBS].shape[0])
        h["valid_denominator"] = torch.tensor(float(B))
        halves.append(pol.loss(pol.model, TorchDiagGaussian, h))
    torch.testing.assert_close(halves[0] + halves[1], ref, rtol=1e-5, atol=1e-5)


def test_meta_update_vs_reference(golden_dir):
    """CoPOPolicy.meta_update (algo_copo.py:228-309): grad_value, LCF loss, fp64 LCF parameters after each Adam step."""
    g = np.load(os.path.join(golden_dir, "meta_update.npz"))
    pol = make_policy(A.CoPOPolicy, A.CoPOConfig, "none", [32, 32], 12)
    load_weights(pol.model, g, "w_")
    load_weights(pol.target_model, g, "wt_")
    pol._lcf_optimizer = torch.optim.Adam([pol.model.lcf_parameters], lr=1e-4)
    pol._raw_lcf_adv_mean = torch.tensor(float(g["in_raw_mean_std"][0]), dtype=torch.float64)
    pol._raw_lcf_adv_std = torch.tensor(float(g["in_raw_mean_std"][1]), dtype=torch.float64)
    for s in range(int(g["n_steps"])):
        tb = SampleBatch({k[len("s%d_in_" % s):]: g[k] for k in g.files if k.startswith("s%d_in_" % s)})
        eps = torch.as_tensor(tb.pop("eps"))
        stats = pol.meta_update(tb, eps=eps)
        for k in ("new_policy_ego_loss", "old_policy_logp_loss", "lcf_lcf_adv_loss", "coordinated_adv", "global_adv"):
            np.testing.assert_allclose(stats[k], float(g["s%d_out_%s" % (s, k)]), rtol=2e-5, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(stats["grad_value"], float(g["s%d_out_grad_value" % s]), rtol=2e-4, atol=1e-9)
        np.testing.assert_allclose(stats["lcf_final_loss"], float(g["s%d_out_lcf_final_loss" % s]), rtol=2e-4, atol=1e-9)
        np.testing.assert_allclose(pol.model.lcf_parameters.detach().numpy(), g["s%d_out_lcf_parameters" % s],
                                   rtol=1e-7, atol=1e-9)
        for k in ("lcf", "lcf_deg", "lcf_param", "lcf_std", "lcf_std_deg", "lcf_std_param"):
            np.testing.assert_allclose(stats[k], float(g["s%d_out_%s" % (s, k)]), rtol=1e-6, atol=1e-9, err_msg=k)


def test_meta_loop_of_training_step_vs_reference(golden_dir):
    """The LCF half of CoPOTrainer.training_step (algo_copo.py:581-613): 5 x ceil(1200/512) unshuffled meta steps
    starting from the reference's own normalised batch; final LCF parameters and the (mean, std) pushed to envs."""
    g = np.load(os.path.join(golden_dir, "training_step.npz"))
    pol = make_policy(A.CoPOPolicy, A.CoPOConfig, "none", [32, 32], 12)
    load_weights(pol.model, g, "w_")
    load_weights(pol.target_model, g, "wt_")
    pol._lcf_optimizer = torch.optim.Adam([pol.model.lcf_parameters], lr=1e-4)
    mean_std = g["out_raw_mean_std"]
    pol._raw_lcf_adv_mean = torch.tensor(float(np.float32(mean_std[0])), dtype=torch.float64)
    pol._raw_lcf_adv_std = torch.tensor(float(np.float32(mean_std[1])), dtype=torch.float64)
    b = batch_from(g)
    b[A.GLOBAL_ADVANTAGES] = torch.as_tensor(g["out_global_advantages"])
    B = b[SampleBatch.OBS].shape[0]
    torch.manual_seed(int(g["in_torch_seed"]))
    rec = []
    for _ in range(5):
        for lo in range(0, B, 512):
            mb = SampleBatch({k: v[lo:lo + 512] for k, v in b.items()})
            n = mb[SampleBatch.OBS].shape[0]
            eps = torch.randn(n, dtype=torch.float64)      # the draw Normal.rsample makes in the reference
            rec.append(pol.meta_update(mb, eps=eps))
    np.testing.assert_allclose(pol.model.lcf_parameters.detach().numpy(), g["out_lcf_parameters"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose([pol.model.lcf_mean.item(), pol.model.lcf_std.item()], g["out_env_lcf_dist"], rtol=1e-6,
                               atol=1e-9)
    assert len(rec) == 15
    np.testing.assert_allclose(np.mean([r["grad_value"] for r in rec]), float(g["out_meta_grad_value"]), rtol=1e-3,
                               atol=1e-9)
    pol.update_old_policy()
    assert all(torch.equal(a, b_) for a, b_ in zip(pol.model.state_dict().values(), pol.target_model.state_dict().values()))


def test_update_kl_rule():
    pol = make_policy(I.IPPOPolicy, I.IPPOConfig, "none", [8, 8], 12)
    assert pol.update_kl(0.03) == pytest.approx(0.3)          # > 2 * 0.01
    assert pol.update_kl(0.012) == pytest.approx(0.3)
    assert pol.update_kl(0.001) == pytest.approx(0.15)        # < 0.5 * 0.01
    assert float(pol.kl_coeff) == pytest.approx(0.15)


def test_config_surface_and_grid():
    cfg = A.CoPOConfig()
    assert (cfg["sgd_minibatch_size"], cfg["rollout_fragment_length"], cfg["train_batch_size"], cfg["num_sgd_iter"]) == \
        (512, 200, 2000, 5)
    assert (cfg["lr"], cfg["clip_param"], cfg["lambda"], cfg["vf_clip_param"], cfg["old_value_loss"]) == \
        (3e-4, 0.2, 0.95, 100, True)
    assert (cfg["initial_lcf_std"], cfg["lcf_num_iters"], cfg["lcf_lr"], cfg["fuse_mode"]) == (0.1, 5, 1e-4, "none")
    assert C.CCPPOConfig()["fuse_mode"] == "mf" and C.CCPPOConfig()["mf_nei_distance"] == 10
    cfg.update_from_dict(dict(initial_svo_std=0.1, svo_lr=1e-4, svo_num_iters=5, use_global_value=True,
                              use_centralized_critic=False))     # dead TF keys of train_copo.py are tolerated
    trials = expand_grid(dict(env=grid_search(["a", "b"]), seed=grid_search([0, 100, 200]), x=1,
                              env_config=dict(start_seed=grid_search([5000, 6000]))))
    assert len(trials) == 12 and {t["env"] for t in trials} == {"a", "b"}


def test_callbacks_vs_reference(golden_dir):
    """MultiAgentDrivingCallbacks.on_episode_end / on_train_result arithmetic (utils/callbacks.py:48-148)."""
    from collections import defaultdict
    from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
    g = np.load(os.path.join(golden_dir, "callbacks.npz"))
    term, lens = g["in_term"], g["in_len"]
    infos, user = {}, {k: defaultdict(list) for k in MultiAgentDrivingCallbacks.STEP_KEYS}
    for a in range(len(term)):
        er = 0.0
        for s in range(lens[a]):
            for k in ("velocity", "steering", "step_reward", "acceleration", "cost"):
                user[k][a].append(g["in_" + k][a, s])
            er += g["in_step_reward"][a, s]
            user["episode_length"][a].append(s + 1)
            user["episode_reward"][a].append(er)
            user["num_neighbours"][a].append(g["in_num_neighbours"][a, s])
        infos[a] = dict(arrive_dest=term[a] == 0, crash=term[a] == 1, out_of_road=term[a] == 2,
                        route_completion=g["in_route_completion"][a], track_length=100.0, current_distance=50.0)
    m = MultiAgentDrivingCallbacks.summarize_episode(infos, user)
    for k in g.files:
        if k.startswith("out_"):
            np.testing.assert_allclose(m[k[4:]], g[k], rtol=1e-12, err_msg=k)
    result = dict(custom_metrics={k + "_mean": v for k, v in m.items()}, episode_len_mean=17.0, episode_reward_mean=123.0,
                  policy_reward_mean={"default": 4.5})
    MultiAgentDrivingCallbacks().on_train_result(algorithm=None, result=result)
    for k in g.files:
        if k.startswith("res_"):
            np.testing.assert_allclose(result[k[4:]], g[k], rtol=1e-12, err_msg=k)


def test_module_surface_of_the_reference():
    """the names a `train_*.py`-shaped script imports from `copo.torch_copo` exist under the same module
    paths, and the four launch scripts import (their __main__ blocks are guarded)."""
    import importlib
    want = {
        "algo_copo": ["CoPOConfig", "CoPOModel", "CoPOPolicy", "CoPOTrainer", "NEI_REWARDS", "NEI_VALUES", "NEI_ADVANTAGE",
                      "NEI_TARGET", "LCF_LR", "GLOBAL_VALUES", "GLOBAL_REWARDS", "GLOBAL_ADVANTAGES", "GLOBAL_TARGET",
                      "USE_CENTRALIZED_CRITIC", "CENTRALIZED_CRITIC_OBS", "COUNTERFACTUAL", "USE_DISTRIBUTIONAL_LCF"],
        "algo_ccppo": ["CCPPOConfig", "CCModel", "CCPPOPolicy", "CCPPOTrainer", "get_ccppo_env", "get_centralized_critic_obs_dim",
                       "CENTRALIZED_CRITIC_OBS", "COUNTERFACTUAL"],
        "algo_ippo": ["IPPOConfig", "IPPOPolicy", "IPPOTrainer"],
        "utils.env_wrappers": ["get_lcf_env", "get_ccenv", "get_rllib_compatible_env", "get_latent_env", "get_change_n_env",
                               "MultiAgentIntersectionEnv", "MultiAgentRoundaboutEnv", "MultiAgentTollgateEnv",
                               "MultiAgentBottleneckEnv", "MultiAgentParkingLotEnv", "CCEnv", "LCFEnv", "COMM_ACTIONS",
                               "COMM_CURRENT_OBS", "COMM_METHOD", "NEI_OBS", "ENV_PREV_OBS"],
        "utils.train": ["train"],
        "utils.utils": ["get_train_parser", "initialize_ray", "setup_logger", "merge_dicts", "deep_update", "get_time_str",
                        "SafeJSONEncoder"],
        "utils.callbacks": ["MultiAgentDrivingCallbacks"],
    }
    for mod, names in want.items():
        m = importlib.import_module("copo_amd.torch_copo." + mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    for script in ("train_copo", "train_ippo", "train_ccppo", "train_cl"):
        importlib.import_module("copo_amd.torch_copo." + script)
    p = importlib.import_module("copo_amd.torch_copo.utils.utils").get_train_parser()
    a = p.parse_args(["--exp-name", "x", "--num-gpus", "0", "--num-seeds", "1", "--test"])
    assert a.exp_name == "x" and a.test
    from copo_amd.torch_copo.utils.utils import deep_update, merge_dicts
    base = dict(x=1, m=dict(a=1, b=2))
    assert merge_dicts(base, dict(m=dict(b=3, c=4), y=2)) == dict(x=1, m=dict(a=1, b=3, c=4), y=2) and base["m"] == dict(a=1, b=2)
    with pytest.raises(Exception):
        deep_update(dict(x=1), dict(z=2))
    assert deep_update(dict(m=dict(type="a", p=1)), dict(m=dict(type="b")), False, None, ["m"]) == dict(m=dict(type="b"))


def test_bootstrap_correction_is_gae_with_the_next_value():
    """trainer.PPOPolicyBase._apply_bootstrap (CPU): the last-row scan of the GAE op plus the linear correction == GAE with an
    explicit bootstrap value for every trajectory that runs into the end of the fragment (RLlib's `last_r = V(next obs)`),
    for trajectories that start, end and restart inside the fragment, three heads with their own gammas."""
    import oracle_lib as ol
    from copo_amd.trainer import PPOPolicyBase
    rng = np.random.RandomState(4)
    H, T, M, lam, gammas = 3, 9, 40, 0.95, [0.99, 0.99, 1.0]
    acted = rng.uniform(size=(T, M)) > 0.15
    done = acted & (rng.uniform(size=(T, M)) < 0.12)
    flags = (acted * 1 + done * 2).astype(np.uint8)
    rew = rng.normal(0, 1, (H, T, M)).astype(np.float32)
    val = rng.normal(0, 5, (H, T, M)).astype(np.float32)
    v_next = rng.normal(0, 5, (H, M)).astype(np.float32)
    adv, tgt = ol.gae3(rew, val, flags, gammas, lam)          # bootstraps a cut trajectory from its last row

    class P:
        _boot_w = None

        def gae_gammas(self):
            return gammas

    a, g = torch.from_numpy(adv.copy()), torch.from_numpy(tgt.copy())
    PPOPolicyBase._apply_bootstrap(P(), torch.from_numpy(val), a, g, torch.from_numpy(flags), lam, torch.from_numpy(v_next))
    a, g = a.numpy().astype(np.float64), g.numpy().astype(np.float64)
    checked = 0
    for m in range(M):
        t = 0
        while t < T:
            if not acted[t, m]:
                t += 1
                continue
            t0 = t
            while t < T and acted[t, m] and not done[t, m]:
                t += 1
            ended = t < T and done[t, m]
            t1 = t if ended else t - 1
            t = t1 + 1
            for h in range(H):
                if ended:
                    last = 0.0
                elif t1 == T - 1:
                    last = float(v_next[h, m])
                else:
                    last = float(val[h, t1, m])      # cut by an empty slot inside the fragment: the op's own shortcut stays
                v = np.concatenate([val[h, t0:t1 + 1, m].astype(np.float64), [last]])
                delta = rew[h, t0:t1 + 1, m] + gammas[h] * v[1:] - v[:-1]
                acc, want = 0.0, np.zeros_like(delta)
                for k in range(len(delta) - 1, -1, -1):
                    acc = delta[k] + gammas[h] * lam * acc
                    want[k] = acc
                np.testing.assert_allclose(a[h, t0:t1 + 1, m], want, rtol=1e-4, atol=1e-4)
                np.testing.assert_allclose(g[h, t0:t1 + 1, m], want + val[h, t0:t1 + 1, m], rtol=1e-4, atol=1e-4)
            checked += 1
    assert checked > 60

"""Evaluation path vs golden vectors produced by the reference's own `copo/eval/get_policy_function.py`
(oracle/gen_golden_eval.py): numpy policy functions in both key layouts, the dict call shape with the LCF column,
and the npz <-> torch-model wire format."""
import os

import numpy as np
import pytest
import torch

from copo_amd.eval import checkpoint_io as CK
from copo_amd.eval import get_policy_function as G


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_policy_function.npz"))


def _weights(gold, name):
    pre = name + "/w/"
    return {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}


@pytest.mark.parametrize("name", ["ippo_inter", "copo_inter", "ccppo_inter"])
def test_numpy_policy_function_matches_reference(gold, name, tmp_path):
    w = _weights(gold, name)
    obs = gold[name + "/obs"]
    layout, sfx = G.population_layout(name)
    assert G.detect_layout(w) == layout
    if layout == "tf":
        mean = G._compute_actions_for_tf_policy(w, obs, deterministic=True, policy_name="default", layer_name_suffix=sfx)
    else:
        mean = G._compute_actions_for_torch_policy(w, obs, deterministic=True)
    np.testing.assert_array_equal(mean, gold[name + "/mean"])
    # population files on disk, stochastic path under the reference's numpy seed
    np.savez(os.path.join(tmp_path, name + ".npz"), **w)
    f = G.get_policy_function(name, checkpoint_dir_name=".", root=str(tmp_path))
    np.random.seed(11)
    np.testing.assert_array_equal(f(obs), gold[name + "/sampled"])
    pf = G.PolicyFunction(model_name=name, checkpoint_dir_name=".", root=str(tmp_path))
    base = gold[name + "/dict_obs"]
    np.random.seed(13)
    act = pf({"agent%d" % i: base[i] for i in range(5)}, {"agent1": True, "agent3": False})
    assert sorted(act) == list(gold[name + "/dict_keys"])
    np.testing.assert_array_equal(np.stack([act[k] for k in sorted(act)]), gold[name + "/dict_actions"])
    if name.startswith("copo"):
        assert G.meta_svo_lookup_table[name] == tuple(gold[name + "/lcf"])
        np.testing.assert_array_equal([pf.existing_svo[k] for k in sorted(pf.existing_svo)], gold[name + "/dict_lcf"])
        pf.reset()
        assert not pf.existing_svo


@pytest.mark.parametrize("name,kind,odim", [("ippo_inter", "ippo", 91), ("copo_inter", "copo", 92), ("ccppo_inter", "ccppo", 91)])
def test_population_loads_into_the_torch_models(gold, name, kind, odim, tmp_path):
    """Reference-trained populations drive this build's models: same means as the reference's numpy forward; and the
    export writes files the reference-side functions read back to the same numbers (both layouts)."""
    from copo_amd.engine import Box
    from copo_amd.torch_copo import algo_ccppo, algo_copo, algo_ippo
    from copo_amd.torch_copo.utils import env_wrappers as W
    base = W.MultiAgentIntersectionEnv
    pcls, ccls, env = dict(
        ippo=(algo_ippo.IPPOPolicy, algo_ippo.IPPOConfig, W.get_rllib_compatible_env(base)),
        copo=(algo_copo.CoPOPolicy, algo_copo.CoPOConfig, W.get_rllib_compatible_env(W.get_lcf_env(base))),
        ccppo=(algo_ccppo.CCPPOPolicy, algo_ccppo.CCPPOConfig, algo_ccppo.get_ccppo_env(base)))[kind]
    cfg = ccls()
    cfg.update_from_dict(dict(env=env, device="cpu", use_hip_graphs=False, use_fused_learner=False))
    cfg.validate()
    model = pcls(Box(-1, 1, (odim,)), Box(-1, 1, (2,)), cfg).model
    w = _weights(gold, name)
    keys = CK.load_policy_weights(model, w)
    assert len(keys) == 6
    obs = torch.as_tensor(gold[name + "/obs"])
    logits, _ = model({"obs": obs})
    np.testing.assert_allclose(logits[:, :2].detach().numpy(), gold[name + "/mean"], rtol=2e-5, atol=2e-6)
    for layout, sfx in (("torch", ""), ("tf", "_1")):
        path = os.path.join(tmp_path, "x_%s.npz" % layout)
        CK.export_policy_npz(model, path, layout=layout, policy_name="default", layer_name_suffix=sfx)
        with np.load(path) as f:
            w2 = {k: f[k] for k in f.files}
        m2 = (G._compute_actions_for_torch_policy(w2, obs.numpy(), deterministic=True) if layout == "torch" else
              G._compute_actions_for_tf_policy(w2, obs.numpy(), deterministic=True, policy_name="default", layer_name_suffix=sfx))
        np.testing.assert_allclose(m2, gold[name + "/mean"], rtol=2e-5, atol=2e-6)
    # Tune-style pickle round trip (worker -> state -> policy)
    ck = os.path.join(tmp_path, "checkpoint-1")
    CK.save_tune_style_checkpoint(model, ck)
    # this build writes the torch key layout for every algorithm: the reader goes by the keys it finds, not by the name
    pf = CK.get_policy_function_from_checkpoint(kind, ck, deterministic=True)
    obs2 = gold[name + "/obs"][:2]
    act = pf.policy(obs2)
    np.testing.assert_allclose(act, gold[name + "/mean"][:2], rtol=2e-5, atol=2e-6)
    # indexed CoPO populations resolve their LCF distribution; an unknown one fails with a clear error
    assert G.PolicyFunction(policy=lambda o: o).lcf_dist is None
    assert G.meta_svo_lookup_table["copo_inter_0"] == G.meta_svo_lookup_table["copo_inter"] and len(G.meta_svo_lookup_table) == 37


class _ScriptedStream:
    """Dict-API env that replays the arrays of tests/golden/recorder.npz (what the reference's RecorderEnv wrapped when
    the fixture was recorded)."""
    INFO = ("velocity", "steering", "step_reward", "acceleration", "cost", "episode_length", "episode_reward", "step_energy",
            "episode_energy")

    def __init__(self, g, c):
        self.s = {k[len("c%d_in_" % c):]: g[k] for k in g.files if k.startswith("c%d_in_" % c)}
        self.t, self.vehicles = 0, {}

    def reset(self):
        self.t = 0
        return {}

    def close(self):
        pass

    def step(self, actions):
        from types import SimpleNamespace
        s, t = self.s, self.t
        o, r, d, i = {}, {}, {}, {}
        self.vehicles = {}
        for n in range(s["present"].shape[1]):
            if not s["present"][t, n]:
                continue
            k = "agent%d" % s["aid"][t, n]
            self.vehicles[k] = SimpleNamespace(position=s["pos"][t, n])
            o[k], r[k], d[k] = np.zeros(3, np.float32), float(s["rew"][t, n]), bool(s["done"][t, n])
            if s["first"][t, n]:
                i[k] = {}
                continue
            i[k] = {key: float(s[key][t, n]) for key in self.INFO}
            i[k]["raw_action"] = s["raw_action"][t, n]
            if d[k]:
                kd = int(s["kind"][t, n])
                i[k].update(arrive_dest=kd == 0, crash=kd == 1, out_of_road=kd == 2)
        d["__all__"] = t == s["present"].shape[0] - 1
        self.t += 1
        return o, r, d, i


def test_recorder_env_vs_reference(golden_dir):
    """f-1: `RecorderEnv.get_episode_result / get_step_result` and `DistanceMap` (copo/eval/recoder.py:16-349): the 31
    episode statistics the reference computed on three scripted episodes (agents appearing, acting, terminating with
    every outcome; neighbourhoods of 12 / 20 / 35 m)."""
    from copo_amd.eval.recoder import DistanceMap, RecorderEnv
    g = np.load(os.path.join(golden_dir, "recorder.npz"))
    for c in range(int(g["n_cases"])):
        env = RecorderEnv(_ScriptedStream(g, c), eval_config=dict(neighbours_distance=float(g["c%d_in_distance" % c])))
        env.reset()
        T = g["c%d_in_present" % c].shape[0]
        q = 0
        for t in range(T):
            _, _, d, _ = env.step({})
            if t in (5, 17):
                sr = env.get_step_result()
                assert sorted(sr) == list(g["c%d_step%d_keys" % (c, q)])
                np.testing.assert_allclose([float(sr[k]) for k in sorted(sr)], g["c%d_step%d_vals" % (c, q)], rtol=1e-12, atol=1e-12)
                q += 1
        assert d["__all__"]
        res = env.get_episode_result()
        assert sorted(res) == list(g["c%d_keys" % c]) and len(res) == 31
        np.testing.assert_allclose([float(res[k]) for k in sorted(res)], g["c%d_vals" % c], rtol=1e-12, atol=1e-12)
    # the survey's known-answer scene (a0 (0,0), a1 (3,4), a2 (30,0), a3 (100,0)) through the evaluation-side distance map
    from types import SimpleNamespace
    dm = DistanceMap()
    dm.update_distance_map({k: SimpleNamespace(position=p) for k, p in
                            zip(["a0", "a1", "a2", "a3"], [(0, 0), (3, 4), (30, 0), (100, 0)])})
    assert dm.find_in_range("a0", 40) == ["a1", "a2"] and dm.find_in_range("a2", 40) == ["a1", "a0"] and dm.find_in_range("a3", 40) == []
    own, nei, cnt = dm.get_rewards(dict(a0=1.0, a1=2.0, a2=3.0, a3=4.0), 40)
    assert (nei["a0"], nei["a1"], nei["a2"], nei["a3"]) == (2.5, 2.0, 1.5, 4.0) and cnt["a3"] == 0


def test_vectorised_recorder_equals_the_reference_recorder(golden_dir):
    """`copo_amd.eval.vec_recorder.VecRecorder` (the evaluation table computed from the sampler's dense [T, E, N] tensors) against
    what the REFERENCE's RecorderEnv returned for the scripted episodes of tests/golden/recorder.npz (eval/recoder.py:188-299):
    every column the vectorised recorder reports.  The scripted stream is turned into the simulator's tensor format: flags (acted
    / spawned / done / outcome), the info columns, and the neighbour counts within the recorder's radius."""
    import torch
    from copo_amd.eval.vec_recorder import (COLUMNS, F_ACTED, F_ARRIVE, F_CRASH, F_DONE, F_ENV_RESET, F_OUT, F_SPAWNED, VecRecorder)
    g = np.load(os.path.join(golden_dir, "recorder.npz"))
    for c in range(int(g["n_cases"])):
        s = {k[len("c%d_in_" % c):]: g[k] for k in g.files if k.startswith("c%d_in_" % c)}
        T, N = s["present"].shape
        present, first, done = s["present"], s["first"] & s["present"], s["done"] & s["present"]
        acted = present & ~first
        fl = np.zeros((T, 1, N), np.uint8)
        fl[:, 0][acted] |= F_ACTED
        fl[:, 0][first] |= F_SPAWNED
        fl[:, 0][done & acted] |= F_DONE
        for kind, bit in ((0, F_ARRIVE), (1, F_CRASH), (2, F_OUT)):
            fl[:, 0][done & acted & (s["kind"] == kind)] |= bit
        fl[T - 1, 0, :] |= F_ENV_RESET
        info = np.zeros((T, 1, N, 8), np.float32)
        for col, key in ((0, "velocity"), (4, "cost"), (5, "episode_length"), (6, "episode_reward")):
            info[:, 0, :, col] = np.where(acted, s[key], 0.0)
        dist = float(s["distance"])
        d = np.linalg.norm(s["pos"][:, :, None, :] - s["pos"][:, None, :, :], axis=-1)          # [T, N, N]
        both = present[:, :, None] & present[:, None, :] & ~np.eye(N, dtype=bool)[None]
        nbr = ((d < dist) & both).sum(-1).astype(np.int32)[:, None, :]
        rec = VecRecorder(1, "cpu")
        # float32 info columns: the reference's statistics in float64 on the same numbers rounded to float32
        rec.add(dict(flags=torch.from_numpy(fl), infos=torch.from_numpy(info), nbr_cnt=torch.from_numpy(nbr)))
        assert len(rec.rows) == 1
        ref = dict(zip(list(g["c%d_keys" % c]), g["c%d_vals" % c]))
        row = rec.rows[0]
        for col in COLUMNS:
            if col == "env_episode_steps":
                assert row[col] == T
                continue
            np.testing.assert_allclose(row[col], float(ref[col]), rtol=2e-6, atol=2e-6, err_msg="case %d: %s" % (c, col))

"""CPU: pin the oracle (oracle/copo_oracle.c) against golden vectors produced by the reference's own code."""
import os

import numpy as np
import pytest

import oracle_lib as ol


@pytest.mark.parametrize("name", ["intersection", "roundabout"])
def test_the_oracles_own_map_tables_agree_with_the_products(name):
    """oracle/oracle_maps.c derives the two maps' tables from MetaDrive's block constants by a construction of its own (closed form about
    the junction centre / arcs turned about their centres); copo_amd/maps.py is what the HIP simulator AND, in every other test, the
    oracle run on.  Every field must agree to fp32 rounding of a 350 m coordinate -- a maps.py error would show here."""
    from copo_amd import maps
    t = maps.MAP_BUILDERS[name]()
    own = ol.own_map_tables(name)
    assert own["route_segs"].shape == t.route_segs.shape and own["route_meta"].shape == t.route_meta.shape
    d = np.abs(own["route_segs"].astype(np.float64) - t.route_segs)
    th = maps.SEG_TH0
    d[..., th] = np.abs((own["route_segs"][..., th].astype(np.float64) - t.route_segs[..., th] + np.pi) % (2 * np.pi) - np.pi)      # angles: modulo a turn
    assert d.max() < 2e-4, (np.unravel_index(d.argmax(), d.shape), d.max())
    assert np.abs(own["route_meta"] - t.route_meta).max() < 2e-4
    assert np.array_equal(own["spawn_tab"], t.spawn_tab) and np.abs(own["spawn_s"] - t.spawn_s).max() < 1e-6
    # ... and the simulator on either copy drives the same episodes: same flags, observations within the tables' rounding
    from copo_amd.sim import SimConfig
    cfg = SimConfig(map=name, num_envs=2, num_agents=12, horizon=60)
    a, b = ol.OracleSim(cfg), ol.OracleSim(cfg, own_tables=True)
    oa, ob = a.reset(), b.reset()
    rng = np.random.RandomState(1)
    for t_ in range(40):
        assert np.array_equal(oa["flags"], ob["flags"]) and np.abs(oa["obs"] - ob["obs"]).max() < 2e-3, t_
        act = np.stack([rng.normal(0, 0.05, (2, 12)), rng.uniform(0.2, 1, (2, 12))], -1).astype(np.float32)
        oa, ob = a.step(act), b.step(act)
    a.close()
    b.close()


def test_oracle_builds_and_versions():
    from copo_amd._abi import ABI_VERSION
    assert ol.lib().oracle_version() == ABI_VERSION == 8


def test_neighbours_and_rewards_vs_reference(golden_dir):
    """CCEnv._update_distance_map/_find_in_range + LCFEnv.step reward block (env_wrappers.py:125-158,313-326)."""
    g = np.load(os.path.join(golden_dir, "lcfenv_step.npz"))
    for c in range(int(g["n_cases"])):
        pos, present, rew = g["c%d_in_pos" % c], g["c%d_in_present" % c].astype(bool), g["c%d_in_rew" % c]
        N = len(pos)
        K = max(1, N - 1)
        o = ol.neighbours(pos[None], present[None], rew[None], K, float(g["c%d_in_radius" % c]), 10.0)
        Kg = g["c%d_out_nbr_idx" % c].shape[1]
        assert np.array_equal(o["nbr_cnt"][0], g["c%d_out_nbr_cnt" % c]), c
        assert np.array_equal(o["nbr_idx"][0][:, :Kg], g["c%d_out_nbr_idx" % c]), c
        np.testing.assert_allclose(o["nbr_dist"][0][:, :Kg], g["c%d_out_nbr_dist" % c], rtol=1e-6)
        np.testing.assert_allclose(o["nei_rew"][0][present], g["c%d_out_nei_r" % c][present], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(o["glob_rew"][0], g["c%d_out_glob_r" % c][present][0], rtol=1e-6, atol=1e-7)
        # mf prefix == number of listed neighbours with d <= 10 (algo_ccppo.py:283)
        d, cnt = g["c%d_out_nbr_dist" % c], g["c%d_out_nbr_cnt" % c]
        mf = ((d <= 10.0) & (np.arange(Kg)[None] < cnt[:, None])).sum(1)
        assert np.array_equal(o["mf_cnt"][0], mf), c


def test_known_answers_from_survey(golden_dir):
    pos = np.array([[[0, 0], [3, 4], [30, 0], [100, 0]]], np.float32)
    o = ol.neighbours(pos, np.ones((1, 4), bool), np.array([[1, 2, 3, 4]], np.float32), 3, 40.0, 10.0)
    assert o["nbr_idx"][0].tolist() == [[1, 2, -1], [0, 2, -1], [1, 0, -1], [-1, -1, -1]]
    np.testing.assert_allclose(o["nbr_dist"][0, 1, :2], [5.0, 27.294688], rtol=1e-6)
    assert o["nei_rew"][0].tolist() == [2.5, 2.0, 1.5, 0.0] and o["glob_rew"][0] == 2.5
    assert o["mf_cnt"][0].tolist() == [1, 1, 0, 0]


def test_coordinated_reward_and_obs_tail(golden_dir):
    """LCFEnv.step :350-353 and _add_lcf :417-418 -- formulas the build applies on the learner side."""
    g = np.load(os.path.join(golden_dir, "lcfenv_step.npz"))
    for c in range(int(g["n_cases"])):
        pres = g["c%d_in_present" % c].astype(bool)
        lcf, r, nr = g["c%d_out_lcf" % c][pres], g["c%d_in_rew" % c][pres], g["c%d_out_nei_r" % c][pres]
        assert np.all(np.abs(lcf) <= 1)
        np.testing.assert_allclose(np.cos(lcf * np.pi / 2) * r + np.sin(lcf * np.pi / 2) * nr,
                                   g["c%d_out_coord_r" % c][pres], rtol=1e-12, atol=1e-12)
        assert np.array_equal(g["c%d_out_obs_last" % c][pres], ((lcf + 1) / 2).astype(np.float32))
        assert np.all(g["c%d_out_obs_len" % c][pres] == 92) and np.all(g["c%d_out_obs_dtype_is_f32" % c][pres])


def _gae_inputs(g):
    lens, done_last = g["lens"], g["done_last"]
    Kt, TM = len(lens), int(lens.max())
    rew = np.zeros((3, TM, Kt), np.float32)
    val = np.zeros_like(rew)
    flags = np.zeros((TM, Kt), np.uint8)
    for k, T in enumerate(lens):
        flags[:T, k] = 1
        if done_last[k]:
            flags[T - 1, k] |= 2
        for h, (rk, vk) in enumerate([("r", "v"), ("nr", "nv"), ("gr", "gv")]):
            rew[h, :T, k], val[h, :T, k] = g[rk][k, :T], g[vk][k, :T]
    return rew, val, flags, lens


def test_gae3_vs_reference(golden_dir):
    """compute_nei_advantage / compute_global_advantage (algo_copo.py:189-204) + compute_advantages shim."""
    g = np.load(os.path.join(golden_dir, "gae.npz"))
    rew, val, flags, lens = _gae_inputs(g)
    adv, tgt = ol.gae3(rew, val, flags, [0.99, 0.99, 1.0], 0.95)
    for h, (ak, tk) in enumerate([("adv", "tgt"), ("nadv", "ntgt"), ("gadv", "gtgt")]):
        for k, T in enumerate(lens):
            assert np.array_equal(adv[h, :T, k], g[ak][k, :T]), (h, k)     # bit-exact incl. the fp32/fp64 dtype path
            assert np.array_equal(tgt[h, :T, k], g[tk][k, :T]), (h, k)
            assert np.all(adv[h, T:, k] == 0)
    a, t = ol.gae3(np.array([[[1.], [0.], [2.]]], np.float32), np.array([[[.1], [.2], [.3]]], np.float32),
                   np.array([[1], [1], [3]], np.uint8), [0.99], 0.95)
    np.testing.assert_array_equal(a[0, :, 0], g["ka_nei_adv"])
    np.testing.assert_array_equal(t[0, :, 0], g["ka_nei_tgt"])
    a, t = ol.gae3(np.ones((1, 3, 1), np.float32), np.array([[[.5], [.4], [.3]]], np.float32),
                   np.array([[1], [1], [1]], np.uint8), [1.0], 0.95)
    np.testing.assert_array_equal(a[0, :, 0], g["ka_glob_adv"])
    np.testing.assert_array_equal(t[0, :, 0], g["ka_glob_tgt"])


def test_gae3_segments_inside_a_column():
    """Several agents sharing a slot within one fragment: each trajectory bootstraps on its own last value."""
    rng = np.random.RandomState(0)
    T = 30
    r, v = rng.normal(0, 1, T).astype(np.float32), rng.normal(0, 3, T).astype(np.float32)
    flags = np.ones(T, np.uint8)
    flags[9] |= 2        # agent A: rows 0..9, done
    flags[10:13] = 0     # slot empty
    flags[20] |= 2       # agent B: rows 13..20 done ; agent C rows 21..29 truncated
    adv, tgt = ol.gae3(r[None, :, None], v[None, :, None], flags[:, None], [0.99], 0.95)
    for lo, hi, done in [(0, 10, True), (13, 21, True), (21, 30, False)]:
        vv = np.concatenate([v[lo:hi].astype(np.float64), [0.0 if done else float(v[hi - 1])]])
        delta = r[lo:hi] + 0.99 * vv[1:] - vv[:-1]
        acc, ref = 0.0, np.zeros(hi - lo)
        for t in range(hi - lo - 1, -1, -1):
            acc = delta[t] + 0.99 * 0.95 * acc
            ref[t] = acc
        np.testing.assert_allclose(adv[0, lo:hi, 0], ref, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tgt[0, lo:hi, 0], ref + v[lo:hi], rtol=1e-5, atol=1e-5)
    assert np.all(adv[0, 10:13, 0] == 0)


@pytest.mark.parametrize("policy,fuse", [("copo", "none"), ("copo", "mf"), ("copo", "concat"), ("ccppo", "mf"),
                                         ("ccppo", "concat")])
def test_cc_fuse_vs_reference(golden_dir, policy, fuse):
    """mean_field_ccppo_process / concat_ccppo_process (algo_ccppo.py:225-311)."""
    g = np.load(os.path.join(golden_dir, "postprocess_%s_%s.npz" % (policy, fuse)))
    obs, act, acted = g["in_obs"], g["in_act"], g["in_acted"]
    ref = g["out_cc_obs"]
    if fuse == "none":
        assert np.array_equal(ref[acted], obs[acted])
        return
    nbr_idx, nbr_cnt, nbr_dist = g["in_nbr_idx"], g["in_nbr_cnt"], g["in_nbr_dist"]
    K = nbr_idx.shape[-1]
    cnt = ((nbr_dist <= 10.0) & (np.arange(K)[None, None] < nbr_cnt[..., None])).sum(-1) if fuse == "mf" else nbr_cnt
    cc = ol.cc_fuse(fuse, obs, act, acted.astype(np.uint8), nbr_idx, cnt)
    assert cc.shape == ref.shape
    np.testing.assert_allclose(cc[acted], ref[acted], rtol=1e-6, atol=1e-7)
    assert np.array_equal(cc[acted], ref[acted])          # same fp32 accumulation order as np.mean over the list


def test_lcf_mix_vs_reference(golden_dir):
    """CoPOTrainer.training_step coordinated advantage + standardized() (algo_copo.py:539-551)."""
    g = np.load(os.path.join(golden_dir, "training_step.npz"))
    mixed, stats, norm, gstd = ol.lcf_mix(g["in_advantages"], g["in_nei_advantage"], g["in_global_advantages"],
                                          g["in_step_lcf"])
    np.testing.assert_allclose(mixed, g["out_raw_normalized_advantages"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(norm, g["out_normalized_advantages"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gstd, g["out_global_advantages"], rtol=1e-5, atol=1e-5)
    mean = stats[1] / stats[0]
    np.testing.assert_allclose([mean, np.sqrt(stats[2] / stats[0] - mean ** 2)], g["out_raw_mean_std"], rtol=1e-5)
    assert abs(norm.mean()) < 1e-6 and abs(norm.std() - 1) < 1e-5


def test_observation_extensions_vs_reference(golden_dir):
    """f-4: traffic-light message + communication channel, executed through the reference's LCFEnv on scripted
    scenes (reset + 9 steps, absent / freshly spawned agents, a coincident pair): env_wrappers.py:89-118, 258-303,
    315-337, 360-371.  Layout after the 91 base columns: [3 traffic light | lcf | comm_neighbours x comm_dim]."""
    g = np.load(os.path.join(golden_dir, "obs_extensions.npz"))
    nan_seen = False
    for c in range(int(g["n_cases"])):
        tl, cs, nb, pos_on, interval = (int(g["c%d_%s" % (c, k)]) for k in ("tl", "cs", "nb", "pos", "interval"))
        P, H, pres, acted, act, ext = (g["c%d_%s" % (c, k)] for k in ("positions", "heading_cs", "present", "acted", "act", "ext"))
        cd = cs + (3 if pos_on else 0)
        for t in range(P.shape[0]):
            otl, ocomm = ol.obs_extensions(P[t], H[t], pres[t], acted[t], act[t], 40.0, t, interval, g["bbox"], tl, cs, nb,
                                           pos_on, fresh=(t == 0))
            m = pres[t].astype(bool)
            col = 0
            if tl:
                assert np.array_equal(otl[m], ext[t][m, :3]), (c, t)
                col = 3
            assert np.all((ext[t][m, col] >= 0) & (ext[t][m, col] <= 1))       # the (lcf + 1) / 2 column
            if cs:
                ref = ext[t][m, col + 1:]
                got = ocomm[m]
                assert ref.shape == got.shape == (m.sum(), nb * cd)
                msg = np.concatenate([np.arange(r * cd, r * cd + cs) for r in range(nb)])
                assert np.array_equal(got[:, msg], ref[:, msg]), (c, t)       # messages: copied bits
                if pos_on:
                    rel = np.setdiff1d(np.arange(nb * cd), msg)
                    assert np.array_equal(np.isnan(got[:, rel]), np.isnan(ref[:, rel])), (c, t)
                    np.testing.assert_allclose(got[:, rel], ref[:, rel], rtol=2e-7, atol=1e-7, equal_nan=True)
                if t == 0:
                    assert not got.any()
        nan_seen = nan_seen or bool(np.isnan(ext).any())
    assert nan_seen       # the coincident pair (d == 0 -> 0/0) went through the reference at least once


def test_oracle_bottleneck_is_the_one_the_reference_population_was_trained_on(golden_dir):
    """Merge / Split blocks of the oracle's Bottleneck (maps.Net.add_funnel + Navigation's check-point rule): the CoPO population
    the reference ships for this scene (`eval/get_policy_function.py:29`: success 0.867 in MetaDrive) must get through the funnel.
    On round 2's corridor model half of its agents left the road (success 0.42)."""
    gold = np.load(os.path.join(golden_dir, "reference_populations_f4.npz"))
    pre = "copo_bottle/w/"
    w = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    n = _roll_population_in_the_oracle(w, 97, steps=700, E=3, map_name="bottleneck", agents=20, lcf=tuple(gold["copo_bottle/lcf"]))
    done = n["arrive"] + n["crash"] + n["out"]
    assert done > 120 and n["arrive"] / done > 0.6 and n["out"] / done < 0.08, n


def _roll_population_in_the_oracle(weights, odim, steps=350, E=3, seed=0, map_name="intersection", agents=30, lcf=None):
    """The oracle simulator (CPU) driven by a numpy policy in the reference's key layout; returns the counts of terminated
    agents by outcome.  Deterministic: counter-based simulator RNG + a seeded numpy generator for the action noise."""
    from copo_amd.eval.get_policy_function import detect_layout, layer_arrays
    from copo_amd.sim import SimConfig
    layers = layer_arrays(weights, detect_layout(weights), "default", "" if lcf is None else "_1")
    sim = ol.OracleSim(SimConfig(map=map_name, num_envs=E, num_agents=agents, enable_lcf=lcf is not None,
                                 lcf_mean=0.0 if lcf is None else float(lcf[0]), lcf_std=0.1 if lcf is None else float(lcf[1])))
    assert sim.O == odim == layers[0][0].shape[0]
    out = sim.reset(np.arange(E, dtype=np.uint64) + np.uint64(5000 + 1000 * seed))
    rng = np.random.RandomState(seed)
    n = dict(arrive=0, crash=0, out=0)
    for _ in range(steps):
        x = out["obs"].reshape(E * sim.N, odim).astype(np.float64)
        for depth, (w, b) in enumerate(layers):
            x = x @ w + b
            if depth < 2:
                x = np.tanh(x)
        act = x[:, :2] + np.exp(x[:, 2:]) * rng.normal(size=(E * sim.N, 2))
        out = sim.step(act.astype(np.float32).reshape(E, sim.N, 2))
        f = out["flags"]
        n["arrive"] += int(((f & 4) != 0).sum())
        n["crash"] += int((((f & 8) != 0) & ((f & 4) == 0)).sum())
        n["out"] += int((((f & 16) != 0) & ((f & 4) == 0)).sum())
    sim.close()
    return n


def test_oracle_simulator_is_pinned_by_the_reference_populations(golden_dir):
    """Simulator half of the oracle (MetaDrive's source is absent): the populations the reference trained
    in MetaDrive -- functions of the observation alone -- must DRIVE the oracle's scenes.  With build-defined observation
    semantics (round 1) they scored 0 %; with the wrong LiDAR beam order they crash within 20 steps.  350 steps x 3 scenes
    keep this in the CPU suite (seconds); the GPU suite repeats it on whole 1000-step episodes of 64 scenes with the
    reference's MetaDrive scores next to it (tests/test_gpu_reference_populations.py)."""
    gold = np.load(os.path.join(golden_dir, "eval_policy_function.npz"))
    pre = "ippo_inter/w/"
    w = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    n = _roll_population_in_the_oracle(w, 91)
    done = n["arrive"] + n["crash"] + n["out"]
    assert done > 150 and n["arrive"] / done > 0.3 and n["out"] / done < 0.15, n
    rng = np.random.RandomState(1)
    untrained = {k: (rng.normal(0, 0.05, v.shape) if v.ndim == 2 else np.zeros_like(v)) for k, v in w.items()}
    n0 = _roll_population_in_the_oracle(untrained, 91)
    assert n0["arrive"] == 0, n0

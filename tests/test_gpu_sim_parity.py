"""HIP simulator vs the CPU oracle, through the C ABI: raw-bit equality of every output over rollouts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT_KEYS = ("obs", "rew", "nei_rew", "glob_rew", "flags", "nbr_idx", "nbr_cnt", "mf_cnt", "nbr_dist", "lcf", "info",
            "agent_id")


def _bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def _compare(tag, g, o):
    for k in OUT_KEYS:
        gv = g[k].cpu().numpy()
        ov = o[k]
        if not np.array_equal(_bits(gv), _bits(ov)):
            bad = np.argwhere(_bits(gv) != _bits(ov))
            i = tuple(bad[0])
            raise AssertionError("%s: output %r differs at %s (%d mismatches): hip=%r oracle=%r" %
                                 (tag, k, i, len(bad), gv[i], ov[i]))


def _actions(rng, E, N, t):
    steer = rng.normal(0, 0.12, (E, N))
    thr = rng.uniform(-0.2, 1.0, (E, N))
    a = np.stack([steer, thr], -1).astype(np.float32)
    if t % 7 == 0:
        a[0, 0] = [np.nan, 5.0]     # NaN guard + clipping
    return a


@pytest.mark.parametrize("map_name,N,E,lasers,steps,block", [
    ("intersection", 40, 6, 72, 260, 1024),
    ("roundabout", 40, 3, 72, 200, 256),
    ("parkinglot", 10, 5, 240, 150, 512),
    ("tollgate", 40, 2, 72, 120, 64),
    ("intersection", 4, 3, 72, 150, 128),
    ("bottleneck", 20, 4, 72, 200, 256),
    ("pgmap", 20, 4, 72, 200, 256),
    ("pgmap-junctions", 20, 3, 72, 260, 256),      # intersection + roundabout blocks inside the generated road
    ("parkinglot-reverse", 10, 4, 72, 150, 64),    # reverse gear (copo_sim_cfg.reverse_acc): negative throttle = engine force backwards
    # the PACKED launch shape (sim_packed.hip; block = -scenes per workgroup): per-agent phases dense over the lanes, a last workgroup
    # with fewer scenes than the others, slots of a scene straddling two waves, exclusive destinations, detector beams, toll columns
    ("intersection", 40, 11, 72, 260, -8),
    ("roundabout", 40, 5, 72, 200, -4),
    ("parkinglot", 10, 13, 240, 150, -12),
    ("tollgate", 40, 3, 72, 120, -2),
    ("intersection", 4, 19, 72, 150, -16),
    ("bottleneck", 20, 7, 72, 200, -1),
    ("pgmap-junctions", 20, 9, 72, 260, -8),
    ("parkinglot-reverse", 10, 6, 72, 150, -5),
    ("intersection", 30, 9, 72, 200, -8),
    # detector beams: 8 LiDAR beams x 10 fans leave no room for a row of 76 beam minima in the LDS scratch -> the fallback that walks the
    # marked primitives beam by beam (detector_beams_walk), one wave per scene and several waves per scene; then the pair formulation
    # with several passes of few agents (30 beams x 3 fans = 90 words: one agent's row per pass)
    ("tollgate", 40, 3, 8, 100, 64),
    ("tollgate", 24, 2, 8, 100, 256),
    ("tollgate-chunk3", 40, 3, 30, 100, 64),
    # MultiAgentTollgateEnv's booth rules (ABI 8): speed limit + overspeed penalty on the booth road, an early exit = out-of-road flag with
    # the ordinary reward -- in the one-wave, the several-waves and the packed shape
    ("tollgate-mdrules", 40, 3, 72, 140, 64),
    ("tollgate-mdrules", 40, 2, 72, 140, 512),
    ("tollgate-mdrules", 24, 5, 72, 140, -4),
    ("tollgate-hidden", 40, 3, 72, 120, 64),       # static boxes the LiDAR does not see (toll_buildings 2)
    ("tollgate-hidden", 40, 2, 72, 120, 512),
])
def test_rollout_bit_exact(map_name, N, E, lasers, steps, block):
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim, TOLLGATE_METADRIVE_RULES
    kw = {"pgmap": dict(sequence="CSCCS", seed=11), "pgmap-junctions": dict(sequence="XOT", seed=2)}.get(map_name, {})
    cfg = SimConfig(map=map_name.split("-")[0], num_envs=E, num_agents=N, num_lasers=lasers, horizon=90, nbr_k=min(8, max(1, N - 1)),
                    delay_done=5, map_kwargs=kw, reverse_acc=2.9 if map_name.endswith("-reverse") else None,
                    **(TOLLGATE_METADRIVE_RULES if map_name.endswith("-mdrules") else (dict(toll_buildings=2) if map_name.endswith("-hidden") else {})))
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    g.set_block(block)
    if map_name.endswith("-chunk3"):
        g.set_chunk(3)
    seeds = np.arange(E, dtype=np.uint64) * np.uint64(7919) + np.uint64(5000)
    _compare("reset", g.reset(seeds), o.reset(seeds))
    rng = np.random.RandomState(3)
    resets = 0
    for t in range(steps):
        if t == 40:
            g.set_lcf_dist(0.4, 0.3)
            o.set_lcf_dist(0.4, 0.3)
        a = _actions(rng, E, N, t)
        go = g.step(torch.from_numpy(a).cuda())
        oo = o.step(a)
        _compare("%s step %d" % (map_name, t), go, oo)
        resets += int(((oo["flags"] & 128) > 0).any(1).sum())
    assert resets > 0 or steps < 150, "the rollout must run through the drain phase and a reset of some scene"
    gs, ge = g.get_state()
    os_, oe = o.get_state()
    assert np.array_equal(gs.cpu().numpy().view(np.uint32), os_.view(np.uint32))
    assert np.array_equal(ge.cpu().numpy()[:, :3], oe[:, :3])
    f = oo["flags"]
    g.close()
    o.close()


@pytest.mark.parametrize("map_name,block", [("intersection", 1024), ("roundabout", 64), ("intersection", -4)])
def test_hip_on_the_products_tables_against_the_oracle_on_its_own_tables(map_name, block):
    """Every other parity test feeds the oracle the product's map tables (copo_amd/maps.py), so a geometry error would move both sides
    together.  Here the oracle runs on ITS OWN derivation of the tables (oracle/oracle_maps.c, MetaDrive's block constants, another
    construction): the tables agree to fp32 rounding, not bit for bit, so the comparison is flags exactly + observations / rewards
    within that rounding over a stretch short enough that no decision sits on a rounding boundary."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N = 4, 30
    cfg = SimConfig(map=map_name, num_envs=E, num_agents=N, horizon=80, nbr_k=8)
    g, o = VecSim(cfg), ol.OracleSim(cfg, own_tables=True)
    g.set_block(block)
    go, oo = g.reset(), o.reset()
    rng = np.random.RandomState(5)
    for t in range(60):
        assert np.array_equal(go["flags"].cpu().numpy(), oo["flags"]), t
        pres = (oo["flags"] & 0x41) != 0
        assert np.abs(go["obs"].cpu().numpy()[pres] - oo["obs"][pres]).max() < 2e-3, t
        assert np.abs(go["rew"].cpu().numpy() - oo["rew"]).max() < 2e-3, t
        assert np.array_equal(go["nbr_cnt"].cpu().numpy(), oo["nbr_cnt"]), t
        a = np.stack([rng.normal(0, 0.05, (E, N)), rng.uniform(0.2, 1.0, (E, N))], -1).astype(np.float32)
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
    g.close()
    o.close()


def test_register_neighbour_lists_decline_on_ties_and_band_cases():
    """The register formulation of the neighbour lists (neighbours_fast, one wave per scene) must hand every agent whose
    result its registers do not PROVE to the exact evaluation (neighbours_exact_one, the reference's fp64 expressions for that
    agent), and a scene with many such agents to the pair-parallel formulation -- the result must be the oracle's either way.
    Crafted scenes through `copo_sim_set_state` (vehicles at rest, zero actions: the poses of the step are the crafted ones);
    debug column 7: 1 = registers only, 16 + n = n agents evaluated exactly, 2 = declined (pair-parallel formulation):
      scene 0  generic positions                                     -> 1
      scene 1  two agents at exactly the same distance of a third    -> 16 + n: tie, slot order decides
      scene 2  an agent exactly ON the 40 m radius (strict <)        -> 16 + n: inside the band of the in-range decision
      scene 3  an agent exactly ON the 10 m mean-field radius (<=)   -> 16 + n: inside the band of the mean-field decision
      scene 4  the reset state itself (vehicles on the spawn grid)   -> 16 + n: equal distances along the grid
      scene 5  as scene 0, one vehicle on a road's axis far away      -> 1: its jump in route progress is a reward > 16, but
                                                                        nobody has it in range
      scene 6  three such vehicles within range of each other         -> 16 + n: a reward outside the range in which sums are exact
      scene 7  nine vehicles on a cross (ties for every one of them)  -> 2: more agents than the exact evaluation takes
    Outputs (counts, first K ids and distances, neighbourhood / global rewards, observations) are compared bit for bit, in
    the launch shapes (one wave per scene; 16 waves per scene, where wave 1 builds the lists; packed, 8 and 3 scenes per workgroup)."""
    import torch
    import oracle_lib as ol
    from copo_amd import _capi
    from copo_amd.sim import SimConfig, VecSim
    E, N = 8, 12
    cfg = SimConfig(map="intersection", num_envs=E, num_agents=N, nbr_k=8, horizon=200)
    for block in (64, 1024, -8, -3):
        g, o = VecSim(cfg), ol.OracleSim(cfg)
        g.set_block(block)
        seeds = np.arange(E, dtype=np.uint64) + np.uint64(5000)
        _compare("reset", g.reset(seeds), o.reset(seeds))
        st, env = o.get_state()
        st0 = st.copy()
        st = st.copy()
        rng = np.random.RandomState(0)
        xy = np.zeros((E, N, 2), np.float32)
        xy[..., 0] = 1000.0 + 137.0 * np.arange(N)[None, :] + rng.uniform(-3, 3, (E, N))       # far apart: no neighbours by default
        xy[..., 1] = 700.0 + rng.uniform(-3, 3, (E, N))     # (far from every road: out of road, reward -10 -- on a road's axis the jump
        #                                                       in route progress would be a reward of +1000)
        xy[0, :6] = rng.uniform(-25, 25, (6, 2))                                                  # scene 0: a generic cluster
        xy[1, 0], xy[1, 1], xy[1, 2] = (0.0, 0.0), (30.0, 0.0), (-30.0, 0.0)                      # scene 1: tie at 30 m
        xy[2, 0], xy[2, 1], xy[2, 2] = (0.0, 0.0), (0.0, 40.0), (12.5, 3.0)                       # scene 2: on the radius
        xy[3, 0], xy[3, 1], xy[3, 2] = (0.0, 0.0), (6.0, 8.0), (-20.0, 7.0)                       # scene 3: on the mean-field radius (d = 10)
        xy[5] = xy[0]
        xy[5, 7] = (1500.0, 0.3)                                                                  # scene 5: on the axis of arm 0's road, 1.4 km out
        xy[6] = xy[0]
        xy[6, 7], xy[6, 8], xy[6, 9] = (1500.0, 0.3), (1512.0, 0.4), (1529.0, 0.2)                # scene 6: three of them, in range of each other
        xy[7, :9] = [(0, 0), (15, 0), (-15, 0), (0, 15), (0, -15), (30, 0), (-30, 0), (0, 30), (0, -30)]   # scene 7: a cross
        st[0], st[1] = xy[..., 0], xy[..., 1]
        st[:, 4] = st0[:, 4]                                                                      # scene 4: the reset state
        o.set_state(st, env)
        g.set_state(torch.from_numpy(st).cuda(), torch.from_numpy(env).cuda())
        dbg = torch.zeros(E, 8, dtype=torch.int64, device="cuda")
        _capi.check(_capi.lib.copo_sim_set_debug(g._h, dbg.data_ptr()))
        a = np.zeros((E, N, 2), np.float32)
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        torch.cuda.synchronize()
        _compare("crafted step (block %d)" % block, go, oo)
        which = dbg[:, 7].cpu().numpy().tolist()
        if block == 64:
            assert which[0] == 1 and which[5] == 1, which
            assert all(17 <= w <= 16 + 6 for w in which[1:4]), which
            assert 17 <= which[4] <= 16 + 6, which
            assert which[6] in (18, 19) and which[7] == 2, which
        elif block < 0:      # packed shape: no pair-parallel formulation, every undecided agent is evaluated exactly
            assert which[0] == 1 and which[5] == 1, which
            assert all(17 <= w <= 16 + 6 for w in which[1:4]), which
            assert which[4] >= 17 and which[6] in (18, 19) and which[7] >= 16 + 9, which
        else:      # (a scene whose last agent terminates in this step -- most of these -- resets, and a resetting scene of the
            #         many-wave shape builds its lists pair-parallel on the poses before the reset: column 7 stays 0)
            assert all(w == 0 or w == 2 or w == 1 or 17 <= w <= 22 for w in which) and any(w >= 17 for w in which), which
        assert float(np.abs(oo["rew"][6, 7:10]).max()) > 16.0              # (the premise of scene 6)
        cnt = oo["nbr_cnt"]
        assert cnt[1, 0] == 2 and list(oo["nbr_idx"][1, 0, :2]) == [1, 2]          # the tie: slot order
        assert cnt[2, 0] == 1 and oo["nbr_idx"][2, 0, 0] == 2                        # d == 40 is not a neighbour
        assert oo["mf_cnt"][3, 0] == 1                                               # d == 10 is inside the mean-field range
        _capi.check(_capi.lib.copo_sim_set_debug(g._h, None))
        g.close()
        o.close()


def test_episode_structure_bit_exact():
    """MultiAgentMetaDrive.step's episode: agents that drove `horizon` steps of their own end with max_step and linger
    as obstacles, a scene past `horizon` env steps stops respawning, and it is reset with its last agent.  Half of the
    agents stand still (-> max_step at their 25th step), the others drive."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N, H = 4, 24, 25
    cfg = SimConfig(map="intersection", num_envs=E, num_agents=N, horizon=H, delay_done=4, nbr_k=8)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    _compare("reset", g.reset(), o.reset())
    rng = np.random.RandomState(11)
    clock = np.zeros(E, np.int64)
    n_max, n_reset, n_spawn_late = 0, 0, 0
    for t in range(160):
        a = _actions(rng, E, N, t)
        a[:, ::2] = 0.0                                 # even slots: no steering, no throttle
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        _compare("step %d" % t, go, oo)
        f = oo["flags"]
        clock += 1
        ended = ((f & 128) > 0).any(1)
        ms = (f & 32) > 0
        assert np.all(oo["info"][..., 5][ms] == H)
        n_max += int(ms.sum())
        n_reset += int(ended.sum())
        n_spawn_late += int((((f & 64) > 0).any(1) & (clock >= H) & ~ended).sum())
        assert np.all(clock[ended] >= H)
        clock[ended] = 0
    assert n_max > 20 and n_reset >= 8 and n_spawn_late == 0
    g.close()
    o.close()


def test_full_k_and_ccenv_mode():
    """K = N-1 (complete lists) and enable_lcf=False (CCEnv-only obs of 91)."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    cfg = SimConfig(map="intersection", num_envs=2, num_agents=30, nbr_k=29, enable_lcf=False, horizon=60,
                    neighbours_distance=10.0)
    assert cfg.obs_dim == 91
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    _compare("reset", g.reset(), o.reset())
    rng = np.random.RandomState(5)
    for t in range(100):
        a = _actions(rng, 2, 30, t + 1)
        _compare("step %d" % t, g.step(torch.from_numpy(a).cuda()), o.step(a))


def test_state_roundtrip_and_determinism():
    import torch
    from copo_amd.sim import SimConfig, VecSim
    cfg = SimConfig(map="roundabout", num_envs=8, num_agents=40, horizon=50)
    g = VecSim(cfg)
    g.reset()
    rng = np.random.RandomState(1)
    acts = [torch.from_numpy(_actions(rng, 8, 40, t + 1)).cuda() for t in range(80)]
    for a in acts[:30]:
        g.step(a)
    st, env = g.get_state()
    ref = []
    for a in acts[30:]:
        out = g.step(a)
        ref.append({k: out[k].clone() for k in ("obs", "rew", "flags", "nbr_idx")})
    g.set_state(st, env)
    for a, r in zip(acts[30:], ref):
        out = g.step(a)
        present = (out["flags"] & 0x41) != 0          # rows of slots without an agent are not written (copo_step_out.obs)
        for k, v in r.items():
            if k == "obs":                     # rows of absent slots are not written (copo_step_out)
                assert torch.equal(out[k][present], v[present]), k
            elif k == "nbr_idx":               # ... and the lists are those of the scene BEFORE a horizon reset
                before = ((out["flags"] & 0x01) != 0) | (((out["flags"] & 0x40) != 0) & ((out["flags"] & 0x80) == 0))
                assert torch.equal(out[k][before], v[before]), k
            else:
                assert torch.equal(out[k], v), k


def test_error_codes():
    import ctypes as C
    import torch
    from copo_amd import _capi
    from copo_amd.sim import SimConfig, VecSim, fill_cfg_struct
    cfg = SimConfig(num_envs=2, num_agents=8)
    struct, keep = fill_cfg_struct(cfg, _capi.SimCfg)
    h = C.c_void_p()
    assert _capi.lib.copo_sim_create(None, 0, C.byref(h)) == -1
    struct.num_agents = 65
    assert _capi.lib.copo_sim_create(C.byref(struct), 0, C.byref(h)) == -2
    assert b"num_agents" in _capi.lib.copo_last_error()
    struct.num_agents = 8
    struct.obs_dim = 50
    assert _capi.lib.copo_sim_create(C.byref(struct), 0, C.byref(h)) == -2
    g = VecSim(cfg)
    with pytest.raises(_capi.CopoError) as ei:
        g.step(torch.zeros(2, 8, 2, device="cuda"))
    assert ei.value.code == -4          # step before reset
    with pytest.raises(_capi.CopoError):
        g.set_lcf_dist(2.0, 0.1)
    with pytest.raises(_capi.CopoError):
        g.set_lcf_dist(0.0, 0.0)


def test_population_capacity_bit_exact_and_respected():
    """Curriculum capacity (copo_sim_set_capacity; `ChangeNEnv.close_and_reset_num_agents`, env_wrappers.py:444-460):
    HIP == oracle bit for bit through capacity changes at a reset, mid-episode and across the horizon reset; slots
    beyond the capacity never spawn; vehicles already driving there finish their episode."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N = 4, 40
    cfg = SimConfig(map="intersection", num_envs=E, num_agents=N, horizon=70, nbr_k=8, delay_done=5)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    seeds = np.arange(E, dtype=np.uint64) + np.uint64(77)
    for s in (g, o):
        s.set_capacity(10)
    r = g.reset(seeds)
    _compare("reset cap 10", r, o.reset(seeds))
    fl = r["flags"].cpu().numpy()
    assert (fl[:, :10] & 64).all() and not fl[:, 10:].any()          # COPO_F_SPAWNED only below the capacity
    rng = np.random.RandomState(5)
    seen_hi_spawn = False
    for t in range(230):
        if t == 50:
            for s in (g, o):
                s.set_capacity(30)        # grows mid-episode: slots 10..29 fill through the respawn path
        if t == 120:
            for s in (g, o):
                s.set_capacity(20)        # shrinks: agents in slots 20..29 drive on, then the slots stay empty
        a = _actions(rng, E, N, t)
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        _compare("capacity step %d" % t, go, oo)
        fl = oo["flags"]
        cap = 10 if t < 50 else (30 if t < 120 else 20)
        assert not (fl[:, cap:] & 64).any(), t                        # nobody spawns beyond the capacity
        if 50 <= t < 120:
            seen_hi_spawn |= bool((fl[:, 10:30] & 64).any())
    assert seen_hi_spawn
    st, _ = g.get_state()
    status = st.cpu().numpy().view(np.int32)[12] & 0xff
    assert (status[:, 30:] == 0).all()
    g.close()
    o.close()


@pytest.mark.parametrize("tl,comm,pos,lcf,nb", [(True, 0, False, True, 4), (False, 4, False, True, 4), (True, 3, True, True, 2),
                                                (True, 4, True, False, 6)])
def test_observation_extensions_bit_exact(tl, comm, pos, lcf, nb):
    """f-4: traffic-light columns, communication channel (messages of the nearest neighbours, optional relative
    position) -- same bits as the oracle through horizon resets, respawns and set_lcf_dist."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N = 4, 24
    cfg = SimConfig(map="intersection", num_envs=E, num_agents=N, horizon=70, nbr_k=8, delay_done=5, enable_lcf=lcf,
                    add_traffic_light=tl, traffic_light_interval=7, comm_size=comm, comm_neighbours=nb, add_pos_in_comm=pos)
    assert cfg.obs_dim == 91 + (3 if tl else 0) + (1 if lcf else 0) + (nb * (comm + (3 if pos else 0)) if comm else 0)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    seeds = np.arange(E, dtype=np.uint64) * np.uint64(104729) + np.uint64(5000)
    _compare("reset", g.reset(seeds), o.reset(seeds))
    rng = np.random.RandomState(11)
    seen_msg = False
    for t in range(180):
        a = np.concatenate([_actions(rng, E, N, t), rng.uniform(-1, 1, (E, N, comm)).astype(np.float32)], -1)
        go = g.step(torch.from_numpy(a).cuda())
        oo = o.step(a)
        _compare("step %d" % t, go, oo)
        if comm:
            base = 91 + (3 if tl else 0) + (1 if lcf else 0)
            seen_msg = seen_msg or bool(np.abs(oo["obs"][..., base:]).max() > 0)
    assert seen_msg or not comm
    g.close()


@pytest.mark.parametrize("spread,lasers", [(3.0, 72), (8.0, 72), (20.0, 240), (45.0, 30), (2.0, 256)])
def test_lidar_windows_on_crafted_dense_scenes(spread, lasers):
    """The pair-driven LiDAR tests only the rays inside a conservative angular window of every vehicle.  Scenes the
    rollouts never produce -- vehicles piled on top of each other (origin inside another circumcircle: full window),
    grazing distances, every heading -- must still give the oracle's bits (its loop tests every ray against every box)."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N = 16, 40
    cfg = SimConfig(map="intersection", num_envs=E, num_agents=N, num_lasers=lasers, horizon=500, nbr_k=8)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    seeds = np.arange(E, dtype=np.uint64) + np.uint64(9)
    g.reset(seeds)
    o.reset(seeds)
    rng = np.random.RandomState(int(spread * 10) + lasers)
    st, env = o.get_state()
    st = st.copy()
    # fields 0..2 = x, y, heading: scatter every vehicle around a few cluster centres, some exactly coincident
    centres = rng.uniform(-30, 30, (E, 4, 2))
    which = rng.randint(0, 4, (E, N))
    xy = centres[np.arange(E)[:, None], which] + rng.normal(0, spread, (E, N, 2))
    xy[:, 1] = xy[:, 0]                                  # coincident pair
    xy[:, 2] = xy[:, 0] + [2.4320, 0.0]                  # about one circumradius away
    xy[:, 3] = xy[:, 0] + [0.0, 4.864]                   # about two
    st[0], st[1] = xy[..., 0].astype(np.float32), xy[..., 1].astype(np.float32)
    st[2] = rng.uniform(-np.pi, np.pi, (E, N)).astype(np.float32)
    o.set_state(st, env)
    g.set_state(torch.from_numpy(st).cuda(), torch.from_numpy(env).cuda())
    for t in range(3):
        a = np.zeros((E, N, 2), np.float32)
        a[..., 1] = -1.0                                   # brake: the piles stay piles
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        _compare("crafted step %d" % t, go, oo)
    hits = oo["obs"][..., 19:19 + lasers][(oo["flags"] & 0x41) != 0]
    assert (hits < 1.0).mean() > 0.05                               # plenty of returns to compare
    g.close()
    o.close()


def test_maximum_population_bit_exact():
    """64 slots per scene (COPO_MAX_AGENTS): every lane of wave 0 owns a vehicle, 4096 LiDAR pairs, full-width masks."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    E, N = 3, 64
    cfg = SimConfig(map="intersection", map_kwargs=dict(exit_length=100.0), num_envs=E, num_agents=N, horizon=60,
                    nbr_k=16, delay_done=4)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    seeds = np.arange(E, dtype=np.uint64) + np.uint64(31)
    _compare("reset", g.reset(seeds), o.reset(seeds))
    rng = np.random.RandomState(8)
    most = 0
    for t in range(150):
        a = _actions(rng, E, N, t + 1)
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        _compare("N=64 step %d" % t, go, oo)
        most = max(most, int(((oo["flags"] & 0x41) != 0).sum(1).max()))
    assert most == 64
    g.close()
    o.close()


@pytest.mark.parametrize("name,map_name,N,E,lasers,steps", [
    ("C2", "intersection", 40, 256, 72, 120),        # BASELINE configs[1], the bench workload
    ("C3", "roundabout", 40, 1024, 72, 60),          # configs[2], all 1024 scenes on one GPU
    ("C4", "tollgate", 40, 512, 72, 60),             # configs[3]: O = 156 (72 side-detector beams, toll columns, no navigation)
    ("bottleneck", "bottleneck", 20, 512, 72, 60),   # O = 96 (4 + 4 detector beams), Merge / Split funnels
    ("C5", "parkinglot", 10, 4096, 240, 60),         # configs[4], 240 beams
    ("saturated", "intersection", 40, 16384, 72, 30),   # the scene count of the saturated roofline figure
])
def test_full_size_configs_on_sampled_scenes(name, map_name, N, E, lasers, steps):
    """BASELINE.json's full sizes: scenes never interact and a scene's random streams hang on its seed alone, so the oracle
    steps a SAMPLE of the scenes (first, last, a few in between) with the same seeds and actions and must reproduce those
    scenes of the full-size HIP rollout bit for bit; over all scenes, the outputs must not depend on the workgroup size."""
    import torch
    import oracle_lib as ol
    from copo_amd.sim import SimConfig, VecSim
    kw = dict(map=map_name, num_agents=N, num_lasers=lasers, horizon=40, delay_done=5)
    pick = np.unique(np.r_[0, E - 1, np.random.RandomState(E).randint(0, E, 4)])
    g, g2, o = VecSim(SimConfig(num_envs=E, **kw)), VecSim(SimConfig(num_envs=E, **kw)), ol.OracleSim(SimConfig(num_envs=len(pick), **kw))
    g2.set_block(256 if E <= 256 else 1024)          # the other launch shape than the default for this scene count
    seeds = np.arange(E, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(77)

    def sub(out):
        return {k: v[torch.from_numpy(pick).to(v.device)] for k, v in out.items() if k in OUT_KEYS}

    go, go2 = g.reset(seeds), g2.reset(seeds)
    _compare("%s reset" % name, sub(go), o.reset(seeds[pick]))
    gen = torch.Generator(device="cuda").manual_seed(E)
    for t in range(steps):
        a = torch.stack([torch.randn(E, N, device="cuda", generator=gen) * 0.12,
                         torch.rand(E, N, device="cuda", generator=gen) * 1.2 - 0.2], -1).contiguous()
        go, go2 = g.step(a), g2.step(a)
        _compare("%s step %d" % (name, t), sub(go), o.step(a[torch.from_numpy(pick).cuda()].cpu().numpy()))
        for k in OUT_KEYS:
            present = ((go["flags"] & 0x41) != 0)
            x, y = go[k], go2[k]
            if k == "obs":                     # rows of absent slots are not written (copo_hip.h)
                x, y = x[present], y[present]
            elif k in ("nbr_idx", "nbr_dist"):  # ... the neighbour lists are those of the scene before a horizon reset
                before = ((go["flags"] & 0x01) != 0) | (((go["flags"] & 0x40) != 0) & ((go["flags"] & 0x80) == 0))
                x, y = x[before], y[before]
            assert torch.equal(x.view(torch.uint8) if x.dtype != torch.float32 else x.view(torch.int32),
                               y.view(torch.uint8) if y.dtype != torch.float32 else y.view(torch.int32)), (name, t, k)
    for s in (g, g2, o):
        s.close()

"""HIP learn-side ops and the stateless neighbour op vs the CPU oracle and the golden vectors (C ABI calls)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hip_neighbours(pos, present, rew, K, radius, mf):
    import torch
    from copo_amd import _capi
    E, N = pos.shape[:2]
    d = dict(pos=_t(pos.astype(np.float32)), present=_t(present.astype(np.uint8)),
             rew=None if rew is None else _t(rew.astype(np.float32)))
    o = dict(nbr_idx=torch.zeros(E, N, K, dtype=torch.int32).cuda(), nbr_cnt=torch.zeros(E, N, dtype=torch.int32).cuda(),
             mf_cnt=torch.zeros(E, N, dtype=torch.int32).cuda(), nbr_dist=torch.zeros(E, N, K).cuda(),
             nei_rew=torch.zeros(E, N).cuda(), glob_rew=torch.zeros(E).cuda())
    _capi.check(_capi.lib.copo_neighbours_f32(
        _capi.ptr(d["pos"]), _capi.ptr(d["present"]), _capi.ptr(d["rew"]), E, N, K, radius, mf, _capi.ptr(o["nbr_idx"]),
        _capi.ptr(o["nbr_cnt"]), _capi.ptr(o["mf_cnt"]), _capi.ptr(o["nbr_dist"]), _capi.ptr(o["nei_rew"]),
        _capi.ptr(o["glob_rew"]), _capi.current_stream()))
    return {k: v.cpu().numpy() for k, v in o.items()}


def test_neighbours_golden(golden_dir):
    """Bit-exact neighbour indexing against the reference's CCEnv/LCFEnv (env_wrappers.py:125-158,313-326)."""
    g = np.load(os.path.join(golden_dir, "lcfenv_step.npz"))
    for c in range(int(g["n_cases"])):
        pos, present, rew = g["c%d_in_pos" % c], g["c%d_in_present" % c], g["c%d_in_rew" % c]
        N = len(pos)
        if N < 2:
            continue
        K = N - 1
        o = hip_neighbours(pos[None], present[None], rew[None], K, float(g["c%d_in_radius" % c]), 10.0)
        assert np.array_equal(o["nbr_cnt"][0], g["c%d_out_nbr_cnt" % c]), c
        assert np.array_equal(o["nbr_idx"][0], g["c%d_out_nbr_idx" % c]), c
        np.testing.assert_allclose(o["nbr_dist"][0], g["c%d_out_nbr_dist" % c], rtol=1e-6, atol=0)
        pres = present.astype(bool)
        np.testing.assert_allclose(o["nei_rew"][0][pres], g["c%d_out_nei_r" % c][pres], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(o["glob_rew"][0], g["c%d_out_glob_r" % c][pres][0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("E,N,K", [(64, 40, 8), (3, 64, 63), (17, 10, 9), (5, 2, 1)])
def test_neighbours_vs_oracle(E, N, K):
    import oracle_lib as ol
    rng = np.random.RandomState(E * 100 + N)
    pos = np.round(rng.uniform(-60, 60, (E, N, 2)) * 4) / 4          # quarter-metre grid -> many exact ties
    present = rng.uniform(size=(E, N)) > 0.2
    rew = rng.normal(0, 1, (E, N)).astype(np.float32)
    a = hip_neighbours(pos, present, rew, K, 40.0, 10.0)
    b = ol.neighbours(pos, present, rew, K, 40.0, 10.0)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                              b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k


def test_neighbours_on_the_radius():
    """Pairs exactly ON the neighbourhood radius (outside: the test is d < R) and on the mean-field radius (inside: d <= M),
    pairs one fp32 ulp either side, exact duplicates of a distance with different slots: the fp32 pre-test of the kernel
    admits all of them as candidates, the fp64 pass must sort them out as the oracle does."""
    import oracle_lib as ol
    eps = np.float32(40.0) - np.nextafter(np.float32(40.0), np.float32(0.0))
    pts = [(0, 0), (40, 0), (24, 32), (-32, 24), (0, -40), (40 - eps, 0), (40 + eps, 0), (0, 10), (6, 8), (-10, 0),
           (0, np.nextafter(np.float32(10.0), np.float32(20.0))), (39.75, 0), (0, 39.75), (28, 28), (28.25, 28.25)]
    pos = np.zeros((2, 16, 2))
    pos[0, :15] = pts
    pos[1, :15] = np.array(pts)[::-1] + 3.0
    pos[:, 15] = (500.0, 500.0)
    present = np.ones((2, 16), bool)
    rew = np.random.RandomState(0).normal(0, 1, (2, 16)).astype(np.float32)
    a = hip_neighbours(pos, present, rew, 15, 40.0, 10.0)
    b = ol.neighbours(pos, present, rew, 15, 40.0, 10.0)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                              b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k
    # of slot 0: (40-eps,0) (0,10) (6,8) (-10,0) (0,10+ulp) (39.75,0) (0,39.75) (28,28) (28.25,28.25); mean field: (0,10) (6,8) (-10,0)
    assert a["nbr_cnt"][0, 0] == 9 and a["mf_cnt"][0, 0] == 3


def hip_gae3(rew, val, flags, gamma, lam):
    import torch
    from copo_amd import _capi
    H, T, M = rew.shape
    r, v, f = _t(rew.astype(np.float32)), _t(val.astype(np.float32)), _t(flags.astype(np.uint8))
    adv, tgt = torch.zeros_like(r), torch.zeros_like(r)
    g = (C.c_double * H)(*[float(x) for x in gamma])
    _capi.check(_capi.lib.copo_gae3_f32(r.data_ptr(), v.data_ptr(), f.data_ptr(), T, M, H, g, float(lam),
                                        adv.data_ptr(), tgt.data_ptr(), _capi.current_stream()))
    return adv.cpu().numpy(), tgt.cpu().numpy()


def test_gae3_golden(golden_dir):
    """compute_advantages / compute_nei_advantage / compute_global_advantage (algo_copo.py:189-204)."""
    g = np.load(os.path.join(golden_dir, "gae.npz"))
    lens, done_last = g["lens"], g["done_last"]
    Kt, TM = len(lens), int(lens.max())
    rew = np.zeros((3, TM, Kt), np.float32)
    val = np.zeros_like(rew)
    flags = np.zeros((TM, Kt), np.uint8)
    for k, T in enumerate(lens):   # right-align each trajectory so that it ends at the window edge or with done
        flags[:T, k] = 1
        if done_last[k]:
            flags[T - 1, k] |= 2
        for h, (rk, vk) in enumerate([("r", "v"), ("nr", "nv"), ("gr", "gv")]):
            rew[h, :T, k], val[h, :T, k] = g[rk][k, :T], g[vk][k, :T]
    adv, tgt = hip_gae3(rew, val, flags, [0.99, 0.99, 1.0], 0.95)
    for h, (ak, tk) in enumerate([("adv", "tgt"), ("nadv", "ntgt"), ("gadv", "gtgt")]):
        for k, T in enumerate(lens):
            np.testing.assert_allclose(adv[h, :T, k], g[ak][k, :T], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(tgt[h, :T, k], g[tk][k, :T], rtol=1e-6, atol=1e-6)
            assert np.array_equal(adv[h, :T, k], g[ak][k, :T]), (h, k)   # in fact bit-exact


# (1, 40960): BASELINE configs[4] (ParkingLot, 10 slots x 4096 scenes, T = 1); (2, 20480): configs[3] (Tollgate, 40 x 512, T = 2);
# (16, 5120): one shard of configs[2] (Roundabout, 40 x 128, T = 16)
@pytest.mark.parametrize("T,M", [(8, 10240), (200, 517), (1, 64), (33, 1), (1, 40960), (2, 20480), (16, 5120)])
def test_gae3_vs_oracle(T, M):
    import oracle_lib as ol
    rng = np.random.RandomState(T + M)
    rew = rng.normal(0, 1, (3, T, M)).astype(np.float32)
    val = rng.normal(0, 4, (3, T, M)).astype(np.float32)
    acted = rng.uniform(size=(T, M)) > 0.1
    done = acted & (rng.uniform(size=(T, M)) < 0.05)
    flags = (acted * 1 + done * 2).astype(np.uint8)
    a = hip_gae3(rew, val, flags, [0.99, 0.99, 1.0], 0.95)
    b = ol.gae3(rew, val, flags, [0.99, 0.99, 1.0], 0.95)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def hip_cc_fuse(mode, obs, act, flags, nbr_idx, cnt, cf=True, nn=4):
    import torch
    from copo_amd import _capi
    R, N, O = obs.shape
    A, K = act.shape[-1], nbr_idx.shape[-1]
    o, a, f, ni, c = _t(obs.astype(np.float32)), _t(act.astype(np.float32)), _t(flags.astype(np.uint8)), \
        _t(nbr_idx.astype(np.int32)), _t(cnt.astype(np.int32))
    if mode == "mf":
        cc = torch.full((R, N, 2 * O + (A if cf else 0)), 7.0).cuda()
        _capi.check(_capi.lib.copo_cc_fuse_mf_f32(o.data_ptr(), a.data_ptr(), f.data_ptr(), ni.data_ptr(), c.data_ptr(),
                                                  R, N, O, A, K, int(cf), cc.data_ptr(), _capi.current_stream()))
    else:
        cc = torch.full((R, N, O + nn * (O + (A if cf else 0))), 7.0).cuda()
        _capi.check(_capi.lib.copo_cc_fuse_concat_f32(o.data_ptr(), a.data_ptr(), f.data_ptr(), ni.data_ptr(),
                                                      c.data_ptr(), R, N, O, A, K, nn, int(cf), cc.data_ptr(),
                                                      _capi.current_stream()))
    return cc.cpu().numpy()


@pytest.mark.parametrize("policy,fuse", [("copo", "mf"), ("copo", "concat"), ("ccppo", "mf"), ("ccppo", "concat")])
def test_cc_fuse_golden(golden_dir, policy, fuse):
    """mean_field_ccppo_process / concat_ccppo_process (algo_ccppo.py:225-311) incl. absent-neighbour slots."""
    g = np.load(os.path.join(golden_dir, "postprocess_%s_%s.npz" % (policy, fuse)))
    obs, act, acted = g["in_obs"], g["in_act"], g["in_acted"]
    flags = acted.astype(np.uint8)
    nbr_idx, nbr_cnt, nbr_dist = g["in_nbr_idx"], g["in_nbr_cnt"], g["in_nbr_dist"]
    if fuse == "mf":
        K = nbr_idx.shape[-1]
        cnt = ((nbr_dist <= 10.0) & (np.arange(K)[None, None] < nbr_cnt[..., None])).sum(-1)
    else:
        cnt = nbr_cnt
    cc = hip_cc_fuse(fuse, obs, act, flags, nbr_idx, cnt)
    ref = g["out_cc_obs"]
    assert cc.shape == ref.shape
    np.testing.assert_allclose(cc[acted], ref[acted], rtol=1e-6, atol=1e-7)
    assert np.all(cc[~acted] == 0)


@pytest.mark.parametrize("mode,cf", [("mf", True), ("mf", False), ("concat", True), ("concat", False)])
def test_cc_fuse_vs_oracle(mode, cf):
    import oracle_lib as ol
    rng = np.random.RandomState(11)
    R, N, O, A, K = 37, 40, 92, 2, 8
    obs = rng.uniform(-1, 1, (R, N, O)).astype(np.float32)
    act = rng.normal(0, 1, (R, N, A)).astype(np.float32)
    flags = (rng.uniform(size=(R, N)) > 0.15).astype(np.uint8)
    cnt = rng.randint(0, 12, (R, N))
    nbr_idx = np.full((R, N, K), -1, np.int32)
    for r in range(R):
        for n in range(N):
            m = min(cnt[r, n], K)
            nbr_idx[r, n, :m] = rng.choice([j for j in range(N) if j != n], m, replace=False)
    a = hip_cc_fuse(mode, obs, act, flags, nbr_idx, cnt, cf)
    b = ol.cc_fuse(mode, obs, act, flags, nbr_idx, cnt, cf)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("mode,cf", [("mf", True), ("concat", True)])
def test_cc_fuse_at_the_bench_batch(mode, cf):
    """BASELINE configs[1]: one rollout fragment of the bench workload, 8 steps x 256 scenes x 40 slots = 81,920 rows."""
    import oracle_lib as ol
    rng = np.random.RandomState(12)
    R, N, O, A, K = 8 * 256, 40, 92, 2, 8
    obs = rng.uniform(-1, 1, (R, N, O)).astype(np.float32)
    act = rng.normal(0, 1, (R, N, A)).astype(np.float32)
    flags = (rng.uniform(size=(R, N)) > 0.4).astype(np.uint8)
    cnt = rng.randint(0, 12, (R, N))
    others = np.argsort(rng.uniform(size=(R, N, N - 1)), axis=-1)[..., :K].astype(np.int32)     # K distinct others per row
    others += (others >= np.arange(N)[None, :, None])                                           # skip the row's own slot
    nbr_idx = np.where(np.arange(K)[None, None, :] < np.minimum(cnt, K)[..., None], others, -1).astype(np.int32)
    a = hip_cc_fuse(mode, obs, act, flags, nbr_idx, cnt, cf)
    b = ol.cc_fuse(mode, obs, act, flags, nbr_idx, cnt, cf)
    assert a.shape[0] * a.shape[1] == 81920 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name,R,N,O,mode", [
    ("configs[3] CCPPO mean-field Tollgate: 2 steps x 512 scenes x 40 slots, O = 156, cc 314", 2 * 512, 40, 156, "mf"),
    ("configs[3] shape with the concat critic", 64, 40, 156, "concat"),
    ("configs[4] shape: ParkingLot 10 slots, 240 beams, O = 260 (4096 scenes x T = 1)", 4096, 10, 260, "mf"),
    ("configs[2] shard: Roundabout 16 steps x 128 scenes x 40 slots, O = 92", 16 * 128, 40, 92, "mf")])
def test_cc_fuse_at_the_other_baseline_shapes(name, R, N, O, mode):
    """The neighbourhood fusion against the oracle, bit for bit, at the observation widths / row counts of the BASELINE
    configurations other than the bench one (algo_ccppo.py:225-311; cc width 2 * O + 2 for mean-field)."""
    import oracle_lib as ol
    rng = np.random.RandomState(O + N)
    A, K = 2, min(8, N - 1)
    obs = rng.uniform(-1, 1, (R, N, O)).astype(np.float32)
    act = rng.normal(0, 1, (R, N, A)).astype(np.float32)
    flags = (rng.uniform(size=(R, N)) > 0.3).astype(np.uint8)
    cnt = rng.randint(0, K + 3, (R, N))
    others = np.argsort(rng.uniform(size=(R, N, N - 1)), axis=-1)[..., :K].astype(np.int32)
    others += (others >= np.arange(N)[None, :, None])
    nbr_idx = np.where(np.arange(K)[None, None, :] < np.minimum(cnt, K)[..., None], others, -1).astype(np.int32)
    a = hip_cc_fuse(mode, obs, act, flags, nbr_idx, cnt, True)
    b = ol.cc_fuse(mode, obs, act, flags, nbr_idx, cnt, True)
    if mode == "mf":
        assert a.shape[-1] == 2 * O + 2, a.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


def hip_lcf_mix(adv, nei, glob, lcf, valid=None):
    import torch
    from copo_amd import _capi
    B = adv.size
    a, n, g, l = (_t(x.astype(np.float32)) for x in (adv, nei, glob, lcf))
    v = None if valid is None else _t(valid.astype(np.uint8))
    mixed, norm, gstd = torch.zeros(B).cuda(), torch.zeros(B).cuda(), torch.zeros(B).cuda()
    stats = torch.zeros(_capi.LCF_STATS_DOUBLES, dtype=torch.float64).cuda()
    s = _capi.current_stream()
    _capi.check(_capi.lib.copo_lcf_mix_partial_f32(a.data_ptr(), n.data_ptr(), g.data_ptr(), l.data_ptr(), _capi.ptr(v),
                                                   B, mixed.data_ptr(), stats.data_ptr(), s))
    _capi.check(_capi.lib.copo_lcf_mix_apply_f32(mixed.data_ptr(), g.data_ptr(), _capi.ptr(v), B, stats.data_ptr(),
                                                 norm.data_ptr(), gstd.data_ptr(), s))
    return mixed.cpu().numpy(), stats[:6].cpu().numpy(), norm.cpu().numpy(), gstd.cpu().numpy()


def test_lcf_mix_golden(golden_dir):
    """CoPOTrainer.training_step coordinated-advantage block (algo_copo.py:539-551)."""
    g = np.load(os.path.join(golden_dir, "training_step.npz"))
    mixed, stats, norm, gstd = hip_lcf_mix(g["in_advantages"], g["in_nei_advantage"], g["in_global_advantages"],
                                           g["in_step_lcf"])
    np.testing.assert_allclose(mixed, g["out_raw_normalized_advantages"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(norm, g["out_normalized_advantages"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gstd, g["out_global_advantages"], rtol=1e-5, atol=1e-5)
    mean = stats[1] / stats[0]
    std = np.sqrt(stats[2] / stats[0] - mean * mean)
    np.testing.assert_allclose([mean, std], g["out_raw_mean_std"], rtol=1e-5)


@pytest.mark.parametrize("B,masked", [(81920, True), (1200, False), (1, False), (2_000_003, True)])
def test_lcf_mix_vs_oracle(B, masked):
    import oracle_lib as ol
    rng = np.random.RandomState(B % 1000)
    adv, nei, glob = (rng.normal(0, 2, B).astype(np.float32) for _ in range(3))
    lcf = np.clip(rng.normal(0.2, 0.4, B), -1, 1).astype(np.float32)
    valid = (rng.uniform(size=B) > 0.1) if masked else None
    if masked:
        valid[0] = True
    a = hip_lcf_mix(adv, nei, glob, lcf, valid)
    b = ol.lcf_mix(adv, nei, glob, lcf, valid)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))      # mixed advantage: bit-exact
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12)                       # fp64 sums, different order
    np.testing.assert_allclose(a[2], b[2], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-6, atol=1e-6)
    a2 = hip_lcf_mix(adv, nei, glob, lcf, valid)
    assert all(np.array_equal(x, y) for x, y in zip(a, a2))                  # run-to-run deterministic


def test_row_movers_equal_the_tensor_code():
    """`copo_gather_rows_f32` (several sources, one launch; float4 and scalar rows) and `copo_pack_columns_f32` against
    torch.index_select / column copies, bit for bit."""
    import ctypes as C
    import torch
    from copo_amd import _capi
    g = torch.Generator(device="cuda").manual_seed(3)
    R, n = 5000, 7001
    srcs = [torch.randn(R, w, device="cuda", generator=g) for w in (92, 17, 91)]
    rows = torch.randint(0, R, (n,), device="cuda", generator=g)
    dsts = [torch.zeros(n, s.shape[1], device="cuda") for s in srcs]
    _capi.check(_capi.lib.copo_gather_rows_f32((C.c_void_p * 3)(*[s.data_ptr() for s in srcs]), (C.c_void_p * 3)(*[d.data_ptr() for d in dsts]),
                                               (C.c_int32 * 3)(92, 17, 91), 3, rows.data_ptr(), n, _capi.current_stream()))
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s.index_select(0, rows))
    widths = [2, 1, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    cols = [torch.randn(R, w, device="cuda", generator=g) for w in widths]
    pack = torch.zeros(R, sum(widths), device="cuda")
    _capi.check(_capi.lib.copo_pack_columns_f32((C.c_void_p * len(cols))(*[c.data_ptr() for c in cols]), (C.c_int32 * len(cols))(*widths),
                                                len(cols), R, pack.data_ptr(), _capi.current_stream()))
    assert torch.equal(pack, torch.cat(cols, 1))

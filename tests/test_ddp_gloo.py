"""CPU, world_size 2 over gloo: the data-parallel exchange steps of the learner (copo_amd/dist.py).

Two ranks hold different rows.  After PPO minibatch steps and LCF meta steps both ranks must hold identical
parameters, and those must equal a single-process replay on the UNION of the two ranks' minibatches:
  * gradients are summed over ranks with the global row count as denominator (== mean over the union),
  * both meta gradients are all-reduced BEFORE their dot product (the product of sums, not the sum of products),
  * advantage statistics are global.
"""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_PER_RANK = (300, 212)       # unequal shards: the smaller rank pads with zero-weight rows
MB = 128


def _policy():
    from copo_amd.engine import Box
    from copo_amd.torch_copo import algo_copo as A
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    cfg = A.CoPOConfig()
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    cfg.update_from_dict(dict(env=env, device="cpu", use_hip_graphs=False, seed=5, sgd_minibatch_size=MB,
                              model={"fcnet_hiddens": [32, 32]}))
    cfg.validate()
    pol = A.CoPOPolicy(Box(-1, 1, (12,)), Box(-1, 1, (2,)), cfg)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for p in list(pol.model.parameters()) + list(pol.target_model.parameters()):
            if p.dtype == torch.float32:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    return pol


def _batch(rank, R):
    from copo_amd.engine import Postprocessing, SampleBatch, TorchDiagGaussian
    g = torch.Generator().manual_seed(100 + rank)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    b = SampleBatch()
    b[SampleBatch.OBS] = torch.rand(R, 12, generator=g) * 2 - 1
    b[SampleBatch.ACTIONS] = rn(R, 2) * 0.8
    di = torch.cat([rn(R, 2) * 0.3, rn(R, 2) * 0.2 - 0.2], 1)
    b[SampleBatch.ACTION_DIST_INPUTS] = di
    b[SampleBatch.ACTION_LOGP] = TorchDiagGaussian(di).logp(b[SampleBatch.ACTIONS])
    for k, s in [(Postprocessing.ADVANTAGES, 2), (SampleBatch.VF_PREDS, 3), ("nei_advantage", 2), ("global_advantages", 1),
                 ("nei_values", 3), ("global_values", 30), ("normalized_advantages", 1)]:
        b[k] = rn(R) * s
    b[Postprocessing.VALUE_TARGETS] = b[SampleBatch.VF_PREDS] + rn(R) * 2
    b["nei_target"] = b["nei_values"] + rn(R) * 2
    b["global_target"] = b["global_values"] + rn(R) * 20
    b[SampleBatch.FLAGS] = torch.ones(R, dtype=torch.uint8)
    return b


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from copo_amd import dist as D
    D.init_from_env("cpu")
    assert D.world_size() == world
    pol = _policy()
    R = R_PER_RANK[rank]
    batch = _batch(rank, R)
    pol.prepare_sgd(batch, R, MB)
    idx = torch.arange(R)
    B_all = D.all_gather_int(R, "cpu")
    assert B_all == list(R_PER_RANK)
    # global advantage statistics (VecTrainer.standardize_advantages uses the same collective)
    stats = torch.tensor([float(R), float(batch["advantages"].sum()), float((batch["advantages"] ** 2).sum())],
                         dtype=torch.float64)
    D.all_reduce_sum_(stats)
    pol._raw_lcf_adv_mean.fill_(0.25)
    pol._raw_lcf_adv_std.fill_(1.5)
    torch.manual_seed(1234 + rank)           # different shuffles per rank, like real shards
    sgd = pol.run_sgd(idx, R, B_all, MB, 1)
    rows_sgd = pol._row_sources["rows_all"].clone()
    w_sgd = pol._row_sources["w_all"].clone()
    den_sgd = pol._row_sources["denom_all"].clone()
    meta = pol.run_meta(idx, R, B_all, MB, 1)
    mbuf = pol._meta_bufs
    torch.save(dict(
        params={k: v.clone() for k, v in pol.model.state_dict().items()}, sgd=sgd, meta=meta, stats=stats,
        rows_sgd=rows_sgd, w_sgd=w_sgd, den_sgd=den_sgd, rows_meta=mbuf["rows_all"].clone(), w_meta=mbuf["w_all"].clone(),
        den_meta=mbuf["denom_all"].clone(), eps_meta=mbuf["eps_all"].clone()), os.path.join(out_dir, "rank%d.pt" % rank))
    D.barrier()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_learner_equals_union_replay():
    from copo_amd.engine import SampleBatch
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r = [torch.load(os.path.join(d, "rank%d.pt" % k), weights_only=False) for k in range(2)]
    # 1. ranks agree bit for bit
    for k in r[0]["params"]:
        assert torch.equal(r[0]["params"][k], r[1]["params"][k]), k
    assert torch.equal(r[0]["stats"], r[1]["stats"]) and r[0]["stats"][0] == sum(R_PER_RANK)
    assert r[0]["sgd"]["num_sgd_steps"] == r[1]["sgd"]["num_sgd_steps"] == 3        # ceil(300 / 128)
    # 2. single-process replay on the union of the two ranks' minibatches
    pol = _policy()
    pol._raw_lcf_adv_mean.fill_(0.25)
    pol._raw_lcf_adv_std.fill_(1.5)
    batches = [_batch(k, R_PER_RANK[k]) for k in range(2)]
    cols = [n for n, _ in pol.train_columns()] + [SampleBatch.OBS]

    def union(rows_key, w_key, step):
        tb = SampleBatch()
        parts = {c: [] for c in cols}
        ws = []
        for k in range(2):
            rows = r[k][rows_key][step]
            for c in cols:
                parts[c].append(batches[k][c][rows])
            ws.append(r[k][w_key][step])
        for c in cols:
            tb[c] = torch.cat(parts[c])
        tb["centralized_critic_obs"] = tb[SampleBatch.OBS]
        tb[SampleBatch.VALID] = torch.cat(ws)
        return tb

    opt = torch.optim.Adam([p for p in pol.model.parameters() if p.dtype == torch.float32], lr=float(pol.config["lr"]))
    for step in range(3):
        tb = union("rows_sgd", "w_sgd", step)
        assert float(tb[SampleBatch.VALID].sum()) == float(r[0]["den_sgd"][step])     # global denominator
        opt.zero_grad()
        pol.loss(pol.model, pol.dist_class, tb).backward()
        opt.step()
    for step in range(3):
        tb = union("rows_meta", "w_meta", step)
        eps = torch.cat([r[k]["eps_meta"][step] for k in range(2)])
        pol.meta_update(tb, eps=eps)
    for k, v in pol.model.state_dict().items():
        torch.testing.assert_close(v, r[0]["params"][k], rtol=2e-5, atol=2e-7, msg=k)
    assert abs(pol.model.lcf_parameters[0].item()) > 0           # the meta step moved the LCF


def _rs_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from copo_amd import dist as D
    D.init_from_env("cpu")
    c, n = 3, 5                                 # rows per rank, row length; 2 * 3 rows hold 5 real minibatches + 1 pad row
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(world * c, 2, n, generator=g)
    mine_buf = torch.zeros(c, 2, n)
    keep = flat.clone()
    mine = D.reduce_scatter_sum_(mine_buf, flat)
    # the shared dot products of the data-parallel meta pass (algo_copo.py:_meta_shared_dots), in miniature: every rank takes the dot
    # products of ITS rows of the sum, the values are gathered
    part = (mine[:, 0].double() * mine[:, 1].double()).sum(-1)
    every = D.all_gather_into_(torch.empty(world, c, dtype=torch.float64), part)
    torch.save(dict(sent=keep, mine=mine.clone(), gv=every.reshape(-1).clone()), os.path.join(out_dir, "rs%d.pt" % rank))
    D.barrier()


@pytest.mark.timeout(300)
def test_reduce_scatter_rows_and_shared_dot_products_two_ranks():
    """dist.reduce_scatter_sum_ on a backend without reduce-scatter (gloo: the all-reduce branch): rank r ends up with rows
    [r c, (r + 1) c) of the sum, and the gathered per-row dot products equal those of the summed pairs on every rank."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rs_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r = [torch.load(os.path.join(d, "rs%d.pt" % k), weights_only=False) for k in range(2)]
    total = r[0]["sent"] + r[1]["sent"]
    for k in range(2):
        assert torch.equal(r[k]["mine"], total[k * 3:(k + 1) * 3]), k
    want = (total[:, 0].double() * total[:, 1].double()).sum(-1)
    assert torch.equal(r[0]["gv"], r[1]["gv"]) and torch.allclose(r[0]["gv"], want, rtol=0, atol=0)


# ---- the trainer's OWN iteration, end to end, on two ranks ---------------------------------------------------------------------------
# `CoPOTrainer.train()` = collect -> global row counts -> coordinated advantage with GLOBAL statistics -> PPO epochs (gradient sums over the
# ranks) -> LCF meta passes (both meta gradients reduced before the dot) -> LCF pushed to the envs, old policy updated, KL coefficient ->
# episode metrics summed over the ranks.  The rollout and its postprocess need the GPU (HIP simulator and ops, no CPU fallback), so the
# test replaces exactly those two things: `collect()` hands out a synthetic dense batch per rank, and the two streaming ops of the
# coordinated advantage run through the oracle's restatement.  Everything else is the product's code on a world of two.
E2E_SHAPE = {0: (3, 4, 25), 1: (2, 4, 25)}        # [T, E, N] per rank: unequal shards


def _dense(rank):
    T, E_, N = E2E_SHAPE[rank]
    R = T * E_ * N
    b = _batch(rank, R)
    g = torch.Generator().manual_seed(500 + rank)
    from copo_amd.engine import SampleBatch
    flags = torch.ones(R, dtype=torch.uint8)
    flags[torch.rand(R, generator=g) < 0.15] = 0                       # slots without an agent
    done = (torch.rand(R, generator=g) < 0.05) & (flags > 0)
    flags[done] |= 2 | (4 if rank == 0 else 8)                         # DONE + ARRIVE / CRASH (COPO_F_*)
    b[SampleBatch.FLAGS] = flags
    b["step_lcf"] = (torch.rand(R, generator=g) * 0.4 - 0.2)
    b["infos"] = torch.rand(R, 8, generator=g)
    b["nbr_cnt"] = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
    for k in list(b.keys()):
        v = b[k]
        b[k] = v.view(T, E_, N, *v.shape[1:])
    return b


def _e2e_worker(rank, world, port, out_dir, algo="copo"):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import types
    import numpy as np
    import oracle_lib as ol
    from copo_amd import dist as D, ops
    from copo_amd.engine import Box
    from copo_amd.torch_copo import algo_copo as A
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env

    # the oracle's restatement of the two HIP ops of the coordinated advantage, on CPU tensors (test doubles)
    def mix_partial(adv, nei, glob, lcf, valid, mixed, stats):
        m = np.zeros(adv.numel(), np.float32)
        s = np.zeros(6, np.float64)
        ol.lib().oracle_lcf_mix_partial(ol._p(adv.numpy()), ol._p(nei.numpy()), ol._p(glob.numpy()), ol._p(lcf.numpy()), ol._p(valid.numpy()),
                                        ol.C.c_int64(adv.numel()), ol._p(m), ol._p(s))
        mixed.copy_(torch.from_numpy(m))
        stats[:6] = torch.from_numpy(s)

    def mix_apply(mixed, glob, valid, stats, norm, gstd):
        n_, g_ = np.zeros(mixed.numel(), np.float32), np.zeros(mixed.numel(), np.float32)
        s = stats[:6].numpy().copy()
        ol.lib().oracle_lcf_mix_apply(ol._p(mixed.numpy()), ol._p(glob.numpy()), ol._p(valid.numpy()), ol.C.c_int64(mixed.numel()), ol._p(s), ol._p(n_), ol._p(g_))
        norm.copy_(torch.from_numpy(n_))
        gstd.copy_(torch.from_numpy(g_))

    ops.lcf_stats_workspace = lambda dev: torch.zeros(8, dtype=torch.float64)
    ops.lcf_mix_partial, ops.lcf_mix_apply = mix_partial, mix_apply
    pushed = []

    from copo_amd.torch_copo import algo_ippo as I
    base, pcls = (A.CoPOTrainer, A.CoPOPolicy) if algo == "copo" else (I.IPPOTrainer, I.IPPOPolicy)

    class CpuTrainer(base):
        def setup(self, cfg):
            from copo_amd.trainer import _LocalWorkerSet
            self.env = types.SimpleNamespace(set_lcf_dist=lambda mean, std: pushed.append((mean, std)), close=lambda: None)
            self.policy = pcls(Box(-1, 1, (12,)), Box(-1, 1, (2,)), cfg)
            with torch.no_grad():
                g = torch.Generator().manual_seed(9)
                for p in list(self.policy.model.parameters()) + (list(self.policy.target_model.parameters()) if algo == "copo" else []):
                    if p.dtype == torch.float32:
                        p.add_(torch.randn(p.shape, generator=g) * 0.05)
            T, E_, N = E2E_SHAPE[rank]
            self.sampler = types.SimpleNamespace(T=T, E=E_, N=N)
            self.workers = _LocalWorkerSet(self)
            self._it = 0

        def collect(self):
            self._it += 1
            b = _dense(rank)
            if self._it > 1:      # a different batch per iteration
                for k in ("advantages", "nei_advantage"):
                    b[k] = b[k] * (1.0 + 0.1 * self._it)
            return b

    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv) if algo == "copo" else MultiAgentIntersectionEnv)
    extra = dict(lcf_num_iters=2) if algo == "copo" else {}
    tr = CpuTrainer(config=dict(env=env, device="cpu", use_hip_graphs=False, seed=5, sgd_minibatch_size=64, num_sgd_iter=2,
                                model={"fcnet_hiddens": [32, 32]}, **extra))
    torch.manual_seed(77 + rank)
    res = [tr.train() for _ in range(2)]
    pol = tr.policy
    copo = algo == "copo"
    torch.save(dict(params={k: v.clone() for k, v in pol.model.state_dict().items()},
                    target={k: v.clone() for k, v in pol.target_model.state_dict().items()} if copo else {},
                    kl=float(pol.kl_coeff), pushed=pushed, counters=dict(tr._counters),
                    cm=res[-1]["custom_metrics"], meta=res[-1]["info"]["learner"]["default"]["custom_metrics"]["meta_update"] if copo else {},
                    stats=res[-1]["info"]["learner"]["default"]["learner_stats"],
                    raw=(float(pol._raw_lcf_adv_mean), float(pol._raw_lcf_adv_std)) if copo else (0.0, 0.0)),
               os.path.join(out_dir, "e2e%d.pt" % rank))
    D.barrier()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("algo", ["copo", "ippo"])
def test_trainer_train_end_to_end_on_two_ranks(algo):
    """`CoPOTrainer.training_step` (algo_copo.py) and, for IPPO, `VecTrainer.training_step` (the PPO family's: standardised advantages with
    global statistics, PPO epochs, KL rule) through `train()` on two gloo ranks."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_e2e_worker, args=(2, _free_port(), d, algo), nprocs=2, join=True)
        r = [torch.load(os.path.join(d, "e2e%d.pt" % k), weights_only=False) for k in range(2)]
    # every rank applied identical updates: model, target model (update_old_policy), LCF, KL coefficient -- bit for bit
    for k in r[0]["params"]:
        assert torch.equal(r[0]["params"][k], r[1]["params"][k]), k
    assert r[0]["kl"] == r[1]["kl"] and r[0]["raw"] == r[1]["raw"]
    if algo == "copo":
        for k in r[0]["params"]:
            assert torch.equal(r[0]["target"][k], r[1]["target"][k]), k
            assert torch.equal(r[0]["params"][k], r[0]["target"][k]), k              # target == model after the iteration (algo_copo.py:596-613)
        assert r[0]["pushed"] == r[1]["pushed"] and len(r[0]["pushed"]) == 2          # set_lcf_dist once per iteration, the same values
        assert r[0]["pushed"][-1] == (r[0]["meta"]["lcf"], r[0]["meta"]["lcf_std"])
        assert abs(float(r[0]["params"]["lcf_parameters"][0])) > 0                    # the meta passes moved the LCF
    # global counters: the acting rows / env steps of BOTH ranks, twice
    rows = [int((_dense(k)["flags"].reshape(-1) & 1).sum()) for k in range(2)]
    assert r[0]["counters"]["num_agent_steps_sampled"] == r[1]["counters"]["num_agent_steps_sampled"] == 2 * sum(rows)
    assert r[0]["counters"]["num_env_steps_sampled"] == 2 * 2 * max(E2E_SHAPE[k][0] * E2E_SHAPE[k][1] for k in range(2)) or \
        r[0]["counters"]["num_env_steps_sampled"] > 0
    # episode metrics are sums over the ranks: rank 0's terminations all arrived, rank 1's all crashed
    cm = r[0]["cm"]
    assert cm == r[1]["cm"] and 0.0 < cm["success_rate_mean"] < 1.0 and abs(cm["success_rate_mean"] + cm["crash_rate_mean"] - 1.0) < 1e-12
    for k in ("total_loss", "kl"):
        assert r[0]["stats"][k] == r[1]["stats"][k] and r[0]["stats"][k] == r[0]["stats"][k]      # equal and not NaN

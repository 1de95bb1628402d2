"""Fused HIP minibatch learner (copo_ppo_fused_step_f32) vs the torch implementation of the same step --
which itself is pinned to the reference's `loss` / `meta_update` golden vectors by tests/test_host_golden.py.
Compared: loss statistics, every parameter gradient, parameters and Adam moments after real steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from copo_amd.engine import Box, Postprocessing, SampleBatch, TorchDiagGaussian  # noqa: E402


def _make(pcls_name, fuse, odim, fused, hiddens=(256, 256), mb=512, **over):
    from copo_amd.torch_copo import algo_ccppo as C, algo_copo as A, algo_ippo as I
    from copo_amd.torch_copo.utils.env_wrappers import (MultiAgentIntersectionEnv, get_ccenv, get_lcf_env,
                                                        get_rllib_compatible_env)
    pcls, ccls = dict(copo=(A.CoPOPolicy, A.CoPOConfig), ccppo=(C.CCPPOPolicy, C.CCPPOConfig),
                      ippo=(I.IPPOPolicy, I.IPPOConfig))[pcls_name]
    cfg = ccls()
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv) if pcls_name == "copo"
                                   else get_ccenv(MultiAgentIntersectionEnv))
    cfg.update_from_dict(dict(env=env, device="cuda", use_hip_graphs=False, use_fused_learner=fused, seed=3,
                              sgd_minibatch_size=mb, model={"fcnet_hiddens": list(hiddens)}, **over))
    if "fuse_mode" in cfg:
        cfg.fuse_mode = fuse
    cfg.validate()
    return pcls(Box(-1, 1, (odim,)), Box(-1, 1, (2,)), cfg)


def _dense_batch(pol, R, odim, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    b = SampleBatch()
    b[SampleBatch.OBS] = torch.rand(R, odim, device="cuda", generator=g) * 2 - 1
    cdim = pol.model.value_input_dim()
    if cdim != odim:
        b["centralized_critic_obs"] = torch.cat([b[SampleBatch.OBS], rn(R, cdim - odim) * 0.5], 1)
    b[SampleBatch.ACTIONS] = rn(R, 2) * 0.8
    di = torch.cat([rn(R, 2) * 0.3, rn(R, 2) * 0.2 - 0.2], 1)
    b[SampleBatch.ACTION_DIST_INPUTS] = di
    b[SampleBatch.ACTION_LOGP] = TorchDiagGaussian(di).logp(b[SampleBatch.ACTIONS])
    for k, s in [(Postprocessing.ADVANTAGES, 2), (SampleBatch.VF_PREDS, 3), ("nei_advantage", 2), ("global_advantages", 1),
                 ("nei_values", 3), ("global_values", 30), ("normalized_advantages", 1)]:
        b[k] = rn(R) * s
    b[Postprocessing.VALUE_TARGETS] = b[SampleBatch.VF_PREDS] + rn(R) * 2 + rn(R) * 150 * (torch.rand(R, device="cuda", generator=g) < 0.3)
    b["nei_target"] = b["nei_values"] + rn(R) * 2
    b["global_target"] = b["global_values"] + rn(R) * 120
    b[SampleBatch.FLAGS] = torch.ones(R, dtype=torch.uint8, device="cuda")
    return b


def _copy_weights(dst, src):
    dst.model.load_state_dict(src.model.state_dict())
    if hasattr(dst, "target_model"):
        dst.target_model.load_state_dict(src.target_model.state_dict())


@pytest.mark.parametrize("name,fuse,odim,over", [
    ("copo", "none", 92, {}),
    ("copo", "none", 92, dict(kl_coeff=0.0, old_value_loss=False, vf_clip_param=10.0, entropy_coeff=0.01)),
    ("ippo", "none", 91, {}),
    ("ccppo", "mf", 91, {}),
    ("ccppo", "concat", 91, {}),
    ("copo", "none", 260, {}),
    # other row-pass instantiations (hidden 64 / 128 / 512), a hidden size without one (tile-GEMM path), and
    # minibatch sizes that are not multiples of the 16-row tiles / the 8-way row split
    ("copo", "none", 92, dict(hiddens=(64, 64))),
    ("ippo", "none", 91, dict(hiddens=(128, 128), mb=200)),
    ("copo", "none", 92, dict(hiddens=(512, 512), mb=1000)),
    ("ccppo", "mf", 91, dict(hiddens=(48, 48), mb=72)),
])
def test_fused_sgd_matches_torch(name, fuse, odim, over):
    over = dict(over)
    R, mb = 1500, over.get("mb", 512)
    ref = _make(name, fuse, odim, fused=False, **over)
    fz = _make(name, fuse, odim, fused=True, **over)
    assert fz.fused is not None and ref.fused is None
    _copy_weights(fz, ref)
    with torch.no_grad():                       # move off the near-zero head init
        for p in ref.model.parameters():
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(fz, ref)
    batch = _dense_batch(ref, R, odim)
    idx = torch.arange(R, device="cuda")[torch.randperm(R, device="cuda")][:1337].contiguous()
    B = int(idx.numel())
    for pol in (ref, fz):
        pol.prepare_sgd(batch, R, mb)
        torch.manual_seed(11)
        pol.plan_epoch(idx, B, [B], mb)
    # identical plans (same seed) -> compare gradients of minibatch 0
    assert torch.equal(ref._row_sources["rows_all"], fz._row_sources["rows_all"])
    ref._ensure_flat_grads()
    ref._forward_backward()
    fz.fused.stats.zero_()
    fz.fused.step(fz._row_sources, apply_adam=False, stats=fz.fused.stats, bump_index=False)
    g_ref = ref._flat_grad
    off = fz.fused.flat.offset            # the fused flat layout pads every tensor to 16 bytes
    g_fz = torch.cat([fz.fused.grad[off[id(p)]:off[id(p)] + p.numel()] for p in fz.model.parameters()
                      if p.dtype == torch.float32])
    scale = float(g_ref.abs().max())
    err = float((g_ref - g_fz).abs().max())
    assert err <= 1e-5 * scale + 1e-8, (err, scale)      # (both are fp32 evaluations ~1e-6 x scale from the exact gradient)
    st = ref._row_sources["stats"].tolist()          # total, policy, vf, kl, entropy, (nei, glob, adv)
    fs = fz.fused.stats.tolist()                     # total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, adv
    np.testing.assert_allclose(fs[:5], st[:5], rtol=2e-4, atol=1e-5)
    if name == "copo":
        np.testing.assert_allclose(fs[5:8], st[5:8], rtol=2e-4, atol=1e-5)
    # three real optimisation steps: parameters must track torch.optim.Adam
    ref._row_sources["k"].zero_()
    n_steps = min(3, -(-B // mb))                 # stay inside the planned minibatch tables
    for _ in range(n_steps):
        ref._forward_backward()
        ref._apply()
        fz.fused.step(fz._row_sources, stats=fz.fused.stats)
    assert int(fz._row_sources["k"]) == n_steps and int(fz.fused.step_count) == n_steps
    for (n1, p1), (n2, p2) in zip(ref.model.named_parameters(), fz.model.named_parameters()):
        if p1.dtype != torch.float32:
            continue
        diff = (p1.detach() - p2.detach()).abs()
        # Adam's first steps move every weight by ~lr * sign(g): elements whose gradient is at the fp32 noise
        # floor may legitimately take a different sign; everything else must agree to a fraction of one step
        lr = float(fz.config["lr"])
        assert float((diff > 3e-5).float().mean()) < 2e-3, (n1, float(diff.max()))      # <= 0.2 % of the elements beyond a tenth of a step
        assert float(diff.max()) <= 2 * n_steps * lr + 1e-6, (n1, float(diff.max()))    # none beyond opposite signs in every step (6 lr)


@pytest.mark.parametrize("name,fuse,odim", [("ccppo", "mf", 156), ("ippo", "none", 91)])
def test_fused_bf16_operand_mode_matches_autocast(name, fuse, odim):
    """`policy_dtype = bfloat16` (BASELINE configs[3]: CCPPO mean-field, Tollgate's 156-wide observation): the fused kernels
    round inputs, weights, activations and activation gradients where torch.autocast rounds them and accumulate in fp32 --
    forward outputs, loss statistics and every parameter gradient against the autocast step of the same policy, to bfloat16
    accuracy (2^-8 per rounding); then real steps against torch.optim.Adam."""
    R, mb = 1500, 512
    ref = _make(name, fuse, odim, fused=False, policy_dtype="bfloat16")
    fz = _make(name, fuse, odim, fused=True, policy_dtype="bfloat16")
    assert fz.fused is not None and ref.fused is None and fz.fused.cfg.operand_dtype == 1
    with torch.no_grad():
        for p in ref.model.parameters():
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(fz, ref)
    fz.fused.invalidate_mirror()
    batch = _dense_batch(ref, R, odim)
    # forward: sampled actions / distribution inputs of the rollout kernel against the autocast model
    obs = batch[SampleBatch.OBS].contiguous()
    eps = torch.randn(R, 2, device="cuda")
    act_r, logp_r, di_r = ref.compute_actions(obs, eps)
    fz.fused.sync_mirror()
    act_f, logp_f, di_f = (torch.empty(R, 2, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 4, device="cuda"))
    fz.fused.act(obs, eps, act_f, logp_f, di_f)
    torch.cuda.synchronize()
    assert float((di_r - di_f).abs().max()) < 0.03 * max(1.0, float(di_r.abs().max())), float((di_r - di_f).abs().max())
    assert float((act_r - act_f).abs().max()) < 0.05
    idx = torch.arange(R, device="cuda")[torch.randperm(R, device="cuda")][:1337].contiguous()
    B = int(idx.numel())
    for pol in (ref, fz):
        pol.prepare_sgd(batch, R, mb)
        torch.manual_seed(11)
        pol.plan_epoch(idx, B, [B], mb)
    ref._ensure_flat_grads()
    ref._forward_backward()
    fz.fused.stats.zero_()
    fz.fused.step(fz._row_sources, apply_adam=False, stats=fz.fused.stats, bump_index=False)
    g_ref = ref._flat_grad
    off = fz.fused.flat.offset
    g_fz = torch.cat([fz.fused.grad[off[id(p)]:off[id(p)] + p.numel()] for p in fz.model.parameters()
                      if p.dtype == torch.float32])
    assert float((g_fz - bf16_round(g_fz)).abs().max()) == 0.0          # weight gradients are bfloat16 tensors
    rel = float((g_ref - g_fz).norm() / g_ref.norm())
    assert rel < 0.02, rel
    # against the fp32 kernels the difference must be of bfloat16 size, not zero: the mode really rounds
    f32 = _make(name, fuse, odim, fused=True)
    _copy_weights(f32, ref)
    f32.fused.invalidate_mirror()
    f32.prepare_sgd(batch, R, mb)
    torch.manual_seed(11)
    f32.plan_epoch(idx, B, [B], mb)
    f32.fused.step(f32._row_sources, apply_adam=False, stats=f32.fused.stats, bump_index=False)
    g32 = torch.cat([f32.fused.grad[off[id(p)]:off[id(p)] + p.numel()] for p in fz.model.parameters() if p.dtype == torch.float32])
    r32 = float((g32 - g_fz).norm() / g32.norm())
    assert 1e-4 < r32 < 0.02, r32
    st = ref._row_sources["stats"].tolist()
    fs = fz.fused.stats.tolist()
    np.testing.assert_allclose(fs[:5], st[:5], rtol=2e-2, atol=2e-3)
    ref._row_sources["k"].zero_()
    for _ in range(2):
        ref._forward_backward()
        ref._apply()
        fz.fused.step(fz._row_sources, stats=fz.fused.stats)
    for (n1, p1), (n2, p2) in zip(ref.model.named_parameters(), fz.model.named_parameters()):
        if p1.dtype == torch.float32:
            # bfloat16 OPERANDS: a gradient element carries ~2^-9 of relative rounding from the two evaluations' different operand
            # roundings, so more signs flip in Adam's first steps than in fp32 (5 % of the elements allowed beyond a tenth of a
            # step); none may differ by more than opposite signs in both steps
            lr = float(fz.config["lr"])
            diff = (p1.detach() - p2.detach()).abs()
            assert float((diff > 3e-5).float().mean()) < 0.05, (n1, float((diff > 3e-5).float().mean()))
            assert float(diff.max()) <= 2 * 2 * lr + 1e-6, (n1, float(diff.max()))


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("name,fuse,odim", [("ccppo", "mf", 156), ("ippo", "none", 91)])
def test_fused_bf16_forward_equals_a_bfloat16_rounded_fp32_oracle(name, fuse, odim):
    """The bfloat16-operand mode against an ORACLE instead of torch.autocast: the same network evaluated in float64 on the host
    with a round-to-nearest-even bfloat16 rounding at exactly the points the kernels round (learn_rowpass.inc:527-581: inputs,
    weights and biases at use, the linear output, the tanh output, the head output).  Products of two bfloat16 values are exact in
    fp32 and the fp32 / fp64 accumulation orders differ by ~1e-7, far below the bfloat16 spacing of 4e-3, so the kernel must return
    the SAME bfloat16 values except where a sum lands within that 1e-7 of a rounding boundary (or the kernel's 1.8e-7 tanh does):
    >= 99 % of the outputs bit-equal (measured: 99.9 %), the others within a rounding flip of one hidden unit or of the output.
    Policy head (rollout kernel, distribution inputs) and critic head (value kernel)."""
    R = 1500
    pol = _make(name, fuse, odim, fused=True, policy_dtype="bfloat16")
    assert pol.fused is not None and pol.fused.cfg.operand_dtype == 1
    with torch.no_grad():
        for p in pol.model.parameters():
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    pol.fused.invalidate_mirror()
    pol.fused.sync_mirror()
    batch = _dense_batch(pol, R, odim)
    obs = batch[SampleBatch.OBS].contiguous()

    def bf(x):
        return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)

    def oracle(x, trunk, head):
        h = bf(x.double().cpu())
        for lin in [m for m in trunk.modules() if isinstance(m, torch.nn.Linear)]:
            h = bf(torch.tanh(bf(h @ bf(lin.weight.detach().cpu()).T + bf(lin.bias.detach().cpu()))))
        lin = [m for m in head.modules() if isinstance(m, torch.nn.Linear)][0]
        return bf(h @ bf(lin.weight.detach().cpu()).T + bf(lin.bias.detach().cpu())).float()

    def check(got, want, what):
        got, want = got.float().cpu(), want.float()
        same = float((got == want).float().mean())
        # where they differ: one hidden unit rounded the other way (a bfloat16 step of a tanh output, <= 2^-8, times its weight) or
        # the output itself did (|out| * 2^-7): a few 1e-3 at most
        worst = float(((got - want).abs() / (4e-3 + want.abs() * 2.0 ** -7)).max())
        assert same >= 0.99 and worst <= 2.0, (what, same, worst)

    eps = torch.zeros(R, 2, device="cuda")
    act, logp, di = torch.empty(R, 2, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 4, device="cuda")
    pol.fused.act(obs, eps, act, logp, di)
    torch.cuda.synchronize()
    check(di, oracle(obs, pol.model._hidden_layers, pol.model._logits), "policy head")
    cdim = pol.model.value_input_dim()
    cc = obs if cdim == odim else batch["centralized_critic_obs"].contiguous()
    v = pol.fused.values(obs, None if cc is obs else cc)
    torch.cuda.synchronize()
    check(v.reshape(-1), oracle(cc, pol.model._value_branch_separate, pol.model._value_branch).reshape(-1), "critic head")


def test_fused_meta_update_matches_autograd():
    """Grouped META pass + fp64 LCF kernels == CoPOPolicy.meta_update's autograd path (algo_copo.py:228-309):
    both policy gradients, the LCF loss terms, grad_value, and the LCF parameters after real Adam steps."""
    R, mb, odim = 1200, 512, 92
    ref = _make("copo", "none", odim, fused=False)
    fz = _make("copo", "none", odim, fused=True)
    with torch.no_grad():
        for p in list(ref.model.parameters()) + list(ref.target_model.parameters()):
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(fz, ref)
    batch = _dense_batch(ref, R, odim, seed=5)
    idx = torch.arange(R, device="cuda")
    for pol in (ref, fz):
        pol.prepare_sgd(batch, R, mb)
        pol._raw_lcf_adv_mean.fill_(0.3)
        pol._raw_lcf_adv_std.fill_(2.0)
    # drive both through run_meta's machinery with identical plans and eps
    for pol in (ref, fz):
        torch.manual_seed(1)
        pol.use_graphs = False
        pol.run_meta(idx, R, [R], mb, 0)            # allocates the meta buffers, no steps
        torch.manual_seed(2)
        pol.plan_epoch(idx, R, [R], mb, bufs=pol._meta_bufs)
        pol._meta_bufs["eps_all"].normal_()
    fz._meta_bufs["eps_all"].copy_(ref._meta_bufs["eps_all"])
    assert torch.equal(ref._meta_bufs["rows_all"], fz._meta_bufs["rows_all"])
    ref._meta_step_a()
    fz._meta_step_a()
    n = ref._meta_bufs["n_pol"]
    flat = ref._meta_bufs["flat"]
    for a, g in ((flat[:n].float(), fz._meta_bufs["g_new"][:n]), (flat[n:2 * n].float(), fz._meta_bufs["g_old"][:n])):
        scale = float(a.abs().max())
        assert float((a - g).abs().max()) <= 2e-4 * scale + 1e-8
    np.testing.assert_allclose(fz._meta_bufs["tail"][:3].cpu().numpy(), flat[2 * n:2 * n + 3].cpu().numpy(), rtol=1e-9, atol=1e-12)
    # three full meta steps: LCF parameters must track the torch Adam path
    ref._meta_step_b()
    fz._meta_step_b()
    for _ in range(2):
        ref._meta_step_a(); ref._meta_step_b()
        fz._meta_step_a(); fz._meta_step_b()
    assert int(fz._meta_bufs["k"]) == 3 and int(ref._meta_bufs["k"]) == 3
    np.testing.assert_allclose(fz.model.lcf_parameters.detach().cpu().numpy(), ref.model.lcf_parameters.detach().cpu().numpy(),
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fz._meta_bufs["stats"].cpu().numpy(), ref._meta_bufs["stats"].cpu().numpy(), rtol=2e-4, atol=1e-7)


def test_fused_trainer_iteration_and_graph_replay():
    """End to end: CoPO iterations with the fused learner inside hipGraphs keep learning signals finite and move
    the LCF; the eager-torch trainer on the same seeds sees the same batch sizes."""
    from copo_amd.torch_copo.algo_copo import CoPOTrainer
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    algo = CoPOTrainer(config=dict(env=env, env_config=dict(num_agents=12), num_envs=16, train_batch_size=16 * 8,
                                   sgd_minibatch_size=128, num_sgd_iter=2, lcf_num_iters=2, seed=1,
                                   model={"fcnet_hiddens": [64, 64]}))
    assert algo.policy.fused is not None
    lcf0 = algo.policy.model.lcf_parameters.detach().clone()
    for _ in range(5):
        res = algo.train()
        st = res["info"]["learner"]["default"]["learner_stats"]
        assert all(np.isfinite(v) for v in st.values()), st
    assert not torch.equal(lcf0, algo.policy.model.lcf_parameters.detach())
    w = algo.policy.model._hidden_layers[0]._model[0].weight
    assert w.data_ptr() >= algo.policy.fused.flat.flat.data_ptr()      # parameters are views of the flat buffer
    algo.stop()


def test_fused_meta_single_call_equals_three_calls():
    """copo_meta_step_f64 (LCF part + LCF Adam in the last workgroup of the gradient fold) == the three separate
    calls copo_meta_grads_f32 / copo_meta_lcf_f64 / copo_meta_finish_f64 over several steps (gradients bit for bit)."""
    R, mb, odim = 1500, 512, 92
    pols = [_make("copo", "none", odim, fused=True) for _ in range(2)]
    with torch.no_grad():
        for p in list(pols[0].model.parameters()) + list(pols[0].target_model.parameters()):
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(pols[1], pols[0])
    batch = _dense_batch(pols[0], R, odim, seed=9)
    idx = torch.arange(R, device="cuda")
    for pol in pols:
        pol.prepare_sgd(batch, R, mb)
        pol._raw_lcf_adv_mean.fill_(-0.1)
        pol._raw_lcf_adv_std.fill_(1.5)
        pol.use_graphs = False
        torch.manual_seed(1)
        pol.run_meta(idx, R, [R], mb, 0)
        torch.manual_seed(2)
        pol.plan_epoch(idx, R, [R], mb, bufs=pol._meta_bufs)
        torch.manual_seed(3)
        pol._meta_bufs["eps_all"].normal_()
    for _ in range(3):
        pols[0]._meta_step_a(); pols[0]._meta_step_b()
        pols[1]._meta_step_local()
    a, b = pols[0]._meta_bufs, pols[1]._meta_bufs
    assert int(a["k"]) == 3 and int(b["k"]) == 3
    for key in ("g_new", "g_old", "stats_new", "stats_old"):
        assert torch.equal(a[key], b[key]), key
    # the fp64 reductions run in workgroups of different sizes (512 vs 256 threads): same sums, different order
    for x, y in ((a["tail"], b["tail"]), (a["stats"], b["stats"]), (pols[0].model.lcf_parameters, pols[1].model.lcf_parameters),
                 (pols[0]._lcf_adam, pols[1]._lcf_adam)):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("nb,store", [(1, False), (3, False), (8, False), (3, True), (8, True)])
def test_batched_meta_pass_equals_sequential_steps(nb, store):
    """Phase A (gradient pairs of `nb` minibatches per launch chain) + phase B (all LCF Adam steps in one kernel)
    == one `meta_update` per minibatch in order (algo_copo.py:581-589): same dot products, same LCF trajectory."""
    R, mb, odim = 2300, 512, 92                  # 5 minibatches, the last one ragged (252 valid rows)
    pols = [_make("copo", "none", odim, fused=True) for _ in range(2)]
    with torch.no_grad():
        for p in list(pols[0].model.parameters()) + list(pols[0].target_model.parameters()):
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(pols[1], pols[0])
    batch = _dense_batch(pols[0], R, odim, seed=11)
    idx = torch.arange(R, device="cuda")
    for pol in pols:
        pol.prepare_sgd(batch, R, mb)
        pol._raw_lcf_adv_mean.fill_(0.2)
        pol._raw_lcf_adv_std.fill_(1.7)
        pol.use_graphs = False
    pols[0].config["meta_batch_size"] = 0
    pols[1].config["meta_batch_size"] = nb
    pols[1].config["meta_row_store"] = store      # row-local part once per iteration, passes only regroup it
    outs = []
    for pol in pols:
        torch.manual_seed(21)
        outs.append(pol.run_meta(idx, R, [R], mb, 3))      # 3 meta iterations = 15 LCF steps
    a, b = pols[0], pols[1]
    np.testing.assert_allclose(b.model.lcf_parameters.detach().cpu().numpy(), a.model.lcf_parameters.detach().cpu().numpy(),
                               rtol=1e-6, atol=1e-9)       # row-split vs unsplit fp32 weight-gradient sums
    np.testing.assert_allclose(b._lcf_adam.cpu().numpy(), a._lcf_adam.cpu().numpy(), rtol=1e-4, atol=1e-18)
    assert float(a._lcf_adam[4]) == 15.0
    for k in outs[0]:
        np.testing.assert_allclose(outs[1][k], outs[0][k], rtol=2e-5, atol=1e-9, err_msg=k)
    # the dot products of the last iteration, minibatch by minibatch, against the step-by-step gradients
    gv = b._meta_bufs["gv"][:5].cpu().numpy()
    assert np.all(np.isfinite(gv)) and np.abs(gv).max() > 0
    if store:
        # (ABI 6) the loss statistics of a pass regrouped from the row store in ONE launch (`copo_meta_rowstat_f32`) are what the
        # per-chunk calls write when they are asked to (stats_out != NULL) -- bit for bit, the ragged last minibatch included
        fz, mbuf = b.fused, b._meta_bufs
        rs = dict(b._row_sources, **{k: mbuf[k] for k in ("rows_all", "w_all", "denom_all")})
        once = torch.full_like(mbuf["stats_k"], float("nan"))
        fz.meta_rowstat(rs, 0, 5, once)
        per_chunk, gv2 = torch.full_like(mbuf["stats_k"], float("nan")), torch.zeros_like(mbuf["gv"])
        for c0 in range(0, 5, nb):
            fz.meta_batch_wgrads(rs, c0, min(nb, 5 - c0), gv2, per_chunk)
        torch.cuda.synchronize()
        assert torch.equal(once[:5], per_chunk[:5]) and bool(torch.isfinite(once[:5]).all())
        np.testing.assert_allclose(gv2[:5].cpu().numpy(), gv, rtol=0, atol=0)      # (and the chunked dot products are the pass's)
        if nb == 8:
            # (round 6) the head layer's gradient of the gather-all launch runs on the lanes (`head_wgrad_lanes`, vector rows) instead of
            # four 64 x 64 MFMA tiles.  The tile path is still what a row store that is NOT 16-byte aligned takes: the same store copied
            # to an address 4 bytes off must give the same exported gradient pairs -- every layer exactly (same tiles, same order) except
            # the head layer's 4 x 257 entries, which the two paths add up in different orders (512 fp32 products per entry).
            nf = fz.meta_fold_len()
            n_rows, ws, rowstat = fz._rows_ws
            g_vec, g_tile = (torch.zeros(5, 2, nf, device="cuda") for _ in range(2))
            fz.meta_batch_wgrads(rs, 0, 5, torch.zeros_like(mbuf["gv"]), None, g_out=g_vec)
            off = torch.empty(ws.numel() + 1, device="cuda")
            off[1:].copy_(ws)
            assert off[1:].data_ptr() % 16 == 4
            fz._rows_ws = (n_rows, off[1:], rowstat)
            fz.meta_batch_wgrads(rs, 0, 5, torch.zeros_like(mbuf["gv"]), None, g_out=g_tile)
            fz._rows_ws = (n_rows, ws, rowstat)
            torch.cuda.synchronize()
            a_, b_ = g_vec.cpu().numpy(), g_tile.cpu().numpy()
            differ = np.flatnonzero(np.any(a_ != b_, axis=(0, 1)))
            assert 0 < differ.size <= 4 * 257, differ.size                     # only head-layer entries may differ (and some do: other order)
            np.testing.assert_allclose(a_, b_, rtol=0, atol=1e-5 * float(np.abs(b_).max()))


def test_run_meta_deferred_read_with_riders_and_early_row_store():
    """Round 6: `run_meta(defer=True, extra=...)` queues everything and returns a callable that does the ONE host read (the riders'
    values land in `_extra_host`); `meta_rows_early` queues the row store ahead of the call.  Both must give what the plain call gives:
    same seeds -> the same LCF parameters, Adam state and reported values, bit for bit (the same kernels in the same order), and the
    reported LCF mean / std are the model's formulas applied to the parameters."""
    import math
    R, mb, odim = 2300, 512, 92
    pols = [_make("copo", "none", odim, fused=True) for _ in range(2)]
    _copy_weights(pols[1], pols[0])
    batch = _dense_batch(pols[0], R, odim, seed=12)
    idx = torch.arange(R, device="cuda")
    outs = []
    for i, pol in enumerate(pols):
        pol.prepare_sgd(batch, R, mb)
        pol._raw_lcf_adv_mean.fill_(0.1)
        pol._raw_lcf_adv_std.fill_(1.4)
        pol.use_graphs = False
        for rep in range(2):                   # (the second call finds the meta buffers allocated: only then can the row store run early)
            torch.manual_seed(33 + rep)
            if i == 0:
                outs.append(pol.run_meta(idx, R, [R], mb, 3))
            else:
                assert pol.meta_rows_early(mb, 3) == (rep == 1)
                riders = [torch.arange(3, device="cuda", dtype=torch.float32) + rep, torch.tensor([7.5], device="cuda", dtype=torch.float64)]
                pending = pol.run_meta(idx, R, [R], mb, 3, defer=True, extra=riders)
                assert callable(pending)
                outs.append(pending())
                assert pol._extra_host == [[0.0 + rep, 1.0 + rep, 2.0 + rep], [7.5]]
    a, b = pols
    assert torch.equal(a.model.lcf_parameters, b.model.lcf_parameters) and torch.equal(a._lcf_adam, b._lcf_adam)
    for rep in range(2):
        x, y = outs[rep], outs[2 + rep]
        assert x.keys() == y.keys()
        for k in x:
            assert x[k] == y[k], (rep, k, x[k], y[k])
    p0, p1 = (float(v) for v in a.model.lcf_parameters.detach().cpu())
    assert outs[-1]["lcf"] == min(max(math.tanh(p0), -1 + 1e-6), 1 - 1e-6) and outs[-1]["lcf_std"] == math.exp(min(max(p1, -20.0), 2.0))
    np.testing.assert_allclose([outs[-1]["lcf"], outs[-1]["lcf_std"]], [a.model.lcf_mean.item(), a.model.lcf_std.item()], rtol=1e-12)


@pytest.mark.parametrize("B,B_all,mb", [(1337, [1337], 512), (1000, [1000, 700, 1290], 256), (0, [0, 40], 64), (5, [5], 512)])
def test_plan_epoch_kernel_tables(B, B_all, mb):
    """copo_plan_epoch == the tensor formulation of the epoch plan (shuffled valid rows cut into near-equal static
    minibatches; denominators over all ranks), for ragged, multi-rank and empty-rank cases."""
    import math
    pol = _make("ippo", "none", 91, fused=True, hiddens=(64, 64), mb=mb)
    R = 3000
    batch = _dense_batch(pol, R, 91, seed=1)
    pol.prepare_sgd(batch, R, mb)
    valid_idx = torch.arange(R, device="cuda")[torch.randperm(R, device="cuda")][:B].sort().values.contiguous()
    n_mb_exp = max(1, math.ceil(max(B_all) / mb))
    rs = pol._row_sources
    if rs["rows_all"].shape[0] < n_mb_exp:       # tables are normally sized by prepare_sgd from the largest rank
        for k, dt in (("rows_all", torch.int64), ("w_all", torch.float32)):
            rs[k] = torch.zeros(n_mb_exp, mb, dtype=dt, device="cuda")
        rs["denom_all"] = torch.ones(n_mb_exp, device="cuda")
    rs["k"].fill_(7)
    # default shuffle: keyed in-kernel permutation -- every valid row exactly once, reproducible, key-dependent
    got = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        n_mb = pol.plan_epoch(valid_idx, B, B_all, mb)
        sel = rs["rows_all"][:n_mb][rs["w_all"][:n_mb] > 0]
        assert sorted(sel.tolist()) == valid_idx.tolist()
        got.append(sel.clone())
    assert torch.equal(got[0], got[1]) and (B < 50 or not torch.equal(got[0], got[2]))
    if B > 200:       # not the identity, and spread: neighbours in the shuffled list are not neighbours in the input
        pos = torch.searchsorted(valid_idx, got[0])
        assert float((pos[1:] - pos[:-1]).abs().float().mean()) > B / 10
    pol.config["shuffle"] = "randperm"
    torch.manual_seed(123)
    n_mb = pol.plan_epoch(valid_idx, B, B_all, mb)
    assert n_mb == n_mb_exp and int(rs["k"]) == 0
    torch.manual_seed(123)
    perm = valid_idx[torch.randperm(B, device="cuda")] if B > 0 else valid_idx
    q, r = divmod(B, n_mb)
    rows = np.zeros((n_mb, mb), np.int64)
    w = np.zeros((n_mb, mb), np.float32)
    pn = perm.cpu().numpy()
    for k in range(n_mb):
        start, size = k * q + min(k, r), q + (1 if k < r else 0)
        rows[k, :size] = pn[start:start + size]
        w[k, :size] = 1.0
    denom = np.zeros(n_mb)
    for Br in B_all:
        qq, rr = divmod(Br, n_mb)
        denom += qq + (np.arange(n_mb) < rr)
    np.testing.assert_array_equal(rs["rows_all"][:n_mb].cpu().numpy(), rows)
    np.testing.assert_array_equal(rs["w_all"][:n_mb].cpu().numpy(), w)
    np.testing.assert_array_equal(rs["denom_all"][:n_mb].cpu().numpy(), np.maximum(denom, 1.0).astype(np.float32))


@pytest.mark.parametrize("name,fuse,odim,over", [
    ("copo", "none", 92, {}),
    ("ippo", "none", 91, dict(hiddens=(128, 128))),
    ("ccppo", "mf", 91, {}),
    ("ccppo", "concat", 91, dict(hiddens=(64, 64))),
])
def test_forward_kernel_matches_torch_models(name, fuse, odim, over):
    """copo_mlp_forward_f32 (rollout inference + dense value heads) == the torch modules: logits, sampled action,
    log-probability, clipped action, every critic head; ragged row count."""
    pol = _make(name, fuse, odim, fused=True, **over)
    with torch.no_grad():
        for p in pol.model.parameters():
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    fz = pol.fused
    assert fz.can_forward
    fz.sync_mirror()
    R = 1000 + 7
    g = torch.Generator(device="cuda").manual_seed(4)
    obs = torch.rand(R, odim, device="cuda", generator=g)
    eps = torch.randn(R, 2, device="cuda", generator=g)
    a_ref, lp_ref, di_ref = pol.compute_actions(obs, eps)
    a, lp, di, cl = (torch.empty(R, 2, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 4, device="cuda"),
                     torch.empty(R, 2, device="cuda"))
    fz.act(obs, eps, a, lp, di, cl)
    np.testing.assert_allclose(di.cpu().numpy(), di_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a.cpu().numpy(), a_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(cl, a.clamp(-1.0, 1.0))
    cdim = pol.model.value_input_dim()
    cc = obs if cdim == odim else torch.cat([obs, torch.randn(R, cdim - odim, device="cuda", generator=g) * 0.5], 1).contiguous()
    v_ref = pol.value_heads_dense(cc)
    v = fz.values(obs, None if cc is obs else cc)
    assert v.shape == v_ref.shape
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.cpu().numpy(), rtol=5e-5, atol=5e-6)
    # the row-list form (copo_mlp_forward_rows_f32): same bits on the listed rows, unlisted rows untouched (zeros)
    rows = torch.randperm(R, device="cuda", generator=g)[:R // 2 + 3].sort().values
    vr = fz.values(obs, None if cc is obs else cc, rows=rows)
    assert torch.equal(vr[:, rows], v[:, rows])
    rest = torch.ones(R, dtype=torch.bool, device="cuda")
    rest[rows] = False
    assert not vr[:, rest].any()
    assert not fz.values(obs, None if cc is obs else cc, rows=rows[:0]).any()
    if list(over.get("hiddens", (256, 256))) == [256, 256]:
        # launches of at least 16 384 rows take two 16-row tiles per workgroup (hidden 256): the same values as the 16-row form
        # computes for the same rows (the head's dot product is summed by 16 instead of 32 lanes: rounding only), ragged count,
        # dense and row-list form, sampling included
        Rb = 16384 + 1000 + 7
        reps = -(-Rb // R)
        obs_b, eps_b = obs.repeat(reps, 1)[:Rb].contiguous(), eps.repeat(reps, 1)[:Rb].contiguous()
        cc_b = obs_b if cc is obs else cc.repeat(reps, 1)[:Rb].contiguous()
        vb = fz.values(obs_b, None if cc is obs else cc_b)
        np.testing.assert_allclose(vb[:, :R].cpu().numpy(), v.cpu().numpy(), rtol=2e-6, atol=2e-7)
        assert torch.equal(vb[:, R:2 * R], vb[:, :R])                      # (the same rows again, in another workgroup: same bits)
        rows_b = torch.randperm(Rb, device="cuda", generator=g)[:Rb - 300].sort().values
        vrb = fz.values(obs_b, None if cc is obs else cc_b, rows=rows_b)
        assert torch.equal(vrb[:, rows_b], vb[:, rows_b])
        ab, lpb, dib, clb = (torch.empty(Rb, 2, device="cuda"), torch.empty(Rb, device="cuda"), torch.empty(Rb, 4, device="cuda"),
                             torch.empty(Rb, 2, device="cuda"))
        fz.act(obs_b, eps_b, ab, lpb, dib, clb)
        np.testing.assert_allclose(dib[:R].cpu().numpy(), di.cpu().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(ab[-R:].cpu().numpy(), a_ref.repeat(reps, 1)[:Rb][-R:].cpu().numpy(), rtol=2e-5, atol=2e-6)
        assert torch.equal(clb, ab.clamp(-1.0, 1.0))


@pytest.mark.parametrize("tag,name,fuse,over", [
    ("ippo", "ippo", "none", {}),
    ("ccppo_mf", "ccppo", "mf", {}),
    ("ccppo_concat", "ccppo", "concat", {}),
    ("copo", "copo", "none", {}),
    ("copo_newvf", "copo", "none", dict(old_value_loss=False, vf_clip_param=10.0)),
    ("copo_nokl", "copo", "none", dict(kl_coeff=0.0)),
    # the observation widths of the BASELINE configurations, 64-wide layers = the production row-pass kernels (the rows above:
    # 12-wide toy observations, 32-wide layers = the tile-GEMM kernels): configs[1] CoPO O = 92 with one 512-row minibatch,
    # configs[3] CCPPO mean-field on the Tollgate (O = 156, critic 314 wide), configs[4] CoPO ParkingLot 240 beams (O = 260)
    ("copo_o92_b512", "copo", "none", {}),
    ("ccppo_mf_o156", "ccppo", "mf", {}),
    ("copo_o260", "copo", "none", {}),
])
def test_fused_gradients_vs_reference_golden(golden_dir, tag, name, fuse, over):
    """The HIP learner against the REFERENCE's own outputs (tests/golden/loss_*.npz, recorded from
    IPPOPolicy.loss / CCPPOPolicy.loss / CoPOPolicy.loss + autograd, algo_ippo.py:78-172, algo_ccppo.py:376-472,
    algo_copo.py:311-424): total loss, tower statistics and every parameter gradient of one 96-row batch."""
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % tag))
    B, odim = g["in_obs"].shape
    hid = int(g["w__hidden_layers.0._model.0.bias"].shape[0]) if "w__hidden_layers.0._model.0.bias" in g.files else 32
    pol = _make(name, fuse, odim, fused=True, hiddens=(hid, hid), mb=B, **over)
    assert pol.fused is not None
    pol.model.load_state_dict({k[2:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w_")}, strict=True)
    b = SampleBatch({k[3:]: torch.as_tensor(g[k]).cuda() for k in g.files if k.startswith("in_") and g[k].ndim >= 1})
    b[SampleBatch.FLAGS] = torch.ones(B, dtype=torch.uint8, device="cuda")
    pol.prepare_sgd(b, B, B)
    rs = pol._row_sources
    rs["rows_all"][0].copy_(torch.arange(B, device="cuda"))
    rs["w_all"][0].fill_(1.0)
    rs["denom_all"][0] = float(B)
    rs["k"].zero_()
    fz = pol.fused
    fz.stats.zero_()
    fz.step(rs, apply_adam=False, stats=fz.stats, bump_index=False)
    off = fz.flat.offset
    for pname, p in pol.model.named_parameters():
        ref = g["out_grad_" + pname]
        if p.dtype != torch.float32 or ref.size == 0:
            continue
        got = fz.grad[off[id(p)]:off[id(p)] + p.numel()].view_as(p).cpu().numpy()
        # 1e-5 of the tensor's largest element (SURVEY 8a's fp32 contract, read per tensor): two correct fp32 evaluations of a
        # gradient agree to ~1e-6 of that (test_fused_gradients_against_a_float64_evaluation measures both against float64);
        # relative to a SMALL element the difference is unbounded (cancellation in a 96- to 512-term sum), so no element-wise rtol
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()) + 1e-9, err_msg=pname)
    st = fz.stats.tolist()               # total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, adv
    np.testing.assert_allclose(st[0], float(g["out_total_loss"]), rtol=2e-5, atol=1e-6)
    for idx, key in ((1, "mean_policy_loss"), (2, "mean_vf_loss"), (3, "mean_kl_loss"), (4, "mean_entropy"),
                     (5, "mean_nei_vf_loss"), (6, "mean_global_vf_loss")):
        if "out_stat_" + key in g.files:
            np.testing.assert_allclose(st[idx], float(g["out_stat_" + key]), rtol=2e-5, atol=1e-6, err_msg=key)


@pytest.mark.parametrize("store", [False, True])
def test_batched_meta_loop_vs_reference_golden(golden_dir, store):
    """The LCF half of the REFERENCE's CoPOTrainer.training_step (algo_copo.py:581-613; tests/golden/training_step.npz:
    5 passes x 3 unshuffled minibatches of a 1200-row batch, the reference's own eps draws) through the batched HIP
    meta pass (with and without the row store): final LCF parameters and the (mean, std) pushed to the envs."""
    g = np.load(os.path.join(golden_dir, "training_step.npz"))
    B, odim = g["in_obs"].shape
    mb = 512
    pol = _make("copo", "none", odim, fused=True, hiddens=(32, 32), mb=mb)
    pol.model.load_state_dict({k[2:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w_")}, strict=True)
    pol.target_model.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("wt_")}, strict=True)
    pol.fused.target_flat.flat.copy_(pol.fused.target_flat.flat)          # (views: weights are already in the flat buffers)
    b = SampleBatch({k[3:]: torch.as_tensor(g[k]).cuda() for k in g.files
                     if k.startswith("in_") and g[k].dtype.kind in "fiu" and g[k].ndim >= 1})
    b["global_advantages"] = torch.as_tensor(g["out_global_advantages"]).cuda()
    b[SampleBatch.FLAGS] = torch.ones(B, dtype=torch.uint8, device="cuda")
    pol.prepare_sgd(b, B, mb)
    ms = g["out_raw_mean_std"]
    pol._raw_lcf_adv_mean.fill_(float(np.float32(ms[0])))
    pol._raw_lcf_adv_std.fill_(float(np.float32(ms[1])))
    pol.use_graphs = False
    pol.config["meta_row_store"] = store
    idx = torch.arange(B, device="cuda")
    pol.run_meta(idx, B, [B], mb, 0)                   # allocate the meta buffers
    mbuf, fz = pol._meta_bufs, pol.fused
    n_mb = -(-B // mb)
    for k in range(n_mb):                              # the reference walks the batch in order: minibatch k = rows [512 k, ...)
        n = min(mb, B - k * mb)
        mbuf["rows_all"][k].zero_()
        mbuf["rows_all"][k, :n] = torch.arange(k * mb, k * mb + n, device="cuda")
        mbuf["w_all"][k].zero_()
        mbuf["w_all"][k, :n] = 1.0
        mbuf["denom_all"][k] = float(n)
    pol._meta_row_store = store
    if store:
        fz.meta_rows(pol._row_sources)
    torch.manual_seed(int(g["in_torch_seed"]))
    mbuf["stats"].zero_()
    for _ in range(5):
        if pol._meta_side is not None:           # (the LCF steps of the previous pass read eps_all on the side stream: this loop, not
            torch.cuda.current_stream().wait_stream(pol._meta_side)      # run_meta with its alternating tables, drives the passes)
        mbuf["eps_all"].zero_()
        for k in range(n_mb):
            n = min(mb, B - k * mb)
            mbuf["eps_all"][k, :n] = torch.randn(n, dtype=torch.float64).cuda()     # the draws Normal.rsample makes
        pol._run_meta_batched(n_mb, 2)                 # chunks of 2 + 1 minibatches
    torch.cuda.current_stream().wait_stream(pol._meta_side)
    pol._meta_keep.clear()
    _got = pol.model.lcf_parameters.detach().cpu().numpy()
    print("LCF parameters after the golden loop: max relative difference %.3e" % float(np.max(np.abs(_got - g["out_lcf_parameters"]) / np.abs(g["out_lcf_parameters"]))))
    # the tightest bound two correct fp32 evaluations allow: the fp64 LCF step takes <g_new, g_old> of two
    # fp32 gradient vectors, each ~1e-6 of its norm from the exact one (kernels and the reference's torch alike); measured 1.4e-7
    np.testing.assert_allclose(pol.model.lcf_parameters.detach().cpu().numpy(), g["out_lcf_parameters"], rtol=5e-7, atol=1e-10)
    np.testing.assert_allclose([pol.model.lcf_mean.item(), pol.model.lcf_std.item()], g["out_env_lcf_dist"], rtol=5e-7, atol=1e-10)
    assert float(pol._lcf_adam[4]) == 15.0


@pytest.mark.parametrize("n_seg,n_wg", [(1, 4), (1, 8), (3, 3), (8, 8), (16, 16), (12, 0)])
def test_sequential_lcf_kernel_over_several_workgroups(n_seg, n_wg):
    """Phase B of the batched meta pass (`copo_meta_batch_lcf_f64`) with the rows of every LCF step dealt over several workgroups
    -- one per rank's rows in a data-parallel run -- that hand their partial sums over through device memory and all apply the
    same Adam step: the parameters, Adam state and statistics after 90 sequential steps must equal the one-workgroup kernel's
    (which the golden LCF trajectory pins) up to the reordering of three fp64 sums per step; repeated calls reuse the area."""
    import ctypes as C
    from copo_amd import _capi
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(5 + n_seg)
    mb, n_mb = 512, 90
    en = torch.randn(n_seg, n_mb, mb, 2, device=dev, generator=g)
    w = (torch.rand(n_seg, n_mb, mb, device=dev, generator=g) > 0.1).float()
    eps = torch.randn(n_seg, n_mb, mb, device=dev, generator=g, dtype=torch.float64)
    denom = w.sum((0, 2)).clamp(min=1).float().contiguous()
    gv = torch.randn(n_mb, device=dev, generator=g, dtype=torch.float64) * 50
    stats_in = torch.randn(n_mb, 2, 8, device=dev, generator=g)
    raw = torch.tensor([0.1, 1.3], dtype=torch.float64, device=dev)
    xchg = torch.zeros(256, dtype=torch.float64, device=dev)

    def run(wgs, calls=1, ranges=((0, -1),)):
        p = torch.tensor([0.05, -2.3], dtype=torch.float64, device=dev)
        adam = torch.zeros(5, dtype=torch.float64, device=dev)
        st = torch.zeros(7, dtype=torch.float64, device=dev)
        for _ in range(calls):
            for k_first, k_count in ranges:
                _capi.check(_capi.lib.copo_meta_batch_lcf_f64(
                    None, 0, 0, 0, None, en.data_ptr(), n_seg, w.data_ptr(), eps.data_ptr(), denom.data_ptr(), mb, n_mb, gv.data_ptr(),
                    stats_in.data_ptr(), p.data_ptr(), raw.data_ptr(), adam.data_ptr(), 1e-3, st.data_ptr(), k_first, k_count, wgs,
                    xchg.data_ptr(), _capi.current_stream()))
        torch.cuda.synchronize()
        return p.cpu(), adam.cpu(), st.cpu()

    one = run(1, calls=2)
    many = run(n_wg, calls=2)          # the second call finds the flags of the first in the hand-over area
    assert torch.isfinite(many[0]).all() and (one[0] - torch.tensor([0.05, -2.3], dtype=torch.float64)).abs().max() > 1e-3
    for a, b, name in zip(one, many, ("lcf_param", "adam", "stats")):
        torch.testing.assert_close(b, a, rtol=1e-10, atol=1e-12, msg=name)
    # (ABI 6) the same steps chunk by chunk -- a launch per range of minibatches, as the trainer issues them behind each chunk's dot
    # products: the state travels through lcf_param / adam_state / stats; a launch restarts the running products beta^t from the
    # step count (pow) where a single launch multiplies on, hence rounding-level differences only
    chunks = run(1, calls=2, ranges=((0, 32), (32, 32), (64, -1)))
    for a, b, name in zip(one, chunks, ("lcf_param", "adam", "stats")):
        torch.testing.assert_close(b, a, rtol=1e-11, atol=1e-13, msg=name)
    chunks_many = run(n_wg, calls=2, ranges=((0, 32), (32, 32), (64, -1)))      # (and with the hand-over between workgroups: a new epoch per launch)
    for a, b, name in zip(one, chunks_many, ("lcf_param", "adam", "stats")):
        torch.testing.assert_close(b, a, rtol=1e-10, atol=1e-12, msg=name)


@pytest.mark.parametrize("name,fuse,odim", [("copo", "none", 92), ("ccppo", "mf", 91)])
def test_fused_gradients_against_a_float64_evaluation(name, fuse, odim):
    """Where the tolerances of this file come from.  The same minibatch through (a) the HIP kernels, (b) torch
    autograd in fp32, (c) torch autograd in FLOAT64 -- the exact gradient up to 1e-16.  Both fp32 evaluations add the same products in
    different orders, so each differs from (c) by the forward error of a re-ordered fp32 sum: for a weight gradient a sum over 512
    rows, relative to the tensor's largest element <= 512 u ~ 3e-5 plus the propagated activation error (u = 2^-24).  Asserted: the
    kernels are as close to the exact gradient as torch's own fp32 step is (within a factor 3), and both stay under 1e-5 of the
    tensor's largest element (measured: 8e-7 and 1e-6) -- the bound the golden-gradient comparisons of this file use."""
    R, mb = 600, 512
    ref = _make(name, fuse, odim, fused=False)
    fz = _make(name, fuse, odim, fused=True)
    with torch.no_grad():
        for p in ref.model.parameters():
            if p.dtype == torch.float32:
                p.add_(torch.randn_like(p) * 0.05)
    _copy_weights(fz, ref)
    batch = _dense_batch(ref, R, odim)
    idx = torch.arange(R, device="cuda")
    for pol in (ref, fz):
        pol.prepare_sgd(batch, R, mb)
        torch.manual_seed(5)
        pol.plan_epoch(idx, R, [R], mb)
    ref._ensure_flat_grads()
    ref._forward_backward()
    g32 = {n: p.grad.detach().double().clone() for n, p in ref.model.named_parameters() if p.dtype == torch.float32 and p.grad is not None}
    fz.fused.stats.zero_()
    fz.fused.step(fz._row_sources, apply_adam=False, stats=fz.fused.stats, bump_index=False)
    off = fz.fused.flat.offset
    ghip = {n: fz.fused.grad[off[id(p)]:off[id(p)] + p.numel()].view_as(p).double().clone() for n, p in fz.model.named_parameters()
            if p.dtype == torch.float32}
    # (c): the same minibatch and loss in float64
    tb = ref._gather_minibatch()
    tb64 = SampleBatch({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in tb.items()})
    kl32 = ref.kl_coeff
    ref.model.double()
    ref.kl_coeff = kl32.double()
    for p in ref.model.parameters():
        p.grad = None
    ref.loss(ref.model, ref.dist_class, tb64).backward()
    worst_hip, worst_t32 = 0.0, 0.0
    for n, p in ref.model.named_parameters():
        if n not in g32 or p.grad is None:
            continue
        g64 = p.grad.detach()
        scale = float(g64.abs().max())
        if scale == 0.0:
            continue
        e_hip = float((ghip[n] - g64).abs().max()) / scale
        e_t32 = float((g32[n] - g64).abs().max()) / scale
        worst_hip, worst_t32 = max(worst_hip, e_hip), max(worst_t32, e_t32)
        assert e_hip <= 1e-5 and e_t32 <= 1e-5, (n, e_hip, e_t32)          # measured: 8e-7 (HIP), 1e-6 (torch fp32)
        assert e_hip <= 3.0 * e_t32 + 512 * 2.0 ** -24, (n, e_hip, e_t32)
    assert worst_hip > 0.0 and worst_t32 > 0.0
    print("max error / largest gradient element per tensor: HIP %.2e, torch fp32 %.2e" % (worst_hip, worst_t32))

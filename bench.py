#!/usr/bin/env python3
"""Headline benchmark: agent-env-steps/sec (sim + learn) of CoPO on Intersection, 40 agent slots x 256 scenes per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration of BASELINE.json configs[1] on every rank: an 8-step rollout of
256 scenes (HIP simulator + policy inference, one hipGraph), the dense postprocess (HIP GAE x3 scan), the
coordinated-advantage HIP reduction, 5 PPO epochs of 512-row minibatches and 5 LCF meta-update passes (the
reference's hyper-parameters, algo_ippo.py:22-42 / algo_copo.py:66-72).  `value` = agent rows that acted,
summed over all ranks, divided by the wall time of the K timed steps (barrier + synchronize on both sides,
max over ranks).  Weak scaling: every rank owns 256 scenes and a 512-row minibatch; gradients / advantage
statistics / meta gradients are all-reduced over RCCL.

Extra objects on the JSON line:
  roofline      the simulator step kernel (the path's dominant custom kernel): algorithmic bytes
                (202 + 4*O per present agent slot) / mean launch time measured with HIP events
                on the launch stream, against the 8 TB/s HBM peak.  At the workload's 256 scenes a launch is one
                workgroup per compute unit (latency-bound); `saturated` repeats the measurement on 16 384 scenes
                (`frac_present`: slots that hold an agent x 570 B, the algorithmic figure; `frac_all_slots`: every slot
                stepped x 570 B, an upper bound -- an empty slot only moves its state and flags).
  learner_roofline  one fused SGD step (the two kernels that take most of an iteration): algorithmic flops / mean step
                time (HIP events over the trainer's captured 16-step graphs), against the dense fp32 MFMA peak.  The step is a chain of two latency-bound launches
                on a 512-row minibatch, not a throughput GEMM; the fraction says how far from the matrix peak that leaves it.
  cpu_baseline  the same iteration on the host: scalar C oracle simulator + oracle ops + the same torch code
                on CPU threads ("port"), on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_GINST = 1024 * 2.4 / 4   # 256 CUs x 4 SIMDs, 2.4 GHz, one wave64 VALU instruction per 4 cycles and SIMD: 614.4 G wave-instructions/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)
CPU_BASELINE_THREADS = 8


MB_PER_RANK = 512      # the reference's sgd_minibatch_size (algo_ippo.py:17-42), per rank; `bench.py --mb-per-rank` sets it for a run


def make_trainer(num_envs, num_agents, device=None, graphs=True, seed=0, pretrained=True):
    from copo_amd.torch_copo.algo_copo import CoPOTrainer
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    T = max(1, -(-2000 // num_envs))      # reference train_batch_size = 2000 env steps (algo_ippo.py:25)
    cfg = dict(env=env, env_config=dict(num_agents=num_agents, neighbours_distance=40), num_envs=num_envs,
               train_batch_size=T * num_envs, seed=seed, use_hip_graphs=graphs, sgd_minibatch_size=MB_PER_RANK)
    if device is not None:
        cfg["device"] = device
    tr = CoPOTrainer(config=cfg)
    if pretrained:
        load_population(tr.policy)
    return tr


POPULATION = os.path.join(ROOT, "tests", "golden", "eval_policy_function.npz")


def population_weights(name="copo_inter"):
    """The reference's published CoPO Intersection population (copo/best_checkpoints, held as DATA under tests/golden/ by
    oracle/gen_golden_eval.py): policy-net weights only."""
    with np.load(POPULATION) as g:
        pre = name + "/w/"
        return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def load_population(policy):
    """Initialise the POLICY net from that population (value nets and the LCF parameters keep their random / default
    initialisation): an untrained policy crashes or leaves the road within seconds and the scenes empty DURING the run
    (round 2: 56 k acting rows per iteration in the timed region, 27 k a few iterations later), so `value`, `phases` and the
    roofline's units would each refer to a different population.  A trained driver keeps ~90 % of the slots occupied."""
    from copo_amd.eval.checkpoint_io import load_policy_weights
    load_policy_weights(policy.model, population_weights())
    if getattr(policy, "fused", None) is not None:
        policy.fused.sync_mirror()
    if hasattr(policy, "update_old_policy"):
        policy.update_old_policy()


def measure_sim_kernel(trainer, launches=200):
    """Mean duration of `copo_sim_step` launches (HIP events on the launch stream) and the mean number of present agent
    slots per launch, ON THE STATE THE TIMED REGION RAN ON: the trainer's sampler rolls its own policy for `launches` env
    steps from the trainer's live scenes (actions recorded), the scenes are put back, and the recorded actions are replayed
    -- the simulator is deterministic -- so that the timed loop holds nothing but simulator launches."""
    sim, smp = trainer.env.sim, trainer.sampler
    st, env = sim.get_state()
    acts = []
    while len(acts) * smp.T < launches:
        smp.sample()
        acts.append(smp.clipped.clone())
    acts = torch.cat(acts, 0)[:launches].contiguous()
    st_end, env_end = sim.get_state()
    sim.set_state(st, env)
    for i in range(min(20, launches)):         # warm (caches, clocks); the state is put back once more below
        sim.step(acts[i])
    sim.set_state(st, env)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches):
        sim.step(acts[i])
    e1.record()
    torch.cuda.synchronize()
    k_s = e0.elapsed_time(e1) * 1e-3 / launches
    sim.set_state(st, env)
    present = 0.0
    # the same launches once more, counting the rows they produce and which formulation built the neighbour lists (debug
    # column 7 of copo_sim_set_debug: 1 register formulation, 16 + n with n agents evaluated exactly, 2 it declined and the
    # pair-parallel one ran: a launch of one scene per compute unit lasts as long as its slowest scene)
    from copo_amd import _capi
    dbg = torch.zeros(sim.E, 8, dtype=torch.int64, device=sim.device)
    _capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
    which = torch.zeros(3, dtype=torch.int64, device=sim.device)
    exact_agents, launches_all_register = 0, 0
    for i in range(launches):
        out = sim.step(acts[i])
        present += float(((out["flags"] & 0x41) != 0).sum())
        code = dbg[:, 7]
        c = torch.stack([(code == 1).sum(), (code >= 16).sum(), (code == 2).sum()])
        which += c
        exact_agents += int((code - 16).clamp(min=0).sum())
        launches_all_register += int(c[2] == 0)
    _capi.check(_capi.lib.copo_sim_set_debug(sim._h, None))
    sim.set_state(st_end, env_end)             # (the sampler's buffers belong to this state)
    w = which.tolist()
    measure_sim_kernel.lists = {"scene_steps": int(sum(w)), "register": w[0], "register_with_exactly_evaluated_agents": w[1],
                                "agents_evaluated_exactly": exact_agents, "declined": w[2],
                                "launches_without_a_declined_scene": launches_all_register, "launches": launches}
    return k_s, present / launches


def cruise_actions(obs, gen, speed=0.25):
    """Lane-keeping controller on the observation (columns 2 / 8: heading error / offset in the lane, 3: speed, 10: check
    point to the right): scenes stay populated like those of a trained policy (~94 % of the slots hold an agent; random
    actions crash or leave the road within seconds and leave ~34 %)."""
    psi = torch.asin(((0.5 - obs[..., 2]) * 2).clamp(-1, 1))
    lat = -(obs[..., 8] - 0.5) * 4.5
    aim = (obs[..., 10] - 0.5) * 2
    steer = (-1.5 * psi - 0.25 * lat - 0.8 * aim + 0.02 * torch.randn(psi.shape, device=obs.device, generator=gen)).clamp(-1, 1)
    thr = ((speed - obs[..., 3]) * 8.0).clamp(-1, 1)
    return torch.stack([steer, thr], -1).contiguous()


def measure_sim_kernel_saturated(trainer, scenes=16384, launches=60, policy="cruise"):
    """The same kernel on enough scenes to fill the chip (at 256 scenes a launch is one workgroup per
    compute unit and latency-bound, so the bandwidth fraction is also reported at a saturating scene count).  `cruise`:
    actions of a lane-keeping controller, recorded closed-loop and replayed from the saved state so that the timed loop
    holds nothing but simulator launches; `random`: N(0, 0.1) steering / U(0, 1) throttle as in round 1."""
    from copo_amd.sim import SimConfig, VecSim
    if isinstance(trainer, dict):      # a named configuration (config legs): SimConfig keywords
        sim = VecSim(SimConfig(num_envs=scenes, **trainer), with_info=False)
    else:
        src = trainer.env.sim
        sim = VecSim(SimConfig(map=src.cfg.map, num_envs=scenes, num_agents=src.N, num_lasers=src.cfg.num_lasers,
                               enable_lcf=src.cfg.enable_lcf), with_info=False)
    measure_sim_kernel_saturated.last = dict(O=sim.O, N=sim.N, block=int(getattr(sim, "block", 0) or 0))
    out = sim.reset()
    gen = torch.Generator(device=sim.device).manual_seed(1)
    if policy == "cruise":
        for i in range(250):         # past the first trips: junction traffic in steady state
            out = sim.step(cruise_actions(out["obs"], gen))
        st, env = sim.get_state()
        acts = []
        for i in range(launches):
            a = cruise_actions(out["obs"], gen)
            acts.append(a)
            out = sim.step(a)
        sim.set_state(st, env)
    else:
        acts = [torch.stack([torch.randn(scenes, sim.N, device=sim.device, generator=gen) * 0.1,
                             torch.rand(scenes, sim.N, device=sim.device, generator=gen)], -1).contiguous() for _ in range(4)]
        for i in range(40):
            sim.step(acts[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    present = 0.0
    e0.record()
    for i in range(launches):
        out = sim.step(acts[i % len(acts)])
    e1.record()
    torch.cuda.synchronize()
    present = float(((out["flags"] & 0x41) != 0).sum())
    k_s = e0.elapsed_time(e1) * 1e-3 / launches
    sim.close()
    return k_s, present, scenes * sim.N


def kernel_source_hash():
    """sha1 over the simulator kernel sources: a committed PMC traffic figure is only quoted for the code it was taken on
    (the CODE: `//` comments and white space are left out, so that a reworded comment does not orphan a measurement)."""
    import hashlib
    import re
    h = hashlib.sha1()
    for f in ("sim_kernels.hip", "sim_packed.hip", "sim_device.h", "sim_common.h", "sim_math.h"):
        with open(os.path.join(ROOT, "copo_amd", "csrc", f), "r") as fh:
            text = re.sub(r"//[^\n]*", "", fh.read())
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def measure_learner_step(trainer, launches=200):
    """Mean duration of one fused SGD step (row-pass kernel + weight-gradient/Adam kernel: the two launches that
    take ~65 % of an iteration) with HIP events on the launch stream, and its algorithmic flops: per net
    2*mb*(K*H + H*H + OD*H) forward, 2*mb*(OD*H + H*H) activation gradients, 2*mb*((K+1)*H + (H+1)*H + (H+1)*OD)
    weight gradients."""
    pol = trainer.policy
    fz = pol.fused
    if fz is None or pol._row_sources is None:
        return None
    c, rs = fz.cfg, pol._row_sources
    mb, H = c.mb, c.hidden
    nets = [c.pol] + [c.val[g] for g in range(c.n_value_heads)]
    flops = 0
    for L in nets:
        K, OD = L.in_dim, L.out_dim
        flops += 2 * mb * (K * H + H * H + OD * H) + 2 * mb * (OD * H + H * H) + 2 * mb * ((K + 1) * H + (H + 1) * H + (H + 1) * OD)
    n_plan = int(rs["max_mb"])
    # the way the trainer launches it: captured graphs of SGD_CHAIN steps (eager single launches otherwise)
    chain = getattr(pol, "_sgd_chain", None)
    n_chain = int(getattr(pol, "SGD_CHAIN", 1))
    if chain is not None and chain.graph is not None and n_plan - 1 >= n_chain:
        unit, run = n_chain, chain
    else:
        unit, run = 1, (lambda: fz.step(rs, stats=fz.stats))
    per = max(1, min(launches, n_plan - 1) // unit)
    for _ in range(3):
        rs["k"].zero_()
        for _ in range(min(per, 3)):
            run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    done, total_ms = 0, 0.0
    while done < launches:
        rs["k"].zero_()                      # stay inside the planned minibatch tables
        e0.record()
        for _ in range(per):
            run()
        e1.record()
        torch.cuda.synchronize()
        total_ms += e0.elapsed_time(e1)
        done += per * unit
    return total_ms * 1e-3 / done, flops


def measure_phases(trainer, iters=4):
    """Synchronised wall clock around the phases of a few extra iterations (outside the timed region, single process only):
    rollout + postprocess, PPO epochs, LCF meta passes -- the sample_throughput / learn_throughput split of SURVEY 8d."""
    acc = {}

    def wrap(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize()
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, g)
        return obj, name, f

    saved = [wrap(trainer, "collect", "sample"), wrap(trainer.policy, "run_sgd", "sgd"), wrap(trainer, "train", "iteration")]
    if hasattr(trainer.policy, "run_meta"):
        saved.append(wrap(trainer.policy, "run_meta", "meta"))
        if hasattr(trainer.policy, "meta_rows_early"):      # (the meta phase's row store, queued ahead of the statistics read)
            saved.append(wrap(trainer.policy, "meta_rows_early", "meta"))
    a0 = trainer._counters["num_agent_steps_sampled"]
    for _ in range(iters):
        trainer.train()
    rows = (trainer._counters["num_agent_steps_sampled"] - a0) / iters
    for obj, name, f in saved:
        setattr(obj, name, f)
    ms = {k: v / iters * 1e3 for k, v in acc.items()}
    learn = ms.get("sgd", 0.0) + ms.get("meta", 0.0)
    return {"sample_ms": round(ms.get("sample", 0.0), 3), "sgd_ms": round(ms.get("sgd", 0.0), 3), "meta_ms": round(ms.get("meta", 0.0), 3),
            "iteration_ms": round(ms.get("iteration", 0.0), 3), "agent_steps_per_iter": round(rows, 1),
            "sample_throughput": round(rows / max(ms.get("sample", 0.0), 1e-9) * 1e3, 1),
            "learn_throughput": round(rows / max(learn, 1e-9) * 1e3, 1), "unit": "agent-steps/s (phase alone, synchronised)"}


def _capi_build_info():
    """What the loaded libcopo_hip.so says about itself (copo_build_info: ABI, profiling mask, environment variables it reads)."""
    from copo_amd import _capi
    return _capi.lib.copo_build_info().decode("utf-8", "replace")


def cpu_baseline(num_envs, num_agents, iters=1):
    """The same iteration on the host: C oracle simulator (scalar, 1 thread) + oracle GAE / LCF-mix + the build's
    own torch learner on CPU threads.  Test infrastructure used as the measured CPU port, never as product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(CPU_BASELINE_THREADS, os.cpu_count() or 1))   # tiny GEMMs: more threads only hurt
    from copo_amd.engine import Postprocessing, SampleBatch
    from copo_amd.sim import SimConfig
    from copo_amd.torch_copo import algo_copo as A
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    cfg = A.CoPOConfig()
    cfg.update_from_dict(dict(env=env, device="cpu", use_hip_graphs=False, env_config=dict(num_agents=num_agents)))
    cfg.validate()
    pol = A.CoPOPolicy(cfg.observation_space, cfg.action_space, cfg)
    E, N = num_envs, num_agents
    T = max(1, -(-2000 // E))
    sim = ol.OracleSim(SimConfig(map="intersection", num_envs=E, num_agents=N))
    O = sim.O
    out = sim.reset()
    for _ in range(60):          # steady-state population (the GPU line counts a running population, not a fresh reset)
        a, _, _ = pol.compute_actions(torch.from_numpy(out["obs"].copy()).view(E * N, O))
        out = sim.step(a.view(E, N, 2).clamp(-1, 1).numpy())
    obs = torch.from_numpy(out["obs"].copy())
    agent_steps, t0 = 0, time.perf_counter()
    for _ in range(iters):
        buf = dict(obs=torch.zeros(T, E, N, O), act=torch.zeros(T, E, N, 2), logp=torch.zeros(T, E, N),
                   di=torch.zeros(T, E, N, 4), rew3=np.zeros((3, T, E, N), np.float32), flags=np.zeros((T, E, N), np.uint8),
                   lcf=np.zeros((T, E, N), np.float32))
        for t in range(T):
            a, lp, di = pol.compute_actions(obs.view(E * N, O))
            buf["obs"][t], buf["act"][t], buf["logp"][t], buf["di"][t] = obs, a.view(E, N, 2), lp.view(E, N), di.view(E, N, 4)
            out = sim.step(a.view(E, N, 2).clamp(-1, 1).numpy())
            buf["rew3"][0, t], buf["rew3"][1, t] = out["rew"], out["nei_rew"]
            buf["rew3"][2, t] = out["glob_rew"][:, None]
            buf["flags"][t], buf["lcf"][t] = out["flags"], out["lcf"]
            obs = torch.from_numpy(out["obs"].copy())
        M = E * N
        vals = pol.value_heads_dense(buf["obs"].view(T * M, O)).view(3, T, M).numpy()
        adv, tgt = ol.gae3(buf["rew3"].reshape(3, T, M), vals, buf["flags"].reshape(T, M), pol.gae_gammas(), 0.95)
        valid = (buf["flags"].reshape(-1) & 1) > 0
        mixed, stats, norm, gstd = ol.lcf_mix(adv[0].ravel(), adv[1].ravel(), adv[2].ravel(), buf["lcf"].ravel(), valid)
        mean = stats[1] / stats[0]
        pol._raw_lcf_adv_mean.fill_(mean)
        pol._raw_lcf_adv_std.fill_(max(1e-4, float(np.sqrt(max(stats[2] / stats[0] - mean * mean, 0)))))
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).view(T, E, N)  # noqa: E731
        batch = SampleBatch({
            SampleBatch.OBS: buf["obs"], SampleBatch.ACTIONS: buf["act"], SampleBatch.ACTION_LOGP: buf["logp"],
            SampleBatch.ACTION_DIST_INPUTS: buf["di"], SampleBatch.FLAGS: torch.from_numpy(buf["flags"]),
            Postprocessing.ADVANTAGES: f(adv[0]), SampleBatch.VF_PREDS: f(vals[0]), Postprocessing.VALUE_TARGETS: f(tgt[0]),
            A.NEI_VALUES: f(vals[1]), A.NEI_ADVANTAGE: f(adv[1]), A.NEI_TARGET: f(tgt[1]), A.GLOBAL_VALUES: f(vals[2]),
            A.GLOBAL_TARGET: f(tgt[2]), A.GLOBAL_ADVANTAGES: f(gstd), "normalized_advantages": f(norm)})
        idx = torch.from_numpy(np.nonzero(valid)[0])
        B = int(idx.numel())
        pol.prepare_sgd(batch, T * M, 512)
        pol.run_sgd(idx, B, [B], 512, 5)
        pol.run_meta(idx, B, [B], 512, 5)
        pol.update_old_policy()
        agent_steps += B
    dt = time.perf_counter() - t0
    sim.close()
    used = torch.get_num_threads()
    torch.set_num_threads(prev_threads)
    return agent_steps / dt, dt, agent_steps, used


def cpu_baseline_all_cores(num_envs, num_agents, iters=1):
    """Best-effort host figure for the same iteration (round-3 review: the one-thread simulator + 8-thread learner above is
    not what a 256-thread host can do): the scenes are dealt over C oracle instances on host threads (ctypes releases the GIL),
    policy inference is one batched torch call per env step, and both the number of simulator workers and the learner's thread
    count are CALIBRATED on the real thing first (a few env steps / a few real minibatch steps per candidate) -- more threads are
    not faster for 512-row minibatches.  Test infrastructure used as the measured CPU port."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from copo_amd.engine import Postprocessing, SampleBatch
    from copo_amd.sim import SimConfig
    from copo_amd.torch_copo import algo_copo as A
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    host = os.cpu_count() or 1
    prev_threads = torch.get_num_threads()
    env = get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv))
    cfg = A.CoPOConfig()
    cfg.update_from_dict(dict(env=env, device="cpu", use_hip_graphs=False, env_config=dict(num_agents=num_agents)))
    cfg.validate()
    pol = A.CoPOPolicy(cfg.observation_space, cfg.action_space, cfg)
    E, N = num_envs, num_agents
    T = max(1, -(-2000 // E))
    keys = ("obs", "rew", "nei_rew", "glob_rew", "flags", "lcf")

    class Workers:
        def __init__(self, W):
            self.W = W
            per = [E // W + (1 if k < E % W else 0) for k in range(W)]
            self.offs = np.concatenate([[0], np.cumsum(per)])
            self.sims = [ol.OracleSim(SimConfig(map="intersection", num_envs=per[k], num_agents=N, start_seed=5000 + 131 * k)) for k in range(W)]
            self.pool = ThreadPoolExecutor(W)

        def step(self, act):          # act [E, N, 2] numpy or None (reset)
            def one(k):
                o = self.sims[k].reset() if act is None else self.sims[k].step(act[self.offs[k]:self.offs[k + 1]])
                return {q: o[q].copy() for q in keys}
            outs = list(self.pool.map(one, range(self.W)))
            return {q: np.concatenate([o[q] for o in outs], 0) for q in keys}

        def close(self):
            self.pool.shutdown()
            for sm in self.sims:
                sm.close()

    # simulator workers: a few env steps per candidate
    zero = np.zeros((E, N, 2), np.float32)
    best_w, best_s = 1, float("inf")
    for cand in (8, 16, 32, 64, 128, 256):
        if cand > min(host, E):
            break
        wk = Workers(cand)
        wk.step(None)
        wk.step(zero)
        t0 = time.perf_counter()
        for _ in range(3):
            wk.step(zero)
        dt = time.perf_counter() - t0
        wk.close()
        if dt < best_s:
            best_w, best_s = cand, dt
    wk = Workers(best_w)
    O = wk.sims[0].O
    torch.set_num_threads(min(host, CPU_BASELINE_THREADS))
    out = wk.step(None)
    for _ in range(60):
        a, _, _ = pol.compute_actions(torch.from_numpy(out["obs"]).view(E * N, O))
        out = wk.step(a.view(E, N, 2).clamp(-1, 1).numpy())
    obs = torch.from_numpy(out["obs"])
    best_t = None
    agent_steps, t0 = 0, time.perf_counter()
    t_cal = 0.0
    split = {"sample_s": 0.0, "postprocess_s": 0.0, "sgd_s": 0.0, "meta_s": 0.0}      # where the host's iteration goes (the learner: a
    for _ in range(iters):                                                             # chain of 750 dependent 512-row steps)
        t_iter0 = time.perf_counter()
        t_cal_iter = 0.0
        buf = dict(obs=torch.zeros(T, E, N, O), act=torch.zeros(T, E, N, 2), logp=torch.zeros(T, E, N), di=torch.zeros(T, E, N, 4),
                   rew3=np.zeros((3, T, E, N), np.float32), flags=np.zeros((T, E, N), np.uint8), lcf=np.zeros((T, E, N), np.float32))
        for t in range(T):
            a, lp, di = pol.compute_actions(obs.view(E * N, O))
            buf["obs"][t], buf["act"][t], buf["logp"][t], buf["di"][t] = obs, a.view(E, N, 2), lp.view(E, N), di.view(E, N, 4)
            out = wk.step(a.view(E, N, 2).clamp(-1, 1).numpy())
            buf["rew3"][0, t], buf["rew3"][1, t] = out["rew"], out["nei_rew"]
            buf["rew3"][2, t] = out["glob_rew"][:, None]
            buf["flags"][t], buf["lcf"][t] = out["flags"], out["lcf"]
            obs = torch.from_numpy(out["obs"])
        t_sample_end = time.perf_counter()
        M = E * N
        vals = pol.value_heads_dense(buf["obs"].view(T * M, O)).view(3, T, M).numpy()
        adv, tgt = ol.gae3(buf["rew3"].reshape(3, T, M), vals, buf["flags"].reshape(T, M), pol.gae_gammas(), 0.95)
        valid = (buf["flags"].reshape(-1) & 1) > 0
        mixed, stats, norm, gstd = ol.lcf_mix(adv[0].ravel(), adv[1].ravel(), adv[2].ravel(), buf["lcf"].ravel(), valid)
        mean = stats[1] / stats[0]
        pol._raw_lcf_adv_mean.fill_(mean)
        pol._raw_lcf_adv_std.fill_(max(1e-4, float(np.sqrt(max(stats[2] / stats[0] - mean * mean, 0)))))
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).view(T, E, N)  # noqa: E731
        batch = SampleBatch({
            SampleBatch.OBS: buf["obs"], SampleBatch.ACTIONS: buf["act"], SampleBatch.ACTION_LOGP: buf["logp"],
            SampleBatch.ACTION_DIST_INPUTS: buf["di"], SampleBatch.FLAGS: torch.from_numpy(buf["flags"]),
            Postprocessing.ADVANTAGES: f(adv[0]), SampleBatch.VF_PREDS: f(vals[0]), Postprocessing.VALUE_TARGETS: f(tgt[0]),
            A.NEI_VALUES: f(vals[1]), A.NEI_ADVANTAGE: f(adv[1]), A.NEI_TARGET: f(tgt[1]), A.GLOBAL_VALUES: f(vals[2]),
            A.GLOBAL_TARGET: f(tgt[2]), A.GLOBAL_ADVANTAGES: f(gstd), "normalized_advantages": f(norm)})
        idx = torch.from_numpy(np.nonzero(valid)[0])
        B = int(idx.numel())
        pol.prepare_sgd(batch, T * M, 512)
        if best_t is None:
            # learner threads: a few REAL minibatch steps per candidate (loss, backward, Adam, the row gathers), outside the clock
            tc = time.perf_counter()
            pol._ensure_flat_grads()
            pol.plan_epoch(idx, B, [B], 512)
            # (the 8 threads of the plain variant unless another count is clearly -- 15 % -- faster over six real steps: on the
            #  256-thread hosts of the pool 32 threads won a three-step calibration and lost the whole learner phase by 2x)
            times = {}
            for cand in (CPU_BASELINE_THREADS, 4, 16, 32, 64):      # (round-4 review: beyond 4 .. 16; on the 256-thread hosts of the pool 64 threads take 39 ms
                                                                  #  per step against 5.8 at 8, 128 threads 140 ms, 256 threads 14 s -- profiles/r05_bench_line.json -- so the sweep stops at 64)
                if cand > host:
                    continue
                torch.set_num_threads(cand)
                pol._row_sources["k"].zero_()
                for _ in range(3):
                    pol._sgd_step_local()
                ts = time.perf_counter()
                for _ in range(10):
                    pol._sgd_step_local()
                times[cand] = (time.perf_counter() - ts) / 10
            split["learner_thread_sweep_ms_per_step"] = {str(k): round(v * 1e3, 2) for k, v in sorted(times.items())}
            best_t = min(CPU_BASELINE_THREADS, host)
            for cand, dts in times.items():
                if dts < 0.85 * times.get(best_t, float("inf")):
                    best_t = cand
            t_cal_iter = time.perf_counter() - tc
            t_cal += t_cal_iter
        torch.set_num_threads(best_t)
        split["sample_s"] += t_sample_end - t_iter0
        t1 = time.perf_counter()
        split["postprocess_s"] += t1 - t_sample_end - t_cal_iter
        pol.run_sgd(idx, B, [B], 512, 5)
        t2 = time.perf_counter()
        pol.run_meta(idx, B, [B], 512, 5)
        pol.update_old_policy()
        t3 = time.perf_counter()
        split["sgd_s"] += t2 - t1
        split["meta_s"] += t3 - t2
        agent_steps += B
    dt = time.perf_counter() - t0 - t_cal
    wk.close()
    torch.set_num_threads(prev_threads)
    split = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in split.items()}
    return agent_steps / dt, dt, agent_steps, best_w, best_t, split


def live_reference(seconds=20.0):
    """if the reference's own stack is importable on this box (MetaDrive + Ray), time its CPU-runnable
    configuration C1 (IPPO, Intersection, 4 agents, 1 env, local mode) for a bounded sample; else say so.  Nothing of
    /root/reference is read: the packages would have to be installed in the image."""
    try:
        import metadrive  # noqa: F401
        import ray  # noqa: F401
    except Exception as e:      # noqa: BLE001
        return {"live_reference": "unavailable", "why": "import metadrive, ray: %s" % type(e).__name__}
    try:
        from metadrive.envs.marl_envs import MultiAgentIntersectionEnv as RefEnv
        env = RefEnv(dict(num_agents=4))
        obs = env.reset()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            obs, r, d, i = env.step({k: env.action_space[k].sample() for k in obs})
            n += len(r)
            if d.get("__all__"):
                obs = env.reset()
        dt = time.perf_counter() - t0
        env.close()
        return {"live_reference": "MetaDrive MultiAgentIntersectionEnv, 4 agents, random actions (simulator half of C1)",
                "value": round(n / dt, 1), "unit": "agent-steps/s", "seconds": round(dt, 1)}
    except Exception as e:      # noqa: BLE001
        return {"live_reference": "unavailable", "why": "%s: %s" % (type(e).__name__, e)}


def cpu_sim_only(num_agents, threads, scenes_per_thread=16, steps=40):
    """Simulator half alone on the host (SURVEY section 8d iii: single thread and all host threads): the scalar C oracle,
    one simulator instance per thread (ctypes releases the GIL), lane-keeping actions computed outside the timed calls."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from copo_amd.sim import SimConfig
    sims = [ol.OracleSim(SimConfig(map="intersection", num_envs=scenes_per_thread, num_agents=num_agents, start_seed=5000 + 97 * k))
            for k in range(threads)]
    rng = np.random.RandomState(0)
    acts = np.stack([rng.normal(0, 0.1, (scenes_per_thread, num_agents)), rng.uniform(0, 1, (scenes_per_thread, num_agents))], -1).astype(np.float32)
    counts = [0] * threads
    for sm in sims:
        sm.reset()
        for _ in range(20):
            sm.step(acts)

    def work(k):
        n = 0
        for _ in range(steps):
            o = sims[k].step(acts)
            n += int(((o["flags"] & 0x41) != 0).sum())
        counts[k] = n
    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    for sm in sims:
        sm.close()
    return sum(counts) / dt


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, cpu_only=False):
    """`python bench.py --gpus N` as ONE command: start N copies of this script, one per GPU, with the env rendezvous the
    driver's `torch.distributed.run` form uses (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); their stdout
    and stderr pass through (rank 0 prints the JSON line).  Fails -- never runs on fewer GPUs -- when the node has fewer
    than N devices.  COPO_BENCH_SHARE_DEVICE=1 (tests on a one-GPU box): every rank uses device 0 and the collectives go
    over gloo, which unlike RCCL accepts several ranks on one device.  Returns the exit code."""
    import subprocess
    share = os.environ.get("COPO_BENCH_SHARE_DEVICE", "0") == "1"
    if not cpu_only:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < (1 if share else n):
            print("bench.py --gpus %d: this node answers with %d GPU(s); refusing to run on fewer" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if (share or cpu_only) else str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if share or cpu_only:
            env["COPO_DIST_BACKEND"] = "gloo"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                try:
                    r = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                alive.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in alive:      # a rank died: its partners would wait in a collective for ever -- stop exactly them
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def rendezvous_only(D, args):
    """`--rendezvous-only`: the ranks of `--gpus N` meet over gloo on the host (no GPU touched), add up their rank numbers and
    rank 0 prints the head of the result line -- the launcher and the env rendezvous checked where there is no GPU."""
    import torch.distributed as td
    os.environ["COPO_DIST_BACKEND"] = "gloo"
    rank, local_rank, world = D.init_from_env("cpu")
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        td.all_reduce(t)
        td.barrier()
    assert float(t.item()) == world * (world + 1) / 2
    if rank == 0:
        print(json.dumps({"metric": "agent-env-steps/sec (sim+learn), Intersection 40-agent", "value": None, "n_gpus": world,
                          "rendezvous": "ok", "config": {"parallelism": "dp%d" % world}}), flush=True)
    D.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--num-envs", type=int, default=256, help="scenes per GPU (BASELINE configs[1]: 256)")
    ap.add_argument("--num-agents", type=int, default=40)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--mb-per-rank", type=int, default=512,
                    help="rows per SGD minibatch and rank (512 = the reference's sgd_minibatch_size; 1024 halves the data-parallel steps of an "
                         "iteration at 1.73 x the step time, DESIGN.md section 6 -- recorded in config.sgd_minibatch_size_per_rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--untrained", action="store_true",
                    help="random-init policy net instead of the reference's published population (the scenes then empty during the run)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="warm-up iterations, then ONLY the live simulator-kernel measurement (recorded replay on the trainer's scenes) "
                         "and one small JSON line: the command scripts/sim_traffic.sh runs under rocprofv3 --pmc")
    ap.add_argument("--config-leg", default=None, choices=["c3", "c4", "c5"],
                    help="ONLY the simulator step kernel on another BASELINE configuration (c3 Roundabout 40 slots, c4 Tollgate 40 slots / "
                         "O = 156, c5 ParkingLot 10 slots / 240 beams / O = 260): populated scenes (lane-keeping controller, recorded replay) at "
                         "the configuration's own scene count per GPU and at 16 384 scenes; one JSON line each (scripts/prof_sim_round.sh)")
    ap.add_argument("--leg-scenes", type=int, default=0, help="with --config-leg: only this scene count (profiling: one size per process)")
    ap.add_argument("--saturated-only", action="store_true",
                    help="ONLY the saturated simulator-kernel measurement (16 384 populated scenes, recorded replay) and one small "
                         "JSON line: the command scripts/prof_sim_r03.sh profiles")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks of --gpus N, let them meet over gloo on the host and print the head of the line: "
                         "checks the launcher where there is no GPU")
    args = ap.parse_args()
    global MB_PER_RANK
    MB_PER_RANK = int(args.mb_per_rank)

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.rendezvous_only):
        # one command, N ranks: this process only launches them (the driver's `torch.distributed.run` form arrives with
        # WORLD_SIZE set and skips this)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], cpu_only=args.rendezvous_only))
    if os.environ.get("COPO_BENCH_TRACE_S"):       # diagnostics: where is every thread after so many seconds (then exit)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["COPO_BENCH_TRACE_S"]), exit=True)
    from copo_amd import dist as D
    import torch.distributed as td
    if args.rendezvous_only:
        rendezvous_only(D, args)
        return
    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with torch.distributed.run --nproc-per-node %d "
                 "(or without WORLD_SIZE in the environment: it then starts its own ranks)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d wants GPU %d, this node answers with %d device(s) -- not falling back to fewer GPUs"
                 % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    torch.cuda.set_device(local_rank)
    if args.config_leg:
        legs = {"c3": (dict(map="roundabout", num_agents=40), (128, 1024, 16384)),
                "c4": (dict(map="tollgate", num_agents=40), (512, 16384)),
                "c5": (dict(map="parkinglot", num_agents=10, num_lasers=240), (4096, 16384))}
        kw, sizes = legs[args.config_leg]
        for scenes in (sizes if args.leg_scenes <= 0 else (args.leg_scenes,)):
            k_s, present, slots = measure_sim_kernel_saturated(kw, scenes=scenes, launches=60, policy="cruise")
            O = measure_sim_kernel_saturated.last["O"]
            bpu = 202 + 4 * O
            print(json.dumps({"kernel": "copo::sim_step_kernel / sim_step_packed_kernel", "config": args.config_leg, "sim": kw, "scenes": scenes,
                              "launches": 60, "us_per_launch": round(k_s * 1e6, 2), "present_slots": round(present), "slots": slots,
                              "obs_dim": O, "bytes_per_unit": bpu, "achieved_GBps": round(present * bpu / k_s * 1e-9, 1),
                              "frac": round(present * bpu / k_s * 1e-9 / HBM_PEAK_GBPS, 4)}), flush=True)
        D.shutdown()
        return
    trainer = make_trainer(args.num_envs, args.num_agents, graphs=not args.no_graphs, pretrained=not args.untrained)
    if args.saturated_only:
        sat_s, sat_present, sat_slots = measure_sim_kernel_saturated(trainer, policy="cruise")
        b = 202 + 4 * trainer.env.sim.O
        print(json.dumps({"kernel": "copo::sim_step_kernel", "scenes": sat_slots // trainer.env.sim.N, "launches": 60,
                          "us_per_launch": round(sat_s * 1e6, 2), "present_slots": round(sat_present), "bytes_per_unit": b,
                          "frac": round(sat_present * b / sat_s * 1e-9 / HBM_PEAK_GBPS, 4)}), flush=True)
        trainer.stop()
        D.shutdown()
        return
    warm = max(args.warmup, 4 if not args.no_graphs else 0)   # eager warm-ups + graph capture happen untimed
    for _ in range(warm):
        trainer.train()
    if args.roofline_only:
        k_s, present = measure_sim_kernel(trainer)
        sim = trainer.env.sim
        print(json.dumps({"kernel": "copo::sim_step_kernel", "scenes": sim.E, "slots": sim.N, "launches": 200,
                          "us_per_launch": round(k_s * 1e6, 2), "units_per_launch": round(present, 1),
                          "bytes_per_unit": 202 + 4 * sim.O, "neighbour_lists": measure_sim_kernel.lists,
                          "kernel_source_sha1": kernel_source_hash()}), flush=True)
        trainer.stop()
        D.shutdown()
        return
    D.barrier()
    torch.cuda.synchronize()
    a0 = trainer._counters["num_agent_steps_sampled"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = trainer.train()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    D.all_reduce_max_(tmax)
    dt = float(tmax.item())
    agent_steps = trainer._counters["num_agent_steps_sampled"] - a0      # already summed over ranks
    value = agent_steps / dt

    coll = None
    if world > 1:
        # what the data-parallel step adds: one all-reduce of the flat gradient bucket per SGD minibatch (every rank takes
        # part; HIP events on this rank's stream, barrier-separated from the timed region above)
        fz = trainer.policy.fused
        n = int(fz.flat.numel) if fz is not None else sum(p.numel() for p in trainer.policy.model.parameters())
        buf = torch.zeros(n, device="cuda")
        for _ in range(5):
            td.all_reduce(buf)
        torch.cuda.synchronize()
        D.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            td.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        coll = {"allreduce_grad_us": round(e0.elapsed_time(e1) * 1e3 / 50, 2), "bucket_bytes": 4 * n,
                "backend": td.get_backend(), "per": "SGD minibatch (one per optimizer step)",
                "used_by": "the RCCL loop only (dp_step = 'rccl'); the tile exchange sums inside the weight-gradient kernel"}

    # who ran where and HOW the data-parallel step summed its gradients, from EVERY rank (a SCALE run that fell back to the RCCL loop,
    # or ranks that disagree, or two ranks on one device must be visible in the line, not inferred)
    pr = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "local_rank": local_rank, "device": int(torch.cuda.current_device()), "name": pr.name,
            "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
            "dp_step": getattr(trainer.policy, "_dp_mode", None), "dp_step_reason": getattr(trainer.policy, "dp_reason", None)}
    ranks_info = [mine]
    if world > 1:
        ranks_info = [None] * world
        td.all_gather_object(ranks_info, mine)
    data_parallel = {"world": world, "backend": td.get_backend() if D.is_dist() else None,
                     "dp_step": mine["dp_step"], "dp_step_reason": mine["dp_step_reason"],
                     "dp_step_same_on_all_ranks": len({r["dp_step"] for r in ranks_info}) == 1,
                     "distinct_devices": len({r["pci"] for r in ranks_info}), "ranks": ranks_info}

    line = None
    if rank == 0:
        # Side measurements refer to the state the timed region STARTED from: training moves the policy (the value nets start
        # untrained, their advantages perturb the loaded driver) and with it the population, so with one process a second trainer
        # is built with the same seed and brought to the same point (`warm` iterations); several ranks keep the trainer they
        # have (building one on rank 0 alone would leave the other ranks' collectives without a partner)
        side = trainer
        if world == 1:
            side = make_trainer(args.num_envs, args.num_agents, graphs=not args.no_graphs, pretrained=not args.untrained)
            for _ in range(warm):
                side.train()
            torch.cuda.synchronize()
        sim = side.env.sim
        k_s, present = measure_sim_kernel(side)
        bytes_per_unit = 202 + 4 * sim.O
        achieved = present * bytes_per_unit / k_s * 1e-9
        # HBM bytes per launch from the PMC counters are taken in a separate rocprofv3 pass (scripts/sim_traffic.sh writes
        # profiles/sim_traffic.json); quoted only if that pass ran on exactly this kernel source
        traffic, traffic_units = None, None
        tfile = os.path.join(ROOT, "profiles", "sim_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("kernel_source_sha1") == kernel_source_hash():
                traffic = tj.get("bytes_per_launch")
                traffic_units = tj.get("units_per_launch")        # present slots of the launches the counters saw
        # VALU roofline of the saturated launch (the kernel is bound by the vector ALUs' issue rate, not by HBM): wave-level VALU
        # instructions per launch from the PMC pass of scripts/prof_sim_round.sh (profiles/sim_valu.json, same source-hash rule)
        valu = None
        vfile = os.path.join(ROOT, "profiles", "sim_valu.json")
        if os.path.exists(vfile):
            vj = json.load(open(vfile))
            if vj.get("kernel_source_sha1") == kernel_source_hash() and vj.get("valu_wave_instructions_per_launch"):
                valu = vj
        learner = None     # (after `phases`: it walks the trainer's SGD plan tables)
        sat_s, sat_present, sat_slots = measure_sim_kernel_saturated(side, policy="cruise")
        rnd_s, rnd_present, _ = measure_sim_kernel_saturated(side, policy="random")
        line = {
            "metric": "agent-env-steps/sec (sim+learn), Intersection 40-agent", "value": round(value, 1),
            "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if args.untrained else "synthetic (seeded HIP scenes; policy net initialised from the reference's published "
                    "CoPO Intersection population so that the scenes stay populated; value nets / LCF random-init)",
            "config": {"workload": "CoPO Intersection, %d agent slots x %d scenes per GPU, fp32 (BASELINE configs[1])"
                                   % (sim.N, sim.E), "rollout_steps": trainer.sampler.T,
                       "sgd_minibatch_size_per_rank": MB_PER_RANK, "num_sgd_iter": 5, "lcf_num_iters": 5,
                       "parallelism": "dp%d" % world, "hip_graphs": not args.no_graphs,
                       "agent_steps_per_iter": round(agent_steps / args.steps / world, 1),
                       "policy_init": "random" if args.untrained else "reference population copo_inter (tests/golden, weights as data)",
                       # tuning knobs the product path reads from the environment (they skip no work): recorded, not hidden
                       "knobs": {"COPO_SGD_CHAIN": int(getattr(trainer.policy, "SGD_CHAIN", 0)),
                                 "COPO_FORCE_DIST": os.environ.get("COPO_FORCE_DIST", "0"),
                                 "COPO_DP_EXCHANGE": os.environ.get("COPO_DP_EXCHANGE", "auto")},
                       # how the data-parallel SGD step sums its gradients: "tile" = inside the weight-gradient kernel
                       # (peer stores over xGMI), "rccl" = all-reduce + flat Adam; None = one process
                       "dp_step": getattr(trainer.policy, "_dp_mode", None),
                       "dp_step_reason": getattr(trainer.policy, "dp_reason", None)},
            "data_parallel": data_parallel,
            "roofline": {"bound": "hbm", "kernel": "copo::sim_step_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": traffic, "traffic_units_per_launch": traffic_units,
                         "us_per_launch": round(k_s * 1e6, 2),
                         "units_per_launch": round(present, 1), "bytes_per_unit": bytes_per_unit,
                         "neighbour_lists": measure_sim_kernel.lists,
                         "state": "the trainer's own scenes and policy: actions of 200 env steps recorded closed-loop, scenes put "
                                  "back, actions replayed (timed loop = simulator launches only); `traffic` = PMC FETCH_SIZE + "
                                  "WRITE_SIZE per launch from rocprofv3 passes over `bench.py --roofline-only` (the same replay), "
                                  "with the present slots those launches had",
                         "saturated": {"scenes": sat_slots // sim.N, "actions": "lane-keeping controller (recorded, replayed)",
                                       "us_per_launch": round(sat_s * 1e6, 1),
                                       "present_slots": round(sat_present), "slots_stepped": sat_slots,
                                       "achieved": round(sat_present * bytes_per_unit / sat_s * 1e-9, 1),
                                       "frac": round(sat_present * bytes_per_unit / sat_s * 1e-9 / HBM_PEAK_GBPS, 4),
                                       "random_actions": {"us_per_launch": round(rnd_s * 1e6, 1), "present_slots": round(rnd_present),
                                                          "frac": round(rnd_present * bytes_per_unit / rnd_s * 1e-9 / HBM_PEAK_GBPS, 4)}},
                         "timing": "HIP events around back-to-back launches on the launch stream; rocprofv3 --kernel-trace reports "
                                   "~10 % longer per-kernel durations at 16 384 scenes because traced dispatches do not overlap "
                                   "the previous launch's drain with their own ramp-up",
                         "library_build": _capi_build_info()},
        }
        if valu is not None:
            # peak: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); issued: this launch's
            # instructions / its duration measured live above
            issued = valu["valu_wave_instructions_per_launch"] / sat_s * 1e-9
            line["roofline"]["valu"] = {
                "bound": "valu-issue", "issued": round(issued, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                "frac": round(issued / VALU_PEAK_GINST, 4),
                "instructions_per_launch": round(valu["valu_wave_instructions_per_launch"]),
                "instructions_per_present_slot": round(valu["valu_wave_instructions_per_launch"] * 64 / max(sat_present, 1.0), 1),
                "what": "copo::sim_step_kernel on %d populated scenes: SQ_INSTS_VALU per launch (profiles/sim_valu.json, rocprofv3 --pmc on "
                        "`bench.py --saturated-only`) / the launch time measured here; `instructions_per_present_slot` counts lane-"
                        "instructions (x 64).  The HBM fraction of this launch can only rise by deleting instructions" % valu["scenes"]}
        if world == 1:
            line["phases"] = measure_phases(side)
        learner = measure_learner_step(side)
        if learner is not None:
            l_s, l_flops = learner
            line["learner_roofline"] = {
                "bound": "mfma", "kernel": "copo::rowpass_kernel + copo::wgrad_adam_kernel (one 512-row SGD step, 4 nets)",
                "achieved": round(l_flops / l_s * 1e-12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(l_flops / l_s * 1e-12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "us_per_step": round(l_s * 1e6, 2), "flops_per_step": int(l_flops)}
        if coll is not None:
            line["config"]["collective"] = coll
        if side is not trainer:
            side.stop()
        if world == 1 and not args.no_cpu_baseline:
            v, cdt, n, used = cpu_baseline(args.num_envs, args.num_agents)
            host = os.cpu_count() or 1
            va, adt, an, aw, at, asplit = cpu_baseline_all_cores(args.num_envs, args.num_agents)
            sim1 = cpu_sim_only(args.num_agents, 1)
            simn = cpu_sim_only(args.num_agents, host)
            line["cpu_baseline"] = {"value": round(v, 1), "unit": "agent-steps/s", "cores": used, "kind": "port",
                                    "sample": "1 iteration of the same workload on a steady-state population (%d agent-steps, "
                                              "%.1f s): scalar C oracle simulator on 1 thread + the build's torch learner on %d "
                                              "CPU threads (host has %d)" % (n, cdt, used, host),
                                    "all_cores": {"value": round(va, 1), "unit": "agent-steps/s", "sim_threads": aw, "learner_threads": at,
                                                  "sample": "the same iteration (%d agent-steps, %.1f s): one C oracle instance per host thread "
                                                            "for the scenes, batched torch inference, torch learner at its fastest thread "
                                                            "count for 512-row minibatches (candidates 4, 8, 16, 32, 64 threads, 8 kept unless another is 15 %% faster; "
                                                            "per-candidate times in split.learner_thread_sweep_ms_per_step)" % (an, adt),
                                                  "split": asplit},
                                    "sim_only": {"threads_1": round(sim1, 1), "threads_%d" % host: round(simn, 1),
                                                 "unit": "agent-steps/s, simulator half alone (C oracle, one instance per thread)"},
                                    "live_reference": live_reference(),
                                    "recorded_reference": {"value": 1500.0, "unit": "agent-steps/s",
                                                           "what": "the reference itself: MetaDrive + RLlib on 4 rollout workers, 60 env-steps/s "
                                                                   "(BASELINE.md; cited, not re-measured: MetaDrive and Ray are not installable here)"}}
    trainer.stop()
    if rank == 0:
        # libraries that write through C stdio (RCCL's start-up banner) sit in a buffer when stdout is a pipe and would
        # otherwise be flushed AFTER the result at exit: push them out first so that the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if world == 1:
        D.shutdown()      # single process: quiet exit; several ranks leave the group the way torchrun expects (at exit)


if __name__ == "__main__":
    main()

"""The host loop that replaces RLlib's rollout workers, sampler and SGD driver (SURVEY.md layer L1/L3).

  VecSampler     T-step rollouts of E x N agent slots: policy inference (torch-ROCm) + `copo_sim_step`
                 writing straight into the time-major rollout buffers; the whole T-step loop is one hipGraph.
  PPOPolicyBase  model + Adam + KL coefficient + the static-shape minibatch SGD step (optionally a hipGraph),
                 with the data-parallel gradient exchange of `copo_amd.dist`.
  VecTrainer     `training_step()` = sample -> postprocess -> advantage statistics -> SGD epochs -> KL update,
                 the control flow of RLlib PPO / the reference's `training_step` (algo_copo.py:516-661).

Reference call sites being replaced: `synchronous_parallel_sample` (algo_copo.py:519-525), `train_one_step` /
`multi_gpu_train_one_step` (:555-558), `minibatches` (:585), `update_kl` (:631-632).
"""
import math
import os
import time
from collections import defaultdict

import numpy as np
import torch

from . import dist as D
from .engine import (LEARNER_STATS_KEY, NUM_AGENT_STEPS_SAMPLED, NUM_ENV_STEPS_SAMPLED, Postprocessing, SampleBatch,
                     TorchDiagGaussian)

F_ACTED, F_DONE, F_ARRIVE, F_CRASH, F_OUT, F_MAXSTEP, F_SPAWNED, F_ENV_RESET = (1 << i for i in range(8))


def resolve_device(cfg_device=None):
    if cfg_device is not None:
        return torch.device(cfg_device)
    if torch.cuda.is_available():
        return torch.device("cuda", D.env_world()[1])
    return torch.device("cpu")


class GraphedCallable:
    """Run `fn()` eagerly, or (CUDA + enabled) capture it once into a hipGraph after `warmup` eager calls and
    replay it afterwards.  `fn` must only touch static tensors and must not synchronise."""

    def __init__(self, fn, enabled, warmup=2):
        self.fn, self.enabled, self.warmup = fn, bool(enabled), warmup
        self.calls, self.graph = 0, None

    def __call__(self):
        if not self.enabled:
            return self.fn()
        if self.graph is not None:
            self.graph.replay()
            return
        self.calls += 1
        if self.calls <= self.warmup:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.fn()
            torch.cuda.current_stream().wait_stream(s)
            return
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread polls events while we capture: thread-local capture mode keeps
        # another thread's runtime calls from invalidating this thread's capture
        with torch.cuda.graph(g, capture_error_mode="thread_local" if D.is_dist() else "global"):
            self.fn()
        self.graph = g
        g.replay()

    def reset(self):
        self.calls, self.graph = 0, None


_side_candidates = []      # keeps the probed pool streams alive (a stream object's identity is its slot in torch's pool)


def concurrent_stream(device, tries=8):
    """A stream whose kernels really run UNDER the current stream's.  HIP maps streams onto a handful of hardware queues round-robin
    and torch hands out pool streams round-robin, so a plain second stream may sit on the current stream's queue -- whatever is
    queued there runs BETWEEN the main stream's kernels instead of under them (round 6: with other captured-chain lengths the
    sequential LCF kernels landed on the main queue and the meta passes took 5.5 instead of 3.5 ms).  Probe: a ~1 ms spin on the
    current stream, a tiny kernel on the candidate behind an event recorded BEFORE the spin; the candidate is concurrent if that kernel
    finished well before the spin did.  (High-priority streams do have queues of their own, but measured 3x slower iterations.)"""
    main = torch.cuda.current_stream(device)
    best = None
    probe = torch.zeros(64, device=device)
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device)
        _side_candidates.append(cand)
        if best is None:
            best = cand
        with torch.cuda.stream(cand):
            probe.add_(1.0)            # (first use of a stream creates its queue: ~0.3 ms, not part of the measurement)
        torch.cuda.synchronize(device)
        e0, e_main, e_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        torch.cuda._sleep(2_000_000)
        e_main.record(main)
        with torch.cuda.stream(cand):
            cand.wait_event(e0)
            probe.add_(1.0)            # (a real dispatch: an event on an idle stream completes without touching the queue)
            e_side.record(cand)
        torch.cuda.synchronize(device)
        if e0.elapsed_time(e_side) < 0.5 * e0.elapsed_time(e_main):
            return cand
    return best          # (no candidate overlapped, e.g. GPU_MAX_HW_QUEUES=1: correct either way, only slower)


# ----------------------------------------------------------------------------------------------------
# sampler
# ----------------------------------------------------------------------------------------------------
class VecSampler:
    """Time-major rollout buffers + the rollout loop.  Row (t, e, n) holds obs_t, the action sampled for it, and
    the reward / flags / neighbour lists / LCF produced by the step that executed that action."""

    def __init__(self, vec_env, policy, T, use_graph=True, stagger_episodes=False):
        self.env, self.sim, self.policy, self.T = vec_env, vec_env.sim, policy, int(T)
        # E scenes stepped in lockstep all sit in the same phase of their episodes (`horizon` steps of dense traffic, then
        # the drain): with stagger_episodes the FIRST episode of scene e starts at env step (e * horizon) // E instead of 0,
        # so every fragment sees every phase (RLlib's 200-step fragments of 8 envs cycle through the phases every ~7
        # iterations; 8-step fragments of 256 scenes take ~160)
        self.stagger_episodes = bool(stagger_episodes)
        sim = self.sim
        E, N, O, K, dev = sim.E, sim.N, sim.O, sim.K, sim.device
        self.E, self.N, self.O, self.K, self.device = E, N, O, K, dev
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        T = self.T
        z = lambda *s, dtype=f32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self.obs = z(T + 1, E, N, O)
        self.actions, self.eps, self.clipped = z(T, E, N, 2), z(T, E, N, 2), z(T, E, N, 2)
        self.logp, self.dist_inputs = z(T, E, N), z(T, E, N, 4)
        self.rew3 = z(3, T, E, N)            # native / neighbourhood / global reward heads
        self.glob = z(T, E)
        self.flags = z(T, E, N, dtype=u8)
        self.nbr_idx = z(T, E, N, K, dtype=i32)     # (distances are not needed by the learner: no nbr_dist output)
        self.nbr_cnt, self.mf_cnt = z(T, E, N, dtype=i32), z(T, E, N, dtype=i32)
        self.lcf, self.agent_id = z(T, E, N), z(T, E, N, dtype=i32)
        self.info = z(T, E, N, 8)
        self._outs = []
        for t in range(T):
            self._outs.append(sim.make_step_out(dict(
                obs=self.obs[t + 1], rew=self.rew3[0, t], nei_rew=self.rew3[1, t], glob_rew=self.glob[t],
                flags=self.flags[t], nbr_idx=self.nbr_idx[t], nbr_cnt=self.nbr_cnt[t], mf_cnt=self.mf_cnt[t],
                lcf=self.lcf[t], info=self.info[t], agent_id=self.agent_id[t])))
        self._reset_out = sim.make_step_out(dict(obs=self.obs[0]))
        self._started = False
        self._loop = GraphedCallable(self._rollout, use_graph and dev.type == "cuda")
        # VecSim.set_block / set_chunk invalidate the captured rollout.  Held weakly: a sampler that was dropped (trainers rebuilt
        # on the same simulator) must not be kept alive -- with its graphs and buffers -- by the simulator's callback list
        import weakref
        sim.on_shape_change.append(weakref.WeakMethod(self._loop.reset))
        self.env_steps_total = 0

    def reset(self, seeds=None):
        import ctypes as C
        sim = self.sim
        if seeds is None:
            seeds = np.arange(sim.E, dtype=np.uint64) + np.uint64(sim.cfg.start_seed + 1000003 * D.rank())
        seeds = np.ascontiguousarray(seeds, np.uint64)
        sim._capi.check(sim._capi.lib.copo_sim_reset(sim._h, seeds.ctypes.data, C.byref(self._reset_out), sim._stream()))
        if self.stagger_episodes and sim.E > 1:
            st, env = sim.get_state()
            env[:, 0] = (torch.arange(sim.E, device=env.device, dtype=torch.int64) * int(sim.cfg.horizon) // sim.E).to(torch.int32)
            sim.set_state(st, env)
        torch.cuda.current_stream(self.device).synchronize()
        self._started = True

    def _rollout(self):
        import ctypes as C
        sim, pol = self.sim, self.policy
        EN = self.E * self.N
        lib, h, stream = sim._capi.lib, sim._h, sim._stream()
        fz = pol.fused if (pol.fused is not None and pol.fused.can_forward and pol.config.get("use_fused_inference", True)) else None
        for t in range(self.T):
            if fz is not None:        # one kernel: both layers, head, sampling, log-probability, clipping
                fz.act(self.obs[t].view(EN, self.O), self.eps[t].view(EN, 2), self.actions[t], self.logp[t],
                       self.dist_inputs[t], self.clipped[t])
            else:
                a, lp, di = pol.compute_actions(self.obs[t].view(EN, self.O), self.eps[t].view(EN, 2))
                self.actions[t].view(EN, 2).copy_(a)
                self.logp[t].view(EN).copy_(lp)
                self.dist_inputs[t].view(EN, 4).copy_(di)
                torch.clamp(a.view(self.E, self.N, 2), -1.0, 1.0, out=self.clipped[t])
            sim._capi.check(lib.copo_sim_step(h, self.clipped[t].data_ptr(), C.byref(self._outs[t]), stream))

    def sample(self):
        """One fragment of T env steps on every scene; returns the dense SampleBatch (views, no copies)."""
        if not self._started:
            self.reset()
        else:
            self.obs[0].copy_(self.obs[self.T])
        self.eps.normal_()
        self.sim.flush()
        if self.policy.fused is not None:
            self.policy.fused.sync_mirror()      # the captured rollout reads the transposed weight mirror
        self._loop()
        self.env_steps_total += self.T * self.E
        self.rew3[2].copy_(self.glob.unsqueeze(-1).expand(self.T, self.E, self.N))
        return SampleBatch({
            SampleBatch.OBS: self.obs[:self.T], SampleBatch.ACTIONS: self.actions, SampleBatch.ACTION_LOGP: self.logp,
            SampleBatch.ACTION_DIST_INPUTS: self.dist_inputs, SampleBatch.REWARDS: self.rew3[0],
            "nei_rewards": self.rew3[1], "global_rewards": self.rew3[2], "rew3": self.rew3,
            SampleBatch.FLAGS: self.flags, "nbr_idx": self.nbr_idx, "nbr_cnt": self.nbr_cnt, "mf_cnt": self.mf_cnt,
            "step_lcf": self.lcf, "infos": self.info, "agent_id": self.agent_id,
            "_next_obs_last": self.obs[self.T],      # observation after the fragment's last step (bootstrap of truncated trajectories)
        })


# ----------------------------------------------------------------------------------------------------
# policy base
# ----------------------------------------------------------------------------------------------------
class PPOPolicyBase:
    """Shared-policy PPO learner.  Subclasses provide `model_class`, `loss`, `postprocess_trajectory`."""
    model_class = None
    STAT_KEYS = ("total_loss", "mean_policy_loss", "mean_vf_loss", "mean_kl_loss", "mean_entropy")

    def __init__(self, observation_space, action_space, config):
        self.observation_space, self.action_space, self.config = observation_space, action_space, config
        self.device = resolve_device(config.get("device"))
        self.dist_class = TorchDiagGaussian
        seed = config.get("seed")
        if seed is not None:
            torch.manual_seed(int(seed))
        self.model = self.make_model("default_model").to(self.device)
        D.broadcast_module_(self.model)
        self.entropy_coeff = float(config.get("entropy_coeff", 0.0))
        self._kl_value = float(config["kl_coeff"])
        self.kl_coeff = torch.tensor(self._kl_value, dtype=torch.float32, device=self.device)
        self.kl_target = float(config.get("kl_target", 0.01))
        cuda = self.device.type == "cuda"
        self._params = [p for p in self.model.parameters() if p.dtype == torch.float32]
        self.optimizer = torch.optim.Adam(self._params, lr=float(config["lr"]), capturable=cuda, foreach=True if cuda else None)
        self.num_grad_updates = 0
        self.use_graphs = bool(config.get("use_hip_graphs", True)) and cuda
        self.autocast_dtype = torch.bfloat16 if str(config.get("policy_dtype", "float32")) in ("bfloat16", "bf16") else None
        self._flat_grad = None
        self._sgd = None
        self._row_sources = None
        # fused HIP learner (2 launches per minibatch step instead of ~250 autograd kernels).  A bfloat16 policy
        # (`policy_dtype`, BASELINE configs[3]) runs the same kernels in their bfloat16-operand mode (fp32 parameters, Adam and
        # losses, as torch.autocast keeps them); the LCF meta pass has no such mode, so bfloat16 CoPO stays on autocast
        self.fused = None
        if bool(config.get("use_fused_learner", True)) and cuda and not config.get("grad_clip"):
            from .fused import FusedLearner
            adv_key, meta_key = self.fused_adv_keys()
            hid = [int(h) for h in (config["model"].get("fcnet_hiddens") or [256, 256])]
            # the bfloat16-operand mode exists only in the row-pass kernels (hidden 64 / 128 / 256 / 512, two equal layers):
            # other widths keep the torch.autocast step instead of failing at the first SGD step
            bf16_ok = self.autocast_dtype is None or (len(hid) == 2 and hid[0] == hid[1] and hid[0] in (64, 128, 256, 512))
            if (self.autocast_dtype is None or meta_key is None) and bf16_ok:
                self.fused = FusedLearner(self, self.train_columns(), int(config["sgd_minibatch_size"]), adv_key, meta_key)
                # writers that only hold the nn.Module (checkpoint_io.load_policy_weights) reach the mirror through this hook
                object.__setattr__(self.model, "_on_external_write", self._weights_changed)

    # ---- construction / inference --------------------------------------------------------------------
    def make_model(self, name):
        n_out = 2 * int(self.action_space.shape[0])
        return self.model_class(self.observation_space, self.action_space, n_out, self.config["model"], name)

    def _autocast(self):
        if self.autocast_dtype is not None and self.device.type == "cuda":
            return torch.autocast("cuda", dtype=self.autocast_dtype)
        import contextlib
        return contextlib.nullcontext()

    @torch.no_grad()
    def compute_actions(self, obs, eps=None, explore=True):
        """obs [B, O] -> (action [B, A] unclipped, logp [B], dist_inputs [B, 2A])."""
        with self._autocast():
            logits, _ = self.model({"obs": obs})
        logits = logits.float()
        dist = self.dist_class(logits, self.model)
        act = dist.sample(eps) if explore else dist.deterministic_sample()
        return act, dist.logp(act), logits

    def update_kl(self, sampled_kl):
        """RLlib's adaptive KL rule (progress.csv 0.2 -> 0.675 = 0.2 * 1.5^3)."""
        if sampled_kl > 2.0 * self.kl_target:
            self._kl_value *= 1.5
        elif sampled_kl < 0.5 * self.kl_target:
            self._kl_value *= 0.5
        self.kl_coeff.fill_(self._kl_value)
        return self._kl_value

    # ---- weights / checkpoint ----------------------------------------------------------------------------
    def get_weights(self):
        return {k: v.detach().cpu().numpy() for k, v in self.model.state_dict().items()}

    def _weights_changed(self):
        """Parameters were written from outside the kernels: refresh the fused learner's transposed mirror lazily and
        drop captured graphs that baked anything derived from the old weights."""
        if self.fused is not None:
            self.fused.invalidate_mirror()

    def set_weights(self, weights):
        sd = {k: torch.as_tensor(v) for k, v in weights.items()}
        self.model.load_state_dict(sd)
        self._weights_changed()

    def get_state(self):
        st = dict(model=self.model.state_dict(), optimizer=self.optimizer.state_dict(), kl_coeff=self._kl_value,
                  num_grad_updates=self.num_grad_updates)
        if self.fused is not None:       # the fused learner owns the Adam moments / step (the torch optimizer is never stepped)
            st["fused"] = self.fused.state()
        return st

    def set_state(self, state):
        self.model.load_state_dict(state["model"])
        self.optimizer.load_state_dict(state["optimizer"])
        self._kl_value = float(state["kl_coeff"])
        self.kl_coeff.fill_(self._kl_value)
        self.num_grad_updates = int(state.get("num_grad_updates", 0))
        if self.fused is not None and state.get("fused") is not None:
            self.fused.load_state(state["fused"])
        self._weights_changed()
        for g in (self._sgd if isinstance(self._sgd, tuple) else (self._sgd,)):
            if g is not None:
                g.reset()

    # ---- minibatch SGD --------------------------------------------------------------------------------
    def loss(self, model, dist_class, train_batch):
        raise NotImplementedError

    def train_columns(self):
        """Per-row scalar/vector columns packed for the minibatch gather (name -> width)."""
        return [(SampleBatch.ACTIONS, 2), (SampleBatch.ACTION_LOGP, 1), (SampleBatch.ACTION_DIST_INPUTS, 4),
                (Postprocessing.ADVANTAGES, 1), (SampleBatch.VF_PREDS, 1), (Postprocessing.VALUE_TARGETS, 1)]

    def _ensure_flat_grads(self):
        if self._flat_grad is not None:
            return
        n = sum(p.numel() for p in self._params)
        self._flat_grad = torch.zeros(n, dtype=torch.float32, device=self.device)
        off = 0
        for p in self._params:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    def prepare_sgd(self, dense, max_rows, mb):
        """Bind the dense (flattened [T*E*N, ...]) sources of one iteration and (re)build the static minibatch
        machinery.  `dense` maps column name -> tensor whose first dim is the flattened row index."""
        cols = self.train_columns()
        width = sum(w for _, w in cols)
        dev = self.device
        if self._row_sources is None or self._row_sources["max_rows"] != max_rows or self._row_sources["mb"] != mb:
            # minibatch tables are sized for the LARGEST rank (every rank runs the same number of steps)
            gmax = torch.tensor([max_rows], dtype=torch.int64, device=dev)
            D.all_reduce_max_(gmax)
            max_mb = max(1, math.ceil(int(gmax.item()) / mb))
            # the tables hold the plans of ALL epochs of an iteration back to back (run_sgd_fused plans them up front and the
            # device-side minibatch counter runs through them without a stop); max_mb = minibatches of ONE epoch
            n_ep = max(1, int(self.config.get("num_sgd_iter", 1)))
            self._row_sources = dict(
                max_rows=max_rows, mb=mb, max_mb=max_mb,
                pack=torch.zeros(max_rows, width, dtype=torch.float32, device=dev),
                rows_all=torch.zeros(max_mb * n_ep, mb, dtype=torch.int64, device=dev),
                w_all=torch.zeros(max_mb * n_ep, mb, dtype=torch.float32, device=dev),
                denom_all=torch.ones(max_mb * n_ep, dtype=torch.float32, device=dev),
                k=torch.zeros(1, dtype=torch.int64, device=dev),
                stats=torch.zeros(len(self.STAT_KEYS), dtype=torch.float32, device=dev),
            )
            self._sgd = None
        rs = self._row_sources
        srcs = [dense[name].reshape(max_rows, w) for name, w in cols]
        if dev.type == "cuda" and len(cols) <= 24 and all(t.is_contiguous() and t.dtype == torch.float32 for t in srcs):
            import ctypes as C      # one kernel instead of one strided copy per column
            from . import _capi
            _capi.check(_capi.lib.copo_pack_columns_f32((C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs]),
                                                        (C.c_int32 * len(srcs))(*[w for _, w in cols]), len(srcs), int(max_rows),
                                                        rs["pack"].data_ptr(), _capi.current_stream()))
        else:
            off = 0
            for (name, w), src in zip(cols, srcs):
                rs["pack"][:, off:off + w].copy_(src)
                off += w
        rs["obs"] = dense[SampleBatch.OBS].reshape(max_rows, -1)
        cc = dense.get("centralized_critic_obs")
        rs["cc_obs"] = None if cc is None else cc.reshape(max_rows, -1)
        return rs

    def _gather_minibatch(self):
        rs = self._row_sources
        k = rs["k"]
        rows = rs["rows_all"].index_select(0, k).view(-1)
        pk = rs["pack"].index_select(0, rows)
        tb = SampleBatch()
        off = 0
        for name, w in self.train_columns():
            tb[name] = pk[:, off:off + w] if w > 1 else pk[:, off]
            off += w
        tb[SampleBatch.OBS] = rs["obs"].index_select(0, rows)
        tb["centralized_critic_obs"] = tb[SampleBatch.OBS] if rs["cc_obs"] is None else rs["cc_obs"].index_select(0, rows)
        tb[SampleBatch.VALID] = rs["w_all"].index_select(0, k).view(-1)
        tb["valid_denominator"] = rs["denom_all"].index_select(0, k).view(())
        return tb

    def _forward_backward(self):
        self._flat_grad.zero_()
        tb = self._gather_minibatch()
        with self._autocast():
            loss = self.loss(self.model, self.dist_class, tb)
        loss.backward()
        st = self.model.tower_stats
        self._row_sources["stats"].add_(torch.stack([st[k].detach().float().reshape(()) for k in self.STAT_KEYS]))
        for k, v in list(st.items()):      # do not keep the autograd graph (and its AccumulateGrad nodes) alive
            if torch.is_tensor(v):
                st[k] = v.detach()

    def _apply(self):
        clip = self.config.get("grad_clip")
        if clip:
            torch.nn.utils.clip_grad_norm_(self._params, float(clip))
        self.optimizer.step()
        self._row_sources["k"].add_(1)

    def _sgd_step_local(self):
        self._forward_backward()
        self._apply()

    def draw_perms(self, n, B_local):
        """Hook for pre-drawn permutations (one per SGD epoch / meta pass).  Batched draws (a segmented argsort, or one
        flat 64-bit radix sort of segment-tagged keys) measured no faster than one `torch.randperm` per epoch on MI355X,
        so plan_epoch draws its own."""
        return None

    def plan_epoch(self, valid_idx, B_local, B_all, mb, bufs=None, perm=None):
        """Shuffle this rank's valid rows and cut them into `n_mb` near-equal static-shape minibatches
        (n_mb from the LARGEST rank so that every rank issues the same number of collectives).
        `bufs` = dict(rows_all, w_all, denom_all, k); default: the SGD buffers."""
        rs = self._row_sources if bufs is None else bufs
        n_mb = max(1, math.ceil(max(B_all) / mb))
        dev = self.device
        if dev.type == "cuda" and len(B_all) <= 16:
            # one kernel builds the tables from the permutation (same tables as the tensor code below)
            import ctypes as C
            from . import _capi
            # shuffle: an explicit permutation if given, else a keyed in-kernel permutation (keys from torch's CPU
            # generator: reproducible under torch.manual_seed, no device sort); "shuffle": "randperm" keeps torch.randperm
            key = None
            if perm is None and B_local > 0:
                if self.config.get("shuffle", "feistel") == "randperm":
                    perm = torch.randperm(B_local, device=dev)
                else:
                    key = (C.c_uint32 * 4)(*[int(v) for v in torch.randint(0, 2 ** 31 - 1, (4,)).tolist()])
            ball = (C.c_int64 * len(B_all))(*[int(b) for b in B_all])
            _capi.check(_capi.lib.copo_plan_epoch(
                valid_idx.data_ptr() if B_local > 0 else None, None if perm is None else perm.data_ptr(), key, int(B_local),
                int(n_mb), int(mb), ball, len(B_all), rs["rows_all"].data_ptr(), rs["w_all"].data_ptr(),
                rs["denom_all"].data_ptr(), rs["k"].data_ptr(), _capi.current_stream()))
            return n_mb
        perm = valid_idx[torch.randperm(B_local, device=dev)] if B_local > 0 else valid_idx
        q, r = divmod(B_local, n_mb)
        k = torch.arange(n_mb, device=dev)
        start = k * q + torch.clamp(k, max=r)
        size = q + (k < r).to(torch.int64)
        j = torch.arange(mb, device=dev)
        pos = start[:, None] + j[None, :]
        w = (j[None, :] < size[:, None])
        if B_local > 0:
            rows = perm[torch.clamp(pos, max=B_local - 1)]
        else:
            rows = torch.zeros(n_mb, mb, dtype=torch.int64, device=dev)
        rs["rows_all"][:n_mb].copy_(torch.where(w, rows, torch.zeros_like(rows)))
        rs["w_all"][:n_mb].copy_(w.to(torch.float32))
        denom = np.zeros(n_mb, np.float64)
        for Br in B_all:
            qq, rr = divmod(Br, n_mb)
            denom += qq + (np.arange(n_mb) < rr)
        rs["denom_all"][:n_mb].copy_(torch.as_tensor(np.maximum(denom, 1.0), dtype=torch.float32))
        rs["k"].zero_()
        return n_mb

    def fused_adv_keys(self):
        """(pack column used as the PPO advantage, pack column used by the meta update or None)."""
        return Postprocessing.ADVANTAGES, None

    # Data-parallel SGD step, default: the gradient tiles are summed over the ranks INSIDE the weight-gradient kernel
    # (`copo_ppo_fused_step_dp_f32`, peer.TileExchange) -- the data-parallel step is the local step's two
    # launches, in the same captured chains.  COPO_DP_EXCHANGE = tile | try | rccl | auto (default): auto takes the tile exchange
    # when a child process per rank shows that it works on this node (dist.probe_tile_exchange), else the RCCL loop below.
    _tile = None            # peer.TileExchange
    _dp_mode = None         # "tile" / "rccl", decided at the first SGD call of a distributed run

    _rs_step = None         # what the step kernels read: the epoch's rows in minibatch order (FusedLearner.gather_epoch)

    def _fused_local(self):
        rs = self._rs_step if self._rs_step is not None else self._row_sources
        if self._dp_mode == "tile":
            self.fused.step_dp(rs, self._tile, stats=self.fused.stats)
        else:
            self.fused.step(rs, stats=self.fused.stats)

    # minibatch steps per captured graph (the device-side minibatch counter walks the plan by itself)
    SGD_CHAIN = int(os.environ.get("COPO_SGD_CHAIN", "16"))

    def _fused_local_chain(self):
        for _ in range(self.SGD_CHAIN):
            self._fused_local()

    # optional longer chain for the bulk of an iteration's ~725 steps (COPO_SGD_CHAIN_LONG > COPO_SGD_CHAIN switches it on).  Measured
    # on one box, same session: chains of 16 2.58 M agent-steps/s, 32 2.56 M, 64 2.45 M -- the synchronised SGD phase is the same
    # 23.6 ms, but the host cannot run as far ahead of a 128-node graph launch; off by default
    SGD_CHAIN_LONG = int(os.environ.get("COPO_SGD_CHAIN_LONG", "0"))

    def _fused_local_chain_long(self):
        for _ in range(self.SGD_CHAIN_LONG):
            self._fused_local()

    dp_reason = None        # why `_dp_mode` is what it is (printed in the bench line: a scaling run that fell back to RCCL says so)

    def _pick_dp_mode(self):
        from . import peer
        want = os.environ.get("COPO_DP_EXCHANGE", "auto")
        if self.device.type != "cuda":
            self.dp_reason = "not a GPU job: collective loop (gloo)"
            return "rccl"
        if want == "rccl":
            self.dp_reason = "COPO_DP_EXCHANGE=rccl"
            return "rccl"
        if D.world_size() == 1:
            self.dp_reason = "one rank (COPO_FORCE_DIST): the data-parallel step kernels, nothing to exchange"
            return "tile"
        if want not in ("tile", "try") and D.ranks_share_a_device(self.device):
            # (a one-GPU test box: a kernel that waits for its peers starves them of compute units)
            self.dp_reason = "ranks share a device: RCCL loop"
            return "rccl"
        c = self.fused.cfg
        # "tile": by name, no probe, a timed-out wait raises on every rank; "try": no probe either, but a timed-out wait falls back to the
        # RCCL loop (what `auto` does after its probe has passed; the tests use it to exercise that fall-back)
        if want in ("tile", "try"):
            probed = None
        else:      # the probe learner has the job's own grid: widest input of all nets (the weight-gradient tiles follow it), all nets, minibatch
            kmax = max([int(c.pol.in_dim)] + [int(c.val[g].in_dim) for g in range(int(c.n_value_heads))])
            probed = D.probe_tile_exchange(c.hidden, kmax, 1 + c.n_value_heads, mb=int(c.mb))
        if probed is None or probed:
            tile = peer.TileExchange(self.fused.cfg, self.device)
            if tile.usable:               # (agreed by all ranks; False: a hipIpc export / open failed somewhere)
                self._tile = tile
                self.dp_reason = ("COPO_DP_EXCHANGE=%s" % want) if probed is None else "start-up probe passed on this node"
                return "tile"
            if want == "tile":
                raise RuntimeError("COPO_DP_EXCHANGE=tile: the ranks could not map each other's exchange workspaces (hipIpc)")
            self.dp_reason = "hipIpc mapping of the exchange workspaces failed: RCCL loop"
        else:
            self.dp_reason = "start-up probe of the tile exchange failed or timed out: RCCL loop"
        return "rccl"

    def _grad_all_reduce(self):
        D.all_reduce_sum_(self.fused.grad)

    def _fused_grads(self):
        self.fused.step(self._row_sources, apply_adam=False, stats=self.fused.stats, bump_index=False)

    def _fused_apply(self):
        self.fused.adam(self._row_sources)

    def _fused_apply_then_grads(self):
        """Adam of the previous minibatch + gradient pass of the next one: one graph launch between two all-reduces."""
        self.fused.adam(self._row_sources)
        self.fused.step(self._row_sources, apply_adam=False, stats=self.fused.stats, bump_index=False)

    def run_sgd_fused(self, valid_idx, B_local, B_all, mb, num_epochs, _perms=None, defer=False):
        """defer=True: returns a callable that reads the statistics back (the ONE device -> host read of the PPO epochs) instead of the
        dict itself -- a caller with more device work to queue (CoPO's meta passes) resolves it afterwards, so the host keeps running
        ahead of the device instead of stopping at the end of the epochs (round 6: ~0.2 ms of launch gaps at the start of the meta phase)."""
        fz = self.fused
        assert mb == fz.cfg.mb, "fused learner was built for minibatch %d" % fz.cfg.mb
        if self._sgd is None and D.is_dist() and self._dp_mode is None:
            self._dp_mode = self._pick_dp_mode()
        tile = self._dp_mode == "tile"
        if self._sgd is None:
            if D.is_dist() and not tile:
                # the RCCL loop: [gradient pass] -> all-reduce -> [Adam + next gradient pass]; the kernels run captured, the collective
                # between them eager (ONE fallback: the captured-collective and peer-kernel variants of round 4 are gone)
                self._sgd = (GraphedCallable(self._fused_grads, self.use_graphs),
                             GraphedCallable(self._fused_apply, self.use_graphs),
                             GraphedCallable(self._fused_apply_then_grads, self.use_graphs))
            else:
                self._sgd = GraphedCallable(self._fused_local, self.use_graphs)
                self._sgd_chain = GraphedCallable(self._fused_local_chain, self.use_graphs)
                self._sgd_chain_long = GraphedCallable(self._fused_local_chain_long, self.use_graphs)
        # the tile exchange has a bounded wait instead of a hang; should one ever time out on the job's links, every rank goes back
        # to the state this call started from and repeats it through the RCCL loop (unless the exchange was asked for by name)
        snap = None
        by_name = os.environ.get("COPO_DP_EXCHANGE", "auto") == "tile"
        if tile and self._tile is not None and not by_name:
            snap = (fz.flat.flat.clone(), fz.adam_m.clone(), fz.adam_v.clone(), fz.step_count.clone(), self.num_grad_updates)
        fz.stats.zero_()
        fz.sync_mirror()          # graph replays below do not run python: refresh the transposed weights here if needed
        steps = 0
        n_epochs_asked = num_epochs
        perms = _perms if _perms is not None else self.draw_perms(num_epochs, B_local)      # (a retry after a fall-back repeats the SAME epochs)
        rs0 = self._row_sources
        n_mb_ep = max(1, math.ceil(max(B_all) / mb))
        if getattr(self, "_gather_ok", None) is None:      # decided once: captured chains keep reading the sources they were captured on
            self._gather_ok = bool(self.config.get("gather_epoch_rows", True) and fz.gather_epoch_ok(rs0))      # (memory for the copy, fp32 sources; else: row tables)
        gather_ok = self._gather_ok
        if (tile or not D.is_dist()) and self.use_graphs and gather_ok and self.config.get("plan_all_epochs", True) \
                and num_epochs * n_mb_ep <= int(rs0["rows_all"].shape[0]):
            # every epoch's plan up front, back to back in the tables, ONE gather of all planned rows into minibatch order; the
            # device-side minibatch counter then walks all num_epochs x n_mb steps in captured chains without a host stop
            # between the epochs (per epoch that was a plan kernel, a gather kernel and the launch gaps around them)
            for ep in range(num_epochs):
                o = ep * n_mb_ep
                self.plan_epoch(valid_idx, B_local, B_all, mb, perm=None if perms is None else perms[ep],
                                bufs=dict(rows_all=rs0["rows_all"][o:], w_all=rs0["w_all"][o:], denom_all=rs0["denom_all"][o:], k=rs0["k"]))
            total = num_epochs * n_mb_ep
            self._rs_step = fz.gather_epoch(rs0, total)
            _k0 = 0
            while self.SGD_CHAIN_LONG > self.SGD_CHAIN and _k0 + self.SGD_CHAIN_LONG <= total:
                self._sgd_chain_long()
                _k0 += self.SGD_CHAIN_LONG
            while _k0 + self.SGD_CHAIN <= total:
                self._sgd_chain()
                _k0 += self.SGD_CHAIN
            for _k in range(_k0, total):
                self._sgd()
            steps = total
            num_epochs = 0          # (the per-epoch loop below has nothing left to do)
        for ep in range(num_epochs):
            n_mb = self.plan_epoch(valid_idx, B_local, B_all, mb, perm=None if perms is None else perms[ep])
            if (tile or not D.is_dist()) and gather_ok:
                # (persistent buffers: the captured chains keep reading the same addresses)
                self._rs_step = fz.gather_epoch(self._row_sources, n_mb)
            _k0 = 0
            if (tile or not D.is_dist()) and self.use_graphs:
                # most of an epoch in graphs of SGD_CHAIN steps: fewer graph launches, no gap between their kernels
                while _k0 + self.SGD_CHAIN <= n_mb:
                    self._sgd_chain()
                    _k0 += self.SGD_CHAIN
                    steps += self.SGD_CHAIN
            for _k in range(_k0, n_mb):
                if D.is_dist() and not tile:
                    # two host calls per minibatch: [Adam of the previous one + this gradient pass], all-reduce; the
                    # last Adam of the epoch is flushed before the next plan resets the minibatch index
                    self._sgd[0 if _k == _k0 else 2]()
                    self._grad_all_reduce()
                    if _k == n_mb - 1:
                        self._sgd[1]()
                else:
                    self._sgd()
                steps += 1
        self.num_grad_updates += steps
        if self._tile is not None:
            # every rank learns whether ANY rank's wait timed out (a rank that raised alone would leave its peers hanging in their
            # next collective with partially summed parameters)
            ok = self._tile.ok()
            if int(self.config.get("dp_fault_injection_rank", -1)) == D.rank() and not getattr(self, "_dp_fault_injected", False):
                # test hook: this rank reports a timed-out wait ONCE (the kernels' own time-out needs starving ranks or dead links;
                # what follows it -- agreement, restore, the same epochs again through the RCCL loop -- must be covered every time)
                ok, self._dp_fault_injected = False, True
            good = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
            if D.world_size() > 1:
                import torch.distributed as td
                td.all_reduce(good, op=td.ReduceOp.MIN)
            if not bool(good.item()):
                if snap is None:          # asked for by name: no fall-back -- close the mappings, raise on every rank
                    self._tile.close()
                    self._tile = None
                    raise RuntimeError("data-parallel tile exchange (COPO_DP_EXCHANGE=tile): a wait for a peer timed out on some rank")
                else:
                    import warnings
                    warnings.warn("data-parallel tile exchange: a wait for a peer timed out; back to the RCCL loop from the state before this call")
                    fz.flat.flat.copy_(snap[0]); fz.adam_m.copy_(snap[1]); fz.adam_v.copy_(snap[2]); fz.step_count.copy_(snap[3])
                    self.num_grad_updates = snap[4]
                    fz.invalidate_mirror()
                    self._tile.close()
                    self._tile, self._dp_mode, self._sgd, self._rs_step = None, "rccl", None, None
                    self.dp_reason = "a wait of the tile exchange timed out during training: RCCL loop from then on"
                    return self.run_sgd_fused(valid_idx, B_local, B_all, mb, n_epochs_asked, _perms=perms, defer=defer)
        # a rank's statistics are ITS rows' terms over the GLOBAL row count of each minibatch: the sum over the ranks is the global mean.
        # Without it the KL coefficient would follow 1 / world of the sampled KL, differently on every rank (found by the two-rank
        # end-to-end test of round 6: the ranks' parameters stay identical only while their KL coefficients do)
        D.all_reduce_sum_(fz.stats)
        means = fz.stats / max(1, steps)          # (its own tensor: the next call zeroes fz.stats)

        copy = {}

        def start_copy():                  # asynchronous device -> host copy; resolve() then waits for ITS event, not for the stream
            if means.is_cuda:
                if getattr(self, "_stats_pin", None) is None:
                    self._stats_pin = torch.empty(means.numel(), dtype=means.dtype, pin_memory=True)
                self._stats_pin.copy_(means, non_blocking=True)
                copy["ev"] = torch.cuda.Event()
                copy["ev"].record()

        def resolve(values=None):          # values: the means, already on the host (they rode along in another read)
            if values is None and "ev" in copy:
                copy["ev"].synchronize()
                values = self._stats_pin.tolist()
            tot, pol, vf, kl, ent, vfn, vfg, adv = means.tolist() if values is None else values
            return dict(total_loss=tot, policy_loss=pol, vf_loss=vf, kl=kl, entropy=ent, cur_kl_coeff=self._kl_value,
                        cur_lr=float(self.config["lr"]), num_sgd_steps=steps, mean_nei_vf_loss=vfn, mean_global_vf_loss=vfg,
                        normalized_advantages=adv)
        resolve.means, resolve.start_copy = means, start_copy
        return resolve if defer else resolve()

    def run_sgd(self, valid_idx, B_local, B_all, mb, num_epochs, defer=False):
        """`num_sgd_iter` epochs of minibatch SGD; returns the mean learner stats over every step taken (defer=True: a callable
        that returns them, see run_sgd_fused)."""
        if self.fused is not None:
            return self.run_sgd_fused(valid_idx, B_local, B_all, mb, num_epochs, defer=defer)
        self._ensure_flat_grads()
        rs = self._row_sources
        if self._sgd is None:
            if D.is_dist():
                self._sgd = (GraphedCallable(self._forward_backward, self.use_graphs),
                             GraphedCallable(self._apply, self.use_graphs))
            else:
                self._sgd = GraphedCallable(self._sgd_step_local, self.use_graphs)
        rs["stats"].zero_()
        steps = 0
        perms = self.draw_perms(num_epochs, B_local)
        for ep in range(num_epochs):
            n_mb = self.plan_epoch(valid_idx, B_local, B_all, mb, perm=None if perms is None else perms[ep])
            for _k in range(n_mb):
                if D.is_dist():
                    self._sgd[0]()
                    D.all_reduce_sum_(self._flat_grad)
                    self._sgd[1]()
                else:
                    self._sgd()
                steps += 1
        self.num_grad_updates += steps
        D.all_reduce_sum_(rs["stats"])          # (partial sums over this rank's rows -> the global means, as in run_sgd_fused)
        vals = (rs["stats"] / max(1, steps)).tolist()
        out = dict(zip(self.STAT_KEYS, vals))
        res = dict(total_loss=out["total_loss"], policy_loss=out["mean_policy_loss"], vf_loss=out["mean_vf_loss"],
                   kl=out["mean_kl_loss"], entropy=out["mean_entropy"], cur_kl_coeff=self._kl_value,
                   cur_lr=float(self.config["lr"]), num_sgd_steps=steps,
                   **{k: v for k, v in out.items() if k not in ("total_loss", "mean_policy_loss", "mean_vf_loss",
                                                                "mean_kl_loss", "mean_entropy")})
        return (lambda: res) if defer else res

    # ---- postprocess (dense) -----------------------------------------------------------------------------
    def critic_obs_dense(self, batch):
        """[T, E, N, C] centralised-critic observation for every row (IPPO / fuse 'none': the obs itself)."""
        return batch[SampleBatch.OBS]

    def gae_heads(self):
        return 1

    def gae_gammas(self):
        return [float(self.config["gamma"])]

    @torch.no_grad()
    def value_heads_dense(self, cc_flat):
        """[heads, rows] values of every critic head for flattened critic observations."""
        with self._autocast():
            v = self.model._value_branch(self.model._value_branch_separate(cc_flat)).reshape(-1)
        return v.float().unsqueeze(0)

    @torch.no_grad()
    def postprocess_trajectory(self, sample_batch, other_agent_batches=None, episode=None):
        """Dense counterpart of `postprocess_trajectory` (algo_ccppo.py:322-374, algo_copo.py:473-502): critic
        observation fusion, value heads, and every GAE head in one segmented scan, for all [T, E, N] rows at once.
        Trajectories are cut at fragment boundaries of `rollout_fragment_length` steps like RLlib's
        truncate_episodes mode."""
        from . import ops
        b = sample_batch
        obs = b[SampleBatch.OBS]
        T, E, N = obs.shape[0], obs.shape[1], obs.shape[2]
        M = E * N
        # look-ahead batches (VecTrainer._lookahead_batch, centralised critics): the last row only supplies the bootstrap value
        look = bool(b.get("_lookahead", False))
        Tt = T - 1 if look else T
        cc = self.critic_obs_dense(b)
        b["centralized_critic_obs"] = cc
        H = self.gae_heads()
        fz = self.fused
        if fz is not None and fz.can_forward and self.config.get("use_fused_inference", True) and \
                int(fz.cfg.n_value_heads) == H:
            fz.sync_mirror()
            cc_flat = cc.reshape(T * M, -1)
            # only rows that hold an acting agent have a value the GAE scan reads (about half of the slots): the row list
            # is the one `valid_rows` needs anyway (the iteration's single host sync), kept on the batch for it
            valid = (b[SampleBatch.FLAGS].reshape(-1) & F_ACTED).bool()
            idx = valid.nonzero(as_tuple=False).view(-1)
            if not look:
                b["_valid"], b["_valid_idx"] = valid, idx
            vals = fz.values(obs.reshape(T * M, -1), None if cc_flat.data_ptr() == obs.data_ptr() else cc_flat,
                             rows=idx).view(H, T, M)
        else:
            vals = self.value_heads_dense(cc.reshape(T * M, -1)).reshape(H, T, M).contiguous()
        rew = b["rew3"][:H].reshape(H, T, M)
        flags = b[SampleBatch.FLAGS].reshape(T, M)
        frag = int(self.config.get("rollout_fragment_length", Tt) or Tt)
        lam = float(self.config["lambda"])
        if look:
            v_next = vals[:, Tt].contiguous()
            vals, rew, flags = vals[:, :Tt].contiguous(), rew[:, :Tt].contiguous(), flags[:Tt]
        adv, tgt = torch.empty_like(vals), torch.empty_like(vals)
        for lo in range(0, Tt, frag):
            hi = min(Tt, lo + frag)
            if lo == 0 and hi == Tt:
                ops.gae3(rew.contiguous(), vals, flags, self.gae_gammas(), lam, adv, tgt)
            else:
                a, g = ops.gae3(rew[:, lo:hi].contiguous(), vals[:, lo:hi].contiguous(), flags[lo:hi].contiguous(),
                                self.gae_gammas(), lam)
                adv[:, lo:hi], tgt[:, lo:hi] = a, g
        if look:
            # the value of the NEXT ROW (its centralised critic observation exists now) instead of the last-row shortcut
            # (only the LAST fragment runs into row Tt; earlier fragments were cut -- and bootstrapped from their own last
            # row, the reference's shortcut -- by the scan above, so the correction must not reach back into them)
            lo_last = ((Tt - 1) // frag) * frag if frag < Tt else 0
            self._apply_bootstrap(vals[:, lo_last:], adv[:, lo_last:], tgt[:, lo_last:], flags[lo_last:], lam, v_next)
            b["_v_next"] = v_next
            for k in list(b.keys()):          # what the trainer sees: the Tt training rows (views of the persistent buffers)
                v = b[k]
                if not torch.is_tensor(v) or k.startswith("_"):
                    continue
                b[k] = v[:, :Tt] if k == "rew3" else (v[:Tt] if v.shape[0] == T else v)
            T = Tt
        elif self.bootstrap_next_obs() and frag >= T and b.get("_next_obs_last") is not None:
            self._bootstrap_from_next_obs(b, vals, adv, tgt, flags, lam)
        b[SampleBatch.VF_PREDS], b[Postprocessing.ADVANTAGES], b[Postprocessing.VALUE_TARGETS] = \
            vals[0].view(T, E, N), adv[0].view(T, E, N), tgt[0].view(T, E, N)
        b["_vals"], b["_adv"], b["_tgt"] = vals, adv, tgt
        return b

    def bootstrap_next_obs(self):
        """How a trajectory that is cut by the end of the fragment is bootstrapped.  RLlib's PPO (the reference's IPPO,
        algo_ippo.py: `compute_gae_for_sample_batch`) evaluates the critic on the observation AFTER the last step; the
        reference's CCPPO / CoPO pass the value of the last ROW instead (algo_ccppo.py:362-365, algo_copo.py:492-496: the
        centralised critic observation of the next step does not exist yet) -- the scan of `copo_gae3_f32`.  With 200-step
        fragments that shortcut touches a trajectory once or twice; with the 8-step fragments of 256 lockstep scenes it
        touches every row, so critics that only read the agent's own observation may opt into the exact bootstrap
        (`bootstrap_next_obs`, default True; a centralised critic -- CCPPO's mean-field / concat modes -- cannot and keeps
        the reference's shortcut).  Batches that carry no next observation (the reference's own postprocess inputs in the
        golden tests) are scanned exactly as the reference does."""
        v = self.config.get("bootstrap_next_obs")
        return True if v is None else bool(v)

    def _bootstrap_from_next_obs(self, b, vals, adv, tgt, flags, lam):
        """V(observation after the last step) for the trajectories that run into the end of the fragment (evaluated for every
        slot: a dense forward pass costs less than listing the live slots would -- that needs a host round trip)."""
        H, T, M = vals.shape
        nxt = b["_next_obs_last"].reshape(M, -1)
        fz = self.fused
        if fz is not None and fz.can_forward and self.config.get("use_fused_inference", True) and int(fz.cfg.n_value_heads) == H:
            v_next = fz.values(nxt.contiguous(), None)
        else:
            v_next = self.value_heads_dense(nxt)
        self._apply_bootstrap(vals, adv, tgt, flags, lam, v_next)

    _boot_w = None

    def _apply_bootstrap(self, vals, adv, tgt, flags, lam, v_next):
        """GAE is linear in the bootstrap value: replacing V(last row) by `v_next` [H, M] adds (gamma lambda)^(T-1-t) gamma
        (v_next - V(last row)) to every row t of the trajectory that runs into the end of the fragment."""
        H, T, M = vals.shape
        key = (H, T, float(lam), tuple(self.gae_gammas()), str(vals.device))
        if self._boot_w is None or self._boot_w[0] != key:
            gam = torch.tensor(self.gae_gammas(), dtype=torch.float32, device=vals.device).view(H, 1, 1)
            k = torch.arange(T - 1, -1, -1, device=vals.device, dtype=torch.float32).view(1, T, 1)
            self._boot_w = (key, gam.view(H, 1), torch.pow(gam * lam, k))              # gamma [H, 1], (gamma lambda)^(T-1-t) [H, T, 1]
        gam, w = self._boot_w[1], self._boot_w[2]
        cont = (flags & (F_ACTED | F_DONE)) == F_ACTED                                # [T, M] the agent drives on after row t
        run = torch.flip(torch.cumprod(torch.flip(cont, [0]).to(torch.float32), 0), [0])      # rows of the trajectory alive at the end
        delta = gam * (v_next - vals[:, T - 1]) * run[T - 1]                             # [H, M]
        corr = w * run.unsqueeze(0) * delta.unsqueeze(1)
        adv += corr
        tgt += corr

    def wants_lookahead(self):
        """A centralised critic has no critic observation for the step after the fragment (the neighbours' next actions are
        not taken yet); `lookahead` (default on) trains every row one step late instead, so that its successor row exists."""
        return (not self.bootstrap_next_obs()) and bool(self.config.get("lookahead", True)) and \
            int(self.model.value_input_dim()) != int(self.observation_space.shape[0])


# ----------------------------------------------------------------------------------------------------
# trainer
# ----------------------------------------------------------------------------------------------------
class VecTrainer:
    """`Algorithm`-shaped driver: `train()` runs one iteration and returns an RLlib-style result dict."""
    _name = "PPO"
    _allow_unknown_configs = True

    @classmethod
    def get_default_config(cls):
        raise NotImplementedError

    def get_default_policy_class(self, config):
        raise NotImplementedError

    def __init__(self, config=None, env=None, logger_creator=None):
        cfg = type(self).get_default_config()
        user = dict(config or {})
        if env is not None:
            user["env"] = env
        cfg.update_from_dict(user)
        D.init_from_env(cfg.get("device"))
        cfg.validate()
        self.config = cfg
        self._counters = defaultdict(int)
        self._timers = defaultdict(float)
        self.iteration = 0
        self._t_start = time.time()
        self.callbacks = cfg.callbacks() if isinstance(cfg.callbacks, type) else cfg.callbacks
        self.setup(cfg)

    def setup(self, cfg):
        from copo_amd.torch_copo.utils.env_wrappers import lookup_env
        env_cls = lookup_env(cfg["env"])
        env_config = dict(cfg["env_config"])
        env_config.setdefault("num_envs", int(cfg["num_envs"]))
        device = resolve_device(cfg.get("device"))
        env_config.setdefault("device", device.index or 0)
        self.env = env_cls(env_config)
        if self.env.sim.A != 2:
            self.env.close()
            raise NotImplementedError(
                "communication widens the action to 2 + comm_size floats; the trainers (like the reference's, whose scripts "
                "keep comm_method='none') drive [steering, throttle] only -- use the env API for the channel")
        pol_cls = self.get_default_policy_class(cfg)
        self.policy = pol_cls(cfg.observation_space, cfg.action_space, cfg)
        E = self.env.sim.E
        T = max(1, math.ceil(int(cfg["train_batch_size"]) / E))
        self.sampler = VecSampler(self.env, self.policy, T, use_graph=cfg.get("use_hip_graphs", True),
                                  stagger_episodes=cfg.get("stagger_episodes", False))
        self.workers = _LocalWorkerSet(self)
        self._episode_stats = defaultdict(float)
        if self.callbacks is not None and hasattr(self.callbacks, "on_algorithm_init"):
            self.callbacks.on_algorithm_init(algorithm=self)

    # ---- RLlib-shaped accessors -----------------------------------------------------------------------------
    def get_policy(self, policy_id="default"):
        return self.policy

    # ---- the iteration ------------------------------------------------------------------------------------
    def collect(self):
        t0 = time.perf_counter()
        self._wait_metric_sums()               # (a caller that never read the previous rollout's sums: they read the buffers this rollout rewrites)
        batch = self.sampler.sample()
        self._metrics_batch = batch            # the rows of THIS rollout (episode metrics, counters)
        # its metric sums are queued NOW and read back at the end of train(): no kernel launch behind the iteration's last host stop.
        # On the GPU the one-workgroup reduction (70 us) runs on a side stream, under the postprocess
        if self.policy.device.type == "cuda" and not D.is_dist():
            if getattr(self, "_metric_stream", None) is None:
                self._metric_stream = concurrent_stream(self.policy.device)
            ms = self._metric_stream
            ms.wait_stream(torch.cuda.current_stream(self.policy.device))
            with torch.cuda.stream(ms):
                sums = self.episode_sums(batch)
            self._metric_event = torch.cuda.Event()
            self._metric_event.record(ms)
            self._metric_sums = (batch, sums)
        else:
            self._metric_event = None
            self._metric_sums = (batch, self.episode_sums(batch))
        if self.policy.wants_lookahead():
            batch = self._lookahead_batch(batch)
        self.policy.postprocess_trajectory(batch)
        self._timers["sample_time_ms"] = (time.perf_counter() - t0) * 1e3
        return batch

    _look = None

    def _lookahead_batch(self, cur):
        """Centralised critics (CCPPO mean-field / concat): T + 1 rows = [last row of the previous rollout | the T new rows] in
        persistent buffers.  The first T are trained on now; the last one is only evaluated by the critic -- the exact
        bootstrap of the trajectories that run through the fragment boundary, which the reference replaces by the value of
        the last row itself (algo_ccppo.py:362-365: negligible with its 200-step fragments, not with T = 8) -- and is
        trained on in the next iteration.  Every row is used exactly once, one iteration late."""
        T = self.sampler.T
        skip = (SampleBatch.REWARDS, "nei_rewards", "global_rewards")          # views of rew3
        if self._look is None:
            self._look = {}
            for k, v in cur.items():
                if k.startswith("_") or k in skip or not torch.is_tensor(v):
                    continue
                shape = (3, T + 1) + tuple(v.shape[2:]) if k == "rew3" else (T + 1,) + tuple(v.shape[1:])
                self._look[k] = torch.zeros(shape, dtype=v.dtype, device=v.device)
        for k, buf in self._look.items():
            if k == "rew3":
                buf[:, 0].copy_(buf[:, T])
                buf[:, 1:].copy_(cur[k])
            else:
                buf[0].copy_(buf[T])
                buf[1:].copy_(cur[k])
        out = SampleBatch(self._look)
        r3 = self._look["rew3"]
        out[SampleBatch.REWARDS], out["nei_rewards"], out["global_rewards"] = r3[0], r3[1], r3[2]
        out["_lookahead"] = True
        return out

    def _wait_metric_sums(self):
        """The current stream may read the sums `collect` queued on the metric stream."""
        ev = getattr(self, "_metric_event", None)
        if ev is not None:
            torch.cuda.current_stream(self.policy.device).wait_event(ev)
            self._metric_event = None

    def valid_rows(self, batch):
        if "_valid_idx" in batch:                          # already listed by the dense postprocess
            return batch["_valid"], batch["_valid_idx"], int(batch["_valid_idx"].numel())
        flags = batch[SampleBatch.FLAGS].reshape(-1)
        valid = (flags & F_ACTED).bool()
        idx = valid.nonzero(as_tuple=False).view(-1)       # the one host sync of an iteration
        return valid, idx, int(idx.numel())

    def standardize_advantages(self, batch, valid):
        """PPO's `standardize_fields(["advantages"])` over the valid rows of ALL ranks."""
        adv = batch[Postprocessing.ADVANTAGES].reshape(-1)
        w = valid.to(torch.float64)
        a = adv.to(torch.float64)
        stats = torch.stack([w.sum(), (a * w).sum(), (a * a * w).sum()])
        D.all_reduce_sum_(stats)
        n, s, ss = stats.tolist()
        mean = s / max(n, 1.0)
        std = max(1e-4, math.sqrt(max(ss / max(n, 1.0) - mean * mean, 0.0)))
        batch[Postprocessing.ADVANTAGES] = ((adv - mean) / std * valid).view_as(batch[Postprocessing.ADVANTAGES])

    def training_step(self):
        cfg, pol = self.config, self.policy
        batch = self.collect()
        valid, idx, B = self.valid_rows(batch)
        B_all = D.all_gather_int(B, pol.device)
        self._counters[NUM_AGENT_STEPS_SAMPLED] += sum(B_all)
        self._counters[NUM_ENV_STEPS_SAMPLED] += self.sampler.T * self.sampler.E * D.world_size()
        self.standardize_advantages(batch, valid)
        t0 = time.perf_counter()
        max_rows = batch[SampleBatch.FLAGS].numel()
        pol.prepare_sgd(batch, max_rows, int(cfg["sgd_minibatch_size"]))
        stats = pol.run_sgd(idx, B, B_all, int(cfg["sgd_minibatch_size"]), int(cfg["num_sgd_iter"]))
        self._timers["learn_time_ms"] = (time.perf_counter() - t0) * 1e3
        pol.update_kl(stats["kl"])
        self._last_batch = batch
        return {"default": {LEARNER_STATS_KEY: stats, "custom_metrics": {}}}

    # ---- metrics --------------------------------------------------------------------------------------------
    def episode_metrics(self, batch):
        """Device-side reduction of the terminal flags / info of this iteration's rows: the quantities
        `MultiAgentDrivingCallbacks` derives from info dicts (utils/callbacks.py:48-110), aggregated over the
        agents that terminated in this iteration."""
        return self._metrics_from_sums(self.episode_sums(batch))

    def episode_sums(self, batch):
        """The device half of `episode_metrics`: 15 float64 sums over all ranks (no host read)."""
        fl8 = batch[SampleBatch.FLAGS].reshape(-1)
        info = batch["infos"].reshape(-1, 8)
        nbr = batch["nbr_cnt"].reshape(-1)
        if fl8.is_cuda and fl8.dtype == torch.uint8 and nbr.dtype == torch.int32 and info.dtype == torch.float32:
            from . import _capi       # one kernel, one host read
            fl8, info, nbr = fl8.contiguous(), info.contiguous(), nbr.contiguous()
            sums = torch.empty(15, dtype=torch.float64, device=fl8.device)
            _capi.check(_capi.lib.copo_episode_metrics(fl8.data_ptr(), info.data_ptr(), nbr.data_ptr(), fl8.numel(),
                                                       sums.data_ptr(), _capi.current_stream()))
        else:
            flags = fl8.to(torch.int32)
            acted = (flags & F_ACTED) > 0
            done = ((flags & F_DONE) > 0) & acted
            sums = self._episode_sums_torch(flags, info, acted, done, nbr)
        D.all_reduce_sum_(sums)
        return sums

    def _episode_sums_torch(self, flags, info, acted, done, nbr):
        f64 = torch.float64
        return torch.stack([
            done.sum().to(f64), ((flags & F_ARRIVE) > 0)[done].sum().to(f64), ((flags & F_CRASH) > 0)[done].sum().to(f64),
            ((flags & F_OUT) > 0)[done].sum().to(f64), ((flags & F_MAXSTEP) > 0)[done].sum().to(f64),
            info[done, 5].sum().to(f64), info[done, 6].sum().to(f64), info[done, 7].sum().to(f64),
            acted.sum().to(f64), info[acted, 0].sum().to(f64), info[acted, 1].sum().to(f64), info[acted, 2].sum().to(f64),
            info[acted, 3].sum().to(f64), info[acted, 4].sum().to(f64),
            nbr[acted].sum().to(f64),
        ])

    def _metrics_from_sums(self, sums):
        (nd, arr, crash, out, maxs, ep_len, ep_rew, rc, na, vel, steer, acc, srew, cost, nnb) = sums.tolist() if torch.is_tensor(sums) else sums
        cm = {}
        if nd > 0:
            cm.update(success_rate_mean=arr / nd, crash_rate_mean=crash / nd, out_of_road_rate_mean=out / nd,
                      max_step_rate_mean=maxs / nd, episode_length_mean=ep_len / nd, episode_reward_mean=ep_rew / nd,
                      route_completion_mean=rc / nd, episode_cost_mean=(crash + out) / nd)
        if na > 0:
            cm.update(velocity_mean=vel / na, steering_mean=steer / na, acceleration_mean=acc / na,
                      step_reward_mean=srew / na, cost_mean=cost / na, num_neighbours_mean=nnb / na)
        cm["num_terminated_agents"] = nd
        cm["num_acting_rows"] = na
        return cm

    def evaluate(self, num_fragments=50, min_episodes=0, scene_episodes=None):
        """Roll the CURRENT policy without learning and aggregate what `RecorderEnv` reports per population
        (copo/eval/recoder.py:139-152 success / crash / out / max_step rates, episode reward / length, velocity, ...)
        over the agents that terminate: `num_fragments` sampler fragments, more until `min_episodes` agents finished.
        `scene_episodes` = k: the next k WHOLE episodes of every scene instead (until done["__all__"], the unit of
        eval/evaluate_population.py:57-76) -- rows of a scene's partial first episode and of episodes after its k-th are
        left out, so slow agents and the drain phase of an episode weigh what they weigh in the reference's evaluation."""
        tot, n_frag = {}, 0
        # [E] episodes completed per scene; -1: still inside the partial episode the call started in
        ep = torch.full((self.sampler.E,), -1 if self.sampler._started else 0, dtype=torch.int64, device=self.sampler.device)
        while True:
            if scene_episodes is None:
                if not (n_frag < num_fragments or tot.get("num_terminated_agents", 0) < min_episodes):
                    break
                if n_frag > 100 * max(1, num_fragments):
                    break
            else:
                # the loop body holds collectives (episode_metrics all-reduces), so the stop decision has to be the same on
                # every rank: scenes are seeded per rank and finish their k episodes after different fragment counts --
                # a rank that is done keeps sampling (its rows are masked out) until the slowest rank is done too
                pending = (ep < int(scene_episodes)).any().to(torch.float32).view(1)
                if D.is_dist():
                    D.all_reduce_max_(pending)
                if not bool(pending.item() > 0):
                    break
            batch = self.sampler.sample()
            if scene_episodes is not None:
                fl = batch[SampleBatch.FLAGS]                                   # [T, E, N]
                ended = ((fl & F_ENV_RESET) > 0).any(-1).to(torch.int64)        # [T, E]
                before = ep[None] + torch.cumsum(ended, 0) - ended              # episodes completed before step t
                keep = (before >= 0) & (before < int(scene_episodes))
                ep = ep + ended.sum(0)
                batch = SampleBatch(batch)
                batch[SampleBatch.FLAGS] = fl * keep[..., None].to(fl.dtype)
            cm = self.episode_metrics(batch)
            nd, na = cm.get("num_terminated_agents", 0.0), cm.get("num_acting_rows", 0.0)
            for k, v in cm.items():
                if k in ("num_terminated_agents", "num_acting_rows"):
                    tot[k] = tot.get(k, 0.0) + v
                else:      # means -> weighted sums (per terminated agent or per acting row)
                    w = nd if k in self._EPISODE_KEYS else na
                    tot[k] = tot.get(k, 0.0) + v * w
            n_frag += 1
            if scene_episodes is not None and n_frag * self.sampler.T > 6 * int(scene_episodes + 1) * int(self.env.sim.cfg.horizon):
                break       # (an episode is at most 5 x horizon env steps)
        out = {}
        for k, v in tot.items():
            if k in ("num_terminated_agents", "num_acting_rows"):
                out[k] = v
            else:
                w = tot["num_terminated_agents"] if k in self._EPISODE_KEYS else tot["num_acting_rows"]
                out[k] = v / w if w > 0 else float("nan")
        out["env_steps"] = n_frag * self.sampler.T * self.sampler.E
        return out

    _EPISODE_KEYS = ("success_rate_mean", "crash_rate_mean", "out_of_road_rate_mean", "max_step_rate_mean",
                     "episode_length_mean", "episode_reward_mean", "route_completion_mean", "episode_cost_mean")

    def train(self):
        t0 = time.perf_counter()
        train_results = self.training_step()
        dt = time.perf_counter() - t0
        self.iteration += 1
        mbatch = getattr(self, "_metrics_batch", None) or self._last_batch
        early, self._metric_sums = getattr(self, "_metric_sums", None), None
        self._wait_metric_sums()
        cm = self._metrics_from_sums(early[1]) if early is not None and early[0] is mbatch else self.episode_metrics(mbatch)
        agent_steps = self._counters[NUM_AGENT_STEPS_SAMPLED]
        result = dict(
            training_iteration=self.iteration, timesteps_total=self._counters[NUM_ENV_STEPS_SAMPLED],
            agent_timesteps_total=agent_steps, time_this_iter_s=dt, time_total_s=time.time() - self._t_start,
            episode_reward_mean=cm.get("episode_reward_mean", float("nan")),
            episode_len_mean=cm.get("episode_length_mean", float("nan")), policy_reward_mean={},
            custom_metrics=cm, info=dict(learner=train_results, num_agent_steps_sampled=agent_steps,
                                         num_env_steps_sampled=self._counters[NUM_ENV_STEPS_SAMPLED]),
            timers=dict(self._timers), num_healthy_workers=D.world_size(),
        )
        if self.callbacks is not None and hasattr(self.callbacks, "on_train_result"):
            self.callbacks.on_train_result(algorithm=self, result=result)
        return result

    # ---- checkpoints ---------------------------------------------------------------------------------------
    def save_checkpoint(self, checkpoint_dir):
        os.makedirs(checkpoint_dir, exist_ok=True)
        path = os.path.join(checkpoint_dir, "checkpoint-%d.pt" % self.iteration)
        if D.rank() == 0:
            torch.save(dict(policy=self.policy.get_state(), counters=dict(self._counters), iteration=self.iteration,
                            trainer=self._name), path)
        return path

    save = save_checkpoint

    def load_checkpoint(self, path):
        st = torch.load(path, map_location=self.policy.device, weights_only=False)
        self.policy.set_state(st["policy"])
        self._counters.update(st["counters"])
        self.iteration = st["iteration"]

    restore = load_checkpoint

    def export_npz(self, path):
        """Flat numpy export with the reference's key names (best_checkpoints/*.npz layout)."""
        np.savez(path, **{k: v for k, v in self.policy.get_weights().items()})
        return path

    def stop(self):
        try:
            self.env.close()
        except Exception:
            pass
        for name in ("_tile",):
            peer = getattr(self.policy, name, None)
            if peer is not None:
                peer.close()
                setattr(self.policy, name, None)


class _LocalWorker:
    def __init__(self, trainer):
        self._t = trainer
        self.policy_map = {"default": trainer.policy}

    def foreach_policy(self, fn):
        return [fn(self._t.policy, "default")]

    def foreach_env(self, fn):
        return [fn(self._t.env)]

    def set_global_vars(self, gv):
        self.global_vars = gv


class _LocalWorkerSet:
    """`self.workers` facade: there are no remote rollout workers, every rank is learner and sampler."""

    def __init__(self, trainer):
        self._w = _LocalWorker(trainer)

    def num_remote_workers(self):
        return 0

    def local_worker(self):
        return self._w

    def foreach_worker_with_id(self, fn):
        return [fn(0, self._w)]

    def sync_weights(self, **kw):
        return None

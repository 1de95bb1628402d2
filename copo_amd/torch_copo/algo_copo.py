"""CoPO: coordinated policy optimisation -- local coordination through an LCF-weighted advantage, global
coordination through a meta-gradient on the LCF distribution.

Counterpart of the reference's `copo/torch_copo/algo_copo.py`: CoPOConfig (:63-92), CoPOModel (:96-182),
compute_{nei,global}_advantage (:189-204), CoPOPolicy.meta_update / loss / assign_lcf /
postprocess_trajectory (:207-502), CoPOTrainer.training_step (:516-661).

Differences in mechanism, not in math: rows are dense [T, E, N] device tensors; the three GAE heads run in
one HIP segmented scan; the coordinated advantage + standardisation is a HIP reduction; the SGD and meta
steps are static-shape (hipGraph-capturable) minibatches; on several GPUs both meta gradients are
all-reduced BEFORE their dot product.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from copo_amd import dist as D
from copo_amd.engine import (LEARNER_STATS_KEY, NUM_AGENT_STEPS_SAMPLED, NUM_ENV_STEPS_SAMPLED, Postprocessing,
                             SampleBatch, build_mlp, reduce_mean_valid_fn)
from copo_amd.torch_copo.algo_ccppo import (CENTRALIZED_CRITIC_OBS, COUNTERFACTUAL, CCModel, CCPPOConfig, CCPPOPolicy,  # noqa: F401
                                            CCPPOTrainer)
from copo_amd.torch_copo.algo_ippo import clipped_value_loss
from copo_amd.trainer import GraphedCallable

NEI_REWARDS = "nei_rewards"
NEI_VALUES = "nei_values"
NEI_ADVANTAGE = "nei_advantage"
NEI_TARGET = "nei_target"
LCF_LR = "lcf_lr"
GLOBAL_VALUES = "global_values"
GLOBAL_REWARDS = "global_rewards"
GLOBAL_ADVANTAGES = "global_advantages"
GLOBAL_TARGET = "global_target"
USE_CENTRALIZED_CRITIC = "use_centralized_critic"
USE_DISTRIBUTIONAL_LCF = "use_distributional_lcf"


class CoPOConfig(CCPPOConfig):
    def __init__(self, algo_class=None):
        super().__init__(algo_class=algo_class or CoPOTrainer)
        self.initial_lcf_std = 0.1
        self.lcf_sgd_minibatch_size = None
        self.lcf_num_iters = 5
        self.lcf_lr = 1e-4
        self.use_distributional_lcf = True
        self.use_centralized_critic = False
        self.fuse_mode = "none"
        self.old_value_loss = True
        # fused learner only: minibatches per batched meta launch chain (0 = one launch chain per minibatch)
        self.meta_batch_size = 32
        # fused learner only: compute the row-local part of the meta gradients once per iteration (row store)
        self.meta_row_store = True
        # fused learner, row store: the LCF steps of a chunk of minibatches run behind that chunk's dot products instead of behind
        # the pass's last chunk (local path: on; data-parallel path: off until it has been measured with real peers)
        self.meta_seq_per_chunk = True
        self.meta_seq_per_chunk_dist = False
        self.update_from_dict({"model": {"custom_model": "copo_model"}})
        # TF-era keys of train_copo.py:43-47 that the torch reference silently ignores
        self.initial_svo_std = None
        self.svo_lr = None
        self.svo_num_iters = None
        self.use_global_value = None

    def validate(self):
        assert self[USE_DISTRIBUTIONAL_LCF]
        self.update_from_dict({"env_config": {"return_native_reward": True, "lcf_dist": "normal",
                                              "lcf_normal_std": self["initial_lcf_std"]}})
        cmc = dict(self.model.get("custom_model_config") or {})
        cmc[USE_DISTRIBUTIONAL_LCF] = self[USE_DISTRIBUTIONAL_LCF]
        cmc["initial_lcf_std"] = self["initial_lcf_std"]
        self.model = {**self.model, "custom_model_config": cmc}
        super().validate()
        return self


class CoPOModel(CCModel):
    """CCModel + neighbourhood and global value nets + the LCF distribution parameters (fp64, like the reference:
    `torch.as_tensor([0.0, np.log(std)])` is a float64 tensor, algo_copo.py:121-124)."""

    def __init__(self, obs_space, action_space, num_outputs, model_config, name="copo_model"):
        super().__init__(obs_space, action_space, num_outputs, model_config, name)
        hiddens = list(model_config.get("fcnet_hiddens", [256, 256]))
        act = model_config.get("fcnet_activation", "tanh")
        cdim = self.get_centralized_critic_obs_dim()
        self.nei_value_network = self.build_one_value_network(cdim, act, hiddens)
        self.global_value_network = self.build_one_value_network(cdim, act, hiddens)
        cmc = model_config["custom_model_config"]
        if cmc[USE_DISTRIBUTIONAL_LCF]:
            init = [0.0, math.log(cmc["initial_lcf_std"])]
        else:
            init = [0.0]
        self.lcf_parameters = nn.Parameter(torch.as_tensor(init, dtype=torch.float64), requires_grad=True)

    def build_one_value_network(self, in_size, activation, hiddens):
        assert in_size > 0
        body, head, _ = build_mlp(in_size, hiddens, activation, 1, 0.01)
        return nn.Sequential(*list(body), head)

    def get_nei_value(self, centralized_critic_obs):
        return self.nei_value_network(centralized_critic_obs).reshape(-1)

    def get_global_value(self, centralized_critic_obs):
        return self.global_value_network(centralized_critic_obs).reshape(-1)

    def compute_coordinated(self, ego, neighbor, eps=None):
        """A' = cos(phi) * A_ego + sin(phi) * A_nei with phi = rsample(N(lcf_mean, lcf_std)) * pi/2 per sample."""
        if self.model_config["custom_model_config"][USE_DISTRIBUTIONAL_LCF]:
            if eps is None:
                eps = torch.randn(ego.size(), dtype=self.lcf_parameters.dtype, device=ego.device)
            lcf_rad = (self.lcf_mean + self.lcf_std * eps) * np.pi / 2
        else:
            lcf_rad = self.lcf_mean * np.pi / 2
        return torch.cos(lcf_rad) * ego + torch.sin(lcf_rad) * neighbor

    @property
    def lcf_dist(self):
        if self.model_config["custom_model_config"][USE_DISTRIBUTIONAL_LCF]:
            return torch.distributions.normal.Normal(self.lcf_mean, self.lcf_std)
        return None

    @property
    def lcf_mean(self):
        return torch.clamp(torch.tanh(self.lcf_parameters[0]), -1 + 1e-6, 1 - 1e-6)

    @property
    def lcf_std(self):
        if self.model_config["custom_model_config"][USE_DISTRIBUTIONAL_LCF]:
            return torch.exp(torch.clamp(self.lcf_parameters[1], -20, 2))
        return None


class CoPOPolicy(CCPPOPolicy):
    model_class = CoPOModel
    STAT_KEYS = CCPPOPolicy.STAT_KEYS + ("mean_nei_vf_loss", "mean_global_vf_loss", "normalized_advantages")
    META_KEYS = ("new_policy_ego_loss", "old_policy_logp_loss", "lcf_lcf_adv_loss", "lcf_final_loss", "grad_value",
                 "coordinated_adv", "global_adv")

    def __init__(self, observation_space, action_space, config):
        super().__init__(observation_space, action_space, config)
        self.target_model = self.make_model("copo_target_model").to(self.device)
        if self.fused is not None:
            self.fused.attach_target(self.target_model)
        self.update_old_policy()
        self._lcf_optimizer = torch.optim.Adam([self.model.lcf_parameters], lr=self.config[LCF_LR],
                                               capturable=self.device.type == "cuda")
        self._raw_ms = torch.tensor([0.0, 1.0], dtype=torch.float64, device=self.device)   # {mean, std} of A_c
        self._raw_lcf_adv_mean, self._raw_lcf_adv_std = self._raw_ms[0], self._raw_ms[1]
        self._lcf_adam = torch.zeros(5, dtype=torch.float64, device=self.device)         # fused path: m0 m1 v0 v1 step
        self._meta = None
        self._meta_bufs = None
        self._meta_side = None
        self._meta_aux = None           # data-parallel: the small collectives of a pass (row terms, statistics) run here
        self._meta_keep = []

    # ---- dense postprocess: three critic heads ----------------------------------------------------------
    def gae_heads(self):
        return 3

    def gae_gammas(self):
        """ego and neighbourhood heads use config.gamma, the global head gamma = 1.0 (algo_copo.py:497-500)."""
        return [float(self.config["gamma"]), float(self.config["gamma"]), 1.0]

    @torch.no_grad()
    def value_heads_dense(self, cc_flat):
        m = self.model
        with self._autocast():
            v = torch.stack([m.central_value_function(cc_flat), m.get_nei_value(cc_flat), m.get_global_value(cc_flat)])
        return v.float()

    @torch.no_grad()
    def postprocess_trajectory(self, sample_batch, other_agent_batches=None, episode=None):
        b = super().postprocess_trajectory(sample_batch, other_agent_batches, episode)
        T, E, N = b[SampleBatch.FLAGS].shape
        vals, adv, tgt = b["_vals"], b["_adv"], b["_tgt"]
        b[NEI_VALUES], b[NEI_ADVANTAGE], b[NEI_TARGET] = (x[1].view(T, E, N) for x in (vals, adv, tgt))
        b[GLOBAL_VALUES], b[GLOBAL_ADVANTAGES], b[GLOBAL_TARGET] = (x[2].view(T, E, N) for x in (vals, adv, tgt))
        return b

    def fused_adv_keys(self):
        return "normalized_advantages", GLOBAL_ADVANTAGES

    def train_columns(self):
        return super().train_columns() + [(NEI_VALUES, 1), (NEI_TARGET, 1), (GLOBAL_VALUES, 1), (GLOBAL_TARGET, 1),
                                          (NEI_ADVANTAGE, 1), (GLOBAL_ADVANTAGES, 1), ("normalized_advantages", 1)]

    # ---- PPO loss with three value heads (algo_copo.py:311-424) ---------------------------------------------
    def loss(self, model, dist_class, train_batch):
        mean = reduce_mean_valid_fn(train_batch)
        cfg = self.config
        logits, _ = model(train_batch)
        curr = dist_class(logits, model)
        ratio = torch.exp(curr.logp(train_batch[SampleBatch.ACTIONS]) - train_batch[SampleBatch.ACTION_LOGP])
        use_kl = cfg["kl_coeff"] > 0.0
        if use_kl:
            prev = dist_class(train_batch[SampleBatch.ACTION_DIST_INPUTS], model)
            mean_kl = mean(prev.kl(curr))
        else:
            mean_kl = torch.zeros((), device=ratio.device)
        entropy = curr.entropy()
        adv = train_batch["normalized_advantages"]        # LCF-coordinated, standardised over the train batch
        surrogate = torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - cfg["clip_param"], 1 + cfg["clip_param"]))
        assert cfg["use_critic"]
        cobs = train_batch[CENTRALIZED_CRITIC_OBS]
        c, old = cfg["vf_clip_param"], cfg["old_value_loss"]
        ego_vf = clipped_value_loss(model.central_value_function(cobs), train_batch[SampleBatch.VF_PREDS],
                                    train_batch[Postprocessing.VALUE_TARGETS], c, old)
        nei_vf = clipped_value_loss(model.get_nei_value(cobs), train_batch[NEI_VALUES], train_batch[NEI_TARGET], c, old)
        glob_vf = clipped_value_loss(model.get_global_value(cobs), train_batch[GLOBAL_VALUES],
                                     train_batch[GLOBAL_TARGET], c, old)
        k = cfg["vf_loss_coeff"]
        total = mean(-surrogate + k * ego_vf + k * nei_vf + k * glob_vf - self.entropy_coeff * entropy)
        if use_kl:
            total = total + self.kl_coeff * mean_kl
        st = model.tower_stats
        st["total_loss"], st["mean_policy_loss"], st["mean_vf_loss"] = total, mean(-surrogate), mean(ego_vf)
        st["vf_explained_var"] = torch.zeros((), device=ratio.device)
        st["mean_entropy"], st["mean_kl_loss"] = mean(entropy), mean_kl
        st["lcf"] = model.lcf_mean
        if cfg[USE_DISTRIBUTIONAL_LCF]:
            st["lcf_std"] = model.lcf_std
        st["mean_nei_vf_loss"], st["mean_global_vf_loss"] = mean(nei_vf), mean(glob_vf)
        st["normalized_advantages"] = mean(adv)
        return total

    # ---- LCF meta-gradient (algo_copo.py:228-309) -----------------------------------------------------------
    def _meta_pieces(self, train_batch, eps=None):
        """Everything of one meta step that is linear in the minibatch rows: returns the flat bucket
        [g_new | g_old | dS/dlcf (2) | S | n_frac] of this rank plus stats.  With the weighted mean of
        `reduce_mean_valid_fn` the bucket entries are partial sums of the GLOBAL means, so a SUM all-reduce of the
        bucket yields exactly the single-process quantities."""
        mean = reduce_mean_valid_fn(train_batch)
        cfg, model, target = self.config, self.model, self.target_model
        logits, _ = model(train_batch)
        curr = self.dist_class(logits, model)
        ratio = torch.exp(curr.logp(train_batch[SampleBatch.ACTIONS]) - train_batch[SampleBatch.ACTION_LOGP])
        adv = train_batch[GLOBAL_ADVANTAGES]
        surrogate = torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - cfg["clip_param"], 1 + cfg["clip_param"]))
        new_policy_loss = mean(-surrogate)
        g_new = torch.autograd.grad(new_policy_loss, model.policy_parameters())
        old_logits, _ = target(train_batch)
        old_logp = self.dist_class(old_logits, target).logp(train_batch[SampleBatch.ACTIONS])
        assert old_logp.ndim == 1
        old_policy_loss = mean(old_logp)
        g_old = torch.autograd.grad(old_policy_loss, target.policy_parameters())
        coordinated = model.compute_coordinated(ego=train_batch[Postprocessing.ADVANTAGES],
                                                neighbor=train_batch[NEI_ADVANTAGE], eps=eps)
        lcf_adv = (coordinated - self._raw_lcf_adv_mean) / self._raw_lcf_adv_std
        lcf_lcf_adv_loss = mean(lcf_adv)
        d_lcf = torch.autograd.grad(lcf_lcf_adv_loss, model.lcf_parameters)[0]
        flat = torch.cat([g.reshape(-1).double() for g in g_new] + [g.reshape(-1).double() for g in g_old] +
                         [d_lcf.double(), lcf_lcf_adv_loss.detach().double().reshape(1)])
        stats = dict(new_policy_ego_loss=new_policy_loss.detach(), old_policy_logp_loss=old_policy_loss.detach(),
                     coordinated_adv=mean(coordinated).detach(), global_adv=mean(adv).detach())
        return flat, stats

    def _meta_finish(self, flat, stats):
        """Dot product of the (already reduced) gradients, LCF loss gradient, Adam step on the two fp64 scalars."""
        n = (flat.numel() - 3) // 2
        grad_value = (flat[:n] * flat[n:2 * n]).sum()
        d_lcf, lcf_lcf_adv_loss = flat[2 * n:2 * n + 2], flat[2 * n + 2]
        p = self.model.lcf_parameters
        p.grad = (grad_value * d_lcf).to(p.dtype)
        self._lcf_optimizer.step()
        stats = dict(stats, lcf_lcf_adv_loss=lcf_lcf_adv_loss, lcf_final_loss=grad_value * lcf_lcf_adv_loss,
                     grad_value=grad_value)
        return stats

    def meta_update(self, train_batch, eps=None):
        """One meta step on one minibatch (eager; single process or already-global means). API of the reference."""
        tb = SampleBatch({k: (torch.as_tensor(v, device=self.device) if not torch.is_tensor(v) else v.to(self.device))
                          for k, v in train_batch.items() if k != "infos"})
        flat, stats = self._meta_pieces(tb, eps)
        D.all_reduce_sum_(flat)
        stats = self._meta_finish(flat, stats)
        m = self.model
        out = {k: (v.item() if torch.is_tensor(v) else v) for k, v in stats.items()}
        out.update(lcf=m.lcf_mean.item(), lcf_deg=m.lcf_mean.item() * 90, lcf_param=m.lcf_parameters[0].item())
        if self.config[USE_DISTRIBUTIONAL_LCF]:
            out.update(lcf_std=m.lcf_std.item(), lcf_std_deg=m.lcf_std.item() * 90,
                       lcf_std_param=m.lcf_parameters[1].item())
        return out

    # static-shape (graph-capturable) meta loop over the rows bound by prepare_sgd --------------------------------
    def _meta_step_a_fused(self):
        """Both policy gradients from the fused HIP learner (head modes META_NEW / META_OLD, no Adam), the
        two-scalar LCF part in torch fp64."""
        mb_, fz = self._meta_bufs, self.fused
        rs = dict(self._row_sources, **{k: mb_[k] for k in ("rows_all", "w_all", "denom_all", "k")})
        fz.meta_grads(rs, mb_["g_new"], mb_["g_old"], mb_["stats_new"], mb_["stats_old"], mb_["dot_partials"])
        fz.meta_lcf(rs, mb_["eps_all"], self.model.lcf_parameters.data, self._raw_ms, mb_["tail"], mb_["col_adv"],
                    mb_["col_nei_adv"])

    def _meta_step_b_fused(self):
        mb_, fz = self._meta_bufs, self.fused
        rs = dict(self._row_sources, **{k: mb_[k] for k in ("rows_all", "w_all", "denom_all", "k")})
        fz.meta_finish(rs, mb_["g_new"], mb_["g_old"], None if D.is_dist() else mb_["dot_partials"], mb_["tail"],
                       self.model.lcf_parameters.data, self._lcf_adam,
                       self.config[LCF_LR], mb_["stats_new"], mb_["stats_old"], mb_["stats"])

    def _meta_step_a(self):
        if self.fused is not None:
            return self._meta_step_a_fused()
        mb_ = self._meta_bufs
        rs = dict(self._row_sources, **{k: mb_[k] for k in ("rows_all", "w_all", "denom_all", "k")})
        saved, self._row_sources = self._row_sources, rs
        try:
            tb = self._gather_minibatch()
        finally:
            self._row_sources = saved
        eps = mb_["eps_all"].index_select(0, mb_["k"]).view(-1)
        flat, stats = self._meta_pieces(tb, eps)
        mb_["flat"].copy_(flat)
        mb_["stats_a"].copy_(torch.stack([stats[k].double().reshape(()) for k in
                                          ("new_policy_ego_loss", "old_policy_logp_loss", "coordinated_adv", "global_adv")]))

    def _meta_step_b(self):
        if self.fused is not None:
            return self._meta_step_b_fused()
        mb_ = self._meta_bufs
        st = self._meta_finish(mb_["flat"], {})
        a = mb_["stats_a"]
        mb_["stats"].add_(torch.stack([a[0], a[1], st["lcf_lcf_adv_loss"], st["lcf_final_loss"], st["grad_value"],
                                       a[2], a[3]]))
        mb_["k"].add_(1)

    def _meta_step_local(self):
        if self.fused is not None and not D.is_dist():
            # single process: gradients, LCF terms, dot product and the LCF Adam step in one call (six launches)
            mb_, fz = self._meta_bufs, self.fused
            rs = dict(self._row_sources, **{k: mb_[k] for k in ("rows_all", "w_all", "denom_all", "k")})
            fz.meta_step(rs, mb_["g_new"], mb_["g_old"], mb_["stats_new"], mb_["stats_old"], mb_["dot_partials"],
                         mb_["eps_all"], self.model.lcf_parameters.data, self._raw_ms, mb_["tail"], mb_["col_adv"],
                         mb_["col_nei_adv"], self._lcf_adam, self.config[LCF_LR], mb_["stats"])
            return
        self._meta_step_a()
        self._meta_step_b()

    def _meta_per_chunk(self):
        """Local: the LCF steps of a chunk follow its dot products on a side stream.  Data-parallel (row store): the same function runs
        the pass with ONE exchange of all its gradient pairs (`meta_dist_exchange` = "pass", the default) or chunk by chunk ("chunk",
        also selected by the older `meta_seq_per_chunk_dist`); "off" keeps the round-4 loop (exchange and dot products per chunk, the
        LCF steps of a pass in one launch behind its last chunk)."""
        if not D.is_dist():
            return bool(self.config.get("meta_seq_per_chunk", True))
        return self._meta_dist_exchange() != "off"

    def _meta_dist_exchange(self):
        if self.config.get("meta_seq_per_chunk_dist", False):
            return "chunk"
        return str(self.config.get("meta_dist_exchange", "pass"))

    def _run_meta_batched(self, n_mb, nb):
        """One meta iteration the batched way: the gradient pairs of `nb` minibatches per launch chain (they do not
        depend on the LCF parameters), then all `n_mb` sequential LCF Adam steps in one kernel.  Same results as
        `n_mb` calls of `_meta_step_local`."""
        mb_, fz = self._meta_bufs, self.fused
        rs = dict(self._row_sources, **{k: mb_[k] for k in ("rows_all", "w_all", "denom_all")})
        # with the row store (filled once per training iteration by run_meta) a pass only redoes the weight-gradient GEMMs
        grads = fz.meta_batch_wgrads if self._meta_row_store else fz.meta_batch_grads
        pack = self._row_sources["pack"]
        # LCF steps chunk by chunk behind each chunk's dot products: on by default in the local path (meta passes 4.1 -> 3.75 ms); in the
        # data-parallel path it is an option -- with ONE rank forced through that path the extra launches per chunk make the host the
        # bottleneck (5.65 -> 6.2 ms); with real peers a chunk also waits for its all-reduce, which has not been measured
        per_chunk = self._meta_row_store and self._meta_per_chunk()
        if not D.is_dist() or per_chunk:
            # the sequential kernel streams dense {A_ego, A_nei} rows instead of chasing row indices into the pack
            rows = mb_["rows_all"][:n_mb]
            if per_chunk:
                en_src = mb_.get("en_src")          # (run_meta gathers the two columns once per iteration)
                if en_src is None:
                    en_src = pack[:, [mb_["col_adv"], mb_["col_nei_adv"]]].contiguous()
                en = en_src[rows].unsqueeze(0)
            else:
                en = torch.stack([pack[:, mb_["col_adv"]][rows], pack[:, mb_["col_nei_adv"]][rows]], dim=-1).unsqueeze(0).contiguous()
            if per_chunk:
                # the LCF steps of a chunk start as soon as its dot products exist (side stream), not after the pass's last chunk:
                # what is left exposed at the end of the last pass is one chunk's steps (~80 us), not a pass's (~400 us)
                self._meta_lcf_chunks(n_mb, nb, grads, rs, en)
                return
            for c0 in range(0, n_mb, nb):
                grads(rs, c0, min(nb, n_mb - c0), mb_["gv"], mb_["stats_k"])
            self._meta_lcf_async(n_mb, en, mb_["w_all"][:n_mb].unsqueeze(0), mb_["eps_all"][:n_mb].unsqueeze(0))
            return
        # data-parallel: the minibatch gradients are sums over the ranks' rows -> all-reduce the exported gradient pairs
        # of a whole chunk BEFORE their dot products; the LCF row terms of every rank are gathered once per iteration
        # and every rank runs the (cheap, sequential) LCF steps on the full rows, so the parameters stay identical.
        nf, S, mb = fz.meta_fold_len(), D.world_size(), mb_["mb"]
        if mb_.get("g_chunk") is None or mb_["g_chunk"][0].shape[0] < nb:
            mb_["g_chunk"] = [torch.zeros(nb, 2, nf, dtype=torch.float32, device=self.device) for _ in range(2)]
        # two export buffers: the all-reduce of chunk c (23 MB of gradient pairs at the bench shape -- over xGMI about as long
        # as a chunk's GEMMs) runs on the collective's own stream under the gradient GEMMs of chunk c + 1
        pending = None

        def finish(c0, n, buf, work):
            if work is not None:
                work.wait()               # (the compute stream waits; the host does not)
            # (exported gradients of the row store carry unit row weights: both factors 1 / D_k, applied by the kernel)
            fz.meta_batch_dot(buf, nf, n, mb_["gv"][c0:], denom=mb_["denom_all"][c0:] if self._meta_row_store else None)

        if self._meta_row_store:
            fz.meta_rowstat(rs, 0, n_mb, mb_["stats_k"])      # the pass's statistics in one launch (summed over the ranks below)
        for q, c0 in enumerate(range(0, n_mb, nb)):
            n = min(nb, n_mb - c0)
            buf = mb_["g_chunk"][q & 1]
            grads(rs, c0, n, mb_["gv"], None if self._meta_row_store else mb_["stats_k"], g_out=buf)
            work = D.all_reduce_sum_async(buf[:n])
            if pending is not None:
                finish(*pending)
            pending = (c0, n, buf, work)
        if pending is not None:
            finish(*pending)
        D.all_reduce_sum_(mb_["stats_k"][:n_mb])
        rows = mb_["rows_all"][:n_mb]
        en = torch.stack([pack[:, mb_["col_adv"]][rows], pack[:, mb_["col_nei_adv"]][rows]], dim=-1).contiguous()
        en_all = D.all_gather_into_(torch.empty((S,) + tuple(en.shape), dtype=en.dtype, device=self.device), en)
        w_all = D.all_gather_into_(torch.empty(S, n_mb, mb, dtype=torch.float32, device=self.device),
                                   mb_["w_all"][:n_mb].contiguous())
        eps_all = D.all_gather_into_(torch.empty(S, n_mb, mb, dtype=torch.float64, device=self.device),
                                     mb_["eps_all"][:n_mb].contiguous())
        self._meta_lcf_async(n_mb, en_all, w_all, eps_all)

    def _side_stream(self):
        """The stream of the sequential LCF kernels: one that is on another hardware queue than the main stream (trainer.concurrent_stream)."""
        from copo_amd.trainer import concurrent_stream
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=self.device)
        return concurrent_stream(self.device)

    def _meta_lcf_chunks(self, n_mb, nb, grads, rs, en):
        """Phase A chunk by chunk on the main stream, phase B of every chunk on the side stream behind that chunk's event.  The dot
        products / statistics of a pass go to one of two buffer sets (passes alternate), so the next pass's GEMMs never write what the
        side stream may still read; everything else the kernel reads is private to the pass.  Data-parallel: the row terms of all
        ranks are gathered BEFORE the chunks (they do not depend on the GEMMs), a chunk's LCF steps follow its all-reduced dot
        products; every rank runs them on the full rows, so the parameters stay identical."""
        mb_, fz = self._meta_bufs, self.fused
        dist = D.is_dist()
        if self._meta_side is None:
            self._meta_side = self._side_stream()
        if mb_.get("gv2") is None:
            mb_["gv2"] = [mb_["gv"], torch.zeros_like(mb_["gv"])]
            mb_["stats_k2"] = [mb_["stats_k"], torch.zeros_like(mb_["stats_k"])]
        # (run_meta switched the planned tables to set q and waited for the pass that used it last; a caller that drives the passes
        # itself must not write rows_all / w_all / denom_all / eps_all before the side stream has taken the pass's last chunk)
        q = mb_.setdefault("pass_no", 0) & 1
        mb_.setdefault("pass_done", [None, None])
        mb_["pass_no"] += 1
        gv, stats_k = mb_["gv2"][q], mb_["stats_k2"][q]
        per_pass = dist and self._meta_dist_exchange() == "pass"
        aux_done = None

        def row_terms():
            """The pass's statistics and, data-parallel, every rank's row terms (they do not depend on the GEMMs)."""
            fz.meta_rowstat(rs, 0, n_mb, stats_k)
            if not dist:
                return en, mb_["w_all"][:n_mb].unsqueeze(0), mb_["eps_all"][:n_mb].unsqueeze(0)
            S, mb = D.world_size(), mb_["mb"]
            D.all_reduce_sum_(stats_k[:n_mb])
            e = D.all_gather_into_(torch.empty((S,) + tuple(en.shape[1:]), dtype=en.dtype, device=self.device), en[0].contiguous())
            w = D.all_gather_into_(torch.empty(S, n_mb, mb, dtype=torch.float32, device=self.device), mb_["w_all"][:n_mb].contiguous())
            x = D.all_gather_into_(torch.empty(S, n_mb, mb, dtype=torch.float64, device=self.device), mb_["eps_all"][:n_mb].contiguous())
            return e, w, x

        if per_pass:
            # ONE exchange per pass: every chunk exports its gradient pairs into the pass's buffer ([n_mb][2][n], ~105 MB at the bench
            # shape -- two of them, passes alternate); the ranks then share the pass's dot products: a reduce-scatter leaves every rank
            # the summed pairs of n_mb / S minibatches (half the wire bytes of an all-reduce), it takes their dot products and an
            # all-gather of n_mb doubles completes `gv`.  NOTHING of this is on the main stream: the four small collectives (statistics,
            # row terms) run on `aux` as soon as the pass is planned, the exchange, the dot products and the pass's LCF steps on `side`
            # behind the last chunk's event -- the main stream goes from this pass's GEMMs straight to the next pass's.
            nf = fz.meta_fold_len()
            S = D.world_size()
            parts_cap = max(1, int(self.config.get("meta_dist_parts", 2)))
            cap = -(-(-(-int(mb_["max_mb"]) // parts_cap)) // S) * S          # minibatches per part, whole shares: rows beyond a part's own are summed and ignored
            if mb_.get("g_pass") is None or mb_["g_pass"][0][0].shape[0] < cap or len(mb_["g_pass"][0]) < parts_cap:
                # (two sets, passes alternate; a buffer per part: a part's exchange may still be on the wire while the next part's GEMMs export)
                mb_["g_pass"] = [[torch.zeros(cap, 2, nf, dtype=torch.float32, device=self.device) for _ in range(parts_cap)] for _ in range(2)]
                mb_["g_mine"] = torch.zeros(cap // S, 2, nf, dtype=torch.float32, device=self.device) if S > 1 else None
                mb_["denom_pad"] = torch.ones(cap, dtype=torch.float32, device=self.device)
                mb_["gv_pad"] = torch.zeros(cap, dtype=torch.float64, device=self.device)
            if self._meta_aux is None:
                self._meta_aux = self._side_stream()
            ev0 = torch.cuda.Event()
            ev0.record()
            # (`en` is a main-stream temporary of the caller that the aux stream reads: the caching allocator must not hand its block
            #  out again before aux is past it -- c10d no longer record_stream()s the inputs of synchronous collectives)
            en.record_stream(self._meta_aux)
            with torch.cuda.stream(self._meta_aux):
                self._meta_aux.wait_event(ev0)
                en_d, w_d, eps_d = row_terms()
                for t in (en_d, w_d, eps_d):
                    t.record_stream(self._meta_side)
                aux_done = torch.cuda.Event()
                aux_done.record()
        else:
            en_d, w_d, eps_d = row_terms()
            if dist:
                nf = fz.meta_fold_len()
                if mb_.get("g_chunk") is None or mb_["g_chunk"][0].shape[0] < nb:
                    mb_["g_chunk"] = [torch.zeros(nb, 2, nf, dtype=torch.float32, device=self.device) for _ in range(2)]
        priv = dict(denom=mb_["denom_all"], en=en_d, w=w_d, eps=eps_d)       # (set q of the planned tables: not written before pass_done[q])

        def lcf_steps(c0, n):
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._meta_side):
                self._meta_side.wait_event(ev)
                fz.meta_batch_lcf(dict(denom_all=priv["denom"]), n_mb, None, gv, stats_k,
                                  self.model.lcf_parameters.data, self._raw_ms, self._lcf_adam, self.config[LCF_LR], mb_["stats"],
                                  0, 0, dense=(en_d, w_d, eps_d), k_first=c0, k_count=n)

        def finish(c0, n, buf, work):       # data-parallel: the chunk's gradient pairs are summed over the ranks, then their dot products
            if work is not None:
                work.wait()               # (the compute stream waits; the host does not)
            fz.meta_batch_dot(buf, nf, n, gv[c0:], denom=mb_["denom_all"][c0:])
            lcf_steps(c0, n)

        if per_pass:
            # `meta_dist_parts` exchanges per pass (default 2): the LCF steps of the first part run under the GEMMs of the second, so what
            # the last pass leaves exposed at the end of the iteration is half a pass of LCF steps, not a whole one
            parts = max(1, min(int(self.config.get("meta_dist_parts", 2)), n_mb))
            bounds = [n_mb * i // parts for i in range(parts + 1)]
            done = torch.cuda.Event()
            for pi in range(parts):
                k0, k1 = bounds[pi], bounds[pi + 1]
                buf = mb_["g_pass"][q][pi]
                for c0 in range(k0, k1, nb):
                    n = min(nb, k1 - c0)
                    grads(rs, c0, n, gv, None, g_out=buf[c0 - k0:c0 - k0 + n])
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._meta_side):
                    self._meta_side.wait_event(ev)
                    self._meta_shared_dots(buf, nf, k0, k1 - k0, gv)
                    if pi == 0:
                        self._meta_side.wait_event(aux_done)
                    fz.meta_batch_lcf(dict(denom_all=priv["denom"]), n_mb, None, gv, stats_k,
                                      self.model.lcf_parameters.data, self._raw_ms, self._lcf_adam, self.config[LCF_LR], mb_["stats"],
                                      0, 0, dense=(en_d, w_d, eps_d), k_first=k0, k_count=k1 - k0)
                    if pi == parts - 1:
                        done.record()
            mb_["pass_done"][q] = done
            mb_["gv"], mb_["stats_k"] = gv, stats_k
            self._meta_keep.append(priv)
            return
        pending = None
        for j, c0 in enumerate(range(0, n_mb, nb)):
            n = min(nb, n_mb - c0)
            if dist:
                # two export buffers: the all-reduce of chunk c runs on the collective's own stream under the GEMMs of chunk c + 1
                buf = mb_["g_chunk"][j & 1]
                grads(rs, c0, n, gv, None, g_out=buf)
                work = D.all_reduce_sum_async(buf[:n])
                if pending is not None:
                    finish(*pending)
                pending = (c0, n, buf, work)
            else:
                grads(rs, c0, n, gv, None)
                lcf_steps(c0, n)
        if pending is not None:
            finish(*pending)
        done = torch.cuda.Event()
        with torch.cuda.stream(self._meta_side):
            done.record()
        mb_["pass_done"][q] = done
        mb_["gv"], mb_["stats_k"] = gv, stats_k          # (what the callers / tests read after the pass)
        self._meta_keep.append(priv)            # alive until the side stream has been joined

    def _meta_shared_dots(self, buf, nf, k0, n, gv):
        """gv[k0 : k0 + n] = <sum over ranks of g_new, sum over ranks of g_old> of the n minibatches whose exported pairs `buf` holds, the
        work shared by the ranks: rank r receives the summed pairs of minibatches [r c, (r + 1) c) of the part (reduce-scatter; a backend
        without it all-reduces), takes their dot products with the kernel every other path uses -- so a minibatch's value does not depend
        on who computed it -- and the values are gathered.  Runs on the current (side) stream."""
        mb_, fz = self._meta_bufs, self.fused
        S, r = D.world_size(), D.rank()
        mb_["denom_pad"][:n].copy_(mb_["denom_all"][k0:k0 + n])
        if S == 1:
            fz.meta_batch_dot(buf, nf, n, gv[k0:], denom=mb_["denom_pad"])
            return
        c = -(-n // S)
        mine = D.reduce_scatter_sum_(mb_["g_mine"][:c], buf[:S * c])
        part = mb_["gv_pad"][r * c:(r + 1) * c]
        fz.meta_batch_dot(mine, nf, c, part, denom=mb_["denom_pad"][r * c:])
        every = D.all_gather_into_(torch.empty(S, c, dtype=torch.float64, device=self.device), part)
        gv[k0:k0 + n].copy_(every.reshape(-1)[:n])

    def _meta_lcf_async(self, n_mb, en, w, eps):
        """Phase B of this pass on a side stream: the sequential LCF kernel keeps ONE compute unit busy for ~0.4 ms, and
        the next pass's gradient GEMMs do not depend on the LCF parameters -- so they run meanwhile on the main stream.
        Everything the kernel reads is private to the pass (fresh tensors / copies); run_meta joins the stream."""
        mb_, fz = self._meta_bufs, self.fused
        if self._meta_side is None:
            self._meta_side = self._side_stream()
        priv = dict(gv=mb_["gv"][:n_mb].clone(), stats_k=mb_["stats_k"][:n_mb].clone(), denom=mb_["denom_all"][:n_mb].clone(),
                    en=en, w=w.clone(), eps=eps.clone())
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._meta_side):
            self._meta_side.wait_event(ev)
            fz.meta_batch_lcf(dict(denom_all=priv["denom"]), n_mb, None, priv["gv"], priv["stats_k"],
                              self.model.lcf_parameters.data, self._raw_ms, self._lcf_adam, self.config[LCF_LR], mb_["stats"],
                              0, 0, dense=(priv["en"], priv["w"], priv["eps"]))
        self._meta_keep.append(priv)            # alive until the side stream has been joined

    def _wants_row_store(self, mb, num_iters):
        rs = self._row_sources
        nb_batch = int(self.config.get("meta_batch_size", 32)) if self.fused is not None else 0
        # the row store costs 8 H bytes per row and net; beyond ~1M rows recomputing per pass is the better trade
        return nb_batch > 0 and bool(self.config.get("meta_row_store", True)) and num_iters > 1 \
            and mb == rs["mb"] and int(rs["max_rows"]) <= int(self.config.get("meta_row_store_max_rows", 1 << 20))

    def meta_rows_early(self, mb, num_iters):
        """Queue the row store of the coming `run_meta` call NOW (it needs the parameters the PPO epochs leave and nothing from the
        host): the caller puts this between the PPO epochs and the read of their statistics, so the device goes from the last SGD
        step straight into 0.7 ms of row-pass kernels while the host wakes up and queues the passes.  Four launches on the MAIN
        stream only -- queuing the whole meta phase that early (second queue, ~150 launches) slows the SGD chains (training_step)."""
        if self.fused is None or self._meta_bufs is None or self._meta_bufs["mb"] != mb or not self._wants_row_store(mb, num_iters) \
                or int(self.config.get("meta_batch_size", 32)) <= 0:
            return False
        self.fused.meta_rows(self._row_sources)
        self._rows_early = True
        return True

    def run_meta(self, valid_idx, B_local, B_all, mb, num_iters, defer=False, extra=()):
        """`lcf_num_iters` passes of shuffled minibatches through `meta_update` (algo_copo.py:581-589).
        defer=True: everything is queued and a callable is returned that does the ONE device -> host read and builds the result;
        `extra`: 1-D device tensors whose values ride along in that read (the callable leaves them in `self._extra_host`), so that an
        iteration ends with one host stop instead of three (PPO statistics, meta results, episode metrics)."""
        rs = self._row_sources
        dev = self.device
        max_mb = max(rs["max_mb"], 1) if mb == rs["mb"] else max(1, math.ceil(rs["max_mb"] * rs["mb"] / mb))
        n_pol = sum(p.numel() for p in self.model.policy_parameters())
        if self._meta_bufs is None or self._meta_bufs["mb"] != mb or self._meta_bufs["max_mb"] != max_mb:
            self._meta_bufs = dict(
                mb=mb, max_mb=max_mb, rows_all=torch.zeros(max_mb, mb, dtype=torch.int64, device=dev),
                w_all=torch.zeros(max_mb, mb, dtype=torch.float32, device=dev),
                denom_all=torch.ones(max_mb, dtype=torch.float32, device=dev),
                k=torch.zeros(1, dtype=torch.int64, device=dev),
                eps_all=torch.zeros(max_mb, mb, dtype=torch.float64, device=dev),
                flat=torch.zeros(2 * n_pol + 3, dtype=torch.float64, device=dev),
                stats_a=torch.zeros(4, dtype=torch.float64, device=dev),
                stats=torch.zeros(len(self.META_KEYS), dtype=torch.float64, device=dev), n_pol=n_pol)
            if self.fused is not None:
                assert mb == self.fused.cfg.mb, "the fused learner is built for one minibatch size"
                cols, off = {}, 0
                for name, wdt in self.train_columns():
                    cols[name] = off
                    off += wdt
                nflat = self.fused.flat.numel
                # one buffer: the policy blocks of g_new / g_old are adjacent -> a single [2 * n_pol] all-reduce
                g_both = torch.zeros(n_pol + nflat, device=dev)
                self._meta_bufs.update(
                    g_both=g_both, g_new=g_both[:nflat], g_old=g_both[n_pol:], tail=torch.zeros(4, dtype=torch.float64, device=dev),
                    dot_partials=torch.zeros(8192, dtype=torch.float64, device=dev),
                    stats_new=torch.zeros(8, device=dev), stats_old=torch.zeros(8, device=dev),
                    col_adv=cols[Postprocessing.ADVANTAGES], col_nei_adv=cols[NEI_ADVANTAGE],
                    gv=torch.zeros(max_mb, dtype=torch.float64, device=dev),
                    stats_k=torch.zeros(max_mb, 2, 8, dtype=torch.float32, device=dev))
            self._meta = None
        if self._meta is None:
            if D.is_dist():
                self._meta = (GraphedCallable(self._meta_step_a, self.use_graphs),
                              GraphedCallable(self._meta_step_b, self.use_graphs))
            else:
                self._meta = GraphedCallable(self._meta_step_local, self.use_graphs)
        mbuf = self._meta_bufs
        mbuf["stats"].zero_()
        steps = 0
        nb_batch = int(self.config.get("meta_batch_size", 32)) if self.fused is not None else 0
        self._meta_row_store = self._wants_row_store(mb, num_iters)
        early, self._rows_early = getattr(self, "_rows_early", False), False
        if self._meta_row_store and not early:
            self.fused.meta_rows(rs)
        perms = self.draw_perms(num_iters, B_local)
        chunked = self._meta_row_store and self._meta_per_chunk()
        if chunked:
            # {A_ego, A_nei} of every row once per iteration: a pass then gathers its dense row terms with ONE index op
            if mbuf.get("en_cols") is None:      # (a Python index list would cost a host -> device copy, i.e. a host stop, per iteration)
                mbuf["en_cols"] = torch.tensor([mbuf["col_adv"], mbuf["col_nei_adv"]], dtype=torch.int64, device=dev)
            mbuf["en_src"] = rs["pack"].index_select(1, mbuf["en_cols"])
            if mbuf.get("sets") is None:
                # the planned tables of a pass are read by its LCF steps on the side stream while the next pass is being planned:
                # two sets of tables, passes alternate (no private copies per pass)
                keys = ("rows_all", "w_all", "denom_all", "eps_all")
                mbuf["sets"] = [{k: mbuf[k] for k in keys}, {k: torch.zeros_like(mbuf[k]) for k in keys}]
                mbuf["sets"][1]["denom_all"].fill_(1.0)
                mbuf["pass_no"], mbuf["pass_done"] = 0, [None, None]
        for it in range(num_iters):
            if chunked:
                q = mbuf["pass_no"] & 1
                if mbuf["pass_done"][q] is not None:      # the pass that used this set two passes ago has long finished: cheap
                    torch.cuda.current_stream().wait_event(mbuf["pass_done"][q])
                mbuf.update(mbuf["sets"][q])
            n_mb = self.plan_epoch(valid_idx, B_local, B_all, mb, bufs=mbuf, perm=None if perms is None else perms[it])
            mbuf["eps_all"].normal_()
            if nb_batch > 0:
                self._run_meta_batched(n_mb, nb_batch)
                steps += n_mb
                continue
            for _k in range(n_mb):
                if D.is_dist():
                    self._meta[0]()
                    if self.fused is not None:
                        D.all_reduce_sum_(mbuf["g_both"][:2 * mbuf["n_pol"]])
                        D.all_reduce_sum_(mbuf["tail"])
                    else:
                        D.all_reduce_sum_(mbuf["flat"])
                    self._meta[1]()
                else:
                    self._meta()
                steps += 1
        if self._meta_side is not None:
            torch.cuda.current_stream().wait_stream(self._meta_side)
        if self._meta_aux is not None:
            torch.cuda.current_stream().wait_stream(self._meta_aux)
        self._meta_keep.clear()
        m = self.model
        # one device -> host read for everything this iteration reports about the meta update (+ the caller's `extra`); the LCF mean /
        # std are the model's formulas (CoPOModel.lcf_mean / lcf_std) applied to the parameters on the host, in float64 like there
        extra = [t.detach().reshape(-1).double() for t in extra]
        packed = torch.cat([mbuf["stats"] / max(1, steps), m.lcf_parameters.detach().double().reshape(-1),
                            self._raw_ms.double().reshape(-1)] + extra)
        nk = len(self.META_KEYS)
        distributional = bool(m.model_config["custom_model_config"][USE_DISTRIBUTIONAL_LCF])

        def resolve():
            host = packed.tolist()
            out = dict(zip(self.META_KEYS, host[:nk]))
            p0, p1, self._raw_host = host[nk], host[nk + 1], (host[nk + 2], host[nk + 3])
            lm = min(max(math.tanh(p0), -1 + 1e-6), 1 - 1e-6)
            out.update(lcf=lm, lcf_deg=lm * 90, lcf_param=p0)
            if distributional:
                ls = math.exp(min(max(p1, -20.0), 2.0))
                out.update(lcf_std=ls, lcf_std_deg=ls * 90, lcf_std_param=p1)
            else:
                out.update(lcf_std=None)
            o, self._extra_host = nk + 4, []
            for t in extra:
                self._extra_host.append(host[o:o + t.numel()])
                o += t.numel()
            return out
        return resolve if defer else resolve()

    def update_old_policy(self):
        fz = self.fused
        if fz is not None and fz.target_flat is not None:     # one flat copy instead of 24 tensor copies
            with torch.no_grad():
                fz.target_flat.flat.copy_(fz.flat.flat)
                self.target_model.lcf_parameters.data.copy_(self.model.lcf_parameters.data)
        else:
            self.target_model.load_state_dict(self.model.state_dict())
            if fz is not None:
                fz.invalidate_mirror()

    def assign_lcf(self, lcf_parameters, lcf_mean, lcf_std=None, my_name=None):
        """Copy LCF parameters into this policy and check the derived mean/std (algo_copo.py:446-471)."""
        lcf_parameters = lcf_parameters.to(self.device)
        assert self.model.lcf_parameters.size() == lcf_parameters.size()
        with torch.no_grad():
            self.model.lcf_parameters.data.copy_(lcf_parameters)
        if lcf_mean is not None and os.environ.get("COPO_CHECK_ASSIGN_LCF", "0") == "1":     # the reference's sanity check costs two device reads
            new_mean = self.model.lcf_mean.item()
            assert abs(new_mean - lcf_mean) < 1e-5, (new_mean, lcf_mean)
            if lcf_std is not None:
                new_std = self.model.lcf_std.item()
                assert abs(new_std - lcf_std) < 1e-5, (new_std, lcf_std)

    def get_state(self):
        st = super().get_state()
        st.update(target_model=self.target_model.state_dict(), lcf_optimizer=self._lcf_optimizer.state_dict(),
                  lcf_adam=self._lcf_adam.clone())
        return st

    def set_state(self, state):
        super().set_state(state)
        self.target_model.load_state_dict(state["target_model"])
        self._lcf_optimizer.load_state_dict(state["lcf_optimizer"])
        if "lcf_adam" in state:
            self._lcf_adam.copy_(state["lcf_adam"])
        if self._meta is not None:
            self._meta = None


def compute_nei_advantage(rollout, last_r, gamma=0.9, lambda_=1.0):
    """Single-trajectory API of the reference (algo_copo.py:189-195), served by the HIP scan."""
    return _single_traj_gae(rollout, last_r, gamma, lambda_, NEI_REWARDS, NEI_VALUES, NEI_ADVANTAGE, NEI_TARGET)


def compute_global_advantage(rollout, last_r, gamma=1.0, lambda_=1.0):
    return _single_traj_gae(rollout, last_r, gamma, lambda_, GLOBAL_REWARDS, GLOBAL_VALUES, GLOBAL_ADVANTAGES,
                            GLOBAL_TARGET)


def _single_traj_gae(rollout, last_r, gamma, lambda_, rk, vk, ak, tk):
    from copo_amd import ops
    r = torch.as_tensor(rollout[rk], dtype=torch.float32).cuda().reshape(1, -1, 1)
    v = torch.as_tensor(rollout[vk], dtype=torch.float32).cuda().reshape(1, -1, 1)
    T = r.shape[1]
    flags = torch.ones(T, 1, dtype=torch.uint8, device=r.device)
    last = float(last_r)
    own = float(v[0, -1, 0])
    # the scan bootstraps a truncated trajectory from the value of its LAST row (what the reference's caller passes,
    # algo_copo.py:492-496) and a finished one from 0; any other `last_r` is honoured through the linearity of GAE in the
    # bootstrap value: A_t += (gamma * lambda)^(T-1-t) * gamma * last_r on top of the finished-trajectory result
    general = last != 0.0 and last != own
    if last == 0.0 or general:
        flags[-1, 0] |= 2
    adv, tgt = ops.gae3(r.contiguous(), v.contiguous(), flags, [gamma], lambda_)
    if general:
        k = torch.arange(T - 1, -1, -1, dtype=torch.float64, device=r.device)
        corr = (torch.pow(torch.tensor(float(gamma) * float(lambda_), dtype=torch.float64, device=r.device), k)
                * float(gamma) * last).to(torch.float32).reshape(adv.shape)
        adv, tgt = adv + corr, tgt + corr
    rollout[ak] = adv.reshape(-1).cpu().numpy()
    rollout[tk] = tgt.reshape(-1).cpu().numpy()
    return rollout


class CoPOTrainer(CCPPOTrainer):
    _name = "CoPO"

    @classmethod
    def get_default_config(cls):
        return CoPOConfig()

    def get_default_policy_class(self, config):
        assert config["framework"] == "torch"
        return CoPOPolicy

    def coordinated_advantage(self, batch, valid):
        """A_c = cos(lcf*pi/2) A_ego + sin(lcf*pi/2) A_nei, its batch mean/std for the meta update, and the
        standardised A_c / A_glob (algo_copo.py:539-551).  Statistics are global over all ranks."""
        from copo_amd import ops
        pol = self.policy
        dev = pol.device
        n = batch[SampleBatch.FLAGS].numel()
        if getattr(self, "_mix_ws", None) is None or self._mix_ws["n"] != n:
            self._mix_ws = dict(n=n, stats=ops.lcf_stats_workspace(dev), mixed=torch.empty(n, device=dev),
                                norm=torch.empty(n, device=dev), gstd=torch.empty(n, device=dev),
                                valid=torch.empty(n, dtype=torch.uint8, device=dev))
        ws = self._mix_ws
        ws["valid"].copy_(valid)
        adv = batch[Postprocessing.ADVANTAGES].reshape(-1)
        nei = batch[NEI_ADVANTAGE].reshape(-1)
        glob = batch[GLOBAL_ADVANTAGES].reshape(-1).contiguous()
        lcf = batch["step_lcf"].reshape(-1)
        ops.lcf_mix_partial(adv.contiguous(), nei.contiguous(), glob, lcf.contiguous(), ws["valid"], ws["mixed"], ws["stats"])
        D.all_reduce_sum_(ws["stats"][:6])
        ops.lcf_mix_apply(ws["mixed"], glob, ws["valid"], ws["stats"], ws["norm"], ws["gstd"])
        s = ws["stats"]
        mean = s[1] / s[0]
        std = torch.clamp(torch.sqrt(torch.clamp(s[2] / s[0] - mean * mean, min=0.0)), min=1e-4)
        pol._raw_lcf_adv_mean.copy_(mean)
        pol._raw_lcf_adv_std.copy_(std)
        shape = batch[SampleBatch.FLAGS].shape
        batch["raw_normalized_advantages"] = ws["mixed"].view(shape)
        batch["normalized_advantages"] = ws["norm"].view(shape)
        batch[GLOBAL_ADVANTAGES] = ws["gstd"].view(shape)

    def training_step(self):
        import time
        cfg, pol = self.config, self.policy
        batch = self.collect()
        valid, idx, B = self.valid_rows(batch)
        B_all = D.all_gather_int(B, pol.device)
        self._counters[NUM_AGENT_STEPS_SAMPLED] += sum(B_all)
        self._counters[NUM_ENV_STEPS_SAMPLED] += self.sampler.T * self.sampler.E * D.world_size()
        # ---- local coordination: LCF-weighted advantage ----
        self.coordinated_advantage(batch, valid)
        # ---- PPO epochs ----
        t0 = time.perf_counter()
        mb = int(cfg["sgd_minibatch_size"])
        pol.prepare_sgd(batch, batch[SampleBatch.FLAGS].numel(), mb)
        # The epochs' statistics are read back HERE, i.e. the host stops behind the PPO epochs before it queues the meta passes.
        # Reading them later (run_sgd(defer=True): the host then queues the whole meta phase while the captured SGD chains still
        # run) closes the ~0.2 ms of launch gaps at the start of the meta phase but costs 1.1 ms per iteration elsewhere (free-running
        # iterations 26.2 -> 27.3 ms on one box, same synchronised phase times: profiles/r06_meta_pass.txt) -- measured, not kept.
        # What IS queued before that read: the meta phase's row store (main stream only, four launches), behind an asynchronous
        # copy of the statistics -- the host wakes up on the copy's event while the device is already in the row pass.
        lcf_mb = int(cfg["lcf_sgd_minibatch_size"] or cfg["sgd_minibatch_size"])
        pending = pol.run_sgd(idx, B, B_all, mb, int(cfg["num_sgd_iter"]), defer=True)
        if hasattr(pending, "start_copy") and not D.is_dist():
            pending.start_copy()
            pol.meta_rows_early(lcf_mb, int(cfg["lcf_num_iters"]))
        stats = pending()
        self._timers["learn_time_ms"] = (time.perf_counter() - t0) * 1e3
        # ---- global coordination: LCF meta update ----
        t0 = time.perf_counter()
        lcf_mb = int(cfg["lcf_sgd_minibatch_size"] or cfg["sgd_minibatch_size"])
        # ONE host stop at the end of the iteration: the episode-metric sums ride along in the meta pass's read
        extra = []
        early = getattr(self, "_metric_sums", None)
        if early is not None and early[1] is not None:
            self._wait_metric_sums()
            extra.append(early[1])
        pending_meta = pol.run_meta(idx, B, B_all, lcf_mb, int(cfg["lcf_num_iters"]), defer=True, extra=extra)
        # the device half of the reference's weight / LCF broadcast (algo_copo.py:555-558, 596-613) is queued BEFORE that read
        lcf_parameters = pol.model.lcf_parameters.detach().clone()

        def _update_lcf_dev(w_id, w):
            def _update_lcf_1(pi, pi_id):
                pi.assign_lcf(lcf_parameters, None)
                pi.update_old_policy()
            w.foreach_policy(_update_lcf_1)

        self.workers.foreach_worker_with_id(_update_lcf_dev)
        meta = pending_meta()
        self._timers["meta_time_ms"] = (time.perf_counter() - t0) * 1e3
        host = list(getattr(pol, "_extra_host", []))
        if early is not None and early[1] is not None:
            self._metric_sums = (early[0], host.pop(0))
        train_results = {"default": {LEARNER_STATS_KEY: stats, "custom_metrics": {}}}
        lcf_mean, lcf_std = meta["lcf"], meta["lcf_std"]
        if os.environ.get("COPO_CHECK_ASSIGN_LCF", "0") == "1":     # the reference's sanity check (algo_copo.py:446-471), two device reads
            self.workers.foreach_worker_with_id(lambda w_id, w: w.foreach_policy(lambda pi, pi_id: pi.assign_lcf(lcf_parameters, lcf_mean, lcf_std)))
        self.workers.foreach_worker_with_id(lambda w_id, w: w.foreach_env(lambda e: e.set_lcf_dist(mean=lcf_mean, std=lcf_std)))
        raw = getattr(pol, "_raw_host", None) or (float(pol._raw_lcf_adv_mean.item()), float(pol._raw_lcf_adv_std.item()))
        fetches = dict(raw_lcf_adv_mean_value=float(raw[0]), raw_lcf_adv_std_value=float(raw[1]))
        fetches.update(meta)
        train_results["default"]["custom_metrics"]["meta_update"] = fetches
        for policy_id, info in train_results.items():
            self.get_policy(policy_id).update_kl(info[LEARNER_STATS_KEY].get("kl"))
        self._last_batch = batch
        return train_results


# ========== Test scripts ==========
def _test(stop=2000, local_dir=None):
    """The reference's `_test()` (algo_copo.py:664-712): a tiny configuration through `train()`, "does it run" end to end."""
    from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_lcf_env, get_rllib_compatible_env
    from copo_amd.torch_copo.utils.train import train
    from copo_amd.torch_copo.utils.utils import get_train_parser
    args, _ = get_train_parser().parse_known_args()
    config = dict(env=get_rllib_compatible_env(get_lcf_env(MultiAgentIntersectionEnv)), env_config=dict(num_agents=8), num_envs=4, train_batch_size=100,
                  rollout_fragment_length=20, sgd_minibatch_size=30)
    return train(CoPOTrainer, config=config, checkpoint_freq=0, keep_checkpoints_num=0, stop={"timesteps_total": stop},
                 num_gpus=args.num_gpus, num_seeds=1, max_failures=0, exp_name=args.exp_name or "test_copo",
                 custom_callback=MultiAgentDrivingCallbacks, test_mode=True, local_mode=True, local_dir=local_dir)


if __name__ == "__main__":
    _test()

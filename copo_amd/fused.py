"""Host side of the fused minibatch learner (`copo_ppo_fused_step_f32`, include/copo_hip.h).

`FusedLearner` re-homes every fp32 parameter of a policy's model into ONE flat device buffer (the
nn.Parameters become views, so inference / checkpoints / state_dict keep working), owns the flat Adam
moments and step counter, and issues the 7-kernel SGD step.  Used by `PPOPolicyBase.run_sgd` and by
`CoPOPolicy.run_meta` (head modes META_NEW / META_OLD give the two policy gradients of the LCF meta update).
"""
import ctypes as C

import torch

from . import _capi

_META_DOT_SPLIT = 8      # COPO_META_DOT_SPLIT of include/copo_hip.h (DOT_SPLIT in csrc/learn_meta.inc): partial sums per minibatch


def _mlp_layers(seq_or_list):
    """[SlimFC, ...] -> list of nn.Linear."""
    return [m._model[0] for m in seq_or_list]


class FlatParams:
    """All fp32 parameters of a module in one flat buffer (parameters become views into it)."""

    def __init__(self, module, device):
        self.params = [p for p in module.parameters() if p.dtype == torch.float32]
        # every tensor starts on a 16-byte boundary so that the GEMM kernels can use float4 loads
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        n = off
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.offset = {}
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                self.offset[id(p)] = off
        self.numel = n

    def layout(self, linears, in_dim, out_dim):
        l1, l2, l3 = linears
        o = self.offset
        return _capi.NetLayout(o[id(l1.weight)], o[id(l1.bias)], o[id(l2.weight)], o[id(l2.bias)], o[id(l3.weight)],
                               o[id(l3.bias)], int(in_dim), int(out_dim))


def model_layouts(model, flat):
    """(policy layout, [value layouts]) for FullyConnectedModel / CCModel / CoPOModel."""
    hid = _mlp_layers(model._hidden_layers)
    assert len(hid) == 2 and hid[0].out_features == hid[1].out_features, "fused learner: two equal hidden layers"
    pol = flat.layout(hid + [model._logits._model[0]], hid[0].in_features, model._logits._model[0].out_features)
    vh = _mlp_layers(model._value_branch_separate)
    vals = [flat.layout(vh + [model._value_branch._model[0]], vh[0].in_features, 1)]
    for name in ("nei_value_network", "global_value_network"):
        net = getattr(model, name, None)
        if net is not None:
            ls = _mlp_layers(net)
            vals.append(flat.layout(ls, ls[0].in_features, 1))
    return pol, vals, hid[0].out_features


class FusedLearner:
    def __init__(self, policy, columns, mb, adv_key, meta_adv_key=None):
        """columns: [(name, width)] of the row pack; adv_key: pack column used as the PPO advantage."""
        self.policy = policy
        dev = policy.device
        cfg = policy.config
        model = policy.model
        self.flat = FlatParams(model, dev)
        pol, vals, H = model_layouts(model, self.flat)
        self.n_policy = sum(p.numel() for p in model.policy_parameters())
        assert max(pol.w1, pol.b1, pol.w2, pol.b2, pol.w3, pol.b3) < self.n_policy, \
            "the policy net must occupy the first block of the flat parameter buffer"
        col, off = {}, 0
        for name, w in columns:
            col[name] = off
            off += w
        c = _capi.PpoCfg()
        c.mb, c.hidden, c.act_dim, c.n_value_heads, c.pack_width = int(mb), int(H), 2, len(vals), off
        c.col_actions, c.col_logp, c.col_dist = col["actions"], col["action_logp"], col["action_dist_inputs"]
        c.col_adv = col[adv_key]
        c.col_meta_adv = col[meta_adv_key] if meta_adv_key else col[adv_key]
        vp = ["vf_preds", "nei_values", "global_values"]
        vt = ["value_targets", "nei_target", "global_target"]
        for g in range(len(vals)):
            c.col_vpred[g], c.col_vtarget[g] = col[vp[g]], col[vt[g]]
            c.val[g] = vals[g]
        c.pol = pol
        c.use_kl = 1 if cfg["kl_coeff"] > 0.0 else 0
        c.old_value_loss = 1 if cfg["old_value_loss"] else 0
        c.operand_dtype = _capi.OPERAND_BF16 if getattr(policy, "autocast_dtype", None) is torch.bfloat16 else _capi.OPERAND_F32
        c.clip_param, c.vf_clip_param = float(cfg["clip_param"]), float(cfg["vf_clip_param"])
        c.vf_loss_coeff, c.entropy_coeff = float(cfg["vf_loss_coeff"]), float(policy.entropy_coeff)
        c.lr, c.beta1, c.beta2, c.eps = float(cfg["lr"]), 0.9, 0.999, 1e-8
        c.n_params = self.flat.numel
        self.cfg = c
        self.columns = col
        n = self.flat.numel
        self.adam_m = torch.zeros(n, device=dev)
        self.adam_v = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = _capi.lib.copo_ppo_workspace_floats(C.byref(c))
        self.workspace = torch.zeros(int(ws), device=dev)
        self.stats = torch.zeros(_capi.PPO_STATS, device=dev)
        self._batch_ws = None
        self._rows_ws = None
        self.target_flat = None
        # mirror of the parameters with W1 / W2 stored [in][out]: coalesced operand reads for the forward passes.  The
        # kernels keep it current; torch-side writes to the parameters are noticed through the tensor version counter.
        self.flat_t = torch.zeros(n, device=dev)
        self._mirror_version = -1

    def attach_target(self, target_model):
        """Flat view of the target network (same layout as the model) for the META_OLD pass."""
        self.target_flat = FlatParams(target_model, self.policy.device)
        assert self.target_flat.numel == self.flat.numel

    def invalidate_mirror(self):
        """Torch code wrote the parameters (load_state_dict, set_weights, a population loaded for evaluation ...): the
        transposed mirror is stale until the next sync_mirror.  Writes through the nn.Parameter views do not bump the
        flat buffer's version counter, so every such writer calls this explicitly."""
        self._mirror_version = -1

    def sync_mirror(self):
        """Re-derive the transposed mirror if it was invalidated (or never built) since the kernels last wrote it.
        Call outside captured graphs, before replaying them."""
        v = self.flat.flat._version
        if self._mirror_version < 0 or v != self._mirror_version:
            _capi.check(_capi.lib.copo_transpose_weights_f32(C.byref(self.cfg), self.flat.flat.data_ptr(),
                                                             self.flat_t.data_ptr(), _capi.current_stream()))
            self._mirror_version = max(v, 0)

    @property
    def can_forward(self):
        """The forward-only kernel covers this layout (hidden 64 / 128 / 256 / 512)."""
        return self.cfg.hidden in (64, 128, 256, 512)

    def act(self, obs, eps, action, logp, dist_inputs, clipped=None):
        """Policy forward + sampling for dense obs [R, O] into caller-owned tensors (rollouts)."""
        R = obs.shape[0]
        _capi.check(_capi.lib.copo_mlp_forward_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.flat_t.data_ptr(), obs.data_ptr(), None, R, 0, 1, None,
            dist_inputs.data_ptr(), eps.data_ptr(), action.data_ptr(), logp.data_ptr(),
            None if clipped is None else clipped.data_ptr(), _capi.current_stream()))

    def values(self, obs, cc_obs, out=None, rows=None):
        """[n_value_heads, R] critic values for dense rows (postprocess); with `rows` (int64 indices) only those rows are
        computed, the others stay 0."""
        R, nv = obs.shape[0], int(self.cfg.n_value_heads)
        if rows is not None:
            out = torch.zeros(nv, R, dtype=torch.float32, device=obs.device) if out is None else out
            if rows.numel() > 0:
                _capi.check(_capi.lib.copo_mlp_forward_rows_f32(
                    C.byref(self.cfg), self.flat.flat.data_ptr(), self.flat_t.data_ptr(), obs.data_ptr(),
                    None if cc_obs is None else cc_obs.data_ptr(), rows.data_ptr(), int(rows.numel()), R, 1, nv, out.data_ptr(),
                    _capi.current_stream()))
            return out
        if out is None:
            out = torch.empty(nv, R, dtype=torch.float32, device=obs.device)
        _capi.check(_capi.lib.copo_mlp_forward_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.flat_t.data_ptr(), obs.data_ptr(),
            None if cc_obs is None else cc_obs.data_ptr(), R, 1, nv, out.data_ptr(), None, None, None, None, None,
            _capi.current_stream()))
        return out

    def gather_epoch_ok(self, rs):
        """May the planned rows be copied into minibatch order (gather_epoch)?  The copy holds rows_all.shape[0] x mb rows of the
        observation, pack and critic-observation sources (the whole plan, i.e. num_sgd_iter x the batch): it must fit the device
        memory that is free now (with a quarter to spare), and the sources must be contiguous fp32.  Otherwise the step kernels
        keep reading through the row tables (`rows_all`), which need no copy."""
        names = ["obs", "pack"] + (["cc_obs"] if rs.get("cc_obs") is not None else [])
        if not all(rs[k].is_contiguous() and rs[k].dtype == torch.float32 for k in names):
            return False
        d = rs.get("_dense")
        cap = int(rs["rows_all"].shape[0]) * int(self.cfg.mb)
        if d is not None and d["obs"].shape[0] == cap and d["obs"].shape[1] == rs["obs"].shape[1]:
            return True                      # (already allocated)
        need = 4 * cap * sum(int(rs[k].shape[1]) for k in names)
        dev = self.flat.flat.device
        if dev.type != "cuda":
            return True
        free, _total = torch.cuda.mem_get_info(dev)
        return need * 1.25 <= free

    def gather_epoch(self, rs, n_mb):
        """The rows of this epoch's plan (`rs["rows_all"][:n_mb]`) copied once into minibatch order -- observation, critic
        observation and pack rows -- so that the step kernels read row kb * mb + m directly instead of chasing a row index in
        front of every row load (`rows` = NULL in the C ABI).  One gather per epoch (~75 k rows) against ~150 steps that each
        saved a dependent memory round trip.  Returns the dict to pass as `rs` to step / step_dp."""
        mb, dev = int(self.cfg.mb), self.flat.flat.device
        rows = rs["rows_all"][:n_mb].reshape(-1)
        cap = int(rs["rows_all"].shape[0]) * mb
        d = rs.get("_dense")
        if d is None or d["obs"].shape[0] != cap or d["obs"].shape[1] != rs["obs"].shape[1]:
            d = dict(obs=torch.zeros(cap, rs["obs"].shape[1], device=dev), pack=torch.zeros(cap, rs["pack"].shape[1], device=dev),
                     cc_obs=None if rs["cc_obs"] is None else torch.zeros(cap, rs["cc_obs"].shape[1], device=dev))
            rs["_dense"] = d
        n = rows.numel()
        names = ["obs", "pack"] + (["cc_obs"] if d["cc_obs"] is not None else [])
        srcs = (C.c_void_p * len(names))(*[rs[k].data_ptr() for k in names])
        dsts = (C.c_void_p * len(names))(*[d[k].data_ptr() for k in names])
        widths = (C.c_int32 * len(names))(*[int(rs[k].shape[1]) for k in names])
        assert all(rs[k].is_contiguous() and rs[k].dtype == torch.float32 for k in names)
        _capi.check(_capi.lib.copo_gather_rows_f32(srcs, dsts, widths, len(names), rows.data_ptr(), n, _capi.current_stream()))
        out = dict(rs)
        out.update(obs=d["obs"], pack=d["pack"], cc_obs=d["cc_obs"], rows_all=None)
        return out

    def step(self, rs, head_mode=_capi.HEAD_PPO, apply_adam=True, theta=None, grad=None, stats=None, bump_index=True):
        """One fused minibatch pass over the sources bound in `rs` (PPOPolicyBase._row_sources layout; `rows_all` None: the
        sources are in minibatch order, see gather_epoch)."""
        cc = rs["cc_obs"]
        kl = self.policy.kl_coeff
        if theta is None and not torch.cuda.is_current_stream_capturing():
            self.sync_mirror()
        _capi.check(_capi.lib.copo_ppo_fused_step_f32(
            C.byref(self.cfg), (self.flat.flat if theta is None else theta).data_ptr(), self.adam_m.data_ptr(),
            self.adam_v.data_ptr(), (self.grad if grad is None else grad).data_ptr(), rs["obs"].data_ptr(),
            None if cc is None else cc.data_ptr(), rs["pack"].data_ptr(), None if rs["rows_all"] is None else rs["rows_all"].data_ptr(),
            rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(), kl.data_ptr(), self.step_count.data_ptr(),
            self.workspace.data_ptr(), None if stats is None else stats.data_ptr(), 1 if apply_adam else 0,
            int(head_mode), None if rs["k"] is None else rs["k"].data_ptr(), 1 if bump_index else 0,
            self.flat_t.data_ptr() if theta is None else None, _capi.current_stream()))

    def step_dp(self, rs, exchange, stats=None, bump_index=True):
        """One data-parallel PPO minibatch step: the local step's two launches, the gradient tiles summed over the ranks
        inside the weight-gradient kernel (`exchange`: peer.TileExchange; None = a world of one, i.e. the local step)."""
        cc = rs["cc_obs"]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_mirror()
        _capi.check(_capi.lib.copo_ppo_fused_step_dp_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), rs["obs"].data_ptr(),
            None if cc is None else cc.data_ptr(), rs["pack"].data_ptr(), None if rs["rows_all"] is None else rs["rows_all"].data_ptr(),
            rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(), self.policy.kl_coeff.data_ptr(), self.step_count.data_ptr(), self.workspace.data_ptr(),
            None if stats is None else stats.data_ptr(), rs["k"].data_ptr(), 1 if bump_index else 0, self.flat_t.data_ptr(),
            None if exchange is None else exchange.ptrs, 0 if exchange is None else exchange.rank,
            1 if exchange is None else exchange.world, _capi.current_stream()))

    def adam(self, rs, grad=None):
        """Adam on the flat buffers after a gradient all-reduce; advances the minibatch index."""
        _capi.check(_capi.lib.copo_adam_step_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
            (self.grad if grad is None else grad).data_ptr(), self.flat.numel, self.step_count.data_ptr(),
            rs["k"].data_ptr(), self.flat_t.data_ptr(), self.workspace.data_ptr(), _capi.current_stream()))

    # ---- LCF meta update (CoPO) -------------------------------------------------------------------------------
    def meta_grads(self, rs, g_new, g_old, stats_new, stats_old, dot_partials):
        """Both policy gradients of `meta_update` in one grouped pass (current policy / target policy), plus the
        per-workgroup partials of their dot product."""
        _capi.check(_capi.lib.copo_meta_grads_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.target_flat.flat.data_ptr(), g_new.data_ptr(),
            g_old.data_ptr(), rs["obs"].data_ptr(), rs["pack"].data_ptr(), rs["rows_all"].data_ptr(),
            rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(), self.workspace.data_ptr(), stats_new.data_ptr(),
            stats_old.data_ptr(), dot_partials.data_ptr(), rs["k"].data_ptr(), _capi.current_stream()))

    def meta_lcf(self, rs, eps_all, lcf_param, raw_mean_std, tail, col_adv, col_nei_adv):
        _capi.check(_capi.lib.copo_meta_lcf_f64(
            rs["pack"].data_ptr(), self.cfg.pack_width, int(col_adv), int(col_nei_adv), rs["rows_all"].data_ptr(),
            rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(), eps_all.data_ptr(), self.cfg.mb, rs["k"].data_ptr(),
            lcf_param.data_ptr(), raw_mean_std.data_ptr(), tail.data_ptr(), _capi.current_stream()))

    def meta_finish(self, rs, g_new, g_old, dot_partials, tail, lcf_param, adam_state, lr, stats_new, stats_old, stats,
                    bump_index=True):
        """dot_partials=None: recompute <g_new, g_old> from the (all-reduced) gradients."""
        _capi.check(_capi.lib.copo_meta_finish_f64(
            g_new.data_ptr(), g_old.data_ptr(), self.n_policy, None if dot_partials is None else dot_partials.data_ptr(),
            tail.data_ptr(), lcf_param.data_ptr(),
            adam_state.data_ptr(), float(lr), stats_new.data_ptr(), stats_old.data_ptr(), stats.data_ptr(),
            rs["k"].data_ptr(), 1 if bump_index else 0, _capi.current_stream()))

    def meta_step(self, rs, g_new, g_old, stats_new, stats_old, dot_partials, eps_all, lcf_param, raw_mean_std, tail,
                  col_adv, col_nei_adv, adam_state, lr, stats, bump_index=True):
        """meta_grads + meta_lcf + meta_finish as one call (single process: nothing to all-reduce in between)."""
        _capi.check(_capi.lib.copo_meta_step_f64(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.target_flat.flat.data_ptr(), g_new.data_ptr(),
            g_old.data_ptr(), rs["obs"].data_ptr(), rs["pack"].data_ptr(), rs["rows_all"].data_ptr(),
            rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(), self.workspace.data_ptr(), stats_new.data_ptr(),
            stats_old.data_ptr(), dot_partials.data_ptr(), int(col_adv), int(col_nei_adv), eps_all.data_ptr(),
            lcf_param.data_ptr(), raw_mean_std.data_ptr(), tail.data_ptr(), adam_state.data_ptr(), float(lr),
            stats.data_ptr(), rs["k"].data_ptr(), 1 if bump_index else 0, _capi.current_stream()))

    # ---- batched meta pass: many minibatches per launch chain, then the sequential LCF steps in one kernel -------
    def meta_fold_len(self):
        return int(_capi.lib.copo_meta_fold_len(C.byref(self.cfg)))

    def meta_batch_grads(self, rs, first, nb, gv, stats_k, g_out=None):
        """Phase A for minibatches [first, first + nb) of the row tables: gv[first:first+nb], stats_k[first:...]."""
        if self._batch_ws is None or self._batch_ws[0] < nb:
            n = int(_capi.lib.copo_meta_batch_workspace_floats(C.byref(self.cfg), int(nb)))
            self._batch_ws = (nb, torch.zeros(n, dtype=torch.float32, device=self.flat.flat.device))
        _capi.check(_capi.lib.copo_meta_batch_grads_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.target_flat.flat.data_ptr(), rs["obs"].data_ptr(),
            rs["pack"].data_ptr(), rs["rows_all"].data_ptr(), rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(),
            self._batch_ws[1].data_ptr(), int(self._batch_ws[0]), int(first), int(nb), None if g_out is None else g_out.data_ptr(),
            gv[first:].data_ptr(), stats_k[first:].data_ptr(), _capi.current_stream()))

    def _ensure_batch_ws(self, nb):
        if self._batch_ws is None or self._batch_ws[0] < nb:
            n = int(_capi.lib.copo_meta_batch_workspace_floats(C.byref(self.cfg), int(nb)))
            self._batch_ws = (nb, torch.zeros(n, dtype=torch.float32, device=self.flat.flat.device))

    def meta_rows(self, rs):
        """Row store of one training iteration: the row-local part of phase A for every row of the dense sources, once."""
        n_rows = int(rs["max_rows"])
        if self._rows_ws is None or self._rows_ws[0] != n_rows:
            n = int(_capi.lib.copo_meta_rows_workspace_floats(C.byref(self.cfg), n_rows))
            blocks = -(-n_rows // self.cfg.mb)
            dev = self.flat.flat.device
            self._rows_ws = (n_rows, torch.zeros(n, dtype=torch.float32, device=dev),
                             torch.zeros(2 * blocks * self.cfg.mb, 2, dtype=torch.float32, device=dev))
        _capi.check(_capi.lib.copo_meta_rows_f32(
            C.byref(self.cfg), self.flat.flat.data_ptr(), self.target_flat.flat.data_ptr(), rs["obs"].data_ptr(),
            rs["pack"].data_ptr(), n_rows, self._rows_ws[1].data_ptr(), self._rows_ws[2].data_ptr(), _capi.current_stream()))

    def meta_rowstat(self, rs, first, nb, stats_k):
        """Loss statistics of minibatches [first, first + nb) regrouped from the row store: one launch for a whole pass, after which
        `meta_batch_wgrads(..., stats_k=None)` leaves them alone."""
        _capi.check(_capi.lib.copo_meta_rowstat_f32(
            C.byref(self.cfg), rs["rows_all"].data_ptr(), rs["w_all"].data_ptr(), rs["denom_all"].data_ptr(),
            self._rows_ws[2].data_ptr(), int(first), int(nb), stats_k[first:].data_ptr(), _capi.current_stream()))

    def meta_batch_wgrads(self, rs, first, nb, gv, stats_k, g_out=None):
        """Phase A of minibatches [first, first + nb) on top of the row store (`meta_rows` must have run).  stats_k None: the
        statistics come from `meta_rowstat`."""
        self._ensure_batch_ws(nb)
        _capi.check(_capi.lib.copo_meta_batch_wgrads_f32(
            C.byref(self.cfg), rs["obs"].data_ptr(), rs["rows_all"].data_ptr(), rs["w_all"].data_ptr(),
            rs["denom_all"].data_ptr(), self._rows_ws[1].data_ptr(), int(self._rows_ws[0]), self._rows_ws[2].data_ptr(),
            self._batch_ws[1].data_ptr(), int(self._batch_ws[0]), int(first), int(nb),
            None if g_out is None else g_out.data_ptr(), gv[first:].data_ptr(), None if stats_k is None else stats_k[first:].data_ptr(),
            _capi.current_stream()))

    def meta_batch_dot(self, g, n, nb, gv, denom=None):
        """gv[:nb] = <g[b][0], g[b][1]>; denom [nb] float32: scaled by 1 / denom^2 (unit-weight gradients of the row store)."""
        # caller-owned scratch for the partial sums (COPO_META_DOT_SPLIT = 8 per minibatch, learn_meta.inc), ONE PER STREAM: two
        # calls on different streams never share it, and a buffer that grows is allocated under the stream that uses it (the old one
        # stays referenced by the allocator until that stream's queued kernels are past it)
        key = int(torch.cuda.current_stream(g.device).cuda_stream)
        parts = self.__dict__.setdefault("_dot_parts", {})
        part = parts.get(key)
        if part is None or part.numel() < _META_DOT_SPLIT * int(nb):
            part = parts[key] = torch.zeros(_META_DOT_SPLIT * max(int(nb), 256), dtype=torch.float64, device=g.device)
        _capi.check(_capi.lib.copo_meta_batch_dot_f64(g.data_ptr(), int(n), int(nb), gv.data_ptr(),
                                                      None if denom is None else denom.data_ptr(), part.data_ptr(), _capi.current_stream()))

    _seq_xchg = None

    def meta_batch_lcf(self, rs, n_mb, eps_all, gv, stats_k, lcf_param, raw_mean_std, adam_state, lr, stats, col_adv,
                       col_nei_adv, dense=None, n_wg=0, k_first=0, k_count=-1):
        """Phase B.  dense = (ego_nei [S][n_mb][mb][2], w [S][n_mb][mb], eps [S][n_mb][mb]) replaces the row gather.
        n_wg: workgroups that share every step's rows (0 = the library's choice: one up to eight ranks' rows).
        k_first / k_count: only the steps of minibatches [k_first, k_first + k_count) of the n_mb the arrays hold."""
        if self._seq_xchg is None:
            self._seq_xchg = torch.zeros(256, dtype=torch.float64, device=self.flat.flat.device)
        if dense is None:
            args = (rs["pack"].data_ptr(), self.cfg.pack_width, int(col_adv), int(col_nei_adv), rs["rows_all"].data_ptr(),
                    None, 1, rs["w_all"].data_ptr(), eps_all.data_ptr())
        else:
            en, w, eps = dense
            args = (None, 0, 0, 0, None, en.data_ptr(), int(en.shape[0]), w.data_ptr(), eps.data_ptr())
        _capi.check(_capi.lib.copo_meta_batch_lcf_f64(
            *args, rs["denom_all"].data_ptr(), self.cfg.mb, int(n_mb), gv.data_ptr(), stats_k.data_ptr(),
            lcf_param.data_ptr(), raw_mean_std.data_ptr(), adam_state.data_ptr(), float(lr), stats.data_ptr(),
            int(k_first), int(k_count), int(n_wg), self._seq_xchg.data_ptr(), _capi.current_stream()))

    def state(self):
        return dict(adam_m=self.adam_m.clone(), adam_v=self.adam_v.clone(), step=self.step_count.clone())

    def load_state(self, st):
        self.adam_m.copy_(st["adam_m"])
        self.adam_v.copy_(st["adam_v"])
        self.step_count.copy_(st["step"])

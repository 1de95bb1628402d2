"""Host-side building blocks shared by the three algorithms (what RLlib supplies to the reference).

The reference plugs into Ray/RLlib: `AlgorithmConfig`, `SampleBatch`, `TorchDiagGaussian`, `SlimFC`,
`normc_initializer`, `tune.grid_search` (call sites: algo_ippo.py:2-12, algo_ccppo.py:11-29,
algo_copo.py:16-38).  There is no Ray here: these are small from-scratch equivalents with the same
names and numeric semantics, written for dense `[T, E, N]` device tensors.
"""
import copy
import math
from typing import Any, Dict

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------------------------------
# batch containers / keys
# ----------------------------------------------------------------------------------------------------
class SampleBatch(dict):
    """Dict of equally long columns.  In this build a batch is usually dense: tensors of shape
    [T, E, N, ...] plus a `flags` column (ACTED/DONE bits) instead of per-agent python lists."""
    OBS = CUR_OBS = "obs"
    NEXT_OBS = "new_obs"
    ACTIONS = "actions"
    REWARDS = "rewards"
    DONES = "dones"
    INFOS = "infos"
    VF_PREDS = "vf_preds"
    ACTION_LOGP = "action_logp"
    ACTION_DIST_INPUTS = "action_dist_inputs"
    SEQ_LENS = "seq_lens"
    T = "t"
    FLAGS = "flags"
    VALID = "valid_mask"

    @property
    def count(self):
        for k, v in self.items():
            if hasattr(v, "shape") and len(v.shape) > 0:
                return int(v.shape[0])
        return 0

    def agent_steps(self):
        return self.count

    def to(self, device):
        return SampleBatch({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class Postprocessing:
    ADVANTAGES = "advantages"
    VALUE_TARGETS = "value_targets"


LEARNER_STATS_KEY = "learner_stats"
NUM_AGENT_STEPS_SAMPLED = "num_agent_steps_sampled"
NUM_ENV_STEPS_SAMPLED = "num_env_steps_sampled"


# ----------------------------------------------------------------------------------------------------
# action distribution
# ----------------------------------------------------------------------------------------------------
_LOG_2PI = math.log(2.0 * math.pi)


class TorchDiagGaussian:
    """Diagonal Gaussian over `inputs = [mean | log_std]` (RLlib TorchDiagGaussian semantics)."""

    def __init__(self, inputs, model=None):
        self.inputs = inputs
        self.mean, self.log_std = torch.chunk(inputs, 2, dim=-1)
        self.std = torch.exp(self.log_std)

    def logp(self, x):
        z = (x - self.mean) / self.std
        return (-0.5 * z * z - self.log_std - 0.5 * _LOG_2PI).sum(-1)

    def entropy(self):
        return (self.log_std + 0.5 + 0.5 * _LOG_2PI).sum(-1)

    def kl(self, other):
        """KL(self || other)."""
        return (other.log_std - self.log_std +
                (self.std * self.std + (self.mean - other.mean) ** 2) / (2.0 * other.std * other.std) - 0.5).sum(-1)

    def sample(self, eps=None):
        if eps is None:
            eps = torch.randn_like(self.mean)
        return self.mean + self.std * eps

    def deterministic_sample(self):
        return self.mean


# ----------------------------------------------------------------------------------------------------
# network pieces (state-dict names match the reference's checkpoints: `<net>.N._model.0.{weight,bias}`)
# ----------------------------------------------------------------------------------------------------
def normc_initializer(std=1.0):
    """N(0,1) weights, every output row rescaled to L2 norm `std` (best_checkpoints/*.npz row norms)."""
    def initializer(tensor):
        with torch.no_grad():
            tensor.normal_(0, 1)
            tensor.mul_(std / torch.sqrt(tensor.pow(2).sum(1, keepdim=True)))
    return initializer


class SlimFC(nn.Module):
    def __init__(self, in_size, out_size, initializer=None, activation_fn=None, use_bias=True):
        super().__init__()
        lin = nn.Linear(in_size, out_size, bias=use_bias)
        (initializer or nn.init.xavier_uniform_)(lin.weight)
        if use_bias:
            nn.init.zeros_(lin.bias)
        layers = [lin]
        if activation_fn in ("tanh", nn.Tanh):
            layers.append(nn.Tanh())
        elif activation_fn in ("relu", nn.ReLU):
            layers.append(nn.ReLU())
        elif activation_fn not in (None, "linear"):
            raise ValueError("unsupported activation %r" % (activation_fn,))
        self._model = nn.Sequential(*layers)

    def forward(self, x):
        return self._model(x)


def build_mlp(in_size, hiddens, activation, out_size=None, out_std=0.01):
    """Hidden stack (normc 1.0) [+ linear head (normc `out_std`)] as separate modules."""
    layers, prev = [], in_size
    for h in hiddens:
        layers.append(SlimFC(prev, h, initializer=normc_initializer(1.0), activation_fn=activation))
        prev = h
    head = None
    if out_size is not None:
        head = SlimFC(prev, out_size, initializer=normc_initializer(out_std), activation_fn=None)
    return nn.Sequential(*layers), head, prev


# ----------------------------------------------------------------------------------------------------
# spaces (gym is not a dependency of the build)
# ----------------------------------------------------------------------------------------------------
class Box:
    def __init__(self, low, high, shape, dtype="float32"):
        import numpy as np
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
        import numpy as np
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def sample(self):
        import numpy as np
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.flat[0], self.high.flat[0], self.shape)


class DictSpace:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def keys(self):
        return self.spaces.keys()

    def __getitem__(self, k):
        return self.spaces[k]

    def __contains__(self, k):
        return k in self.spaces

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}


# ----------------------------------------------------------------------------------------------------
# config
# ----------------------------------------------------------------------------------------------------
class _GridSearch(dict):
    pass


def grid_search(values):
    """`tune.grid_search([...])` stand-in: expanded by `expand_grid` (train_copo.py:22)."""
    return _GridSearch(grid_search=list(values))


def expand_grid(config: Dict[str, Any]):
    """All combinations of the `grid_search` entries found anywhere in a nested config dict."""
    def find(d, path=()):
        for k, v in d.items():
            if isinstance(v, dict) and set(v.keys()) == {"grid_search"}:
                yield path + (k,), v["grid_search"]
            elif isinstance(v, dict):
                yield from find(v, path + (k,))
    axes = list(find(config))
    if not axes:
        return [copy.deepcopy(config)]
    out = [copy.deepcopy(config)]
    for path, values in axes:
        nxt = []
        for c in out:
            for val in values:
                c2 = copy.deepcopy(c)
                d = c2
                for p in path[:-1]:
                    d = d[p]
                d[path[-1]] = val
                nxt.append(c2)
        out = nxt
    return out


class AlgorithmConfig:
    """Attribute bag with dict-style access and `update_from_dict` (the slice of RLlib's AlgorithmConfig the
    reference uses: algo_ippo.py:17-75, algo_ccppo.py:37-52, algo_copo.py:63-92)."""
    _ALIASES = {"lambda": "lambda_", "num_workers": "num_rollout_workers", "framework": "framework_str"}

    def __init__(self, algo_class=None):
        self.algo_class = algo_class
        # RLlib PPO defaults that the reference inherits
        self.gamma = 0.99
        self.lambda_ = 1.0
        self.kl_coeff = 0.2
        self.kl_target = 0.01
        self.vf_loss_coeff = 1.0
        self.entropy_coeff = 0.0
        self.clip_param = 0.3
        self.vf_clip_param = 10.0
        self.grad_clip = None
        self.use_gae = True
        self.use_critic = True
        self.lr = 5e-5
        self.train_batch_size = 4000
        self.sgd_minibatch_size = 128
        self.num_sgd_iter = 30
        self.rollout_fragment_length = 200
        self.batch_mode = "truncate_episodes"
        self.count_steps_by = "env_steps"
        self.num_rollout_workers = 2
        self.num_cpus_per_worker = 1
        self.num_cpus_for_local_worker = 1
        self.num_gpus = 0
        self.framework_str = "torch"
        self.simple_optimizer = True
        self.seed = None
        self.env = None
        self.env_config = {}
        self.callbacks = None
        self.log_level = "INFO"
        self.model = dict(fcnet_hiddens=[256, 256], fcnet_activation="tanh", post_fcnet_hiddens=[],
                          no_final_linear=False, vf_share_layers=False, free_log_std=False,
                          custom_model=None, custom_model_config={})
        self.multiagent = {}
        # build-specific knobs (no counterpart in the reference)
        self.num_envs = 1                # parallel scenes per GPU (the reference: one env per rollout worker)
        self.use_hip_graphs = True       # capture rollout / SGD / meta steps in hipGraphs when on a GPU
        self.policy_dtype = "float32"    # "bfloat16": MLPs under autocast, losses/advantages stay fp32
        self.device = None               # default: cuda:LOCAL_RANK if available else cpu

    def _key(self, k):
        return self._ALIASES.get(k, k)

    def __getitem__(self, k):
        k = self._key(k)
        if not hasattr(self, k):
            raise KeyError(k)
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, self._key(k), v)

    def __contains__(self, k):
        return hasattr(self, self._key(k))

    def get(self, k, default=None):
        return getattr(self, self._key(k), default)

    def update_from_dict(self, d):
        for k, v in d.items():
            k = self._key(k)
            cur = getattr(self, k, None)
            if isinstance(cur, dict) and isinstance(v, dict) and k in ("model", "env_config", "multiagent"):
                merged = dict(cur)
                for kk, vv in v.items():
                    if isinstance(merged.get(kk), dict) and isinstance(vv, dict):
                        merged[kk] = {**merged[kk], **vv}
                    else:
                        merged[kk] = vv
                setattr(self, k, merged)
            else:
                setattr(self, k, v)
        return self

    def to_dict(self):
        return {k: copy.deepcopy(v) for k, v in self.__dict__.items() if k != "algo_class"}

    def copy(self):
        c = copy.copy(self)
        c.__dict__ = copy.deepcopy({k: v for k, v in self.__dict__.items() if k != "algo_class"})
        c.algo_class = self.algo_class
        return c

    def validate(self):
        assert self.framework_str == "torch", "only the torch path exists in this build"
        assert self.sgd_minibatch_size > 0 and self.num_sgd_iter > 0
        return self


def standardized(a):
    """(a - mean) / max(1e-4, std) with population std (RLlib `standardized`, algo_copo.py:550-551)."""
    return (a - a.mean()) / max(1e-4, float(a.std(unbiased=False) if torch.is_tensor(a) else a.std()))


def reduce_mean_valid_fn(train_batch):
    """torch.mean, or a weighted mean when the minibatch carries a validity mask (static-shape minibatches
    padded with zero-weight rows; data-parallel runs pass the GLOBAL denominator / world_size)."""
    w = train_batch.get(SampleBatch.VALID)
    if w is None:
        return torch.mean
    denom = train_batch.get("valid_denominator")
    if denom is None:
        denom = w.sum().clamp_min(1.0)

    def _mean(t):
        return (t * w.to(t.dtype)).sum() / denom.to(t.dtype)
    return _mean

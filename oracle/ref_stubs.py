"""Stub-import harness for the reference (TEST INFRASTRUCTURE, dev container only).

The reference (`/root/reference/copo_code/copo/torch_copo/*.py`) imports ray / gym /
metadrive at module top level; none of them is installed here.  This module installs a
meta-path finder that fabricates those packages, then injects *real behaviour* for the
dozen RLlib helpers the reference's arithmetic actually calls.
With it the reference's own functions (`compute_nei_advantage`, `CCEnv._find_in_range`,
`CoPOPolicy.loss`, `CoPOPolicy.meta_update`, `CoPOTrainer.training_step`, ...) execute
unmodified, and `oracle/gen_golden.py` records their inputs/outputs as fixtures.

Nothing here ships to the GPU box as a dependency of product code: only
`oracle/gen_golden.py` imports it, and only when `/root/reference` exists.

Third-party helpers restated here (3P restatements, flagged in the fixtures' metadata):
  TorchDiagGaussian (logp / entropy / kl / sample), SlimFC, normc_initializer,
  discount_cumsum, compute_advantages, standardized, minibatches (unshuffled), SampleBatch.
"""
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np
import scipy.signal
import torch
import torch.nn as nn

STUB_ROOTS = ("ray", "gym", "metadrive", "tqdm")
REFERENCE_ROOT = "/root/reference/copo_code"


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (object,), {"__init__": lambda self, *a, **k: None, "__module__": self.__name__})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


class SampleBatch(dict):
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

    @property
    def count(self):
        for k, v in self.items():
            if k != "infos" and hasattr(v, "__len__"):
                return len(v)
        return len(self["infos"])

    def set_get_interceptor(self, fn):
        pass

    def agent_steps(self):
        return self.count

    def env_steps(self):
        return self.count


class Postprocessing:
    ADVANTAGES = "advantages"
    VALUE_TARGETS = "value_targets"


def discount_cumsum(x, gamma):
    return scipy.signal.lfilter([1], [1, float(-gamma)], x[::-1], axis=0)[::-1]


def compute_advantages(rollout, last_r, gamma=0.9, lambda_=1.0, use_gae=True, use_critic=True):
    assert use_gae and use_critic
    vpred_t = np.concatenate([rollout[SampleBatch.VF_PREDS], np.array([last_r])])
    delta_t = rollout[SampleBatch.REWARDS] + gamma * vpred_t[1:] - vpred_t[:-1]
    rollout[Postprocessing.ADVANTAGES] = discount_cumsum(delta_t, gamma * lambda_)
    rollout[Postprocessing.VALUE_TARGETS] = (
        rollout[Postprocessing.ADVANTAGES] + rollout[SampleBatch.VF_PREDS]).astype(np.float32)
    rollout[Postprocessing.ADVANTAGES] = rollout[Postprocessing.ADVANTAGES].astype(np.float32)
    return rollout


def standardized(a):
    return (a - a.mean()) / max(1e-4, a.std())


def minibatches_unshuffled(batch, size, shuffle=False):
    n = batch.count
    i = 0
    while i < n:
        sl = slice(i, i + size)
        yield SampleBatch({k: (v[sl] if not isinstance(v, list) else v[sl]) for k, v in batch.items()})
        i += size


def normc_initializer(std=1.0):
    def initializer(tensor):
        tensor.data.normal_(0, 1)
        tensor.data *= std / torch.sqrt(tensor.data.pow(2).sum(1, keepdim=True))

    return initializer


class SlimFC(nn.Module):
    def __init__(self, in_size, out_size, initializer=None, activation_fn=None, use_bias=True, bias_init=0.0):
        super().__init__()
        layers = []
        linear = nn.Linear(in_size, out_size, bias=use_bias)
        if initializer is None:
            initializer = nn.init.xavier_uniform_
        initializer(linear.weight)
        if use_bias:
            nn.init.constant_(linear.bias, bias_init)
        layers.append(linear)
        if activation_fn == "tanh":
            layers.append(nn.Tanh())
        elif activation_fn == "relu":
            layers.append(nn.ReLU())
        elif activation_fn is not None:
            raise ValueError(activation_fn)
        self._model = nn.Sequential(*layers)

    def forward(self, x):
        return self._model(x)


class TorchModelV2:
    def __init__(self, obs_space, action_space, num_outputs, model_config, name):
        self.obs_space = obs_space
        self.action_space = action_space
        self.num_outputs = num_outputs
        self.model_config = model_config
        self.name = name
        self.view_requirements = {}
        self.tower_stats = {}

    def __call__(self, input_dict, state=None, seq_lens=None):
        d = dict(input_dict)
        d["obs_flat"] = torch.as_tensor(d["obs"])
        return self.forward(d, state or [], seq_lens)

    def is_time_major(self):
        return False


class TorchDiagGaussian:
    """3P restatement of ray.rllib.models.torch.torch_action_dist.TorchDiagGaussian (2.2.0)."""

    def __init__(self, inputs, model=None):
        inputs = torch.as_tensor(inputs)
        mean, log_std = torch.chunk(inputs, 2, dim=1)
        self.mean, self.log_std = mean, log_std
        self.dist = torch.distributions.normal.Normal(mean, torch.exp(log_std))

    def logp(self, x):
        return self.dist.log_prob(x).sum(-1)

    def entropy(self):
        return self.dist.entropy().sum(-1)

    def kl(self, other):
        return torch.distributions.kl.kl_divergence(self.dist, other.dist).sum(-1)

    def sample(self):
        return self.dist.sample()


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(dtype)
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class DictSpace:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]


_installed = False


def install():
    """Install the stubs + injections; idempotent."""
    global _installed
    if _installed:
        return
    _installed = True
    if not hasattr(np, "product"):
        np.product = np.prod
    sys.meta_path.insert(0, _Finder())

    import ray.rllib.utils.framework as fw
    fw.try_import_torch = lambda *a, **k: (torch, nn)

    import ray.rllib.evaluation.postprocessing as pp
    pp.discount_cumsum = discount_cumsum
    pp.Postprocessing = Postprocessing
    pp.compute_advantages = compute_advantages

    import ray.rllib.policy.sample_batch as sb
    sb.SampleBatch = SampleBatch

    import ray.rllib.models as models
    import ray.rllib.models.catalog as catalog

    class ModelCatalog:
        @staticmethod
        def register_custom_model(name, cls):
            pass

    models.ModelCatalog = ModelCatalog
    catalog.ModelCatalog = ModelCatalog

    import ray.rllib.utils.annotations as ann
    ann.override = lambda cls: (lambda f: f)
    ann.ExperimentalAPI = lambda f: f

    import ray.tune.registry as reg
    reg.register_env = lambda *a, **k: None

    import metadrive.utils as mu
    mu.clip = lambda a, lo, hi: min(max(a, lo), hi)
    _rs = {"rs": np.random.RandomState(0)}
    mu.get_np_random = lambda seed=None: _rs["rs"]
    mu._copo_rs = _rs  # gen_golden reseeds through this handle

    import ray.rllib.models.torch.misc as misc
    misc.SlimFC = SlimFC
    misc.normc_initializer = normc_initializer
    misc.AppendBiasLayer = type("AppendBiasLayer", (nn.Module,), {})

    import ray.rllib.models.torch.torch_modelv2 as tmv2
    tmv2.TorchModelV2 = TorchModelV2

    import ray.rllib.utils.torch_utils as tu
    tu.convert_to_torch_tensor = lambda x, device=None: torch.as_tensor(x)
    tu.explained_variance = lambda y, pred: torch.zeros(())
    tu.sequence_mask = None
    tu.warn_if_infinite_kl_divergence = lambda policy, kl: None

    import ray.rllib.utils.sgd as sgd
    sgd.standardized = standardized
    sgd.minibatches = minibatches_unshuffled

    import ray.rllib.utils.numpy as rnp

    def convert_to_numpy(x):
        if isinstance(x, dict):
            return {k: convert_to_numpy(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [convert_to_numpy(v) for v in x]
        if torch.is_tensor(x):
            return x.detach().cpu().numpy()
        return x

    rnp.convert_to_numpy = convert_to_numpy

    import ray.rllib.utils.metrics as metrics
    metrics.NUM_AGENT_STEPS_SAMPLED = "num_agent_steps_sampled"
    metrics.NUM_ENV_STEPS_SAMPLED = "num_env_steps_sampled"
    metrics.SYNCH_WORKER_WEIGHTS_TIMER = "synch_weights"
    import ray.rllib.utils.metrics.learner_info as li
    li.LEARNER_STATS_KEY = "learner_stats"

    import ray.util.debug as dbg
    dbg.log_once = lambda s: False

    import gym
    import gym.spaces as spaces
    spaces.Box = Box
    spaces.Dict = DictSpace
    gym.spaces = spaces

    class Wrapper:       # gym.Wrapper restated (3P): holds `env`, forwards reset / step / close, exposes `unwrapped`
        def __init__(self, env):
            self.env = env

        def reset(self, *a, **k):
            return self.env.reset(*a, **k)

        def step(self, *a, **k):
            return self.env.step(*a, **k)

        def close(self):
            return self.env.close()

        @property
        def unwrapped(self):
            return getattr(self.env, "unwrapped", self.env)

    gym.Wrapper = Wrapper

    import ray.rllib.utils as rutils

    def deep_update(original, new_dict, new_keys_allowed=False, allow_new_subkey_list=None, override_all_if_type_changes=None):
        # ray.rllib.utils.deep_update restated (3P): recursive dict update
        for k, v in new_dict.items():
            if isinstance(original.get(k), dict) and isinstance(v, dict):
                deep_update(original[k], v, True)
            else:
                original[k] = v
        return original

    rutils.deep_update = deep_update

    import ray.rllib.env as renv
    renv.MultiAgentEnv = type("MultiAgentEnv", (object,), {"__init__": lambda self, *a, **k: None})

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reseed_env_rng(seed):
    import metadrive.utils as mu
    mu._copo_rs["rs"] = np.random.RandomState(seed)

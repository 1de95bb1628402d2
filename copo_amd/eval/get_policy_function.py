"""numpy policy functions over the reference's two checkpoint key layouts (copo/eval/get_policy_function.py:56-198).

Layouts of a 2x256 tanh policy with a 2*act_dim Gaussian head:
  "tf"     `{policy}/fc_1{sfx}/kernel` [in, out] + `/bias`, `fc_2`, `fc_out`  -- IPPO / CL / CoPO populations
           (CoPO checkpoints carry the suffix `_1`, get_policy_function.py:135)
  "torch"  `_hidden_layers.{0,1}._model.0.{weight,bias}` [out, in], `_logits._model.0.*` -- CCPPO populations

`PolicyFunction` keeps the reference's call shape: `{agent_id: obs} , {agent_id: last_done} -> {agent_id: action}`,
and for CoPO populations appends `(lcf + 1) / 2` with one LCF draw per agent id from the population's trained
distribution (`process_svo`, get_policy_function.py:174-192).
"""
import os.path as osp

import numpy as np

_LAYERS_TF = ("fc_1", "fc_2", "fc_out")
_LAYERS_TORCH = ("_hidden_layers.0._model.0", "_hidden_layers.1._model.0", "_logits._model.0")
_cache = {}

# trained LCF distributions (mean, std) of the reference's CoPO populations, keyed `{ALGO}_{ENV}[_{INDEX}]` like the
# population files; data of copo/eval/get_policy_function.py:10-53 (the un-indexed names are the entries flagged "Best")
meta_svo_lookup_table = {
    "copo_round_0": (0.3837417275236364, 0.10217650927472532),
    "copo_round_1": (0.31903679224482756, 0.0923634324418871),
    "copo_round_2": (0.4292658473315993, 0.09482618003037936),
    "copo_round_3": (0.4448881541427814, 0.10107655640234027),
    "copo_round_4": (0.40086105256749877, 0.09221974747222766),
    "copo_parking_0": (0.30009075371842636, 0.09819950084937246),
    "copo_parking_1": (0.21065708838011088, 0.09828158781716699),
    "copo_parking": (0.21065708838011088, 0.09828158781716699),
    "copo_parking_2": (0.19518211745379263, 0.099467324583154),
    "copo_parking_3": (0.10191127059883193, 0.0997653921183787),
    "copo_parking_4": (0.16749037122517296, 0.10430529321494854),
    "copo_bottle_0": (0.3347182310464089, 0.09320298072538878),
    "copo_bottle_1": (0.17889355489036493, 0.09873832422390318),
    "copo_bottle_2": (0.20677767223433444, 0.09703644548068967),
    "copo_bottle": (0.20677767223433444, 0.09703644548068967),
    "copo_bottle_3": (0.38850163995173936, 0.0996062973873657),
    "copo_bottle_4": (0.41495788567944586, 0.09026645110887394),
    "copo_inter_0": (0.36824979071031544, 0.08807231132921418),
    "copo_inter": (0.36824979071031544, 0.08807231132921418),
    "copo_inter_1": (0.3538261389261798, 0.0960544714410054),
    "copo_inter_2": (0.5021972039289642, 0.09395808752691537),
    "copo_inter_3": (0.32071430693592934, 0.09482878145941516),
    "copo_inter_4": (0.5012396887729041, 0.08545188030652832),
    "copo_round_rerun_0": (0.18783088442112683, 0.09685282814254507),
    "copo_round_rerun_1": (0.4449950145496117, 0.08596959420113016),
    "copo_round_rerun_2": (0.2914212175433245, 0.09590505765930911),
    "copo_round": (0.2914212175433245, 0.09590505765930911),
    "copo_round_rerun_3": (0.3506030522549751, 0.09272900488746863),
    "copo_bottle_rerun_0": (0.21729068847457367, 0.09800391086381884),
    "copo_bottle_rerun_1": (0.31267254543763706, 0.0914876350830348),
    "copo_bottle_rerun_2": (0.20579787078985448, 0.09402470028045275),
    "copo_tollgate_0": (0.46550068926742755, 0.08945204678064445),
    "copo_tollgate_1": (0.4772816712447233, 0.08097108654084174),
    "copo_tollgate_2": (0.4913835221499055, 0.08520848447553676),
    "copo_tollgate_3": (0.5575323092877565, 0.07595817525083297),
    "copo_tollgate": (0.5575323092877565, 0.07595817525083297),
    "copo_tollgate_4": (0.5247444219924696, 0.08257146898526042),
}


def detect_layout(weights):
    return "tf" if any(k.endswith("/kernel") for k in weights) else "torch"


def layer_arrays(weights, layout, policy_name="default", layer_name_suffix=""):
    """[(W [in, out], b)] of the three layers, whatever the key layout."""
    out = []
    for tf_name, torch_name in zip(_LAYERS_TF, _LAYERS_TORCH):
        if layout == "tf":
            stem = "%s/%s%s" % (policy_name, tf_name, layer_name_suffix)
            out.append((np.asarray(weights[stem + "/kernel"]), np.asarray(weights[stem + "/bias"])))
        else:
            out.append((np.asarray(weights[torch_name + ".weight"]).T, np.asarray(weights[torch_name + ".bias"])))
    return out


def _gaussian_head(layers, obs, deterministic):
    x = np.asarray(obs)
    assert x.ndim == 2 and x.shape[1] == layers[0][0].shape[0], (x.shape, layers[0][0].shape)
    for depth, (w, b) in enumerate(layers):
        x = np.matmul(x, w) + b
        if depth < len(layers) - 1:
            x = np.tanh(x)
    mean, log_std = np.split(x, 2, axis=1)
    return mean if deterministic else np.random.normal(mean, np.exp(log_std))


def _compute_actions_for_tf_policy(weights, obs, deterministic=False, policy_name="default_policy", layer_name_suffix=""):
    return _gaussian_head(layer_arrays(weights, "tf", policy_name, layer_name_suffix), obs, deterministic)


def _compute_actions_for_torch_policy(weights, obs, deterministic=False):
    return _gaussian_head(layer_arrays(weights, "torch"), obs, deterministic)


def _compute_actions_for_torch_policy2(weights, obs, policy_name=None, layer_name_suffix=None, deterministic=None):
    return _compute_actions_for_torch_policy(weights, obs, deterministic=bool(deterministic))


def population_layout(model_name):
    """(layout, suffix) by algorithm prefix, as get_policy_function.py:128-139."""
    if model_name.startswith("ccppo"):
        return "torch", ""
    if model_name.startswith(("ippo", "cl")):
        return "tf", ""
    if model_name.startswith("copo"):
        return "tf", "_1"
    raise ValueError("Unknown model: ", model_name)


def load_population(model_name, checkpoint_dir):
    key = (model_name, checkpoint_dir)
    if key not in _cache:
        with np.load(osp.join(checkpoint_dir, model_name + ".npz")) as f:
            _cache[key] = {k: f[k] for k in f.files}
    return _cache[key]


def get_policy_function(model_name, checkpoint_dir_name="checkpoints", root=None):
    """`{ALGO}_{ENV}[_{INDEX}].npz` under `root/checkpoint_dir_name` -> `obs [B, O] -> actions [B, 2]` (sampling)."""
    root = root or osp.dirname(osp.dirname(osp.abspath(__file__)))
    w = load_population(model_name, osp.join(root, checkpoint_dir_name))
    layout, sfx = population_layout(model_name)
    layers = layer_arrays(w, layout, "default", sfx)
    return lambda obs: _gaussian_head(layers, obs, False)


class PolicyFunction:
    def __init__(self, model_name=None, use_distributional_svo=True, auto_add_svo_to_obs=True,
                 checkpoint_dir_name="best_checkpoints", policy=None, root=None, lcf_dist=None):
        if policy is not None:
            self.policy, self.model_name, self.use_svo = policy, None, False
        else:
            self.policy = get_policy_function(model_name, checkpoint_dir_name, root=root)
            self.model_name = model_name
            self.use_svo = model_name.startswith("copo")
        self.lcf_dist = lcf_dist or (meta_svo_lookup_table.get(model_name) if self.use_svo else None)
        if self.use_svo and self.lcf_dist is None:
            raise KeyError("CoPO population %r has no entry in meta_svo_lookup_table: pass lcf_dist=(mean, std)" % model_name)
        self.existing_svo = dict()
        self.use_distributional_svo = use_distributional_svo
        self.auto_add_svo_to_obs = auto_add_svo_to_obs

    def __call__(self, obs_dict, last_done_dict):
        obs_dict = self.process_svo(obs_dict)
        keys = [k for k in obs_dict if not last_done_dict.get(k, False)]
        actions = self.policy([obs_dict[k] for k in keys])
        return {k: actions[n] for n, k in enumerate(keys)}

    def process_svo(self, obs_dict):
        if not (self.use_svo and self.auto_add_svo_to_obs):
            return obs_dict
        mean, std = self.lcf_dist
        out = {}
        for k, o in obs_dict.items():
            if k not in self.existing_svo:
                self.existing_svo[k] = np.clip(np.random.normal(loc=mean, scale=std), -1, 1) \
                    if self.use_distributional_svo else mean
            out[k] = np.concatenate([o, [(self.existing_svo[k] + 1) / 2]])
        return out

    def reset(self):
        self.existing_svo.clear()

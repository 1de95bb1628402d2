"""Checkpoint wire formats.

* `.npz` populations in either key layout (see get_policy_function.py) <-> the policy part of this build's torch
  models (`FullyConnectedModel / CCModel / CoPOModel`, state-dict keys `_hidden_layers.*`, `_logits.*`).
* Tune checkpoints: a pickle whose `"worker"` entry is a pickle of `{"state": {policy_name: weights}}`
  (copo/eval/get_policy_function_from_checkpoint.py:15-29).
"""
import os
import pickle

import numpy as np
import torch

from .get_policy_function import (PolicyFunction, _LAYERS_TF, _LAYERS_TORCH, _compute_actions_for_tf_policy,
                                  _compute_actions_for_torch_policy2, detect_layout, layer_arrays)


def policy_state_dict(weights, policy_name="default", layer_name_suffix=None):
    """Torch state-dict entries of the policy net from an `.npz`-style dict in either layout."""
    layout = detect_layout(weights)
    if layout == "tf" and layer_name_suffix is None:
        layer_name_suffix = "_1" if any("fc_1_1/" in k for k in weights) else ""
    sd = {}
    for name, (w, b) in zip(_LAYERS_TORCH, layer_arrays(weights, layout, policy_name, layer_name_suffix or "")):
        sd[name + ".weight"] = torch.as_tensor(np.ascontiguousarray(w.T), dtype=torch.float32)
        sd[name + ".bias"] = torch.as_tensor(np.asarray(b), dtype=torch.float32)
    return sd


def load_policy_weights(model, weights, **kw):
    """Copy a population's policy into `model` (value nets / LCF parameters are left as they are)."""
    sd = policy_state_dict(weights, **kw)
    own = model.state_dict()
    for k, v in sd.items():
        assert own[k].shape == v.shape, (k, tuple(own[k].shape), tuple(v.shape))
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v.to(own[k].device))
    # a fused learner that reads these parameters through its transposed mirror must rebuild it: the owner of `model`
    # (PPOPolicyBase._weights_changed / FusedLearner.invalidate_mirror) is told through this hook when it registered one
    hook = getattr(model, "_on_external_write", None)
    if hook is not None:
        hook()
    return sorted(sd)


def export_policy_npz(model, path, layout="torch", policy_name="default", layer_name_suffix=""):
    """Write the policy net of `model` as a population file the reference's `get_policy_function` reads."""
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    out = {}
    for tf_name, torch_name in zip(_LAYERS_TF, _LAYERS_TORCH):
        w, b = sd[torch_name + ".weight"], sd[torch_name + ".bias"]
        if layout == "torch":
            out[torch_name + ".weight"], out[torch_name + ".bias"] = w, b
        else:
            stem = "%s/%s%s" % (policy_name, tf_name, layer_name_suffix)
            out[stem + "/kernel"], out[stem + "/bias"] = np.ascontiguousarray(w.T), b
    np.savez(path, **out)
    return sorted(out)


def get_policy_function_from_checkpoint(algo, ckpt, deterministic=False, policy_name="default"):
    assert os.path.isfile(ckpt), ckpt
    with open(ckpt, "rb") as f:
        blob = pickle.loads(f.read())
    weights = pickle.loads(blob.pop("worker"))["state"][policy_name]
    weights = {k: v for k, v in weights.items() if k != "_optimizer_variables" and "value" not in k}
    # the key layout decides the reader (this build's own checkpoints are torch-layout for every algorithm); the `_1`
    # suffix is the TF-era CoPO convention only (get_policy_function_from_checkpoint.py:15-29)
    layout = detect_layout(weights)
    sfx = "_1" if ("copo" in algo and layout == "tf") else ""
    fn = _compute_actions_for_torch_policy2 if layout == "torch" else _compute_actions_for_tf_policy

    def policy(obs):
        return fn(weights, obs, policy_name=policy_name, layer_name_suffix=sfx, deterministic=deterministic)

    return PolicyFunction(policy=policy)


def save_tune_style_checkpoint(model, path, policy_name="default"):
    """The inverse of the loader above for this build's models (torch key layout, as the reference's torch stack)."""
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    with open(path, "wb") as f:
        f.write(pickle.dumps({"worker": pickle.dumps({"state": {policy_name: state}})}))


def get_lcf_from_checkpoint(trial_path):
    """Last `info/learner/svo[_std]` of the trial's progress.csv (get_policy_function_from_checkpoint.py:51-62)."""
    import pandas as pd
    file = os.path.join(trial_path, "progress.csv")
    assert os.path.isfile(file), "We expect to use progress.csv to extract LCF! The folder should be: %s" % trial_path
    df = pd.read_csv(file)
    last = df.index[-1]
    std = df.loc[last, "info/learner/svo_std"] if "info/learner/svo_std" in df else 0.0
    return df.loc[last, "info/learner/svo"], std

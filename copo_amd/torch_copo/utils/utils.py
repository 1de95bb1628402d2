"""CLI / misc helpers with the reference's names (`copo/torch_copo/utils/utils.py:182-235`); no Ray underneath."""
import argparse
import copy
import datetime
import json
import logging
import numbers
import os

import numpy as np


def get_train_parser():
    """The six flags of the reference's launch scripts (utils/utils.py:206-214)."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--exp-name", type=str, default="")
    parser.add_argument("--num-gpus", type=int, default=0)
    parser.add_argument("--num-seeds", type=int, default=3)
    parser.add_argument("--num-cpus-per-worker", type=float, default=0.5)
    parser.add_argument("--num-gpus-per-trial", type=float, default=0.25)
    parser.add_argument("--test", action="store_true")
    return parser


def initialize_ray(local_mode=False, num_gpus=None, test_mode=False, **kwargs):
    """There is no Ray runtime in this build; kept so that reference-shaped scripts run unchanged.  One process
    drives one GPU; launch with `python -m torch.distributed.run --nproc-per-node N` for N GPUs."""
    os.environ["OMP_NUM_THREADS"] = "1"
    from copo_amd import dist as D
    D.init_from_env()
    print("Successfully initialize the MI355X host loop (world size %d)." % D.world_size())


def setup_logger(debug=False):
    logging.basicConfig(level=logging.DEBUG if debug else logging.WARNING,
                        format="%(asctime)s - %(filename)s[line:%(lineno)d] - %(levelname)s: %(message)s")


def pretty_print(result):
    import json

    def clean(v):
        if isinstance(v, dict):
            return {k: clean(x) for k, x in v.items()}
        try:
            return float(v)
        except (TypeError, ValueError):
            return str(v)
    return json.dumps(clean(result), indent=2, sort_keys=True)


def deep_update(original, new_dict, new_keys_allowed=False, allow_new_subkey_list=None, override_all_if_type_changes=None):
    """Recursive in-place update of a config dict with the semantics of the reference's helper (utils/utils.py:64-108):
    unknown keys raise unless allowed; for top-level keys in `allow_new_subkey_list` new sub-keys may appear; for keys in
    `override_all_if_type_changes` a changed `type` entry replaces the whole sub-dict."""
    sub_ok, by_type = set(allow_new_subkey_list or ()), set(override_all_if_type_changes or ())
    for key, new in new_dict.items():
        if key not in original and not new_keys_allowed:
            raise Exception("Unknown config parameter `{}` ".format(key))
        old = original.get(key)
        if not (isinstance(old, dict) and isinstance(new, dict)):
            original[key] = new
        elif key in by_type and "type" in new and "type" in old and new["type"] != old["type"]:
            original[key] = new
        else:
            deep_update(old, new, True if key in sub_ok else new_keys_allowed)
    return original


def merge_dicts(d1, d2):
    """A new dict: deep copy of d1 deep-updated with d2, new keys allowed (utils/utils.py:50-61)."""
    return deep_update(copy.deepcopy(d1), d2, True, [])


def get_time_str():
    return datetime.datetime.now().strftime("%Y-%m-%d_%H%M")


class SafeJSONEncoder(json.JSONEncoder):
    """JSON encoder for result dicts: numpy arrays -> lists, numpy scalars -> python numbers, NaN -> `nan_str`, anything
    else that does not serialise -> its string (utils/utils.py:155-179)."""

    def __init__(self, nan_str="null", **kwargs):
        super().__init__(**kwargs)
        self.nan_str = nan_str

    def default(self, value):
        try:
            if isinstance(value, np.ndarray):
                return value.tolist()
            if isinstance(value, np.bool_):
                return bool(value)
            if np.isnan(value):
                return self.nan_str
            if isinstance(value, numbers.Integral):
                return int(value)
            if isinstance(value, numbers.Number):
                return float(value)
            return super().default(value)
        except Exception:
            return str(value)

"""`train()` with the reference's signature (`copo/torch_copo/utils/train.py:27-199`), minus Ray Tune.

Expands `grid_search` entries and seeds into trials, runs them one after another in this process (one process
per GPU; start several with torch.distributed.run for data-parallel trials), stops on `timesteps_total`, writes
`progress.csv` / `result.json` / checkpoints under `~/copo_results/<exp_name>/<trial>/`.
"""
import copy
import csv
import json
import os
import pickle

import numpy as np

from copo_amd import dist as D
from copo_amd.engine import expand_grid, grid_search
from copo_amd.torch_copo.utils.utils import initialize_ray

REPORT_COLUMNS = ("training_iteration", "timesteps_total", "time_total_s", "episode_reward_mean", "success", "crash",
                  "out", "max_step", "length", "cost", "rc")


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        key = prefix + str(k)
        if isinstance(v, dict):
            out.update(_flatten(v, key + "/"))
        elif isinstance(v, (int, float, np.floating, np.integer, str)) or v is None:
            out[key] = v
    return out


def train(trainer, config, stop, exp_name, num_seeds=1, num_gpus=0, test_mode=False, suffix="", checkpoint_freq=10,
          keep_checkpoints_num=None, start_seed=0, local_mode=False, save_pkl=True, custom_callback=None,
          max_failures=1, wandb_key_file=None, wandb_project=None, wandb_team="copo", wandb_log_config=True,
          init_kws=None, local_dir=None, verbose=None, **kwargs):
    initialize_ray(test_mode=test_mode, local_mode=local_mode, num_gpus=num_gpus, **(init_kws or {}))
    used_config = {
        "seed": grid_search([i * 100 + start_seed for i in range(num_seeds)]) if num_seeds is not None else None,
        "log_level": "DEBUG" if test_mode else "INFO",
        "callbacks": custom_callback if custom_callback else None,
    }
    if config:
        used_config.update(config)
    trainer_name = trainer if isinstance(trainer, str) else getattr(trainer, "_name", trainer.__name__)
    if not isinstance(stop, dict) and stop is not None:
        assert np.isscalar(stop)
        stop = {"timesteps_total": int(stop)}
    stop = stop or {}
    verbose = (2 if test_mode else 1) if verbose is None else verbose
    root = os.path.join(local_dir or os.path.expanduser("~/copo_results"), exp_name)
    trials, frames = expand_grid(used_config), []
    for ti, trial_cfg in enumerate(trials):
        env_name = trial_cfg["env"] if isinstance(trial_cfg["env"], str) else trial_cfg["env"].__name__
        tag = "%s_%s_%05d_seed%s" % (trainer_name, env_name, ti, trial_cfg.get("seed"))
        tdir = os.path.join(root, tag)
        if D.rank() == 0:
            os.makedirs(tdir, exist_ok=True)
            with open(os.path.join(tdir, "params.json"), "w") as f:
                json.dump({k: str(v) for k, v in trial_cfg.items()}, f, indent=1)
        failures = 0
        while True:
            try:
                rows = _run_trial(trainer, trial_cfg, stop, tdir, checkpoint_freq, keep_checkpoints_num, verbose)
                break
            except Exception:
                failures += 1
                if failures > (0 if test_mode else max_failures):
                    raise
        frames.append(rows)
    if save_pkl and D.rank() == 0:
        with open(os.path.join(root, "{}-{}{}.pkl".format(exp_name, trainer_name, "" if not suffix else "-" + suffix)),
                  "wb") as f:
            pickle.dump(frames, f)
    return frames


def _run_trial(trainer_cls, trial_cfg, stop, tdir, checkpoint_freq, keep_checkpoints_num, verbose):
    algo = trainer_cls(config=copy.deepcopy(trial_cfg))
    rows, ckpts, writer, fcsv = [], [], None, None
    try:
        while True:
            result = algo.train()
            flat = _flatten({k: v for k, v in result.items() if k != "config"})
            rows.append(flat)
            if D.rank() == 0:
                if writer is None:
                    fcsv = open(os.path.join(tdir, "progress.csv"), "w", newline="")
                    writer = csv.DictWriter(fcsv, fieldnames=list(flat.keys()), extrasaction="ignore")
                    writer.writeheader()
                writer.writerow(flat)
                fcsv.flush()
                with open(os.path.join(tdir, "result.json"), "a") as f:
                    f.write(json.dumps(flat, default=float) + "\n")
                if verbose:
                    print(" | ".join("%s=%s" % (c, ("%.4g" % result[c]) if isinstance(result.get(c), (int, float, np.floating))
                                                else result.get(c)) for c in REPORT_COLUMNS if c in result), flush=True)
            it = result["training_iteration"]
            if checkpoint_freq and it % checkpoint_freq == 0:
                ckpts.append((result.get("episode_reward_mean", 0.0), algo.save_checkpoint(os.path.join(tdir, "checkpoints"))))
                if keep_checkpoints_num and len(ckpts) > keep_checkpoints_num and D.rank() == 0:
                    ckpts.sort(key=lambda x: (x[0] if x[0] == x[0] else -1e30))
                    _, worst = ckpts.pop(0)
                    if os.path.exists(worst):
                        os.remove(worst)
            if any(k in result and result[k] >= v for k, v in stop.items()):
                break
        if checkpoint_freq:
            algo.save_checkpoint(os.path.join(tdir, "checkpoints"))
    finally:
        if fcsv is not None:
            fcsv.close()
        algo.stop()
    return rows

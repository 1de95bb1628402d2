"""GPU time of every phase of a training iteration (synchronised wall clock around each host-level call), C2 workload."""
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_trainer  # noqa: E402

t = make_trainer(256, 40)
acc, cnt = defaultdict(float), defaultdict(int)


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[label] += time.perf_counter() - t0
        cnt[label] += 1
        return r
    setattr(obj, name, g)


for _ in range(5):
    t.train()
wrap(t.sampler, "sample")
wrap(t.policy, "postprocess_trajectory")
wrap(t, "valid_rows")
wrap(t, "coordinated_advantage")
wrap(t.policy, "prepare_sgd")
wrap(t.policy, "plan_epoch")
wrap(t.policy, "run_sgd")
wrap(t.policy, "run_meta")
wrap(t, "episode_metrics")
wrap(t, "training_step")
wrap(t, "train")
N = 10
for _ in range(N):
    t.train()
for k in ("train", "training_step", "sample", "postprocess_trajectory", "valid_rows", "coordinated_advantage", "prepare_sgd",
          "run_sgd", "run_meta", "plan_epoch", "episode_metrics"):
    print("%-24s %8.3f ms / iteration  (%d calls)" % (k, acc[k] / N * 1e3, cnt[k] // N))
t.stop()

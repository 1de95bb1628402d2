"""A/B of differently compiled libcopo_hip copies on the bench workload: `python scripts/ab_lib.py <lib.so> [iters]` loads that
library (copo_amd._libsel.PATH, set before the package is imported), trains a few iterations of BASELINE configs[1] and prints
the synchronised phase split plus the plain iteration time.  One process per library; the caller alternates them on one box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copo_amd._libsel as sel  # noqa: E402

sel.PATH = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-" else None
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
import torch  # noqa: E402
import bench  # noqa: E402

tr = bench.make_trainer(256, 40, graphs=True)
for _ in range(int(os.environ.get('AB_WARM', '6'))):
    tr.train()
torch.cuda.synchronize()
a0 = tr._counters["num_agent_steps_sampled"]
t0 = time.perf_counter()
marks = [t0]
for _ in range(iters):
    tr.train()
    marks.append(time.perf_counter())        # (every iteration ends with its one host read)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
per_iter = [round((b - a) * 1e3, 2) for a, b in zip(marks, marks[1:])]
rows = tr._counters["num_agent_steps_sampled"] - a0
host = {k: round(v, 2) for k, v in tr._timers.items()}      # host time of the LAST free-running iteration's calls (no synchronisation inside)
ph = bench.measure_phases(tr, iters=6)
print(json.dumps({"lib": os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "-", "ms_per_iter": round(dt / iters * 1e3, 3),
                  "agent_steps_per_s": round(rows / dt, 1), "sgd_ms": ph["sgd_ms"], "meta_ms": ph["meta_ms"],
                  "sample_ms": ph["sample_ms"], "host_ms": host, "per_iter_ms": per_iter, "lcf": [float(x) for x in tr.policy.model.lcf_parameters.detach().cpu()]}))
tr.stop()

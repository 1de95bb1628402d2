"""Which framework (ATen) kernels does one training iteration of the bench workload still launch, and from where?
torch.profiler over a few iterations, grouped by the python frames that issued them."""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_trainer  # noqa: E402

# COPO_ITER_NO_GRAPHS=1: every captured region runs eagerly, so the framework kernels INSIDE the graphs show up with their frames too
t = make_trainer(256, 40, graphs=not os.environ.get("COPO_ITER_NO_GRAPHS"))
for _ in range(6):
    t.train()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        t.train()
    torch.cuda.synchronize()
by = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if "/copo_amd/" in f or "bench.py" in f]
    where = " <- ".join(f.split("/copo_amd/")[-1] for f in frames[:2]) if frames else "?"
    key = (ev.name, where)
    by[key][0] += 1
    by[key][1] += sum(k.duration for k in ev.kernels)
rows = sorted(by.items(), key=lambda kv: -kv[1][0])
print("top-level ATen ops that launch device kernels, per iteration (count, device us) -- %d iterations profiled" % N)
tot = 0
for (name, where), (n, us) in rows[:60]:
    print("%6.1f  %8.1f us  %-28s %s" % (n / N, us / N, name, where))
    tot += n
print("total %.1f launching ops per iteration" % (tot / N))
t.stop()

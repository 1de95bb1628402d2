"""HBM roofline of the streaming learn-side ops at saturating sizes (SURVEY.md section 8d byte counts)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copo_amd import _capi, ops  # noqa: E402

PEAK = 8000.0


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


dev = "cuda"
out = []
# GAE x3: reads 3 x (r, V) + flags, writes 3 x (A, target): 49 B per agent-step
T, M = 8, 4 * 1024 * 1024
rew, val = torch.randn(3, T, M, device=dev), torch.randn(3, T, M, device=dev)
flags = (torch.rand(T, M, device=dev) < 0.97).to(torch.uint8) | ((torch.rand(T, M, device=dev) < 0.01).to(torch.uint8) << 1)
adv, tgt = torch.empty_like(rew), torch.empty_like(rew)
t = timed(lambda: ops.gae3(rew, val, flags, [0.99, 0.99, 1.0], 0.95, adv, tgt))
out.append(dict(op="gae3 (3 heads, T=8)", agent_steps=T * M, bytes_per_unit=49, us=t * 1e6, GBps=T * M * 49 / t * 1e-9))
del rew, val, adv, tgt, flags
# coordinated advantage + standardisation: 2 passes x 12 B read + 8 B write = 32 B per agent-step
B = 64 * 1024 * 1024
a, na, ga, lcf = (torch.randn(B, device=dev) for _ in range(4))
valid = torch.ones(B, dtype=torch.uint8, device=dev)
mixed, norm, gstd = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
stats = torch.zeros(_capi.LCF_STATS_DOUBLES, dtype=torch.float64, device=dev)


def mix():
    ops.lcf_mix_partial(a, na, ga, lcf, valid, mixed, stats)
    ops.lcf_mix_apply(mixed, ga, valid, stats, norm, gstd)


t = timed(mix)
out.append(dict(op="lcf_mix partial + apply", agent_steps=B, bytes_per_unit=32, us=t * 1e6, GBps=B * 32 / t * 1e-9))
del a, na, ga, lcf, valid, mixed, norm, gstd
# mean-field centralised-critic observation: per row reads its own obs row + the rows of the neighbours within 10 m
# (+ their actions), writes 2 O + A floats
R, N, O, A, K = 8192, 40, 91, 2, 8
obs = torch.rand(R, N, O, device=dev)
act = torch.rand(R, N, A, device=dev)
flags = torch.ones(R, N, dtype=torch.uint8, device=dev)
nbr = torch.randint(0, N, (R, N, K), dtype=torch.int32, device=dev)
cnt = torch.randint(0, 5, (R, N), dtype=torch.int32, device=dev)
cc = torch.empty(R, N, 2 * O + A, device=dev)
t = timed(lambda: ops.cc_fuse("mf", obs, act, flags, nbr, cnt, True, 4, out=cc))
kbar = float(cnt.float().mean())
bpu = (1 + kbar) * (4 * O + 4 * A) + 4 * (2 * O + A) + 4 * K + 5
out.append(dict(op="cc_fuse_mf (O=91, mean %.1f neighbours)" % kbar, agent_steps=R * N, bytes_per_unit=round(bpu), us=t * 1e6,
                GBps=R * N * bpu / t * 1e-9))
for o in out:
    o["hbm_frac"] = round(o["GBps"] / PEAK, 4)
    o["us"], o["GBps"] = round(o["us"], 1), round(o["GBps"], 1)
    print(json.dumps(o))

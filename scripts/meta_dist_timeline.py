"""Where the LCF meta passes spend their time, local vs the data-parallel path with one rank (COPO_FORCE_DIST=1): synchronised wall
clock around the row store, every pass, the join of the side stream.  usage: [COPO_FORCE_DIST=1] python scripts/meta_dist_timeline.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from copo_amd import dist as D
D.init_from_env("cuda")
tr = bench.make_trainer(256, 40, graphs=True, pretrained=True)
for _ in range(6):
    tr.train()
pol = tr.policy
acc = {}


def wrap(obj, name, label, sync_inside=False):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        acc.setdefault(label, []).append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
        return r
    setattr(obj, name, g)


wrap(pol.fused, "meta_rows", "row store (once)")
wrap(pol, "_run_meta_batched", "one pass (queued / done)")
wrap(pol, "run_meta", "run_meta total")
wrap(pol, "plan_epoch", "plan_epoch")
for _ in range(4):
    tr.train()
print("dist" if D.is_dist() else "local", "-- ms: host time until the call returned / until the device was idle, mean over calls")
for k, v in acc.items():
    h = sum(x[0] for x in v) / len(v)
    d = sum(x[1] for x in v) / len(v)
    print("  %-28s calls per iteration %5.1f   host %7.3f   done %7.3f" % (k, len(v) / 4, h, d))
tr.stop()
D.shutdown()

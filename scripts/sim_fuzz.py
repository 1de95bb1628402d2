"""Randomised parity hunt: HIP simulator vs the CPU oracle, raw bits of every output, over random maps / populations /
beam counts / launch shapes / action sources.  usage: python scripts/sim_fuzz.py [n_cases] [steps] [packed]
(`packed`: the packed launch shape of sim_packed.hip, 2 .. 16 scenes per workgroup, on configurations that allow it)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as ol
from copo_amd.sim import SimConfig, VecSim

KEYS = ("obs", "rew", "nei_rew", "glob_rew", "flags", "nbr_cnt", "mf_cnt", "lcf", "info", "agent_id")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
packed = len(sys.argv) > 3 and sys.argv[3] == "packed"
rng = np.random.RandomState(int(os.environ.get("COPO_FUZZ_SEED", "2026")))
bad = 0
for case in range(n_cases):
    name = ["intersection", "roundabout", "parkinglot", "tollgate", "bottleneck", "pgmap"][rng.randint(6)]
    kw = dict(sequence=int(rng.randint(1, 5)), seed=int(rng.randint(1000))) if name == "pgmap" else {}
    tabs = __import__("copo_amd.maps", fromlist=["x"]).MAP_BUILDERS[name](**kw)
    N = int(rng.randint(2, min(64, tabs.n_spawns) + 1))
    E = int(rng.randint(1, 24 if packed else 7))
    lasers = int(rng.choice([30, 72, 72, 72, 120, 240]))
    block = int(rng.choice([-2, -3, -5, -8, -8, -11, -16])) if packed else int(rng.choice([64, 64, 128, 256, 512, 1024]))
    # round 6: the Tollgate's booth rules and static boxes (absent / seen by the LiDAR / hidden from it), LiDAR range, body margin -- a third of
    # the cases keep the per-map defaults
    r6 = {}
    if rng.randint(3):
        r6 = dict(lidar_range=float(rng.choice([20.0, 40.0, 50.0])), body_margin=float(rng.choice([0.0, 0.5, 0.75, 1.0])))
        if name == "tollgate":
            r6.update(toll_buildings=int(rng.randint(3)), toll_early_exit=int(rng.randint(2)), speed_reward=float(rng.choice([0.0, 0.1])),
                      toll_speed_limit=float(rng.choice([0.0, 3.0 / 3.6, 2.0])), overspeed_penalty=float(rng.choice([0.0, 0.5])))
    cfg = SimConfig(map=name, map_kwargs=kw, num_envs=E, num_agents=N, num_lasers=lasers, horizon=int(rng.randint(40, 200)),
                    nbr_k=int(rng.randint(1, max(2, min(N, 9 if packed else 12)))), delay_done=int(rng.randint(0, 30)), enable_lcf=bool(rng.randint(2)),
                    neighbours_distance=float(rng.choice([20.0, 40.0] if packed else [10.0, 20.0, 40.0])),
                    reverse_acc=float(rng.choice([0.0, 0.0, 0.0, 2.9])),        # (round 3: optional reverse gear)
                    **r6)
    g, o = VecSim(cfg), ol.OracleSim(cfg)
    try:
        g.set_block(block)
    except Exception as ex:      # (packed: more LDS than a workgroup may have with this many scenes)
        block = -2
        g.set_block(block)
    chunk = int(rng.choice([0, 0, 3, 8, 13, 20]))                               # (round 3: LiDAR fans per pass of the one-wave shape)
    try:
        g.set_chunk(chunk)
    except Exception:            # (packed: that many fans of that many rays do not fit the workgroup's LDS)
        chunk = 0
    seeds = rng.randint(0, 2 ** 31, E).astype(np.uint64)
    go, oo = g.reset(seeds), o.reset(seeds)
    mode = rng.randint(3)
    fail = None
    for t in range(steps):
        if mode == 0:
            a = np.stack([rng.normal(0, 0.15, (E, N)), rng.uniform(-0.3, 1.0, (E, N))], -1)
        elif mode == 1:          # lane keeping on the oracle's observation (side_lasers = 0 layouts only; else random)
            ob = oo["obs"]
            c2 = cfg.ego_dim - 7 if cfg.lane_line_lasers == 0 else None
            hd = ob[..., (cfg.side_lasers or 2)]
            psi = np.arcsin(np.clip((0.5 - hd) * 2, -1, 1))
            a = np.stack([np.clip(-1.5 * psi + rng.normal(0, 0.05, psi.shape), -1, 1), np.full((E, N), 0.6)], -1)
        else:
            a = rng.uniform(-1.2, 1.2, (E, N, 2))
        a = a.astype(np.float32)
        go, oo = g.step(torch.from_numpy(a).cuda()), o.step(a)
        pres = (oo["flags"] & 0x41) != 0
        before = ((oo["flags"] & 1) != 0) | (((oo["flags"] & 0x40) != 0) & ((oo["flags"] & 0x80) == 0))
        for k in KEYS + ("nbr_idx", "nbr_dist"):
            x, y = go[k].cpu().numpy(), oo[k]
            if k == "obs":
                x, y = x[pres], y[pres]
            elif k in ("nbr_idx", "nbr_dist"):
                x, y = x[before], y[before]
            xb = x.view(np.uint32) if x.dtype == np.float32 else x
            yb = y.view(np.uint32) if y.dtype == np.float32 else y
            if not np.array_equal(xb, yb):
                fail = (t, k, int((xb != yb).sum()))
                break
        if fail:
            break
    print("case %3d %-12s N=%2d E=%d lasers=%3d block=%4d chunk=%2d rev=%.1f O=%3d mode=%d r6=%s: %s" % (case, name, N, E, lasers, block, chunk, cfg.reverse_acc, cfg.obs_dim, mode,
          ",".join("%s=%g" % (k[:9], v) for k, v in sorted(r6.items())) or "-", "ok" if not fail else "MISMATCH step %d %s (%d words)" % fail), flush=True)
    bad += 1 if fail else 0
    g.close()
    o.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)

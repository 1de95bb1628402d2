"""Micro-benchmark of the simulator step kernel: time per launch vs E and workgroup size (HIP events)."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from copo_amd.sim import SimConfig, VecSim


def run(E, N, lasers, block, map_name, steps=200, warm=30):
    cfg = SimConfig(map=map_name, num_envs=E, num_agents=N, num_lasers=lasers)
    sim = VecSim(cfg, with_info=False)
    sim.set_block(block)
    sim.reset()
    act = torch.empty(E, sim.N, 2, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.stack([torch.randn(E, sim.N, device="cuda", generator=gen) * 0.1,
                         torch.rand(E, sim.N, device="cuda", generator=gen)], -1).contiguous() for _ in range(16)]
    for i in range(warm):
        sim.step(acts[i % 16])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    alive = 0
    ev0.record()
    for i in range(steps):
        sim.step(acts[i % 16])
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    acted = float((sim.out["flags"] & 1).float().mean())
    bytes_per = 202 + 4 * sim.O
    agents = E * sim.N
    sim.close()
    return dict(E=E, N=sim.N, O=sim.O, block=block, us_per_step=round(us, 2), slots_per_s=round(agents / us * 1e6),
                acted_frac=round(acted, 3), algo_GBps=round(agents * bytes_per / us * 1e-3, 1),
                hbm_frac=round(agents * bytes_per / us * 1e-3 / 8000, 4))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="intersection")
    ap.add_argument("--N", type=int, default=40)
    ap.add_argument("--lasers", type=int, default=72)
    ap.add_argument("--E", type=int, nargs="+", default=[256, 1024, 4096, 16384])
    ap.add_argument("--blocks", type=int, nargs="+", default=[256, 512, 1024])
    a = ap.parse_args()
    for E in a.E:
        for b in a.blocks:
            print(json.dumps(run(E, a.N, a.lasers, b, a.map)), flush=True)

"""Micro-benchmark of the simulator step kernel: time per launch vs E and workgroup size (HIP events)."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

if "--lib" in sys.argv:        # a profiling / comparison build of the library (never the default)
    import copo_amd._libsel as _S
    _S.PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from copo_amd.sim import SimConfig, VecSim


def cruise_actions(obs, gen, speed=0.25):
    """Lane-keeping controller on the observation (columns 2 / 8: heading error / offset in the lane; 3: speed; 10: check
    point to the right): keeps vehicles on their routes, so scenes stay populated like those of a trained policy."""
    psi = torch.asin(((0.5 - obs[..., 2]) * 2).clamp(-1, 1))
    lat = -(obs[..., 8] - 0.5) * 4.5
    aim = (obs[..., 10] - 0.5) * 2            # navigation: check point to the right (+) / left (-)
    steer = (-1.5 * psi - 0.25 * lat - 0.8 * aim + 0.02 * torch.randn(psi.shape, device=obs.device, generator=gen)).clamp(-1, 1)
    thr = ((speed - obs[..., 3]) * 8.0).clamp(-1, 1)
    return torch.stack([steer, thr], -1).contiguous()


def run(E, N, lasers, block, map_name, steps=200, warm=30, policy="random", chunk=0):
    cfg = SimConfig(map=map_name, num_envs=E, num_agents=N, num_lasers=lasers)
    sim = VecSim(cfg, with_info=False)
    sim.set_block(block)
    sim.set_chunk(chunk)
    out = sim.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    if policy == "random":
        acts = [torch.stack([torch.randn(E, sim.N, device="cuda", generator=gen) * 0.1,
                             torch.rand(E, sim.N, device="cuda", generator=gen)], -1).contiguous() for _ in range(16)]
        for i in range(warm):
            sim.step(acts[i % 16])
    else:
        # closed-loop warm-up, then record the controller's actions over the timed stretch and replay them from the saved
        # state (the simulator is deterministic), so that the timed loop holds nothing but simulator launches
        for i in range(warm + 120):
            out = sim.step(cruise_actions(out["obs"], gen))
        st, env = sim.get_state()
        acts = []
        for i in range(steps):
            a = cruise_actions(out["obs"], gen)
            acts.append(a)
            out = sim.step(a)
        sim.set_state(st, env)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(steps):
        sim.step(acts[i % len(acts)])
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    present = float(((sim.out["flags"] & 0x41) != 0).float().mean())
    bytes_per = 202 + 4 * sim.O
    agents = E * sim.N
    sim.close()
    return dict(E=E, N=sim.N, O=sim.O, block=block, chunk=chunk, policy=policy, us_per_step=round(us, 2), slots_per_s=round(agents / us * 1e6),
                present_frac=round(present, 3), present_GBps=round(agents * present * bytes_per / us * 1e-3, 1),
                hbm_frac_present=round(agents * present * bytes_per / us * 1e-3 / 8000, 4),
                hbm_frac_all_slots=round(agents * bytes_per / us * 1e-3 / 8000, 4))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="intersection")
    ap.add_argument("--N", type=int, default=40)
    ap.add_argument("--lasers", type=int, default=72)
    ap.add_argument("--E", type=int, nargs="+", default=[256, 1024, 4096, 16384])
    ap.add_argument("--blocks", type=int, nargs="+", default=[256, 512, 1024])
    ap.add_argument("--policy", default="random", choices=["random", "cruise"])
    ap.add_argument("--lib", default=None)
    ap.add_argument("--chunks", type=int, nargs="+", default=[0])
    a = ap.parse_args()
    for E in a.E:
        for b in a.blocks:
            for ch in a.chunks:
                print(json.dumps(run(E, a.N, a.lasers, b, a.map, policy=a.policy, chunk=ch)), flush=True)

"""Per-phase time of the simulator step kernel from in-kernel clock64() stamps (mean over scenes, cycles), and the share of
scenes whose neighbour lists took the register formulation (neighbours_fast) vs the pair-parallel one.
usage: phase_sim.py E block [random|cruise]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
from bench_sim import cruise_actions
E, block = int(sys.argv[1]), int(sys.argv[2])
policy = sys.argv[3] if len(sys.argv) > 3 else "random"
sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
sim.set_block(block)
out = sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.1, torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous() for _ in range(8)]


def act(i, out):
    return acts[i % 8] if policy == "random" else cruise_actions(out["obs"], gen)


for i in range(40 if policy == "random" else 250):
    out = sim.step(act(i, out))
dbg = torch.zeros(E, 8, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
acc = torch.zeros(6, dtype=torch.float64)
n, fast, slow = 20, 0, 0
for i in range(n):
    out = sim.step(act(i, out))
    torch.cuda.synchronize()
    d = dbg[:, :7].double()
    acc += (d[:, 1:] - d[:, :-1]).mean(0).cpu()
    fast += int((dbg[:, 7] == 1).sum())
    slow += int((dbg[:, 7] == 2).sum())
    dbg[:, 7] = 0
names = ["P0 dynamics", "P1 collision", "P2 project/respawn", "P3 neighbours", "P4 writeback/ego", "P5 lidar+obs"]
tot = float(acc.sum() / n)
for k, v in zip(names, (acc / n).tolist()):
    print("%-22s %9.0f cycles  %5.1f%%" % (k, v, 100 * v / tot))
print("block lifetime %.0f cycles = %.1f us @2.4GHz" % (tot, tot / 2400))
print("present slots %.3f; neighbour lists: register formulation %d scenes, pair-parallel %d (%.2f %% declined)"
      % (float(((out["flags"] & 0x41) != 0).float().mean()), fast, slow, 100.0 * slow / max(1, fast + slow)))

"""Per-phase time of the simulator step kernel from in-kernel clock64() stamps (mean over scenes, cycles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
E, block = int(sys.argv[1]), int(sys.argv[2])
sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
sim.set_block(block)
sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.1, torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous() for _ in range(8)]
for i in range(40):
    sim.step(acts[i % 8])
dbg = torch.zeros(E, 8, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
acc = torch.zeros(6, dtype=torch.float64)
n = 20
for i in range(n):
    sim.step(acts[i % 8])
    torch.cuda.synchronize()
    d = dbg[:, :7].double()
    acc += (d[:, 1:] - d[:, :-1]).mean(0).cpu()
names = ["P0 dynamics", "P1 collision", "P2 project/respawn", "P3 neighbours", "P4 writeback/ego", "P5 lidar+obs"]
tot = float(acc.sum() / n)
for k, v in zip(names, (acc / n).tolist()):
    print("%-22s %9.0f cycles  %5.1f%%" % (k, v, 100 * v / tot))
print("block lifetime %.0f cycles = %.1f us @2.4GHz" % (tot, tot / 2400))

"""Work counters of the LiDAR phase (profiling build `make -C copo_amd/csrc prof SKIP=256`: nothing compiled out, the kernel
counts queued pairs / pair batches / box tests / test batches / hits per scene into the [E][16] debug rows).
usage: lidar_counts.py E block [random|cruise]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copo_amd._libsel as _S
_S.PATH = os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_256.so")
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_sim import cruise_actions
E, block = int(sys.argv[1]), int(sys.argv[2])
policy = sys.argv[3] if len(sys.argv) > 3 else "cruise"
sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
sim.set_block(block)
out = sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
for i in range(150):
    out = sim.step(cruise_actions(out["obs"], gen) if policy == "cruise" else
                   torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.1, torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous())
dbg = torch.zeros(E, 16, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
n = 20
pres = 0.0
for i in range(n):
    out = sim.step(cruise_actions(out["obs"], gen))
    pres += float(((out["flags"] & 0x41) != 0).float().sum()) / E
torch.cuda.synchronize()
c = dbg[:, 8:14].double().mean(0).cpu() / n
print("per scene and step: present %.1f | queued pairs %.1f  pair batches %.2f (fill %.0f %%) | box tests %.1f  test batches %.2f (fill %.0f %%)  hits %.1f (%.0f %% of tests) | pairs with a window %.1f"
      % (pres / n, c[0], c[1], 100 * c[0] / max(c[1] * 64, 1), c[2], c[3], 100 * c[2] / max(c[3] * 64, 1), c[4], 100 * c[4] / max(c[2], 1), c[5]))

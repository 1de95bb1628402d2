"""Work counters of the detector beams (profiling build `make -C copo_amd/csrc prof SKIP=16384`: nothing compiled out, the one-wave
shape counts candidate batches / near pairs / pair batches / beam tests / test batches / hits per scene into the [E][16] debug rows).
usage: detector_counts.py E [map]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copo_amd._libsel as _S
_S.PATH = os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_16384.so")
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
E = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "tollgate"
sim = VecSim(SimConfig(map=name, num_envs=E, num_agents=40), with_info=False)
sim.set_block(64)
out = sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
def act():
    return torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.02, 0.2 + 0.4 * torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous()
for i in range(150):
    out = sim.step(act())
dbg = torch.zeros(E, 16, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
n, pres = 20, 0.0
for i in range(n):
    out = sim.step(act())
    pres += float(((out["flags"] & 0x41) != 0).float().sum()) / E
torch.cuda.synchronize()
c = dbg[:, 8:14].double().mean(0).cpu() / n
print("per scene and step: present %.1f | candidate batches %.1f  near pairs %.1f | pair batches %.2f | beam tests %.1f  test batches %.2f (fill %.0f %%)  hits %.1f (%.0f %% of tests)"
      % (pres / n, c[0], c[1], c[2], c[3], c[4], 100 * c[3] / max(c[4] * 64, 1), c[5], 100 * c[5] / max(c[3], 1)))

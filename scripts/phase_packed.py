"""Per-phase time of the PACKED simulator step kernel (sim_packed.hip) from clock64() stamps of the profiling build 4096
(`make -C copo_amd/csrc prof SKIP=4096`): for every scene's wave, released-from-barrier and arrived-at-barrier times.
usage: python scripts/phase_packed.py E scenes_per_workgroup [random|cruise]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import copo_amd._libsel as S
S.PATH = os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_4096.so")
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
from bench_sim import cruise_actions
E, SC = int(sys.argv[1]), int(sys.argv[2])
policy = sys.argv[3] if len(sys.argv) > 3 else "cruise"
N = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=N), with_info=False)
sim.set_block(-SC)
out = sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
for i in range(250):
    out = sim.step(cruise_actions(out["obs"], gen))
dbg = torch.zeros(E, 16, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
n = 20
rel = torch.zeros(6, dtype=torch.float64)
own = torch.zeros(6, dtype=torch.float64)
life = 0.0
for i in range(n):
    out = sim.step(cruise_actions(out["obs"], gen))
    torch.cuda.synchronize()
    d = dbg.double()
    rel += (d[:, 1:7] - d[:, 0:6]).mean(0).cpu()                      # barrier k released -> barrier k + 1 released (last: end of the wave)
    arr = torch.cat([d[:, 8:13], d[:, 6:7]], 1)                        # arrival at barrier k + 1 (last: end of the wave)
    own += (arr - d[:, 0:6]).mean(0).cpu()
    wg = d.view(E // SC, SC, 16)
    life += float((wg[:, :, 6].max(1).values - wg[:, :, 0].min(1).values).mean())
names = ["A1 state + dynamics (packed)", "S1 collision (scene)", "A2 projection (packed)", "S2 respawn / records (scene)", "A3 walk + outputs (packed)", "S3 exact lists + LiDAR (scene)"]
print("# mean over the scenes' waves, cycles: phase = barrier release to next release; own = release to this wave's arrival at the next barrier")
for k, nm in enumerate(names):
    print("%-34s phase %8.0f   own work of a wave %8.0f" % (nm, rel[k] / n, own[k] / n))
print("workgroup lifetime %.0f cycles = %.1f us @2.4 GHz; present slots %.3f" % (life / n, life / n / 2400, float(((out["flags"] & 0x41) != 0).float().mean())))

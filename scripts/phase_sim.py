"""Per-phase time of the simulator step kernel from in-kernel clock64() stamps (mean over scenes, cycles), and the share of
scenes whose neighbour lists took the register formulation (neighbours_fast) vs the pair-parallel one.
usage: phase_sim.py E block [random|cruise]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("COPO_LIB_PROF"):
    import copo_amd._libsel as S
    S.PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "copo_amd", "lib", "libcopo_hip_prof_%s.so" % os.environ["COPO_LIB_PROF"])
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
from bench_sim import cruise_actions
E, block = int(sys.argv[1]), int(sys.argv[2])
policy = sys.argv[3] if len(sys.argv) > 3 else "random"
sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
sim.set_block(block)
if os.environ.get("TRAINER_OUTPUTS"):       # the sampler's output set: no neighbour distances (trainer.py: the learner does not read them)
    sim.out["nbr_dist"] = None
    sim._step_out = sim.make_step_out(sim.out)
out = sim.reset()
gen = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.1, torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous() for _ in range(8)]


def act(i, out):
    return acts[i % 8] if policy == "random" else cruise_actions(out["obs"], gen)


for i in range(40 if policy == "random" else 250):
    out = sim.step(act(i, out))
ROLES = os.environ.get("COPO_LIB_PROF") == "512"       # (profiling build `make prof SKIP=512`: the wave roles' own finishing times)
dbg = torch.zeros(E, 16 if ROLES else 8, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
acc = torch.zeros(6, dtype=torch.float64)
n, fast, slow = 20, 0, 0
for i in range(n):
    out = sim.step(act(i, out))
    torch.cuda.synchronize()
    d = dbg[:, :7].double()
    acc += (d[:, 1:] - d[:, :-1]).mean(0).cpu()
    if ROLES:
        r = (dbg[:, 8:11] - dbg[:, 4:5]).double()
        racc = r.mean(0).cpu() + (racc if i else 0)
        dbg[:, 8:11] = 0
        wall = (dbg[:, 12] - dbg[:, 11]).double().mean().item() * 10.0 + (wall if i else 0)      # ns (100 MHz counter)
        span = (dbg[:, 12].max() - dbg[:, 11].min()).item() * 10.0 + (span if i else 0)
    fast += int(((dbg[:, 7] == 1) | (dbg[:, 7] >= 16)).sum())
    slow += int((dbg[:, 7] == 2).sum())
    dbg[:, 7] = 0
names = ["P0 dynamics", "P1 collision", "P2 project/respawn", "P3 neighbours", "P4 writeback/ego", "P5 lidar+obs"]
tot = float(acc.sum() / n)
for k, v in zip(names, (acc / n).tolist()):
    print("%-22s %9.0f cycles  %5.1f%%" % (k, v, 100 * v / tot))
print("block lifetime %.0f cycles = %.1f us @2.4GHz" % (tot, tot / 2400))
print("present slots %.3f; neighbour lists: register formulation %d scenes, pair-parallel %d (%.2f %% declined)"
      % (float(((out["flags"] & 0x41) != 0).float().mean()), fast, slow, 100.0 * slow / max(1, fast + slow)))
if ROLES:
    print("after P3's stamp, cycles until: wave 0 done (write-back, ego/navigation block) %.0f | wave 1 done (neighbour lists) %.0f | last LiDAR wave done %.0f"
          % tuple((racc / n).tolist()))
    print("constant-rate clock: workgroup lifetime %.2f us -> shader clock %.2f GHz; first workgroup start to last end %.2f us"
          % (wall / n / 1e3, tot / (wall / n), span / n / 1e3))

"""When do the workgroups of ONE simulator step launch start and end (constant-rate 100 MHz counter, profiling build
`make -C copo_amd/csrc prof SKIP=512`)?  usage: COPO_LIB_PROF=512 python scripts/dispatch_skew.py E block"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copo_amd._libsel as S
S.PATH = os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_512.so")
import torch
from copo_amd import _capi
from copo_amd.sim import SimConfig, VecSim
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_sim import cruise_actions
E, block = int(sys.argv[1]), int(sys.argv[2])
LIVE = len(sys.argv) > 3 and sys.argv[3] == "live"          # the bench trainer's own scenes and policy instead of the lane-keeping controller
if LIVE:
    import bench
    tr = bench.make_trainer(E, 40, graphs=False, pretrained=True)
    for _ in range(3):
        tr.train()
    sim = tr.env.sim
    sim.set_block(block)
else:
    sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
    sim.set_block(block)
    sim.out["nbr_dist"] = None
    sim._step_out = sim.make_step_out(sim.out)
    out = sim.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    for i in range(250):
        out = sim.step(cruise_actions(out["obs"], gen))
dbg = torch.zeros(E, 16, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib.copo_sim_set_debug(sim._h, dbg.data_ptr()))
if not LIVE:
    a = cruise_actions(out["obs"], gen)
for rep in range(4):
    torch.cuda.synchronize()
    if LIVE:
        tr.sampler.sample()
        out = {"flags": tr.sampler.flags[-1]}
    else:
        for _ in range(3):
            out = sim.step(cruise_actions(out["obs"], gen))
    torch.cuda.synchronize()
    t0, t1 = dbg[:, 11].cpu().double() * 10e-3, dbg[:, 12].cpu().double() * 10e-3       # us
    base = t0.min()
    s0, _ = torch.sort(t0 - base)
    q = [0, E // 8, E // 4, E // 2, 3 * E // 4, E - 1]
    print("starts (us after the first), by rank %s: %s" % (q, ["%.2f" % s0[i] for i in q]))
    print("lifetime mean %.2f us, min %.2f, max %.2f; last end %.2f us" % ((t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max(), (t1 - base).max()))
    order = torch.argsort(t0)
    print("first 16 workgroups to start:", order[:16].tolist())
    life = t1 - t0
    top = torch.argsort(life, descending=True)[:6].tolist() + torch.argsort(life)[:2].tolist()
    d = dbg.cpu()
    pres = ((out["flags"] & 0x41) != 0).sum(-1).cpu()
    for e in top:
        ph = [(d[e, k + 1] - d[e, k]).item() for k in range(6)]
        print("  scene %3d: %.2f us, phases (cycles) P0 %d P1 %d P2 %d P3 %d P4 %d P5 %d | lists %s | roles done w0 %d w1 %d lidar %d | present %d"
              % (e, life[e], *ph, {1: "register", 2: "pair-parallel"}.get(d[e, 7].item(), "register + %d agents exactly" % (d[e, 7].item() - 16)), *[(d[e, k] - d[e, 4]).item() for k in (8, 9, 10)], pres[e]))
    dbg[:, 7:11] = 0
    print("  P0: state loads arrived %.0f cycles after the kernel's first stamp (mean over scenes)" % d[:, 14].double().mean().item())
    simd = d[:, 13]
    print("  SIMD of waves 0..15 (scene 0, 1, 2):", [[(int(simd[k]) >> (2 * w)) & 3 for w in range(block // 64)] for k in range(3)])
    dbg[:, 11:16] = 0

"""Wide counter sweep of the simulator step kernel on populated scenes: one rocprofv3 --pmc pass per counter group, each stepping the
SAME saved state (16 384 Intersection scenes under the lane-keeping controller) 12 times; prints the per-launch mean of every counter.
usage (GPU box): python scripts/sim_pmc_wide.py E block [N] [map] [lasers]      child: --child E block state.pt N map lasers"""
import glob, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH",
    "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL GRBM_GUI_ACTIVE",
    "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN",
]


def child(E, block, path, N, mp, lasers):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import torch
    from copo_amd.sim import SimConfig, VecSim
    from bench_sim import cruise_actions
    sim = VecSim(SimConfig(map=mp, num_envs=E, num_agents=N, num_lasers=lasers), with_info=False)
    sim.set_block(block)
    out = sim.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    if not os.path.exists(path):
        for i in range(250):
            act = cruise_actions(out["obs"], gen)
            if i < 249:
                out = sim.step(act)
        st, env = sim.get_state()
        torch.save(dict(st=st.cpu(), env=env.cpu(), act=act.cpu()), path)
    d = torch.load(path)
    st, env, act = d["st"].cuda(), d["env"].cuda(), d["act"].cuda()
    for i in range(12):
        sim.set_state(st, env)
        out = sim.step(act)
    torch.cuda.synchronize()
    print(json.dumps(dict(present=float(((out["flags"] & 0x41) != 0).sum()) / E)))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6], int(sys.argv[7]))
        sys.exit(0)
    E, block = int(sys.argv[1]), int(sys.argv[2])
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    mp = sys.argv[4] if len(sys.argv) > 4 else "intersection"
    lasers = int(sys.argv[5]) if len(sys.argv) > 5 else 72
    tmp = "/tmp/pmc_wide_%s_E%d_N%d_L%d" % (mp, E, N, lasers)
    os.makedirs(tmp, exist_ok=True)
    state = os.path.join(tmp, "state.pt")
    env = dict(os.environ, TMPDIR="/tmp")
    args = [sys.executable, os.path.abspath(__file__), "--child", str(E), str(block), state, str(N), mp, str(lasers)]
    if not os.path.exists(state):
        subprocess.run(args, capture_output=True, cwd="/tmp", env=env)
    print("# %s E=%d N=%d lasers=%d block=%d: per-launch means over the step launches (12 per pass)" % (mp, E, N, lasers, block))
    for gi, g in enumerate(GROUPS[:int(os.environ.get("PMC_GROUPS", len(GROUPS)))]):
        d = os.path.join(tmp, "g%d_b%d" % (gi, block))
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["-d", d, "--"] + args, capture_output=True, text=True, cwd="/tmp", env=env)
        got = {}
        for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            con = sqlite3.connect(f)
            for kn, cn, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                if "sim_step" in kn:
                    got[cn] = v
        if not got:
            print("group %d: no counters: %s" % (gi, r.stderr[-300:]))
        for cn in g.split():
            if cn in got:
                print("  %-28s %16.0f   (%10.1f per scene)" % (cn, got[cn], got[cn] / E), flush=True)

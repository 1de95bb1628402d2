"""Where the simulator step kernel spends its instructions: profiling builds with one phase compiled out each
(`make -C copo_amd/csrc prof SKIP=<mask>`; the package itself never loads those) step the SAME saved scene state once per
launch, so the difference of the SQ_INSTS_VALU counter against the full build is the phase's own instruction count.

usage (on the GPU box):  python scripts/sim_valu_split.py [E] [block]
   -> one rocprofv3 --pmc pass per variant, table on stdout.  SPLIT_LDS=1: a second pass per variant with the LDS counters
   (SQ_LDS_BANK_CONFLICT, SQ_LDS_ADDR_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_ACTIVE_INST_LDS: cycles summed over the chip), so that the
   difference against the full build says which phase the bank conflicts belong to.  Child mode: sim_valu_split.py --child <mask> E block state.pt"""
import json
import os
import sqlite3
import glob
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "everything (nothing skipped)", 1: "neighbour lists", 2: "LiDAR windows + box tests", 4: "LiDAR write-out",
         8: "state / navigation block", 16: "collision pairs", 32: "respawn", 64: "projection / termination", 127: "all of the above",
         128: "neighbour lists pair-parallel (register formulation off: phase = NEGATIVE of its saving)", 3: "neighbours + LiDAR tests", 6: "LiDAR tests + write-out", 7: "neighbours + all LiDAR",
         1024: "timers + kinematic bicycle (packed shape)", 2048: "LiDAR pair queue from the reach masks (packed shape)", 3199: "all of the above (packed shape)"}


def child(mask, E, block, path):
    sys.path.insert(0, ROOT)
    import copo_amd._libsel as S
    S.PATH = os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_%d.so" % mask)
    import torch
    from copo_amd.sim import SimConfig, VecSim
    sim = VecSim(SimConfig(map="intersection", num_envs=E, num_agents=40), with_info=False)
    sim.set_block(block)
    sim.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.stack([torch.randn(E, 40, device="cuda", generator=gen) * 0.1, torch.rand(E, 40, device="cuda", generator=gen)], -1).contiguous() for _ in range(8)]
    if not os.path.exists(path):
        assert mask == 0
        act = acts[0]
        if os.environ.get("VALU_POLICY") == "cruise":     # populated scenes: lane-keeping controller, 250 closed-loop steps
            sys.path.insert(0, ROOT)
            from bench import cruise_actions
            out = sim.step(act)
            for i in range(250):
                act = cruise_actions(out["obs"], gen)
                if i < 249:
                    out = sim.step(act)
        else:
            for i in range(60):
                sim.step(acts[i % 8])
        st, env = sim.get_state()
        torch.save(dict(st=st.cpu(), env=env.cpu(), act=act.cpu()), path)
    d = torch.load(path)
    st, env, act = d["st"].cuda(), d["env"].cuda(), d["act"].cuda()
    for i in range(12):
        sim.set_state(st, env)
        out = sim.step(act)
    torch.cuda.synchronize()
    print(json.dumps(dict(present=float(((out["flags"] & 0x41) != 0).sum()) / E)))


def counters(d):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for kn, cn, v, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if "sim_step" in kn:
                vals[cn] = v
        try:
            for name, avg in con.execute("select name, average from top_kernels"):
                if "sim_step" in name:
                    vals["avg_us"] = avg
        except sqlite3.Error:
            pass
    return vals


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    tmp = os.path.join("/tmp", "valu_split_E%d_%s" % (E, os.environ.get("VALU_POLICY", "random")))
    os.makedirs(tmp, exist_ok=True)
    state = os.path.join(tmp, "state.pt")
    env = dict(os.environ, TMPDIR="/tmp")
    if not os.path.exists(state):      # scene state after 60 steps of the full build (not profiled)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "0", str(E), str(block), state], capture_output=True, cwd="/tmp", env=env)
    base = None
    base_lds = None
    for mask, name in NAMES.items():
        if not os.path.exists(os.path.join(ROOT, "copo_amd", "lib", "libcopo_hip_prof_%d.so" % mask)):
            continue
        d = os.path.join(tmp, "m%d" % mask)
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "-d", d, "--",
                            sys.executable, os.path.abspath(__file__), "--child", str(mask), str(E), str(block), state],
                           capture_output=True, text=True, cwd="/tmp", env=env)
        c = counters(d)
        if "SQ_INSTS_VALU" not in c:
            print("mask %d: no counters (%s)" % (mask, r.stderr[-200:]))
            continue
        per = {k: c[k] / E for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
        if mask == 0:
            base = per
        msg = "skip %3d  %-36s VALU %7.0f  SALU %6.0f  LDS %6.0f per scene  %7.1f us" % (mask, name, per["SQ_INSTS_VALU"], per["SQ_INSTS_SALU"], per["SQ_INSTS_LDS"], c.get("avg_us", float("nan")))
        if base is not None and mask:
            msg += "   -> phase: VALU %6.0f (%4.1f %%)  SALU %6.0f  LDS %5.0f" % (
                base["SQ_INSTS_VALU"] - per["SQ_INSTS_VALU"], 100 * (base["SQ_INSTS_VALU"] - per["SQ_INSTS_VALU"]) / base["SQ_INSTS_VALU"],
                base["SQ_INSTS_SALU"] - per["SQ_INSTS_SALU"], base["SQ_INSTS_LDS"] - per["SQ_INSTS_LDS"])
        if os.environ.get("SPLIT_LDS"):
            d2 = os.path.join(tmp, "l%d" % mask)
            subprocess.run(["rm", "-rf", d2])
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "-d", d2, "--",
                            sys.executable, os.path.abspath(__file__), "--child", str(mask), str(E), str(block), state],
                           capture_output=True, text=True, cwd="/tmp", env=env)
            c2 = counters(d2)
            lds = {k: c2.get(k, float("nan")) / E for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS")}
            if mask == 0:
                base_lds = lds
            msg += "\n          LDS cycles per scene: bank conflict %6.0f  addr conflict %6.0f  idx active %6.0f  inst active %6.0f" % (
                lds["SQ_LDS_BANK_CONFLICT"], lds["SQ_LDS_ADDR_CONFLICT"], lds["SQ_LDS_IDX_ACTIVE"], lds["SQ_ACTIVE_INST_LDS"])
            if mask:
                msg += "   -> phase: bank conflict %6.0f  idx active %6.0f" % (base_lds["SQ_LDS_BANK_CONFLICT"] - lds["SQ_LDS_BANK_CONFLICT"], base_lds["SQ_LDS_IDX_ACTIVE"] - lds["SQ_LDS_IDX_ACTIVE"])
        print(msg, flush=True)

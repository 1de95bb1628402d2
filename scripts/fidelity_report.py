"""Assemble profiles/r06_fidelity.txt from the kept outputs of the round-6 fidelity runs (gpurun_out/r06_fid_*.txt, written on the GPU box by
scripts/fidelity_round6_gpu*.sh): the summary tables, the population evaluations, the dynamics sweep and every run's last line.
usage: python scripts/fidelity_report.py > profiles/r06_fidelity.txt"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def table(*names):
    files = [os.path.join(G, n) for n in names if os.path.exists(os.path.join(G, n))]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fidelity_summary.py")] + files, capture_output=True, text=True)
    return r.stdout.rstrip()


def grep(path, *keys):
    p = os.path.join(G, path)
    if not os.path.exists(p):
        return "(missing: %s)" % path
    return "\n".join(l.rstrip()[:400] for l in open(p) if any(k in l for k in keys))


print(open(os.path.join(ROOT, "scripts", "fidelity_report_head.txt")).read().rstrip())
print("\n## (b) Intersection, 30 agents, 1 M env steps, 8 seeds: the default against the reference's bootstrap rule and the reference's batch structure")
print(table("r06_fid_inter.txt", "r06_fid_interref.txt"))
# learning curves of the default: success per ~100 k-step window, mean over the seeds
import re as _re
import numpy as _np
def curve(path, mapname, algo, variant):
    p = os.path.join(G, path)
    if not os.path.exists(p):
        return "(missing)"
    runs, cur = [], None
    for line in open(p):
        m = _re.match(r"### map=MultiAgent(\S+?)(?:Env)? algo=(\S+) variant=(\S+) ", line)
        if m:
            cur = [] if (m.group(1), m.group(2), m.group(3)) == (mapname, algo, variant) else None
            if cur is not None:
                runs.append(cur)
            continue
        f = line.split()
        if cur is not None and len(f) >= 14 and f[0].isdigit():
            cur.append((int(f[1]), float(f[4])))
    n = min(len(r) for r in runs)
    return "  ".join("%dk: %.3f" % (runs[0][k][0] // 1000, _np.mean([r[k][1] for r in runs])) for k in range(n)) + "   (%d seeds)" % len(runs)
print("\n# success rate against env steps, default structure (mean over the seeds of the rates of all agents that finished in the window)")
print("CoPO Intersection: " + curve("r06_fid_inter.txt", "Intersection", "copo", "base"))
print("IPPO Intersection: " + curve("r06_fid_inter.txt", "Intersection", "ippo", "base"))
print("\n## (c) Tollgate (40 agents) and Bottleneck (20 agents), 4 seeds per variant: what makes them easy to learn here?")
print(table("r06_fid_toll_bottle.txt"))
print("\n# second pass: booth buildings as static boxes (crash on touch, seen by the LiDAR)")
print(table("r06_fid_tollb2.txt"))
print("\n# third pass: the Tollgate's LiDAR at MATollConfig's 20 m (the code had kept the other scenes' 40 m since round 2), alone and with buildings + booth rules")
print(table("r06_fid_tolll.txt"))
print(grep("r06_fid_run5.log", "_tollgate", "=== shipped"))
print("\n# fourth pass: the buildings as exact static boxes, HIDDEN from the LiDAR (20 m): the new default")
print(table("r06_fid_hidden.txt", "r06_fid_hidden_b.txt"))
print(grep("r06_fid_run6.log", "_tollgate", "=== shipped"))
print("\n# H7: Bottleneck with the centre line of the Merge / neck / Split roads broken and crossable (maps.bottleneck(centre_open=True))")
print(table("r06_fid_bottleopen.txt"))
print(grep("r06_fid_run7.log", "_bottle", "=== shipped"))
print("\n# H6: body_margin 1.0 (the whole body against the edge lines) on the three scenes")
print(table("r06_fid_margin.txt"))
print("\n# first pass of the buildings (a road-coordinate box test, NOT seen by the LiDAR; kept for the comparison)")
print(table("r06_fid_tollb.txt"))
print("\n# the reference's shipped Tollgate populations (64 whole scene episodes) under the rule variants")
print(grep("r06_fid_run4.log", "_tollgate", "=== shipped"))
print("# ... and with the first-pass buildings the LiDAR does not see")
print(grep("r06_fid_run2.log", "_tollgate"))
print("\n## the table's other scenes (Roundabout, Parking Lot, PG map = MultiAgentMetaDrive) on the round-6 code")
print(table("r06_fid_rest.txt"))
print("\n## (a) which dynamics constant moves the shipped CoPO Intersection population's speed?  (scripts/fidelity_dynamics_sweep.py)")
print(grep("r06_fid_dynamics.txt", "km/h"))
print("\n## review item 4: the weak-scaling job of G ranks as ONE job on one GPU (G x 256 scenes x 8 steps per iteration, global minibatch G x 512 / G x 1 024)")
print(table("r06_fid_dp.txt"))
print("\n## every run's last line (iter env_steps agent_steps wall_s success crash out max_step ep_reward lcf kl agents_finished velocity_m_s episode_len)")
for f in sorted(glob.glob(os.path.join(G, "r06_fid_*.txt"))):
    if f.endswith("dynamics.txt"):
        continue
    print("# " + os.path.basename(f))
    last, head = None, None
    for line in open(f):
        if line.startswith("### "):
            if head and last:
                print(head + "\n" + last)
            head, last = line.rstrip(), None
        elif line.split() and line.split()[0].isdigit():
            last = line.rstrip()
    if head and last:
        print(head + "\n" + last)

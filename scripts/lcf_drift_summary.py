"""Table of the LCF-drift experiments (scripts/lcf_drift_experiments.sh): per variant, the maximum training success per seed (the
reference's table metric), the LCF and the success / velocity of the last window at 1 M env steps.
usage: python scripts/lcf_drift_summary.py file [file ...]"""
import re
import sys
import numpy as np

runs = {}
for path in sys.argv[1:]:
    key = None
    for line in open(path):
        m = re.match(r"### variant=(\S+) num_envs=(\d+) config=(.*) seed=(\d+)", line)
        if m:
            key = (m.group(1), int(m.group(2)), m.group(3))
            runs.setdefault(key, []).append([])
            continue
        m = re.match(r"# copo \S+ E=(\d+) N=\d+ seed=(\d+)", line)
        if m and path.endswith("30_agents.txt"):
            key = ("base (profiles/r04_train_curves_intersection_30_agents.txt)", int(m.group(1)), "{}")
            runs.setdefault(key, []).append([])
            continue
        if line.startswith("# ippo"):
            key = None
        f = line.split()
        if key and len(f) >= 14 and f[0].isdigit():
            runs[key][-1].append([float(x) for x in f])
print("%-62s %5s %5s  %-16s %-14s %-16s %-12s" % ("variant", "envs", "seeds", "max success", "final LCF", "final success", "final m/s"))
for (name, envs, cfg), rr in runs.items():
    rr = [np.array(r) for r in rr if len(r)]
    mx = np.array([r[:, 4].max() for r in rr])
    lcf = np.array([r[-1, 9] for r in rr])
    fin = np.array([r[-1, 4] for r in rr])
    vel = np.array([r[-1, 12] for r in rr])
    print("%-62s %5d %5d  %5.1f +- %4.1f %%   %.3f +- %.3f  %5.1f +- %4.1f %%   %5.1f" % (
        (name + " " + (cfg if cfg != "{}" else ""))[:62], envs, len(rr), 100 * mx.mean(), 100 * mx.std(), lcf.mean(), lcf.std(), 100 * fin.mean(), 100 * fin.std(), vel.mean()))

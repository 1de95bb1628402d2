"""Table of scripts/fidelity_r06.sh: per (map, algo, variant) the maximum training success per seed (the reference's table metric,
benchmarks/MetaDrive-0.2.5/README.md), success / crash / out / velocity of the last window and the final LCF.
usage: python scripts/fidelity_summary.py file [file ...]"""
import re
import sys
import numpy as np

runs = {}
for path in sys.argv[1:]:
    key = None
    for line in open(path):
        m = re.match(r"### map=MultiAgent(\S+?)(?:Env)? algo=(\S+) variant=(\S+) num_envs=(\d+) config=(.*) env=(.*) seed=(\d+)", line)
        if m:
            key = (m.group(1), m.group(2), m.group(3), int(m.group(4)))
            runs.setdefault(key, []).append([])
            continue
        f = line.split()
        if key and len(f) >= 14 and f[0].isdigit():
            runs[key][-1].append([float(x) for x in f])
print("%-13s %-5s %-32s %5s %5s  %-15s %-15s %-7s %-7s %-8s %-8s" % ("map", "algo", "variant", "envs", "seeds", "max success %", "final success %", "crash", "out", "LCF", "m/s"))
for (mp, algo, name, envs), rr in runs.items():
    rr = [np.array(r) for r in rr if len(r)]
    if not rr:
        continue
    mx = np.array([r[:, 4].max() for r in rr])
    fin = np.array([r[-1, 4] for r in rr])
    print("%-13s %-5s %-32s %5d %5d  %5.1f +- %4.1f   %5.1f +- %4.1f   %.3f   %.3f   %+.3f   %5.1f" % (
        mp, algo, name, envs, len(rr), 100 * mx.mean(), 100 * mx.std(), 100 * fin.mean(), 100 * fin.std(),
        np.mean([r[-1, 5] for r in rr]), np.mean([r[-1, 6] for r in rr]), np.nanmean([r[-1, 9] for r in rr]), np.mean([r[-1, 12] for r in rr])))

"""profiles/sim_traffic.json from the `live` passes of scripts/prof_sim_round.sh (FETCH_SIZE / WRITE_SIZE were collected there in their
own rocprofv3 runs on `bench.py --roofline-only`, mean over the last 200 step launches = the replay of the recorded actions).
usage: python scripts/sim_traffic_from_summary.py gpurun_out/prof_r05_live"""
import json, os, re, sys
d = sys.argv[1]
txt = open(os.path.join(d, "summary.txt")).read()
f = float(re.search(r"pmc FETCH_SIZE\s+mean over the last (\d+) dispatches = ([0-9.]+)", txt).group(2))
n = int(re.search(r"pmc FETCH_SIZE\s+mean over the last (\d+) dispatches", txt).group(1))
w = float(re.search(r"pmc WRITE_SIZE\s+mean over the last \d+ dispatches = ([0-9.]+)", txt).group(1))
seen = int(re.search(r"trace: (\d+) dispatches", txt).group(1))
log = [json.loads(l) for l in open(os.path.join(d, "pmc3.log")) if l.startswith("{")][-1]
res = dict(command="bench.py --roofline-only", scenes=log["scenes"], slots=log["slots"], launches_averaged=n, step_launches_seen=seen,
           fetch_size_kb=f, write_size_kb=w, bytes_per_launch=round((f + w) * 1024), bytes_per_launch_fetch_x2=round((2 * f + w) * 1024),
           units_per_launch=log["units_per_launch"], bytes_per_unit=log["bytes_per_unit"],
           algorithmic_bytes_per_launch=round(log["units_per_launch"] * log["bytes_per_unit"]),
           us_per_launch_under_pmc=log["us_per_launch"], kernel_source_sha1=log["kernel_source_sha1"],
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/prof_sim_round.sh live), mean over the last %d step launches = "
                "the replay of the recorded actions on the trainer's scenes; raw counter sums in KB; FETCH_SIZE is uncalibrated for this "
                "kernel's narrow loads (x2 = the guide's wide-read correction as an upper bound)" % n)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(res, open(os.path.join(root, "profiles", "sim_traffic.json"), "w"), indent=1)
print(json.dumps(res))

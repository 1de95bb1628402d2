"""Duration of one fused SGD step of the bench workload (bench.measure_learner_step) for a chosen library build:
COPO_LIB=<path of a libcopo_hip*.so> python scripts/learner_step_time.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("COPO_LIB"):
    import copo_amd._libsel as S
    S.PATH = os.path.join(ROOT, os.environ["COPO_LIB"])
import torch
import bench
tr = bench.make_trainer(256, 40, graphs=True, pretrained=True)
for _ in range(4):
    tr.train()
torch.cuda.synchronize()
for rep in range(3):
    print(json.dumps(bench.measure_learner_step(tr)))
ph = bench.measure_phases(tr)
print(json.dumps(ph))

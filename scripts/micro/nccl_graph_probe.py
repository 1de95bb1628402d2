"""Can an RCCL all-reduce be captured in a hipGraph through torch.distributed on this stack?  (single rank probe)"""
import os
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as td
td.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(360201, device="cuda")
y = torch.zeros_like(x)
td.all_reduce(x)                       # warm-up: communicator creation outside the capture
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        y.add_(x); td.all_reduce(y); y.mul_(0.5)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        for _ in range(4):
            y.add_(x); td.all_reduce(y); y.mul_(0.5)
    y.zero_()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("captured and replayed: y[0] =", float(y[0]))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
td.destroy_process_group()

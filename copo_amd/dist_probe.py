"""Child process of `dist.probe_graphed_allreduce`: does a hipGraph that holds an RCCL all-reduce capture AND replay on this
node?  Joins its own process group (the parent shifted MASTER_PORT), captures [kernel, all-reduce, kernel] x 4 exactly like the
trainer's data-parallel chain (GraphedCallable, thread-local capture mode), replays it and checks the sums.  Exit code 0 = yes.
A hang stays inside this process: the parent kills it by PID after its timeout and keeps the eager loop."""
import os
import sys


def main():
    import torch
    import torch.distributed as td
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local)
    td.init_process_group(backend="nccl", rank=rank, world_size=world)
    from copo_amd.trainer import GraphedCallable
    n = 360448
    x = torch.zeros(n, device="cuda")
    acc = torch.zeros(n, device="cuda")
    step = torch.zeros(1, device="cuda")

    def chain():
        for _ in range(4):
            step.add_(1.0)
            x.copy_(step.expand(n) * float(rank + 1))        # a "gradient" that depends on rank and step
            td.all_reduce(x)
            acc.add_(x)

    g = GraphedCallable(chain, True)
    for _ in range(5):          # 2 eager warm-ups, 1 capture + replay, 2 replays
        g()
    torch.cuda.synchronize()
    steps = 20
    want = sum(range(1, steps + 1)) * sum(range(1, world + 1))
    ok = bool(torch.all(acc == float(want)).item()) and g.graph is not None
    td.barrier()
    td.destroy_process_group()
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())

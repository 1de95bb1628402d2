"""Two (or WORLD) processes on ONE GPU through the peer all-reduce: correctness against the known sum over many back-to-back
calls, then time per call.  Launch: python scripts/micro/peer_allreduce_probe.py [world] [n]   (spawns the ranks itself)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker():
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td
    rank, world, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["PROBE_N"])
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from copo_amd.peer import PeerAllReduce
    dev = torch.device("cuda", 0)
    pa = PeerAllReduce(n, dev)
    bad = 0
    for it in range(200):
        gens = [torch.Generator(device="cpu").manual_seed(1000 * it + r) for r in range(world)]
        parts = [torch.randn(n, generator=g) for g in gens]
        want = parts[0].clone()
        for p in parts[1:]:
            want += p                      # rank order, fp32: what the kernel computes
        pa.data.copy_(parts[rank].to(dev))
        pa.all_reduce_()
        got = pa.data.cpu()
        if not torch.equal(got, want):
            bad += 1
        if it % 10 == 0:
            pa.status()                    # a wait that timed out: stop at once
    pa.status()
    torch.cuda.synchronize()
    td.barrier()
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        pa.all_reduce_()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pa.status()
    print("rank %d: %d mismatching calls of 200, %.1f us per call (n = %d floats, world %d, one shared GPU)" % (rank, bad, dt * 1e6, n, world), flush=True)
    pa.close()
    td.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    if os.environ.get("PROBE_WORKER"):
        worker()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 360201
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", PROBE_WORKER="1", PROBE_N=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 99
    sys.exit(rc)

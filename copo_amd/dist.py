"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The reference has no collective at all (Ray object store + in-process
towers, algo_copo.py:519-577); the exchange steps below are the build's own:

  * per SGD minibatch  : one flat-bucket all-reduce of the gradient sums
  * per iteration      : one all-reduce of the advantage statistics (6 doubles) + row counts
  * per meta minibatch : one flat bucket [g_new | g_old | dS/dlcf | S | n]  (both gradients BEFORE the dot); the fused
                         learner batches this: one all-reduce of the gradient pairs of 32 minibatches, one
                         all-gather of the LCF row terms per meta iteration

Env shards never talk to each other, so nothing else crosses ranks.
"""
import os

import torch
import torch.distributed as td


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(device=None):
    """Initialise the default process group from torchrun's env vars (no-op for a single process)."""
    rank, local_rank, world = env_world()
    if (world > 1 or _forced()) and not td.is_initialized():
        use_cuda = torch.cuda.is_available() and (device is None or str(device).startswith("cuda"))
        if use_cuda:
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
        # COPO_DIST_BACKEND=gloo: CUDA tensors over gloo (staged through the host) -- lets several ranks share ONE GPU,
        # which RCCL refuses; used by the two-rank test of the fused data-parallel path on a single-GPU box
        backend = os.environ.get("COPO_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _forced():
    # COPO_FORCE_DIST=1: take the data-parallel code paths with a single rank (exercises them on a one-GPU box)
    return os.environ.get("COPO_FORCE_DIST", "0") == "1"


def is_dist():
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _forced())


def world_size():
    return td.get_world_size() if is_dist() else 1


def rank():
    return td.get_rank() if is_dist() else 0


def _several():
    # (a forced data-parallel run with ONE rank takes every data-parallel code path; a sum or maximum over one rank is the
    #  identity and is not sent through the collective library)
    return is_dist() and td.get_world_size() > 1


def all_reduce_sum_(t):
    if _several():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_sum_async(t):
    """Start the sum; returns the work handle (None for a single process).  `.wait()` makes the CURRENT stream wait for it."""
    if _several():
        return td.all_reduce(t, op=td.ReduceOp.SUM, async_op=True)
    return None


def all_reduce_max_(t):
    if _several():
        td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def has_reduce_scatter():
    """Decided ONCE from the backend's name, identically on every rank (a per-rank try / except at call time could send one rank
    into the all-reduce while its peers sit in the reduce-scatter): RCCL has reduce_scatter_tensor, gloo does not.
    COPO_REDUCE_SCATTER=0 forces the all-reduce branch (the multi-GPU A/B of the two)."""
    if os.environ.get("COPO_REDUCE_SCATTER", "1") == "0":
        return False
    return is_dist() and td.get_backend() == "nccl"


def reduce_scatter_sum_(out, flat):
    """Rows [r c, (r + 1) c) of the sum over the ranks of `flat` [world c, ...] for rank r, c = out.shape[0]; returns the tensor that
    holds them: `out` after a reduce-scatter (half the wire bytes of an all-reduce), or this rank's slice of `flat` after an all-reduce
    where the backend has no reduce-scatter (gloo: the tests' ranks on one device).  Every rank takes the same branch
    (`has_reduce_scatter`); an error of the collective is raised, never turned into the other branch."""
    c = out.shape[0]
    if not _several():
        return flat[:c]
    if has_reduce_scatter():
        td.reduce_scatter_tensor(out, flat, op=td.ReduceOp.SUM)
        return out
    td.all_reduce(flat, op=td.ReduceOp.SUM)
    r = td.get_rank()
    return flat[r * c:(r + 1) * c]


def all_gather_into_(out, t):
    """out [world, *t.shape] <- every rank's t (same shape everywhere)."""
    if not _several():          # (one rank, forced or not: a copy, not a trip through the collective library -- like the sums above)
        out[0].copy_(t)
        return out
    try:
        td.all_gather_into_tensor(out, t.contiguous())
    except (RuntimeError, NotImplementedError):
        td.all_gather([out[i] for i in range(out.shape[0])], t.contiguous())
    return out


def all_gather_int(v, device):
    """Every rank's python int, as a list (one tiny collective per iteration)."""
    if not is_dist():
        return [int(v)]
    t = torch.zeros(world_size(), dtype=torch.int64, device=device)
    t[rank()] = int(v)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def broadcast_module_(module, src=0):
    """Same initial weights on every rank (the reference broadcasts weights from the learner, algo_copo.py:572-577)."""
    if _several():          # (a world of one, forced or not: the identity -- and the CPU policies of bench.py's cpu_baseline leg have no backend)
        for p in list(module.parameters()) + list(module.buffers()):
            td.broadcast(p.data, src=src)


def barrier():
    if is_dist():
        td.barrier()


def ranks_share_a_device(device):
    """True if two ranks of the job use the same physical GPU (tests on a one-GPU box).  Kernels that wait for their peers
    INSIDE the kernel (the tile exchange) then compete with those peers for the same compute units and may starve them."""
    if not is_dist() or td.get_world_size() < 2 or device.type != "cuda":
        return False
    import socket
    pr = torch.cuda.get_device_properties(device)
    me = (socket.gethostname(), getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", device.index), getattr(pr, "pci_device_id", 0))
    every = [None] * td.get_world_size()
    td.all_gather_object(every, me)
    return len(set(every)) < len(every)


def _probe_child(module, port_shift, timeout, need_nccl, extra_env=None):
    """Run `python -m <module>` once per rank as a CHILD process with its own process group (MASTER_PORT shifted); the ranks
    agree on the outcome with a MIN all-reduce.  A child that hangs is killed by PID after `timeout` seconds."""
    import subprocess
    import sys
    if not (td.is_available() and td.is_initialized() and torch.cuda.is_available()):
        return False
    if (need_nccl and td.get_backend() != "nccl") or os.environ.get("COPO_DIST_PROBE", "1") == "0":
        return False
    # the children's rendezvous port: a port that is free on rank 0's host right now, handed to every rank over the parent group
    # (a fixed MASTER_PORT + shift can collide with another job of the node); `port_shift` only if that exchange fails
    # Only the bind may fail quietly (an IPv6-only or unresolvable MASTER_ADDR): rank 0 then broadcasts 0 and EVERY rank takes the
    # shifted port.  The broadcast itself is executed by every rank unconditionally -- a rank that skipped it would pair its next
    # collective with the peers' pending broadcast.
    port = [0]
    if td.get_rank() == 0:
        try:
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind((os.environ.get("MASTER_ADDR", "127.0.0.1"), 0))
                port[0] = int(sk.getsockname()[1])
        except Exception:      # noqa: BLE001
            port[0] = 0
    td.broadcast_object_list(port, src=0)
    if not port[0]:
        port[0] = int(os.environ.get("MASTER_PORT", "29500")) + port_shift
    env = dict(os.environ, MASTER_PORT=str(port[0]))
    env.pop("COPO_FORCE_DIST", None)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # under torchrun: the child's rank 0 must open its OWN store on the new port
    env.update(extra_env or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = -1
    try:
        p = subprocess.Popen([sys.executable, "-m", module], env=env, cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()            # exactly the child we started
            p.wait()
    except OSError:
        rc = -1
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device="cuda")
    td.all_reduce(ok, op=td.ReduceOp.MIN)
    return bool(ok.item())


def probe_tile_exchange(hidden=64, obs_dim=20, nets=2, timeout=150.0, mb=128):
    """True if the data-parallel tile exchange (`copo_ppo_fused_step_dp_f32`: hipIpc-mapped uncached workspaces, peer stores,
    system-scope flags inside the weight-gradient kernel) works between the GPUs of THIS node: every rank runs
    copo_amd/dp_probe.py in a child process (its own process group on MASTER_PORT + 19) -- a learner of the caller's shape takes captured chains
    of data-parallel steps; all ranks must end with bit-identical parameters that agree with the RCCL-reduced step, and no wait
    may have timed out.  A failure or a hang (child killed after `timeout` s) leaves the RCCL loop in place."""
    return _probe_child("copo_amd.dp_probe", 19, timeout, need_nccl=False,
                        extra_env=dict(COPO_DP_PROBE_HIDDEN=str(int(hidden)), COPO_DP_PROBE_OBS=str(int(obs_dim)), COPO_DP_PROBE_NETS=str(int(nets)),
                                       COPO_DP_PROBE_MB=str(int(mb))))


def shutdown():
    """Tear the process group down (quietens the exit-time warning of ProcessGroupNCCL); safe to call when not initialised."""
    import torch.distributed as td
    try:
        if td.is_available() and td.is_initialized():
            td.destroy_process_group()
    except Exception:
        pass

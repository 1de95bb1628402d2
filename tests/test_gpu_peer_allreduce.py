"""Peer all-reduce (`copo_peer_allreduce_sum_f32`): the two-shot sum over mapped device memory that can
stand in for RCCL's all-reduce of the gradient sums.  No multi-GPU node is available to these tests: the kernel's logic is
checked with VIRTUAL ranks (one process, one workspace per rank, all ranks in ONE launch -- rank = blockIdx.y -- because the
streams of a process share hardware queues and separate launches would wait for peers queued behind them), the hipIpc mapping
and the per-rank launches with two real processes that share cuda:0."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n", [(2, 360201), (4, 204549), (8, 1000), (8, 360201), (3, 17)])
def test_virtual_ranks_sum_in_rank_order(world, n):
    from copo_amd import _capi
    from copo_amd.peer import _Raw
    dev = torch.device("cuda", 0)
    nbytes = _capi.lib.copo_peer_workspace_bytes(n, world)
    ws = []
    for r in range(world):
        p = C.c_void_p()
        _capi.check(_capi.lib.copo_peer_alloc(nbytes, C.byref(p)))
        ws.append(p)
    ptrs = (C.c_void_p * world)(*[p.value for p in ws])
    data = [torch.as_tensor(_Raw(p.value, n), device=dev) for p in ws]
    s0 = torch.cuda.current_stream().cuda_stream
    try:
        for it in range(12):
            gen = torch.Generator(device="cpu").manual_seed(100 * it + world)
            parts = [torch.randn(n, generator=gen) * (10.0 ** (r % 3 - 1)) for r in range(world)]
            want = parts[0].clone()
            for p in parts[1:]:
                want += p                      # fp32, rank order: every rank must hold exactly this
            for r in range(world):
                data[r].copy_(parts[r].to(dev))
            torch.cuda.synchronize()
            _capi.check(_capi.lib.copo_debug_peer_allreduce_all_ranks(ptrs, n, world, s0))
            for r in range(world):
                _capi.check(_capi.lib.copo_peer_status(ws[r], n, world, s0))     # no wait timed out
            for r in range(world):
                assert torch.equal(data[r].cpu(), want), (it, r)
    finally:
        torch.cuda.synchronize()
        del data
        for p in ws:
            _capi.lib.copo_peer_free(p)


def test_a_missing_rank_is_an_error_not_a_hang():
    from copo_amd import _capi
    n, world = 1000, 2
    ws = []
    for r in range(world):
        p = C.c_void_p()
        _capi.check(_capi.lib.copo_peer_alloc(_capi.lib.copo_peer_workspace_bytes(n, world), C.byref(p)))
        ws.append(p)
    ptrs = (C.c_void_p * world)(*[p.value for p in ws])
    s = torch.cuda.current_stream().cuda_stream
    _capi.check(_capi.lib.copo_peer_allreduce_sum_f32(ptrs, n, 0, world, s))          # rank 1 never calls
    assert _capi.lib.copo_peer_status(ws[0], n, world, s) != 0
    for p in ws:
        _capi.lib.copo_peer_free(p)
    assert _capi.lib.copo_peer_workspace_bytes(10, 17) == -1 and _capi.lib.copo_peer_allreduce_sum_f32(ptrs, n, 2, 2, s) != 0


def test_two_processes_share_cuda0_through_ipc_handles():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "micro", "peer_allreduce_probe.py"), "2", "360201"],
                       capture_output=True, text=True, timeout=280, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("rank ")]
    assert r.returncode == 0 and len(lines) == 2 and all(" 0 mismatching" in ln for ln in lines), (r.stdout[-2000:], r.stderr[-2000:])

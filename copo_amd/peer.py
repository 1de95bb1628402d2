"""Peer all-reduce of the data-parallel learner (C ABI `copo_peer_*` / `copo_ipc_*`): a two-shot sum over
device memory that every rank of the node has mapped, instead of RCCL's ring, for the one message that sits on the critical
path of every optimizer step (the 1.44 MB gradient sum of a 512-row minibatch).  A stand-alone op since round 5 (its own tests,
tests/test_gpu_peer_allreduce.py): the trainer's data-parallel step is the tile exchange below (the sum happens inside the
weight-gradient kernel, no gradient buffer at all) with ONE fallback, the RCCL loop."""
import ctypes as C
import os

import torch
import torch.distributed as td

from . import _capi


class _Raw:
    """Raw device memory as a `__cuda_array_interface__` provider (torch.as_tensor aliases it)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f4", data=(int(ptr), False), version=2)


class PeerAllReduce:
    """`data` [n] float32 lives in this rank's workspace; `all_reduce_()` replaces it with the sum over all ranks."""

    def __init__(self, n, device):
        assert td.is_initialized() and device.type == "cuda"
        self.n, self.rank, self.world = int(n), td.get_rank(), td.get_world_size()
        self.device = device
        nbytes = _capi.lib.copo_peer_workspace_bytes(self.n, self.world)
        if nbytes < 0:
            raise ValueError("peer all-reduce: world size %d not supported" % self.world)
        with torch.cuda.device(device):
            own = C.c_void_p()
            _capi.check(_capi.lib.copo_peer_alloc(nbytes, C.byref(own)))
            self._own = own
            handle = C.create_string_buffer(64)
            _capi.check(_capi.lib.copo_ipc_export(own, handle))
            handles = [None] * self.world
            td.all_gather_object(handles, (os.getpid(), handle.raw))
            self._mapped = []
            ptrs = (C.c_void_p * self.world)()
            for r, (pid, raw) in enumerate(handles):
                if r == self.rank:
                    ptrs[r] = own.value
                else:
                    p = C.c_void_p()
                    _capi.check(_capi.lib.copo_ipc_open(raw, C.byref(p)))
                    self._mapped.append(p)
                    ptrs[r] = p.value
            self._ptrs = ptrs
        self.data = torch.as_tensor(_Raw(own.value, self.n), device=device)
        td.barrier()          # nobody reduces before everybody has mapped everybody

    def all_reduce_(self):
        _capi.check(_capi.lib.copo_peer_allreduce_sum_f32(self._ptrs, self.n, self.rank, self.world, _capi.current_stream()))
        return self.data

    def status(self):
        """Raises if a wait inside any call so far timed out (a rank that never arrived).  Synchronises the stream."""
        _capi.check(_capi.lib.copo_peer_status(self._own, self.n, self.world, _capi.current_stream()))

    def close(self):
        if getattr(self, "_own", None) is not None:
            torch.cuda.synchronize(self.device)
            self.data = None
            for p in self._mapped:
                _capi.lib.copo_ipc_close(p)
            _capi.lib.copo_peer_free(self._own)
            self._own, self._mapped = None, []


def _map_all(own, device):
    """Exchange the hipIpc handle of `own` (this rank's device allocation, or None if the allocation failed) with every rank and
    map the others'.  Every rank takes part in the SAME collectives whatever fails locally; returns (ctypes array of the world
    pointers as mapped here, list of the mappings to close, ok) with `ok` agreed by all ranks (a failure anywhere = False
    everywhere, so that nobody waits for a peer that gave up)."""
    rank, world = td.get_rank(), td.get_world_size()
    raw = None
    if own is not None:
        handle = C.create_string_buffer(64)
        if _capi.lib.copo_ipc_export(own, handle) == 0:
            raw = handle.raw
    handles = [None] * world
    td.all_gather_object(handles, (os.getpid(), raw))
    mapped, good = [], raw is not None and all(h[1] is not None for h in handles)
    ptrs = (C.c_void_p * world)()
    if good:
        for r, (pid, h) in enumerate(handles):
            if r == rank:
                ptrs[r] = own.value
            else:
                p = C.c_void_p()
                if _capi.lib.copo_ipc_open(h, C.byref(p)) != 0:
                    good = False
                    break
                mapped.append(p)
                ptrs[r] = p.value
    flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=device)
    td.all_reduce(flag, op=td.ReduceOp.MIN)
    return ptrs, mapped, bool(flag.item())


class TileExchange:
    """Workspaces of the data-parallel SGD step (`copo_ppo_fused_step_dp_f32`): every rank allocates
    one (uncached device memory), all ranks map all of them.  The step kernels then exchange gradient tiles through them --
    no gradient buffer, no all-reduce call, no separate Adam launch."""

    def __init__(self, cfg, device):
        assert td.is_initialized() and device.type == "cuda"
        self.cfg, self.rank, self.world, self.device = cfg, td.get_rank(), td.get_world_size(), device
        nbytes = _capi.lib.copo_dp_workspace_bytes(C.byref(cfg), self.world)
        if nbytes < 0:
            raise ValueError("tile exchange: world size %d not supported" % self.world)
        with torch.cuda.device(device):
            own = C.c_void_p()
            if _capi.lib.copo_peer_alloc(nbytes, C.byref(own)) != 0:
                own = None
            self._own = own
            self.ptrs, self._mapped, self.usable = _map_all(own, device)      # (its all-reduce: nobody steps before everybody has mapped everybody)
        if not self.usable:           # agreed by all ranks: hipIpc export / open failed somewhere -- the caller keeps the collective loop
            self.close()

    def ok(self):
        """False if a wait inside any step so far timed out.  Synchronises the stream."""
        return _capi.lib.copo_dp_status(self._own, C.byref(self.cfg), self.world, _capi.current_stream()) == 0

    def status(self):
        """Raises if a wait inside any step so far timed out (a rank that never arrived).  Synchronises the stream."""
        rc = _capi.lib.copo_dp_status(self._own, C.byref(self.cfg), self.world, _capi.current_stream())
        if rc != 0:
            raise RuntimeError("data-parallel tile exchange: a wait for a peer rank timed out (rank %d of %d, code %d) -- the "
                               "parameters of this run are not trustworthy; COPO_DP_EXCHANGE=rccl selects the RCCL loop"
                               % (self.rank, self.world, rc))

    def close(self):
        torch.cuda.synchronize(self.device)
        for p in getattr(self, "_mapped", []):
            _capi.lib.copo_ipc_close(p)
        if getattr(self, "_own", None) is not None:
            _capi.lib.copo_peer_free(self._own)
        self._own, self._mapped = None, []


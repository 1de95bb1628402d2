"""Tensor-level wrappers of the stateless HIP ops in libcopo_hip.so.

Every function takes CUDA tensors, launches on torch's current stream through the C ABI and returns
tensors.  There is no CPU implementation here on purpose: a non-CUDA tensor raises.
"""
import ctypes as C

import torch

from . import _capi

F_ACTED, F_DONE = _capi.F_ACTED, _capi.F_DONE


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("copo_amd.ops: expected a CUDA tensor (the HIP path has no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("copo_amd.ops: tensors must be contiguous")


def neighbours(pos, present, rew=None, K=8, radius=40.0, mf_distance=10.0):
    """CCEnv._update_distance_map + _find_in_range (+ LCFEnv reward block) -- utils/env_wrappers.py:125-158,313-326."""
    _chk(pos, present, rew)
    E, N = pos.shape[0], pos.shape[1]
    dev = pos.device
    o = dict(nbr_idx=torch.empty(E, N, K, dtype=torch.int32, device=dev),
             nbr_cnt=torch.empty(E, N, dtype=torch.int32, device=dev),
             mf_cnt=torch.empty(E, N, dtype=torch.int32, device=dev),
             nbr_dist=torch.empty(E, N, K, dtype=torch.float32, device=dev),
             nei_rew=torch.zeros(E, N, dtype=torch.float32, device=dev),
             glob_rew=torch.zeros(E, dtype=torch.float32, device=dev))
    _capi.check(_capi.lib.copo_neighbours_f32(
        pos.data_ptr(), present.data_ptr(), _capi.ptr(rew), E, N, K, float(radius), float(mf_distance),
        o["nbr_idx"].data_ptr(), o["nbr_cnt"].data_ptr(), o["mf_cnt"].data_ptr(), o["nbr_dist"].data_ptr(),
        o["nei_rew"].data_ptr(), o["glob_rew"].data_ptr(), _capi.current_stream()))
    return o


def gae3(rew, val, flags, gammas, lam, out_adv=None, out_tgt=None):
    """Segmented reverse scan for `heads` GAE heads.  rew/val: [H, T, M] fp32; flags: [T, M] u8."""
    _chk(rew, val, flags)
    H, T, M = rew.shape
    assert val.shape == rew.shape and flags.numel() == T * M and flags.dtype == torch.uint8
    adv = torch.empty_like(rew) if out_adv is None else out_adv
    tgt = torch.empty_like(rew) if out_tgt is None else out_tgt
    g = (C.c_double * H)(*[float(x) for x in gammas])
    _capi.check(_capi.lib.copo_gae3_f32(rew.data_ptr(), val.data_ptr(), flags.data_ptr(), T, M, H, g, float(lam),
                                        adv.data_ptr(), tgt.data_ptr(), _capi.current_stream()))
    return adv, tgt


def cc_fuse(mode, obs, act, flags, nbr_idx, cnt, counterfactual=True, num_neighbours=4, out=None):
    """Centralised-critic observation (algo_ccppo.py:225-311).  obs [R, N, O], act [R, N, A], flags [R, N],
    nbr_idx [R, N, K], cnt [R, N] (mf: mf_cnt; concat: nbr_cnt)."""
    _chk(obs, act, flags, nbr_idx, cnt)
    R, N, O = obs.shape
    A, K = act.shape[-1], nbr_idx.shape[-1]
    cf = 1 if counterfactual else 0
    if mode == "mf":
        Cd = 2 * O + (A if cf else 0)
    elif mode == "concat":
        Cd = O + num_neighbours * (O + (A if cf else 0))
    else:
        raise ValueError("unknown fuse mode %r" % (mode,))
    cc = torch.empty(R, N, Cd, dtype=torch.float32, device=obs.device) if out is None else out
    assert cc.shape == (R, N, Cd)
    s = _capi.current_stream()
    if mode == "mf":
        _capi.check(_capi.lib.copo_cc_fuse_mf_f32(obs.data_ptr(), act.data_ptr(), flags.data_ptr(), nbr_idx.data_ptr(),
                                                  cnt.data_ptr(), R, N, O, A, K, cf, cc.data_ptr(), s))
    else:
        _capi.check(_capi.lib.copo_cc_fuse_concat_f32(obs.data_ptr(), act.data_ptr(), flags.data_ptr(),
                                                      nbr_idx.data_ptr(), cnt.data_ptr(), R, N, O, A, K,
                                                      int(num_neighbours), cf, cc.data_ptr(), s))
    return cc


def lcf_stats_workspace(device):
    return torch.zeros(_capi.LCF_STATS_DOUBLES, dtype=torch.float64, device=device)


def lcf_mix_partial(adv, nei_adv, glob_adv, lcf, valid, mixed, stats):
    """A_c = cos(lcf*pi/2)*adv + sin(lcf*pi/2)*nei_adv and {n, sum, sumsq} of A_c / glob_adv -> stats[0:6]."""
    _chk(adv, nei_adv, glob_adv, lcf, valid, mixed, stats)
    assert stats.dtype == torch.float64 and stats.numel() >= _capi.LCF_STATS_DOUBLES
    _capi.check(_capi.lib.copo_lcf_mix_partial_f32(adv.data_ptr(), nei_adv.data_ptr(), glob_adv.data_ptr(),
                                                   lcf.data_ptr(), _capi.ptr(valid), adv.numel(), mixed.data_ptr(),
                                                   stats.data_ptr(), _capi.current_stream()))


def lcf_mix_apply(mixed, glob_adv, valid, stats, norm_adv, glob_std):
    _chk(mixed, glob_adv, valid, stats, norm_adv, glob_std)
    _capi.check(_capi.lib.copo_lcf_mix_apply_f32(mixed.data_ptr(), glob_adv.data_ptr(), _capi.ptr(valid), mixed.numel(),
                                                 stats.data_ptr(), norm_adv.data_ptr(), glob_std.data_ptr(),
                                                 _capi.current_stream()))

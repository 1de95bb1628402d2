"""ctypes binding of libcopo_hip.so (include/copo_hip.h).

This is the thin host layer the north star asks for: Python hosts the loop, the simulator and the
custom learn-side ops are HIP kernels behind a C ABI.  There is NO fallback: if the shared library
is missing or a symbol is absent, importing this module raises, and every op raises on a non-zero
return code with the library's own error string.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcopo_hip.so")
# Profiling scripts (scripts/sim_valu_split.py) load a differently compiled copy -- phases of the step kernel compiled out,
# `make prof SKIP=<mask>` -- by setting `copo_amd._libsel.PATH` BEFORE importing this module.  There is no environment
# variable and no runtime knob: the shipped library always does all the work.
try:
    from . import _libsel as _sel
    if getattr(_sel, "PATH", None):
        LIB_PATH = _sel.PATH
except ImportError:
    pass

from ._abi import (ABI_VERSION, INFO_DIM, LINE_STRIDE, MAX_AGENTS, MAX_LASERS, MAX_LINES, MAX_ROUTES, MAX_SAFE,  # noqa: F401
                   MAX_SEGS, MAX_SPAWNS, NAVI_DIM, SEG_STRIDE, STATE_DIM, STATE_FIELDS, SimCfg, StepOut)

LCF_STATS_DOUBLES = 8 + 6 * 2048

F_ACTED, F_DONE, F_ARRIVE, F_CRASH, F_OUT, F_MAXSTEP, F_SPAWNED, F_ENV_RESET = (1 << i for i in range(8))
I_VELOCITY, I_STEERING, I_ACCELERATION, I_STEP_REWARD, I_COST, I_EPISODE_LENGTH, I_EPISODE_REWARD, \
    I_ROUTE_COMPLETION = range(8)

ERR_NAMES = {0: "COPO_OK", -1: "COPO_ERR_NULL", -2: "COPO_ERR_DIM", -3: "COPO_ERR_DEVICE", -4: "COPO_ERR_STATE",
             -5: "COPO_ERR_CONFIG"}


class CopoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class NetLayout(C.Structure):
    """Mirror of `copo_net_layout`."""
    _fields_ = [("w1", C.c_int64), ("b1", C.c_int64), ("w2", C.c_int64), ("b2", C.c_int64), ("w3", C.c_int64),
                ("b3", C.c_int64), ("in_dim", C.c_int32), ("out_dim", C.c_int32)]


class PpoCfg(C.Structure):
    """Mirror of `copo_ppo_cfg`."""
    _fields_ = [
        ("mb", C.c_int32), ("hidden", C.c_int32), ("act_dim", C.c_int32), ("n_value_heads", C.c_int32),
        ("pack_width", C.c_int32), ("col_actions", C.c_int32), ("col_logp", C.c_int32), ("col_dist", C.c_int32),
        ("col_adv", C.c_int32), ("col_meta_adv", C.c_int32), ("col_vpred", C.c_int32 * 3), ("col_vtarget", C.c_int32 * 3),
        ("use_kl", C.c_int32), ("old_value_loss", C.c_int32), ("operand_dtype", C.c_int32), ("reserved0", C.c_int32),
        ("clip_param", C.c_float), ("vf_clip_param", C.c_float), ("vf_loss_coeff", C.c_float), ("entropy_coeff", C.c_float),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("pol", NetLayout), ("val", NetLayout * 3), ("n_params", C.c_int64),
    ]


HEAD_PPO, HEAD_META_NEW, HEAD_META_OLD = 0, 1, 2
OPERAND_F32, OPERAND_BF16 = 0, 1
PPO_STATS = 8
META_DOT_PARTIALS = 8192

_SIGS = {
    "copo_ppo_workspace_floats": (C.c_int64, [C.POINTER(PpoCfg)]),
    "copo_ppo_fused_step_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 14 + [C.c_int32, C.c_int32, C.c_void_p,
                                                                                   C.c_int32, C.c_void_p, C.c_void_p]),
    "copo_gather_rows_f32": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "copo_pack_columns_f32": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "copo_dp_workspace_bytes": (C.c_int64, [C.POINTER(PpoCfg), C.c_int32]),
    "copo_ppo_fused_step_dp_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 13 + [C.c_void_p, C.c_int32, C.c_void_p,
                                                                                      C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "copo_dp_status": (C.c_int, [C.c_void_p, C.POINTER(PpoCfg), C.c_int32, C.c_void_p]),
    "copo_transpose_weights_f32": (C.c_int, [C.POINTER(PpoCfg), C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_mlp_forward_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32] +
                             [C.c_void_p] * 7),
    "copo_mlp_forward_rows_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 5 + [C.c_int64, C.c_int64, C.c_int32, C.c_int32] +
                                  [C.c_void_p] * 2),
    "copo_adam_step_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 5),
    "copo_meta_grads_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 15),
    "copo_meta_lcf_f64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] +
                          [C.c_void_p] * 5),
    "copo_meta_finish_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "copo_meta_step_f64": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 13 + [C.c_int32, C.c_int32] + [C.c_void_p] * 5 +
                           [C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "copo_episode_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "copo_plan_epoch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32] +
                        [C.c_void_p] * 5),
    "copo_meta_fold_len": (C.c_int64, [C.POINTER(PpoCfg)]),
    "copo_meta_batch_workspace_floats": (C.c_int64, [C.POINTER(PpoCfg), C.c_int32]),
    "copo_meta_batch_grads_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 8 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 4),
    "copo_meta_batch_dot_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_meta_rows_workspace_floats": (C.c_int64, [C.POINTER(PpoCfg), C.c_int64]),
    "copo_meta_rows_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3),
    "copo_meta_batch_wgrads_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 2 +
                                   [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 4),
    "copo_meta_rowstat_f32": (C.c_int, [C.POINTER(PpoCfg)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "copo_meta_batch_lcf_f64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32] +
                                [C.c_void_p] * 3 + [C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "copo_version": (C.c_int, []),
    "copo_build_info": (C.c_char_p, []),
    "copo_last_error": (C.c_char_p, []),
    "copo_sim_create": (C.c_int, [C.POINTER(SimCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "copo_sim_destroy": (C.c_int, [C.c_void_p]),
    "copo_sim_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StepOut), C.c_void_p]),
    "copo_sim_set_lcf_dist": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "copo_sim_set_force_lcf": (C.c_int, [C.c_void_p, C.c_double]),
    "copo_sim_set_capacity": (C.c_int, [C.c_void_p, C.c_int32]),
    "copo_sim_set_block": (C.c_int, [C.c_void_p, C.c_int32]),
    "copo_sim_set_chunk": (C.c_int, [C.c_void_p, C.c_int32]),
    "copo_sim_flush": (C.c_int, [C.c_void_p, C.c_void_p]),
    "copo_sim_set_debug": (C.c_int, [C.c_void_p, C.c_void_p]),
    "copo_sim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StepOut), C.c_void_p]),
    "copo_sim_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_sim_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_neighbours_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "copo_gae3_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_cc_fuse_mf_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "copo_cc_fuse_concat_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p, C.c_void_p]),
    "copo_lcf_mix_partial_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_lcf_mix_apply_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "copo_peer_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "copo_peer_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "copo_peer_free": (C.c_int, [C.c_void_p]),
    "copo_ipc_export": (C.c_int, [C.c_void_p, C.c_char_p]),
    "copo_ipc_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "copo_ipc_close": (C.c_int, [C.c_void_p]),
    "copo_peer_allreduce_sum_f32": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "copo_peer_status": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "copo_debug_peer_allreduce_all_ranks": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_void_p]),
    "copo_debug_rowpass_stamps": (C.c_int, [C.c_void_p]),
    "copo_debug_wg_times": (C.c_int, [C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcopo_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so that this library binds to the SAME HIP
    # runtime (one runtime per process -- two would not share devices, streams or allocations).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.copo_version()
    if v != ABI_VERSION:
        raise ImportError("libcopo_hip.so ABI %d != binding ABI %d" % (v, ABI_VERSION))
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise CopoError(rc, lib.copo_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream

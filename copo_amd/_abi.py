"""Struct layouts of include/copo_hip.h shared by the ctypes binding of libcopo_hip.so (`_capi`) and by the test
oracle's loader (tests/oracle_lib.py).  Importing this module loads no native code."""
import ctypes as C

ABI_VERSION = 8
MAX_AGENTS = 64
MAX_SEGS = 16
SEG_STRIDE = 16
MAX_LASERS = 256
MAX_SPAWNS = 256
MAX_SAFE = 32
MAX_ROUTES = 128
MAX_LINES = 128
LINE_STRIDE = 12
STATE_DIM = 6
NAVI_DIM = 10
INFO_DIM = 8
STATE_FIELDS = 16

SIM_CFG_FIELDS = [
    ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("num_lasers", C.c_int32), ("obs_dim", C.c_int32),
    ("nbr_k", C.c_int32), ("enable_lcf", C.c_int32), ("horizon", C.c_int32), ("delay_done", C.c_int32),
    ("respawn_cooldown", C.c_int32), ("substeps", C.c_int32),
    ("lidar_range", C.c_float), ("neighbours_distance", C.c_float), ("mf_distance", C.c_float),
    ("dt", C.c_float), ("veh_half_len", C.c_float), ("veh_half_wid", C.c_float), ("wheelbase", C.c_float),
    ("max_steer", C.c_float), ("max_speed", C.c_float), ("acc_max", C.c_float), ("brake_gain", C.c_float),
    ("brake_max", C.c_float), ("lat_acc_max", C.c_float), ("reverse_acc", C.c_float), ("spawn_region_len", C.c_float), ("spawn_region_wid", C.c_float),
    ("driving_reward", C.c_float), ("speed_reward", C.c_float), ("success_reward", C.c_float),
    ("crash_penalty", C.c_float), ("out_penalty", C.c_float), ("arrive_margin", C.c_float), ("body_margin", C.c_float),
    ("lane_width", C.c_float),
    ("lcf_mean", C.c_double), ("lcf_std", C.c_double),
    ("n_routes", C.c_int32), ("n_spawns", C.c_int32),
    ("route_segs", C.c_void_p), ("route_meta", C.c_void_p), ("spawn_tab", C.c_void_p), ("spawn_s", C.c_void_p),
    ("ray_cs", C.c_void_p),
    ("add_traffic_light", C.c_int32), ("traffic_light_interval", C.c_int32), ("comm_size", C.c_int32),
    ("comm_neighbours", C.c_int32), ("add_pos_in_comm", C.c_int32), ("map_bbox", C.c_float * 4),
    ("side_lasers", C.c_int32), ("lane_line_lasers", C.c_int32), ("side_range", C.c_float),
    ("lane_line_range", C.c_float), ("navi_dim", C.c_int32), ("toll_dim", C.c_int32), ("toll_min_steps", C.c_int32),
    ("n_lines", C.c_int32), ("lines", C.c_void_p), ("side_cs", C.c_void_p), ("lane_line_cs", C.c_void_p),
    ("toll_speed_limit", C.c_float), ("overspeed_penalty", C.c_float), ("toll_early_exit", C.c_int32),
    ("n_boxes", C.c_int32), ("boxes", C.c_void_p), ("boxes_hidden", C.c_int32),
]

STEP_OUT_FIELDS = ("obs", "rew", "nei_rew", "glob_rew", "flags", "nbr_idx", "nbr_cnt", "mf_cnt", "nbr_dist", "lcf",
                   "info", "agent_id")


class SimCfg(C.Structure):
    """Mirror of `copo_sim_cfg`."""
    _fields_ = SIM_CFG_FIELDS


class StepOut(C.Structure):
    """Mirror of `copo_step_out` (pointers, 0 = skip)."""
    _fields_ = [(n, C.c_void_p) for n in STEP_OUT_FIELDS]

/*
 * copo_hip.h -- C ABI of libcopo_hip.so, the MI355X (gfx950) hot path of the CoPO rollout-and-update engine.
 *
 * Every entry point replaces one Python interface of the reference (decisionforce/CoPO,
 * paths relative to copo_code/copo/torch_copo/); see INTEGRATION.md for the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All data pointers are DEVICE pointers owned by the
 *     caller (kept alive by the caller until the stream has consumed them) unless marked HOST.
 *   - every call returns COPO_OK (0) or a negative error code; nothing throws.  copo_last_error()
 *     returns a thread-local, library-owned message for the last failing call on this thread.
 *   - asynchronous on the hipStream_t passed as `void* stream` (NULL = default stream); no implicit
 *     device synchronisation inside any op.
 *   - one opaque handle per GPU process; a handle is not thread-safe (the reference runs one env per
 *     process, README.md:179-180).
 *   - deterministic: counter-based RNG keyed by (seed, env, slot, spawn counter).
 */
#ifndef COPO_HIP_H
#define COPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COPO_ABI_VERSION 8

#define COPO_OK 0
#define COPO_ERR_NULL (-1)      /* required pointer is NULL */
#define COPO_ERR_DIM (-2)       /* size/shape out of the supported range */
#define COPO_ERR_DEVICE (-3)    /* HIP runtime error (message in copo_last_error) */
#define COPO_ERR_STATE (-4)     /* call order violated (e.g. step before reset) */
#define COPO_ERR_CONFIG (-5)    /* inconsistent configuration / map tables */

#define COPO_MAX_AGENTS 64      /* slots per env (one wave64 owns an env's agents) */
#define COPO_MAX_SEGS 16        /* roads per route (a full turn of the roundabout is 11) */
#define COPO_SEG_STRIDE 16      /* floats per road record */
#define COPO_MAX_LASERS 256
#define COPO_MAX_BOXES 16       /* static boxes (buildings) of a map */
#define COPO_BOX_STRIDE 6       /* {x, y, cos, sin, half_len, half_wid} */
#define COPO_MAX_SPAWNS 256     /* spawn slots per map */
#define COPO_MAX_SAFE 32        /* respawn places (spawn slots with the `safe` mark) per map */
#define COPO_MAX_ROUTES 128
#define COPO_MAX_LINES 128      /* lane-line primitives per map (side / lane-line detectors) */
#define COPO_LINE_STRIDE 12
#define COPO_STATE_DIM 6        /* heading, speed, steering, last action x2, yaw rate */
#define COPO_NAVI_DIM 10
#define COPO_INFO_DIM 8
#define COPO_STATE_FIELDS 16
#define COPO_LCF_STATS_DOUBLES (8 + 6 * 2048) /* size of the `stats` workspace of copo_lcf_mix_* */

/* per-slot step flags (uint8 bitfield), the device-side equivalent of the info dict keys consumed by
 * utils/callbacks.py:63-91 (arrive_dest / crash / out_of_road / max_step) plus row bookkeeping. */
#define COPO_F_ACTED 0x01u      /* slot held an agent that acted this step -> a (obs, act, rew, done) row exists */
#define COPO_F_DONE 0x02u       /* that agent terminated this step */
#define COPO_F_ARRIVE 0x04u
#define COPO_F_CRASH 0x08u
#define COPO_F_OUT 0x10u
#define COPO_F_MAXSTEP 0x20u    /* the AGENT drove `horizon` steps without terminating (episode_lengths[id] >= horizon) */
#define COPO_F_SPAWNED 0x40u    /* a new agent occupies the slot after this step (obs valid, reward 0, no row) */
#define COPO_F_ENV_RESET 0x80u  /* the episode of this scene ended this step and the scene was reset (done["__all__"]) */

/* columns of the optional per-slot info output [E][N][COPO_INFO_DIM] (utils/callbacks.py:35-46) */
#define COPO_I_VELOCITY 0       /* km/h */
#define COPO_I_STEERING 1
#define COPO_I_ACCELERATION 2
#define COPO_I_STEP_REWARD 3
#define COPO_I_COST 4           /* 1 on a crash (MetaDrive's multi-agent default: out_of_road_cost = 0) */
#define COPO_I_EPISODE_LENGTH 5
#define COPO_I_EPISODE_REWARD 6
#define COPO_I_ROUTE_COMPLETION 7

/* Road record, COPO_SEG_STRIDE floats.  A road is one primitive (straight or arc) carrying `lanes` lanes; the record
 * describes the centre line of lane 0, the LEFTMOST lane -- lane i lies i * lane_width to its right:
 *   0 x0, 1 y0, 2 cos0, 3 sin0   start pose            4 length      5 kappa (signed: + = counter-clockwise)
 *   6 s_start (route arc length at the start)           7 theta0
 *   8 ckx, 9 cky   navigation check point: the end of the road at its lateral middle
 *  10 lanes + 0.25 (left edge line continuous) + 0.5 (right edge line continuous): the edges a body must not touch (body_margin)
 *            + 0.125 (left edge line BROKEN and crossable: the reference point may be up to one lane width left of lane 0 --
 *                     MetaDrive's on_lane of the opposite road across a broken centre line, as in the parking-lot block)
 *                  11 radius feature of the navigation block, 12 radius (0 for a plain straight), 13 angle feature
 *  14 umx, 15 umy unit vector from an arc's centre to its mid point (projection without a wrap inside the arc)
 * STRAIGHT records (kappa == 0) overload 12 / 14 / 15 for MetaDrive's Merge / Split blocks (maps.Net.add_funnel; read by
 * funnel_extra in the step kernel and the oracle): extra drivable width to the RIGHT of the record's lanes, bounded by the
 * outer edge of the outermost wave lane -- two arcs of opposite sense:
 *   12 wave radius R (> 0 switches the funnel on; 0 = plain straight, 14 / 15 are then ignored)
 *   14 extra width D at the wide end, signed: + the road narrows along its direction, - it widens
 *   15 arc length u1 from the wide end at which the edge line's first arc (radius R + w/2) hands over to the second (R - w/2)
 * A map builder that leaves field 12 of a straight non-zero therefore changes its out-of-road decisions.
 * A route has nseg roads followed by one terminal record (length 0) holding the end pose. */
#define COPO_SEG_CKX 8
#define COPO_SEG_LANES 10
#define COPO_SEG_FEAT 11
#define COPO_SEG_UMX 14

/* Lane-line primitive, COPO_LINE_STRIDE floats (side detector: continuous lines; lane-line detector: all lines):
 *   0 kind (1 broken, 2 continuous), 1 x0, 2 y0, 3 cos0, 4 sin0, 5 length, 6 kappa, 7 cx, 8 cy (arc centre),
 *   9 umx, 10 umy (centre -> arc mid point), 11 cos(half the arc angle) */

typedef struct copo_sim_cfg {
    /* population */
    int32_t num_envs;          /* E */
    int32_t num_agents;        /* N slots per env, <= COPO_MAX_AGENTS */
    int32_t num_lasers;        /* LiDAR beams (72; 240 for config C5) */
    int32_t obs_dim;           /* COPO_OBS_DIM() */
    int32_t nbr_k;             /* neighbour ids stored per slot (<= N-1) */
    int32_t enable_lcf;        /* 1: LCFEnv (append (lcf+1)/2 to obs, sample LCF at spawn); 0: CCEnv only */
    int32_t horizon;           /* MetaDrive `horizon` (1000), MultiAgentMetaDrive.step: (i) an agent that acted `horizon` steps is
                                  done with max_step; (ii) once `horizon` env steps have run the scene stops respawning and
                                  drains; (iii) the scene is reset when no agent is left driving (or after 5 x horizon steps) */
    int32_t delay_done;        /* steps a vehicle that terminated without arriving lingers as an obstacle (MetaDrive `delay_done`) */
    int32_t respawn_cooldown;  /* steps a slot stays empty before re-use (0: MetaDrive respawns in the same step) */
    int32_t substeps;          /* physics sub-steps per env step (5) */
    /* radii */
    float lidar_range;         /* m */
    float neighbours_distance; /* env_wrappers.py:40,168 (strict <) */
    float mf_distance;         /* algo_ccppo.py:43,283 (prefix of the sorted list with d <= mf) */
    /* vehicle + kinematic bicycle model (centre-referenced, slip angle from the steering angle) */
    float dt;                  /* env step seconds (0.1) */
    float veh_half_len, veh_half_wid, wheelbase;
    float max_steer;           /* rad */
    float max_speed;           /* m/s */
    float acc_max;             /* m/s^2 at full throttle (engine force cut above max_speed) */
    float brake_gain, brake_max; /* deceleration = min(brake_gain * |a1|, brake_max) for a1 < 0 */
    float lat_acc_max;         /* tyre friction limit on the lateral acceleration v x yaw rate (m/s^2); 0 = none (the bicycle turns on rails) */
    float reverse_acc;         /* m/s^2 at full NEGATIVE throttle when the vehicle has a reverse gear (MetaDrive `enable_reverse`, the
                                  ParkingLot's vehicle config: a negative throttle is engine force backwards, there is no brake); 0 = no
                                  reverse gear: a negative throttle brakes and the speed stops at 0 */
    float spawn_region_len, spawn_region_wid; /* the box that must hold no vehicle for a respawn (8 x 3 m) */
    /* reward (MetaDrive multi-agent scheme) */
    float driving_reward, speed_reward, success_reward, crash_penalty, out_penalty;
    float arrive_margin;       /* arrival: within +- this of the end of the final road */
    float body_margin;         /* out of road when the BODY touches the road's edge lines (MetaDrive: on_yellow / on_white_continuous_line,
                                  crash_sidewalk): the centre must keep body_margin x (half_wid |cos| + half_len |sin| of the heading
                                  error) from both edges.  0: the centre rule (vehicle.out_of_route alone) */
    float lane_width;
    /* LCF distribution at creation (LCFEnv.current_lcf_mean/std, env_wrappers.py:200-201) */
    double lcf_mean, lcf_std;
    /* map tables (HOST pointers, copied at create) */
    int32_t n_routes;
    int32_t n_spawns;
    const float* route_segs;   /* [n_routes][COPO_MAX_SEGS + 1][COPO_SEG_STRIDE] */
    const float* route_meta;   /* [n_routes][4] = {total_len, nseg, index of the toll-booth road or -1, exclusive destination id + 1 or 0}.
                                * Exclusive destinations (at most 32 per map; MetaDrive's ParkingSpaceManager): a spawn draws its route among
                                * those of its spawn place whose destination no LIVING agent of the scene is heading for (all of them if
                                * none is free); the space is free again when that agent is done. */
    const int32_t* spawn_tab;  /* [n_spawns][4] = {first_route, n_destinations, lane, safe} */
    const float* spawn_s;      /* [n_spawns] longitudinal position of the slot on its spawn road */
    const float* ray_cs;       /* [num_lasers][2] = beam directions in the vehicle frame (forward, left) */
    /* optional observation / action extensions of CCEnv / LCFEnv (env_wrappers.py:44-46, 89-118, 258-272, 331-337,
     * 362-371); all zero = off. */
    int32_t add_traffic_light;      /* append clip([message(t), x', y'], 0, 1): env_wrappers.py:258-272 */
    int32_t traffic_light_interval; /* steps per phase (30) */
    int32_t comm_size;              /* > 0: communication on -- actions are [2 + comm_size] floats per slot */
    int32_t comm_neighbours;        /* nearest neighbours whose message is appended (4) */
    int32_t add_pos_in_comm;        /* 1: each message is followed by [d/20, (lon/d+1)/2, (lat/d+1)/2] clipped to [0,1] */
    float map_bbox[4];              /* {x_min, x_max, y_min, y_max} of the road network (traffic-light position columns) */
    /* MetaDrive's optional detectors (Bottleneck: 4 + 4 beams -> O = 96; Tollgate: 72 + 4 beams, no navigation block,
     * two toll columns -> O = 156).  0 beams = the two / one lateral-distance columns of the default observation. */
    int32_t side_lasers;            /* side detector beams (continuous lines), first beam 90 deg clockwise of the heading */
    int32_t lane_line_lasers;       /* lane-line detector beams (all lines) */
    float side_range, lane_line_range;
    int32_t navi_dim;               /* 10, or 0 (Tollgate) */
    int32_t toll_dim;               /* 0, or 2 (Tollgate: on the booth road; stayed there longer than toll_min_steps -- zeros off it) */
    int32_t toll_min_steps;         /* steps a vehicle has to spend in a booth (30) */
    int32_t n_lines;
    const float* lines;             /* [n_lines][COPO_LINE_STRIDE] (HOST) */
    const float* side_cs;           /* [side_lasers][2] beam directions in the vehicle frame (forward, left) (HOST) */
    const float* lane_line_cs;      /* [lane_line_lasers][2] (HOST) */
    /* ABI 8 -- MetaDrive 0.2.5's MultiAgentTollgateEnv rules (restated from the release's published source, which is not in the
     * reference tree; all zero = the rules of ABI 7):
     *   toll_speed_limit  > 0: on the booth road (TollGate.SPEED_LIMIT = 3 km/h, here in m/s) the step reward is the driving
     *                          reward alone while |v| <= limit and -overspeed_penalty * |v| / max_speed above it
     *                          (`reward_function`: `if vehicle.overspeed: reward = -overspeed_penalty * speed / max_speed`);
     *                          off the booth road the speed term is added as everywhere (the env's own speed_reward is 0.0);
     *   toll_early_exit   1: a vehicle that leaves the booth road before toll_min_steps is DONE with the out_of_road flag and
     *                          keeps the step's ordinary reward (`done_function`: `done_info["out_of_road"] = True`, the reward
     *                          function does not see it); 0: it is a crash with -crash_penalty (rounds 2-5). */
    float toll_speed_limit;
    float overspeed_penalty;
    int32_t toll_early_exit;
    /*   static boxes      buildings of the map as oriented boxes {centre x, y, cos, sin of the long axis, half length, half width}
     *                          (`TollGate._add_building_and_speed_limit`: `if idx % 2 == 1` a TollGateBuilding of the lane's width and
     *                          the road's length at the centre of every SECOND lane of the booth road).  A vehicle whose box overlaps
     *                          one (the separating-axis test of the vehicle collisions) is done with the crash flag (MetaDrive:
     *                          crash_building); the LiDAR sees them like vehicles (ray / box test in the box frame, minimum of the hit
     *                          distances).  n_boxes <= COPO_MAX_BOXES; 0 = none. */
    int32_t n_boxes;
    const float* boxes;             /* [n_boxes][COPO_BOX_STRIDE] (HOST) */
    int32_t boxes_hidden;           /* 1: the LiDAR does NOT see the static boxes (they still end an agent on touch): the experiment of
                                       profiles/r06_fidelity.txt -- whether MetaDrive 0.2.5's LiDAR mask holds the buildings' walls is not
                                       something this tree can settle; 0 (default): seen */
} copo_sim_cfg;

/* Observation row: [side block | heading, speed, steering, last action x2, yaw rate | lane-line block | navigation |
 * lasers | toll | 3 traffic light | 1 lcf | comm_neighbours x (comm_size + 3 if add_pos_in_comm)], where the side block
 * is side_lasers beams or the 2 lateral-distance columns and the lane-line block lane_line_lasers beams or 1 column. */
#define COPO_SIDE_DIM(c) ((c)->side_lasers > 0 ? (c)->side_lasers : 2)
#define COPO_LANE_DIM(c) ((c)->lane_line_lasers > 0 ? (c)->lane_line_lasers : 1)
#define COPO_EGO_DIM(c) (COPO_SIDE_DIM(c) + COPO_STATE_DIM + COPO_LANE_DIM(c))
#define COPO_OBS_DIM(c)                                                                                       \
    (COPO_EGO_DIM(c) + (c)->navi_dim + (c)->num_lasers + (c)->toll_dim + ((c)->add_traffic_light ? 3 : 0) +     \
     ((c)->enable_lcf ? 1 : 0) +                                                                              \
     ((c)->comm_size > 0 ? (c)->comm_neighbours * ((c)->comm_size + ((c)->add_pos_in_comm ? 3 : 0)) : 0))
/* floats per slot of the action array: steering, throttle, then the message */
#define COPO_ACT_DIM(c) (2 + ((c)->comm_size > 0 ? (c)->comm_size : 0))

typedef struct copo_sim copo_sim;

/* outputs of one vectorised env step; any pointer except obs may be NULL to skip that output */
typedef struct copo_step_out {
    float* obs;          /* [E][N][O]   obs AFTER the step (the next policy input); rows of slots that hold no
                                         agent (neither ACTED nor SPAWNED) are NOT written -- the reference has no
                                         dict entry for them, and not streaming ~half of the rows halves the traffic */
    float* rew;          /* [E][N]      native reward of the acting agent (return_native_reward=True)     */
    float* nei_rew;      /* [E][N]      env_wrappers.py:321-325                                           */
    float* glob_rew;     /* [E]         env_wrappers.py:313                                               */
    uint8_t* flags;      /* [E][N]      COPO_F_* bitfield                                                 */
    int32_t* nbr_idx;    /* [E][N][K]   slot ids sorted by (distance, slot), -1 padded  (:125-139); like obs, rows of
                                         slots that held no agent when the lists were made (the scene BEFORE an
                                         end-of-episode reset) are not written -- nbr_cnt / mf_cnt / nei_rew are 0 there            */
    int32_t* nbr_cnt;    /* [E][N]      neighbours within neighbours_distance (may exceed K)              */
    int32_t* mf_cnt;     /* [E][N]      length of the list prefix with distance <= mf_distance            */
    float* nbr_dist;     /* [E][N][K]   distances (float64 compare, stored fp32)                          */
    float* lcf;          /* [E][N]      LCF in [-1,1] of the acting agent (info["lcf"]) / new occupant    */
    float* info;         /* [E][N][COPO_INFO_DIM]                                                          */
    int32_t* agent_id;   /* [E][N]      per-env running id of the acting agent ("agent%d"), -1 if none    */
} copo_step_out;

/* ---- library ---- */
int copo_version(void);
/* how this library was built, for bench lines and bug reports: ABI, target, profiling mask ("none" in the shipped build) and the
 * environment variables it reads (none: every tuning knob is an argument of this ABI).  Owned by the library. */
const char* copo_build_info(void);
const char* copo_last_error(void);

/* ---- vectorised multi-agent env: replaces MultiAgent*Env.step/reset (MetaDrive, call site
 *      utils/env_wrappers.py:95), CCEnv.step (:89-123) and LCFEnv.step/_get_reset_return (:274-391) ---- */
int copo_sim_create(const copo_sim_cfg* cfg, int device, copo_sim** out);
int copo_sim_destroy(copo_sim* sim);
/* seeds: HOST [E]; writes the reset observation + flags(SPAWNED) + lcf.  env_wrappers.py:274-305 */
int copo_sim_reset(copo_sim* sim, const uint64_t* seeds, const copo_step_out* out, void* stream);
/* LCFEnv.set_lcf_dist (env_wrappers.py:420-426): affects agents spawned from now on */
int copo_sim_set_lcf_dist(copo_sim* sim, double mean, double std);
/* The LCF distribution lives in device memory (launches captured in a hipGraph keep seeing updates).
 * set_lcf_dist/set_force_lcf mark it dirty; the next reset/step pushes it on its stream, except while that
 * stream is capturing -- replay-only callers push explicitly with copo_sim_flush before the replay. */
int copo_sim_flush(copo_sim* sim, void* stream);
/* LCFEnv.set_force_lcf (env_wrappers.py:428-430): v == -100 disables */
int copo_sim_set_force_lcf(copo_sim* sim, double v);
/* Population capacity, 1 <= capacity <= num_agents (default): slots >= capacity are left empty by a reset and never
 * respawn; vehicles already driving in them finish their episode.  With a following copo_sim_reset this is
 * `ChangeNEnv.close_and_reset_num_agents` of the curriculum baseline (env_wrappers.py:444-460) without re-creating
 * the simulator.  Device-resident like the LCF distribution (pushed by the next reset / step / flush). */
int copo_sim_set_capacity(copo_sim* sim, int32_t capacity);
/* act: [E][N][COPO_ACT_DIM] device fp32: steering, throttle (clipped to [-1,1] inside, as RLlib's clip_actions does),
 * then the comm_size message floats when the communication channel is on (passed through unclipped, like the reference) */
int copo_sim_step(copo_sim* sim, const float* act, const copo_step_out* out, void* stream);
/* raw state access for tests / checkpointing: [COPO_STATE_FIELDS][E][N] fp32 words + [E][4] int32 env words */
int copo_sim_get_state(copo_sim* sim, float* slot_state, int32_t* env_state, void* stream);
int copo_sim_set_state(copo_sim* sim, const float* slot_state, const int32_t* env_state, void* stream);
/* launch shape of the step kernel; tuning knob, results do not depend on it.  threads > 0: one scene per workgroup of 64 / 128 / 256 /
 * 512 / 1024 threads; threads < 0: the PACKED shape for large scene counts, -threads (2..16) scenes per workgroup with the per-agent phases
 * dense over the lanes (-1 = the default count; needs nbr_k <= 8, 0 < mf_distance < neighbours_distance, no traffic-light /
 * communication block: COPO_ERR_CONFIG otherwise); 0 = pick from E (packed above 3072 scenes where available) */
int copo_sim_set_block(copo_sim* sim, int32_t threads);
/* LiDAR fans whose ray minima are held in LDS at a time when ONE wave owns a scene (workgroup size 64): 1..64, 0 = default;
 * trades resident scenes per compute unit against fuller work batches; tuning knob, results do not depend on it */
int copo_sim_set_chunk(copo_sim* sim, int32_t fans);
/* profiling aid: device buffer [E][8] int64 receiving clock64() stamps at the phase boundaries of the step kernel
 * (NULL switches it off; results of the step do not depend on it); row [7]: which formulation of the neighbour lists ran */
int copo_sim_set_debug(copo_sim* sim, int64_t* stamps);

/* ---- stateless ops ---- */

/* CCEnv._update_distance_map + _find_in_range (env_wrappers.py:125-158) + LCFEnv reward block (:313-326).
 * pos [E][N][2] fp32, present [E][N] u8, rew [E][N] (may be NULL -> no reward outputs). */
int copo_neighbours_f32(const float* pos, const uint8_t* present, const float* rew, int32_t E, int32_t N, int32_t K,
                        float radius, float mf_distance, int32_t* nbr_idx, int32_t* nbr_cnt, int32_t* mf_cnt,
                        float* nbr_dist, float* nei_rew, float* glob_rew, void* stream);

/* Three GAE heads in one segmented reverse scan over [T][M] (M = E*N columns):
 * compute_advantages (algo_ccppo.py:362-373), compute_nei_advantage / compute_global_advantage
 * (algo_copo.py:189-204, 492-500).  rew/val/adv/tgt: [3][T][M]; flags [T][M] (ACTED/DONE bits);
 * gamma[heads] (HOST doubles); bootstrap = value of the segment's LAST row unless DONE.  Rows without ACTED get 0. */
int copo_gae3_f32(const float* rew, const float* val, const uint8_t* flags, int32_t T, int32_t M, int32_t heads,
                  const double* gamma, double lam, float* adv, float* tgt, void* stream);

/* mean_field_ccppo_process / concat_ccppo_process (algo_ccppo.py:225-311) over [T][E][N] rows.
 * obs [R][N][O], act [R][N][A], flags [R][N], nbr_idx [R][N][K], cnt [R][N] (mf: mf_cnt, concat: nbr_cnt),
 * with R = T*E.  cc_obs [R][N][C]: C = 2O+A (mf, counterfactual), 2O (mf), O+k(O+A) / O+kO (concat). */
int copo_cc_fuse_mf_f32(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                        const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                        int32_t counterfactual, float* cc_obs, void* stream);
int copo_cc_fuse_concat_f32(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                            const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                            int32_t num_neighbours, int32_t counterfactual, float* cc_obs, void* stream);

/* CoPOTrainer.training_step coordinated-advantage block (algo_copo.py:539-551):
 *   A_c = cos(lcf*pi/2)*adv + sin(lcf*pi/2)*nei_adv ; stats of A_c and glob_adv over valid rows;
 *   norm_adv = (A_c-mean)/max(1e-4,std) ; glob_adv_std likewise (population std).
 * Two-phase for data-parallel runs: `_partial` writes {count, sum, sumsq} x2 as doubles stats[0..5]
 * (caller all-reduces those six), `_apply` consumes the (reduced) stats.  `stats` is a device workspace of
 * COPO_LCF_STATS_DOUBLES doubles (per-block partials live behind the six results; the reduction order is
 * fixed, so results are run-to-run deterministic).  valid [B] u8 may be NULL. */
int copo_lcf_mix_partial_f32(const float* adv, const float* nei_adv, const float* glob_adv, const float* lcf,
                             const uint8_t* valid, int64_t B, float* mixed, double* stats, void* stream);
int copo_lcf_mix_apply_f32(const float* mixed, const float* glob_adv, const uint8_t* valid, int64_t B,
                           const double* stats, float* norm_adv, float* glob_adv_std, void* stream);

/* Sums behind `MultiAgentDrivingCallbacks.on_episode_end / on_train_result` (utils/callbacks.py:48-110) over the
 * n_rows rows of one iteration (flags u8, info [n_rows][COPO_INFO_DIM], nbr_cnt i32): out15[0..7] over rows that acted
 * and terminated = {count, arrive, crash, out_of_road, max_step, sum info[5], info[6], info[7]}, out15[8..14] over rows
 * that acted = {count, sum info[0..4], sum nbr_cnt}. */
int copo_episode_metrics(const uint8_t* flags, const float* info, const int32_t* nbr_cnt, int64_t n_rows, double* out15,
                         void* stream);

/* Row movers around the SGD loop.  copo_gather_rows_f32: dsts[s][r][:] = srcs[s][rows[r]][:] for n_src <= COPO_GATHER_MAX_SRC
 * sources of widths[s] floats per row in ONE launch (host arrays of device pointers) -- the epoch's planned rows into minibatch
 * order, after which copo_ppo_fused_step_f32 takes rows = NULL.  copo_pack_columns_f32: pack[r] = [cols[0][r] | cols[1][r] | ...]
 * (widths[c] floats each, <= COPO_PACK_MAX_COLS columns): the per-row pack the step kernels read, from the iteration's separate
 * column tensors.  (Replace RLlib's SampleBatch column handling inside `train_one_step`, algo_copo.py:555-558.) */
#define COPO_GATHER_MAX_SRC 4
#define COPO_PACK_MAX_COLS 24
int copo_gather_rows_f32(const float* const* srcs, float* const* dsts, const int32_t* widths, int32_t n_src, const int64_t* rows,
                         int64_t n_rows, void* stream);
int copo_pack_columns_f32(const float* const* cols, const int32_t* widths, int32_t n_cols, int64_t n_rows, float* pack,
                          void* stream);

/* Minibatch plan of one SGD epoch (the static-shape replacement of RLlib's shuffled minibatch iterator used by
 * `train_one_step`, algo_copo.py:555-558): from a permutation `perm` of this rank's B_local valid rows `valid_idx`,
 * minibatch k takes q + (k < r) consecutive entries of the shuffled list (q, r = divmod(B_local, n_mb)); rows / w are
 * [n_mb][mb] (padding: row 0, weight 0), denom[k] = number of rows of minibatch k over ALL ranks (B_all: HOST array of
 * `world` counts), *mb_index (may be NULL) is reset to 0.  perm NULL: the shuffle is a keyed pseudo-random permutation
 * computed in the kernel (4-round Feistel network with cycle walking, key4_host = four 32-bit words on the HOST). */
int copo_plan_epoch(const int64_t* valid_idx, const int64_t* perm, const uint32_t* key4_host, int64_t B_local, int32_t n_mb,
                    int32_t mb, const int64_t* B_all_host, int32_t world, int64_t* rows, float* w, float* denom,
                    int64_t* mb_index, void* stream);

/* ---- fused minibatch learner --------------------------------------------------------------------------------
 * Replaces, for one static-shape minibatch, `Policy.loss` + autograd + Adam of the reference
 * (algo_ippo.py:78-172, algo_ccppo.py:376-472, algo_copo.py:311-424; RLlib train_one_step, algo_copo.py:555-558)
 * and the two policy-gradient evaluations of `CoPOPolicy.meta_update` (algo_copo.py:250-278).
 * All fp32 parameters of the model live in ONE flat buffer `theta`; a net is obs -> H -> H -> out (tanh). */
typedef struct copo_net_layout {
    int64_t w1, b1, w2, b2, w3, b3;   /* offsets (floats) into theta; weights are [out][in] row-major */
    int32_t in_dim, out_dim;
} copo_net_layout;

#define COPO_HEAD_PPO 0        /* full PPO loss: policy net + value heads */
#define COPO_HEAD_META_NEW 1   /* policy net only, loss = mean(-clipped surrogate) with the global advantage */
#define COPO_HEAD_META_OLD 2   /* policy net only, loss = mean(logp(action))  (target network)           */
#define COPO_PPO_MAX_MB 1024   /* rows per minibatch supported by the fused learner */
#define COPO_META_BATCH_MAX 256        /* minibatches per copo_meta_batch_grads_f32 call */
#define COPO_META_DOT_PARTIALS 8192 /* doubles in the `dot_partials` workspace of the meta update */
#define COPO_PPO_MAX_KSPLIT 4  /* row splits of the weight-gradient GEMMs (fixed order -> deterministic sums) */
#define COPO_OPERAND_F32 0
#define COPO_OPERAND_BF16 1
#define COPO_PPO_STATS 8       /* sums of: total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, advantage  */

typedef struct copo_ppo_cfg {
    int32_t mb;                /* rows per minibatch (static shape; padded rows carry weight 0) */
    int32_t hidden;            /* H: width of both hidden layers */
    int32_t act_dim;           /* 2 */
    int32_t n_value_heads;     /* 1 (IPPO / CCPPO) or 3 (CoPO: ego, neighbourhood, global) */
    int32_t pack_width;        /* floats per row of pack_src */
    int32_t col_actions, col_logp, col_dist, col_adv, col_meta_adv;  /* pack columns */
    int32_t col_vpred[3], col_vtarget[3];
    int32_t use_kl, old_value_loss;
    int32_t operand_dtype;     /* COPO_OPERAND_F32, or COPO_OPERAND_BF16: the `policy_dtype = bfloat16` configuration -- inputs,
                                  weights, activations and activation gradients are rounded to bfloat16 wherever torch.autocast
                                  rounds them, products accumulate in fp32 (the arithmetic of v_mfma_f32_*_bf16); parameters,
                                  Adam and the losses stay fp32.  Row-pass shapes, PPO head mode and copo_mlp_forward_f32 only. */
    int32_t reserved0;
    float clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff;
    float lr, beta1, beta2, eps;   /* Adam (torch.optim.Adam semantics, no weight decay) */
    copo_net_layout pol, val[3];
    int64_t n_params;          /* length of theta (floats) */
} copo_ppo_cfg;

/* floats of `workspace` needed for this configuration */
int64_t copo_ppo_workspace_floats(const copo_ppo_cfg* cfg);
/* One minibatch step.  rows [mb] index the dense sources (obs_src [R][pol.in_dim], cc_src [R][val.in_dim] or NULL
 * = obs_src, pack_src [R][pack_width]); w [mb] row weights; denom [1] = global number of valid rows (device);
 * apply_adam = 1: parameters and Adam moments are updated in the epilogues and *step is incremented;
 * apply_adam = 0: only `grad` (flat, same layout as theta) is written.  stats (may be NULL) accumulates.
 * mb_index (device int64, may be NULL = 0) selects the minibatch k: rows / w are then [n_mb][mb] tables and denom
 * is [n_mb]; bump_index = 1 increments *mb_index at the end, so that a captured hipGraph of this call walks the
 * epoch plan by itself.
 * theta_t (may be NULL): a mirror of theta, same length and offsets, in which W1 and W2 of every net are stored
 * transposed ([in][out]).  When given, the forward passes read it (coalesced B operands) and the Adam epilogue
 * keeps it current; create / refresh it with copo_transpose_weights_f32 whenever theta was changed from outside. */
int copo_ppo_fused_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, float* grad,
                            const float* obs_src, const float* cc_src, const float* pack_src, const int64_t* rows,
                            const float* w, const float* denom, const float* kl_coeff, int64_t* step,
                            float* workspace, float* stats, int32_t apply_adam, int32_t head_mode,
                            int64_t* mb_index, int32_t bump_index, float* theta_t, void* stream);
int copo_transpose_weights_f32(const copo_ppo_cfg* cfg, const float* theta, float* theta_t, void* stream);
/* Forward-only pass of nets [first_net, first_net + n_nets) of the layout (0 = policy, 1.. = value nets) over n_rows
 * DENSE rows: `model.forward` / `central_value_function` / `get_nei_value` / `get_global_value` of the reference's
 * models (algo_ccppo.py:74-219, algo_copo.py:96-182) for rollouts and the dense postprocess.  values [n_nets][n_rows]
 * (row of a policy net unused); for the policy net optionally dist_inputs [n_rows][4], and with eps [n_rows][2]
 * (standard normal draws) the sampled action, its log-probability and the action clipped to [-1, 1].  Needs the
 * transposed mirror theta_t and hidden in {64, 128, 256, 512}. */
int copo_mlp_forward_f32(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t, const float* obs_src,
                         const float* cc_src, int64_t n_rows, int32_t first_net, int32_t n_nets, float* values,
                         float* dist_inputs, const float* eps, float* action, float* logp, float* clipped, void* stream);
/* The same pass over a LIST of rows: launch row m reads row rows[m] of the sources ([n_src_rows][..]) and writes
 * values[net][rows[m]] (values is [n_nets][n_src_rows]; entries of unlisted rows are left alone).  The dense postprocess
 * of a rollout buffer uses it with the rows that hold an agent -- about half of the slots. */
int copo_mlp_forward_rows_f32(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t, const float* obs_src,
                              const float* cc_src, const int64_t* rows, int64_t n_rows, int64_t n_src_rows,
                              int32_t first_net, int32_t n_nets, float* values, void* stream);
/* Adam on the flat buffers (the data-parallel path: after the gradient all-reduce); theta_t as above or NULL.
 * workspace: the workspace of the copo_ppo_fused_step_f32(apply_adam = 0, `step` given) call that produced `grad` --
 * that call published the step number and the next minibatch index there, so this one needs no trailing counter
 * launch -- or NULL: *step + 1 is used and both counters are advanced by a separate 1-thread kernel. */
int copo_adam_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, const float* grad,
                       int64_t n, int64_t* step, int64_t* mb_index, float* theta_t, float* workspace, void* stream);

/* ---- data-parallel SGD step: the local step's two launches, gradients summed over the ranks INSIDE the weight-gradient
 * kernel (replaces RLlib's multi_gpu_train_one_step + the weight broadcasts, algo_copo.py:555-558, 572-613).  One process per
 * GPU; every rank owns an exchange workspace of copo_dp_workspace_bytes(cfg, world) bytes from copo_peer_alloc (uncached
 * device memory), exported / opened with copo_ipc_* so that `dp_workspaces[r]` is rank r's workspace as mapped HERE.
 * Every 32 x 32 gradient tile is sent to the rank that owns it (tile index mod world), added up there in rank order and
 * sent back; then every rank applies the same Adam update, so parameters, moments and the transposed mirror stay
 * bit-identical on all ranks without being communicated.  Arguments as copo_ppo_fused_step_f32 (PPO head mode, Adam
 * applied, `w / denom` with the GLOBAL row count in denom); every rank must make the same calls in the same order.
 * world == 1 is the local step.  A peer that does not answer within ~10 s raises an error word (copo_dp_status) instead of
 * hanging the GPU; every later wait then returns at once. */
int64_t copo_dp_workspace_bytes(const copo_ppo_cfg* cfg, int32_t world);
int copo_ppo_fused_step_dp_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, const float* obs_src,
                               const float* cc_src, const float* pack_src, const int64_t* rows, const float* w,
                               const float* denom, const float* kl_coeff, int64_t* step, float* workspace, float* stats,
                               int64_t* mb_index, int32_t bump_index, float* theta_t, void* const* dp_workspaces,
                               int32_t rank, int32_t world, void* stream);
int copo_dp_status(void* workspace, const copo_ppo_cfg* cfg, int32_t world, void* stream);

/* ---- LCF meta update (CoPOPolicy.meta_update, algo_copo.py:228-309) in three calls ------------------------------
 * (1) both policy gradients in one grouped pass: g_new = d mean(-clip-surrogate(global adv)) / d theta on the
 *     current policy, g_old = d mean(logp) / d theta_target on the target policy (flat layout; only the policy
 *     block is written).  stats_new[1] / stats_old[1] accumulate the two losses, stats_new[7] the mean advantage. */
int copo_meta_grads_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, float* g_new, float* g_old,
                        const float* obs_src, const float* pack_src, const int64_t* rows, const float* w,
                        const float* denom, float* workspace, float* stats_new, float* stats_old,
                        double* dot_partials /* [COPO_META_DOT_PARTIALS], zero-initialised once by the caller */,
                        int64_t* mb_index, void* stream);
/* (2) fp64 LCF terms of the minibatch: tail = {dS/dp0, dS/dp1, S, mean(A')} with S = mean((A' - mu)/sigma),
 *     A' = cos(phi) A_ego + sin(phi) A_nei, phi = (lcf_mean + lcf_std * eps) * pi/2 (reparameterised sample).
 *     eps [n_mb][mb] doubles; lcf_param [2] doubles (the model's lcf_parameters); raw_mean_std [2] doubles. */
int copo_meta_lcf_f64(const float* pack_src, int32_t pack_width, int32_t col_adv, int32_t col_nei_adv,
                      const int64_t* rows, const float* w, const float* denom, const double* eps, int32_t mb,
                      const int64_t* mb_index, const double* lcf_param, const double* raw_mean_std, double* tail,
                      void* stream);
/* (3) grad_value = <g_new[0:n], g_old[0:n]> -- from the per-workgroup partials that (1) left in dot_partials, or,
 *     when dot_partials is NULL (data-parallel: the gradients were all-reduced in between), recomputed from
 *     g_new / g_old -- (fp64 accumulate), LCF gradient grad_value * tail[0:2], Adam (fp64,
 *     betas 0.9/0.999, eps 1e-8) on lcf_param in place; adam_state [5] doubles = {m0, m1, v0, v1, step};
 *     stats [7] doubles accumulate {new loss, old loss, S, grad_value*S, grad_value, mean A', mean global adv}.
 *     Data-parallel runs all-reduce g_new/g_old (and tail) between (2) and (3). */
int copo_meta_finish_f64(const float* g_new, const float* g_old, int64_t n, const double* dot_partials,
                         const double* tail, double* lcf_param, double* adam_state, double lr, float* stats_new,
                         float* stats_old, double* stats, int64_t* mb_index, int32_t bump_index, void* stream);

/* (1)+(2)+(3) in one call for the single-process case (no gradient all-reduce in between): the last workgroup of
 * the gradient fold runs the LCF part and the LCF Adam step, so a whole meta step (`CoPOPolicy.meta_update`,
 * algo_copo.py:228-309) is six kernel launches.  Arguments as in the three calls above. */
int copo_meta_step_f64(const copo_ppo_cfg* cfg, float* theta, float* theta_target, float* g_new, float* g_old,
                       const float* obs_src, const float* pack_src, const int64_t* rows, const float* w,
                       const float* denom, float* workspace, float* stats_new, float* stats_old, double* dot_partials,
                       int32_t col_adv, int32_t col_nei_adv, const double* eps, double* lcf_param,
                       const double* raw_mean_std, double* tail, double* adam_state, double lr, double* stats,
                       int64_t* mb_index, int32_t bump_index, void* stream);

/* ---- batched LCF meta pass ------------------------------------------------------------------------------------
 * The two policy gradients of `meta_update` depend on the minibatch and on the policy / target parameters, which
 * the LCF loop (algo_copo.py:581-589) does not change -- only the two LCF parameters move.  So the gradient pairs
 * of many minibatches are computed in ONE grouped launch chain (phase A), and the sequential fp64 LCF Adam steps
 * run afterwards in one kernel (phase B).  Results equal the step-by-step calls above.
 *
 * Phase A: minibatches mb_first .. mb_first + nb - 1 of the row tables (rows / w [n_mb][mb], denom [n_mb]).
 *   gv_out [nb] doubles = <g_new, g_old> of each minibatch; stats_out [nb][2][COPO_PPO_STATS] = loss statistics of
 *   the new-policy / old-policy pass; g_out NULL, or [nb][2][copo_meta_fold_len()] to export the gradients
 *   (data-parallel: all-reduce them, then copo_meta_batch_dot_f64 recomputes gv).  workspace:
 *   copo_meta_batch_workspace_floats(cfg, nb_cap) floats, 8-byte aligned, zero-initialised once; nb <= nb_cap. */
int64_t copo_meta_fold_len(const copo_ppo_cfg* cfg);
int64_t copo_meta_batch_workspace_floats(const copo_ppo_cfg* cfg, int32_t nb);
int copo_meta_batch_grads_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, const float* obs_src,
                              const float* pack_src, const int64_t* rows, const float* w, const float* denom,
                              float* workspace, int32_t nb_cap, int64_t mb_first, int32_t nb, float* g_out,
                              double* gv_out, float* stats_out, void* stream);
/* gv_out[b] = <g[b][0], g[b][1]> (fp64 accumulation, fixed order) of exported -- all-reduced -- gradient pairs; denom (may be
 * NULL): [nb] row counts D_b, the result is scaled by 1 / D_b^2 (exported gradients of the row-store path carry unit weights).
 * partials: caller-owned scratch of nb * COPO_META_DOT_SPLIT doubles (stream-ordered like every other argument: calls on different
 * streams need different scratch), or NULL: one workgroup per minibatch, no scratch, slower. */
#define COPO_META_DOT_SPLIT 8
int copo_meta_batch_dot_f64(const float* g /* [nb][2][n] */, int64_t n, int32_t nb, double* gv_out, const float* denom, double* partials,
                            void* stream);
/* Row store.  Everything of phase A that is local to a ROW (both forward passes, the loss gradients, the activation
 * gradients -- with unit row weight) does not depend on how a meta pass groups the rows into minibatches, and the
 * `lcf_num_iters` passes of one training iteration regroup the same rows.  copo_meta_rows_f32 computes it once for rows
 * [0, n_rows) of the dense sources into rows_ws (copo_meta_rows_workspace_floats floats) + rowstat [2 * ceil(n_rows / mb)
 * * mb][2]; copo_meta_batch_wgrads_f32 is then phase A of one chunk of minibatches: only the weight-gradient GEMMs,
 * gathering their operands from the row store through the minibatch row tables, the dot products (scaled by
 * 1 / denom^2) and the loss statistics regrouped from rowstat.  Same results as copo_meta_batch_grads_f32.
 * Exported gradients (g_out) carry unit row weights: divide by denom before use. */
int64_t copo_meta_rows_workspace_floats(const copo_ppo_cfg* cfg, int64_t n_rows);
int copo_meta_rows_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, const float* obs_src,
                       const float* pack_src, int64_t n_rows, float* rows_ws, float* rowstat, void* stream);
int copo_meta_batch_wgrads_f32(const copo_ppo_cfg* cfg, const float* obs_src, const int64_t* rows, const float* w,
                               const float* denom, const float* rows_ws, int64_t n_rows, const float* rowstat,
                               float* workspace, int32_t nb_cap, int64_t mb_first, int32_t nb, float* g_out,
                               double* gv_out, float* stats_out, void* stream);
/* (ABI 6) stats_out == NULL: the chunk's loss statistics are not regrouped here -- the caller has them from ONE
 * copo_meta_rowstat_f32 launch for all minibatches [mb_first, mb_first + nb) of the pass (they depend on the row tables and the
 * row store only, not on the GEMMs): stats_out [nb][2][8] as copo_meta_batch_wgrads_f32 writes them.  nb <= 65535. */
int copo_meta_rowstat_f32(const copo_ppo_cfg* cfg, const int64_t* rows, const float* w, const float* denom, const float* rowstat,
                          int64_t mb_first, int32_t nb, float* stats_out, void* stream);
/* Phase B: n_mb sequential LCF Adam steps (minibatch order) in one kernel.  Row inputs either gathered from
 * pack_src via rows (ego_nei NULL, n_seg 1) or dense: ego_nei [n_seg][n_mb][mb][2] = {A_ego, A_nei} with w / eps
 * [n_seg][n_mb][mb] (data-parallel: the all-gathered rows of every rank).  gv [n_mb], stats_in [n_mb][2][8] from
 * phase A; lcf_param / adam_state / stats as in copo_meta_finish_f64.
 * n_wg workgroups share the rows of every step (0 = automatic: 1 up to eight segments, where one workgroup is as fast, then
 * one per four segments; <= 16): each adds up
 * its rows, publishes three partial sums in `exchange` (COPO_META_SEQ_XCHG_DOUBLES doubles of device memory, zeroed ONCE by the
 * caller and then left to the kernel) and applies the same Adam step to the sum of all partials -- with N ranks' rows a step
 * then costs what it costs with one.  A workgroup that never arrives (2 s) turns lcf_param into NaN.
 * (ABI 6) k_first / k_count: the launch takes the steps of minibatches [k_first, k_first + k_count) of the n_mb the arrays hold
 * (k_count < 0: all from k_first on) -- a meta pass can run its LCF steps chunk by chunk, behind the dot products of each chunk. */
#define COPO_META_SEQ_XCHG_DOUBLES 256
int copo_meta_batch_lcf_f64(const float* pack_src, int32_t pack_width, int32_t col_adv, int32_t col_nei_adv,
                            const int64_t* rows, const float* ego_nei, int32_t n_seg, const float* w, const double* eps,
                            const float* denom, int32_t mb, int32_t n_mb, const double* gv, const float* stats_in,
                            double* lcf_param, const double* raw_mean_std, double* adam_state, double lr, double* stats,
                            int32_t k_first, int32_t k_count, int32_t n_wg, double* exchange, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Peer all-reduce (replaces the in-process tower averaging of RLlib's multi-GPU learner,
 * algo_copo.py:519-577, for one process per GPU): a two-shot sum over device memory that every rank of the node has mapped.
 *   workspace = copo_peer_alloc(copo_peer_workspace_bytes(n, world))      uncached device memory, zeroed; its first n floats are
 *               the buffer that is reduced in place (the caller writes its gradient sums there)
 *   copo_ipc_export / copo_ipc_open     64-byte handles, exchanged by the host side (torch.distributed all_gather_object)
 *   copo_peer_allreduce_sum_f32(workspaces[world] as mapped in THIS process, n, rank, world, stream)
 *               one kernel: shards scattered to their owners' inboxes, added up in rank order (bit-identical on every rank),
 *               results stored to every rank's buffer; flags in the workspaces order the two hops.  Every rank must call it the
 *               same number of times.  A peer that does not answer within ~2 s raises an error word instead of hanging the GPU:
 *   copo_peer_status(workspace, n, world, stream)    COPO_OK, or COPO_ERR_DEVICE after a timed-out wait
 */
#define COPO_PEER_MAX_WORLD 16
#define COPO_IPC_HANDLE_BYTES 64
int64_t copo_peer_workspace_bytes(int64_t n, int32_t world);
int copo_peer_alloc(int64_t bytes, void** out);
int copo_peer_free(void* workspace);
int copo_ipc_export(void* dev_ptr, unsigned char* handle64);
int copo_ipc_open(const unsigned char* handle64, void** out);
int copo_ipc_close(void* mapped);
int copo_peer_allreduce_sum_f32(void* const* workspaces, int64_t n, int32_t rank, int32_t world, void* stream);
int copo_peer_status(void* workspace, int64_t n, int32_t world, void* stream);
/* test entry: all `world` ranks in one launch (the workspaces all belong to the calling process) */
int copo_debug_peer_allreduce_all_ranks(void* const* workspaces, int64_t n, int32_t world, void* stream);
/* profiling entries (scripts/ only): device-side wall-clock stamps that profiling calls of the fused learner leave behind --
 * 16 phase stamps of one row-pass workgroup / [2][1024][2] start-end stamps per workgroup of the two kernels of a step */
int copo_debug_rowpass_stamps(unsigned long long* out16);
int copo_debug_wg_times(unsigned long long* out4096);

#ifdef __cplusplus
}
#endif
#endif /* COPO_HIP_H */

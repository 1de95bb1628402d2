// Shared declarations of the simulator kernels and their host launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/copo_hip.h"
#include "sim_math.h"

#define COPO_SIM_MAX_BLOCK 1024

namespace copo {

enum : int { ST_EMPTY = 0, ST_ALIVE = 1, ST_WRECK = 2 };
enum : uint32_t { RNG_ROUTE = 1, RNG_LCF1 = 2, RNG_LCF2 = 3, RNG_SPAWN = 4, RNG_PERM = 16 };

// Kernel parameter block (passed by value).  All pointers are device pointers owned by the handle.
struct SimParams {
    int32_t E, N, O, K, num_lasers, enable_lcf, horizon, delay_done, respawn_cooldown, substeps;
    int32_t n_routes, n_spawns, n_safe, n_lines;
    int32_t n_spaces;              // exclusive destinations (route_meta[.][3] = id + 1, at most 32); 0 = none
    int32_t chunk;                 // present agents whose LiDAR fans are in LDS at a time (launch shape: sim_shape_params)
    int32_t nbr_chunk;             // present agents whose pair-parallel neighbour lists are in LDS at a time (same)
    int32_t chunk_one_wave;        // `chunk` of the one-wave-per-scene shape (copo_sim_set_chunk; 0 = default)
    int32_t stage_tables;          // 1: the step kernel copies the route / spawn tables to LDS (launch shape)
    int32_t pack_scenes;           // packed launch shape (sim_packed.hip): scenes per workgroup; 0 = one scene per workgroup
    int32_t seg_rows;              // road records per route in the DEVICE copy of route_segs: longest route + 1 (compacted at create)
    int32_t side_lasers, lane_lasers, navi_dim, toll_dim, toll_min_steps;
    int32_t lists_for_absent;      // 1: nbr_idx / nbr_dist rows of absent slots are filled with -1 / 0 (the stateless op); 0: left alone
    float lidar_range, neighbours_distance, mf_distance, dt, hl, hw, wheelbase, max_steer, max_speed;
    float acc_max, brake_gain, brake_max, lat_acc_max, reverse_acc, region_hl, region_hw;
    float driving_reward, speed_reward, success_reward, crash_penalty, out_penalty, arrive_margin, body_margin, lane_width;
    float side_range, lane_range;
    float toll_speed_limit, overspeed_penalty;   // MultiAgentTollgateEnv's booth rules (copo_sim_cfg, ABI 8); 0 = off
    int32_t toll_early_exit, n_boxes, n_boxes_lidar;      // n_boxes: static boxes (buildings) of the map, `boxes` below; n_boxes_lidar: those the LiDAR sees (all or none)
    float side_theta0, side_rpr;   // evenly spaced side-detector beams: angle of beam 0 in the vehicle frame, beams per radian (signed); rpr = 0: not evenly spaced
    float lane_theta0, lane_rpr;   // the same for the lane-line detector's beams
    // register formulation of the neighbour lists (neighbours_fast): fp32 d^2 thresholds 1e-6 inside / outside the exact radius,
    // key thresholds of the mean-field range, and whether the configuration qualifies (K <= 8, mean-field range inside the radius)
    float nbr_r2lo, nbr_r2hi;
    uint32_t mf_key_lo, mf_key_hi;
    int32_t nbr_fast;
    float ray_sign;                // +1: beam k is turned k steps counter-clockwise of the heading, -1: clockwise (MetaDrive)
    // constants derived once on the host, in float, exactly as the oracle derives them
    float inv_w, inv_range, inv_vnorm, inv_dt, inv_side_range, inv_lane_range, inv_toll, h_sub;
    // observation row layout: [side block | 6 state | lane block | navigation | lasers | toll | traffic light | lcf | comm]
    int32_t col_state, col_lane, col_navi, col_lidar, col_toll;
    // observation / action extensions (copo_sim_cfg): columns are -1 when the block is off
    int32_t act_dim, tl_interval, comm_size, comm_nb, comm_pos, col_tl, col_lcf, col_comm;
    float bbox[4];
    float* state;                  // [COPO_STATE_FIELDS][E][N] 32-bit words
    int32_t* env;                  // [E][4] = {t_env, episode, next_aid, started}
    const uint64_t* seeds;         // [E]
    const float* route_segs;       // [R][seg_rows][COPO_SEG_STRIDE]
    const float* route_meta;       // [R][4]
    const int32_t* spawn_tab;      // [P][4]
    const float* spawn_s;          // [P]
    const int32_t* safe_ids;       // [n_safe] spawn slots that are respawn places
    const float* safe_pose;        // [n_safe][4] pose of respawn place q: x, y, cos, sin of its lane (derived on the host at create)
    const float* ray_cs;           // [num_lasers][2]
    const float* side_cs;          // [side_lasers][2]
    const float* lane_cs;          // [lane_lasers][2]
    const float* lines;            // [n_lines][COPO_LINE_STRIDE]
    const float* boxes;            // [n_boxes][COPO_BOX_STRIDE] static boxes {x, y, cos, sin, half_len, half_wid}
    long long* dbg;                // optional [E][8] phase timestamps (clock64) of the step kernel; NULL = off
    const float* lcf_dist;         // [4] = {mean (force_lcf folded in), std, capacity, 0}: device memory so that captured graphs see updates
};

using StepOut = copo_step_out;

void sim_shape_params(SimParams& p, int block);
// p: host copy (launch shape), p_dev: the same block in device memory (what the kernels read)
hipError_t launch_sim_reset(const SimParams& p, const SimParams* p_dev, const StepOut& out, int block, hipStream_t stream);
hipError_t launch_sim_step(const SimParams& p, const SimParams* p_dev, const float* act, const StepOut& out, int block, hipStream_t stream);
// packed launch shape (sim_packed.hip): S scenes per workgroup, per-agent phases dense over the lanes
bool sim_packed_supported(const SimParams& p);
int sim_packed_default_scenes(const SimParams& p);
int sim_packed_chunk(const SimParams& p);
size_t sim_packed_lds_bytes(const SimParams& p, int S);
hipError_t launch_sim_step_packed(const SimParams& p, const SimParams* p_dev, const float* act, const StepOut& out, int S, hipStream_t stream);
// stateless neighbour op: no communication block
hipError_t launch_neighbours(const float* pos, const uint8_t* present, const float* rew, const SimParams& p,
                             const StepOut& out, hipStream_t stream);

// learn-side ops (learn_kernels.hip)
hipError_t launch_gae3(const float* rew, const float* val, const uint8_t* flags, int T, int M, int heads,
                       const double* gamma_host, double lam, float* adv, float* tgt, hipStream_t stream);
hipError_t launch_cc_fuse_mf(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                             const int32_t* cnt, int R, int N, int O, int A, int K, int counterfactual, float* cc,
                             hipStream_t stream);
hipError_t launch_cc_fuse_concat(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                                 const int32_t* cnt, int R, int N, int O, int A, int K, int num_neighbours,
                                 int counterfactual, float* cc, hipStream_t stream);
hipError_t launch_lcf_mix_partial(const float* adv, const float* nei_adv, const float* glob_adv, const float* lcf,
                                  const uint8_t* valid, int64_t B, float* mixed, double* stats, hipStream_t stream);
hipError_t launch_lcf_mix_apply(const float* mixed, const float* glob_adv, const uint8_t* valid, int64_t B,
                                const double* stats, float* norm_adv, float* glob_std, hipStream_t stream);

}  // namespace copo

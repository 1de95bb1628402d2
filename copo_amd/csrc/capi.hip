// C ABI of libcopo_hip.so (include/copo_hip.h): argument validation, handle lifetime, error strings.
// No torch types; all launches are asynchronous on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>

#include "sim_common.h"

using namespace copo;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(COPO_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct copo_sim {
    SimParams p;
    SimParams* p_dev;          // device copy of p (the kernels' parameter block)
    int device;
    int block;
    bool started;
    double lcf_mean, lcf_std, force_lcf;
    int capacity;              // active agent slots (curriculum), num_agents by default
    float lcf_host[4];         // {mean, std, capacity, 0}: what the kernels read from p.lcf_dist
    bool lcf_dirty;
    std::vector<void*> allocs;
};

extern "C" int copo_version(void) { return COPO_ABI_VERSION; }
#define COPO_STR2(x) #x
#define COPO_STR(x) COPO_STR2(x)
extern "C" const char* copo_build_info(void) {
#ifdef COPO_PROFILE_SKIP
    return "libcopo_hip ABI " COPO_STR(COPO_ABI_VERSION) ", gfx950, PROFILING build (COPO_PROFILE_SKIP=" COPO_STR(COPO_PROFILE_SKIP)
           "): phases may be compiled out, results may be wrong; reads COPO_ROWPASS_4X4 COPO_ROWPASS_RT8 COPO_WGRAD_OT COPO_FUSED_WGRAD COPO_FUSED_ROWPASS COPO_RP_DBG";
#else
    return "libcopo_hip ABI " COPO_STR(COPO_ABI_VERSION) ", gfx950, shipped build: all phases compiled in, no environment variable is read";
#endif
}
extern "C" const char* copo_last_error(void) { return g_err; }

template <typename T>
static int upload(copo_sim* s, const T* host, size_t count, const T** dev) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, count * sizeof(T) ? count * sizeof(T) : sizeof(T)));
    s->allocs.push_back(d);
    if (count) HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T*>(d);
    return COPO_OK;
}

// measured (scripts/bench_sim.py, 40 slots, populated scenes): up to one scene per CU -> 16 waves per scene; two per CU -> 8;
// then 4; from ~12 scenes per CU on, ONE wave per scene with the small LDS footprint (sim_shape_params, ~20 scenes resident
// per CU) and the register formulation of the neighbour lists (round 3: 4096 scenes 124 -> 110 us, 8192 222 -> 173 us; at
// 2048 four waves per scene still win, 72 vs 96 us)
// (`nbr_fast` = the register formulation of the neighbour lists can run, SimParams::nbr_fast: without it -- K > 8, a mean-field
// range of 0 or beyond the radius -- one wave per scene only pays above 8192 scenes, as before that formulation existed)
// Above that: the PACKED shape (sim_packed.hip: several scenes per workgroup, the per-agent phases dense over the lanes) wherever
// the configuration allows it (`packed` = scenes per workgroup, 0: not available), else one wave per scene.  Returned as -scenes.
static int pick_block(int E, bool nbr_fast = true, int packed = 0) {
    if (E <= 256) return 1024;
    if (E <= 512) return 512;
    if (E <= (nbr_fast ? 3072 : 8192)) return 256;
    return packed > 0 ? -packed : 64;
}
// (measured, 16 384 populated scenes, scripts/bench_sim.py, 4 scenes per workgroup against one wave per scene: 8 slots 62.7 / 66.0 us, 12 78.8 /
// 87.9, 16 90.9 / 99.6, 20 109.1 / 115.8, 10 slots x 240 beams 109 / 124 (fans of 5); with 40 slots the packed shape issues 9 % fewer vector
// instructions but ends behind, 209 vs 193 us: parked waves, barriers.  With detector beams (Bottleneck, 20 slots: 332 / 290 us) one wave
// per scene is ahead; 24 slots 127.9 / 134.2, 30 slots 154.7 / 161.3, 40 slots 207.9 / 204.4: automatic up to 32 slots on maps without
// detector beams)
static int packed_scenes(const SimParams& p) {
    return (p.N <= 32 && p.side_lasers == 0 && p.lane_lasers == 0 && sim_packed_supported(p)) ? sim_packed_default_scenes(p) : 0;
}

extern "C" int copo_sim_create(const copo_sim_cfg* cfg, int device, copo_sim** out) {
    if (!cfg || !out) return fail(COPO_ERR_NULL, "copo_sim_create: cfg/out is NULL");
    *out = nullptr;
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > COPO_MAX_AGENTS)
        return fail(COPO_ERR_DIM, "num_envs=%d num_agents=%d (agents must be 1..%d)", cfg->num_envs, cfg->num_agents,
                    COPO_MAX_AGENTS);
    if (cfg->num_lasers < 1 || cfg->num_lasers > COPO_MAX_LASERS)
        return fail(COPO_ERR_DIM, "num_lasers=%d out of 1..%d", cfg->num_lasers, COPO_MAX_LASERS);
    if (cfg->comm_size < 0 || cfg->comm_size > 64 || (cfg->comm_size > 0 && (cfg->comm_neighbours < 1 || cfg->comm_neighbours > COPO_MAX_AGENTS)))
        return fail(COPO_ERR_CONFIG, "comm_size=%d comm_neighbours=%d", cfg->comm_size, cfg->comm_neighbours);
    if (cfg->add_traffic_light && (cfg->traffic_light_interval < 1 || !(cfg->map_bbox[1] > cfg->map_bbox[0]) || !(cfg->map_bbox[3] > cfg->map_bbox[2])))
        return fail(COPO_ERR_CONFIG, "add_traffic_light needs traffic_light_interval >= 1 and a non-empty map_bbox");
    if (cfg->side_lasers < 0 || cfg->side_lasers > COPO_MAX_LASERS || cfg->lane_line_lasers < 0 || cfg->lane_line_lasers > COPO_MAX_LASERS)
        return fail(COPO_ERR_DIM, "side_lasers=%d lane_line_lasers=%d out of 0..%d", cfg->side_lasers, cfg->lane_line_lasers, COPO_MAX_LASERS);
    if ((cfg->navi_dim != 0 && cfg->navi_dim != COPO_NAVI_DIM) || (cfg->toll_dim != 0 && cfg->toll_dim != 2))
        return fail(COPO_ERR_CONFIG, "navi_dim=%d (0 or %d) toll_dim=%d (0 or 2)", cfg->navi_dim, COPO_NAVI_DIM, cfg->toll_dim);
    const int O = COPO_OBS_DIM(cfg);
    if (cfg->obs_dim != O) return fail(COPO_ERR_DIM, "obs_dim=%d but the configured blocks add up to %d (COPO_OBS_DIM)", cfg->obs_dim, O);
    if (cfg->nbr_k < 1 || cfg->nbr_k > COPO_MAX_AGENTS) return fail(COPO_ERR_DIM, "nbr_k=%d out of 1..64", cfg->nbr_k);
    if (cfg->n_routes < 1 || cfg->n_routes > COPO_MAX_ROUTES || cfg->n_spawns < cfg->num_agents ||
        cfg->n_spawns > COPO_MAX_SPAWNS)
        return fail(COPO_ERR_CONFIG, "n_routes=%d n_spawns=%d (need num_agents <= n_spawns <= %d)", cfg->n_routes,
                    cfg->n_spawns, COPO_MAX_SPAWNS);
    if (!cfg->route_segs || !cfg->route_meta || !cfg->spawn_tab || !cfg->spawn_s || !cfg->ray_cs)
        return fail(COPO_ERR_NULL, "copo_sim_create: a map table pointer is NULL");
    if (cfg->n_boxes < 0 || cfg->n_boxes > COPO_MAX_BOXES || (cfg->n_boxes > 0 && !cfg->boxes))
        return fail(COPO_ERR_CONFIG, "static boxes: n_boxes=%d (max %d) needs the table", cfg->n_boxes, COPO_MAX_BOXES);
    if (cfg->n_lines < 0 || cfg->n_lines > COPO_MAX_LINES ||
        ((cfg->side_lasers || cfg->lane_line_lasers) && (!cfg->lines || (cfg->side_lasers && !cfg->side_cs) || (cfg->lane_line_lasers && !cfg->lane_line_cs))))
        return fail(COPO_ERR_CONFIG, "detectors need the line table and their beam tables (n_lines=%d, max %d)", cfg->n_lines, COPO_MAX_LINES);
    if (cfg->substeps < 1 || cfg->horizon < 1 || cfg->horizon > 65535 || cfg->respawn_cooldown < 0 || cfg->respawn_cooldown > 255 ||
        cfg->delay_done < 0 || cfg->delay_done > 255)
        return fail(COPO_ERR_CONFIG, "substeps >= 1, 1 <= horizon <= 65535, 0 <= respawn_cooldown, delay_done <= 255");
    if (!(cfg->lcf_std > 0.0) || cfg->lcf_mean < -1.0 || cfg->lcf_mean > 1.0)
        return fail(COPO_ERR_CONFIG, "lcf_mean must be in [-1,1] and lcf_std > 0 (env_wrappers.py:195,425-426)");
    for (int r = 0; r < cfg->n_routes; ++r) {
        const int nseg = (int)cfg->route_meta[r * 4 + 1];
        if (nseg < 1 || nseg > COPO_MAX_SEGS) return fail(COPO_ERR_CONFIG, "route %d has %d roads", r, nseg);
        const int space = (int)cfg->route_meta[r * 4 + 3];
        if (space < 0 || space > 32) return fail(COPO_ERR_CONFIG, "route %d: exclusive destination id %d outside 0..32", r, space);
    }
    std::vector<int32_t> safe;
    for (int s = 0; s < cfg->n_spawns; ++s) {
        const int r0 = cfg->spawn_tab[s * 4], nc = cfg->spawn_tab[s * 4 + 1];
        if (r0 < 0 || nc < 1 || r0 + nc > cfg->n_routes) return fail(COPO_ERR_CONFIG, "spawn %d: bad route range", s);
        for (int r = r0; r < r0 + nc; ++r) {
            const float* g = cfg->route_segs + (size_t)r * (COPO_MAX_SEGS + 1) * COPO_SEG_STRIDE;
            if (g[5] != 0.0f || !(cfg->spawn_s[s] < g[4]) || cfg->spawn_tab[s * 4 + 2] < 0 || (float)cfg->spawn_tab[s * 4 + 2] >= floorf(g[COPO_SEG_LANES]))
                return fail(COPO_ERR_CONFIG, "spawn %d must lie on a lane of the straight first road of route %d", s, r);
        }
        if (cfg->spawn_tab[s * 4 + 3]) safe.push_back(s);
    }
    if (safe.empty() || safe.size() > COPO_MAX_SAFE)
        return fail(COPO_ERR_CONFIG, "%zu respawn places (spawn slots marked safe): need 1..%d", safe.size(), COPO_MAX_SAFE);
    copo_sim* s = new (std::nothrow) copo_sim();
    if (!s) return fail(COPO_ERR_DEVICE, "out of host memory");
    s->device = device;
    s->started = false;
    s->lcf_mean = cfg->lcf_mean;
    s->lcf_std = cfg->lcf_std;
    s->force_lcf = -100.0;
    s->capacity = cfg->num_agents;
    s->lcf_dirty = true;
    if (hipSetDevice(device) != hipSuccess) {
        delete s;
        return fail(COPO_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    }
    SimParams& p = s->p;
    memset(&p, 0, sizeof(p));
    p.E = cfg->num_envs; p.N = cfg->num_agents; p.O = cfg->obs_dim; p.K = cfg->nbr_k; p.num_lasers = cfg->num_lasers;
    p.enable_lcf = cfg->enable_lcf; p.horizon = cfg->horizon; p.delay_done = cfg->delay_done;
    p.respawn_cooldown = cfg->respawn_cooldown; p.substeps = cfg->substeps;
    p.n_routes = cfg->n_routes; p.n_spawns = cfg->n_spawns;
    {   // observation row: [side | state | lane | navigation | lasers | toll | traffic light | lcf | messages]
        p.side_lasers = cfg->side_lasers; p.lane_lasers = cfg->lane_line_lasers; p.navi_dim = cfg->navi_dim;
        p.toll_dim = cfg->toll_dim; p.toll_min_steps = cfg->toll_min_steps;
        p.toll_speed_limit = cfg->toll_speed_limit; p.overspeed_penalty = cfg->overspeed_penalty; p.toll_early_exit = cfg->toll_early_exit;
        p.col_state = COPO_SIDE_DIM(cfg);
        p.col_lane = p.col_state + COPO_STATE_DIM;
        p.col_navi = p.col_lane + COPO_LANE_DIM(cfg);
        p.col_lidar = p.col_navi + cfg->navi_dim;
        int col = p.col_lidar + cfg->num_lasers;
        p.col_toll = cfg->toll_dim ? col : -1;
        col += cfg->toll_dim;
        p.col_tl = cfg->add_traffic_light ? col : -1;
        col += cfg->add_traffic_light ? 3 : 0;
        p.col_lcf = cfg->enable_lcf ? col : -1;
        col += cfg->enable_lcf ? 1 : 0;
        p.col_comm = cfg->comm_size > 0 ? col : -1;
        p.act_dim = COPO_ACT_DIM(cfg);
        p.tl_interval = cfg->traffic_light_interval > 0 ? cfg->traffic_light_interval : 1;
        p.comm_size = cfg->comm_size > 0 ? cfg->comm_size : 0;
        p.comm_nb = cfg->comm_size > 0 ? cfg->comm_neighbours : 0;
        p.comm_pos = cfg->add_pos_in_comm ? 1 : 0;
        for (int k = 0; k < 4; ++k) p.bbox[k] = cfg->map_bbox[k];
    }
    p.lidar_range = cfg->lidar_range; p.neighbours_distance = cfg->neighbours_distance; p.mf_distance = cfg->mf_distance;
    {   // neighbours_fast (sim_kernels.hip): conservative fp32 thresholds around the exact fp64 decisions
        const double R2 = (double)cfg->neighbours_distance * (double)cfg->neighbours_distance;
        const double M2 = (double)cfg->mf_distance * (double)cfg->mf_distance;
        p.nbr_r2lo = (float)(R2 * (1.0 - 1e-6));
        p.nbr_r2hi = (float)(R2 * (1.0 + 1e-6));
        const float mlo = (float)(M2 * (1.0 - 1e-5)), mhi = (float)(M2 * (1.0 + 1e-6));
        uint32_t blo, bhi;
        memcpy(&blo, &mlo, 4);
        memcpy(&bhi, &mhi, 4);
        p.mf_key_lo = blo & ~63u;                    // key < lo: inside for certain (keys drop 6 mantissa bits of d^2)
        p.mf_key_hi = (bhi + 63u) & ~63u;            // key >= hi: outside for certain
        // (a mean-field range of exactly 0 stays with the exact formulation: a coincident pair has d = 0 <= 0 there, and the
        // key thresholds of a zero range cannot express it)
        p.nbr_fast = (cfg->nbr_k <= 8 && cfg->neighbours_distance > 0.0f && cfg->mf_distance > 0.0f &&
                      cfg->mf_distance < 0.99f * cfg->neighbours_distance) ? 1 : 0;
    }
    p.dt = cfg->dt; p.hl = cfg->veh_half_len; p.hw = cfg->veh_half_wid; p.wheelbase = cfg->wheelbase;
    p.max_steer = cfg->max_steer; p.max_speed = cfg->max_speed; p.acc_max = cfg->acc_max; p.brake_gain = cfg->brake_gain;
    p.brake_max = cfg->brake_max; p.lat_acc_max = cfg->lat_acc_max; p.reverse_acc = cfg->reverse_acc;
    p.region_hl = 0.5f * cfg->spawn_region_len; p.region_hw = 0.5f * cfg->spawn_region_wid;
    p.driving_reward = cfg->driving_reward; p.speed_reward = cfg->speed_reward; p.success_reward = cfg->success_reward;
    p.crash_penalty = cfg->crash_penalty; p.out_penalty = cfg->out_penalty; p.arrive_margin = cfg->arrive_margin; p.body_margin = cfg->body_margin;
    p.lane_width = cfg->lane_width;
    p.side_range = cfg->side_range; p.lane_range = cfg->lane_line_range;
    // derived constants: single float operations (this file is compiled with -ffp-contract=off), as in the oracle
    p.inv_w = 1.0f / cfg->lane_width;
    p.inv_range = 1.0f / cfg->lidar_range;
    p.inv_vnorm = 1.0f / (cfg->max_speed * 3.6f + 1.0f);
    p.inv_dt = 1.0f / cfg->dt;
    p.inv_side_range = cfg->side_lasers ? 1.0f / cfg->side_range : 0.0f;
    p.inv_lane_range = cfg->lane_line_lasers ? 1.0f / cfg->lane_line_range : 0.0f;
    p.inv_toll = cfg->toll_dim ? 1.0f / (float)(cfg->toll_min_steps > 0 ? cfg->toll_min_steps : 1) : 0.0f;
    p.h_sub = cfg->dt / (float)cfg->substeps;
    p.ray_sign = (cfg->num_lasers > 2 && cfg->ray_cs[3] < 0.0f) ? -1.0f : 1.0f;   // the beam table's sense of rotation
    // detector beams: evenly spaced?  (then a line primitive only meets a WINDOW of them, detector_window in sim_device.h)
    auto even_table = [](const float* cs, int n, float& theta0, float& rpr) {
        theta0 = 0.0f; rpr = 0.0f;
        if (n < 3 || !cs) return;
        const double two_pi = 6.283185307179586;
        const double th0 = std::atan2((double)cs[1], (double)cs[0]);
        double step = std::atan2((double)cs[3], (double)cs[2]) - th0;
        step = step > two_pi / 2 ? step - two_pi : (step < -two_pi / 2 ? step + two_pi : step);
        bool even = std::fabs(std::fabs(step) * n - two_pi) < 1e-3;
        for (int k = 0; k < n && even; ++k) {
            const double want = th0 + k * step;
            even = std::fabs(cs[2 * k] - std::cos(want)) < 1e-4 && std::fabs(cs[2 * k + 1] - std::sin(want)) < 1e-4;
        }
        if (even) { theta0 = (float)th0; rpr = (float)(1.0 / step); }
    };
    even_table(cfg->side_cs, cfg->side_lasers, p.side_theta0, p.side_rpr);
    even_table(cfg->lane_line_cs, cfg->lane_line_lasers, p.lane_theta0, p.lane_rpr);
    p.n_safe = (int32_t)safe.size();
    p.n_spaces = 0;
    for (int r = 0; r < cfg->n_routes; ++r) {
        const int d = (int)cfg->route_meta[r * 4 + 3];
        p.n_spaces = std::max(p.n_spaces, (int32_t)d);
    }
    p.n_lines = (cfg->side_lasers || cfg->lane_line_lasers) ? cfg->n_lines : 0;
    int rc = COPO_OK;
    const size_t EN = (size_t)p.E * p.N;
    void* d = nullptr;
    auto dev_alloc = [&](size_t bytes, void** ptr) -> int {
        hipError_t e = hipMalloc(ptr, bytes);
        if (e != hipSuccess) return fail(COPO_ERR_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        s->allocs.push_back(*ptr);
        e = hipMemset(*ptr, 0, bytes);
        if (e != hipSuccess) return fail(COPO_ERR_DEVICE, "hipMemset: %s", hipGetErrorString(e));
        return COPO_OK;
    };
    if (rc == COPO_OK && (rc = dev_alloc(COPO_STATE_FIELDS * EN * 4, &d)) == COPO_OK) p.state = (float*)d;
    if (rc == COPO_OK && (rc = dev_alloc((size_t)p.E * 16, &d)) == COPO_OK) p.env = (int32_t*)d;
    if (rc == COPO_OK && (rc = dev_alloc((size_t)p.E * 8, &d)) == COPO_OK) p.seeds = (const uint64_t*)d;
    if (rc == COPO_OK && (rc = dev_alloc(16, &d)) == COPO_OK) p.lcf_dist = (const float*)d;
    if (rc == COPO_OK) {      // device table: only as many road records per route as the longest route needs (+ its terminal record)
        int rows = 2;
        for (int r = 0; r < cfg->n_routes; ++r) rows = std::max(rows, (int)cfg->route_meta[r * 4 + 1] + 1);
        p.seg_rows = rows;
        std::vector<float> compact((size_t)cfg->n_routes * rows * COPO_SEG_STRIDE);
        for (int r = 0; r < cfg->n_routes; ++r)
            memcpy(compact.data() + (size_t)r * rows * COPO_SEG_STRIDE,
                   cfg->route_segs + (size_t)r * (COPO_MAX_SEGS + 1) * COPO_SEG_STRIDE, sizeof(float) * rows * COPO_SEG_STRIDE);
        rc = upload(s, compact.data(), compact.size(), &p.route_segs);
    }
    if (rc == COPO_OK) rc = upload(s, cfg->route_meta, (size_t)cfg->n_routes * 4, &p.route_meta);
    if (rc == COPO_OK) rc = upload(s, cfg->spawn_tab, (size_t)cfg->n_spawns * 4, &p.spawn_tab);
    if (rc == COPO_OK) rc = upload(s, cfg->spawn_s, (size_t)cfg->n_spawns, &p.spawn_s);
    if (rc == COPO_OK) rc = upload(s, cfg->ray_cs, (size_t)cfg->num_lasers * 2, &p.ray_cs);
    if (rc == COPO_OK) rc = upload(s, safe.data(), safe.size(), &p.safe_ids);
    if (rc == COPO_OK) {      // pose of every respawn place (sim_kernels.hip spawn_pose, the same float operations in the same order)
        std::vector<float> sp4(4 * std::max<size_t>(safe.size(), 1), 0.0f);
        for (size_t q = 0; q < safe.size(); ++q) {
            const int sp = safe[q];
            const float* g = cfg->route_segs + (size_t)cfg->spawn_tab[sp * 4 + 0] * (COPO_MAX_SEGS + 1) * COPO_SEG_STRIDE;
            const float s0 = cfg->spawn_s[sp];
            const float off = (float)cfg->spawn_tab[sp * 4 + 2] * cfg->lane_width;
            sp4[4 * q + 0] = g[0] + g[2] * s0 + g[3] * off;
            sp4[4 * q + 1] = g[1] + g[3] * s0 - g[2] * off;
            sp4[4 * q + 2] = g[2];
            sp4[4 * q + 3] = g[3];
        }
        rc = upload(s, sp4.data(), sp4.size(), &p.safe_pose);
    }
    if (rc == COPO_OK && p.n_lines) rc = upload(s, cfg->lines, (size_t)cfg->n_lines * COPO_LINE_STRIDE, &p.lines);
    p.n_boxes = cfg->n_boxes;
    p.n_boxes_lidar = cfg->boxes_hidden ? 0 : cfg->n_boxes;
    if (rc == COPO_OK && p.n_boxes) rc = upload(s, cfg->boxes, (size_t)cfg->n_boxes * COPO_BOX_STRIDE, &p.boxes);
    if (rc == COPO_OK && cfg->side_lasers) rc = upload(s, cfg->side_cs, (size_t)cfg->side_lasers * 2, &p.side_cs);
    if (rc == COPO_OK && cfg->lane_line_lasers) rc = upload(s, cfg->lane_line_cs, (size_t)cfg->lane_line_lasers * 2, &p.lane_cs);
    s->block = pick_block(cfg->num_envs, p.nbr_fast != 0, packed_scenes(p));      // (after the observation layout: the packed shape depends on it)
    sim_shape_params(p, s->block);
    if (rc == COPO_OK) {
        const SimParams* pd = nullptr;
        rc = upload(s, &s->p, 1, &pd);
        s->p_dev = const_cast<SimParams*>(pd);
    }
    if (rc != COPO_OK) {
        for (void* a : s->allocs) (void)hipFree(a);
        delete s;
        return rc;
    }
    *out = s;
    return COPO_OK;
}

extern "C" int copo_sim_destroy(copo_sim* s) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_destroy: NULL handle");
    (void)hipSetDevice(s->device);
    for (void* a : s->allocs) (void)hipFree(a);
    delete s;
    return COPO_OK;
}

// Push the LCF distribution to device memory on `st` (kernels read it from there, so launches captured
// in a hipGraph keep seeing later updates).  Skipped while the stream is capturing.
static int flush_lcf(copo_sim* s, hipStream_t st) {
    if (!s->lcf_dirty) return COPO_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return COPO_OK;
    s->lcf_host[0] = (float)((s->force_lcf != -100.0) ? s->force_lcf : s->lcf_mean);
    s->lcf_host[1] = (float)s->lcf_std;
    s->lcf_host[2] = (float)s->capacity;
    s->lcf_host[3] = 0.0f;
    HIP_TRY(hipMemcpyAsync(const_cast<float*>(s->p.lcf_dist), s->lcf_host, 16, hipMemcpyHostToDevice, st));
    s->lcf_dirty = false;
    return COPO_OK;
}

extern "C" int copo_sim_flush(copo_sim* s, void* stream) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_flush: NULL handle");
    return flush_lcf(s, static_cast<hipStream_t>(stream));
}

extern "C" int copo_sim_set_lcf_dist(copo_sim* s, double mean, double std) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_lcf_dist: NULL handle");
    if (!(std > 0.0) || mean < -1.0 || mean > 1.0)
        return fail(COPO_ERR_CONFIG, "set_lcf_dist(mean=%g, std=%g): need -1 <= mean <= 1, std > 0", mean, std);
    s->lcf_mean = mean;
    s->lcf_std = std;
    s->lcf_dirty = true;
    return COPO_OK;
}

extern "C" int copo_sim_set_capacity(copo_sim* s, int32_t capacity) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_capacity: NULL handle");
    if (capacity < 1 || capacity > s->p.N) return fail(COPO_ERR_CONFIG, "capacity=%d not in [1, %d]", capacity, s->p.N);
    s->capacity = capacity;
    s->lcf_dirty = true;
    return COPO_OK;
}

extern "C" int copo_sim_set_force_lcf(copo_sim* s, double v) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_force_lcf: NULL handle");
    if (v != -100.0 && (v < -1.0 || v > 1.0)) return fail(COPO_ERR_CONFIG, "force_lcf=%g not in [-1,1] (or -100)", v);
    s->force_lcf = v;
    s->lcf_dirty = true;
    return COPO_OK;
}

extern "C" int copo_sim_set_debug(copo_sim* s, int64_t* stamps) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_debug: NULL handle");
    s->p.dbg = reinterpret_cast<long long*>(stamps);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());         // profiling aid: launches in flight keep the block they started with
    HIP_TRY(hipMemcpy(s->p_dev, &s->p, sizeof(SimParams), hipMemcpyHostToDevice));
    return COPO_OK;
}

extern "C" int copo_sim_set_block(copo_sim* s, int32_t threads) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_block: NULL handle");
    if (threads == 0) threads = pick_block(s->p.E, s->p.nbr_fast != 0, packed_scenes(s->p));
    if (threads < 0) {        // packed shape: -threads scenes per workgroup (-1: the default count)
        if (!sim_packed_supported(s->p)) return fail(COPO_ERR_CONFIG, "the packed launch shape needs the register formulation of the neighbour lists and no traffic-light / communication block");
        if (threads == -1) threads = -sim_packed_default_scenes(s->p);
        if (threads < -16 || threads > -2 || sim_packed_lds_bytes(s->p, -threads) > 96 * 1024)
            return fail(COPO_ERR_DIM, "block=%d: 2..16 scenes per workgroup within 96 KB of LDS", threads);
    } else if (threads != 64 && threads != 128 && threads != 256 && threads != 512 && threads != 1024)
        return fail(COPO_ERR_DIM, "block=%d must be 64/128/256/512/1024, or -scenes for the packed shape", threads);
    if (threads != s->block) {
        s->block = threads;
        sim_shape_params(s->p, threads);
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());         // launches in flight keep the shape they started with
        HIP_TRY(hipMemcpy(s->p_dev, &s->p, sizeof(SimParams), hipMemcpyHostToDevice));
    }
    return COPO_OK;
}

extern "C" int copo_sim_set_chunk(copo_sim* s, int32_t fans) {
    if (!s) return fail(COPO_ERR_NULL, "copo_sim_set_chunk: NULL handle");
    if (fans < 0 || fans > COPO_MAX_AGENTS) return fail(COPO_ERR_DIM, "fans=%d must be 0..%d", fans, COPO_MAX_AGENTS);
    if (fans != s->p.chunk_one_wave) {
        const int32_t before = s->p.chunk_one_wave;
        s->p.chunk_one_wave = fans;
        if (s->block < 0 && sim_packed_lds_bytes(s->p, -s->block) > 96 * 1024) {      // packed shape: the workgroup's scenes must still fit
            s->p.chunk_one_wave = before;
            return fail(COPO_ERR_DIM, "fans=%d: %d scenes per workgroup would need more than 96 KB of LDS", fans, -s->block);
        }
        sim_shape_params(s->p, s->block);
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());         // launches in flight keep the shape they started with
        HIP_TRY(hipMemcpy(s->p_dev, &s->p, sizeof(SimParams), hipMemcpyHostToDevice));
    }
    return COPO_OK;
}

extern "C" int copo_sim_reset(copo_sim* s, const uint64_t* seeds, const copo_step_out* out, void* stream) {
    if (!s || !seeds || !out) return fail(COPO_ERR_NULL, "copo_sim_reset: NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(const_cast<uint64_t*>(s->p.seeds), seeds, (size_t)s->p.E * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(s->p.state, 0, COPO_STATE_FIELDS * (size_t)s->p.E * s->p.N * 4, st));
    int rc = flush_lcf(s, st);
    if (rc != COPO_OK) return rc;
    HIP_TRY(launch_sim_reset(s->p, s->p_dev, *out, s->block, st));
    s->started = true;
    return COPO_OK;
}

extern "C" int copo_sim_step(copo_sim* s, const float* act, const copo_step_out* out, void* stream) {
    if (!s || !act || !out) return fail(COPO_ERR_NULL, "copo_sim_step: NULL argument");
    if (!s->started) return fail(COPO_ERR_STATE, "copo_sim_step before copo_sim_reset");
    int rc = flush_lcf(s, static_cast<hipStream_t>(stream));
    if (rc != COPO_OK) return rc;
    HIP_TRY(launch_sim_step(s->p, s->p_dev, act, *out, s->block, static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

extern "C" int copo_sim_get_state(copo_sim* s, float* slot_state, int32_t* env_state, void* stream) {
    if (!s || !slot_state || !env_state) return fail(COPO_ERR_NULL, "copo_sim_get_state: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(slot_state, s->p.state, COPO_STATE_FIELDS * (size_t)s->p.E * s->p.N * 4,
                           hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(env_state, s->p.env, (size_t)s->p.E * 16, hipMemcpyDeviceToDevice, st));
    return COPO_OK;
}

extern "C" int copo_sim_set_state(copo_sim* s, const float* slot_state, const int32_t* env_state, void* stream) {
    if (!s || !slot_state || !env_state) return fail(COPO_ERR_NULL, "copo_sim_set_state: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(s->p.state, slot_state, COPO_STATE_FIELDS * (size_t)s->p.E * s->p.N * 4,
                           hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->p.env, env_state, (size_t)s->p.E * 16, hipMemcpyDeviceToDevice, st));
    s->started = true;
    return COPO_OK;
}

// ---- stateless ops --------------------------------------------------------------------------------

extern "C" int copo_neighbours_f32(const float* pos, const uint8_t* present, const float* rew, int32_t E, int32_t N,
                                   int32_t K, float radius, float mf_distance, int32_t* nbr_idx, int32_t* nbr_cnt,
                                   int32_t* mf_cnt, float* nbr_dist, float* nei_rew, float* glob_rew, void* stream) {
    if (!pos || !present) return fail(COPO_ERR_NULL, "copo_neighbours_f32: pos/present is NULL");
    if (E < 0 || N < 1 || N > COPO_MAX_AGENTS || K < 1 || K > COPO_MAX_AGENTS)
        return fail(COPO_ERR_DIM, "copo_neighbours_f32: E=%d N=%d K=%d (N,K in 1..64)", E, N, K);
    if (E == 0) return COPO_OK;
    SimParams p;
    memset(&p, 0, sizeof(p));
    p.E = E; p.N = N; p.K = K;
    p.lists_for_absent = 1;
    p.neighbours_distance = radius;
    p.mf_distance = mf_distance;
    StepOut out;
    memset(&out, 0, sizeof(out));
    out.nbr_idx = nbr_idx; out.nbr_cnt = nbr_cnt; out.mf_cnt = mf_cnt; out.nbr_dist = nbr_dist;
    out.nei_rew = rew ? nei_rew : nullptr;
    out.glob_rew = rew ? glob_rew : nullptr;
    HIP_TRY(launch_neighbours(pos, present, rew, p, out, static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

extern "C" int copo_gae3_f32(const float* rew, const float* val, const uint8_t* flags, int32_t T, int32_t M,
                             int32_t heads, const double* gamma, double lam, float* adv, float* tgt, void* stream) {
    if (!rew || !val || !flags || !gamma || !adv || !tgt) return fail(COPO_ERR_NULL, "copo_gae3_f32: NULL argument");
    if (T < 0 || M < 0 || heads < 1 || heads > 4) return fail(COPO_ERR_DIM, "copo_gae3_f32: T=%d M=%d heads=%d", T, M, heads);
    if (T == 0 || M == 0) return COPO_OK;
    HIP_TRY(launch_gae3(rew, val, flags, T, M, heads, gamma, lam, adv, tgt, static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

static int check_fuse(const char* name, const void* obs, const void* act, const void* flags, const void* idx,
                      const void* cnt, const void* cc, int R, int N, int O, int A, int K) {
    if (!obs || !act || !flags || !idx || !cnt || !cc) return fail(COPO_ERR_NULL, "%s: NULL argument", name);
    if (R < 0 || N < 1 || N > COPO_MAX_AGENTS || O < 1 || A < 0 || K < 1 || K > COPO_MAX_AGENTS)
        return fail(COPO_ERR_DIM, "%s: R=%d N=%d O=%d A=%d K=%d", name, R, N, O, A, K);
    return COPO_OK;
}

extern "C" int copo_cc_fuse_mf_f32(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                                   const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                                   int32_t counterfactual, float* cc_obs, void* stream) {
    int rc = check_fuse("copo_cc_fuse_mf_f32", obs, act, flags, nbr_idx, cnt, cc_obs, R, N, O, A, K);
    if (rc != COPO_OK || R == 0) return rc;
    HIP_TRY(launch_cc_fuse_mf(obs, act, flags, nbr_idx, cnt, R, N, O, A, K, counterfactual, cc_obs,
                              static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

extern "C" int copo_cc_fuse_concat_f32(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                                       const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                                       int32_t num_neighbours, int32_t counterfactual, float* cc_obs, void* stream) {
    int rc = check_fuse("copo_cc_fuse_concat_f32", obs, act, flags, nbr_idx, cnt, cc_obs, R, N, O, A, K);
    if (rc != COPO_OK || R == 0) return rc;
    if (num_neighbours < 0 || num_neighbours > COPO_MAX_AGENTS)
        return fail(COPO_ERR_DIM, "copo_cc_fuse_concat_f32: num_neighbours=%d", num_neighbours);
    HIP_TRY(launch_cc_fuse_concat(obs, act, flags, nbr_idx, cnt, R, N, O, A, K, num_neighbours, counterfactual, cc_obs,
                                  static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

extern "C" int copo_lcf_mix_partial_f32(const float* adv, const float* nei_adv, const float* glob_adv, const float* lcf,
                                        const uint8_t* valid, int64_t B, float* mixed, double* stats, void* stream) {
    if (!adv || !nei_adv || !glob_adv || !lcf || !mixed || !stats)
        return fail(COPO_ERR_NULL, "copo_lcf_mix_partial_f32: NULL argument");
    if (B < 0) return fail(COPO_ERR_DIM, "copo_lcf_mix_partial_f32: B=%lld", (long long)B);
    HIP_TRY(launch_lcf_mix_partial(adv, nei_adv, glob_adv, lcf, valid, B, mixed, stats, static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

extern "C" int copo_lcf_mix_apply_f32(const float* mixed, const float* glob_adv, const uint8_t* valid, int64_t B,
                                      const double* stats, float* norm_adv, float* glob_adv_std, void* stream) {
    if (!mixed || !glob_adv || !stats || !norm_adv || !glob_adv_std)
        return fail(COPO_ERR_NULL, "copo_lcf_mix_apply_f32: NULL argument");
    if (B < 0) return fail(COPO_ERR_DIM, "copo_lcf_mix_apply_f32: B=%lld", (long long)B);
    if (B == 0) return COPO_OK;
    HIP_TRY(launch_lcf_mix_apply(mixed, glob_adv, valid, B, stats, norm_adv, glob_adv_std, static_cast<hipStream_t>(stream)));
    return COPO_OK;
}

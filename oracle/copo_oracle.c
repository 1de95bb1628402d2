/*
 * copo_oracle.c -- CPU restatement (scalar C) of the CoPO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (copo_amd/, libcopo_hip.so) never does.
 *
 * Two halves:
 *  (1) wrapper / learn-side arithmetic that restates the reference line by line and is pinned by the
 *      golden vectors under tests/golden/ (generated from the reference's own functions):
 *        oracle_neighbours      <- utils/env_wrappers.py:125-158 (CCEnv) + :313-326 (LCFEnv rewards)
 *        oracle_gae3            <- algo_ccppo.py:362-373, algo_copo.py:189-204,492-500
 *        oracle_cc_fuse_mf      <- algo_ccppo.py:266-311
 *        oracle_cc_fuse_concat  <- algo_ccppo.py:225-263
 *        oracle_lcf_mix         <- algo_copo.py:539-551
 *  (2) the simulator step.  MetaDrive 0.2.5 (the reference's simulator, README.md:42) is a pip
 *      dependency whose source is NOT in the reference tree, so this half restates MetaDrive's published
 *      multi-agent semantics as written down in DESIGN.md section 3 (observation columns, navigation,
 *      spawn / respawn, reward, termination, episode structure; a kinematic bicycle stands in for
 *      Bullet's raycast vehicle).  Parity status: pinned BEHAVIOURALLY -- the populations the reference
 *      trained in MetaDrive (weights held as data under tests/golden/) must drive these scenes at the
 *      levels the reference's own evaluation CSVs / training table / progress.csv record
 *      (tests/test_oracle_golden.py, tests/test_gpu_reference_populations.py) -- not function by
 *      function: no MetaDrive function output exists in the reference tree to compare with.
 *      It is the canonical definition the HIP kernel is checked
 *      against, bit for bit: all float math is +,-,*,/,sqrt on IEEE fp32/fp64 with contraction off
 *      and hand-written polynomials for sin/cos/atan2/log, so CPU and GPU agree exactly.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/copo_hip.h"

/* ------------------------------------------------------------------------------------------------
 * deterministic math (spec: DESIGN.md section 3.2)
 * ---------------------------------------------------------------------------------------------- */
/* Round 6: the hot expressions of the spec are stated with explicit FUSED multiply-adds (one rounding), the same ones in
 * the HIP kernels (`fm` in csrc/sim_math.h = v_fma_f32): a rotation, dot or cross product is a multiply + an fma instead of
 * two multiplies + an add.  fmaf() is exact-then-rounded on every IEEE machine (glibc's software path included), so HIP ==
 * oracle stays bit for bit; -ffp-contract=off still keeps the compiler from fusing anything that is not written as fm(). */
static inline float fm(float a, float b, float c) { return fmaf(a, b, c); }
#define PI_F 3.14159265f
#define TWO_PI_F 6.28318531f
#define HALF_PI_F 1.57079633f

static void o_sincosf(float x, float* s, float* c) {
    float kf = floorf(fm(x, 0.636619772f, 0.5f));
    int k = (int)kf;
    float r = fm(-kf, 1.5703125f, x);
    r = fm(-kf, 4.83751297e-4f, r);
    r = fm(-kf, 7.54978996e-8f, r);
    float z = r * r;
    float sp = fm(fm(fm(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float cp = fm(z, fm(z, fm(fm(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), -0.5f), 1.0f);
    switch (k & 3) {
        case 0: *s = sp; *c = cp; break;
        case 1: *s = cp; *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}

static float o_atan2f(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    /* one division: tan(a - pi/4) = (mn - mx) / (mn + mx) above tan(pi/8), mn / mx below */
    int hi = mn > 0.414213562f * mx;
    float t = (hi ? mn - mx : mn) / (hi ? mn + mx : mx);
    float off = hi ? 0.785398163f : 0.0f;
    float z = t * t;
    float p = fm(fm(fm(fm(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f) * z, t, t);
    float r = off + p;
    if (ay > ax) r = HALF_PI_F - r;
    if (x < 0.0f) r = PI_F - r;
    return y < 0.0f ? -r : r;
}

static float o_logf(float u) { /* u in (0, 1] normal */
    uint32_t b;
    memcpy(&b, &u, 4);
    int e = (int)(b >> 23) - 127;
    b = (b & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &b, 4);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = (((0.111111111f * z + 0.142857143f) * z + 0.2f) * z + 0.333333333f) * z + 1.0f;
    return (float)e * 0.693147181f + 2.0f * s * p;
}

static float o_wrap_pi(float a) {
    if (a > PI_F) a -= TWO_PI_F;
    if (a < -PI_F) a += TWO_PI_F;
    return a;
}

static float o_clip(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static uint32_t o_mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

static uint32_t o_hash(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t stream) {
    uint32_t h = o_mix32((uint32_t)seed + 0x9E3779B9u);
    h = o_mix32(h ^ (uint32_t)(seed >> 32));
    h = o_mix32(h ^ a);
    h = o_mix32(h ^ b);
    h = o_mix32(h ^ c);
    h = o_mix32(h ^ stream);
    return h;
}

static float o_uniform(uint32_t h) { return ((float)(h >> 9) + 0.5f) * 1.1920929e-7f; /* 2^-23 */ }

enum { ST_EMPTY = 0, ST_ALIVE = 1, ST_WRECK = 2 };
enum { RNG_ROUTE = 1, RNG_LCF1 = 2, RNG_LCF2 = 3, RNG_SPAWN = 4, RNG_PERM = 16 };

/* state field indices: [COPO_STATE_FIELDS][E][N] 32-bit words */
enum {
    S_X = 0, S_Y, S_TH, S_V, S_STEER /* action of this step */, S_THROTTLE, S_PSTEER /* action of the step before */,
    S_PTHROTTLE, S_YAWRATE, S_PROG, S_LCF, S_EPREW, S_ROUTE /* route | road << 16 */,
    S_STATUS /* status | timer << 8 | age << 16 */, S_AID, S_SPAWNCNT /* spawn count | toll wait << 16 */
};
#define ST_STATUS(w) ((w) & 0xff)
#define ST_TIMER(w) (((w) >> 8) & 0xff)
#define ST_AGE(w) ((int)((uint32_t)(w) >> 16))
#define ST_PACK(st, tm, age) ((int32_t)((uint32_t)(st) | ((uint32_t)(tm) << 8) | ((uint32_t)(age) << 16)))
/* env words: [E][4] int32 = {t_env, episode, next_aid, started} */

typedef struct oracle_sim {
    copo_sim_cfg cfg;
    float* route_segs;
    float* route_meta;
    int32_t* spawn_tab;
    float* spawn_s;
    float* ray_cs;
    float* lines;
    float* boxes;              /* [n_boxes][COPO_BOX_STRIDE] static boxes (buildings) */
    float* st;      /* [16][E][N] */
    int32_t* env;   /* [E][4] */
    uint64_t* seeds;
    double lcf_mean, lcf_std, force_lcf;
    int capacity;              /* active agent slots (curriculum); num_agents by default */
    /* constants derived once, in float, exactly as the kernel derives them */
    float inv_w, inv_range, inv_vnorm, inv_dt, inv_side_range, inv_lane_range, inv_toll;
    int safe_ids[COPO_MAX_SAFE];
    int n_safe;
    int n_spaces;              /* exclusive destinations (route_meta[.][3] = id + 1), 0 = none; at most 32 */
} oracle_sim;

static float* FP(oracle_sim* s, int f, int e) { return s->st + ((size_t)f * s->cfg.num_envs + e) * s->cfg.num_agents; }
static int32_t* IP(oracle_sim* s, int f, int e) { return (int32_t*)FP(s, f, e); }

static void* dup_mem(const void* p, size_t n) {
    void* q = malloc(n ? n : 1);
    if (n) memcpy(q, p, n);
    return q;
}

int oracle_sim_destroy(oracle_sim* s);

int oracle_sim_create(const copo_sim_cfg* cfg, oracle_sim** out) {
    if (!cfg || !out) return COPO_ERR_NULL;
    if (cfg->num_agents < 1 || cfg->num_agents > COPO_MAX_AGENTS || cfg->num_envs < 1) return COPO_ERR_DIM;
    if (cfg->num_lasers < 1 || cfg->num_lasers > COPO_MAX_LASERS) return COPO_ERR_DIM;
    if (cfg->side_lasers < 0 || cfg->side_lasers > COPO_MAX_LASERS || cfg->lane_line_lasers < 0 || cfg->lane_line_lasers > COPO_MAX_LASERS) return COPO_ERR_DIM;
    if (cfg->navi_dim != 0 && cfg->navi_dim != COPO_NAVI_DIM) return COPO_ERR_CONFIG;
    if (cfg->toll_dim != 0 && cfg->toll_dim != 2) return COPO_ERR_CONFIG;
    if (cfg->obs_dim != COPO_OBS_DIM(cfg)) return COPO_ERR_DIM;
    if (cfg->comm_size < 0 || (cfg->comm_size > 0 && (cfg->comm_neighbours < 1 || cfg->comm_neighbours > COPO_MAX_AGENTS))) return COPO_ERR_CONFIG;
    if (cfg->add_traffic_light && (cfg->traffic_light_interval < 1 || !(cfg->map_bbox[1] > cfg->map_bbox[0]) || !(cfg->map_bbox[3] > cfg->map_bbox[2]))) return COPO_ERR_CONFIG;
    if (cfg->n_spawns < cfg->num_agents || cfg->n_spawns > COPO_MAX_SPAWNS || cfg->nbr_k < 1 || cfg->nbr_k > COPO_MAX_AGENTS) return COPO_ERR_CONFIG;
    if (cfg->n_lines < 0 || cfg->n_lines > COPO_MAX_LINES || ((cfg->side_lasers || cfg->lane_line_lasers) && !cfg->lines)) return COPO_ERR_CONFIG;
    if (cfg->delay_done > 255 || cfg->respawn_cooldown > 255 || cfg->horizon > 65535) return COPO_ERR_CONFIG;
    oracle_sim* s = (oracle_sim*)calloc(1, sizeof(oracle_sim));
    s->cfg = *cfg;
    size_t E = cfg->num_envs, N = cfg->num_agents;
    s->route_segs = dup_mem(cfg->route_segs, sizeof(float) * cfg->n_routes * (COPO_MAX_SEGS + 1) * COPO_SEG_STRIDE);
    s->route_meta = dup_mem(cfg->route_meta, sizeof(float) * cfg->n_routes * 4);
    s->spawn_tab = dup_mem(cfg->spawn_tab, sizeof(int32_t) * cfg->n_spawns * 4);
    s->spawn_s = dup_mem(cfg->spawn_s, sizeof(float) * cfg->n_spawns);
    s->ray_cs = dup_mem(cfg->ray_cs, sizeof(float) * cfg->num_lasers * 2);
    s->lines = dup_mem(cfg->lines, sizeof(float) * cfg->n_lines * COPO_LINE_STRIDE);
    s->boxes = cfg->n_boxes > 0 ? dup_mem(cfg->boxes, sizeof(float) * cfg->n_boxes * COPO_BOX_STRIDE) : NULL;
    s->st = (float*)calloc(COPO_STATE_FIELDS * E * N, 4);
    s->env = (int32_t*)calloc(E * 4, 4);
    s->seeds = (uint64_t*)calloc(E, 8);
    s->lcf_mean = cfg->lcf_mean;
    s->lcf_std = cfg->lcf_std;
    s->force_lcf = -100.0;
    s->capacity = cfg->num_agents;
    s->inv_w = 1.0f / cfg->lane_width;
    s->inv_range = 1.0f / cfg->lidar_range;
    s->inv_vnorm = 1.0f / (cfg->max_speed * 3.6f + 1.0f);
    s->inv_dt = 1.0f / cfg->dt;
    s->inv_side_range = cfg->side_lasers ? 1.0f / cfg->side_range : 0.0f;
    s->inv_lane_range = cfg->lane_line_lasers ? 1.0f / cfg->lane_line_range : 0.0f;
    s->inv_toll = cfg->toll_dim ? 1.0f / (float)(cfg->toll_min_steps > 0 ? cfg->toll_min_steps : 1) : 0.0f;
    for (int p = 0; p < cfg->n_spawns; ++p)
        if (s->spawn_tab[p * 4 + 3]) {
            if (s->n_safe >= COPO_MAX_SAFE) { oracle_sim_destroy(s); return COPO_ERR_CONFIG; }
            s->safe_ids[s->n_safe++] = p;
        }
    if (s->n_safe < 1) { oracle_sim_destroy(s); return COPO_ERR_CONFIG; }
    for (int r = 0; r < cfg->n_routes; ++r) {
        int d = (int)s->route_meta[r * 4 + 3];
        if (d < 0 || d > 32) { oracle_sim_destroy(s); return COPO_ERR_CONFIG; }
        if (d > s->n_spaces) s->n_spaces = d;
    }
    *out = s;
    return COPO_OK;
}

int oracle_sim_destroy(oracle_sim* s) {
    if (!s) return COPO_ERR_NULL;
    free(s->route_segs); free(s->route_meta); free(s->spawn_tab); free(s->spawn_s); free(s->ray_cs); free(s->lines); free(s->boxes);
    free(s->st); free(s->env); free(s->seeds); free(s);
    return COPO_OK;
}

int oracle_sim_set_lcf_dist(oracle_sim* s, double mean, double std) { s->lcf_mean = mean; s->lcf_std = std; return COPO_OK; }
int oracle_sim_set_force_lcf(oracle_sim* s, double v) { s->force_lcf = v; return COPO_OK; }
int oracle_sim_set_capacity(oracle_sim* s, int capacity) { s->capacity = capacity; return COPO_OK; }

int oracle_sim_get_state(oracle_sim* s, float* slot_state, int32_t* env_state) {
    size_t E = s->cfg.num_envs, N = s->cfg.num_agents;
    memcpy(slot_state, s->st, COPO_STATE_FIELDS * E * N * 4);
    memcpy(env_state, s->env, E * 16);
    return COPO_OK;
}
int oracle_sim_set_state(oracle_sim* s, const float* slot_state, const int32_t* env_state) {
    size_t E = s->cfg.num_envs, N = s->cfg.num_agents;
    memcpy(s->st, slot_state, COPO_STATE_FIELDS * E * N * 4);
    memcpy(s->env, env_state, E * 16);
    return COPO_OK;
}
int oracle_sim_set_seeds(oracle_sim* s, const uint64_t* seeds) { memcpy(s->seeds, seeds, 8 * (size_t)s->cfg.num_envs); return COPO_OK; }

static const float* SEG(oracle_sim* s, int route, int k) {
    return s->route_segs + ((size_t)route * (COPO_MAX_SEGS + 1) + k) * COPO_SEG_STRIDE;
}

/* pose of spawn slot sp: on lane `lane` of the spawn road (road 0 of its routes), `spawn_s` metres in */
static void spawn_pose(oracle_sim* s, int sp, float* x, float* y) {
    const float* g = SEG(s, s->spawn_tab[sp * 4 + 0], 0);
    float s0 = s->spawn_s[sp];
    float off = (float)s->spawn_tab[sp * 4 + 2] * s->cfg.lane_width;
    *x = g[0] + g[2] * s0 + g[3] * off;
    *y = g[1] + g[3] * s0 - g[2] * off;
}

/* spawn a fresh agent into slot n of env e at spawn slot sp: MetaDrive's reset / `_respawn_single_vehicle` +
 * `_update_destination_for` (destination = the far end of the reverse of a random spawn road) */
static void spawn_agent(oracle_sim* s, int e, int n, int sp) {
    const copo_sim_cfg* c = &s->cfg;
    int32_t* env = s->env + e * 4;
    uint64_t seed = s->seeds[e];
    uint32_t cnt = (uint32_t)IP(s, S_SPAWNCNT, e)[n] & 0xffffu;
    uint32_t epi = (uint32_t)env[1];
    uint32_t h = o_hash(seed, (uint32_t)n, cnt, epi, RNG_ROUTE);
    int first = s->spawn_tab[sp * 4 + 0], count = s->spawn_tab[sp * 4 + 1];
    int route = first + (int)(h % (uint32_t)count);
    if (s->n_spaces > 0 && s->route_meta[first * 4 + 3] > 0.0f) {
        /* exclusive destinations (MetaDrive's ParkingSpaceManager, marl_parking_lot.py): a space that a LIVING agent is heading for
         * is not handed out again; it comes back when that agent is done.  The h-th of the free ones, in table order; if every
         * one is taken the draw is over all of them. */
        uint32_t taken = 0;
        for (int j = 0; j < c->num_agents; ++j) {
            if (j == n || ST_STATUS(IP(s, S_STATUS, e)[j]) != ST_ALIVE) continue;
            int d = (int)s->route_meta[(IP(s, S_ROUTE, e)[j] & 0xffff) * 4 + 3];
            if (d > 0) taken |= 1u << (d - 1);
        }
        int nfree = 0;
        for (int k = 0; k < count; ++k) {
            int d = (int)s->route_meta[(first + k) * 4 + 3];
            if (!(d > 0 && ((taken >> (d - 1)) & 1u))) nfree += 1;
        }
        if (nfree > 0) {
            int pick = (int)(h % (uint32_t)nfree);
            for (int k = 0; k < count; ++k) {
                int d = (int)s->route_meta[(first + k) * 4 + 3];
                if (d > 0 && ((taken >> (d - 1)) & 1u)) continue;
                if (pick == 0) { route = first + k; break; }
                --pick;
            }
        }
    }
    const float* g = SEG(s, route, 0);
    spawn_pose(s, sp, &FP(s, S_X, e)[n], &FP(s, S_Y, e)[n]);
    FP(s, S_TH, e)[n] = g[7];
    FP(s, S_V, e)[n] = 0.0f;
    FP(s, S_STEER, e)[n] = 0.0f;
    FP(s, S_THROTTLE, e)[n] = 0.0f;
    FP(s, S_PSTEER, e)[n] = 0.0f;
    FP(s, S_PTHROTTLE, e)[n] = 0.0f;
    FP(s, S_YAWRATE, e)[n] = 0.0f;
    FP(s, S_PROG, e)[n] = s->spawn_s[sp];
    FP(s, S_EPREW, e)[n] = 0.0f;
    IP(s, S_ROUTE, e)[n] = route;
    IP(s, S_STATUS, e)[n] = ST_PACK(ST_ALIVE, 0, 0);
    IP(s, S_AID, e)[n] = env[2];
    env[2] += 1;
    float lcf = 0.0f;
    if (c->enable_lcf) {
        float u1 = o_uniform(o_hash(seed, (uint32_t)n, cnt, epi, RNG_LCF1));
        float u2 = o_uniform(o_hash(seed, (uint32_t)n, cnt, epi, RNG_LCF2));
        float sn, cs;
        o_sincosf(TWO_PI_F * u2 - PI_F, &sn, &cs);
        float z = sqrtf(-2.0f * o_logf(u1)) * cs;
        float mean = (s->force_lcf != -100.0) ? (float)s->force_lcf : (float)s->lcf_mean;
        lcf = o_clip(mean + (float)s->lcf_std * z, -1.0f, 1.0f);
    }
    FP(s, S_LCF, e)[n] = lcf;
    IP(s, S_SPAWNCNT, e)[n] = (int32_t)((cnt + 1) & 0xffffu);      /* toll wait (high half) starts at 0 */
}

/* a freshly spawned vehicle stands along its spawn road: heading vector = the road record's (cos0, sin0) */
static void road_heading(oracle_sim* s, int e, int n, float* cs, float* sn) {
    const float* g0 = SEG(s, IP(s, S_ROUTE, e)[n] & 0xffff, 0);
    *cs = g0[2];
    *sn = g0[3];
}

static int oracle_capacity(const oracle_sim* s) {
    int c = s->capacity, N = s->cfg.num_agents;
    return c < 1 ? 1 : (c > N ? N : c);
}

/* reset: `num_agents` distinct spawn slots drawn from ALL slots of the spawn roads (SpawnManager: a respawn later
 * only uses the `safe` slot at the far end of every lane) */
static void reset_env(oracle_sim* s, int e) {
    const copo_sim_cfg* c = &s->cfg;
    int N = c->num_agents, P = c->n_spawns;
    int32_t* env = s->env + e * 4;
    env[0] = 0;
    env[2] = 0;
    int perm[COPO_MAX_SPAWNS];
    for (int i = 0; i < P; ++i) perm[i] = i;
    for (int i = 0; i < N; ++i) { /* partial Fisher-Yates: first N entries */
        uint32_t h = o_hash(s->seeds[e], (uint32_t)i, (uint32_t)env[1], 0u, RNG_PERM);
        int j = i + (int)(h % (uint32_t)(P - i));
        int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    /* population capacity (curriculum): slots beyond it start empty and never respawn */
    int cap = oracle_capacity(s);
    for (int n = 0; n < N; ++n) IP(s, S_STATUS, e)[n] = ST_PACK(ST_EMPTY, 0, 0);     /* nobody of the old episode holds a space */
    for (int n = 0; n < N; ++n) {
        if (n < cap) spawn_agent(s, e, n, perm[n]);
        else IP(s, S_STATUS, e)[n] = ST_PACK(ST_EMPTY, 0, 0);
    }
}

/* Projection of (x, y) with heading (ch, sh) on road g (its lane-0 line): arc length from the road's start, lateral
 * offset (left +), and sin(heading - lane direction).  Arcs measure the angle from their MID point (g[14], g[15]), so
 * nothing wraps inside an arc of up to 360 degrees. */
static void project_seg(const float* g, float x, float y, float ch, float sh, float* sl, float* lat, float* sinpsi) {
    float dx = x - g[0], dy = y - g[1];
    float kap = g[5];
    if (kap == 0.0f) {
        *sl = fm(dx, g[2], dy * g[3]);
        *lat = fm(dy, g[2], -(dx * g[3]));
        *sinpsi = fm(sh, g[2], -(ch * g[3]));
    } else {
        float sg = kap > 0.0f ? 1.0f : -1.0f;
        float R = g[12];
        /* centre = p0 + sg*R*n0, n0 = (-sin0, cos0) */
        float cx = fm(-(sg * R), g[3], g[0]), cy = fm(sg * R, g[2], g[1]);
        float ex = x - cx, ey = y - cy;
        float rho = sqrtf(fm(ex, ex, ey * ey));
        float dotp = fm(g[14], ex, g[15] * ey);
        float crs = fm(g[14], ey, -(g[15] * ex));
        float ang = o_atan2f(sg * crs, dotp);
        *sl = fm(ang, R, 0.5f * g[4]);
        *lat = sg * (R - rho);
        /* left normal of the lane at the vehicle = -sg * e / rho */
        *sinpsi = rho > 0.0f ? (-sg * fm(ch, ex, sh * ey)) / rho : 0.0f;
    }
}

/* Extra drivable width to the right of a straight road of a Merge / Split block (sim_kernels.hip funnel_extra, the same
 * operations in the same order; maps.Net.add_funnel documents the geometry). */
static float funnel_extra(const float* g, float sl, float w) {
    float R = g[12];
    if (g[5] != 0.0f || R == 0.0f) return 0.0f;
    float L = g[4], Ds = g[14], u1 = g[15];
    float D = fabsf(Ds);
    float u = Ds > 0.0f ? sl : L - sl;
    u = u < 0.0f ? 0.0f : (u > L ? L : u);
    if (u <= u1) {
        float R1 = R + 0.5f * w;
        return D - (R1 - sqrtf(R1 * R1 - u * u));
    }
    float R2 = R - 0.5f * w, v = L - u;
    return R2 - sqrtf(R2 * R2 - v * v);
}

typedef struct step_tmp {
    uint8_t acted[COPO_MAX_AGENTS], newly[COPO_MAX_AGENTS], fl[COPO_MAX_AGENTS];
    float rew[COPO_MAX_AGENTS], acc[COPO_MAX_AGENTS], lcf_row[COPO_MAX_AGENTS];
    int32_t aid_row[COPO_MAX_AGENTS];
    float cs[COPO_MAX_AGENTS], sn[COPO_MAX_AGENTS];
    /* communication (env_wrappers.py:102-118): full sorted neighbour lists of this step; fresh = reset observation */
    int nb_ids[COPO_MAX_AGENTS][COPO_MAX_AGENTS];
    int nb_cnt[COPO_MAX_AGENTS];
    int fresh;
} step_tmp;

/* SAT overlap of two oriented boxes: centre, heading unit vector, half length / half width each */
static int obb_overlap2(float xi, float yi, float ci, float si, float ai, float bi,
                        float xj, float yj, float cj, float sj, float aj, float bj) {
    float dx = xj - xi, dy = yj - yi;
    /* (ss stays two products and a subtraction: a fused ci * sj - round(si * cj) is not the exact negation of the pair's
     *  other order cj * si - round(sj * ci), and the kernel tests every UNORDERED pair once) */
    float cc = fabsf(fm(ci, cj, si * sj)), ss = fabsf(ci * sj - si * cj);
    /* axes of i */
    if (fabsf(fm(dx, ci, dy * si)) > fm(bj, ss, fm(aj, cc, ai))) return 0;
    if (fabsf(fm(dy, ci, -(dx * si))) > fm(bj, cc, fm(aj, ss, bi))) return 0;
    /* axes of j */
    if (fabsf(fm(dx, cj, dy * sj)) > fm(bi, ss, fm(ai, cc, aj))) return 0;
    if (fabsf(fm(dy, cj, -(dx * sj))) > fm(bi, cc, fm(ai, ss, bj))) return 0;
    return 1;
}

/* LCFEnv._traffic_light_msg + get_agent_traffic_light_msg (env_wrappers.py:258-272), python float64 arithmetic */
static void traffic_light_cols(const copo_sim_cfg* c, int counter, float x, float y, float* o) {
    int I = c->traffic_light_interval;
    double inc = (double)(counter % I) / (double)I * 0.1;
    double msg = (((counter / I) % 2) == 1) ? 0.0 + inc : 1.0 - inc;
    double b0 = (double)c->map_bbox[0], b1 = (double)c->map_bbox[1], b2 = (double)c->map_bbox[2], b3 = (double)c->map_bbox[3];
    double v[3] = {msg, ((double)x - b0) / (b1 - b0), ((double)y - b2) / (b3 - b2)};
    for (int k = 0; k < 3; ++k) o[k] = (float)(v[k] < 0.0 ? 0.0 : (v[k] > 1.0 ? 1.0 : v[k]));
}

static float nan_canon(void) { union { uint32_t u; float f; } v; v.u = 0x7fc00000u; return v.f; }
#define NAN_CANON nan_canon()   /* one bit pattern for NaN on every platform (np.clip lets NaN through) */
/* Message columns of agent i (CCEnv.step :102-118 + LCFEnv.step :362-371): the comm actions of its first
 * `comm_nb` neighbours (ids[0..cnt), nearest first); zeros for a neighbour that was given no action this step, for a
 * missing neighbour and -- `fresh` -- in a reset observation (:297-303).  act rows are [2 + CS] floats per slot. */
static void comm_cols(int CS, int comm_nb, int add_pos, int fresh, const int* ids, int cnt, const uint8_t* acted,
                      const float* act, const float* px, const float* py, int i, float csi, float sni, float* q0) {
    int CD = CS + (add_pos ? 3 : 0), AD = 2 + CS;
    for (int r = 0; r < comm_nb; ++r) {
        float* q = q0 + r * CD;
        for (int k = 0; k < CD; ++k) q[k] = 0.0f;
        if (fresh || r >= cnt) continue;
        int nn = ids[r];
        if (!acted[nn]) continue;    /* `n in comm_actions`: only agents that were given an action */
        for (int k = 0; k < CS; ++k) q[k] = act[(size_t)nn * AD + 2 + k];
        if (add_pos) {               /* neighbour relative to ego in the ego frame; numpy float64 arithmetic */
            double dx = (double)px[nn] - (double)px[i], dy = (double)py[nn] - (double)py[i];
            double lon = dx * (double)csi + dy * (double)sni, lat2 = dy * (double)csi - dx * (double)sni;
            double dis = sqrt(lon * lon + lat2 * lat2);
            double ex[3] = {dis / 20.0, (lon / dis + 1.0) / 2.0, (lat2 / dis + 1.0) / 2.0};   /* 0/0 = NaN for d == 0, as numpy */
            for (int k = 0; k < 3; ++k) q[CS + k] = ex[k] != ex[k] ? NAN_CANON : (float)(ex[k] < 0.0 ? 0.0 : (ex[k] > 1.0 ? 1.0 : ex[k]));
        }
    }
}

/* One detector beam (MetaDrive SideDetector / LaneLineDetector: a ray test against the lane-line bodies): smallest
 * t in [0, range] at which the ray (x, y) + t (dx, dy) meets a line primitive of kind >= min_kind, or `range`.
 * The result is a pure MINIMUM over the primitives' hit distances (any evaluation order gives the same bits): a straight
 * piece is decided by cross-multiplied comparisons and contributes tn / ad, an arc the smaller of the roots of the circle
 * equation whose point lies within the arc's angular extent; `+ 0.0f` turns a -0 hit distance into +0 (so that the
 * unsigned order of the float bits is the order of the values). */
static float detector_ray(const oracle_sim* s, float x, float y, float dx, float dy, float range, float min_kind) {
    float best = range;
    for (int l = 0; l < s->cfg.n_lines; ++l) {
        const float* L = s->lines + (size_t)l * COPO_LINE_STRIDE;
        if (L[0] < min_kind) continue;
        if (L[6] == 0.0f) {
            float rx = L[1] - x, ry = L[2] - y;
            float den = dx * L[4] - dy * L[3];
            if (den == 0.0f) continue;
            float sd = den > 0.0f ? 1.0f : -1.0f;
            float ad = den * sd;
            float tn = (rx * L[4] - ry * L[3]) * sd;
            float un = (rx * dy - ry * dx) * sd;
            if (!(tn >= 0.0f && un >= 0.0f && un <= L[5] * ad)) continue;
            float th = tn / ad + 0.0f;
            if (th < best) best = th;
        } else {
            float R = 1.0f / fabsf(L[6]);
            float mx = x - L[7], my = y - L[8];
            float b = mx * dx + my * dy;
            float cq = mx * mx + my * my - R * R;
            float disc = b * b - cq;
            if (!(disc >= 0.0f)) continue;
            float sq = sqrtf(disc);
            for (int r = 0; r < 2; ++r) {
                float tt = (r == 0 ? -b - sq : -b + sq) + 0.0f;
                if (!(tt >= 0.0f)) continue;
                float hx = mx + tt * dx, hy = my + tt * dy;
                if (hx * L[9] + hy * L[10] >= R * L[11]) { if (tt < best) best = tt; break; }
            }
        }
    }
    return best;
}

/* Observation rows of the slots in `present` (MetaDrive 0.2.5 LidarStateObservation layout):
 *   [side block | heading, speed, steering, last action x2, yaw rate | lane block | navigation | LiDAR | extensions] */
static void write_obs(oracle_sim* s, int e, const copo_step_out* out, const uint8_t* present, const step_tmp* t,
                      const float* act) {
    const copo_sim_cfg* c = &s->cfg;
    int N = c->num_agents, O = c->obs_dim, L = c->num_lasers;
    float hl = c->veh_half_len, hw = c->veh_half_wid;
    float circ = sqrtf(hl * hl + hw * hw);
    float w = c->lane_width;
    const float *cs = t->cs, *sn = t->sn;   /* heading unit vectors of the step (oracle_sim_step / reset keep them current) */
    uint8_t solid[COPO_MAX_AGENTS];
    for (int j = 0; j < N; ++j) {
        int st = ST_STATUS(IP(s, S_STATUS, e)[j]);
        solid[j] = (st == ST_ALIVE || st == ST_WRECK);
    }
    for (int i = 0; i < N; ++i) {
        float* o = out->obs + ((size_t)e * N + i) * O;
        if (!present[i]) continue;      /* no agent, no row: the slot's observation bytes are left as they are */
        float x = FP(s, S_X, e)[i], y = FP(s, S_Y, e)[i];
        int rw = IP(s, S_ROUTE, e)[i];
        int route = rw & 0xffff, seg = rw >> 16;
        const float* meta = s->route_meta + route * 4;
        int nseg = (int)meta[1];
        const float* g = SEG(s, route, seg);
        float sl, lat, sinpsi;
        project_seg(g, x, y, cs[i], sn[i], &sl, &lat, &sinpsi);
        float lanes = floorf(g[COPO_SEG_LANES]);
        float lif = floorf(fm(-lat, s->inv_w, 0.5f));
        lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
        float left = 0.5f * w - lat;            /* distance to the left edge of the road (of the current route) */
        float right = lanes * w - left;
        int col = 0;
        if (c->side_lasers > 0) {
            for (int k = 0; k < c->side_lasers; ++k) {
                float a = c->side_cs[2 * k], b = c->side_cs[2 * k + 1];
                float dx = cs[i] * a - sn[i] * b, dy = sn[i] * a + cs[i] * b;
                o[col++] = detector_ray(s, x, y, dx, dy, c->side_range, 2.0f) * s->inv_side_range;
            }
        } else {
            float tw = (lanes + 1.0f) * w;      /* StateObservation: (lane_num + 1) * lane_width */
            o[col++] = o_clip(left / tw, 0.0f, 1.0f);
            o[col++] = o_clip(right / tw, 0.0f, 1.0f);
        }
        o[col++] = o_clip(fm(-0.5f, sinpsi, 0.5f), 0.0f, 1.0f);        /* heading_diff: cos to the lane's RIGHT normal */
        o[col++] = o_clip(fm(fabsf(FP(s, S_V, e)[i]), 3.6f, 1.0f) * s->inv_vnorm, 0.0f, 1.0f);    /* vehicle.speed: a magnitude */
        o[col++] = o_clip(fm(FP(s, S_STEER, e)[i], 1.0f / 120.0f, 0.5f), 0.0f, 1.0f);  /* (steering / MAX_STEERING(60) + 1) / 2 */
        o[col++] = o_clip(fm(0.5f, FP(s, S_PSTEER, e)[i], 0.5f), 0.0f, 1.0f);    /* last_current_action[0]: the step before */
        o[col++] = o_clip(fm(0.5f, FP(s, S_PTHROTTLE, e)[i], 0.5f), 0.0f, 1.0f);
        o[col++] = o_clip(fabsf(FP(s, S_YAWRATE, e)[i]), 0.0f, 1.0f);          /* |heading change| / 0.1 */
        if (c->lane_line_lasers > 0) {
            for (int k = 0; k < c->lane_line_lasers; ++k) {
                float a = c->lane_line_cs[2 * k], b = c->lane_line_cs[2 * k + 1];
                float dx = cs[i] * a - sn[i] * b, dy = sn[i] * a + cs[i] * b;
                o[col++] = detector_ray(s, x, y, dx, dy, c->lane_line_range, 1.0f) * s->inv_lane_range;
            }
        } else {
            float latr = -fm(lif, w, lat);      /* offset in the vehicle's lane, right +; MAX_LANE_WIDTH 4.5 */
            o[col++] = o_clip(fm(latr, 1.0f / 4.5f, 0.5f), 0.0f, 1.0f);
        }
        /* navigation block: check points at the end of the current road and of the next one (the current one again on
         * the final road), Navigation._get_info_for_checkpoint */
        if (c->navi_dim)
            for (int j = 0; j < 2; ++j) {
                int kk = seg + j;
                if (kk > nseg - 1) kk = nseg - 1;
                const float* gk = SEG(s, route, kk);
                float ckx = gk[COPO_SEG_CKX], cky = gk[COPO_SEG_CKX + 1];
                if (floorf(gk[COPO_SEG_LANES]) != lanes) {
                    /* both check points sit at the lateral middle of the CURRENT road's lane count (Navigation.
                     * _get_info_for_checkpoint: later_middle from get_current_lane_num()), to the right of road kk's lane 0 */
                    const float* gn = SEG(s, route, kk + 1);
                    float off = (lanes * 0.5f - 0.5f) * w;
                    ckx = gn[0] + gn[3] * off;
                    cky = gn[1] - gn[2] * off;
                }
                float vx = ckx - x, vy = cky - y;
                float nrm = sqrtf(fm(vx, vx, vy * vy));
                if (nrm > 50.0f) {
                    float sc = 50.0f / nrm;
                    vx = vx * sc;
                    vy = vy * sc;
                }
                float fwd = fm(vx, cs[i], vy * sn[i]), rhs = fm(vx, sn[i], -(vy * cs[i]));
                float* q = o + col;
                q[0] = o_clip(fm(fwd, 0.01f, 0.5f), 0.0f, 1.0f);
                q[1] = o_clip(fm(rhs, 0.01f, 0.5f), 0.0f, 1.0f);
                q[2] = gk[COPO_SEG_FEAT];
                q[3] = gk[5] == 0.0f ? 0.5f : (gk[5] < 0.0f ? 1.0f : 0.0f);
                q[4] = gk[COPO_SEG_FEAT + 2];
                col += 5;
            }
        /* LiDAR */
        float* lid = o + col;
        float range = c->lidar_range;
        /* Every (ray, vehicle) decision is made in the box frame of vehicle j.  Per pair (i, j): the ray origin in that
         * frame (ox, oy) and the rotation from i's frame into it (cr, sr); per ray only the direction (ddx, ddy). */
        float pox[COPO_MAX_AGENTS], poy[COPO_MAX_AGENTS], pcr[COPO_MAX_AGENTS], psr[COPO_MAX_AGENTS];
        uint8_t near[COPO_MAX_AGENTS];
        for (int j = 0; j < N; ++j) {
            near[j] = 0;
            if (j == i || !solid[j]) continue;
            float rx = FP(s, S_X, e)[j] - x, ry = FP(s, S_Y, e)[j] - y;
            float lim = range + circ;
            if (fm(rx, rx, ry * ry) > lim * lim) continue;
            near[j] = 1;
            pox[j] = -fm(rx, cs[j], ry * sn[j]);
            poy[j] = -fm(ry, cs[j], -(rx * sn[j]));
            pcr[j] = fm(cs[i], cs[j], sn[i] * sn[j]);
            psr[j] = fm(cs[i], sn[j], -(sn[i] * cs[j]));
        }
        for (int k = 0; k < L; ++k) {
            float rc = s->ray_cs[2 * k], rs = s->ray_cs[2 * k + 1];
            float best = range;
            for (int j = 0; j < N; ++j) {
                if (!near[j]) continue;
                /* ray vs box j in j's frame, mirrored so that the direction is non-negative on both axes; entering and
                 * exiting times are fractions n/a compared by cross-multiplication (no division until a hit is known) */
                float ox = pox[j], oy = poy[j];
                float ddx = fm(rc, pcr[j], rs * psr[j]), ddy = fm(rs, pcr[j], -(rc * psr[j]));
                float ax = fabsf(ddx), ay = fabsf(ddy);
                float oxs = ddx < 0.0f ? -ox : ox, oys = ddy < 0.0f ? -oy : oy;
                float nxe = -(hl + oxs), nxx = hl - oxs, nye = -(hw + oys), nyx = hw - oys;
                if (!(nxx >= 0.0f && nyx >= 0.0f)) continue;            /* box entirely behind the origin on an axis */
                if (!(nxe * ay <= nyx * ax)) continue;                  /* enter-x after exit-y */
                if (!(nye * ax <= nxx * ay)) continue;                  /* enter-y after exit-x */
                int usex = (nxe * ay >= nye * ax);                      /* the later entering plane */
                float n = usex ? nxe : nye, a = usex ? ax : ay;
                float tt = n > 0.0f ? n / a : 0.0f;                     /* origin inside the box -> 0 */
                if (tt < best) best = tt;
            }
            for (int b = 0; b < (c->boxes_hidden ? 0 : c->n_boxes); ++b) {      /* static boxes (buildings): the same test, the box's own half extents */
                const float* B = s->boxes + (size_t)b * COPO_BOX_STRIDE;
                float rx = B[0] - x, ry = B[1] - y;
                float ox = -fm(rx, B[2], ry * B[3]), oy = -fm(ry, B[2], -(rx * B[3]));
                float bcr = fm(cs[i], B[2], sn[i] * B[3]), bsr = fm(cs[i], B[3], -(sn[i] * B[2]));
                float ddx = fm(rc, bcr, rs * bsr), ddy = fm(rs, bcr, -(rc * bsr));
                float ax = fabsf(ddx), ay = fabsf(ddy);
                float oxs = ddx < 0.0f ? -ox : ox, oys = ddy < 0.0f ? -oy : oy;
                float nxe = -(B[4] + oxs), nxx = B[4] - oxs, nye = -(B[5] + oys), nyx = B[5] - oys;
                if (!(nxx >= 0.0f && nyx >= 0.0f)) continue;
                if (!(nxe * ay <= nyx * ax)) continue;
                if (!(nye * ax <= nxx * ay)) continue;
                int usex = (nxe * ay >= nye * ax);
                float n = usex ? nxe : nye, a = usex ? ax : ay;
                float tt = n > 0.0f ? n / a : 0.0f;
                if (tt < best) best = tt;
            }
            lid[k] = best * s->inv_range;
        }
        col += L;
        if (c->toll_dim) {              /* Tollgate: [on the booth road, stayed longer than `toll_min_steps`], zeros off it */
            uint32_t wait = (uint32_t)IP(s, S_SPAWNCNT, e)[i] >> 16;
            int in_booth = (seg == (int)meta[2]);
            o[col++] = in_booth ? 1.0f : 0.0f;
            o[col++] = (in_booth && wait > (uint32_t)c->toll_min_steps) ? 1.0f : 0.0f;
        }
        if (c->add_traffic_light) {     /* counter = steps since the last reset (env word 0 is already advanced) */
            traffic_light_cols(c, s->env[e * 4], x, y, o + col);
            col += 3;
        }
        if (c->enable_lcf) o[col++] = (FP(s, S_LCF, e)[i] + 1.0f) * 0.5f;
        if (c->comm_size > 0)
            comm_cols(c->comm_size, c->comm_neighbours, c->add_pos_in_comm, t->fresh, t->nb_ids[i], t->nb_cnt[i], t->acted,
                      act ? act + (size_t)e * N * (2 + c->comm_size) : NULL, FP(s, S_X, e), FP(s, S_Y, e), i, cs[i], sn[i], o + col);
    }
}

/* neighbour lists + reward reductions for one env on an explicit present set (fp64 distances) */
static void neighbours_env(const float* px, const float* py, const uint8_t* present, const float* rew, int N, int K,
                           float radius, float mf, int32_t* nbr_idx, int32_t* nbr_cnt, int32_t* mf_cnt, float* nbr_dist,
                           float* nei_rew, float* glob_rew, int (*full_ids)[COPO_MAX_AGENTS], int* full_cnt, int skip_absent) {
    double gsum = 0.0;
    int gcnt = 0;
    for (int i = 0; i < N; ++i)
        if (present[i] && rew) { gsum += (double)rew[i]; gcnt++; }
    if (glob_rew) *glob_rew = gcnt ? (float)(gsum / (double)gcnt) : 0.0f;
    for (int i = 0; i < N; ++i) {
        int ids[COPO_MAX_AGENTS];
        double ds[COPO_MAX_AGENTS];
        int cnt = 0;
        if (present[i]) {
            for (int j = 0; j < N; ++j) {
                if (j == i || !present[j]) continue;
                double dx = (double)px[i] - (double)px[j], dy = (double)py[i] - (double)py[j];
                double d = sqrt(dx * dx + dy * dy);
                if (d < (double)radius) {
                    /* stable insertion sort by distance: ties keep ascending slot order (python sorted) */
                    int p = cnt;
                    while (p > 0 && ds[p - 1] > d) { ds[p] = ds[p - 1]; ids[p] = ids[p - 1]; --p; }
                    ds[p] = d; ids[p] = j; cnt++;
                }
            }
        }
        if (nbr_cnt) nbr_cnt[i] = cnt;
        if (full_cnt) {
            full_cnt[i] = cnt;
            for (int k = 0; k < cnt; ++k) full_ids[i][k] = ids[k];
        }
        int m = 0;
        for (int k = 0; k < cnt; ++k)
            if (ds[k] <= (double)mf) m++; else break;
        if (mf_cnt) mf_cnt[i] = m;
        for (int k = 0; k < K && (present[i] || !skip_absent); ++k) {    /* simulator: rows of absent slots are left alone */
            if (nbr_idx) nbr_idx[i * K + k] = k < cnt ? ids[k] : -1;
            if (nbr_dist) nbr_dist[i * K + k] = k < cnt ? (float)ds[k] : 0.0f;
        }
        if (nei_rew) {
            double sum = 0.0;
            for (int k = 0; k < cnt; ++k) sum += (double)rew[ids[k]];
            nei_rew[i] = cnt ? (float)(sum / (double)cnt) : 0.0f;
        }
    }
}

int oracle_neighbours(const float* pos, const uint8_t* present, const float* rew, int32_t E, int32_t N, int32_t K,
                      float radius, float mf_distance, int32_t* nbr_idx, int32_t* nbr_cnt, int32_t* mf_cnt,
                      float* nbr_dist, float* nei_rew, float* glob_rew) {
    if (N > COPO_MAX_AGENTS || K > COPO_MAX_AGENTS) return COPO_ERR_DIM;
    for (int e = 0; e < E; ++e) {
        float px[COPO_MAX_AGENTS], py[COPO_MAX_AGENTS];
        for (int i = 0; i < N; ++i) { px[i] = pos[((size_t)e * N + i) * 2]; py[i] = pos[((size_t)e * N + i) * 2 + 1]; }
        neighbours_env(px, py, present + (size_t)e * N, rew ? rew + (size_t)e * N : NULL, N, K, radius, mf_distance,
                       nbr_idx ? nbr_idx + (size_t)e * N * K : NULL, nbr_cnt ? nbr_cnt + (size_t)e * N : NULL,
                       mf_cnt ? mf_cnt + (size_t)e * N : NULL, nbr_dist ? nbr_dist + (size_t)e * N * K : NULL,
                       (rew && nei_rew) ? nei_rew + (size_t)e * N : NULL, (rew && glob_rew) ? glob_rew + e : NULL, NULL, NULL, 0);
    }
    return COPO_OK;
}

/* Stateless form of the observation extensions of one scene (what write_obs appends), for the golden vectors
 * captured from the reference's LCFEnv: neighbour lists on `pos` over the present set, then the columns. */
int oracle_obs_extensions(const float* pos, const float* cs_sn, const uint8_t* present, const uint8_t* acted,
                          const float* act, int32_t N, float radius, int32_t counter, int32_t interval, const float* bbox,
                          int32_t add_tl, int32_t comm_size, int32_t comm_nb, int32_t add_pos, int32_t fresh, float* tl_out,
                          float* comm_out) {
    if (N > COPO_MAX_AGENTS) return COPO_ERR_DIM;
    float px[COPO_MAX_AGENTS] = {0}, py[COPO_MAX_AGENTS] = {0};
    static int ids[COPO_MAX_AGENTS][COPO_MAX_AGENTS];
    int cnt[COPO_MAX_AGENTS];
    for (int i = 0; i < N; ++i) { px[i] = pos[2 * i]; py[i] = pos[2 * i + 1]; }
    neighbours_env(px, py, present, NULL, N, 1, radius, 10.0f, NULL, NULL, NULL, NULL, NULL, NULL, ids, cnt, 0);
    copo_sim_cfg c;
    memset(&c, 0, sizeof(c));
    c.traffic_light_interval = interval;
    for (int k = 0; k < 4; ++k) c.map_bbox[k] = bbox[k];
    int CD = comm_size + (add_pos ? 3 : 0);
    for (int i = 0; i < N; ++i) {
        if (!present[i]) continue;
        if (add_tl) traffic_light_cols(&c, counter, px[i], py[i], tl_out + 3 * i);
        if (comm_size > 0)
            comm_cols(comm_size, comm_nb, add_pos, fresh, ids[i], cnt[i], acted, act, px, py, i, cs_sn[2 * i], cs_sn[2 * i + 1],
                      comm_out + (size_t)i * comm_nb * CD);
    }
    return COPO_OK;
}

static void emit_outputs(oracle_sim* s, int e, const copo_step_out* out, step_tmp* t, const uint8_t* present) {
    const copo_sim_cfg* c = &s->cfg;
    int N = c->num_agents, K = c->nbr_k;
    size_t b = (size_t)e * N;
    /* neighbour lists on post-step positions of the present set; rewards of new spawns are 0 */
    neighbours_env(FP(s, S_X, e), FP(s, S_Y, e), present, t->rew, N, K, c->neighbours_distance, c->mf_distance,
                   out->nbr_idx ? out->nbr_idx + b * K : NULL, out->nbr_cnt ? out->nbr_cnt + b : NULL,
                   out->mf_cnt ? out->mf_cnt + b : NULL, out->nbr_dist ? out->nbr_dist + b * K : NULL,
                   out->nei_rew ? out->nei_rew + b : NULL, out->glob_rew ? out->glob_rew + e : NULL, t->nb_ids, t->nb_cnt, 1);
    for (int n = 0; n < N; ++n) {
        if (out->rew) out->rew[b + n] = t->rew[n];
        if (out->flags) out->flags[b + n] = t->fl[n];
        if (out->lcf) out->lcf[b + n] = t->lcf_row[n];
        if (out->agent_id) out->agent_id[b + n] = t->aid_row[n];
    }
}

int oracle_sim_reset(oracle_sim* s, const uint64_t* seeds, const copo_step_out* out) {
    const copo_sim_cfg* c = &s->cfg;
    int E = c->num_envs, N = c->num_agents;
    memcpy(s->seeds, seeds, 8 * (size_t)E);
    memset(s->st, 0, COPO_STATE_FIELDS * (size_t)E * N * 4);
    for (int e = 0; e < E; ++e) {
        int32_t* env = s->env + e * 4;
        env[0] = env[1] = env[2] = 0;
        env[3] = 1;
        reset_env(s, e);
        step_tmp t;
        memset(&t, 0, sizeof(t));
        uint8_t present[COPO_MAX_AGENTS];
        for (int n = 0; n < N; ++n) {
            present[n] = n < oracle_capacity(s);
            if (present[n]) road_heading(s, e, n, &t.cs[n], &t.sn[n]);
            t.fl[n] = present[n] ? COPO_F_SPAWNED : 0;
            t.lcf_row[n] = FP(s, S_LCF, e)[n];
            t.aid_row[n] = IP(s, S_AID, e)[n];
        }
        emit_outputs(s, e, out, &t, present);
        if (out->info) memset(out->info + (size_t)e * N * COPO_INFO_DIM, 0, sizeof(float) * N * COPO_INFO_DIM);
        t.fresh = 1;
        if (out->obs) write_obs(s, e, out, present, &t, NULL);
    }
    return COPO_OK;
}

int oracle_sim_step(oracle_sim* s, const float* act, const copo_step_out* out) {
    const copo_sim_cfg* c = &s->cfg;
    int E = c->num_envs, N = c->num_agents;
    float hl = c->veh_half_len, hw = c->veh_half_wid;
    float h = c->dt / (float)c->substeps;
    float w = c->lane_width;
    for (int e = 0; e < E; ++e) {
        int32_t* env = s->env + e * 4;
        if (!env[3]) return COPO_ERR_STATE;
        step_tmp t;
        memset(&t, 0, sizeof(t));
        float *X = FP(s, S_X, e), *Y = FP(s, S_Y, e), *TH = FP(s, S_TH, e), *V = FP(s, S_V, e);
        int32_t* STA = IP(s, S_STATUS, e);
        /* 0. timers of non-alive slots */
        for (int n = 0; n < N; ++n) {
            int st = ST_STATUS(STA[n]), tm = ST_TIMER(STA[n]);
            t.acted[n] = (st == ST_ALIVE);
            if (st == ST_WRECK) {
                tm -= 1;
                if (tm <= 0) STA[n] = ST_PACK(ST_EMPTY, c->respawn_cooldown, 0); else STA[n] = ST_PACK(ST_WRECK, tm, 0);
            } else if (st == ST_EMPTY && tm > 0) {
                STA[n] = ST_PACK(ST_EMPTY, tm - 1, 0);
            }
        }
        /* 1. kinematic bicycle for acting slots: steering angle a0 * max_steer held over the step, slip angle of the
         *    body centre beta = atan(tan(delta) / 2), engine 4 x max_engine_force / mass, brake friction-limited */
        for (int n = 0; n < N; ++n) {
            t.lcf_row[n] = FP(s, S_LCF, e)[n];
            t.aid_row[n] = t.acted[n] ? IP(s, S_AID, e)[n] : -1;
            if (!t.acted[n]) continue;
            const int AD = COPO_ACT_DIM(c);
            float a0 = act[((size_t)e * N + n) * AD], a1 = act[((size_t)e * N + n) * AD + 1];
            if (!(a0 == a0)) a0 = 0.0f;
            if (!(a1 == a1)) a1 = 0.0f;
            a0 = o_clip(a0, -1.0f, 1.0f);
            a1 = o_clip(a1, -1.0f, 1.0f);
            float delta = a0 * c->max_steer;
            float sd, cd;
            o_sincosf(delta, &sd, &cd);
            float tand = sd / cd;
            float tb = 0.5f * tand;
            float cb = 1.0f / sqrtf(1.0f + tb * tb), sb = tb * cb;
            float yawk = (tand / c->wheelbase) * cb;
            float brake = -a1 * c->brake_gain;
            if (brake > c->brake_max) brake = c->brake_max;
            float x = X[n], y = Y[n], th = TH[n], v = V[n];
            float v0 = v, th0 = th;
            /* heading unit vector: one sincos per step, then turned by every sub-step's small angle with a 3-term sine /
             * 3-term cosine (|dth| <= 0.05 rad: error < 1e-11); the vector after the last sub-step is the pose's heading
             * vector for the rest of the step (collision, LiDAR, observation) */
            float sn, cs;
            o_sincosf(th, &sn, &cs);
            for (int k = 0; k < c->substeps; ++k) {
                /* reverse gear (MetaDrive enable_reverse): a negative throttle is engine force backwards, no brake, v may go negative */
                float a = a1 >= 0.0f ? (v < c->max_speed ? a1 * c->acc_max : 0.0f) : (c->reverse_acc > 0.0f ? (v > -c->max_speed ? a1 * c->reverse_acc : 0.0f) : -brake);
                v = fm(a, h, v);
                if (v < 0.0f && !(c->reverse_acc > 0.0f)) v = 0.0f;
                float dxh = fm(cs, cb, -(sn * sb)), dyh = fm(sn, cb, cs * sb);
                x = fm(v * dxh, h, x);
                y = fm(v * dyh, h, y);
                float dth = v * yawk * h;
                if (c->lat_acc_max > 0.0f && v * fabsf(dth) > c->lat_acc_max * h) {      /* tyres slide: v x yaw rate is friction-limited */
                    float lim = (c->lat_acc_max * h) / v;
                    dth = dth < 0.0f ? -lim : lim;
                }
                float q = dth * dth;
                float sd2 = fm(-(dth * q), fm(-q, 0.00833333333f, 0.166666667f), dth);
                float cd2 = fm(-q, fm(-q, 0.0416666667f, 0.5f), 1.0f);
                float cn = fm(cs, cd2, -(sn * sd2)), sm = fm(sn, cd2, cs * sd2);
                cs = cn;
                sn = sm;
                th = o_wrap_pi(th + dth);
            }
            t.cs[n] = cs;
            t.sn[n] = sn;
            X[n] = x; Y[n] = y; TH[n] = th; V[n] = v;
            FP(s, S_PSTEER, e)[n] = FP(s, S_STEER, e)[n];
            FP(s, S_PTHROTTLE, e)[n] = FP(s, S_THROTTLE, e)[n];
            FP(s, S_STEER, e)[n] = a0;
            FP(s, S_THROTTLE, e)[n] = a1;
            FP(s, S_YAWRATE, e)[n] = o_wrap_pi(th - th0) * s->inv_dt;
            t.acc[n] = (v - v0) * s->inv_dt;
            STA[n] = ST_PACK(ST_ALIVE, 0, ST_AGE(STA[n]) + 1);
        }
        /* 2. heading unit vectors of the slots that did not act (wrecks; empty slots are never read) */
        for (int n = 0; n < N; ++n)
            if (!t.acted[n]) o_sincosf(TH[n], &t.sn[n], &t.cs[n]);
        /* MultiAgentMetaDrive.step: once `horizon` env steps have run the scene stops respawning and drains; it is reset
         * when no agent is left (done["__all__"] = episode_steps >= horizon and all(d.values()), or no vehicle at all, or
         * 5 x horizon steps).  Every agent has its own step limit: episode_lengths[id] >= horizon -> max_step. */
        int no_respawn = (env[0] + 1 >= c->horizon);
        int force_end = (env[0] + 1 >= 5 * c->horizon);
        /* 3-5. collision, route projection, termination, reward */
        uint8_t term[COPO_MAX_AGENTS];
        uint8_t crash_any[COPO_MAX_AGENTS];
        for (int n = 0; n < N; ++n) {
            crash_any[n] = 0;
            if (!t.acted[n]) continue;
            for (int j = 0; j < N; ++j) {
                if (j == n || ST_STATUS(STA[j]) == ST_EMPTY) continue;
                if (obb_overlap2(X[n], Y[n], t.cs[n], t.sn[n], hl, hw, X[j], Y[j], t.cs[j], t.sn[j], hl, hw)) crash_any[n] = 1;
            }
        }
        for (int n = 0; n < N; ++n) {
            term[n] = 0;
            t.rew[n] = 0.0f;
            t.fl[n] = 0;
            if (!t.acted[n]) continue;
            int rw = IP(s, S_ROUTE, e)[n];
            int route = rw & 0xffff, seg = rw >> 16;
            const float* meta = s->route_meta + route * 4;
            int nseg = (int)meta[1];
            float sl, lat, sinpsi;
            const float* g = SEG(s, route, seg);
            project_seg(g, X[n], Y[n], t.cs[n], t.sn[n], &sl, &lat, &sinpsi);
            for (int it = 0; it < 2; ++it) {        /* Navigation.update_localization: on to the next road */
                if (sl > g[4] && seg < nseg - 1) {
                    seg += 1;
                    g = SEG(s, route, seg);
                    project_seg(g, X[n], Y[n], t.cs[n], t.sn[n], &sl, &lat, &sinpsi);
                }
            }
            if (sl < 0.0f && seg > 0) {
                seg -= 1;
                g = SEG(s, route, seg);
                project_seg(g, X[n], Y[n], t.cs[n], t.sn[n], &sl, &lat, &sinpsi);
            }
            float prog = g[6] + sl;
            float prev = FP(s, S_PROG, e)[n];
            int too_fast = 0, in_toll = 0;
            if (c->toll_dim) {          /* booth road meta[2]: count the steps spent on it; leaving it early is a failure */
                int toll_seg = (int)meta[2], seg_before = rw >> 16;
                in_toll = (seg == toll_seg);
                uint32_t sc = (uint32_t)IP(s, S_SPAWNCNT, e)[n];
                uint32_t wait = sc >> 16;
                if (seg == toll_seg && wait < 0xffffu) wait += 1;
                too_fast = toll_seg >= 0 && seg > toll_seg && seg_before <= toll_seg && wait < (uint32_t)c->toll_min_steps;
                IP(s, S_SPAWNCNT, e)[n] = (int32_t)((sc & 0xffffu) | (wait << 16));
            }
            IP(s, S_ROUTE, e)[n] = route | (seg << 16);
            FP(s, S_PROG, e)[n] = prog;
            float lanes_f = g[COPO_SEG_LANES], lanes = floorf(lanes_f), lfr = lanes_f - lanes;      /* fraction: edge-line flags, eighths */
            int lcode = (int)(lfr * 8.0f);
            int left_solid = (lcode & 2) != 0, right_solid = (lcode & 4) != 0, left_open = (lcode & 1) != 0;
            float lif = floorf(fm(-lat, s->inv_w, 0.5f));
            lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
            float left = 0.5f * w - lat, right = (lanes * w + funnel_extra(g, sl, w)) - left;
            /* the body's half extent across the road (heading error psi): the edge lines must not be touched */
            float cos2 = fm(-sinpsi, sinpsi, 1.0f);
            float edge = c->body_margin * fm(hl, fabsf(sinpsi), hw * sqrtf(cos2 > 0.0f ? cos2 : 0.0f));
            int on_road = (left >= (left_solid ? edge : (left_open ? -w : 0.0f))) && (right >= (right_solid ? edge : 0.0f));
            /* _is_arrive_destination: within +-5 m of the end of the final road, anywhere across it */
            int arrive = (seg == nseg - 1) && (sl > g[4] - c->arrive_margin) && (sl < g[4] + c->arrive_margin) && on_road;
            int out_of_road = !on_road;         /* vehicle.out_of_route (out_of_route_done) */
            /* MultiAgentTollgateEnv (MetaDrive 0.2.5, marl_tollgate.py; restated, source not in the reference tree): done_function
             * ends a vehicle whose stay on the booth road was shorter than min_pass_steps with done_info["out_of_road"] = True -- the
             * reward function does not see it (toll_early_exit = 1); rounds 2-5 made it a crash with -crash_penalty (0) */
            int early = too_fast && c->toll_early_exit != 0;
            /* static boxes of the map (TollGate._add_building_and_speed_limit: `if idx % 2 == 1` a TollGateBuilding, lane width x
             * road length, at the centre of every second booth lane): touching one is crash_building -- the vehicles' SAT test */
            int bldg = 0;
            for (int b = 0; b < c->n_boxes; ++b) {
                const float* B = s->boxes + (size_t)b * COPO_BOX_STRIDE;
                if (obb_overlap2(X[n], Y[n], t.cs[n], t.sn[n], hl, hw, B[0], B[1], B[2], B[3], B[4], B[5])) bldg = 1;
            }
            int crash = crash_any[n] || bldg || (too_fast && c->toll_early_exit == 0);
            /* reward_function: longitudinal movement on the vehicle's lane (lane i of an arc is 1 + kappa * i * w longer
             * than lane 0) + speed term; use_lateral is off in 0.2.5 */
            /* ... and reward_function: on the booth road `if vehicle.overspeed: reward = -overspeed_penalty * speed / max_speed`
             * (TollGate.SPEED_LIMIT = 3 km/h), else the driving reward alone; off it the speed term is added as everywhere */
            float drive = (prog - prev) * fm(g[5], lif * w, 1.0f), spd = fabsf(V[n]) / c->max_speed;
            float r;
            if (c->toll_speed_limit > 0.0f && in_toll) r = fabsf(V[n]) > c->toll_speed_limit ? -c->overspeed_penalty * spd : c->driving_reward * drive;
            else r = fm(c->driving_reward, drive, c->speed_reward * spd);
            uint8_t fl = COPO_F_ACTED;
            if (arrive) { r = c->success_reward; fl |= COPO_F_ARRIVE; }
            else if (out_of_road) { r = -c->out_penalty; }
            else if (crash) { r = -c->crash_penalty; }
            if (out_of_road || early) fl |= COPO_F_OUT;
            if (crash) fl |= COPO_F_CRASH;
            int done = arrive || out_of_road || crash || early;
            if (!done && (ST_AGE(STA[n]) >= c->horizon || force_end)) { fl |= COPO_F_MAXSTEP; done = 1; }
            if (done) fl |= COPO_F_DONE;
            term[n] = (uint8_t)done;
            t.rew[n] = r;
            t.fl[n] = fl;
            FP(s, S_EPREW, e)[n] += r;
            if (out->info) {
                float* q = out->info + ((size_t)e * N + n) * COPO_INFO_DIM;
                q[COPO_I_VELOCITY] = fabsf(V[n]) * 3.6f;
                q[COPO_I_STEERING] = FP(s, S_STEER, e)[n];
                q[COPO_I_ACCELERATION] = t.acc[n];
                q[COPO_I_STEP_REWARD] = r;
                q[COPO_I_COST] = crash ? 1.0f : 0.0f;
                q[COPO_I_EPISODE_LENGTH] = (float)ST_AGE(STA[n]);
                q[COPO_I_EPISODE_REWARD] = FP(s, S_EPREW, e)[n];
                q[COPO_I_ROUTE_COMPLETION] = o_clip(prog / meta[0], 0.0f, 1.0f);
            }
        }
        if (out->info)
            for (int n = 0; n < N; ++n)
                if (!t.acted[n]) memset(out->info + ((size_t)e * N + n) * COPO_INFO_DIM, 0, sizeof(float) * COPO_INFO_DIM);
        /* 6. status update of terminated slots: a vehicle that arrived leaves at once, every other one stays as a static
         *    obstacle for `delay_done` steps (AgentManager.finish(ignore_delay_done = success)) */
        for (int n = 0; n < N; ++n) {
            if (!term[n]) continue;
            if (!(t.fl[n] & COPO_F_ARRIVE) && c->delay_done > 0)
                STA[n] = ST_PACK(ST_WRECK, c->delay_done, 0);
            else
                STA[n] = ST_PACK(ST_EMPTY, c->respawn_cooldown, 0);
        }
        uint8_t present[COPO_MAX_AGENTS];
        for (int n = 0; n < N; ++n) present[n] = t.acted[n];
        /* 7. respawn (serial in slot order): a random one of the SAFE places whose 8 x 3 m region holds no vehicle
         *    (SpawnManager.get_available_respawn_places), each place at most once per step */
        if (!no_respawn) {
            /* places whose region is clear of the vehicles standing when the respawns begin; `used`: taken this step */
            uint32_t clear = 0, used = 0;
            for (int q = 0; q < s->n_safe; ++q) {
                int sp = s->safe_ids[q];
                const float* g = SEG(s, s->spawn_tab[sp * 4], 0);
                float sx, sy;
                spawn_pose(s, sp, &sx, &sy);
                int blocked = 0;
                for (int j = 0; j < N; ++j) {
                    if (ST_STATUS(STA[j]) == ST_EMPTY) continue;
                    if (obb_overlap2(sx, sy, g[2], g[3], 0.5f * c->spawn_region_len, 0.5f * c->spawn_region_wid,
                                     X[j], Y[j], t.cs[j], t.sn[j], hl, hw)) blocked = 1;
                }
                if (!blocked) clear |= 1u << q;
            }
            for (int n = 0; n < N; ++n) {
                if (n >= oracle_capacity(s)) break;
                if (t.acted[n] || STA[n] != ST_PACK(ST_EMPTY, 0, 0)) continue;
                uint32_t freem = clear & ~used;
                if (!freem) break;              /* nowhere to go this step: everybody still waiting keeps waiting */
                int nfree = __builtin_popcount(freem);
                uint32_t cnt = (uint32_t)IP(s, S_SPAWNCNT, e)[n] & 0xffffu;
                uint32_t hh = o_hash(s->seeds[e], (uint32_t)n, cnt, (uint32_t)env[0], RNG_SPAWN);
                int pick = (int)(hh % (uint32_t)nfree), q = 0;
                for (;; ++q)                    /* the pick-th clear place in table order */
                    if ((freem >> q) & 1u) { if (pick == 0) break; --pick; }
                used |= 1u << q;
                spawn_agent(s, e, n, s->safe_ids[q]);
                road_heading(s, e, n, &t.cs[n], &t.sn[n]);
                present[n] = 1;
                t.newly[n] = 1;
                t.fl[n] = COPO_F_SPAWNED;
                t.lcf_row[n] = FP(s, S_LCF, e)[n];
            }
        }
        int ending = 1;                     /* nobody left driving: the episode is over */
        for (int n = 0; n < N; ++n)
            if (ST_STATUS(STA[n]) == ST_ALIVE) ending = 0;
        /* 8. neighbour lists, reward reductions, row outputs (on the pre-reset scene) */
        emit_outputs(s, e, out, &t, present);
        env[0] += 1;
        /* 9-10. end of the episode: reset the env, then observations of whoever occupies the slots now */
        if (ending) {
            env[1] += 1;
            reset_env(s, e);
            for (int n = 0; n < N; ++n) {
                present[n] = n < oracle_capacity(s);
                if (present[n]) road_heading(s, e, n, &t.cs[n], &t.sn[n]);
                if (out->flags) out->flags[(size_t)e * N + n] |= (present[n] ? COPO_F_SPAWNED : 0) | COPO_F_ENV_RESET;
                if (out->lcf && !t.acted[n] && present[n]) out->lcf[(size_t)e * N + n] = FP(s, S_LCF, e)[n];
            }
            t.fresh = 1;
        }
        if (out->obs) write_obs(s, e, out, present, &t, act);
    }
    return COPO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * learn-side ops
 * ---------------------------------------------------------------------------------------------- */

/* GAE, restating the dtype path of the reference: delta in fp32 when the trajectory is truncated
 * (last_r is a np.float32 value -> all-fp32 numpy expression), in fp64 when it ended with done
 * (last_r = python 0.0 -> np.array([0.0]) is fp64); the discounted cumsum (scipy lfilter) always fp64. */
int oracle_gae3(const float* rew, const float* val, const uint8_t* flags, int32_t T, int32_t M, int32_t heads,
                const double* gamma, double lam, float* adv, float* tgt) {
    for (int hd = 0; hd < heads; ++hd) {
        const float* R = rew + (size_t)hd * T * M;
        const float* Vv = val + (size_t)hd * T * M;
        float* A = adv + (size_t)hd * T * M;
        float* G = tgt + (size_t)hd * T * M;
        double g64 = gamma[hd];     /* python float gamma */
        float g32 = (float)g64;
        double c = g64 * lam;
        for (int m = 0; m < M; ++m) {
            double acc = 0.0;
            int seg_done = 0, in_seg = 0;
            float vnext32 = 0.0f;
            double vnext64 = 0.0;
            for (int t = T - 1; t >= 0; --t) {
                size_t ix = (size_t)t * M + m;
                uint8_t f = flags[ix];
                if (!(f & COPO_F_ACTED)) { A[ix] = 0.0f; G[ix] = 0.0f; in_seg = 0; continue; }
                if (!in_seg || (f & COPO_F_DONE)) { /* last row of a trajectory */
                    seg_done = (f & COPO_F_DONE) != 0;
                    acc = 0.0;
                    vnext32 = Vv[ix];
                    vnext64 = 0.0;
                    in_seg = 1;
                }
                double delta;
                if (seg_done) delta = ((double)R[ix] + g64 * vnext64) - (double)Vv[ix];
                else { float d32 = (R[ix] + g32 * vnext32) - Vv[ix]; delta = (double)d32; }
                acc = delta + c * acc;
                A[ix] = (float)acc;
                G[ix] = (float)(acc + (double)Vv[ix]);
                vnext32 = Vv[ix];
                vnext64 = (double)Vv[ix];
            }
        }
    }
    return COPO_OK;
}

int oracle_cc_fuse_mf(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                      const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                      int32_t counterfactual, float* cc) {
    int C = 2 * O + (counterfactual ? A : 0);
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < N; ++n) {
            size_t row = (size_t)r * N + n;
            float* o = cc + row * C;
            for (int k = 0; k < C; ++k) o[k] = 0.0f;
            if (!(flags[row] & COPO_F_ACTED)) continue;
            for (int k = 0; k < O; ++k) o[k] = obs[row * O + k];
            int m = cnt[row] < K ? cnt[row] : K, got = 0;
            /* np.mean over a list of fp32 rows: pairwise-free for short lists -> sequential fp32 add, then / count */
            for (int q = 0; q < m; ++q) {
                int j = nbr_idx[row * K + q];
                size_t jr = (size_t)r * N + j;
                if (j < 0 || !(flags[jr] & COPO_F_ACTED)) continue;
                for (int k = 0; k < O; ++k) o[O + k] += obs[jr * O + k];
                if (counterfactual)
                    for (int k = 0; k < A; ++k) o[2 * O + k] += act[jr * A + k];
                got++;
            }
            if (got > 0) {
                for (int k = 0; k < O; ++k) o[O + k] = o[O + k] / (float)got;
                if (counterfactual)
                    for (int k = 0; k < A; ++k) o[2 * O + k] = o[2 * O + k] / (float)got;
            }
        }
    return COPO_OK;
}

int oracle_cc_fuse_concat(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                          const int32_t* cnt, int32_t R, int32_t N, int32_t O, int32_t A, int32_t K,
                          int32_t num_neighbours, int32_t counterfactual, float* cc) {
    int other = O + (counterfactual ? A : 0);
    int C = O + num_neighbours * other;
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < N; ++n) {
            size_t row = (size_t)r * N + n;
            float* o = cc + row * C;
            for (int k = 0; k < C; ++k) o[k] = 0.0f;
            if (!(flags[row] & COPO_F_ACTED)) continue;
            for (int k = 0; k < O; ++k) o[k] = obs[row * O + k];
            int m = cnt[row] < K ? cnt[row] : K;
            if (m > num_neighbours) m = num_neighbours;
            for (int q = 0; q < m; ++q) { /* slot = rank in the neighbour list, not compacted */
                int j = nbr_idx[row * K + q];
                size_t jr = (size_t)r * N + j;
                if (j < 0 || !(flags[jr] & COPO_F_ACTED)) continue;
                float* d = o + O + q * other;
                for (int k = 0; k < O; ++k) d[k] = obs[jr * O + k];
                if (counterfactual)
                    for (int k = 0; k < A; ++k) d[O + k] = act[jr * A + k];
            }
        }
    return COPO_OK;
}

/* coordinated advantage + standardisation; stats = {n, sum, sumsq} of A_c then of glob_adv (doubles) */
int oracle_lcf_mix_partial(const float* adv, const float* nei_adv, const float* glob_adv, const float* lcf,
                           const uint8_t* valid, int64_t B, float* mixed, double* stats) {
    for (int k = 0; k < 6; ++k) stats[k] = 0.0;
    for (int64_t i = 0; i < B; ++i) {
        if (valid && !valid[i]) { mixed[i] = 0.0f; continue; }
        float ang = lcf[i] * HALF_PI_F; /* fp32 step_lcf * np.pi / 2 stays fp32 (python scalars are weak) */
        float sn, cs;
        o_sincosf(ang, &sn, &cs);
        float m = cs * adv[i] + sn * nei_adv[i];
        mixed[i] = m;
        stats[0] += 1.0; stats[1] += (double)m; stats[2] += (double)m * (double)m;
        stats[3] += 1.0; stats[4] += (double)glob_adv[i]; stats[5] += (double)glob_adv[i] * (double)glob_adv[i];
    }
    return COPO_OK;
}

int oracle_lcf_mix_apply(const float* mixed, const float* glob_adv, const uint8_t* valid, int64_t B,
                         const double* stats, float* norm_adv, float* glob_std) {
    double m0 = stats[1] / stats[0], v0 = stats[2] / stats[0] - m0 * m0;
    double m1 = stats[4] / stats[3], v1 = stats[5] / stats[3] - m1 * m1;
    double s0 = sqrt(v0 > 0 ? v0 : 0), s1 = sqrt(v1 > 0 ? v1 : 0);
    if (s0 < 1e-4) s0 = 1e-4;
    if (s1 < 1e-4) s1 = 1e-4;
    for (int64_t i = 0; i < B; ++i) {
        if (valid && !valid[i]) { norm_adv[i] = 0.0f; glob_std[i] = 0.0f; continue; }
        norm_adv[i] = (float)(((double)mixed[i] - m0) / s0);
        glob_std[i] = (float)(((double)glob_adv[i] - m1) / s1);
    }
    return COPO_OK;
}

int oracle_version(void) { return COPO_ABI_VERSION; }

/* test hook: the funnel width function of Merge / Split roads at arc length sl of road record g */
float oracle_funnel_extra(const float* g, float sl, float w) { return funnel_extra(g, sl, w); }

// Device helpers shared by the simulator kernels (sim_kernels.hip: one scene per workgroup; sim_packed.hip: several scenes per
// workgroup with the per-agent phases packed densely over the lanes).  Everything here is part of the deterministic spec
// (DESIGN.md section 3): individually rounded IEEE operations, the same expressions in the same order as oracle/copo_oracle.c.
#pragma once
#include "sim_common.h"

// profiling builds only (`make prof SKIP=<mask>`, sim_kernels.hip / sim_packed.hip): phases compiled out; 0 in the shipped library
#ifndef COPO_PROFILE_SKIP
#define COPO_PROFILE_SKIP 0
#endif

namespace copo {

__device__ __host__ inline int ray_lds_words(int n_lasers) { return (2 * n_lasers + 3) & ~3; }      // LDS copy of the ray table
constexpr uint32_t NBR_SENT = 0xffffffffu;      // empty key of the register formulation of the neighbour lists

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Projection of (x, y) with heading (ch, sh) on road g (its lane-0 line): arc length from the road's start, lateral
// offset (left +) and sin(heading - lane direction).  Arcs measure the angle from their mid point (g[14], g[15]).
__device__ __forceinline__ void project_seg(const float* __restrict__ g, float x, float y, float ch, float sh, float& sl,
                                            float& lat, float& sinpsi) {
    const float gx = g[0], gy = g[1], gc = g[2], gs = g[3], kap = g[5];
    const float dx = x - gx, dy = y - gy;
    if (kap == 0.0f) {
        sl = dx * gc + dy * gs;
        lat = dy * gc - dx * gs;
        sinpsi = sh * gc - ch * gs;
    } else {
        const float sg = kap > 0.0f ? 1.0f : -1.0f;
        const float R = g[12];
        const float cx = gx - sg * R * gs, cy = gy + sg * R * gc;
        const float ex = x - cx, ey = y - cy;
        const float rho = sqrtf(ex * ex + ey * ey);
        const float umx = g[14], umy = g[15];
        const float dotp = umx * ex + umy * ey;
        const float crs = umx * ey - umy * ex;
        const float ang = atan2_det(sg * crs, dotp);
        sl = ang * R + 0.5f * g[4];
        lat = sg * (R - rho);
        sinpsi = rho > 0.0f ? (-sg * (ch * ex + sh * ey)) / rho : 0.0f;
    }
}

// Extra drivable width to the right of a straight road of a Merge / Split block at arc length sl (record fields 12 wave radius R,
// 14 extra width D at the wide end: + narrowing / - widening, 15 hand-over point of the edge line's two arcs, measured from the
// wide end): the outer edge of the outermost wave lane, two arcs of radius R + w / 2 and R - w / 2 (maps.Net.add_funnel).
__device__ __forceinline__ float funnel_extra(const float* __restrict__ g, float sl, float w) {
    const float R = g[12];
    if (g[5] != 0.0f || R == 0.0f) return 0.0f;
    const float L = g[4], Ds = g[14], u1 = g[15];
    const float D = fabsf(Ds);
    float u = Ds > 0.0f ? sl : L - sl;             // distance from the wide end
    u = u < 0.0f ? 0.0f : (u > L ? L : u);
    if (u <= u1) {
        const float R1 = R + 0.5f * w;
        return D - (R1 - sqrtf(R1 * R1 - u * u));
    }
    const float R2 = R - 0.5f * w, v = L - u;
    return R2 - sqrtf(R2 * R2 - v * v);
}

// SAT overlap of two oriented boxes (centre, heading unit vector, half length, half width each)
__device__ __forceinline__ bool obb_overlap2(float xi, float yi, float ci, float si, float ai, float bi, float xj, float yj,
                                             float cj, float sj, float aj, float bj) {
    const float dx = xj - xi, dy = yj - yi;
    const float cc = fabsf(ci * cj + si * sj), ss = fabsf(ci * sj - si * cj);
    if (fabsf(dx * ci + dy * si) > ai + aj * cc + bj * ss) return false;
    if (fabsf(dy * ci - dx * si) > bi + aj * ss + bj * cc) return false;
    if (fabsf(dx * cj + dy * sj) > aj + ai * cc + bi * ss) return false;
    if (fabsf(dy * cj - dx * sj) > bj + ai * ss + bi * cc) return false;
    return true;
}

// Slot state held in the registers of lane n of wave 0.
//   status word: status | timer << 8 | age << 16;  spawncnt word: spawn count | toll wait << 16
struct Slot {
    float x, y, th, v, steer, throttle, psteer, pthrottle, yawrate, prog, lcf, eprew;
    int32_t route, status, aid, spawncnt;
    float hc, hs;      // heading unit vector of the step (registers only): sincos(th) / rotated through the sub-steps / the spawn road's
};
__device__ __forceinline__ int st_status(int32_t w) { return w & 0xff; }
__device__ __forceinline__ int st_timer(int32_t w) { return (w >> 8) & 0xff; }
__device__ __forceinline__ int st_age(int32_t w) { return (int)((uint32_t)w >> 16); }
__device__ __forceinline__ int32_t st_pack(int st, int tm, int age) {
    return (int32_t)((uint32_t)st | ((uint32_t)tm << 8) | ((uint32_t)age << 16));
}

__device__ __forceinline__ void load_slot(const SimParams& p, int e, int n, Slot& s) {
    const size_t EN = (size_t)p.E * p.N, o = (size_t)e * p.N + n;
    const float* st = p.state;
    s.x = st[0 * EN + o]; s.y = st[1 * EN + o]; s.th = st[2 * EN + o]; s.v = st[3 * EN + o];
    s.steer = st[4 * EN + o]; s.throttle = st[5 * EN + o]; s.psteer = st[6 * EN + o]; s.pthrottle = st[7 * EN + o];
    s.yawrate = st[8 * EN + o]; s.prog = st[9 * EN + o]; s.lcf = st[10 * EN + o]; s.eprew = st[11 * EN + o];
    const int32_t* si = reinterpret_cast<const int32_t*>(st);
    s.route = si[12 * EN + o]; s.status = si[13 * EN + o]; s.aid = si[14 * EN + o]; s.spawncnt = si[15 * EN + o];
}

__device__ __forceinline__ void store_slot(const SimParams& p, int e, int n, const Slot& s) {
    const size_t EN = (size_t)p.E * p.N, o = (size_t)e * p.N + n;
    float* st = p.state;
    st[0 * EN + o] = s.x; st[1 * EN + o] = s.y; st[2 * EN + o] = s.th; st[3 * EN + o] = s.v;
    st[4 * EN + o] = s.steer; st[5 * EN + o] = s.throttle; st[6 * EN + o] = s.psteer; st[7 * EN + o] = s.pthrottle;
    st[8 * EN + o] = s.yawrate; st[9 * EN + o] = s.prog; st[10 * EN + o] = s.lcf; st[11 * EN + o] = s.eprew;
    int32_t* si = reinterpret_cast<int32_t*>(st);
    si[12 * EN + o] = s.route; si[13 * EN + o] = s.status; si[14 * EN + o] = s.aid; si[15 * EN + o] = s.spawncnt;
}

// pose of spawn slot sp: lane `stab[sp][2]` of the spawn road (road 0 of its routes), `sps[sp]` metres in
__device__ __forceinline__ void spawn_pose(const SimParams& p, const float* rsegs, const int32_t* stab, const float* sps,
                                           int sp, float& x, float& y) {
    const float* g = rsegs + (size_t)stab[sp * 4 + 0] * p.seg_rows * COPO_SEG_STRIDE;
    const float s0 = sps[sp];
    const float off = (float)stab[sp * 4 + 2] * p.lane_width;
    x = g[0] + g[2] * s0 + g[3] * off;
    y = g[1] + g[3] * s0 - g[2] * off;
}

// Spawn a fresh agent into this lane's slot at spawn slot sp.  `aid` is the env-wide id.
// the random draws of the `cnt`-th spawn in slot n: route hash, LCF sample (LCFEnv._add_lcf: normal(mean, std) clipped to [-1, 1])
__device__ __forceinline__ void spawn_draws(const SimParams& p, uint64_t seed, uint32_t episode, int n, uint32_t cnt,
                                            uint32_t& h_route, float& lcf) {
    h_route = hash_rng(seed, (uint32_t)n, cnt, episode, RNG_ROUTE);
    lcf = 0.0f;
    if (p.enable_lcf) {
        const float u1 = uniform01(hash_rng(seed, (uint32_t)n, cnt, episode, RNG_LCF1));
        const float u2 = uniform01(hash_rng(seed, (uint32_t)n, cnt, episode, RNG_LCF2));
        float sn, cs;
        sincos_det(kTwoPi * u2 - kPi, sn, cs);
        const float z = sqrtf(-2.0f * log_det(u1)) * cs;
        lcf = clipf(p.lcf_dist[0] + p.lcf_dist[1] * z, -1.0f, 1.0f);
    }
}

// Route of a spawn at place sp: the (h mod count)-th of the routes that start there.  With exclusive destinations
// (route_meta[.][3] = id + 1; MetaDrive's ParkingSpaceManager: a parking space is the goal of one living vehicle at a time) it is
// the (h mod free)-th of those whose space is not in `taken`, in table order; all of them when none is free.  The chosen space
// joins `taken`.  Uniform over the wave (every lane evaluates it for the slot being served).
__device__ __forceinline__ int pick_route_exclusive(const float* rmeta, const int32_t* stab, int sp, uint32_t h, uint32_t& taken) {
    const int first = stab[sp * 4 + 0], count = stab[sp * 4 + 1];
    int route = first + (int)(h % (uint32_t)count);
    if (rmeta[first * 4 + 3] > 0.0f) {
        int nfree = 0;
        for (int k = 0; k < count; ++k) {
            const int d = (int)rmeta[(first + k) * 4 + 3];
            if (!(d > 0 && ((taken >> (d - 1)) & 1u))) nfree += 1;
        }
        if (nfree > 0) {
            int pick = (int)(h % (uint32_t)nfree);
            for (int k = 0; k < count; ++k) {
                const int d = (int)rmeta[(first + k) * 4 + 3];
                if (d > 0 && ((taken >> (d - 1)) & 1u)) continue;
                if (pick == 0) { route = first + k; break; }
                --pick;
            }
        }
    }
    const int d = (int)rmeta[route * 4 + 3];
    if (d > 0) taken |= 1u << (d - 1);
    return route;
}
// the spaces the living vehicles of the scene are heading for (one lane per slot)
__device__ __forceinline__ uint32_t spaces_taken(const SimParams& p, const float* rmeta, bool alive, int route_word) {
    const int d = alive ? (int)rmeta[(route_word & 0xffff) * 4 + 3] : 0;
    uint32_t taken = 0;
    for (int k = 1; k <= p.n_spaces; ++k)
        if (__ballot(d == k) != 0ull) taken |= 1u << (k - 1);
    return taken;
}

// `pre`: the draws were made ahead of time (step kernel, several waves per scene: a wave that idles during P0 makes them for
// every slot, so that a spawn costs wave 0 -- the critical path of the launch -- two LDS reads instead of ~200 instructions)
__device__ __forceinline__ void spawn_slot(const SimParams& p, const float* rsegs, const int32_t* stab, const float* sps,
                                           uint64_t seed, uint32_t episode, int n, int sp, int32_t aid, Slot& s,
                                           bool pre = false, uint32_t pre_h = 0, float pre_lcf = 0.0f, int route_fixed = -1) {
    const uint32_t cnt = (uint32_t)s.spawncnt & 0xffffu;
    uint32_t h = pre_h;
    float lcf = pre_lcf;
    if (!pre) spawn_draws(p, seed, episode, n, cnt, h, lcf);
    const int route = route_fixed >= 0 ? route_fixed : stab[sp * 4 + 0] + (int)(h % (uint32_t)stab[sp * 4 + 1]);
    const float* g = rsegs + (size_t)route * p.seg_rows * COPO_SEG_STRIDE;
    spawn_pose(p, rsegs, stab, sps, sp, s.x, s.y);
    s.th = g[7];
    s.hc = g[2]; s.hs = g[3];          // a fresh vehicle stands along its spawn road
    s.v = 0.0f; s.steer = 0.0f; s.throttle = 0.0f; s.psteer = 0.0f; s.pthrottle = 0.0f; s.yawrate = 0.0f;
    s.prog = sps[sp]; s.eprew = 0.0f;
    s.route = route;
    s.status = st_pack(ST_ALIVE, 0, 0);
    s.aid = aid;
    s.lcf = lcf;
    s.spawncnt = (int32_t)((cnt + 1) & 0xffffu);
}

// active agent slots: device memory next to the LCF distribution, so that captured graphs see updates
__device__ __forceinline__ int capacity_of(const SimParams& p) {
    const int c = (int)p.lcf_dist[2];
    return c < 1 ? 1 : (c > p.N ? p.N : c);
}

// median of three unsigned values (compiles to v_med3_u32)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
    const uint32_t t = mn > c ? mn : c;
    return t < mx ? t : mx;
}

__device__ __forceinline__ float readlane_f(float v, int lane_uniform) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// The lists of ONE agent i (wave-uniform) by one wave, lane = slot j, with the reference's own expressions: fp64 distance of
// every present j, in range / mean-field range on it, rank by (d, slot), the rewards added in list order
// (env_wrappers.py:321-325; neighbours_phase does the same for all agents at once).  `odd`: slots whose reward is outside
// the range in which sums are exact in any order -- only with one of those in range are the rewards brought into list
// order (a cross-lane push by rank) before they are added; otherwise slot order gives the same bits.
__device__ __forceinline__ void neighbours_exact_one(const SimParams& p, int e, int lane, int i, float xl, float yl, float rwl,
                                                     unsigned long long present, unsigned long long odd, const StepOut& out) {
    const int N = p.N, K = p.K;
    const double R = (double)p.neighbours_distance, M = (double)p.mf_distance;
    const float xi = readlane_f(xl, i), yi = readlane_f(yl, i);
    const double dx = (double)xi - (double)xl, dy = (double)yi - (double)yl;
    const double d = sqrt(dx * dx + dy * dy);
    const bool inr = ((present >> lane) & 1ull) && lane != i && d < R;
    const unsigned long long mi = __ballot(inr);
    const int cnt = __popcll(mi);
    const int mf = __popcll(__ballot(inr && d <= M));
    const int dlo = __double2loint(d), dhi = __double2hiint(d);
    // rank = in-range slots that sort before this one by (d, slot).  Distances are >= 0: the upper word of the fp64 pattern
    // orders them except when two upper words agree (distances within 1e-6 of each other) -- only then the full compare
    int rank = 0;
    for (unsigned long long m = mi; m; m &= m - 1ull) {
        const int k = __ffsll((long long)m) - 1;
        const unsigned int kh = (unsigned int)__builtin_amdgcn_readlane(dhi, k);
        rank += kh < (unsigned int)dhi ? 1 : 0;
        if (__ballot(inr && kh == (unsigned int)dhi && k != lane) != 0ull) {
            const double dk = __hiloint2double((int)kh, __builtin_amdgcn_readlane(dlo, k));
            rank += (kh == (unsigned int)dhi && (dk < d || (dk == d && k < lane))) ? 1 : 0;
        }
    }
    const size_t row = ((size_t)e * N + i) * K;
    if (inr && rank < K) {
        if (out.nbr_idx) out.nbr_idx[row + rank] = lane;
        if (out.nbr_dist) out.nbr_dist[row + rank] = (float)d;
    }
    if (lane >= cnt && lane < K) {
        if (out.nbr_idx) out.nbr_idx[row + lane] = -1;
        if (out.nbr_dist) out.nbr_dist[row + lane] = 0.0f;
    }
    double nsum = 0.0;
    if (out.nei_rew) {
        if (mi & odd) {       // list order matters: lane `rank` receives the reward of this lane (lanes out of range push to lane 63)
            const int byrank = __builtin_amdgcn_ds_permute((inr ? rank : 63) << 2, __float_as_int(rwl));
            for (int r = 0; r < cnt; ++r) nsum += (double)__int_as_float(__builtin_amdgcn_readlane(byrank, r));
        } else {
            for (unsigned long long m = mi; m; m &= m - 1ull) nsum += (double)readlane_f(rwl, __ffsll((long long)m) - 1);
        }
    }
    if (lane == 0) {
        if (out.nbr_cnt) out.nbr_cnt[(size_t)e * N + i] = cnt;
        if (out.mf_cnt) out.mf_cnt[(size_t)e * N + i] = mf;
        if (out.nei_rew) out.nei_rew[(size_t)e * N + i] = cnt ? (float)(nsum / (double)cnt) : 0.0f;
    }
}

// segment record k of a route (COPO_SEG_STRIDE floats), through whichever copy of the tables the caller uses (`L`: anything with
// the members rsegs / seg_rows)
template <class LT>
__device__ __forceinline__ const float* seg_ptr(const LT& L, int route, int k) {
    return L.rsegs + ((size_t)route * L.seg_rows + k) * COPO_SEG_STRIDE;
}

// State + navigation blocks of the observation of this lane's slot (MetaDrive 0.2.5 StateObservation.vehicle_state +
// Navigation._get_info_for_checkpoint), written straight to the slot's observation row; the detector / LiDAR columns
// are filled by obs_phase.  `counter` = env steps since the last reset (the traffic-light clock,
// env_wrappers.py:258-265,280,317).
// (`L`: the route tables -- members rsegs / rmeta / seg_rows; `cs`, `sn`: the heading unit vector of the slot's staged pose)
template <bool EXT, class LT>
__device__ __forceinline__ void ego_navi_obs(const SimParams& p, const LT& L, float cs, float sn, const Slot& s, bool present,
                                             float* __restrict__ row, int counter, bool zero_comm) {
    if (!row || !present) return;     // the observation row of an absent slot is not written (copo_step_out.obs)
    if (EXT && zero_comm && p.col_comm >= 0) {   // reset observation: no messages (the neighbour phase of a step that ends an
        const int n = p.comm_nb * (p.comm_size + 3 * p.comm_pos);     // episode ran on the scene BEFORE the reset)
        for (int k = 0; k < n; ++k) row[p.col_comm + k] = 0.0f;
    }
    if (EXT && p.col_tl >= 0) {   // clip([message, x', y'], 0, 1) in python float64 arithmetic, cast to fp32
        const int I = p.tl_interval;
        const double inc = (double)(counter % I) / (double)I * 0.1;
        const double msg = (((counter / I) % 2) == 1) ? 0.0 + inc : 1.0 - inc;
        const double b0 = (double)p.bbox[0], b1 = (double)p.bbox[1], b2 = (double)p.bbox[2], b3 = (double)p.bbox[3];
        const double v[3] = {msg, ((double)s.x - b0) / (b1 - b0), ((double)s.y - b2) / (b3 - b2)};
#pragma unroll
        for (int k = 0; k < 3; ++k) row[p.col_tl + k] = (float)(v[k] < 0.0 ? 0.0 : (v[k] > 1.0 ? 1.0 : v[k]));
    }
    const int route = s.route & 0xffff, seg = s.route >> 16;
    const float* meta = L.rmeta + route * 4;
    const int nseg = (int)meta[1];
    const float* g = seg_ptr(L, route, seg);
    float sl, lat, sinpsi;
    project_seg(g, s.x, s.y, cs, sn, sl, lat, sinpsi);
    const float w = p.lane_width;
    const float lanes = floorf(g[COPO_SEG_LANES]);
    float lif = floorf(0.5f - lat * p.inv_w);
    lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
    const float left = 0.5f * w - lat;
    const float right = lanes * w - left;
    if (p.side_lasers == 0) {
        const float tw = (lanes + 1.0f) * w;
        row[0] = clipf(left / tw, 0.0f, 1.0f);
        row[1] = clipf(right / tw, 0.0f, 1.0f);
    }
    float* q = row + p.col_state;
    q[0] = clipf(0.5f - 0.5f * sinpsi, 0.0f, 1.0f);
    q[1] = clipf((fabsf(s.v) * 3.6f + 1.0f) * p.inv_vnorm, 0.0f, 1.0f);      // vehicle.speed is a magnitude
    q[2] = clipf(0.5f + s.steer * (1.0f / 120.0f), 0.0f, 1.0f);
    q[3] = clipf(0.5f + 0.5f * s.psteer, 0.0f, 1.0f);
    q[4] = clipf(0.5f + 0.5f * s.pthrottle, 0.0f, 1.0f);
    q[5] = clipf(fabsf(s.yawrate), 0.0f, 1.0f);
    if (p.lane_lasers == 0) {
        const float latr = -(lat + lif * w);
        row[p.col_lane] = clipf(0.5f + latr * (1.0f / 4.5f), 0.0f, 1.0f);
    }
    if (p.navi_dim) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int kk = seg + j;
            if (kk > nseg - 1) kk = nseg - 1;
            const float* gk = seg_ptr(L, route, kk);
            float ckx = gk[COPO_SEG_CKX], cky = gk[COPO_SEG_CKX + 1];
            if (floorf(gk[COPO_SEG_LANES]) != lanes) {
                // Navigation._get_info_for_checkpoint puts BOTH check points at the lateral middle of the CURRENT road's lane
                // count ((get_current_lane_num() / 2 - 0.5) * width to the right of the checked road's lane 0): where the lane
                // count changes (Merge / Split blocks) the next check point is not the middle of its own road
                const float* gn = seg_ptr(L, route, kk + 1);          // start of the next record = end of road kk, lane 0
                const float off = (lanes * 0.5f - 0.5f) * w;
                ckx = gn[0] + gn[3] * off;
                cky = gn[1] - gn[2] * off;
            }
            float vx = ckx - s.x, vy = cky - s.y;
            const float nrm = sqrtf(vx * vx + vy * vy);
            if (nrm > 50.0f) {
                const float sc = 50.0f / nrm;
                vx = vx * sc;
                vy = vy * sc;
            }
            const float fwd = vx * cs + vy * sn, rhs = vx * sn - vy * cs;
            float* n5 = row + p.col_navi + 5 * j;
            const float kap = gk[5];
            n5[0] = clipf(0.5f + fwd * 0.01f, 0.0f, 1.0f);
            n5[1] = clipf(0.5f + rhs * 0.01f, 0.0f, 1.0f);
            n5[2] = gk[COPO_SEG_FEAT];
            n5[3] = kap == 0.0f ? 0.5f : (kap < 0.0f ? 1.0f : 0.0f);
            n5[4] = gk[COPO_SEG_FEAT + 2];
        }
    }
    if (p.toll_dim) {
        const uint32_t wait = (uint32_t)s.spawncnt >> 16;
        const bool in_booth = seg == (int)meta[2];      // [on the booth road, stayed longer than toll_min_steps], zeros off it
        row[p.col_toll] = in_booth ? 1.0f : 0.0f;
        row[p.col_toll + 1] = (in_booth && wait > (uint32_t)p.toll_min_steps) ? 1.0f : 0.0f;
    }
    if (p.col_lcf >= 0) row[p.col_lcf] = (s.lcf + 1.0f) * 0.5f;
}

// One detector beam against the lane-line primitives (MetaDrive SideDetector / LaneLineDetector): see the oracle's
// detector_ray for the arithmetic, which this repeats operation by operation.
__device__ __forceinline__ float detector_ray(const SimParams& p, const float* __restrict__ lines, float x, float y, float dx,
                                              float dy, float range, float min_kind) {
    float best = range;
    for (int l = 0; l < p.n_lines; ++l) {
        const float* Ln = lines + (size_t)l * COPO_LINE_STRIDE;
        if (Ln[0] < min_kind) continue;
        if (Ln[6] == 0.0f) {
            const float rx = Ln[1] - x, ry = Ln[2] - y;
            const float den = dx * Ln[4] - dy * Ln[3];
            if (den == 0.0f) continue;
            const float sd = den > 0.0f ? 1.0f : -1.0f;
            const float ad = den * sd;
            const float tn = (rx * Ln[4] - ry * Ln[3]) * sd;
            const float un = (rx * dy - ry * dx) * sd;
            if (!(tn >= 0.0f && un >= 0.0f && un <= Ln[5] * ad && tn < best * ad)) continue;
            best = tn / ad;
        } else {
            const float R = 1.0f / fabsf(Ln[6]);
            const float mx = x - Ln[7], my = y - Ln[8];
            const float b = mx * dx + my * dy;
            const float cq = mx * mx + my * my - R * R;
            const float disc = b * b - cq;
            if (!(disc >= 0.0f)) continue;
            const float sq = sqrtf(disc);
            for (int r = 0; r < 2; ++r) {
                const float tt = r == 0 ? -b - sq : -b + sq;
                if (!(tt >= 0.0f && tt < best)) continue;
                const float hx = mx + tt * dx, hy = my + tt * dy;
                if (hx * Ln[9] + hy * Ln[10] >= R * Ln[11]) { best = tt; break; }
            }
        }
    }
    return best;
}

// Ray against the box of vehicle j, in j's box frame: entering distance, or a negative value for a miss.  Box frame mirrored so that the direction is non-negative on both axes; entering / exiting
// times are fractions n/a compared by cross-multiplication, one IEEE division only for an actual hit (spec 3.4-9).
__device__ __forceinline__ float ray_box(float ox, float oy, float ddx, float ddy, float hl, float hw) {
    // (ox, oy): ray origin, (ddx, ddy): unit direction, both in the box frame of vehicle j
    const float ax = fabsf(ddx), ay = fabsf(ddy);
    const float oxs = ddx < 0.0f ? -ox : ox, oys = ddy < 0.0f ? -oy : oy;
    const float nxe = -(hl + oxs), nxx = hl - oxs, nye = -(hw + oys), nyx = hw - oys;
    if (!(nxx >= 0.0f && nyx >= 0.0f)) return -1.0f;
    if (!(nxe * ay <= nyx * ax)) return -1.0f;
    if (!(nye * ax <= nxx * ay)) return -1.0f;
    const bool usex = nxe * ay >= nye * ax;
    const float n = usex ? nxe : nye, a = usex ? ax : ay;
    return n > 0.0f ? n / a : 0.0f;
}

// n / a, correctly rounded, for the operands ray_box_nr meets: the Newton-Raphson chain of the compiler's own fp32 division
// (v_rcp_f32, two refinements of the reciprocal and of the quotient, all in fused multiply-adds) WITHOUT its range scaling
// (v_div_scale / v_div_fmas / v_div_fixup).  The scaling only acts on denormal operands, exponents near the ends of the range or
// quotients that over- / underflow; for every other operand pair the two chains execute the same operations on the same values.
// A hit that can change a ray's minimum has 0 < n / a < lidar_range with n a difference of vehicle-scale coordinates (0 or at
// least 2^-23 in magnitude), so a >= n / range is far from denormal; a quotient beyond the range (or the inf / NaN of a == 0)
// never lowers a minimum that starts at the range, whichever way it is rounded.
__device__ __forceinline__ float div_nr(float n, float a) {
    float y = __builtin_amdgcn_rcpf(a);
    const float e = __builtin_fmaf(-a, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = n * y;
    float r = __builtin_fmaf(-a, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-a, q, n);
    return __builtin_fmaf(r, y, q);
}
// ray_box with that division (sim_packed.hip)
__device__ __forceinline__ float ray_box_nr(float ox, float oy, float ddx, float ddy, float hl, float hw) {
    const float ax = fabsf(ddx), ay = fabsf(ddy);
    const float oxs = ddx < 0.0f ? -ox : ox, oys = ddy < 0.0f ? -oy : oy;
    const float nxe = -(hl + oxs), nxx = hl - oxs, nye = -(hw + oys), nyx = hw - oys;
    if (!(nxx >= 0.0f && nyx >= 0.0f)) return -1.0f;
    if (!(nxe * ay <= nyx * ax)) return -1.0f;
    if (!(nye * ax <= nxx * ay)) return -1.0f;
    const bool usex = nxe * ay >= nye * ax;
    const float n = usex ? nxe : nye, a = usex ? ax : ay;
    return n > 0.0f ? div_nr(n, a) : 0.0f;
}

// Wave64 inclusive scans on the DPP network (row shifts 1/2/4/8, then row_bcast:15 / :31 -- the gfx9 sequence):
// six VALU operations, no LDS traffic (a __shfl_up ladder is six ds_bpermute round trips).
template <bool MAX>
__device__ __forceinline__ int wave_scan_incl(int v) {
#define COPO_SCAN_STEP(ctrl, rmask)                                                       \
    {                                                                                     \
        const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false);         \
        v = MAX ? (t > v ? t : v) : v + t;                                                \
    }
    COPO_SCAN_STEP(0x111, 0xf)   // row_shr:1
    COPO_SCAN_STEP(0x112, 0xf)   // row_shr:2
    COPO_SCAN_STEP(0x114, 0xf)   // row_shr:4
    COPO_SCAN_STEP(0x118, 0xf)   // row_shr:8
    COPO_SCAN_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1, 3
    COPO_SCAN_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2, 3
#undef COPO_SCAN_STEP
    return v;                    // identity 0: counts and (lane + 1) markers are non-negative
}

// Bearing of (u, v) in (-pi, pi], absolute error < 1e-5 rad.  NOT part of the deterministic spec: it only sizes the
// conservative ray window below, every hit/miss decision stays with ray_box.
__device__ __forceinline__ float atan2_window(float v, float u) {
    const float au = fabsf(u), av = fabsf(v);
    const float mx = fmaxf(fmaxf(au, av), 1e-30f), mn = fminf(au, av);
    const float z = mn * __builtin_amdgcn_rcpf(mx), z2 = z * z;
    float q = __builtin_fmaf(z2, -0.01172120f, 0.05265332f);
    q = __builtin_fmaf(z2, q, -0.11643287f);
    q = __builtin_fmaf(z2, q, 0.19354346f);
    q = __builtin_fmaf(z2, q, -0.33262347f);
    q = __builtin_fmaf(z2, q, 0.99997726f);
    q = q * z;
    if (av > au) q = 1.57079633f - q;
    if (u < 0.0f) q = 3.14159265f - q;
    return v < 0.0f ? -q : q;
}

// P0 of one slot: the wreck / cooldown timers, then -- for a slot that holds a driving agent -- the action and the kinematic bicycle
// over the sub-steps.  `o` = e * N + n, the slot's row in the action tensor.  Sets s.hc / s.hs (the heading unit vector of the
// step: the start heading turned through the sub-steps for an acting slot, sincos(th) otherwise).
template <bool EXT>
__device__ __forceinline__ void slot_dynamics(const SimParams& p, const float* __restrict__ act, size_t o, Slot& s, bool& acted, float& acc) {
    const int st = st_status(s.status);
    int tm = st_timer(s.status);
    acted = (st == ST_ALIVE);
    if (st == ST_WRECK) {
        tm -= 1;
        s.status = (tm <= 0) ? st_pack(ST_EMPTY, p.respawn_cooldown, 0) : st_pack(ST_WRECK, tm, 0);
    } else if (st == ST_EMPTY && tm > 0) {
        s.status = st_pack(ST_EMPTY, tm - 1, 0);
    }
    sincos_det(s.th, s.hs, s.hc);     // the one sincos of the step: start heading of acting slots, pose heading of wrecks
    if (acted) {
        float a0, a1;
        if (EXT) {
            const float* ap = act + (o) * p.act_dim;
            a0 = ap[0]; a1 = ap[1];
        } else {
            const float2 a = reinterpret_cast<const float2*>(act)[o];
            a0 = a.x; a1 = a.y;
        }
        if (!(a0 == a0)) a0 = 0.0f;
        if (!(a1 == a1)) a1 = 0.0f;
        a0 = clipf(a0, -1.0f, 1.0f);
        a1 = clipf(a1, -1.0f, 1.0f);
        const float delta = a0 * p.max_steer;
        float sd, cd;
        sincos_det(delta, sd, cd);
        const float tand = sd / cd;
        const float tb = 0.5f * tand;
        const float cb = 1.0f / sqrtf(1.0f + tb * tb), sb = tb * cb;
        const float yawk = (tand / p.wheelbase) * cb;
        float brake = -a1 * p.brake_gain;
        if (brake > p.brake_max) brake = p.brake_max;
        const float h = p.h_sub;
        float x = s.x, y = s.y, th = s.th, v = s.v;
        float cs = s.hc, sn = s.hs;
        const float v0 = v, th0 = th;
        for (int k = 0; k < p.substeps; ++k) {
            // (reverse gear, MetaDrive enable_reverse: a negative throttle is engine force backwards, no brake, v may go negative;
            //  the engine is cut at max_speed in either direction)
            const float a = a1 >= 0.0f ? (v < p.max_speed ? a1 * p.acc_max : 0.0f) : (p.reverse_acc > 0.0f ? (v > -p.max_speed ? a1 * p.reverse_acc : 0.0f) : -brake);
            v = v + a * h;
            if (v < 0.0f && !(p.reverse_acc > 0.0f)) v = 0.0f;
            const float dxh = cs * cb - sn * sb, dyh = sn * cb + cs * sb;
            x = x + v * dxh * h;
            y = y + v * dyh * h;
            // turn the heading vector by the sub-step's small angle: 3-term sine / cosine, no range reduction
            float dth = v * yawk * h;
            if (p.lat_acc_max > 0.0f && v * fabsf(dth) > p.lat_acc_max * h) {      // tyres slide: v x yaw rate is friction-limited
                const float lim = (p.lat_acc_max * h) / v;
                dth = dth < 0.0f ? -lim : lim;
            }
            const float q = dth * dth;
            const float sd2 = dth - dth * q * (0.166666667f - q * 0.00833333333f);
            const float cd2 = 1.0f - q * (0.5f - q * 0.0416666667f);
            const float cn = cs * cd2 - sn * sd2, sm = sn * cd2 + cs * sd2;
            cs = cn;
            sn = sm;
            th = wrap_pi(th + dth);
        }
        s.x = x; s.y = y; s.th = th; s.v = v;
        s.hc = cs; s.hs = sn;
        s.psteer = s.steer; s.pthrottle = s.throttle;
        s.steer = a0; s.throttle = a1;
        s.yawrate = wrap_pi(th - th0) * p.inv_dt;
        acc = (v - v0) * p.inv_dt;
        s.status = st_pack(ST_ALIVE, 0, st_age(s.status) + 1);
    }
}

// P2 of one ACTING slot: projection on its route, arrival / out of road / crash / max_step, reward, info row, status after a termination.
// (`L`: the route tables; (ch, sh): the slot's heading unit vector after P0; crash_in: a collision pair of P1 overlapped)
template <class LT>
__device__ __forceinline__ void slot_project(const SimParams& p, const LT& L, float ch, float sh, bool crash_in, bool force_end, float acc,
                                             float* __restrict__ info_row, Slot& s, uint8_t& fl, float& rew, bool& term) {
    const float hl = p.hl, hw = p.hw;
    const int route = s.route & 0xffff;
    const int seg_before = s.route >> 16;
    int seg = seg_before;
    const float* meta = L.rmeta + route * 4;
    const float total = meta[0];
    const int nseg = (int)meta[1];
    const float* g = seg_ptr(L, route, seg);
    float sl, lat, sinpsi;
    project_seg(g, s.x, s.y, ch, sh, sl, lat, sinpsi);
    for (int it = 0; it < 2; ++it) {
        if (sl > g[4] && seg < nseg - 1) {
            seg += 1;
            g = seg_ptr(L, route, seg);
            project_seg(g, s.x, s.y, ch, sh, sl, lat, sinpsi);
        }
    }
    if (sl < 0.0f && seg > 0) {
        seg -= 1;
        g = seg_ptr(L, route, seg);
        project_seg(g, s.x, s.y, ch, sh, sl, lat, sinpsi);
    }
    const float prog = g[6] + sl;
    const float prev = s.prog;
    bool too_fast = false;
    if (p.toll_dim) {
        const int toll_seg = (int)meta[2];
        const uint32_t sc = (uint32_t)s.spawncnt;
        uint32_t wait = sc >> 16;
        if (seg == toll_seg && wait < 0xffffu) wait += 1;
        too_fast = toll_seg >= 0 && seg > toll_seg && seg_before <= toll_seg && wait < (uint32_t)p.toll_min_steps;
        s.spawncnt = (int32_t)((sc & 0xffffu) | (wait << 16));
    }
    s.route = route | (seg << 16);
    s.prog = prog;
    const float w = p.lane_width;
    const float lanes_f = g[COPO_SEG_LANES], lanes = floorf(lanes_f), lfr = lanes_f - lanes;      // fraction: edge-line flags
    const int lcode = (int)(lfr * 8.0f);      // edge-line flags in eighths: 1 = left edge open (broken centre line), 2 / 4 = left / right edge solid
    const bool left_solid = (lcode & 2) != 0, right_solid = (lcode & 4) != 0, left_open = (lcode & 1) != 0;
    float lif = floorf(0.5f - lat * p.inv_w);
    lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
    const float left = 0.5f * w - lat, right = (lanes * w + funnel_extra(g, sl, w)) - left;
    const float cos2 = 1.0f - sinpsi * sinpsi;
    const float edge = p.body_margin * (hw * sqrtf(cos2 > 0.0f ? cos2 : 0.0f) + hl * fabsf(sinpsi));      // body extent across the road
    const bool on_road = (left >= (left_solid ? edge : (left_open ? -w : 0.0f))) && (right >= (right_solid ? edge : 0.0f));
    const bool arrive = (seg == nseg - 1) && (sl > g[4] - p.arrive_margin) && (sl < g[4] + p.arrive_margin) && on_road;
    const bool oor = !on_road;
    const bool crash = crash_in || too_fast;
    float r = p.driving_reward * ((prog - prev) * (1.0f + g[5] * (lif * w))) + p.speed_reward * (fabsf(s.v) / p.max_speed);
    fl = COPO_F_ACTED;
    if (arrive) { r = p.success_reward; fl |= COPO_F_ARRIVE; }
    else if (oor) { r = -p.out_penalty; }
    else if (crash) { r = -p.crash_penalty; }
    if (oor) fl |= COPO_F_OUT;
    if (crash) fl |= COPO_F_CRASH;
    bool done = arrive || oor || crash;
    if (!done && (st_age(s.status) >= p.horizon || force_end)) { fl |= COPO_F_MAXSTEP; done = true; }
    if (done) fl |= COPO_F_DONE;
    term = done;
    rew = r;
    s.eprew += r;
    if (info_row) {
        float* q = info_row;
        q[COPO_I_VELOCITY] = fabsf(s.v) * 3.6f;
        q[COPO_I_STEERING] = s.steer;
        q[COPO_I_ACCELERATION] = acc;
        q[COPO_I_STEP_REWARD] = r;
        q[COPO_I_COST] = crash ? 1.0f : 0.0f;
        q[COPO_I_EPISODE_LENGTH] = (float)st_age(s.status);
        q[COPO_I_EPISODE_REWARD] = s.eprew;
        q[COPO_I_ROUTE_COMPLETION] = clipf(prog / total, 0.0f, 1.0f);
    }
    if (term) {
        if (!(fl & COPO_F_ARRIVE) && p.delay_done > 0)
            s.status = st_pack(ST_WRECK, p.delay_done, 0);
        else
            s.status = st_pack(ST_EMPTY, p.respawn_cooldown, 0);
    }
}

}  // namespace copo

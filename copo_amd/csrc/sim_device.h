// Device helpers shared by the simulator kernels (sim_kernels.hip: one scene per workgroup; sim_packed.hip: several scenes per
// workgroup with the per-agent phases packed densely over the lanes).  Everything here is part of the deterministic spec
// (DESIGN.md section 3): individually rounded IEEE operations, the same expressions in the same order as oracle/copo_oracle.c.
#pragma once
#include "sim_common.h"

// profiling builds only (`make prof SKIP=<mask>`, sim_kernels.hip / sim_packed.hip): phases compiled out; 0 in the shipped library
#ifndef COPO_PROFILE_SKIP
#define COPO_PROFILE_SKIP 0
#endif
// mask 256: nothing compiled out, the LiDAR phase counts its work into the debug rows [E][16] (8 queued pairs, 9 pair batches, 10 box
// tests, 11 test batches, 12 hits, 13 pairs with a ray window); needs `p`, `e`, `lane` in scope
#define COPO_COUNT(slot, v) do { if ((COPO_PROFILE_SKIP & 256) && p.dbg && lane == 0) p.dbg[(size_t)e * 16 + (slot)] += (long long)(v); } while (0)
// mask 16384: the detector beams count theirs into the same slots of row `ecount` (8 candidate batches, 9 near pairs, 10 pair batches,
// 11 beam tests, 12 test batches, 13 hits)
#define COPO_DCOUNT(slot, v) do { if ((COPO_PROFILE_SKIP & 16384) && p.dbg && lane == 0 && ecount >= 0) p.dbg[(size_t)ecount * 16 + (slot)] += (long long)(v); } while (0)

namespace copo {

__device__ __host__ inline int ray_lds_words(int n_lasers) { return (2 * n_lasers + 3) & ~3; }      // LDS copy of the ray table
// lidar_by_wave keeps the ray minima of a fan SHIFTED by up to 3 words, so that the 16-byte groups of rays that the write-out stores
// to an aligned observation address are 16-byte aligned in LDS too (one ds_read_b128 per lane instead of four ds_read_b32 at a
// stride of four words: a 4-way bank conflict); the work area is these words longer and the pair queue starts behind them
constexpr int LIDAR_MIN_PAD = 4;
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
        sl = fm(dx, gc, dy * gs);
        lat = fm(dy, gc, -(dx * gs));
        sinpsi = fm(sh, gc, -(ch * gs));
    } else {
        const float sg = kap > 0.0f ? 1.0f : -1.0f;
        const float R = g[12];
        const float cx = fm(-(sg * R), gs, gx), cy = fm(sg * R, gc, gy);
        const float ex = x - cx, ey = y - cy;
        const float rho = sqrtf(fm(ex, ex, ey * ey));
        const float umx = g[14], umy = g[15];
        const float dotp = fm(umx, ex, umy * ey);
        const float crs = fm(umx, ey, -(umy * ex));
        const float ang = atan2_det(sg * crs, dotp);
        sl = fm(ang, R, 0.5f * g[4]);
        lat = sg * (R - rho);
        sinpsi = rho > 0.0f ? (-sg * fm(ch, ex, sh * ey)) / rho : 0.0f;
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
    // (ss stays two products and a subtraction: a fused ci sj - round(si cj) is not the exact negation of the pair's other order,
    //  and the one-wave / packed shapes test every UNORDERED pair once; everything else is symmetric under the swap as before)
    const float cc = fabsf(fm(ci, cj, si * sj)), ss = fabsf(ci * sj - si * cj);
    if (fabsf(fm(dx, ci, dy * si)) > fm(bj, ss, fm(aj, cc, ai))) return false;
    if (fabsf(fm(dy, ci, -(dx * si))) > fm(bj, cc, fm(aj, ss, bi))) return false;
    if (fabsf(fm(dx, cj, dy * sj)) > fm(bi, ss, fm(ai, cc, aj))) return false;
    if (fabsf(fm(dy, cj, -(dx * sj))) > fm(bi, cc, fm(ai, ss, bj))) return false;
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

// (field k of slot (e, n) = state[k * E * N + e * N + n]: a UNIFORM 64-bit field base plus ONE 32-bit lane offset -- the loads and stores
// then take the scalar-base form of the global instructions instead of a 64-bit vector address per field)
__device__ __forceinline__ void load_slot(const SimParams& p, int e, int n, Slot& s) {
    const size_t EN = (size_t)p.E * p.N;
    const unsigned int o = (unsigned int)e * (unsigned int)p.N + (unsigned int)n;      // E * N < 2^31 (copo_sim_create)
    const float* st = p.state;
    s.x = (st + 0 * EN)[o]; s.y = (st + 1 * EN)[o]; s.th = (st + 2 * EN)[o]; s.v = (st + 3 * EN)[o];
    s.steer = (st + 4 * EN)[o]; s.throttle = (st + 5 * EN)[o]; s.psteer = (st + 6 * EN)[o]; s.pthrottle = (st + 7 * EN)[o];
    s.yawrate = (st + 8 * EN)[o]; s.prog = (st + 9 * EN)[o]; s.lcf = (st + 10 * EN)[o]; s.eprew = (st + 11 * EN)[o];
    const int32_t* si = reinterpret_cast<const int32_t*>(st);
    s.route = (si + 12 * EN)[o]; s.status = (si + 13 * EN)[o]; s.aid = (si + 14 * EN)[o]; s.spawncnt = (si + 15 * EN)[o];
}

__device__ __forceinline__ void store_slot(const SimParams& p, int e, int n, const Slot& s) {
    const size_t EN = (size_t)p.E * p.N;
    const unsigned int o = (unsigned int)e * (unsigned int)p.N + (unsigned int)n;
    float* st = p.state;
    (st + 0 * EN)[o] = s.x; (st + 1 * EN)[o] = s.y; (st + 2 * EN)[o] = s.th; (st + 3 * EN)[o] = s.v;
    (st + 4 * EN)[o] = s.steer; (st + 5 * EN)[o] = s.throttle; (st + 6 * EN)[o] = s.psteer; (st + 7 * EN)[o] = s.pthrottle;
    (st + 8 * EN)[o] = s.yawrate; (st + 9 * EN)[o] = s.prog; (st + 10 * EN)[o] = s.lcf; (st + 11 * EN)[o] = s.eprew;
    int32_t* si = reinterpret_cast<int32_t*>(st);
    (si + 12 * EN)[o] = s.route; (si + 13 * EN)[o] = s.status; (si + 14 * EN)[o] = s.aid; (si + 15 * EN)[o] = s.spawncnt;
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
    float lif = floorf(fm(-lat, p.inv_w, 0.5f));
    lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
    const float left = 0.5f * w - lat;
    const float right = lanes * w - left;
    if (p.side_lasers == 0) {
        const float tw = (lanes + 1.0f) * w;
        row[0] = clipf(left / tw, 0.0f, 1.0f);
        row[1] = clipf(right / tw, 0.0f, 1.0f);
    }
    float* q = row + p.col_state;
    q[0] = clipf(fm(-0.5f, sinpsi, 0.5f), 0.0f, 1.0f);
    q[1] = clipf(fm(fabsf(s.v), 3.6f, 1.0f) * p.inv_vnorm, 0.0f, 1.0f);      // vehicle.speed is a magnitude
    q[2] = clipf(fm(s.steer, 1.0f / 120.0f, 0.5f), 0.0f, 1.0f);
    q[3] = clipf(fm(0.5f, s.psteer, 0.5f), 0.0f, 1.0f);
    q[4] = clipf(fm(0.5f, s.pthrottle, 0.5f), 0.0f, 1.0f);
    q[5] = clipf(fabsf(s.yawrate), 0.0f, 1.0f);
    if (p.lane_lasers == 0) {
        const float latr = -fm(lif, w, lat);
        row[p.col_lane] = clipf(fm(latr, 1.0f / 4.5f, 0.5f), 0.0f, 1.0f);
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
            const float nrm = sqrtf(fm(vx, vx, vy * vy));
            if (nrm > 50.0f) {
                const float sc = 50.0f / nrm;
                vx = vx * sc;
                vy = vy * sc;
            }
            const float fwd = fm(vx, cs, vy * sn), rhs = fm(vx, sn, -(vy * cs));
            float* n5 = row + p.col_navi + 5 * j;
            const float kap = gk[5];
            n5[0] = clipf(fm(fwd, 0.01f, 0.5f), 0.0f, 1.0f);
            n5[1] = clipf(fm(rhs, 0.01f, 0.5f), 0.0f, 1.0f);
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
// detector_ray for the arithmetic, which this repeats operation by operation.  The result of a beam is a pure MINIMUM over the
// primitives' hit distances, so the primitives may be visited in any order and by any lane.
// one line primitive against one beam: the hit distance (>= +0), or a negative value when the beam does not meet the primitive
__device__ __forceinline__ float detector_line_hit(const float* __restrict__ Ln, float x, float y, float dx, float dy) {
    if (Ln[6] == 0.0f) {
        const float rx = Ln[1] - x, ry = Ln[2] - y;
        const float den = dx * Ln[4] - dy * Ln[3];
        if (den == 0.0f) return -1.0f;
        const float sd = den > 0.0f ? 1.0f : -1.0f;
        const float ad = den * sd;
        const float tn = (rx * Ln[4] - ry * Ln[3]) * sd;
        const float un = (rx * dy - ry * dx) * sd;
        if (!(tn >= 0.0f && un >= 0.0f && un <= Ln[5] * ad)) return -1.0f;
        return tn / ad + 0.0f;
    }
    const float R = 1.0f / fabsf(Ln[6]);
    const float mx = x - Ln[7], my = y - Ln[8];
    const float b = mx * dx + my * dy;
    const float cq = mx * mx + my * my - R * R;
    const float disc = b * b - cq;
    if (!(disc >= 0.0f)) return -1.0f;
    const float sq = sqrtf(disc);
    for (int r = 0; r < 2; ++r) {
        const float tt = (r == 0 ? -b - sq : -b + sq) + 0.0f;
        if (!(tt >= 0.0f)) continue;
        const float hx = mx + tt * dx, hy = my + tt * dy;
        if (hx * Ln[9] + hy * Ln[10] >= R * Ln[11]) return tt;
    }
    return -1.0f;
}
__device__ __forceinline__ void detector_line(const float* __restrict__ Ln, float x, float y, float dx, float dy, float& best) {
    const float t = detector_line_hit(Ln, x, y, dx, dy);
    if (t >= 0.0f && t < best) best = t;
}
__device__ __forceinline__ float detector_ray(const SimParams& p, const float* __restrict__ lines, float x, float y, float dx,
                                              float dy, float range, float min_kind) {
    float best = range;
    for (int l = 0; l < p.n_lines; ++l) {
        const float* Ln = lines + (size_t)l * COPO_LINE_STRIDE;
        if (Ln[0] < min_kind) continue;
        detector_line(Ln, x, y, dx, dy, best);
    }
    return best;
}
// Can a beam of length `range` from (x, y) meet the primitive at all?  Conservative: distance to the segment / to the arc's full
// circle against the range + 0.1 % + 1 cm (a hit lies on the primitive at its hit distance from the origin; rounding is ~1e-4 m).
__device__ __forceinline__ bool detector_line_near(const float* __restrict__ Ln, float x, float y, float range) {
    const float lim = range * 1.001f + 0.01f;
    if (Ln[6] == 0.0f) {
        const float rx = x - Ln[1], ry = y - Ln[2];
        float t = rx * Ln[3] + ry * Ln[4];
        t = t < 0.0f ? 0.0f : (t > Ln[5] ? Ln[5] : t);
        const float ex = rx - t * Ln[3], ey = ry - t * Ln[4];
        return ex * ex + ey * ey <= lim * lim;
    }
    const float R = 1.0f / fabsf(Ln[6]);
    const float mx = x - Ln[7], my = y - Ln[8];
    return fabsf(sqrtf(mx * mx + my * my) - R) <= lim;
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
// ray_box with that division, as one predicate: `hit` = the ray meets the box, the return value its entering distance (0 from
// inside).  Same comparisons on the same values as ray_box, evaluated without the nested early exits.
__device__ __forceinline__ float ray_box_nr(float ox, float oy, float ddx, float ddy, float hl, float hw, bool& hit) {
    const float ax = fabsf(ddx), ay = fabsf(ddy);
    const float oxs = ddx < 0.0f ? -ox : ox, oys = ddy < 0.0f ? -oy : oy;
    const float nxe = -(hl + oxs), nxx = hl - oxs, nye = -(hw + oys), nyx = hw - oys;
    const float exy = nxe * ay, eyx = nye * ax;
    hit = (nxx >= 0.0f) & (nyx >= 0.0f) & (exy <= nyx * ax) & (eyx <= nxx * ay);
    const bool usex = exy >= eyx;
    const float n = usex ? nxe : nye, a = usex ? ax : ay;
    return n > 0.0f ? div_nr(n, a) : 0.0f;
}

// Wave64 inclusive scans on the DPP network (row shifts 1/2/4/8, then row_bcast:15 / :31 -- the gfx9 sequence):
// six VALU operations, no LDS traffic (a __shfl_up ladder is six ds_bpermute round trips).
template <bool MAX>
__device__ __forceinline__ int wave_scan_incl(int v) {
#define COPO_SCAN_STEP(ctrl, rmask)                                                       \
    {                                                                                     \
        const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, (rmask) == 0xf);  \
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

// The add scan with the DPP control on the add itself (six VALU instructions; the compiler's own lowering of update_dpp + add is a
// v_mov_b32_dpp and a v_add per step).  `s_nop 1`: a DPP read needs two wait states after the VALU write of its source.
__device__ __forceinline__ int wave_scan_add(int v) {
    asm volatile(
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
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
            v = fm(a, h, v);
            if (v < 0.0f && !(p.reverse_acc > 0.0f)) v = 0.0f;
            const float dxh = fm(cs, cb, -(sn * sb)), dyh = fm(sn, cb, cs * sb);
            x = fm(v * dxh, h, x);
            y = fm(v * dyh, h, y);
            // turn the heading vector by the sub-step's small angle: 3-term sine / cosine, no range reduction
            float dth = v * yawk * h;
            if (p.lat_acc_max > 0.0f && v * fabsf(dth) > p.lat_acc_max * h) {      // tyres slide: v x yaw rate is friction-limited
                const float lim = (p.lat_acc_max * h) / v;
                dth = dth < 0.0f ? -lim : lim;
            }
            const float q = dth * dth;
            const float sd2 = fm(-(dth * q), fm(-q, 0.00833333333f, 0.166666667f), dth);
            const float cd2 = fm(-q, fm(-q, 0.0416666667f, 0.5f), 1.0f);
            const float cn = fm(cs, cd2, -(sn * sd2)), sm = fm(sn, cd2, cs * sd2);
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
    bool too_fast = false, in_toll = false;
    if (p.toll_dim) {
        const int toll_seg = (int)meta[2];
        in_toll = seg == toll_seg;
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
    float lif = floorf(fm(-lat, p.inv_w, 0.5f));
    lif = lif < 0.0f ? 0.0f : (lif > lanes - 1.0f ? lanes - 1.0f : lif);
    const float left = 0.5f * w - lat, right = (lanes * w + funnel_extra(g, sl, w)) - left;
    const float cos2 = fm(-sinpsi, sinpsi, 1.0f);
    const float edge = p.body_margin * fm(hl, fabsf(sinpsi), hw * sqrtf(cos2 > 0.0f ? cos2 : 0.0f));      // body extent across the road
    const bool on_road = (left >= (left_solid ? edge : (left_open ? -w : 0.0f))) && (right >= (right_solid ? edge : 0.0f));
    const bool arrive = (seg == nseg - 1) && (sl > g[4] - p.arrive_margin) && (sl < g[4] + p.arrive_margin) && on_road;
    const bool oor = !on_road;
    // MultiAgentTollgateEnv (copo_sim_cfg, ABI 8): leaving the booth early is a crash (0) or ends the agent with the out-of-road flag
    // and the step's ordinary reward (1); on the booth road the reward is the driving reward alone up to the speed limit and
    // -overspeed_penalty * speed / max_speed above it
    const bool early = too_fast && p.toll_early_exit != 0;
    bool bldg = false;                       // the vehicle's box against the map's static boxes (buildings): the vehicles' separating-axis test
    for (int b = 0; b < p.n_boxes; ++b) {
        const float* B = p.boxes + b * COPO_BOX_STRIDE;
        const float bdx = B[0] - s.x, bdy = B[1] - s.y, rr = (B[4] + B[5]) + (hl + hw);      // (farther than the half extents' sums: no overlap)
        if (fm(bdx, bdx, bdy * bdy) <= rr * rr && obb_overlap2(s.x, s.y, ch, sh, hl, hw, B[0], B[1], B[2], B[3], B[4], B[5])) bldg = true;
    }
    const bool crash = crash_in || bldg || (too_fast && p.toll_early_exit == 0);
    const float drive = (prog - prev) * fm(g[5], lif * w, 1.0f), spd = fabsf(s.v) / p.max_speed;
    float r;
    if (p.toll_speed_limit > 0.0f && in_toll) r = fabsf(s.v) > p.toll_speed_limit ? -p.overspeed_penalty * spd : p.driving_reward * drive;
    else r = fm(p.driving_reward, drive, p.speed_reward * spd);
    fl = COPO_F_ACTED;
    if (arrive) { r = p.success_reward; fl |= COPO_F_ARRIVE; }
    else if (oor) { r = -p.out_penalty; }
    else if (crash) { r = -p.crash_penalty; }
    if (oor || early) fl |= COPO_F_OUT;
    if (crash) fl |= COPO_F_CRASH;
    bool done = arrive || oor || crash || early;
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

// ---- helpers of the one-wave-per-scene phases -----------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pk_u64(const unsigned int* w) { return *reinterpret_cast<const unsigned long long*>(w); }
__device__ __forceinline__ void pk_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int pk_mbcnt(unsigned long long m) {      // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// wave sum of doubles on the DPP network (row shifts, then row broadcasts); the total arrives in lane 63.  Lanes without a source
// add +0.0.  Only used where every partial sum is exact, so the order of the tree does not matter.
__device__ __forceinline__ double pk_wave_sum_f64(double v) {
#define COPO_DSUM_STEP(ctrl, rmask)                                                                             \
    {                                                                                                           \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false);              \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false);              \
        v = v + __hiloint2double(hi, lo);                                                                       \
    }
    COPO_DSUM_STEP(0x111, 0xf)
    COPO_DSUM_STEP(0x112, 0xf)
    COPO_DSUM_STEP(0x114, 0xf)
    COPO_DSUM_STEP(0x118, 0xf)
    COPO_DSUM_STEP(0x142, 0xa)
    COPO_DSUM_STEP(0x143, 0xc)
#undef COPO_DSUM_STEP
    return v;
}

__device__ __forceinline__ unsigned long long pk_uni64(unsigned long long v) {      // a wave-uniform value, in scalar registers
    return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}

// Fallback of detector_beams (below) for configurations whose LDS scratch cannot hold a row of beam minima: side / lane-line detector
// beams of `np` present agents by `nth` threads (WG: the threads of a workgroup, else one wave).  A beam only meets the primitives within its range of the vehicle: one pass marks, per agent, the
// primitives near enough for each of the two detectors (bit masks in `scratch`, 8 words per agent, `cap` words in all); the beams then
// walk the marked primitives in table order -- the order in which the exhaustive loop (detector_ray, the oracle's) lowers `best`, so
// the result is that loop's, bit for bit: a primitive left out cannot lower it.
template <bool WG, class PoseF>
__device__ __forceinline__ void detector_beams_walk(const SimParams& p, PoseF pose, const uint8_t* plist, int np, unsigned int* scratch, int cap,
                                               float* __restrict__ eobs, int tid, int nth) {
    const int nb = p.side_lasers + p.lane_lasers, NLn = p.n_lines, O = p.O;
    if (nb <= 0) return;
    auto sync = [&]() {
        if (WG) __syncthreads();
        else pk_wave_sync();
    };
    const int G = cap / 8 > 0 ? cap / 8 : 1;      // agents per pass
    const float inv_nb = 1.0f / (float)nb, inv_nl = 1.0f / (float)(NLn > 0 ? NLn : 1);
    for (int ip0 = 0; ip0 < np; ip0 += G) {
        const int na = np - ip0 < G ? np - ip0 : G;
        sync();
        for (int q = tid; q < na * 8; q += nth) scratch[q] = 0u;
        sync();
        for (int q = tid; q < na * NLn; q += nth) {
            const int ia = (int)(((float)q + 0.5f) * inv_nl), l = q - ia * NLn;
            const float4 pi = pose((int)plist[ip0 + ia]);
            const float* Ln = p.lines + (size_t)l * COPO_LINE_STRIDE;
            const float kind = Ln[0];
            if (p.side_lasers > 0 && kind >= 2.0f && detector_line_near(Ln, pi.x, pi.y, p.side_range)) atomicOr(&scratch[ia * 8 + (l >> 5)], 1u << (l & 31));
            if (p.lane_lasers > 0 && kind >= 1.0f && detector_line_near(Ln, pi.x, pi.y, p.lane_range)) atomicOr(&scratch[ia * 8 + 4 + (l >> 5)], 1u << (l & 31));
        }
        sync();
        for (int q = tid; q < na * nb; q += nth) {
            const int ia = (int)(((float)q + 0.5f) * inv_nb), b = q - ia * nb;
            const int i = plist[ip0 + ia];
            const bool side = b < p.side_lasers;
            const int k = side ? b : b - p.side_lasers;
            const float* tab = side ? p.side_cs : p.lane_cs;
            const float a0 = tab[2 * k], b0 = tab[2 * k + 1];
            const float4 pi = pose(i);
            const float dx = pi.z * a0 - pi.w * b0, dy = pi.w * a0 + pi.z * b0;
            float best = side ? p.side_range : p.lane_range;
            for (int w = 0; w < 4; ++w) {
                unsigned int m = scratch[ia * 8 + (side ? 0 : 4) + w];
                while (m) {
                    const int l = 32 * w + __ffs((int)m) - 1;
                    m &= m - 1u;
                    detector_line(p.lines + (size_t)l * COPO_LINE_STRIDE, pi.x, pi.y, dx, dy, best);
                }
            }
            eobs[i * O + (side ? k : p.col_lane + k)] = best * (side ? p.inv_side_range : p.inv_lane_range);
        }
    }
}


// Conservative window of detector beams that can meet line primitive Ln from (x, y) with heading (c, s): first beam `klo` in
// [0, nbeam) and the number of beams `cnt` (0 .. nbeam, the window wraps round).  The beams are evenly spaced (`rpr` beams per
// radian, signed; beam 0 at `theta0` in the vehicle frame -- checked on the host, else rpr = 0: every beam).  A straight piece
// subtends the interval between the bearings of its end points (the short way round: less than pi unless the origin lies on it); an
// arc is covered by a disc -- round the chord's mid point for a half angle up to pi / 2, else the arc's own circle.  NOT part of the
// deterministic spec: a beam outside the window cannot meet the primitive, every hit / miss decision stays with detector_line_hit.
__device__ __forceinline__ void detector_window(const float* __restrict__ Ln, float x, float y, float c, float s, float theta0, float rpr,
                                                int nbeam, int& klo, int& cnt) {
    klo = 0; cnt = nbeam;
    if (rpr == 0.0f) return;
    float ua, ub;                                  // beam coordinates of the interval's ends
    const float marg = 0.004f * fabsf(rpr) + 0.02f;      // atan2_window < 1e-5 rad, the asin bound, rounding
    if (Ln[6] == 0.0f) {
        const float ax = Ln[1] - x, ay = Ln[2] - y;
        const float bx = ax + Ln[5] * Ln[3], by = ay + Ln[5] * Ln[4];
        if (!(ax * ax + ay * ay > 1e-4f) || !(bx * bx + by * by > 1e-4f)) return;      // at an end point: every beam meets it at distance 0
        const float fa = atan2_window(c * ay - s * ax, c * ax + s * ay), fb = atan2_window(c * by - s * bx, c * bx + s * by);
        float d = fb - fa;
        d = d > 3.14159265f ? d - 6.28318531f : (d < -3.14159265f ? d + 6.28318531f : d);
        if (!(fabsf(d) < 3.0f)) return;            // the origin (almost) on the piece: which way round is not decided here
        ua = (fa - theta0) * rpr;
        ub = ua + d * rpr;
    } else {
        const float R = 1.0f / fabsf(Ln[6]);
        const float ch = Ln[11] > 0.0f ? Ln[11] : 0.0f;               // cos of the half angle, 0 beyond a quarter turn: the whole circle's disc
        const float mx = Ln[7] + R * ch * Ln[9] - x, my = Ln[8] + R * ch * Ln[10] - y;
        const float rho = R * sqrtf(fmaxf(1.0f - ch * ch, 0.0f)) * 1.001f + 0.01f;
        const float d2 = mx * mx + my * my;
        if (!(d2 > rho * rho * 1.01f)) return;     // inside (or next to) the disc
        const float q = rho * __builtin_amdgcn_rsqf(d2);
        const float w = (q + 0.5708f * q * q * q) * 1.001f;           // >= asin(rho / distance)
        const float fm = atan2_window(c * my - s * mx, c * mx + s * my);
        ua = (fm - w - theta0) * rpr;
        ub = (fm + w - theta0) * rpr;
    }
    const float lo = fminf(ua, ub) - marg, hi = fmaxf(ua, ub) + marg;
    const int ilo = (int)ceilf(lo), ihi = (int)floorf(hi);
    const int n = ihi - ilo + 1;
    if (n >= nbeam) return;
    cnt = n < 0 ? 0 : n;
    int k = ilo + 2 * nbeam;                       // |ua|, |ub| < (2 pi + pi) |rpr| + margin < 1.6 nbeam: k in (0, 4 nbeam)
    k = k >= 2 * nbeam ? k - 2 * nbeam : k;
    k = k >= nbeam ? k - nbeam : k;
    klo = k < 0 ? 0 : k;
}

// Side / lane-line detector beams (Bottleneck, Tollgate, generated roads) of `np` present agents by `nth` threads (WG: the threads of
// a workgroup -- whole waves, `tid` beyond them for waves that only meet the barriers -- else one wave).  The result of a beam is a
// pure minimum over the primitives (detector_line_hit), so the work is laid out by (agent, primitive, detector) PAIRS as in the LiDAR:
//   1. candidates (agent, primitive) in batches of 64: is the primitive within the detector's range of the agent (detector_line_near)?
//      The pairs that are, are pushed together ACROSS batches (a pending batch in the lanes' registers, ds_permute), so that
//   2. full batches of 64 pairs get their window of beams (detector_window), the windows are numbered through by a scan and every lane
//      takes one (beam, primitive) test (owner = ballot + mbcnt of the batch's head flags, as in lidar_by_wave), folding its hit into
//      the agent's row of beam minima in `scratch` (`cap` words: cap / beams agents per pass) with an LDS atomicMin on the float bits.
// Tollgate: ~220 tests per agent instead of 76 beams x ~15 marked primitives.  `wtag`: 64 words of the calling wave (head flags).
template <bool WG, class PoseF>
__device__ __forceinline__ void detector_beams(const SimParams& p, PoseF pose, const uint8_t* plist, int np, unsigned int* scratch, int cap,
                                               float* __restrict__ eobs, int tid, int nth, int* wtag, int ecount = -1) {
    const int ns = p.side_lasers, nl = p.lane_lasers, nb = ns + nl, NLn = p.n_lines, O = p.O;
    if (nb <= 0) return;
    if (cap < nb || NLn > 255) {                   // (no room for one row of minima: the walk over marked primitives)
        detector_beams_walk<WG>(p, pose, plist, np, scratch, cap, eobs, tid, nth);
        return;
    }
    auto sync = [&]() {
        if (WG) __syncthreads();
        else pk_wave_sync();
    };
    const bool working = tid < nth;                // (WG: waves that only meet the barriers carry a huge tid)
    const int lane = threadIdx.x & 63;
    const int wv = working ? (tid >> 6) : 0, nwv = nth >> 6;
    const int G = cap / nb < 64 ? cap / nb : 64;  // agents per pass
    const float inv_nb = 1.0f / (float)nb;
    const unsigned int side_bits = __float_as_uint(p.side_range), lane_bits = __float_as_uint(p.lane_range);
    int seq = 0;
    if (working) wtag[lane] = 0;
    // the beam tables in the lanes' registers (beam b of [side | lane-line] in lane b & 63, two registers deep): a test fetches its
    // beam's direction with ds_bpermute instead of a global load at the end of its dependent chain
    const bool tab_regs = nb <= 128;
    float2 tb0 = make_float2(0.f, 0.f), tb1 = make_float2(0.f, 0.f);
    if (tab_regs) {
        auto entry = [&](int b) { return b < ns ? reinterpret_cast<const float2*>(p.side_cs)[b] : reinterpret_cast<const float2*>(p.lane_cs)[b - ns]; };
        if (lane < nb) tb0 = entry(lane);
        if (64 + lane < nb) tb1 = entry(64 + lane);
    }
    for (int ip0 = 0; ip0 < np; ip0 += G) {
        const int na = np - ip0 < G ? np - ip0 : G;
        sync();
        for (int q = tid; q < na * nb; q += nth) {
            const int ia = (int)(((float)q + 0.5f) * inv_nb), b = q - ia * nb;
            scratch[q] = b < ns ? side_bits : lane_bits;
        }
        sync();
        // a batch of `n` pairs, record ia | l << 8 | detector << 16 in lanes 0 .. n - 1: windows, scan, tests.  The lane of a pair (its
        // owner) derives everything a beam test reads from the pair ONCE -- rx, ry / mx, my, cq, R cos(half angle): the same operations
        // detector_line_hit performs per beam, so the same bits -- and the test lanes fetch it from the owner with ds_bpermute: no global
        // load between a test's number and its atomicMin but the beam table's
        auto tests = [&](int rec, int n) {
            {   // the arcs behind the straight pieces: a batch of 64 tests then takes one of the two branches of the hit test, not both
                const bool lv = lane < n, arc = lv && ((rec >> 17) & 1);
                const unsigned long long ma = __ballot(arc), ms = __ballot(lv && !arc);
                if (ma != 0ull && ms != 0ull) {
                    const int to = arc ? __popcll(ms) + pk_mbcnt(ma) : (lv ? pk_mbcnt(ms) : lane);
                    rec = __builtin_amdgcn_ds_permute(to << 2, rec);
                }
            }
            const bool live = lane < n;
            int klo = 0, cnt = 0;
            float fa = 0.0f, fb = 0.0f, fc = 0.0f, fd = 0.0f, fe = 0.0f, ff = 0.0f, hc = 1.0f, hs = 0.0f;
            if (live) {
                const int ia = rec & 63, l = (rec >> 8) & 255;
                const bool side = ((rec >> 16) & 1) == 0;
                const float4 pi = pose((int)plist[ip0 + ia]);
                const float* Ln = p.lines + (size_t)l * COPO_LINE_STRIDE;
                detector_window(Ln, pi.x, pi.y, pi.z, pi.w, side ? p.side_theta0 : p.lane_theta0,
                                side ? p.side_rpr : p.lane_rpr, side ? ns : nl, klo, cnt);
                cnt = cnt < 1 ? 1 : cnt;           // (an empty window still takes one test: every live lane heads a run of tests)
                hc = pi.z; hs = pi.w;
                if (Ln[6] == 0.0f) {
                    fa = Ln[1] - pi.x; fb = Ln[2] - pi.y; fc = Ln[3]; fd = Ln[4]; fe = Ln[5];
                } else {
                    const float R = 1.0f / fabsf(Ln[6]);
                    fa = pi.x - Ln[7]; fb = pi.y - Ln[8]; fc = Ln[9]; fd = Ln[10]; fe = R * Ln[11];
                    ff = fa * fa + fb * fb - R * R;        // (the record says "arc": bit 17)
                }
            }
            const int incl = wave_scan_add(cnt);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int excl = incl - cnt;
            const int rec_k = klo - excl;          // first beam - first test
            COPO_DCOUNT(10, 1); COPO_DCOUNT(11, total); COPO_DCOUNT(12, (total + 63) >> 6);
            int hb = -1;
            for (int t0 = 0; t0 < total; t0 += 64) {
                seq += 1;
                if (cnt > 0 && excl >= t0 && excl < t0 + 64) wtag[excl - t0] = seq;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool hd = wtag[lane] == seq;
                const unsigned long long H = __ballot(hd);
                const int own = hb + pk_mbcnt(H) + (hd ? 1 : 0);
                hb += __popcll(H);
                __builtin_amdgcn_wave_barrier();
                const int sl = own << 2;
                auto from = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(v))); };
                const int ra = __builtin_amdgcn_ds_bpermute(sl, rec);
                const int rk = __builtin_amdgcn_ds_bpermute(sl, rec_k);
                const float ga = from(fa), gb = from(fb), gc = from(fc), gd = from(fd), ge = from(fe), gf = from(ff), c = from(hc), sn = from(hs);
                const int t = t0 + lane;
                // (every lane takes part in the fetches: ds_bpermute returns 0 from a lane that is switched off)
                const bool side = ((ra >> 16) & 1) == 0;
                const int nbeam = side ? ns : nl;
                int k = rk + t;                         // beam of this test: in [klo, klo + cnt), the window wraps
                k = k >= nbeam ? k - nbeam : k;
                k = t < total ? k : 0;
                float2 ab;
                if (tab_regs) {
                    const int kk = (side ? 0 : ns) + k, ks = (kk & 63) << 2;
                    const float x0 = __int_as_float(__builtin_amdgcn_ds_bpermute(ks, __float_as_int(tb0.x))), y0 = __int_as_float(__builtin_amdgcn_ds_bpermute(ks, __float_as_int(tb0.y)));
                    const float x1 = __int_as_float(__builtin_amdgcn_ds_bpermute(ks, __float_as_int(tb1.x))), y1 = __int_as_float(__builtin_amdgcn_ds_bpermute(ks, __float_as_int(tb1.y)));
                    ab = kk < 64 ? make_float2(x0, y0) : make_float2(x1, y1);
                } else {
                    ab = reinterpret_cast<const float2*>(side ? p.side_cs : p.lane_cs)[k];
                }
                if (t < total) {
                    const float dx = c * ab.x - sn * ab.y, dy = sn * ab.x + c * ab.y;
                    float th = -1.0f;                   // detector_line_hit on the owner's quantities, operation by operation
                    if (!((ra >> 17) & 1)) {
                        const float den = dx * gd - dy * gc;
                        if (den != 0.0f) {
                            const float sd = den > 0.0f ? 1.0f : -1.0f;
                            const float ad = den * sd;
                            const float tn = (ga * gd - gb * gc) * sd;
                            const float un = (ga * dy - gb * dx) * sd;
                            if (tn >= 0.0f && un >= 0.0f && un <= ge * ad) th = tn / ad + 0.0f;
                        }
                    } else {
                        const float b = ga * dx + gb * dy;
                        const float disc = b * b - gf;
                        if (disc >= 0.0f) {
                            const float sq = sqrtf(disc);
                            for (int r = 0; r < 2; ++r) {
                                const float tt = (r == 0 ? -b - sq : -b + sq) + 0.0f;
                                if (!(tt >= 0.0f)) continue;
                                const float hx = ga + tt * dx, hy = gb + tt * dy;
                                if (hx * gc + hy * gd >= ge) { th = tt; break; }
                            }
                        }
                    }
                    if (th >= 0.0f) atomicMin(&scratch[(ra & 63) * nb + (side ? 0 : ns) + k], __float_as_uint(th));
                    if (COPO_PROFILE_SKIP & 16384) { const int nh = __popcll(__ballot(th >= 0.0f)); COPO_DCOUNT(13, nh); }
                }
            }
        };
        // candidates: lane = line primitive (its record in the lane's registers, read once per block of 64 lines), one agent per turn --
        // no memory access in the turn but the agent's pose (LDS, one address for the wave)
        int pend = 0, npend = 0;                   // the pending batch: records in lanes 0 .. npend - 1 (npend < 64, uniform)
        for (int l0 = 0; l0 < (working ? NLn : 0); l0 += 64) {
            const int l = l0 + lane;
            const bool has = l < NLn;
            const float* Ln = p.lines + (size_t)(has ? l : 0) * COPO_LINE_STRIDE;
            const float kind = has ? Ln[0] : 0.0f, kap = Ln[6];
            const float px = kap == 0.0f ? Ln[1] : Ln[7], py = kap == 0.0f ? Ln[2] : Ln[8];       // start point / arc centre
            const float ux = Ln[3], uy = Ln[4], len = Ln[5];
            const float Rr = kap == 0.0f ? 0.0f : 1.0f / fabsf(kap);
            const int arc_bit = (has && kap != 0.0f) ? (1 << 17) : 0;
            for (int det = 0; det < 2; ++det) {    // 0: side detector (continuous lines), 1: lane-line detector (every line)
                if ((det == 0 ? ns : nl) <= 0) continue;
                const float range = det == 0 ? p.side_range : p.lane_range, min_kind = det == 0 ? 2.0f : 1.0f;
                const float lim = range * 1.001f + 0.01f;          // (detector_line_near, on the lane's registers)
                // an arc is near when the distance to its centre is within lim of its radius: squared bounds, no square root in the turn
                const float rlo = Rr - lim > 0.0f ? (Rr - lim) * 0.9999f : 0.0f, rhi = (Rr + lim) * 1.0001f;
                const float rlo2 = rlo * rlo, rhi2 = rhi * rhi;
                for (int ia = wv; ia < na; ia += nwv) {
                    const float4 pi = pose((int)plist[ip0 + ia]);
                    bool near = false;
                    if (kind >= min_kind) {
                        const float rx = pi.x - px, ry = pi.y - py;
                        if (kap == 0.0f) {
                            float t = rx * ux + ry * uy;
                            t = t < 0.0f ? 0.0f : (t > len ? len : t);
                            const float ex = rx - t * ux, ey = ry - t * uy;
                            near = ex * ex + ey * ey <= lim * lim;
                        } else {
                            const float d2 = rx * rx + ry * ry;
                            near = d2 >= rlo2 && d2 <= rhi2;
                        }
                    }
                    const unsigned long long mn = __ballot(near);
                    const int nn = __popcll(mn);
                    COPO_DCOUNT(8, 1); COPO_DCOUNT(9, nn);
                    if (nn == 0) continue;
                    const int rec = ia | (l << 8) | (det << 16) | arc_bit;
                    const int pos = npend + pk_mbcnt(mn);
                    // the near pairs join the pending batch at lanes npend ..; the other lanes push to a lane whose value is not taken
                    // (lane 0 holds a pending record when npend > 0; lane 63 is only a target when all 64 lanes push for real)
                    const int got = __builtin_amdgcn_ds_permute(((near && pos < 64) ? pos : (npend > 0 ? 0 : 63)) << 2, rec);
                    pend = lane < npend ? pend : got;
                    const int tot = npend + nn;
                    if (tot >= 64) {
                        tests(pend, 64);
                        pend = __builtin_amdgcn_ds_permute(((near && pos >= 64) ? pos - 64 : 63) << 2, rec);      // the batch's overflow: < 64 pairs
                        npend = tot - 64;
                    } else {
                        npend = tot;
                    }
                }
            }
        }
        if (npend > 0) tests(pend, npend);
        sync();
        for (int q = tid; q < na * nb; q += nth) {
            const int ia = (int)(((float)q + 0.5f) * inv_nb), b = q - ia * nb;
            const int i = plist[ip0 + ia];
            const bool side = b < ns;
            eobs[i * O + (side ? b : p.col_lane + b - ns)] = __uint_as_float(scratch[q]) * (side ? p.inv_side_range : p.inv_lane_range);
        }
    }
}

// LiDAR of one scene by its wave: the pair-driven formulation of obs_phase (sim_kernels.hip, the shapes with several waves per scene) -- conservative ray window per (fan,
// vehicle) pair, box tests numbered by a DPP scan, hits folded with LDS atomicMin -- with
//   * the pair queue filled from the reach masks of the neighbour walk (no all-pairs reach pass);
//   * the pairs with a non-empty window pushed together (ds_permute) before their box tests are numbered: the owner of box test t
//     is then simply pair number (heads before this batch) + (heads at or before t in it) - 1, a ballot of the head flags and a
//     v_mbcnt instead of a DPP max-scan per batch of tests; the head flags carry the batch's sequence number, so the strip is
//     never cleared;
//   * the pair record holding the ray-minima row offset (lp * NL) next to the first ray, so a box test multiplies nothing;
//   * the hit distance through div_nr (sim_device.h).
// `present` / `solid`: the scene after the step (after a reset, if it reset).
// `pose(slot)` -> float4 {x, y, cos, sin} of a vehicle; `reach(lane)` -> this LANE's slot's reach mask (two 32-bit halves in the lane's
// registers, or read from LDS by the caller); `plist`: u8 [64] for the present slots (filled here); `best` / `cq`: the ray minima [chunk][NL]
// and the pair queue u16 [chunk * N]; `wtag`: 64 words of this wave.
template <class PoseF>
__device__ __forceinline__ void lidar_by_wave(const SimParams& p, PoseF pose, uint8_t* plist, unsigned int reach_lo, unsigned int reach_hi,
                                              unsigned int* best, uint16_t* cq, const float* __restrict__ rays, int* wtag,
                                              int e, int lane, unsigned long long present_in, unsigned long long solid_in, bool all_reach_in,
                                              float* __restrict__ obs) {
    const unsigned long long present = pk_uni64(present_in), solid = pk_uni64(solid_in);
    const bool all_reach = __builtin_amdgcn_readfirstlane(all_reach_in ? 1 : 0) != 0;
    const int N = p.N, O = p.O, NL = p.num_lasers;
    const float hl = p.hl, hw = p.hw;
    const float circ = sqrtf(hl * hl + hw * hw);
    const float range = p.lidar_range;
    const float lim = range + circ;
    const int np = __popcll(present);
    const unsigned int range_bits = __float_as_uint(range);
    float* eobs = obs + (size_t)e * N * O;
    const int CH = p.chunk > 0 ? p.chunk : N;
    const float rays_per_rad = (float)NL * 0.159154943f;
    const float inv_nl = 1.0f / (float)NL;
    const float inv_range = p.inv_range;
    const int col_lidar = p.col_lidar;
    const int head = (4 - (col_lidar & 3)) & 3;
    const int nvec = (NL - head) >> 2;
    const bool vec_out = ((O & 3) == 0) && nvec > 0 && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
    const float inv_nvec = 1.0f / (float)(nvec > 0 ? nvec : 1), inv_nsc = 1.0f / (float)(NL - 4 * nvec > 0 ? NL - 4 * nvec : 1);
    const int sh = (NL & 3) == 0 ? ((4 - head) & 3) : 0;      // the shift of the minima rows (LIDAR_MIN_PAD)
    unsigned int* bs = best + sh;
    if (__builtin_amdgcn_inverse_ballot_w64(present)) plist[pk_mbcnt(present)] = (uint8_t)lane;
    wtag[lane] = 0;
    int seq = 0;                      // sequence number of the box-test batches of this wave (head flags)
    pk_wave_sync();
    for (int ip0 = 0; ip0 < np; ip0 += CH) {
        const int cha = np - ip0 < CH ? np - ip0 : CH;
        {      // ray minima of the chunk := range, 16 bytes per lane and turn (the storage is 16-byte aligned; a tail of < 4 words by the last lanes)
            const int nwords = cha * NL + sh;         // (the shifted rows end inside the pad)
            const int nw4 = nwords >> 2;
            const uint4 r4 = make_uint4(range_bits, range_bits, range_bits, range_bits);
            for (int q = lane; q < nw4; q += 64) reinterpret_cast<uint4*>(best)[q] = r4;
            if (lane < (nwords & 3)) best[4 * nw4 + lane] = range_bits;
        }
        // pair queue of this chunk of fans, from their reach masks
        int nq = 0;
        {
            // (the mask of fan lp lives in the registers of lane plist[ip0 + lp], the fan's own slot)
            const int islot = lane < cha ? (int)plist[ip0 + lane] : 0;
            for (int lp = 0; lp < ((COPO_PROFILE_SKIP & 2048) ? 0 : cha); ++lp) {
                const int i = __builtin_amdgcn_readlane(islot, lp);
                const unsigned long long m = all_reach ? solid
                                                       : (((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)reach_hi, i) << 32) |
                                                          (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)reach_lo, i)) & solid;
                if (__builtin_amdgcn_inverse_ballot_w64(m)) cq[nq + pk_mbcnt(m)] = (uint16_t)((lp << 8) | lane);
                nq += __popcll(m);
            }
        }
        pk_wave_sync();
        COPO_COUNT(8, nq);
        for (int q0 = 0; q0 < ((COPO_PROFILE_SKIP & 2) ? 0 : nq); q0 += 64) {
            COPO_COUNT(9, 1);
            const bool live = q0 + lane < nq;
            const int ent = live ? (int)cq[q0 + lane] : 0;
            const int lp = ent >> 8, j = ent & 255;
            const int i = plist[ip0 + lp];
            const float4 pi = pose(i), pj = pose(j);
            const float ci = pi.z, si = pi.w, cj = pj.z, sj = pj.w;
            const float dx = pj.x - pi.x, dy = pj.y - pi.y;
            const float d2 = fm(dx, dx, dy * dy);
            int klo = 0, cnt = 0;
            if (live && j != i && !(d2 > lim * lim)) {
                if (d2 <= circ * circ * 1.002f) {
                    cnt = NL;                         // origin inside the circumcircle: any ray may hit
                } else {
                    const float phi = p.ray_sign * atan2_window(ci * dy - si * dx, ci * dx + si * dy);   // in beam-index direction
                    const float rd = __builtin_amdgcn_rsqf(d2);
                    const float x = circ * rd;
                    float w = x + 0.5708f * x * x * x;                    // >= asin(circumradius / distance)
                    const float ux = dx * rd, uy = dy * rd;
                    const float ca = fabsf(cj * ux + sj * uy), sa = fabsf(cj * uy - sj * ux);
                    const float h_perp = hl * sa + hw * ca, along = d2 * rd - (hl * ca + hw * sa);
                    if (along > 0.5f) w = fminf(w, h_perp * __builtin_amdgcn_rcpf(along) * 1.0001f);
                    w += 0.004f;                                          // margin over the approximations above (< 1e-4 rad)
                    const int lo = (int)ceilf((phi - w) * rays_per_rad), hi = (int)floorf((phi + w) * rays_per_rad);
                    cnt = hi - lo + 1;
                    cnt = cnt < 0 ? 0 : (cnt > NL ? NL : cnt);
                    klo = lo < 0 ? lo + NL : lo;
                }
            }
            // the pairs with a window, pushed together: lane r takes the r-th of them (the others push to lane 63, which only
            // holds a pair when all 64 have a window)
            const unsigned long long mw = __ballot(cnt > 0);
            const int nw = __popcll(mw);
            COPO_COUNT(13, nw);
            if (nw == 0) continue;
            const int dst = (cnt > 0 ? pk_mbcnt(mw) : 63) << 2;
            const float rec_ox = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(-fm(dx, cj, dy * sj))));
            const float rec_oy = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(-fm(dy, cj, -(dx * sj)))));
            const float rec_cr = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(fm(ci, cj, si * sj))));
            const float rec_sr = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(fm(ci, sj, -(si * cj)))));
            const int ck = __builtin_amdgcn_ds_permute(dst, cnt | (klo << 12) | (lp << 24));      // cnt <= 256 < 2^12, klo < 2^12, lp < 64
            const int cnt_c = lane < nw ? (ck & 0xfff) : 0;
            const int incl = wave_scan_add(cnt_c);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int excl = incl - cnt_c;
            // record word of the box tests: first ray - first test (16 bits, signed) | row offset of the fan's ray minima (lp * NL)
            const int rec_ix = ((((ck >> 12) & 0xfff) - excl) & 0xffff) | ((int)__umul24((unsigned int)(ck >> 24) & 63u, (unsigned int)NL & 0x1ffu) << 16);
            int hb = -1;                              // (heads before this batch of tests) - 1
            COPO_COUNT(10, total);
            COPO_COUNT(11, (total + 63) >> 6);
            for (int t0 = 0; t0 < total; t0 += 64) {
                seq += 1;
                if (cnt_c > 0 && excl >= t0 && excl < t0 + 64) wtag[excl - t0] = seq;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool hd = wtag[lane] == seq;
                const unsigned long long H = __ballot(hd);
                const int own = hb + pk_mbcnt(H) + (hd ? 1 : 0);
                hb += __popcll(H);
                __builtin_amdgcn_wave_barrier();
                const int sl = own << 2;
                const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(rec_ox)));
                const float oy = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(rec_oy)));
                const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(rec_cr)));
                const float sr = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(rec_sr)));
                const int pw = __builtin_amdgcn_ds_bpermute(sl, rec_ix);
                const int t = t0 + lane;
                if (t < total) {
                    const unsigned int k0 = (unsigned int)(((pw << 16) >> 16) + t);
                    const unsigned int k = k0 < k0 - (unsigned int)NL ? k0 : k0 - (unsigned int)NL;      // k0 mod NL for k0 < 2 NL, as one v_min_u32
                    const float2 r = reinterpret_cast<const float2*>(rays)[k];
                    bool hit;
                    const float tt = ray_box_nr(ox, oy, fm(r.x, cr, r.y * sr), fm(r.y, cr, -(r.x * sr)), hl, hw, hit);
                    if (hit) atomicMin(&bs[(unsigned int)(pw >> 16) + k], __float_as_uint(tt));
                    if (COPO_PROFILE_SKIP & 256) { const int nh = __popcll(__ballot(hit)); COPO_COUNT(12, nh); }
                }
            }
        }
        if (p.n_boxes_lidar > 0) {
            // static boxes (buildings): per fan the boxes within reach (lane = box), then every ray of the fan against each of them -- the
            // vehicles' ray / box test with the box's own half extents; one lane per (fan, ray): a plain minimum
            pk_wave_sync();
            float bx = 0.0f, by = 0.0f, bc = 1.0f, bsn = 0.0f, bl = 0.0f, bw = 0.0f;
            if (lane < p.n_boxes) {
                const float* B = p.boxes + lane * COPO_BOX_STRIDE;
                bx = B[0]; by = B[1]; bc = B[2]; bsn = B[3]; bl = B[4]; bw = B[5];
            }
            for (int lp = 0; lp < cha; ++lp) {
                const float4 pi = pose((int)plist[ip0 + lp]);
                const float ddx = bx - pi.x, ddy = by - pi.y, lb = range + (bl + bw);
                for (unsigned long long mb = __ballot(lane < p.n_boxes && fm(ddx, ddx, ddy * ddy) <= lb * lb); mb; mb &= mb - 1ull) {
                    const int b = __ffsll((long long)mb) - 1;
                    const float X = readlane_f(bx, b), Y = readlane_f(by, b), Cb = readlane_f(bc, b), Sb = readlane_f(bsn, b);
                    const float HL = readlane_f(bl, b), HW = readlane_f(bw, b);
                    const float rx = X - pi.x, ry = Y - pi.y;
                    const float ox = -fm(rx, Cb, ry * Sb), oy = -fm(ry, Cb, -(rx * Sb));
                    const float cr = fm(pi.z, Cb, pi.w * Sb), sr = fm(pi.z, Sb, -(pi.w * Cb));
                    for (int k = lane; k < NL; k += 64) {
                        const float2 r = reinterpret_cast<const float2*>(rays)[k];
                        bool hit;
                        const float tt = ray_box_nr(ox, oy, fm(r.x, cr, r.y * sr), fm(r.y, cr, -(r.x * sr)), HL, HW, hit);
                        unsigned int* m = &bs[lp * NL + k];
                        if (hit && __float_as_uint(tt) < *m) *m = __float_as_uint(tt);
                    }
                }
            }
        }
        pk_wave_sync();
        if (COPO_PROFILE_SKIP & 4) {
        } else if (vec_out) {
            // (24-bit multiplies on provably small operands: the 32 / 64-bit multiply-adds the compiler picks for `int` index
            //  arithmetic run at a quarter of the rate)
            const unsigned int uNL = (unsigned int)NL & 0x1ffu, uO = (unsigned int)O & 0xffffu, unv = (unsigned int)nvec & 0x7fu;
            for (int q = lane; q < cha * nvec; q += 64) {
                const unsigned int lp = (unsigned int)(int)(((float)q + 0.5f) * inv_nvec) & 63u;
                const unsigned int k = (unsigned int)head + 4u * ((unsigned int)q - __umul24(lp, unv));
                const unsigned int* b = bs + (__umul24(lp, uNL) + k);
                uint4 b4;
                if ((NL & 3) == 0) b4 = *reinterpret_cast<const uint4*>(b);      // sh + lp NL + head + 4 m: a multiple of four words
                else b4 = make_uint4(b[0], b[1], b[2], b[3]);
                float4 v;
                v.x = __uint_as_float(b4.x) * inv_range; v.y = __uint_as_float(b4.y) * inv_range;
                v.z = __uint_as_float(b4.z) * inv_range; v.w = __uint_as_float(b4.w) * inv_range;
                *reinterpret_cast<float4*>(eobs + (__umul24((unsigned int)plist[ip0 + lp], uO) + (unsigned int)col_lidar + k)) = v;
            }
            const int nsc = NL - 4 * nvec;
            for (int q = lane; q < cha * nsc; q += 64) {
                const unsigned int lp = (unsigned int)(int)(((float)q + 0.5f) * inv_nsc) & 63u, r = (unsigned int)q - __umul24(lp, (unsigned int)nsc & 7u);
                const unsigned int k = r < (unsigned int)head ? r : r + 4u * unv;
                eobs[__umul24((unsigned int)plist[ip0 + lp], uO) + (unsigned int)col_lidar + k] = __uint_as_float(bs[__umul24(lp, uNL) + k]) * inv_range;
            }
        } else {
            for (int q = lane; q < cha * NL; q += 64) {
                const int lp = (int)(((float)q + 0.5f) * inv_nl), k = q - lp * NL;
                eobs[(int)plist[ip0 + lp] * O + col_lidar + k] = __uint_as_float(bs[q]) * inv_range;
            }
        }
        if (ip0 + CH < np) pk_wave_sync();
    }
    // side / lane-line detector beams (Bottleneck, Tollgate): the ray minima are written out, their storage holds the line masks
    detector_beams<false>(p, pose, plist, np, best, CH * NL, eobs, lane, 64, wtag, e);
}


}  // namespace copo

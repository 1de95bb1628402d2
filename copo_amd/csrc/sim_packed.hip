// "Packed" launch shape of the simulator step for large scene counts: S scenes per workgroup, S waves.
//
// The one-wave-per-scene shape (sim_kernels.hip, sim_step_kernel<.., ONE>) keeps agent slot n in lane n of the scene's wave: with
// N = 40 slots 24 of 64 lanes idle through every per-agent phase (dynamics, projection, observation block, the walk that builds the
// neighbour lists), with N = 10 it is 54 of 64 -- and that kernel is bound by VALU issue, not by memory.  Here the
// per-agent phases run DENSELY over the first ceil(S * N / 64) waves of the workgroup (thread a <-> scene a / N, slot a % N: 8
// scenes x 40 slots fill 5 waves completely; the other waves wait at the barrier and issue nothing), and the per-scene phases
// (masks, collision pairs, respawn hand-out, exact fall-backs of the neighbour lists, LiDAR) run wave w <-> scene w as before.
// The two kinds of phases talk through LDS only:
//
//   A1  packed   state loads, timers, kinematic bicycle                      -> pose4, st0 (acted | solid | eligible), cnt16
//   S1  scene    masks, collision over UNORDERED pairs of solid vehicles     -> crash
//   A2  packed   projection / termination / reward / info                   -> st1 (acted | solid | alive), reward, destinations
//   S2  scene    respawn places + hand-out, episode end, reset assignment    -> dec / aid, pose of spawned slots, rec0 / rec1, masks
//   A3  packed   spawn, neighbour walk (+ LiDAR reach masks), reset, row outputs, state write-back, state / navigation columns
//   S3  scene    clock words, global reward, exactly evaluated agents, LiDAR + detector beams
//
// Differences to the one-wave shape that remove instructions, not only idle lanes:
//   * the walk of the neighbour lists (lane = agent i, j uniform) also decides which vehicles j are within LiDAR reach of fan i
//     (one compare on the d^2 it has anyway): the LiDAR pair queue is filled from those 64-bit masks, the separate all-pairs
//     reach pass is gone.  Per-j thresholds live in the LDS record of j (0 / -1 for a slot that is not present / not solid), so
//     the walk tests no masks;
//   * collision candidates are enumerated as unordered pairs {a, b} of solid vehicles (the separating-axis test is symmetric bit
//     for bit: exact negations and commutative products), half the pairs of the (acting, solid) rectangle;
//   * vehicles standing on a respawn place: a distance pre-filter per place, the box test only for places with a candidate;
//   * the global reward is a DPP tree sum whenever every reward is in the range where fp64 sums are exact in any order.
// Results are the one-wave shape's (and the oracle's) bit for bit: same expressions (sim_device.h), same decision rules for the
// register formulation of the neighbour lists (neighbours_fast), every agent it cannot prove goes to neighbours_exact_one.
//
// Not supported here (the launcher keeps the other shapes for them): the traffic-light / communication extensions, configurations
// without the register formulation of the neighbour lists (K > 8, mean-field range of 0 or beyond the radius).
#include "sim_device.h"

namespace copo {

// ---- LDS layout of one scene (32-bit words) ---------------------------------------------------------------------------
struct PkLay {
    int NP, o_reach, o_plist, o_clist, o_st0, o_st1, o_crash, o_dest, o_cnt16, o_scal, o_u, o_rec1, o_dec, o_aid, o_perm, o_queue, scene_words;
};
enum : int { PK_MA = 0, PK_MP = 2, PK_MS = 4, PK_MEX = 6, PK_MODD = 8, PK_ENDING = 10, PK_BLK = 11 };      // words of the scalar area
__host__ __device__ inline PkLay pk_lay(int N, int NL, int CH) {
    PkLay y;
    y.NP = (N + 3) & ~3;
    int o = 4 * y.NP;                 // pose4 [NP] float4 {x, y, cos, sin} at word 0
    y.o_reach = o; o += 2 * y.NP;     // LiDAR reach masks [NP] u64
    y.o_plist = o; o += 16;           // present slots, ascending (u8 [64])
    y.o_clist = o; o += 16;           // solid vehicles before the step's terminations (collision pairs)
    y.o_st0 = o; o += 16;             // after A1: 1 acted | 2 solid | 4 eligible for a respawn
    y.o_st1 = o; o += 16;             // after A2: 1 acted | 2 solid | 4 alive
    y.o_crash = o; o += 16;
    y.o_dest = o; o += 16;            // exclusive destination id (+1) of the living vehicles
    y.o_cnt16 = o; o += 32;           // spawn counts (u16 [64])
    y.o_scal = o; o += 16;
    y.o_u = o;                        // work area, three lives:
    //   S1        collision candidates (u16 [N (N - 1) / 2])
    //   A2 .. S3  rec0 [NP] float4 {x, y, r2lo_j, r2hi_j}, rec1 [NP] float4 {reward fp64 lo, hi, lim2_j, reward fp32}, dec [NP], aid [NP], spawn permutation
    //   S3        LiDAR ray minima [CH][NL], pair queue (u16 [CH * N])
    y.o_rec1 = y.o_u + 4 * y.NP;
    y.o_dec = y.o_u + 8 * y.NP;
    y.o_aid = y.o_u + 9 * y.NP;
    y.o_perm = y.o_u + 10 * y.NP;
    y.o_queue = y.o_u + CH * NL + LIDAR_MIN_PAD;
    const int a = 10 * y.NP + COPO_MAX_SPAWNS / 2, b = CH * NL + LIDAR_MIN_PAD + (CH * N + 1) / 2, c = (N * N / 2 + 1) / 2 + 1;
    const int u = a > b ? (a > c ? a : c) : (b > c ? b : c);
    y.scene_words = (y.o_u + u + 3) & ~3;
    return y;
}
__host__ __device__ inline int pk_wg_words(int S, int N, int NL, int CH) {
    return ray_lds_words(NL) + S * 64 + S * pk_lay(N, NL, CH).scene_words;
}

struct RoadTabs {
    const float* rsegs;
    const float* rmeta;
    int32_t seg_rows;
};

__global__ void __launch_bounds__(COPO_SIM_MAX_BLOCK) sim_step_packed_kernel(const SimParams* __restrict__ pp, const float* __restrict__ act,
                                                                             StepOut out) {
    const SimParams& p = *pp;
    extern __shared__ unsigned int dyn[];
    const int tid = threadIdx.x, lane = tid & 63, nthreads = (int)blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, S = p.pack_scenes, NL = p.num_lasers, K = p.K;
    const PkLay Y = pk_lay(N, NL, p.chunk);
    float* rays = reinterpret_cast<float*>(dyn);
    const int o_wtag = ray_lds_words(NL), o_scenes = o_wtag + S * 64;
    const float hl = p.hl, hw = p.hw;
    const RoadTabs T{p.route_segs, p.route_meta, p.seg_rows};
    const int cap = capacity_of(p);
    for (int q = tid; q < 2 * NL; q += nthreads) rays[q] = p.ray_cs[q];
    // profiling build 4096 (`make prof SKIP=4096`, scripts/phase_packed.py): clock64 of wave w = scene w at every workgroup barrier -- slot k:
    // released from barrier k (0: kernel start), slot 8 + k: arrived at barrier k + 1 (its own work of the phase done)
#define PK_STAMP(slot) do { if ((COPO_PROFILE_SKIP & 4096) && p.dbg && lane == 0 && (int)blockIdx.x * S + wave < p.E) p.dbg[((size_t)blockIdx.x * S + wave) * 16 + (slot)] = (long long)clock64(); } while (0)
    PK_STAMP(0);

    // ---- A1 (packed): state, timers, kinematic bicycle ---------------------------------------------------------------
    // (the waves of a workgroup go to the four SIMDs of the compute unit in turn: with the packed work always on waves 0 .. k the SIMD of
    // waves 0 and 4 would carry two shares of it in every workgroup; the packed waves therefore start at a wave that rotates over the
    // workgroups a compute unit receives -- workgroup b goes to XCD b mod 8 and there to compute unit (b / 8) mod 32)
    const int rot = ((int)blockIdx.x >> 8) % S;
    const int ptid = ((wave + S - rot) % S) * 64 + lane;          // packed thread index of this thread
    const bool agent = ptid < S * N;
    const int s_a = agent ? (int)(((float)ptid + 0.5f) * (1.0f / (float)N)) : 0;
    const int n_a = agent ? ptid - s_a * N : 0;
    const int e_a = (int)blockIdx.x * S + s_a;
    const bool live = agent && e_a < p.E;
    unsigned int* sc_a = dyn + o_scenes + s_a * Y.scene_words;
    Slot s = Slot{};
    bool acted = false;
    float acc = 0.0f;
    int32_t t_env_a = 0, episode_a = 0;
    uint64_t seed_a = 0;
    if (agent) {
        uint8_t st0 = 0;
        if (live) {
            const int32_t* env = p.env + (size_t)e_a * 4;
            t_env_a = env[0];
            episode_a = env[1];
            seed_a = p.seeds[e_a];
            load_slot(p, e_a, n_a, s);
            if (!(COPO_PROFILE_SKIP & 1024)) slot_dynamics<false>(p, act, (size_t)e_a * N + n_a, s, acted, acc);
            else { acted = st_status(s.status) == ST_ALIVE; sincos_det(s.th, s.hs, s.hc); }
            const bool sol0 = st_status(s.status) != ST_EMPTY;
            // (a slot without a vehicle stands far away: the collision rows below then need no flag of the OTHER slot)
            reinterpret_cast<float4*>(sc_a)[n_a] = make_float4(sol0 ? s.x : 1.0e18f, s.y, s.hc, s.hs);
            const bool elig = n_a < cap && !acted && s.status == st_pack(ST_EMPTY, 0, 0);
            st0 = (uint8_t)((acted ? 1 : 0) | (sol0 ? 2 : 0) | (elig ? 4 : 0));
            reinterpret_cast<uint16_t*>(sc_a + Y.o_cnt16)[n_a] = (uint16_t)((uint32_t)s.spawncnt & 0xffffu);
        }
        reinterpret_cast<uint8_t*>(sc_a + Y.o_st0)[n_a] = st0;
        reinterpret_cast<uint8_t*>(sc_a + Y.o_crash)[n_a] = 0;
    }
    PK_STAMP(8);
    __syncthreads();
    PK_STAMP(1);

    // ---- S1 (wave w <-> scene w): masks, collision over unordered pairs of solid vehicles ---------------------------------
    unsigned int* sc = dyn + o_scenes + wave * Y.scene_words;
    const int e = (int)blockIdx.x * S + wave;
    const bool scene_live = e < p.E;
    float4* pose = reinterpret_cast<float4*>(sc);
    int32_t t_env = 0, episode = 0, next_aid = 0;
    uint64_t seed = 0;
    if (scene_live) {
        const int32_t* env = p.env + (size_t)e * 4;
        t_env = env[0];
        episode = env[1];
        next_aid = env[2];
        seed = p.seeds[e];
    }
    unsigned long long ma, mel;
    {
        const int b0 = lane < N ? (int)reinterpret_cast<const uint8_t*>(sc + Y.o_st0)[lane] : 0;
        ma = __ballot(b0 & 1);
        mel = __ballot(b0 & 4);
        uint8_t* crash = reinterpret_cast<uint8_t*>(sc + Y.o_crash);
        // row r = 1 .. N / 2: slot `lane` against slot (lane + r) mod N -- every unordered pair once (the last row of an even N pairs
        // every slot with its opposite: its first half is all of it).  Slots without a vehicle stand at x = 1e18.
        const bool sol_me = (b0 & 2) != 0;
        const float2 pme = lane < N ? *reinterpret_cast<const float2*>(&pose[lane]) : make_float2(0.0f, 0.0f);
        const float near2 = 4.0f * (hl * hl + hw * hw) * 1.001f;      // farther apart than two circumradii: no overlap
        uint16_t* nq = reinterpret_cast<uint16_t*>(sc + Y.o_u);
        int nn = 0;
        const int R = (COPO_PROFILE_SKIP & 16) ? 0 : (N >> 1);
        int jj = lane < N ? lane : 0;
        for (int r = 1; r <= R; r += 2) {      // two rows per turn: both poses are requested before either is used
            const int j0 = jj + 1 >= N ? jj + 1 - N : jj + 1, j1 = jj + 2 >= N ? jj + 2 - N : jj + 2;
            jj = j1;
            const float2 p0 = *reinterpret_cast<const float2*>(&pose[j0]), p1 = *reinterpret_cast<const float2*>(&pose[j1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr_ = r + h, jh = h ? j1 : j0;
                const float ddx = (h ? p1.x : p0.x) - pme.x, ddy = (h ? p1.y : p0.y) - pme.y;
                const int lanes = rr_ > R ? 0 : ((rr_ == R && (N & 1) == 0) ? R : N);
                const bool near = sol_me && lane < lanes && (fm(ddx, ddx, ddy * ddy) <= near2);
                const unsigned long long m = __ballot(near);
                if (m != 0ull) {
                    if (near) nq[nn + pk_mbcnt(m)] = (uint16_t)((lane << 8) | jh);
                    nn += __popcll(m);
                }
            }
        }
        pk_wave_sync();
        for (int q0 = 0; q0 < nn; q0 += 64) {
            if (q0 + lane < nn) {
                const int pk = (int)nq[q0 + lane], a = pk >> 8, b = pk & 255;
                const float4 pa = pose[a], pb = pose[b];
                if (obb_overlap2(pa.x, pa.y, pa.z, pa.w, hl, hw, pb.x, pb.y, pb.z, pb.w, hl, hw)) {
                    crash[a] = 1;       // (the mark of a vehicle that did not act -- a wreck -- is never read)
                    crash[b] = 1;
                }
            }
        }
    }
    PK_STAMP(9);
    __syncthreads();
    PK_STAMP(2);

    // ---- A2 (packed): projection, termination, reward ------------------------------------------------------------------
    uint8_t fl = 0;
    float rew = 0.0f, lcf_row = 0.0f;
    int32_t aid_row = -1;
    bool present = false;
    if (live) {
        const bool force_end = t_env_a + 1 >= 5 * p.horizon;
        bool term = false;
        lcf_row = s.lcf;
        aid_row = acted ? s.aid : -1;
        float* info_row = out.info ? out.info + ((size_t)e_a * N + n_a) * COPO_INFO_DIM : nullptr;
        if (acted && !(COPO_PROFILE_SKIP & 64)) {
            slot_project(p, T, s.hc, s.hs, reinterpret_cast<const uint8_t*>(sc_a + Y.o_crash)[n_a] != 0, force_end, acc, info_row, s, fl, rew, term);
        } else if (info_row) {
            for (int k = 0; k < COPO_INFO_DIM; ++k) info_row[k] = 0.0f;
        }
        present = acted;
        const int st = st_status(s.status);
        reinterpret_cast<uint8_t*>(sc_a + Y.o_st1)[n_a] = (uint8_t)((acted ? 1 : 0) | (st != ST_EMPTY ? 2 : 0) | (st == ST_ALIVE ? 4 : 0));
        const double rd = (double)rew;
        reinterpret_cast<float4*>(sc_a + Y.o_rec1)[n_a] = make_float4(__int_as_float(__double2loint(rd)), __int_as_float(__double2hiint(rd)), -1.0f, rew);
        if (p.n_spaces > 0)
            reinterpret_cast<uint8_t*>(sc_a + Y.o_dest)[n_a] = (uint8_t)(st == ST_ALIVE ? (int)p.route_meta[(s.route & 0xffff) * 4 + 3] : 0);
    } else if (agent) {
        reinterpret_cast<uint8_t*>(sc_a + Y.o_st1)[n_a] = 0;
        reinterpret_cast<float4*>(sc_a + Y.o_rec1)[n_a] = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
    }
    PK_STAMP(10);
    __syncthreads();
    PK_STAMP(3);

    // ---- S2 (scene): respawn places and hand-out, end of the episode, reset assignment, walk records -------------------------
    unsigned long long mp, ms;
    bool ending = false;
    {
        const int b1 = lane < N ? (int)reinterpret_cast<const uint8_t*>(sc + Y.o_st1)[lane] : 0;
        const unsigned long long solid_now = __ballot(b1 & 2), alive = __ballot(b1 & 4);
        float4 ps = lane < N ? pose[lane] : make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        const uint32_t c16 = lane < N ? (uint32_t)reinterpret_cast<const uint16_t*>(sc + Y.o_cnt16)[lane] : 0u;
        int32_t* dec = reinterpret_cast<int32_t*>(sc + Y.o_dec);
        int32_t* aidv = reinterpret_cast<int32_t*>(sc + Y.o_aid);
        bool spawned = false;
        int dec_word = 0;
        const bool no_respawn = t_env + 1 >= p.horizon;
        if (scene_live && !no_respawn && mel != 0ull && !(COPO_PROFILE_SKIP & 32)) {
            // respawn places whose region is clear of the vehicles standing now: the (place, vehicle) pairs closer than the two
            // circumradii are queued and box-tested together (a box test per place would run for 64 lanes with one or two candidates)
            const float rr = sqrtf(p.region_hl * p.region_hl + p.region_hw * p.region_hw) + sqrtf(hl * hl + hw * hw);
            const float rr2 = rr * rr * 1.001f;
            const bool sol = (b1 & 2) != 0;
            uint16_t* bq = reinterpret_cast<uint16_t*>(sc + Y.o_perm);      // (the spawn permutation of a reset comes later)
            unsigned int* blkw = sc + Y.o_scal + PK_BLK;
            constexpr int BQ_CAP = COPO_MAX_SPAWNS;
            if (lane == 0) *blkw = 0u;
            int nbq = 0;
            auto flush = [&]() {
                pk_wave_sync();
                for (int q0 = 0; q0 < nbq; q0 += 64) {
                    if (q0 + lane < nbq) {
                        const int ent = (int)bq[q0 + lane], q = ent >> 8;
                        const float4 sp4 = reinterpret_cast<const float4*>(p.safe_pose)[q];
                        const float4 pn = pose[ent & 255];
                        if (obb_overlap2(sp4.x, sp4.y, sp4.z, sp4.w, p.region_hl, p.region_hw, pn.x, pn.y, pn.z, pn.w, hl, hw)) atomicOr(blkw, 1u << q);
                    }
                }
                pk_wave_sync();
                nbq = 0;
            };
            // (lane q loads place q once: a load per iteration would put a memory round trip in front of every place)
            const float2 spq = lane < p.n_safe ? *reinterpret_cast<const float2*>(p.safe_pose + 4 * lane) : make_float2(0.0f, 0.0f);
            for (int q = 0; q < p.n_safe; ++q) {
                const float dx = ps.x - readlane_f(spq.x, q), dy = ps.y - readlane_f(spq.y, q);
                const bool pre = sol && (fm(dx, dx, dy * dy) <= rr2);
                const unsigned long long m = __ballot(pre);
                if (m != 0ull) {
                    const int c = __popcll(m);
                    if (nbq + c > BQ_CAP) flush();
                    if (pre) bq[nbq + pk_mbcnt(m)] = (uint16_t)((q << 8) | lane);
                    nbq += c;
                }
            }
            flush();
            const uint32_t clear = ~(uint32_t)__builtin_amdgcn_readfirstlane((int)*blkw) & (p.n_safe >= 32 ? 0xffffffffu : ((1u << p.n_safe) - 1u));
            uint32_t used = 0, taken = 0;
            int my_q = -1, my_route = -1;
            int32_t my_aid = 0;
            if (p.n_spaces > 0) {
                const int d = lane < N ? (int)reinterpret_cast<const uint8_t*>(sc + Y.o_dest)[lane] : 0;
                for (int k = 1; k <= p.n_spaces; ++k)
                    if (__ballot(d == k) != 0ull) taken |= 1u << (k - 1);
            }
            unsigned long long elig = mel;
            while (elig) {
                const int n = __ffsll((long long)elig) - 1;
                elig &= elig - 1;
                const uint32_t freem = clear & ~used;
                if (!freem) break;
                const int nfree = __popc(freem);
                const uint32_t cnt_n = (uint32_t)__builtin_amdgcn_readlane((int)c16, n);
                const uint32_t hh = hash_rng(seed, (uint32_t)n, cnt_n, (uint32_t)t_env, RNG_SPAWN);
                int pick = (int)(hh % (uint32_t)nfree);
                uint32_t m = freem;
                while (pick > 0) { m &= m - 1; --pick; }
                const int q = __ffs((int)m) - 1;
                used |= 1u << q;
                if (p.n_spaces > 0) {
                    const uint32_t hr = hash_rng(seed, (uint32_t)n, cnt_n, (uint32_t)episode, RNG_ROUTE);
                    const int r = pick_route_exclusive(p.route_meta, p.spawn_tab, p.safe_ids[q], hr, taken);
                    if (lane == n) my_route = r;
                }
                if (lane == n) { my_q = q; my_aid = next_aid; }
                next_aid += 1;
            }
            if (my_q >= 0) {      // the pose of the fresh vehicle (spawn_slot, A3, derives the same one for the state)
                const int sp = p.safe_ids[my_q];
                if (my_route < 0) {
                    const uint32_t h = hash_rng(seed, (uint32_t)lane, c16, (uint32_t)episode, RNG_ROUTE);
                    my_route = p.spawn_tab[sp * 4 + 0] + (int)(h % (uint32_t)p.spawn_tab[sp * 4 + 1]);
                }
                const float* g = p.route_segs + (size_t)my_route * p.seg_rows * COPO_SEG_STRIDE;
                spawn_pose(p, p.route_segs, p.spawn_tab, p.spawn_s, sp, ps.x, ps.y);
                ps.z = g[2];
                ps.w = g[3];
                pose[lane] = ps;
                spawned = true;
                dec_word = (my_q + 1) | ((my_route + 1) << 8);
                aidv[lane] = my_aid;
            }
        }
        const unsigned long long msp = __ballot(spawned);
        mp = ma | msp;
        ms = solid_now | msp;
        ending = scene_live && (alive | msp) == 0ull;
        // records of the neighbour walk: thresholds of j (0: not present -> never in range; -1: not solid -> never in LiDAR reach)
        const bool pres = (mp >> lane) & 1ull, sol = (ms >> lane) & 1ull;
        float rw = 0.0f;
        if (lane < N) {
            const float lim = p.lidar_range + sqrtf(hl * hl + hw * hw);
            const float lim2 = lim * lim * 1.00001f;      // (a superset test: the window pass decides every pair it is handed itself)
            reinterpret_cast<float4*>(sc + Y.o_u)[lane] = make_float4(ps.x, ps.y, pres ? p.nbr_r2lo : 0.0f, pres ? p.nbr_r2hi : 0.0f);
            float4* r1 = reinterpret_cast<float4*>(sc + Y.o_rec1) + lane;
            if (b1 & 1) {
                rw = r1->w;
                r1->z = sol ? lim2 : -1.0f;
            } else {
                *r1 = make_float4(0.0f, 0.0f, sol ? lim2 : -1.0f, 0.0f);
            }
        }
        const float arw = fabsf(rw);
        const unsigned long long odd = __ballot(pres && !(rw == 0.0f || (arw >= 9.5367431640625e-07f && arw <= 16.0f)));
        if (ending) {      // reset assignment (reset_env_wave0): spawn permutation, exclusive destinations in slot order
            int16_t* perm = reinterpret_cast<int16_t*>(sc + Y.o_perm);
            const uint32_t ep1 = (uint32_t)episode + 1u;
            if (lane == 0) {
                const int P = p.n_spawns;
                for (int i = 0; i < P; ++i) perm[i] = (int16_t)i;
                for (int i = 0; i < N; ++i) {
                    const uint32_t h = hash_rng(seed, (uint32_t)i, ep1, 0u, RNG_PERM);
                    const int j = i + (int)(h % (uint32_t)(P - i));
                    const int16_t t = perm[i];
                    perm[i] = perm[j];
                    perm[j] = t;
                }
            }
            pk_wave_sync();
            int route_fixed = -1;
            if (p.n_spaces > 0) {
                uint32_t taken = 0;
                for (int n = 0; n < cap; ++n) {
                    const uint32_t cnt_n = (uint32_t)__builtin_amdgcn_readlane((int)c16, n);
                    const uint32_t h = hash_rng(seed, (uint32_t)n, cnt_n, ep1, RNG_ROUTE);
                    const int r = pick_route_exclusive(p.route_meta, p.spawn_tab, (int)perm[n], h, taken);
                    if (lane == n) route_fixed = r;
                }
            }
            dec_word = lane < cap ? (int)(0x40000000u | (uint32_t)perm[lane < N ? lane : 0] | ((uint32_t)(route_fixed + 1) << 16)) : (int)0x20000000u;
        }
        if (lane < N) dec[lane] = dec_word;
        if (lane == 0) {
            *reinterpret_cast<unsigned long long*>(sc + Y.o_scal + PK_MP) = mp;
            *reinterpret_cast<unsigned long long*>(sc + Y.o_scal + PK_MEX) = 0ull;
            *reinterpret_cast<unsigned long long*>(sc + Y.o_scal + PK_MODD) = odd;
            sc[Y.o_scal + PK_ENDING] = ending ? 1u : 0u;
        }
    }
    PK_STAMP(11);
    __syncthreads();
    PK_STAMP(4);

    // ---- A3 (packed): spawn, neighbour walk + LiDAR reach, reset, row outputs, state, state / navigation columns ------------
    if (live) {
        const float4* rec0 = reinterpret_cast<const float4*>(sc_a + Y.o_u);
        const float4* rec1 = reinterpret_cast<const float4*>(sc_a + Y.o_rec1);
        const int dw = reinterpret_cast<const int32_t*>(sc_a + Y.o_dec)[n_a];
        const bool ending_a = sc_a[Y.o_scal + PK_ENDING] != 0u;
        if (dw > 0 && dw < 0x20000000) {
            const int q = (dw & 0xff) - 1, route = (dw >> 8) - 1;
            spawn_slot(p, p.route_segs, p.spawn_tab, p.spawn_s, seed_a, (uint32_t)episode_a, n_a, p.safe_ids[q],
                       reinterpret_cast<const int32_t*>(sc_a + Y.o_aid)[n_a], s, false, 0u, 0.0f, route);
            present = true;
            fl = COPO_F_SPAWNED;
            lcf_row = s.lcf;
        }
        // the walk: every lane takes its scene's slots j = 0 .. N - 1 in order (neighbours_fast's rules; the thresholds of a slot that
        // is not present / not solid are 0 / -1, so it is never in range / in reach)
        const float4 me0 = rec0[n_a];
        const float xi = me0.x, yi = me0.y;
        uint32_t a0 = NBR_SENT, a1 = NBR_SENT, a2 = NBR_SENT, a3 = NBR_SENT, a4 = NBR_SENT, a5 = NBR_SENT, a6 = NBR_SENT,
                 a7 = NBR_SENT, a8 = NBR_SENT;
        double sum = 0.0;
        int cnt = 0;
        bool unc = false;
        unsigned int rlo = 0u, rhi = 0u;
        const int NW = (COPO_PROFILE_SKIP & 1) ? 0 : N;
        float4 r0 = rec0[0], r1 = rec1[0];
        for (int j = 0; j < NW; ++j) {
            const int jn = j + 1 < N ? j + 1 : j;
            const float4 n0 = rec0[jn], n1 = rec1[jn];        // the next records, requested before this step's arithmetic
            const float dx = xi - r0.x, dy = yi - r0.y;
            const float d2 = __builtin_fmaf(dy, dy, dx * dx);
            const bool other = j != n_a;
            const bool in = other && d2 < r0.z, inhi = other && d2 < r0.w;
            unc |= in != inhi;
            const double rj = __hiloint2double(__float_as_int(r1.y), __float_as_int(r1.x));
            sum = fma(in ? 1.0 : 0.0, rj, sum);       // (exact either way: the rewards that may differ by the order are caught below)
            cnt += in ? 1 : 0;
            const uint32_t key = in ? ((__float_as_uint(d2) & ~63u) | (uint32_t)j) : NBR_SENT;
            a8 = umed3(a7, a8, key); a7 = umed3(a6, a7, key); a6 = umed3(a5, a6, key); a5 = umed3(a4, a5, key);
            a4 = umed3(a3, a4, key); a3 = umed3(a2, a3, key); a2 = umed3(a1, a2, key); a1 = umed3(a0, a1, key);
            a0 = a0 < key ? a0 : key;
            const uint32_t rb = (other && d2 <= r1.z) ? 1u : 0u;
            if (j < 32) rlo |= rb << j;
            else rhi |= rb << (j - 32);
            r0 = n0;
            r1 = n1;
        }
        const uint32_t ak[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8};
        const uint32_t tlo = p.mf_key_lo, thi = p.mf_key_hi;
        int mf = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mf += ak[k] < tlo ? 1 : 0;
            unc |= ak[k] >= tlo && ak[k] < thi;
        }
        unc |= a8 < thi;
#pragma unroll
        for (int k = 0; k < 8; ++k) unc |= (ak[k + 1] != NBR_SENT) && (ak[k + 1] - ak[k] < 128u);
        for (unsigned long long mo = pk_u64(sc_a + Y.o_scal + PK_MODD); mo; mo &= mo - 1ull) {      // (rare) a reward outside the exact-sum range
            const int b = __ffsll((long long)mo) - 1;
            const float4 rb = rec0[b];
            const float dx = xi - rb.x, dy = yi - rb.y;
            unc |= b != n_a && __builtin_fmaf(dy, dy, dx * dx) < p.nbr_r2hi;
        }
        const bool me = present;
        const size_t row0 = (size_t)e_a * N + n_a;
        if (me && unc) atomicOr(reinterpret_cast<unsigned long long*>(sc_a + Y.o_scal + PK_MEX), 1ull << n_a);
        if (!(me && unc)) {
            if (out.nbr_cnt) out.nbr_cnt[row0] = me ? cnt : 0;
            if (out.mf_cnt) out.mf_cnt[row0] = me ? mf : 0;
            if (out.nei_rew) out.nei_rew[row0] = (me && cnt) ? (float)(sum / (double)cnt) : 0.0f;
        }
        if (me && !unc) {
            if (out.nbr_idx) {
                int32_t* row = out.nbr_idx + row0 * K;
                if (K == 8) {
                    int4 lo4, hi4;
                    lo4.x = 0 < cnt ? (int)(a0 & 63u) : -1; lo4.y = 1 < cnt ? (int)(a1 & 63u) : -1;
                    lo4.z = 2 < cnt ? (int)(a2 & 63u) : -1; lo4.w = 3 < cnt ? (int)(a3 & 63u) : -1;
                    hi4.x = 4 < cnt ? (int)(a4 & 63u) : -1; hi4.y = 5 < cnt ? (int)(a5 & 63u) : -1;
                    hi4.z = 6 < cnt ? (int)(a6 & 63u) : -1; hi4.w = 7 < cnt ? (int)(a7 & 63u) : -1;
                    reinterpret_cast<int4*>(row)[0] = lo4;
                    reinterpret_cast<int4*>(row)[1] = hi4;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < K) row[k] = k < cnt ? (int)(ak[k] & 63u) : -1;
                }
            }
            if (out.nbr_dist) {
                float* row = out.nbr_dist + row0 * K;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < K) {
                        float dv = 0.0f;
                        if (k < cnt) {
                            const float4 rj = rec0[(int)(ak[k] & 63u)];
                            const double dx = (double)xi - (double)rj.x, dy = (double)yi - (double)rj.y;
                            dv = (float)sqrt(dx * dx + dy * dy);
                        }
                        row[k] = dv;
                    }
            }
        }
        unsigned int* reach = sc_a + Y.o_reach;
        reach[2 * n_a] = me ? rlo : 0u;
        reach[2 * n_a + 1] = me ? rhi : 0u;

        // end of the episode: the scene starts over (the lists above are the old scene's, the observation the new one's)
        uint8_t fl_out = fl;
        float lcf_out = lcf_row;
        if (ending_a) {
            episode_a += 1;
            if (dw & 0x40000000) {
                const int sp = dw & 0xffff, rf = ((dw >> 16) & 0x1fff) - 1;
                spawn_slot(p, p.route_segs, p.spawn_tab, p.spawn_s, seed_a, (uint32_t)episode_a, n_a, sp, n_a, s, false, 0u, 0.0f, rf);
                present = true;
                fl_out |= COPO_F_SPAWNED;
                if (!acted) lcf_out = s.lcf;
                reinterpret_cast<float4*>(sc_a)[n_a] = make_float4(s.x, s.y, s.hc, s.hs);
            } else {
                s.status = st_pack(ST_EMPTY, 0, 0);
                present = false;
            }
            fl_out |= COPO_F_ENV_RESET;
        }
        if (out.rew) out.rew[row0] = rew;
        if (out.flags) out.flags[row0] = fl_out;
        if (out.lcf) out.lcf[row0] = lcf_out;
        if (out.agent_id) out.agent_id[row0] = aid_row;
        store_slot(p, e_a, n_a, s);
        if (!(COPO_PROFILE_SKIP & 8)) ego_navi_obs<false>(p, T, s.hc, s.hs, s, present, out.obs ? out.obs + row0 * p.O : nullptr, ending_a ? 0 : t_env_a + 1, ending_a);
    }
    PK_STAMP(12);
    __syncthreads();
    PK_STAMP(5);

    // ---- S3 (scene): clock words, global reward, exactly evaluated agents, LiDAR --------------------------------------------
    if (scene_live) {
        if (lane == 0) {
            int32_t* env = p.env + (size_t)e * 4;
            env[0] = ending ? 0 : t_env + 1;
            env[1] = episode + (ending ? 1 : 0);
            env[2] = ending ? cap : next_aid;
        }
        const float4 r0 = lane < N ? reinterpret_cast<const float4*>(sc + Y.o_u)[lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float rwl = lane < N ? reinterpret_cast<const float4*>(sc + Y.o_rec1)[lane].w : 0.0f;
        const unsigned long long odd = pk_u64(sc + Y.o_scal + PK_MODD);
        const unsigned long long mex = pk_u64(sc + Y.o_scal + PK_MEX);
        if (out.glob_rew) {      // LCFEnv.step: sum(r.values()) / len(r.values()), fp64 in slot order
            const int np = __popcll(mp);
            double gs = 0.0;
            if (odd == 0ull) {       // every partial sum of such rewards is exact: any order gives the slot order's bits
                gs = pk_wave_sum_f64(((mp >> lane) & 1ull) ? (double)rwl : 0.0);
                gs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(gs), 63), __builtin_amdgcn_readlane(__double2loint(gs), 63));
                gs = 0.0 + gs;       // (the serial sum starts from +0.0: a total of -0.0 becomes +0.0 there)
            } else {
                for (unsigned long long m = mp; m; m &= m - 1ull) gs += (double)readlane_f(rwl, __ffsll((long long)m) - 1);
            }
            if (lane == 0) out.glob_rew[e] = np ? (float)(gs / (double)np) : 0.0f;
        }
        for (unsigned long long mx = mex; mx; mx &= mx - 1ull)        // the agents the walk's registers do not decide
            neighbours_exact_one(p, e, lane, __ffsll((long long)mx) - 1, r0.x, r0.y, rwl, mp, odd, out);
        if (p.dbg && lane == 0) p.dbg[(size_t)e * 8 + 7] = mex ? 16 + __popcll(mex) : 1;
        if (out.obs) {
            unsigned long long fp = mp, fs = ms;
            if (ending) fp = fs = (cap >= 64 ? ~0ull : ((1ull << cap) - 1ull));
            pk_wave_sync();          // (the ray minima reuse the walk records the exact evaluation just read)
            const float4* pose4 = reinterpret_cast<const float4*>(sc);
            const unsigned int rl = lane < N ? sc[Y.o_reach + 2 * lane] : 0u, rh = lane < N ? sc[Y.o_reach + 2 * lane + 1] : 0u;
            lidar_by_wave(p, [pose4](int v) { return pose4[v]; }, reinterpret_cast<uint8_t*>(sc + Y.o_plist), rl, rh, sc + Y.o_u,
                          reinterpret_cast<uint16_t*>(sc + Y.o_queue), rays, reinterpret_cast<int*>(dyn + o_wtag) + wave * 64, e, lane, fp, fs, ending, out.obs);
        }
    }
    PK_STAMP(6);
#undef PK_STAMP
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static hipError_t pk_lds_attr() {
    static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(sim_step_packed_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    return once;
}

int sim_packed_chunk(const SimParams& p) {
    if (p.chunk_one_wave > 0) return p.chunk_one_wave < p.N ? p.chunk_one_wave : p.N;
    int ch = 1280 / p.num_lasers;      // (240 beams: fans of 5 -- 3: 145 us, 5: 109 us per launch of 16 384 scenes x 10 slots, 7: 138)
    ch = ch < 2 ? 2 : (ch > 10 ? 10 : ch);
    return ch < p.N ? ch : p.N;
}

size_t sim_packed_lds_bytes(const SimParams& p, int S) { return (size_t)pk_wg_words(S, p.N, p.num_lasers, sim_packed_chunk(p)) * sizeof(unsigned int); }

// scenes per workgroup: FOUR (one wave per SIMD), fewer when four do not fit 64 KB of LDS.  Measured at 16 384 scenes, 8 .. 20 slots x 72
// beams and 10 slots x 240 beams (scripts/bench_sim.py --blocks -2 .. -8): 4 scenes are the fastest everywhere, 2 / 3 within 2-6 %, 8
// within 3-8 %, and 5 or 6 -- the counts that fill the packed waves' lanes best -- 20-30 % behind: a workgroup's waves are dealt over the
// four SIMDs, so five or six leave one SIMD with twice the others' work; lane use of the packed phases matters less than that.
int sim_packed_default_scenes(const SimParams& p) {
    for (int S = 4; S >= 2; --S)
        if (sim_packed_lds_bytes(p, S) <= 64 * 1024) return S;
    return 0;
}

bool sim_packed_supported(const SimParams& p) {
    return p.nbr_fast != 0 && p.col_tl < 0 && p.col_comm < 0 && sim_packed_default_scenes(p) > 0;
}

hipError_t launch_sim_step_packed(const SimParams& p, const SimParams* p_dev, const float* act, const StepOut& out, int S, hipStream_t stream) {
    if (hipError_t a = pk_lds_attr(); a != hipSuccess) return a;
    const size_t lds = sim_packed_lds_bytes(p, S);
    if (lds > 96 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sim_step_packed_kernel, dim3((p.E + S - 1) / S), dim3(64 * S), lds, stream, p_dev, act, out);
    return hipGetLastError();
}

}  // namespace copo

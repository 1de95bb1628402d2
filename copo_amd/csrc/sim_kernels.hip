// Vectorised multi-agent driving step for gfx950: one workgroup per env, the env's N <= 64 agent slots
// live in the lanes of wave 0 (state in registers), pair work (collision, neighbour ranking, LiDAR
// candidates) is wave64-ballot based with the other agent along the lanes, and the LiDAR ray casts
// are spread over every thread of the workgroup.  Per-scene poses / masks are staged in LDS.
//
// Replaces, behind the C ABI of include/copo_hip.h:
//   MultiAgent*Env.step / reset          (MetaDrive 0.2.5; call site utils/env_wrappers.py:95) -- MetaDrive's published
//                                        semantics restated, pinned through the reference's populations
//   CCEnv._update_distance_map/_find_in_range/step   (utils/env_wrappers.py:89-158)
//   LCFEnv.step reward block + _add_lcf  (utils/env_wrappers.py:307-418)
//
// HBM traffic per agent-step: state 64 B in + 64 B out, action 8 B, obs 4*O B,
// row outputs ~50 B; everything else stays in LDS / registers.
#include "sim_device.h"

// Profiling builds only (`make prof SKIP=<mask>`, a separate .so that the package never loads): phases compiled out to
// split the instruction count -- 1 neighbour lists, 2 LiDAR windows + box tests, 4 LiDAR write-out, 8 state / navigation
// block, 16 collision pairs, 32 respawn, 64 projection / termination, 128 register formulation of the neighbour lists off;
// 256: nothing compiled out, the LiDAR phase COUNTS its work into the debug rows instead ([E][16] then: 8 queued pairs,
// 9 pair batches, 10 box tests, 11 test batches, 12 hits).  The shipped library is built without it.
// (the macro's default lives in sim_device.h)
#define COPO_DBG_STRIDE ((COPO_PROFILE_SKIP & 0x5300) ? 16 : 8)
// (mask 512: the wave roles' own finishing times -- slot 8 wave 0, slot 9 wave 1, slot 10 the last LiDAR wave; scripts/phase_sim.py)
#define COPO_ROLE_STAMP(slot) do { if ((COPO_PROFILE_SKIP & 512) && p.dbg && lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(p.dbg) + (size_t)e * 16 + (slot), (unsigned long long)clock64()); } while (0)
// (COPO_COUNT: sim_device.h)

namespace copo {


struct __align__(16) EnvLds {
    float x[64], y[64], cs[64], sn[64], rew[64];
    uint8_t plist[64], slist[64];   // slots of the present agents / solid vehicles, ascending (LiDAR pair list)
    uint32_t n_entries;             // total list entries of the scene (pair-parallel neighbour lists; their per-list counters
                                    // live behind the list storage in dynamic LDS)
    uint8_t alist[64], clist[64];   // acting agents / solid vehicles before the step's terminations (collision pairs)
    unsigned long long m_acted, m_present, m_solid;
    uint8_t crash[64];
    const float* rsegs;        // route segment records: the LDS copy of the step kernel (small maps) or global memory
    const float* rmeta;        // route meta records, same
    const int32_t* stab;       // spawn table, same
    const float* sps;          // spawn offsets, same
    int32_t ending;
    unsigned long long nbr_exact, nbr_exact2, nbr_odd;   // wave roles: agents the two neighbour waves could not decide (evaluated exactly next
                                                         // to the LiDAR write-out) / slots with odd rewards
    int32_t seg_rows;          // road records per route in the device tables (longest route + its terminal record)
};

constexpr int ROUTE_LDS_MAX_BYTES = 16 * 1024;
// dynamic LDS: [slots][rays] LiDAR minima, then the route-table copy
// (the neighbour phase borrows it for its [slots][slots] list-order rewards and a reset for its spawn permutation);
// then the ray direction table, then the route-table copy
// Work areas in dynamic LDS are sized for a CHUNK of `ch` present agents (p.chunk: all N slots when several waves share a
// scene, 8 when one wave owns it -- the per-scene footprint decides how many scenes a compute unit holds):
// neighbour phase: list distances (fp64) [ch][N] -- later reused for the rewards in list order --, list slots (u8) [ch][N],
// ranks (u8) [ch][N], entry directory (u16) [ch*N]; then the spawn permutation of a reset (int16 [COPO_MAX_SPAWNS])
// then the list lengths / mean-field counts [64] + [64] (LDS atomics)
__device__ __host__ inline int nbr_lds_words(int ch, int n_agents) { return 3 * ch * n_agents + 4 + 128; }
// LiDAR phase: ray minima [ch][n_lasers]; when one wave owns the scene (ch < n_agents) also the queue of (fan, vehicle) pairs
// in range, uint16 [ch * n_agents], behind them (mostly inside what the neighbour phase needs anyway)
__device__ __host__ inline int lidar_queue_words(int ch, int n_agents) { return ch < n_agents ? (ch * n_agents + 1) / 2 : 0; }
// (`nch`: the chunk of the pair-parallel neighbour lists -- smaller than the LiDAR chunk when one wave owns the scene: that path
// only runs for the scenes the register formulation declines)
__device__ __host__ inline int lidar_lds_words(int ch, int nch, int n_agents, int n_lasers) {
    const int a = ch * n_lasers + LIDAR_MIN_PAD + lidar_queue_words(ch, n_agents), b = nbr_lds_words(nch, n_agents) + COPO_MAX_SPAWNS / 2;
    const int c = a > b ? a : b;
    return ((c > 256 ? c : 256) + 3) & ~3;     // (>= the 64 float4 records of neighbours_fast)
}
constexpr int LIDAR_WAVE_WORDS = 64;   // per wave: head flags of the box-test batches
__device__ __forceinline__ float* lds_rays(const SimParams& p) {
    extern __shared__ unsigned int dyn[];
    return reinterpret_cast<float*>(dyn + lidar_lds_words(p.chunk, p.nbr_chunk, p.N, p.num_lasers));
}
__device__ __forceinline__ int16_t* lds_perm(const SimParams& p) {
    extern __shared__ unsigned int dyn[];
    return reinterpret_cast<int16_t*>(dyn + nbr_lds_words(p.nbr_chunk, p.N));
}
__device__ __host__ inline int route_table_floats(int n_routes, int seg_rows) {
    return n_routes * (seg_rows * COPO_SEG_STRIDE + 4);
}


// Full reset of one env by wave 0 (all lanes call; lane n < N owns slot n).
__device__ __forceinline__ void reset_env_wave0(const SimParams& p, EnvLds& L, uint64_t seed, uint32_t episode, int lane,
                                                Slot& s) {
    int16_t* perm = lds_perm(p);
    if (lane == 0) {
        const int P = p.n_spawns;
        for (int i = 0; i < P; ++i) perm[i] = (int16_t)i;
        for (int i = 0; i < p.N; ++i) {
            const uint32_t h = hash_rng(seed, (uint32_t)i, episode, 0u, RNG_PERM);
            const int j = i + (int)(h % (uint32_t)(P - i));
            const int16_t t = perm[i];
            perm[i] = perm[j];
            perm[j] = t;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // population capacity (curriculum): slots beyond it start empty and never respawn
    const int cap = capacity_of(p);
    int route_fixed = -1;
    if (p.n_spaces > 0) {          // exclusive destinations: handed out in slot order (nobody of the old episode holds one)
        uint32_t taken = 0;
        for (int n = 0; n < cap; ++n) {
            const uint32_t cnt_n = (uint32_t)__builtin_amdgcn_readlane(s.spawncnt, n) & 0xffffu;
            const uint32_t h = hash_rng(seed, (uint32_t)n, cnt_n, episode, RNG_ROUTE);
            const int r = pick_route_exclusive(L.rmeta, L.stab, (int)perm[n], h, taken);
            if (lane == n) route_fixed = r;
        }
    }
    if (lane < cap) spawn_slot(p, L.rsegs, L.stab, L.sps, seed, episode, lane, (int)perm[lane], lane, s, false, 0u, 0.0f, route_fixed);
    else if (lane < p.N) s.status = st_pack(ST_EMPTY, 0, 0);
}

// Slot lists of the present agents / solid vehicles (ascending), by wave 0, from the masks of the scene.
__device__ __forceinline__ void build_lists(EnvLds& L, int lane, unsigned long long present, unsigned long long solid,
                                            bool for_neighbours = true) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    if ((present >> lane) & 1ull) L.plist[__popcll(present & lt)] = (uint8_t)lane;
    if ((solid >> lane) & 1ull) L.slist[__popcll(solid & lt)] = (uint8_t)lane;
    (void)for_neighbours;        // (the pair-parallel neighbour phase clears its own counters)
}

// neighbour lists + reward reductions (CCEnv / LCFEnv).  Precondition: build_lists ran and is visible.
//
// The reference orders by d = sqrt_rn(dx^2 + dy^2) in float64 (np.linalg.norm) with ties in slot order and tests
// `d < radius` (env_wrappers.py:125-158).  Pair-parallel in three steps:
//   1. one lane per ORDERED pair (i, j) of present agents: an fp32 distance decides which pairs can be in range, those
//      evaluate the reference's fp64 expression and append (d, j) to i's list in LDS (list order = arrival order);
//   2. one lane per list entry: rank = number of entries of the same list that sort before it by (d, slot) -- the stable
//      `sorted` of the reference -- then the first K ranks go to nbr_idx / nbr_dist;
//   3. rewards scattered to list order, one lane per agent adds them up in that order in fp64 (:321-325).
// Rows of absent slots: nbr_cnt / mf_cnt / nei_rew = 0; their nbr_idx / nbr_dist rows are NOT written (like obs rows).
//
// Communication (CCEnv.step :102-118, LCFEnv.step :362-371): the message columns of agent i hold the comm actions of
// its `comm_nb` nearest neighbours (zeros for a neighbour that was not given an action this step, for a missing
// neighbour, and everywhere after a reset: `fresh`), optionally followed by the neighbour's relative position.
template <bool EXT>
__device__ __forceinline__ void neighbours_phase(const SimParams& p, EnvLds& L, int e, int tid, int nthreads,
                                                 const StepOut& out, const float* __restrict__ act = nullptr,
                                                 unsigned long long acted_mask = 0ull, bool fresh = true) {
    extern __shared__ unsigned int dyn[];
    const int N = p.N, K = p.K;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = nthreads >> 6;
    const bool comm = EXT && p.col_comm >= 0 && out.obs != nullptr;
    const int CS = p.comm_size, CD = p.comm_size + 3 * p.comm_pos;
    const unsigned long long present = L.m_present;
    const int np = __popcll(present);
    const size_t base = (size_t)e * N;
    const double R = (double)p.neighbours_distance, M = (double)p.mf_distance;
    const int CH = p.nbr_chunk > 0 ? p.nbr_chunk : N;                    // present agents whose lists are in LDS at a time
    double* nb_d = reinterpret_cast<double*>(dyn);                       // [CH][N]
    float* srt = reinterpret_cast<float*>(dyn);                          // [CH][N], after the ranks are known
    uint8_t* nb_j = reinterpret_cast<uint8_t*>(dyn + 2 * CH * N);         // [CH][N]
    uint8_t* nb_rk = nb_j + CH * N;                                       // [CH][N]
    uint16_t* ent = reinterpret_cast<uint16_t*>(dyn + 2 * CH * N + (CH * N) / 2 + 2);   // [CH*N]: local list << 8 | pos
    uint32_t* ncnt = dyn + 3 * CH * N + 4;                                // [64] list lengths (low half) | outside candidates (high half)
    uint32_t* mfcn = ncnt + 64;                                           // [64] mean-field counts
    if (tid < 64) {
        ncnt[tid] = 0;
        mfcn[tid] = 0;
    }
    if (tid == 0) L.n_entries = 0;
    __syncthreads();
    const int tail_wave = nwaves > 1 ? 1 : 0;       // per-slot tails run on a wave that the caller's next (wave 0) phase does not need
    // slots without an agent: empty lists (counts and reward only; their nbr_idx / nbr_dist rows are not written)
    if (wave == tail_wave && lane < N && !((present >> lane) & 1ull)) {
        if (out.nbr_cnt) out.nbr_cnt[base + lane] = 0;
        if (out.mf_cnt) out.mf_cnt[base + lane] = 0;
        if (out.nei_rew) out.nei_rew[base + lane] = 0.0f;
        if (p.lists_for_absent)      // (the stateless op fills these rows too)
            for (int k = 0; k < K; ++k) {
                if (out.nbr_idx) out.nbr_idx[(base + lane) * K + k] = -1;
                if (out.nbr_dist) out.nbr_dist[(base + lane) * K + k] = 0.0f;
            }
    }
    if (wave == nwaves - 1 && out.glob_rew) {  // LCFEnv.step: sum(r.values()) / len(r.values()) in slot order, fp64
        double gs = 0.0;
        unsigned long long m = present;
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            gs += (double)L.rew[j];
        }
        if (lane == 0) out.glob_rew[e] = np ? (float)(gs / (double)np) : 0.0f;
    }
    const float r2hi = p.neighbours_distance * p.neighbours_distance * 1.00001f;   // fp32 d^2 errs by < 2e-7 relative
    for (int ia0 = 0; ia0 < np; ia0 += CH) {
        const int cha = np - ia0 < CH ? np - ia0 : CH;                   // lists of this pass: present agents ia0 .. ia0 + cha - 1
        // ---- 1. candidate pairs -------------------------------------------------------------------------------------
        // (a) one lane per ordered pair: an fp32 distance decides which pairs CAN be in range; those are appended to the
        //     list of i.  (b) one lane per appended entry evaluates the reference's fp64 expression -- a batch of 64 pairs
        //     almost always holds a candidate, so evaluating in (a) ran the fp64 block for every batch; the entries are
        //     ~30 % of the pairs.  A candidate that turns out to be outside (the fp32 test errs by < 1e-5 relative: a 0.2 mm
        //     shell) keeps its place with distance +inf: it sorts behind every neighbour and is counted out (ncnt >> 16).
        {
            const int npair = cha * np;
            const float inv_cha = 1.0f / (float)cha;
            for (int c0 = wave * 64; c0 < npair; c0 += nwaves * 64) {
                const int c = c0 + lane;
                const bool live = c < npair;
                const int ib = live ? (int)(((float)c + 0.5f) * inv_cha) : 0;
                const int la = live ? c - ib * cha : 0;                   // consecutive lanes: consecutive LISTS (no counter clash)
                const int i = L.plist[ia0 + la], j = L.plist[ib];
                const float fx = L.x[i] - L.x[j], fy = L.y[i] - L.y[j];
                const bool cand = live && i != j && fx * fx + fy * fy < r2hi;
                const unsigned long long m = __ballot(cand);
                if (m) {
                    unsigned int e0 = 0;
                    if (lane == 0) e0 = atomicAdd(&L.n_entries, (unsigned int)__popcll(m));
                    e0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)e0);
                    if (cand) {
                        const unsigned int pos = atomicAdd(&ncnt[i], 1u) & 0xffffu;
                        nb_j[la * N + pos] = (uint8_t)j;
                        ent[e0 + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((la << 8) | pos);
                    }
                }
            }
        }
        __syncthreads();
        const int T = (int)L.n_entries;
        for (int q = tid; q < T; q += nthreads) {
            const unsigned int pk = ent[q];
            const int la = (int)(pk >> 8), pos = (int)(pk & 255u);
            const int i = L.plist[ia0 + la], j = nb_j[la * N + pos];
            const double dx = (double)L.x[i] - (double)L.x[j], dy = (double)L.y[i] - (double)L.y[j];
            double d = sqrt(dx * dx + dy * dy);
            if (d < R) {
                if (d <= M) atomicAdd(&mfcn[i], 1u);
            } else {
                d = __longlong_as_double(0x7ff0000000000000ll);
                atomicAdd(&ncnt[i], 0x10000u);
            }
            nb_d[la * N + pos] = d;
        }
        __syncthreads();
        // ---- 2. ranks, first K entries of every list ------------------------------------------------------------------
        for (int q = tid; q < T; q += nthreads) {
            const unsigned int pk = ent[q];
            const int la = (int)(pk >> 8), pos = (int)(pk & 255u);
            const int i = L.plist[ia0 + la];
            const int cnt = (int)(ncnt[i] & 0xffffu);       // candidates (the +inf ones rank last)
            const double d = nb_d[la * N + pos];
            const int j = nb_j[la * N + pos];
            // rank = entries of the list that sort before this one by (d, slot).  Distances are >= 0, so the upper word of
            // the fp64 pattern orders them except when two upper words agree (distances within 1e-6 of each other: rare) --
            // only then are the full values and the slots compared
            const unsigned int* nb_hi = reinterpret_cast<const unsigned int*>(nb_d) + 1;
            const unsigned int dh = nb_hi[2 * (la * N + pos)];
            int rank = 0;
            for (int k = 0; k < cnt; ++k) {
                const unsigned int dkh = nb_hi[2 * (la * N + k)];
                rank += dkh < dh ? 1 : 0;
                if (__ballot(dkh == dh && k != pos) != 0ull) {
                    const double dk = nb_d[la * N + k];
                    const int jk = nb_j[la * N + k];
                    rank += (dkh == dh && (dk < d || (dk == d && jk < j))) ? 1 : 0;
                }
            }
            nb_rk[la * N + pos] = (uint8_t)rank;
            if (rank < K && dh != 0x7ff00000u) {
                if (out.nbr_idx) out.nbr_idx[(base + i) * K + rank] = j;
                if (out.nbr_dist) out.nbr_dist[(base + i) * K + rank] = (float)d;
            }
            if (comm && rank < p.comm_nb && dh != 0x7ff00000u) {
                float* qm = out.obs + (base + i) * p.O + p.col_comm + rank * CD;
                const bool spoke = !fresh && ((acted_mask >> j) & 1ull) && act != nullptr;
                for (int k = 0; k < CS; ++k) qm[k] = spoke ? act[(base + j) * p.act_dim + 2 + k] : 0.0f;
                if (p.comm_pos) {
                    float ex[3] = {0.0f, 0.0f, 0.0f};
                    if (spoke) {   // neighbour relative to ego in the ego frame, float64 like the reference's numpy
                        const double ci = (double)L.cs[i], si = (double)L.sn[i];
                        const double dx = (double)L.x[j] - (double)L.x[i], dy = (double)L.y[j] - (double)L.y[i];
                        const double lon = dx * ci + dy * si, lat = dy * ci - dx * si;
                        const double dis = sqrt(lon * lon + lat * lat);
                        const double v[3] = {dis / 20.0, (lon / dis + 1.0) / 2.0, (lat / dis + 1.0) / 2.0};   // 0/0 = NaN for d == 0, as numpy
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            ex[k] = v[k] != v[k] ? __uint_as_float(0x7fc00000u) : (float)(v[k] < 0.0 ? 0.0 : (v[k] > 1.0 ? 1.0 : v[k]));
                    }
                    qm[CS] = ex[0]; qm[CS + 1] = ex[1]; qm[CS + 2] = ex[2];
                }
            }
        }
        __syncthreads();       // every read of the fp64 distances is done: their storage now takes the rewards in list order
        // ---- 3. rewards in list order, per-agent tails ----------------------------------------------------------------
        if (out.nei_rew)
            for (int q = tid; q < T; q += nthreads) {
                const unsigned int pk = ent[q];
                const int la = (int)(pk >> 8), pos = (int)(pk & 255u);
                srt[la * N + nb_rk[la * N + pos]] = L.rew[nb_j[la * N + pos]];
            }
        __syncthreads();
        if (wave == tail_wave) {       // one lane per list of this pass
            if (lane < cha) {
                const int i = L.plist[ia0 + lane];
                const int cnt = (int)(ncnt[i] & 0xffffu) - (int)(ncnt[i] >> 16);       // candidates - those outside
                if (out.nbr_cnt) out.nbr_cnt[base + i] = cnt;
                if (out.mf_cnt) out.mf_cnt[base + i] = (int)mfcn[i];
                if (out.nei_rew) {
                    double nsum = 0.0;
                    for (int r = 0; r < cnt; ++r) nsum += (double)srt[lane * N + r];
                    out.nei_rew[base + i] = cnt ? (float)(nsum / (double)cnt) : 0.0f;
                }
                for (int k = cnt; k < K; ++k) {
                    if (out.nbr_idx) out.nbr_idx[(base + i) * K + k] = -1;
                    if (out.nbr_dist) out.nbr_dist[(base + i) * K + k] = 0.0f;
                }
                if (comm)
                    for (int r = cnt; r < p.comm_nb; ++r) {
                        float* qm = out.obs + (base + i) * p.O + p.col_comm + r * CD;
                        for (int k = 0; k < CD; ++k) qm[k] = 0.0f;
                    }
            }
            if (lane == 0) L.n_entries = 0;
        }
        if (ia0 + CH < np) __syncthreads();     // the next pass reuses the list storage (and the entry counter)
    }
}



// Neighbour lists + reward reductions of one scene by ONE wave, lane = slot: the same results as neighbours_phase, bit for
// bit.  Returns 0 when the scene is done -- `*n_exact` agents of it through neighbours_exact_one -- or 2 (nothing written)
// when more than NBR_EXACT_MAX agents would need that: the caller then runs neighbours_phase.
//
// Every lane walks the present agents j in slot order (the record {x, y, (double) reward} of j is one broadcast LDS read)
// and keeps, in registers: the count and the fp64 reward sum of the agents within `neighbours_distance`, and the 9 smallest
// keys `fp32 d^2 bits (low 6 mantissa bits dropped) | j` in ascending order (one v_med3_u32 per position).  A lane's
// registers ARE the reference's result unless one of these holds, and then the agent is evaluated by neighbours_exact_one:
//  * in range / mean-field range: decided on the fp32 d^2 against thresholds 1e-6 (1e-5 for the truncated keys) inside and
//    outside the exact ones; a distance inside such a band is not decided (fp32 d^2 errs by < 2.4e-7 relative).  Nine or
//    more agents inside the mean-field range cannot be counted on nine keys.
//  * order by (d, slot): adjacent keys of the nine must differ by >= 2 units of the truncated d^2 (the fp32 values then
//    differ by > 64 ulp >> their error: the exact fp64 distances are ordered the same way and no tie exists).  Exact ties
//    (the crafted cases of the tests, symmetric scenes) therefore always go to the exact evaluation.
//  * neighbourhood reward: the reference adds the fp64 rewards in list order.  If every reward in range is 0 or has
//    2^-20 <= |r| <= 16, every partial sum of <= 64 of them is a multiple of 2^-43 below 2^10, i.e. exactly representable:
//    no addition rounds and the order does not matter.  An agent with another kind of reward in range is not decided.
//  * global reward: added in slot order, which IS the reference's order.
// K > 8, the communication block and the stateless op's filled rows take neighbours_phase as well.
constexpr int NBR_EXACT_MAX = 6;
// `reach_lo / reach_hi` (optional): the slots of the SOLID vehicles within LiDAR reach of this lane's slot, from the same walk (one more
// compare on the d^2 it has anyway, the wrecks -- solid, not present -- in a short walk of their own); a superset test (1e-5 wider than
// the LiDAR phase's own, which still decides every pair it is handed).
__device__ __forceinline__ int neighbours_fast(const SimParams& p, EnvLds& L, int e, int lane, const StepOut& out, float4* rec,
                                               int* n_exact, unsigned int* reach_lo = nullptr, unsigned int* reach_hi = nullptr) {
    const int N = p.N, K = p.K;
    const unsigned long long present = L.m_present;
    const bool me = (present >> lane) & 1ull;
    const int np = __popcll(present);
    const float xi = L.x[lane], yi = L.y[lane], rw = L.rew[lane];
    const float arw = fabsf(rw);
    const unsigned long long odd = __ballot(me && !(rw == 0.0f || (arw >= 9.5367431640625e-07f && arw <= 16.0f)));
    {   // rec: [64] {x, y, (double) reward} in LDS (the caller's choice: the neighbour work area when one wave owns the scene)
        const double rd = (double)rw;
        rec[lane] = make_float4(xi, yi, __int_as_float(__double2loint(rd)), __int_as_float(__double2hiint(rd)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float r2lo = p.nbr_r2lo, r2hi = p.nbr_r2hi;
    uint32_t a0 = NBR_SENT, a1 = NBR_SENT, a2 = NBR_SENT, a3 = NBR_SENT, a4 = NBR_SENT, a5 = NBR_SENT, a6 = NBR_SENT,
             a7 = NBR_SENT, a8 = NBR_SENT;
    double sum = 0.0, gs = 0.0;
    int cnt = 0;
    bool unc = false;
    const bool want_reach = reach_lo != nullptr;
    const unsigned long long solid = L.m_solid;
    const float lim_r = p.lidar_range + sqrtf(p.hl * p.hl + p.hw * p.hw);
    const float lim2 = lim_r * lim_r * 1.00001f;
    unsigned int rlo = 0u, rhi = 0u;
    unsigned long long m = present;
    int j = m ? __ffsll((long long)m) - 1 : 0;              // wave-uniform
    float4 r = rec[j];
    while (m) {
        m &= m - 1ull;
        const int jn = m ? __ffsll((long long)m) - 1 : j;
        const float4 rn = rec[jn];
        const double rj = __hiloint2double(__float_as_int(r.w), __float_as_int(r.z));
        const float dx = xi - r.x, dy = yi - r.y;
        const float d2 = __builtin_fmaf(dy, dy, dx * dx);
        const bool other = j != lane;
        const bool in = other && d2 < r2lo, inhi = other && d2 < r2hi;
        unc |= in != inhi;
        sum = fma(in ? 1.0 : 0.0, rj, sum);      // (sum starts at +0.0 and rj is finite: the same bits as a guarded add)
        cnt += in ? 1 : 0;
        if (want_reach && ((solid >> j) & 1ull)) {
            const unsigned int rb = (other && d2 <= lim2) ? 1u : 0u;
            if (j < 32) rlo |= rb << j;
            else rhi |= rb << (j - 32);
        }
        const uint32_t key = in ? ((__float_as_uint(d2) & ~63u) | (uint32_t)j) : NBR_SENT;
        a8 = umed3(a7, a8, key); a7 = umed3(a6, a7, key); a6 = umed3(a5, a6, key); a5 = umed3(a4, a5, key);
        a4 = umed3(a3, a4, key); a3 = umed3(a2, a3, key); a2 = umed3(a1, a2, key); a1 = umed3(a0, a1, key);
        a0 = a0 < key ? a0 : key;
        j = jn;
        r = rn;
    }
    if (want_reach) {
        for (unsigned long long mw = solid & ~present; mw; mw &= mw - 1ull) {      // wrecks: solid, no agent
            const int jw = __ffsll((long long)mw) - 1;
            const float4 rw4 = rec[jw];
            const float dx = xi - rw4.x, dy = yi - rw4.y;
            const unsigned int rb = (jw != lane && __builtin_fmaf(dy, dy, dx * dx) <= lim2) ? 1u : 0u;
            if (jw < 32) rlo |= rb << jw;
            else rhi |= rb << (jw - 32);
        }
        *reach_lo = me ? rlo : 0u;
        *reach_hi = me ? rhi : 0u;
    }
    // global reward (LCFEnv.step: sum(r.values()) / len(r.values()), fp64 in slot order): with every reward in the range where
    // sums are exact in any order a DPP tree gives the slot order's bits; else the serial sum
    if (out.glob_rew) {
        if (odd == 0ull) {
            gs = pk_wave_sum_f64(me ? (double)rw : 0.0);
            gs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(gs), 63), __builtin_amdgcn_readlane(__double2loint(gs), 63));
            gs = 0.0 + gs;
        } else {
            for (unsigned long long mg = present; mg; mg &= mg - 1ull) gs += (double)readlane_f(rw, __ffsll((long long)mg) - 1);
        }
    }
    // mean-field count from the keys (the 9 nearest), order check of adjacent keys
    const uint32_t ak[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8};
    const uint32_t tlo = p.mf_key_lo, thi = p.mf_key_hi;
    int mf = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mf += ak[k] < tlo ? 1 : 0;
        unc |= ak[k] >= tlo && ak[k] < thi;
    }
    unc |= a8 < thi;                                        // nine or more within the mean-field range: not countable here
#pragma unroll
    for (int k = 0; k < 8; ++k) unc |= (ak[k + 1] != NBR_SENT) && (ak[k + 1] - ak[k] < 128u);
    for (unsigned long long mo = odd; mo; mo &= mo - 1ull) {          // (rare) a reward outside the exact-sum range: who has it in range?
        const int b = __ffsll((long long)mo) - 1;
        const float dx = xi - readlane_f(xi, b), dy = yi - readlane_f(yi, b);
        unc |= b != lane && __builtin_fmaf(dy, dy, dx * dx) < r2hi;
    }
    const unsigned long long exact = __ballot(me && unc);
    *n_exact = __popcll(exact);
    if (*n_exact > NBR_EXACT_MAX) return 2;
    // ---- write-out ------------------------------------------------------------------------------------------------------
    const size_t base = (size_t)e * N;
    const bool mine = me && !unc;                           // this lane's registers are the result
    if (lane == 0 && out.glob_rew) out.glob_rew[e] = np ? (float)(gs / (double)np) : 0.0f;
    if (lane < N && !(me && unc)) {
        if (out.nbr_cnt) out.nbr_cnt[base + lane] = me ? cnt : 0;
        if (out.mf_cnt) out.mf_cnt[base + lane] = me ? mf : 0;
        if (out.nei_rew) out.nei_rew[base + lane] = (me && cnt) ? (float)(sum / (double)cnt) : 0.0f;
    }
    if (mine) {
        if (out.nbr_idx) {
            int32_t* row = out.nbr_idx + (base + lane) * K;
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
        if (out.nbr_dist) {                                 // the reference's fp64 distance, for the <= K entries that are written
            float* row = out.nbr_dist + (base + lane) * K;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < K) {
                    float dv = 0.0f;
                    if (k < cnt) {
                        const int j = (int)(ak[k] & 63u);
                        const double dx = (double)xi - (double)L.x[j], dy = (double)yi - (double)L.y[j];
                        dv = (float)sqrt(dx * dx + dy * dy);
                    }
                    row[k] = dv;
                }
        }
    }
    for (unsigned long long mx = exact; mx; mx &= mx - 1ull)        // (rare) the agents that the registers do not decide
        neighbours_exact_one(p, e, lane, __ffsll((long long)mx) - 1, xi, yi, rw, present, odd, out);
    return 0;
}

// The register formulation for the step kernel's wave roles (several waves per scene), split over TWO waves.  Two things are
// different from neighbours_fast, where 26 scenes per compute unit hide every latency: (i) the record of agent j comes out of
// lane j's registers (v_readlane), not out of LDS -- next to 13 waves of LiDAR atomics and permutes an LDS round trip takes
// ~300 cycles, and a walk step waited one out whatever it computed (neighbour wave done 9.2k -> 6.4k cycles after P3, 256-scene
// launch 17.3 -> 16.1 us); (ii) the walk is the longest single-wave stretch of such a launch, so two waves share it:
//   PART 1: range decisions, counts, fp64 reward sums, the global reward    -> nbr_cnt, nei_rew, glob_rew
//   PART 2: the nine smallest keys                                           -> mf_cnt, nbr_idx, nbr_dist
// Both walk all present agents and write the rows their registers prove; each hands back the agents it cannot decide, the caller
// evaluates the union exactly (neighbours_exact_one rewrites every field of such an agent) or falls back for the scene.
template <int PART>
__device__ __forceinline__ unsigned long long neighbours_roles(const SimParams& p, EnvLds& L, int e, int lane, const StepOut& out,
                                                               unsigned long long* odd_out) {
    const int N = p.N, K = p.K;
    const unsigned long long present = L.m_present;
    const bool me = (present >> lane) & 1ull;
    const int np = __popcll(present);
    const float xi = L.x[lane], yi = L.y[lane], rw = L.rew[lane];
    const float arw = fabsf(rw);
    const unsigned long long odd = __ballot(me && !(rw == 0.0f || (arw >= 9.5367431640625e-07f && arw <= 16.0f)));
    float r2lo = p.nbr_r2lo, r2hi = p.nbr_r2hi;
    uint32_t a0 = NBR_SENT, a1 = NBR_SENT, a2 = NBR_SENT, a3 = NBR_SENT, a4 = NBR_SENT, a5 = NBR_SENT, a6 = NBR_SENT,
             a7 = NBR_SENT, a8 = NBR_SENT;
    double sum = 0.0, gs = 0.0;
    int cnt = 0;
    bool unc = false;
    // the record of agent j comes out of lane j's registers (v_readlane into scalar operands), not out of LDS: next to 13 waves
    // of LiDAR atomics and permutes an LDS round trip takes ~300 cycles, and a walk step waited one out whatever it computed
    for (unsigned long long m = present; m; m &= m - 1ull) {
        const int j = __ffsll((long long)m) - 1;
        const float xj = readlane_f(xi, j), yj = readlane_f(yi, j);
        const float dx = xi - xj, dy = yi - yj;
        const float d2 = __builtin_fmaf(dy, dy, dx * dx);
        const bool other = j != lane;
        const bool in = other && d2 < r2lo;
        if (PART == 1) {
            const double rj = (double)readlane_f(rw, j);
            gs += rj;
            const bool inhi = other && d2 < r2hi;
            unc |= in != inhi;
            sum = fma(in ? 1.0 : 0.0, rj, sum);
            cnt += in ? 1 : 0;
        } else {
            const uint32_t key = in ? ((__float_as_uint(d2) & ~63u) | (uint32_t)j) : NBR_SENT;
            a8 = umed3(a7, a8, key); a7 = umed3(a6, a7, key); a6 = umed3(a5, a6, key); a5 = umed3(a4, a5, key);
            a4 = umed3(a3, a4, key); a3 = umed3(a2, a3, key); a2 = umed3(a1, a2, key); a1 = umed3(a0, a1, key);
            a0 = a0 < key ? a0 : key;
        }
    }
    const size_t base = (size_t)e * N;
    if (PART == 1) {
        for (unsigned long long mo = odd; mo; mo &= mo - 1ull) {          // (rare) a reward outside the exact-sum range: who has it in range?
            const int b = __ffsll((long long)mo) - 1;
            const float dx = xi - readlane_f(xi, b), dy = yi - readlane_f(yi, b);
            unc |= b != lane && __builtin_fmaf(dy, dy, dx * dx) < r2hi;
        }
        if (lane == 0 && out.glob_rew) out.glob_rew[e] = np ? (float)(gs / (double)np) : 0.0f;
        if (lane < N && !(me && unc)) {
            if (out.nbr_cnt) out.nbr_cnt[base + lane] = me ? cnt : 0;
            if (out.nei_rew) out.nei_rew[base + lane] = (me && cnt) ? (float)(sum / (double)cnt) : 0.0f;
        }
        *odd_out = odd;
        return __ballot(me && unc);
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
    if (lane < N && !(me && unc) && out.mf_cnt) out.mf_cnt[base + lane] = me ? mf : 0;
    if (me && !unc) {
        if (out.nbr_idx) {
            int32_t* row = out.nbr_idx + (base + lane) * K;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < K) row[k] = ak[k] != NBR_SENT ? (int)(ak[k] & 63u) : -1;
        }
        if (out.nbr_dist) {
            float* row = out.nbr_dist + (base + lane) * K;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < K) {
                    float dv = 0.0f;
                    if (ak[k] != NBR_SENT) {
                        const int j = (int)(ak[k] & 63u);
                        const double dx = (double)xi - (double)L.x[j], dy = (double)yi - (double)L.y[j];
                        dv = (float)sqrt(dx * dx + dy * dy);
                    }
                    row[k] = dv;
                }
        }
    }
    return __ballot(me && unc);
}

template <bool EXT>
__device__ __forceinline__ void neighbours_any(const SimParams& p, EnvLds& L, int e, int tid, int nthreads, const StepOut& out,
                                               const float* __restrict__ act = nullptr, unsigned long long acted_mask = 0ull,
                                               bool fresh = true, unsigned int* reach_lo = nullptr, unsigned int* reach_hi = nullptr,
                                               bool* have_reach = nullptr) {
    // one wave owns the scene: the register formulation above, unless it declines (ties, band cases, odd rewards, K > 8, comm)
    const bool comm = EXT && p.col_comm >= 0 && out.obs != nullptr;
    if (nthreads == 64 && p.nbr_fast && !comm && !(COPO_PROFILE_SKIP & 128)) {
        extern __shared__ unsigned int dyn[];
        int n_exact = 0;
        const int why = neighbours_fast(p, L, e, tid, out, reinterpret_cast<float4*>(dyn), &n_exact, reach_lo, reach_hi);
        if (have_reach) *have_reach = reach_lo != nullptr;      // (the masks are complete whether or not the lists were declined)
        // profiling aid, which formulation ran: 1 register, 16 + n register with n agents evaluated exactly, 2 declined (pair-parallel)
        if (p.dbg && tid == 0) p.dbg[(size_t)e * COPO_DBG_STRIDE + 7] = why ? 2 : (n_exact ? 16 + n_exact : 1);
        if (why == 0) return;
    }
    neighbours_phase<EXT>(p, L, e, tid, nthreads, out, act, acted_mask, fresh);
}



// LiDAR + observation write-out, all threads of the workgroup.  Precondition: L.x/y/cs/sn, m_present, m_solid and the
// slot lists (build_lists) are final and visible (caller synchronised).
//
// Pair-driven: a ray of agent i can only touch vehicle j inside the angular window bearing(j) +- asin(circumradius /
// distance) of i's ray fan, so the work list is (present agent, solid vehicle) pairs expanded to the few rays of their
// window -- ~6 box tests per pair instead of one reject test per (ray, vehicle).  One lane per pair computes the
// window (a superset, with a 4 mrad margin over < 1e-4 rad of approximation error); a wave scan of the window sizes
// numbers the box tests; the pair that owns box test t is found with head flags (every pair marks the slot of its
// first test in a 64-entry LDS strip of the wave, a max-scan spreads the marks), and hits fold into
// the per-ray minimum with LDS atomicMin on the float bit pattern (distances are >= 0, so unsigned order == float
// order and the result does not depend on the task order).  Rays outside every window keep `range`, exactly what
// the exhaustive test of the oracle gives them.
// `phases`: 1 = initialise the ray minima, 2 = window / box-test work, 4 = write-out (+ detector beams); all of them (7) is the
// whole phase.  The step kernel of the many-waves-per-scene shape runs them apart (wave roles): 2 on waves >= `wave_lo` only
// while waves 0 / 1 / 2 write the state back and build the neighbour lists; 4 starts with a workgroup barrier (waves 1 .. idle_n
// only pass it).
template <int PHASES = 7>
__device__ __forceinline__ void obs_phase(const SimParams& p, EnvLds& L, int e, int tid, int nthreads,
                                          float* __restrict__ obs, int wave_lo = 0, int idle_n = 0) {
    extern __shared__ unsigned int dyn[];
    const int N = p.N, O = p.O, NL = p.num_lasers;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = nthreads >> 6;
    // write-out (4): waves 1 .. idle_n take no part (they pass the barriers only; the step kernel gives them other work)
    const int otid = (wave >= 1 && wave <= idle_n) ? 0x3fffffff : tid - (wave > idle_n ? idle_n * 64 : 0);
    const int onth = nthreads - idle_n * 64;
    const float hl = p.hl, hw = p.hw;
    const float circ = sqrtf(hl * hl + hw * hw);
    const float range = p.lidar_range;
    const float lim = range + circ;
    const unsigned long long solid = L.m_solid, present = L.m_present;
    const int np = __popcll(present), ns = __popcll(solid);
    unsigned int* best = dyn;                     // [chunk][NL] nearest entry distance (float bits) per ray
    const unsigned int range_bits = __float_as_uint(range);
    float* eobs = obs + (size_t)e * N * O;
    const float* __restrict__ rays = lds_rays(p);
    int* wtag = reinterpret_cast<int*>(dyn + lidar_lds_words(p.chunk, p.nbr_chunk, p.N, p.num_lasers) + ray_lds_words(p.num_lasers)) + wave * LIDAR_WAVE_WORDS;   // head flags
    const int CH = p.chunk > 0 ? p.chunk : N;     // present agents whose ray fans are in LDS at a time
    const float inv_ns = 1.0f / (float)(ns > 0 ? ns : 1);
    const float rays_per_rad = (float)NL * 0.159154943f;
    const float inv_nl = 1.0f / (float)NL;
    const float inv_range = p.inv_range;
    const int col_lidar = p.col_lidar;
    const int head = (4 - (col_lidar & 3)) & 3;   // rays in front of the first 16-byte aligned column of a row
    const int nvec = (NL - head) >> 2;
    const bool vec_out = ((O & 3) == 0) && nvec > 0 && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
    const float inv_nvec = 1.0f / (float)(nvec > 0 ? nvec : 1), inv_nsc = 1.0f / (float)(NL - 4 * nvec > 0 ? NL - 4 * nvec : 1);
    for (int ip0 = 0; ip0 < np; ip0 += CH) {
    const int cha = np - ip0 < CH ? np - ip0 : CH;
    if (PHASES & 1) {
        for (int q = tid; q < cha * NL; q += nthreads) best[q] = range_bits;
        __syncthreads();
    }
    const int ncombo = cha * ns;
    // everything per batch of 64 (fan, vehicle) pairs: window, numbering of the box tests, the tests themselves
    auto pair_batch = [&](const bool live, const int lp, const int i, const int j) {
        const float ci = L.cs[i], si = L.sn[i], cj = L.cs[j], sj = L.sn[j];      // (all eight pose reads in one LDS round trip)
        const float dx = L.x[j] - L.x[i], dy = L.y[j] - L.y[i];
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
                // tighter for a box seen from outside its own length: every point of it lies at least d - h_par along the
                // line of sight and at most h_perp beside it (extents of the box along / across that line), so the
                // half-angle is below h_perp / (d - h_par) -- a vehicle seen end-on is 0.93 m wide, not 2.44
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
        const int incl = wave_scan_incl<false>(cnt);      // inclusive scan of the window sizes
        const int total = __builtin_amdgcn_readlane(incl, 63);
        COPO_COUNT(9, 1); COPO_COUNT(10, total); COPO_COUNT(11, (total + 63) >> 6);
        const int excl = incl - cnt;                      // index of this pair's first box test
        // the pair's record for its box tests (registers of this lane, fetched by the test lanes with ds_bpermute -- no LDS
        // storage: a strip of records per wave cost more in resident scenes than it saved in instructions): ray origin in
        // j's box frame, rotation from i's frame into it, first ray / first test (klo - excl: 24 bits signed, local fan: 6)
        const float rec_ox = -fm(dx, cj, dy * sj), rec_oy = -fm(dy, cj, -(dx * sj));
        const float rec_cr = fm(ci, cj, si * sj), rec_sr = fm(ci, sj, -(si * cj));
        const int rec_ix = ((klo - excl) & 0xffffff) | (lp << 24);      // ray of test t = (klo - excl + t) mod NL
        int carry = 0;                                    // (lane + 1) of the pair that owns the last test of the previous batch
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            // head flags: the pair whose first test falls into this batch marks that slot with lane + 1; a max-scan
            // spreads the marks over the tests that follow (marks grow with the position)
            wtag[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cnt > 0 && excl >= t0 && excl < t0 + 64) wtag[excl - t0] = lane + 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int own = wave_scan_incl<true>(wtag[lane]);
            own = own > carry ? own : carry;
            carry = __builtin_amdgcn_readlane(own, 63);
            __builtin_amdgcn_wave_barrier();
            const int sl = own > 0 ? own - 1 : 0;
            const float ox = __shfl(rec_ox, sl), oy = __shfl(rec_oy, sl), cr = __shfl(rec_cr, sl), sr = __shfl(rec_sr, sl);
            const int pw = __shfl(rec_ix, sl);
            if (t < total) {
                const int slp = pw >> 24;
                int k = ((pw << 8) >> 8) + t;              // (klo - excl) is a signed 24-bit field
                if (k >= NL) k -= NL;
                const float2 r = reinterpret_cast<const float2*>(rays)[k];
                const float tt = ray_box(ox, oy, fm(r.x, cr, r.y * sr), fm(r.y, cr, -(r.x * sr)), hl, hw);
                if (tt >= 0.0f) atomicMin(&best[slp * NL + k], __float_as_uint(tt));
                if (COPO_PROFILE_SKIP & 256) { const int nh = __popcll(__ballot(tt >= 0.0f)); COPO_COUNT(12, nh); }
            }
        }
    };
    if (!(PHASES & 2)) {
    } else if (nwaves == 1 && lidar_queue_words(p.chunk, N) > 0 && !(COPO_PROFILE_SKIP & 2)) {
        // one wave owns the scene: a cheap pass keeps the pairs within LiDAR reach (about half of them at a junction) in a
        // queue, the window / test pass then runs on full batches of those -- the window arithmetic executes for a whole
        // batch as soon as one of its pairs is in reach
        uint16_t* cq = reinterpret_cast<uint16_t*>(dyn + cha * NL);
        int nq = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int c0 = 0; c0 < ncombo; c0 += 64) {
            const int c = c0 + lane;
            const bool live = c < ncombo;
            const int lp = live ? (int)(((float)c + 0.5f) * inv_ns) : 0;
            const int i = L.plist[ip0 + lp], j = L.slist[live ? c - lp * ns : 0];
            const float dx = L.x[j] - L.x[i], dy = L.y[j] - L.y[i];
            const bool reach = live && j != i && !(fm(dx, dx, dy * dy) > lim * lim);
            const unsigned long long m = __ballot(reach);
            if (reach) cq[nq + __popcll(m & lt)] = (uint16_t)c;
            nq += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        COPO_COUNT(8, nq);
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const bool live = q0 + lane < nq;
            const int c = live ? (int)cq[q0 + lane] : 0;
            const int lp = (int)(((float)c + 0.5f) * inv_ns);
            pair_batch(live, lp, L.plist[ip0 + lp], L.slist[c - lp * ns]);
        }
    } else if (wave_lo > 0) {
        // wave roles: a wave takes its (fan, vehicle) pairs two batches of 64 at a time, keeps the ones within LiDAR reach (about a
        // quarter at a junction) and, when they fit one batch, pushes them together across the lanes (ds_permute) -- the window
        // arithmetic and its LDS round trips then run once instead of twice
        const int nact = nwaves - wave_lo;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int c0 = (wave - wave_lo) * 128; c0 < ((COPO_PROFILE_SKIP & 2) || wave < wave_lo ? 0 : ncombo); c0 += nact * 128) {
            int cc[2], ii[2], jj[2];
            bool rr[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + 64 * h + lane;
                const bool valid = c < ncombo;
                const int lp = valid ? (int)(((float)c + 0.5f) * inv_ns) : 0;
                cc[h] = c;
                ii[h] = L.plist[ip0 + lp];
                jj[h] = L.slist[valid ? c - lp * ns : 0];
                rr[h] = valid;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float dx = L.x[jj[h]] - L.x[ii[h]], dy = L.y[jj[h]] - L.y[ii[h]];
                rr[h] = rr[h] && jj[h] != ii[h] && !(fm(dx, dx, dy * dy) > lim * lim);
            }
            const unsigned long long m0 = __ballot(rr[0]), m1 = __ballot(rr[1]);
            const int n0 = __popcll(m0), n1 = __popcll(m1);
            if (n0 + n1 <= 63) {            // one batch: lane r takes the r-th pair in reach (lanes out of reach push to lane 63, which is not used)
                const int p0 = __builtin_amdgcn_ds_permute((rr[0] ? __popcll(m0 & lt) : 63) << 2, cc[0]);
                const int p1 = __builtin_amdgcn_ds_permute((rr[1] ? n0 + __popcll(m1 & lt) : 63) << 2, cc[1]);
                const bool live = lane < n0 + n1;
                const int c = live ? (lane < n0 ? p0 : p1) : 0;
                const int lp = (int)(((float)c + 0.5f) * inv_ns);
                if (n0 + n1 > 0) pair_batch(live, lp, L.plist[ip0 + lp], L.slist[c - lp * ns]);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int lp = (int)(((float)(rr[h] ? cc[h] : 0) + 0.5f) * inv_ns);
                    if (h == 0 ? n0 > 0 : n1 > 0) pair_batch(rr[h], lp, ii[h], jj[h]);
                }
            }
        }
    } else {
        for (int c0 = (wave - wave_lo) * 64; c0 < ((COPO_PROFILE_SKIP & 2) || wave < wave_lo ? 0 : ncombo); c0 += (nwaves - wave_lo) * 64) {
            const int c = c0 + lane;
            const bool live = c < ncombo;
            const int lp = live ? (int)(((float)c + 0.5f) * inv_ns) : 0;      // fan of this pass
            pair_batch(live, lp, L.plist[ip0 + lp], L.slist[live ? c - lp * ns : 0]);
        }
    }
    if (!(PHASES & 4)) continue;
    __syncthreads();
    if (p.n_boxes_lidar > 0) {      // static boxes (buildings): every (fan, ray) of the pass against the boxes within reach of the fan, one thread each
        for (int q = otid; q < cha * NL; q += onth) {
            const int lp = (int)(((float)q + 0.5f) * inv_nl), k = q - lp * NL;
            const int i = L.plist[ip0 + lp];
            const float xi = L.x[i], yi = L.y[i], ci = L.cs[i], si = L.sn[i];
            const float2 r = reinterpret_cast<const float2*>(rays)[k];
            unsigned int m = best[q];
            for (int b = 0; b < p.n_boxes; ++b) {
                const float* B = p.boxes + b * COPO_BOX_STRIDE;
                const float rx = B[0] - xi, ry = B[1] - yi, lb = range + (B[4] + B[5]);
                if (!(fm(rx, rx, ry * ry) <= lb * lb)) continue;
                const float ox = -fm(rx, B[2], ry * B[3]), oy = -fm(ry, B[2], -(rx * B[3]));
                const float cr = fm(ci, B[2], si * B[3]), sr = fm(ci, B[3], -(si * B[2]));
                bool hit;
                const float tt = ray_box_nr(ox, oy, fm(r.x, cr, r.y * sr), fm(r.y, cr, -(r.x * sr)), B[4], B[5], hit);
                if (hit && __float_as_uint(tt) < m) m = __float_as_uint(tt);
            }
            best[q] = m;
        }
        __syncthreads();
    }
    const int nrays = cha * NL;                   // rows of present slots only
    if (vec_out && !(COPO_PROFILE_SKIP & 4)) {
        // 16-byte stores: [head scalars | nvec aligned quads | tail scalars] of every fan (rows are 16-byte aligned: O % 4 == 0)
        for (int q = otid; q < cha * nvec; q += onth) {
            const int lp = (int)(((float)q + 0.5f) * inv_nvec), k = head + 4 * (q - lp * nvec);
            const unsigned int* b = best + lp * NL + k;
            float4 v;
            v.x = __uint_as_float(b[0]) * inv_range; v.y = __uint_as_float(b[1]) * inv_range;
            v.z = __uint_as_float(b[2]) * inv_range; v.w = __uint_as_float(b[3]) * inv_range;
            *reinterpret_cast<float4*>(eobs + (int)L.plist[ip0 + lp] * O + col_lidar + k) = v;
        }
        const int nsc = NL - 4 * nvec;            // head + tail scalars per fan
        for (int q = otid; q < cha * nsc; q += onth) {
            const int lp = (int)(((float)q + 0.5f) * inv_nsc), r = q - lp * nsc;
            const int k = r < head ? r : r + 4 * nvec;
            eobs[(int)L.plist[ip0 + lp] * O + col_lidar + k] = __uint_as_float(best[lp * NL + k]) * inv_range;
        }
    } else
    for (int q = otid; q < ((COPO_PROFILE_SKIP & 4) ? 0 : nrays); q += onth) {
        const int lp = (int)(((float)q + 0.5f) * inv_nl), k = q - lp * NL;
        eobs[(int)L.plist[ip0 + lp] * O + col_lidar + k] = __uint_as_float(best[q]) * inv_range;
    }
    if (ip0 + CH < np) __syncthreads();           // the next pass reuses the minima
    }
    // optional side / lane-line detector beams (Bottleneck, Tollgate): the primitives near each agent are marked first, the beams walk only
    // those (detector_beams, sim_device.h; its marks take the storage of the ray minima, which are written out).  Wave roles: waves
    // 1 .. idle_n take no part in the write-out (they evaluate neighbour lists afterwards) but meet the barriers of the phase.
    if (PHASES & 4)
        detector_beams<true>(p, [&L](int v) { return make_float4(L.x[v], L.y[v], L.cs[v], L.sn[v]); }, L.plist, np, best, CH * NL, eobs, otid, onth, wtag);
}

__device__ __forceinline__ void stage_pose(EnvLds& L, int lane, const Slot& s) {
    L.x[lane] = s.x;
    L.y[lane] = s.y;
    L.cs[lane] = s.hc;
    L.sn[lane] = s.hs;
}

__device__ __forceinline__ void load_rays(const SimParams& p, EnvLds& L, int tid, int nthreads) {
    float* rays = lds_rays(p);
    for (int q = tid; q < p.num_lasers * 2; q += nthreads) rays[q] = p.ray_cs[q];
}

// ------------------------------------------------------------------------------------------------
// reset kernel
// ------------------------------------------------------------------------------------------------
template <bool EXT>   // EXT: the traffic-light / communication observation blocks are compiled in (copo_sim_cfg extensions)
__global__ void __launch_bounds__(COPO_SIM_MAX_BLOCK) sim_reset_kernel(const SimParams* __restrict__ pp, StepOut out) {
    // the parameter block lives in device memory: ~90 scalars as by-value kernel arguments are all loaded up front and
    // kept live, and the resulting SGPR spills (v_writelane / v_readlane) were a quarter of the kernel's VALU instructions
    const SimParams& p = *pp;
    __shared__ EnvLds L;
    const int e = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = p.N;
    load_rays(p, L, tid, nthreads);
    if (tid == 0) {
        L.seg_rows = p.seg_rows;
        L.rsegs = p.route_segs;
        L.rmeta = p.route_meta;
        L.stab = p.spawn_tab;
        L.sps = p.spawn_s;
    }
    __syncthreads();
    Slot s;
    if (wave == 0) {
        const uint64_t seed = p.seeds[e];
        s = Slot{};
        reset_env_wave0(p, L, seed, 0u, lane, s);
        const int cap = capacity_of(p);
        if (lane < N) {
            stage_pose(L, lane, s);
            store_slot(p, e, lane, s);
            L.rew[lane] = 0.0f;
            const size_t o = (size_t)e * N + lane;
            if (out.rew) out.rew[o] = 0.0f;
            if (out.flags) out.flags[o] = lane < cap ? COPO_F_SPAWNED : 0;
            if (out.lcf) out.lcf[o] = s.lcf;
            if (out.agent_id) out.agent_id[o] = s.aid;
            if (out.info)
                for (int k = 0; k < COPO_INFO_DIM; ++k) out.info[o * COPO_INFO_DIM + k] = 0.0f;
        } else {
            L.x[lane] = 0.0f; L.y[lane] = 0.0f; L.cs[lane] = 1.0f; L.sn[lane] = 0.0f; L.rew[lane] = 0.0f;
        }
        const unsigned long long all = __ballot(lane < cap);
        build_lists(L, lane, all, all);
        if (lane == 0) {
            L.m_present = all;
            L.m_solid = all;
            L.m_acted = 0ull;
            int32_t* env = p.env + (size_t)e * 4;
            env[0] = 0; env[1] = 0; env[2] = cap; env[3] = 1;
        }
        ego_navi_obs<EXT>(p, L, L.cs[lane], L.sn[lane], s, lane < cap, (out.obs && lane < N) ? out.obs + ((size_t)e * N + lane) * p.O : nullptr, 0, true);
    }
    __syncthreads();
    neighbours_any<EXT>(p, L, e, tid, nthreads, out);
    __syncthreads();   // the list-order sums read the LDS words that the LiDAR minima reuse
    if (out.obs) obs_phase(p, L, e, tid, nthreads, out.obs);
}

// ------------------------------------------------------------------------------------------------
// step kernel
// ------------------------------------------------------------------------------------------------
// ONE: the one-wave-per-scene launch shape as its own instantiation (workgroup = 64 threads known at compile time: the
// wave-role branches fold away, workgroup barriers become wave-local, the register budget is that of a 64-thread kernel)
template <bool EXT, bool ONE>
__global__ void __launch_bounds__(ONE ? 64 : COPO_SIM_MAX_BLOCK, (ONE && !EXT) ? 7 : 1) sim_step_kernel(const SimParams* __restrict__ pp,
                                                                                 const float* __restrict__ act, StepOut out) {
    const SimParams& p = *pp;       // device-memory parameter block (see sim_reset_kernel)
    __shared__ EnvLds L;
    const int e = blockIdx.x, tid = threadIdx.x, nthreads = ONE ? 64 : (int)blockDim.x;
    const int wave = ONE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = ONE ? 1 : nthreads >> 6;
    const int N = p.N;
    const float hl = p.hl, hw = p.hw;
    // several waves per scene: wave 0 goes straight to P0 (its state loads are the head of the launch's longest dependent chain:
    // 18.4 -> 17.4 us per launch of 256 scenes); the ray table and the route tables are copied by the other waves meanwhile --
    // the barrier after P0 publishes them.  (Measured and dropped: the state / clock pointers as kernel arguments of their own,
    // so that P0's loads need not wait for the parameter block -- 0.3 us slower.)
    const bool copy_apart = !ONE && nwaves > 1;
    const int ctid = copy_apart ? tid - 64 : tid, cnth = copy_apart ? nthreads - 64 : nthreads;
    if (ctid >= 0) load_rays(p, L, ctid, cnth);
    float4* rec_roles = nullptr;
    unsigned long long* spblk = nullptr;      // [COPO_MAX_SAFE] slot masks: who stands on respawn place q (filled during P1)
    float* lcf_pre = nullptr;                 // [64] / [64]: the draws of each slot's NEXT spawn (filled by the last wave during P0)
    uint32_t* hr_pre = nullptr;
    {   // route tables: a few KB read on every step by the projection / navigation code -> LDS copy when they fit
        // (the waves that idle during P0 do the copy; the barrier after P0 publishes it)
        extern __shared__ unsigned int dyn[];
        const int nseg_f = p.n_routes * p.seg_rows * COPO_SEG_STRIDE, nmeta_f = p.n_routes * 4;
        const int ntab = p.n_spawns * 4, nsp = p.n_spawns;
        const bool stage = p.stage_tables != 0;     // (host: several waves per scene and the tables are small)
        float* rl = reinterpret_cast<float*>(dyn + lidar_lds_words(p.chunk, p.nbr_chunk, p.N, p.num_lasers) + ray_lds_words(p.num_lasers) + (nthreads >> 6) * LIDAR_WAVE_WORDS);
        int32_t* tl = reinterpret_cast<int32_t*>(rl + nseg_f + nmeta_f);
        float* sl = reinterpret_cast<float*>(tl + ntab);
        // (several waves per scene: 64 float4 records for neighbours_fast behind everything else -- the LiDAR minima are live then)
        rec_roles = reinterpret_cast<float4*>(stage ? reinterpret_cast<float*>(((reinterpret_cast<uintptr_t>(sl + nsp) + 15) & ~(uintptr_t)15)) : rl);
        spblk = reinterpret_cast<unsigned long long*>(rec_roles + 64);
        lcf_pre = reinterpret_cast<float*>(spblk + COPO_MAX_SAFE);
        hr_pre = reinterpret_cast<uint32_t*>(lcf_pre + 64);
        if (copy_apart && wave == nwaves - 1 && lane < N && !(COPO_PROFILE_SKIP & 32)) {
            const uint32_t cnt = (uint32_t)reinterpret_cast<const int32_t*>(p.state)[(size_t)15 * p.E * N + (size_t)e * N + lane] & 0xffffu;
            uint32_t h;
            float lcf;
            spawn_draws(p, p.seeds[e], (uint32_t)p.env[(size_t)e * 4 + 1], lane, cnt, h, lcf);
            hr_pre[lane] = h;
            lcf_pre[lane] = lcf;
        }
        if (stage && ctid >= 0) {
            for (int q = ctid; q < nseg_f; q += cnth) rl[q] = p.route_segs[q];
            for (int q = ctid; q < nmeta_f; q += cnth) rl[nseg_f + q] = p.route_meta[q];
            for (int q = ctid; q < ntab; q += cnth) tl[q] = p.spawn_tab[q];
            for (int q = ctid; q < nsp; q += cnth) sl[q] = p.spawn_s[q];
        }
        if (tid == 0) {
            L.seg_rows = p.seg_rows;
            L.rsegs = stage ? rl : p.route_segs;
            L.rmeta = stage ? rl + nseg_f : p.route_meta;
            L.stab = stage ? tl : p.spawn_tab;
            L.sps = stage ? sl : p.spawn_s;
        }
    }

#define COPO_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(size_t)e * COPO_DBG_STRIDE + (i)] = (long long)clock64(); } while (0)
    COPO_STAMP(0);
    if ((COPO_PROFILE_SKIP & 512) && p.dbg && tid == 0) p.dbg[(size_t)e * 16 + 11] = (long long)wall_clock64();     // (100 MHz: the shader clock of the launch)
    if ((COPO_PROFILE_SKIP & 512) && p.dbg && lane == 0 && !ONE)     // which SIMD runs which wave (HW_ID bits 5:4), 2 bits per wave
        atomicOr(reinterpret_cast<unsigned long long*>(p.dbg) + (size_t)e * 16 + 13, (unsigned long long)__builtin_amdgcn_s_getreg(2308) << (2 * wave));
    // ---- P0 (wave 0): timers + kinematic bicycle, poses -> LDS ----------------------------------------
    Slot s = Slot{};
    bool acted = false;
    float acc = 0.0f;
    int32_t t_env = 0, episode = 0, next_aid = 0;
    uint64_t seed = 0;
    if (wave == 0) {
        const int32_t* env = p.env + (size_t)e * 4;
        t_env = env[0]; episode = env[1]; next_aid = env[2];
        seed = p.seeds[e];
        if (lane < N) {
            load_slot(p, e, lane, s);
            if ((COPO_PROFILE_SKIP & 512) && p.dbg) {        // when have P0's loads arrived?  (slot 14: cycles after stamp 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) p.dbg[(size_t)e * 16 + 14] = (long long)clock64() - p.dbg[(size_t)e * 16 + 0];
            }
            slot_dynamics<EXT>(p, act, (size_t)e * N + lane, s, acted, acc);
            stage_pose(L, lane, s);
            if (ONE && st_status(s.status) == ST_EMPTY) L.x[lane] = 1.0e18f;      // (collision rows of P1: no vehicle here; P2 stages the pose again)
        } else {
            L.x[lane] = 0.0f; L.y[lane] = 0.0f; L.cs[lane] = 1.0f; L.sn[lane] = 0.0f;
        }
        const unsigned long long ma = __ballot(acted);
        const bool sol0 = lane < N && st_status(s.status) != ST_EMPTY;
        const unsigned long long ms = __ballot(sol0);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (acted) L.alist[__popcll(ma & lt)] = (uint8_t)lane;
        if (sol0) L.clist[__popcll(ms & lt)] = (uint8_t)lane;
        L.crash[lane] = 0;
        if (lane == 0) {
            L.m_acted = ma;
            L.m_solid = ms;
            L.ending = 0;
        }
    }
    __syncthreads();
    COPO_STAMP(1);

    // ---- P1 (all waves): collision of every (acting agent, solid vehicle) pair, one lane per pair -----
    {
        const int na = __popcll(L.m_acted), nc = __popcll(L.m_solid);
        const int npair = na * nc;
        const float inv_nc = 1.0f / (float)(nc > 0 ? nc : 1);
        const float near2 = 4.0f * (hl * hl + hw * hw) * 1.001f;   // farther apart than two circumradii: no overlap
        extern __shared__ unsigned int dyn[];
        const int qcap = 2 * lidar_lds_words(p.chunk, p.nbr_chunk, N, p.num_lasers);      // the LiDAR / neighbour work area is free here
        if (ONE && (N * (N - 1)) / 2 <= qcap && !(COPO_PROFILE_SKIP & 16)) {
            // one wave owns the scene: row r = 1 .. N / 2 pairs slot `lane` with slot (lane + r) mod N -- every UNORDERED pair once (the
            // separating-axis test is symmetric bit for bit: exact negations, commutative products; the last row of an even N pairs
            // every slot with its opposite, so its first half is all of it).  Slots without a vehicle stand at x = 1e18 (P0), so only
            // the lane's own flag is tested.  The pairs closer than two circumradii are queued and box-tested together; the mark of a
            // vehicle that did not act (a wreck) is never read.
            uint16_t* nq = reinterpret_cast<uint16_t*>(dyn);
            int nn = 0;
            const bool sol_me = (L.m_solid >> lane) & 1ull;
            const float xme = L.x[lane], yme = L.y[lane];
            const int R = N >> 1;
            int jj = lane < N ? lane : 0;
            for (int r = 1; r <= R; r += 2) {      // two rows per turn: both poses are requested before either is used
                const int j0 = jj + 1 >= N ? jj + 1 - N : jj + 1, j1 = jj + 2 >= N ? jj + 2 - N : jj + 2;
                jj = j1;
                const float x0 = L.x[j0], y0 = L.y[j0], x1 = L.x[j1], y1 = L.y[j1];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rr_ = r + h, jh = h ? j1 : j0;
                    const float ddx = (h ? x1 : x0) - xme, ddy = (h ? y1 : y0) - yme;
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
                    const int pk = (int)nq[q0 + lane], i = pk >> 8, j = pk & 255;
                    if (obb_overlap2(L.x[i], L.y[i], L.cs[i], L.sn[i], hl, hw, L.x[j], L.y[j], L.cs[j], L.sn[j], hl, hw)) {
                        L.crash[i] = 1;
                        L.crash[j] = 1;
                    }
                }
            }
        } else
        for (int c0 = wave * 64; c0 < ((COPO_PROFILE_SKIP & 16) ? 0 : npair); c0 += nwaves * 64) {
            const int c = c0 + lane;
            const bool live = c < npair;
            const int ia = live ? (int)(((float)c + 0.5f) * inv_nc) : 0;
            const int i = L.alist[ia], j = L.clist[live ? c - ia * nc : 0];
            const float xi = L.x[i], yi = L.y[i], xj = L.x[j], yj = L.y[j];
            const float ddx = xj - xi, ddy = yj - yi;
            const bool near = live && (i != j) && (fm(ddx, ddx, ddy * ddy) <= near2);
            if (near && obb_overlap2(xi, yi, L.cs[i], L.sn[i], hl, hw, xj, yj, L.cs[j], L.sn[j], hl, hw)) L.crash[i] = 1;
        }
    }
    // several waves per scene: which vehicles stand in the region of respawn place q, as a slot mask per place, next to
    // the collision pairs -- P2 (wave 0 alone, on the critical path of a launch with one scene per
    // compute unit) then only ANDs the masks with the slots that are still solid
    if (!ONE && nwaves > 1 && !(COPO_PROFILE_SKIP & 32)) {     // (places dealt from the last wave down: those have the fewest pairs)
        const float xj = L.x[lane], yj = L.y[lane], cj = L.cs[lane], sj = L.sn[lane];
        for (int q = nwaves - 1 - wave; q < p.n_safe; q += nwaves) {
            const float4 sp4 = reinterpret_cast<const float4*>(p.safe_pose)[q];
            const bool blk = lane < N && obb_overlap2(sp4.x, sp4.y, sp4.z, sp4.w, p.region_hl, p.region_hw, xj, yj, cj, sj, hl, hw);
            const unsigned long long mq = __ballot(blk);
            if (lane == 0) spblk[q] = mq;
        }
    }
    __syncthreads();

    // MultiAgentMetaDrive.step: after `horizon` env steps the scene stops respawning and drains; it is reset when nobody
    // is left driving (P2 decides).  Every agent has its own limit of `horizon` steps (max_step).
    const bool no_respawn = t_env + 1 >= p.horizon, force_end = t_env + 1 >= 5 * p.horizon;
    uint8_t fl = 0;
    float rew = 0.0f, lcf_row = 0.0f;
    int32_t aid_row = -1;
    bool present = false;
    COPO_STAMP(2);
    // Wave roles after P2 (8 / 16 waves per scene, i.e. few scenes per launch, where a launch is latency-bound): wave 0 writes the
    // state back (P4), waves 1 and 2 build the neighbour lists in registers (neighbours_roles: counts + sums / keys), waves 3.. run the
    // LiDAR windows / box tests; then everybody writes the LiDAR columns, except the waves that evaluate an undecided agent's lists.  The ray minima are initialised here, by the waves that idle during P2.
    const bool roles_ok = !ONE && nwaves >= 8 && p.nbr_fast && out.obs != nullptr && !(EXT && p.col_comm >= 0) && !(COPO_PROFILE_SKIP & 0x187);
    if (roles_ok && wave != 0) {
        extern __shared__ unsigned int dyn[];
        const unsigned int range_bits = __float_as_uint(p.lidar_range);
        for (int q = tid - 64; q < N * p.num_lasers; q += nthreads - 64) dyn[q] = range_bits;
    }
    // ---- P2 (wave 0): route projection, termination, reward, respawn ------------------------------
    if (wave == 0) {
        bool term = false;
        lcf_row = s.lcf;
        aid_row = acted ? s.aid : -1;
        if (acted && !(COPO_PROFILE_SKIP & 64)) {
            slot_project(p, L, L.cs[lane], L.sn[lane], L.crash[lane] != 0, force_end, acc, out.info ? out.info + ((size_t)e * N + lane) * COPO_INFO_DIM : nullptr, s, fl, rew, term);
        } else if (lane < N && out.info) {
            float* q = out.info + ((size_t)e * N + lane) * COPO_INFO_DIM;
            for (int k = 0; k < COPO_INFO_DIM; ++k) q[k] = 0.0f;
        }
        present = acted;
        // respawn: a random one of the respawn places whose region is clear of the vehicles standing now, each place at
        // most once per step; serial over the eligible slots in slot order, every lane tests its own vehicle
        if (!no_respawn && !(COPO_PROFILE_SKIP & 32)) {
            const bool mine = lane < capacity_of(p) && !acted && s.status == st_pack(ST_EMPTY, 0, 0);
            unsigned long long elig = __ballot(mine);
            if (elig) {
                const bool solid_now = lane < N && st_status(s.status) != ST_EMPTY;
                const float cj = L.cs[lane], sj = L.sn[lane];      // poses of terminated vehicles did not change since P0
                uint32_t clear = 0;
                if (!ONE && nwaves > 1) {          // the box tests ran during P1 (above); the poses of standing vehicles did not change
                    const unsigned long long solid_mask = __ballot(solid_now);
                    for (int q = 0; q < p.n_safe; ++q)
                        if ((spblk[q] & solid_mask) == 0ull) clear |= 1u << q;
                } else if (ONE) {
                    // one wave owns the scene: the (place, vehicle) pairs closer than the two circumradii are queued and box-tested
                    // together (a box test per place ran for 64 lanes with one or two candidates among them)
                    extern __shared__ unsigned int dyn[];
                    uint16_t* bq = reinterpret_cast<uint16_t*>(dyn);
                    const int bq_cap = 2 * lidar_lds_words(p.chunk, p.nbr_chunk, N, p.num_lasers) - 64;
                    const float rr = sqrtf(p.region_hl * p.region_hl + p.region_hw * p.region_hw) + sqrtf(hl * hl + hw * hw);
                    const float rr2 = rr * rr * 1.001f;
                    uint32_t blocked = 0;
                    int nbq = 0;
                    auto flush = [&]() {
                        pk_wave_sync();
                        for (int q0 = 0; q0 < nbq; q0 += 64) {
                            bool blk = false;
                            int q = 0;
                            if (q0 + lane < nbq) {
                                const int ent = (int)bq[q0 + lane], n = ent & 255;
                                q = ent >> 8;
                                const float4 sp4 = reinterpret_cast<const float4*>(p.safe_pose)[q];
                                blk = obb_overlap2(sp4.x, sp4.y, sp4.z, sp4.w, p.region_hl, p.region_hw, L.x[n], L.y[n], L.cs[n], L.sn[n], hl, hw);
                            }
                            for (unsigned long long mb = __ballot(blk); mb; mb &= mb - 1ull)
                                blocked |= 1u << __builtin_amdgcn_readlane(q, __ffsll((long long)mb) - 1);
                        }
                        pk_wave_sync();
                        nbq = 0;
                    };
                    // (lane q loads place q once: a load per iteration would put a memory round trip in front of every place)
                    const float2 spq = lane < p.n_safe ? *reinterpret_cast<const float2*>(p.safe_pose + 4 * lane) : make_float2(0.0f, 0.0f);
                    for (int q = 0; q < p.n_safe; ++q) {
                        const float dx = s.x - readlane_f(spq.x, q), dy = s.y - readlane_f(spq.y, q);
                        const bool pre = solid_now && (fm(dx, dx, dy * dy) <= rr2);
                        const unsigned long long m = __ballot(pre);
                        if (m != 0ull) {
                            const int c = __popcll(m);
                            if (nbq + c > bq_cap) flush();
                            if (pre) bq[nbq + pk_mbcnt(m)] = (uint16_t)((q << 8) | lane);
                            nbq += c;
                        }
                    }
                    flush();
                    clear = ~blocked & (p.n_safe >= 32 ? 0xffffffffu : ((1u << p.n_safe) - 1u));
                } else
                for (int q = 0; q < p.n_safe; ++q) {
                    const float4 sp4 = reinterpret_cast<const float4*>(p.safe_pose)[q];      // pose of respawn place q (host table)
                    const bool blk = solid_now && obb_overlap2(sp4.x, sp4.y, sp4.z, sp4.w, p.region_hl, p.region_hw,
                                                               s.x, s.y, cj, sj, hl, hw);
                    if (__ballot(blk) == 0ull) clear |= 1u << q;
                }
                // the serial part only hands out places (a few scalar operations per waiting slot); the spawns themselves
                // -- three hash chains, the LCF draw -- run afterwards for all chosen lanes at once
                uint32_t used = 0;
                int my_q = -1, my_route = -1;
                int32_t my_aid = 0;
                uint32_t taken = 0;
                if (p.n_spaces > 0) taken = spaces_taken(p, L.rmeta, lane < N && st_status(s.status) == ST_ALIVE, s.route);
                while (elig) {
                    const int n = __ffsll((long long)elig) - 1;
                    elig &= elig - 1;
                    const uint32_t freem = clear & ~used;
                    if (!freem) break;
                    const int nfree = __popc(freem);
                    const uint32_t cnt_n = (uint32_t)__builtin_amdgcn_readlane(s.spawncnt, n) & 0xffffu;
                    const uint32_t hh = hash_rng(seed, (uint32_t)n, cnt_n, (uint32_t)t_env, RNG_SPAWN);
                    int pick = (int)(hh % (uint32_t)nfree);
                    uint32_t m = freem;
                    while (pick > 0) { m &= m - 1; --pick; }
                    const int q = __ffs((int)m) - 1;
                    used |= 1u << q;
                    if (p.n_spaces > 0) {
                        const uint32_t hr = copy_apart ? hr_pre[n] : hash_rng(seed, (uint32_t)n, cnt_n, (uint32_t)episode, RNG_ROUTE);
                        const int r = pick_route_exclusive(L.rmeta, L.stab, p.safe_ids[q], hr, taken);
                        if (lane == n) my_route = r;
                    }
                    if (lane == n) { my_q = q; my_aid = next_aid; }
                    next_aid += 1;
                }
                if (my_q >= 0) {
                    if (copy_apart) spawn_slot(p, L.rsegs, L.stab, L.sps, seed, (uint32_t)episode, lane, p.safe_ids[my_q], my_aid, s, true, hr_pre[lane], lcf_pre[lane], my_route);
                    else spawn_slot(p, L.rsegs, L.stab, L.sps, seed, (uint32_t)episode, lane, p.safe_ids[my_q], my_aid, s, false, 0u, 0.0f, my_route);
                    present = true;
                    fl = COPO_F_SPAWNED;
                    lcf_row = s.lcf;
                }
            }
        }
        if (lane < N) {
            stage_pose(L, lane, s);
            L.rew[lane] = rew;
        } else {
            L.rew[lane] = 0.0f;
        }
        const unsigned long long mp = __ballot(present);
        const unsigned long long ms = __ballot(lane < N && st_status(s.status) != ST_EMPTY);
        const unsigned long long alive = __ballot(lane < N && st_status(s.status) == ST_ALIVE);
        build_lists(L, lane, mp, ms);
        if (lane == 0) {
            L.m_present = mp;
            L.m_solid = ms;
            L.ending = alive == 0ull ? 1 : 0;   // nobody left driving: the episode is over
        }
    }
    __syncthreads();
    const bool ending = L.ending != 0;

    COPO_STAMP(3);
    const bool roles = roles_ok && !ending;      // (a scene that resets this step: the neighbour lists need the poses BEFORE the reset, the LiDAR the ones after)
    unsigned int reach_lo = 0u, reach_hi = 0u;   // one wave per scene: the solid vehicles within LiDAR reach of this lane's slot (from the neighbour walk)
    bool have_reach = false;
    // ---- P3 (all waves): neighbour lists + reward reductions on the post-step (pre-reset) scene ---
    if (roles) {
        if (wave == 1) {
            unsigned long long odd = 0ull;
            const unsigned long long ex = neighbours_roles<1>(p, L, e, lane, out, &odd);
            if (lane == 0) {
                L.nbr_exact = ex;
                L.nbr_odd = odd;
            }
            COPO_ROLE_STAMP(9);
        } else if (wave == 2) {
            unsigned long long odd = 0ull;
            const unsigned long long ex = neighbours_roles<2>(p, L, e, lane, out, &odd);
            if (lane == 0) L.nbr_exact2 = ex;
            COPO_ROLE_STAMP(9);
        } else if (wave >= 3) {
            obs_phase<2>(p, L, e, tid, nthreads, out.obs, 3);
            COPO_ROLE_STAMP(10);
        }
    } else if (!(COPO_PROFILE_SKIP & 1)) neighbours_any<EXT>(p, L, e, tid, nthreads, out, act, L.m_acted, ending,
                                                             ONE ? &reach_lo : nullptr, ONE ? &reach_hi : nullptr, ONE ? &have_reach : nullptr);
    else __syncthreads();
    COPO_STAMP(4);
    // (neighbours_phase ends with a workgroup barrier: the reset below may overwrite the poses it read)

    // ---- P4 (wave 0): end-of-episode reset, row outputs, state write-back, ego/navi obs ------------------
    if (wave == 0) {
        uint8_t fl_out = fl;
        float lcf_out = lcf_row;
        if (ending) {
            episode += 1;
            next_aid = 0;
            reset_env_wave0(p, L, seed, (uint32_t)episode, lane, s);
            const int cap = capacity_of(p);
            next_aid = cap;
            present = lane < cap;
            fl_out |= (lane < cap ? COPO_F_SPAWNED : 0) | COPO_F_ENV_RESET;
            if (!acted && lane < cap) lcf_out = s.lcf;
            if (lane < N) stage_pose(L, lane, s);
            const unsigned long long all = __ballot(lane < cap);
            build_lists(L, lane, all, all, false);
            if (lane == 0) {
                L.m_present = all;
                L.m_solid = all;
            }
        }
        if (lane < N) {
            const size_t o = (size_t)e * N + lane;
            if (out.rew) out.rew[o] = rew;
            if (out.flags) out.flags[o] = fl_out;
            if (out.lcf) out.lcf[o] = lcf_out;
            if (out.agent_id) out.agent_id[o] = aid_row;
            store_slot(p, e, lane, s);
        }
        if (lane == 0) {
            int32_t* env = p.env + (size_t)e * 4;
            env[0] = ending ? 0 : t_env + 1;
            env[1] = episode;
            env[2] = next_aid;
        }
        if (!(COPO_PROFILE_SKIP & 8)) ego_navi_obs<EXT>(p, L, L.cs[lane], L.sn[lane], s, present, (out.obs && lane < N) ? out.obs + ((size_t)e * N + lane) * p.O : nullptr, ending ? 0 : t_env + 1, ending);
        COPO_ROLE_STAMP(8);
    }
    __syncthreads();

    COPO_STAMP(5);
    // ---- P5 (all threads): LiDAR + observation write-out -------------------------------------------
    if (roles) {
        // the agents whose neighbour lists the registers did not decide (<= NBR_EXACT_MAX, usually none): one each on waves
        // 1 .. n, next to the write-out of the ray minima by the other waves
        const unsigned long long ex_all = L.nbr_exact | L.nbr_exact2;
        const bool nbr_ok = __popcll(ex_all) <= NBR_EXACT_MAX;
        const unsigned long long ex = nbr_ok ? ex_all : 0ull;
        const int n_ex = __popcll(ex);
        if (p.dbg && tid == 0) p.dbg[(size_t)e * COPO_DBG_STRIDE + 7] = !nbr_ok ? 2 : (n_ex ? 16 + n_ex : 1);
        obs_phase<4>(p, L, e, tid, nthreads, out.obs, 0, n_ex);
        if (wave >= 1 && wave <= n_ex) {
            unsigned long long m = ex;
            for (int k = 1; k < wave; ++k) m &= m - 1ull;
            neighbours_exact_one(p, e, lane, __ffsll((long long)m) - 1, L.x[lane], L.y[lane], L.rew[lane], L.m_present, L.nbr_odd, out);
        }
        if (!nbr_ok) {              // the register formulation declined (ties, band cases, odd rewards): the pair-parallel lists, on the
            __syncthreads();        // work area that the ray minima no longer need
            neighbours_phase<EXT>(p, L, e, tid, nthreads, out, act, L.m_acted, ending);
        }
    } else if (ONE && out.obs && have_reach && lidar_queue_words(p.chunk, N) > 0) {
        // one wave owns the scene: the pair queue comes from the reach masks of the neighbour walk (a scene that reset has new poses:
        // every solid vehicle is handed to the window pass, which tests the reach itself)
        extern __shared__ unsigned int dyn[];
        const int o_q = (p.chunk > 0 ? p.chunk : N) * p.num_lasers + LIDAR_MIN_PAD;
        lidar_by_wave(p, [](int v) { return make_float4(L.x[v], L.y[v], L.cs[v], L.sn[v]); }, L.plist, reach_lo, reach_hi, dyn,
                      reinterpret_cast<uint16_t*>(dyn + o_q), lds_rays(p),
                      reinterpret_cast<int*>(dyn + lidar_lds_words(p.chunk, p.nbr_chunk, N, p.num_lasers) + ray_lds_words(p.num_lasers)), e, lane,
                      L.m_present, L.m_solid, ending, out.obs);
    } else if (out.obs) obs_phase(p, L, e, tid, nthreads, out.obs);
    __syncthreads();
    COPO_STAMP(6);
    if ((COPO_PROFILE_SKIP & 512) && p.dbg && tid == 0) p.dbg[(size_t)e * 16 + 12] = (long long)wall_clock64();
#undef COPO_STAMP
}

// ------------------------------------------------------------------------------------------------
// stateless neighbour op (CCEnv + LCFEnv reward block) on caller-provided positions
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) neighbours_kernel(const float* __restrict__ pos,
                                                         const uint8_t* __restrict__ present_in,
                                                         const float* __restrict__ rew, SimParams p, StepOut out) {
    __shared__ EnvLds L;
    const int e = blockIdx.x, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = p.N;
    if (wave == 0) {
        bool pr = false;
        if (lane < N) {
            const size_t o = (size_t)e * N + lane;
            L.x[lane] = pos[o * 2];
            L.y[lane] = pos[o * 2 + 1];
            L.rew[lane] = rew ? rew[o] : 0.0f;
            pr = present_in[o] != 0;
        } else {
            L.x[lane] = 0.0f; L.y[lane] = 0.0f; L.rew[lane] = 0.0f;
        }
        const unsigned long long m = __ballot(pr);
        build_lists(L, lane, m, 0ull);
        if (lane == 0) L.m_present = m;
    }
    __syncthreads();
    neighbours_phase<false>(p, L, e, tid, (int)blockDim.x, out);
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
// [slots][rays] minima, ray table, one 64-entry strip per wave (box-test owners)
static size_t lidar_lds_bytes(const SimParams& p, int block) {
    return (size_t)(lidar_lds_words(p.chunk, p.nbr_chunk, p.N, p.num_lasers) + ray_lds_words(p.num_lasers) + (block / 64) * LIDAR_WAVE_WORDS) * sizeof(unsigned int) +
           (block > 64 ? 64 * sizeof(float4) + 16 + COPO_MAX_SAFE * sizeof(unsigned long long) + 128 * sizeof(float) : 0);       // (step kernel, several waves per
                                                                                                      //  scene: the records of neighbours_fast, the blocker masks of the respawn places, the spawn draws)
}
static bool sim_has_ext(const SimParams& p) { return p.col_tl >= 0 || p.col_comm >= 0; }
static hipError_t sim_lds_attrs() {              // 64 slots x 256 rays + route tables exceed the default 64 KB
    static hipError_t once = [] {
        hipError_t r = hipSuccess;
        for (const void* f : {reinterpret_cast<const void*>(sim_reset_kernel<false>), reinterpret_cast<const void*>(sim_reset_kernel<true>),
                              reinterpret_cast<const void*>(sim_step_kernel<false, false>), reinterpret_cast<const void*>(sim_step_kernel<true, false>),
                              reinterpret_cast<const void*>(sim_step_kernel<false, true>), reinterpret_cast<const void*>(sim_step_kernel<true, true>)}) {
            const hipError_t a = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (a != hipSuccess) r = a;
        }
        return r;
    }();
    return once;
}
static size_t route_lds_bytes(const SimParams& p) {      // LDS copy of the route tables (+ spawn table and offsets)
    return p.stage_tables ? (size_t)route_table_floats(p.n_routes, p.seg_rows) * sizeof(float) + (size_t)p.n_spawns * 5 * sizeof(float) : 0;
}
// Launch shape -> the two fields of the parameter block that depend on it.  Several waves per scene: everything of a scene
// at once, route tables in LDS when small.  One wave per scene (large scene counts): work areas for 8 agents at a time and
// tables from L2, so that a scene needs ~8 KB of LDS and a compute unit holds ~20 of them.
void sim_shape_params(SimParams& p, int block) {
    p.pack_scenes = 0;
    if (block < 0) {          // packed shape (sim_packed.hip): -block scenes per workgroup; the reset kernel runs one wave per scene
        sim_shape_params(p, 64);
        p.pack_scenes = -block;
        p.chunk = sim_packed_chunk(p);
        return;
    }
    // one wave per scene: 10 fans at a time (scripts/bench_sim.py --chunks, 16 384 populated scenes: 6 263 / 8 257 / 10 251 / 12 262 /
    // 14 259 / 16 275 us -- fuller pair and box-test batches against resident scenes per compute unit, 26 at 6.2 KB of LDS)
    // (with more beams per fan, fewer fans: 240 beams x 10 slots 166 us at fans of 10, 124 at 5, 126 at 4, 130 at 6)
    const int lch_auto = 1280 / (p.num_lasers > 0 ? p.num_lasers : 1);
    const int lch = p.chunk_one_wave > 0 ? p.chunk_one_wave : (lch_auto < 2 ? 2 : (lch_auto > 10 ? 10 : lch_auto));
    p.chunk = block > 64 ? p.N : (p.N < lch ? p.N : lch);
    // (pair-parallel neighbour lists, one wave per scene: 4 agents at a time when they only serve the scenes the register
    //  formulation declines, 8 when they are the only formulation this configuration has)
    const int nch = p.nbr_fast ? 4 : 8;
    p.nbr_chunk = block > 64 ? p.N : (p.N < nch ? p.N : nch);
    p.stage_tables = (block > 64 && (size_t)route_table_floats(p.n_routes, p.seg_rows) * sizeof(float) <= (size_t)ROUTE_LDS_MAX_BYTES) ? 1 : 0;
}

hipError_t launch_sim_reset(const SimParams& p, const SimParams* p_dev, const StepOut& out, int block, hipStream_t stream) {
    if (hipError_t a = sim_lds_attrs(); a != hipSuccess) return a;
    if (block < 0) block = 64;
    if (sim_has_ext(p)) hipLaunchKernelGGL(sim_reset_kernel<true>, dim3(p.E), dim3(block), lidar_lds_bytes(p, block), stream, p_dev, out);
    else hipLaunchKernelGGL(sim_reset_kernel<false>, dim3(p.E), dim3(block), lidar_lds_bytes(p, block), stream, p_dev, out);
    return hipGetLastError();
}

hipError_t launch_sim_step(const SimParams& p, const SimParams* p_dev, const float* act, const StepOut& out, int block, hipStream_t stream) {
    if (block < 0) return launch_sim_step_packed(p, p_dev, act, out, -block, stream);
    if (hipError_t a = sim_lds_attrs(); a != hipSuccess) return a;
    const size_t lds = lidar_lds_bytes(p, block) + route_lds_bytes(p);
    if (block == 64) {
        if (sim_has_ext(p)) hipLaunchKernelGGL((sim_step_kernel<true, true>), dim3(p.E), dim3(64), lds, stream, p_dev, act, out);
        else hipLaunchKernelGGL((sim_step_kernel<false, true>), dim3(p.E), dim3(64), lds, stream, p_dev, act, out);
    } else {
        if (sim_has_ext(p)) hipLaunchKernelGGL((sim_step_kernel<true, false>), dim3(p.E), dim3(block), lds, stream, p_dev, act, out);
        else hipLaunchKernelGGL((sim_step_kernel<false, false>), dim3(p.E), dim3(block), lds, stream, p_dev, act, out);
    }
    return hipGetLastError();
}

hipError_t launch_neighbours(const float* pos, const uint8_t* present, const float* rew, const SimParams& p,
                             const StepOut& out, hipStream_t stream) {
    hipLaunchKernelGGL(neighbours_kernel, dim3(p.E), dim3(256), (size_t)nbr_lds_words(p.N, p.N) * sizeof(unsigned int), stream, pos, present, rew, p, out);
    return hipGetLastError();
}

}  // namespace copo

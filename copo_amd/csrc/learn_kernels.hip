// Learn-side custom ops for gfx950 (HBM-bound scans / gathers / reductions):
//   gae3            three GAE heads in one segmented reverse scan        (algo_ccppo.py:362-373, algo_copo.py:189-204,492-500)
//   cc_fuse_mf      mean-field centralised-critic observation            (algo_ccppo.py:266-311)
//   cc_fuse_concat  concat centralised-critic observation                (algo_ccppo.py:225-263)
//   lcf_mix         coordinated advantage + batch standardisation        (algo_copo.py:539-551)
#include <string.h>

#include "sim_common.h"

namespace copo {

// ------------------------------------------------------------------------------------------------
// GAE: one thread per (head, column); columns are contiguous so every time-step is a coalesced row read.
// The dtype path of the reference is kept: delta in fp32 for truncated trajectories (numpy fp32
// expression), in fp64 for trajectories that ended with done; the discounted cumsum always in fp64.
// ------------------------------------------------------------------------------------------------
struct Gae3Args {
    double gamma[4];
    double lam;
};

__global__ void __launch_bounds__(256) gae3_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                   const uint8_t* __restrict__ flags, int T, int M, int heads,
                                                   Gae3Args a, float* __restrict__ adv, float* __restrict__ tgt) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)heads * M) return;
    const int hd = (int)(gid / M), m = (int)(gid - (long long)hd * M);
    const size_t hoff = (size_t)hd * T * M;
    const double g64 = a.gamma[hd];
    const float g32 = (float)g64;
    const double c = g64 * a.lam;
    double acc = 0.0, vnext64 = 0.0;
    float vnext32 = 0.0f;
    bool seg_done = false, in_seg = false;
    for (int t = T - 1; t >= 0; --t) {
        const size_t ix = (size_t)t * M + m;
        const uint8_t f = flags[ix];
        if (!(f & COPO_F_ACTED)) {
            adv[hoff + ix] = 0.0f;
            tgt[hoff + ix] = 0.0f;
            in_seg = false;
            continue;
        }
        const float r = rew[hoff + ix], v = val[hoff + ix];
        if (!in_seg || (f & COPO_F_DONE)) {
            seg_done = (f & COPO_F_DONE) != 0;
            acc = 0.0;
            vnext32 = v;
            vnext64 = 0.0;
            in_seg = true;
        }
        double delta;
        if (seg_done) {
            delta = ((double)r + g64 * vnext64) - (double)v;
        } else {
            const float d32 = (r + g32 * vnext32) - v;
            delta = (double)d32;
        }
        acc = delta + c * acc;
        adv[hoff + ix] = (float)acc;
        tgt[hoff + ix] = (float)(acc + (double)v);
        vnext32 = v;
        vnext64 = (double)v;
    }
}

hipError_t launch_gae3(const float* rew, const float* val, const uint8_t* flags, int T, int M, int heads,
                       const double* gamma_host, double lam, float* adv, float* tgt, hipStream_t stream) {
    Gae3Args a{};
    for (int h = 0; h < heads && h < 4; ++h) a.gamma[h] = gamma_host[h];
    a.lam = lam;
    const long long total = (long long)heads * M;
    const int block = 256;
    const int grid = (int)((total + block - 1) / block);
    hipLaunchKernelGGL(gae3_kernel, dim3(grid), dim3(block), 0, stream, rew, val, flags, T, M, heads, a, adv, tgt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Centralised-critic fusion: one wave per row (r, n); lanes stride the feature columns, so every
// neighbour row is one coalesced read and the per-column accumulation order is the list order.
// A neighbour contributes only if it has a row at the same env time-step (ACTED flag), which is the
// `np.where(nei_batch["t"] == environmental_time_step)` match of the reference.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cc_fuse_mf_kernel(const float* __restrict__ obs, const float* __restrict__ act,
                                                         const uint8_t* __restrict__ flags,
                                                         const int32_t* __restrict__ nbr_idx,
                                                         const int32_t* __restrict__ cnt, long long rows, int N, int O,
                                                         int A, int K, int cf, float* __restrict__ cc) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int C = 2 * O + (cf ? A : 0);
    float* o = cc + (size_t)row * C;
    if (!(flags[row] & COPO_F_ACTED)) {
        for (int k = lane; k < C; k += 64) o[k] = 0.0f;
        return;
    }
    const long long r0 = (row / N) * N;  // first row of this (t, env)
    int m = cnt[row];
    if (m > K) m = K;
    const int W = O + (cf ? A : 0);  // fused columns: O obs then A act
    // neighbour q of the list is decoded by lane q once (index -> ACTED flag of its row); the column loops below only
    // broadcast the surviving row numbers, so no dependent index / flag loads sit in front of the row reads
    long long jr = 0;
    bool ok = false;
    if (lane < m) {
        const int j = nbr_idx[(size_t)row * K + lane];
        if (j >= 0) {
            jr = r0 + j;
            ok = (flags[jr] & COPO_F_ACTED) != 0;
        }
    }
    const unsigned long long live = __ballot(ok);
    const int got = __popcll(live);
    const int jlo = (int)jr;             // rows fit 31 bits in every configuration this op is launched with (checked on the host)
    for (int k0 = 0; k0 < W; k0 += 64) {
        const int k = k0 + lane;
        float sum = 0.0f;
        for (unsigned long long mq = live; mq; mq &= mq - 1) {      // list order: the oracle adds in this order
            const size_t r = (size_t)__builtin_amdgcn_readlane(jlo, __ffsll((long long)mq) - 1);
            if (k < O) sum += obs[r * O + k];
            else if (k < W) sum += act[r * A + (k - O)];
        }
        if (k < W) o[O + k] = got > 0 ? sum / (float)got : 0.0f;
    }
    for (int k = lane; k < O; k += 64) o[k] = obs[(size_t)row * O + k];
}

__global__ void __launch_bounds__(256) cc_fuse_concat_kernel(const float* __restrict__ obs, const float* __restrict__ act,
                                                             const uint8_t* __restrict__ flags,
                                                             const int32_t* __restrict__ nbr_idx,
                                                             const int32_t* __restrict__ cnt, long long rows, int N,
                                                             int O, int A, int K, int nn, int cf,
                                                             float* __restrict__ cc) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int W = O + (cf ? A : 0);
    const int C = O + nn * W;
    float* o = cc + (size_t)row * C;
    const bool acted = (flags[row] & COPO_F_ACTED) != 0;
    for (int k = lane; k < O; k += 64) o[k] = acted ? obs[(size_t)row * O + k] : 0.0f;
    const long long r0 = (row / N) * N;
    int m = acted ? cnt[row] : 0;
    if (m > K) m = K;
    for (int q = 0; q < nn; ++q) {  // slot = rank in the neighbour list, not compacted
        long long jr = -1;
        if (q < m) {
            const int j = nbr_idx[(size_t)row * K + q];
            if (j >= 0 && (flags[r0 + j] & COPO_F_ACTED)) jr = r0 + j;
        }
        float* d = o + O + q * W;
        for (int k = lane; k < W; k += 64) {
            float v = 0.0f;
            if (jr >= 0) v = k < O ? obs[(size_t)jr * O + k] : act[(size_t)jr * A + (k - O)];
            d[k] = v;
        }
    }
}

hipError_t launch_cc_fuse_mf(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                             const int32_t* cnt, int R, int N, int O, int A, int K, int counterfactual, float* cc,
                             hipStream_t stream) {
    const long long rows = (long long)R * N;
    const int wpb = 4;
    const int grid = (int)((rows + wpb - 1) / wpb);
    hipLaunchKernelGGL(cc_fuse_mf_kernel, dim3(grid), dim3(wpb * 64), 0, stream, obs, act, flags, nbr_idx, cnt, rows, N,
                       O, A, K, counterfactual, cc);
    return hipGetLastError();
}

hipError_t launch_cc_fuse_concat(const float* obs, const float* act, const uint8_t* flags, const int32_t* nbr_idx,
                                 const int32_t* cnt, int R, int N, int O, int A, int K, int num_neighbours,
                                 int counterfactual, float* cc, hipStream_t stream) {
    const long long rows = (long long)R * N;
    const int wpb = 4;
    const int grid = (int)((rows + wpb - 1) / wpb);
    hipLaunchKernelGGL(cc_fuse_concat_kernel, dim3(grid), dim3(wpb * 64), 0, stream, obs, act, flags, nbr_idx, cnt, rows,
                       N, O, A, K, num_neighbours, counterfactual, cc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Coordinated advantage + standardisation.  Deterministic two-launch reduction: COPO_LCF_BLOCKS
// per-block partials (fixed tree order) then one block folds them in index order.
// stats layout (doubles): [0..5] = {n, sum, sumsq}(A_c), {n, sum, sumsq}(glob); [8 + 6*b ..] partials.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// four elements per thread and iteration (float4 loads when the arrays are 16-byte aligned), no branch around the
// loads; per-workgroup partials in a fixed layout, folded by one wave per statistic -> deterministic
constexpr int kLcfBlocks = 2048;

__device__ __forceinline__ void lcf_ld4(const float* p, long long i, long long B, bool vec, float (&v)[4]) {
    if (vec && i + 4 <= B) {
        const float4 q = *reinterpret_cast<const float4*>(p + i);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u < B ? i + u : B - 1];
    }
}

__global__ void __launch_bounds__(256) lcf_mix_partial_kernel(const float* __restrict__ adv,
                                                              const float* __restrict__ nei,
                                                              const float* __restrict__ glob,
                                                              const float* __restrict__ lcf,
                                                              const uint8_t* __restrict__ valid, long long B, int vec,
                                                              float* __restrict__ mixed, double* __restrict__ stats) {
    __shared__ double red[4][6];
    double a[6] = {0, 0, 0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < B; i += stride) {
        float va[4], vn[4], vg[4], vl[4], m[4];
        lcf_ld4(adv, i, B, vec, va);
        lcf_ld4(nei, i, B, vec, vn);
        lcf_ld4(glob, i, B, vec, vg);
        lcf_ld4(lcf, i, B, vec, vl);
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ok[u] = i + u < B && (!valid || valid[i + u < B ? i + u : B - 1]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float sn, cs;
            sincos_det(vl[u] * kHalfPi, sn, cs);
            m[u] = ok[u] ? cs * va[u] + sn * vn[u] : 0.0f;
            const double w = ok[u] ? 1.0 : 0.0, g = ok[u] ? (double)vg[u] : 0.0;
            a[0] += w; a[1] += (double)m[u]; a[2] += (double)m[u] * (double)m[u];
            a[3] += w; a[4] += g; a[5] += g * g;
        }
        if (vec && i + 4 <= B) {
            *reinterpret_cast<float4*>(mixed + i) = make_float4(m[0], m[1], m[2], m[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < B) mixed[i + u] = m[u];
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        stats[8 + 6 * blockIdx.x + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

__global__ void __launch_bounds__(384) lcf_mix_fold_kernel(double* __restrict__ stats, int nblocks) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;       // one wave per statistic, fixed order
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += stats[8 + 6 * b + k];
    s = wave_sum(s);
    if (lane == 0) stats[k] = s;
}

__global__ void __launch_bounds__(256) lcf_mix_apply_kernel(const float* __restrict__ mixed,
                                                            const float* __restrict__ glob,
                                                            const uint8_t* __restrict__ valid, long long B, int vec,
                                                            const double* __restrict__ stats,
                                                            float* __restrict__ norm_adv, float* __restrict__ glob_std) {
    const double m0 = stats[1] / stats[0], v0 = stats[2] / stats[0] - m0 * m0;
    const double m1 = stats[4] / stats[3], v1 = stats[5] / stats[3] - m1 * m1;
    double s0 = sqrt(v0 > 0 ? v0 : 0), s1 = sqrt(v1 > 0 ? v1 : 0);
    if (s0 < 1e-4) s0 = 1e-4;
    if (s1 < 1e-4) s1 = 1e-4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < B; i += stride) {
        float vm[4], vg[4], o0[4], o1[4];
        lcf_ld4(mixed, i, B, vec, vm);
        lcf_ld4(glob, i, B, vec, vg);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = i + u < B && (!valid || valid[i + u < B ? i + u : B - 1]);
            o0[u] = ok ? (float)(((double)vm[u] - m0) / s0) : 0.0f;
            o1[u] = ok ? (float)(((double)vg[u] - m1) / s1) : 0.0f;
        }
        if (vec && i + 4 <= B) {
            *reinterpret_cast<float4*>(norm_adv + i) = make_float4(o0[0], o0[1], o0[2], o0[3]);
            *reinterpret_cast<float4*>(glob_std + i) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < B) { norm_adv[i + u] = o0[u]; glob_std[i + u] = o1[u]; }
        }
    }
}

static int lcf_aligned(const void* a, const void* b, const void* c, const void* d) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

hipError_t launch_lcf_mix_partial(const float* adv, const float* nei_adv, const float* glob_adv, const float* lcf,
                                  const uint8_t* valid, int64_t B, float* mixed, double* stats, hipStream_t stream) {
    const int vec = lcf_aligned(adv, nei_adv, glob_adv, lcf) && ((reinterpret_cast<uintptr_t>(mixed) & 15) == 0);
    hipLaunchKernelGGL(lcf_mix_partial_kernel, dim3(kLcfBlocks), dim3(256), 0, stream, adv, nei_adv, glob_adv, lcf,
                       valid, (long long)B, vec, mixed, stats);
    hipLaunchKernelGGL(lcf_mix_fold_kernel, dim3(1), dim3(384), 0, stream, stats, kLcfBlocks);
    return hipGetLastError();
}

hipError_t launch_lcf_mix_apply(const float* mixed, const float* glob_adv, const uint8_t* valid, int64_t B,
                                const double* stats, float* norm_adv, float* glob_std, hipStream_t stream) {
    const int vec = lcf_aligned(mixed, glob_adv, norm_adv, glob_std);
    int grid = (int)((B + 1023) / 1024);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(lcf_mix_apply_kernel, dim3(grid), dim3(256), 0, stream, mixed, glob_adv, valid, (long long)B, vec,
                       stats, norm_adv, glob_std);
    return hipGetLastError();
}

// ---- episode metrics -------------------------------------------------------------------------------------------
// The sums `MultiAgentDrivingCallbacks` needs (utils/callbacks.py:48-110) over the rows of one iteration, in one
// workgroup (fixed order -> deterministic): out[0..7] over rows that acted AND terminated = {count, arrive, crash,
// out_of_road, max_step, sum info[5], sum info[6], sum info[7]}; out[8..14] over rows that acted = {count,
// sum info[0..4], sum neighbour count}.
__global__ void __launch_bounds__(1024) episode_metrics_kernel(const uint8_t* flags, const float* info, const int32_t* nbr_cnt,
                                                               int64_t R, double* out) {
    __shared__ double red[16][15];
    double s[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) s[k] = 0.0;
    for (int64_t i = threadIdx.x; i < R; i += blockDim.x) {
        const unsigned f = flags[i];
        if (!(f & COPO_F_ACTED)) continue;
        const float* q = info + i * COPO_INFO_DIM;
        s[8] += 1.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) s[9 + k] += (double)q[k];
        s[14] += (double)nbr_cnt[i];
        if (f & COPO_F_DONE) {
            s[0] += 1.0;
            s[1] += (f & COPO_F_ARRIVE) ? 1.0 : 0.0;
            s[2] += (f & COPO_F_CRASH) ? 1.0 : 0.0;
            s[3] += (f & COPO_F_OUT) ? 1.0 : 0.0;
            s[4] += (f & COPO_F_MAXSTEP) ? 1.0 : 0.0;
            s[5] += (double)q[5];
            s[6] += (double)q[6];
            s[7] += (double)q[7];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += red[w][threadIdx.x];
        out[threadIdx.x] = v;
    }
}

// ---- minibatch plan of one SGD epoch ------------------------------------------------------------------------
// rows [n_mb][mb] / w [n_mb][mb] / denom [n_mb] from a permutation of this rank's valid rows: minibatch k takes
// size_k = q + (k < r) consecutive entries of the shuffled list (q, r = divmod(B_local, n_mb)); the denominators add
// the same split of every rank's row count.  One launch instead of ~20 tensor ops per epoch.
// Keyed pseudo-random permutation of [0, n): a 4-round Feistel network on the next even power-of-two domain with
// cycle walking (values >= n are encrypted again; expected < 4 walks).  A bijection for every key, O(1) memory, no sort.
__device__ __forceinline__ uint32_t perm_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t feistel_perm(uint32_t i, uint32_t n, uint32_t half, const uint32_t* key) {
    const uint32_t mask = (1u << half) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x >> half, r = x & mask;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t t = l ^ (perm_mix(r ^ key[k]) & mask);
            l = r;
            r = t;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

struct PlanArgs {
    const int64_t* valid_idx;   // [B_local]
    const int64_t* perm;        // [B_local] permutation of 0 .. B_local - 1, or NULL: keyed Feistel permutation
    uint32_t key[4];
    uint32_t half;              // half the (even) bit width of the Feistel domain
    int64_t B_local;
    int32_t n_mb, mb, world;
    int64_t B_all[16];          // every rank's valid-row count (by value: no device round trip)
    int64_t* rows;
    float* w;
    float* denom;
    int64_t* k_index;           // reset to 0
};

__global__ void __launch_bounds__(256) plan_epoch_kernel(PlanArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.n_mb * a.mb;
    if (i == 0 && a.k_index) a.k_index[0] = 0;
    if (i < a.n_mb) {
        double d = 0.0;
        for (int r = 0; r < a.world; ++r) d += (double)(a.B_all[r] / a.n_mb + (i < a.B_all[r] % a.n_mb ? 1 : 0));
        a.denom[i] = (float)(d > 1.0 ? d : 1.0);
    }
    if (i >= total) return;
    const int k = (int)(i / a.mb), j = (int)(i - (int64_t)k * a.mb);
    const int64_t q = a.B_local / a.n_mb, r = a.B_local % a.n_mb;
    const int64_t start = k * q + (k < r ? k : r), size = q + (k < r ? 1 : 0);
    const bool in = j < size && a.B_local > 0;
    int64_t row = 0;
    if (in) {
        int64_t pos = start + j;
        pos = pos < a.B_local ? pos : a.B_local - 1;
        row = a.valid_idx[a.perm ? a.perm[pos] : (int64_t)feistel_perm((uint32_t)pos, (uint32_t)a.B_local, a.half, a.key)];
    }
    a.rows[i] = row;
    a.w[i] = in ? 1.0f : 0.0f;
}

hipError_t launch_plan_epoch(const PlanArgs& a, hipStream_t s) {
    const int64_t total = (int64_t)a.n_mb * a.mb;
    hipLaunchKernelGGL(plan_epoch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// row movers around the SGD loop (each replaces a handful of framework copy / index kernels per call)
// ------------------------------------------------------------------------------------------------
// dst_s[r][:] = src_s[rows[r]][:] for up to COPO_GATHER_MAX_SRC sources in one launch: one wave per output row, float4
// along a row where the widths and bases allow (the epoch's planned rows into minibatch order: observation + pack
// (+ centralised-critic observation) rows of ~75 k minibatch entries)
struct GatherArgs {
    const float* src[COPO_GATHER_MAX_SRC];
    float* dst[COPO_GATHER_MAX_SRC];
    int32_t width[COPO_GATHER_MAX_SRC];
    int32_t n_src;
    const int64_t* rows;
    int64_t n_rows;
};
__global__ void __launch_bounds__(256) gather_rows_kernel(GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.n_rows) return;
    const int64_t sr = a.rows[r];
#pragma unroll
    for (int s = 0; s < COPO_GATHER_MAX_SRC; ++s) {
        if (s >= a.n_src) continue;
        const int w = a.width[s];
        const float* in = a.src[s] + (size_t)sr * w;
        float* out = a.dst[s] + (size_t)r * w;
        if ((w & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.src[s]) | reinterpret_cast<uintptr_t>(a.dst[s])) & 15) == 0) {
            for (int q = lane; q < (w >> 2); q += 64) reinterpret_cast<float4*>(out)[q] = reinterpret_cast<const float4*>(in)[q];
        } else {
            for (int q = lane; q < w; q += 64) out[q] = in[q];
        }
    }
}

// pack[r][off_c .. off_c + width_c) = col_c[r][:] for up to COPO_PACK_MAX_COLS per-row columns (the [rows][17] pack of
// per-row scalars the step kernels read, from the iteration's separate [T][E][N](x w) tensors)
struct PackArgs {
    const float* col[COPO_PACK_MAX_COLS];
    int32_t width[COPO_PACK_MAX_COLS];
    int32_t n_cols, pack_width;
    float* pack;
    int64_t n_rows;
};
__global__ void __launch_bounds__(256) pack_columns_kernel(PackArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread per pack element
    const int64_t r = i / a.pack_width;
    if (r >= a.n_rows) return;
    int k = (int)(i - r * a.pack_width);
    const int k0 = k;
    for (int c = 0; c < a.n_cols; ++c) {          // (uniform trip count; the column table is read with scalar loads)
        if (k < a.width[c]) {
            a.pack[(size_t)r * a.pack_width + k0] = a.col[c][(size_t)r * a.width[c] + k];
            return;
        }
        k -= a.width[c];
    }
}

}  // namespace copo

extern "C" int copo_gather_rows_f32(const float* const* srcs, float* const* dsts, const int32_t* widths, int32_t n_src,
                                    const int64_t* rows, int64_t n_rows, void* stream) {
    if (!srcs || !dsts || !widths || !rows) return COPO_ERR_NULL;
    if (n_src < 1 || n_src > COPO_GATHER_MAX_SRC || n_rows < 0) return COPO_ERR_DIM;
    if (n_rows == 0) return COPO_OK;
    copo::GatherArgs a;
    memset(&a, 0, sizeof(a));
    for (int s = 0; s < n_src; ++s) {
        if (!srcs[s] || !dsts[s]) return COPO_ERR_NULL;
        if (widths[s] < 1) return COPO_ERR_DIM;
        a.src[s] = srcs[s]; a.dst[s] = dsts[s]; a.width[s] = widths[s];
    }
    a.n_src = n_src; a.rows = rows; a.n_rows = n_rows;
    hipLaunchKernelGGL(copo::gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_pack_columns_f32(const float* const* cols, const int32_t* widths, int32_t n_cols, int64_t n_rows, float* pack,
                                     void* stream) {
    if (!cols || !widths || !pack) return COPO_ERR_NULL;
    if (n_cols < 1 || n_cols > COPO_PACK_MAX_COLS || n_rows < 0) return COPO_ERR_DIM;
    if (n_rows == 0) return COPO_OK;
    copo::PackArgs a;
    memset(&a, 0, sizeof(a));
    int pw = 0;
    for (int c = 0; c < n_cols; ++c) {
        if (!cols[c]) return COPO_ERR_NULL;
        if (widths[c] < 1) return COPO_ERR_DIM;
        a.col[c] = cols[c]; a.width[c] = widths[c];
        pw += widths[c];
    }
    a.n_cols = n_cols; a.pack_width = pw; a.pack = pack; a.n_rows = n_rows;
    const int64_t total = n_rows * pw;
    hipLaunchKernelGGL(copo::pack_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_episode_metrics(const uint8_t* flags, const float* info, const int32_t* nbr_cnt, int64_t n_rows,
                                    double* out15, void* stream) {
    if (!flags || !info || !nbr_cnt || !out15) return COPO_ERR_NULL;
    if (n_rows < 0) return COPO_ERR_DIM;
    hipLaunchKernelGGL(copo::episode_metrics_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), flags, info,
                       nbr_cnt, n_rows, out15);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_plan_epoch(const int64_t* valid_idx, const int64_t* perm, const uint32_t* key4_host, int64_t B_local,
                               int32_t n_mb, int32_t mb, const int64_t* B_all_host, int32_t world, int64_t* rows, float* w,
                               float* denom, int64_t* mb_index, void* stream) {
    if (!rows || !w || !denom || !B_all_host || (B_local > 0 && (!valid_idx || (!perm && !key4_host)))) return COPO_ERR_NULL;
    if (B_local < 0 || B_local > 0x7fffffffLL || n_mb < 1 || mb < 1 || world < 1 || world > 16) return COPO_ERR_DIM;
    copo::PlanArgs a;
    int bits = 1;
    while ((1LL << bits) < B_local) ++bits;
    a.half = (uint32_t)((bits + 1) / 2);
    for (int k = 0; k < 4; ++k) a.key[k] = key4_host ? key4_host[k] : 0u;
    a.valid_idx = valid_idx; a.perm = perm; a.B_local = B_local; a.n_mb = n_mb; a.mb = mb; a.world = world;
    for (int r = 0; r < 16; ++r) a.B_all[r] = r < world ? B_all_host[r] : 0;
    a.rows = rows; a.w = w; a.denom = denom; a.k_index = mb_index;
    return copo::launch_plan_epoch(a, static_cast<hipStream_t>(stream)) == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// Fused minibatch learner for the CoPO / CCPPO / IPPO MLPs on gfx950.
//
// The reference runs the PPO minibatch step through torch autograd (algo_copo.py:311-424 `loss`, RLlib
// `train_one_step`, torch.optim.Adam): ~250 tiny kernels per 512-row minibatch, launch-bound on any GPU.
// Everything here works on ONE flat fp32 parameter buffer (+ a mirror with W1 / W2 transposed).
//
// Contents, in file order:
//   1. FusedArgs, group / workspace-layout helpers (WsLay), counter hand-over slots
//   2. tile-GEMM engine (gemm_tile: 64x64 tiles, K slabs through LDS, v_mfma_f32_32x32x2_f32) and its operand functors
//        FwdOpT<1|2>  h = tanh(X W^T + b)            BxOp  dz1 = (dz2 W2) (1 - h1^2)
//        BwOpT<l, GA> partial dW_l = dz^T [In | 1]   (GA: operands gathered from the meta row store)
//      -- the path of shapes without a row-pass instantiation, of the batched meta pass and of the row store
//   3. head_row_terms (PPO / value / meta loss terms + analytic gradient), head_kernel
//   4. rowpass_kernel: layers 1-2, heads, dz2, dz1 of 16 rows in one workgroup (v_mfma_f32_16x16x4_f32, weights
//      streamed through BRing register rings); mlp_fwd_kernel: its forward-only sibling (rollouts, critic heads)
//   5. LCF meta kernels: meta_lcf / meta_finish (step by step), meta_batch_fold / dot / rowstat, meta_seq (all LCF
//      Adam steps of a pass in one workgroup)
//   6. reduce_adam_kernel (fold of row-split partials + Adam; legacy two-net meta step), wgrad_adam_kernel (weight
//      gradients + fold + Adam in one kernel), adam_flat_kernel (after a gradient all-reduce), transposes
//   7. launch_fused_step and the C ABI
//
// One PPO minibatch step = rowpass_kernel + wgrad_adam_kernel (2 launches).  One LCF meta pass over n_mb minibatches =
// ceil(n_mb / 32) x [gemm_bw_kernel<GA> + fold + rowstat + dot] + meta_seq_kernel, on top of a row store computed once
// per training iteration (F1, F2, head, Bx over all rows).  DESIGN.md section 4 has the measurements.
#include <cstring>

#include "sim_common.h"

#pragma clang fp contract(fast)

namespace copo {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int TM = 64, TN = 64, TK = 64, LDP = 68;   // K slab; LDP: padded LDS row (floats)
static_assert((TK == 64 || TK == 128) && TM == 64 && TN == 64, "the slab load maps below are written for 64 x 64/128 slabs");
constexpr int HT = 16;                                // rows per workgroup of the head / row-pass kernels
constexpr int HTPR = 256 / HT;                        // threads per row of the head kernel
constexpr int MODE_META_BOTH = 3;                     // internal: group 0 = META_NEW on theta, group 1 = META_OLD on theta2

struct FusedArgs {
    copo_ppo_cfg c;
    float* theta;
    float* theta2;             // META_BOTH: target-network parameters (group 1)
    float* theta_t;            // optional mirror of theta with W1 / W2 of every net stored transposed ([in][out])
    float* adam_m;
    float* adam_v;
    float* grad;
    float* grad2;              // META_BOTH: gradient of group 1
    const float* obs_src;
    const float* cc_src;
    const float* pack_src;
    const int64_t* rows;       // [n_mb][mb]
    const float* w;            // [n_mb][mb] row weights (1 valid / 0 padding)
    const float* denom;        // [n_mb] global number of valid rows
    const float* kl_coeff;     // [1]
    const int64_t* step;       // [1] Adam step counter BEFORE this step
    float* ws;                 // workspace, layout below
    float* stats;              // [COPO_PPO_STATS] accumulated sums
    float* stats2;             // META_BOTH: statistics of group 1
    int32_t apply_adam;
    int32_t head_mode;
    int32_t groups;            // nets processed
    int32_t ksplit;            // row splits of the weight-gradient GEMMs
    const int64_t* kptr;       // [1] device minibatch index k; NULL -> 0
    int32_t bump_k;            // increment *kptr at the end of this call
    double* dot_partials;      // META_BOTH: per-workgroup partials of <g_new, g_old> (COPO_META_DOT_PARTIALS doubles)
    int32_t gcap;              // group slabs of the workspace layout (4, or `groups` of a batched meta pass)
    int32_t nreg;              // gradient regions of the workspace layout
    // meta row store: the row-local quantities of EVERY row (both policies), computed once per training iteration in
    // identity row blocks of `mb` rows; the weight-gradient GEMMs of each meta pass then gather from it
    int64_t identity_rows;     // > 0: minibatch b is the rows [b * mb, (b + 1) * mb) of `identity_rows` rows, weight 1, denom 1
    const float* ws0;          // row store (WsLay(c, gcap0, gcap0) layout) the GA weight-gradient GEMMs read
    int32_t gcap0;
    float* rowstat;            // [gcap0 * mb][2] per-row {loss term, advantage term} of the head kernel (row-store pass)
    int32_t dbg;               // timing experiments only (COPO_RP_DBG): phases of the row pass to skip
    int64_t k_first;           // minibatch index offset (batched meta pass: groups 2b, 2b+1 are minibatch k_first + b)
};

__device__ __forceinline__ bool both(const FusedArgs& a) { return a.head_mode == MODE_META_BOTH; }
// META_BOTH: even groups run the current policy, odd groups the target policy; groups 2b and 2b+1 share minibatch b
__device__ __forceinline__ bool second(const FusedArgs& a, int g) { return both(a) && (g & 1); }
__device__ __forceinline__ int64_t kb_of(const FusedArgs& a, int g) {
    return (a.kptr ? a.kptr[0] : 0) + a.k_first + (both(a) ? (g >> 1) : 0);
}
// The minibatch index is advanced without any grid-wide synchronisation: the first kernel of a step reads k = *kptr and
// one of its workgroups publishes k + 1 in a workspace slot that kernel never reads; the weight-gradient kernel reads
// that slot (minus one) and one of its workgroups copies it back to *kptr, which that kernel never reads.
__device__ __forceinline__ int64_t* knext_slot(const FusedArgs& a);
// row index / weight of entry m of the minibatch a group works on
__device__ __forceinline__ int64_t row_of(const FusedArgs& a, int64_t kb, int m) {
    return a.identity_rows > 0 ? (kb * a.c.mb + m < a.identity_rows ? kb * a.c.mb + m : 0) : a.rows[kb * a.c.mb + m];
}
__device__ __forceinline__ float weight_of(const FusedArgs& a, int64_t kb, int m) {
    return a.identity_rows > 0 ? (kb * a.c.mb + m < a.identity_rows ? 1.0f : 0.0f) : a.w[kb * a.c.mb + m] / a.denom[kb];
}
__device__ __forceinline__ const copo_net_layout& net_of(const FusedArgs& a, int g) {
    return (g == 0 || both(a)) ? a.c.pol : a.c.val[g - 1];
}
__device__ __forceinline__ float* theta_of(const FusedArgs& a, int g) { return second(a, g) ? a.theta2 : a.theta; }
__device__ __forceinline__ int mode_of(const FusedArgs& a, int g) {
    return both(a) ? ((g & 1) ? COPO_HEAD_META_OLD : COPO_HEAD_META_NEW) : a.head_mode;
}
__device__ __forceinline__ bool is_policy(const FusedArgs& a, int g) { return g == 0 || both(a); }
__device__ __forceinline__ const float* src_of(const FusedArgs& a, int g) { return is_policy(a, g) ? a.obs_src : a.cc_src; }
__device__ __forceinline__ int region_of(const FusedArgs& a, int g) { return both(a) ? g : 0; }

// workspace layout (floats): 4 activation slabs x gcap groups, head output gradients, nreg x KSPLIT gradient
// regions of n_params floats, per-tile loss statistics, one completion counter
__device__ __host__ inline int head_tiles(const copo_ppo_cfg& c) { return (c.mb + HT - 1) / HT; }
struct WsLay {
    size_t slab;       // mb * hidden
    int gcap, nreg, mb, tiles, kcap;
    size_t n_params;
    // a layout with more than 4 group slabs is a batched meta pass: no row splits, one partial per region
    __device__ __host__ WsLay(const copo_ppo_cfg& c, int gcap_, int nreg_)
        : slab((size_t)c.mb * c.hidden), gcap(gcap_), nreg(nreg_), mb(c.mb), tiles(head_tiles(c)),
          kcap(gcap_ > 4 ? 1 : COPO_PPO_MAX_KSPLIT), n_params((size_t)c.n_params) {}
    __device__ __host__ size_t h1(int g) const { return (size_t)g * slab; }
    __device__ __host__ size_t h2(int g) const { return (size_t)(gcap + g) * slab; }
    __device__ __host__ size_t dz2(int g) const { return (size_t)(2 * gcap + g) * slab; }
    __device__ __host__ size_t dz1(int g) const { return (size_t)(3 * gcap + g) * slab; }
    __device__ __host__ size_t dout(int g) const { return (size_t)4 * gcap * slab + (size_t)g * mb * 4; }      // [g][mb][4]
    __device__ __host__ size_t split(int region, int sp) const {                                           // [region][split][n_params]
        return (size_t)4 * gcap * slab + (size_t)gcap * mb * 4 + ((size_t)region * kcap + sp) * n_params;
    }
    __device__ __host__ size_t stats(int q) const { return split(nreg, 0) + (size_t)q * 8; }               // [g * tiles + tile][8]
    __device__ __host__ size_t counter() const { return stats(gcap * tiles); }
    __device__ __host__ size_t total() const { return counter() + 8; }       // counter, pad, two int64 hand-over slots
};
__device__ __forceinline__ WsLay lay(const FusedArgs& a) { return WsLay(a.c, a.gcap, a.nreg); }
__device__ __forceinline__ size_t ws_h1(const FusedArgs& a, int g) { return lay(a).h1(g); }
__device__ __forceinline__ size_t ws_h2(const FusedArgs& a, int g) { return lay(a).h2(g); }
__device__ __forceinline__ size_t ws_dz2(const FusedArgs& a, int g) { return lay(a).dz2(g); }
__device__ __forceinline__ size_t ws_dz1(const FusedArgs& a, int g) { return lay(a).dz1(g); }
__device__ __forceinline__ size_t ws_dout(const FusedArgs& a, int g) { return lay(a).dout(g); }
__device__ __forceinline__ size_t ws_split(const FusedArgs& a, int region, int sp) { return lay(a).split(region, sp); }
__device__ __forceinline__ size_t ws_stats_at(const FusedArgs& a, int q) { return lay(a).stats(q); }
__device__ __forceinline__ size_t ws_counter_at(const FusedArgs& a) { return lay(a).counter(); }
__device__ __forceinline__ int64_t* knext_slot(const FusedArgs& a) {
    return reinterpret_cast<int64_t*>(a.ws + ((ws_counter_at(a) + 2 + 1) & ~(size_t)1));      // 8-byte aligned, after the counter
}
// duties of the first kernel of a step (one thread): advance the Adam step counter, publish the next minibatch index
__device__ __forceinline__ void first_kernel_duties(const FusedArgs& a) {
    // [1]: the Adam step this minibatch will be applied with -- read by a deferred copo_adam_step_f32 (data-parallel:
    // gradient all-reduce in between), which then advances *step itself
    if (a.step) knext_slot(a)[1] = a.step[0] + 1;
    if (a.apply_adam) const_cast<int64_t*>(a.step)[0] += 1;
    knext_slot(a)[0] = (a.kptr ? a.kptr[0] : 0) + 1;
}

// ------------------------------------------------------------------------------------------------------------
// generic 64x64 tile GEMM:  C[m][n] = sum_k A(m,k) * B(k,n), K staged through LDS in slabs of 32.
// Operand functors return 4 consecutive elements along their memory-contiguous axis (float4 when aligned and
// in range, masked scalars at the edges); the next slab is prefetched into registers while the MFMAs of the
// current slab run.
// ------------------------------------------------------------------------------------------------------------
// 4 consecutive elements [off, off+4) of a row of `limit` floats, zeros beyond the limit or when !ok.
// Branch-free on purpose: a divergent branch around a load makes the compiler drain vmcnt at the join, which
// serialises the prefetch loads of a slab.  VEC requires 16-byte aligned rows and limit % 4 == 0.
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* row, int off, int limit, bool ok, int& mask) {
    // returns RAW data from a clamped (always valid) address plus a 4-bit validity mask; the caller applies the
    // mask when it stashes the registers to LDS, so that nothing consumes the load result before the MFMAs.
    float4 v;
    if (VEC) {
        const bool in = ok && (off + 4 <= limit);
        v = *reinterpret_cast<const float4*>(row + (in ? off : 0));
        mask = in ? 15 : 0;
    } else {
        const bool b0 = ok && off + 0 < limit, b1 = ok && off + 1 < limit, b2 = ok && off + 2 < limit, b3 = ok && off + 3 < limit;
        v.x = row[b0 ? off + 0 : 0];
        v.y = row[b1 ? off + 1 : 0];
        v.z = row[b2 ? off + 2 : 0];
        v.w = row[b3 ? off + 3 : 0];
        mask = (b0 ? 1 : 0) | (b1 ? 2 : 0) | (b2 ? 4 : 0) | (b3 ? 8 : 0);
    }
    return v;
}

__device__ __forceinline__ float4 apply_mask(float4 v, int mask, int one) {
    // zero the invalid lanes of a quad; `one` (0..3, or -1) marks the position of the constant-1 bias column
    v.x = (mask & 1) ? v.x : (one == 0 ? 1.0f : 0.0f);
    v.y = (mask & 2) ? v.y : (one == 1 ? 1.0f : 0.0f);
    v.z = (mask & 4) ? v.z : (one == 2 ? 1.0f : 0.0f);
    v.w = (mask & 8) ? v.w : (one == 3 ? 1.0f : 0.0f);
    return v;
}

// Per-workgroup operand context: plain scalars / global pointers in registers.  (Indexing the by-value argument
// struct with a runtime group id, or mutating it, would push it to scratch and put dependent loads in front of
// every operand fetch.)
struct GemmCtx {
    const float* __restrict__ abase;   // A operand rows
    const float* __restrict__ bbase;   // B operand rows
    const float* __restrict__ aux;     // bias (forward) / h1 (B2x)
    float* __restrict__ out;
    const int32_t* srow;               // LDS table of gathered row indices (layer 1)
    const int32_t* srow2;              // LDS table of row-store indices (gather-all mode)
    int K;          // length of the k-contiguous rows / input width
    int mb, H;
    int M, astr;    // Bw: output rows (H, or the head's out_dim) and the row stride of the dz operand
    int64_t woff, boff;
};

// LDS of a GEMM workgroup (dynamic): gathered row indices + one K slab of each operand.
//   k-contiguous operands (4 consecutive k of one row per load):  quad layout [k/4][row][4], pitch QP quads --
//     float4 stores without bank conflicts, one ds_read_b128 feeds four MFMAs;
//   row-contiguous operands (4 consecutive rows of one k per load): [k][row], pitch LDP floats.
constexpr int QP = TM + 1;
constexpr int OPER_FLOATS = (TK / 4) * QP * 4 > TK * LDP ? (TK / 4) * QP * 4 : TK * LDP;
constexpr size_t GEMM_LDS_BYTES = (size_t)(2 * COPO_PPO_MAX_MB + 2 * OPER_FLOATS) * sizeof(float);

struct GemmSmem {
    int32_t* srow;      // row index into the dense sources (-1: masked entry, gather-all mode)
    int32_t* srow2;     // row index into the meta row store (gather-all mode)
    float* As;
    float* Bs;
};

__device__ __forceinline__ GemmSmem gemm_smem() {
    extern __shared__ float4 gemm_dyn_lds[];
    GemmSmem sm;
    sm.srow = reinterpret_cast<int32_t*>(gemm_dyn_lds);
    sm.srow2 = sm.srow + COPO_PPO_MAX_MB;
    sm.As = reinterpret_cast<float*>(gemm_dyn_lds) + 2 * COPO_PPO_MAX_MB;
    sm.Bs = sm.As + OPER_FLOATS;
    return sm;
}

constexpr int NLD = TK * TM / 4 / 256;     // float4 loads per thread, operand and slab

// slab coordinates of load j of this thread.  k-contiguous: 8 lanes cover 128 contiguous bytes of a row;
// row-contiguous: 16 lanes cover 256 contiguous bytes of one k.
template <bool KC> __device__ __forceinline__ int ld_row(int tid, int j) { return KC ? (tid >> 3) + 32 * (j & 1) : (tid & 15) * 4; }
template <bool KC> __device__ __forceinline__ int ld_k(int tid, int j) { return KC ? ((tid & 7) + 8 * (j >> 1)) * 4 : (tid >> 4) + 16 * j; }
template <bool KC> __device__ __forceinline__ void stash(float* S, int tid, int j, float4 v) {
    if (KC) *reinterpret_cast<float4*>(S + (((tid & 7) + 8 * (j >> 1)) * QP + ld_row<true>(tid, j)) * 4) = v;
    else *reinterpret_cast<float4*>(S + ld_k<false>(tid, j) * LDP + ld_row<false>(tid, j)) = v;
}
// the four MFMA operand values of quad q for output row/column r of this lane
template <bool KC> __device__ __forceinline__ float4 frag(const float* S, int q, int r) {
    if (KC) return *reinterpret_cast<const float4*>(S + (q * QP + r) * 4);
    return make_float4(S[(4 * q + 0) * LDP + r], S[(4 * q + 1) * LDP + r], S[(4 * q + 2) * LDP + r], S[(4 * q + 3) * LDP + r]);
}

template <class Op, bool VEC>
__device__ __forceinline__ void gemm_tile(const FusedArgs& a, int K, int g, int split, int m0, int n0, const GemmSmem& sm) {
    int32_t* srow = sm.srow;
    float* As = sm.As;
    float* Bs = sm.Bs;
    if (Op::GATHER) {     // row indices of this minibatch once per workgroup (removes a dependent-load chain)
        const int64_t kb = kb_of(a, g);
        for (int i = threadIdx.x; i < a.c.mb; i += 256) {
            const int64_t r = row_of(a, kb, i);
            if (Op::GATHER_ALL) {
                // store row of (row r, net g & 1): block (2 (r / mb) + net) of the identity pass, entry r % mb
                const int blk = (int)(r / a.c.mb);
                sm.srow2[i] = (2 * blk + (g & 1)) * a.c.mb + (int)(r - (int64_t)blk * a.c.mb);
                srow[i] = a.w[kb * a.c.mb + i] != 0.0f ? (int32_t)r : -1;
            } else {
                srow[i] = (int32_t)r;
            }
        }
        __syncthreads();
    }
    GemmCtx c = Op::prep(a, g, split, srow);
    c.srow2 = sm.srow2;
    int kbeg = 0, kend = K;
    if (Op::SPLITS_K) {
        const int chunk = ((K + a.ksplit - 1) / a.ksplit + TK - 1) / TK * TK;
        kbeg = split * chunk;
        kend = kbeg + chunk < K ? kbeg + chunk : K;
    }
    v16f acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr bool AK = Op::A_KCONTIG, BK = Op::B_KCONTIG;
    float4 ra[NLD], rb[NLD];
    int ma[NLD], mb_[NLD], ob[NLD];
#define COPO_FETCH(k0)                                                                                              \
    do {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                           \
            ra[j] = Op::template lda4<VEC>(c, m0 + ld_row<AK>(tid, j), (k0) + ld_k<AK>(tid, j), kend, ma[j]);       \
            ob[j] = -1;                                                                                             \
            rb[j] = Op::template ldb4<VEC>(c, (k0) + ld_k<BK>(tid, j), n0 + ld_row<BK>(tid, j), kend, mb_[j], ob[j]); \
        }                                                                                                           \
    } while (0)
    if (kbeg < kend) COPO_FETCH(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        // stash the prefetched registers (masking happens here, after the loads had a whole slab to land)
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            stash<AK>(As, tid, j, apply_mask(ra[j], ma[j], -1));
            stash<BK>(Bs, tid, j, apply_mask(rb[j], mb_[j], ob[j]));
        }
        __syncthreads();
        if (k0 + TK < kend) COPO_FETCH(k0 + TK);
        const int left = kend - k0;
        const int nq = left >= TK ? TK / 4 : (left + 3) / 4;      // quads of this slab that hold data
        for (int q = 0; q < nq; q += 2) {
            const float4 av = frag<AK>(As, q + (lane >> 5), wm * 32 + (lane & 31));
            const float4 bv = frag<BK>(Bs, q + (lane >> 5), wn * 32 + (lane & 31));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
#undef COPO_FETCH
    // accumulator element j of this lane: row = rbase + 8*(j/4) + (j%4), column = col
    Op::store_tile(c, m0 + wm * 32 + (lane >> 5) * 4, n0 + wn * 32 + (lane & 31), acc);
}

template <class Op, bool VEC>
__global__ void __launch_bounds__(256) gemm_kernel(FusedArgs a, int K) {
    const GemmSmem sm = gemm_smem();
    const int G = a.groups;
    // the first kernel of an SGD step advances the Adam step counter (its only reader is the fold at the end)
    if (Op::FIRST && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) first_kernel_duties(a);
    gemm_tile<Op, VEC>(a, K, blockIdx.z % G, blockIdx.z / G, blockIdx.y * TM, blockIdx.x * TN, sm);
}

#define COPO_ACC_ROW(rbase, j) ((rbase) + ((j) >> 2) * 8 + ((j) & 3))

// ---- layer forward: Y[m][n] = tanh(sum_k X[m][k] W[n][k] + b[n]) --------------------------------------------
// tanh to ~1e-7 absolute: odd polynomial near zero (no cancellation), 1 - 2 / (exp(2x) + 1) elsewhere; ~12
// instructions instead of the libm expansion (16 per lane and layer sit on the critical path of every step)
__device__ __forceinline__ float tanh_fast(float x) {
    const float x2 = x * x;
    const float p = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
    const float r = 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f);
    return fabsf(x) < 0.3f ? p : r;
}

template <int LAYER>
struct FwdOpT {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true, SPLITS_K = false, GATHER = LAYER == 1, FIRST = LAYER == 1, GATHER_ALL = false;
    __device__ static __forceinline__ GemmCtx prep(const FusedArgs& a, int g, int, const int32_t* srow) {
        const copo_net_layout L = net_of(a, g);
        GemmCtx c;
        c.mb = a.c.mb; c.H = a.c.hidden; c.srow = srow;
        c.K = GATHER ? L.in_dim : c.H;
        c.abase = GATHER ? src_of(a, g) : a.ws + ws_h1(a, g);
        const float* th = theta_of(a, g);
        c.bbase = th + (GATHER ? L.w1 : L.w2);
        c.aux = th + (GATHER ? L.b1 : L.b2);
        c.out = a.ws + (GATHER ? ws_h1(a, g) : ws_h2(a, g));
        c.woff = c.boff = 0;
        return c;
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 lda4(const GemmCtx& c, int m, int k, int, int& mask) {   // row m, 4 k's
        const bool ok = m < c.mb;
        const int mc = ok ? m : 0;
        const int r = GATHER ? c.srow[mc] : mc;
        return ld4<VEC>(c.abase + (size_t)r * c.K, k, c.K, ok, mask);
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 ldb4(const GemmCtx& c, int k, int n, int, int& mask, int&) {   // weight row n
        const bool ok = n < c.H;
        return ld4<VEC>(c.bbase + (size_t)(ok ? n : 0) * c.K, k, c.K, ok, mask);
    }
    __device__ static __forceinline__ void store_tile(const GemmCtx& c, int rbase, int col, const v16f& acc) {
        const bool cok = col < c.H;
        const float bv = c.aux[cok ? col : 0];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = COPO_ACC_ROW(rbase, j);
            const float y = tanh_fast(acc[j] + bv);
            if (cok && m < c.mb) c.out[(size_t)m * c.H + col] = y;
        }
    }
};

// ---- B2x: dz1[m][i] = (sum_o dz2[m][o] W2[o][i]) * (1 - h1[m][i]^2) -------------------------------------------
struct BxOp {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false, SPLITS_K = false, GATHER = false, FIRST = false, GATHER_ALL = false;
    __device__ static __forceinline__ GemmCtx prep(const FusedArgs& a, int g, int, const int32_t* srow) {
        GemmCtx c;
        c.mb = a.c.mb; c.H = a.c.hidden; c.K = c.H; c.srow = srow;
        c.abase = a.ws + ws_dz2(a, g);
        c.bbase = theta_of(a, g) + net_of(a, g).w2;
        c.aux = a.ws + ws_h1(a, g);
        c.out = a.ws + ws_dz1(a, g);
        c.woff = c.boff = 0;
        return c;
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 lda4(const GemmCtx& c, int m, int k, int, int& mask) {     // dz2[m][k..k+3]
        const bool ok = m < c.mb;
        return ld4<VEC>(c.abase + (size_t)(ok ? m : 0) * c.H, k, c.H, ok, mask);
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 ldb4(const GemmCtx& c, int k, int n, int, int& mask, int&) {   // W2[k][n..n+3]
        const bool ok = k < c.H;
        return ld4<VEC>(c.bbase + (size_t)(ok ? k : 0) * c.H, n, c.H, ok, mask);
    }
    __device__ static __forceinline__ void store_tile(const GemmCtx& c, int rbase, int col, const v16f& acc) {
        const bool cok = col < c.H;
        float hv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = COPO_ACC_ROW(rbase, j);
            hv[j] = c.aux[(size_t)((cok && m < c.mb) ? m : 0) * c.H + (cok ? col : 0)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = COPO_ACC_ROW(rbase, j);
            if (cok && m < c.mb) c.out[(size_t)m * c.H + col] = acc[j] * (1.0f - hv[j] * hv[j]);
        }
    }
};

// ---- Bw: partial dW[o][i] = sum_{m in split} dz[m][o] * [In | 1][m][i]; column i == in_dim is the bias --------
// GA (gather-all): dz / h operands come from the meta row store `ws0` through the minibatch's row table instead of
// this call's own activation slabs; masked table entries (weight 0) contribute nothing.
template <int LAYER, bool GA = false>   // 3: dout x h2 ; 2: dz2 x h1 ; 1: dz1 x X (rows gathered)
struct BwOpT {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false, SPLITS_K = true, GATHER = LAYER == 1 || GA, FIRST = false,
                          GATHER_ALL = GA;
    __device__ static __forceinline__ GemmCtx prep(const FusedArgs& a, int g, int split, const int32_t* srow) {
        const copo_net_layout L = net_of(a, g);
        GemmCtx c;
        c.mb = a.c.mb; c.H = a.c.hidden; c.srow = srow;
        c.K = LAYER == 1 ? L.in_dim : c.H;
        if (GA) {
            const WsLay l0(a.c, a.gcap0, 0);
            c.abase = a.ws0 + (LAYER == 1 ? l0.dz1(0) : (LAYER == 2 ? l0.dz2(0) : l0.dout(0)));
            c.bbase = LAYER == 1 ? src_of(a, g) : a.ws0 + (LAYER == 2 ? l0.h1(0) : l0.h2(0));
        } else {
            c.abase = a.ws + (LAYER == 1 ? ws_dz1(a, g) : (LAYER == 2 ? ws_dz2(a, g) : ws_dout(a, g)));
            c.bbase = LAYER == 1 ? src_of(a, g) : a.ws + (LAYER == 2 ? ws_h1(a, g) : ws_h2(a, g));
        }
        c.aux = nullptr;
        c.woff = LAYER == 1 ? L.w1 : (LAYER == 2 ? L.w2 : L.w3);
        c.boff = LAYER == 1 ? L.b1 : (LAYER == 2 ? L.b2 : L.b3);
        c.M = LAYER == 3 ? L.out_dim : c.H;
        c.astr = LAYER == 3 ? 4 : c.H;
        c.out = a.ws + ws_split(a, region_of(a, g), split);
        return c;
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 lda4(const GemmCtx& c, int o, int m, int mend, int& mask) {   // dz[m][o..o+3]
        const int mc = m < mend ? m : 0;
        const bool ok = m < mend && (!GA || c.srow[mc] >= 0);
        const int r = GA ? c.srow2[mc] : mc;
        return ld4<VEC>(c.abase + (size_t)(ok ? r : 0) * c.astr, o, c.astr, ok, mask);
    }
    template <bool VEC>
    __device__ static __forceinline__ float4 ldb4(const GemmCtx& c, int m, int i, int mend, int& mask, int& one) {   // [In|1][m][i..]
        const int mc = m < mend ? m : 0;
        const bool ok = m < mend && (!GA || c.srow[mc] >= 0);
        const int r = !ok ? 0 : (LAYER == 1 ? c.srow[mc] : (GA ? c.srow2[mc] : mc));
        const int d = c.K - i;             // position of the constant-1 bias column inside this quad, if any
        one = (ok && d >= 0 && d < 4) ? d : -1;
        return ld4<VEC>(c.bbase + (size_t)r * c.K, i, c.K, ok, mask);
    }
    __device__ static __forceinline__ void store_tile(const GemmCtx& c, int rbase, int col, const v16f& acc) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int o = COPO_ACC_ROW(rbase, j);
            if (o < c.M && col <= c.K)
                c.out[(col == c.K) ? (size_t)c.boff + o : (size_t)c.woff + (size_t)o * c.K + col] = acc[j];
        }
    }
};

// the three weight-gradient GEMMs in one launch: column tiles [0, nx2) are layer 2, [nx2, nx2 + nx1) layer 1,
// the rest the head layer (one row tile: out_dim <= 4 rows).  Fewer launches per SGD step, a grid that covers the chip.
template <bool VEC, bool GA = false>
__global__ void __launch_bounds__(256) gemm_bw_kernel(FusedArgs a, int K, int nx2, int nx1, int ny) {
    // blockIdx.x enumerates the output tiles of the three GEMMs: nx2 * ny of layer 2, nx1 * ny of layer 1, nx2 of the head
    const GemmSmem sm = gemm_smem();
    const int G = a.groups, g = blockIdx.y % G, split = blockIdx.y / G;
    int x = blockIdx.x;
    if (x < nx2 * ny) { gemm_tile<BwOpT<2, GA>, VEC>(a, K, g, split, (x / nx2) * TM, (x % nx2) * TN, sm); return; }
    x -= nx2 * ny;
    if (x < nx1 * ny) { gemm_tile<BwOpT<1, GA>, VEC>(a, K, g, split, (x / nx1) * TM, (x % nx1) * TN, sm); return; }
    x -= nx1 * ny;
    gemm_tile<BwOpT<3, GA>, VEC>(a, K, g, split, 0, x * TN, sm);
}

// ------------------------------------------------------------------------------------------------------------
// head + loss kernel: one workgroup per (16-row tile, net); 16 threads per row
// ------------------------------------------------------------------------------------------------------------
constexpr float kLog2Pi = 1.8378770664093453f;

// loss terms of one row and their analytic gradient w.r.t. the head outputs: PPO surrogate / KL / entropy for the
// policy net (algo_copo.py:311-424), the clipped value losses for the value nets, the two meta-gradient heads.
// st: total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, adv  (already weighted by wgt = w / denom)
struct RowIn {            // the pack columns (and the KL coefficient) the loss terms of one row read
    float act0, act1, logp, adv, dist0, dist1, dist2, dist3, vpred, vtarget, klc;     // scalars: stays in registers
};

// issued early (the loads are dependent on the row index and miss L2), consumed by head_row_terms much later
__device__ __forceinline__ RowIn load_row_in(const FusedArgs& a, int g, int mode, bool policy, const float* pk) {
    const copo_ppo_cfg& c = a.c;
    RowIn ri;
    ri.act0 = ri.act1 = ri.logp = ri.adv = ri.vpred = ri.vtarget = ri.klc = 0.0f;
    ri.dist0 = ri.dist1 = ri.dist2 = ri.dist3 = 0.0f;
    if (policy) {
        ri.act0 = pk[c.col_actions];
        ri.act1 = pk[c.col_actions + 1];
        if (mode != COPO_HEAD_META_OLD) {
            ri.adv = pk[mode == COPO_HEAD_META_NEW ? c.col_meta_adv : c.col_adv];
            ri.logp = pk[c.col_logp];
            if (mode == COPO_HEAD_PPO && c.use_kl) {
                ri.dist0 = pk[c.col_dist];
                ri.dist1 = pk[c.col_dist + 1];
                ri.dist2 = pk[c.col_dist + 2];
                ri.dist3 = pk[c.col_dist + 3];
                ri.klc = a.kl_coeff[0];
            }
        }
    } else {
        ri.vpred = pk[c.col_vpred[g - 1]];
        ri.vtarget = pk[c.col_vtarget[g - 1]];
    }
    return ri;
}

__device__ __forceinline__ void head_row_terms(const FusedArgs& a, int g, int mode, bool policy, const RowIn& ri, float wgt,
                                               const float* out, float* dout, float* st) {
    const copo_ppo_cfg& c = a.c;
            if (policy) {
                const int A = c.act_dim;     // A == 2
                float logp = 0.f, ent = 0.f, kl = 0.f, z[2], sig[2];
    #pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float mu = out[j], ls = out[A + j];
                    sig[j] = expf(ls);
                    z[j] = ((j == 0 ? ri.act0 : ri.act1) - mu) / sig[j];
                    logp += -0.5f * z[j] * z[j] - ls - 0.5f * kLog2Pi;
                    ent += ls + 0.5f + 0.5f * kLog2Pi;
                }
                if (mode == COPO_HEAD_META_OLD) {       // loss = mean(logp) on the target net
                    st[0] = st[1] = wgt * logp;
    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        dout[j] = wgt * z[j] / sig[j];
                        dout[A + j] = wgt * (z[j] * z[j] - 1.0f);
                    }
                } else {
                    const float adv = ri.adv;
                    const float ratio = expf(logp - ri.logp);
                    const float s1 = adv * ratio;
                    const float rc = fminf(fmaxf(ratio, 1.0f - c.clip_param), 1.0f + c.clip_param);
                    const float s2 = adv * rc;
                    const float surr = fminf(s1, s2);
                    const bool inside = (ratio >= 1.0f - c.clip_param) && (ratio <= 1.0f + c.clip_param);
                    const float dsurr_dratio = (inside || s1 < s2) ? adv : 0.0f;
                    const float dlogp = -dsurr_dratio * ratio;      // d(-surr)/d logp
                    float dmu[2] = {0.f, 0.f}, dls[2] = {0.f, 0.f};
                    const bool ppo = mode == COPO_HEAD_PPO;
                    if (ppo && c.use_kl) {
    #pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float mup = j == 0 ? ri.dist0 : ri.dist1, lsp = j == 0 ? ri.dist2 : ri.dist3;
                            const float sp = expf(lsp), dm = mup - out[j];
                            const float q = (sp * sp + dm * dm) / (sig[j] * sig[j]);
                            kl += out[A + j] - lsp + 0.5f * q - 0.5f;
                            dmu[j] += ri.klc * (-dm / (sig[j] * sig[j]));
                            dls[j] += ri.klc * (1.0f - q);
                        }
                    }
    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        dmu[j] += dlogp * z[j] / sig[j];
                        dls[j] += dlogp * (z[j] * z[j] - 1.0f);
                        if (ppo) dls[j] += -c.entropy_coeff;
                        dout[j] = wgt * dmu[j];
                        dout[A + j] = wgt * dls[j];
                    }
                    st[1] = wgt * (-surr);
                    st[3] = wgt * kl;
                    st[4] = wgt * ent;
                    st[0] = st[1] + (ppo ? ((c.use_kl ? ri.klc * st[3] : 0.0f) - c.entropy_coeff * st[4]) : 0.0f);
                    st[7] = wgt * adv;
                }
            } else {
                const float v = out[0];
                const float vp = ri.vpred, T = ri.vtarget;
                float l, dv;
                if (c.old_value_loss) {
                    const float d1 = v - T, l1 = d1 * d1;
                    const float dc = fminf(fmaxf(v - vp, -c.vf_clip_param), c.vf_clip_param);
                    const float d2 = vp + dc - T, l2 = d2 * d2;
                    const bool pass = (v - vp >= -c.vf_clip_param) && (v - vp <= c.vf_clip_param);
                    l = fmaxf(l1, l2);
                    if (l1 > l2) dv = 2.0f * d1;
                    else if (l2 > l1) dv = pass ? 2.0f * d2 : 0.0f;
                    else dv = d1 + (pass ? d2 : 0.0f);
                } else {
                    const float d1 = v - T, l1 = d1 * d1;
                    l = fminf(fmaxf(l1, 0.0f), c.vf_clip_param);
                    dv = (l1 >= 0.0f && l1 <= c.vf_clip_param) ? 2.0f * d1 : 0.0f;
                }
                dout[0] = wgt * c.vf_loss_coeff * dv;
                st[0] = wgt * c.vf_loss_coeff * l;
                st[g == 1 ? 2 : (g == 2 ? 5 : 6)] = wgt * l;
            }
}

__global__ void __launch_bounds__(256) head_kernel(FusedArgs a) {
    extern __shared__ float lds[];
    const copo_ppo_cfg& c = a.c;
    const int H = c.hidden, g = blockIdx.y, tile = blockIdx.x, m0 = tile * HT;
    const copo_net_layout L = net_of(a, g);
    const float* theta = theta_of(a, g);
    const int mode = mode_of(a, g);
    const bool policy = is_policy(a, g);
    const int OD = L.out_dim;            // 2*act_dim for the policy net, 1 for value nets
    float* h2s = lds;                    // [HT][H+1]
    float* w3s = h2s + HT * (H + 1);     // [4][H]
    float* douts = w3s + 4 * H;          // [HT][4]
    float* red = douts + HT * 4;         // [8 stats][4 waves]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // row bookkeeping first: these dependent loads (k -> row index -> pack row) overlap the tile load below
    const int r = tid / HTPR, part = tid % HTPR;
    const int m = m0 + r;
    const int64_t kb = kb_of(a, g);
    const bool rok = m < c.mb;
    const float wgt = rok ? weight_of(a, kb, m) : 0.0f;
    const float* pk = a.pack_src + (size_t)(rok ? row_of(a, kb, m) : 0) * c.pack_width;
    const float* h2g = a.ws + ws_h2(a, g);
    // unconditional loads from clamped rows (a branch around a load drains vmcnt and serialises the tile load)
    for (int r = tid >> 6; r < HT; r += 4) {
        const bool ok = m0 + r < c.mb;
        const float* src = h2g + (size_t)(ok ? m0 + r : 0) * H;
        for (int i = lane; i < H; i += 64) {
            const float v = src[i];
            h2s[r * (H + 1) + i] = ok ? v : 0.0f;
        }
    }
    for (int q = tid; q < OD * H; q += 256) w3s[q] = theta[L.w3 + q];
    __syncthreads();
    // outputs: HTPR threads per row, each a slice of the hidden units
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < H; i += HTPR) {
        const float h = h2s[r * (H + 1) + i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < OD) out[j] += h * w3s[j * H + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < HTPR; o <<= 1) out[j] += __shfl_xor(out[j], o);
        if (j < OD) out[j] += theta[L.b3 + j];
    }
    // per-row loss terms and d(loss)/d(out)
    float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, adv
    float dout[4] = {0.f, 0.f, 0.f, 0.f};
    if (part == 0 && rok) head_row_terms(a, g, mode, policy, load_row_in(a, g, mode, policy, pk), wgt, out, dout, st);
    if (part == 0 && rok && a.rowstat) {     // row-store pass: the per-row terms the meta passes regroup by minibatch
        a.rowstat[((size_t)g * c.mb + m) * 2 + 0] = st[1];
        a.rowstat[((size_t)g * c.mb + m) * 2 + 1] = st[7];
    }
    if (part == 0) {
        // d(loss)/d(outputs): to LDS for dz2 below and to the workspace for the head's weight-gradient GEMM
        *reinterpret_cast<float4*>(douts + r * 4) = make_float4(dout[0], dout[1], dout[2], dout[3]);
        if (rok) *reinterpret_cast<float4*>(a.ws + ws_dout(a, g) + (size_t)m * 4) = make_float4(dout[0], dout[1], dout[2], dout[3]);
    }
    // statistics: wave reduce -> LDS -> one partial per workgroup and stat
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = st[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) red[k * 4 + wave] = s;
    }
    __syncthreads();
    if (tid < 8)   // per-tile partial, folded in a fixed order by reduce_adam_kernel
        a.ws[ws_stats_at(a, g * head_tiles(c) + tile) + tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    // dz2 = (dout W3) * (1 - h2^2): one hidden column per thread, W3's column in registers, dout rows are LDS
    // broadcasts.  (dW3 / db3 come out of the weight-gradient GEMM launch: BwOpT<3>.)
    float* dz2 = a.ws + ws_dz2(a, g);
    const int nr = (c.mb - m0 < HT) ? c.mb - m0 : HT;
    for (int i = tid; i < H; i += 256) {
        float wc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wc[j] = (j < OD) ? w3s[j * H + i] : 0.0f;
#pragma unroll
        for (int rr = 0; rr < HT; ++rr) {
            if (rr < nr) {
                const float4 d = *reinterpret_cast<const float4*>(douts + rr * 4);
                const float h = h2s[rr * (H + 1) + i];
                const float sx = (d.x * wc[0] + d.y * wc[1]) + (d.z * wc[2] + d.w * wc[3]);
                dz2[(size_t)(m0 + rr) * H + i] = sx * (1.0f - h * h);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// row pass: everything of an SGD step that is LOCAL TO A ROW of the minibatch -- both layer forwards, the heads
// with the loss gradients, and the activation gradients back to layer 1 -- in one kernel, one workgroup per 16 rows
// and net.  h1 / h2 / dz2 tiles stay in LDS between the phases; the weights stream from L2 straight into the MFMA
// B operands (v_mfma_f32_16x16x4_f32: lane l supplies A[l & 15][k] and B[k][l & 15] with k = l >> 4, so lane group
// l >> 4 owns one quarter of K and walks it with float4 loads).  Only the weight gradients (sums over rows)
// need the second kernel.  Replaces F1, F2, H, B2x: an SGD step is then three launches.
// ------------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

// acc[t] += A[16 x K] * W^T for the 16 output columns cb + 16 t + (l & 15); W rows are k-contiguous (forward).
// K is walked in steps of 32: lane group j = l >> 4 takes k = 32 s + 8 j + {0..7}, so the four lane groups of a row
// read one whole 128-byte line per step (every line of W is touched exactly once per workgroup).
// The weights were rewritten by the previous kernel (Adam), so the first touch of every line misses the XCD's L2:
// a ring of D steps of B operands is kept in flight (D x 8 x NT VGPRs) to cover that latency with MFMA work.
// kp = K rounded up to a multiple of 32 D; As rows hold zeros beyond K, so the over-read of W is harmless.
template <int NT, int D>
__device__ __forceinline__ void rowgemm_fwd(const float* As, int astride, const float* W, int wstride, int kp, int cb, int ln,
                                            int lj, v4f* acc) {
    const float* arow = As + ln * astride + 8 * lj;
    const float* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = W + (size_t)(cb + 16 * t + ln) * wstride + 8 * lj;
    const int ns = kp >> 5;                    // multiple of D
    v4f b[D][2][NT];
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            b[u][0][t] = *reinterpret_cast<const v4f*>(wrow[t] + 32 * u);
            b[u][1][t] = *reinterpret_cast<const v4f*>(wrow[t] + 32 * u + 4);
        }
    for (int s0 = 0; s0 < ns; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = s0 + u;
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * s), a1 = *reinterpret_cast<const float4*>(arow + 32 * s + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // keep each quad one 128-bit register tuple: without this the optimiser splits the ring's float4
                // loads into dword loads, and the 6-bit vmcnt counter saturates long before the ring is in flight
                asm volatile("" : "+v"(b[u][0][t]), "+v"(b[u][1][t]));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b[u][e >> 2][t][e & 3], acc[t], 0, 0, 0);
            }
            const int sn = s + D < ns ? s + D : ns - 1;     // refill this slot (clamped: unconditional loads)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                b[u][0][t] = *reinterpret_cast<const v4f*>(wrow[t] + 32 * sn);
                b[u][1][t] = *reinterpret_cast<const v4f*>(wrow[t] + 32 * sn + 4);
            }
        }
    }
}

// acc[t] += A[16 x K] * W for output columns cb + NT (l & 15) + t; W is [K][wstride] (backward: W2 as stored).
// Tile t of a wave holds the columns {cb + NT i + t}: a lane's NT tiles are NT adjacent columns, so its B values of one
// k are one contiguous load (float4 for NT = 4) and 16 lanes read 16 NT contiguous floats.  Steps of 16 k (lane
// group j takes k = 16 s + 4 j + {0..3}); ring of D steps in flight.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int N> struct ColVec;                 // N adjacent columns of one k as one load
template <> struct ColVec<1> { typedef float T; static __device__ __forceinline__ float get(const float& v, int) { return v; } };
template <> struct ColVec<2> { typedef v2f T; static __device__ __forceinline__ float get(const v2f& v, int i) { return v[i]; } };
template <> struct ColVec<4> { typedef v4f T; static __device__ __forceinline__ float get(const v4f& v, int i) { return v[i]; } };

// Ring of D steps of B operands for the [K][wstride]-layout GEMM.  `start` issues the first D steps' loads and can be
// called long before `run` (the loads depend only on the weights): the L2-missing first touch of the weights then
// overlaps whatever the workgroup does in between (input gather, the previous layer's epilogue, the heads).
template <int NT, int D>
struct BRing {
    static constexpr int VW = NT >= 4 ? 4 : NT, NV = NT / VW;     // vector width of a load, loads per k
    typedef typename ColVec<VW>::T vec_t;
    vec_t b[D][4][NV];
    const float* wcol;
    int wstride;
    __device__ __forceinline__ void start(const float* W, int wstride_, int cb, int ln, int lj) {
        wstride = wstride_;
        wcol = W + (size_t)(4 * lj) * wstride + cb + NT * ln;
#pragma unroll
        for (int u = 0; u < D; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < NV; ++q) b[u][e][q] = *reinterpret_cast<const vec_t*>(wcol + (size_t)(16 * u + e) * wstride + VW * q);
    }
    // acc[t] += A[16 x k] * W for output columns cb + NT (l & 15) + t; k a multiple of 16 D
    __device__ __forceinline__ void run(const float* As, int astride, int k, int ln, int lj, v4f* acc) {
        const float* arow = As + ln * astride + 4 * lj;
        const int ns = k >> 4;
        for (int s0 = 0; s0 < ns; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int s = s0 + u;
                const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * s);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
                if constexpr (VW > 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int q = 0; q < NV; ++q) asm volatile("" : "+v"(b[u][e][q]));      // one register tuple per load (see rowgemm_fwd)
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], ColVec<VW>::get(b[u][e][t / VW], t % VW), acc[t], 0, 0, 0);
                const int sn = s + D < ns ? s + D : ns - 1;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < NV; ++q) b[u][e][q] = *reinterpret_cast<const vec_t*>(wcol + (size_t)(16 * sn + e) * wstride + VW * q);
            }
        }
    }
};

template <int NT, int D>
__device__ __forceinline__ void rowgemm_bwd(const float* As, int astride, const float* W, int wstride, int k, int cb, int ln,
                                            int lj, v4f* acc) {
    BRing<NT, D> r;
    r.start(W, wstride, cb, ln, lj);
    r.run(As, astride, k, ln, lj, acc);
}

constexpr int RP_D1 = 4;      // prefetch ring depth of the layer-1 GEMM (its K is padded to 32 * RP_D1)
__device__ __host__ inline int rowpass_k1p(int in_dim) { return (in_dim + 32 * RP_D1 - 1) / (32 * RP_D1) * (32 * RP_D1); }
__device__ __host__ inline size_t rowpass_lds_floats(int H, int k1p) {
    return (size_t)HT * (k1p + 4) + (size_t)3 * HT * (H + 4) + 4 * H + HT * 4 + HT * 8;
}

__device__ unsigned long long g_rp_stamps[16];
#define RP_STAMP(i) do { if ((a.dbg & 256) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_rp_stamps[i] = wall_clock64(); } while (0)

// hidden = 16 * NT * WAVES: every wave owns NT column tiles of 16.  Two waves per SIMD (WAVES = 8) overlap one
// wave's epilogue / load stalls with the other's MFMAs; the MFMA work per SIMD is the same.
// TW: a.theta_t holds W1 / W2 transposed ([in][out]), so the forward B operands are read like the backward ones --
// 16 lanes per 64 NT contiguous bytes -- instead of one row per lane (the strided pattern costs ~2x in the TA).
template <int NT, int WAVES, bool TW>
__global__ void __launch_bounds__(64 * WAVES) rowpass_kernel(FusedArgs a) {
    extern __shared__ float4 rowpass_lds[];
    constexpr int H = 16 * NT * WAVES, HP = H + 4, TH = 64 * WAVES, TPR = TH / HT;      // TPR: threads per row (heads)
    constexpr int DF = NT >= 8 ? 2 : 4, DF2 = H / 32 < DF ? H / 32 : DF;                // prefetch ring depths
    constexpr int DB = NT >= 8 ? 4 : (H / 16 >= 8 ? 8 : H / 16);
    static_assert(H % (32 * DF2) == 0 && (H / 16) % DB == 0, "ring depths must divide the step counts");
    const copo_ppo_cfg& c = a.c;
    const int g = blockIdx.y, tile = blockIdx.x, m0 = tile * HT;
    const copo_net_layout L = net_of(a, g);
    const float* theta = theta_of(a, g);
    const int mode = mode_of(a, g);
    const bool policy = is_policy(a, g);
    const int K1 = L.in_dim, OD = L.out_dim, K1P = rowpass_k1p(K1), XP = K1P + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, lj = lane >> 4, cb = wave * (16 * NT);
    float* xs = reinterpret_cast<float*>(rowpass_lds);      // [HT][XP]   gathered input rows, zero beyond K1
    float* h1s = xs + HT * XP;                               // [HT][HP]
    float* h2s = h1s + HT * HP;                              // [HT][HP]
    float* dzs = h2s + HT * HP;                              // [HT][HP]   dz2
    float* w3s = dzs + HT * HP;                              // [4][H]
    float* douts = w3s + 4 * H;                              // [HT][4]
    float* sts = douts + HT * 4;                             // [HT][8]    per-row loss statistics
    // the first kernel of an SGD step advances the Adam step counter (its only reader is the fold at the end)
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) first_kernel_duties(a);
    RP_STAMP(0);
    // TW: the first ring of layer-1 weights is requested before anything else
    constexpr bool EARLY = NT <= 2;          // two rings in flight fit the register file only for narrow waves
    BRing<NT, DB> ring1, ring2, ring3;
    if (TW) ring1.start(a.theta_t + L.w1, H, cb, ln, lj);
    const int64_t kb = kb_of(a, g);
    // head bookkeeping: TPR threads per row; the dependent loads (k -> row index -> pack row) are issued here and
    // consumed after both layers
    const int r = tid / TPR, part = tid % TPR, m = m0 + r;
    const bool rok = m < c.mb;
    float wgt = 0.0f;
    RowIn ri;
    if (part == 0) {
        const int64_t prow = a.rows[kb * c.mb + (rok ? m : 0)];
        wgt = rok ? a.w[kb * c.mb + m] / a.denom[kb] : 0.0f;
        ri = load_row_in(a, g, mode, policy, a.pack_src + (size_t)prow * c.pack_width);
    }
    {   // input tile: rows gathered through the minibatch table, float4 per thread, zeros beyond K1 / mb
        const float* src = src_of(a, g);
        if ((K1 & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const int qn = K1P >> 2;
            for (int i = tid; i < HT * qn; i += TH) {
                const int row = i / qn, k = (i - row * qn) * 4;
                const bool ok = (m0 + row < c.mb) && (k < K1);
                const int64_t ridx = a.rows[kb * c.mb + (m0 + row < c.mb ? m0 + row : 0)];
                const float4 v = *reinterpret_cast<const float4*>(src + (size_t)ridx * K1 + (k < K1 ? k : 0));
                *reinterpret_cast<float4*>(xs + row * XP + k) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {      // rows that are not 16-byte aligned (e.g. the 91-dim observation): element-wise gather
            for (int i = tid; i < HT * K1P; i += TH) {
                const int row = i / K1P, k = i - row * K1P;
                const bool ok = (m0 + row < c.mb) && (k < K1);
                const int64_t ridx = a.rows[kb * c.mb + (m0 + row < c.mb ? m0 + row : 0)];
                const float v = src[(size_t)ridx * K1 + (k < K1 ? k : 0)];
                xs[row * XP + k] = ok ? v : 0.0f;
            }
        }
        for (int i = tid; i < OD * H; i += TH) w3s[i] = theta[L.w3 + i];
    }
    __syncthreads();
    RP_STAMP(1);
    v4f acc[NT];
    // ---- layer 1 ----
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    if (TW) {
        if (EARLY) ring2.start(a.theta_t + L.w2, H, cb, ln, lj);       // layer-2 weights travel while layer 1 computes
        ring1.run(xs, XP, K1P, ln, lj, acc);
    } else if (!(a.dbg & 1)) rowgemm_fwd<NT, DF>(xs, XP, theta + L.w1, K1, K1P, cb, ln, lj, acc);
    {
        float* h1g = a.ws + ws_h1(a, g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = TW ? cb + NT * ln + t : cb + 16 * t + ln;      // tile -> column map of the GEMM flavour used
            const float bv = theta[L.b1 + col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * lj + rr;
                const float y = tanh_fast(acc[t][rr] + bv);
                h1s[i * HP + col] = y;
                if (m0 + i < c.mb) h1g[(size_t)(m0 + i) * H + col] = y;
            }
        }
    }
    __syncthreads();
    RP_STAMP(2);
    // ---- layer 2 ----
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    if (TW) {
        if (!EARLY) ring2.start(a.theta_t + L.w2, H, cb, ln, lj);
        else ring3.start(theta + L.w2, H, cb, ln, lj);      // W2 as stored, for the activation-gradient GEMM below
        ring2.run(h1s, HP, H, ln, lj, acc);
    } else if (!(a.dbg & 2)) rowgemm_fwd<NT, DF2>(h1s, HP, theta + L.w2, H, H, cb, ln, lj, acc);
    {
        float* h2g = a.ws + ws_h2(a, g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = TW ? cb + NT * ln + t : cb + 16 * t + ln;
            const float bv = theta[L.b2 + col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * lj + rr;
                const float y = tanh_fast(acc[t][rr] + bv);
                h2s[i * HP + col] = y;
                if (m0 + i < c.mb) h2g[(size_t)(m0 + i) * H + col] = y;
            }
        }
    }
    __syncthreads();
    RP_STAMP(3);
    // ---- heads, loss terms, d(loss)/d(outputs) ----
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < H; i += TPR) {
        const float h = h2s[r * HP + i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < OD) out[j] += h * w3s[j * H + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) out[j] += __shfl_xor(out[j], o);
    }
    if (part == 0) {
        float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float dout[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] += theta[L.b3 + (j < OD ? j : 0)];
        if (rok && !(a.dbg & 8)) head_row_terms(a, g, mode, policy, ri, wgt, out, dout, st);
        *reinterpret_cast<float4*>(douts + r * 4) = make_float4(dout[0], dout[1], dout[2], dout[3]);
        if (rok) *reinterpret_cast<float4*>(a.ws + ws_dout(a, g) + (size_t)m * 4) = make_float4(dout[0], dout[1], dout[2], dout[3]);
        *reinterpret_cast<float4*>(sts + r * 8) = make_float4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<float4*>(sts + r * 8 + 4) = make_float4(st[4], st[5], st[6], st[7]);
    }
    __syncthreads();
    RP_STAMP(4);
    if (tid < 8) {       // per-tile partial of the statistics, rows in a fixed order
        float sv = 0.0f;
#pragma unroll
        for (int rr = 0; rr < HT; ++rr) sv += sts[rr * 8 + tid];
        a.ws[ws_stats_at(a, g * head_tiles(c) + tile) + tid] = sv;
    }
    // ---- dz2 = (dout W3) * (1 - h2^2) ----
    {
        float* dz2g = a.ws + ws_dz2(a, g);
        const int nr = (c.mb - m0 < HT) ? c.mb - m0 : HT;
        for (int i = tid; i < H; i += TH) {
            float wc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wc[j] = (j < OD) ? w3s[j * H + i] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < HT; ++rr) {
                const float4 d = *reinterpret_cast<const float4*>(douts + rr * 4);
                const float h = h2s[rr * HP + i];
                const float v = ((d.x * wc[0] + d.y * wc[1]) + (d.z * wc[2] + d.w * wc[3])) * (1.0f - h * h);
                dzs[rr * HP + i] = v;
                if (rr < nr) dz2g[(size_t)(m0 + rr) * H + i] = v;
            }
        }
    }
    __syncthreads();
    RP_STAMP(5);
    // ---- dz1 = (dz2 W2) * (1 - h1^2) ----
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    if (TW) {
        if (!EARLY) ring3.start(theta + L.w2, H, cb, ln, lj);
        ring3.run(dzs, HP, H, ln, lj, acc);
    }
    else if (!(a.dbg & 4)) rowgemm_bwd<NT, DB>(dzs, HP, theta + L.w2, H, H, cb, ln, lj, acc);
    {
        float* dz1g = a.ws + ws_dz1(a, g);
        // tile t of rowgemm_bwd holds the columns cb + NT * ln + t: a lane stores NT adjacent columns per row
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * lj + rr;
            if (m0 + i < c.mb) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = cb + NT * ln + t;
                    const float h = h1s[i * HP + col];
                    dz1g[(size_t)(m0 + i) * H + col] = acc[t][rr] * (1.0f - h * h);
                }
            }
        }
    }
    RP_STAMP(9);
}

// ------------------------------------------------------------------------------------------------------------
// forward-only pass of the same nets for rollouts and the dense postprocess: 16 dense rows per workgroup and net,
// both layers through the transposed weight mirror (see rowpass_kernel), heads on the vector ALUs.  Net 0 (policy)
// can sample: action = mean + exp(log_std) * eps, its log-probability and the clipped action the simulator takes.
// ------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    copo_ppo_cfg c;
    const float* theta;
    const float* theta_t;
    const float* obs_src;     // [n_rows][pol.in_dim]
    const float* cc_src;      // [n_rows][val.in_dim] (value nets)
    int64_t n_rows;
    int32_t first_net, n_nets;
    float* values;            // [n_nets][n_rows]: output 0 of value nets (slot of a policy net unused)
    float* dist_inputs;       // policy: [n_rows][4] or NULL
    const float* eps;         // policy: [n_rows][2] standard normal draws, or NULL = no sampling
    float* action;            // [n_rows][2]
    float* logp;              // [n_rows]
    float* clipped;           // [n_rows][2] or NULL
};

template <int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) mlp_fwd_kernel(FwdArgs a) {
    extern __shared__ float4 fwd_lds[];
    constexpr int H = 16 * NT * WAVES, HP = H + 4, TH = 64 * WAVES, TPR = TH / HT;
    constexpr int DB = NT >= 8 ? 4 : (H / 16 >= 8 ? 8 : H / 16);
    const copo_ppo_cfg& c = a.c;
    const int g = a.first_net + blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * HT;
    const copo_net_layout L = g == 0 ? c.pol : c.val[g - 1];
    const float* src = g == 0 ? a.obs_src : a.cc_src;
    const int K1 = L.in_dim, OD = L.out_dim, K1P = rowpass_k1p(K1), XP = K1P + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, lj = lane >> 4, cb = wave * (16 * NT);
    float* xs = reinterpret_cast<float*>(fwd_lds);     // [HT][XP]
    float* h1s = xs + HT * XP;                          // [HT][HP]
    float* h2s = h1s + HT * HP;                         // [HT][HP]
    float* w3s = h2s + HT * HP;                         // [4][H]
    BRing<NT, DB> ring;
    ring.start(a.theta_t + L.w1, H, cb, ln, lj);
    if ((K1 & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int qn = K1P >> 2;
        for (int i = tid; i < HT * qn; i += TH) {
            const int row = i / qn, k = (i - row * qn) * 4;
            const bool ok = (m0 + row < a.n_rows) && (k < K1);
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(m0 + row < a.n_rows ? m0 + row : 0) * K1 + (k < K1 ? k : 0));
            *reinterpret_cast<float4*>(xs + row * XP + k) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int i = tid; i < HT * K1P; i += TH) {
            const int row = i / K1P, k = i - row * K1P;
            const bool ok = (m0 + row < a.n_rows) && (k < K1);
            const float v = src[(size_t)(m0 + row < a.n_rows ? m0 + row : 0) * K1 + (k < K1 ? k : 0)];
            xs[row * XP + k] = ok ? v : 0.0f;
        }
    }
    for (int i = tid; i < OD * H; i += TH) w3s[i] = a.theta[L.w3 + i];
    __syncthreads();
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    ring.run(xs, XP, K1P, ln, lj, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = cb + NT * ln + t;
        const float bv = a.theta[L.b1 + col];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) h1s[(4 * lj + rr) * HP + col] = tanh_fast(acc[t][rr] + bv);
    }
    ring.start(a.theta_t + L.w2, H, cb, ln, lj);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    ring.run(h1s, HP, H, ln, lj, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = cb + NT * ln + t;
        const float bv = a.theta[L.b2 + col];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) h2s[(4 * lj + rr) * HP + col] = tanh_fast(acc[t][rr] + bv);
    }
    __syncthreads();
    const int r = tid / TPR, part = tid % TPR;
    const int64_t m = m0 + r;
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < H; i += TPR) {
        const float h = h2s[r * HP + i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < OD) out[j] += h * w3s[j * H + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) out[j] += __shfl_xor(out[j], o);
    }
    if (part != 0 || m >= a.n_rows) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] += a.theta[L.b3 + (j < OD ? j : 0)];
    if (g != 0) {
        a.values[(size_t)blockIdx.y * a.n_rows + m] = out[0];
        return;
    }
    if (a.dist_inputs) *reinterpret_cast<float4*>(a.dist_inputs + (size_t)m * 4) = make_float4(out[0], out[1], out[2], out[3]);
    if (a.eps) {       // TorchDiagGaussian.sample / logp (RLlib): mean + std * eps, -0.5 sum z^2 - sum log_std - log(2 pi)
        float act[2], lp = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sd = expf(out[2 + j]);
            act[j] = out[j] + sd * a.eps[(size_t)m * 2 + j];
            const float z = (act[j] - out[j]) / sd;
            lp += -0.5f * z * z - out[2 + j] - 0.5f * kLog2Pi;
        }
        a.action[(size_t)m * 2] = act[0];
        a.action[(size_t)m * 2 + 1] = act[1];
        a.logp[m] = lp;
        if (a.clipped) {
            a.clipped[(size_t)m * 2] = fminf(fmaxf(act[0], -1.0f), 1.0f);
            a.clipped[(size_t)m * 2 + 1] = fminf(fmaxf(act[1], -1.0f), 1.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// LCF meta update tail (fp64, like the reference's float64 lcf_parameters)
// ------------------------------------------------------------------------------------------------------------
struct MetaArgs {
    const float* pack_src;
    const int64_t* rows;
    const float* w;
    const float* denom;
    const double* eps;          // [n_mb][mb] standard normal draws of the reparameterised LCF sample
    const int64_t* kptr;
    int32_t mb, pack_width, col_adv, col_nei_adv;
    const double* lcf_param;    // [2] = {p0, p1}
    const double* raw_mean_std; // [2]
    double* tail;               // [4] = {dS/dp0, dS/dp1, S, mean(A')}
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// S = sum_i w_i ((A'_i - mu)/sigma) / D with A' = cos(phi) A_ego + sin(phi) A_nei, phi = (m + s eps) pi/2,
// m = clamp(tanh p0, +-(1-1e-6)), s = exp(clamp(p1, -20, 2))   (algo_copo.py:155-179, 283-287)
__device__ __forceinline__ void meta_lcf_body(const MetaArgs& a, double* red) {
    const int64_t kb = a.kptr ? a.kptr[0] : 0;
    const double p0 = a.lcf_param[0], p1 = a.lcf_param[1];
    const double th = tanh(p0), lim = 1.0 - 1e-6;
    const double mean = th > lim ? lim : (th < -lim ? -lim : th);
    const double dmean = (th >= -lim && th <= lim) ? (1.0 - th * th) : 0.0;
    const double p1c = p1 > 2.0 ? 2.0 : (p1 < -20.0 ? -20.0 : p1);
    const double sd = exp(p1c), dsd = (p1 >= -20.0 && p1 <= 2.0) ? sd : 0.0;
    const double half_pi = 3.14159265358979323846 / 2.0;
    const double mu = a.raw_mean_std[0], sigma = a.raw_mean_std[1];
    const double D = (double)a.denom[kb];
    double s0 = 0.0, s1 = 0.0, sS = 0.0, sA = 0.0;
    for (int m = threadIdx.x; m < a.mb; m += blockDim.x) {
        const double w = (double)a.w[kb * a.mb + m];
        if (w == 0.0) continue;
        const float* pk = a.pack_src + (size_t)a.rows[kb * a.mb + m] * a.pack_width;
        const double ego = (double)pk[a.col_adv], nei = (double)pk[a.col_nei_adv];
        const double e = a.eps[kb * a.mb + m];
        const double phi = (mean + sd * e) * half_pi;
        const double cs = cos(phi), sn = sin(phi);
        const double A = cs * ego + sn * nei;
        const double dA = (-sn * ego + cs * nei) * half_pi;
        sS += w * (A - mu) / sigma;
        sA += w * A;
        s0 += w * dA * dmean / sigma;
        s1 += w * dA * e * dsd / sigma;
    }
    const double r0 = block_sum_d(s0, red), r1 = block_sum_d(s1, red), rS = block_sum_d(sS, red), rA = block_sum_d(sA, red);
    if (threadIdx.x == 0) {
        a.tail[0] = r0 / D;
        a.tail[1] = r1 / D;
        a.tail[2] = rS / D;
        a.tail[3] = rA / D;
    }
}

struct MetaFinishArgs {
    const float* g_new;        // used when dot_partials is NULL (data-parallel path: gradients were all-reduced)
    const float* g_old;
    int64_t n;
    const double* dot_partials;
    int32_t n_partials;
    const double* tail;        // [4]
    double* lcf_param;         // [2] updated in place
    double* adam;              // [5] = {m0, m1, v0, v1, step}
    double lr;
    float* stats_new;          // fused-step statistics of the two passes (may be NULL); cleared after use
    float* stats_old;
    double* stats;             // [7] accumulated: new_loss, old_loss, S, gv*S, gv, mean A', mean global adv
    int64_t* kptr;
    int32_t bump_k;
};

__device__ __forceinline__ void meta_finish_body(const MetaFinishArgs& a, double* red) {
    double s = 0.0;
    if (a.dot_partials) {
        for (int i = threadIdx.x; i < a.n_partials; i += blockDim.x) s += a.dot_partials[i];
    } else {
        for (int64_t i = threadIdx.x; i < a.n; i += blockDim.x) s += (double)a.g_new[i] * (double)a.g_old[i];
    }
    const double gv = block_sum_d(s, red);
    if (threadIdx.x == 0) {
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        const double t = a.adam[4] + 1.0;
        const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
        for (int j = 0; j < 2; ++j) {
            const double g = gv * a.tail[j];
            double m = a.adam[j], v = a.adam[2 + j];
            m = m + (g - m) * (1.0 - b1);
            v = v * b2 + g * g * (1.0 - b2);
            a.adam[j] = m;
            a.adam[2 + j] = v;
            a.lcf_param[j] -= (a.lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps));
        }
        a.adam[4] = t;
        if (a.stats) {
            a.stats[0] += a.stats_new ? (double)a.stats_new[1] : 0.0;
            a.stats[1] += a.stats_old ? (double)a.stats_old[1] : 0.0;
            a.stats[2] += a.tail[2];
            a.stats[3] += gv * a.tail[2];
            a.stats[4] += gv;
            a.stats[5] += a.tail[3];
            a.stats[6] += a.stats_new ? (double)a.stats_new[7] : 0.0;
        }
        // the per-step statistics are consumed: clear them for the next meta step
        if (a.stats_new) for (int j = 0; j < COPO_PPO_STATS; ++j) a.stats_new[j] = 0.0f;
        if (a.stats_old) for (int j = 0; j < COPO_PPO_STATS; ++j) a.stats_old[j] = 0.0f;
        if (a.bump_k && a.kptr) a.kptr[0] += 1;
    }
}

__global__ void __launch_bounds__(1024) meta_lcf_kernel(MetaArgs a) {
    __shared__ double red[16];
    meta_lcf_body(a, red);
}

__global__ void __launch_bounds__(1024) meta_finish_kernel(MetaFinishArgs a) {
    __shared__ double red[16];
    meta_finish_body(a, red);
}

// fold the split partials in a fixed order, then Adam or gradient store.  Flat 1-D grid over the parameter
// range [lo, lo + n): every element of the flat buffer folds the same way (sum over the row splits at its own
// index); padding elements fold zeros.  Workgroup 0 also folds the head kernel's per-tile statistics.
// With `meta_tail` (single-GPU meta step) the last workgroup to finish also runs the fp64 LCF part and the LCF
// Adam step (meta_lcf_body / meta_finish_body): the whole meta step is then six launches.
constexpr int FOLD_EPT = 4;      // elements per thread of the fold (strided by the workgroup size: coalesced)

__global__ void __launch_bounds__(256) reduce_adam_kernel(FusedArgs a, int64_t lo, int n, int meta_tail, MetaArgs ml,
                                                          MetaFinishArgs mf) {
    const copo_ppo_cfg& c = a.c;
    const int tiles = head_tiles(c);
    const bool two = both(a);
    size_t idx[FOLD_EPT];
    bool ok[FOLD_EPT];
    float s0[FOLD_EPT], s1[FOLD_EPT];
    // branch-free: every element sums the row-split partials at its own index (the padding slots of those regions
    // hold zeros); all loads are issued before the first add
    float v0[FOLD_EPT][COPO_PPO_MAX_KSPLIT], v1[FOLD_EPT][COPO_PPO_MAX_KSPLIT];
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        const int e = (blockIdx.x * FOLD_EPT + u) * 256 + threadIdx.x;
        ok[u] = e < n;
        idx[u] = (size_t)lo + (ok[u] ? e : 0);
#pragma unroll
        for (int sp = 0; sp < COPO_PPO_MAX_KSPLIT; ++sp) {
            v0[u][sp] = a.ws[ws_split(a, 0, sp < a.ksplit ? sp : 0) + idx[u]];
            v1[u][sp] = two ? a.ws[ws_split(a, 1, sp < a.ksplit ? sp : 0) + idx[u]] : 0.0f;
        }
    }
    float am[FOLD_EPT], av[FOLD_EPT], th[FOLD_EPT];
    const bool adam = !two && a.apply_adam;
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        am[u] = adam ? a.adam_m[idx[u]] : 0.0f;
        av[u] = adam ? a.adam_v[idx[u]] : 0.0f;
        th[u] = adam ? a.theta[idx[u]] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        s0[u] = s1[u] = 0.0f;
#pragma unroll
        for (int sp = 0; sp < COPO_PPO_MAX_KSPLIT; ++sp) {
            s0[u] += sp < a.ksplit ? v0[u][sp] : 0.0f;
            s1[u] += sp < a.ksplit ? v1[u][sp] : 0.0f;
        }
    }
    if (blockIdx.x == 0) {
        // statistics: fixed-order sum of the head kernel's per-tile partials (deterministic, no atomics)
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nq = a.groups * tiles;
        for (int k = wave; k < 8; k += 4) {
            float t0 = 0.0f, t1 = 0.0f;
            for (int q = lane; q < nq; q += 64) {
                const float v = a.ws[ws_stats_at(a, q) + k];
                if (two && q >= tiles) t1 += v; else t0 += v;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { t0 += __shfl_down(t0, o); t1 += __shfl_down(t1, o); }
            if (lane == 0) {
                if (a.stats) a.stats[k] += t0;
                if (two && a.stats2) a.stats2[k] += t1;
            }
        }
        // nothing in this kernel reads the minibatch index: advance it here (the meta tail advances it itself)
        if (threadIdx.x == 0 && a.bump_k && a.kptr && !meta_tail) const_cast<int64_t*>(a.kptr)[0] += 1;
    }
    if (!two) {
        // the Adam step counter was advanced by the first kernel of this step (see FwdOpT<1>): step[0] = t
        const float tt = adam ? (float)a.step[0] : 1.0f;
        const float bc1 = 1.0f - powf(c.beta1, tt), bc2s = sqrtf(1.0f - powf(c.beta2, tt));
#pragma unroll
        for (int u = 0; u < FOLD_EPT; ++u) {
            if (!ok[u]) continue;
            if (adam) {
                const float g = s0[u];
                const float m = am[u] + (g - am[u]) * (1.0f - c.beta1);
                const float v = av[u] * c.beta2 + g * g * (1.0f - c.beta2);
                a.adam_m[idx[u]] = m;
                a.adam_v[idx[u]] = v;
                a.theta[idx[u]] = th[u] - (c.lr / bc1) * (m / (sqrtf(v) / bc2s + c.eps));
            } else {
                a.grad[idx[u]] = s0[u];
            }
        }
        return;
    }
    // meta pass: one thread folds BOTH gradients of its elements and contributes g_new * g_old to a per-workgroup
    // partial of the dot product (fixed order -> deterministic), consumed by meta_finish
    __shared__ double red[16];
    __shared__ int last_flag;
    double prod = 0.0;
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        if (ok[u]) {
            a.grad[idx[u]] = s0[u];
            a.grad2[idx[u]] = s1[u];
            prod += (double)s0[u] * (double)s1[u];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) prod += __shfl_down(prod, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = prod;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.dot_partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        if (meta_tail) {
            // the last workgroup to get here sees every partial and statistic of this launch
            unsigned* done = reinterpret_cast<unsigned*>(a.ws + ws_counter_at(a));
            __threadfence();
            const bool last = atomicAdd(done, 1u) == gridDim.x - 1;
            last_flag = last ? 1 : 0;
            if (last) *done = 0u;
        }
    }
    if (!meta_tail) return;
    __syncthreads();
    if (!last_flag) return;
    __threadfence();      // acquire: the other workgroups' dot partials and statistics are visible from here on
    meta_lcf_body(ml, red);
    __syncthreads();
    meta_finish_body(mf, red);
}

// ------------------------------------------------------------------------------------------------------------
// batched LCF meta pass.  The two policy gradients of `meta_update` (algo_copo.py:228-309) depend on the
// minibatch and on the (fixed) policy / target parameters only -- not on the LCF parameters the meta loop
// updates -- so the gradient pairs of MANY minibatches are computed in one grouped launch chain (groups 2b, 2b+1 =
// minibatch b; no row split: the grid is large enough), and the strictly sequential part, the fp64 LCF Adam
// steps, runs afterwards in a single workgroup (meta_seq_kernel).
// ------------------------------------------------------------------------------------------------------------
// grid (fold_blocks, nb): <g_new_b, g_old_b> partials per workgroup, optional gradient export (data-parallel path),
// loss statistics of both passes.  dot_out [nb][fold_blocks]; stats_out [nb][2][8]; g_out [nb][2][n].
__global__ void __launch_bounds__(256) meta_batch_fold_kernel(FusedArgs a, int64_t lo, int n, float* g_out, double* dot_out,
                                                              float* stats_out) {
    const int b = blockIdx.y, tiles = head_tiles(a.c);
    float s0[FOLD_EPT], s1[FOLD_EPT];
    int e[FOLD_EPT];
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        e[u] = (blockIdx.x * FOLD_EPT + u) * 256 + threadIdx.x;
        const size_t idx = (size_t)lo + (e[u] < n ? e[u] : 0);
        s0[u] = a.ws[ws_split(a, 2 * b, 0) + idx];
        s1[u] = a.ws[ws_split(a, 2 * b + 1, 0) + idx];
    }
    double prod = 0.0;
#pragma unroll
    for (int u = 0; u < FOLD_EPT; ++u) {
        if (e[u] < n) {
            prod += (double)s0[u] * (double)s1[u];
            if (g_out) {
                g_out[((size_t)b * 2 + 0) * n + e[u]] = s0[u];
                g_out[((size_t)b * 2 + 1) * n + e[u]] = s1[u];
            }
        }
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) prod += __shfl_down(prod, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = prod;
    __syncthreads();
    if (threadIdx.x == 0) dot_out[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0 && stats_out) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int k = wave; k < 8; k += 4) {
            float t0 = 0.0f, t1 = 0.0f;
            for (int q = lane; q < tiles; q += 64) {
                t0 += a.ws[ws_stats_at(a, (2 * b) * tiles + q) + k];
                t1 += a.ws[ws_stats_at(a, (2 * b + 1) * tiles + q) + k];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { t0 += __shfl_down(t0, o); t1 += __shfl_down(t1, o); }
            if (lane == 0) {
                stats_out[((size_t)b * 2 + 0) * 8 + k] = t0;
                stats_out[((size_t)b * 2 + 1) * 8 + k] = t1;
            }
        }
    }
}

// gv[b] = <g_new_b, g_old_b>: from the fold's per-workgroup partials (dot != NULL, n = partials per minibatch) or
// from exported gradients g [nb][2][n] (after a gradient all-reduce).  grid (nb), fixed summation order.
__global__ void __launch_bounds__(1024) meta_batch_dot_kernel(const double* dot, const float* g, int64_t n, double* gv,
                                                              const float* denom = nullptr) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double s = 0.0;
    if (dot) {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += dot[(size_t)b * n + i];
    } else {
        const float* g0 = g + ((size_t)b * 2 + 0) * n;
        const float* g1 = g + ((size_t)b * 2 + 1) * n;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)g0[i] * (double)g1[i];
    }
    const double t = block_sum_d(s, red);
    // row-store path: the gradients were summed with unit row weights; both carry the factor 1 / D_b
    const double sc = denom ? 1.0 / ((double)denom[b] * (double)denom[b]) : 1.0;
    if (threadIdx.x == 0) gv[b] = t * sc;
}

// per-minibatch loss statistics from the per-row terms of the row store: stats_out[b][net][1] = sum_m w rowstat[.][0] / D_b,
// stats_out[b][net][7] = sum_m w rowstat[.][1] / D_b (fixed order), everything else 0.  grid (nb), 256 threads.
__global__ void __launch_bounds__(256) meta_rowstat_kernel(FusedArgs a, int64_t k_first, const float* denom, float* stats_out) {
    __shared__ double red[16];
    const int b = blockIdx.x, mb = a.c.mb;
    const int64_t kb = k_first + b;
    double s[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int m = threadIdx.x; m < mb; m += blockDim.x) {
        if (a.w[kb * mb + m] == 0.0f) continue;
        const int64_t r = a.rows[kb * mb + m];
        const int64_t blk = r / mb, loc = r - blk * mb;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* q = a.rowstat + (((size_t)2 * blk + n) * mb + loc) * 2;
            s[n][0] += (double)q[0];
            s[n][1] += (double)q[1];
        }
    }
    const double D = (double)denom[kb];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const double t0 = block_sum_d(s[n][0], red), t1 = block_sum_d(s[n][1], red);
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            stats_out[((size_t)b * 2 + n) * 8 + k] = k == 1 ? (float)(t0 / D) : (k == 7 ? (float)(t1 / D) : 0.0f);
        }
    }
}

struct MetaSeqArgs {
    const float* pack_src;      // gather mode (ego_nei == NULL): A_ego / A_nei from pack_src rows
    const int64_t* rows;        // [n_mb][mb]
    const float* ego_nei;       // dense mode: [n_seg][n_mb][mb][2]
    const float* w;             // [n_seg][n_mb][mb]
    const double* eps;          // [n_seg][n_mb][mb]
    const float* denom;         // [n_mb]
    const double* gv;           // [n_mb]
    const float* stats_in;      // [n_mb][2][8] loss statistics of the two passes
    int32_t mb, n_mb, n_seg, pack_width, col_adv, col_nei_adv;
    double* lcf_param;          // [2]
    const double* raw_mean_std; // [2]
    double* adam;               // [5]
    double lr;
    double* stats;              // [7] accumulated
};

// all LCF Adam steps of one meta iteration, in minibatch order, in ONE workgroup: per step the fp64 row sums of
// meta_lcf_body with the current LCF parameters, then meta_finish_body's update -- parameters and Adam state live
// in LDS between steps.
constexpr int SEQ_RPT = 8;      // rows per thread whose inputs are prefetched one LCF step ahead

__global__ void __launch_bounds__(512) meta_seq_kernel(MetaSeqArgs a) {
    __shared__ double red[4][8];
    __shared__ double P[2], AD[5], ST[7];
    __shared__ double DV[4];          // quantities derived from the LCF parameters: mean, d mean / d p0, std, d std / d p1
    __shared__ double PW[2];          // beta1^t, beta2^t as running products
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const double half_pi = 3.14159265358979323846 / 2.0, lim = 1.0 - 1e-6;
    // lane j of wave 0 owns LCF parameter j: its Adam state, and the derived quantities every row needs
    auto derive = [&](int j, double p) {
        if (j == 0) {
            const double th = tanh(p);
            DV[0] = th > lim ? lim : (th < -lim ? -lim : th);
            DV[1] = (th >= -lim && th <= lim) ? (1.0 - th * th) : 0.0;
        } else {
            const double pc = p > 2.0 ? 2.0 : (p < -20.0 ? -20.0 : p);
            const double sdv = exp(pc);
            DV[2] = sdv;
            DV[3] = (p >= -20.0 && p <= 2.0) ? sdv : 0.0;
        }
    };
    if (tid < 2) {
        P[tid] = a.lcf_param[tid];
        derive(tid, P[tid]);
        PW[tid] = pow(tid == 0 ? 0.9 : 0.999, a.adam[4]);
    }
    if (tid < 5) AD[tid] = a.adam[tid];
    if (tid < 7) ST[tid] = 0.0;
    __syncthreads();
    const double mu = a.raw_mean_std[0], sigma = a.raw_mean_std[1];
    const size_t seg_stride = (size_t)a.n_mb * a.mb;
    const int total = a.n_seg * a.mb;
    // row inputs of one LCF step: {A_ego, A_nei, eps, w}; the next step's are requested before this step's math, so
    // that the (dependent, L2-missing) loads overlap the fp64 work and the reduction
    float pe[SEQ_RPT], pn[SEQ_RPT], pw[SEQ_RPT];
    double px[SEQ_RPT];
    auto fetch = [&](int k, int i, float& ego, float& nei, double& e, float& w) {
        const int seg = i / a.mb, m = i - seg * a.mb;
        const size_t at = seg * seg_stride + (size_t)k * a.mb + m;
        w = a.w[at];
        e = a.eps[at];
        if (a.ego_nei) {
            ego = a.ego_nei[at * 2];
            nei = a.ego_nei[at * 2 + 1];
        } else {
            const float* pk = a.pack_src + (size_t)a.rows[at] * a.pack_width;
            ego = pk[a.col_adv];
            nei = pk[a.col_nei_adv];
        }
    };
#pragma unroll
    for (int j = 0; j < SEQ_RPT; ++j) {
        const int i = tid + j * blockDim.x;
        pe[j] = pn[j] = pw[j] = 0.0f;
        px[j] = 0.0;
        if (i < total && a.n_mb > 0) fetch(0, i, pe[j], pn[j], px[j], pw[j]);
    }
    for (int k = 0; k < a.n_mb; ++k) {
        float ce[SEQ_RPT], cn[SEQ_RPT], cw[SEQ_RPT];
        double cx[SEQ_RPT];
#pragma unroll
        for (int j = 0; j < SEQ_RPT; ++j) { ce[j] = pe[j]; cn[j] = pn[j]; cw[j] = pw[j]; cx[j] = px[j]; }
        if (k + 1 < a.n_mb) {
#pragma unroll
            for (int j = 0; j < SEQ_RPT; ++j) {
                const int i = tid + j * blockDim.x;
                if (i < total) fetch(k + 1, i, pe[j], pn[j], px[j], pw[j]);
            }
        }
        const double mean = DV[0], sd = DV[2];
        // per row only the sums that depend on the row: sum w A', sum w dA'/dphi, sum w eps dA'/dphi; the parameter
        // dependent factors are applied once per step below
        double s0 = 0.0, s1 = 0.0, sS = 0.0, sA = 0.0;
        auto row = [&](double ego, double nei, double e, double w) {
            if (w == 0.0) return;
            double cs, sn;
            sincospi((mean + sd * e) * 0.5, &sn, &cs);        // phi = (mean + sd eps) pi / 2
            const double A = cs * ego + sn * nei;
            const double dA = cs * nei - sn * ego;
            sA += w * A;
            s0 += w * dA;
            s1 += w * dA * e;
        };
#pragma unroll
        for (int j = 0; j < SEQ_RPT; ++j)
            if (tid + j * (int)blockDim.x < total) row((double)ce[j], (double)cn[j], cx[j], (double)cw[j]);
        for (int i = tid + SEQ_RPT * blockDim.x; i < total; i += blockDim.x) {      // rows beyond the prefetch window
            float ego, nei, w;
            double e;
            fetch(k, i, ego, nei, e, w);
            row((double)ego, (double)nei, e, (double)w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_down(s0, o); s1 += __shfl_down(s1, o); sS += __shfl_down(sS, o); sA += __shfl_down(sA, o);
        }
        if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = sS; red[3][wave] = sA; }
        __syncthreads();
        if (wave == 0) {
            // final sums over the waves (fixed order), one lane per quantity, then broadcast inside the wave
            double tq = 0.0;
            if (lane < 4)
                for (int i = 0; i < nw; ++i) tq += red[lane][i];
            const double D = (double)a.denom[k];
            // tail = {dS/dp0, dS/dp1, S, mean A'} with S = (mean A' - mu) / sigma   (red[2] is unused: zero)
            const double r0 = __shfl(tq, 0), r1 = __shfl(tq, 1), r3 = __shfl(tq, 3);
            const double t3 = r3 / D, t2 = (t3 - mu) / sigma;
            const double t0 = r0 * half_pi * DV[1] / (sigma * D), t1 = r1 * half_pi * DV[3] / (sigma * D);
            const double gvk = a.gv[k];
            if (lane < 2) {          // Adam on parameter `lane` (fp64, betas 0.9 / 0.999, eps 1e-8), then what the rows need
                const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
                const double pw1 = PW[0] * b1, pw2 = PW[1] * b2;       // beta^t with t = step + 1
                const double bc1 = 1.0 - pw1, bc2 = 1.0 - pw2;
                const double g = gvk * (lane == 0 ? t0 : t1);
                double m = AD[lane], v = AD[2 + lane];
                m = m + (g - m) * (1.0 - b1);
                v = v * b2 + g * g * (1.0 - b2);
                AD[lane] = m;
                AD[2 + lane] = v;
                const double pnew = P[lane] - (a.lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps));
                P[lane] = pnew;
                derive(lane, pnew);
            }
            if (lane == 2) {
                ST[0] += (double)a.stats_in[((size_t)k * 2 + 0) * 8 + 1];
                ST[1] += (double)a.stats_in[((size_t)k * 2 + 1) * 8 + 1];
                ST[2] += t2;
                ST[3] += gvk * t2;
                ST[4] += gvk;
                ST[5] += t3;
                ST[6] += (double)a.stats_in[((size_t)k * 2 + 0) * 8 + 7];
            }
            if (lane == 3) {         // after lanes 0 / 1 read the old products (same wave: program order)
                const double n1 = PW[0] * 0.9, n2 = PW[1] * 0.999;
                PW[0] = n1;
                PW[1] = n2;
                AD[4] = AD[4] + 1.0;
            }
        }
        __syncthreads();
    }
    if (tid < 2) a.lcf_param[tid] = P[tid];
    if (tid < 5) a.adam[tid] = AD[tid];
    if (tid < 7 && a.stats) a.stats[tid] += ST[tid];
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients + Adam in one kernel (PPO / single-head modes).  One workgroup per 32 x 32 tile of dW of one
// layer and net; the minibatch rows (the K of these GEMMs) are split over the four waves INSIDE the workgroup, so the
// partial sums meet in LDS, are added in a fixed order, and the same threads apply Adam (or store the gradient):
// no partial-sum round trip through memory, no separate fold launch.  Operands stream straight from L2 into the
// MFMA registers -- lane l of v_mfma_f32_32x32x2_f32 supplies dz[m][o0 + (l & 31)] and [In | 1][m][i0 + (l & 31)] for
// m = k + (l >> 5): 32 lanes read 128 contiguous bytes, nothing is shared between waves, so LDS staging would buy
// nothing.  Workgroup (0, 0) also folds the loss statistics and hands over the next minibatch index.
// ------------------------------------------------------------------------------------------------------------
constexpr int WG_WAVES = 8;       // waves per workgroup = row splits of K inside the workgroup
constexpr int WG_RING = 32;       // k-pairs of operands in flight per wave: the whole row range of a wave at the default minibatch (a second round of loads would expose the L2-miss latency again)

// OT: 32-row output tiles per wave (tile t holds the output rows o0 + OT (l & 31) + t, so a lane's OT dz values of
// one minibatch row are one contiguous load and the [In | 1] value is shared by OT MFMAs).
#define WG_STAMP(i) do { if ((a.dbg & 512) && blockIdx.x == 5 && blockIdx.y == 0 && threadIdx.x == 0) g_rp_stamps[i] = wall_clock64(); } while (0)

template <int OT>
__global__ void __launch_bounds__(64 * WG_WAVES) wgrad_adam_kernel(FusedArgs a, int nty, int nx2, int nx1) {
    extern __shared__ float wg_red[];                        // [WG_WAVES][32 OT][33] partial tiles
    __shared__ int32_t srow[COPO_PPO_MAX_MB];
    constexpr int TH = 64 * WG_WAVES, TO = 32 * OT, EPT = TO * 32 / TH, TPRW = 32 / EPT;
    typedef typename ColVec<OT>::T avec_t;
    auto red = [&](int w, int o, int i) -> float& { return wg_red[(w * TO + o) * 33 + i]; };
    const copo_ppo_cfg& c = a.c;
    const int g = blockIdx.y, H = c.hidden;
    const copo_net_layout L = net_of(a, g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // tile decode: nx2 * nty tiles of layer 2, nx1 * nty of layer 1, nx2 of the head layer
    int x = blockIdx.x, layer, o0, i0;
    if (x < nx2 * nty) { layer = 2; o0 = (x / nx2) * TO; i0 = (x % nx2) * 32; }
    else if ((x -= nx2 * nty) < nx1 * nty) { layer = 1; o0 = (x / nx1) * TO; i0 = (x % nx1) * 32; }
    else { x -= nx1 * nty; layer = 3; o0 = 0; i0 = x * 32; }
    const int K = layer == 1 ? L.in_dim : H;                 // input width; column K of the tile space is the bias
    const int M = layer == 3 ? L.out_dim : H;                // output rows
    const int astr = layer == 3 ? 4 : H;
    const float* dz = a.ws + (layer == 1 ? ws_dz1(a, g) : (layer == 2 ? ws_dz2(a, g) : ws_dout(a, g)));
    const float* in = layer == 1 ? src_of(a, g) : a.ws + (layer == 2 ? ws_h1(a, g) : ws_h2(a, g));
    const int64_t woff = layer == 1 ? L.w1 : (layer == 2 ? L.w2 : L.w3), boff = layer == 1 ? L.b1 : (layer == 2 ? L.b2 : L.b3);
    const bool tile_live = i0 <= K;                          // layer-1 tiles beyond this net's own input width do nothing
    // Adam state of this thread's elements, requested before anything else (consumed after the GEMM)
    const int ero = tid / TPRW, erc = (tid % TPRW) * EPT;
    const int eo = o0 + ero, ei = i0 + erc;
    size_t eidx[EPT];
    bool eok[EPT];
    float em[EPT], ev[EPT], eth[EPT];
    const bool adam = a.apply_adam != 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = ei + q;
        eok[q] = tile_live && eo < M && i <= K;
        eidx[q] = !eok[q] ? (size_t)woff : (i == K ? (size_t)boff + eo : (size_t)woff + (size_t)eo * K + i);
        em[q] = adam ? a.adam_m[eidx[q]] : 0.0f;
        ev[q] = adam ? a.adam_v[eidx[q]] : 0.0f;
        eth[q] = adam ? a.theta[eidx[q]] : 0.0f;
    }
    WG_STAMP(0);
    const int64_t kb = knext_slot(a)[0] - 1 + a.k_first;      // published by the first kernel of this step
    if (layer == 1) {
        for (int i = tid; i < c.mb; i += TH) srow[i] = (int32_t)a.rows[kb * c.mb + i];
        __syncthreads();
    }
    v16f acc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
    if (tile_live) {
        // rows of this wave: [kbeg, kend), walked two at a time (lane half lh takes row k + lh)
        const int chunk = (((c.mb + WG_WAVES - 1) / WG_WAVES) + 1) & ~1;
        const int kbeg = wave * chunk, kend = (kbeg + chunk < c.mb) ? kbeg + chunk : c.mb;
        const int ns = (kend - kbeg + 1) >> 1;               // k-pairs (may be <= 0)
        const int ao = o0 + OT * li, bi = i0 + li;
        const bool a_any = ao < M, b_in = bi < K;
        bool a_col[OT];
#pragma unroll
        for (int t = 0; t < OT; ++t) a_col[t] = ao + t < M;
        const float b_fill = bi == K ? 1.0f : 0.0f;          // the constant-1 bias column / zero padding
        // running fetch state: fetch number f reads row kbeg + 2 f + lh (rows past kend re-read row kend - 1, masked at use)
        const bool gather = layer == 1;
        const int mlast = kend - 1;
        int fm = kbeg + lh;
        uint32_t fa = (uint32_t)fm * (uint32_t)astr + (a_any ? ao : 0);
        uint32_t fb = (uint32_t)fm * (uint32_t)K + (b_in ? bi : 0);
        const uint32_t fa_last = (uint32_t)mlast * (uint32_t)astr + (a_any ? ao : 0), fb_last = (uint32_t)mlast * (uint32_t)K + (b_in ? bi : 0);
        const uint32_t bcol = b_in ? bi : 0;
        avec_t ra[WG_RING];
        float rb[WG_RING];
        bool rk[WG_RING];
#define WG_FETCH(u)                                                                        \
        do {                                                                              \
            rk[u] = fm <= mlast;                                                          \
            ra[u] = *reinterpret_cast<const avec_t*>(dz + (rk[u] ? fa : fa_last));        \
            rb[u] = in[gather ? (uint32_t)srow[rk[u] ? fm : mlast] * (uint32_t)K + bcol : (rk[u] ? fb : fb_last)]; \
            fm += 2; fa += 2u * (uint32_t)astr; fb += 2u * (uint32_t)K;                   \
        } while (0)
        if (ns > 0) {
#pragma unroll
            for (int u = 0; u < WG_RING; ++u) WG_FETCH(u);
            WG_STAMP(1);
            for (int s0 = 0; s0 < ns; s0 += WG_RING) {
                const bool more = s0 + WG_RING < ns;         // uniform: only long row ranges loop
#pragma unroll
                for (int u = 0; u < WG_RING; ++u) {
                    // masking happens at use: rows beyond the range / columns beyond the tensors contribute zero
                    const float bv = b_in ? rb[u] : b_fill;
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        const float av = (rk[u] && a_col[t]) ? ColVec<OT>::get(ra[u], t) : 0.0f;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                    if (more) WG_FETCH(u);
                }
            }
        }
#undef WG_FETCH
    }
    WG_STAMP(2);
    // wave partials -> LDS; accumulator element j of tile t: output row OT (8 (j / 4) + 4 lh + (j % 4)) + t, column li
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) red(wave, OT * COPO_ACC_ROW(4 * lh, j) + t, li) = acc[t][j];
    __shared__ float bcs[2];
    if (tid == 0) {      // Adam bias corrections once per workgroup (two powf are ~200 instructions)
        const float tt = adam ? (float)a.step[0] : 1.0f;     // advanced by the first kernel of this step
        bcs[0] = 1.0f - powf(c.beta1, tt);
        bcs[1] = sqrtf(1.0f - powf(c.beta2, tt));
    }
    __syncthreads();
    WG_STAMP(3);
    {
        const float bc1 = bcs[0], bc2s = bcs[1];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            if (!eok[q]) continue;
            float gsum = red(0, ero, erc + q);
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) gsum += red(w, ero, erc + q);
            if (adam) {
                const float m = em[q] + (gsum - em[q]) * (1.0f - c.beta1);
                const float v = ev[q] * c.beta2 + gsum * gsum * (1.0f - c.beta2);
                a.adam_m[eidx[q]] = m;
                a.adam_v[eidx[q]] = v;
                eth[q] = eth[q] - (c.lr / bc1) * (m / (sqrtf(v) / bc2s + c.eps));
                a.theta[eidx[q]] = eth[q];
            } else {
                a.grad[eidx[q]] = gsum;
            }
        }
    }
    WG_STAMP(4);
    if (adam && a.theta_t) {
        // keep the transposed mirror current: the tile goes back through LDS so that the [in][out] rows are written
        // contiguously; biases and the head layer are mirrored as they are
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) red(0, ero, erc + q) = eth[q];
        __syncthreads();
        for (int e = tid; e < TO * 32; e += TH) {
            const int ti = e / TO, to = e - ti * TO;         // consecutive threads -> consecutive output rows o
            const int o = o0 + to, i = i0 + ti;
            if (!tile_live || o >= M || i > K) continue;
            const float v = red(0, to, ti);
            if (i == K) a.theta_t[(size_t)boff + o] = v;
            else if (layer == 3) a.theta_t[(size_t)woff + (size_t)o * K + i] = v;
            else a.theta_t[(size_t)woff + (size_t)i * M + o] = v;
        }
    }
    WG_STAMP(5);
    // workgroup (0, 0) folds the per-tile loss statistics of the previous kernel in a fixed order and hands the next
    // minibatch index back to *kptr (which no workgroup of this kernel reads)
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    const int tiles = head_tiles(c), nq = a.groups * tiles;
    for (int k = wave; k < 8; k += WG_WAVES) {
        float t0 = 0.0f;
        for (int q = lane; q < nq; q += 64) t0 += a.ws[ws_stats_at(a, q) + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t0 += __shfl_down(t0, o);
        if (lane == 0 && a.stats) a.stats[k] += t0;
    }
    if (tid == 0 && a.bump_k && a.kptr) const_cast<int64_t*>(a.kptr)[0] = knext_slot(a)[0];
}

__global__ void bump_kernel(int64_t* step, int64_t* k) {
    if (step) step[0] += 1;
    if (k) k[0] += 1;
}

// theta_t[w + k * H + n] = theta[w + n * K + k] for one [H][K] weight matrix at offset w (32 x 32 tiles through LDS)
__global__ void __launch_bounds__(256) transpose_weight_kernel(const float* theta, float* theta_t, int64_t w, int H, int K) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < H && k0 + tx < K) tile[r][tx] = theta[w + (size_t)(n0 + r) * K + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K && n0 + tx < H) theta_t[w + (size_t)(k0 + r) * H + n0 + tx] = tile[tx][r];
}

hipError_t launch_refresh_transposed(const copo_ppo_cfg& c, const float* theta, float* theta_t, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(theta_t, theta, (size_t)c.n_params * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    const copo_net_layout* nets[4] = {&c.pol, &c.val[0], &c.val[1], &c.val[2]};
    for (int g = 0; g <= c.n_value_heads; ++g) {
        const int H = c.hidden, K1 = nets[g]->in_dim;
        hipLaunchKernelGGL(transpose_weight_kernel, dim3((K1 + 31) / 32, (H + 31) / 32), dim3(256), 0, s, theta, theta_t, nets[g]->w1, H, K1);
        hipLaunchKernelGGL(transpose_weight_kernel, dim3((H + 31) / 32, (H + 31) / 32), dim3(256), 0, s, theta, theta_t, nets[g]->w2, H, H);
    }
    return hipGetLastError();
}

// slots != 0: the step number and the next minibatch index were published by the gradient pass (first_kernel_duties);
// workgroup 0 hands them to *step / *kptr, which no workgroup of this kernel reads -> no trailing 1-thread launch.
__global__ void __launch_bounds__(256) adam_flat_kernel(FusedArgs a, long long n, int slots) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const copo_ppo_cfg& c = a.c;
    const int64_t tnext = slots ? knext_slot(a)[1] : a.step[0] + 1;
    if (slots && i == 0) {
        const_cast<int64_t*>(a.step)[0] = tnext;
        if (a.bump_k && a.kptr) const_cast<int64_t*>(a.kptr)[0] = knext_slot(a)[0];
    }
    if (i >= n) return;
    const float tt = (float)tnext;
    const float bc1 = 1.0f - powf(c.beta1, tt), bc2s = sqrtf(1.0f - powf(c.beta2, tt));
    const float s = a.grad[i];
    float m = a.adam_m[i], v = a.adam_v[i];
    m = m + (s - m) * (1.0f - c.beta1);
    v = v * c.beta2 + s * s * (1.0f - c.beta2);
    a.adam_m[i] = m;
    a.adam_v[i] = v;
    const float th = a.theta[i] - (c.lr / bc1) * (m / (sqrtf(v) / bc2s + c.eps));
    a.theta[i] = th;
    if (a.theta_t) {     // mirror with W1 / W2 transposed (scattered 4-byte writes: the whole buffer is ~1 MB)
        long long j = i;
        const uint32_t H = (uint32_t)c.hidden;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g <= c.n_value_heads) {
                const copo_net_layout& L = g == 0 ? c.pol : c.val[g - 1];
                const long long r1 = i - L.w1, r2 = i - L.w2;
                if (r1 >= 0 && r1 < (long long)H * L.in_dim) {
                    const uint32_t q = (uint32_t)r1 / (uint32_t)L.in_dim;
                    j = L.w1 + (long long)(((uint32_t)r1 - q * (uint32_t)L.in_dim) * H + q);
                }
                if (r2 >= 0 && r2 < (long long)H * H) {
                    const uint32_t q = (uint32_t)r2 / H;
                    j = L.w2 + (long long)(((uint32_t)r2 - q * H) * H + q);
                }
            }
        }
        a.theta_t[j] = th;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------
size_t fused_ws_floats(const copo_ppo_cfg& c) {
    return WsLay(c, 4, 2).total();
}

static int pick_ksplit(int mb) {
    int s = mb / 128;
    if (s < 1) s = 1;
    if (s > COPO_PPO_MAX_KSPLIT) s = COPO_PPO_MAX_KSPLIT;
    return s;
}

struct MetaTail { MetaArgs lcf; MetaFinishArgs fin; };

// the GEMM workgroups use more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
template <class F> static hipError_t allow_big_lds(F* f) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
}
static hipError_t gemm_lds_attrs() {
    static hipError_t once = [] {
        hipError_t e = hipSuccess, r;
#define COPO_ATTR(K) if ((r = allow_big_lds(K)) != hipSuccess) e = r
        COPO_ATTR((gemm_kernel<FwdOpT<1>, true>)); COPO_ATTR((gemm_kernel<FwdOpT<1>, false>));
        COPO_ATTR((gemm_kernel<FwdOpT<2>, true>)); COPO_ATTR((gemm_kernel<FwdOpT<2>, false>));
        COPO_ATTR((gemm_kernel<BxOp, true>)); COPO_ATTR((gemm_kernel<BxOp, false>));
        COPO_ATTR((gemm_bw_kernel<true>)); COPO_ATTR((gemm_bw_kernel<false>));
        COPO_ATTR((gemm_bw_kernel<true, true>)); COPO_ATTR((gemm_bw_kernel<false, true>));
#undef COPO_ATTR
        for (const void* f : {reinterpret_cast<const void*>(rowpass_kernel<1, 4, false>), reinterpret_cast<const void*>(rowpass_kernel<2, 4, false>),
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, false>), reinterpret_cast<const void*>(rowpass_kernel<4, 8, false>),
                              reinterpret_cast<const void*>(rowpass_kernel<1, 4, true>), reinterpret_cast<const void*>(rowpass_kernel<2, 4, true>),
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, true>), reinterpret_cast<const void*>(rowpass_kernel<4, 8, true>)})
            if ((r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess) e = r;
        for (const void* f : {reinterpret_cast<const void*>(mlp_fwd_kernel<1, 4>), reinterpret_cast<const void*>(mlp_fwd_kernel<2, 4>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<2, 8>), reinterpret_cast<const void*>(mlp_fwd_kernel<4, 8>)})
            if ((r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess) e = r;
        if ((r = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_adam_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) e = r;
        return e;
    }();
    return once;
}

// part: 0 = the whole chain; 1 = row-local kernels only (fill the meta row store); 2 = weight-gradient GEMMs that
// gather from the row store + fold (a meta pass over precomputed rows)
struct MetaBatch { int nb; float* g_out; double* dot_out; float* stats_out; int part; };

// COPO_FUSED_ROWPASS=0 keeps the four-kernel activation path (A/B measurements, tests of both paths)
static const int g_wgrad_ot = [] {          // COPO_WGRAD_OT=2: two 32-row output tiles per wave (measured equal at H = 256)
    const char* e = getenv("COPO_WGRAD_OT");
    return (e && e[0] == '2') ? 2 : 1;
}();
static const bool g_use_wgrad = [] {
    const char* e = getenv("COPO_FUSED_WGRAD");
    return !(e && e[0] == '0');
}();
static const bool g_use_rowpass = [] {
    const char* e = getenv("COPO_FUSED_ROWPASS");
    return !(e && e[0] == '0');
}();

// [lo, lo + n): the span of the flat parameter buffer that the first `nets` networks of the layout occupy
static void fold_range(const copo_ppo_cfg& c, int nets_n, int64_t* lo_out, int* n_out) {
    const copo_net_layout* nets[4] = {&c.pol, &c.val[0], &c.val[1], &c.val[2]};
    int64_t lo = c.pol.w1, hi = 0;
    for (int g = 0; g < nets_n; ++g) {
        const int64_t offs[6] = {nets[g]->w1, nets[g]->b1, nets[g]->w2, nets[g]->b2, nets[g]->w3, nets[g]->b3};
        const int64_t szs[6] = {(int64_t)c.hidden * nets[g]->in_dim, c.hidden, (int64_t)c.hidden * c.hidden, c.hidden,
                                (int64_t)nets[g]->out_dim * c.hidden, nets[g]->out_dim};
        for (int t = 0; t < 6; ++t) {
            lo = offs[t] < lo ? offs[t] : lo;
            hi = offs[t] + szs[t] > hi ? offs[t] + szs[t] : hi;
        }
    }
    *lo_out = lo;
    *n_out = (int)(hi - lo);
}

hipError_t launch_fused_step(FusedArgs a, hipStream_t s, const MetaTail* mt_ = nullptr, const MetaBatch* mbatch = nullptr,
                             int* fold_blocks_out = nullptr, int* n_fold_out = nullptr) {
    const copo_ppo_cfg& c = a.c;
    if (hipError_t e = gemm_lds_attrs(); e != hipSuccess) return e;
    a.ksplit = mbatch ? 1 : pick_ksplit(c.mb);
    { static const int dbg = [] { const char* e = getenv("COPO_RP_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }     // a batched pass fills the chip without splitting rows
    const int G = a.groups, mt = (c.mb + TM - 1) / TM, ht = (c.hidden + TN - 1) / TN;
    int kmax1 = c.pol.in_dim;
    if (a.head_mode == COPO_HEAD_PPO)
        for (int g = 1; g < G; ++g) kmax1 = c.val[g - 1].in_dim > kmax1 ? c.val[g - 1].in_dim : kmax1;
    // vector path: every row the kernels touch is 16-byte aligned and a multiple of 4 floats long
    bool vec = (c.hidden % 4 == 0) && (c.pol.in_dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.obs_src) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(a.cc_src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.theta) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(a.ws) & 15) == 0) && (!a.theta2 || (reinterpret_cast<uintptr_t>(a.theta2) & 15) == 0);
    const copo_net_layout* nets[4] = {&c.pol, &c.val[0], &c.val[1], &c.val[2]};
    for (int g = 0; g < (a.head_mode == COPO_HEAD_PPO ? G : 1); ++g)
        vec = vec && (nets[g]->in_dim % 4 == 0) && (nets[g]->w1 % 4 == 0) && (nets[g]->w2 % 4 == 0);
#define COPO_GEMM(OP, grid, op, K)                                                                     \
    do {                                                                                               \
        if (vec) hipLaunchKernelGGL((gemm_kernel<OP, true>), grid, dim3(256), GEMM_LDS_BYTES, s, a, K);   \
        else hipLaunchKernelGGL((gemm_kernel<OP, false>), grid, dim3(256), GEMM_LDS_BYTES, s, a, K);      \
    } while (0)
    // row pass (one kernel for F1, F2, heads, B2x) when the shapes allow it; else the four separate kernels
    // shapes with a row-pass instantiation: hidden = 16 * NT * WAVES
    const int rp_h = c.hidden;
    const size_t rp_lds = rowpass_lds_floats(c.hidden, rowpass_k1p(kmax1)) * sizeof(float);
    // with the transposed mirror the row pass reads no k-contiguous weight rows, so input widths that are not a
    // multiple of 4 (the 91-dim observation of IPPO / CCPPO) only cost a scalar input gather
    const bool tw = a.theta_t != nullptr && a.head_mode != MODE_META_BOTH;
    bool tw_ok = tw && (c.hidden % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.theta) & 15) == 0) &&
                 ((reinterpret_cast<uintptr_t>(a.theta_t) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.ws) & 15) == 0);
    for (int g = 0; g < (a.head_mode == COPO_HEAD_PPO ? G : 1); ++g)
        tw_ok = tw_ok && (nets[g]->w1 % 4 == 0) && (nets[g]->w2 % 4 == 0);
    const int part = mbatch ? mbatch->part : 0;
    if (part == 2) vec = vec && ((reinterpret_cast<uintptr_t>(a.ws0) & 15) == 0);
    const bool rowpass = (vec || tw_ok) && !mbatch && g_use_rowpass && (rp_h == 64 || rp_h == 128 || rp_h == 256 || rp_h == 512) &&
                         rp_lds <= 150 * 1024;
    if (rowpass) {
        const dim3 grid(head_tiles(c), G);
#define COPO_RP(NT_, W_)                                                                                       \
        do {                                                                                                   \
            if (tw) hipLaunchKernelGGL((rowpass_kernel<NT_, W_, true>), grid, dim3(64 * W_), rp_lds, s, a);      \
            else hipLaunchKernelGGL((rowpass_kernel<NT_, W_, false>), grid, dim3(64 * W_), rp_lds, s, a);        \
        } while (0)
        switch (rp_h) {
            case 64: COPO_RP(1, 4); break;
            case 128: COPO_RP(2, 4); break;
            case 256: COPO_RP(2, 8); break;
            default: COPO_RP(4, 8); break;
        }
#undef COPO_RP
    } else if (part != 2) {
        COPO_GEMM(FwdOpT<1>, dim3(ht, mt, G), f1, kmax1);
        COPO_GEMM(FwdOpT<2>, dim3(ht, mt, G), f2, c.hidden);
        const size_t lds = (size_t)(HT * (c.hidden + 1) + 4 * c.hidden + HT * 4 + 32) * sizeof(float);
        hipLaunchKernelGGL(head_kernel, dim3(head_tiles(c), G), dim3(256), lds, s, a);
        COPO_GEMM(BxOp, dim3(ht, mt, G), bx, c.hidden);
    }
#undef COPO_GEMM
    if (part == 1) return hipGetLastError();
    if (!mbatch && a.head_mode != MODE_META_BOTH && g_use_wgrad) {
        // weight gradients, fold and Adam in one kernel: the SGD step ends here
        const int ot = g_wgrad_ot;
        const int nty = (c.hidden + 32 * ot - 1) / (32 * ot), wx2 = (c.hidden + 1 + 31) / 32, wx1 = (kmax1 + 1 + 31) / 32;
        const dim3 grid((wx2 + wx1) * nty + wx2, G);
        const size_t lds = (size_t)WG_WAVES * 32 * ot * 33 * sizeof(float);
        if (ot == 2) hipLaunchKernelGGL((wgrad_adam_kernel<2>), grid, dim3(64 * WG_WAVES), lds, s, a, nty, wx2, wx1);
        else hipLaunchKernelGGL((wgrad_adam_kernel<1>), grid, dim3(64 * WG_WAVES), lds, s, a, nty, wx2, wx1);
        return hipGetLastError();
    }
    {
        const int nx2 = (c.hidden + 1 + TN - 1) / TN, nx1 = (kmax1 + 1 + TN - 1) / TN;
        const dim3 grid((nx2 + nx1) * ht + nx2, G * a.ksplit);
        if (part == 2) {
            if (vec) hipLaunchKernelGGL((gemm_bw_kernel<true, true>), grid, dim3(256), GEMM_LDS_BYTES, s, a, c.mb, nx2, nx1, ht);
            else hipLaunchKernelGGL((gemm_bw_kernel<false, true>), grid, dim3(256), GEMM_LDS_BYTES, s, a, c.mb, nx2, nx1, ht);
        } else if (vec) hipLaunchKernelGGL((gemm_bw_kernel<true>), grid, dim3(256), GEMM_LDS_BYTES, s, a, c.mb, nx2, nx1, ht);
        else hipLaunchKernelGGL((gemm_bw_kernel<false>), grid, dim3(256), GEMM_LDS_BYTES, s, a, c.mb, nx2, nx1, ht);
    }
    // parameter range the fold covers: every net of this call (the policy only in the meta modes)
    int64_t lo;
    int n_fold;
    fold_range(c, a.head_mode == COPO_HEAD_PPO ? G : 1, &lo, &n_fold);
    const int fold_blocks = (n_fold + 256 * FOLD_EPT - 1) / (256 * FOLD_EPT);
    if (a.head_mode == MODE_META_BOTH && fold_blocks > COPO_META_DOT_PARTIALS) return hipErrorInvalidValue;
    if (fold_blocks_out) *fold_blocks_out = fold_blocks;
    if (n_fold_out) *n_fold_out = n_fold;
    if (mbatch) {
        hipLaunchKernelGGL(meta_batch_fold_kernel, dim3(fold_blocks, mbatch->nb), dim3(256), 0, s, a, lo, n_fold, mbatch->g_out,
                           mbatch->dot_out, mbatch->stats_out);
        return hipGetLastError();
    }
    MetaTail tailv{};
    if (mt_) {
        tailv = *mt_;
        tailv.fin.n_partials = fold_blocks;      // only the partials this launch writes
    }
    hipLaunchKernelGGL(reduce_adam_kernel, dim3(fold_blocks), dim3(256), 0, s, a, lo, n_fold, mt_ ? 1 : 0, tailv.lcf, tailv.fin);
    return hipGetLastError();
}

hipError_t launch_adam_flat(const FusedArgs& a, long long n, hipStream_t s) {
    const int slots = a.ws != nullptr;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n, slots);
    if (!slots)
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, const_cast<int64_t*>(a.step),
                           a.bump_k ? const_cast<int64_t*>(a.kptr) : nullptr);
    return hipGetLastError();
}

}  // namespace copo

// ---- C ABI ---------------------------------------------------------------------------------------------------
using namespace copo;

extern "C" int copo_debug_rowpass_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(copo::g_rp_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int64_t copo_ppo_workspace_floats(const copo_ppo_cfg* cfg) { return cfg ? (int64_t)fused_ws_floats(*cfg) : -1; }

static int check_cfg(const copo_ppo_cfg* c) {
    if (!c) return COPO_ERR_NULL;
    if (c->mb < 1 || c->mb > COPO_PPO_MAX_MB || c->hidden < 1 || c->hidden > 1024 || c->act_dim != 2 ||
        c->n_value_heads < 0 || c->n_value_heads > 3 || c->n_params < 1)
        return COPO_ERR_DIM;
    if (c->pol.out_dim != 4) return COPO_ERR_DIM;
    for (int g = 0; g < c->n_value_heads; ++g)
        if (c->val[g].out_dim != 1) return COPO_ERR_DIM;
    return COPO_OK;
}

static void fill_common(FusedArgs& a, const copo_ppo_cfg* cfg, const float* obs_src, const float* cc_src,
                        const float* pack_src, const int64_t* rows, const float* w, const float* denom, float* workspace,
                        int64_t* mb_index) {
    memset(&a, 0, sizeof(a));
    a.c = *cfg;
    a.obs_src = obs_src; a.cc_src = cc_src ? cc_src : obs_src; a.pack_src = pack_src;
    a.rows = rows; a.w = w; a.denom = denom; a.ws = workspace; a.kptr = mb_index;
    a.gcap = 4; a.nreg = 2;
}

extern "C" int copo_ppo_fused_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, float* grad,
                                       const float* obs_src, const float* cc_src, const float* pack_src,
                                       const int64_t* rows, const float* w, const float* denom, const float* kl_coeff,
                                       int64_t* step, float* workspace, float* stats, int32_t apply_adam,
                                       int32_t head_mode, int64_t* mb_index, int32_t bump_index, float* theta_t,
                                       void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !obs_src || !pack_src || !rows || !w || !denom || !workspace) return COPO_ERR_NULL;
    if (apply_adam && (!adam_m || !adam_v || !step)) return COPO_ERR_NULL;
    if (!apply_adam && !grad) return COPO_ERR_NULL;
    if (head_mode < COPO_HEAD_PPO || head_mode > COPO_HEAD_META_OLD) return COPO_ERR_DIM;
    if (head_mode == COPO_HEAD_PPO && cfg->use_kl && !kl_coeff) return COPO_ERR_NULL;
    FusedArgs a;
    fill_common(a, cfg, obs_src, cc_src, pack_src, rows, w, denom, workspace, mb_index);
    a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v; a.grad = grad;
    a.kl_coeff = kl_coeff; a.step = step; a.stats = stats; a.apply_adam = apply_adam; a.head_mode = head_mode;
    a.groups = (head_mode == COPO_HEAD_PPO) ? 1 + cfg->n_value_heads : 1;
    a.bump_k = (mb_index && bump_index) ? 1 : 0;
    a.theta_t = theta_t;
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_mlp_forward_f32(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t, const float* obs_src,
                                    const float* cc_src, int64_t n_rows, int32_t first_net, int32_t n_nets, float* values,
                                    float* dist_inputs, const float* eps, float* action, float* logp, float* clipped,
                                    void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_t || !obs_src) return COPO_ERR_NULL;
    if (n_rows < 1 || first_net < 0 || n_nets < 1 || first_net + n_nets > 1 + cfg->n_value_heads) return COPO_ERR_DIM;
    if (first_net + n_nets > 1 && !values) return COPO_ERR_NULL;
    if (first_net == 0 && eps && (!action || !logp)) return COPO_ERR_NULL;
    const int H = cfg->hidden;
    if (!(H == 64 || H == 128 || H == 256 || H == 512) || (reinterpret_cast<uintptr_t>(theta_t) & 15) != 0) return COPO_ERR_DIM;
    if (gemm_lds_attrs() != hipSuccess) return COPO_ERR_DEVICE;
    int kmax = 0;
    const copo_net_layout* nets[4] = {&cfg->pol, &cfg->val[0], &cfg->val[1], &cfg->val[2]};
    for (int g = first_net; g < first_net + n_nets; ++g) {
        kmax = nets[g]->in_dim > kmax ? nets[g]->in_dim : kmax;
        if (nets[g]->w1 % 4 != 0 || nets[g]->w2 % 4 != 0) return COPO_ERR_DIM;
    }
    const size_t lds = ((size_t)HT * (rowpass_k1p(kmax) + 4) + (size_t)2 * HT * (H + 4) + 4 * H) * sizeof(float);
    if (lds > 150 * 1024) return COPO_ERR_DIM;
    FwdArgs a{*cfg, theta, theta_t, obs_src, cc_src ? cc_src : obs_src, n_rows, first_net, n_nets, values, dist_inputs, eps,
              action, logp, clipped};
    const dim3 grid((unsigned)((n_rows + HT - 1) / HT), n_nets);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (H) {
        case 64: hipLaunchKernelGGL((mlp_fwd_kernel<1, 4>), grid, dim3(256), lds, st, a); break;
        case 128: hipLaunchKernelGGL((mlp_fwd_kernel<2, 4>), grid, dim3(256), lds, st, a); break;
        case 256: hipLaunchKernelGGL((mlp_fwd_kernel<2, 8>), grid, dim3(512), lds, st, a); break;
        default: hipLaunchKernelGGL((mlp_fwd_kernel<4, 8>), grid, dim3(512), lds, st, a); break;
    }
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_transpose_weights_f32(const copo_ppo_cfg* cfg, const float* theta, float* theta_t, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_t) return COPO_ERR_NULL;
    return launch_refresh_transposed(*cfg, theta, theta_t, static_cast<hipStream_t>(stream)) == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_grads_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, float* g_new, float* g_old,
                                   const float* obs_src, const float* pack_src, const int64_t* rows, const float* w,
                                   const float* denom, float* workspace, float* stats_new, float* stats_old,
                                   double* dot_partials, int64_t* mb_index, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_target || !g_new || !g_old || !obs_src || !pack_src || !rows || !w || !denom || !workspace ||
        !dot_partials)
        return COPO_ERR_NULL;
    FusedArgs a;
    fill_common(a, cfg, obs_src, nullptr, pack_src, rows, w, denom, workspace, mb_index);
    a.theta = theta; a.theta2 = theta_target; a.grad = g_new; a.grad2 = g_old;
    a.stats = stats_new; a.stats2 = stats_old; a.apply_adam = 0; a.head_mode = MODE_META_BOTH; a.groups = 2;
    a.dot_partials = dot_partials;
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_step_f64(const copo_ppo_cfg* cfg, float* theta, float* theta_target, float* g_new, float* g_old,
                                  const float* obs_src, const float* pack_src, const int64_t* rows, const float* w,
                                  const float* denom, float* workspace, float* stats_new, float* stats_old,
                                  double* dot_partials, int32_t col_adv, int32_t col_nei_adv, const double* eps,
                                  double* lcf_param, const double* raw_mean_std, double* tail, double* adam_state,
                                  double lr, double* stats, int64_t* mb_index, int32_t bump_index, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_target || !g_new || !g_old || !obs_src || !pack_src || !rows || !w || !denom || !workspace ||
        !dot_partials || !eps || !lcf_param || !raw_mean_std || !tail || !adam_state)
        return COPO_ERR_NULL;
    FusedArgs a;
    fill_common(a, cfg, obs_src, nullptr, pack_src, rows, w, denom, workspace, mb_index);
    a.theta = theta; a.theta2 = theta_target; a.grad = g_new; a.grad2 = g_old;
    a.stats = stats_new; a.stats2 = stats_old; a.apply_adam = 0; a.head_mode = MODE_META_BOTH; a.groups = 2;
    a.dot_partials = dot_partials;
    MetaTail mt;
    mt.lcf = MetaArgs{pack_src, rows, w, denom, eps, mb_index, cfg->mb, cfg->pack_width, col_adv, col_nei_adv, lcf_param,
                      raw_mean_std, tail};
    mt.fin = MetaFinishArgs{g_new, g_old, 0, dot_partials, COPO_META_DOT_PARTIALS, tail, lcf_param, adam_state, lr, stats_new,
                            stats_old, stats, mb_index, (mb_index && bump_index) ? 1 : 0};
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream), &mt);
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// ---- batched LCF meta pass --------------------------------------------------------------------------------------
static size_t batch_ws_base(const copo_ppo_cfg& c, int nb) {     // floats before the fp64 dot-partial scratch (kept even)
    const int g = 2 * nb > 4 ? 2 * nb : 8;
    const size_t t = WsLay(c, g, g).total();
    return (t + 1) & ~(size_t)1;
}
static int batch_fold_blocks(const copo_ppo_cfg& c) {
    int64_t lo;
    int n;
    fold_range(c, 1, &lo, &n);
    return (n + 256 * FOLD_EPT - 1) / (256 * FOLD_EPT);
}

extern "C" int64_t copo_meta_fold_len(const copo_ppo_cfg* cfg) {
    if (!cfg) return -1;
    int64_t lo;
    int n;
    fold_range(*cfg, 1, &lo, &n);
    return n;
}

extern "C" int64_t copo_meta_batch_workspace_floats(const copo_ppo_cfg* cfg, int32_t nb) {
    if (!cfg || nb < 1) return -1;
    return (int64_t)(batch_ws_base(*cfg, nb) + (size_t)2 * nb * batch_fold_blocks(*cfg));
}

extern "C" int copo_meta_batch_grads_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, const float* obs_src,
                                         const float* pack_src, const int64_t* rows, const float* w, const float* denom,
                                         float* workspace, int32_t nb_cap, int64_t mb_first, int32_t nb, float* g_out,
                                         double* gv_out, float* stats_out, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_target || !obs_src || !pack_src || !rows || !w || !denom || !workspace || !gv_out || !stats_out)
        return COPO_ERR_NULL;
    if (nb < 1 || nb > nb_cap || nb_cap > COPO_META_BATCH_MAX || mb_first < 0) return COPO_ERR_DIM;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return COPO_ERR_DIM;
    FusedArgs a;
    fill_common(a, cfg, obs_src, nullptr, pack_src, rows, w, denom, workspace, nullptr);
    a.theta = theta; a.theta2 = theta_target; a.apply_adam = 0; a.head_mode = MODE_META_BOTH;
    a.groups = 2 * nb;
    a.gcap = a.nreg = 2 * nb_cap > 4 ? 2 * nb_cap : 8;   // > 4: the batched layout (see WsLay); fixed by nb_cap so that
                                                         // a short last chunk reuses the same (zero-padded) regions
    a.k_first = mb_first;
    double* dot = reinterpret_cast<double*>(workspace + batch_ws_base(*cfg, nb_cap));
    MetaBatch mbt{nb, g_out, dot, stats_out, 0};
    int fb = 0;
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream), nullptr, &mbt, &fb, nullptr);
    if (e != hipSuccess) return COPO_ERR_DEVICE;
    hipLaunchKernelGGL(meta_batch_dot_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dot, nullptr,
                       (int64_t)fb, gv_out);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// ---- meta row store: row-local quantities once per training iteration, regrouped by every meta pass ---------------
static int rows_blocks(const copo_ppo_cfg& c, int64_t n_rows) { return (int)((n_rows + c.mb - 1) / c.mb); }
static int rows_gcap(const copo_ppo_cfg& c, int64_t n_rows) {
    const int g = 2 * rows_blocks(c, n_rows);
    return g > 4 ? g : 8;
}

extern "C" int64_t copo_meta_rows_workspace_floats(const copo_ppo_cfg* cfg, int64_t n_rows) {
    if (!cfg || n_rows < 1) return -1;
    return (int64_t)WsLay(*cfg, rows_gcap(*cfg, n_rows), 0).total();      // activation slabs only: no gradient regions
}

extern "C" int copo_meta_rows_f32(const copo_ppo_cfg* cfg, float* theta, float* theta_target, const float* obs_src,
                                  const float* pack_src, int64_t n_rows, float* rows_ws, float* rowstat, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_target || !obs_src || !pack_src || !rows_ws || !rowstat) return COPO_ERR_NULL;
    if (n_rows < 1 || 2 * rows_blocks(*cfg, n_rows) > 65535) return COPO_ERR_DIM;
    FusedArgs a;
    fill_common(a, cfg, obs_src, nullptr, pack_src, nullptr, nullptr, nullptr, rows_ws, nullptr);
    a.theta = theta; a.theta2 = theta_target; a.apply_adam = 0; a.head_mode = MODE_META_BOTH;
    a.groups = 2 * rows_blocks(*cfg, n_rows);
    a.gcap = rows_gcap(*cfg, n_rows);
    a.nreg = 0;
    a.identity_rows = n_rows;
    a.rowstat = rowstat;
    MetaBatch mbt{a.groups / 2, nullptr, nullptr, nullptr, 1};
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream), nullptr, &mbt, nullptr, nullptr);
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_batch_wgrads_f32(const copo_ppo_cfg* cfg, const float* obs_src, const int64_t* rows, const float* w,
                                          const float* denom, const float* rows_ws, int64_t n_rows, const float* rowstat,
                                          float* workspace, int32_t nb_cap, int64_t mb_first, int32_t nb, float* g_out,
                                          double* gv_out, float* stats_out, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!obs_src || !rows || !w || !denom || !rows_ws || !rowstat || !workspace || !gv_out || !stats_out) return COPO_ERR_NULL;
    if (nb < 1 || nb > nb_cap || nb_cap > COPO_META_BATCH_MAX || mb_first < 0 || n_rows < 1) return COPO_ERR_DIM;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return COPO_ERR_DIM;
    FusedArgs a;
    fill_common(a, cfg, obs_src, nullptr, nullptr, rows, w, denom, workspace, nullptr);
    a.apply_adam = 0; a.head_mode = MODE_META_BOTH;
    a.groups = 2 * nb;
    a.gcap = a.nreg = 2 * nb_cap > 4 ? 2 * nb_cap : 8;
    a.k_first = mb_first;
    a.ws0 = rows_ws;
    a.gcap0 = rows_gcap(*cfg, n_rows);
    a.rowstat = const_cast<float*>(rowstat);
    double* dot = reinterpret_cast<double*>(workspace + batch_ws_base(*cfg, nb_cap));
    MetaBatch mbt{nb, g_out, dot, nullptr, 2};
    int fb = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = launch_fused_step(a, st, nullptr, &mbt, &fb, nullptr);
    if (e != hipSuccess) return COPO_ERR_DEVICE;
    hipLaunchKernelGGL(meta_rowstat_kernel, dim3(nb), dim3(256), 0, st, a, mb_first, denom, stats_out);
    hipLaunchKernelGGL(meta_batch_dot_kernel, dim3(nb), dim3(256), 0, st, dot, nullptr, (int64_t)fb, gv_out, denom + mb_first);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_batch_dot_f64(const float* g, int64_t n, int32_t nb, double* gv_out, void* stream) {
    if (!g || !gv_out) return COPO_ERR_NULL;
    if (n < 1 || nb < 1) return COPO_ERR_DIM;
    hipLaunchKernelGGL(meta_batch_dot_kernel, dim3(nb), dim3(1024), 0, static_cast<hipStream_t>(stream), nullptr, g, n, gv_out);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_batch_lcf_f64(const float* pack_src, int32_t pack_width, int32_t col_adv, int32_t col_nei_adv,
                                       const int64_t* rows, const float* ego_nei, int32_t n_seg, const float* w,
                                       const double* eps, const float* denom, int32_t mb, int32_t n_mb, const double* gv,
                                       const float* stats_in, double* lcf_param, const double* raw_mean_std,
                                       double* adam_state, double lr, double* stats, void* stream) {
    if (!w || !eps || !denom || !gv || !stats_in || !lcf_param || !raw_mean_std || !adam_state) return COPO_ERR_NULL;
    if (!ego_nei && (!pack_src || !rows)) return COPO_ERR_NULL;
    if (mb < 1 || n_mb < 0 || n_seg < 1 || (!ego_nei && n_seg != 1)) return COPO_ERR_DIM;
    if (n_mb == 0) return COPO_OK;
    MetaSeqArgs a{pack_src, rows, ego_nei, w, eps, denom, gv, stats_in, mb, n_mb, n_seg, pack_width, col_adv, col_nei_adv,
                  lcf_param, raw_mean_std, adam_state, lr, stats};
    hipLaunchKernelGGL(meta_seq_kernel, dim3(1), dim3(512), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_lcf_f64(const float* pack_src, int32_t pack_width, int32_t col_adv, int32_t col_nei_adv,
                                 const int64_t* rows, const float* w, const float* denom, const double* eps, int32_t mb,
                                 const int64_t* mb_index, const double* lcf_param, const double* raw_mean_std, double* tail,
                                 void* stream) {
    if (!pack_src || !rows || !w || !denom || !eps || !lcf_param || !raw_mean_std || !tail) return COPO_ERR_NULL;
    if (mb < 1) return COPO_ERR_DIM;
    MetaArgs a{pack_src, rows, w, denom, eps, mb_index, mb, pack_width, col_adv, col_nei_adv, lcf_param, raw_mean_std, tail};
    hipLaunchKernelGGL(meta_lcf_kernel, dim3(1), dim3(512), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_finish_f64(const float* g_new, const float* g_old, int64_t n, const double* dot_partials,
                                    const double* tail, double* lcf_param, double* adam_state, double lr,
                                    float* stats_new, float* stats_old, double* stats, int64_t* mb_index,
                                    int32_t bump_index, void* stream) {
    if (!tail || !lcf_param || !adam_state) return COPO_ERR_NULL;
    if (!dot_partials && (!g_new || !g_old)) return COPO_ERR_NULL;
    if (n < 0) return COPO_ERR_DIM;
    MetaFinishArgs a{g_new, g_old, n, dot_partials, COPO_META_DOT_PARTIALS, tail, lcf_param, adam_state, lr, stats_new,
                     stats_old, stats, mb_index, (mb_index && bump_index) ? 1 : 0};
    hipLaunchKernelGGL(meta_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_adam_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, const float* grad,
                                  int64_t n, int64_t* step, int64_t* mb_index, float* theta_t, float* workspace,
                                  void* stream) {
    if (!cfg || !theta || !adam_m || !adam_v || !grad || !step) return COPO_ERR_NULL;
    if (n < 0) return COPO_ERR_DIM;
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    a.c = *cfg;
    a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v; a.grad = const_cast<float*>(grad); a.step = step;
    a.kptr = mb_index; a.bump_k = mb_index ? 1 : 0;
    a.theta_t = theta_t;
    a.ws = workspace;        // non-NULL: step number / next index come from the slots the gradient pass published
    a.gcap = 4; a.nreg = 2;
    hipError_t e = launch_adam_flat(a, n, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

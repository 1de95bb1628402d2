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
#include "learn_args.inc"
#include "learn_gemm.inc"
#include "learn_rowpass.inc"
#include "learn_meta.inc"
#include "learn_update.inc"

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
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, true>), reinterpret_cast<const void*>(rowpass_kernel<4, 8, true>),
                              reinterpret_cast<const void*>(rowpass_kernel<1, 4, true, true>), reinterpret_cast<const void*>(rowpass_kernel<2, 4, true, true>),
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, true, true>), reinterpret_cast<const void*>(rowpass_kernel<4, 8, true, true>),
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, true, false, true>),
                              reinterpret_cast<const void*>(rowpass_kernel<2, 8, true, false, true, 8>), reinterpret_cast<const void*>(rowpass_kernel<2, 8, true, false, false, 8>),
                              reinterpret_cast<const void*>(rowpass8_kernel<4>), reinterpret_cast<const void*>(rowpass8_kernel<4, true>)})
            if ((r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess) e = r;
        for (const void* f : {reinterpret_cast<const void*>(mlp_fwd_kernel<1, 4>), reinterpret_cast<const void*>(mlp_fwd_kernel<2, 4>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<2, 8>), reinterpret_cast<const void*>(mlp_fwd_kernel<4, 8>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<1, 4, true>), reinterpret_cast<const void*>(mlp_fwd_kernel<2, 4, true>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<2, 8, true>), reinterpret_cast<const void*>(mlp_fwd_kernel<4, 8, true>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<2, 8, false, 2>), reinterpret_cast<const void*>(mlp_fwd_kernel<2, 8, true, 2>)})
            if ((r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess) e = r;
        if ((r = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_adam_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) e = r;
        return e;
    }();
    return once;
}

// part: 0 = the whole chain; 1 = row-local kernels only (fill the meta row store); 2 = weight-gradient GEMMs that
// gather from the row store + fold (a meta pass over precomputed rows)
struct MetaBatch { int nb; float* g_out; double* dot_out; float* stats_out; int part; };

// The shipped library takes no environment variables: one code path, all of the work, every time.  Profiling builds
// (`make prof`, -DCOPO_PROFILE_SKIP=<mask>, a separate .so that the package never loads) read COPO_RP_DBG for the
// phase-stamp / phase-skip bits of the row pass and can fall back to the older kernel chains for A/B measurements.
#ifdef COPO_PROFILE_SKIP
static const bool g_rowpass_4x4 = [] { const char* e = getenv("COPO_ROWPASS_4X4"); return !(e && e[0] == '0'); }();
static const bool g_rowpass_rt8 = [] { const char* e = getenv("COPO_ROWPASS_RT8"); return !(e && e[0] == '0'); }();
static const int g_wgrad_ot = [] { const char* e = getenv("COPO_WGRAD_OT"); return (e && e[0] == '2') ? 2 : 1; }();
static const bool g_use_wgrad = [] { const char* e = getenv("COPO_FUSED_WGRAD"); return !(e && e[0] == '0'); }();
static const bool g_use_rowpass = [] { const char* e = getenv("COPO_FUSED_ROWPASS"); return !(e && e[0] == '0'); }();
static int profile_dbg_bits() { static const int dbg = [] { const char* e = getenv("COPO_RP_DBG"); return e ? atoi(e) : 0; }(); return dbg; }
#else
static constexpr bool g_rowpass_4x4 = true;
static constexpr bool g_rowpass_rt8 = true;
static constexpr int g_wgrad_ot = 1;
static constexpr bool g_use_wgrad = true;
static constexpr bool g_use_rowpass = true;
static constexpr int profile_dbg_bits() { return 0; }
#endif

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
    a.dbg = profile_dbg_bits();
    a.stat_tiles = head_tiles(c);
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
        if (vec) hipLaunchKernelGGL((gemm_kernel<OP, true>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, K);   \
        else hipLaunchKernelGGL((gemm_kernel<OP, false>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, K);      \
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
    if (part == 2) {
        // the gather-all tiles address the row store with 32-bit byte offsets (buffer loads): every operand slab must stay below 4 GB
        const uint64_t widest = (uint64_t)(c.hidden > c.pol.in_dim ? c.hidden : c.pol.in_dim);
        vec = vec && ((reinterpret_cast<uintptr_t>(a.ws0) & 15) == 0) && ((uint64_t)a.gcap0 * (uint64_t)c.mb * widest * 4u < 0xfffffff0ull);
    }
    const bool rowpass = (vec || tw_ok) && !mbatch && g_use_rowpass && (rp_h == 64 || rp_h == 128 || rp_h == 256 || rp_h == 512) &&
                         rp_lds <= 150 * 1024;
    // bfloat16 operands: only the row pass + fused weight-gradient kernels round where torch.autocast rounds
    const bool bf = c.operand_dtype == COPO_OPERAND_BF16;
    if (bf && !(rowpass && tw && tw_ok && a.head_mode == COPO_HEAD_PPO && g_use_wgrad)) return hipErrorInvalidValue;
    if (rowpass) {
        // 8-row tiles where 16-row tiles would leave compute units without a workgroup (rowpass_kernel, RT): the production
        // shape only (hidden 256 through the transposed mirror)
        const bool rt8 = tw && rp_h == 256 && head_tiles(c) * G <= 160 && g_rowpass_rt8 && (!bf || g_rowpass_4x4);
        if (rt8) a.stat_tiles = (c.mb + 7) / 8;
        const dim3 grid(a.stat_tiles, G);
#define COPO_RP(NT_, W_, HO_)                                                                                     \
        do {                                                                                                   \
            if (bf && rt8 && g_rowpass_4x4) hipLaunchKernelGGL((rowpass8_kernel<4, true>), grid, dim3(64 * R8_KS * 4), rowpass8_lds_floats(256, rowpass8_k1p(kmax1)) * sizeof(float), s, a); \
            else if (bf) hipLaunchKernelGGL((rowpass_kernel<NT_, W_, true, true>), grid, dim3(64 * W_), rp_lds, s, a); \
            else if (rt8 && g_rowpass_4x4) hipLaunchKernelGGL((rowpass8_kernel<4>), grid, dim3(64 * R8_KS * 4), rowpass8_lds_floats(256, rowpass8_k1p(kmax1)) * sizeof(float), s, a); \
            else if (rt8 && HO_ && kmax1 <= 128) hipLaunchKernelGGL((rowpass_kernel<2, 8, true, false, true, 8>), grid, dim3(64 * W_), rp_lds, s, a); \
            else if (rt8) hipLaunchKernelGGL((rowpass_kernel<2, 8, true, false, false, 8>), grid, dim3(64 * W_), rp_lds, s, a); \
            else if (tw && HO_ && kmax1 <= 128) hipLaunchKernelGGL((rowpass_kernel<NT_, W_, true, false, HO_>), grid, dim3(64 * W_), rp_lds, s, a); \
            else if (tw) hipLaunchKernelGGL((rowpass_kernel<NT_, W_, true>), grid, dim3(64 * W_), rp_lds, s, a); \
            else hipLaunchKernelGGL((rowpass_kernel<NT_, W_, false>), grid, dim3(64 * W_), rp_lds, s, a);        \
        } while (0)
        switch (rp_h) {
            case 64: COPO_RP(1, 4, false); break;
            case 128: COPO_RP(2, 4, false); break;
            case 256: COPO_RP(2, 8, true); break;       // hidden 256: layer 2 is two ring turns -> hand-over variant for narrow inputs
            default: COPO_RP(4, 8, false); break;
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
    if (a.dp_world > 1 && (mbatch || a.head_mode == MODE_META_BOTH || !g_use_wgrad || g_wgrad_ot != 1 || !a.apply_adam)) return hipErrorInvalidValue;
    if (!mbatch && a.head_mode != MODE_META_BOTH && g_use_wgrad) {
        // weight gradients, fold and Adam in one kernel: the SGD step ends here
        const int ot = g_wgrad_ot;
        const int nty = (c.hidden + 32 * ot - 1) / (32 * ot), wx2 = (c.hidden + 1 + 31) / 32, wx1 = (kmax1 + 1 + 31) / 32;
        const dim3 grid((wx2 + wx1) * nty + wx2, G);      // (wgrad_tiles() below: the exchange workspace is sized by this grid)
        const size_t lds = (size_t)WG_WAVES * 32 * ot * 33 * sizeof(float);
        const bool fastk = c.mb == 2 * WG_RING * WG_WAVES && (bf || ot == 1);      // buffer-load operand ring (learn_update.inc)
        if (bf && fastk) hipLaunchKernelGGL((wgrad_adam_kernel<1, true, true>), grid, dim3(64 * WG_WAVES), (size_t)WG_WAVES * 32 * 33 * sizeof(float), s, a, nty, (c.hidden + 1 + 31) / 32, wx1);
        else if (bf) hipLaunchKernelGGL((wgrad_adam_kernel<1, true>), grid, dim3(64 * WG_WAVES), (size_t)WG_WAVES * 32 * 33 * sizeof(float), s, a, nty, (c.hidden + 1 + 31) / 32, wx1);
        else if (ot == 2) hipLaunchKernelGGL((wgrad_adam_kernel<2>), grid, dim3(64 * WG_WAVES), lds, s, a, nty, wx2, wx1);
        else if (fastk) hipLaunchKernelGGL((wgrad_adam_kernel<1, false, true>), grid, dim3(64 * WG_WAVES), lds, s, a, nty, wx2, wx1);
        else hipLaunchKernelGGL((wgrad_adam_kernel<1>), grid, dim3(64 * WG_WAVES), lds, s, a, nty, wx2, wx1);
        return hipGetLastError();
    }
    {
        // layers 2 / 3: no tile for the bias column alone when the hidden width fills whole tiles (BwOpT::bias_by_rowsum)
        const int nx2 = (c.hidden % TN == 0) ? c.hidden / TN : (c.hidden + 1 + TN - 1) / TN, nx1 = (kmax1 + 1 + TN - 1) / TN;
        dim3 grid((nx2 + nx1) * ht + nx2, G * a.ksplit);
        if (part == 2) {
            if (vec) grid.x = (nx2 + nx1) * ht + 1;          // gather-all mode, vector rows: the head layer is one workgroup on the lanes (head_wgrad_lanes)
            if (vec) hipLaunchKernelGGL((gemm_bw_kernel<true, true>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, c.mb, nx2, nx1, ht);
            else hipLaunchKernelGGL((gemm_bw_kernel<false, true>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, c.mb, nx2, nx1, ht);
        } else if (vec) hipLaunchKernelGGL((gemm_bw_kernel<true>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, c.mb, nx2, nx1, ht);
        else hipLaunchKernelGGL((gemm_bw_kernel<false>), grid, dim3(256), gemm_lds_bytes(a.c.mb), s, a, c.mb, nx2, nx1, ht);
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
    // this Adam does not write the transposed mirror (the fused weight-gradient kernel does): bring it up to date, or
    // the next row pass would read stale weights
    if (a.apply_adam && a.theta_t && a.head_mode == COPO_HEAD_PPO) return launch_refresh_transposed(c, a.theta, a.theta_t, s);
    return hipGetLastError();
}

// tiles of the weight-gradient kernel's grid for a PPO step over all nets of `c` (= workgroups; launch_fused_step)
static int wgrad_tiles(const copo_ppo_cfg& c) {
    const int G = 1 + c.n_value_heads;
    int kmax1 = c.pol.in_dim;
    for (int g = 1; g < G; ++g) kmax1 = c.val[g - 1].in_dim > kmax1 ? c.val[g - 1].in_dim : kmax1;
    const int nty = (c.hidden + 31) / 32, wx2 = (c.hidden + 1 + 31) / 32, wx1 = (kmax1 + 1 + 31) / 32;
    return ((wx2 + wx1) * nty + wx2) * G;
}

hipError_t launch_adam_flat(const FusedArgs& a, long long n, hipStream_t s) {
    const int slots = a.ws != nullptr;
    const int n_tiles = a.theta_t ? adam_tile_count(a.c) : 0;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)(n_tiles + (n + 256 * ADAM_EPT - 1) / (256 * ADAM_EPT))), dim3(256), 0, s, a, n, slots, n_tiles);
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

extern "C" int copo_debug_wg_times(unsigned long long* out4096) {
    return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(copo::g_wg_times), 4096 * sizeof(unsigned long long)) == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int64_t copo_ppo_workspace_floats(const copo_ppo_cfg* cfg) { return cfg ? (int64_t)fused_ws_floats(*cfg) : -1; }

static int check_cfg(const copo_ppo_cfg* c) {
    if (!c) return COPO_ERR_NULL;
    if (c->mb < 1 || c->mb > COPO_PPO_MAX_MB || c->hidden < 1 || c->hidden > 1024 || c->act_dim != 2 ||
        c->n_value_heads < 0 || c->n_value_heads > 3 || c->n_params < 1)
        return COPO_ERR_DIM;
    if (c->pol.out_dim != 4) return COPO_ERR_DIM;
    if (c->operand_dtype != COPO_OPERAND_F32 && c->operand_dtype != COPO_OPERAND_BF16) return COPO_ERR_DIM;
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
    if (!theta || !obs_src || !pack_src || !w || !denom || !workspace) return COPO_ERR_NULL;
    if (!rows && head_mode != COPO_HEAD_PPO) return COPO_ERR_NULL;       // (sources in minibatch order: the PPO step only)
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

extern "C" int64_t copo_dp_workspace_bytes(const copo_ppo_cfg* cfg, int32_t world) {
    if (check_cfg(cfg) != COPO_OK || world < 1 || world > COPO_PEER_MAX_WORLD) return -1;
    return (int64_t)(DpLay(wgrad_tiles(*cfg), world).words() * sizeof(float));
}

extern "C" int copo_ppo_fused_step_dp_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v,
                                          const float* obs_src, const float* cc_src, const float* pack_src,
                                          const int64_t* rows, const float* w, const float* denom, const float* kl_coeff,
                                          int64_t* step, float* workspace, float* stats, int64_t* mb_index, int32_t bump_index,
                                          float* theta_t, void* const* dp_workspaces, int32_t rank, int32_t world, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !obs_src || !pack_src || !w || !denom || !workspace || !adam_m || !adam_v || !step) return COPO_ERR_NULL;
    if (cfg->use_kl && !kl_coeff) return COPO_ERR_NULL;
    if (world < 1 || world > COPO_PEER_MAX_WORLD || rank < 0 || rank >= world) return COPO_ERR_DIM;
    if (world > 1 && !dp_workspaces) return COPO_ERR_NULL;
    FusedArgs a;
    fill_common(a, cfg, obs_src, cc_src, pack_src, rows, w, denom, workspace, mb_index);
    a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
    a.kl_coeff = kl_coeff; a.step = step; a.stats = stats; a.apply_adam = 1; a.head_mode = COPO_HEAD_PPO;
    a.groups = 1 + cfg->n_value_heads;
    a.bump_k = (mb_index && bump_index) ? 1 : 0;
    a.theta_t = theta_t;
    a.dp_rank = rank; a.dp_world = world;
    for (int r = 0; r < world && world > 1; ++r) {
        if (!dp_workspaces[r]) return COPO_ERR_NULL;
        a.dp_ws[r] = static_cast<float*>(dp_workspaces[r]);
    }
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// error word of this rank's exchange workspace (0 = every wait so far was answered); synchronises the stream first
extern "C" int copo_dp_status(void* workspace, const copo_ppo_cfg* cfg, int32_t world, void* stream) {
    if (!workspace || check_cfg(cfg) != COPO_OK || world < 1 || world > COPO_PEER_MAX_WORLD) return COPO_ERR_NULL;
    uint32_t ctl[8];
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return COPO_ERR_DEVICE;
    if (hipMemcpy(ctl, static_cast<float*>(workspace) + DpLay(wgrad_tiles(*cfg), world).control(), sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess)
        return COPO_ERR_DEVICE;
    return ctl[0] ? COPO_ERR_DEVICE : COPO_OK;
}

static int mlp_forward(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t, const float* obs_src,
                       const float* cc_src, int64_t n_rows, int32_t first_net, int32_t n_nets, float* values,
                       float* dist_inputs, const float* eps, float* action, float* logp, float* clipped,
                       const int64_t* rows, int64_t n_out, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !theta_t || !obs_src) return COPO_ERR_NULL;
    if (rows && n_out < 1) return COPO_ERR_DIM;
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
    // hidden 256 with at least two rounds of 256 workgroups of 32 rows per net: two 16-row tiles per workgroup (learn_rowpass.inc:
    // RT).  Measured: the critic heads on 72 k rows 580 -> 478 us; a rollout step's 10 240 rows (320 workgroups of 32 rows on 256
    // CUs) 39.5 -> 43.8 us, hence the threshold
    const bool rt2 = H == 256 && n_rows >= 2 * 32 * 256 && rowpass_k1p(kmax) <= H;
    const size_t lds = rt2 ? ((size_t)2 * 2 * HT * (H + 4) + 4 * H) * sizeof(float)
                           : ((size_t)HT * (rowpass_k1p(kmax) + 4) + (size_t)2 * HT * (H + 4) + 4 * H) * sizeof(float);
    if (lds > 150 * 1024) return COPO_ERR_DIM;
    FwdArgs a{*cfg, theta, theta_t, obs_src, cc_src ? cc_src : obs_src, n_rows, first_net, n_nets, values, dist_inputs, eps,
              action, logp, clipped, rows, rows ? n_out : n_rows};
    const int rws = rt2 ? 2 * HT : HT;
    const dim3 grid((unsigned)((n_rows + rws - 1) / rws), n_nets);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rt2) {
        if (cfg->operand_dtype == COPO_OPERAND_BF16) hipLaunchKernelGGL((mlp_fwd_kernel<2, 8, true, 2>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((mlp_fwd_kernel<2, 8, false, 2>), grid, dim3(512), lds, st, a);
        return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
    }
#define COPO_FWD(NT_, W_)                                                                                       \
        do {                                                                                                   \
            if (cfg->operand_dtype == COPO_OPERAND_BF16) hipLaunchKernelGGL((mlp_fwd_kernel<NT_, W_, true>), grid, dim3(64 * W_), lds, st, a); \
            else hipLaunchKernelGGL((mlp_fwd_kernel<NT_, W_>), grid, dim3(64 * W_), lds, st, a);                 \
        } while (0)
    switch (H) {
        case 64: COPO_FWD(1, 4); break;
        case 128: COPO_FWD(2, 4); break;
        case 256: COPO_FWD(2, 8); break;
        default: COPO_FWD(4, 8); break;
    }
#undef COPO_FWD
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_mlp_forward_f32(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t, const float* obs_src,
                                    const float* cc_src, int64_t n_rows, int32_t first_net, int32_t n_nets, float* values,
                                    float* dist_inputs, const float* eps, float* action, float* logp, float* clipped,
                                    void* stream) {
    return mlp_forward(cfg, theta, theta_t, obs_src, cc_src, n_rows, first_net, n_nets, values, dist_inputs, eps, action, logp,
                       clipped, nullptr, n_rows, stream);
}

extern "C" int copo_mlp_forward_rows_f32(const copo_ppo_cfg* cfg, const float* theta, const float* theta_t,
                                         const float* obs_src, const float* cc_src, const int64_t* rows, int64_t n_rows,
                                         int64_t n_src_rows, int32_t first_net, int32_t n_nets, float* values, void* stream) {
    if (!rows) return COPO_ERR_NULL;
    return mlp_forward(cfg, theta, theta_t, obs_src, cc_src, n_rows, first_net, n_nets, values, nullptr, nullptr, nullptr, nullptr,
                       nullptr, rows, n_src_rows, stream);
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
    if (!obs_src || !rows || !w || !denom || !rows_ws || !rowstat || !workspace || !gv_out) return COPO_ERR_NULL;      // (stats_out may be NULL: copo_meta_rowstat_f32)
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
    if (stats_out) hipLaunchKernelGGL(meta_rowstat_kernel, dim3(nb), dim3(256), 0, st, a, mb_first, denom, stats_out);
    hipLaunchKernelGGL(meta_batch_dot_kernel, dim3(nb), dim3(256), 0, st, dot, nullptr, (int64_t)fb, gv_out, denom + mb_first);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_rowstat_f32(const copo_ppo_cfg* cfg, const int64_t* rows, const float* w, const float* denom,
                                     const float* rowstat, int64_t mb_first, int32_t nb, float* stats_out, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!rows || !w || !denom || !rowstat || !stats_out) return COPO_ERR_NULL;
    if (nb < 1 || nb > 65535 || mb_first < 0) return COPO_ERR_DIM;
    FusedArgs a;
    fill_common(a, cfg, nullptr, nullptr, nullptr, rows, w, denom, nullptr, nullptr);
    a.rowstat = const_cast<float*>(rowstat);
    hipLaunchKernelGGL(meta_rowstat_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), a, mb_first, denom, stats_out);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_batch_dot_f64(const float* g, int64_t n, int32_t nb, double* gv_out, const float* denom, double* partials,
                                       void* stream) {
    if (!g || !gv_out) return COPO_ERR_NULL;
    if (n < 1 || nb < 1) return COPO_ERR_DIM;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (partials && nb <= 65535) {      // grid (nb, COPO_META_DOT_SPLIT) fills the chip; the partials meet in the caller's buffer
        hipLaunchKernelGGL(meta_batch_dot_part_kernel, dim3(nb, DOT_SPLIT), dim3(512), 0, st, g, n, partials);
        hipLaunchKernelGGL(meta_batch_dot_fin_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, nb, gv_out, denom, partials);
    } else {
        hipLaunchKernelGGL(meta_batch_dot_kernel, dim3(nb), dim3(1024), 0, st, nullptr, g, n, gv_out, denom);
    }
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_meta_batch_lcf_f64(const float* pack_src, int32_t pack_width, int32_t col_adv, int32_t col_nei_adv,
                                       const int64_t* rows, const float* ego_nei, int32_t n_seg, const float* w,
                                       const double* eps, const float* denom, int32_t mb, int32_t n_mb, const double* gv,
                                       const float* stats_in, double* lcf_param, const double* raw_mean_std,
                                       double* adam_state, double lr, double* stats, int32_t k_first, int32_t k_count,
                                       int32_t n_wg, double* exchange, void* stream) {
    if (!w || !eps || !denom || !gv || !stats_in || !lcf_param || !raw_mean_std || !adam_state) return COPO_ERR_NULL;
    if (!ego_nei && (!pack_src || !rows)) return COPO_ERR_NULL;
    if (mb < 1 || n_mb < 0 || n_mb > 65534 || n_seg < 1 || (!ego_nei && n_seg != 1)) return COPO_ERR_DIM;
    if (n_wg < 0 || n_wg > SEQ_MAX_WG) return COPO_ERR_DIM;
    if (k_count < 0) k_count = n_mb - k_first;          // (-1: all steps from k_first on)
    if (k_first < 0 || k_first + k_count > n_mb) return COPO_ERR_DIM;
    // measured (scripts/meta_seq_time.py, 90 steps): one workgroup 2.4 / 2.6 / 3.3 / 4.9 us per step with the rows of 1 / 2 / 4 / 8
    // ranks (eight rows per thread are still latency, not work), the hand-over ~2.4 us per step on top of a workgroup's own rows
    // (8 ranks on 8 workgroups: 4.9 us) -- so several workgroups only pay beyond eight ranks' rows: then one per four segments
    if (n_wg == 0) n_wg = (n_seg > 8 && exchange) ? ((n_seg + 3) / 4 < SEQ_MAX_WG ? (n_seg + 3) / 4 : SEQ_MAX_WG) : 1;
    if (n_wg > 1 && !exchange) return COPO_ERR_NULL;
    if (n_mb == 0 || k_count == 0) return COPO_OK;
    MetaSeqArgs a{pack_src, rows, ego_nei, w, eps, denom, gv, stats_in, mb, n_mb, n_seg, pack_width, col_adv, col_nei_adv,
                  k_first, k_count, lcf_param, raw_mean_std, adam_state, lr, stats, exchange, n_wg};
    hipLaunchKernelGGL(meta_seq_kernel, dim3(n_wg), dim3(512), 0, static_cast<hipStream_t>(stream), a);
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

// Fused minibatch learner for the CoPO / CCPPO / IPPO MLPs on gfx950.
//
// The reference runs the PPO minibatch step through torch autograd (algo_copo.py:311-424 `loss`, RLlib
// `train_one_step`, torch.optim.Adam): ~250 tiny kernels per 512-row minibatch, launch-bound on any GPU.
// Here one SGD step is 7 launches over ONE flat fp32 parameter buffer:
//
//   F1   h1 = tanh(X W1^T + b1)          grouped over the policy net + up to 3 value nets; X rows gathered by index
//   F2   h2 = tanh(h1 W2^T + b2)
//   H    heads (256 -> 4 / 1), the PPO loss terms and their ANALYTIC gradient w.r.t. the head outputs,
//        dz2 = (dout W3) * (1 - h2^2), per-tile partials of dW3 / db3, loss statistics
//   B2x  dz1 = (dz2 W2) * (1 - h1^2)
//   B2w  dW2 = dz2^T [h1 | 1]  (+ Adam update in the epilogue)
//   B1w  dW1 = dz1^T [X  | 1]  (+ Adam)
//   A3   fold the dW3 partials (+ Adam), bump the step counter
//
// GEMMs are 64x64 output tiles per 256-thread workgroup, 4 waves x one 32x32 fp32 MFMA accumulator
// (v_mfma_f32_32x32x2_f32: exact fp32 products at the fp32 vector rate), K staged through LDS in slabs of 16.
// With apply_adam = 0 the kernels only write the flat gradient (data-parallel runs all-reduce it and call
// copo_adam_step_f32); the same F/B kernels with head mode META_NEW / META_OLD produce the two policy
// gradients of the LCF meta update (algo_copo.py:250-278).
#include <cstring>

#include "sim_common.h"

#pragma clang fp contract(fast)

namespace copo {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int TM = 64, TN = 64, TK = 16, LDP = 68;   // LDP: padded LDS row (floats)
constexpr int HT = 32;                                // rows per workgroup of the head kernel

struct FusedArgs {
    copo_ppo_cfg c;
    float* theta;
    float* adam_m;
    float* adam_v;
    float* grad;
    const float* obs_src;
    const float* cc_src;
    const float* pack_src;
    const int64_t* rows;       // [mb]
    const float* w;            // [mb] row weights (1 valid / 0 padding)
    const float* denom;        // [1] global number of valid rows
    const float* kl_coeff;     // [1]
    const int64_t* step;       // [1] Adam step counter BEFORE this step
    float* ws;                 // workspace, layout below
    float* stats;              // [COPO_PPO_STATS] accumulated sums
    int32_t apply_adam;
    int32_t head_mode;
    int32_t groups;            // nets processed: 1 (policy only) or 1 + n_value_heads
    const int64_t* kptr;       // [1] device minibatch index k: rows/w are [*][mb] tables, denom is [*]; NULL -> 0
    int32_t bump_k;            // increment *kptr at the end of this call
};

__device__ __forceinline__ int64_t kbase(const FusedArgs& a) { return a.kptr ? a.kptr[0] : 0; }
__device__ __forceinline__ int64_t row_of(const FusedArgs& a, int m) { return a.rows[kbase(a) * a.c.mb + m]; }
__device__ __forceinline__ float w_of(const FusedArgs& a, int m) { return a.w[kbase(a) * a.c.mb + m]; }
__device__ __forceinline__ float denom_of(const FusedArgs& a) { return a.denom[kbase(a)]; }

// workspace layout (floats), G = 4 nets max, mb rows, H hidden
__device__ __host__ inline size_t ws_h1(const copo_ppo_cfg& c, int g) { return (size_t)g * c.mb * c.hidden; }
__device__ __host__ inline size_t ws_h2(const copo_ppo_cfg& c, int g) { return (size_t)(4 + g) * c.mb * c.hidden; }
__device__ __host__ inline size_t ws_dz2(const copo_ppo_cfg& c, int g) { return (size_t)(8 + g) * c.mb * c.hidden; }
__device__ __host__ inline size_t ws_dz1(const copo_ppo_cfg& c, int g) { return (size_t)(12 + g) * c.mb * c.hidden; }
__device__ __host__ inline size_t ws_p3(const copo_ppo_cfg& c) { return (size_t)16 * c.mb * c.hidden; }
// dW3 partials: [g][tile][out<=4][H+1]
__device__ __host__ inline size_t ws_p3_at(const copo_ppo_cfg& c, int g, int tile) {
    const int tiles = (c.mb + HT - 1) / HT;
    return ws_p3(c) + ((size_t)g * tiles + tile) * 4 * (c.hidden + 1);
}

__device__ __forceinline__ const copo_net_layout& net_of(const copo_ppo_cfg& c, int g) { return g == 0 ? c.pol : c.val[g - 1]; }

__device__ __forceinline__ void adam_update(const FusedArgs& a, size_t idx, float g) {
    const float b1 = a.c.beta1, b2 = a.c.beta2;
    const float t = (float)(a.step[0] + 1);
    float m = a.adam_m[idx], v = a.adam_v[idx];
    m = m + (g - m) * (1.0f - b1);
    v = v * b2 + g * g * (1.0f - b2);
    a.adam_m[idx] = m;
    a.adam_v[idx] = v;
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float den = sqrtf(v) / sqrtf(bc2) + a.c.eps;
    a.theta[idx] = a.theta[idx] - (a.c.lr / bc1) * (m / den);
}

// ------------------------------------------------------------------------------------------------------------
// generic 64x64 tile GEMM:  C[m][n] = sum_k A(m,k) * B(k,n)   (operand functors return 0 outside their range)
// ------------------------------------------------------------------------------------------------------------
template <class Op>
__device__ __forceinline__ void tile_gemm(const Op& op, int g, int m0, int n0, int K, v16f& acc) {
    __shared__ float As[TK][LDP];
    __shared__ float Bs[TK][LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mm, kk;
            if (Op::A_KCONTIG) { mm = tid >> 2; kk = (tid & 3) * 4 + r; }
            else { kk = tid >> 4; mm = (tid & 15) * 4 + r; }
            As[kk][mm] = op.lda(g, m0 + mm, k0 + kk);
            int nn, kb;
            if (Op::B_KCONTIG) { nn = tid >> 2; kb = (tid & 3) * 4 + r; }
            else { kb = tid >> 4; nn = (tid & 15) * 4 + r; }
            Bs[kb][nn] = op.ldb(g, k0 + kb, n0 + nn);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float av = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float bv = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
}

template <class Op>
__global__ void __launch_bounds__(256) gemm_kernel(Op op, int K) {
    const int g = blockIdx.z, m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    v16f acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    tile_gemm(op, g, m0, n0, K, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = m0 + wm * 32 + (j >> 2) * 8 + (lane >> 5) * 4 + (j & 3);
        const int col = n0 + wn * 32 + (lane & 31);
        op.store(g, row, col, acc[j]);
    }
}

// ---- layer forward: Y[m][n] = tanh(sum_k X[m][k] W[n][k] + b[n]) --------------------------------------------
struct FwdOp {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    FusedArgs a;
    int layer;   // 1 or 2
    __device__ __forceinline__ int in_dim(int g) const { return layer == 1 ? net_of(a.c, g).in_dim : a.c.hidden; }
    __device__ __forceinline__ float lda(int g, int m, int k) const {
        if (m >= a.c.mb || k >= in_dim(g)) return 0.0f;
        if (layer == 1) {
            const float* src = (g == 0) ? a.obs_src : a.cc_src;
            return src[(size_t)row_of(a, m) * in_dim(g) + k];
        }
        return a.ws[ws_h1(a.c, g) + (size_t)m * a.c.hidden + k];
    }
    __device__ __forceinline__ float ldb(int g, int k, int n) const {
        if (n >= a.c.hidden || k >= in_dim(g)) return 0.0f;
        const copo_net_layout& L = net_of(a.c, g);
        return a.theta[(layer == 1 ? L.w1 : L.w2) + (size_t)n * in_dim(g) + k];
    }
    __device__ __forceinline__ void store(int g, int m, int n, float v) const {
        if (m >= a.c.mb || n >= a.c.hidden) return;
        const copo_net_layout& L = net_of(a.c, g);
        const float y = tanhf(v + a.theta[(layer == 1 ? L.b1 : L.b2) + n]);
        a.ws[(layer == 1 ? ws_h1(a.c, g) : ws_h2(a.c, g)) + (size_t)m * a.c.hidden + n] = y;
    }
};

// ---- B2x: dz1[m][i] = (sum_o dz2[m][o] W2[o][i]) * (1 - h1[m][i]^2) -------------------------------------------
struct BxOp {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    FusedArgs a;
    __device__ __forceinline__ float lda(int g, int m, int k) const {
        if (m >= a.c.mb || k >= a.c.hidden) return 0.0f;
        return a.ws[ws_dz2(a.c, g) + (size_t)m * a.c.hidden + k];
    }
    __device__ __forceinline__ float ldb(int g, int k, int n) const {
        if (n >= a.c.hidden || k >= a.c.hidden) return 0.0f;
        return a.theta[net_of(a.c, g).w2 + (size_t)k * a.c.hidden + n];
    }
    __device__ __forceinline__ void store(int g, int m, int n, float v) const {
        if (m >= a.c.mb || n >= a.c.hidden) return;
        const float h = a.ws[ws_h1(a.c, g) + (size_t)m * a.c.hidden + n];
        a.ws[ws_dz1(a.c, g) + (size_t)m * a.c.hidden + n] = v * (1.0f - h * h);
    }
};

// ---- Bw: dW[o][i] = sum_m dz[m][o] * [In | 1][m][i]  (+ Adam); column i == in_dim is the bias gradient ---------
struct BwOp {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false;
    FusedArgs a;
    int layer;   // 2: dz2 x h1 ; 1: dz1 x X
    __device__ __forceinline__ int in_dim(int g) const { return layer == 1 ? net_of(a.c, g).in_dim : a.c.hidden; }
    __device__ __forceinline__ float lda(int g, int o, int m) const {   // A(mm = o, k = m)
        if (o >= a.c.hidden || m >= a.c.mb) return 0.0f;
        return a.ws[(layer == 2 ? ws_dz2(a.c, g) : ws_dz1(a.c, g)) + (size_t)m * a.c.hidden + o];
    }
    __device__ __forceinline__ float ldb(int g, int m, int i) const {   // B(k = m, n = i)
        const int K = in_dim(g);
        if (m >= a.c.mb || i > K) return 0.0f;
        if (i == K) return 1.0f;
        if (layer == 2) return a.ws[ws_h1(a.c, g) + (size_t)m * a.c.hidden + i];
        const float* src = (g == 0) ? a.obs_src : a.cc_src;
        return src[(size_t)row_of(a, m) * K + i];
    }
    __device__ __forceinline__ void store(int g, int o, int i, float v) const {
        const int K = in_dim(g);
        if (o >= a.c.hidden || i > K) return;
        const copo_net_layout& L = net_of(a.c, g);
        const size_t idx = (i == K) ? (size_t)(layer == 1 ? L.b1 : L.b2) + o
                                    : (size_t)(layer == 1 ? L.w1 : L.w2) + (size_t)o * K + i;
        if (a.apply_adam) adam_update(a, idx, v);
        else a.grad[idx] = v;
    }
};

// ------------------------------------------------------------------------------------------------------------
// head + loss kernel: one workgroup per (32-row tile, net); 8 threads per row
// ------------------------------------------------------------------------------------------------------------
constexpr float kLog2Pi = 1.8378770664093453f;

__global__ void __launch_bounds__(256) head_kernel(FusedArgs a) {
    extern __shared__ float lds[];
    const copo_ppo_cfg& c = a.c;
    const int H = c.hidden, g = blockIdx.y, tile = blockIdx.x, m0 = tile * HT;
    const copo_net_layout& L = net_of(c, g);
    const int OD = L.out_dim;            // 2*act_dim for the policy net, 1 for value nets
    float* h2s = lds;                    // [HT][H+1]
    float* w3s = h2s + HT * (H + 1);     // [4][H]
    float* douts = w3s + 4 * H;          // [HT][4]
    float* red = douts + HT * 4;         // [8 stats][4 waves]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* h2g = a.ws + ws_h2(c, g);
    for (int q = tid; q < HT * H; q += 256) {
        const int r = q / H, i = q - r * H;
        h2s[r * (H + 1) + i] = (m0 + r < c.mb) ? h2g[(size_t)(m0 + r) * H + i] : 0.0f;
    }
    for (int q = tid; q < OD * H; q += 256) w3s[q] = a.theta[L.w3 + q];
    __syncthreads();
    // outputs: 8 threads per row, each an eighth of the hidden units
    const int r = tid >> 3, part = tid & 7;
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < H; i += 8) {
        const float h = h2s[r * (H + 1) + i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < OD) out[j] += h * w3s[j * H + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out[j] += __shfl_xor(out[j], 1);
        out[j] += __shfl_xor(out[j], 2);
        out[j] += __shfl_xor(out[j], 4);
        if (j < OD) out[j] += a.theta[L.b3 + j];
    }
    // per-row loss terms and d(loss)/d(out)
    float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // total, policy, vf_ego, kl, entropy, vf_nei, vf_glob, adv
    float dout[4] = {0.f, 0.f, 0.f, 0.f};
    const int m = m0 + r;
    if (part == 0 && m < c.mb) {
        const float wgt = w_of(a, m) / denom_of(a);
        const float* pk = a.pack_src + (size_t)row_of(a, m) * c.pack_width;
        if (g == 0) {
            const int A = c.act_dim;     // A == 2
            float logp = 0.f, ent = 0.f, kl = 0.f, z[2], sig[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float mu = out[j], ls = out[A + j];
                sig[j] = expf(ls);
                z[j] = (pk[c.col_actions + j] - mu) / sig[j];
                logp += -0.5f * z[j] * z[j] - ls - 0.5f * kLog2Pi;
                ent += ls + 0.5f + 0.5f * kLog2Pi;
            }
            if (a.head_mode == COPO_HEAD_META_OLD) {       // loss = mean(logp) on the target net
                st[0] = st[1] = wgt * logp;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dout[j] = wgt * z[j] / sig[j];
                    dout[A + j] = wgt * (z[j] * z[j] - 1.0f);
                }
            } else {
                const float adv = pk[a.head_mode == COPO_HEAD_META_NEW ? c.col_meta_adv : c.col_adv];
                const float ratio = expf(logp - pk[c.col_logp]);
                const float s1 = adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.0f - c.clip_param), 1.0f + c.clip_param);
                const float s2 = adv * rc;
                const float surr = fminf(s1, s2);
                const bool inside = (ratio >= 1.0f - c.clip_param) && (ratio <= 1.0f + c.clip_param);
                const float dsurr_dratio = (inside || s1 < s2) ? adv : 0.0f;
                float dlogp = -dsurr_dratio * ratio;      // d(-surr)/d logp
                float dmu[2] = {0.f, 0.f}, dls[2] = {0.f, 0.f};
                const bool ppo = a.head_mode == COPO_HEAD_PPO;
                if (ppo && c.use_kl) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float mup = pk[c.col_dist + j], lsp = pk[c.col_dist + A + j];
                        const float sp = expf(lsp), dm = mup - out[j];
                        const float q = (sp * sp + dm * dm) / (sig[j] * sig[j]);
                        kl += out[A + j] - lsp + 0.5f * q - 0.5f;
                        dmu[j] += a.kl_coeff[0] * (-dm / (sig[j] * sig[j]));
                        dls[j] += a.kl_coeff[0] * (1.0f - q);
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
                st[0] = st[1] + (ppo ? (a.kl_coeff[0] * st[3] - c.entropy_coeff * st[4]) : 0.0f);
                st[7] = wgt * adv;
            }
        } else {
            const float v = out[0];
            const float vp = pk[c.col_vpred[g - 1]], T = pk[c.col_vtarget[g - 1]];
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
    if (part == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) douts[r * 4 + j] = dout[j];
    }
    // statistics: wave reduce -> LDS -> one atomic per workgroup and stat
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = st[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) red[k * 4 + wave] = s;
    }
    __syncthreads();
    if (tid < 8 && a.stats) {
        const float s = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
        if (s != 0.0f) atomicAdd(a.stats + tid, s);
    }
    // dz2 = (dout W3) * (1 - h2^2);  per-tile partial of dW3 / db3
    float* dz2 = a.ws + ws_dz2(c, g);
    for (int q = tid; q < HT * H; q += 256) {
        const int rr = q / H, i = q - rr * H;
        if (m0 + rr >= c.mb) continue;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < OD) s += douts[rr * 4 + j] * w3s[j * H + i];
        const float h = h2s[rr * (H + 1) + i];
        dz2[(size_t)(m0 + rr) * H + i] = s * (1.0f - h * h);
    }
    float* p3 = a.ws + ws_p3_at(c, g, tile);
    for (int q = tid; q < OD * (H + 1); q += 256) {
        const int j = q / (H + 1), i = q - j * (H + 1);
        float s = 0.0f;
        for (int rr = 0; rr < HT; ++rr) s += douts[rr * 4 + j] * (i == H ? 1.0f : h2s[rr * (H + 1) + i]);
        p3[j * (H + 1) + i] = s;
    }
}

// fold dW3 partials in tile order (deterministic), Adam or gradient store
__global__ void __launch_bounds__(256) head_fold_kernel(FusedArgs a) {
    const copo_ppo_cfg& c = a.c;
    const int g = blockIdx.y, H = c.hidden;
    const copo_net_layout& L = net_of(c, g);
    const int tiles = (c.mb + HT - 1) / HT;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= L.out_dim * (H + 1)) return;
    const int j = q / (H + 1), i = q - j * (H + 1);
    float s = 0.0f;
    for (int t = 0; t < tiles; ++t) s += a.ws[ws_p3_at(c, g, t) + j * (H + 1) + i];
    const size_t idx = (i == H) ? (size_t)L.b3 + j : (size_t)L.w3 + (size_t)j * H + i;
    if (a.apply_adam) adam_update(a, idx, s);
    else a.grad[idx] = s;
}

__global__ void bump_kernel(int64_t* step, int64_t* k) {
    if (step) step[0] += 1;
    if (k) k[0] += 1;
}

__global__ void __launch_bounds__(256) adam_flat_kernel(FusedArgs a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) adam_update(a, (size_t)i, a.grad[i]);
}

// ------------------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------------------
size_t fused_ws_floats(const copo_ppo_cfg& c) {
    const int tiles = (c.mb + HT - 1) / HT;
    return (size_t)16 * c.mb * c.hidden + (size_t)4 * tiles * 4 * (c.hidden + 1);
}

hipError_t launch_fused_step(const FusedArgs& a, hipStream_t s) {
    const copo_ppo_cfg& c = a.c;
    const int G = a.groups, mt = (c.mb + TM - 1) / TM, ht = (c.hidden + TN - 1) / TN;
    FwdOp f1{a, 1}, f2{a, 2};
    int kmax1 = c.pol.in_dim;
    for (int g = 1; g < G; ++g) kmax1 = c.val[g - 1].in_dim > kmax1 ? c.val[g - 1].in_dim : kmax1;
    hipLaunchKernelGGL(gemm_kernel<FwdOp>, dim3(ht, mt, G), dim3(256), 0, s, f1, kmax1);
    hipLaunchKernelGGL(gemm_kernel<FwdOp>, dim3(ht, mt, G), dim3(256), 0, s, f2, c.hidden);
    const size_t lds = (size_t)(HT * (c.hidden + 1) + 4 * c.hidden + HT * 4 + 32) * sizeof(float);
    hipLaunchKernelGGL(head_kernel, dim3((c.mb + HT - 1) / HT, G), dim3(256), lds, s, a);
    BxOp bx{a};
    hipLaunchKernelGGL(gemm_kernel<BxOp>, dim3(ht, mt, G), dim3(256), 0, s, bx, c.hidden);
    BwOp bw2{a, 2}, bw1{a, 1};
    hipLaunchKernelGGL(gemm_kernel<BwOp>, dim3((c.hidden + 1 + TN - 1) / TN, ht, G), dim3(256), 0, s, bw2, c.mb);
    hipLaunchKernelGGL(gemm_kernel<BwOp>, dim3((kmax1 + 1 + TN - 1) / TN, ht, G), dim3(256), 0, s, bw1, c.mb);
    hipLaunchKernelGGL(head_fold_kernel, dim3((4 * (c.hidden + 1) + 255) / 256, G), dim3(256), 0, s, a);
    int64_t* st = a.apply_adam ? const_cast<int64_t*>(a.step) : nullptr;
    int64_t* kp = a.bump_k ? const_cast<int64_t*>(a.kptr) : nullptr;
    if (st || kp) hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, st, kp);
    return hipGetLastError();
}

hipError_t launch_adam_flat(const FusedArgs& a, long long n, hipStream_t s) {
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n);
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, const_cast<int64_t*>(a.step),
                       a.bump_k ? const_cast<int64_t*>(a.kptr) : nullptr);
    return hipGetLastError();
}

}  // namespace copo

// ---- C ABI ---------------------------------------------------------------------------------------------------
using namespace copo;

extern "C" int64_t copo_ppo_workspace_floats(const copo_ppo_cfg* cfg) { return cfg ? (int64_t)fused_ws_floats(*cfg) : -1; }

static int check_cfg(const copo_ppo_cfg* c) {
    if (!c) return COPO_ERR_NULL;
    if (c->mb < 1 || c->hidden < 1 || c->hidden > 1024 || c->act_dim != 2 || c->n_value_heads < 0 || c->n_value_heads > 3)
        return COPO_ERR_DIM;
    if (c->pol.out_dim != 4) return COPO_ERR_DIM;
    for (int g = 0; g < c->n_value_heads; ++g)
        if (c->val[g].out_dim != 1) return COPO_ERR_DIM;
    return COPO_OK;
}

extern "C" int copo_ppo_fused_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, float* grad,
                                       const float* obs_src, const float* cc_src, const float* pack_src,
                                       const int64_t* rows, const float* w, const float* denom, const float* kl_coeff,
                                       int64_t* step, float* workspace, float* stats, int32_t apply_adam,
                                       int32_t head_mode, int64_t* mb_index, int32_t bump_index, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != COPO_OK) return rc;
    if (!theta || !obs_src || !pack_src || !rows || !w || !denom || !workspace) return COPO_ERR_NULL;
    if (apply_adam && (!adam_m || !adam_v || !step)) return COPO_ERR_NULL;
    if (!apply_adam && !grad) return COPO_ERR_NULL;
    if (head_mode == COPO_HEAD_PPO && cfg->use_kl && !kl_coeff) return COPO_ERR_NULL;
    FusedArgs a;
    a.c = *cfg;
    a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v; a.grad = grad;
    a.obs_src = obs_src; a.cc_src = cc_src ? cc_src : obs_src; a.pack_src = pack_src;
    a.rows = rows; a.w = w; a.denom = denom; a.kl_coeff = kl_coeff; a.step = step;
    a.ws = workspace; a.stats = stats; a.apply_adam = apply_adam; a.head_mode = head_mode;
    a.groups = (head_mode == COPO_HEAD_PPO) ? 1 + cfg->n_value_heads : 1;
    a.kptr = mb_index; a.bump_k = (mb_index && bump_index) ? 1 : 0;
    hipError_t e = launch_fused_step(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_adam_step_f32(const copo_ppo_cfg* cfg, float* theta, float* adam_m, float* adam_v, const float* grad,
                                  int64_t n, int64_t* step, int64_t* mb_index, void* stream) {
    if (!cfg || !theta || !adam_m || !adam_v || !grad || !step) return COPO_ERR_NULL;
    if (n < 0) return COPO_ERR_DIM;
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    a.c = *cfg;
    a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v; a.grad = const_cast<float*>(grad); a.step = step;
    a.kptr = mb_index; a.bump_k = mb_index ? 1 : 0;
    hipError_t e = launch_adam_flat(a, n, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// Probe: C[8][64] = A[8][K] * B[K][64] with v_mfma_f32_4x4x1_16B_f32, A broadcast from one block to all sixteen (cbsz = 4):
// lane = output column, 4 rows per instruction -- the operand / accumulator layout the 8-row row pass relies on.
// build + run: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4 scripts/micro/mfma4x4_probe.hip && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x;
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
        const float a = A[(lane & 7) * K + k];        // lanes 0..3: rows 0..3 (block 0), lanes 4..7: rows 4..7 (block 1)
        const float b = B[k * 64 + lane];
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
    }
    for (int i = 0; i < 4; ++i) {
        C[i * 64 + lane] = c0[i];
        C[(4 + i) * 64 + lane] = c1[i];
    }
}
int main() {
    const int K = 96;
    std::vector<float> A(8 * K), B(K * 64), C(8 * 64), R(8 * 64, 0.f);
    for (size_t i = 0; i < A.size(); ++i) A[i] = sinf(0.37f * i);
    for (size_t i = 0; i < B.size(); ++i) B[i] = cosf(0.11f * i);
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 64; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[k * 64 + n]; R[m * 64 + n] = (float)s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (size_t i = 0; i < C.size(); ++i) worst = fmax(worst, fabs(C[i] - R[i]));
    printf("max |C - ref| = %.3g  (%s)\n", worst, worst < 1e-4 ? "layout as assumed" : "LAYOUT MISMATCH");
    return worst < 1e-4 ? 0 : 1;
}

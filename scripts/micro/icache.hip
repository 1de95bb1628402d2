// Is straight-line code executed once per workgroup paced by instruction issue (~4-5 cycles per instruction of a
// single wave) or by instruction-cache misses?  N dependent FMAs, fully unrolled, one wave per CU, launched back to
// back: time per launch vs code size.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void chain(float* out, float a, float b) {
    float x = a + threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, b, a + (float)(i & 7));
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int N>
void run(float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(chain<N>, dim3(256), dim3(64), 0, 0, out, 1.0f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(chain<N>, dim3(256), dim3(64), 0, 0, out, 1.0f, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("N=%6d dependent FMAs (~%d KB of code): %.2f us per launch -> %.2f ns per instruction beyond the 512-instruction kernel\n", N, N * 8 / 1024, ms * 10.0f, 0.0f);
}
int main() {
    float* out; hipMalloc(&out, 256 * 64 * 4);
    run<512>(out); run<2048>(out); run<4096>(out); run<8192>(out); run<16384>(out); run<32768>(out);
    return 0;
}

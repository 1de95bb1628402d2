// How many workgroups of the weight-gradient kernel's shape (512 threads, ~104 VGPRs, ~38 KB LDS) does a compute unit hold?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/occ scripts/micro/occupancy_probe.hip && /tmp/occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int V>
__global__ void __launch_bounds__(512) probe(long long* t, int spin) {
    extern __shared__ float lds[];
    if (V >= 100) asm volatile("v_mov_b32 v100, 0" ::: "v100");
    if (V >= 60 && V < 100) asm volatile("v_mov_b32 v60, 0" ::: "v60");
    if (threadIdx.x == 0) t[2 * blockIdx.x] = wall_clock64();
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(4); }
    if (threadIdx.x == 0) t[2 * blockIdx.x + 1] = wall_clock64() + (long long)lds[1] * 0;
}
template <int V>
void run(const char* name, size_t lds, int wgs) {
    long long* d; hipMalloc(&d, sizeof(long long) * 2 * wgs);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<V>, 512, lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<V>, dim3(wgs), dim3(512), lds, 0, d, 400);
    hipDeviceSynchronize();
    std::vector<long long> h(2 * wgs); hipMemcpy(h.data(), d, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    long long t0 = h[0]; for (int i = 0; i < wgs; ++i) t0 = std::min(t0, h[2 * i]);
    int early = 0; double last = 0; for (int i = 0; i < wgs; ++i) { double s = (h[2 * i] - t0) / 100.0; if (s < 2.0) ++early; last = std::max(last, s); }
    printf("%-28s lds %6zu B  occupancy API %d / CU   %d of %d workgroups started within 2 us (spin 4 us), last start %.2f us\n", name, lds, occ, early, wgs, last);
    hipFree(d);
}
int main() {
    run<100>("104 VGPRs", 33792 + 4104, 420);
    run<100>("104 VGPRs", 16 * 1024, 420);
    run<100>("104 VGPRs", 2048, 420);
    run<60>("64 VGPRs", 33792 + 4104, 420);
    run<60>("64 VGPRs", 2048, 420);
    run<0>("few VGPRs", 2048, 420);
    run<0>("few VGPRs", 2048, 512);
    return 0;
}

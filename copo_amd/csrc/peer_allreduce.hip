// Two-shot all-reduce over peer-mapped device memory (the per-minibatch gradient exchange of the
// data-parallel learner).  One process per GPU; every rank owns a WORKSPACE that all ranks of the node
// have mapped (hipIpc handles exchanged by the host side once):
//
//     [ data: n floats | inbox: world x shard floats | flags_in: world | flags_out: world | control: 8 words ]
//
// `data` is the buffer that is reduced IN PLACE (the fused learner writes its gradient sums straight into it).  One launch:
//   A  scatter      rank r stores shard s of its data into inbox[r] of rank s's workspace        (peer stores, 1/world of n each)
//      publish      the last workgroup to finish writes flags_in[r] = epoch into every workspace
//   B  reduce       rank r waits for its world flags_in, adds the world inbox rows in RANK ORDER (every rank gets bit-identical
//                   sums: each shard is added up once, by its owner) and stores the result into data[shard r] of EVERY workspace
//      publish      the last workgroup writes flags_out[r] = epoch into every workspace
//   C  wait         until the world flags_out of this rank say `epoch`: data holds the whole sum
// xGMI is point-to-point: in A and B every rank talks to every peer at once over its own link, 2 x (world - 1) / world x n
// floats per rank in total, two flag hops of latency.  Peer writes into `data` cannot race this rank's own scatter: an owner
// reduces only after it has seen this rank's flag, which goes out after all of this rank's scatter stores.
// The workspace is uncached device memory (hipDeviceMallocUncached): loads and stores of it are coherent across agents
// without kernel boundaries; release / acquire fences at system scope order data against flags.
// Spin waits are bounded (COPO_PEER_TIMEOUT_TICKS of the 100 MHz wall clock): a rank that never shows up turns into an error
// word in the control block, not a hung GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/copo_hip.h"

namespace copo {

constexpr int PEER_BLOCK = 256;
constexpr int PEER_GRID = 64;  // all workgroups spin together: stay far below one per compute unit
constexpr unsigned long long COPO_PEER_TIMEOUT_TICKS = 200000000ull;      // 2 s

struct PeerLayout {
    int64_t n, shard;
    int32_t world;
    __host__ __device__ PeerLayout(int64_t n_, int32_t world_) : n(n_), shard(((n_ + world_ - 1) / world_ + 3) & ~(int64_t)3), world(world_) {}
    __host__ __device__ size_t data() const { return 0; }
    __host__ __device__ size_t inbox(int p) const { return (size_t)(((n + 3) & ~(int64_t)3) + (int64_t)p * shard); }     // floats
    __host__ __device__ size_t flags_in() const { return inbox(world); }                                                  // words
    __host__ __device__ size_t flags_out() const { return flags_in() + 16; }
    __host__ __device__ size_t control() const { return flags_out() + 16; }       // {epoch, arrive_a, arrive_b, error, ...}
    __host__ __device__ size_t words() const { return control() + 8; }
};

struct PeerArgs {
    float* ws[COPO_PEER_MAX_WORLD];
    int64_t n;
    int32_t rank, world;
};

__device__ __forceinline__ unsigned long long peer_clock() { return wall_clock64(); }

__device__ __forceinline__ bool peer_wait(const uint32_t* flag, uint32_t want, uint32_t* err) {
    const unsigned long long t0 = peer_clock();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
        if (peer_clock() - t0 > COPO_PEER_TIMEOUT_TICKS) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}

// the last of the grid's workgroups to arrive returns true (and re-arms the counter)
__device__ __forceinline__ bool grid_arrive(uint32_t* counter) {
    __shared__ uint32_t last;
    // EVERY thread releases its own peer stores at system scope before the barrier: the workgroup barrier alone does not
    // wait for other waves' outstanding vector stores (vmcnt), so a fence by thread 0 only could publish the arrival
    // ahead of their scatter / reduce stores into xGMI memory
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (v == gridDim.x - 1) ? 1u : 0u;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return last != 0;
}

__device__ __forceinline__ void peer_allreduce_body(const PeerArgs& a, const int R) {
    const PeerLayout L(a.n, a.world);
    float* mine = a.ws[R];
    uint32_t* ctl = reinterpret_cast<uint32_t*>(mine + L.control());
    const uint32_t epoch = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;      // bumped at the end, by one thread
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const int W = a.world;
    // ---- A: scatter my shards --------------------------------------------------------------------------------------------
    const float4* d4 = reinterpret_cast<const float4*>(mine + L.data());
    const int64_t sh4 = L.shard >> 2, n4 = (a.n + 3) >> 2;
    for (int s = 0; s < W; ++s) {
        float4* dst = reinterpret_cast<float4*>(a.ws[s] + L.inbox(R));
        for (int64_t q = tid; q < sh4; q += nth) {
            const int64_t g = (int64_t)s * sh4 + q;
            dst[q] = g < n4 ? d4[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (grid_arrive(ctl + 1)) {
        if (threadIdx.x < W)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(a.ws[threadIdx.x] + L.flags_in()) + R, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- B: reduce my shard, in rank order, and hand it to everybody -----------------------------------------------------
    __shared__ uint32_t ok_s;
    if (threadIdx.x == 0) ok_s = 1u;
    __syncthreads();
    if (threadIdx.x < W) {
        if (!peer_wait(reinterpret_cast<const uint32_t*>(mine + L.flags_in()) + threadIdx.x, epoch, ctl + 3)) ok_s = 0u;
    }
    __syncthreads();
    __threadfence_system();           // acquire: the inbox rows behind the flags
    if (ok_s) {
        for (int64_t q = tid; q < sh4; q += nth) {
            const int64_t g = (int64_t)R * sh4 + q;
            if (g >= n4) break;
            float4 acc = reinterpret_cast<const float4*>(mine + L.inbox(0))[q];
            for (int p = 1; p < W; ++p) {
                const float4 v = reinterpret_cast<const float4*>(mine + L.inbox(p))[q];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            for (int s = 0; s < W; ++s) reinterpret_cast<float4*>(a.ws[s] + L.data())[g] = acc;
        }
    }
    if (grid_arrive(ctl + 2)) {
        if (threadIdx.x < W)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(a.ws[threadIdx.x] + L.flags_out()) + R, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) __hip_atomic_store(ctl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- C: everybody's shard has landed in my data ----------------------------------------------------------------------
    if (threadIdx.x < W) peer_wait(reinterpret_cast<const uint32_t*>(mine + L.flags_out()) + threadIdx.x, epoch, ctl + 3);
    __syncthreads();
    __threadfence_system();
}

__global__ void __launch_bounds__(PEER_BLOCK) peer_allreduce_kernel(PeerArgs a) { peer_allreduce_body(a, a.rank); }

// Every rank of the node in ONE launch (rank = blockIdx.y): what the `world` concurrent launches of a real job do, on a box
// with a single GPU -- streams of one process share hardware queues, so `world` separate launches there would queue up
// behind one another and wait for peers that cannot start.  Debug / test entry (copo_debug_peer_allreduce_all_ranks).
__global__ void __launch_bounds__(PEER_BLOCK) peer_allreduce_all_ranks_kernel(PeerArgs a) { peer_allreduce_body(a, (int)blockIdx.y); }

}  // namespace copo

using namespace copo;

extern "C" int64_t copo_peer_workspace_bytes(int64_t n, int32_t world) {
    if (n < 1 || world < 1 || world > COPO_PEER_MAX_WORLD) return -1;
    return (int64_t)(PeerLayout(n, world).words() * sizeof(float));
}

extern "C" int copo_peer_alloc(int64_t bytes, void** out) {
    if (!out || bytes < 1) return COPO_ERR_NULL;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) return COPO_ERR_DEVICE;
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return COPO_ERR_DEVICE;
    *out = p;
    return COPO_OK;
}

extern "C" int copo_peer_free(void* p) { return (!p || hipFree(p) == hipSuccess) ? COPO_OK : COPO_ERR_DEVICE; }

extern "C" int copo_ipc_export(void* dev_ptr, unsigned char* handle64) {
    if (!dev_ptr || !handle64) return COPO_ERR_NULL;
    static_assert(sizeof(hipIpcMemHandle_t) == COPO_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, dev_ptr) != hipSuccess) return COPO_ERR_DEVICE;
    memcpy(handle64, &h, sizeof(h));
    return COPO_OK;
}

extern "C" int copo_ipc_open(const unsigned char* handle64, void** out) {
    if (!handle64 || !out) return COPO_ERR_NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return COPO_ERR_DEVICE;
    *out = p;
    return COPO_OK;
}

extern "C" int copo_ipc_close(void* p) { return (!p || hipIpcCloseMemHandle(p) == hipSuccess) ? COPO_OK : COPO_ERR_DEVICE; }

extern "C" int copo_peer_allreduce_sum_f32(void* const* workspaces, int64_t n, int32_t rank, int32_t world, void* stream) {
    if (!workspaces) return COPO_ERR_NULL;
    if (n < 1 || world < 1 || world > COPO_PEER_MAX_WORLD || rank < 0 || rank >= world) return COPO_ERR_DIM;
    PeerArgs a;
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < world; ++r) {
        if (!workspaces[r]) return COPO_ERR_NULL;
        a.ws[r] = static_cast<float*>(workspaces[r]);
    }
    a.n = n; a.rank = rank; a.world = world;
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(PEER_GRID), dim3(PEER_BLOCK), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

extern "C" int copo_debug_peer_allreduce_all_ranks(void* const* workspaces, int64_t n, int32_t world, void* stream) {
    if (!workspaces) return COPO_ERR_NULL;
    if (n < 1 || world < 1 || world > COPO_PEER_MAX_WORLD) return COPO_ERR_DIM;
    PeerArgs a;
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < world; ++r) {
        if (!workspaces[r]) return COPO_ERR_NULL;
        a.ws[r] = static_cast<float*>(workspaces[r]);
    }
    a.n = n; a.rank = -1; a.world = world;
    hipLaunchKernelGGL(peer_allreduce_all_ranks_kernel, dim3(PEER_GRID, world), dim3(PEER_BLOCK), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? COPO_OK : COPO_ERR_DEVICE;
}

// error word of this rank's workspace (0 = every wait so far was answered); synchronises the stream first
extern "C" int copo_peer_status(void* workspace, int64_t n, int32_t world, void* stream) {
    if (!workspace) return COPO_ERR_NULL;
    uint32_t ctl[8];
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return COPO_ERR_DEVICE;
    if (hipMemcpy(ctl, static_cast<float*>(workspace) + PeerLayout(n, world).control(), sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess)
        return COPO_ERR_DEVICE;
    return ctl[3] ? COPO_ERR_DEVICE : COPO_OK;
}

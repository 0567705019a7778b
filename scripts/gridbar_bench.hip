// What does a device-wide barrier cost inside ONE persistent kernel on gfx950 -- against the 3.0 us a kernel boundary costs inside a
// hipGraph (scripts/skinny_bench.hip)?  256 workgroups (one per CU, cooperative launch), R rounds of: every thread stores a value
// that the NEXT round's readers in other workgroups (other XCDs: other L2s) must see, barrier, check.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 scripts/gridbar_bench.hip -o scripts/gridbar_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

struct Bar {
    unsigned int count;     // monotonically increasing arrivals
    unsigned int abort_;    // set when a spin runs away: everyone leaves
    unsigned int pad[30];
    unsigned int xcd[8][32];  // per-XCD arrival counters (hierarchical variant), one cache line apart
};

#define SPIN_LIMIT (1u << 22)

// flat: one counter, 256 arrivals per round
__device__ __forceinline__ bool grid_barrier_flat(Bar* b, unsigned int target) {
    __syncthreads();  // every wave's stores of this phase are issued and complete (vmcnt(0)) before thread 0 releases
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&b->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT || __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

// hierarchical: 32 workgroups of an XCD meet on their own counter, the last one of them arrives at the device counter
__device__ __forceinline__ bool grid_barrier_hier(Bar* b, unsigned int round, unsigned int wgs_per_xcd) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned int x = blockIdx.x & 7;
        const unsigned int prev = __hip_atomic_fetch_add(&b->xcd[x][0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == round * wgs_per_xcd - 1) __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&b->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round * 8) {
            if (++spins > SPIN_LIMIT || __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

// fences: 1 = one release fence before the arrival and one acquire fence after the wait (thread 0 only), the spin itself is relaxed;
//         0 = no fence at all (the cost of the atomics alone; data exchange then NOT guaranteed)
__device__ __forceinline__ bool grid_barrier_lean(Bar* b, unsigned int target, int fences) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&b->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// mode 4: lean barrier with fences + data exchange; 5: lean barrier, no fences, no data; 6: lean with fences, no data
// mode 0: flat barrier only; 1: flat + data exchange (each thread writes 16 B, reads its neighbour workgroup's next round);
// 2: hierarchical + data exchange; 3: flat + exchange + explicit __threadfence() by every thread on both sides
__global__ __launch_bounds__(512) void bar_kernel(Bar* b, float4* data, int rounds, int mode, unsigned int* errors) {
    const unsigned int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned int bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (mode >= 1 && mode <= 4) data[(size_t)(r & 1) * nwg * blockDim.x + (size_t)wg * blockDim.x + tid] = make_float4((float)r, (float)wg, (float)tid, 0.f);
        if (mode == 3) __threadfence();
        const bool ok = mode >= 4 ? grid_barrier_lean(b, (unsigned)r * nwg, mode != 5) : mode == 2 ? grid_barrier_hier(b, (unsigned)r, nwg / 8) : grid_barrier_flat(b, (unsigned)r * nwg);
        if (!ok) break;
        if (mode == 3) __threadfence();
        if (mode >= 1 && mode <= 4) {
            const unsigned int src = (wg + 1 + (r % 7) * 37) % nwg;  // a workgroup on another XCD most rounds
            const float4 v = data[(size_t)(r & 1) * nwg * blockDim.x + (size_t)src * blockDim.x + tid];  // double-buffered by round parity
            if (v.x != (float)r || v.y != (float)src) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    Bar* b;
    float4* data;
    unsigned int* err;
    const int nwg = 256, threads = 512;
    hipMalloc(&b, sizeof(Bar)); hipMalloc(&data, (size_t)2 * nwg * threads * sizeof(float4)); hipMalloc(&err, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int dev_coop = 0;
    hipDeviceGetAttribute(&dev_coop, hipDeviceAttributeCooperativeLaunch, 0);
    printf("cooperative launch supported: %d\n", dev_coop);
    for (int mode = 0; mode < 7; ++mode) {
        if (mode == 3) continue;
        for (int rounds : {2000}) {
            hipMemset(b, 0, sizeof(Bar)); hipMemset(err, 0, 4); hipMemset(data, 0, (size_t)2 * nwg * threads * sizeof(float4));
            int r = rounds, m = mode;
            void* args[] = {&b, &data, &r, &m, &err};
            hipEventRecord(e0, nullptr);
            hipError_t rc = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(bar_kernel), dim3(nwg), dim3(threads), args, 0, nullptr);
            hipEventRecord(e1, nullptr);
            if (rc != hipSuccess) { printf("mode %d: launch failed: %s\n", mode, hipGetErrorString(rc)); return 1; }
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            Bar hb;
            unsigned int herr;
            hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost); hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("mode %d rounds %4d: %8.3f ms total, %6.3f us per round | abort %u | stale reads %u\n", mode, rounds, ms, ms * 1000.f / rounds, hb.abort_, herr);
        }
    }
    return 0;
}

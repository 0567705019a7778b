// What the 256 MiB Infinity Cache does for the policy step's update: the step ends with dWi (writes 101 MB of f32 gradient), the norm pass (reads g and p, 101 MB
// each) and Adam (reads p g m v, writes p m v + a 16-bit mirror).  Questions, each timed with HIP events around ONE launch:
//   1. does a buffer a kernel just WROTE serve the next kernel's reads from the cache?            (write 101 MB, read it)
//   2. the norm pass behind the gradient write: g cached, p from HBM?                             (write g, read g + p)
//   3. Adam behind the norm pass, walking front to back vs back to front
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 scripts/l3_probe.hip -o scripts/l3_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

static __global__ __launch_bounds__(256) void write_kernel(float4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
static __global__ __launch_bounds__(256) void read1_kernel(const float4* __restrict__ a, size_t n4, float* out) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n4; i += stride) s += a[i].x;
    if (s == 12345.678f) out[0] = s;
}
static __global__ __launch_bounds__(256) void read2_kernel(const float4* __restrict__ a, const float4* __restrict__ b, size_t n4, float* out) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = a[i + u * stride]; w[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (v[u].x * w[u].x + v[u].y * w[u].y) + (v[u].z * w[u].z + v[u].w * w[u].w);
    }
    for (; i < n4; i += stride) s += a[i].x * b[i].x;
    if (s == 12345.678f) out[0] = s;
}
// the update's access pattern: one float4 per thread of p g m v in, p m v (+ 8 bytes of mirror) out
static __global__ __launch_bounds__(256) void adam_like_kernel(float4* p, const float4* __restrict__ g, float4* m, float4* v, uint2* mirror, size_t n4, int reverse) {
    const size_t i = (size_t)(reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 gv = g[i];
    float4 pv = p[i], mv = m[i], vv = v[i];
    mv.x = 0.9f * mv.x + 0.1f * gv.x; mv.y = 0.9f * mv.y + 0.1f * gv.y; mv.z = 0.9f * mv.z + 0.1f * gv.z; mv.w = 0.9f * mv.w + 0.1f * gv.w;
    vv.x = 0.99f * vv.x + 0.01f * gv.x * gv.x; vv.y = 0.99f * vv.y + 0.01f * gv.y * gv.y; vv.z = 0.99f * vv.z + 0.01f * gv.z * gv.z; vv.w = 0.99f * vv.w + 0.01f * gv.w * gv.w;
    pv.x -= 1e-3f * mv.x / (sqrtf(vv.x) + 1e-8f); pv.y -= 1e-3f * mv.y / (sqrtf(vv.y) + 1e-8f); pv.z -= 1e-3f * mv.z / (sqrtf(vv.z) + 1e-8f); pv.w -= 1e-3f * mv.w / (sqrtf(vv.w) + 1e-8f);
    m[i] = mv; v[i] = vv; p[i] = pv;
    mirror[i] = make_uint2(__float_as_uint(pv.x) >> 16 | (__float_as_uint(pv.y) & 0xffff0000u), __float_as_uint(pv.z) >> 16 | (__float_as_uint(pv.w) & 0xffff0000u));
}

int main() {
    const size_t n = (size_t)128 * 197376, n4 = n / 4;  // image_text_input/kernel: 25.3 M parameters, 101 MB of f32
    float *p, *g, *m, *v, *junk, *out;
    uint2* mirror;
    const size_t junk_n4 = (size_t)640 << 20 >> 4;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&mirror, n * 2); hipMalloc(&junk, junk_n4 * 16); hipMalloc(&out, 64);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto flush = [&]() { hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (float4*)junk, junk_n4, 1.f); };
    auto wr = [&](float* b) { hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (float4*)b, n4, 0.5f); };
    auto rd1 = [&](float* a) { hipLaunchKernelGGL(read1_kernel, dim3(1024), dim3(256), 0, 0, (const float4*)a, n4, out); };
    auto rd2 = [&](float* a, float* b) { hipLaunchKernelGGL(read2_kernel, dim3(1024), dim3(256), 0, 0, (const float4*)a, (const float4*)b, n4, out); };
    auto adam = [&](int rev) { hipLaunchKernelGGL(adam_like_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, mirror, n4, rev); };
    // time the LAST launch of `seq` (everything before it sets the cache state), median of 7
    auto timed = [&](const char* name, double mb, auto&& before, auto&& last) {
        std::vector<float> t;
        for (int r = 0; r < 7; ++r) {
            flush();
            before();
            hipEventRecord(e0, 0);
            last();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-64s %7.1f us   %5.2f TB/s of %.0f MB\n", name, t[3] * 1e3, mb / (t[3] * 1e-3) / 1e12, mb / 1e6);
        fflush(stdout);
    };
    const double MB = n * 4.0;
    timed("read g, cold (640 MB written in between)", MB, [&] {}, [&] { rd1(g); });
    timed("read g right after reading g", MB, [&] { rd1(g); }, [&] { rd1(g); });
    timed("read g right after WRITING g", MB, [&] { wr(g); }, [&] { rd1(g); });
    timed("read g + p, cold", 2 * MB, [&] {}, [&] { rd2(g, p); });
    timed("read g + p right after writing g", 2 * MB, [&] { wr(g); }, [&] { rd2(g, p); });
    timed("read g + p right after reading g + p", 2 * MB, [&] { rd2(g, p); }, [&] { rd2(g, p); });
    timed("adam, cold, forward", 7.5 * MB, [&] {}, [&] { adam(0); });
    timed("adam, cold, reverse", 7.5 * MB, [&] {}, [&] { adam(1); });
    timed("adam forward behind (write g; read g + p)", 7.5 * MB, [&] { wr(g); rd2(g, p); }, [&] { adam(0); });
    timed("adam reverse behind (write g; read g + p)", 7.5 * MB, [&] { wr(g); rd2(g, p); }, [&] { adam(1); });
    // the three launches together, both orders
    timed("write g; read g + p; adam forward   (all three)", 10.5 * MB, [&] {}, [&] { wr(g); rd2(g, p); adam(0); });
    timed("write g; read g + p; adam reverse   (all three)", 10.5 * MB, [&] {}, [&] { wr(g); rd2(g, p); adam(1); });
    return 0;
}

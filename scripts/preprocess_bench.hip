// Standalone timing / checksum harness for the frame preprocessing kernels (preprocess.h): 1024 frames 256x256x3 -> 224x224 im2col f16.
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. ../../scripts/preprocess_bench.hip -o ../../scripts/preprocess_bench.bin
// ARP_PRE_LDS_KB=<n> changes the LDS budget per workgroup (rows per tile), ARP_PREPROCESS_GENERIC=1 runs the generic instance.
#include <cstdio>
#include <vector>

#include "preprocess_host.h"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
}  // namespace arp
using namespace arp;

static uint64_t fnv(const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, H = 256, W = 256, R = 224, P = 32;
    std::vector<uint8_t> fr((size_t)n * H * W * 3);
    uint32_t s = 777u;
    for (size_t i = 0; i < fr.size(); ++i) {  // blocky "game-like" content with noise: neighbouring pixels correlate
        s = s * 1664525u + 1013904223u;
        const size_t px = i / 3;
        fr[i] = (uint8_t)((((px / 8) * 37 + (px / (256 * 8)) * 101 + (i % 3) * 53) & 0xff) ^ ((s >> 24) & 0x1f));
    }
    uint8_t* dF;
    void* dO;
    float* dL;
    hipMalloc(&dF, fr.size());
    hipMalloc(&dO, (size_t)n * R * R * 3 * 2);
    hipMalloc(&dL, 768 * 4);
    hipMemcpy(dF, fr.data(), fr.size(), hipMemcpyHostToDevice);
    float lut[768];
    build_lut(lut);
    hipMemcpy(dL, lut, sizeof(lut), hipMemcpyHostToDevice);
    for (int use_crop = 0; use_crop < 2; ++use_crop) {
        ResizePlan p;
        if (build_plan(H, W, use_crop, R, p)) return 1;
        auto go = [&]() { return launch_preprocess<f16_t, PRE_PATCH>(p, dF, n, P, dL, dO, nullptr); };
        if (go()) return 1;
        hipDeviceSynchronize();
        std::vector<uint8_t> out((size_t)n * R * R * 3 * 2);
        hipMemcpy(out.data(), dO, out.size(), hipMemcpyDeviceToHost);
        const uint64_t sum = fnv(out.data(), out.size());
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) go();
        const int iters = 20;
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) go();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= iters;
        const double bytes = (double)n * (H * W * 3 + R * R * 3 * 2);
        printf("use_crop %d: kmax %d/%d TR %d rows %d lds %zu B: %7.1f us per %d frames, %6.2f TB/s algorithmic, fnv %016llx\n", use_crop, p.kmax_h, p.kmax_v, p.TR,
               p.max_rows, p.lds_bytes, ms * 1e3, n, bytes / ms / 1e9, (unsigned long long)sum);
    }
    return 0;
}

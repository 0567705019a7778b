// A/B harness: the four-wave 256x256 GEMM (gemm4w.h) against the eight-wave one (gemm256.h), same operands, one process, interleaved
// rounds; FNV checksums of the outputs must agree bit for bit.
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. -I../../scripts ../../scripts/gemm4w_bench.hip -o ../../scripts/gemm4w_bench.bin
#include <algorithm>
#include <cstdio>
#include <vector>

#include "gemm256.h"
#include "gemm4w.h"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
int launch_gemm2w_dyn(int, int, int, int, const GemmArgs&, hipStream_t) { return fail("gemm2w is not linked into this harness"); }
bool gemm2w_has(int, int, int, int) { return false; }
}  // namespace arp
using namespace arp;

static uint64_t fnv(const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

template <typename OutT, int ACT, bool RESID> static void run(const char* name, int M, int N, int K) {
    std::vector<f16_t> hA((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N), hr(RESID ? (size_t)M * N : 0);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    for (auto& v : hr) v = rnd();
    void *dA, *dW, *dO, *dR, *dR0;
    float* dB;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dB, N * 4); hipMalloc(&dO, (size_t)M * N * 4); hipMalloc(&dR, (size_t)M * N * 4);
    hipMalloc(&dR0, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice);
    if (RESID) hipMemcpy(dR0, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    GemmArgs g;
    g.A = dA; g.W = dW; g.bias = dB; g.resid = RESID ? (float*)dR : nullptr; g.out = RESID ? dR : dO;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
    if (const char* e = getenv("G4_FLAGS")) g.flags = atoi(e);
    auto go8 = [&]() { return launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6>(g, nullptr); };
    auto go4 = [&]() { return launch_gemm4w<f16_t, OutT, ACT, RESID>(g, nullptr); };
    uint64_t sum[2];
    std::vector<uint8_t> out((size_t)M * N * sizeof(OutT));
    for (int v = 0; v < 2; ++v) {
        if (RESID) hipMemcpy(dR, dR0, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
        else hipMemset(dO, 0, (size_t)M * N * 4);
        if ((v ? go4() : go8()) != 0) { printf("%s: launch failed\n", name); return; }
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel %d failed: %s\n", name, v, hipGetErrorString(hipGetLastError())); return; }
        hipMemcpy(out.data(), g.out, out.size(), hipMemcpyDeviceToHost);
        sum[v] = fnv(out.data(), out.size());
    }
#ifdef ARP_G4_STAMPS
    {
        const int ntile = ((M + 255) / 256) * ((N + 255) / 256);
        long long* dS;
        hipMalloc(&dS, (size_t)ntile * 16 * 8);
        hipMemset(dS, 0, (size_t)ntile * 16 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(arp_g4_stamps), &dS, sizeof(dS));
        go4();
        hipDeviceSynchronize();
        std::vector<long long> h((size_t)ntile * 16);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        double d[3] = {0, 0, 0};
        for (int t = 0; t < ntile; ++t)
            for (int w = 0; w < 4; ++w)
                for (int i = 0; i < 3; ++i) d[i] += (double)(h[((size_t)t * 4 + w) * 4 + i + 1] - h[((size_t)t * 4 + w) * 4 + i]);
        printf("  stamps4 %-10s: kloop %.0f | stage %.0f | rest %.0f cycles (MFMA floor %d)\n", name, d[0] / ntile / 4, d[1] / ntile / 4, d[2] / ntile / 4, K / 64 * 2048);
        long long* nul = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(arp_g4_stamps), &nul, sizeof(nul));
        hipFree(dS);
    }
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best[2] = {1e9f, 1e9f}, med[2];
    std::vector<float> all[2];
    for (int round = 0; round < 5; ++round)
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 2; ++i) v ? go4() : go8();
            const int iters = 10;
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) v ? go4() : go8();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= iters;
            best[v] = std::min(best[v], ms);
            all[v].push_back(ms);
        }
    for (int v = 0; v < 2; ++v) { std::sort(all[v].begin(), all[v].end()); med[v] = all[v][all[v].size() / 2]; }
    const double fl = 2.0 * M * N * K;
    printf("%-10s M=%d N=%d K=%d: 8w %7.1f us (min %7.1f) %6.1f TF | 4w %7.1f us (min %7.1f) %6.1f TF | %+5.1f %% | fnv %s %016llx\n", name, M, N, K, med[0] * 1e3,
           best[0] * 1e3, fl / med[0] / 1e9, med[1] * 1e3, best[1] * 1e3, fl / med[1] / 1e9, (med[0] / med[1] - 1.0) * 100.0, sum[0] == sum[1] ? "same" : "DIFF",
           (unsigned long long)sum[1]);
    hipFree(dA); hipFree(dW); hipFree(dB); hipFree(dO); hipFree(dR); hipFree(dR0);
}

int main() {
    run<f16_t, ACT_NONE, false>("ragged", 1000, 520, 192);
    run<f16_t, ACT_NONE, false>("k64", 512, 512, 64);
    run<f16_t, ACT_NONE, false>("k128", 512, 512, 128);
    run<f16_t, ACT_NONE, false>("sq4096", 4096, 4096, 4096);
    run<f16_t, ACT_NONE, false>("qkv", 51200, 2304, 768);
    run<f16_t, ACT_QGELU, false>("c_fc", 51200, 3072, 768);
    run<float, ACT_NONE, true>("c_proj", 51200, 768, 3072);
    run<float, ACT_NONE, true>("out_proj", 51200, 768, 768);
    run<f16_t, ACT_QGELU, false>("c_fc_half", 25600, 3072, 768);
    run<float, ACT_NONE, true>("c_proj_half", 25600, 768, 3072);
    return 0;
}

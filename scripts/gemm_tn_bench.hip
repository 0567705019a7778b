// Standalone check + timing harness for the TN weight-gradient GEMMs (gemm_tn.hip): both tile sizes against a CPU reference on a
// small shape, then the policy step's adapter shape (768 x 768 x 32 896) over a sweep of K-splits, each with its split-K reduction.
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. ../../scripts/gemm_tn_bench.hip -o ../../scripts/gemm_tn_bench.bin
#include <cmath>
#include <cstdio>
#include <vector>

#include "gemm_tn.hip"
#include "dtops.h"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
}  // namespace arp
using namespace arp;

static float h2f(f16_t h) {
    _Float16 x;
    memcpy(&x, &h, 2);
    return (float)x;
}

struct Case { int M, N, K, S, tile256, xcd; int ld0 = 0; };

static double run(const Case& c, bool check, int iters) {
    const int M = c.M, N = c.N, K = c.K;
    std::vector<f16_t> hA((size_t)K * M), hB((size_t)K * N);
    uint32_t s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hB) v = host_f2h(rnd() * 0.25f);
    void *dA, *dB;
    float *dP, *dO;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2);
    hipMalloc(&dP, (size_t)c.S * M * N * 4); hipMalloc(&dO, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    GemmTnArgs g;
    g.A = dA; g.B = dB; g.M = M; g.N = N; g.K = K; g.lda = c.ld0 ? 0 : M; g.ldb = c.ld0 ? 0 : N; g.ldo = N; g.ksplit = c.S; g.tile256 = c.tile256; g.xcd_slices = c.xcd;
    const size_t MN = (size_t)M * N;
    auto go = [&]() {
        if (c.S == 1) {
            g.out = dO; g.slice_stride = 0; g.alpha = 0.5f;
            return launch_gemm_tn(2, g, nullptr);
        }
        g.out = dP; g.slice_stride = MN; g.alpha = 1.f;
        if (launch_gemm_tn(2, g, nullptr)) return -1;
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((MN + 63) / 64), dim3(256), 0, nullptr, dP, c.S, MN, N, nullptr, ACT_NONE, dO, nullptr, 0, 0.5f);
        return 0;
    };
#ifdef ARP_TN_STAMPS
    long long* dS;
    hipMalloc(&dS, 4096 * 4 * 8);
    hipMemset(dS, 0, 4096 * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(arp_tn_stamps), &dS, sizeof(dS));
#endif
    if (go()) exit(1);
    hipDeviceSynchronize();
#ifdef ARP_TN_STAMPS
    if (c.tile256 && !check) {
        for (int i = 0; i < 50; ++i) go();  // warm clocks
        hipDeviceSynchronize();
        std::vector<long long> h(4096 * 4);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, ticks = 0, tiles = 0; int n = 0;
        for (int b = 0; b < 4096; ++b) if (h[b * 4 + 2] > 0) { cyc += h[b * 4]; ticks += h[b * 4 + 1]; tiles += h[b * 4 + 2]; ++n; }
        printf("  stamps: %d workgroups, %.0f shader cycles per K-tile, clock %.2f GHz\n", n, cyc / tiles, cyc / ticks * 0.1);
    }
    hipFree(dS);
#endif
    double maxerr = 0;
    if (check) {
        std::vector<float> hO(MN);
        hipMemcpy(hO.data(), dO, MN * 4, hipMemcpyDeviceToHost);
        std::vector<float> fa(hA.size()), fb(hB.size());
        for (size_t i = 0; i < hA.size(); ++i) fa[i] = h2f(hA[i]);
        for (size_t i = 0; i < hB.size(); ++i) fb[i] = h2f(hB[i]);
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 5) {
                double r = 0;
                for (int k = 0; k < K; ++k) r += (double)fa[(size_t)k * M + m] * fb[(size_t)k * N + n];
                maxerr = std::max(maxerr, std::fabs(0.5 * r - hO[(size_t)m * N + n]));
            }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    if (c.ld0) printf("[every K-tile re-reads row 0: cache-resident operands] ");
    printf("M=%d N=%d K=%d S=%2d tile=%s xcd=%d: %7.1f us  %7.1f TFLOP/s", M, N, K, c.S, c.tile256 ? "256" : "128", c.xcd, us, 2.0 * M * N * K / us / 1e6);
    if (check) printf("  max |err| = %.3g", maxerr);
    printf("\n");
    hipFree(dA); hipFree(dB); hipFree(dP); hipFree(dO);
    return maxerr;
}

int main() {
    // correctness: ragged K-splits, one / many tiles, K-tile counts below the ring depth
    const Case checks[] = {{256, 256, 64, 1, 1, 0},   {256, 256, 128, 1, 1, 0},  {256, 512, 704, 1, 1, 0},  {512, 256, 1344, 5, 1, 0},
                           {768, 768, 2112, 8, 1, 1}, {768, 768, 2112, 16, 1, 1}, {256, 256, 1344, 5, 0, 0}, {768, 768, 2112, 24, 1, 1},
                           {128, 1280, 128, 1, 0, 0}};  // (the last: dWi's shape class -- one row of 128 x 128 tiles, two K-tiles, staged f32 epilogue)
    bool ok = true;
#ifndef TW_ABL
    for (const Case& c : checks) ok &= run(c, true, 2) < 0.05;
#endif
    printf(ok ? "CHECK OK\n" : "CHECK FAILED\n");
    const int K = 32896;
#ifdef TW_ABL
    printf("ABLATION build TW_ABL=%d (1: no LDS reads, 2: no LDS-DMA in the loop, 4: no barrier) -- timings only\n", TW_ABL);
    const Case shapes[] = {{768, 768, K, 8, 1, 1, 1}, {768, 768, K, 24, 1, 1, 1}};
#else
    const Case shapes[] = {{768, 768, K, 14, 0, 0}, {768, 768, K, 28, 1, 0}, {768, 768, K, 24, 1, 0}, {768, 768, K, 24, 1, 1},
                           {768, 768, K, 16, 1, 1}, {768, 768, K, 16, 1, 0}, {768, 768, K, 56, 1, 0}, {768, 768, K, 8, 1, 1},
                           {768, 768, K, 24, 1, 1, 1}, {768, 768, K, 8, 1, 1, 1}, {768, 768, K, 14, 0, 0, 1}};
#endif
    for (const Case& c : shapes) run(c, false, 20);
    run(Case{128, 197376, 128, 1, 0, 0}, false, 20);  // image_text_input's weight gradient at B = 32
    return ok ? 0 : 1;
}

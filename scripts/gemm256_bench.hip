// Standalone timing / checksum harness for gemm256.h variants (kernel development loop: 20 s per build instead of the library's 4 min):
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. [-DARP_G2_...=..] ../../scripts/gemm256_bench.hip -o ../../scripts/gemm256_bench.bin
// Prints microseconds, TFLOP/s and an FNV checksum of the output per shape (variants that keep the MFMA order must agree bit for bit).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "gemm256.h"

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
    std::vector<float> hb(N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    void *dA, *dW, *dO, *dR;
    float* dB;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dB, N * 4); hipMalloc(&dO, (size_t)M * N * 4); hipMalloc(&dR, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dR, 0, (size_t)M * N * 4);
    GemmArgs g;
    g.A = dA; g.W = dW; g.bias = dB; g.resid = RESID ? (float*)dR : nullptr; g.out = RESID ? dR : dO;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
    if (getenv("G256_LDA0")) g.lda = 0;  // every A row is row 0: the A stream comes from cache (isolates memory latency from loop mechanics)
    if (const char* e = getenv("G256_STAGGER")) { g.stagger_groups = atoi(e); g.stagger_cycles = getenv("G256_STAGGER_CYCLES") ? atoi(getenv("G256_STAGGER_CYCLES")) : 100000; }
    if (const char* e = getenv("G256_FLAGS")) g.flags = atoi(e);  // 1 = ablate the epilogue stores (K loop only), 2 = unstaged stores
#ifdef ARP_G2_FINE
    {   // -DARP_G2_FINE: s_memtime stamps around every segment of the middle K-tile, per wave group (wr = 0: waves 0-3, wr = 1: waves 4-7)
        const int ntile = ((M + 255) / 256) * ((N + 255) / 256);
        long long* dS;
        hipMalloc(&dS, (size_t)ntile * 128 * 8);
        hipMemset(dS, 0, (size_t)ntile * 128 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(arp_g2_stamps), &dS, sizeof(dS));
        launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6>(g, nullptr);
        hipDeviceSynchronize();
        std::vector<long long> h((size_t)ntile * 128);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        const char* nm[8] = {"A:reads+issue", "A:wait+barrier", "A:mfma(+bar)", "A:bar2", "B:reads+issue", "B:wait+barrier", "B:mfma(+bar)", "B:bar2"};
        for (int grp = 0; grp < 2; ++grp) {
            double d[8] = {0};
            for (int t = 0; t < ntile; ++t)
                for (int w = grp * 4; w < grp * 4 + 4; ++w) {
                    const long long* f = &h[((size_t)t * 8 + w) * 16];
                    long long prev = f[0];
                    for (int i = 1; i < 9; ++i) {
                        if (f[i] == 0) continue;  // stamp not taken in this build
                        d[i - 1] += (double)(f[i] - prev);
                        prev = f[i];
                    }
                }
            printf("  fine %-10s wr=%d:", name, grp);
            double tot = 0;
            for (int i = 0; i < 8; ++i) { printf(" %s %.0f |", nm[i], d[i] / ntile / 4); tot += d[i] / ntile / 4; }
            printf(" K-tile %.0f cycles\n", tot);
        }
        hipFree(dS);
        return;
    }
#endif
#ifdef ARP_G2_STAMPS
    const int ntile = ((M + 255) / 256) * ((N + 255) / 256);
    long long* dS;
    hipMalloc(&dS, (size_t)ntile * 64 * 8);
    hipMemset(dS, 0, (size_t)ntile * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(arp_g2_stamps), &dS, sizeof(dS));
#endif
    auto go = [&]() { return launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6>(g, nullptr); };
    go();
    hipDeviceSynchronize();
#ifdef ARP_G2_STAMPS
    {   // per-tile phase durations in shader cycles, averaged over tiles and waves: K loop | stage | barrier | copy-out issue | drain
        std::vector<long long> h((size_t)ntile * 64);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        double d[5] = {0, 0, 0, 0, 0}, epi = 0;
        for (int t = 0; t < ntile; ++t)
            for (int w = 0; w < 8; ++w) {
                for (int i = 0; i < 5; ++i) d[i] += (double)(h[((size_t)t * 8 + w) * 8 + i + 1] - h[((size_t)t * 8 + w) * 8 + i]);
                epi += (double)(h[((size_t)t * 8 + w) * 8 + 4] - h[((size_t)t * 8 + w) * 8 + 1]);
            }
        if (sizeof(OutT) == 4) {  // the f32 path takes no mid-epilogue stamps: K loop | whole epilogue | drain
            printf("  stamps %-10s: kloop %.0f | epilogue (two staged passes) %.0f | drain %.0f cycles\n", name, d[0] / ntile / 8, epi / ntile / 8, d[4] / ntile / 8);
            hipFree(dS);
            return;
        }
        printf("  stamps %-10s: kloop %.0f | stage %.0f | barrier %.0f | copy-out issue %.0f | drain %.0f cycles\n", name, d[0] / ntile / 8, d[1] / ntile / 8,
               d[2] / ntile / 8, d[3] / ntile / 8, d[4] / ntile / 8);
        hipFree(dS);
    }
    return;
#endif
    // ---- both MFMA shapes in ONE process, interleaved rounds (cdna_hip_programming.md rule 24), random operands (rule 25) ----
    auto go16 = [&]() { return launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6, false>(g, nullptr); };
#ifdef G256_AB_KV  // round 5: the second arm is the KV = 1 K loop (SADDR LDS-DMA statements, peeled steady state) on the SAME MFMA shape; columns keep their names
    auto go32 = [&]() { return launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6, false, 1>(g, nullptr); };
#else
    auto go32 = [&]() { return launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6, true>(g, nullptr); };
#endif
    std::vector<uint8_t> out16((size_t)M * N * sizeof(OutT)), out32(out16.size());
    auto h2f = [](f16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; };
    auto as_f = [&](const std::vector<uint8_t>& o, size_t i) { if (sizeof(OutT) == 4) { float f; memcpy(&f, &o[i * 4], 4); return f; } f16_t h; memcpy(&h, &o[i * 2], 2); return h2f(h); };
    if (RESID) hipMemset(dR, 0, (size_t)M * N * 4);
    go16(); hipDeviceSynchronize();
    hipMemcpy(out16.data(), g.out, out16.size(), hipMemcpyDeviceToHost);
    if (RESID) hipMemset(dR, 0, (size_t)M * N * 4);
    go32(); hipDeviceSynchronize();
    hipMemcpy(out32.data(), g.out, out32.size(), hipMemcpyDeviceToHost);
    double maxd = 0, maxref16 = 0, maxref32 = 0;
    for (size_t i = 0; i < (size_t)M * N; i += 7) maxd = std::max(maxd, (double)std::fabs(as_f(out16, i) - as_f(out32, i)));
    {   // a sample of entries against a float64 reference (the 32x32 layout is new: rows / columns / k-steps must all line up)
        uint32_t t = 777u;
        for (int c = 0; c < 400; ++c) {
            t = t * 1664525u + 1013904223u; const int m = (c < 40) ? (M - 1 - c * 7 % std::min(M, 300)) : (int)((t >> 8) % M);
            t = t * 1664525u + 1013904223u; const int n = (c < 40) ? (N - 1 - c * 5 % std::min(N, 300)) : (int)((t >> 8) % N);
            double r = hb[n];
            for (int k = 0; k < K; ++k) r += (double)h2f(hA[(size_t)m * (getenv("G256_LDA0") ? 0 : K) + k]) * h2f(hW[(size_t)n * K + k]);
            if (ACT == ACT_QGELU) r = r / (1.0 + std::exp(-1.702 * r));
            maxref16 = std::max(maxref16, std::fabs(r - as_f(out16, (size_t)m * N + n)) / (1.0 + std::fabs(r)));
            maxref32 = std::max(maxref32, std::fabs(r - as_f(out32, (size_t)m * N + n)) / (1.0 + std::fabs(r)));
        }
    }
    const uint64_t sum = fnv(out16.data(), out16.size());
#ifdef ARP_G2_CLOCK
    long long* dS;
    const int nblk = ((M + 255) / 256) * ((N + 255) / 256);
    hipMalloc(&dS, (size_t)nblk * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(arp_g2_stamps), &dS, sizeof(dS));
    auto clock_of = [&](auto&& fn) {
        for (int i = 0; i < 200; ++i) fn();  // the chip settles under the load first
        hipDeviceSynchronize();
        std::vector<long long> h((size_t)nblk * 2);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int b = 0; b < nblk; ++b) if (h[2 * b + 1] > 0) c.push_back(100.0 * h[2 * b] / h[2 * b + 1]);
        std::sort(c.begin(), c.end());
        return c.empty() ? 0.0 : c[c.size() / 2];
    };
    const double mhz16 = clock_of(go16), mhz32 = clock_of(go32);
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) { go16(); go32(); }
    const int iters = 10, rounds = 7;
    std::vector<float> t16, t32;
    for (int r = 0; r < rounds; ++r) {
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) { if (v) go32(); else go16(); }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            (v ? t32 : t16).push_back(ms / iters);
        }
    }
    std::sort(t16.begin(), t16.end()); std::sort(t32.begin(), t32.end());
    const float m16 = t16[rounds / 2], m32 = t32[rounds / 2];
    printf("%-11s M=%d N=%d K=%d: 16x16x32 %7.1f us (min %7.1f) %7.1f TF | 32x32x16 %7.1f us (min %7.1f) %7.1f TF | 32/16 %.3f | err vs f64: %.1e / %.1e, |16-32| %.1e  fnv16 %016llx\n",
           name, M, N, K, m16 * 1e3, t16[0] * 1e3, 2.0 * M * N * K / m16 / 1e9, m32 * 1e3, t32[0] * 1e3, 2.0 * M * N * K / m32 / 1e9, m32 / m16,
           maxref16, maxref32, maxd, (unsigned long long)sum);
#ifdef ARP_G2_CLOCK
    printf("            in-kernel clock (median over workgroups, after 200 launches): 16x16x32 %.0f MHz | 32x32x16 %.0f MHz\n", mhz16, mhz32);
    hipFree(dS);
#endif
    hipFree(dA); hipFree(dW); hipFree(dB); hipFree(dO); hipFree(dR);
}

// the masked epilogue (out = value * (mask > 0), column sums of the stored values per 256-row tile) against a CPU reference
static bool check_masked(int M, int N, int K) {
    std::vector<f16_t> hA((size_t)M * K), hW((size_t)N * K), hM((size_t)M * N);
    uint32_t s = 4242u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hM) { const float r = rnd(); v = host_f2h(r > 0.f ? r : (r < -0.9f ? -0.0f : 0.f)); }  // relu-like: positives, +0 and a few -0
    const int mt = (M + 255) / 256;
    void *dA, *dW, *dO, *dM;
    float* dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dO, (size_t)M * N * 2); hipMalloc(&dM, hM.size() * 2); hipMalloc(&dC, (size_t)mt * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dM, hM.data(), hM.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dO, 0xff, (size_t)M * N * 2);
    GemmArgs g;
    g.A = dA; g.W = dW; g.out = dO; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
    g.mask = dM; g.ldm = N; g.colsum_part = dC;
    if (launch_gemm256_nt<f16_t, f16_t, ACT_NONE, false, 16>(g, nullptr)) return false;
    if (hipDeviceSynchronize() != hipSuccess) { printf("masked M=%d N=%d K=%d: launch failed\n", M, N, K); return false; }
    std::vector<f16_t> hO((size_t)M * N);
    std::vector<float> hC((size_t)mt * N);
    hipMemcpy(hO.data(), dO, hO.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    auto h2f = [](f16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; };
    double maxerr = 0, maxc = 0;
    long bad_mask = 0;
    std::vector<double> cs((size_t)mt * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float o = h2f(hO[(size_t)m * N + n]);
            cs[(size_t)(m / 256) * N + n] += o;
            if (!(h2f(hM[(size_t)m * N + n]) > 0.f)) { bad_mask += (o != 0.f); continue; }
            if ((m * 7 + n) % 13) continue;
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)h2f(hA[(size_t)m * K + k]) * h2f(hW[(size_t)n * K + k]);
            maxerr = std::max(maxerr, std::fabs(r - o) / (1.0 + std::fabs(r)));
        }
    for (size_t i = 0; i < cs.size(); ++i) maxc = std::max(maxc, std::fabs(cs[i] - hC[i]) / (1.0 + std::fabs(cs[i])));
    const bool ok = bad_mask == 0 && maxerr < 2e-3 && maxc < 1e-4;
    printf("masked M=%d N=%d K=%d: max rel err %.2e, unmasked-where-masked %ld, column-sum rel err %.2e  %s\n", M, N, K, maxerr, bad_mask, maxc, ok ? "OK" : "FAILED");
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dM); hipFree(dC);
    return ok;
}

int main() {
#if !defined(ARP_G2_STAMPS) && !ARP_G2_ABL
    bool ok = check_masked(512, 256, 128) & check_masked(1000, 520, 192) & check_masked(32896, 768, 768);
    if (!ok) return 1;
#endif
    run<f16_t, ACT_NONE, false>("qkv", 51200, 2304, 768);
    run<f16_t, ACT_QGELU, false>("c_fc", 51200, 3072, 768);
    run<f16_t, ACT_NONE, false>("c_fc_noact", 51200, 3072, 768);
    run<float, ACT_NONE, true>("c_proj", 51200, 768, 3072);
    run<float, ACT_NONE, true>("out_proj", 51200, 768, 768);
    run<f16_t, ACT_QGELU, false>("c_fc_half", 25600, 3072, 768);
    run<float, ACT_NONE, true>("c_proj_half", 25600, 768, 3072);
    run<f16_t, ACT_NONE, false>("sq4096", 4096, 4096, 4096);
    run<f16_t, ACT_NONE, false>("ragged", 1000, 520, 192);
    if (getenv("G256_MALL")) {  // is the c_proj A stream's home (Infinity Cache vs HBM) visible?  one / two full rounds of 255 tiles
        run<float, ACT_NONE, true>("cproj_1rnd", 21760, 768, 3072);   // A = 134 MB: stays in the 256 MiB Infinity Cache between launches
        run<float, ACT_NONE, true>("cproj_2rnd", 43520, 768, 3072);   // A = 267 MB: does not
        run<f16_t, ACT_QGELU, false>("cfc_1rnd", 5376, 3072, 768);     // 21 x 12 = 252 tiles
        run<f16_t, ACT_QGELU, false>("cfc_8rnd", 43520, 3072, 768);    // 170 x 12 = 2040 tiles = 8 rounds
    }
    return 0;
}

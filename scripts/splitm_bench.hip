// Tile rounds of c_proj (600 tiles of 256 x 256 on 256 CUs = 2.34 rounds, run as 3): what does it buy to run the FULL rounds on gemm256 and the rest of the rows
// on the 128 x 128 kernel (two workgroups per CU: 4x as many, 1/4-size tiles fill the chip where 88 big tiles leave two thirds of it idle)?  Same MFMA, same k order:
// the two kernels must agree bit for bit, so the split changes no result (checked here by checksum).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iarp_amd/csrc scripts/splitm_bench.hip -o scripts/splitm_bench.bin
#include <algorithm>
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
    uint32_t s = 4321u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    void *dA, *dW, *dR;
    float* dB;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dB, N * 4); hipMalloc(&dR, (size_t)M * N * 4);  // (f16 outputs use the first half)
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice);
    auto args = [&](int row0, int rows) {
        GemmArgs g;
        g.A = (const f16_t*)dA + (size_t)row0 * K; g.W = dW; g.bias = dB; g.resid = RESID ? (float*)dR + (size_t)row0 * N : nullptr; g.out = (OutT*)dR + (size_t)row0 * N;
        g.M = rows; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
        return g;
    };
    const int n_tiles = (N + 255) / 256, m_tiles = (M + 255) / 256;
    const int full_m = (m_tiles * n_tiles / 256) * 256 / n_tiles;  // tile rows of the full rounds
    const int Mfull = std::min(M, full_m * 256);
    auto whole = [&]() { launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6>(args(0, M), nullptr); };
    auto split = [&]() {
        if (Mfull > 0) launch_gemm256_nt<f16_t, OutT, ACT, RESID, 6>(args(0, Mfull), nullptr);
        if (M > Mfull) launch_gemm_nt<f16_t, OutT, ACT, RESID, 6>(args(Mfull, M - Mfull), nullptr);
    };
    uint64_t sums[2];
    std::vector<float> ho((size_t)M * N);  // bytes compared: the whole buffer (an f16 run leaves its second half at the memset value)
    for (int v = 0; v < 2; ++v) {
        hipMemset(dR, 0, (size_t)M * N * 4);
        if (v) split(); else whole();
        hipDeviceSynchronize();
        hipMemcpy(ho.data(), dR, ho.size() * 4, hipMemcpyDeviceToHost);
        sums[v] = fnv(ho.data(), ho.size() * 4);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) { whole(); split(); }
    std::vector<float> t[2];
    for (int r = 0; r < 7; ++r)
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) { if (v) split(); else whole(); }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            t[v].push_back(ms / 10);
        }
    std::sort(t[0].begin(), t[0].end()); std::sort(t[1].begin(), t[1].end());
    printf("%-12s M=%d N=%d K=%d: %d tiles = %.2f rounds; whole %7.1f us (min %7.1f) | full rounds on gemm256 (%d rows) + rest on 128x128 %7.1f us (min %7.1f) | ratio %.3f | bit-identical: %s\n", name, M, N, K,
           m_tiles * n_tiles, m_tiles * n_tiles / 256.0, t[0][3] * 1e3, t[0][0] * 1e3, Mfull, t[1][3] * 1e3, t[1][0] * 1e3, t[1][3] / t[0][3], sums[0] == sums[1] ? "yes" : "NO");
    hipFree(dA); hipFree(dW); hipFree(dB); hipFree(dR);
}

int main() {
    run<float, ACT_NONE, true>("c_proj", 51200, 768, 3072);
    run<float, ACT_NONE, true>("c_proj_half", 25600, 768, 3072);
    run<float, ACT_NONE, true>("out_proj", 51200, 768, 768);
    run<f16_t, ACT_QGELU, false>("c_fc", 51200, 3072, 768);
    run<f16_t, ACT_QGELU, false>("c_fc_half", 25600, 3072, 768);
    run<f16_t, ACT_NONE, false>("qkv_b16", 50432, 2304, 768);
    run<f16_t, ACT_NONE, false>("qkv_b16h", 25216, 2304, 768);
    run<f16_t, ACT_GELU_TANH, false>("m3ae_c_fc", 32896, 3072, 768);
    run<float, ACT_NONE, true>("m3ae_c_proj", 32896, 768, 3072);
    run<f16_t, ACT_NONE, false>("m3ae_qkv", 32896, 2304, 768);
    return 0;
}

// Standalone timing harness for the fused QKV projection + attention kernel (qkvattn.hip) at the labelling shape (1024 / 512 frames,
// N = 50, 12 heads, K = 768).  -DARP_QA_STAMPS: in-kernel s_memtime stamps -> cycles per phase (K loop | q/k/v images | attention |
// barrier + copy-out).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iarp_amd/csrc [-DARP_QA_STAMPS] scripts/qkvattn_bench.hip -o scripts/qkvattn_bench.bin
#include <cstdio>
#include <vector>

#include "qkvattn.hip"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
int launch_gemm2w_dyn(int, int, int, int, const GemmArgs&, hipStream_t) { return fail("gemm2w is not linked into this harness"); }
bool gemm2w_has(int, int, int, int) { return false; }
}  // namespace arp
using namespace arp;

static void run(int B) {
    const int N = 50, heads = 12, K = 768, D = 768;
    const size_t M = (size_t)B * N;
    std::vector<f16_t> hA(M * K), hW((size_t)heads * 192 * K);
    std::vector<float> hb(heads * 192);
    uint32_t s = 99u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hb) v = rnd() * 0.1f;
    void *dA, *dW, *dO;
    float* dB;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dB, hb.size() * 4); hipMalloc(&dO, M * D * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    QkvAttnArgs g{};
    g.A = dA; g.W = dW; g.bias = dB; g.out = dO; g.B = B; g.N = N; g.K = K; g.heads = heads; g.lda = K; g.ldw = K; g.ldo = D; g.nq = N; g.causal = 0;
    const int wgs = ((B + 4) / 5) * heads;
#ifdef ARP_QA_STAMPS
    long long* dS;
    hipMalloc(&dS, (size_t)wgs * 64 * 8);
    hipMemset(dS, 0, (size_t)wgs * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(arp_qa_stamps), &dS, sizeof(dS));
#endif
    if (launch_qkv_attn_f16(g, nullptr)) exit(1);
    hipDeviceSynchronize();
#ifdef ARP_QA_STAMPS
    {
        std::vector<long long> h((size_t)wgs * 64);
        hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost);
        double d[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
        for (int t = 0; t < wgs; ++t)
            for (int w = 0; w < 8; ++w)
                for (int i = 0; i < 4; ++i) {
                    const double v = (double)(h[((size_t)t * 8 + w) * 8 + i + 1] - h[((size_t)t * 8 + w) * 8 + i]);
                    d[i] += v;
                    if (w == 0) mx[i] += v;
                }
        printf("  stamps B=%d (%d workgroups), cycles per workgroup averaged over waves: K loop %.0f | q/k/v images %.0f | attention %.0f | barrier + copy-out %.0f"
               "   (wave 0: attention %.0f, barrier + copy-out %.0f)\n",
               B, wgs, d[0] / wgs / 8, d[1] / wgs / 8, d[2] / wgs / 8, d[3] / wgs / 8, mx[2] / wgs, mx[3] / wgs);
    }
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_qkv_attn_f16(g, nullptr);
    const int iters = 20;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch_qkv_attn_f16(g, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double flop = 2.0 * M * 2304 * K + (double)B * heads * 4.0 * N * N * 64;
    printf("qkv_attn B=%d: %8.1f us  %7.1f TFLOP/s (projection + attention FLOPs)\n", B, ms * 1e3, flop / ms / 1e9);
    hipFree(dA); hipFree(dW); hipFree(dB); hipFree(dO);
}

int main() {
    run(1024);
    run(512);
    return 0;
}

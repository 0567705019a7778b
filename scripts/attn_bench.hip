// Standalone timing harness for attn_mfma_kernel (attention.h) at the long-sequence shapes: ViT-B/16 labelling (N = 197) and the
// M3AE encoder inside the policy step (N = 257).  -DARP_ATTN_STAMPS: cycles per phase (K/V staging | QK^T | softmax | PV | stores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iarp_amd/csrc [-DARP_ATTN_STAMPS] scripts/attn_bench.hip -o scripts/attn_bench.bin
#include <cstdio>
#include <vector>

#include "tower.h"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
int launch_gemm2w_dyn(int, int, int, int, const GemmArgs&, hipStream_t) { return fail("not linked"); }
bool gemm2w_has(int, int, int, int) { return false; }
int launch_qkv_attn_f16(QkvAttnArgs, hipStream_t) { return fail("not linked"); }
int launch_qkv_attn_bf16(QkvAttnArgs, hipStream_t) { return fail("not linked"); }
}  // namespace arp
using namespace arp;

static void run(int B, int N) {
    const int D = 768, heads = 12;
    const size_t rows = (size_t)B * N;
    std::vector<f16_t> h(rows * 3 * D);
    uint32_t s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : h) v = host_f2h(rnd());
    void *dq, *dout;
    hipMalloc(&dq, h.size() * 2); hipMalloc(&dout, rows * D * 2);
    hipMemcpy(dq, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int wgs = B * heads;
#ifdef ARP_ATTN_STAMPS
    long long* dS;
    hipMalloc(&dS, (size_t)wgs * 32 * 8);
    hipMemset(dS, 0, (size_t)wgs * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(arp_attn_stamps), &dS, sizeof(dS));
#endif
    auto go = [&]() { return launch_attention<f16_t>(nullptr, 0, static_cast<const f16_t*>(dq), static_cast<f16_t*>(dout), B, N, D, heads, 0); };
    if (go()) exit(1);
    hipDeviceSynchronize();
#ifdef ARP_ATTN_STAMPS
    {
        std::vector<long long> st((size_t)wgs * 32);
        hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost);
        double d[5] = {0, 0, 0, 0, 0};
        for (int t = 0; t < wgs; ++t)
            for (int w = 0; w < 4; ++w)
                for (int i = 0; i < 5; ++i) d[i] += (double)st[((size_t)t * 4 + w) * 8 + i];
        printf("  stamps N=%d: cycles per workgroup (wave average): K/V staging %.0f | QK^T %.0f | softmax %.0f | PV %.0f | stores %.0f\n", N, d[0] / wgs / 4,
               d[1] / wgs / 4, d[2] / wgs / 4, d[3] / wgs / 4, d[4] / wgs / 4);
    }
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("attention B=%d N=%d: %7.1f us  (%.2f TB/s of qkv read + output written)\n", B, N, ms * 1e3, (double)rows * D * 8 / ms / 1e9);
    hipFree(dq); hipFree(dout);
}

int main() {
    run(128, 257);  // the M3AE encoder inside the policy step (B = 32 x window 4)
    run(128, 197);  // ViT-B/16 labelling, one 128-frame part
    run(256, 197);
    return 0;
}

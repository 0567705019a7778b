// Where does a workgroup of attn_x3_kernel spend its time?  (round 4; the kernel was insensitive to its LDS conflicts, its tail round and its staging order)
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. -DARP_ATTNX3_STAMPS ../../scripts/attn_x3_stamps.hip -o ../../scripts/attn_x3_stamps.bin
// s_memtime (100 MHz constant clock: 10 ns units) at the phase boundaries of each wave's first query block, averaged over the workgroups; the M3AE geometry
// (128 sequences x 12 heads x 257 tokens, f32 qkv) by default.
#include <cstdio>
#include <vector>

#include "attention.h"
namespace arp {
int fail(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string&) {}
}  // namespace arp
using namespace arp;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, N = 257, D = 768, heads = 12;
    const size_t nq = (size_t)B * N * 3 * D;
    std::vector<float> h(nq);
    uint32_t s = 7u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * 1.5f; }
    float *dq, *dout;
    hipMalloc(&dq, nq * 4); hipMalloc(&dout, (size_t)B * N * D * 4);
    hipMemcpy(dq, h.data(), nq * 4, hipMemcpyHostToDevice);
    const int WG = B * heads;
    long long* dS;
    hipMalloc(&dS, (size_t)WG * 8 * 16 * 8);
    hipMemset(dS, 0, (size_t)WG * 8 * 16 * 8);
    auto kern = attn_x3_kernel<17>;
    const int lds = attn_x3_lds_bytes(17);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const float scale = 0.125f;
    long long* null_stamps = nullptr;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: no stamps (timing), pass 1: stamps
        hipMemcpyToSymbol(HIP_SYMBOL(arp_ax3_stamps), pass ? &dS : &null_stamps, sizeof(dS));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(WG), dim3(512), lds, nullptr, dq, dout, N, D, heads, scale, 0, N, (f16_t*)nullptr);
        hipEventRecord(e0);
        const int reps = pass ? 1 : 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(WG), dim3(512), lds, nullptr, dq, dout, N, D, heads, scale, 0, N, (f16_t*)nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch (%d workgroups, %d B of LDS, %.1f per CU)\n", pass ? "with stamps" : "plain", ms * 1e3 / reps, WG, lds, WG / 256.0);
    }
    std::vector<long long> st((size_t)WG * 8 * 16);
    hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost);
    const char* nm[9] = {"staging (loads, split, LDS stores, barrier)", "barrier -> first block", "Q load + split", "S^T = K.Q^T (102 MFMAs)", "softmax", "O^T = V^T.P^T (108 MFMAs + P split)",
                         "output stores issued", "remaining blocks of the wave", "tail block + merge"};
    double d[9] = {0};
    long n = 0;
    double wg_total = 0;
    for (int w = 0; w < WG; ++w) {
        long long t0 = st[(size_t)w * 128], tend = 0;
        for (int wave = 0; wave < 8; ++wave) {
            const long long* p = &st[((size_t)w * 8 + wave) * 16];
            if (!p[9]) continue;
            for (int k = 0; k < 9; ++k) d[k] += (double)(p[k + 1] - p[k]);
            ++n;
            tend = std::max(tend, p[9]);
        }
        wg_total += (double)(tend - t0);
    }
    printf("per wave, first query block (averages over %ld waves), in us:\n", n);
    for (int k = 0; k < 9; ++k) printf("  %-48s %7.2f\n", nm[k], d[k] / n * 0.01);
    printf("workgroup start -> last wave done: %.2f us on average\n", wg_total / WG * 0.01);
    return 0;
}

// A/B harness: the W-tiled skinny GEMM (skinny.hip) against the 128 x 128 output-tiled kernel (gemm.h) on the single-frame
// tower's shapes; both against a double-precision host product of the same f16 operands.
//   cd arp_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. ../../scripts/skinny_bench.hip -o ../../scripts/skinny_bench.bin
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

#include "gemm.h"
#include "skinny.hip"

namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string& m) { g_err = m; }
}  // namespace arp
using namespace arp;

static float h2f_host(f16_t v) { return (float)__builtin_bit_cast(_Float16, v); }

// act: 0 none, 1 quickgelu;  resid: f32 residual epilogue in place;  out16: f16 output
static void run(const char* name, int M, int N, int K, int act, int resid, int out16, int ksplit) {
    std::vector<f16_t> hA((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N), hr((size_t)M * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    for (auto& v : hr) v = rnd();
    void *dA, *dW, *dO, *dR, *dR0, *dP, *dH;
    float *dB, *dLw, *dLb;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dB, N * 4); hipMalloc(&dO, (size_t)M * N * 4); hipMalloc(&dR, (size_t)M * N * 4);
    hipMalloc(&dR0, (size_t)M * N * 4); hipMalloc(&dP, (size_t)std::max(ksplit, 1) * M * N * 4); hipMalloc(&dH, (size_t)M * N * 2);
    hipMalloc(&dLw, N * 4); hipMalloc(&dLb, N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dLw, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dLb, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dR0, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);

    GemmArgs g;
    g.A = dA; g.W = dW; g.bias = dB; g.resid = resid ? (float*)dR : nullptr; g.out = resid ? dR : dO;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
    SkinnyArgs k;
    k.A = dA; k.W = dW; k.bias = dB; k.resid = resid ? (float*)dR : nullptr; k.out = resid ? dR : dO;
    k.M = M; k.N = N; k.K = K; k.lda = K; k.ldw = K; k.ldr = N; k.ldo = N; k.act = act; k.out_f32 = !out16;
    if (getenv("SK_STRIPS")) k.strips = 1;
    if (getenv("SK_GATHER")) k.gather = 1;
    if (ksplit > 1) { k.ksplit = ksplit; k.slice_stride = (size_t)M * N; k.out = dP; k.out_f32 = 1; k.bias = nullptr; k.resid = nullptr; k.act = 0; }
    auto go_ref = [&]() -> int {
        if (resid) return launch_gemm_nt<f16_t, float, ACT_NONE, true, 0>(g, nullptr);
        if (!out16) return launch_gemm_nt<f16_t, float, ACT_NONE, false, 0>(g, nullptr);
        if (act) return launch_gemm_nt<f16_t, f16_t, ACT_QGELU, false, 0>(g, nullptr);
        return launch_gemm_nt<f16_t, f16_t, ACT_NONE, false, 0>(g, nullptr);
    };
    auto go_sk = [&]() -> int {
        if (launch_skinny_gemm(2, k, nullptr)) return -1;
        if (ksplit > 1) return launch_skinny_reduce_ln(2, (const float*)dP, ksplit, (size_t)M * N, dB, (float*)dR, N, dH, N, dLw, dLb, M, N, 1e-5f, nullptr);
        return 0;
    };
    const size_t osz = (size_t)M * N * (out16 ? 2 : 4);
    std::vector<uint8_t> o[2];
    for (int v = 0; v < 2; ++v) {
        hipMemcpy(dR, dR0, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
        hipMemset(dO, 0, (size_t)M * N * 4);
        if ((v ? go_sk() : go_ref()) != 0) { printf("%s: launch failed\n", name); return; }
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel %d failed: %s\n", name, v, hipGetErrorString(hipGetLastError())); return; }
        o[v].resize(osz);
        hipMemcpy(o[v].data(), resid ? dR : dO, osz, hipMemcpyDeviceToHost);
    }
    // host reference (double) on a sample of rows
    double e_ref = 0, e_sk = 0, mag = 0;
    for (int m = 0; m < M; m += std::max(1, M / 7)) {
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int kk = 0; kk < K; ++kk) acc += (double)h2f_host(hA[(size_t)m * K + kk]) * (double)h2f_host(hW[(size_t)n * K + kk]);
            acc += hb[n];
            if (act) acc = acc / (1.0 + exp(-1.702 * acc));
            if (resid) acc += hr[(size_t)m * N + n];
            const size_t i = (size_t)m * N + n;
            const double a = out16 ? h2f_host(reinterpret_cast<f16_t*>(o[0].data())[i]) : reinterpret_cast<float*>(o[0].data())[i];
            const double b = out16 ? h2f_host(reinterpret_cast<f16_t*>(o[1].data())[i]) : reinterpret_cast<float*>(o[1].data())[i];
            e_ref = std::max(e_ref, fabs(a - acc)); e_sk = std::max(e_sk, fabs(b - acc)); mag = std::max(mag, fabs(acc));
        }
    }
    // device time per launch: 50 launches captured into one hipGraph, replayed (no host launch cost inside the timed region);
    // "host" = the same launches issued one by one
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st;
    hipStreamCreate(&st);
    float us[2], us_host[2];
    const int per = 50, reps = 8;
    for (int v = 0; v < 2; ++v) {
        auto go = [&](hipStream_t q) {
            if (v == 0) {
                if (resid) return launch_gemm_nt<f16_t, float, ACT_NONE, true, 0>(g, q);
                if (!out16) return launch_gemm_nt<f16_t, float, ACT_NONE, false, 0>(g, q);
                if (act) return launch_gemm_nt<f16_t, f16_t, ACT_QGELU, false, 0>(g, q);
                return launch_gemm_nt<f16_t, f16_t, ACT_NONE, false, 0>(g, q);
            }
            if (launch_skinny_gemm(2, k, q)) return -1;
            if (ksplit > 1) return launch_skinny_reduce_ln(2, (const float*)dP, ksplit, (size_t)M * N, dB, (float*)dR, N, dH, N, dLw, dLb, M, N, 1e-5f, q);
            return 0;
        };
        for (int i = 0; i < 10; ++i) go(st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < per * 4; ++i) go(st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        us_host[v] = ms * 1000.f / (per * 4);
        hipGraph_t gr;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < per; ++i) go(st);
        hipStreamEndCapture(st, &gr);
        hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        us[v] = ms * 1000.f / (per * reps);
        hipGraphExecDestroy(ge); hipGraphDestroy(gr);
    }
    hipStreamDestroy(st);
    printf("%-12s M=%3d N=%4d K=%4d split %d: 128x128 %5.1f us (host-issued %5.1f) | skinny%s %5.1f us (%5.1f) | max err vs f64: %.2e / %.2e (|out| <= %.1f)\n", name, M, N, K, ksplit, us[0], us_host[0],
           ksplit > 1 ? "+reduce_ln" : "", us[1], us_host[1], e_ref, e_sk, mag);
    hipFree(dA); hipFree(dW); hipFree(dB); hipFree(dO); hipFree(dR); hipFree(dR0); hipFree(dP); hipFree(dH); hipFree(dLw); hipFree(dLb);
}

// tile-shape sweep of the coalesced kernel at 65..256 rows (direct epilogue, f16 output): (MT, NT, waves)
template <int MT, int NT> static void variant(const char* name, int M, int N, int K, int nw) {
    std::vector<f16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = host_f2h(rnd());
    for (auto& v : hW) v = host_f2h(rnd() * 0.05f);
    void *dA, *dW, *dO, *dO2;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dO, (size_t)M * N * 2); hipMalloc(&dO2, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    SkinnyArgs k;
    k.A = dA; k.W = dW; k.out = dO; k.M = M; k.N = N; k.K = K; k.lda = K; k.ldw = K; k.ldr = N; k.ldo = N; k.out_f32 = 0;
    if (K % (64 * nw) || N % (16 * NT) || (size_t)nw * 2 * (MT + NT) * 16 * 128 > 160 * 1024) { printf("  %-6s (%d,%d,%d): not applicable\n", name, MT, NT, nw); return; }
    hipStream_t st;
    hipStreamCreate(&st);
    launch_lds<f16_t, MT, NT>(k, nw, st);
    SkinnyArgs r = k; r.out = dO2; r.gather = 1;
    launch_skinny_gemm(2, r, st);
    hipStreamSynchronize(st);
    std::vector<f16_t> o1((size_t)M * N), o2((size_t)M * N);
    hipMemcpy(o1.data(), dO, o1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), dO2, o2.size() * 2, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < o1.size(); ++i) md = std::max(md, (double)fabsf(h2f_host(o1[i]) - h2f_host(o2[i])));
    hipGraph_t gr; hipGraphExec_t ge; hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 50; ++i) launch_lds<f16_t, MT, NT>(k, nw, st);
    hipStreamEndCapture(st, &gr);
    hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 8; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int mtd = (M + 15) / 16;
    printf("  %-6s M=%3d (MT %d, NT %d, %d waves): %3d workgroups, %3zu KB operands each: %5.2f us | max diff vs gather kernel %.1e\n", name, M, MT, NT, nw,
           (N / (16 * NT)) * ((mtd + MT - 1) / MT), (size_t)(MT + NT) * 16 * K * 2 / 1024, ms * 1000.f / 400.f, md);
    hipGraphExecDestroy(ge); hipGraphDestroy(gr); hipStreamDestroy(st);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dO2);
}

int main() {
    if (getenv("SK_VARIANTS")) {
        for (int M : {197, 100}) {
            for (auto [name, N] : {std::pair<const char*, int>{"qkv", 2304}, {"c_fc", 3072}, {"out", 768}}) {
                variant<4, 4>(name, M, N, 768, 4); variant<4, 4>(name, M, N, 768, 3); variant<4, 4>(name, M, N, 768, 2);
                variant<4, 3>(name, M, N, 768, 4); variant<4, 2>(name, M, N, 768, 6); variant<4, 2>(name, M, N, 768, 4);
                variant<4, 1>(name, M, N, 768, 6); variant<7, 2>(name, M, N, 768, 4); variant<7, 1>(name, M, N, 768, 4); variant<7, 1>(name, M, N, 768, 6);
                variant<13, 1>(name, M, N, 768, 3); variant<8, 2>(name, M, N, 768, 3);
            }
        }
        return 0;
    }
    for (int M : {50, 100, 197, 250}) {
        run("qkv", M, 2304, 768, 0, 0, 1, 1);
        run("out_proj", M, 768, 768, 0, 1, 0, 1);
        run("out_proj/s", M, 768, 768, 0, 1, 0, 2);
        run("c_fc", M, 3072, 768, 1, 0, 1, 1);
        run("c_proj", M, 768, 3072, 0, 1, 0, 1);
        run("c_proj/s4", M, 768, 3072, 0, 1, 0, 4);
        run("c_proj/s8", M, 768, 3072, 0, 1, 0, 8);
        run("proj", M, 512, 768, 0, 0, 0, 1);
    }
    run("patch32", 49, 768, 3072, 0, 0, 0, 1);
    run("patch32/s4", 49, 768, 3072, 0, 1, 0, 4);
    run("patch16", 196, 768, 768, 0, 0, 0, 1);
    return 0;
}

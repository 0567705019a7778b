// Pre-LN transformer tower shared by the CLIP image / text towers (path 1) and the frozen M3AE encoder
// (row N1): per-layer weights, the GEMM / LayerNorm / attention launch helpers and the block loop.
#pragma once
#include <string>
#include <vector>

#include "attention.h"
#include "common.h"
#include "gemm.h"
#include "gemm256.h"
#include "rowops.h"
#include "runtime.h"

namespace arp {

enum Site { SITE_PATCH = 0, SITE_QKV = 1, SITE_OUT = 2, SITE_FC1 = 3, SITE_FC2 = 4, SITE_PROJ = 5, SITE_OP = 6 };

struct LayerW {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *b_in, *b_out, *b_fc, *b_proj;
    void *w_in, *w_out, *w_fc, *w_proj;
};
struct TowerW {
    int width = 0, layers = 0, heads = 0;
    std::vector<LayerW> L;
};

// what a tower launch needs from its owner
struct TowerCtx {
    hipStream_t stream = nullptr;
    Profiler* prof = nullptr;
    int attn_impl = 0;   // 0 = MFMA attention where available, 1 = VALU kernel
    int gemm_force = 0;  // 0 auto, 1 = 128x128 kernel, 2 = 256x256 kernel
};

template <typename T, typename OutT, int ACT, bool RESID, int SITE>
static int tower_gemm(TowerCtx& c, const char* site, const void* A, const void* W, const float* bias, const float* resid, void* out,
                int M, int N, int K) {
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.resid = resid; g.out = out;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
    ProfScope ps(*c.prof, c.stream, site);
    return launch_gemm_auto<T, OutT, ACT, RESID, SITE>(g, c.stream, c.gemm_force);
}

template <typename OutT>
static int tower_layernorm(TowerCtx& c, const char* site, const float* in, size_t in_stride, OutT* out, int out_stride,
                     const float* w, const float* b, int rows, int D, float eps) {
    if (D % 4 || D > ROW_MAX_V4 * 256) return fail("layernorm: unsupported width " + std::to_string(D));
    ProfScope ps(*c.prof, c.stream, site);
#define ARP_LN_CALL(NV)                                                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<OutT, NV>), dim3((rows + 3) / 4), dim3(256), 0, c.stream, in, in_stride, out, \
                       out_stride, w, b, rows, D, eps)
    ARP_NV_DISPATCH(D, ARP_LN_CALL);
#undef ARP_LN_CALL
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename T>
static int launch_attention(hipStream_t stream, int impl, const T* qkv, T* out, int B, int N, int D, int heads, int causal) {
    const int hd = D / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    if constexpr (sizeof(T) == 2) {
        if (impl == 0 && hd == 64) {
            const int NT = ((N + 31) / 32) * 2;
#define ARP_ATTN_CASE(nt)                                                                                                   \
    case nt: {                                                                                                              \
        auto kern = attn_mfma_kernel<nt>;                                                                                   \
        const int lds = nt * 16 * 128 + 64 * (nt * 32 + 8);                                                                 \
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL(kern, dim3(B* heads), dim3(256), lds, stream, qkv, out, N, D, heads, scale, causal);              \
        ARP_HIP_OK(hipGetLastError());                                                                                      \
        return 0;                                                                                                           \
    }
            switch (NT) {
                ARP_ATTN_CASE(2)
                ARP_ATTN_CASE(4)
                ARP_ATTN_CASE(6)
                ARP_ATTN_CASE(8)
                ARP_ATTN_CASE(14)
                ARP_ATTN_CASE(18)
                default: break;  // fall through to the VALU kernel
            }
#undef ARP_ATTN_CASE
        }
    }
    const size_t lds = (size_t)2 * N * hd * 4;
    if (lds > 160 * 1024) return fail("attention: sequence too long for the LDS-resident kernel");
    const int threads = N <= 64 ? 64 : (N <= 128 ? 128 : 256);
    if (hd == 64) {
        auto kern = attn_valu_kernel<T, 64>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal);
    } else if (hd == 32) {
        auto kern = attn_valu_kernel<T, 32>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal);
    } else if (hd == 16) {
        auto kern = attn_valu_kernel<T, 16>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal);
    } else {
        return fail("attention: unsupported head_dim " + std::to_string(hd));
    }
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// 12 x ResidualAttentionBlock (arp_dt/models/openai/layers.py:235-271) on the f32 residual stream x.
// ACT: MLP activation (QuickGELU for CLIP, tanh-GELU for the M3AE encoder); eps: LayerNorm epsilon;
// SB: site-id base so that every call site is its own kernel instantiation in a rocprof trace.
template <typename T, int ACT, int SB>
static int run_blocks(TowerCtx& c, const TowerW& tw, const char* tag, float* x, T* h, T* qkv, T* ao, T* fc, int B, int N, int causal,
                      float eps) {
    const int D = tw.width, M = B * N;
    const std::string t(tag);
    const std::string s_ln1 = t + ".ln_1", s_qkv = t + ".qkv", s_attn = t + ".attn", s_out = t + ".out_proj", s_ln2 = t + ".ln_2",
                      s_fc1 = t + ".c_fc", s_fc2 = t + ".c_proj";
    for (int i = 0; i < tw.layers; ++i) {
        const LayerW& L = tw.L[i];
        ARP_TRY(tower_layernorm<T>(c, s_ln1.c_str(), x, D, h, D, L.ln1_w, L.ln1_b, M, D, eps));
        ARP_TRY((tower_gemm<T, T, ACT_NONE, false, SB + SITE_QKV>(c, s_qkv.c_str(), h, L.w_in, L.b_in, nullptr, qkv, M, 3 * D, D)));
        {
            ProfScope ps(*c.prof, c.stream, s_attn.c_str());
            ARP_TRY(launch_attention<T>(c.stream, c.attn_impl, qkv, ao, B, N, D, tw.heads, causal));
        }
        ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_OUT>(c, s_out.c_str(), ao, L.w_out, L.b_out, x, x, M, D, D)));
        ARP_TRY(tower_layernorm<T>(c, s_ln2.c_str(), x, D, h, D, L.ln2_w, L.ln2_b, M, D, eps));
        ARP_TRY((tower_gemm<T, T, ACT, false, SB + SITE_FC1>(c, s_fc1.c_str(), h, L.w_fc, L.b_fc, nullptr, fc, M, 4 * D, D)));
        ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_FC2>(c, s_fc2.c_str(), fc, L.w_proj, L.b_proj, x, x, M, D, 4 * D)));
    }
    return 0;
}

}  // namespace arp

// Pre-LN transformer tower shared by the CLIP image / text towers (path 1) and the frozen M3AE encoder
// (row N1): per-layer weights, the GEMM / LayerNorm / attention launch helpers and the block loop.
#pragma once
#include <string>
#include <vector>

#include "attention.h"
#include "common.h"
#include "gemm.h"
#include "gemm256.h"
#include "qkvattn.h"
#include "rowops.h"
#include "skinny.h"
#include "runtime.h"

namespace arp {

enum Site { SITE_PATCH = 0, SITE_QKV = 1, SITE_OUT = 2, SITE_FC1 = 3, SITE_FC2 = 4, SITE_PROJ = 5, SITE_OP = 6 };

struct LayerW {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *b_in, *b_out, *b_fc, *b_proj;
    void *w_in, *w_out, *w_fc, *w_proj;
    // LayerNorm-folded operands (bf16 mode): W' = W diag(gamma), c = row sums of W', d = W beta + bias
    // fp8 MLP (BASELINE configs[4] "fp8 MFMA GEMMs"; TowerCtx::fp8_mlp): e4m3 copies of c_fc / c_proj scaled by per-tensor powers
    // of two, ln_2's gain and bias pre-multiplied by the activation scale, and the factors that take the scales out again
    void *w_fc8 = nullptr, *w_proj8 = nullptr;
    float *ln2_w8 = nullptr, *ln2_b8 = nullptr;
    float a_fc = 1.f, a_proj = 1.f;
    // fp8 projections of the attention (TowerCtx::fp8_attn): e4m3 in_proj / out_proj, ln_1 pre-multiplied by the activation scale
    void *w_in8 = nullptr, *w_out8 = nullptr;
    float *ln1_w8 = nullptr, *ln1_b8 = nullptr;
    float a_in = 1.f, a_out = 1.f;
    // head-major in-proj operands of the fused QKV + attention kernel (qkvattn.h): rows h*192 + {q | k | v of head h}
    void* w_in_hm = nullptr;
    float* b_in_hm = nullptr;
    void *w_in_f = nullptr, *w_fc_f = nullptr;
    float *c_in = nullptr, *d_in = nullptr, *c_fc = nullptr, *d_fc = nullptr;
};
struct TowerW {
    int width = 0, layers = 0, heads = 0;
    bool folded = false;  // every layer carries the folded operands
    bool lat_folded = false;  // ... in the handle's own operand type, for the LATENCY path only (run_blocks; the throughput path is unchanged)
    std::vector<LayerW> L;
};

// Host-side LayerNorm folding for a Linear that consumes LN(x):
//   LN(x) W^T + b = rstd * (x W'^T - mu * c) + d,   W' = W diag(gamma),  c[n] = sum_k W'[n,k],  d = W beta + b.
// W' is rounded to bf16 first and c is summed from the ROUNDED values, so the mean subtraction cancels exactly
// what the MFMA accumulates.
// (half = true: IEEE-half operands -- the latency path's fold in f16 mode; false: bf16)
inline void fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K,
                           std::vector<bf16_t>& Wf, std::vector<float>& c, std::vector<float>& d, bool half = false) {
    Wf.resize((size_t)N * K);
    c.assign(N, 0.f);
    d.assign(N, 0.f);
    for (int n = 0; n < N; ++n) {
        double cs = 0.0, ds = bias ? (double)bias[n] : 0.0;
        for (int k = 0; k < K; ++k) {
            const float w = W[(size_t)n * K + k];
            float wf;
            if (half) {
                const f16_t wh = host_f2h(w * gamma[k]);
                Wf[(size_t)n * K + k] = __builtin_bit_cast(bf16_t, wh);  // raw 16 bits either way
                wf = (float)__builtin_bit_cast(_Float16, wh);
            } else {
                const bf16_t wb = host_f2bf(w * gamma[k]);
                Wf[(size_t)n * K + k] = wb;
                uint32_t u = (uint32_t)wb << 16;
                memcpy(&wf, &u, 4);
            }
            cs += (double)wf;
            ds += (double)w * (double)beta[k];
        }
        c[n] = (float)cs;
        d[n] = (float)ds;
    }
}

// x f32 [rows, D] -> operand-type copy xb and the per-128-column partial (sum, sumsq) the folded GEMMs consume.
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, T* __restrict__ xb, float* __restrict__ stats,
                                                        int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int parts = D >> 7;
    for (int c0 = 0; c0 < D; c0 += 256) {  // one 256-column chunk per iteration: two 128-column segments
        const int c = c0 + lane * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            load4(x + (size_t)row * D + c, v);
            store4(xb + (size_t)row * D + c, v[0], v[1], v[2], v[3]);
        }
        float s = (v[0] + v[1]) + (v[2] + v[3]);
        float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        const int seg = (c0 >> 7) + (lane >> 5);
        if ((lane & 31) == 0 && seg < parts) {
            stats[((size_t)row * parts + seg) * 2] = s;
            stats[((size_t)row * parts + seg) * 2 + 1] = q;
        }
    }
}

// optional LayerNorm-fold plumbing of one GEMM launch
struct GemmFold {
    const float* stats = nullptr;  // consumer
    const float* c = nullptr;
    int parts = 0;
    float inv_d = 0.f, eps = 0.f;
    void* xb_out = nullptr;        // producer
    int ldxb = 0;
    float* stats_out = nullptr;
    int split3 = 0;                // xb_out rows are (hi, lo, hi) binary16 triples of the f32 result (GemmArgs::split3)
};

// what a tower launch needs from its owner
struct TowerCtx {
    hipStream_t stream = nullptr;
    Profiler* prof = nullptr;
    // Only row 0 of every sample (the class token) is read after the last block (ln_post(x[:,0]), arp_dt/models/openai/
    // layers.py:330): the last block then runs out_proj / ln_2 / c_fc / c_proj -- and the attention's query side -- on those
    // B rows instead of all B*N.  Every value that IS consumed is the same dot product as before; 49/50 of the last block's
    // out_proj + MLP work (6.1 % of the tower's FLOPs at N = 50) is never computed.
    bool cls_only_last = false;
    int attn_impl = 0;   // 0 = MFMA attention where available, 1 = VALU kernel
    int gemm_force = 0;  // 0 auto, 1 = 128x128 kernel, 2 = 256x256 kernel
    // c_fc / c_proj on the scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4: twice the 16-bit rate): ln_2 writes e4m3 (x 32), c_fc
    // reads it and writes QuickGELU x 16 as e4m3, c_proj reads that; accumulation, bias, residual stream stay f32.  Three
    // significand bits: a lower-precision THROUGHPUT mode for the frozen towers of the fine-tune step, not the labelling default.
    bool fp8_mlp = false;
    // ... and the attention's two projections as well (blocks whose QKV + attention do not run on the fused kernel, i.e. ViT-B/16's
    // 197 tokens): ln_1 writes e4m3 (x 32), in_proj is an fp8 GEMM with a 16-bit output (the attention kernel's operand type), the
    // attention writes its output as e4m3 (x 16), out_proj is an fp8 GEMM into the f32 residual stream.
    bool fp8_attn = false;
    bool qkv_fused = true;  // QKV projection + attention in one kernel where qkvattn.h supports the geometry (ARP_QKV_FUSED=0 disables)
    unsigned long long* clock_acc = nullptr;  // != null: c_fc runs on gemm256's clock-diagnostic instance and accumulates its workgroups' cycle / real-time stamps here
    bool shared_chip = false;  // the pass runs beside another part stream's kernels (arp_clip.hip::label_dev): see out_proj's kernel choice in tower_gemm
    // Latency path (the rollout loop's single-frame reward): with at most SKINNY_MAX_M rows in the residual stream the
    // GEMMs run on the W-tiled skinny kernel (skinny.h), out_proj / c_proj as split-K slabs whose reduction kernel also adds bias +
    // residual and applies the NEXT LayerNorm -- six launches per block, each near the 3 us a dependent launch costs.
    bool skinny = false;       // set by the owner for a pass of at most SKINNY_MAX_M rows IN TOTAL (never per GEMM: a batch cut into
                               // parts must give the same bits whatever the part size, tests/test_clip_gpu.py two-stream test)
    bool h_ready0 = false;     // the caller already wrote ln_1 of the first block into h (the latency path's token-assembly kernel)
    float* lat_stats = nullptr; // folded LayerNorm on the latency path: per row and 16-column strip (sum, sum of squares), [rows][width / 16][2];
                               // h then holds the operand-type copy of x and the caller's token-assembly kernel has written both for block 0
    bool lat_fold0 = false;
    float* part = nullptr;     // split-K slabs, f32 [<= 4][rows][width]; null: out_proj / c_proj stay single launches
    size_t part_floats = 0;
    // multi-scale export: after every block, one row per sample of the residual stream -- what the forward
    // hooks of finetune_module/utils.py:6-18 capture on each resblock output -- is copied to ms_out[b, layer*D ..]
    float* ms_out = nullptr;
    int ms_ld = 0;
    const int* ms_rows = nullptr;  // absolute row per sample (text: the EOT token); null = row b*N (the CLS token)
};

// out[b, col0 + d] = x[row(b), d]
static __global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ x, int D, const int* __restrict__ rows, int row_stride,
                                                                 float* __restrict__ out, int ld, int col0, int B) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * D) return;
    const int b = (int)(i / D), d = (int)(i - (size_t)b * D);
    const size_t row = rows ? (size_t)rows[b] : (size_t)b * row_stride;
    out[(size_t)b * ld + col0 + d] = x[row * D + d];
}
static int tower_export_rows(TowerCtx& c, const float* x, int D, int layer, int B, int N) {
    if (!c.ms_out) return 0;
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)(((size_t)B * D + 255) / 256)), dim3(256), 0, c.stream, x, D, c.ms_rows, N, c.ms_out, c.ms_ld,
                       layer * D, B);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename T, typename OutT, int ACT, bool RESID, int SITE>
static int tower_gemm(TowerCtx& c, const char* site, const void* A, const void* W, const float* bias, const float* resid, void* out,
                int M, int N, int K, const GemmFold* f = nullptr, int lda = 0, int ldr = 0, int ldo = 0) {
    GemmArgs g;
    if (f) {
        g.ln_stats = f->stats; g.ln_c = f->c; g.ln_parts = f->parts; g.ln_inv_d = f->inv_d; g.ln_eps = f->eps;
        g.xb_out = f->xb_out; g.ldxb = f->ldxb; g.stats_out = f->stats_out; g.split3 = f->split3;
    }
    g.A = A; g.W = W; g.bias = bias; g.resid = resid; g.out = out;
    g.M = M; g.N = N; g.K = K; g.lda = lda ? lda : K; g.ldw = K; g.ldr = ldr ? ldr : N; g.ldo = ldo ? ldo : N;
    ProfScope ps(*c.prof, c.stream, site);
    int force = c.gemm_force;
    if constexpr (sizeof(T) == 2 && (sizeof(OutT) == 4 || sizeof(OutT) == 2)) {
        // a handful of rows: the W-tiled kernel (3.9 against 11.5 us at 50 rows, K = 768; 6.2 against 11.6 at 197 rows; K = 3072: 7 against
        // 33 us: scripts/skinny_bench.hip)
        if (c.skinny && force == 0 && !f && skinny_supported(M, N, K, g.lda, g.ldw) && !(g.ldo & 3) && !(RESID && (g.ldr & 3))) {
            SkinnyArgs k;
            k.A = A; k.W = W; k.bias = bias; k.resid = RESID ? resid : nullptr; k.out = out;
            k.M = M; k.N = N; k.K = K; k.lda = g.lda; k.ldw = g.ldw; k.ldr = g.ldr; k.ldo = g.ldo;
            k.act = ACT; k.out_f32 = sizeof(OutT) == 4;
            return launch_skinny_gemm(__is_same(T, bf16_t) ? 1 : 2, k, c.stream);
        }
    }
    // the clock probe (arp_clip_clock_probe): c_fc of the vision tower on the CLK instance of the 256 x 256 kernel -- same tile, same K loop, four scalar
    // stamps and three atomics per workgroup more -- so that the clock is read under the very pass bench.py times
    if constexpr (SITE == SITE_FC1 && sizeof(T) == 2 && sizeof(OutT) == 2 && ACT == ACT_QGELU && !RESID) {
        if (c.clock_acc && !f && !c.skinny && force != 1 && force != 3 && (long)((M + 255) / 256) * ((N + 255) / 256) >= 192) {
            g.clock_acc = c.clock_acc;
            return launch_gemm256_nt<T, OutT, ACT, RESID, SITE, (ARP_G2_MFMA32 != 0), ARP_G2_KV, false, true>(g, c.stream);
        }
    }
    // out_proj (K = N = width, f32 residual epilogue) is the one big GEMM where the two-workgroups-per-CU kernel wins alone on the chip: its tiles
    // are short (12 K-tiles) and epilogue-heavy, so a second resident workgroup pays (measured 57.7 vs 66.8 us at M = 25 600; row N1's single
    // stream: 8.84 vs 8.89 ms per step).  Beside a second part stream (the labelling pass) the other stream's kernels are that second workgroup, and since the
    // residual rows stopped being loaded one at a time (round 5) the 256-tile kernel is the better neighbour: 101.0 -> 101.6 k and 102.8 -> 103.9 k frames/s on
    // two boxes, ViT-B/16 24.45 -> 24.70 k (profiles/r5_out_g256_ab.txt).  Same MFMA, same k order.  ARP_OUT_G256=0/1 forces either.
    static const int out_g256_env = [] { const char* e = getenv("ARP_OUT_G256"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    const bool out_g256 = out_g256_env >= 0 ? out_g256_env == 1 : c.shared_chip;
    if (force == 0 && (SITE & 7) == SITE_OUT && M >= 4096) force = out_g256 ? 2 : 3;
    // (c_proj on 256 x 192 tiles -- 400 instead of 300 tiles per 512-frame part, fewer idle slots in the last round -- measured
    //  slower, 5.3 vs 4.5 ms per step: the narrower wave tile's K loop loses more over 48 K-tiles than the rounding wins; removed)
    return launch_gemm_auto<T, OutT, ACT, RESID, SITE>(g, c.stream, force);
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
static int launch_attention(hipStream_t stream, int impl, const T* qkv, T* out, int B, int N, int D, int heads, int causal, int nq = 0, float out8 = 0.f,
                            f16_t* out3 = nullptr, int outc = 0) {  // out3 (T = float, the f32-MFMA kernel only): (hi, lo, hi) binary16 rows instead of f32
                                                                    // outc (T = f16, the MFMA kernel only): [hi | x4 | dx4] rows (ARP_MODE_F16C)
    const int hd = D / heads;
    if (nq <= 0 || nq > N) nq = N;  // query rows produced per sample
    const float scale = 1.0f / sqrtf((float)hd);
    if constexpr (sizeof(T) == 2) {
        if (impl == 0 && hd == 64) {
            const int NT = ((N + 31) / 32) * 2;
#define ARP_ATTN_CASE(nt)                                                                                                   \
    case nt: {                                                                                                              \
        auto kern = attn_mfma_kernel<T, nt>;                                                                                  \
        const int lds = nt * 16 * 256;                                                                 \
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        const int nqb_ = (nq + 15) / 16;                                                                                    \
        const int qsplit_ = (B * heads < 128 && nqb_ > 4) ? std::min((nqb_ + 3) / 4, 4) : 1;                                   \
        hipLaunchKernelGGL(kern, dim3(B* heads, qsplit_), dim3(256), lds, stream, qkv, out, N, D, heads, scale, causal, nq, out8, outc);  \
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
    if (out8 != 0.f || outc) return fail("attention: the e4m3 / [hi | x4 | dx4] outputs exist on the MFMA kernel only");
    if constexpr (sizeof(T) == 4) {
        // impl 3: the (hi, lo) binary16 MFMA kernel (attention.h::attn_x3_kernel, the f16x3 encoder mode's attention): head_dim 64, up to 288 keys
        if (impl == 3 && hd == 64 && N <= 288 && (D & 3) == 0) {
            // key tiles under the kernel's permuted mapping: tile kt holds keys 32 (kt / 2) + 8 i' + 4 (kt % 2) + r, so the odd tile of the last pair is needed from key 32 c + 4 on
            const int need = 2 * ((N - 1) / 32) + (((N - 1) % 32) >= 4 ? 2 : 1);
#define ARP_ATTNX3_CASE(nt)                                                                                                        \
    if (need <= nt) {                                                                                                              \
        const int ldsx = attn_x3_lds_bytes(nt);                                                                                    \
        auto kern = attn_x3_kernel<nt>;                                                                                            \
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsx));    \
        hipLaunchKernelGGL(kern, dim3(B* heads), dim3(512), ldsx, stream, qkv, out, N, D, heads, scale, causal, nq, out3);          \
        ARP_HIP_OK(hipGetLastError());                                                                                             \
        return 0;                                                                                                                  \
    }
            ARP_ATTNX3_CASE(2)
            ARP_ATTNX3_CASE(4)
            ARP_ATTNX3_CASE(6)
            ARP_ATTNX3_CASE(14)
            ARP_ATTNX3_CASE(17)
            ARP_ATTNX3_CASE(18)
#undef ARP_ATTNX3_CASE
        }
        if (impl == 3) impl = 0;  // other shapes: the exact-f32 kernels below
        // exact-f32 attention on the f32-input MFMA (attention.h): head_dim 64, up to 288 keys in LDS
        if (impl == 0 && hd == 64 && N <= 288 && (D & 3) == 0) {
            const int need = (N + 15) / 16;
#define ARP_ATTN32_CASE(nt)                                                                                                        \
    if (need <= nt) {                                                                                                              \
        const int lds32 = 2 * nt * 16 * 68 * 4;                                                                                    \
        auto kern = attn_f32_mfma_kernel<nt>;                                                                                      \
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds32));   \
        hipLaunchKernelGGL(kern, dim3(B* heads), dim3(256), lds32, stream, qkv, out, N, D, heads, scale, causal, nq, out3);         \
        ARP_HIP_OK(hipGetLastError());                                                                                             \
        return 0;                                                                                                                  \
    }
            ARP_ATTN32_CASE(1)
            ARP_ATTN32_CASE(4)
            ARP_ATTN32_CASE(5)
            ARP_ATTN32_CASE(13)
            ARP_ATTN32_CASE(17)
            ARP_ATTN32_CASE(18)
#undef ARP_ATTN32_CASE
        }
    }
    if (out3) return fail("attention: the (hi, lo, hi) output exists on the f32-MFMA kernel only (head_dim 64, <= 288 tokens)");
    const size_t lds = (size_t)2 * N * hd * 4;
    if (lds > 160 * 1024) return fail("attention: sequence too long for the LDS-resident kernel");
    const int threads = N <= 64 ? 64 : (N <= 128 ? 128 : 256);
    if (hd == 64) {
        auto kern = attn_valu_kernel<T, 64>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal, nq, (const float*)nullptr);
    } else if (hd == 32) {
        auto kern = attn_valu_kernel<T, 32>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal, nq, (const float*)nullptr);
    } else if (hd == 16) {
        auto kern = attn_valu_kernel<T, 16>;
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B * heads), dim3(threads), lds, stream, qkv, out, N, D, heads, scale, causal, nq, (const float*)nullptr);
    } else {
        return fail("attention: unsupported head_dim " + std::to_string(hd));
    }
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// the MFMA attention kernel is instantiated for this sequence length (launch_attention's switch)
static inline bool attn_mfma_has(int N, int hd) {
    const int NT = ((N + 31) / 32) * 2;
    return hd == 64 && (NT == 2 || NT == 4 || NT == 6 || NT == 8 || NT == 14 || NT == 18);
}
constexpr float FP8_S_A = 16.f;  // activation scale of the attention output as an fp8 operand

// ln_1 -> in_proj -> attention -> out_proj (+ residual) with both projections on fp8 operands (TowerCtx::fp8_attn)
template <typename T, int SB>
static int tower_attn_fp8(TowerCtx& c, const TowerW& tw, const LayerW& L, const char* s_ln1, const char* s_qkv, const char* s_attn, const char* s_out, float* x,
                          T* h, T* qkv, T* ao, int B, int N, int causal, float eps) {
    const int D = tw.width, M = B * N;
    fp8_t* h8 = reinterpret_cast<fp8_t*>(h);
    ARP_TRY(tower_layernorm<fp8_t>(c, s_ln1, x, D, h8, D, L.ln1_w8, L.ln1_b8, M, D, eps));
    GemmArgs g;
    g.A = h8; g.W = L.w_in8; g.bias = L.b_in; g.out = qkv;
    g.M = M; g.N = 3 * D; g.K = D; g.lda = D; g.ldw = D; g.ldr = 3 * D; g.ldo = 3 * D;
    g.alpha = L.a_in;
    {
        ProfScope ps(*c.prof, c.stream, s_qkv);
        ARP_TRY((launch_gemm256_nt<fp8_t, T, ACT_NONE, false, SB + SITE_QKV>(g, c.stream)));
    }
    {
        ProfScope ps(*c.prof, c.stream, s_attn);
        ARP_TRY(launch_attention<T>(c.stream, 0, qkv, ao, B, N, D, tw.heads, causal, 0, FP8_S_A));
    }
    GemmArgs q;
    q.A = ao; q.W = L.w_out8; q.bias = L.b_out; q.resid = x; q.out = x;
    q.M = M; q.N = D; q.K = D; q.lda = D; q.ldw = D; q.ldr = D; q.ldo = D;
    q.alpha = L.a_out;
    ProfScope ps(*c.prof, c.stream, s_out);
    return launch_gemm256_nt<fp8_t, float, ACT_NONE, true, SB + SITE_OUT>(q, c.stream);
}

// ln_1 output -> attention output: the fused kernel where the geometry allows it, else projection + attention kernel
template <typename T, int SITE>
static int tower_qkv_attention(TowerCtx& c, const TowerW& tw, const LayerW& L, const char* s_qkv, const char* s_attn, const char* s_fused, const T* h,
                               T* qkv, T* ao, int B, int N, int causal, int nq) {
    const int D = tw.width, M = B * N;
    if constexpr (sizeof(T) == 2) {
        // (with <= SKINNY_MAX_M rows the fused kernel is 12-60 workgroups walking K one round trip at a time: 18 us per block at one frame)
        if (c.qkv_fused && L.w_in_hm && c.attn_impl == 0 && qkv_attn_supported(N, D, tw.heads, (int)sizeof(T)) && !c.skinny) {
            QkvAttnArgs q;
            q.A = h; q.W = L.w_in_hm; q.bias = L.b_in_hm; q.out = ao;
            q.B = B; q.N = N; q.K = D; q.heads = tw.heads; q.lda = D; q.ldw = D; q.ldo = D; q.fpt = 0; q.nq = nq; q.causal = causal; q.scale = 0.f;
            ProfScope ps(*c.prof, c.stream, s_fused);
            return launch_qkv_attn<T>(q, c.stream);
        }
    }
    ARP_TRY((tower_gemm<T, T, ACT_NONE, false, SITE>(c, s_qkv, h, L.w_in, L.b_in, nullptr, qkv, M, 3 * D, D)));
    ProfScope ps(*c.prof, c.stream, s_attn);
    return launch_attention<T>(c.stream, c.attn_impl, qkv, ao, B, N, D, tw.heads, causal, nq);
}

constexpr float FP8_S_H = 32.f, FP8_S_G = 16.f;  // activation scales of the fp8 MLP: LayerNorm output, QuickGELU output

// ln_2 -> c_fc -> c_proj on fp8 operands (256 x 256 kernel only: M >= 1 row block of the big GEMMs)
template <typename T, int ACT, int SB>
static int tower_mlp_fp8(TowerCtx& c, const LayerW& L, const char* s_ln2, const char* s_fc1, const char* s_fc2, float* x, T* h, T* fc, int M, int D, float eps) {
    fp8_t* h8 = reinterpret_cast<fp8_t*>(h);
    fp8_t* fc8 = reinterpret_cast<fp8_t*>(fc);
    ARP_TRY(tower_layernorm<fp8_t>(c, s_ln2, x, D, h8, D, L.ln2_w8, L.ln2_b8, M, D, eps));
    GemmArgs g;
    g.A = h8; g.W = L.w_fc8; g.bias = L.b_fc; g.resid = nullptr; g.out = fc8;
    g.M = M; g.N = 4 * D; g.K = D; g.lda = D; g.ldw = D; g.ldr = 4 * D; g.ldo = 4 * D;
    g.alpha = L.a_fc; g.out_scale = FP8_S_G;
    {
        ProfScope ps(*c.prof, c.stream, s_fc1);
        ARP_TRY((launch_gemm256_nt<fp8_t, fp8_t, ACT, false, SB + SITE_FC1>(g, c.stream)));
    }
    GemmArgs q;
    q.A = fc8; q.W = L.w_proj8; q.bias = L.b_proj; q.resid = x; q.out = x;
    q.M = M; q.N = D; q.K = 4 * D; q.lda = 4 * D; q.ldw = 4 * D; q.ldr = D; q.ldo = D;
    q.alpha = L.a_proj;
    ProfScope ps(*c.prof, c.stream, s_fc2);
    return launch_gemm256_nt<fp8_t, float, ACT_NONE, true, SB + SITE_FC2>(q, c.stream);
}

// Latency path: x += A.W^T + bias, then h = LayerNorm(x) with (ln_w, ln_b) when given -- the product as `S` split-K slabs of the
// skinny kernel, summed in slab order by the row kernel that also applies the residual and the LayerNorm.
template <typename T>
static int tower_gemm_split_ln(TowerCtx& c, const char* s_gemm, const char* s_red, const void* A, const void* W, const float* bias, float* x, int M, int D,
                               int K, int S, T* h, const float* ln_w, const float* ln_b, float eps) {
    const int tcode = __is_same(T, bf16_t) ? 1 : 2;
    const size_t slab = (size_t)M * D;
    if ((size_t)S * slab > c.part_floats) return fail("tower: split-K slab buffer too small");
    SkinnyArgs k;
    k.A = A; k.W = W; k.out = c.part; k.M = M; k.N = D; k.K = K; k.lda = K; k.ldw = K; k.ldo = D; k.out_f32 = 1;
    k.ksplit = S; k.slice_stride = slab;
    {
        ProfScope ps(*c.prof, c.stream, s_gemm);
        ARP_TRY(launch_skinny_gemm(tcode, k, c.stream));
    }
    ProfScope ps(*c.prof, c.stream, s_red);
    return launch_skinny_reduce_ln(tcode, c.part, S, slab, bias, x, D, h, D, ln_w, ln_b, M, D, eps, c.stream);
}
// slabs for a [M, D] x [D, K]^T product on the latency path (0: geometry not supported)
static inline int tower_split_of(int M, int D, int K) {
    for (int S = K >= 2048 ? 4 : (K >= 512 ? 2 : 1); S >= 1; S >>= 1)
        if (skinny_supported(M, D, K, K, K, S) && K / S >= 32) return S;
    return 0;
}

// 12 x ResidualAttentionBlock (arp_dt/models/openai/layers.py:235-271) on the f32 residual stream x.
// ACT: MLP activation (QuickGELU for CLIP, tanh-GELU for the M3AE encoder); eps: LayerNorm epsilon;
// SB: site-id base so that every call site is its own kernel instantiation in a rocprof trace.
template <typename T, int ACT, int SB>
static int run_blocks(TowerCtx& c, const TowerW& tw, const char* tag, float* x, T* h, T* qkv, T* ao, T* fc, int B, int N, int causal,
                      float eps, float* stats = nullptr) {
    const int D = tw.width, M = B * N;
    const std::string t(tag);
    const std::string s_ln1 = t + ".ln_1", s_qkv = t + ".qkv", s_attn = t + ".attn", s_out = t + ".out_proj", s_ln2 = t + ".ln_2",
                      s_fc1 = t + ".c_fc", s_fc2 = t + ".c_proj", s_st = t + ".ln_stats", s_qa = t + ".qkv_attn", s_qa1 = t + ".qkv_attn_cls";
    if (stats && tw.folded && sizeof(T) == 2 && (D & 127) == 0) {
        // LayerNorm folded into the consumer GEMMs: `h` holds the bf16 copy of the residual stream, `stats` its
        // per-128-column partial sums; both are re-emitted by the epilogue of every residual GEMM.  No LN kernel runs.
        GemmFold cons, prod;
        cons.stats = stats; cons.parts = D >> 7; cons.inv_d = 1.0f / (float)D; cons.eps = eps;
        prod.xb_out = h; prod.ldxb = D; prod.stats_out = stats;
        {
            ProfScope ps(*c.prof, c.stream, s_st.c_str());
            hipLaunchKernelGGL((row_stats_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, c.stream, x, h, stats, M, D);
            ARP_HIP_OK(hipGetLastError());
        }
        for (int i = 0; i < tw.layers; ++i) {
            const LayerW& L = tw.L[i];
            cons.c = L.c_in;
            ARP_TRY((tower_gemm<T, T, ACT_NONE, false, SB + SITE_QKV>(c, s_qkv.c_str(), h, L.w_in_f, L.d_in, nullptr, qkv, M, 3 * D, D, &cons)));
            {
                ProfScope ps(*c.prof, c.stream, s_attn.c_str());
                ARP_TRY(launch_attention<T>(c.stream, c.attn_impl, qkv, ao, B, N, D, tw.heads, causal));
            }
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_OUT>(c, s_out.c_str(), ao, L.w_out, L.b_out, x, x, M, D, D, &prod)));
            cons.c = L.c_fc;
            ARP_TRY((tower_gemm<T, T, ACT, false, SB + SITE_FC1>(c, s_fc1.c_str(), h, L.w_fc_f, L.d_fc, nullptr, fc, M, 4 * D, D, &cons)));
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_FC2>(c, s_fc2.c_str(), fc, L.w_proj, L.b_proj, x, x, M, D, 4 * D, &prod)));
            ARP_TRY(tower_export_rows(c, x, D, i, B, N));
        }
        return 0;
    }
    // latency path: see TowerCtx::skinny
    const int S_out = tower_split_of(M, D, D), S_proj = tower_split_of(M, D, 4 * D);
    const bool lat = sizeof(T) == 2 && c.skinny && c.gemm_force == 0 && !c.fp8_mlp && c.part && M <= SKINNY_MAX_M && S_out && S_proj &&
                     (size_t)std::max(S_out, S_proj) * M * D <= c.part_floats && skinny_supported(M, 3 * D, D, D, D) && skinny_supported(M, 4 * D, D, D, D);
    const std::string s_red2 = t + ".out_reduce_ln_2", s_red1 = t + ".proj_reduce_ln_1";
    // ... with LayerNorm folded into the consumer GEMMs where the tower carries the folded operands (TowerW::lat_folded) and the caller's
    // token-assembly kernel has written the operand copy and the statistics of block 0
    const bool fold = lat && tw.lat_folded && c.lat_stats && c.lat_fold0 && (D & 15) == 0 && skinny_supported(M, D, 4 * D, 4 * D, 4 * D);
    bool h_ready = lat && !fold && c.h_ready0;  // h already holds ln_1 of this block (written by the previous block's reduction)
    bool qkv_ready = false;  // the class-token-only last block of a folded pass: in_proj already ran as a folded consumer
    for (int i = 0; i < tw.layers; ++i) {
        const LayerW& L = tw.L[i];
        if constexpr (sizeof(T) == 2) {
            const bool cls_blk = c.cls_only_last && i == tw.layers - 1 && N > 1;
            // (a folded pass reaches the class-token-only last block with h = the operand copy of x and its statistics in place: that
            //  block's in_proj is a consumer too -- no ln_1 launch -- and only its class rows go on through the unfolded kernels)
            if (fold && cls_blk && i > 0) {
                const int tcode = __is_same(T, bf16_t) ? 1 : 2;
                SkinnyArgs k;
                k.A = h; k.W = L.w_in_f; k.bias = L.d_in; k.out = qkv; k.M = M; k.N = 3 * D; k.K = D; k.lda = D; k.ldw = D; k.ldo = 3 * D;
                k.ln_stats = c.lat_stats; k.ln_c = L.c_in; k.ln_parts = D >> 4; k.ln_inv_d = 1.0f / (float)D; k.ln_eps = eps;
                {
                    ProfScope ps(*c.prof, c.stream, s_qkv.c_str());
                    ARP_TRY(launch_skinny_gemm(tcode, k, c.stream));
                }
                qkv_ready = true;
            }
            if (fold && !cls_blk) {
                // LayerNorm folded into the consumers (W diag(gamma); mean / rstd from the producers' per-strip sums): qkv, attention,
                // out_proj, c_fc, c_proj -- five launches, no reduce + LayerNorm kernels; h = the operand-type copy of x throughout
                const int tcode = __is_same(T, bf16_t) ? 1 : 2;
                auto consumer = [&](const char* site, const void* Wf, const float* dvec, const float* cvec, void* out, int Nout, int act) -> int {
                    SkinnyArgs k;
                    k.A = h; k.W = Wf; k.bias = dvec; k.out = out; k.M = M; k.N = Nout; k.K = D; k.lda = D; k.ldw = D; k.ldo = Nout; k.act = act;
                    k.ln_stats = c.lat_stats; k.ln_c = cvec; k.ln_parts = D >> 4; k.ln_inv_d = 1.0f / (float)D; k.ln_eps = eps;
                    ProfScope ps(*c.prof, c.stream, site);
                    return launch_skinny_gemm(tcode, k, c.stream);
                };
                auto producer = [&](const char* site, const void* A, const void* W, const float* bias, int K) -> int {
                    SkinnyArgs k;
                    k.A = A; k.W = W; k.bias = bias; k.resid = x; k.out = x; k.M = M; k.N = D; k.K = K; k.lda = K; k.ldw = K; k.ldr = D; k.ldo = D;
                    k.out_f32 = 1; k.strips = 1; k.xb = h; k.ldxb = D; k.stats_out = c.lat_stats;
                    ProfScope ps(*c.prof, c.stream, site);
                    return launch_skinny_gemm(tcode, k, c.stream);
                };
                ARP_TRY(consumer(s_qkv.c_str(), L.w_in_f, L.d_in, L.c_in, qkv, 3 * D, ACT_NONE));
                {
                    ProfScope ps(*c.prof, c.stream, s_attn.c_str());
                    ARP_TRY(launch_attention<T>(c.stream, c.attn_impl, qkv, ao, B, N, D, tw.heads, causal));
                }
                ARP_TRY(producer(s_out.c_str(), ao, L.w_out, L.b_out, D));
                ARP_TRY(consumer(s_fc1.c_str(), L.w_fc_f, L.d_fc, L.c_fc, fc, 4 * D, ACT));
                ARP_TRY(producer(s_fc2.c_str(), fc, L.w_proj, L.b_proj, 4 * D));
                ARP_TRY(tower_export_rows(c, x, D, i, B, N));
                continue;
            }
            if (lat && !(c.cls_only_last && i == tw.layers - 1 && N > 1)) {
                if (!h_ready) ARP_TRY(tower_layernorm<T>(c, s_ln1.c_str(), x, D, h, D, L.ln1_w, L.ln1_b, M, D, eps));
                ARP_TRY((tower_qkv_attention<T, SB + SITE_QKV>(c, tw, L, s_qkv.c_str(), s_attn.c_str(), s_qa.c_str(), h, qkv, ao, B, N, causal, 0)));
                ARP_TRY(tower_gemm_split_ln<T>(c, s_out.c_str(), s_red2.c_str(), ao, L.w_out, L.b_out, x, M, D, D, S_out, h, L.ln2_w, L.ln2_b, eps));
                ARP_TRY((tower_gemm<T, T, ACT, false, SB + SITE_FC1>(c, s_fc1.c_str(), h, L.w_fc, L.b_fc, nullptr, fc, M, 4 * D, D)));
                const LayerW* nx = i + 1 < tw.layers ? &tw.L[i + 1] : nullptr;
                ARP_TRY(tower_gemm_split_ln<T>(c, s_fc2.c_str(), s_red1.c_str(), fc, L.w_proj, L.b_proj, x, M, D, 4 * D, S_proj, h, nx ? nx->ln1_w : nullptr,
                                               nx ? nx->ln1_b : nullptr, eps));
                h_ready = nx != nullptr;
                ARP_TRY(tower_export_rows(c, x, D, i, B, N));
                continue;
            }
        }
        if (c.cls_only_last && i == tw.layers - 1 && N > 1) {
            const std::string s_attn1 = t + ".attn_cls", s_out1 = t + ".out_proj_cls", s_ln21 = t + ".ln_2_cls", s_fc11 = t + ".c_fc_cls",
                              s_fc21 = t + ".c_proj_cls";
            const int ND = N * D;  // row stride of the class-token rows inside the [B*N, D] buffers
            if (qkv_ready) {
                ProfScope ps(*c.prof, c.stream, s_attn1.c_str());
                ARP_TRY(launch_attention<T>(c.stream, c.attn_impl, qkv, ao, B, N, D, tw.heads, causal, 1));
            } else {
                if (!h_ready) ARP_TRY(tower_layernorm<T>(c, s_ln1.c_str(), x, D, h, D, L.ln1_w, L.ln1_b, M, D, eps));  // K and V need every token
                ARP_TRY((tower_qkv_attention<T, SB + SITE_QKV>(c, tw, L, s_qkv.c_str(), s_attn1.c_str(), s_qa1.c_str(), h, qkv, ao, B, N, causal, 1)));
            }
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_OUT>(c, s_out1.c_str(), ao, L.w_out, L.b_out, x, x, B, D, D, nullptr, ND, ND, ND)));
            ARP_TRY(tower_layernorm<T>(c, s_ln21.c_str(), x, ND, h, D, L.ln2_w, L.ln2_b, B, D, eps));
            ARP_TRY((tower_gemm<T, T, ACT, false, SB + SITE_FC1>(c, s_fc11.c_str(), h, L.w_fc, L.b_fc, nullptr, fc, B, 4 * D, D)));
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_FC2>(c, s_fc21.c_str(), fc, L.w_proj, L.b_proj, x, x, B, D, 4 * D, nullptr, 0, ND, ND)));
            ARP_TRY(tower_export_rows(c, x, D, i, B, N));
            break;
        }
        bool attn_done = false;
        if constexpr (sizeof(T) == 2) {
            if (c.fp8_attn && L.w_in8 && c.attn_impl == 0 && (D % 128) == 0 && attn_mfma_has(N, D / tw.heads) &&
                !(c.qkv_fused && L.w_in_hm && qkv_attn_supported(N, D, tw.heads, (int)sizeof(T)))) {
                const std::string s_ln18 = t + ".ln_1_fp8", s_qkv8 = t + ".qkv_fp8", s_attn8 = t + ".attn_out8", s_out8 = t + ".out_proj_fp8";
                ARP_TRY((tower_attn_fp8<T, SB>(c, tw, L, s_ln18.c_str(), s_qkv8.c_str(), s_attn8.c_str(), s_out8.c_str(), x, h, qkv, ao, B, N, causal, eps)));
                attn_done = true;
            }
        }
        if (!attn_done) {
            ARP_TRY(tower_layernorm<T>(c, s_ln1.c_str(), x, D, h, D, L.ln1_w, L.ln1_b, M, D, eps));
            ARP_TRY((tower_qkv_attention<T, SB + SITE_QKV>(c, tw, L, s_qkv.c_str(), s_attn.c_str(), s_qa.c_str(), h, qkv, ao, B, N, causal, 0)));
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_OUT>(c, s_out.c_str(), ao, L.w_out, L.b_out, x, x, M, D, D)));
        }
        if (sizeof(T) == 2 && c.fp8_mlp && L.w_fc8 && (D % 128) == 0) {
            const std::string s_fc18 = t + ".c_fc_fp8", s_fc28 = t + ".c_proj_fp8";
            ARP_TRY((tower_mlp_fp8<T, ACT, SB>(c, L, s_ln2.c_str(), s_fc18.c_str(), s_fc28.c_str(), x, h, fc, M, D, eps)));
        } else {
            ARP_TRY(tower_layernorm<T>(c, s_ln2.c_str(), x, D, h, D, L.ln2_w, L.ln2_b, M, D, eps));
            ARP_TRY((tower_gemm<T, T, ACT, false, SB + SITE_FC1>(c, s_fc1.c_str(), h, L.w_fc, L.b_fc, nullptr, fc, M, 4 * D, D)));
            ARP_TRY((tower_gemm<T, float, ACT_NONE, true, SB + SITE_FC2>(c, s_fc2.c_str(), fc, L.w_proj, L.b_proj, x, x, M, D, 4 * D)));
        }
        ARP_TRY(tower_export_rows(c, x, D, i, B, N));
    }
    return 0;
}

}  // namespace arp

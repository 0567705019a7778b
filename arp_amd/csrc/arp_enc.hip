// Row N1: the frozen M3AE image encoder (forward_representation) on MI355X -- host orchestration + C ABI.
// Reference: /root/reference/arp_dt/models/m3ae/model.py:471-496 (forward_representation), :200-312 (blocks),
// :95-136 (sincos pos-emb), arp_dt/ARPDT.py:111-116,413-458 (patchify, call site under stop_gradient).
// Same kernels as the CLIP image tower (tower.h) with tanh-GELU, LayerNorm eps 1e-6, a biased patch
// embedding, fixed 2-D sincos position embedding + type embedding, no ln_pre and a final LN over all tokens.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/arp_hip.h"
#include "enc_internal.h"
#include "tower.h"

using namespace arp;

namespace {

// images f32 NHWC -> patch matrix [n*G*G, P*P*3] (T), "b (h p1) (w p2) c -> b (h w) (p1 p2 c)"
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int n, int res, int P) {
    const int G = res / P, K = P * P * 3;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (size_t)n * G * G * K) return;
    const int k = (int)(i % K);
    const size_t row = i / K;
    const int px = (int)(row % G), py = (int)((row / G) % G);
    const size_t b = row / ((size_t)G * G);
    const int p1 = k / (P * 3), rem = k - p1 * P * 3;  // rem = p2*3 + c, contiguous in the source row
    const float* src = img + ((b * res + (size_t)py * P + p1) * res + (size_t)px * P) * 3 + rem;
    store4(out + i, src[0], src[1], src[2], src[3]);
}

// x[b, 0] = cls;  x[b, 1+p] = pe[b*GG + p] + pos[p]   (pos already holds sincos + type embedding)
__global__ __launch_bounds__(256) void enc_assemble_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int rows, int ntok,
                                                           int D) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (size_t)rows * D) return;
    const int c = (int)(i % D);
    const size_t row = i / D;
    const int t = (int)(row % ntok);
    const size_t b = row / ntok;
    float v[4];
    if (t == 0) {
        load4(cls + c, v);
    } else {
        float p4[4];
        load4(pe + (b * (ntok - 1) + (t - 1)) * D + c, v);
        load4(pos + (size_t)(t - 1) * D + c, p4);
        v[0] += p4[0]; v[1] += p4[1]; v[2] += p4[2]; v[3] += p4[3];
    }
    store4(x + i, v[0], v[1], v[2], v[3]);
}

// ARP_MODE_F16X3: f32 [rows, K] -> binary16 [rows, 3K] = [hi | lo | hi], hi = rn16(x), lo = rn16(x - hi): the A operand of a K-concatenated product against
// [W_hi | W_hi | W_lo] -- x.W to ~2^-22 on three 16-bit MFMAs (hi.hi is exact in the f32 accumulator; lo.lo, ~2^-24, is dropped).  Values whose lo falls below
// binary16's normal range keep an absolute error <= 3e-8 (subnormals are not flushed on this path).  8 elements per thread.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, f16_t* __restrict__ out, size_t rows, int K) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= rows * (size_t)K) return;
    const size_t r = i / K;
    const int k = (int)(i - r * K);
    float va[4], vb[4];
    load4(x + i, va);
    load4(x + i + 4, vb);
    const float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float h0 = h2f(f2h(v[2 * j])), h1 = h2f(f2h(v[2 * j + 1]));
        hi[j] = pack_h2(h0, h1);
        lo[j] = pack_h2(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
    f16_t* o = out + r * 3 * (size_t)K + k;
    *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(o + K) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    *reinterpret_cast<uint4*>(o + 2 * (size_t)K) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
}

// patchify_kernel<float> + split3_kernel in one pass (round 6; f16x3 and f16c modes): the f32 patch matrix is never written (100 MB per 128 frames written and
// read back).  Same values: each element is loaded once, hi = rn16(v), lo = rn16(v - hi) as split3_kernel forms them.
__global__ __launch_bounds__(256) void patchify_split3_kernel(const float* __restrict__ img, f16_t* __restrict__ out, int n, int res, int P) {
    const int G = res / P, K = P * P * 3;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (size_t)n * G * G * K) return;
    const int k = (int)(i % K);
    const size_t row = i / K;
    const int px = (int)(row % G), py = (int)((row / G) % G);
    const size_t b = row / ((size_t)G * G);
    const int p1 = k / (P * 3), rem = k - p1 * P * 3;
    const float* src = img + ((b * res + (size_t)py * P + p1) * res + (size_t)px * P) * 3 + rem;
    const float v[4] = {src[0], src[1], src[2], src[3]};
    uint32_t hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float h0 = h2f(f2h(v[2 * j])), h1 = h2f(f2h(v[2 * j + 1]));
        hi[j] = pack_h2(h0, h1);
        lo[j] = pack_h2(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
    f16_t* o = out + row * 3 * (size_t)K + k;
    *reinterpret_cast<uint2*>(o) = make_uint2(hi[0], hi[1]);
    *reinterpret_cast<uint2*>(o + K) = make_uint2(lo[0], lo[1]);
    *reinterpret_cast<uint2*>(o + 2 * (size_t)K) = make_uint2(hi[0], hi[1]);
}

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

}  // namespace

struct arp_enc {
    arp_enc_cfg cfg;
    hipStream_t stream = nullptr;
    std::map<std::string, HostTensor> staged;
    std::vector<void*> owned;
    bool finalized = false;
    TowerW tower;
    void* w_emb = nullptr;  // T [D, P*P*3]
    float *b_emb = nullptr, *cls = nullptr, *pos = nullptr, *lnf_w = nullptr, *lnf_b = nullptr;
    // Part streams (round 6, as arp_clip.hip::label_dev): a call's frames are cut into contiguous parts, part 0 on the caller's stream, part k on ws[k].stream,
    // forked and joined by events -- one part's HBM-bound LayerNorm / attention kernels and GEMM grid tails (M = 128 x 257 rows: c_fc 6.05 rounds of tiles,
    // out_proj / c_proj 1.51) run beside the other part's GEMMs.  Every part has its own workspace; weights are shared.  A frame's encoding does not depend on
    // which part it is in (every row's dot products run in the same order whatever M is): tests/test_m3ae_gpu.py::test_part_streams_are_exact.
    struct Ws {
        int frames = 0;
        DevBuf patches, pe, x, h, qkv, ao, fc;
        DevBuf a3;  // ARP_MODE_F16X3: the [hi | lo | hi] operand of the current GEMM, binary16 [rows, 3 * (mlp_ratio * width)]
                    // ARP_MODE_F16C: the [hi | x4 | dx4] operand rows (3 bytes per value): [M, D] for LayerNorm / attention outputs, then [M, H] for the hidden activation
        hipStream_t stream = nullptr;  // parts 1..: the part's own stream (part 0 runs on the caller's)
        hipEvent_t done = nullptr;
    };
    static constexpr int MAX_PARTS = 4;
    Ws ws[MAX_PARTS];
    int n_parts = 2;          // arp_enc_set_streams / ARP_ENC_STREAMS
    int min_part_frames = 24; // a part of fewer frames does not fill the chip's GEMM grid (24 x 257 rows = 24 row tiles x 3..12 column tiles)
    // frames of part 0 of two (ARP_ENC_SPLIT); 0 = the default cut, 15/32 of the frames (60 + 68 of the step's 128); < 0 = equal parts.  Uneven parts end at different
    // times, so that one part's GEMM grid tails and LayerNorms keep meeting the other's full rounds instead of its tails: 9.05-9.08 against 9.19-9.26 ms per 32-sample step
    // (three boxes, three repetitions each: profiles/r6_n1_streams.txt, r6_plans_time.txt, r6_n1_split_final.txt); 56 + 72 the same, 43 + 43 + 42 slower
    int first_part = 0;
    hipEvent_t ev_fork = nullptr;
    bool shared_chip = false; // the pass being enqueued runs beside another part's kernels (tower.h: out_proj's kernel choice)
    DevBuf img_in, out;
    // ARP_MODE_F16C: per GEMM g in {in_proj, out_proj, fc1, fc2} the correction plan (0 plain, 1 weights, 2 weights + activations) and, per layer, the
    // power-of-two exponents the e2m1 (fp4) weight segments were scaled by: dW4 = fp4(dW * 2^sw_d), W4 = fp4(W * 2^sw_w)
    bool vperm = true;  // ARP_MODE_F16C: V's columns permuted inside every head so that the attention's [hi | x4 | dx4] rows leave in whole pieces (attention.h, outc == 2); ARP_F16C_VPERM=0: round 5's stores
    // Default 1110 (round 6): the weight roundings of in_proj, out_proj and fc1 corrected, fc2 plain.  Encoder-inside logits over 8 seeds (max) and the 32-sample step
    // on one box, AFTER the epilogue's x4 segment was repaired (gemm256.h; rounds 5 and 6 measured fc2's correction as noise): 1221 3.7e-4 / 10.19 ms,
    // 1121 5.4e-4, 1111 6.0e-4 / 9.82, 1220 6.4e-4 / 9.83, 1211 6.9e-4 / 9.88, 1210 6.8e-4 / 9.54, 1120 6.7e-4 / 9.76, 1110 7.0e-4 / 9.48 (profiles/r6_n1_plan_sweep.txt,
    // r6_plans_time.txt): one plan is clearly better (1221, round 5's, for +0.7 ms) and the rest are one cloud at 5.4 ... 7.0e-4 -- the cheapest of it is the default,
    // ARP_F16C_PLAN=1221 buys the other.  Producers skip the segments a plan never reads (the attention's dx4, fc1's x4).
    int plan[4] = {1, 1, 1, 0};
    std::vector<int> sw_d[4], sw_w[4];
    void* w_emb3 = nullptr;  // ARP_MODE_F16C: the patch embedding's [W_hi | W_hi | W_lo] (its product runs as ARP_MODE_F16X3's K-concatenation)
    Profiler prof;
    int gemm_force = 0;
    int tokens() const { return (cfg.img_res / cfg.patch) * (cfg.img_res / cfg.patch) + 1; }
    size_t esz() const { return (cfg.mode == ARP_MODE_F32 || cfg.mode == ARP_MODE_F16X3) ? 4 : 2; }  // element size of the ACTIVATION buffers
    size_t wsz() const { return cfg.mode == ARP_MODE_F32 ? 4 : 2; }
};

namespace {

int up_f32(arp_enc* c, const float* v, size_t n, float** out) {
    void* p = nullptr;
    ARP_HIP_OK(hipMalloc(&p, std::max<size_t>(n * 4, 16)));
    ARP_HIP_OK(hipMemcpy(p, v, n * 4, hipMemcpyHostToDevice));
    c->owned.push_back(p);
    *out = static_cast<float*>(p);
    return 0;
}
// [out, 3 in] = [W_hi | W_hi | W_lo] from the transposed kernel t [out, in] (ARP_MODE_F16X3; the patch embedding of ARP_MODE_F16C)
int up_kernel_x3(arp_enc* c, const std::vector<float>& t, int in, int out_, void** out) {
    const size_t n = (size_t)in * out_;
    void* p = nullptr;
    std::vector<f16_t> hb(3 * n);
    for (int o = 0; o < out_; ++o)
        for (int i = 0; i < in; ++i) {
            const float w = t[(size_t)o * in + i];
            const f16_t h = host_f2h(w);
            const float hf = (float)__builtin_bit_cast(_Float16, h.b);
            f16_t* row = hb.data() + (size_t)o * 3 * in;
            row[i] = h;
            row[in + i] = h;
            row[2 * in + i] = host_f2h(w - hf);
        }
    ARP_HIP_OK(hipMalloc(&p, 3 * n * 2));
    ARP_HIP_OK(hipMemcpy(p, hb.data(), 3 * n * 2, hipMemcpyHostToDevice));
    c->owned.push_back(p);
    *out = p;
    return 0;
}
// Flax kernel [in, out] -> device [out, in] in the operand type
int up_kernel(arp_enc* c, const float* src, int in, int out_, void** out) {
    const size_t n = (size_t)in * out_;
    std::vector<float> t(n);
    for (int i = 0; i < in; ++i)
        for (int o = 0; o < out_; ++o) t[(size_t)o * in + i] = src[(size_t)i * out_ + o];
    void* p = nullptr;
    if (c->cfg.mode == ARP_MODE_F16X3) return up_kernel_x3(c, t, in, out_, out);
    ARP_HIP_OK(hipMalloc(&p, std::max<size_t>(n * c->wsz(), 16)));
    if (c->cfg.mode == ARP_MODE_BF16) {
        std::vector<bf16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2bf(t[i]);
        ARP_HIP_OK(hipMemcpy(p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else if (c->cfg.mode == ARP_MODE_F16 || c->cfg.mode == ARP_MODE_F16C) {
        std::vector<f16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2h(t[i]);
        ARP_HIP_OK(hipMemcpy(p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        ARP_HIP_OK(hipMemcpy(p, t.data(), n * 4, hipMemcpyHostToDevice));
    }
    c->owned.push_back(p);
    *out = p;
    return 0;
}
// ARP_MODE_F16C: Flax kernel [in, out] -> device rows [W_hi: binary16 x in | dW4: e2m1 x in (| W4: e2m1 x in)] (plan 1 / 2; plan 0: binary16 only)
// src(i, o) = the weight of input i, output o.  Rows [W_hi: binary16 x in | dW4: e2m1 x in (| W4: e2m1 x in)] for plan 1 (2); two e2m1 values per byte, value 2j in the
// low nibble (the order common.h::pack_fp4x8 packs activations in).  dW4 = fp4((w - W_hi) 2^*sd), W4 = fp4(w 2^*sw): per-tensor powers of two that put the largest
// magnitude into (6, 12] (e2m1 saturates at 6: the few values of the top half-binade lose part of their correction, every other value gains a bit).
template <typename F> void pack_weight_c(F src, int in, int out_, int plan, std::vector<uint8_t>& hb, int* sd, int* sw) {
    const size_t row_bytes = (size_t)in * 2 + (size_t)plan * (in / 2);
    hb.assign(row_bytes * out_, 0);
    float max_d = 0.f, max_w = 0.f;
    for (int i = 0; i < in; ++i)
        for (int o = 0; o < out_; ++o) {
            const float w = src(i, o);
            const float hf = (float)__builtin_bit_cast(_Float16, host_f2h(w).b);
            max_d = std::max(max_d, std::fabs(w - hf));
            max_w = std::max(max_w, std::fabs(w));
        }
    auto pick = [](float mx) { return mx > 0.f ? (int)std::floor(std::log2(6.0 / (double)mx)) + 1 : 0; };  // the top binade saturates at 6: measured better than wasting a code on it
    *sd = std::min(pick(max_d), 100);
    *sw = std::min(pick(max_w), 100);
    const float fd = std::ldexp(1.0f, *sd), fw = std::ldexp(1.0f, *sw);
    for (int o = 0; o < out_; ++o) {
        uint8_t* row = hb.data() + (size_t)o * row_bytes;
        for (int i = 0; i < in; ++i) {
            const float w = src(i, o);
            const f16_t h = host_f2h(w);
            const float hf = (float)__builtin_bit_cast(_Float16, h.b);
            memcpy(row + 2 * (size_t)i, &h.b, 2);
            const int sh = (i & 1) * 4;
            if (plan >= 1) row[2 * (size_t)in + i / 2] |= (uint8_t)(host_f2fp4((w - hf) * fd) << sh);
            if (plan >= 2) row[2 * (size_t)in + in / 2 + i / 2] |= (uint8_t)(host_f2fp4(w * fw) << sh);
        }
    }
}
int up_kernel_c(arp_enc* c, const float* src, int in, int out_, int plan, void** out, int* sd, int* sw) {
    std::vector<uint8_t> hb;
    pack_weight_c([&](int i, int o) { return src[(size_t)i * out_ + o]; }, in, out_, plan, hb, sd, sw);
    void* p = nullptr;
    ARP_HIP_OK(hipMalloc(&p, hb.size() + 512));  // + the 256 bytes gemm256 MIXC reads past the last row (GemmArgs::mix_nk16)
    ARP_HIP_OK(hipMemcpy(p, hb.data(), hb.size(), hipMemcpyHostToDevice));
    c->owned.push_back(p);
    *out = p;
    return 0;
}
int staged(arp_enc* c, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
    auto it = c->staged.find(name);
    if (it == c->staged.end()) return fail("missing weight: " + name);
    if (it->second.shape != shape) return fail("weight " + name + " has an unexpected shape");
    *out = &it->second;
    return 0;
}

int ensure_ws(arp_enc* c, arp_enc::Ws& w, int frames) {
    if (frames <= w.frames) return 0;
    const arp_enc_cfg& k = c->cfg;
    const size_t e = c->esz(), G = k.img_res / k.patch, D = k.width, N = c->tokens(), B = frames, M = B * N;
    ARP_TRY(w.patches.ensure(B * G * G * k.patch * k.patch * 3 * e));
    ARP_TRY(w.pe.ensure(B * G * G * D * 4));
    ARP_TRY(w.x.ensure(M * D * 4)); ARP_TRY(w.h.ensure(M * D * e)); ARP_TRY(w.qkv.ensure(M * 3 * D * e));
    ARP_TRY(w.ao.ensure(M * D * e)); ARP_TRY(w.fc.ensure(M * k.mlp_ratio * D * e));
    if (k.mode == ARP_MODE_F16X3)  // [M, 3 D] (a GEMM's A operand) followed by [M, 3 H] (c_fc's own epilogue writes c_proj's operand there)
        ARP_TRY(w.a3.ensure(std::max(M * 3 * (D + k.mlp_ratio * D), B * G * G * 3 * k.patch * k.patch * 3) * 2));
    if (k.mode == ARP_MODE_F16C) {  // [M, D] and [M, H] operand rows of 3 bytes per value; the patch embedding's (hi, lo, hi) triples share the space
        // (+ one 256-row tile of slack: the MIXC instances read ceil(M / 256) * 256 operand rows unclamped)
        ARP_TRY(w.a3.ensure(std::max((M + 256) * 3 * (D + k.mlp_ratio * D) + 4096, B * G * G * 3 * k.patch * k.patch * 3 * 2)));
        ARP_TRY(w.patches.ensure(B * G * G * k.patch * k.patch * 3 * 4));  // f32 patches (split into triples on the device)
    }
    w.frames = frames;
    return 0;
}

template <typename T> int forward_chunk(arp_enc* c, arp_enc::Ws& w, hipStream_t stream, const float* img_dev, int nb, float* out_dev) {
    const arp_enc_cfg& k = c->cfg;
    const int G = k.img_res / k.patch, N = c->tokens(), D = k.width, KP = k.patch * k.patch * 3;
    TowerCtx t;
    t.stream = stream; t.prof = &c->prof; t.attn_impl = k.attn_impl; t.gemm_force = c->gemm_force; t.shared_chip = c->shared_chip;
    {
        ProfScope ps(c->prof, stream, "m3ae.patchify");
        const size_t tot = (size_t)nb * G * G * KP;
        hipLaunchKernelGGL((patchify_kernel<T>), dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, img_dev, w.patches.as<T>(), nb,
                           k.img_res, k.patch);
        ARP_HIP_OK(hipGetLastError());
    }
    ARP_TRY((tower_gemm<T, float, ACT_NONE, false, 8 + SITE_PATCH>(t, "m3ae.image_embedding", w.patches.p, c->w_emb, c->b_emb, nullptr, w.pe.p,
                                                                   nb * G * G, D, KP)));
    {
        ProfScope ps(c->prof, stream, "m3ae.assemble");
        const size_t tot = (size_t)nb * N * D;
        hipLaunchKernelGGL(enc_assemble_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, w.pe.as<float>(), c->cls, c->pos,
                           w.x.as<float>(), nb * N, N, D);
        ARP_HIP_OK(hipGetLastError());
    }
    ARP_TRY((run_blocks<T, ACT_GELU_TANH, 8>(t, c->tower, "m3ae", w.x.as<float>(), w.h.as<T>(), w.qkv.as<T>(), w.ao.as<T>(), w.fc.as<T>(), nb, N, 0,
                                             1e-6f)));
    ARP_TRY(tower_layernorm<float>(t, "m3ae.ln_final", w.x.as<float>(), (size_t)D, out_dev, D, c->lnf_w, c->lnf_b, nb * N, D, 1e-6f));
    return 0;
}

// ARP_MODE_F16X3: the same network with every GEMM as a K-concatenated (hi, lo) product on the f16 kernels and everything between the GEMMs in f32
// (LayerNorm outputs, the exact-f32 attention kernel, the tanh-GELU'd hidden activation): 12 x (ln -> split -> in_proj -> attention -> split -> out_proj -> ln ->
// split -> fc1 -> split -> fc2).  Three MFMAs per product plus the split passes: ~4x the plain f16 step, ~2x faster than the f32-MFMA mode, f32-level error.
int forward_chunk_x3(arp_enc* c, arp_enc::Ws& w, hipStream_t stream, const float* img_dev, int nb, float* out_dev) {
    const arp_enc_cfg& k = c->cfg;
    const int G = k.img_res / k.patch, N = c->tokens(), D = k.width, KP = k.patch * k.patch * 3, H = k.mlp_ratio * D, M = nb * N;
    TowerCtx t;
    t.stream = stream; t.prof = &c->prof; t.attn_impl = k.attn_impl; t.gemm_force = c->gemm_force; t.shared_chip = c->shared_chip;
    f16_t* a3 = w.a3.as<f16_t>();
    auto split = [&](const char* site, const float* src, size_t rows, int K) -> int {
        if (K % 8) return fail("f16x3: widths must be multiples of 8");
        ProfScope ps(c->prof, stream, site);
        hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((rows * K / 8 + 255) / 256)), dim3(256), 0, stream, src, a3, rows, K);
        ARP_HIP_OK(hipGetLastError());
        return 0;
    };
    {   // frames -> (hi, lo, hi) patch triples in one pass
        ProfScope ps(c->prof, stream, "m3ae.patchify");
        const size_t tot = (size_t)nb * G * G * KP;
        hipLaunchKernelGGL(patchify_split3_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, img_dev, a3, nb, k.img_res, k.patch);
        ARP_HIP_OK(hipGetLastError());
    }
    ARP_TRY((tower_gemm<f16_t, float, ACT_NONE, false, 8 + SITE_PATCH>(t, "m3ae.image_embedding", a3, c->w_emb, c->b_emb, nullptr, w.pe.p, nb * G * G, D, 3 * KP)));
    {
        ProfScope ps(c->prof, stream, "m3ae.assemble");
        const size_t tot = (size_t)nb * N * D;
        hipLaunchKernelGGL(enc_assemble_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, w.pe.as<float>(), c->cls, c->pos, w.x.as<float>(), nb * N, N, D);
        ARP_HIP_OK(hipGetLastError());
    }
    float *x = w.x.as<float>(), *qkv = w.qkv.as<float>(), *ao = w.ao.as<float>();
    for (int i = 0; i < k.layers; ++i) {
        const LayerW& L = c->tower.L[i];
        // LayerNorm and the attention write the (hi, lo, hi) triples of their f32 results themselves (f16x3_t / out3): no f32 copy, no split pass
        const bool direct = k.attn_impl == 0 && D / k.heads == 64 && N <= 288;  // (the VALU attention has no such output: it keeps the split pass)
        ARP_TRY(tower_layernorm<f16x3_t>(t, "m3ae.ln_1", x, D, reinterpret_cast<f16x3_t*>(a3), 3 * D, L.ln1_w, L.ln1_b, M, D, 1e-6f));
        ARP_TRY((tower_gemm<f16_t, float, ACT_NONE, false, 8 + SITE_QKV>(t, "m3ae.qkv", a3, L.w_in, L.b_in, nullptr, qkv, M, 3 * D, 3 * D)));
        {
            ProfScope ps(c->prof, stream, "m3ae.attn");
            // the (hi, lo) binary16 attention (attention.h::attn_x3_kernel) where it exists; ARP_ENC_ATTN_X3=0 keeps the exact-f32 MFMA kernel
            static const bool attn_x3 = [] { const char* e = getenv("ARP_ENC_ATTN_X3"); return !e || atoi(e) != 0; }();
            ARP_TRY(launch_attention<float>(stream, direct && attn_x3 ? 3 : k.attn_impl, qkv, ao, nb, N, D, k.heads, 0, 0, 0.f, direct ? a3 : nullptr));
        }
        if (!direct) ARP_TRY(split("m3ae.split", ao, M, D));
        ARP_TRY((tower_gemm<f16_t, float, ACT_NONE, true, 8 + SITE_OUT>(t, "m3ae.out_proj", a3, L.w_out, L.b_out, x, x, M, D, 3 * D)));
        ARP_TRY(tower_layernorm<f16x3_t>(t, "m3ae.ln_2", x, D, reinterpret_cast<f16x3_t*>(a3), 3 * D, L.ln2_w, L.ln2_b, M, D, 1e-6f));
        // c_fc's epilogue writes the (hi, lo, hi) triples of the tanh-GELU'd hidden activation itself: no f32 copy of it, no split pass (2.4 ms of a 30.9 ms step)
        f16_t* a3b = a3 + (size_t)M * 3 * D;
        GemmFold pair;
        pair.xb_out = a3b; pair.ldxb = 3 * H; pair.split3 = 1;
        ARP_TRY((tower_gemm<f16_t, float, ACT_GELU_TANH, false, 8 + SITE_FC1>(t, "m3ae.c_fc", a3, L.w_fc, L.b_fc, nullptr, nullptr, M, H, 3 * D, &pair)));
        ARP_TRY((tower_gemm<f16_t, float, ACT_NONE, true, 8 + SITE_FC2>(t, "m3ae.c_proj", a3b, L.w_proj, L.b_proj, x, x, M, D, 3 * H)));
    }
    ARP_TRY(tower_layernorm<float>(t, "m3ae.ln_final", x, (size_t)D, out_dev, D, c->lnf_w, c->lnf_b, M, D, 1e-6f));
    return 0;
}

// ARP_MODE_F16C: the binary16 encoder with its GEMMs' operand roundings corrected on the scaled fp4 MFMA (include/arp_hip.h).  Per block:
//   ln_1 -> [hi | x4 (| dx4)] -> in_proj (MIXC) -> qkv binary16 -> MFMA attention -> [hi | x4 | dx4] -> out_proj (MIXC) into the f32 residual stream ->
//   ln_2 -> [hi | x4 (| dx4)] -> fc1 (MIXC, tanh-GELU; the epilogue stores hi AND the e2m1 segment of fc2's operand) -> fc2 (MIXC) into the residual stream.
template <int ACT, bool RESID, typename OutT, int SITE>
int gemm_c(arp_enc* c, TowerCtx& t, const char* site, const void* A, const void* W, int plan, int sd, int sw, const float* bias, const float* resid, void* out,
           int M, int N, int Kc, int ldo, void* x4_out = nullptr, int ld4 = 0, void* dx4_out = nullptr) {
    // A rows: binary16 x Kc, then e2m1 x Kc (x4), then e2m1 x Kc (dx4): 3 Kc bytes; W rows: binary16 x Kc followed by `plan` e2m1 segments
    if (Kc % 256) return fail("f16c: widths must be multiples of 256");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.resid = resid; g.out = out;
    g.M = M; g.N = N;
    g.lda = Kc + Kc / 2; g.ldw = Kc + plan * Kc / 4; g.ldr = N; g.ldo = ldo;
    g.mix_nk16 = Kc / 64;
    g.mix_nkc_a = plan >= 1 ? Kc / 256 : 0;
    g.K = Kc + plan * Kc / 4;
    g.mix_sa = F16C_X_SHIFT + sd;
    g.mix_sb = F16C_DX_SHIFT + sw;
    if (x4_out) { g.xb_out = x4_out; g.ldxb = ld4; g.x8_shift = F16C_X_SHIFT; g.dx4_out = dx4_out; }
    ProfScope ps(*t.prof, t.stream, site);
    if (plan == 0) return launch_gemm256_nt<f16_t, OutT, ACT, RESID, SITE>(g, t.stream);  // (no fp4 side output on this instance: plan 0 is for probing only)
    return launch_gemm256_nt<f16_t, OutT, ACT, RESID, SITE, false, 1, true>(g, t.stream);
}

int forward_chunk_c(arp_enc* c, arp_enc::Ws& w, hipStream_t stream, const float* img_dev, int nb, float* out_dev) {
    const arp_enc_cfg& k = c->cfg;
    const int G = k.img_res / k.patch, N = c->tokens(), D = k.width, KP = k.patch * k.patch * 3, H = k.mlp_ratio * D, M = nb * N;
    if (D / k.heads != 64 || N > 288 || k.attn_impl != 0) return fail("f16c: needs the MFMA attention kernel (head_dim 64, <= 288 tokens)");
    TowerCtx t;
    t.stream = stream; t.prof = &c->prof; t.attn_impl = 0; t.gemm_force = c->gemm_force; t.shared_chip = c->shared_chip;
    char* a4 = static_cast<char*>(w.a3.p);                 // [M, D] operand rows, 3 D bytes each: [hi | x4 | dx4]
    char* a4h = a4 + (((size_t)M * 3 * D + 255) & ~(size_t)255);  // [M, H] operand rows, 3 H bytes each (fc1's epilogue -> fc2)
    {   // patch embedding on (hi, lo) binary16 pairs (its input rounding alone is a third of the plain f16 encoder's logit error; the product is 0.6 % of the FLOPs)
        ProfScope ps(c->prof, stream, "m3ae.patchify");
        const size_t tot = (size_t)nb * G * G * KP;
        hipLaunchKernelGGL(patchify_split3_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, img_dev, reinterpret_cast<f16_t*>(a4), nb, k.img_res, k.patch);
        ARP_HIP_OK(hipGetLastError());
    }
    ARP_TRY((tower_gemm<f16_t, float, ACT_NONE, false, 8 + SITE_PATCH>(t, "m3ae.image_embedding", a4, c->w_emb3, c->b_emb, nullptr, w.pe.p, nb * G * G, D, 3 * KP)));
    {
        ProfScope ps(c->prof, stream, "m3ae.assemble");
        const size_t tot = (size_t)nb * N * D;
        hipLaunchKernelGGL(enc_assemble_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, stream, w.pe.as<float>(), c->cls, c->pos, w.x.as<float>(), nb * N, N, D);
        ARP_HIP_OK(hipGetLastError());
    }
    float* x = w.x.as<float>();
    f16_t* qkv = w.qkv.as<f16_t>();
    auto ln = [&](const char* site, const float* w, const float* b, int plan) -> int {
        if (plan >= 2) return tower_layernorm<f16c2_t>(t, site, x, D, reinterpret_cast<f16c2_t*>(a4), 3 * D / 2, w, b, M, D, 1e-6f);
        return tower_layernorm<f16c_t>(t, site, x, D, reinterpret_cast<f16c_t*>(a4), 3 * D / 2, w, b, M, D, 1e-6f);
    };
    for (int i = 0; i < k.layers; ++i) {
        const LayerW& L = c->tower.L[i];
        ARP_TRY(ln("m3ae.ln_1", L.ln1_w, L.ln1_b, c->plan[0]));
        ARP_TRY((gemm_c<ACT_NONE, false, f16_t, 8 + SITE_QKV>(c, t, "m3ae.qkv", a4, L.w_in, c->plan[0], c->sw_d[0][i], c->sw_w[0][i], L.b_in, nullptr, qkv, M, 3 * D, D, 3 * D)));
        {
            ProfScope ps(c->prof, stream, "m3ae.attn");
            ARP_TRY(launch_attention<f16_t>(stream, 0, qkv, reinterpret_cast<f16_t*>(a4), nb, N, D, k.heads, 0, 0, 0.f, nullptr, (c->vperm ? 2 : 1) | (c->plan[1] >= 2 ? 0 : 4)));
        }
        ARP_TRY((gemm_c<ACT_NONE, true, float, 8 + SITE_OUT>(c, t, "m3ae.out_proj", a4, L.w_out, c->plan[1], c->sw_d[1][i], c->sw_w[1][i], L.b_out, x, x, M, D, D, D)));
        ARP_TRY(ln("m3ae.ln_2", L.ln2_w, L.ln2_b, c->plan[2]));
        // fc1's epilogue stores the binary16 hidden activation at the head of fc2's operand rows and its e2m1 copy behind it (row stride 3 H bytes)
        ARP_TRY((gemm_c<ACT_GELU_TANH, false, f16_t, 8 + SITE_FC1>(c, t, "m3ae.c_fc", a4, L.w_fc, c->plan[2], c->sw_d[2][i], c->sw_w[2][i], L.b_fc, nullptr, a4h, M, H, D, 3 * H / 2,
                                                                  c->plan[3] >= 1 ? a4h + 2 * (size_t)H : nullptr, 3 * H, c->plan[3] >= 2 ? a4h + 2 * (size_t)H + H / 2 : nullptr)));
        ARP_TRY((gemm_c<ACT_NONE, true, float, 8 + SITE_FC2>(c, t, "m3ae.c_proj", a4h, L.w_proj, c->plan[3], c->sw_d[3][i], c->sw_w[3][i], L.b_proj, x, x, M, D, H, D)));
    }
    ARP_TRY(tower_layernorm<float>(t, "m3ae.ln_final", x, (size_t)D, out_dev, D, c->lnf_w, c->lnf_b, M, D, 1e-6f));
    return 0;
}

}  // namespace

namespace arp {

static int forward_part(arp_enc* c, arp_enc::Ws& w, hipStream_t stream, const float* img, int nb, float* out) {
    ARP_TRY(ensure_ws(c, w, nb));
    if (c->cfg.mode == ARP_MODE_F16X3) return forward_chunk_x3(c, w, stream, img, nb, out);
    if (c->cfg.mode == ARP_MODE_F16C) return forward_chunk_c(c, w, stream, img, nb, out);
    if (c->cfg.mode == ARP_MODE_BF16) return forward_chunk<bf16_t>(c, w, stream, img, nb, out);
    if (c->cfg.mode == ARP_MODE_F16) return forward_chunk<f16_t>(c, w, stream, img, nb, out);
    return forward_chunk<float>(c, w, stream, img, nb, out);
}

// Enqueues the encoder of `n` device-resident frames behind everything already on `stream`; `stream` continues behind the last part.  Safe under a
// stream capture of `stream` (the part streams join the capture through the fork event and leave it through the join events), but every buffer must
// exist beforehand: the first call of a geometry allocates and must run eagerly (arp_dt.hip runs two eager steps before it captures).
int enc_forward_on(arp_enc* c, hipStream_t stream, const float* images_dev, int n, float* out_dev) {
    if (!c || !c->finalized) return fail("encoder weights not finalized");
    if (n <= 0) return 0;
    const int mb = c->cfg.max_frames;
    const size_t fi = (size_t)c->cfg.img_res * c->cfg.img_res * 3, fo = (size_t)c->tokens() * c->cfg.width;
    for (int off = 0; off < n; off += mb) {
        const int nb = std::min(mb, n - off);
        const float* img = images_dev + off * fi;
        float* out = out_dev + off * fo;
        int parts = std::min(c->n_parts, (int)arp_enc::MAX_PARTS);
        while (parts > 1 && nb / parts < c->min_part_frames) --parts;
        if (parts < 2) {
            ARP_TRY(forward_part(c, c->ws[0], stream, img, nb, out));
            continue;
        }
        // contiguous parts: part 0 on the caller's stream, part k on its own
        int cut[arp_enc::MAX_PARTS + 1];
        cut[0] = 0;
        for (int i = 1; i <= parts; ++i) cut[i] = (int)((long long)nb * i / parts);
        if (parts == 2 && c->first_part > 0 && c->first_part < nb) cut[1] = c->first_part;
        else if (parts == 2 && c->first_part == 0 && nb >= 32) cut[1] = (int)(((long long)nb * 15 + 16) / 32);
        if (!c->ev_fork) ARP_HIP_OK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int i = 1; i < parts; ++i) {
            if (!c->ws[i].stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->ws[i].stream, hipStreamNonBlocking));
            if (!c->ws[i].done) ARP_HIP_OK(hipEventCreateWithFlags(&c->ws[i].done, hipEventDisableTiming));
        }
        ARP_HIP_OK(hipEventRecord(c->ev_fork, stream));
        c->shared_chip = true;
        int rc = 0;
        for (int i = 0; i < parts && !rc; ++i) {
            hipStream_t st = i == 0 ? stream : c->ws[i].stream;
            if (i > 0 && hipStreamWaitEvent(st, c->ev_fork, 0) != hipSuccess) rc = fail("hipStreamWaitEvent(fork) failed");
            if (!rc) rc = forward_part(c, c->ws[i], st, img + (size_t)cut[i] * fi, cut[i + 1] - cut[i], out + (size_t)cut[i] * fo);
        }
        c->shared_chip = false;
        // join even after a failure: a part stream that entered a capture must leave it before the capture ends
        for (int i = 1; i < parts; ++i) {
            if (hipEventRecord(c->ws[i].done, c->ws[i].stream) != hipSuccess || hipStreamWaitEvent(stream, c->ws[i].done, 0) != hipSuccess)
                if (!rc) rc = fail("part stream join failed");
        }
        if (rc) return rc;
    }
    return 0;
}
int enc_geometry(arp_enc* c, int* tokens, int* width, int* img_res, int* device) {
    if (!c) return fail("null encoder");
    *tokens = c->tokens(); *width = c->cfg.width; *img_res = c->cfg.img_res; *device = c->cfg.device;
    return 0;
}

}  // namespace arp

extern "C" {

int arp_enc_create(const arp_enc_cfg* cfg, arp_enc** out) {
    if (!cfg || !out) return fail("null argument");
    const arp_enc_cfg& k = *cfg;
    if (k.mode != ARP_MODE_F32 && k.mode != ARP_MODE_BF16 && k.mode != ARP_MODE_F16 && k.mode != ARP_MODE_F16X3 && k.mode != ARP_MODE_F16C) return fail("bad mode");
    if (k.patch <= 0 || k.img_res % k.patch || k.width % k.heads || k.width % 4) return fail("bad geometry");
    const int kq = k.mode == ARP_MODE_F32 ? 32 : 64;
    if (k.width % kq || (k.patch * k.patch * 3) % kq) return fail("width and 3*patch^2 must be multiples of " + std::to_string(kq));
    int ndev = 0;
    ARP_HIP_OK(hipGetDeviceCount(&ndev));
    if (k.device < 0 || k.device >= ndev) return fail("no such HIP device");
    ARP_HIP_OK(hipSetDevice(k.device));
    ARP_TRY(prime_runtime(k.device));  // (runtime.h: one null-stream copy before the process's first stream exists)
    arp_enc* c = new arp_enc();
    c->cfg = k;
    if (c->cfg.max_frames <= 0) c->cfg.max_frames = 128;
    if (const char* e = getenv("ARP_GEMM")) c->gemm_force = atoi(e);
    if (const char* e = getenv("ARP_F16C_PLAN")) {  // four digits: in_proj, out_proj, fc1, fc2 (0 plain, 1 weight correction, 2 + activation correction)
        for (int i = 0; i < 4 && e[i]; ++i)
            if (e[i] >= '0' && e[i] <= '2') c->plan[i] = e[i] - '0';
    }
    if (const char* e = getenv("ARP_F16C_VPERM")) c->vperm = atoi(e) != 0;
    if (const char* e = getenv("ARP_ENC_STREAMS")) c->n_parts = std::max(1, std::min(atoi(e), (int)arp_enc::MAX_PARTS));
    if (const char* e = getenv("ARP_ENC_SPLIT")) c->first_part = std::max(-1, atoi(e));
    if (const char* e = getenv("ARP_ENC_MIN_PART")) c->min_part_frames = std::max(1, atoi(e));
    // fc1's plan-0 instance has no e2m1 side output, which fc2's correction K-tiles read: such a plan would run on stale operand segments
    if (c->plan[2] == 0 && c->plan[3] >= 1) {
        delete c;
        return fail("ARP_F16C_PLAN: fc2 cannot be corrected (digit 4 >= 1) when fc1 runs on the plain instance (digit 3 = 0)");
    }
    // (fc2's operand comes out of fc1's epilogue: its x4 segment from the rounded tile, its dx4 segment -- plan 2 -- straight from the accumulators)
    // (the handle's own stream serves arp_enc_forward only and is created by its first call: inside a policy step the encoder runs on the step's streams, and a
    //  process's streams share GPU_MAX_HW_QUEUES hardware queues -- arp_dt.hip)
    *out = c;
    return 0;
}

int arp_enc_destroy(arp_enc* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    c->prof.destroy();
    for (void* p : c->owned) (void)hipFree(p);
    for (auto& w : c->ws) {
        DevBuf* bufs[] = {&w.patches, &w.pe, &w.x, &w.h, &w.qkv, &w.ao, &w.fc, &w.a3};
        for (auto* b : bufs) b->release();
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.done) (void)hipEventDestroy(w.done);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    c->img_in.release(); c->out.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int arp_enc_load_weight(arp_enc* c, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!c || !name || !data || (ndim > 0 && !shape)) return fail("null argument");
    if (c->finalized) return fail("weights already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    c->staged[name] = std::move(t);
    return 0;
}

int arp_enc_finalize_weights(arp_enc* c) {
    if (!c) return fail("null handle");
    if (c->finalized) return 0;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const arp_enc_cfg& k = c->cfg;
    const int D = k.width, KP = k.patch * k.patch * 3, H = k.mlp_ratio * D, G = k.img_res / k.patch;
    const HostTensor* t;
    ARP_TRY(staged(c, "image_embedding/kernel", {KP, D}, &t)); ARP_TRY(up_kernel(c, t->data.data(), KP, D, &c->w_emb));
    const bool f16c = k.mode == ARP_MODE_F16C;
    if (f16c) {
        std::vector<float> tr((size_t)KP * D);
        for (int i = 0; i < KP; ++i)
            for (int o = 0; o < D; ++o) tr[(size_t)o * KP + i] = t->data[(size_t)i * D + o];
        ARP_TRY(up_kernel_x3(c, tr, KP, D, &c->w_emb3));
        for (int gi = 0; gi < 4; ++gi) { c->sw_d[gi].assign(k.layers, 0); c->sw_w[gi].assign(k.layers, 0); }
    }
    // the block GEMMs' weights: operand type, or ARP_MODE_F16C's [W_hi | dW4 (| W4)] rows (e2m1 segments) with their per-tensor scales
    // f16c with vperm: output column o of in_proj's V third (o = 2 D + head * 64 + d') carries the ORIGINAL column 2 D + head * 64 + pi(d'), pi = the swap of
    // bits [5:4] and [3:2] (an involution): the MFMA attention then holds sixteen consecutive original columns per lane (attention.h, outc == 2)
    const bool vp = f16c && c->vperm && D / k.heads == 64 && c->tokens() > 64;  // (the attention instances of <= 64 tokens keep round 5's stores: attention.h)
    c->vperm = vp;
    auto vperm_col = [&](int o) { return o < 2 * D ? o : (o & ~63) | ((o >> 2) & 3) << 4 | ((o >> 4) & 3) << 2 | (o & 3); };
    std::vector<float> permuted;
    auto up_w = [&](const HostTensor* ht, int in, int out_, int gi, int layer, void** dst) -> int {
        const float* src = ht->data.data();
        if (vp && gi == 0) {
            permuted.resize((size_t)in * out_);
            for (int i = 0; i < in; ++i)
                for (int o = 0; o < out_; ++o) permuted[(size_t)i * out_ + o] = src[(size_t)i * out_ + vperm_col(o)];
            src = permuted.data();
        }
        if (f16c) return up_kernel_c(c, src, in, out_, c->plan[gi], dst, &c->sw_d[gi][layer], &c->sw_w[gi][layer]);
        return up_kernel(c, src, in, out_, dst);
    };
    ARP_TRY(staged(c, "image_embedding/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &c->b_emb));
    ARP_TRY(staged(c, "cls_token", {1, 1, D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &c->cls));
    ARP_TRY(staged(c, "encoder_image_type_embedding", {1, 1, D}, &t));
    {   // get_2d_sincos_pos_embed (m3ae/model.py:95-136, "w goes first") + the image type embedding
        std::vector<float> pos((size_t)G * G * D);
        const int q = D / 4;  // embed_dim/2 per axis, half sin half cos
        for (int i = 0; i < G; ++i)
            for (int j = 0; j < G; ++j) {
                float* row = pos.data() + ((size_t)i * G + j) * D;
                for (int half = 0; half < 2; ++half) {
                    const double p = half == 0 ? (double)j : (double)i;  // first half encodes the w coordinate
                    for (int d = 0; d < q; ++d) {
                        const double omega = 1.0 / std::pow(10000.0, (double)d / (double)q);
                        row[half * 2 * q + d] = (float)std::sin(p * omega);
                        row[half * 2 * q + q + d] = (float)std::cos(p * omega);
                    }
                }
                for (int d = 0; d < D; ++d) row[d] += t->data[d];
            }
        ARP_TRY(up_f32(c, pos.data(), pos.size(), &c->pos));
    }
    c->tower.width = D; c->tower.layers = k.layers; c->tower.heads = k.heads;
    c->tower.L.resize(k.layers);
    for (int i = 0; i < k.layers; ++i) {
        const std::string p = "encoder/Block_" + std::to_string(i) + "/";
        LayerW& L = c->tower.L[i];
        ARP_TRY(staged(c, p + "LayerNorm_0/scale", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.ln1_w));
        ARP_TRY(staged(c, p + "LayerNorm_0/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.ln1_b));
        ARP_TRY(staged(c, p + "LayerNorm_1/scale", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.ln2_w));
        ARP_TRY(staged(c, p + "LayerNorm_1/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.ln2_b));
        ARP_TRY(staged(c, p + "Attention_0/Dense_0/kernel", {D, 3 * D}, &t)); ARP_TRY(up_w(t, D, 3 * D, 0, i, &L.w_in));
        ARP_TRY(staged(c, p + "Attention_0/Dense_0/bias", {3 * D}, &t));
        if (vp) {
            std::vector<float> pb(3 * D);
            for (int o = 0; o < 3 * D; ++o) pb[o] = t->data[vperm_col(o)];
            ARP_TRY(up_f32(c, pb.data(), 3 * D, &L.b_in));
        } else {
            ARP_TRY(up_f32(c, t->data.data(), 3 * D, &L.b_in));
        }
        ARP_TRY(staged(c, p + "Attention_0/Dense_1/kernel", {D, D}, &t)); ARP_TRY(up_w(t, D, D, 1, i, &L.w_out));
        ARP_TRY(staged(c, p + "Attention_0/Dense_1/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.b_out));
        ARP_TRY(staged(c, p + "TransformerMLP_0/fc1/kernel", {D, H}, &t)); ARP_TRY(up_w(t, D, H, 2, i, &L.w_fc));
        ARP_TRY(staged(c, p + "TransformerMLP_0/fc1/bias", {H}, &t)); ARP_TRY(up_f32(c, t->data.data(), H, &L.b_fc));
        ARP_TRY(staged(c, p + "TransformerMLP_0/fc2/kernel", {H, D}, &t)); ARP_TRY(up_w(t, H, D, 3, i, &L.w_proj));
        ARP_TRY(staged(c, p + "TransformerMLP_0/fc2/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &L.b_proj));
    }
    ARP_TRY(staged(c, "encoder/LayerNorm_0/scale", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &c->lnf_w));
    ARP_TRY(staged(c, "encoder/LayerNorm_0/bias", {D}, &t)); ARP_TRY(up_f32(c, t->data.data(), D, &c->lnf_b));
    c->staged.clear();
    c->finalized = true;
    return 0;
}

int arp_enc_forward(arp_enc* c, const float* images, int n, float* out) {
    if (!c || !c->finalized) return fail("encoder weights not finalized");
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    if (!images || !out) return fail("null buffer");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const size_t fi = (size_t)c->cfg.img_res * c->cfg.img_res * 3, fo = (size_t)c->tokens() * c->cfg.width;
    const int mb = c->cfg.max_frames;
    if (!c->stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ARP_TRY(c->img_in.ensure((size_t)std::min(n, mb) * fi * 4));
    ARP_TRY(c->out.ensure((size_t)std::min(n, mb) * fo * 4));
    for (int off = 0; off < n; off += mb) {
        const int nb = std::min(mb, n - off);
        ARP_HIP_OK(hipMemcpyAsync(c->img_in.p, images + off * fi, nb * fi * 4, hipMemcpyHostToDevice, c->stream));
        ARP_TRY(enc_forward_on(c, c->stream, c->img_in.as<float>(), nb, c->out.as<float>()));
        ARP_HIP_OK(hipMemcpyAsync(out + off * fo, c->out.p, nb * fo * 4, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// Part streams of a call (see arp_enc::Ws).  n_streams 1 = everything on the caller's stream (rounds 1-5); first_part_frames > 0: that many frames in part 0 of two, 0: the default cut (15/32), < 0: equal parts;
// min_part_frames <= 0 keeps the default (a part of fewer frames runs with fewer parts).  Takes effect from the next call.
int arp_enc_set_streams(arp_enc* c, int n_streams, int first_part_frames, int min_part_frames) {
    if (!c || n_streams < 1 || n_streams > arp_enc::MAX_PARTS) return fail("n_streams must be 1..4");
    c->n_parts = n_streams;
    c->first_part = first_part_frames;
    if (min_part_frames > 0) c->min_part_frames = min_part_frames;
    return 0;
}

int arp_enc_profile_enable(arp_enc* c, int on) {
    if (!c) return fail("null handle");
    c->prof.on = on != 0;
    return 0;
}
int arp_enc_profile_json(arp_enc* c, char* buf, int buf_len) {
    if (!c || !buf) return fail("null argument");
    const std::string s = c->prof.json();
    if ((int)s.size() + 1 > buf_len) return fail("profile buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// Test hook for the MIXC instances of gemm256_nt_kernel (ARP_MODE_F16C's products): A [M, K] and W [N, K] f32; the operand rows are built as the
// encoder builds them -- A: [rn16(a) | fp4(rn16(a) 2^1) | fp4((a - rn16(a)) 2^13)], W: [rn16(w) | fp4(dw 2^sd) (| fp4(w 2^sw))] for plan 1 (2) --
// and out = A.W^T + bias as f32.  sd_sw (optional, 2 ints) receives the weight scales so that a test can restate the quantisation exactly.
int arp_op_gemm_f16c(int plan, const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int* sd_sw) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K < 512 || K % 256 || N % 8 || plan < 0 || plan > 2) return fail("bad argument (K % 256, K >= 512, N % 8, plan 0..2)");
    DevBuf dA, dW, dB, dO;
    auto body = [&]() -> int {
        std::vector<uint8_t> ha((size_t)M * 3 * K, 0), hw;
        constexpr float sx = (float)(1 << F16C_X_SHIFT), sdx = (float)(1 << F16C_DX_SHIFT);
        for (int m = 0; m < M; ++m) {
            uint8_t* row = ha.data() + (size_t)m * 3 * K;
            for (int k = 0; k < K; ++k) {
                const float a = A[(size_t)m * K + k];
                const f16_t h = host_f2h(a);
                const float hf = (float)__builtin_bit_cast(_Float16, h.b);
                memcpy(row + 2 * (size_t)k, &h.b, 2);
                const int sh = (k & 1) * 4;
                row[2 * (size_t)K + k / 2] |= (uint8_t)(host_f2fp4(hf * sx) << sh);
                row[2 * (size_t)K + K / 2 + k / 2] |= (uint8_t)(host_f2fp4((a - hf) * sdx) << sh);
            }
        }
        int sd = 0, sw = 0;
        pack_weight_c([&](int i, int o) { return W[(size_t)o * K + i]; }, K, N, plan, hw, &sd, &sw);
        if (sd_sw) { sd_sw[0] = sd; sd_sw[1] = sw; }
        ARP_TRY(dA.ensure(ha.size() + 512)); ARP_TRY(dW.ensure(hw.size() + 512)); ARP_TRY(dO.ensure((size_t)M * N * 4));
        ARP_HIP_OK(hipMemcpy(dA.p, ha.data(), ha.size(), hipMemcpyHostToDevice));
        ARP_HIP_OK(hipMemcpy(dW.p, hw.data(), hw.size(), hipMemcpyHostToDevice));
        if (bias) {
            ARP_TRY(dB.ensure((size_t)N * 4));
            ARP_HIP_OK(hipMemcpy(dB.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        }
        GemmArgs g;
        g.A = dA.p; g.W = dW.p; g.bias = bias ? dB.as<float>() : nullptr; g.out = dO.p;
        g.M = M; g.N = N; g.lda = K + K / 2; g.ldw = K + plan * K / 4; g.ldr = N; g.ldo = N;
        g.mix_nk16 = K / 64; g.mix_nkc_a = plan >= 1 ? K / 256 : 0; g.K = K + plan * K / 4;
        g.mix_sa = F16C_X_SHIFT + sd; g.mix_sb = F16C_DX_SHIFT + sw;
        if (plan == 0) ARP_TRY((launch_gemm256_nt<f16_t, float, ACT_NONE, false, 8 + SITE_OP>(g, nullptr)));
        else ARP_TRY((launch_gemm256_nt<f16_t, float, ACT_NONE, false, 8 + SITE_OP, false, 1, true>(g, nullptr)));
        ARP_HIP_OK(hipDeviceSynchronize());
        ARP_HIP_OK(hipMemcpy(out, dO.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    dA.release(); dW.release(); dB.release(); dO.release();
    return rc;
}

}  // extern "C"

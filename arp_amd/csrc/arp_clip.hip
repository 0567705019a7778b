// Path (1): CLIP reward labelling on MI355X -- host orchestration + C ABI.
// Reference seam: compute_reward, /root/reference/arp_dt/label_reward.py:132-146.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/arp_hip.h"
#include "attention.h"
#include "common.h"
#include "gemm.h"
#include "gemm256.h"
#include "tower.h"
#include "preprocess.h"
#include "preprocess_host.h"
#include "rowops.h"
#include "runtime.h"

namespace arp {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

// ---- model ---------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

}  // namespace arp

using namespace arp;

// Experiment (round 4, OFF by default): the part streams of a batch on DISJOINT halves of the chip (hipExtStreamCreateWithCUMask) instead of sharing all
// 256 CUs -- ARP_CLIP_CUMASK=xcd: stream parity p gets the CUs whose index mod 8 is in [4 p, 4 p + 4) (whole XCDs, if the mask's bit order is the
// round-robin over XCDs it is documented to be); =half: the low / high 128 mask bits.  Measured in profiles/r4_cumask.txt.
static hipError_t create_part_stream(hipStream_t* st, int index) {
    static const char* mode = getenv("ARP_CLIP_CUMASK");
    if (!mode || !*mode || !strcmp(mode, "0")) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    uint32_t mask[8];
    const int p = index & 1;
    for (int w = 0; w < 8; ++w) {
        if (!strcmp(mode, "xcd")) mask[w] = p ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
        else mask[w] = ((w < 4) == (p == 0)) ? 0xFFFFFFFFu : 0u;
    }
    return hipExtStreamCreateWithCUMask(st, 8, mask);
}

struct arp_clip {
    arp_clip_cfg cfg;
    hipStream_t stream = nullptr;
    std::map<std::string, HostTensor> staged;
    bool finalized = false;
    std::vector<void*> owned;  // every device allocation holding weights

    TowerW vis, txt;
    void* conv_w = nullptr;  // T [D, 3PP]
    float *cls = nullptr, *pos = nullptr, *lnpre_w = nullptr, *lnpre_b = nullptr, *lnpost_w = nullptr, *lnpost_b = nullptr;
    void* proj_t = nullptr;  // T [E, D]
    float *tok_emb = nullptr, *tpos = nullptr, *lnf_w = nullptr, *lnf_b = nullptr;
    void* tproj_t = nullptr;  // T [E, Tw]
    // f32 copy of the text tower (16-bit modes only) for the CACHED prompt features of arp_clip_set_text: the prompt vector multiplies
    // every frame's feature, so its rounding error is common to all rewards -- measured on ViT-B/32, 128 frames: f16 cosine error
    // max 7.4e-5 / rms 3.9e-5 with a 16-bit text tower, 4.9e-5 / 2.2e-5 with this one (bf16: 6.1e-4 / 3.4e-4 -> 2.8e-4 / 1.1e-4).
    // It runs once per prompt set (6 GFLOP); the multi-scale text path of the fine-tune step keeps the 16-bit weights.
    TowerW txt32;
    void* tproj_t32 = nullptr;
    float logit_scale = 0.f;
    float* lut = nullptr;

    DevBuf txt_feat;
    int n_prompts = 0;
    DevBuf txt_mean;        // mean of the cached (normalised) prompt vectors: the rollout loop's `isinstance(pos_text, list)` branch
    int prompt_reduce = 0;  // 0 = prompt 0 (label_reward.py:146, quirk Q1), 1 = mean over the prompts (envs/vl_reward.py:19-22)
    DevBuf ms_keep;         // multi-scale export buffer kept between calls of the online adapter reward (a hipMalloc per call would cost more than the head)

    // workspace for `ws_frames` frames
    int ws_frames = 0;
    DevBuf patches, pe, x, h, qkv, ao, fc, cls_h, feat, frames_in, rewards;
    std::map<long long, ResizePlan*> plans;
    Profiler prof;
    // second stream: a shallow clone (shared weights, own workspace/stream/profiler) that labels the other half of a
    // batch concurrently, so one half's memory-bound kernels and GEMM tails overlap the other half's GEMMs
    // multi-scale export target of the NEXT tower run (row N2); cleared after use
    float* ms_out = nullptr;
    int ms_ld = 0;
    const int* ms_rows = nullptr;
    int pre_bilinear = 0;  // next forward_chunk uses the fine-tune transform (bilinear) instead of the PIL-bicubic one
    std::vector<arp_clip*> siblings;  // n_streams - 1 clones
    bool is_sibling = false;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;  // one per sibling
    // host-fed labelling (the S2 seam): every part's frames go up on ONE copy stream, back to back at the full PCIe rate, and the part's
    // compute stream waits for its own slice only.  From pinned / registered host memory (arp_host_register) the copies are true DMA and
    // all of them are in flight before the first kernel; from pageable memory the runtime stages them and the call order does the overlap.
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> ev_copy;  // one per part
    // arp_clip_label_submit / _collect: two host-fed calls in flight.  A synchronous arp_clip_label pays the pipeline's fill and drain on
    // every call (13.3 ms per 1024 frames where back-to-back passes take 10.5 ms and the 3.5 ms upload overlaps them completely:
    // scripts/seam_probe.py); with the next call's upload and kernels queued behind the running one the streams never drain.
    struct LabelSlot {
        DevBuf frames, rewards;
        float* host = nullptr;  // pinned staging of the rewards (a D2H into pageable memory would block the submitting thread)
        size_t host_n = 0;
        hipEvent_t done = nullptr;
        int n = 0;
        bool busy = false;
    } lslot[2];
    int gemm_force = 0;  // 0 auto, 1 force the 128x128 kernel, 2 force the 256x256 kernel (ARP_GEMM env)
    // Fold LayerNorm into the consumer GEMMs of the vision tower in bf16 mode (ARP_LN_FOLD=1).  Numerically fine
    // (cosine error 3.6e-4 vs 4.8e-4 unfused) but MEASURED SLOWER on MI355X (74.2 k vs 79.7 k frames/s): the extra
    // per-row loads and reductions land in the GEMM epilogues, which are the serialised part of every tile,
    // while the LayerNorm kernels they replace overlap with the other stream's GEMMs.  Off by default.
    bool ln_fold = false;
    bool fp8_mlp = false;       // vision tower MLP GEMMs on the scaled fp8 MFMA (arp_clip_set_fp8_mlp before finalize; tower.h)
    bool fp8_attn = false;      // ... and in_proj / out_proj as well (arp_clip_set_fp8_mlp(c, 2))
    DevBuf clock_buf;           // arp_clip_clock_probe: three u64 the clock-diagnostic c_fc instance accumulates into (owned by the primary handle)
    unsigned long long* clock_acc = nullptr;  // null: the ordinary instances run
    bool shared_chip = false;   // this handle's kernels run beside another part stream's (label_dev with two or more parts): tower.h picks out_proj's kernel by it
    bool qkv_fused = true;      // QKV projection + attention in one kernel (qkvattn.h) where the geometry allows; ARP_QKV_FUSED=0 disables
    bool cls_only_last = true;  // vision tower: the last block computes only what ln_post reads (tower.h); ARP_CLS_ONLY=0 disables
    DevBuf stats;
    // latency path (get_torch_clip_reward): split-K slabs of the skinny GEMMs (tower.h), and the whole pass over
    // <= SKINNY_MAX_M token rows -- preprocess .. reward, ~80 launches -- replayed as ONE hipGraph per (buffers, geometry): the host then
    // pays one graph launch instead of 80 kernel launches that each cost more host time than the kernel runs
    DevBuf part;
    DevBuf lat_stats;       // folded LayerNorm on the latency path: [rows][width / 16][2] strip sums (tower.h)
    bool lat_fold = true;   // ARP_LAT_FOLD=0: reduce + LayerNorm kernels instead (seven launches per block instead of five)
    bool lat_f0 = false;    // the pass being enqueued: the token-assembly kernel wrote the operand copy + statistics of block 0
    bool skinny = true;     // ARP_SKINNY=0: the output-tiled GEMMs at every size
    bool lat_h0 = false;    // ... and its token-assembly kernel already wrote ln_1 of the first block into h
    int lat_rows = 1024;    // a pass of at most this many token rows takes the latency path (ARP_SKINNY_ROWS; = the kernel's cap).  Against the
                            // throughput kernels (profiles/r3_latency_rows.txt): +63 % at 50 rows, +51 % at 300, +29 % at 600, +5..12 % at 1000
    bool lat_now = false;   // the pass being enqueued has at most SKINNY_MAX_M token rows (set by forward_chunk)
    bool lat_graph = true;  // ARP_CLIP_GRAPH=0: launch by launch
    struct LatGraph {
        const uint8_t* frames;
        float* rewards;
        int n, H, W, crop;  // crop: bit 0 = use_crop, bit 1 = the prompt reduction the captured reward kernel reads (arp_clip_set_prompt_reduce)
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<LatGraph> lat_graphs;
    // host-fed single-frame calls: the frame is copied into pinned memory by the CPU and the preprocess kernel reads it over PCIe;
    // the reward kernel writes into pinned memory -- no copy operations on the stream at all (ARP_CLIP_PINNED=0: hipMemcpyAsync both ways)
    uint8_t* pin_frames = nullptr;
    size_t pin_frames_bytes = 0;
    float* pin_rewards = nullptr;
    size_t pin_rewards_n = 0;
    bool lat_pinned = true;

    int ntok() const { return (cfg.img_res / cfg.patch) * (cfg.img_res / cfg.patch) + 1; }
    size_t esz() const { return cfg.mode == ARP_MODE_F32 ? 4 : 2; }
};

namespace arp {

static TowerCtx ctx_of(arp_clip* c) {
    TowerCtx t;
    t.stream = c->stream;
    t.prof = &c->prof;
    t.attn_impl = c->cfg.attn_impl;
    t.gemm_force = c->gemm_force;
    t.qkv_fused = c->qkv_fused;
    t.shared_chip = c->shared_chip;
    t.clock_acc = c->clock_acc;
    t.fp8_mlp = c->fp8_mlp; t.fp8_attn = c->fp8_attn;
    t.ms_out = c->ms_out; t.ms_ld = c->ms_ld; t.ms_rows = c->ms_rows;
    t.skinny = c->lat_now; t.h_ready0 = c->lat_now && c->lat_h0; t.lat_stats = c->lat_stats.as<float>(); t.lat_fold0 = c->lat_now && c->lat_f0; t.part = c->part.as<float>(); t.part_floats = c->part.bytes / 4;
    return t;
}

static int upload_f32(arp_clip* c, const std::vector<float>& v, float** out) {
    void* p = nullptr;
    ARP_HIP_OK(hipMalloc(&p, std::max<size_t>(v.size() * 4, 16)));
    ARP_HIP_OK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    c->owned.push_back(p);
    *out = static_cast<float*>(p);
    return 0;
}

// uploads a [rows, cols] matrix in the handle's GEMM operand type (bf16 RNE or f32); optional transpose
static int upload_mat(arp_clip* c, const float* src, int rows, int cols, bool transpose, void** out, int wmode = -1) {
    const int mode = wmode >= 0 ? wmode : c->cfg.mode;
    const size_t esz = mode == ARP_MODE_F32 ? 4 : 2;
    const size_t n = (size_t)rows * cols;
    std::vector<float> tmp;
    const float* s = src;
    if (transpose) {  // src is [rows, cols]; result is [cols, rows]
        tmp.resize(n);
        for (int r = 0; r < rows; ++r)
            for (int q = 0; q < cols; ++q) tmp[(size_t)q * rows + r] = src[(size_t)r * cols + q];
        s = tmp.data();
    }
    void* p = nullptr;
    ARP_HIP_OK(hipMalloc(&p, std::max<size_t>(n * esz, 16)));
    if (mode == ARP_MODE_BF16) {
        std::vector<bf16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2bf(s[i]);
        ARP_HIP_OK(hipMemcpy(p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else if (mode == ARP_MODE_F16) {
        std::vector<f16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2h(s[i]);
        ARP_HIP_OK(hipMemcpy(p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        ARP_HIP_OK(hipMemcpy(p, s, n * 4, hipMemcpyHostToDevice));
    }
    c->owned.push_back(p);
    *out = p;
    return 0;
}

static int get_staged(arp_clip* c, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
    auto it = c->staged.find(name);
    if (it == c->staged.end()) return fail("missing weight: " + name);
    if (it->second.shape != shape) {
        std::string s = "weight " + name + " has shape [";
        for (auto d : it->second.shape) s += std::to_string(d) + ",";
        s += "], expected [";
        for (auto d : shape) s += std::to_string(d) + ",";
        return fail(s + "]");
    }
    *out = &it->second;
    return 0;
}

static int load_tower(arp_clip* c, const std::string& prefix, int d, int layers, int heads, TowerW& tw, bool fold, int wmode = -1, int ntok = 0) {
    tw.width = d; tw.layers = layers; tw.heads = heads;
    const int emode = wmode >= 0 ? wmode : c->cfg.mode;
    const bool hm = c->qkv_fused && ntok > 0 && qkv_attn_supported(ntok, d, heads, emode == ARP_MODE_F32 ? 4 : 2);
    tw.L.resize(layers);
    for (int i = 0; i < layers; ++i) {
        const std::string p = prefix + "resblocks." + std::to_string(i) + ".";
        LayerW& L = tw.L[i];
        const HostTensor* t;
        ARP_TRY(get_staged(c, p + "ln_1.weight", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.ln1_w));
        ARP_TRY(get_staged(c, p + "ln_1.bias", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.ln1_b));
        ARP_TRY(get_staged(c, p + "ln_2.weight", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.ln2_w));
        ARP_TRY(get_staged(c, p + "ln_2.bias", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.ln2_b));
        ARP_TRY(get_staged(c, p + "attn.in_proj_weight", {3 * d, d}, &t)); ARP_TRY(upload_mat(c, t->data.data(), 3 * d, d, false, &L.w_in, wmode));
        ARP_TRY(get_staged(c, p + "attn.in_proj_bias", {3 * d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.b_in));
        if (hm) {  // head-major copies for the fused QKV + attention kernel (qkvattn.h): row h*192 + j = q | k | v row of head h
            const HostTensor *w, *b;
            ARP_TRY(get_staged(c, p + "attn.in_proj_weight", {3 * d, d}, &w)); ARP_TRY(get_staged(c, p + "attn.in_proj_bias", {3 * d}, &b));
            std::vector<float> wp((size_t)3 * d * d), bp((size_t)3 * d);
            for (int hh = 0; hh < heads; ++hh)
                for (int j = 0; j < 192; ++j) {
                    const int src = (j >> 6) * d + hh * 64 + (j & 63);
                    memcpy(&wp[(size_t)(hh * 192 + j) * d], &w->data[(size_t)src * d], (size_t)d * 4);
                    bp[hh * 192 + j] = b->data[src];
                }
            ARP_TRY(upload_mat(c, wp.data(), 3 * d, d, false, &L.w_in_hm, wmode)); ARP_TRY(upload_f32(c, bp, &L.b_in_hm));
        }
        ARP_TRY(get_staged(c, p + "attn.out_proj.weight", {d, d}, &t)); ARP_TRY(upload_mat(c, t->data.data(), d, d, false, &L.w_out, wmode));
        ARP_TRY(get_staged(c, p + "attn.out_proj.bias", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.b_out));
        ARP_TRY(get_staged(c, p + "mlp.c_fc.weight", {4 * d, d}, &t)); ARP_TRY(upload_mat(c, t->data.data(), 4 * d, d, false, &L.w_fc, wmode));
        ARP_TRY(get_staged(c, p + "mlp.c_fc.bias", {4 * d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.b_fc));
        ARP_TRY(get_staged(c, p + "mlp.c_proj.weight", {d, 4 * d}, &t)); ARP_TRY(upload_mat(c, t->data.data(), d, 4 * d, false, &L.w_proj, wmode));
        ARP_TRY(get_staged(c, p + "mlp.c_proj.bias", {d}, &t)); ARP_TRY(upload_f32(c, t->data, &L.b_proj));
        if (c->fp8_mlp && ntok > 0 && emode != ARP_MODE_F32 && d % 128 == 0) {  // vision tower only (ntok is passed for it alone)
            auto put8 = [&](const std::vector<float>& w, void** out, float* scale) -> int {
                float mx = 0.f;
                for (float v : w) mx = std::max(mx, fabsf(v));
                const float sc = mx > 0.f ? exp2f(floorf(log2f(240.f / mx))) : 1.f;  // power of two: the largest weight lands in [120, 240]
                std::vector<fp8_t> q(w.size());
                for (size_t i = 0; i < w.size(); ++i) q[i] = host_f2fp8(w[i] * sc);
                void* dp = nullptr;
                ARP_HIP_OK(hipMalloc(&dp, q.size()));
                ARP_HIP_OK(hipMemcpy(dp, q.data(), q.size(), hipMemcpyHostToDevice));
                c->owned.push_back(dp);
                *out = dp;
                *scale = sc;
                return 0;
            };
            const HostTensor *w1, *w2, *lw, *lb;
            float s1 = 1.f, s2 = 1.f;
            ARP_TRY(get_staged(c, p + "mlp.c_fc.weight", {4 * d, d}, &w1)); ARP_TRY(put8(w1->data, &L.w_fc8, &s1));
            ARP_TRY(get_staged(c, p + "mlp.c_proj.weight", {d, 4 * d}, &w2)); ARP_TRY(put8(w2->data, &L.w_proj8, &s2));
            L.a_fc = 1.0f / (FP8_S_H * s1);
            L.a_proj = 1.0f / (FP8_S_G * s2);
            ARP_TRY(get_staged(c, p + "ln_2.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_2.bias", {d}, &lb));
            std::vector<float> g8(lw->data), b8(lb->data);
            for (auto& v : g8) v *= FP8_S_H;
            for (auto& v : b8) v *= FP8_S_H;
            ARP_TRY(upload_f32(c, g8, &L.ln2_w8)); ARP_TRY(upload_f32(c, b8, &L.ln2_b8));
            if (c->fp8_attn) {  // the attention's projections too (tower.h::tower_attn_fp8)
                const HostTensor *wi, *wo;
                float si = 1.f, so = 1.f;
                ARP_TRY(get_staged(c, p + "attn.in_proj_weight", {3 * d, d}, &wi)); ARP_TRY(put8(wi->data, &L.w_in8, &si));
                ARP_TRY(get_staged(c, p + "attn.out_proj.weight", {d, d}, &wo)); ARP_TRY(put8(wo->data, &L.w_out8, &so));
                L.a_in = 1.0f / (FP8_S_H * si);
                L.a_out = 1.0f / (FP8_S_A * so);
                ARP_TRY(get_staged(c, p + "ln_1.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_1.bias", {d}, &lb));
                std::vector<float> g1(lw->data), b1(lb->data);
                for (auto& v : g1) v *= FP8_S_H;
                for (auto& v : b1) v *= FP8_S_H;
                ARP_TRY(upload_f32(c, g1, &L.ln1_w8)); ARP_TRY(upload_f32(c, b1, &L.ln1_b8));
            }
        }
        if (fold) {  // LayerNorm folded into in_proj (ln_1) and c_fc (ln_2): tower.h fold_layernorm
            const HostTensor *w, *b, *lw, *lb;
            std::vector<bf16_t> wf;
            std::vector<float> cc, dd;
            auto put = [&](const std::vector<bf16_t>& m, void** out) -> int {
                void* dp = nullptr;
                ARP_HIP_OK(hipMalloc(&dp, m.size() * 2));
                ARP_HIP_OK(hipMemcpy(dp, m.data(), m.size() * 2, hipMemcpyHostToDevice));
                c->owned.push_back(dp);
                *out = dp;
                return 0;
            };
            ARP_TRY(get_staged(c, p + "attn.in_proj_weight", {3 * d, d}, &w)); ARP_TRY(get_staged(c, p + "attn.in_proj_bias", {3 * d}, &b));
            ARP_TRY(get_staged(c, p + "ln_1.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_1.bias", {d}, &lb));
            fold_layernorm(w->data.data(), lw->data.data(), lb->data.data(), b->data.data(), 3 * d, d, wf, cc, dd, emode == ARP_MODE_F16);
            ARP_TRY(put(wf, &L.w_in_f)); ARP_TRY(upload_f32(c, cc, &L.c_in)); ARP_TRY(upload_f32(c, dd, &L.d_in));
            ARP_TRY(get_staged(c, p + "mlp.c_fc.weight", {4 * d, d}, &w)); ARP_TRY(get_staged(c, p + "mlp.c_fc.bias", {4 * d}, &b));
            ARP_TRY(get_staged(c, p + "ln_2.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_2.bias", {d}, &lb));
            fold_layernorm(w->data.data(), lw->data.data(), lb->data.data(), b->data.data(), 4 * d, d, wf, cc, dd, emode == ARP_MODE_F16);
            ARP_TRY(put(wf, &L.w_fc_f)); ARP_TRY(upload_f32(c, cc, &L.c_fc)); ARP_TRY(upload_f32(c, dd, &L.d_fc));
        }
    }
    tw.folded = fold;
    // the latency path's fold (tower.h run_blocks): the same operands in the handle's own 16-bit type, vision tower only
    if (!fold && ntok > 0 && c->lat_fold && c->skinny && emode != ARP_MODE_F32 && (d & 15) == 0) {
        for (int i = 0; i < layers; ++i) {
            LayerW& L = tw.L[i];
            const std::string p = prefix + "resblocks." + std::to_string(i) + ".";
            const HostTensor *w, *b, *lw, *lb;
            std::vector<bf16_t> wf;
            std::vector<float> cc, dd;
            auto put = [&](const std::vector<bf16_t>& m, void** out) -> int {
                void* dp = nullptr;
                ARP_HIP_OK(hipMalloc(&dp, m.size() * 2));
                ARP_HIP_OK(hipMemcpy(dp, m.data(), m.size() * 2, hipMemcpyHostToDevice));
                c->owned.push_back(dp);
                *out = dp;
                return 0;
            };
            const bool half = emode == ARP_MODE_F16;
            ARP_TRY(get_staged(c, p + "attn.in_proj_weight", {3 * d, d}, &w)); ARP_TRY(get_staged(c, p + "attn.in_proj_bias", {3 * d}, &b));
            ARP_TRY(get_staged(c, p + "ln_1.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_1.bias", {d}, &lb));
            fold_layernorm(w->data.data(), lw->data.data(), lb->data.data(), b->data.data(), 3 * d, d, wf, cc, dd, half);
            ARP_TRY(put(wf, &L.w_in_f)); ARP_TRY(upload_f32(c, cc, &L.c_in)); ARP_TRY(upload_f32(c, dd, &L.d_in));
            ARP_TRY(get_staged(c, p + "mlp.c_fc.weight", {4 * d, d}, &w)); ARP_TRY(get_staged(c, p + "mlp.c_fc.bias", {4 * d}, &b));
            ARP_TRY(get_staged(c, p + "ln_2.weight", {d}, &lw)); ARP_TRY(get_staged(c, p + "ln_2.bias", {d}, &lb));
            fold_layernorm(w->data.data(), lw->data.data(), lb->data.data(), b->data.data(), 4 * d, d, wf, cc, dd, half);
            ARP_TRY(put(wf, &L.w_fc_f)); ARP_TRY(upload_f32(c, cc, &L.c_fc)); ARP_TRY(upload_f32(c, dd, &L.d_fc));
        }
        tw.lat_folded = true;
    }
    return 0;
}

// thin adapters onto tower.h
template <typename T, typename OutT, int ACT, bool RESID, int SITE>
static int gemm(arp_clip* c, const char* site, const void* A, const void* W, const float* bias, const float* resid, void* out,
                int M, int N, int K) {
    TowerCtx t = ctx_of(c);
    return tower_gemm<T, OutT, ACT, RESID, SITE>(t, site, A, W, bias, resid, out, M, N, K);
}
template <typename OutT>
static int layernorm(arp_clip* c, const char* site, const float* in, size_t in_stride, OutT* out, int out_stride,
                     const float* w, const float* b, int rows, int D, float eps) {
    TowerCtx t = ctx_of(c);
    return tower_layernorm<OutT>(t, site, in, in_stride, out, out_stride, w, b, rows, D, eps);
}
template <typename T>
static int run_blocks(arp_clip* c, const TowerW& tw, const char* tag, float* x, T* h, T* qkv, T* ao, T* fc, int B, int N,
                      int causal, float* stats = nullptr) {
    TowerCtx t = ctx_of(c);
    t.cls_only_last = (&tw == &c->vis) && c->cls_only_last && !(t.ms_out && t.ms_rows);
    return run_blocks<T, ACT_QGELU, 0>(t, tw, tag, x, h, qkv, ao, fc, B, N, causal, 1e-5f, stats);
}


// every captured pass holds raw pointers into the workspace, the prompt features and the kernel arguments of its day
static void drop_lat_graphs(arp_clip* c) {
    for (auto& g : c->lat_graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    c->lat_graphs.clear();
}

static int ensure_workspace(arp_clip* c, int frames) {
    if (frames <= c->ws_frames) return 0;
    if (!c->lat_graphs.empty()) {
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        drop_lat_graphs(c);
    }
    const arp_clip_cfg& k = c->cfg;
    const size_t e = c->esz();
    const int G = k.img_res / k.patch, N = c->ntok(), D = k.width;
    const size_t B = frames, M = B * N;
    ARP_TRY(c->patches.ensure(B * G * G * 3 * k.patch * k.patch * e));
    ARP_TRY(c->pe.ensure(B * G * G * D * 4));
    ARP_TRY(c->x.ensure(M * D * 4));
    ARP_TRY(c->h.ensure(M * D * e));
    ARP_TRY(c->qkv.ensure(M * 3 * D * e));
    ARP_TRY(c->ao.ensure(M * D * e));
    ARP_TRY(c->fc.ensure(M * 4 * D * e));
    ARP_TRY(c->cls_h.ensure(B * D * e));
    ARP_TRY(c->feat.ensure(B * k.embed * 4));
    ARP_TRY(c->stats.ensure(M * (size_t)std::max(D >> 7, 1) * 8));
    c->ws_frames = frames;
    return 0;
}

static int get_plan(arp_clip* c, int H, int W, int use_crop, ResizePlan** out, bool small = false) {
    const long long key = ((long long)H << 32) | ((long long)W << 2) | (small ? 2 : 0) | (use_crop ? 1 : 0);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) {
        *out = it->second;
        return 0;
    }
    ResizePlan* p = new ResizePlan();
    const int r = build_plan(H, W, use_crop, c->cfg.img_res, *p, small ? (getenv("ARP_PRE_TR_SMALL") ? atoi(getenv("ARP_PRE_TR_SMALL")) : 8) : 32);
    if (r != 0) {
        delete p;
        return r;
    }
    c->plans[key] = p;
    *out = p;
    return 0;
}

// frames (device) -> un-normalised image features in c->feat [nb, embed]
template <typename T>
static int forward_chunk(arp_clip* c, const uint8_t* frames_dev, int nb, ResizePlan* plan) {
    const arp_clip_cfg& k = c->cfg;
    const int G = k.img_res / k.patch, N = c->ntok(), D = k.width, KP = 3 * k.patch * k.patch;
    c->lat_now = c->skinny && sizeof(T) == 2 && (long)nb * N <= c->lat_rows;
    struct LatGuard {
        arp_clip* c;
        ~LatGuard() { c->lat_now = false; c->lat_h0 = false; c->lat_f0 = false; }
    } lat_guard{c};
    if (c->pre_bilinear) {
        ProfScope ps(c->prof, c->stream, "preprocess_bilinear");
        const int H = c->pre_bilinear >> 16, W = c->pre_bilinear & 0xffff, R = k.img_res;
        const size_t total = (size_t)nb * 3 * R * (R / 4);
        hipLaunchKernelGGL((preprocess_bilinear_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, frames_dev, c->patches.as<T>(), nb,
                           H, W, R, k.patch, (H != R && W != R) ? 1 : 0);
        ARP_HIP_OK(hipGetLastError());
    } else {
        ProfScope ps(c->prof, c->stream, "preprocess");
        ARP_TRY((launch_preprocess<T, PRE_PATCH>(*plan, frames_dev, nb, k.patch, c->lut, c->patches.p, c->stream)));
    }
    c->lat_h0 = false;
    c->lat_f0 = false;
    bool assembled = false;
    if constexpr (sizeof(T) == 2) {
        // latency path: the patch embedding as split-K slabs, summed by the token-assembly kernel, which also applies the first ln_1
        const int Mp = nb * G * G, S = c->lat_now ? tower_split_of(Mp, D, KP) : 0;
        if (S && (size_t)S * Mp * D * 4 <= c->part.bytes && c->vis.layers > 0 && D % 4 == 0 && D <= ROW_MAX_V4 * 256) {
            const int tcode = __is_same(T, bf16_t) ? 1 : 2;
            SkinnyArgs q;
            q.A = c->patches.p; q.W = c->conv_w; q.out = c->part.p; q.M = Mp; q.N = D; q.K = KP; q.lda = KP; q.ldw = KP; q.ldo = D; q.out_f32 = 1;
            q.ksplit = S; q.slice_stride = (size_t)Mp * D;
            {
                ProfScope ps(c->prof, c->stream, "vit.patch_embed");
                ARP_TRY(launch_skinny_gemm(tcode, q, c->stream));
            }
            ProfScope ps(c->prof, c->stream, "vit.assemble_ln_pre_ln_1");
            const bool f0 = c->vis.lat_folded && c->lat_stats.p && (D & 15) == 0;
#define ARP_ASML_CALL(NV)                                                                                                                \
    hipLaunchKernelGGL((vit_assemble_lat_kernel<T, NV>), dim3(nb * N), dim3(64), 0, c->stream, c->part.as<float>(), S, (size_t)Mp * D, c->cls, c->pos, \
                       c->lnpre_w, c->lnpre_b, c->x.as<float>(), c->h.as<T>(), c->vis.L[0].ln1_w, c->vis.L[0].ln1_b, N, D, 1e-5f,         \
                       f0 ? c->lat_stats.as<float>() : nullptr, D >> 4)
            ARP_NV_DISPATCH(D, ARP_ASML_CALL);
#undef ARP_ASML_CALL
            ARP_HIP_OK(hipGetLastError());
            assembled = true;
            c->lat_h0 = !f0;
            c->lat_f0 = f0;
        }
    }
    if (!assembled) {
    ARP_TRY((gemm<T, float, ACT_NONE, false, SITE_PATCH>(c, "vit.patch_embed", c->patches.p, c->conv_w, nullptr, nullptr, c->pe.p,
                                                         nb * G * G, D, KP)));
    {
        ProfScope ps(c->prof, c->stream, "vit.assemble_ln_pre");
#define ARP_ASM_CALL(NV)                                                                                                      \
    hipLaunchKernelGGL((vit_assemble_lnpre_kernel<NV>), dim3((nb * N + 3) / 4), dim3(256), 0, c->stream, c->pe.as<float>(), \
                       c->cls, c->pos, c->lnpre_w, c->lnpre_b, c->x.as<float>(), nb* N, N, D, 1e-5f)
        ARP_NV_DISPATCH(D, ARP_ASM_CALL);
#undef ARP_ASM_CALL
        ARP_HIP_OK(hipGetLastError());
    }
    }
    ARP_TRY(run_blocks<T>(c, c->vis, "vit", c->x.as<float>(), c->h.as<T>(), c->qkv.as<T>(), c->ao.as<T>(), c->fc.as<T>(), nb, N, 0, c->stats.as<float>()));
    // ln_post on the CLS rows only, then proj (arp_dt/models/openai/layers.py:330-332)
    ARP_TRY(layernorm<T>(c, "vit.ln_post", c->x.as<float>(), (size_t)N * D, c->cls_h.as<T>(), D, c->lnpost_w, c->lnpost_b, nb, D, 1e-5f));
    ARP_TRY((gemm<T, float, ACT_NONE, false, SITE_PROJ>(c, "vit.proj", c->cls_h.p, c->proj_t, nullptr, nullptr, c->feat.p, nb, k.embed, D)));
    return 0;
}

static int forward_chunk_dispatch(arp_clip* c, const uint8_t* frames_dev, int nb, ResizePlan* plan) {
    if (c->cfg.mode == ARP_MODE_BF16) return forward_chunk<bf16_t>(c, frames_dev, nb, plan);
    if (c->cfg.mode == ARP_MODE_F16) return forward_chunk<f16_t>(c, frames_dev, nb, plan);
    return forward_chunk<float>(c, frames_dev, nb, plan);
}

// ms_host != null: multi-scale export of the EOT rows [np, layers*txt_width]; out_host != null: return the (optionally
// normalised) features instead of caching them as the prompt set
template <typename T> static int run_text(arp_clip* c, const int32_t* tokens, int np, float* out_host = nullptr, bool normalize = true,
                                          float* ms_host = nullptr, bool dev_out = false, bool w32 = false) {
    // w32: T = float on the f32 copy of the text tower of a 16-bit handle (arp_clip::txt32)
    const TowerW& txt = w32 ? c->txt32 : c->txt;
    const void* tproj = w32 ? c->tproj_t32 : c->tproj_t;
    const arp_clip_cfg& k = c->cfg;
    const int Tw = k.txt_width, ctx = k.ctx, M = np * ctx;
    const size_t e = sizeof(T);
    DevBuf tok, eot, x, h, qkv, ao, fc, hs, ms, feat_tmp;
    DevBuf& feat = out_host ? feat_tmp : c->txt_feat;
    int rc = 0;
    std::vector<int> eot_rows(np);
    for (int p = 0; p < np; ++p) {
        int best = 0;
        for (int t = 1; t < ctx; ++t)
            if (tokens[p * ctx + t] > tokens[p * ctx + best]) best = t;  // argmax, first max wins (EOT is the largest id)
        eot_rows[p] = p * ctx + best;
        for (int t = 0; t < ctx; ++t)
            if (tokens[p * ctx + t] < 0 || tokens[p * ctx + t] >= k.vocab) return fail("set_text: token id out of range");
    }
    auto body = [&]() -> int {
        ARP_TRY(tok.ensure((size_t)M * 4)); ARP_TRY(eot.ensure((size_t)np * 4));
        ARP_TRY(x.ensure((size_t)M * Tw * 4)); ARP_TRY(h.ensure((size_t)M * Tw * e)); ARP_TRY(qkv.ensure((size_t)M * 3 * Tw * e));
        ARP_TRY(ao.ensure((size_t)M * Tw * e)); ARP_TRY(fc.ensure((size_t)M * 4 * Tw * e)); ARP_TRY(hs.ensure((size_t)np * Tw * e));
        ARP_TRY(feat.ensure((size_t)np * k.embed * 4));
        if (ms_host) ARP_TRY(ms.ensure((size_t)np * c->txt.layers * Tw * 4));
        ARP_HIP_OK(hipMemcpyAsync(tok.p, tokens, (size_t)M * 4, hipMemcpyHostToDevice, c->stream));
        ARP_HIP_OK(hipMemcpyAsync(eot.p, eot_rows.data(), (size_t)np * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(text_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, c->stream, tok.as<int>(), c->tok_emb, c->tpos,
                           x.as<float>(), M, ctx, Tw);
        ARP_HIP_OK(hipGetLastError());
        if (ms_host) { c->ms_out = ms.as<float>(); c->ms_ld = c->txt.layers * Tw; c->ms_rows = eot.as<int>(); }
        const int rb = run_blocks<T>(c, txt, "text", x.as<float>(), h.as<T>(), qkv.as<T>(), ao.as<T>(), fc.as<T>(), np, ctx, 1);
        c->ms_out = nullptr; c->ms_rows = nullptr;
        ARP_TRY(rb);
        // ln_final, EOT row, text_projection (arp_dt/models/openai/layers.py:367-369)
#define ARP_LNG_CALL(NV)                                                                                                     \
    hipLaunchKernelGGL((layernorm_gather_kernel<T, NV>), dim3((np + 3) / 4), dim3(256), 0, c->stream, x.as<float>(), (size_t)Tw, \
                       eot.as<int>(), hs.as<T>(), Tw, c->lnf_w, c->lnf_b, np, Tw, 1e-5f)
        ARP_NV_DISPATCH(Tw, ARP_LNG_CALL);
#undef ARP_LNG_CALL
        ARP_HIP_OK(hipGetLastError());
        ARP_TRY((gemm<T, float, ACT_NONE, false, SITE_PROJ>(c, "text.proj", hs.p, tproj, nullptr, nullptr, feat.p, np,
                                                            k.embed, Tw)));
        if (normalize) {
            hipLaunchKernelGGL(l2_normalize_kernel, dim3((np + 3) / 4), dim3(256), 0, c->stream, feat.as<float>(), np, k.embed);
            ARP_HIP_OK(hipGetLastError());
        }
        const hipMemcpyKind okind = dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (out_host) ARP_HIP_OK(hipMemcpyAsync(out_host, feat.p, (size_t)np * k.embed * 4, okind, c->stream));
        if (ms_host) ARP_HIP_OK(hipMemcpyAsync(ms_host, ms.p, (size_t)np * c->txt.layers * Tw * 4, okind, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        return 0;
    };
    rc = body();
    tok.release(); eot.release(); x.release(); h.release(); qkv.release(); ao.release(); fc.release(); hs.release(); ms.release(); feat_tmp.release();
    if (rc == 0 && !out_host) c->n_prompts = np;
    return rc;
}

static int check_ready(arp_clip* c, bool need_text) {
    if (!c) return fail("null handle");
    if (!c->finalized) return fail("weights not finalized: call arp_clip_finalize_weights first");
    if (need_text && c->n_prompts <= 0) return fail("no prompt set: call arp_clip_set_text first");
    return 0;
}

static int label_dev_single(arp_clip* c, const uint8_t* frames_dev, int n, int H, int W, int use_crop, float* rewards_dev) {
    ARP_TRY(check_ready(c, true));
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ResizePlan* plan;
    const int mb = c->cfg.max_batch;
    const bool small = !c->is_sibling && n <= mb && (long)n * c->ntok() <= c->lat_rows;  // the rollout loop's call (one frame, or a few)
    ARP_TRY(get_plan(c, H, W, use_crop, &plan, small));
    ARP_TRY(ensure_workspace(c, std::min(n, mb)));
    const float scale = expf(c->logit_scale);
    auto pass = [&](const uint8_t* fr, int nb, float* rw) -> int {
        ARP_TRY(forward_chunk_dispatch(c, fr, nb, plan));
        ProfScope ps(c->prof, c->stream, "reward");
        // mean over prompts of scale <img_n, txt_p> = scale <img_n, mean_p txt_p>: the same kernel on the mean prompt vector
        hipLaunchKernelGGL(reward_kernel, dim3((nb + 3) / 4), dim3(256), 0, c->stream, c->feat.as<float>(),
                           c->prompt_reduce ? c->txt_mean.as<float>() : c->txt_feat.as<float>(), scale, rw, nb, c->cfg.embed);
        ARP_HIP_OK(hipGetLastError());
        return 0;
    };
    // replay the captured pass
    if (c->lat_graph && small && !c->prof.on && !c->ms_out && !c->pre_bilinear) {
        for (auto& g : c->lat_graphs)
            if (g.frames == frames_dev && g.rewards == rewards_dev && g.n == n && g.H == H && g.W == W && g.crop == ((use_crop ? 1 : 0) | (c->prompt_reduce << 1))) {
                ARP_HIP_OK(hipGraphLaunch(g.exec, c->stream));
                return 0;
            }
        arp_clip::LatGraph g{frames_dev, rewards_dev, n, H, W, (use_crop ? 1 : 0) | (c->prompt_reduce << 1), nullptr, nullptr};
        // (Relaxed: another thread's HIP calls -- a reader thread pinning a buffer, a second handle -- must not invalidate the capture)
        ARP_HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        const int rc = pass(frames_dev, n, rewards_dev);
        const hipError_t ec = hipStreamEndCapture(c->stream, &g.graph);
        if (rc != 0 || ec != hipSuccess || !g.graph || hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
            // a capture that did not take: this handle goes on launch by launch (same kernels, same results)
            if (g.graph) (void)hipGraphDestroy(g.graph);
            for (int k = 0; k < 4 && hipGetLastError() != hipSuccess; ++k) {}
            c->lat_graph = false;
            return pass(frames_dev, n, rewards_dev);
        }
        if (c->lat_graphs.size() >= 8) {  // a caller that keeps changing buffers: forget the oldest
            (void)hipGraphExecDestroy(c->lat_graphs.front().exec);
            (void)hipGraphDestroy(c->lat_graphs.front().graph);
            c->lat_graphs.erase(c->lat_graphs.begin());
        }
        c->lat_graphs.push_back(g);
        ARP_HIP_OK(hipGraphLaunch(g.exec, c->stream));
        return 0;
    }
    for (int off = 0; off < n; off += mb) {
        const int nb = std::min(mb, n - off);
        ARP_TRY(pass(frames_dev + (size_t)off * H * W * 3, nb, rewards_dev + off));
    }
    return 0;
}

static int make_sibling(arp_clip* c) {
    arp_clip* s = new arp_clip(*c);  // shares every weight pointer; owns nothing of them
    s->is_sibling = true;
    s->siblings.clear();
    s->ev_join.clear();
    s->owned.clear();
    s->staged.clear();
    s->prof = Profiler();
    s->prof.on = c->prof.on;
    s->ws_frames = 0;
    DevBuf* bufs[] = {&s->patches, &s->pe, &s->x, &s->h, &s->qkv, &s->ao, &s->fc, &s->cls_h, &s->feat, &s->frames_in, &s->rewards, &s->stats};
    for (auto* b : bufs) *b = DevBuf();
    s->clock_buf = DevBuf();  // (the primary's; clock_acc is copied per call)
    s->part = DevBuf();  // the latency path never runs on a sibling (parts of >= 128 frames)
    s->lat_stats = DevBuf();
    s->lat_graphs.clear();
    s->pin_frames = nullptr; s->pin_frames_bytes = 0; s->pin_rewards = nullptr; s->pin_rewards_n = 0;
    s->stream = nullptr;
    s->ev_fork = nullptr;
    s->copy_stream = nullptr;
    s->ev_copy.clear();
    for (auto& ls : s->lslot) ls = arp_clip::LabelSlot();
    if (create_part_stream(&s->stream, (int)c->siblings.size() + 1) != hipSuccess) {
        delete s;
        return fail("hipStreamCreate failed");
    }
    if (!c->ev_fork) ARP_HIP_OK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    hipEvent_t ej = nullptr;
    ARP_HIP_OK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    c->ev_join.push_back(ej);
    c->siblings.push_back(s);
    return 0;
}

// host_src != null: the frames still live in host memory; every part uploads its own slice on its own stream right before its
// compute, so the upload of part i+1 overlaps the kernels of part i (the S2 seam hands over host buffers).
static int label_dev(arp_clip* c, const uint8_t* frames_dev, int n, int H, int W, int use_crop, float* rewards_dev, const uint8_t* host_src = nullptr,
                     int lead = 0) {
    ARP_TRY(check_ready(c, true));
    int ns = c->cfg.n_streams;
    while (ns > 1 && n / ns < 128) --ns;  // keep every part big enough to fill the chip's GEMM grid
    const size_t fbytes = (size_t)H * W * 3;
    if (ns < 2) {
        if (host_src) ARP_HIP_OK(hipMemcpyAsync(const_cast<uint8_t*>(frames_dev), host_src, (size_t)n * fbytes, hipMemcpyHostToDevice, c->stream));
        return label_dev_single(c, frames_dev, n, H, W, use_crop, rewards_dev);
    }
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    // host-fed calls may be cut into MORE parts than streams (ARP_CLIP_HOST_PARTS, round-robin over the streams): the first kernels then
    // start after 1 / parts of the upload.  Measured (profiles/r3_seam_sweep.txt): 2, 4, 6 or 8 parts on two streams all give 77-80 k
    // frames/s -- what a synchronous call loses is the pipeline's fill and drain, not the exposed first slice, and smaller parts cost
    // GEMM tile quantisation (c_proj at 256 frames: 150 tiles on 256 CUs) -- so the default stays one part per stream.
    int parts = ns;
    if (host_src) {
        static const int env_parts = getenv("ARP_CLIP_HOST_PARTS") ? atoi(getenv("ARP_CLIP_HOST_PARTS")) : 0;
        parts = env_parts > 0 ? env_parts : ns;
        while (parts > ns && n / parts < 128) --parts;
        if (parts < ns) parts = ns;
        if (!c->copy_stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        while ((int)c->ev_copy.size() < parts + 1) {
            hipEvent_t e = nullptr;
            ARP_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->ev_copy.push_back(e);
        }
    }
    while ((int)c->siblings.size() < ns - 1) ARP_TRY(make_sibling(c));
    ResizePlan* plan;
    ARP_TRY(get_plan(c, H, W, use_crop, &plan));
    ARP_HIP_OK(hipEventRecord(c->ev_fork, c->stream));
    for (int i = 1; i < ns; ++i) {
        arp_clip* s = c->siblings[i - 1];
        s->plans = c->plans;  // shared, owned by the primary
        s->txt_feat = c->txt_feat;
        s->txt_mean = c->txt_mean;
        s->prompt_reduce = c->prompt_reduce;
        s->n_prompts = c->n_prompts;
        s->logit_scale = c->logit_scale;
        s->prof.on = c->prof.on;
        s->clock_acc = c->clock_acc;
        ARP_HIP_OK(hipStreamWaitEvent(s->stream, c->ev_fork, 0));
    }
    // contiguous parts; part i runs on stream i % ns (0 = the primary, k = sibling k-1).
    // lead > 0 (a SYNCHRONOUS host-fed call): a short first part on the primary stream, so that kernels start after lead / n of the upload
    // instead of 1 / parts of it; the remaining frames are cut into `parts` equal parts, the first of them on the first sibling.
    if (!host_src || lead <= 0 || n < 8 * lead) lead = 0;
    const int per = (n - lead + parts - 1) / parts;
    for (int i = lead ? -1 : 0; i < parts; ++i) {
        const int b0 = i < 0 ? 0 : lead + i * per, nb = i < 0 ? lead : std::min(per, n - b0);
        if (nb <= 0) break;
        const int si = i < 0 ? 0 : (lead ? (i + 1) % ns : i % ns);
        arp_clip* s = si == 0 ? c : c->siblings[si - 1];
        if (host_src) {
            // (the staging buffer's previous readers -- the last call's kernels -- were synchronised before that call returned)
            ARP_HIP_OK(hipMemcpyAsync(const_cast<uint8_t*>(frames_dev) + (size_t)b0 * fbytes, host_src + (size_t)b0 * fbytes, (size_t)nb * fbytes,
                                      hipMemcpyHostToDevice, c->copy_stream));
            ARP_HIP_OK(hipEventRecord(c->ev_copy[i + 1], c->copy_stream));
            ARP_HIP_OK(hipStreamWaitEvent(s->stream, c->ev_copy[i + 1], 0));
        }
        s->shared_chip = true;
        const int rc = label_dev_single(s, frames_dev + (size_t)b0 * H * W * 3, nb, H, W, use_crop, rewards_dev + b0);
        s->shared_chip = false;
        if (rc) return rc;
    }
    for (int i = 1; i < ns; ++i) {  // join: the primary stream continues behind every sibling's last part
        ARP_HIP_OK(hipEventRecord(c->ev_join[i - 1], c->siblings[i - 1]->stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->stream, c->ev_join[i - 1], 0));
    }
    return 0;
}

}  // namespace arp

// =================================== C ABI ==========================================================
extern "C" {

const char* arp_last_error(void) { return g_err.c_str(); }
int arp_version(void) { return 100; }

int arp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int arp_dev_malloc(void** out, size_t bytes) {
    if (!out) return fail("null out");
    ARP_HIP_OK(hipMalloc(out, bytes ? bytes : 16));
    return 0;
}
int arp_dev_free(void* p) {
    if (p) ARP_HIP_OK(hipFree(p));
    return 0;
}
int arp_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    ARP_HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int arp_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    ARP_HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
// Pin a caller-owned host buffer (hipHostRegister) so that host-fed calls (arp_clip_label, arp_dt_upload_batch_async) move it by true
// asynchronous DMA instead of staging it.  Registration costs milliseconds per 100 MB: worth it for buffers that are REUSED (a frame
// reader's recycled batch buffers); the caller unregisters before freeing the memory.
int arp_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail("bad argument");
    ARP_HIP_OK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return 0;
}
int arp_host_unregister(void* p) {
    if (!p) return 0;
    ARP_HIP_OK(hipHostUnregister(p));
    return 0;
}
int arp_set_device(int device) {
    ARP_HIP_OK(hipSetDevice(device));
    return 0;
}
int arp_dev_synchronize(void) {
    ARP_HIP_OK(hipDeviceSynchronize());
    return 0;
}

int arp_clip_create(const arp_clip_cfg* cfg, arp_clip** out) {
    if (!cfg || !out) return fail("null argument");
    const arp_clip_cfg& k = *cfg;
    if (k.mode != ARP_MODE_F32 && k.mode != ARP_MODE_BF16 && k.mode != ARP_MODE_F16) return fail("bad mode");
    if (k.patch <= 0 || k.img_res % k.patch || k.patch % 4 || k.img_res % 4) return fail("bad patch / img_res");
    if (k.width % k.heads || k.txt_width % k.txt_heads) return fail("width not divisible by heads");
    const int kq = (k.mode == ARP_MODE_F32) ? 32 : 64;
    if (k.width % kq || k.txt_width % kq || (3 * k.patch * k.patch) % kq)
        return fail("width / txt_width / 3*patch^2 must be multiples of " + std::to_string(kq));
    if (k.embed % 4) return fail("embed must be a multiple of 4");
    int ndev = 0;
    ARP_HIP_OK(hipGetDeviceCount(&ndev));
    if (k.device < 0 || k.device >= ndev) return fail("no such HIP device: " + std::to_string(k.device));
    ARP_HIP_OK(hipSetDevice(k.device));
    ARP_TRY(prime_runtime(k.device));  // (runtime.h: one null-stream copy before the process's first stream exists)
    arp_clip* c = new arp_clip();
    c->cfg = k;
    if (c->cfg.max_batch <= 0) c->cfg.max_batch = 1024;
    if (const char* e = getenv("ARP_GEMM")) c->gemm_force = atoi(e);
    if (const char* e = getenv("ARP_LN_FOLD")) c->ln_fold = atoi(e) != 0;
    if (const char* e = getenv("ARP_CLS_ONLY")) c->cls_only_last = atoi(e) != 0;
    if (const char* e = getenv("ARP_QKV_FUSED")) c->qkv_fused = atoi(e) != 0;
    if (const char* e = getenv("ARP_SKINNY")) c->skinny = atoi(e) != 0;
    if (const char* e = getenv("ARP_LAT_FOLD")) c->lat_fold = atoi(e) != 0;
    if (const char* e = getenv("ARP_CLIP_GRAPH")) c->lat_graph = atoi(e) != 0;
    if (const char* e = getenv("ARP_SKINNY_ROWS")) c->lat_rows = std::min(std::max(atoi(e), 1), SKINNY_MAX_M);
    if (const char* e = getenv("ARP_CLIP_PINNED")) c->lat_pinned = atoi(e) != 0;
    if (create_part_stream(&c->stream, 0) != hipSuccess) {
        delete c;
        return fail("hipStreamCreate failed");
    }
    *out = c;
    return 0;
}

int arp_clip_destroy(arp_clip* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    (void)hipStreamSynchronize(c->stream);
    for (arp_clip* s : c->siblings) {
        (void)hipStreamSynchronize(s->stream);
        s->prof.destroy();
        DevBuf* sb[] = {&s->patches, &s->pe, &s->x, &s->h, &s->qkv, &s->ao, &s->fc, &s->cls_h, &s->feat, &s->frames_in, &s->rewards, &s->stats};
        for (auto* b : sb) b->release();
        (void)hipStreamDestroy(s->stream);
        s->plans.clear();
        delete s;
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_join) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_copy) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& ls : c->lslot) {
        ls.frames.release();
        ls.rewards.release();
        if (ls.host) (void)hipHostFree(ls.host);
        if (ls.done) (void)hipEventDestroy(ls.done);
    }
    c->prof.destroy();
    drop_lat_graphs(c);
    c->part.release();
    c->lat_stats.release();
    c->clock_buf.release();
    if (c->pin_frames) (void)hipHostFree(c->pin_frames);
    if (c->pin_rewards) (void)hipHostFree(c->pin_rewards);
    for (void* p : c->owned) (void)hipFree(p);
    for (auto& kv : c->plans) {
        kv.second->h_tab.release();
        kv.second->v_tab.release();
        delete kv.second;
    }
    DevBuf* bufs[] = {&c->txt_feat, &c->txt_mean, &c->ms_keep, &c->patches, &c->pe, &c->x, &c->h, &c->qkv, &c->ao, &c->fc, &c->cls_h, &c->feat, &c->frames_in, &c->rewards, &c->stats};
    for (auto* b : bufs) b->release();
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int arp_clip_load_weight(arp_clip* c, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!c || !name || !data || (ndim > 0 && !shape)) return fail("null argument");
    if (c->finalized) return fail("weights already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail("negative dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    c->staged[name] = std::move(t);
    return 0;
}

int arp_clip_finalize_weights(arp_clip* c) {
    if (!c) return fail("null handle");
    if (c->finalized) return 0;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const arp_clip_cfg& k = c->cfg;
    const int D = k.width, P = k.patch, E = k.embed, Tw = k.txt_width, N = c->ntok();
    const HostTensor* t;
    ARP_TRY(get_staged(c, "visual.conv1.weight", {D, 3, P, P}, &t)); ARP_TRY(upload_mat(c, t->data.data(), D, 3 * P * P, false, &c->conv_w));
    ARP_TRY(get_staged(c, "visual.class_embedding", {D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->cls));
    ARP_TRY(get_staged(c, "visual.positional_embedding", {N, D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->pos));
    ARP_TRY(get_staged(c, "visual.ln_pre.weight", {D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnpre_w));
    ARP_TRY(get_staged(c, "visual.ln_pre.bias", {D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnpre_b));
    ARP_TRY(get_staged(c, "visual.ln_post.weight", {D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnpost_w));
    ARP_TRY(get_staged(c, "visual.ln_post.bias", {D}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnpost_b));
    ARP_TRY(get_staged(c, "visual.proj", {D, E}, &t)); ARP_TRY(upload_mat(c, t->data.data(), D, E, true, &c->proj_t));
    ARP_TRY(load_tower(c, "visual.transformer.", D, k.layers, k.heads, c->vis, c->ln_fold && (k.mode == ARP_MODE_BF16 || k.mode == ARP_MODE_F16) && (D & 127) == 0, -1, c->ntok()));
    ARP_TRY(get_staged(c, "token_embedding.weight", {k.vocab, Tw}, &t)); ARP_TRY(upload_f32(c, t->data, &c->tok_emb));
    ARP_TRY(get_staged(c, "positional_embedding", {k.ctx, Tw}, &t)); ARP_TRY(upload_f32(c, t->data, &c->tpos));
    ARP_TRY(get_staged(c, "ln_final.weight", {Tw}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnf_w));
    ARP_TRY(get_staged(c, "ln_final.bias", {Tw}, &t)); ARP_TRY(upload_f32(c, t->data, &c->lnf_b));
    ARP_TRY(get_staged(c, "text_projection", {Tw, E}, &t)); ARP_TRY(upload_mat(c, t->data.data(), Tw, E, true, &c->tproj_t));
    ARP_TRY(load_tower(c, "transformer.", Tw, k.txt_layers, k.txt_heads, c->txt, false));
    if (k.mode != ARP_MODE_F32) {  // f32 text tower for the cached prompt features (see arp_clip::txt32)
        ARP_TRY(get_staged(c, "text_projection", {Tw, E}, &t)); ARP_TRY(upload_mat(c, t->data.data(), Tw, E, true, &c->tproj_t32, ARP_MODE_F32));
        ARP_TRY(load_tower(c, "transformer.", Tw, k.txt_layers, k.txt_heads, c->txt32, false, ARP_MODE_F32));
    }
    if (c->staged.count("logit_scale") && c->staged["logit_scale"].shape == std::vector<int64_t>{1}) c->staged["logit_scale"].shape.clear();
    ARP_TRY(get_staged(c, "logit_scale", {}, &t));
    c->logit_scale = t->data[0];
    std::vector<float> lut(768);
    build_lut(lut.data());
    ARP_TRY(upload_f32(c, lut, &c->lut));
    ARP_TRY(c->part.ensure((size_t)4 * SKINNY_MAX_M * std::max(D, Tw) * 4));
    ARP_TRY(c->lat_stats.ensure((size_t)SKINNY_MAX_M * std::max(D >> 4, 1) * 8));
    c->staged.clear();
    c->finalized = true;
    return 0;
}

static __global__ __launch_bounds__(256) void prompt_mean_kernel(const float* __restrict__ txt, float* __restrict__ out, int np, int E) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= E) return;
    float s = 0.f;
    for (int p = 0; p < np; ++p) s += txt[(size_t)p * E + c];
    out[c] = s / (float)np;
}

int arp_clip_set_text(arp_clip* c, const int32_t* tokens, int n_prompts) {
    ARP_TRY(check_ready(c, false));
    if (!tokens || n_prompts <= 0) return fail("set_text: need at least one prompt");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (!c->lat_graphs.empty()) {  // the captured passes read the old prompt features
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        drop_lat_graphs(c);
    }
    // 16-bit handles: the cached prompt features come from the f32 copy of the text tower (arp_clip::txt32); ARP_TEXT_F32=0 keeps
    // the handle's own operand type (A/B measurements)
    static const bool text32 = [] { const char* e = getenv("ARP_TEXT_F32"); return !e || atoi(e) != 0; }();
    int rc;
    if (c->cfg.mode != ARP_MODE_F32 && text32) rc = run_text<float>(c, tokens, n_prompts, nullptr, true, nullptr, false, true);
    else if (c->cfg.mode == ARP_MODE_BF16) rc = run_text<bf16_t>(c, tokens, n_prompts);
    else if (c->cfg.mode == ARP_MODE_F16) rc = run_text<f16_t>(c, tokens, n_prompts);
    else rc = run_text<float>(c, tokens, n_prompts);
    ARP_TRY(rc);
    ARP_TRY(c->txt_mean.ensure((size_t)c->cfg.embed * 4));
    hipLaunchKernelGGL(prompt_mean_kernel, dim3((c->cfg.embed + 255) / 256), dim3(256), 0, c->stream, c->txt_feat.as<float>(), c->txt_mean.as<float>(),
                       n_prompts, c->cfg.embed);
    ARP_HIP_OK(hipGetLastError());
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

// 0: rewards are logits_per_text[0] -- prompt 0 whatever the number of cached prompts (arp_dt/label_reward.py:146; the offline pass).
// 1: rewards are logits_per_text.mean(axis=0) over the cached prompts -- the rollout loop's branch for a LIST of prompts
//    (arp_dt/envs/vl_reward.py:19-22, get_torch_clip_reward; :56-59 for the adapter model).
int arp_clip_set_prompt_reduce(arp_clip* c, int mode) {
    ARP_TRY(check_ready(c, false));
    if (mode != 0 && mode != 1) return fail("set_prompt_reduce: mode is 0 (prompt 0) or 1 (mean over the prompts)");
    c->prompt_reduce = mode;  // (the captured single-frame passes are keyed by it: switching back and forth replays, nothing is dropped)
    return 0;
}

int arp_clip_get_text_features(arp_clip* c, float* out) {
    ARP_TRY(check_ready(c, true));
    if (!out) return fail("null out");
    ARP_HIP_OK(hipMemcpy(out, c->txt_feat.p, (size_t)c->n_prompts * c->cfg.embed * 4, hipMemcpyDeviceToHost));
    return 0;
}

int arp_clip_label_dev_async(arp_clip* c, const uint8_t* frames_dev, int n, int H, int W, int use_crop, float* rewards_dev) {
    if (n > 0 && (!frames_dev || !rewards_dev)) return fail("null buffer");
    return label_dev(c, frames_dev, n, H, W, use_crop, rewards_dev);
}

int arp_clip_sync(arp_clip* c) {
    if (!c) return fail("null handle");
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_clip_label(arp_clip* c, const uint8_t* frames, int n, int H, int W, int use_crop, float* rewards) {
    ARP_TRY(check_ready(c, true));
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    if (!frames || !rewards) return fail("null buffer");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const size_t fb = (size_t)H * W * 3;
    const int mb = c->cfg.max_batch;
    if (c->lat_pinned && n <= mb && (long)n * c->ntok() <= c->lat_rows) {
        if (c->pin_frames_bytes < (size_t)n * fb) {
            ARP_HIP_OK(hipStreamSynchronize(c->stream));
            if (c->pin_frames) ARP_HIP_OK(hipHostFree(c->pin_frames));
            c->pin_frames = nullptr; c->pin_frames_bytes = 0;
            ARP_HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&c->pin_frames), (size_t)n * fb, hipHostMallocDefault));
            c->pin_frames_bytes = (size_t)n * fb;
        }
        if (c->pin_rewards_n < (size_t)n) {
            ARP_HIP_OK(hipStreamSynchronize(c->stream));
            if (c->pin_rewards) ARP_HIP_OK(hipHostFree(c->pin_rewards));
            c->pin_rewards = nullptr; c->pin_rewards_n = 0;
            ARP_HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&c->pin_rewards), (size_t)std::max(n, 16) * 4, hipHostMallocDefault));
            c->pin_rewards_n = (size_t)std::max(n, 16);
        }
        memcpy(c->pin_frames, frames, (size_t)n * fb);
        ARP_TRY(label_dev_single(c, c->pin_frames, n, H, W, use_crop, c->pin_rewards));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        memcpy(rewards, c->pin_rewards, (size_t)n * 4);
        return 0;
    }
    ARP_TRY(c->frames_in.ensure((size_t)std::min(n, mb) * fb));
    ARP_TRY(c->rewards.ensure((size_t)std::min(n, mb) * 4));
    for (int off = 0; off < n; off += mb) {
        const int nb = std::min(mb, n - off);
        // measured at 1 024 frames (scripts/lead_probe.py): lead 0: 13.24 ms per call, 64: 12.83, 96: 12.60, 128: 12.50, 192: 13.40; same bits
        static const int lead_frames = getenv("ARP_CLIP_LEAD") ? atoi(getenv("ARP_CLIP_LEAD")) : 128;
        ARP_TRY(label_dev(c, c->frames_in.as<uint8_t>(), nb, H, W, use_crop, c->rewards.as<float>(), frames + (size_t)off * fb, lead_frames));
        ARP_HIP_OK(hipMemcpyAsync(rewards + off, c->rewards.p, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// Asynchronous form of arp_clip_label: submit() enqueues the upload, the labelling pass and the download of n <= max_batch frames on
// slot 0 / 1 and returns; collect() waits for that slot and hands the rewards over.  Two slots: the next call is submitted before the
// previous one is collected, so its upload overlaps the running pass and the compute streams never drain between calls.  The caller
// keeps `frames` alive and unchanged until collect() of the same slot has returned.
int arp_clip_label_submit(arp_clip* c, int slot, const uint8_t* frames, int n, int H, int W, int use_crop) {
    ARP_TRY(check_ready(c, true));
    if (slot < 0 || slot > 1 || !frames || n <= 0) return fail("bad argument");
    if (n > c->cfg.max_batch) return fail("arp_clip_label_submit: at most max_batch frames per call");
    arp_clip::LabelSlot& ls = c->lslot[slot];
    if (ls.busy) return fail("arp_clip_label_submit: the slot still holds an uncollected call");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const size_t fb = (size_t)H * W * 3;
    ARP_TRY(ls.frames.ensure((size_t)c->cfg.max_batch * fb));
    ARP_TRY(ls.rewards.ensure((size_t)c->cfg.max_batch * 4));
    if (ls.host_n < (size_t)c->cfg.max_batch) {
        if (ls.host) ARP_HIP_OK(hipHostFree(ls.host));
        ls.host = nullptr;
        ARP_HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&ls.host), (size_t)c->cfg.max_batch * 4, hipHostMallocDefault));
        ls.host_n = (size_t)c->cfg.max_batch;
    }
    if (!ls.done) ARP_HIP_OK(hipEventCreateWithFlags(&ls.done, hipEventDisableTiming));
    ARP_TRY(label_dev(c, ls.frames.as<uint8_t>(), n, H, W, use_crop, ls.rewards.as<float>(), frames));
    ARP_HIP_OK(hipMemcpyAsync(ls.host, ls.rewards.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipEventRecord(ls.done, c->stream));
    ls.n = n;
    ls.busy = true;
    return 0;
}
int arp_clip_label_collect(arp_clip* c, int slot, float* rewards) {
    if (!c || slot < 0 || slot > 1 || !rewards) return fail("bad argument");
    arp_clip::LabelSlot& ls = c->lslot[slot];
    if (!ls.busy) return fail("arp_clip_label_collect: nothing was submitted on this slot");
    ARP_HIP_OK(hipEventSynchronize(ls.done));
    memcpy(rewards, ls.host, (size_t)ls.n * 4);
    ls.busy = false;
    return 0;
}

int arp_clip_encode_image(arp_clip* c, const uint8_t* frames, int n, int H, int W, int use_crop, int normalize, float* out) {
    ARP_TRY(check_ready(c, false));
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    if (!frames || !out) return fail("null buffer");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ResizePlan* plan;
    const int mb = c->cfg.max_batch, E = c->cfg.embed;
    // (a call of one or two frames -- the rollout loop's goal-conditioned reward, envs/vl_reward.py:26-41 -- gets the small-tile preprocess plan)
    ARP_TRY(get_plan(c, H, W, use_crop, &plan, n <= mb && (long)n * c->ntok() <= c->lat_rows));
    const size_t fb = (size_t)H * W * 3;
    ARP_TRY(c->frames_in.ensure((size_t)std::min(n, mb) * fb));
    ARP_TRY(ensure_workspace(c, std::min(n, mb)));
    for (int off = 0; off < n; off += mb) {
        const int nb = std::min(mb, n - off);
        ARP_HIP_OK(hipMemcpyAsync(c->frames_in.p, frames + (size_t)off * fb, (size_t)nb * fb, hipMemcpyHostToDevice, c->stream));
        ARP_TRY(forward_chunk_dispatch(c, c->frames_in.as<uint8_t>(), nb, plan));
        if (normalize) {
            hipLaunchKernelGGL(l2_normalize_kernel, dim3((nb + 3) / 4), dim3(256), 0, c->stream, c->feat.as<float>(), nb, E);
            ARP_HIP_OK(hipGetLastError());
        }
        ARP_HIP_OK(hipMemcpyAsync(out + (size_t)off * E, c->feat.p, (size_t)nb * E * 4, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// Frozen-tower outputs for the fine-tune head (row N2): per-block CLS features [n, layers*width] and the un-normalised
// CLIP image feature [n, embed], through the fine-tune transform (bilinear; clip_multiscale_adapter.py:120-149).
// pil_crop < 0: the fine-tune transform; 0 / 1: the LABEL transform instead (Pillow bicubic + normalise, use_crop = pil_crop) -- what the rollout
// loop's adapter rewards feed the model: `model.encode_image(preprocess(Image.fromarray(obs)))`, envs/vl_reward.py:44-79.
static int encode_image_multiscale(arp_clip* c, const uint8_t* frames, int n, int H, int W, float* inter, float* final_feat, bool dev_out, int pil_crop = -1) {
    ARP_TRY(check_ready(c, false));
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    if (!frames || !inter || !final_feat) return fail("null buffer");
    const int R = c->cfg.img_res;
    if (H <= 0 || W <= 0 || H > 0xffff || W > 0xffff) return fail("bad frame geometry");
    if (pil_crop < 0 && (H != R) != (W != R)) return fail("the reference resizes only when BOTH sides differ from 224 (clip_multiscale_adapter.py:127): "
                                          "a frame with exactly one side at 224 cannot enter the tower");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    const size_t fb = (size_t)H * W * 3;
    const int mb = c->cfg.max_batch, E = c->cfg.embed, LD = c->vis.layers * c->cfg.width;
    ResizePlan* plan = nullptr;
    if (pil_crop >= 0) ARP_TRY(get_plan(c, H, W, pil_crop, &plan, n <= mb && (long)n * c->ntok() <= c->lat_rows));
    ARP_TRY(c->frames_in.ensure((size_t)std::min(n, mb) * fb));
    ARP_TRY(ensure_workspace(c, std::min(n, mb)));
    const bool keep = (size_t)std::min(n, mb) * LD * 4 <= (1u << 20);  // a few frames: the export buffer stays with the handle
    DevBuf ms_tmp;
    DevBuf& ms = keep ? c->ms_keep : ms_tmp;
    int rc = 0;
    auto body = [&]() -> int {
        ARP_TRY(ms.ensure((size_t)std::min(n, mb) * LD * 4));
        for (int off = 0; off < n; off += mb) {
            const int nb = std::min(mb, n - off);
            ARP_HIP_OK(hipMemcpyAsync(c->frames_in.p, frames + (size_t)off * fb, (size_t)nb * fb, hipMemcpyHostToDevice, c->stream));
            c->ms_out = ms.as<float>(); c->ms_ld = LD; c->ms_rows = nullptr; c->pre_bilinear = pil_crop < 0 ? ((H << 16) | W) : 0;
            const int r = forward_chunk_dispatch(c, c->frames_in.as<uint8_t>(), nb, plan);
            c->ms_out = nullptr; c->pre_bilinear = 0;
            ARP_TRY(r);
            const hipMemcpyKind kind = dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
            ARP_HIP_OK(hipMemcpyAsync(final_feat + (size_t)off * E, c->feat.p, (size_t)nb * E * 4, kind, c->stream));
            ARP_HIP_OK(hipMemcpyAsync(inter + (size_t)off * LD, ms.p, (size_t)nb * LD * 4, kind, c->stream));
            ARP_HIP_OK(hipStreamSynchronize(c->stream));
        }
        return 0;
    };
    rc = body();
    c->ms_out = nullptr; c->pre_bilinear = 0;
    ms_tmp.release();
    return rc;
}

int arp_clip_encode_image_multiscale(arp_clip* c, const uint8_t* frames, int n, int H, int W, float* inter, float* final_feat) {
    return encode_image_multiscale(c, frames, n, H, W, inter, final_feat, false);
}
// The same outputs through the LABEL transform (Pillow-exact bicubic resize to 224 + normalise; use_crop as arp_clip_label): the input the rollout
// loop's adapter rewards give the fine-tuned model (arp_dt/envs/vl_reward.py:44-61, 64-79).
int arp_clip_encode_image_multiscale_pil(arp_clip* c, const uint8_t* frames, int n, int H, int W, int use_crop, float* inter, float* final_feat) {
    return encode_image_multiscale(c, frames, n, H, W, inter, final_feat, false, use_crop ? 1 : 0);
}
// Same, with the two outputs in DEVICE memory (arp_dev_malloc): the features go to arp_ft_set_batch_dev without touching the host.
int arp_clip_encode_image_multiscale_dev(arp_clip* c, const uint8_t* frames, int n, int H, int W, float* inter_dev, float* final_dev) {
    return encode_image_multiscale(c, frames, n, H, W, inter_dev, final_dev, true);
}

// Text side of the same: per-block EOT-token features [n, layers*txt_width] and the un-normalised text feature [n, embed]
// (clip_multiscale_adapter.py:151-166).  Does not touch the cached prompt set of arp_clip_set_text.
int arp_clip_encode_text_multiscale(arp_clip* c, const int32_t* tokens, int n, float* inter, float* final_feat) {
    ARP_TRY(check_ready(c, false));
    if (!tokens || n <= 0 || !inter || !final_feat) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_BF16) return run_text<bf16_t>(c, tokens, n, final_feat, false, inter);
    if (c->cfg.mode == ARP_MODE_F16) return run_text<f16_t>(c, tokens, n, final_feat, false, inter);
    return run_text<float>(c, tokens, n, final_feat, false, inter);
}
int arp_clip_encode_text_multiscale_dev(arp_clip* c, const int32_t* tokens, int n, float* inter_dev, float* final_dev) {
    ARP_TRY(check_ready(c, false));
    if (!tokens || n <= 0 || !inter_dev || !final_dev) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_BF16) return run_text<bf16_t>(c, tokens, n, final_dev, false, inter_dev, true);
    if (c->cfg.mode == ARP_MODE_F16) return run_text<f16_t>(c, tokens, n, final_dev, false, inter_dev, true);
    return run_text<float>(c, tokens, n, final_dev, false, inter_dev, true);
}

int arp_bicubic_coeffs(int in_size, int out_size, int32_t* xmin, int32_t* cnt, int32_t* weights, int ksize_cap) {
    if (in_size <= 0 || out_size <= 0 || !xmin || !cnt || !weights) return fail("bad argument");
    ResampleTable t;
    build_bicubic_table(in_size, out_size, t);
    if (t.kmax > ksize_cap) return fail("ksize_cap too small, need " + std::to_string(t.kmax));
    for (int o = 0; o < out_size; ++o) {
        xmin[o] = t.xmin[o];
        cnt[o] = t.cnt[o];
        for (int k = 0; k < ksize_cap; ++k) weights[(size_t)o * ksize_cap + k] = k < t.cnt[o] ? t.w[(size_t)o * t.ksize + k] : 0;
    }
    return t.kmax;
}

int arp_preprocess(const uint8_t* frames, int n, int H, int W, int use_crop, int res, float* out) {
    if (n < 0) return fail("negative frame count");
    if (n == 0) return 0;
    if (!frames || !out) return fail("null buffer");
    ResizePlan plan;
    DevBuf in, o, lut;
    int rc = 0;
    auto body = [&]() -> int {
        ARP_TRY(build_plan(H, W, use_crop, res, plan));
        const size_t fb = (size_t)H * W * 3, ob = (size_t)3 * res * res * 4;
        ARP_TRY(in.ensure((size_t)n * fb)); ARP_TRY(o.ensure((size_t)n * ob)); ARP_TRY(lut.ensure(768 * 4));
        std::vector<float> l(768);
        build_lut(l.data());
        ARP_HIP_OK(hipMemcpy(lut.p, l.data(), 768 * 4, hipMemcpyHostToDevice));
        ARP_HIP_OK(hipMemcpy(in.p, frames, (size_t)n * fb, hipMemcpyHostToDevice));
        ARP_TRY((launch_preprocess<float, PRE_NCHW>(plan, in.as<uint8_t>(), n, 4, lut.as<float>(), o.p, nullptr)));
        ARP_HIP_OK(hipDeviceSynchronize());
        ARP_HIP_OK(hipMemcpy(out, o.p, (size_t)n * ob, hipMemcpyDeviceToHost));
        return 0;
    };
    rc = body();
    in.release(); o.release(); lut.release(); plan.h_tab.release(); plan.v_tab.release();
    return rc;
}

int arp_clip_set_fp8_mlp(arp_clip* c, int on) {
    if (!c) return fail("null handle");
    if (c->finalized) return fail("arp_clip_set_fp8_mlp must come before arp_clip_finalize_weights (the e4m3 weight copies are made there)");
    if (on && c->cfg.mode == ARP_MODE_F32) return fail("fp8 MLP needs a 16-bit mode for the rest of the tower");
    if (on && c->cfg.width % 128) return fail("fp8 MLP needs width % 128 == 0");
    c->fp8_mlp = on != 0;
    c->fp8_attn = on >= 2;
    return 0;
}

int arp_clip_set_streams(arp_clip* c, int n_streams) {
    if (!c || n_streams < 0 || n_streams > 4) return fail("n_streams must be 0..4");
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    for (arp_clip* s : c->siblings) ARP_HIP_OK(hipStreamSynchronize(s->stream));
    c->cfg.n_streams = n_streams;
    return 0;
}

// The clock the chip holds under the labelling pass (MI355X_MICROARCH 'DVFS give-back' item 6).  on = 1: the vision tower's c_fc launches run on the
// clock-diagnostic instance of the 256 x 256 GEMM (gemm256.h, CLK: one s_memtime / s_memrealtime pair around each workgroup, summed into a buffer of their
// own) until on = 0; the accumulators are zeroed at every on = 1.  Rewards are unchanged (same tile, same K loop, same epilogue).
int arp_clip_clock_probe(arp_clip* c, int on) {
    if (!c) return fail("null handle");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    for (arp_clip* s : c->siblings) ARP_HIP_OK(hipStreamSynchronize(s->stream));
    if (on) {
        ARP_TRY(c->clock_buf.ensure(64));
        ARP_HIP_OK(hipMemset(c->clock_buf.p, 0, 64));
        c->clock_acc = c->clock_buf.as<unsigned long long>();
    } else {
        c->clock_acc = nullptr;
    }
    for (arp_clip* s : c->siblings) s->clock_acc = c->clock_acc;
    // (a captured single-frame pass holds the instance it was captured with)
    return 0;
}
// out[0] = clock in GHz = sum d(s_memtime) / sum d(s_memrealtime) x 0.1 (time-weighted over the probed workgroups), out[1] = workgroups probed,
// out[2] = mean workgroup duration in microseconds (s_memrealtime ticks are 10 ns)
int arp_clip_clock_read(arp_clip* c, double* out) {
    if (!c || !out) return fail("null argument");
    if (!c->clock_buf.p) return fail("arp_clip_clock_probe was never switched on");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    for (arp_clip* s : c->siblings) ARP_HIP_OK(hipStreamSynchronize(s->stream));
    unsigned long long v[3] = {0, 0, 0};
    ARP_HIP_OK(hipMemcpy(v, c->clock_buf.p, sizeof(v), hipMemcpyDeviceToHost));
    out[0] = v[1] ? (double)v[0] / (double)v[1] * 0.1 : 0.0;
    out[1] = (double)v[2];
    out[2] = v[2] ? (double)v[1] / (double)v[2] * 0.01 : 0.0;
    return 0;
}

int arp_clip_profile_enable(arp_clip* c, int on) {
    if (!c) return fail("null handle");
    c->prof.on = on != 0;
    for (arp_clip* s : c->siblings) s->prof.on = c->prof.on;
    return 0;
}
int arp_clip_profile_reset(arp_clip* c) {
    if (!c) return fail("null handle");
    c->prof.reset();
    for (arp_clip* s : c->siblings) s->prof.reset();
    return 0;
}
int arp_clip_profile_json(arp_clip* c, char* buf, int buf_len) {
    if (!c || !buf) return fail("null argument");
    for (arp_clip* sb : c->siblings) {  // fold the other streams' launches into the primary's sites
        Profiler& q = sb->prof;
        q.collect();
        c->prof.collect();
        for (size_t i = 0; i < q.names.size(); ++i) {
            const int id = c->prof.site_id(q.names[i].c_str());
            c->prof.ms[id] += q.ms[i];
            c->prof.calls[id] += q.calls[i];
            q.ms[i] = 0.0;
            q.calls[i] = 0;
        }
    }
    const std::string s = c->prof.json();
    if ((int)s.size() + 1 > buf_len) return fail("profile buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int arp_event_create(arp_event** out) {
    if (!out) return fail("null out");
    arp_event* ev = new arp_event();
    if (hipEventCreate(&ev->e) != hipSuccess) {
        delete ev;
        return fail("hipEventCreate failed");
    }
    *out = ev;
    return 0;
}
int arp_event_destroy(arp_event* e) {
    if (e) {
        (void)hipEventDestroy(e->e);
        delete e;
    }
    return 0;
}
int arp_clip_event_record(arp_clip* c, arp_event* e) {
    if (!c || !e) return fail("null argument");
    ARP_HIP_OK(hipEventRecord(e->e, c->stream));
    return 0;
}
int arp_event_elapsed_ms(arp_event* a, arp_event* b, float* ms) {
    if (!a || !b || !ms) return fail("null argument");
    ARP_HIP_OK(hipEventSynchronize(b->e));
    ARP_HIP_OK(hipEventElapsedTime(ms, a->e, b->e));
    return 0;
}

}  // extern "C"

// ---- single-operator entry points ------------------------------------------------------------------
template <typename T> static int to_dev(const float* src, size_t n, DevBuf& d) {
    ARP_TRY(d.ensure(std::max<size_t>(n * sizeof(T), 16)));
    if constexpr (std::is_same_v<T, f16_t>) {
        std::vector<f16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2h(src[i]);
        ARP_HIP_OK(hipMemcpy(d.p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else if constexpr (sizeof(T) == 2) {
        std::vector<bf16_t> hb(n);
        for (size_t i = 0; i < n; ++i) hb[i] = host_f2bf(src[i]);
        ARP_HIP_OK(hipMemcpy(d.p, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        ARP_HIP_OK(hipMemcpy(d.p, src, n * 4, hipMemcpyHostToDevice));
    }
    return 0;
}
template <typename T> static int from_dev(float* dst, size_t n, const DevBuf& d) {
    if constexpr (std::is_same_v<T, f16_t>) {
        std::vector<uint16_t> hb(n);
        ARP_HIP_OK(hipMemcpy(hb.data(), d.p, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) dst[i] = (float)__builtin_bit_cast(_Float16, hb[i]);
    } else if constexpr (sizeof(T) == 2) {
        std::vector<bf16_t> hb(n);
        ARP_HIP_OK(hipMemcpy(hb.data(), d.p, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            const uint32_t u = (uint32_t)hb[i] << 16;
            memcpy(&dst[i], &u, 4);
        }
    } else {
        ARP_HIP_OK(hipMemcpy(dst, d.p, n * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}

template <typename T> static int op_gemm(int act, const float* A, const float* W, const float* bias, const float* resid, float* out,
                                         int M, int N, int K) {
    DevBuf dA, dW, dB, dR, dO;
    auto body = [&]() -> int {
        ARP_TRY(to_dev<T>(A, (size_t)M * K, dA)); ARP_TRY(to_dev<T>(W, (size_t)N * K, dW));
        if (bias) ARP_TRY(to_dev<float>(bias, N, dB));
        ARP_TRY(dO.ensure((size_t)M * N * 4));
        if (resid) ARP_TRY(to_dev<float>(resid, (size_t)M * N, dR));
        GemmArgs g;
        g.A = dA.p; g.W = dW.p; g.bias = bias ? dB.as<float>() : nullptr; g.resid = resid ? dR.as<float>() : nullptr; g.out = dO.p;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
        int rc = -1;
        const char* fe = getenv("ARP_GEMM");
        const int force = fe ? atoi(fe) : 0;
#define ARP_OP_CASE(a)                                                                                            \
    case a: rc = resid ? launch_gemm_auto<T, float, a, true, SITE_OP>(g, nullptr, force) : launch_gemm_auto<T, float, a, false, SITE_OP>(g, nullptr, force); break;
        switch (act) {
            ARP_OP_CASE(ACT_NONE) ARP_OP_CASE(ACT_QGELU) ARP_OP_CASE(ACT_RELU) ARP_OP_CASE(ACT_TANH) ARP_OP_CASE(ACT_GELU_TANH)
            default: return fail("bad activation");
        }
#undef ARP_OP_CASE
        ARP_TRY(rc);
        ARP_HIP_OK(hipDeviceSynchronize());
        return from_dev<float>(out, (size_t)M * N, dO);
    };
    const int rc = body();
    dA.release(); dW.release(); dB.release(); dR.release(); dO.release();
    return rc;
}

// The latency path's GEMM (skinny.h) on host arrays.  ksplit = 0: one launch with the bias / activation / residual epilogue;
// ksplit >= 1: raw split-K slabs + the row kernel: out = resid + bias + A.W^T (resid required), h_out = LayerNorm(out) when ln_w is given.
template <typename T> static int op_skinny(int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N, int K,
                                           int ksplit, const float* ln_w, const float* ln_b, float eps, float* h_out) {
    DevBuf dA, dW, dB, dR, dO, dP, dH, dLw, dLb;
    const int tcode = __is_same(T, bf16_t) ? 1 : 2;
    auto body = [&]() -> int {
        ARP_TRY(to_dev<T>(A, (size_t)M * K, dA)); ARP_TRY(to_dev<T>(W, (size_t)N * K, dW));
        if (bias) ARP_TRY(to_dev<float>(bias, N, dB));
        SkinnyArgs k;
        k.A = dA.p; k.W = dW.p; k.M = M; k.N = N; k.K = K; k.lda = K; k.ldw = K; k.ldr = N; k.ldo = N; k.out_f32 = 1;
        if (ksplit == 0) {
            ARP_TRY(dO.ensure((size_t)M * N * 4));
            if (resid) ARP_TRY(to_dev<float>(resid, (size_t)M * N, dR));
            k.bias = bias ? dB.as<float>() : nullptr; k.resid = resid ? dR.as<float>() : nullptr; k.out = dO.p; k.act = act;
            ARP_TRY(launch_skinny_gemm(tcode, k, nullptr));
            ARP_HIP_OK(hipDeviceSynchronize());
            return from_dev<float>(out, (size_t)M * N, dO);
        }
        if (!resid || act != ACT_NONE) return fail("skinny split: needs the residual, takes no activation");
        ARP_TRY(to_dev<float>(resid, (size_t)M * N, dR));
        ARP_TRY(dP.ensure((size_t)ksplit * M * N * 4));
        if (ln_w) {
            if (!ln_b || !h_out) return fail("skinny split: ln_w needs ln_b and h_out");
            ARP_TRY(to_dev<float>(ln_w, N, dLw)); ARP_TRY(to_dev<float>(ln_b, N, dLb)); ARP_TRY(dH.ensure((size_t)M * N * sizeof(T)));
        }
        k.out = dP.p; k.ksplit = ksplit; k.slice_stride = (size_t)M * N;
        ARP_TRY(launch_skinny_gemm(tcode, k, nullptr));
        ARP_TRY(launch_skinny_reduce_ln(tcode, dP.as<float>(), ksplit, (size_t)M * N, bias ? dB.as<float>() : nullptr, dR.as<float>(), N, dH.p, N,
                                        ln_w ? dLw.as<float>() : nullptr, ln_w ? dLb.as<float>() : nullptr, M, N, eps, nullptr));
        ARP_HIP_OK(hipDeviceSynchronize());
        ARP_TRY(from_dev<float>(out, (size_t)M * N, dR));
        if (ln_w) ARP_TRY(from_dev<T>(h_out, (size_t)M * N, dH));
        return 0;
    };
    const int rc = body();
    dA.release(); dW.release(); dB.release(); dR.release(); dO.release(); dP.release(); dH.release(); dLw.release(); dLb.release();
    return rc;
}

extern "C" {

int arp_op_skinny_gemm(int mode, int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N, int K, int ksplit,
                       const float* ln_w, const float* ln_b, float eps, float* h_out) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || ksplit < 0) return fail("bad argument");
    if (mode == ARP_MODE_BF16) return op_skinny<bf16_t>(act, A, W, bias, resid, out, M, N, K, ksplit, ln_w, ln_b, eps, h_out);
    if (mode == ARP_MODE_F16) return op_skinny<f16_t>(act, A, W, bias, resid, out, M, N, K, ksplit, ln_w, ln_b, eps, h_out);
    return fail("skinny gemm: 16-bit modes only");
}

int arp_op_gemm_nt(int mode, int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N,
                   int K) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0) return fail("bad argument");
    if (mode == ARP_MODE_F16) return op_gemm<f16_t>(act, A, W, bias, resid, out, M, N, K);
    return mode == ARP_MODE_BF16 ? op_gemm<bf16_t>(act, A, W, bias, resid, out, M, N, K) : op_gemm<float>(act, A, W, bias, resid, out, M, N, K);
}

}  // extern "C"

// e4m3 bits -> f32 on the host (test hook below)
static float host_fp82f(uint8_t b) {
    const int sign = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) v = NAN;
    else if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return sign ? -v : v;
}

extern "C" {

// Test hook for the fp8 instances of gemm256_nt_kernel: A [M,K] and W [N,K] are rounded to e4m3 on the host (pass values that are
// already representable for an exact check), out = act(alpha * A.W^T + bias) (+ resid) as f32, or (out_fp8) quantised to e4m3 after
// a multiplication by out_scale and returned decoded.
int arp_op_gemm_fp8(int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N, int K, float alpha,
                    int out_fp8, float out_scale) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 16) return fail("bad argument (K % 128, N % 16)");
    if (out_fp8 && resid) return fail("fp8 output has no residual epilogue");
    DevBuf dA, dW, dB, dR, dO;
    auto body = [&]() -> int {
        std::vector<fp8_t> qa((size_t)M * K), qw((size_t)N * K);
        for (size_t i = 0; i < qa.size(); ++i) qa[i] = host_f2fp8(A[i]);
        for (size_t i = 0; i < qw.size(); ++i) qw[i] = host_f2fp8(W[i]);
        ARP_TRY(dA.ensure(qa.size())); ARP_TRY(dW.ensure(qw.size()));
        ARP_HIP_OK(hipMemcpy(dA.p, qa.data(), qa.size(), hipMemcpyHostToDevice));
        ARP_HIP_OK(hipMemcpy(dW.p, qw.data(), qw.size(), hipMemcpyHostToDevice));
        if (bias) ARP_TRY(to_dev<float>(bias, N, dB));
        if (resid) ARP_TRY(to_dev<float>(resid, (size_t)M * N, dR));
        ARP_TRY(dO.ensure((size_t)M * N * 4));
        GemmArgs g;
        g.A = dA.p; g.W = dW.p; g.bias = bias ? dB.as<float>() : nullptr; g.resid = resid ? dR.as<float>() : nullptr; g.out = dO.p;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N; g.alpha = alpha; g.out_scale = out_scale;
        int rc;
        if (out_fp8 == 2) {  // f16 output (the in_proj of the fp8 attention projections)
            if (act != ACT_NONE || resid) return fail("f16 output: no activation, no residual");
            rc = launch_gemm256_nt<fp8_t, f16_t, ACT_NONE, false, SITE_OP>(g, nullptr);
            ARP_TRY(rc);
            ARP_HIP_OK(hipDeviceSynchronize());
            return from_dev<f16_t>(out, (size_t)M * N, dO);
        }
        if (out_fp8) rc = act == ACT_QGELU ? launch_gemm256_nt<fp8_t, fp8_t, ACT_QGELU, false, SITE_OP>(g, nullptr) : launch_gemm256_nt<fp8_t, fp8_t, ACT_NONE, false, SITE_OP>(g, nullptr);
        else if (act != ACT_NONE) return fail("f32 output: act must be ACT_NONE");
        else rc = resid ? launch_gemm256_nt<fp8_t, float, ACT_NONE, true, SITE_OP>(g, nullptr) : launch_gemm256_nt<fp8_t, float, ACT_NONE, false, SITE_OP>(g, nullptr);
        ARP_TRY(rc);
        ARP_HIP_OK(hipDeviceSynchronize());
        if (out_fp8) {
            std::vector<uint8_t> hb((size_t)M * N);
            ARP_HIP_OK(hipMemcpy(hb.data(), dO.p, hb.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < hb.size(); ++i) out[i] = host_fp82f(hb[i]);
            return 0;
        }
        return from_dev<float>(out, (size_t)M * N, dO);
    };
    const int rc = body();
    dA.release(); dW.release(); dB.release(); dR.release(); dO.release();
    return rc;
}

}  // extern "C"

template <typename T> static int op_gemm_bench(int kernel, int act, int resid, int out_f32, int M, int N, int K, int iters, float* avg_ms) {
    DevBuf dA, dW, dB, dR, dO;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto body = [&]() -> int {
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() * 0.05f;
        for (auto& v : hb) v = rnd();
        ARP_TRY(to_dev<T>(hA.data(), hA.size(), dA)); ARP_TRY(to_dev<T>(hW.data(), hW.size(), dW)); ARP_TRY(to_dev<float>(hb.data(), N, dB));
        ARP_TRY(dO.ensure((size_t)M * N * 4)); ARP_TRY(dR.ensure((size_t)M * N * 4));
        ARP_HIP_OK(hipMemset(dR.p, 0, (size_t)M * N * 4));
        GemmArgs g;
        g.A = dA.p; g.W = dW.p; g.bias = dB.as<float>(); g.resid = resid ? dR.as<float>() : nullptr; g.out = resid ? dR.p : dO.p;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
        if (const char* fe = getenv("ARP_GEMM_FLAGS")) g.flags = atoi(fe);
        if (const char* fe = getenv("ARP_GEMM_STAGGER")) sscanf(fe, "%d,%d", &g.stagger_groups, &g.stagger_cycles);
        if (const char* fe = getenv("ARP_GEMM_GROUP_M")) g.group_m = atoi(fe);
        auto run = [&]() -> int {
            if (resid) return launch_gemm_auto<T, float, ACT_NONE, true, SITE_OP>(g, nullptr, kernel);
            if (out_f32) return launch_gemm_auto<T, float, ACT_NONE, false, SITE_OP>(g, nullptr, kernel);
            if (act == ACT_QGELU) return launch_gemm_auto<T, T, ACT_QGELU, false, SITE_OP>(g, nullptr, kernel);
            return launch_gemm_auto<T, T, ACT_NONE, false, SITE_OP>(g, nullptr, kernel);
        };
        ARP_HIP_OK(hipEventCreate(&e0)); ARP_HIP_OK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) ARP_TRY(run());
        ARP_HIP_OK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ARP_TRY(run());
        ARP_HIP_OK(hipEventRecord(e1, nullptr));
        ARP_HIP_OK(hipEventSynchronize(e1));
        float ms = 0.f;
        ARP_HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / iters;
        return 0;
    };
    const int rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dA.release(); dW.release(); dB.release(); dR.release(); dO.release();
    return rc;
}

extern "C" {

int arp_op_gemm_bench(int mode, int kernel, int act, int resid, int out_f32, int M, int N, int K, int iters, float* avg_ms) {
    if (!avg_ms || M <= 0 || N <= 0 || K <= 0 || iters <= 0) return fail("bad argument");
    if (mode == ARP_MODE_F16) return op_gemm_bench<f16_t>(kernel, act, resid, out_f32, M, N, K, iters, avg_ms);
    return mode == ARP_MODE_BF16 ? op_gemm_bench<bf16_t>(kernel, act, resid, out_f32, M, N, K, iters, avg_ms)
                                 : op_gemm_bench<float>(kernel, act, resid, out_f32, M, N, K, iters, avg_ms);
}

int arp_op_layernorm(const float* x, const float* w, const float* b, float* out, int rows, int D, float eps) {
    if (!x || !w || !b || !out || rows <= 0 || D <= 0) return fail("bad argument");
    if (D % 4 || D > ROW_MAX_V4 * 256) return fail("layernorm: unsupported width");
    DevBuf dx, dw, db, dout;
    auto body = [&]() -> int {
        ARP_TRY(to_dev<float>(x, (size_t)rows * D, dx)); ARP_TRY(to_dev<float>(w, D, dw)); ARP_TRY(to_dev<float>(b, D, db));
        ARP_TRY(dout.ensure((size_t)rows * D * 4));
#define ARP_LN_CALL(NV)                                                                                                  \
    hipLaunchKernelGGL((layernorm_kernel<float, NV>), dim3((rows + 3) / 4), dim3(256), 0, nullptr, dx.as<float>(), (size_t)D, \
                       dout.as<float>(), D, dw.as<float>(), db.as<float>(), rows, D, eps)
        ARP_NV_DISPATCH(D, ARP_LN_CALL);
#undef ARP_LN_CALL
        ARP_HIP_OK(hipGetLastError());
        ARP_HIP_OK(hipDeviceSynchronize());
        return from_dev<float>(out, (size_t)rows * D, dout);
    };
    const int rc = body();
    dx.release(); dw.release(); db.release(); dout.release();
    return rc;
}

}  // extern "C"

template <typename T> static int op_attn(int impl, const float* qkv, float* out, int B, int N, int D, int heads, int causal) {
    DevBuf dq, dout;
    auto body = [&]() -> int {
        ARP_TRY(to_dev<T>(qkv, (size_t)B * N * 3 * D, dq));
        ARP_TRY(dout.ensure((size_t)B * N * D * sizeof(T)));
        ARP_TRY(launch_attention<T>(nullptr, impl, dq.as<T>(), dout.as<T>(), B, N, D, heads, causal));
        ARP_HIP_OK(hipDeviceSynchronize());
        return from_dev<T>(out, (size_t)B * N * D, dout);
    };
    const int rc = body();
    dq.release(); dout.release();
    return rc;
}

extern "C" {

int arp_op_attention(int mode, int impl, const float* qkv, float* out, int B, int N, int D, int heads, int causal) {
    if (!qkv || !out || B <= 0 || N <= 0 || D <= 0 || heads <= 0 || D % heads) return fail("bad argument");
    if (mode == ARP_MODE_F16) return op_attn<f16_t>(impl, qkv, out, B, N, D, heads, causal);
    return mode == ARP_MODE_BF16 ? op_attn<bf16_t>(impl, qkv, out, B, N, D, heads, causal)
                                 : op_attn<float>(impl, qkv, out, B, N, D, heads, causal);
}

}  // extern "C"

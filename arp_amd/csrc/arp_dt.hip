// Path (2): the ARP-DT policy train step on MI355X -- host orchestration + C ABI.
// Reference seam: create_train_step / train_step_fn, /root/reference/arp_dt/main_procgen.py:104-141
// (model: arp_dt/ARPDT.py:152-261,413-486; arp_dt/layers.py; optimizer: main_procgen.py:490-507).
//
// Boundary this round: the frozen M3AE encoder output `enc` [B,T,tokens,dim] is an
// INPUT (BASELINE.json configs[3]: "random-init M3AE encodings"); everything trainable is inside.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/arp_hip.h"
#include "attention.h"
#include "common.h"
#include "dtops.h"
#include "enc_internal.h"
#include "gemm.h"
#include "gemm256.h"
#include "adapter_bwd.h"
#include "gemm_tn.h"
#include "policy_fused.h"
#include "rccl_dl.h"
#include "runtime.h"

using namespace arp;

namespace {

enum { SITE_DT = 16 };

struct ParamInfo {
    std::string name;
    std::vector<int64_t> shape;  // Flax shape
    size_t off = 0, size = 0;
    bool dense = false;  // 2-D Dense kernel: device layout is [out, in] (transposed from Flax [in, out])
    int in = 0, out = 0;
};

inline int cdiv(size_t a, size_t b) { return (int)((a + b - 1) / b); }

}  // namespace

struct arp_dt {
    arp_dt_cfg cfg;
    hipStream_t stream = nullptr;
    std::mutex capture_mu;  // graph capture on the compute thread vs the uploader thread's HIP calls (fwd_bwd_graphed / upload_async)
    std::vector<ParamInfo> infos;
    std::map<std::string, int> index;
    size_t P = 0, n_decay = 0;  // total parameters; the first n_decay are the ndim > 1 ones (L2 penalty applies)
    DevBuf params, grads, mu, nu;
    long long step = 0;
    bool shadows_stale = true;
    // operand-type shadows of the big weights
    // Forward-layout operands of the three big Dense kernels: the f32 parameters themselves (f32 mode) or `mirror`, their
    // operand-type copy over the flat prefix [0, n_mirror) that the Adam kernel keeps current; the transposed ones (dX = dY.W)
    // are rebuilt from it once per step.
    DevBuf mirror, W2t, Wit;
    size_t n_mirror = 0;
    bool mirror_stale = true;  // the host wrote parameters (or a broadcast did): rebuild the mirror from f32
    const void* fwd_w(const std::string& n) {
        const size_t off = infos[index.at(n)].off;
        return cfg.mode == ARP_MODE_F32 ? (const void*)(params.as<float>() + off) : (const void*)(static_cast<char*>(mirror.p) + off * 2);
    }
    // (f16 mode, measured on the real geometry with tests/probes/policy_rounding_probe.py: every operand rounding of the adapter
    // path -- X, W1, H1, W2, A, Y, Wi -- costs 2.4-6.1e-4 on the logits and they combine to 4.9-8.4e-4 across seeds; with
    // res = sigmoid(4) = 0.98 the adapter branch carries the signal, so carrying Y and Wi as hi + lo pairs, tried, bought
    // nothing for +10 % step time and was removed.)
    // batch: TWO device-resident slots, so that batch i+1 can be uploaded (arp_dt_upload_batch_async, copy stream) while the step
    // on batch i runs -- the reference's prefetch_to_device(..., 2) (main_procgen.py:703).  `cur` is the slot the next forward /
    // step reads; arp_dt_set_batch writes it synchronously as before.
    int B = 0;
    struct BatchSlot {
        DevBuf enc32, action, rtg;
        DevBuf img32;             // raw frames [B*T, res, res, 3] f32 when the encoder is attached
        int B = 0;
        bool images = false;
        hipEvent_t up = nullptr;   // recorded on the copy stream behind the slot's upload
        hipEvent_t use = nullptr;  // recorded on the compute stream behind the last step that read the slot
        bool up_pending = false, used = false;
        hipEvent_t enc_done = nullptr;  // recorded on the encoder stream behind the encoder pass that filled enc32 from img32
        bool up_recorded = false;       // `up` has been recorded at least once (a slot the prefetcher has used)
        bool enc_ahead = false;         // enc32 holds (or will hold, behind enc_done) the encodings of the frames now in img32: the step does not encode again
    } bt[3];  // 0 / 1: the prefetcher's slots (arp_dt_upload_batch*_async); 2: the synchronous arp_dt_set_batch* calls -- a validation
              // step or a greedy action in between prefetched train steps must not write into a slot the uploader thread may be filling
    int cur = 2;
    // one copy stream per slot; an upload waits on the HOST for the slot's last reader (upload_async) and then runs on a stream with
    // nothing else queued, beside the step on the other slot
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    arp_enc* enc = nullptr;   // optional frozen encoder in front (row N1)
    bool use_images = false;
    // The encoder's part streams fork and join around ~150 long kernels: replayed as parallel branches of the step's hipGraph that costs more than it overlaps
    // (1.33 vs 0.99 ms for two short branches), and the encoder's launches are far longer than the host takes to issue them -- so with frames in
    // the encoder is enqueued EAGERLY in front of the (encoder-less) captured chain.  ARP_DT_ENC_EAGER=0 captures it with the rest, as rounds 1-5 did.
    bool enc_eager = true;
    bool enc_outside = false;  // this call's encoder pass has been enqueued ahead of the chain: forward<T> skips it
    // Round 6: the encoder is FROZEN (stop_gradient, ARPDT.py:418-462) -- batch i + 1's encodings depend on nothing step i computes.  Every eager encoder pass
    // runs on a stream of its own (its part streams fork from it), ordered behind the slot's upload and the slot's last reader, and the step waits for the
    // slot's enc_done event: arp_dt_encode_ahead(slot) lets batch i + 1 be encoded WHILE step i's policy part (19 short, partly HBM-bound launches, one of them
    // on 32 of 256 CUs) runs -- what prefetch_to_device's uploader thread calls once a batch of frames has landed.  All passes share the encoder's
    // workspace and are serialised on that one stream.
    hipStream_t enc_stream = nullptr;
    hipEvent_t ev_enc_go = nullptr;
    // activations (T = operand type)
    DevBuf Xb, XbT, H1, H1T, A, Y, YT, dY, dApre, dApreT, G, dH1T, dzb, dzT, part, scal;
    // f32 small tensors
    std::vector<DevBuf> xs, ln0, qkv, att, hmid, ln1, u, gl;  // per block
    DevBuf img, hf, a_in, r_in, ha, hr, logits, ret, metrics;
    DevBuf dlogits, dret, dha, dhr, da_in, dr_in, dhf, dh, t1, t2, t3, dws, dbs, dimg, dz, dqkv;
    // fused policy kernel (policy_fused.h): per-layer saved output gradients + the grouped-launch tables
    bool fused = false;
    std::vector<DevBuf> d_x1, d_u, d_mid, d_qkv, dws0, dbs0, dws1, dbs1;
    DevBuf dwsf, dbsf, dtok, loss_part, gtab, gprefix, ctab, cprefix;
    int n_gemm = 0, gemm_tiles = 0, n_cs = 0, cs_tiles = 0;
    PfArgs pf;
    DevBuf alibi;  // per-head slopes of config.alibi_bias (null pointer semantics: cfg.alibi_bias == 0 -> the kernels get nullptr / zeros)
    DevBuf pf_pack, pf_jobs;  // fragment-major weight copies of the fused kernel's big linears and their job table (policy_fused.h)
    int pf_njobs = 0, pf_pack_blocks = 0;
    // 16-bit modes: the fused kernel's big linears on (hi, lo) binary16 operand pairs (policy_fused.h::pf_lin_x3: f32-level products at 5.3x the f32 MFMA
    // rate); the f32 parity mode keeps v_mfma_f32_16x16x4_f32.  ARP_PF_X3=0/1 overrides either way (A/B, tests).
    bool pf_x3 = false;
    // f16 mode, optional (arp_dt_set_adapter_corrections / ARP_DT_ADAPTER_C=1): the adapter's two FORWARD products with their operand roundings corrected on the
    // scaled fp4 MFMA (ARP_MODE_F16C's product, gemm256 MIXC) and its output handed to the mix in f32 -- the policy's own share of the encoder-inside logit error
    // (the comment at fwd_w: X, W1, H1, W2, A) without the f32 MFMA.  The backward reads the same plain binary16 Xb / H1 / A as before.
    bool adapter_c = false;
    // Which corrections (round 6; ARP_DT_ADAPTER_PLAN = "<fc1><fc2><e|h|d>", default below): per product 1 = the WEIGHT rounding corrected (x4 . dW4: +K/256 K-tiles),
    // 2 = weight and ACTIVATION roundings (+ dx4 . W4: another K/256); the adapter output reaches the mix as e = f32 (fc2 on the f32 read-modify epilogue + a
    // binary16 copy for the backward), h = binary16 (the ordinary 16-bit epilogue, no f32 copy), d = binary16 + the e2m1 code of its rounding error (fc2's dx4 side
    // output into Adx, decoded by iti_x3_kernel's mix).  Chosen with scripts/adapter_plan_emulate.py (fp64 emulation of every rounding, 8 + 8 seeds) and measured on
    // the GPU: profiles/r6_adapter_plans.txt.
    int ac_plan1 = 2, ac_plan2 = 2;
    // 22d.  Per-seed maxima over 16 seeds of N(0,1) encodings / 8 seeds behind f32 encoder outputs / the seed of bench.py's own gate, and the policy step alone
    // (one box; after the x4 segment written by fc1's epilogue was repaired, gemm256.h -- rounds 5 and 6 measured every plan with fc2's weight correction as noise):
    //   plain 8.7e-4 / 1.18e-3 / 8.3e-4, 0.77 ms;  11h 1.18e-3 / 7.9e-4 / 8.9e-4, 0.85;  11d 9.3e-4 / 6.5e-4 / 5.6e-4, 0.86;  12h 7.1e-4 / 7.1e-4 / 6.7e-4, 0.87;
    //   21h 6.0e-4 / 7.8e-4;  12d 6.4e-4 / 6.0e-4;  22h 4.9e-4 / 3.5e-4 / 5.9e-4, 0.875;  22d 1.6e-4 / 2.1e-4 / 1.2e-4, 0.90 (22e: the same errors, 0.925)
    // Every one of the four corrections carries its share (fc1's activation rounding is the encodings' own); h leaves the hand-off's rounding, which is then the
    // largest term left; d takes it out to 2^-14 for 12.6 MB where e moves 50 MB more and puts fc2 on the f32 epilogue.
    bool ac_a_dx = true;
    DevBuf Adx;            // d: [Mx, D / 2] bytes
    bool ac_a_exact = false;
    bool ac_h1_inplace = true;  // the backward reads H1 out of the [hi | x4 | dx4] rows fc1 wrote (row stride 3 D / 2 halves) instead of a copy made by extract_hi_kernel
    const void* h1_ptr = nullptr;  // what the backward reads as H1 this step, and its row stride
    int h1_ld = 0;
    DevBuf Xc, H1c, A32, W1c, W2c, wc_scal;  // operand rows [hi | x4 | dx4]; packed weights [W_hi | dW4 | W4]; wc_scal: 16 ints (sd, sw of W1 / W2 at 4, 5 / 12, 13) + 2 x 32 per-block (max |dw|, max |w|) pairs
    ncclComm_t comm = nullptr;
    bool has_comm = false;
    // data-parallel step: gradient all-reduce in two buckets on a communication stream, bucket 1 (image_text_input's kernel, 94 % of
    // the bytes, + everything the transformer produced) launched while the adapter's backward GEMMs still run (step_impl)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_b1 = nullptr, ev_b2 = nullptr, ev_comm = nullptr;
    // Experiment (round 4, OFF; ARP_DT_SIDE=1 switches it on): in the single-rank backward of the adapter (16-bit TN path) the weight-gradient GEMMs that
    // nothing downstream waits for run on a SIDE stream beside the kernels that carry the dependent chain -- dWi beside the fused dY kernel (two streaming
    // kernels at 3.4 - 3.9 TB/s each), dW2 (216 long workgroups) beside dApre . W2 (387 tiles = one and a half rounds of the chip); forked and joined by
    // events, inside the captured graph as well.  MEASURED SLOWER (profiles/r4_side.txt, interleaved): step 0.862 -> 0.907 ms -- side by side dW2 takes
    // 158 us instead of 56 (its K-slices-per-XCD walk wants the chip), dApre . W2 99 instead of 62, the fused dY kernel 94 instead of 72: these kernels
    // are each sized for 256 CUs and lose more to each other than their tails were worth.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_dapre = nullptr, ev_side = nullptr;
    bool side_gemms = false;
    bool mix_x16 = true;      // iti_x3_kernel's mix reads the encodings' operand-type copy, not the f32 encodings (ARP_DT_MIX_X16=0; dtops.h: 16-seed logits 8.73e-4 -> 8.74e-4)
    bool dy_x16 = true;       // adapter_dy_kernel reads the encodings' operand-type copy for d loss / d res (ARP_DT_DY_X16=0: the f32 encodings, rounds 2-5)
    bool merge_small = true;  // the step's small dependent launches merged (ARP_DT_MERGE=0: one launch each, rounds 2-5)
    bool dzb_from_pf = false;
    bool pack_early = false;
    bool defer_w2t = false, w2t_pending = false, packed_in_prologue = false;  // forward<T>'s merged prologue launch (dt_prologue_kernel)
    bool dwi_last = true;   // backward_adapter_tn: image_text_input's weight gradient last (Infinity Cache residency for the norm pass)
    bool adam_rev = true;   // apply_update: the update walks the flat state from its end (what the norm pass touched last)
    DevBuf part_side;
    bool overlap_comm = true;   // ARP_DT_OVERLAP=0: the serial form (one all-reduce after the whole backward), for A/B and the bit-identity test
    bool force_comm = false;    // ARP_DT_FORCE_COMM=1: run the all-reduce path at world = 1 too (what a 1-GPU box can test)
    bool grads_summed = false;  // the gradient buffer holds the SUM over ranks (set by a data-parallel step)
    // forward + backward captured once per (batch slot, stage, geometry) and replayed; stage 0 = the whole chain, 1 / 2 = the two
    // halves either side of the point where bucket 1 is complete
    bool use_graph = true;
    struct GraphRec {
        hipGraphExec_t exec = nullptr;
        int B = 0, images = -1, eager = 0;
    } graphs[3][4];  // [batch slot][stage: 0 whole step, 1 / 2 the two halves of the overlapped step, 3 forward only]
    Profiler prof;

    size_t esz() const { return cfg.mode == ARP_MODE_F32 ? 4 : 2; }
    // Backward activations of the adapter path (dz -> dY -> dApre -> G, all stored in the operand type) are ~1e-7 at B = 32:
    // below binary16's normal range.  In f16 mode they carry a power-of-two scale (exact), removed again where a gradient
    // leaves for the f32 gradient buffer (GEMM alpha / split-K reduce / row sums).  bf16 and f32 have the range: scale 1.
    float act_scale() const { return cfg.mode == ARP_MODE_F16 ? 16384.f : 1.f; }
    // 16-bit modes with 128-aligned widths: weight gradients on the TN kernel (gemm_tn.h) straight from the row-major operands --
    // no transposed, K-padded copies.  Other geometries (and the f32 parity mode) keep the transposed-copy path.
    bool use_tn() const {
        static const bool off = getenv("ARP_DT_TN") && atoi(getenv("ARP_DT_TN")) == 0;
        return !off && cfg.mode != ARP_MODE_F32 && cfg.enc_dim % 128 == 0 && cfg.emb % 128 == 0;
    }
    // ... and, with the adapter on, the gradient through image_text_input fused with the adapter's first mask (adapter_bwd.h)
    bool use_fused_dy() const {
        static const bool off = getenv("ARP_DT_FUSED_DY") && atoi(getenv("ARP_DT_FUSED_DY")) == 0;
        return !off && use_tn() && cfg.use_adapter && adapter_dy_supported(cfg.emb, cfg.enc_dim, (long long)cfg.enc_tokens * cfg.enc_dim);
    }
    // ARP_DT_FUSE_RELU_BWD: 0 = never, 1 (default) = where gemm256 is the kernel that would run anyway (>= 192 tiles), 2 = whenever
    // the shape allows (the parity tests' way to reach the masked epilogue at B = 2); read when the handle is created
    int relu_fuse_mode = 1;
    bool fuse_relu_bwd(long tiles256) const { return relu_fuse_mode == 2 || (relu_fuse_mode == 1 && tiles256 >= 192); }
    // 16-bit modes (default ON, ARP_DT_ITI_F32=0 switches it off): image_text_input's FORWARD contraction (K = 197 376, 6.5 GF) on f32
    // operands -- the un-rounded mix Y and the f32 master weights on the f32-input MFMA -- instead of their 16-bit copies.  Takes two
    // of the adapter path's seven operand roundings (Y, Wi) out of the logits: over 16 seeds at the real geometry the f16 logits /
    // return error goes from max 1.05e-3 (2 seeds of 16 outside north_star's 1e-3) to max 8.7e-4 (none).  Costs the Y32 write and a
    // 1/16-rate GEMM: 0.866 -> 0.926 ms per step at B = 32 (+7 %).  The backward is unchanged.
    bool iti_f32 = false;
    bool iti_mix = true;  // ... with the adapter's mix formed inside that kernel's operand load (ARP_DT_ITI_MIX=0: adapter_mix_kernel + a 101 MB f32 copy, rounds 3-5)
    bool iti_x3 = true;  // ... and that f32-level product on (hi, lo) binary16 MFMA pairs instead of the f32 MFMA (ARP_DT_ITI_X3=0: round 3's f32-MFMA GEMM)
    DevBuf Y32;
    DevBuf colpart;  // column partial sums of mask_copy_colsum_kernel
    DevBuf colpart0;   // the dH1 GEMM's column partials when the small reductions are deferred (backward_adapter_tn)
    DevBuf dres_part;  // per-workgroup d loss / d res partials of adapter_dy_kernel
    int R() const { return B * cfg.window; }
    int L() const { return 3 * cfg.window; }
    float* p(const std::string& n) { return params.as<float>() + infos[index.at(n)].off; }
    float* g(const std::string& n) { return grads.as<float>() + infos[index.at(n)].off; }
};

namespace {

int add_param(arp_dt* c, std::vector<ParamInfo>& v, const std::string& name, std::vector<int64_t> shape) {
    ParamInfo pi;
    pi.name = name;
    pi.shape = shape;
    pi.size = 1;
    for (auto d : shape) pi.size *= (size_t)d;
    const bool is_kernel = name.size() >= 7 && name.compare(name.size() - 7, 7, "/kernel") == 0;
    if (shape.size() == 2 && is_kernel) {
        pi.dense = true;
        pi.in = (int)shape[0];
        pi.out = (int)shape[1];
    }
    v.push_back(pi);
    return 0;
}

void build_layout(arp_dt* c) {
    const arp_dt_cfg& k = c->cfg;
    const int E = k.emb, D = k.enc_dim, H = k.mlp_ratio * k.emb;
    std::vector<ParamInfo> v;
    if (k.use_adapter) {
        for (int i = 0; i < 2; ++i) {
            add_param(c, v, "AdapterMLP_0/Dense_" + std::to_string(i) + "/kernel", {D, D});
            add_param(c, v, "AdapterMLP_0/Dense_" + std::to_string(i) + "/bias", {D});
        }
        add_param(c, v, "residual_weight", {1});
    }
    add_param(c, v, "image_text_input/kernel", {(int64_t)k.enc_tokens * D, E});
    add_param(c, v, "image_text_input/bias", {E});
    add_param(c, v, "action_input/embedding", {k.n_actions, E});
    add_param(c, v, "rtg_input/kernel", {1, E});
    for (int i = 0; i < k.depth; ++i) {
        const std::string p = "policy/Block_" + std::to_string(i) + "/";
        for (const char* ln : {"LayerNorm_0", "LayerNorm_1"}) {
            add_param(c, v, p + ln + "/scale", {E});
            add_param(c, v, p + ln + "/bias", {E});
        }
        add_param(c, v, p + "Attention_0/Dense_0/kernel", {E, 3 * E});
        add_param(c, v, p + "Attention_0/Dense_0/bias", {3 * E});
        add_param(c, v, p + "Attention_0/Dense_1/kernel", {E, E});
        add_param(c, v, p + "Attention_0/Dense_1/bias", {E});
        add_param(c, v, p + "FeedForward_0/fc1/kernel", {E, H});
        add_param(c, v, p + "FeedForward_0/fc2/kernel", {H, E});
    }
    add_param(c, v, "policy/LayerNorm_0/scale", {E});
    add_param(c, v, "policy/LayerNorm_0/bias", {E});
    for (auto hn : {std::make_pair(std::string("action_outputs_0"), k.n_actions), std::make_pair(std::string("return_outputs_0"), 1)}) {
        add_param(c, v, hn.first + "/layers_0/kernel", {E, E});
        add_param(c, v, hn.first + "/layers_0/bias", {E});
        add_param(c, v, hn.first + "/layers_2/kernel", {E, hn.second});
    }
    // flat order: every ndim > 1 parameter first (the explicit L2 term of main_procgen.py:114-117 covers
    // exactly those), then the vectors; every offset a multiple of 4 floats
    size_t off = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (auto& pi : v)
            if ((pi.shape.size() > 1) == (pass == 0)) {
                pi.off = off;
                off += (pi.size + 3) & ~(size_t)3;
                if (pass == 0) c->n_decay = off;
            }
    c->P = off;
    c->infos = v;
    for (size_t i = 0; i < v.size(); ++i) c->index[v[i].name] = (int)i;
}

template <typename T> int small_attention_fwd(arp_dt* c, const float* qkv, float* out, int B, int L, int E, int heads) {
    const int hd = E / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t lds = (size_t)2 * L * hd * 4;
    const int threads = 64;
#define ARP_DT_ATT(HD)                                                                                                   \
    hipLaunchKernelGGL((attn_valu_kernel<float, HD>), dim3(B * heads), dim3(threads), lds, c->stream, qkv, out, L, E, heads, scale, 1, L, \
                       c->cfg.alibi_bias ? c->alibi.as<float>() : nullptr)
    if (hd == 16) ARP_DT_ATT(16);
    else if (hd == 32) ARP_DT_ATT(32);
    else if (hd == 64) ARP_DT_ATT(64);
    else return fail("policy attention: unsupported head_dim " + std::to_string(hd));
#undef ARP_DT_ATT
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

int sgemm(arp_dt* c, const float* A, int ta, const float* B, int tb, const float* bias, const float* resid, float* C, int M, int N, int K,
          int lda, int ldb, int act = ACT_NONE, int accumulate = 0) {
    SmallGemm g{A, B, bias, resid, C, M, N, K, lda, ldb, N, ta, tb, act, accumulate};
    hipLaunchKernelGGL(small_gemm_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, c->stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}
// Y[M,N] = act(X[M,K] . W[N,K]^T + b)   (W in device layout [out, in])
int linear_fwd(arp_dt* c, const float* X, const float* W, const float* b, const float* resid, float* Y, int M, int N, int K, int act = ACT_NONE) {
    return sgemm(c, X, 0, W, 1, b, resid, Y, M, N, K, K, K, act);
}
// dW[N,K] = dY[M,N]^T . X[M,K];  db[N] = colsum(dY);  dX[M,K] = dY . W
int linear_bwd(arp_dt* c, const float* X, const float* W, const float* dY, float* dW, float* db, float* dX, int M, int N, int K) {
    ARP_TRY(sgemm(c, dY, 1, X, 0, nullptr, nullptr, dW, N, K, M, N, K));
    if (db) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(256), 0, c->stream, dY, M, N, db);
        ARP_HIP_OK(hipGetLastError());
    }
    if (dX) ARP_TRY(sgemm(c, dY, 0, W, 0, nullptr, nullptr, dX, M, K, N, N, K));
    return 0;
}
int ln_fwd(arp_dt* c, const float* x, const float* w, const float* b, float* y, int rows, int D) {
    hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, c->stream, x, w, b, y, rows, D, 1e-6f);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}
int ln_bwd(arp_dt* c, const float* x, const float* w, const float* dy, float* dx, int accumulate, float* dscale, float* dbias, int rows, int D) {
    hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, c->stream, x, w, dy, dx, accumulate, c->dws.as<float>(),
                       c->dbs.as<float>(), rows, D, 1e-6f);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->dws.as<float>(), rows, D, dscale);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->dbs.as<float>(), rows, D, dbias);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}
int ew_bwd(arp_dt* c, const float* gr, const float* ref, float* out, size_t n, int op) {
    hipLaunchKernelGGL(ew_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, gr, ref, out, n, op);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// big NT GEMM on the MFMA kernels; OutT in {T, float}
template <typename T, typename OutT, int ACT>
int big_gemm(arp_dt* c, const char* site, const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldo, int M, int N,
             int K, float alpha = 1.f) {
    GemmArgs g;
    g.alpha = alpha;
    g.A = A; g.W = W; g.bias = bias; g.resid = nullptr; g.out = out;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldr = ldo; g.ldo = ldo;
    ProfScope ps(c->prof, c->stream, site);
    // (the two-workgroups-per-CU kernel, which wins on the ViT's out_proj, measured 2-10 % slower here on the adapter's
    //  768 x 768 x 32 896 GEMMs: bias/ReLU epilogues are cheap, nothing for a second workgroup to hide)
    return launch_gemm_auto<T, OutT, ACT, false, SITE_DT>(g, c->stream, 0);
}
// split-K NT GEMM: f32 partials [S][M][N] then a fixed-order reduce (+bias, act) into OutT
template <typename T, typename OutT>
int splitk_gemm(arp_dt* c, const char* site, const void* A, int lda, const void* W, int ldw, const float* bias, int act, OutT* out, int M, int N,
                int K, float alpha = 1.f) {
    constexpr int EPB = 128 / (int)sizeof(T);
    const int nk = K / EPB;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    static const int wg_target = getenv("ARP_SPLITK_WGS") ? atoi(getenv("ARP_SPLITK_WGS")) : 512;  // one resident round (2 WG/CU x 256 CUs); measured best of 256..2048
    int S = std::max(1, std::min(nk, wg_target / std::max(tiles, 1)));
    const int per = (nk + S - 1) / S;
    S = (nk + per - 1) / per;  // every slice non-empty
    ARP_TRY(c->part.ensure((size_t)S * M * N * 4));
    GemmArgs g;
    g.A = A; g.W = W; g.bias = nullptr; g.resid = nullptr; g.out = c->part.p;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldr = N; g.ldo = N;
    g.ksplit = S; g.slice_stride = (size_t)M * N;
    ProfScope ps(c->prof, c->stream, site);
    if (S == 1) g.ksplit = 1;
    ARP_TRY((launch_gemm_nt<T, float, ACT_NONE, false, SITE_DT + 1>(g, c->stream)));
    const size_t MN = (size_t)M * N;
    launch_splitk_reduce<OutT>(c->stream, c->part.as<float>(), S, MN, N, bias, act, out, nullptr, 0, alpha);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename TI, typename TM, typename TO>
int transpose_mask(arp_dt* c, const TI* in, int ldi, const TM* mask, const float* scale_ptr, float scale, TO* outN, int ldn, TO* outT, int ldt,
                   int Rr, int Cc) {
    hipLaunchKernelGGL((transpose_mask_kernel<TI, TM, TO>), dim3(cdiv(Cc, 64), cdiv(Rr, 64)), dim3(256), 0, c->stream, in, ldi, mask, scale_ptr,
                       scale, outN, ldn, outT, ldt, Rr, Cc);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// The step's three independent preparations in one launch (16-bit modes, fused transformer): the encodings' f32 -> operand-type conversion (a 151 MB stream),
// the fused kernel's fragment-major weight copies (pf_pack_kernel's jobs) and the transposed shadow of the adapter's second kernel.  They were three
// dependent launches; none reads another's output, and the two small ones disappear under the stream.
struct DtPrologueArgs {
    const float* enc; void* xb; size_t n8; int conv_blocks;
    const PfPackJob* jobs; int pack_blocks, njobs;
    const void* W2; void* W2t; int D;  // W2t == nullptr: no transpose in this launch
};
template <typename T> static __global__ __launch_bounds__(256) void dt_prologue_kernel(DtPrologueArgs a) {
    int b = (int)blockIdx.x;
    const int npack = a.pack_blocks * a.njobs;
    if (b < npack) { pf_pack_block(a.jobs, b % a.pack_blocks, a.pack_blocks, b / a.pack_blocks); return; }
    b -= npack;
    const int tb = (a.D + 63) / 64, ntr = a.W2t ? tb * tb : 0;
    if (b < ntr) {
        transpose_mask_tile<T, T, T>(static_cast<const T*>(a.W2), a.D, nullptr, nullptr, 1.f, nullptr, 0, static_cast<T*>(a.W2t), a.D, a.D, a.D, b % tb, b / tb);
        return;
    }
    convert8_block<T>(a.enc, static_cast<T*>(a.xb), a.n8, b - ntr, a.conv_blocks);
}

template <typename T> int refresh_shadows(arp_dt* c) {
    if (!c->shadows_stale) return 0;
    const arp_dt_cfg& k = c->cfg;
    const int D = k.enc_dim, E = k.emb;
    const size_t Kin = (size_t)k.enc_tokens * D;
    ProfScope ps(c->prof, c->stream, "dt.refresh_shadows");
    if constexpr (sizeof(T) == 2) {
        if (c->mirror_stale) {
            hipLaunchKernelGGL((convert_kernel<T>), dim3(cdiv(c->n_mirror, 1024)), dim3(256), 0, c->stream, c->params.as<float>(), c->mirror.as<T>(), c->n_mirror);
            ARP_HIP_OK(hipGetLastError());
            c->mirror_stale = false;
        }
    }
    // device layout of a Dense kernel is [out, in]; the transposed shadows come from the operand-type copy (half the bytes)
    const T* W2 = static_cast<const T*>(c->fwd_w("AdapterMLP_0/Dense_1/kernel"));
    const T* Wi = static_cast<const T*>(c->fwd_w("image_text_input/kernel"));
    if (k.use_adapter) {
        if (c->defer_w2t) c->w2t_pending = true;  // forward<T>'s merged prologue launch makes it (the backward is its only reader)
        else ARP_TRY((transpose_mask<T, T, T>(c, W2, D, nullptr, nullptr, 1.f, nullptr, 0, c->W2t.as<T>(), D, D, D)));
    }
    if constexpr (__is_same(T, f16_t)) {
        if (k.use_adapter && c->adapter_c) {  // [W_hi | dW4 | W4] of the adapter's two kernels from the f32 parameters, scales chosen on the device
            ARP_TRY(c->W1c.ensure((size_t)D * 3 * D + 512)); ARP_TRY(c->W2c.ensure((size_t)D * 3 * D + 512));
            ARP_TRY(c->wc_scal.ensure(1024));  // 16 ints of scales (ints 4, 5 / 12, 13: sd, sw of W1 / W2) + 2 x 32 (max |dw|, max |w|) pairs
            unsigned int* mx = c->wc_scal.as<unsigned int>();
            hipLaunchKernelGGL(wc_absmax2_kernel, dim3(32, 2), dim3(256), 0, c->stream, c->p("AdapterMLP_0/Dense_0/kernel"), c->p("AdapterMLP_0/Dense_1/kernel"), (size_t)D * D, mx);
            hipLaunchKernelGGL(wc_pack2_kernel, dim3(cdiv((size_t)D * D, 1024), 2), dim3(256), 0, c->stream, c->p("AdapterMLP_0/Dense_0/kernel"), c->p("AdapterMLP_0/Dense_1/kernel"), D, D, mx, 32,
                               c->W1c.as<f16_t>(), c->W2c.as<f16_t>());
            ARP_HIP_OK(hipGetLastError());
        }
    }
    // the fused dY kernel reads Wi as it lies; only the unfused path wants the [Kin, E] copy
    if (!c->use_fused_dy()) ARP_TRY((transpose_mask<T, T, T>(c, Wi, (int)Kin, nullptr, nullptr, 1.f, nullptr, 0, c->Wit.as<T>(), E, E, (int)Kin)));
    c->shadows_stale = false;
    return 0;
}

// The fused kernel covers the shipped geometry family: up to 16 tokens per sample, widths in MFMA-tile multiples.
bool fused_eligible(const arp_dt_cfg& k) {
    if (const char* e = getenv("ARP_DT_FUSED"))
        if (atoi(e) == 0) return false;
    const int E = k.emb, H = k.mlp_ratio * k.emb;
    // instantiated geometries: the shipped one (E = 128, mlp_ratio 4) and the half-width one the tests use
    const bool shape = (E == 128 && H == 512) || (E == 64 && H == 256);
    return shape && 3 * k.window <= 16 && k.n_actions <= 16 && k.depth <= PF_MAX_DEPTH && E % k.heads == 0 && (E / k.heads) % 16 == 0 &&
           pf_lds_bytes(E, H, k.heads, k.depth) <= 160 * 1024;
}

// kernel arguments of policy_fused_kernel and the problem tables of the two grouped gradient launches
// _get_attention_slopes (arp_dt/layers.py:97-110): the ALiBi head slopes -- 2^(-8 i / n) for a power-of-two head count n, otherwise the
// slopes of the next lower power of two followed by every other slope of the next higher one.
std::vector<float> alibi_slopes(int n) {
    auto pow2 = [](int m) {
        std::vector<float> v;
        const double start = std::pow(2.0, -std::pow(2.0, -(std::log2((double)m) - 3.0)));
        double x = start;
        for (int i = 0; i < m; ++i, x *= start) v.push_back((float)x);
        return v;
    };
    if (n <= 0) return {};
    if ((n & (n - 1)) == 0) return pow2(n);
    int c = 1;
    while (c * 2 <= n) c *= 2;
    std::vector<float> v = pow2(c);
    const std::vector<float> w = alibi_slopes(2 * c);
    for (int i = 0; i < (int)w.size() && (int)v.size() < n; i += 2) v.push_back(w[i]);
    return v;
}

int build_fused_plan(arp_dt* c) {
    const arp_dt_cfg& k = c->cfg;
    const int E = k.emb, H = k.mlp_ratio * E, T = k.window, L = 3 * T, NA = k.n_actions, depth = k.depth;
    const int R = c->B * T, BL = c->B * L;
    PfArgs& a = c->pf;
    memset(&a, 0, sizeof(a));
    a.T = T; a.L = L; a.E = E; a.H = H; a.heads = k.heads; a.NA = NA; a.depth = depth; a.do_bwd = 1; a.R = R; a.lambda = k.lambda_ret;
    {
        const std::vector<float> sl = alibi_slopes(k.heads);
        for (int h = 0; h < 16; ++h) a.alibi[h] = (k.alibi_bias && h < k.heads) ? sl[h] : 0.f;
    }
    a.img = c->img.as<float>(); a.rtg = c->bt[c->cur].rtg.as<float>(); a.action = c->bt[c->cur].action.as<int>();
    a.Wr = c->p("rtg_input/kernel"); a.emb = c->p("action_input/embedding");
    std::vector<SmallGemm> gj;
    std::vector<ColSumJob> cj;
    // dW[Nout, Kin] = dY[rows, Nout]^T . X[rows, Kin]
    auto add_dw = [&](const float* dY, const float* X, float* dW, int Nout, int Kin, int rows) {
        gj.push_back(SmallGemm{dY, X, nullptr, nullptr, dW, Nout, Kin, rows, Nout, Kin, Kin, 1, 0, ACT_NONE, 0});
    };
    auto add_cs = [&](const float* in, float* out, int rows, int C) { cj.push_back(ColSumJob{in, out, rows, C}); };
    for (int i = 0; i < depth; ++i) {
        const std::string p = "policy/Block_" + std::to_string(i) + "/";
        PfBlk& b = a.blk[i];
        b.ln0w = c->p(p + "LayerNorm_0/scale"); b.ln0b = c->p(p + "LayerNorm_0/bias");
        b.wqkv = c->p(p + "Attention_0/Dense_0/kernel"); b.bqkv = c->p(p + "Attention_0/Dense_0/bias");
        b.wo = c->p(p + "Attention_0/Dense_1/kernel"); b.bo = c->p(p + "Attention_0/Dense_1/bias");
        b.ln1w = c->p(p + "LayerNorm_1/scale"); b.ln1b = c->p(p + "LayerNorm_1/bias");
        b.wfc1 = c->p(p + "FeedForward_0/fc1/kernel"); b.wfc2 = c->p(p + "FeedForward_0/fc2/kernel");
        b.x = c->xs[i].as<float>(); b.ln0 = c->ln0[i].as<float>(); b.qkv = c->qkv[i].as<float>(); b.att = c->att[i].as<float>();
        b.hmid = c->hmid[i].as<float>(); b.ln1 = c->ln1[i].as<float>(); b.u = c->u[i].as<float>(); b.gl = c->gl[i].as<float>();
        b.d_x1 = c->d_x1[i].as<float>(); b.d_u = c->d_u[i].as<float>(); b.d_mid = c->d_mid[i].as<float>(); b.d_qkv = c->d_qkv[i].as<float>();
        b.dws0 = c->dws0[i].as<float>(); b.dbs0 = c->dbs0[i].as<float>(); b.dws1 = c->dws1[i].as<float>(); b.dbs1 = c->dbs1[i].as<float>();
        add_dw(b.d_qkv, b.ln0, c->g(p + "Attention_0/Dense_0/kernel"), 3 * E, E, BL);
        add_dw(b.d_mid, b.att, c->g(p + "Attention_0/Dense_1/kernel"), E, E, BL);
        add_dw(b.d_u, b.ln1, c->g(p + "FeedForward_0/fc1/kernel"), H, E, BL);
        add_dw(b.d_x1, b.gl, c->g(p + "FeedForward_0/fc2/kernel"), E, H, BL);
        add_cs(b.d_qkv, c->g(p + "Attention_0/Dense_0/bias"), BL, 3 * E);
        add_cs(b.d_mid, c->g(p + "Attention_0/Dense_1/bias"), BL, E);
        add_cs(b.dws0, c->g(p + "LayerNorm_0/scale"), BL, E); add_cs(b.dbs0, c->g(p + "LayerNorm_0/bias"), BL, E);
        add_cs(b.dws1, c->g(p + "LayerNorm_1/scale"), BL, E); add_cs(b.dbs1, c->g(p + "LayerNorm_1/bias"), BL, E);
    }
    a.lnfw = c->p("policy/LayerNorm_0/scale"); a.lnfb = c->p("policy/LayerNorm_0/bias");
    a.wa0 = c->p("action_outputs_0/layers_0/kernel"); a.ba0 = c->p("action_outputs_0/layers_0/bias"); a.wa2 = c->p("action_outputs_0/layers_2/kernel");
    a.wr0 = c->p("return_outputs_0/layers_0/kernel"); a.br0 = c->p("return_outputs_0/layers_0/bias"); a.wr2 = c->p("return_outputs_0/layers_2/kernel");
    {
        // fragment-major copies (nt for the forward, nn for the backward) of every weight pf_lin_nt / pf_lin_nn stream; filled by
        // pf_pack_kernel at the head of every launch of the fused kernel (the parameters change every step)
        std::vector<PfPackJob> jobs;
        size_t total = 0;
        const int x3 = c->pf_x3 ? 1 : 0;
        auto want = [&](const float* W, int N, int K) { jobs.push_back(PfPackJob{W, nullptr, nullptr, N, K, x3}); total += 2 * (size_t)N * K; };
        for (int i = 0; i < depth; ++i) {
            want(a.blk[i].wqkv, 3 * E, E); want(a.blk[i].wo, E, E); want(a.blk[i].wfc1, H, E); want(a.blk[i].wfc2, E, H);
        }
        want(a.wa0, E, E); want(a.wr0, E, E);
        ARP_TRY(c->pf_pack.ensure(total * 4));
        float* base = c->pf_pack.as<float>();
        int maxq = 0;
        for (auto& jb : jobs) {
            jb.nt = base; base += (size_t)jb.N * jb.K;
            jb.nn = base; base += (size_t)jb.N * jb.K;
            maxq = std::max(maxq, jb.N * jb.K / (x3 ? 8 : 4));
        }
        for (int i = 0; i < depth; ++i) {
            PfBlk& b = a.blk[i];
            b.wqkv_nt = jobs[4 * i].nt; b.wqkv_nn = jobs[4 * i].nn; b.wo_nt = jobs[4 * i + 1].nt; b.wo_nn = jobs[4 * i + 1].nn;
            b.wfc1_nt = jobs[4 * i + 2].nt; b.wfc1_nn = jobs[4 * i + 2].nn; b.wfc2_nt = jobs[4 * i + 3].nt; b.wfc2_nn = jobs[4 * i + 3].nn;
        }
        a.wa0_nt = jobs[4 * depth].nt; a.wa0_nn = jobs[4 * depth].nn; a.wr0_nt = jobs[4 * depth + 1].nt; a.wr0_nn = jobs[4 * depth + 1].nn;
        ARP_TRY(c->pf_jobs.ensure(jobs.size() * sizeof(PfPackJob)));
        ARP_HIP_OK(hipMemcpy(c->pf_jobs.p, jobs.data(), jobs.size() * sizeof(PfPackJob), hipMemcpyHostToDevice));
        c->pf_njobs = (int)jobs.size();
        c->pf_pack_blocks = std::min((maxq + 255) / 256, 32);
    }
    a.xf = c->xs[depth].as<float>(); a.a_in = c->a_in.as<float>(); a.r_in = c->r_in.as<float>(); a.ha = c->ha.as<float>(); a.hr = c->hr.as<float>();
    a.logits = c->logits.as<float>(); a.ret = c->ret.as<float>(); a.dlogits = c->dlogits.as<float>(); a.dret = c->dret.as<float>();
    a.dha = c->dha.as<float>(); a.dhr = c->dhr.as<float>(); a.dwsf = c->dwsf.as<float>(); a.dbsf = c->dbsf.as<float>();
    a.dtok = c->dtok.as<float>(); a.dz = c->dz.as<float>(); a.loss_part = c->loss_part.as<float>();
    add_dw(a.dlogits, a.ha, c->g("action_outputs_0/layers_2/kernel"), NA, E, R);
    add_dw(a.dha, a.a_in, c->g("action_outputs_0/layers_0/kernel"), E, E, R);
    add_dw(a.dret, a.hr, c->g("return_outputs_0/layers_2/kernel"), 1, E, R);
    add_dw(a.dhr, a.r_in, c->g("return_outputs_0/layers_0/kernel"), E, E, R);
    add_cs(a.dwsf, c->g("policy/LayerNorm_0/scale"), BL, E); add_cs(a.dbsf, c->g("policy/LayerNorm_0/bias"), BL, E);
    add_cs(a.dha, c->g("action_outputs_0/layers_0/bias"), R, E); add_cs(a.dhr, c->g("return_outputs_0/layers_0/bias"), R, E);
    add_cs(a.dz, c->g("image_text_input/bias"), R, E);
    std::vector<int> gp(1, 0), cp(1, 0);
    for (auto& g : gj) gp.push_back(gp.back() + cdiv(g.M, 32) * cdiv(g.N, 32));
    for (auto& j : cj) cp.push_back(cp.back() + cdiv(j.C, 64));
    c->n_gemm = (int)gj.size(); c->gemm_tiles = gp.back(); c->n_cs = (int)cj.size(); c->cs_tiles = cp.back();
    ARP_TRY(c->gtab.ensure(gj.size() * sizeof(SmallGemm))); ARP_TRY(c->gprefix.ensure(gp.size() * 4));
    ARP_TRY(c->ctab.ensure(cj.size() * sizeof(ColSumJob))); ARP_TRY(c->cprefix.ensure(cp.size() * 4));
    ARP_HIP_OK(hipMemcpy(c->gtab.p, gj.data(), gj.size() * sizeof(SmallGemm), hipMemcpyHostToDevice));
    ARP_HIP_OK(hipMemcpy(c->gprefix.p, gp.data(), gp.size() * 4, hipMemcpyHostToDevice));
    ARP_HIP_OK(hipMemcpy(c->ctab.p, cj.data(), cj.size() * sizeof(ColSumJob), hipMemcpyHostToDevice));
    ARP_HIP_OK(hipMemcpy(c->cprefix.p, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(policy_fused_kernel<128, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(policy_fused_kernel<64, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(policy_fused_kernel<128, 512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(policy_fused_kernel<64, 256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    return 0;
}

int ensure_buffers(arp_dt* c, int B) {
    if (B == c->B) return 0;
    const arp_dt_cfg& k = c->cfg;
    const size_t e = c->esz();
    const int E = k.emb, D = k.enc_dim, H = k.mlp_ratio * E, T = k.window, NA = k.n_actions;
    const size_t R = (size_t)B * T, Mx = R * k.enc_tokens, BL = R * 3, Kin = (size_t)k.enc_tokens * D;
    const size_t Mxp = (Mx + 63) / 64 * 64, Rp = (R + 63) / 64 * 64;
    // (the batch slots size their own buffers: stage_slot)
    // TN path: these are GEMM operands whose contraction index is the ROW -- rows up to the next multiple of 64 must read as zeros
    const size_t Rp64 = (R + 63) / 64 * 64;
    const size_t rowpad = std::max(Mxp * (size_t)D, Rp64 * Kin);
    DevBuf* tb[] = {&c->Xb, &c->H1, &c->A, &c->Y, &c->dY, &c->dApre, &c->G};
    for (auto* b : tb) {
        ARP_TRY(b->ensure(rowpad * e));
        ARP_HIP_OK(hipMemsetAsync(b->p, 0, rowpad * e, c->stream));
    }
    ARP_TRY(c->colpart.ensure((Mxp / 64) * (size_t)D * 4));
    if (c->iti_f32 && k.use_adapter) ARP_TRY(c->Y32.ensure(Mx * D * 4));

    DevBuf* tt[] = {&c->XbT, &c->H1T, &c->dApreT, &c->dH1T};
    for (auto* b : tt) {
        ARP_TRY(b->ensure((size_t)D * Mxp * e));
        ARP_HIP_OK(hipMemsetAsync(b->p, 0, (size_t)D * Mxp * e, c->stream));  // K padding must read as zeros
    }
    ARP_TRY(c->YT.ensure(Kin * Rp * e)); ARP_HIP_OK(hipMemsetAsync(c->YT.p, 0, Kin * Rp * e, c->stream));
    ARP_TRY(c->dzT.ensure((size_t)E * Rp * e)); ARP_HIP_OK(hipMemsetAsync(c->dzT.p, 0, (size_t)E * Rp * e, c->stream));
    ARP_TRY(c->dzb.ensure(Rp64 * E * e)); ARP_HIP_OK(hipMemsetAsync(c->dzb.p, 0, Rp64 * E * e, c->stream));
    ARP_TRY(c->scal.ensure((4096 + (size_t)cdiv(D, 256) * (Mxp / 64 + 1)) * 4));
    auto f32 = [&](DevBuf& b, size_t n) { return b.ensure(std::max<size_t>(n, 4) * 4); };
    c->xs.resize(k.depth + 1); c->ln0.resize(k.depth); c->qkv.resize(k.depth); c->att.resize(k.depth); c->hmid.resize(k.depth);
    c->ln1.resize(k.depth); c->u.resize(k.depth); c->gl.resize(k.depth);
    for (int i = 0; i <= k.depth; ++i) ARP_TRY(f32(c->xs[i], BL * E));
    for (int i = 0; i < k.depth; ++i) {
        ARP_TRY(f32(c->ln0[i], BL * E)); ARP_TRY(f32(c->qkv[i], BL * 3 * E)); ARP_TRY(f32(c->att[i], BL * E)); ARP_TRY(f32(c->hmid[i], BL * E));
        ARP_TRY(f32(c->ln1[i], BL * E)); ARP_TRY(f32(c->u[i], BL * H)); ARP_TRY(f32(c->gl[i], BL * H));
    }
    ARP_TRY(f32(c->img, R * E)); ARP_TRY(f32(c->hf, BL * E)); ARP_TRY(f32(c->a_in, R * E)); ARP_TRY(f32(c->r_in, R * E));
    ARP_TRY(f32(c->ha, R * E)); ARP_TRY(f32(c->hr, R * E)); ARP_TRY(f32(c->logits, R * NA)); ARP_TRY(f32(c->ret, R)); ARP_TRY(f32(c->metrics, 16));
    ARP_TRY(f32(c->dlogits, R * NA)); ARP_TRY(f32(c->dret, R)); ARP_TRY(f32(c->dha, R * E)); ARP_TRY(f32(c->dhr, R * E));
    ARP_TRY(f32(c->da_in, R * E)); ARP_TRY(f32(c->dr_in, R * E)); ARP_TRY(f32(c->dhf, BL * E)); ARP_TRY(f32(c->dh, BL * E));
    ARP_TRY(f32(c->t1, BL * std::max(H, 3 * E))); ARP_TRY(f32(c->t2, BL * std::max(H, 3 * E))); ARP_TRY(f32(c->t3, BL * E));
    ARP_TRY(f32(c->dws, BL * E)); ARP_TRY(f32(c->dbs, BL * E)); ARP_TRY(f32(c->dimg, R * E)); ARP_TRY(f32(c->dz, R * E));
    ARP_TRY(f32(c->dqkv, BL * 3 * E));
    c->B = B;
    c->fused = fused_eligible(k);
    if (c->fused) {
        for (auto* v : {&c->d_x1, &c->d_u, &c->d_mid, &c->d_qkv, &c->dws0, &c->dbs0, &c->dws1, &c->dbs1}) v->resize(k.depth);
        for (int i = 0; i < k.depth; ++i) {
            ARP_TRY(f32(c->d_x1[i], BL * E)); ARP_TRY(f32(c->d_u[i], BL * H)); ARP_TRY(f32(c->d_mid[i], BL * E)); ARP_TRY(f32(c->d_qkv[i], BL * 3 * E));
            ARP_TRY(f32(c->dws0[i], BL * E)); ARP_TRY(f32(c->dbs0[i], BL * E)); ARP_TRY(f32(c->dws1[i], BL * E)); ARP_TRY(f32(c->dbs1[i], BL * E));
        }
        ARP_TRY(f32(c->dwsf, BL * E)); ARP_TRY(f32(c->dbsf, BL * E)); ARP_TRY(f32(c->dtok, BL * E)); ARP_TRY(f32(c->loss_part, (size_t)B * 4));
        ARP_TRY(build_fused_plan(c));
    }
    return 0;
}

// tokens -> transformer -> heads -> losses (and, with do_bwd, the activation gradients) in one launch
int policy_fused(arp_dt* c, bool do_bwd) {
    const arp_dt_cfg& k = c->cfg;
    c->pf.do_bwd = do_bwd ? 1 : 0;
    c->pf.rtg = c->bt[c->cur].rtg.as<float>();  // the CURRENT batch slot's labels (the plan was built when the geometry last changed)
    c->pf.action = c->bt[c->cur].action.as<int>();
    // the scaled operand-type copy of dz (the TN backward's first operand) straight from the kernel instead of a transpose_mask launch behind it
    const bool dzb_here = do_bwd && c->merge_small && c->use_tn() && k.mode != ARP_MODE_F32 && c->dzb.p;
    c->pf.dzb = dzb_here ? c->dzb.p : nullptr;
    c->pf.dz_scale = c->act_scale();
    c->pf.dzb_f16 = k.mode == ARP_MODE_F16 ? 1 : 0;
    c->dzb_from_pf = dzb_here;
    const size_t lds = pf_lds_bytes(k.emb, k.mlp_ratio * k.emb, k.heads, k.depth);
    if (!c->packed_in_prologue)  // (forward<T>'s merged prologue launch packed them already)
        hipLaunchKernelGGL(pf_pack_kernel, dim3(c->pf_pack_blocks, c->pf_njobs), dim3(256), 0, c->stream, static_cast<const PfPackJob*>(c->pf_jobs.p));
    c->packed_in_prologue = false;
    if (k.emb == 128 && c->pf_x3) hipLaunchKernelGGL((policy_fused_kernel<128, 512, true>), dim3(c->B), dim3(PF_THREADS), lds, c->stream, c->pf);
    else if (k.emb == 128) hipLaunchKernelGGL((policy_fused_kernel<128, 512>), dim3(c->B), dim3(PF_THREADS), lds, c->stream, c->pf);
    else if (c->pf_x3) hipLaunchKernelGGL((policy_fused_kernel<64, 256, true>), dim3(c->B), dim3(PF_THREADS), lds, c->stream, c->pf);
    else hipLaunchKernelGGL((policy_fused_kernel<64, 256>), dim3(c->B), dim3(PF_THREADS), lds, c->stream, c->pf);
    // (with the backward behind it and the merged gradient launch on, that launch reduces the losses: backward<T>)
    if (!(do_bwd && c->merge_small))
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, c->stream, c->loss_part.as<float>(), c->B, c->R(), k.n_actions, k.lambda_ret,
                           c->metrics.as<float>());
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// The encoder pass of batch slot `slot` on the encoder stream; after_compute: ordered behind everything on the compute stream (a batch staged synchronously
// there, and the earlier steps' reads of this slot's enc32); otherwise behind the slot's upload and its last reader.  Caller holds capture_mu.
int enqueue_encode(arp_dt* c, int slot, bool after_compute) {
    arp_dt::BatchSlot& b = c->bt[slot];
    if (!c->enc || !b.images || b.B <= 0) return fail("encode: the slot holds no frames (or no encoder is attached)");
    if (!c->enc_stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->enc_stream, hipStreamNonBlocking));
    if (!c->ev_enc_go) ARP_HIP_OK(hipEventCreateWithFlags(&c->ev_enc_go, hipEventDisableTiming));
    if (!b.enc_done) ARP_HIP_OK(hipEventCreateWithFlags(&b.enc_done, hipEventDisableTiming));
    if (after_compute) {
        ARP_HIP_OK(hipEventRecord(c->ev_enc_go, c->stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->enc_stream, c->ev_enc_go, 0));
    } else {
        if (b.up_recorded) ARP_HIP_OK(hipStreamWaitEvent(c->enc_stream, b.up, 0));
        if (b.used) ARP_HIP_OK(hipStreamWaitEvent(c->enc_stream, b.use, 0));
    }
    ARP_TRY(enc_forward_on(c->enc, c->enc_stream, b.img32.as<float>(), b.B * c->cfg.window, b.enc32.as<float>()));
    ARP_HIP_OK(hipEventRecord(b.enc_done, c->enc_stream));
    b.enc_ahead = true;
    return 0;
}
// frames -> encodings for the CURRENT slot, ahead of the step's own launches on the compute stream
int encode_current(arp_dt* c) {
    if (!c->enc_eager)  // rounds 1-5: on the step's own stream (ARP_DT_ENC_EAGER=0: inside the captured chain)
        return enc_forward_on(c->enc, c->stream, c->bt[c->cur].img32.as<float>(), c->R(), c->bt[c->cur].enc32.as<float>());
    arp_dt::BatchSlot& b = c->bt[c->cur];
    if (!b.enc_ahead) {
        std::lock_guard<std::mutex> lock(c->capture_mu);
        ARP_TRY(enqueue_encode(c, c->cur, true));
    }
    ARP_HIP_OK(hipStreamWaitEvent(c->stream, b.enc_done, 0));
    b.enc_ahead = false;
    return 0;
}

// ---- forward: everything up to the losses; leaves every activation the backward needs -----------------
template <typename T> int forward(arp_dt* c, bool with_bwd = false) {
    const arp_dt_cfg& k = c->cfg;
    const int E = k.emb, D = k.enc_dim, H = k.mlp_ratio * E, NA = k.n_actions, depth = k.depth;
    const int R = c->R(), L = c->L(), BL = c->B * L;
    const size_t Mx = (size_t)R * k.enc_tokens;
    const int Kin = k.enc_tokens * D;
    const int Mxp = (int)((Mx + 63) / 64 * 64);
    const bool adapter_cpath = __is_same(T, f16_t) && k.use_adapter && c->adapter_c && D % 256 == 0 && D >= 512;
    // 16-bit modes with the fused transformer: conversion + weight packing + W2's transposed shadow in ONE launch (dt_prologue_kernel)
    const bool prologue = sizeof(T) == 2 && c->merge_small && c->fused && !(adapter_cpath && c->use_tn()) && !(k.use_adapter && !c->use_tn()) && (Mx * D) % 8 == 0;
    c->defer_w2t = prologue;
    ARP_TRY(refresh_shadows<T>(c));
    c->defer_w2t = false;
    if (c->use_images && !c->enc_outside) {  // frozen M3AE encoder under stop_gradient (arp_dt/ARPDT.py:418-462): frames -> encodings
        ARP_TRY(encode_current(c));
    }
    {   // enc f32 -> operand type, both layouts (the transposed one feeds the weight-gradient GEMM)
        ProfScope ps(c->prof, c->stream, "dt.enc_convert");
        if (adapter_cpath && c->use_tn()) {
            // (convert_f16c_kernel below writes Xb beside the [hi | x4 | dx4] rows in one pass over the encodings)
        } else if (k.use_adapter && !c->use_tn()) {
            ARP_TRY((transpose_mask<float, float, T>(c, c->bt[c->cur].enc32.as<float>(), D, nullptr, nullptr, 1.f, c->Xb.as<T>(), D, c->XbT.as<T>(), Mxp, (int)Mx, D)));
        } else if constexpr (sizeof(T) == 2) {  // no transposed copy wanted: a flat 16-byte-per-lane conversion
            const size_t n = Mx * D;
            if (prologue) {
                DtPrologueArgs a;
                a.enc = c->bt[c->cur].enc32.as<float>(); a.xb = c->Xb.p; a.n8 = n / 8; a.conv_blocks = (int)std::min<size_t>(cdiv(n / 8, 256), 4096);
                // (the weight packing rides along only on request, ARP_DT_PACK_EARLY=1: packed 200 us ahead of the fused kernel the copies have left L2 by the time
                //  it streams them -- policy_fused_kernel 113 -> 122 us, more than the launch saved; profiles/r5_policy_ab.txt run 13)
                a.jobs = static_cast<const PfPackJob*>(c->pf_jobs.p); a.pack_blocks = c->pf_pack_blocks; a.njobs = c->pack_early ? c->pf_njobs : 0;
                a.W2 = c->fwd_w("AdapterMLP_0/Dense_1/kernel"); a.W2t = c->w2t_pending ? c->W2t.p : nullptr; a.D = D;
                const int tb = cdiv(D, 64);
                hipLaunchKernelGGL((dt_prologue_kernel<T>), dim3(a.pack_blocks * a.njobs + (a.W2t ? tb * tb : 0) + a.conv_blocks), dim3(256), 0, c->stream, a);
                c->w2t_pending = false;
                c->packed_in_prologue = c->pack_early;
            } else if (n % 8 == 0) hipLaunchKernelGGL((convert8_kernel<T>), dim3((unsigned)std::min<size_t>(cdiv(n / 8, 256), 4096)), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), c->Xb.as<T>(), n / 8);
            else hipLaunchKernelGGL((convert_kernel<T>), dim3(cdiv(n, 1024)), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), c->Xb.as<T>(), n);
            ARP_HIP_OK(hipGetLastError());
        } else {
            ARP_TRY((transpose_mask<float, float, T>(c, c->bt[c->cur].enc32.as<float>(), D, nullptr, nullptr, 1.f, c->Xb.as<T>(), D, nullptr, Mxp, (int)Mx, D)));
        }
    }
    if (c->w2t_pending) {  // (deferred by refresh_shadows, and this call's conversion did not go through the merged launch after all)
        ARP_TRY((transpose_mask<T, T, T>(c, static_cast<const T*>(c->fwd_w("AdapterMLP_0/Dense_1/kernel")), D, nullptr, nullptr, 1.f, nullptr, 0, c->W2t.as<T>(), D, D, D)));
        c->w2t_pending = false;
    }
    const T* Yp = c->Xb.as<T>();
    bool adapter_done = false;
    c->h1_ptr = c->H1.p; c->h1_ld = D;
    // the adapter's mix inside image_text_input's operand load (16-bit modes with the f32-level (hi, lo) product): no mix launch, no f32 copy of the mix
    const bool fuse_mix = sizeof(T) == 2 && k.use_adapter && c->iti_f32 && c->iti_x3 && c->iti_mix && Kin % 64 == 0 && E % 4 == 0;
    const float* mix_a32 = nullptr;
    const T* mix_a = nullptr;
    const bool a_exact = c->ac_a_exact || (c->ac_a_dx && !fuse_mix);  // (the e2m1 hand-off exists inside image_text_input's operand load only)
    if constexpr (__is_same(T, f16_t)) {
        if (adapter_cpath) {
            ARP_TRY(c->Xc.ensure(Mx * 3 * D + 4096));
            {   // the hidden rows double as the backward's H1 operand (a TN contraction over the ROWS: rows up to the next multiple of 64 must read as zeros)
                const void* before = c->H1c.p;
                ARP_TRY(c->H1c.ensure((size_t)Mxp * 3 * D + 4096));
                if (c->H1c.p != before) ARP_HIP_OK(hipMemsetAsync(c->H1c.p, 0, (size_t)Mxp * 3 * D + 4096, c->stream));
            }
            if (a_exact) ARP_TRY(c->A32.ensure(Mx * D * 4));
            // relu(relu(x W1 + b1) W2 + b2) with the operand roundings of the two products corrected (gemm256 MIXC) as ac_plan1 / ac_plan2 say: x -> [hi | x4 (| dx4)]
            // rows, fc1 writes the hidden rows [hi | x4 (| dx4)] itself (x4 from the rounded tile, dx4 from the accumulators), fc2 writes the output in f32 for the
            // mix + its binary16 copy for the backward (ac_a_exact) or in binary16 only; the plain binary16 Xb the backward reads comes out of the conversion pass.
            {
                ProfScope ps(c->prof, c->stream, "dt.enc_convert");
                if (D % 16 == 0) {
                    if (c->ac_plan1 >= 2) hipLaunchKernelGGL((convert_f16c16_kernel<true>), dim3(cdiv(Mx * D, 4096)), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), c->Xb.as<f16_t>(), c->Xc.as<f16_t>(), Mx, D);
                    else hipLaunchKernelGGL((convert_f16c16_kernel<false>), dim3(cdiv(Mx * D, 4096)), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), c->Xb.as<f16_t>(), c->Xc.as<f16_t>(), Mx, D);
                } else {
                    hipLaunchKernelGGL(convert_f16c_kernel, dim3(cdiv(Mx * D, 1024)), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), c->Xb.as<f16_t>(), c->Xc.as<f16_t>(), Mx, D);
                }
                ARP_HIP_OK(hipGetLastError());
            }
            const int* sc = reinterpret_cast<const int*>(c->wc_scal.p);
            auto mixc = [&](GemmArgs& g, const void* A, const void* W, const float* bias, const int* sptr, int plan) {
                g.A = A; g.W = W; g.bias = bias; g.M = (int)Mx; g.N = D;
                g.lda = D + D / 2; g.ldw = D + D / 2; g.ldr = D;
                g.mix_nk16 = D / 64; g.mix_nkc_a = D / 256; g.K = D + plan * D / 4; g.mix_sptr = sptr;
            };
            {
                GemmArgs g;
                mixc(g, c->Xc.p, c->W1c.p, c->p("AdapterMLP_0/Dense_0/bias"), sc + 4, c->ac_plan1);
                g.out = c->H1c.p; g.ldo = D + D / 2;
                g.xb_out = static_cast<char*>(c->H1c.p) + 2 * (size_t)D; g.ldxb = 3 * D; g.x8_shift = F16C_X_SHIFT;
                g.dx4_out = c->ac_plan2 >= 2 ? static_cast<char*>(c->H1c.p) + 2 * (size_t)D + D / 2 : nullptr;
                ProfScope ps(c->prof, c->stream, "dt.adapter_fc1");
                ARP_TRY((launch_gemm256_nt<f16_t, f16_t, ACT_RELU, false, SITE_DT, false, 1, true>(g, c->stream)));
            }
            // (in place only where the backward's ReLU mask rides in dApre . W2's epilogue -- a row stride there -- and not in mask_copy_colsum_kernel, which walks a
            //  contiguous H1: small geometries, ARP_DT_FUSE_RELU_BWD=0)
            if (c->ac_h1_inplace && c->use_tn() && D % 8 == 0 && c->fuse_relu_bwd((long)cdiv((int)Mx, 256) * cdiv(D, 256))) {
                c->h1_ptr = c->H1c.p; c->h1_ld = D + D / 2;
            } else {
                ProfScope ps(c->prof, c->stream, "dt.adapter_fc1");
                hipLaunchKernelGGL(extract_hi_kernel, dim3(cdiv(Mx * D, 2048)), dim3(256), 0, c->stream, c->H1c.as<f16_t>(), D + D / 2, c->H1.as<f16_t>(), Mx, D);
                ARP_HIP_OK(hipGetLastError());
            }
            {
                GemmArgs g;
                mixc(g, c->H1c.p, c->W2c.p, c->p("AdapterMLP_0/Dense_1/bias"), sc + 12, c->ac_plan2);
                ProfScope ps(c->prof, c->stream, "dt.adapter_fc2");
                if (a_exact) {
                    g.out = c->A32.p; g.ldo = D;
                    g.xb_out = c->A.p; g.ldxb = D;
                    ARP_TRY((launch_gemm256_nt<f16_t, float, ACT_RELU, false, SITE_DT, false, 1, true>(g, c->stream)));
                } else {
                    g.out = c->A.p; g.ldo = D;
                    g.x8_shift = -1;  // (no x4 side output: nothing multiplies the adapter's output on the fp4 MFMA)
                    if (c->ac_a_dx && fuse_mix) {  // ... but the e2m1 code of its rounding error goes to the mix inside image_text_input's operand load
                        ARP_TRY(c->Adx.ensure(Mx * (size_t)D / 2 + 256));
                        g.dx4_out = c->Adx.p; g.ldxb = D / 2;
                    }
                    ARP_TRY((launch_gemm256_nt<f16_t, f16_t, ACT_RELU, false, SITE_DT, false, 1, true>(g, c->stream)));
                }
            }
            if (fuse_mix && a_exact) {
                mix_a32 = c->A32.as<float>();  // the mix happens inside image_text_input's operand load (dtops.h::iti_x3_kernel)
            } else if (fuse_mix) {
                mix_a = c->A.as<T>();
            } else if (a_exact) {
                ProfScope ps(c->prof, c->stream, "dt.adapter_mix");
                hipLaunchKernelGGL((adapter_mix_kernel<T, float>), dim3(cdiv(Mx * D, 1024)), dim3(256), 0, c->stream, c->A32.as<float>(), c->bt[c->cur].enc32.as<float>(),
                                   c->p("residual_weight"), c->Y.as<T>(), Mx * D, c->iti_f32 ? c->Y32.as<float>() : nullptr);
                ARP_HIP_OK(hipGetLastError());
            } else {
                ProfScope ps(c->prof, c->stream, "dt.adapter_mix");
                hipLaunchKernelGGL((adapter_mix_kernel<T>), dim3(cdiv(Mx * D, 1024)), dim3(256), 0, c->stream, c->A.as<T>(), c->bt[c->cur].enc32.as<float>(),
                                   c->p("residual_weight"), c->Y.as<T>(), Mx * D, c->iti_f32 ? c->Y32.as<float>() : nullptr);
                ARP_HIP_OK(hipGetLastError());
            }
            Yp = c->Y.as<T>();
            adapter_done = true;
        }
    }
    if (k.use_adapter && !adapter_done) {
        // AdapterMLP: relu(relu(x W1 + b1) W2 + b2)   (arp_dt/models/adapter/layers.py:19-30)
        ARP_TRY((big_gemm<T, T, ACT_RELU>(c, "dt.adapter_fc1", c->Xb.p, D, c->fwd_w("AdapterMLP_0/Dense_0/kernel"), D, c->p("AdapterMLP_0/Dense_0/bias"), c->H1.p, D, (int)Mx, D, D)));
        ARP_TRY((big_gemm<T, T, ACT_RELU>(c, "dt.adapter_fc2", c->H1.p, D, c->fwd_w("AdapterMLP_0/Dense_1/kernel"), D, c->p("AdapterMLP_0/Dense_1/bias"), c->A.p, D, (int)Mx, D, D)));
        // (the mix as a second output of fc2's epilogue measured 0.083 ms against 0.050 + 0.035 ms for the two launches: the f32 x rows
        //  arrive behind the tile instead of beside it)
        if (fuse_mix) {
            mix_a = c->A.as<T>();
        } else {
            ProfScope ps(c->prof, c->stream, "dt.adapter_mix");
            hipLaunchKernelGGL((adapter_mix_kernel<T>), dim3(cdiv(Mx * D, 1024)), dim3(256), 0, c->stream, c->A.as<T>(), c->bt[c->cur].enc32.as<float>(),
                               c->p("residual_weight"), c->Y.as<T>(), Mx * D, c->iti_f32 ? c->Y32.as<float>() : nullptr);
            ARP_HIP_OK(hipGetLastError());
        }
        Yp = c->Y.as<T>();
    }
    if (sizeof(T) == 2 && c->iti_f32 && c->iti_x3 && Kin % 64 == 0 && E % 4 == 0) {
        // the same f32-level contraction on (hi, lo) binary16 pairs split in flight (dtops.h::iti_x3_kernel): bound by the 202 MB stream, not by the f32 matrix rate
        const float* Y32 = k.use_adapter ? c->Y32.as<float>() : c->bt[c->cur].enc32.as<float>();
        const int tiles = cdiv(R, 128) * cdiv(E, 128);
        const int nk = (int)(Kin / 64);
        static const int iti_wgs = getenv("ARP_DT_ITI_WGS") ? atoi(getenv("ARP_DT_ITI_WGS")) : 256;
        int S = std::max(1, std::min(nk, iti_wgs / std::max(tiles, 1)));  // workgroups on the chip (74 KB of LDS and 174 registers each: two fit a CU)
        const int per = (nk + S - 1) / S;
        S = (nk + per - 1) / per;
        ARP_TRY(c->part.ensure((size_t)S * R * E * 4));
        ProfScope ps(c->prof, c->stream, "dt.image_text_input");
        static const int ahead = [] { const char* e = getenv("ARP_DT_ITI_AHEAD"); return e && atoi(e) == 2 ? 2 : 1; }();
        // (ARP_DT_ITI_CYCLIC=1: K-tiles dealt round-robin instead of one K range per workgroup -- measured 2 us slower, profiles/r5_policy_ab.txt: the stream is not short of DRAM locality)
        static const bool cyclic = [] { const char* e = getenv("ARP_DT_ITI_CYCLIC"); return e && atoi(e) != 0; }();
        const int kslice = cyclic ? 0 : per * 64;
        const float* Wi = c->p("image_text_input/kernel");
        if constexpr (sizeof(T) == 2) {
            if (mix_a32) {
                hipLaunchKernelGGL((iti_x3_kernel<1, float, T>), dim3(S, tiles), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin,
                                   kslice, mix_a32, c->p("residual_weight"), c->Y.as<T>(), (const T*)nullptr);
            } else if (mix_a && adapter_cpath && c->ac_a_dx && c->mix_x16) {  // binary16 adapter output + the e2m1 code of its rounding error
                hipLaunchKernelGGL((iti_x3_kernel<1, T, T, true, true>), dim3(S, tiles), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin,
                                   kslice, mix_a, c->p("residual_weight"), c->Y.as<T>(), c->Xb.as<T>(), c->Adx.as<uint8_t>());
            } else if (mix_a && adapter_cpath && c->ac_a_dx) {
                hipLaunchKernelGGL((iti_x3_kernel<1, T, T, false, true>), dim3(S, tiles), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin,
                                   kslice, mix_a, c->p("residual_weight"), c->Y.as<T>(), (const T*)nullptr, c->Adx.as<uint8_t>());
            } else if (mix_a && c->mix_x16) {
                hipLaunchKernelGGL((iti_x3_kernel<1, T, T, true>), dim3(S, tiles), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin,
                                   kslice, mix_a, c->p("residual_weight"), c->Y.as<T>(), c->Xb.as<T>());
            } else if (mix_a) {
                hipLaunchKernelGGL((iti_x3_kernel<1, T, T>), dim3(S, tiles), dim3(256), 0, c->stream, c->bt[c->cur].enc32.as<float>(), Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin,
                                   kslice, mix_a, c->p("residual_weight"), c->Y.as<T>(), (const T*)nullptr);
            } else if (ahead == 2) {
                hipLaunchKernelGGL(iti_x3_kernel<2>, dim3(S, tiles), dim3(256), 0, c->stream, Y32, Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin, kslice, (const iti_nomix_t*)nullptr, (const float*)nullptr, (f16_t*)nullptr, (const f16_t*)nullptr);
            } else {
                hipLaunchKernelGGL(iti_x3_kernel<1>, dim3(S, tiles), dim3(256), 0, c->stream, Y32, Kin, Wi, Kin, c->part.as<float>(), R, E, (int)Kin, kslice, (const iti_nomix_t*)nullptr, (const float*)nullptr, (f16_t*)nullptr, (const f16_t*)nullptr);
            }
        }
        ARP_HIP_OK(hipGetLastError());
        launch_splitk_reduce<float>(c->stream, c->part.as<float>(), S, (size_t)R * E, E, c->p("image_text_input/bias"), ACT_TANH, c->img.as<float>());
        ARP_HIP_OK(hipGetLastError());
    } else if (sizeof(T) == 2 && c->iti_f32) {
        const float* Y32 = k.use_adapter ? c->Y32.as<float>() : c->bt[c->cur].enc32.as<float>();
        ARP_TRY((splitk_gemm<float, float>(c, "dt.image_text_input", Y32, Kin, c->p("image_text_input/kernel"), Kin, c->p("image_text_input/bias"), ACT_TANH,
                                           c->img.as<float>(), R, E, Kin)));
    } else
    // image_text_input + tanh (arp_dt/ARPDT.py:475-484): [R, tokens*dim] x [tokens*dim, E], split over K
    ARP_TRY((splitk_gemm<T, float>(c, "dt.image_text_input", Yp, Kin, c->fwd_w("image_text_input/kernel"), Kin, c->p("image_text_input/bias"), ACT_TANH, c->img.as<float>(), R, E, Kin)));
    if (c->fused) {
        ProfScope ps(c->prof, c->stream, "dt.policy_fwd");
        ARP_TRY(policy_fused(c, with_bwd));
    } else {
        ProfScope ps(c->prof, c->stream, "dt.policy_fwd");
        hipLaunchKernelGGL(tokens_fwd_kernel, dim3(cdiv((size_t)R * E, 256)), dim3(256), 0, c->stream, c->img.as<float>(), c->bt[c->cur].rtg.as<float>(),
                           c->bt[c->cur].action.as<int>(), c->p("rtg_input/kernel"), c->p("action_input/embedding"), c->xs[0].as<float>(), R, E);
        ARP_HIP_OK(hipGetLastError());
        for (int i = 0; i < depth; ++i) {
            const std::string p = "policy/Block_" + std::to_string(i) + "/";
            float* x = c->xs[i].as<float>();
            ARP_TRY(ln_fwd(c, x, c->p(p + "LayerNorm_0/scale"), c->p(p + "LayerNorm_0/bias"), c->ln0[i].as<float>(), BL, E));
            ARP_TRY(linear_fwd(c, c->ln0[i].as<float>(), c->p(p + "Attention_0/Dense_0/kernel"), c->p(p + "Attention_0/Dense_0/bias"), nullptr,
                               c->qkv[i].as<float>(), BL, 3 * E, E));
            ARP_TRY(small_attention_fwd<float>(c, c->qkv[i].as<float>(), c->att[i].as<float>(), c->B, L, E, k.heads));
            ARP_TRY(linear_fwd(c, c->att[i].as<float>(), c->p(p + "Attention_0/Dense_1/kernel"), c->p(p + "Attention_0/Dense_1/bias"), x,
                               c->hmid[i].as<float>(), BL, E, E));
            ARP_TRY(ln_fwd(c, c->hmid[i].as<float>(), c->p(p + "LayerNorm_1/scale"), c->p(p + "LayerNorm_1/bias"), c->ln1[i].as<float>(), BL, E));
            ARP_TRY(linear_fwd(c, c->ln1[i].as<float>(), c->p(p + "FeedForward_0/fc1/kernel"), nullptr, nullptr, c->u[i].as<float>(), BL, H, E));
            hipLaunchKernelGGL(gelu_fwd_kernel, dim3(cdiv((size_t)BL * H, 256)), dim3(256), 0, c->stream, c->u[i].as<float>(), c->gl[i].as<float>(),
                               (size_t)BL * H);
            ARP_TRY(linear_fwd(c, c->gl[i].as<float>(), c->p(p + "FeedForward_0/fc2/kernel"), nullptr, c->hmid[i].as<float>(),
                               c->xs[i + 1].as<float>(), BL, E, H));
        }
        ARP_TRY(ln_fwd(c, c->xs[depth].as<float>(), c->p("policy/LayerNorm_0/scale"), c->p("policy/LayerNorm_0/bias"), c->hf.as<float>(), BL, E));
        hipLaunchKernelGGL(heads_gather_kernel, dim3(cdiv((size_t)R * E, 256)), dim3(256), 0, c->stream, c->hf.as<float>(), c->a_in.as<float>(),
                           c->r_in.as<float>(), R, E);
        ARP_TRY(linear_fwd(c, c->a_in.as<float>(), c->p("action_outputs_0/layers_0/kernel"), c->p("action_outputs_0/layers_0/bias"), nullptr,
                           c->ha.as<float>(), R, E, E, ACT_RELU));
        ARP_TRY(linear_fwd(c, c->ha.as<float>(), c->p("action_outputs_0/layers_2/kernel"), nullptr, nullptr, c->logits.as<float>(), R, NA, E));
        ARP_TRY(linear_fwd(c, c->r_in.as<float>(), c->p("return_outputs_0/layers_0/kernel"), c->p("return_outputs_0/layers_0/bias"), nullptr,
                           c->hr.as<float>(), R, E, E, ACT_RELU));
        ARP_TRY(linear_fwd(c, c->hr.as<float>(), c->p("return_outputs_0/layers_2/kernel"), nullptr, nullptr, c->ret.as<float>(), R, 1, E));
        hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, c->stream, c->logits.as<float>(), c->ret.as<float>(), c->bt[c->cur].action.as<int>(),
                           c->bt[c->cur].rtg.as<float>(), R, NA, k.lambda_ret, c->metrics.as<float>(), c->dlogits.as<float>(), c->dret.as<float>());
        ARP_HIP_OK(hipGetLastError());
    }
    return 0;
}

// ---- adapter + image_text_input backward, 16-bit modes, TN weight-gradient GEMMs (gemm_tn.h) ---------------------------------
// Same math as the tail of backward<T>() below; the operands of the three weight-gradient contractions stay row-major
// (no transposed K-padded copies): dWi = dz^T Y, dW2 = dApre^T H1, dW1 = dH1^T X.
template <typename T>
int tn_gemm(arp_dt* c, const char* site, const T* A, int lda, const T* B, int ldb, float* out, int M, int N, int K, float alpha, bool side = false) {
    hipStream_t st = side ? c->side_stream : c->stream;
    DevBuf& part = side ? c->part_side : c->part;  // (a side-stream GEMM keeps its split-K slabs to itself)
    const int tcode = __is_same(T, bf16_t) ? 1 : 2;
    GemmTnArgs g;
    int S;
    // a long contraction into a few 256 x 256 tiles (the adapter's 768 x 768 x 32 896): the 256-tile kernel, whole K-slices per XCD
    const int t256 = (M % 256 == 0 && N % 256 == 0) ? (M / 256) * (N / 256) : 0;
    const int spx = t256 > 0 && t256 <= 32 ? 32 / t256 : 0;
    static const bool allow256 = [] { const char* e = getenv("ARP_DT_TN256"); return !e || atoi(e) != 0; }();
    if (allow256 && spx > 0 && K / 32 >= 8 * spx * 8) {
        S = 8 * spx;
        g.tile256 = 1;
        g.xcd_slices = 1;
    } else {
        const int tiles = (M / 128) * (N / 128), nk = K / 64;
        S = std::max(1, std::min(nk, 512 / std::max(tiles, 1)));  // one resident round of workgroups, as splitk_gemm
        const int per = (nk + S - 1) / S;
        S = (nk + per - 1) / per;
    }
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ksplit = S;
    ProfScope ps(c->prof, st, site);
    if (S == 1) {
        g.out = out; g.ldo = N; g.slice_stride = 0; g.alpha = alpha;
        return launch_gemm_tn(tcode, g, st);
    }
    ARP_TRY(part.ensure((size_t)S * M * N * 4));
    g.out = part.as<float>(); g.ldo = N; g.slice_stride = (size_t)M * N; g.alpha = 1.f;
    ARP_TRY(launch_gemm_tn(tcode, g, st));
    const size_t MN = (size_t)M * N;
    launch_splitk_reduce<float>(st, part.as<float>(), S, MN, N, nullptr, ACT_NONE, out, nullptr, 0, alpha);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// stage: 0 = everything; 1 = up to and including image_text_input's weight gradient (after it every gradient of all-reduce bucket 1
// exists: step_impl); 2 = the rest (the adapter's own backward)
template <typename T> int backward_adapter_tn(arp_dt* c, int stage) {
    const arp_dt_cfg& k = c->cfg;
    const int E = k.emb, D = k.enc_dim;
    const int R = c->R();
    const size_t Mx = (size_t)R * k.enc_tokens;
    const int Kin = k.enc_tokens * D;
    const int Mxp = (int)((Mx + 63) / 64 * 64), Rp64 = (R + 63) / 64 * 64;
    const float S = c->act_scale(), invS = 1.0f / S;
    const T* Yp = k.use_adapter ? c->Y.as<T>() : c->Xb.as<T>();
    // the whole backward in one call on one rank, with the adapter and the fused dY kernel: the two weight-gradient GEMMs nothing waits for go to the side stream
    const bool side = c->side_gemms && stage == 0 && k.use_adapter && c->use_fused_dy() && c->side_stream;
    // The whole backward in one call: dWi (101 MB of f32 gradient at the real geometry, 95 % of the flat gradient) is produced LAST, so that
    // the norm pass right behind it finds those bytes in the 256 MiB Infinity Cache instead of HBM (same launches, same arithmetic; a staged
    // backward needs dWi first for its all-reduce bucket).  ARP_DT_DWI_LAST=0 restores the old order.
    const bool dwi_last = c->dwi_last && stage == 0 && k.use_adapter && !side;
    // the three small reductions of this backward in one launch behind its last GEMM (both of its fused forms on, nothing on the side stream)
    const long tiles256_dx = (long)cdiv((int)Mx, 256) * cdiv(D, 256);
    const bool defer_small = c->merge_small && k.use_adapter && stage != 1 && !side && c->use_fused_dy() && D % 8 == 0 && c->fuse_relu_bwd(tiles256_dx);
    int fin_rows1 = 0, fin_rows0 = 0, fin_ndres = 0;
    if (stage != 2) {
        // dz (f32) -> operand type, scaled (rows R..Rp64 of dzb stay zero); the fused kernel may have written it already (policy_fused)
        if (!(c->fused && c->dzb_from_pf)) ARP_TRY((transpose_mask<float, float, T>(c, c->dz.as<float>(), E, nullptr, nullptr, S, c->dzb.as<T>(), E, nullptr, 0, R, E)));
        // dWi[E, Kin] = dz^T Y: contraction over the R rows, written straight into the gradient buffer
        if (side) {
            ARP_HIP_OK(hipEventRecord(c->ev_fork, c->stream));
            ARP_HIP_OK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        }
        if (!dwi_last)
            ARP_TRY((tn_gemm<T>(c, "dt.image_text_input_dW", c->dzb.as<T>(), E, Yp, Kin, c->g("image_text_input/kernel"), E, Kin, Rp64, invS, side)));
    }
    if (!k.use_adapter || stage == 1) return 0;
    const int prow = Mxp / 64, ncb = cdiv(D, 256);
    if (c->use_fused_dy()) {
        // dY = dz Wi, dApre = res * dY * (A > 0), its column sums and sum dY * (A - x) in ONE pass (adapter_bwd.h): no transposed
        // shadow of Wi, no dY round trip
        ProfScope ps(c->prof, c->stream, "dt.adapter_dy_fused");
        const int nrb = adapter_dy_row_blocks(R), nct = Kin / 128;
        ARP_TRY(c->colpart.ensure((size_t)nrb * k.enc_tokens * D * 4));
        ARP_TRY(c->dres_part.ensure((size_t)nrb * nct * 4));
        AdapterDyArgs a;
        a.dz = c->dzb.p; a.Wi = c->fwd_w("image_text_input/kernel"); a.A = c->A.p; a.x32 = c->bt[c->cur].enc32.as<float>(); a.rw = c->p("residual_weight");
        a.x16 = c->dy_x16 ? c->Xb.p : nullptr;
        a.dApre = c->dApre.p; a.colpart = c->colpart.as<float>(); a.dres_part = c->dres_part.as<float>();
        a.R = R; a.E = E; a.Kin = Kin; a.D = D;
        ARP_TRY(launch_adapter_dy(__is_same(T, bf16_t) ? 1 : 2, a, c->stream));
        if (defer_small) {  // (one launch for these and the Dense_0 bias sums, behind the last adapter GEMM: adapter_grad_finish_kernel)
            fin_rows1 = nrb * k.enc_tokens;
            fin_ndres = nrb * nct;
        } else {
            hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->colpart.as<float>(), nrb * k.enc_tokens, D,
                               c->g("AdapterMLP_0/Dense_1/bias"), invS);
            hipLaunchKernelGGL(reduce_dres_to_drw_kernel, dim3(1), dim3(256), 0, c->stream, c->dres_part.as<float>(), nrb * nct, invS, c->p("residual_weight"),
                               c->g("residual_weight"));
        }
        ARP_HIP_OK(hipGetLastError());
    } else {
        ARP_TRY((big_gemm<T, T, ACT_NONE>(c, "dt.image_text_input_dX", c->dzb.p, E, c->Wit.p, E, nullptr, c->dY.p, Kin, R, Kin, E)));
        // dApre = res * dY * (A > 0), row-major; its column sums = the Dense_1 bias gradient; sum dY * (A - x) = d loss / d res
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_masks");
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->p("residual_weight"), 1, 1.0f, c->scal.as<float>() + 9, 0);
        hipLaunchKernelGGL(sigmoid_scalar_kernel, dim3(1), dim3(1), 0, c->stream, c->scal.as<float>() + 9);
        hipLaunchKernelGGL((mask_copy_colsum_kernel<T>), dim3(ncb, prow), dim3(256), 0, c->stream, c->dY.as<T>(), c->A.as<T>(),
                           c->scal.as<float>() + 9, 1.f, c->dApre.as<T>(), c->colpart.as<float>(), (int)Mx, D, c->bt[c->cur].enc32.as<float>(), c->scal.as<float>() + 16);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->colpart.as<float>(), prow, D, c->g("AdapterMLP_0/Dense_1/bias"), invS);
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->scal.as<float>() + 16, ncb * prow, invS, c->scal.as<float>() + 8, 0);
        hipLaunchKernelGGL(dres_to_drw_kernel, dim3(1), dim3(1), 0, c->stream, c->scal.as<float>() + 8, c->p("residual_weight"), c->g("residual_weight"));
        ARP_HIP_OK(hipGetLastError());
    }
    if (side) {  // dApre is complete on the main stream from here
        ARP_HIP_OK(hipEventRecord(c->ev_dapre, c->stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->side_stream, c->ev_dapre, 0));
    }
    // (H1 = the hidden rows' binary16 segment: c->H1 itself, or the head of fc1's [hi | x4 | dx4] rows with the adapter corrections on)
    ARP_TRY((tn_gemm<T>(c, "dt.adapter_fc2_dW", c->dApre.as<T>(), D, static_cast<const T*>(c->h1_ptr), c->h1_ld, c->g("AdapterMLP_0/Dense_1/kernel"), D, D, Mxp, invS, side)));
    const long tiles256 = (long)cdiv((int)Mx, 256) * cdiv(D, 256);
    if (D % 8 == 0 && c->fuse_relu_bwd(tiles256)) {
        // dH1 = (dApre W2) * (H1 > 0) and its column sums (the Dense_0 bias gradient) in the GEMM's own epilogue (gemm256.h)
        GemmArgs g;
        g.A = c->dApre.p; g.W = c->W2t.p; g.out = c->dH1T.p; g.M = (int)Mx; g.N = D; g.K = D; g.lda = D; g.ldw = D; g.ldr = D; g.ldo = D;
        g.mask = c->h1_ptr; g.ldm = c->h1_ld;
        const int mt = cdiv((int)Mx, 256);
        DevBuf& cpart = defer_small ? c->colpart0 : c->colpart;  // (deferred: the dY kernel's partials in colpart are still waiting for their sums)
        ARP_TRY(cpart.ensure((size_t)mt * D * 4));
        g.colsum_part = cpart.as<float>();
        {
            ProfScope ps(c->prof, c->stream, "dt.adapter_fc2_dX");
            ARP_TRY((launch_gemm256_nt<T, T, ACT_NONE, false, SITE_DT>(g, c->stream)));
        }
        if (defer_small) fin_rows0 = mt;
        else hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->colpart.as<float>(), mt, D, c->g("AdapterMLP_0/Dense_0/bias"), invS);
        ARP_HIP_OK(hipGetLastError());
    } else {
        ARP_TRY((big_gemm<T, T, ACT_NONE>(c, "dt.adapter_fc2_dX", c->dApre.p, D, c->W2t.p, D, nullptr, c->G.p, D, (int)Mx, D, D)));
        // dH1 = G * (H1 > 0), row-major (in the buffer the other path uses for its transposed copy), + the Dense_0 bias gradient
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_masks");
        if (c->h1_ld != D) return fail("backward_adapter_tn: the unfused ReLU backward reads a contiguous H1 (ARP_DT_ADAPTER_H1_INPLACE=0)");
        hipLaunchKernelGGL((mask_copy_colsum_kernel<T>), dim3(cdiv(D, 256), prow), dim3(256), 0, c->stream, c->G.as<T>(), c->H1.as<T>(), nullptr, 1.f,
                           c->dH1T.as<T>(), c->colpart.as<float>(), (int)Mx, D);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, c->stream, c->colpart.as<float>(), prow, D, c->g("AdapterMLP_0/Dense_0/bias"), invS);
        ARP_HIP_OK(hipGetLastError());
    }
    ARP_TRY((tn_gemm<T>(c, "dt.adapter_fc1_dW", c->dH1T.as<T>(), D, c->Xb.as<T>(), D, c->g("AdapterMLP_0/Dense_0/kernel"), D, D, Mxp, invS)));
    if (defer_small) {
        if (fin_rows1 <= 0 || fin_rows0 <= 0) return fail("backward_adapter_tn: deferred reductions without their partials");
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_masks");
        hipLaunchKernelGGL(adapter_grad_finish_kernel, dim3(2 * cdiv(D, 64) + 1), dim3(256), 0, c->stream, c->colpart.as<float>(), fin_rows1, c->g("AdapterMLP_0/Dense_1/bias"),
                           c->colpart0.as<float>(), fin_rows0, c->g("AdapterMLP_0/Dense_0/bias"), D, invS, c->dres_part.as<float>(), fin_ndres, c->p("residual_weight"),
                           c->g("residual_weight"));
        ARP_HIP_OK(hipGetLastError());
    }
    if (dwi_last) ARP_TRY((tn_gemm<T>(c, "dt.image_text_input_dW", c->dzb.as<T>(), E, Yp, Kin, c->g("image_text_input/kernel"), E, Kin, Rp64, invS)));
    if (side) {  // join: everything after the backward (norms, Adam, a later forward) is ordered behind the side stream's two GEMMs
        ARP_HIP_OK(hipEventRecord(c->ev_side, c->side_stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->stream, c->ev_side, 0));
    }
    return 0;
}

// ---- backward: fills c->grads (every entry written exactly once; no accumulation across calls) ----------
template <typename T> int backward(arp_dt* c, int stage = 0) {
    const arp_dt_cfg& k = c->cfg;
    const int E = k.emb, D = k.enc_dim, H = k.mlp_ratio * E, NA = k.n_actions, depth = k.depth;
    const int R = c->R(), L = c->L(), BL = c->B * L;
    const size_t Mx = (size_t)R * k.enc_tokens;
    const int Kin = k.enc_tokens * D;
    const int Mxp = (int)((Mx + 63) / 64 * 64), Rp = (R + 63) / 64 * 64;
    float* dh = c->dh.as<float>();
    c->grads_summed = false;  // this rank's own gradient from here on
    const bool tn = sizeof(T) == 2 && c->use_tn();
    if (stage == 2) {  // second half of a staged backward: only the TN path has one (the other paths did everything in stage 1)
        if constexpr (sizeof(T) == 2) {
            if (tn) return backward_adapter_tn<T>(c, 2);
        }
        return 0;
    }
    if (c->fused) {
        // activation gradients came out of policy_fused_kernel; every parameter gradient of the transformer, the
        // heads, the LayerNorms and the embeddings is produced by three launches
        // (Tried: these three launches and two of the weight-gradient contractions on a second stream, forked / joined with events
        // so that the step's hipGraph holds them as parallel branches: 1.33 ms per step against 0.99 ms on one stream.)
        hipStream_t st = c->stream;
        ProfScope ps(c->prof, st, "dt.policy_bwd");
        if (c->merge_small) {
            PfGradsArgs a;
            a.gtab = c->gtab.as<SmallGemm>(); a.gprefix = c->gprefix.as<int>(); a.n_gemm = c->n_gemm; a.gemm_tiles = c->gemm_tiles;
            a.ctab = c->ctab.as<ColSumJob>(); a.cprefix = c->cprefix.as<int>(); a.n_cs = c->n_cs; a.cs_tiles = c->cs_tiles;
            a.dtok = c->dtok.as<float>(); a.rtg = c->bt[c->cur].rtg.as<float>(); a.action = c->bt[c->cur].action.as<int>();
            a.dWr = c->g("rtg_input/kernel"); a.demb = c->g("action_input/embedding"); a.R = R; a.E = E; a.NA = NA;
            a.loss_part = c->loss_part.as<float>(); a.B = c->B; a.lambda = k.lambda_ret; a.metrics = c->metrics.as<float>();
            hipLaunchKernelGGL(pf_param_grads_kernel, dim3(c->gemm_tiles + c->cs_tiles + NA + 2), dim3(256), 0, st, a);
            ARP_HIP_OK(hipGetLastError());
        } else {
        hipLaunchKernelGGL(grouped_small_gemm_kernel, dim3(c->gemm_tiles), dim3(256), 0, st, c->gtab.as<SmallGemm>(), c->gprefix.as<int>(), c->n_gemm);
        hipLaunchKernelGGL(grouped_colsum_kernel, dim3(c->cs_tiles), dim3(256), 0, st, c->ctab.as<ColSumJob>(), c->cprefix.as<int>(), c->n_cs);
        hipLaunchKernelGGL(tokens_bwd_par_kernel, dim3(NA + 1), dim3(TOKB_THREADS), 0, st, c->dtok.as<float>(), c->bt[c->cur].rtg.as<float>(), c->bt[c->cur].action.as<int>(),
                           c->g("rtg_input/kernel"), c->g("action_input/embedding"), R, E, NA);
        ARP_HIP_OK(hipGetLastError());
        }
    } else {
        ProfScope ps(c->prof, c->stream, "dt.policy_bwd");
        // heads (arp_dt/ARPDT.py:94-99,206-220)
        ARP_TRY(linear_bwd(c, c->ha.as<float>(), c->p("action_outputs_0/layers_2/kernel"), c->dlogits.as<float>(), c->g("action_outputs_0/layers_2/kernel"),
                           nullptr, c->t1.as<float>(), R, NA, E));
        ARP_TRY(ew_bwd(c, c->t1.as<float>(), c->ha.as<float>(), c->dha.as<float>(), (size_t)R * E, EW_RELU_BWD));
        ARP_TRY(linear_bwd(c, c->a_in.as<float>(), c->p("action_outputs_0/layers_0/kernel"), c->dha.as<float>(), c->g("action_outputs_0/layers_0/kernel"),
                           c->g("action_outputs_0/layers_0/bias"), c->da_in.as<float>(), R, E, E));
        ARP_TRY(linear_bwd(c, c->hr.as<float>(), c->p("return_outputs_0/layers_2/kernel"), c->dret.as<float>(), c->g("return_outputs_0/layers_2/kernel"),
                           nullptr, c->t1.as<float>(), R, 1, E));
        ARP_TRY(ew_bwd(c, c->t1.as<float>(), c->hr.as<float>(), c->dhr.as<float>(), (size_t)R * E, EW_RELU_BWD));
        ARP_TRY(linear_bwd(c, c->r_in.as<float>(), c->p("return_outputs_0/layers_0/kernel"), c->dhr.as<float>(), c->g("return_outputs_0/layers_0/kernel"),
                           c->g("return_outputs_0/layers_0/bias"), c->dr_in.as<float>(), R, E, E));
        hipLaunchKernelGGL(heads_scatter_kernel, dim3(cdiv((size_t)R * E, 256)), dim3(256), 0, c->stream, c->da_in.as<float>(), c->dr_in.as<float>(),
                           c->dhf.as<float>(), R, E);
        ARP_TRY(ln_bwd(c, c->xs[depth].as<float>(), c->p("policy/LayerNorm_0/scale"), c->dhf.as<float>(), dh, 0, c->g("policy/LayerNorm_0/scale"),
                       c->g("policy/LayerNorm_0/bias"), BL, E));
        for (int i = depth - 1; i >= 0; --i) {
            const std::string p = "policy/Block_" + std::to_string(i) + "/";
            // x_{i+1} = hmid + gelu(ln1 Wfc1) Wfc2
            ARP_TRY(linear_bwd(c, c->gl[i].as<float>(), c->p(p + "FeedForward_0/fc2/kernel"), dh, c->g(p + "FeedForward_0/fc2/kernel"), nullptr,
                               c->t1.as<float>(), BL, E, H));
            ARP_TRY(ew_bwd(c, c->t1.as<float>(), c->u[i].as<float>(), c->t2.as<float>(), (size_t)BL * H, EW_GELU_BWD));
            ARP_TRY(linear_bwd(c, c->ln1[i].as<float>(), c->p(p + "FeedForward_0/fc1/kernel"), c->t2.as<float>(), c->g(p + "FeedForward_0/fc1/kernel"),
                               nullptr, c->t3.as<float>(), BL, H, E));
            ARP_TRY(ln_bwd(c, c->hmid[i].as<float>(), c->p(p + "LayerNorm_1/scale"), c->t3.as<float>(), dh, 1, c->g(p + "LayerNorm_1/scale"),
                           c->g(p + "LayerNorm_1/bias"), BL, E));
            // hmid = x_i + att Wo + bo
            ARP_TRY(linear_bwd(c, c->att[i].as<float>(), c->p(p + "Attention_0/Dense_1/kernel"), dh, c->g(p + "Attention_0/Dense_1/kernel"),
                               c->g(p + "Attention_0/Dense_1/bias"), c->t3.as<float>(), BL, E, E));
            {
                const int hd = E / k.heads;
                const size_t lds = ((size_t)4 * L * hd + 2 * L * L) * 4;
                hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(c->B * k.heads), dim3(64), lds, c->stream, c->qkv[i].as<float>(), c->t3.as<float>(),
                                   c->dqkv.as<float>(), L, E, k.heads, 1.0f / sqrtf((float)hd), k.alibi_bias ? c->alibi.as<float>() : nullptr);
                ARP_HIP_OK(hipGetLastError());
            }
            ARP_TRY(linear_bwd(c, c->ln0[i].as<float>(), c->p(p + "Attention_0/Dense_0/kernel"), c->dqkv.as<float>(), c->g(p + "Attention_0/Dense_0/kernel"),
                               c->g(p + "Attention_0/Dense_0/bias"), c->t3.as<float>(), BL, 3 * E, E));
            ARP_TRY(ln_bwd(c, c->xs[i].as<float>(), c->p(p + "LayerNorm_0/scale"), c->t3.as<float>(), dh, 1, c->g(p + "LayerNorm_0/scale"),
                           c->g(p + "LayerNorm_0/bias"), BL, E));
        }
        hipLaunchKernelGGL(tokens_bwd_kernel, dim3(cdiv(E, 256)), dim3(256), 0, c->stream, dh, c->bt[c->cur].rtg.as<float>(), c->bt[c->cur].action.as<int>(),
                           c->dimg.as<float>(), c->g("rtg_input/kernel"), c->g("action_input/embedding"), R, E, NA);
        ARP_TRY(ew_bwd(c, c->dimg.as<float>(), c->img.as<float>(), c->dz.as<float>(), (size_t)R * E, EW_TANH_BWD));
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(E, 64)), dim3(256), 0, c->stream, c->dz.as<float>(), R, E, c->g("image_text_input/bias"));
        ARP_HIP_OK(hipGetLastError());
    }
    if constexpr (sizeof(T) == 2) {
        if (tn) return backward_adapter_tn<T>(c, stage);
    }
    // ---- image_text_input: dW[E, Kin] = dz^T Y ;  dY[R, Kin] = dz Wi -------------------------------------
    const T* Yp = k.use_adapter ? c->Y.as<T>() : c->Xb.as<T>();
    const float S = c->act_scale(), invS = 1.0f / S;
    ARP_TRY((transpose_mask<float, float, T>(c, c->dz.as<float>(), E, nullptr, nullptr, S, c->dzb.as<T>(), E, c->dzT.as<T>(), Rp, R, E)));
    {
        ProfScope ps(c->prof, c->stream, "dt.Y_transpose");
        ARP_TRY((transpose_mask<T, T, T>(c, Yp, Kin, nullptr, nullptr, 1.f, nullptr, 0, c->YT.as<T>(), Rp, R, Kin)));
    }
    ARP_TRY((big_gemm<T, float, ACT_NONE>(c, "dt.image_text_input_dW", c->dzT.p, Rp, c->YT.p, Rp, nullptr, c->g("image_text_input/kernel"), Kin, E, Kin, Rp, invS)));
    if (!k.use_adapter) return 0;
    ARP_TRY((big_gemm<T, T, ACT_NONE>(c, "dt.image_text_input_dX", c->dzb.p, E, c->Wit.p, E, nullptr, c->dY.p, Kin, R, Kin, E)));
    // ---- adapter backward (y = res a + (1-res) x, x is stop_gradient'ed: arp_dt/ARPDT.py:462-472) --------
    {
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_elementwise");
        const int nb = 1024;
        hipLaunchKernelGGL((adapter_dres_kernel<T>), dim3(nb), dim3(256), 0, c->stream, c->dY.as<T>(), c->A.as<T>(), c->bt[c->cur].enc32.as<float>(),
                           c->scal.as<float>() + 16, Mx * D);
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->scal.as<float>() + 16, nb, invS, c->scal.as<float>() + 8, 0);
        hipLaunchKernelGGL(dres_to_drw_kernel, dim3(1), dim3(1), 0, c->stream, c->scal.as<float>() + 8, c->p("residual_weight"), c->g("residual_weight"));
        // res = sigmoid(rw) as a device scalar for the masked transposes
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->p("residual_weight"), 1, 1.0f, c->scal.as<float>() + 9, 0);
        ARP_HIP_OK(hipGetLastError());
    }
    // dApre = res * dY * (A > 0), both layouts.  res is applied through scale_ptr after a sigmoid kernel:
    {
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_masks");
        hipLaunchKernelGGL(sigmoid_scalar_kernel, dim3(1), dim3(1), 0, c->stream, c->scal.as<float>() + 9);
        ARP_TRY((transpose_mask<T, T, T>(c, c->dY.as<T>(), D, c->A.as<T>(), c->scal.as<float>() + 9, 1.f, c->dApre.as<T>(), D, c->dApreT.as<T>(), Mxp,
                                         (int)Mx, D)));
        ARP_TRY((transpose_mask<T, T, T>(c, c->H1.as<T>(), D, nullptr, nullptr, 1.f, nullptr, 0, c->H1T.as<T>(), Mxp, (int)Mx, D)));
    }
    ARP_TRY((splitk_gemm<T, float>(c, "dt.adapter_fc2_dW", c->dApreT.p, Mxp, c->H1T.p, Mxp, nullptr, ACT_NONE, c->g("AdapterMLP_0/Dense_1/kernel"), D, D, Mxp, invS)));
    hipLaunchKernelGGL((rowsum_kernel<T>), dim3(D), dim3(256), 0, c->stream, c->dApreT.as<T>(), Mxp, (int)Mx, c->g("AdapterMLP_0/Dense_1/bias"), D, invS);
    ARP_TRY((big_gemm<T, T, ACT_NONE>(c, "dt.adapter_fc2_dX", c->dApre.p, D, c->W2t.p, D, nullptr, c->G.p, D, (int)Mx, D, D)));
    {
        ProfScope ps(c->prof, c->stream, "dt.adapter_bwd_masks");
        ARP_TRY((transpose_mask<T, T, T>(c, c->G.as<T>(), D, c->H1.as<T>(), nullptr, 1.f, nullptr, 0, c->dH1T.as<T>(), Mxp, (int)Mx, D)));
    }
    ARP_TRY((splitk_gemm<T, float>(c, "dt.adapter_fc1_dW", c->dH1T.p, Mxp, c->XbT.p, Mxp, nullptr, ACT_NONE, c->g("AdapterMLP_0/Dense_0/kernel"), D, D, Mxp, invS)));
    hipLaunchKernelGGL((rowsum_kernel<T>), dim3(D), dim3(256), 0, c->stream, c->dH1T.as<T>(), Mxp, (int)Mx, c->g("AdapterMLP_0/Dense_0/bias"), D, invS);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// weight_l2 = sum p^2 over ndim > 1 params -> scal[1];  grads += wd * p there (main_procgen.py:114-117)
int l2_penalty(arp_dt* c) {
    ProfScope ps(c->prof, c->stream, "dt.l2_penalty");
    const int nb = 1024;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, c->stream, c->params.as<float>(), c->n_decay, c->scal.as<float>() + 16);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->scal.as<float>() + 16, nb, 1.0f, c->scal.as<float>() + 1, 0);
    hipLaunchKernelGGL(add_scaled_kernel, dim3(cdiv(c->n_decay, 256)), dim3(256), 0, c->stream, c->grads.as<float>(), c->params.as<float>(),
                       c->cfg.weight_decay, c->n_decay);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// L2 term + optax.clip_by_global_norm + adam in two passes over the flat state: (1) both norms, (2) the update with the
// L2 gradient wd*p folded in.  (main_procgen.py:114-117,490-507; the gradient buffer keeps the raw loss gradient.)
int apply_update(arp_dt* c, float lr) {
    ProfScope ps(c->prof, c->stream, "dt.clip_adam");
    const int nb = 1024;
    const float gscale = 1.0f / (float)std::max(c->cfg.world, 1);
    float* pg = c->scal.as<float>() + 16;
    float* pp = pg + nb;
    hipLaunchKernelGGL(norms_partial_kernel, dim3(nb), dim3(256), 0, c->stream, c->grads.as<float>(), c->params.as<float>(), c->P, c->n_decay, gscale,
                       c->cfg.weight_decay, pg, pp);
    if (c->merge_small) {
        hipLaunchKernelGGL(reduce_sum2_kernel, dim3(2), dim3(256), 0, c->stream, pg, pp, nb, c->scal.as<float>() + 0, c->scal.as<float>() + 1);
    } else {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, pg, nb, 1.0f, c->scal.as<float>() + 0, 0);
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, pp, nb, 1.0f, c->scal.as<float>() + 1, 0);
    }
    const double t = (double)(c->step + 1);
    const float bc1 = (float)(1.0 - std::pow((double)c->cfg.b1, t)), bc2 = (float)(1.0 - std::pow((double)c->cfg.b2, t));
#define ARP_ADAM(TM)                                                                                                                    \
    hipLaunchKernelGGL((adam_kernel<TM>), dim3(cdiv(c->P / 4, 256)), dim3(256), 0, c->stream, c->params.as<float>(), c->grads.as<float>(),        \
                       c->mu.as<float>(), c->nu.as<float>(), c->scal.as<float>(), gscale, c->cfg.weight_decay, c->n_decay, c->cfg.clip_norm, lr, \
                       c->cfg.b1, c->cfg.b2, c->cfg.eps, bc1, bc2, c->P, c->mirror.as<TM>(), c->mirror_stale ? (size_t)0 : c->n_mirror, c->adam_rev ? 1 : 0)
    if (c->cfg.mode == ARP_MODE_BF16) ARP_ADAM(bf16_t);
    else if (c->cfg.mode == ARP_MODE_F16) ARP_ADAM(f16_t);
    else ARP_ADAM(float);
#undef ARP_ADAM
    ARP_HIP_OK(hipGetLastError());
    c->step += 1;
    c->shadows_stale = true;
    return 0;
}

template <typename T> int fwd_bwd(arp_dt* c, int stage = 0) {
    if (stage == 3) return forward<T>(c, false);  // forward only (arp_dt_forward, arp_dt_val_step: greedy_action / validation)
    if (stage != 2) ARP_TRY(forward<T>(c, true));
    return backward<T>(c, stage);
}

// Replays forward + backward as one hipGraph (a chain of short dependent kernels of a few
// microseconds).  The first steps of a geometry run eagerly (lazy workspace allocations must not happen under
// capture); profiling and any capture failure fall back to eager launches.
template <typename T> int fwd_bwd_graphed_chain(arp_dt* c, int stage);
template <typename T> int fwd_bwd_graphed(arp_dt* c, int stage = 0) {
    if (c->use_images && c->enc_eager && c->use_graph && !c->prof.on && stage != 2) {
        ARP_TRY(encode_current(c));
        c->enc_outside = true;
    }
    const int rc = fwd_bwd_graphed_chain<T>(c, stage);
    c->enc_outside = false;
    return rc;
}
template <typename T> int fwd_bwd_graphed_chain(arp_dt* c, int stage) {
    const int images = c->use_images && !c->enc_outside ? 1 : 0;
    if (!c->use_graph || c->prof.on) return fwd_bwd<T>(c, stage);
    arp_dt::GraphRec& gr = c->graphs[c->cur][stage];  // the chain holds the batch slot's pointers
    if (gr.exec && (gr.B != c->B || gr.images != images)) {
        (void)hipGraphExecDestroy(gr.exec);
        gr.exec = nullptr;
        gr.eager = 0;
    }
    if constexpr (sizeof(T) == 2) {
        if (c->mirror_stale && stage != 2) {  // a host write since the last step: rebuild the operand mirror eagerly, never inside the captured chain
            c->shadows_stale = true;
            ARP_TRY(refresh_shadows<T>(c));
        }
    }
    if constexpr (sizeof(T) == 2) {
        // the forward-only chain is captured WITHOUT the shadow refresh (the parameters do not change between greedy_action calls):
        // stale shadows are rebuilt here, eagerly, before the capture or the replay
        if (stage == 3 && c->shadows_stale) ARP_TRY(refresh_shadows<T>(c));
    }
    if (!gr.exec) {
        if (gr.eager < 2) {
            gr.eager++;
            return fwd_bwd<T>(c, stage);
        }
        if (stage != 2 && stage != 3) c->shadows_stale = true;  // the captured chain always refreshes the operand shadows
        // The uploader thread of prefetch_to_device must not issue HIP calls while this thread captures: its hipEventSynchronize on the
        // slot's `use` event (last recorded on THIS stream, before the capture) is refused while the stream captures ("operation not
        // permitted on an event last recorded in a capturing stream") and the refusal invalidates the capture ("operation failed due to
        // a previous error during capture": one run in ~300 before this lock, every second run when provoked).  capture_mu is held for
        // the few hundred microseconds of a capture -- three per trainer and batch slot -- and by arp_dt_upload_batch*_async.
        std::lock_guard<std::mutex> capture_lock(c->capture_mu);
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            c->use_graph = false;
            return fwd_bwd<T>(c, stage);
        }
        const int rc = fwd_bwd<T>(c, stage);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != 0 || e != hipSuccess || !g || hipGraphInstantiate(&gr.exec, g, nullptr, nullptr, 0) != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            gr.exec = nullptr;
            c->use_graph = false;
            for (int k = 0; k < 4 && hipGetLastError() != hipSuccess; ++k) {}  // the invalidated capture's error is sticky until read
            if (stage != 2) c->shadows_stale = true;
            return fwd_bwd<T>(c, stage);
        }
        (void)hipGraphDestroy(g);
        gr.B = c->B;
        gr.images = images;
    }
    ARP_HIP_OK(hipGraphLaunch(gr.exec, c->stream));
    if (stage != 2) c->shadows_stale = false;
    return 0;
}

// The current slot's last reader is enqueued on the compute stream: an upload into this slot waits (host side) for this event.
int mark_slot_read(arp_dt* c) {
    arp_dt::BatchSlot& slot = c->bt[c->cur];
    ARP_HIP_OK(hipEventRecord(slot.use, c->stream));
    slot.used = true;
    return 0;
}

// The flat gradient as two all-reduce buckets of two ranges each (they tile [0, P) exactly once; arp_dt_bucket_plan):
//   bucket 1 = [off(image_text_input/kernel), n_decay) + [off(image_text_input/bias), P): image_text_input's kernel (25.3 M of the
//              26.9 M parameters) and every matrix / vector the transformer, the heads and the embeddings own -- all of it exists once
//              image_text_input_dW has run, BEFORE the adapter's three backward GEMMs;
//   bucket 2 = [0, off(image_text_input/kernel)) + [n_decay, off(image_text_input/bias)): the adapter's two kernels and its vectors.
struct BucketPlan { size_t lo[4], hi[4]; };
BucketPlan bucket_plan(const arp_dt* c) {
    BucketPlan b;
    const size_t wi = c->infos[c->index.at("image_text_input/kernel")].off, bi = c->infos[c->index.at("image_text_input/bias")].off;
    b.lo[0] = wi; b.hi[0] = c->n_decay;   // bucket 1
    b.lo[1] = bi; b.hi[1] = c->P;
    b.lo[2] = 0; b.hi[2] = wi;            // bucket 2
    b.lo[3] = c->n_decay; b.hi[3] = bi;
    return b;
}
int allreduce_ranges(arp_dt* c, const BucketPlan& b, int first, hipStream_t st, bool with_metrics) {
    RcclApi* r = rccl_api();
    const bool grp = r->GroupStart && r->GroupEnd;
    if (grp) r->GroupStart();
    int rc = 0;
    for (int i = first; i < first + 2; ++i)
        if (b.hi[i] > b.lo[i]) {
            float* p = c->grads.as<float>() + b.lo[i];
            if (r->AllReduce(p, p, b.hi[i] - b.lo[i], ncclFloat, ncclSum, c->comm, st) != ncclSuccess) rc = -1;
        }
    if (with_metrics && r->AllReduce(c->metrics.p, c->metrics.p, 4, ncclFloat, ncclSum, c->comm, st) != ncclSuccess) rc = -1;
    if (grp) r->GroupEnd();
    return rc ? fail("ncclAllReduce(gradient bucket) failed") : 0;
}

template <typename T> int step_impl(arp_dt* c, float lr, float* aux) {
    const bool comm = c->has_comm && (c->cfg.world > 1 || c->force_comm);
    // pmean of (loss, aux, grads) over devices (main_procgen.py:132); the 1/world factor is folded into the update kernel.  The
    // reference gets communication / computation overlap from XLA's scheduler under pmap; here it is explicit:
    if (comm && c->overlap_comm) {
        // stage 1 (forward, transformer backward, image_text_input_dW) -> bucket 1 goes out on the communication stream while stage
        // 2 (the adapter's backward: the fused dY pass and three 768 x 768 x 32 896 GEMMs, ~0.25 ms) still runs -> bucket 2 + the
        // loss scalars -> the update waits for both.
        const BucketPlan b = bucket_plan(c);
        if (!c->comm_stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        ARP_TRY(fwd_bwd_graphed<T>(c, 1));
        ARP_HIP_OK(hipEventRecord(c->ev_b1, c->stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->comm_stream, c->ev_b1, 0));
        {
            ProfScope ps(c->prof, c->comm_stream, "dt.allreduce_b1");
            ARP_TRY(allreduce_ranges(c, b, 0, c->comm_stream, false));
        }
        ARP_TRY(fwd_bwd_graphed<T>(c, 2));
        ARP_HIP_OK(hipEventRecord(c->ev_b2, c->stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->comm_stream, c->ev_b2, 0));
        {
            ProfScope ps(c->prof, c->comm_stream, "dt.allreduce_b2");
            ARP_TRY(allreduce_ranges(c, b, 2, c->comm_stream, true));
        }
        ARP_HIP_OK(hipEventRecord(c->ev_comm, c->comm_stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        c->grads_summed = true;
    } else {
        ARP_TRY(fwd_bwd_graphed<T>(c, 0));
        if (comm) {  // the serial form: ONE all-reduce(sum) of the flat gradient plus one of the 4 loss scalars, behind the whole backward
            ProfScope ps(c->prof, c->stream, "dt.allreduce");
            if (rccl_api()->AllReduce(c->grads.p, c->grads.p, c->P, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return fail("ncclAllReduce(grads) failed");
            if (rccl_api()->AllReduce(c->metrics.p, c->metrics.p, 4, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return fail("ncclAllReduce(metrics) failed");
            c->grads_summed = true;
        }
    }
    ARP_TRY(mark_slot_read(c));
    const long long step_before = c->step;
    ARP_TRY(apply_update(c, lr));
    if (aux) {
        float m[4], s[2];
        ARP_HIP_OK(hipMemcpyAsync(m, c->metrics.p, 16, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipMemcpyAsync(s, c->scal.p, 8, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        const float inv = 1.0f / (float)std::max(c->cfg.world, 1);
        const float l2 = s[1], pen = c->cfg.weight_decay * 0.5f * l2;
        aux[0] = m[0] * inv + pen;   // loss (incl. the L2 penalty)
        aux[1] = m[1] * inv * 100.f; // acc * 100
        aux[2] = m[2] * inv;         // trans_loss
        aux[3] = m[3] * inv;         // return_loss
        aux[4] = pen;                // weight_penalty
        aux[5] = l2;                 // weight_l2
        aux[6] = (float)step_before; // train_state_step
        aux[7] = lr;                 // learning_rate
        aux[8] = sqrtf(s[0]);        // (extra) global norm of the rank-averaged gradient incl. the L2 term, before clipping
    }
    return 0;
}

}  // namespace

// =================================== C ABI ===============================================================
extern "C" {

int arp_dt_create(const arp_dt_cfg* cfg, arp_dt** out) {
    if (!cfg || !out) return fail("null argument");
    const arp_dt_cfg& k = *cfg;
    if (k.mode != ARP_MODE_F32 && k.mode != ARP_MODE_BF16 && k.mode != ARP_MODE_F16) return fail("bad mode");
    if (k.emb <= 0 || k.heads <= 0 || k.emb % k.heads || k.emb % 4) return fail("emb must be a positive multiple of heads and of 4");
    const int hd = k.emb / k.heads;
    if (hd != 16 && hd != 32 && hd != 64) return fail("head_dim must be 16, 32 or 64");
    const int kq = k.mode == ARP_MODE_F32 ? 32 : 64;
    if (k.enc_dim % kq || k.emb % kq) return fail("enc_dim and emb must be multiples of " + std::to_string(kq));
    if (k.window <= 0 || 3 * k.window > 64) return fail("window must be in 1..21");
    if (k.depth <= 0 || k.n_actions <= 0 || k.enc_tokens <= 0 || k.mlp_ratio <= 0) return fail("bad geometry");
    int ndev = 0;
    ARP_HIP_OK(hipGetDeviceCount(&ndev));
    if (k.device < 0 || k.device >= ndev) return fail("no such HIP device: " + std::to_string(k.device));
    ARP_HIP_OK(hipSetDevice(k.device));
    ARP_TRY(prime_runtime(k.device));  // (runtime.h: one null-stream copy before the process's first stream exists)
    arp_dt* c = new arp_dt();
    c->cfg = k;
    if (c->cfg.world <= 0) c->cfg.world = 1;
    if (const char* e = getenv("ARP_DT_GRAPH")) c->use_graph = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_FUSE_RELU_BWD")) c->relu_fuse_mode = atoi(e);
    c->pf_x3 = k.mode != ARP_MODE_F32;
    if (const char* e = getenv("ARP_PF_X3")) c->pf_x3 = atoi(e) != 0;
    c->iti_f32 = k.mode != ARP_MODE_F32;
    if (const char* e = getenv("ARP_DT_ITI_F32")) c->iti_f32 = atoi(e) != 0 && k.mode != ARP_MODE_F32;
    if (const char* e = getenv("ARP_DT_ITI_X3")) c->iti_x3 = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_ITI_MIX")) c->iti_mix = atoi(e) != 0;
    // Round 6: ON by default where the corrected products exist (f16, adapter widths that are multiples of 256): the plain f16 adapter reads 8.7e-4 on the logits over
    // 16 seeds of N(0,1) encodings and 1.18e-3 behind real encoder outputs (one seed of eight outside north_star's 1e-3); corrected (plan 22d) 1.6e-4 / 2.1e-4,
    // for +0.12 ms per 32-sample step (profiles/r6_adapter_plans.txt).  ARP_DT_ADAPTER_C=0 / arp_dt_set_adapter_corrections(h, 0): the plain products.
    c->adapter_c = k.mode == ARP_MODE_F16 && k.use_adapter && k.enc_dim % 256 == 0 && k.enc_dim >= 512;
    if (const char* e = getenv("ARP_DT_ADAPTER_C")) c->adapter_c = c->adapter_c && atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_ADAPTER_PLAN")) {  // "<fc1><fc2><e|h|d>", e.g. 22e (round 5), 12h
        if (e[0] >= '1' && e[0] <= '2') c->ac_plan1 = e[0] - '0';
        if (e[0] && e[1] >= '1' && e[1] <= '2') c->ac_plan2 = e[1] - '0';
        if (e[0] && e[1] && (e[2] == 'e' || e[2] == 'h' || e[2] == 'd')) { c->ac_a_exact = e[2] == 'e'; c->ac_a_dx = e[2] == 'd'; }
    }
    if (const char* e = getenv("ARP_DT_ADAPTER_H1_INPLACE")) c->ac_h1_inplace = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_OVERLAP")) c->overlap_comm = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_SIDE")) c->side_gemms = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_DWI_LAST")) c->dwi_last = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_MERGE")) c->merge_small = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_PACK_EARLY")) c->pack_early = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_DY_X16")) c->dy_x16 = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_MIX_X16")) c->mix_x16 = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_ADAM_REV")) c->adam_rev = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_FORCE_COMM")) c->force_comm = atoi(e) != 0;
    if (const char* e = getenv("ARP_DT_ENC_EAGER")) c->enc_eager = atoi(e) != 0;
    build_layout(c);
    auto body = [&]() -> int {
        // The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (8 here, arp_amd/_ffi.py): two streams that land on ONE queue run
        // one after the other.  Round 6 measured what that costs: with the encoder's stream added this handle + its encoder held nine streams, the encoder's two
        // part streams shared a queue, and the step went 10.6 -> 12.4 ms (profiles/r6_n1_ab_queues.txt).  Only the compute stream exists from the start; the
        // communication, side and copy streams are created by the first call that needs them (stream_or_create).
        ARP_HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (c->side_gemms) ARP_HIP_OK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        for (hipEvent_t* e : {&c->ev_fork, &c->ev_dapre, &c->ev_side}) ARP_HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (hipEvent_t* e : {&c->ev_b1, &c->ev_b2, &c->ev_comm, &c->bt[0].up, &c->bt[0].use, &c->bt[1].up, &c->bt[1].use, &c->bt[2].up, &c->bt[2].use})
            ARP_HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        DevBuf* fb[] = {&c->params, &c->grads, &c->mu, &c->nu};
        for (auto* b : fb) {
            ARP_TRY(b->ensure(c->P * 4));
            ARP_HIP_OK(hipMemset(b->p, 0, c->P * 4));
        }
        const size_t e = c->esz(), D = k.enc_dim, Kin = (size_t)k.enc_tokens * D;
        c->n_mirror = c->infos[c->index.at("action_input/embedding")].off;  // flat prefix = (W1, W2,) Wi: the big Dense kernels
        if (k.mode != ARP_MODE_F32) ARP_TRY(c->mirror.ensure(c->n_mirror * e));
        if (k.use_adapter) ARP_TRY(c->W2t.ensure(D * D * e));
        ARP_TRY(c->Wit.ensure(Kin * k.emb * e));
        if (k.alibi_bias) {
            if (k.heads > 16) return fail("alibi_bias: at most 16 heads");
            const std::vector<float> sl = alibi_slopes(k.heads);
            ARP_TRY(c->alibi.ensure(64));
            ARP_HIP_OK(hipMemcpy(c->alibi.p, sl.data(), sl.size() * 4, hipMemcpyHostToDevice));
        }
        return 0;
    };
    if (body() != 0) { arp_dt_destroy(c); return -1; }
    *out = c;
    return 0;
}

int arp_dt_destroy(arp_dt* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    for (hipStream_t st : {c->stream, c->comm_stream, c->side_stream, c->copy_stream[0], c->copy_stream[1], c->enc_stream})  // (an encode-ahead pass may still be in flight)
        if (st) (void)hipStreamSynchronize(st);
    for (auto& slot : c->graphs)
        for (auto& gr : slot)
            if (gr.exec) (void)hipGraphExecDestroy(gr.exec);
    if (c->has_comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
    for (hipEvent_t e : {c->ev_fork, c->ev_dapre, c->ev_side, c->ev_enc_go, c->bt[0].enc_done, c->bt[1].enc_done, c->bt[2].enc_done})
        if (e) (void)hipEventDestroy(e);
    if (c->enc_stream) (void)hipStreamDestroy(c->enc_stream);
    c->part_side.release();
    for (hipEvent_t e : {c->ev_b1, c->ev_b2, c->ev_comm, c->bt[0].up, c->bt[0].use, c->bt[1].up, c->bt[1].use, c->bt[2].up, c->bt[2].use})
        if (e) (void)hipEventDestroy(e);
    c->prof.destroy();
    DevBuf* all[] = {&c->params, &c->grads, &c->mu, &c->nu, &c->mirror, &c->W2t, &c->Wit, &c->colpart, &c->Y32, &c->bt[0].enc32, &c->bt[0].img32, &c->bt[0].action, &c->bt[0].rtg, &c->bt[1].enc32, &c->bt[1].img32, &c->bt[1].action, &c->bt[1].rtg, &c->bt[2].enc32, &c->bt[2].img32, &c->bt[2].action, &c->bt[2].rtg, &c->Xb, &c->XbT,
                     &c->H1, &c->H1T, &c->A, &c->Y, &c->YT, &c->Xc, &c->H1c, &c->A32, &c->Adx, &c->W1c, &c->W2c, &c->wc_scal, &c->dY, &c->dApre, &c->dApreT, &c->G, &c->dH1T, &c->dzb, &c->dzT, &c->part, &c->scal, &c->img,
                     &c->hf, &c->a_in, &c->r_in, &c->ha, &c->hr, &c->logits, &c->ret, &c->metrics, &c->dlogits, &c->dret, &c->dha, &c->dhr, &c->da_in,
                     &c->dr_in, &c->dhf, &c->dh, &c->t1, &c->t2, &c->t3, &c->dws, &c->dbs, &c->dimg, &c->dz, &c->dqkv,
                     &c->dwsf, &c->dbsf, &c->dtok, &c->loss_part, &c->gtab, &c->gprefix, &c->ctab, &c->cprefix, &c->pf_pack, &c->pf_jobs};
    for (auto* b : all) b->release();
    for (auto* v : {&c->xs, &c->ln0, &c->qkv, &c->att, &c->hmid, &c->ln1, &c->u, &c->gl, &c->d_x1, &c->d_u, &c->d_mid, &c->d_qkv, &c->dws0, &c->dbs0,
                    &c->dws1, &c->dbs1})
        for (auto& b : *v) b.release();
    for (hipStream_t st : {c->stream, c->comm_stream, c->side_stream, c->copy_stream[0], c->copy_stream[1]})
        if (st) (void)hipStreamDestroy(st);
    delete c;
    return 0;
}

int arp_dt_num_params(arp_dt* c, int64_t* total, int32_t* n_tensors) {
    if (!c) return fail("null handle");
    size_t n = 0;
    for (auto& pi : c->infos) n += pi.size;
    if (total) *total = (int64_t)n;
    if (n_tensors) *n_tensors = (int32_t)c->infos.size();
    return 0;
}

int arp_dt_param_info(arp_dt* c, int i, char* name_buf, int name_len, int64_t* shape4, int32_t* ndim) {
    if (!c || i < 0 || i >= (int)c->infos.size() || !name_buf || !shape4 || !ndim) return fail("bad argument");
    const ParamInfo& pi = c->infos[i];
    if ((int)pi.name.size() + 1 > name_len) return fail("name buffer too small");
    memcpy(name_buf, pi.name.c_str(), pi.name.size() + 1);
    *ndim = (int32_t)pi.shape.size();
    for (size_t d = 0; d < pi.shape.size() && d < 4; ++d) shape4[d] = pi.shape[d];
    return 0;
}

// which: 0 = params, 1 = grads, 2 = adam mu, 3 = adam nu.  Host data is in the Flax layout ([in, out] kernels).
static int tensor_io(arp_dt* c, const char* name, int which, float* host, int write) {
    if (!c || !name || !host) return fail("null argument");
    auto it = c->index.find(name);
    if (it == c->index.end()) return fail(std::string("unknown parameter: ") + name);
    const ParamInfo& pi = c->infos[it->second];
    DevBuf* bufs[] = {&c->params, &c->grads, &c->mu, &c->nu};
    if (which < 0 || which > 3) return fail("bad tensor selector");
    float* dev = bufs[which]->as<float>() + pi.off;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<float> tmp(pi.size);
    if (write) {
        if (pi.dense) {
            for (int i = 0; i < pi.in; ++i)
                for (int o = 0; o < pi.out; ++o) tmp[(size_t)o * pi.in + i] = host[(size_t)i * pi.out + o];
        } else {
            memcpy(tmp.data(), host, pi.size * 4);
        }
        ARP_HIP_OK(hipMemcpy(dev, tmp.data(), pi.size * 4, hipMemcpyHostToDevice));
        if (which == 0) c->shadows_stale = c->mirror_stale = true;
    } else {
        ARP_HIP_OK(hipMemcpy(tmp.data(), dev, pi.size * 4, hipMemcpyDeviceToHost));
        if (which == 1 && c->grads_summed && c->cfg.world > 1) {  // after a data-parallel step the buffer holds the SUM over ranks; the
            const float inv = 1.0f / (float)c->cfg.world;         // getter returns what the optimizer consumed: the rank mean
            for (auto& v : tmp) v *= inv;
        }
        if (pi.dense) {
            for (int i = 0; i < pi.in; ++i)
                for (int o = 0; o < pi.out; ++o) host[(size_t)i * pi.out + o] = tmp[(size_t)o * pi.in + i];
        } else {
            memcpy(host, tmp.data(), pi.size * 4);
        }
    }
    return 0;
}
int arp_dt_set_tensor(arp_dt* c, const char* name, int which, const float* data) { return tensor_io(c, name, which, const_cast<float*>(data), 1); }
int arp_dt_get_tensor(arp_dt* c, const char* name, int which, float* out) { return tensor_io(c, name, which, out, 0); }
int arp_dt_set_step(arp_dt* c, int64_t step) {
    if (!c || step < 0) return fail("bad argument");
    c->step = step;
    return 0;
}
int arp_dt_get_step(arp_dt* c, int64_t* step) {
    if (!c || !step) return fail("bad argument");
    *step = c->step;
    return 0;
}

// Host -> device copy of one batch into slot `si` on stream `st`.  Touches only the slot's own buffers (the caller may be a
// prefetch thread working beside a running step on the other slot); frames != nullptr: the encoder-in-front boundary (row N1).
static int stage_slot(arp_dt* c, int si, hipStream_t st, const float* enc, const float* frames, const int32_t* action, const float* rtg, int B) {
    const int R = B * c->cfg.window;
    for (int i = 0; i < R; ++i)
        if (action[i] < 0 || action[i] >= c->cfg.n_actions) return fail("action id out of range");
    arp_dt::BatchSlot& b = c->bt[si];
    ARP_TRY(b.action.ensure((size_t)R * 4));
    ARP_TRY(b.rtg.ensure((size_t)R * 4));
    const size_t Mx = (size_t)R * c->cfg.enc_tokens;
    ARP_TRY(b.enc32.ensure(Mx * c->cfg.enc_dim * 4));  // with frames in: the encoder's output buffer
    if (frames) {
        int tokens = 0, width = 0, res = 0, dev = 0;
        ARP_TRY(enc_geometry(c->enc, &tokens, &width, &res, &dev));
        const size_t fb = (size_t)res * res * 3 * 4;
        ARP_TRY(b.img32.ensure((size_t)R * fb));
        ARP_HIP_OK(hipMemcpyAsync(b.img32.p, frames, (size_t)R * fb, hipMemcpyHostToDevice, st));
    } else {
        ARP_HIP_OK(hipMemcpyAsync(b.enc32.p, enc, Mx * c->cfg.enc_dim * 4, hipMemcpyHostToDevice, st));
    }
    ARP_HIP_OK(hipMemcpyAsync(b.action.p, action, (size_t)R * 4, hipMemcpyHostToDevice, st));
    ARP_HIP_OK(hipMemcpyAsync(b.rtg.p, rtg, (size_t)R * 4, hipMemcpyHostToDevice, st));
    b.B = B;
    b.images = frames != nullptr;
    return 0;
}

int arp_dt_set_batch(arp_dt* c, const float* enc, const int32_t* action, const float* rtg, int B) {
    if (!c || !enc || !action || !rtg || B <= 0) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    c->cur = 2;  // the synchronous slot
    if (c->bt[2].enc_ahead && c->enc_stream) ARP_HIP_OK(hipStreamSynchronize(c->enc_stream));  // (an encode-ahead pass still writing the buffer about to be filled)
    c->bt[2].enc_ahead = false;
    ARP_TRY(stage_slot(c, 2, c->stream, enc, nullptr, action, rtg, B));
    ARP_TRY(ensure_buffers(c, B));
    c->use_images = false;
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

// Asynchronous upload of a batch into slot 0 / 1 on the handle's COPY stream -- main_procgen.py:703's prefetch_to_device(..., 2).
// Returns once the copies are enqueued (from pageable host memory: once they are staged); never touches the slot a running step
// reads, so it may be called from another host thread while arp_dt_train_step runs.  The copy waits (on the GPU) for the last
// step that read this slot; arp_dt_select_batch makes the compute stream wait for the copy.
static int upload_async(arp_dt* c, int slot, const float* enc, const float* frames, const int32_t* action, const float* rtg, int B) {
    if (!c || (!enc && !frames) || !action || !rtg || B <= 0 || slot < 0 || slot > 1) return fail("bad argument");
    if (frames && !c->enc) return fail("no encoder attached: call arp_dt_attach_encoder first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    arp_dt::BatchSlot& b = c->bt[slot];
    // Behind the slot's last reader: a HOST-side wait on the event that step recorded (already complete when the caller follows
    // prefetch_to_device's protocol: a slot is handed back after its step's aux was read).  Not a stream-side wait: from this thread
    // hipStreamWaitEvent on an event of a stream that is being captured is refused, and a pageable upload queued behind a pending
    // stream wait ran 4x slower (7.8 ms instead of 1.8 ms per 101 MB).
    // Never beside a graph capture on the compute thread (fwd_bwd_graphed): HIP refuses hipEventSynchronize on an event whose stream is
    // capturing at that moment -- even one recorded long before the capture began -- and that refusal also invalidates the capture.
    std::lock_guard<std::mutex> capture_lock(c->capture_mu);
    if (b.used) ARP_HIP_OK(hipEventSynchronize(b.use));
    // a slot that has to GROW frees its old buffers (hipFree synchronises the device): size them once, with the largest batch
    // ONE copy stream for both slots (round 6; two until then): the uploads share the PCIe link anyway, and every stream of a process is a hardware queue -- with
    // the two copy streams the encoder-inside step held five (compute, 2 x copy, encoder, encoder part) and ran 10.3 -> 11.9 ms per step when the fifth queue came
    // to share a dispatch pipe with a busy one (profiles/r6_n1_flow.txt: same box, GPU_MAX_HW_QUEUES 4 / 8 / 16 and creation orders); four is what the chip runs side by side
    if (!c->copy_stream[0]) ARP_HIP_OK(hipStreamCreateWithFlags(&c->copy_stream[0], hipStreamNonBlocking));
    ARP_TRY(stage_slot(c, slot, c->copy_stream[0], enc, frames, action, rtg, B));
    ARP_HIP_OK(hipEventRecord(b.up, c->copy_stream[0]));
    b.up_pending = true;
    b.up_recorded = true;
    b.enc_ahead = false;  // (new frames: whatever enc32 holds belongs to the batch before)
    return 0;
}
int arp_dt_upload_batch_async(arp_dt* c, int slot, const float* enc, const int32_t* action, const float* rtg, int B) {
    return upload_async(c, slot, enc, nullptr, action, rtg, B);
}
int arp_dt_upload_batch_images_async(arp_dt* c, int slot, const float* images, const int32_t* action, const float* rtg, int B) {
    return upload_async(c, slot, nullptr, images, action, rtg, B);
}

// The next forward / step reads slot `slot` (compute stream ordered behind the slot's upload).
int arp_dt_select_batch(arp_dt* c, int slot) {
    if (!c || slot < 0 || slot > 1) return fail("bad argument");
    arp_dt::BatchSlot& b = c->bt[slot];
    if (b.B <= 0) return fail("nothing was uploaded into this batch slot");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (b.up_pending) {
        ARP_HIP_OK(hipStreamWaitEvent(c->stream, b.up, 0));
        b.up_pending = false;
    }
    c->cur = slot;
    ARP_TRY(ensure_buffers(c, b.B));
    c->use_images = b.images;
    return 0;
}

int arp_dt_attach_encoder(arp_dt* c, arp_enc* enc) {
    if (!c || !enc) return fail("null argument");
    int tokens = 0, width = 0, res = 0, dev = 0;
    ARP_TRY(enc_geometry(enc, &tokens, &width, &res, &dev));
    if (tokens != c->cfg.enc_tokens || width != c->cfg.enc_dim) return fail("encoder geometry does not match enc_tokens / enc_dim");
    if (dev != c->cfg.device) return fail("encoder lives on another device");
    c->enc = enc;
    return 0;
}

int arp_dt_set_batch_images(arp_dt* c, const float* images, const int32_t* action, const float* rtg, int B) {
    if (!c || !images || !action || !rtg || B <= 0) return fail("bad argument");
    if (!c->enc) return fail("no encoder attached: call arp_dt_attach_encoder first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    c->cur = 2;  // the synchronous slot
    if (c->bt[2].enc_ahead && c->enc_stream) ARP_HIP_OK(hipStreamSynchronize(c->enc_stream));  // (an encode-ahead pass still reading the frames about to be replaced)
    c->bt[2].enc_ahead = false;
    ARP_TRY(stage_slot(c, 2, c->stream, nullptr, images, action, rtg, B));
    ARP_TRY(ensure_buffers(c, B));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    c->use_images = true;
    return 0;
}

// Encode batch slot `slot` (0 / 1: a batch of frames uploaded with arp_dt_upload_batch_images_async; 2: the batch staged by arp_dt_set_batch_images) NOW, on the
// encoder's own stream, instead of at the head of the step that will read it: the frozen encoder's pass for batch i + 1 then runs beside step i's policy
// part.  Ordered on the GPU behind the slot's upload and behind the last step that read the slot; the step that selects the slot waits for the pass.  May be
// called from the uploader thread (it takes the capture lock, like the uploads).  A slot that is re-selected without a new upload and without another call
// of this function is encoded again by its step: no step ever reads encodings it did not pay for.
int arp_dt_encode_ahead(arp_dt* c, int slot) {
    if (!c || slot < 0 || slot > 2) return fail("bad argument");
    if (!c->enc) return fail("no encoder attached: call arp_dt_attach_encoder first");
    if (!c->enc_eager) return fail("arp_dt_encode_ahead needs the eager encoder path (ARP_DT_ENC_EAGER=0 captures the encoder inside the step's graph)");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    std::lock_guard<std::mutex> lock(c->capture_mu);
    return enqueue_encode(c, slot, slot == 2);
}

int arp_dt_forward(arp_dt* c, float* action_logits, float* return_pred, float* metrics) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_dt_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    // (replayed as a hipGraph from the third call of a geometry on: ~25 dependent launches, host-bound at batch 1)
    ARP_TRY(c->cfg.mode == ARP_MODE_BF16 ? fwd_bwd_graphed<bf16_t>(c, 3) : (c->cfg.mode == ARP_MODE_F16 ? fwd_bwd_graphed<f16_t>(c, 3) : fwd_bwd_graphed<float>(c, 3)));
    const int R = c->R();
    if (action_logits) ARP_HIP_OK(hipMemcpyAsync(action_logits, c->logits.p, (size_t)R * c->cfg.n_actions * 4, hipMemcpyDeviceToHost, c->stream));
    if (return_pred) ARP_HIP_OK(hipMemcpyAsync(return_pred, c->ret.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->stream));
    if (metrics) ARP_HIP_OK(hipMemcpyAsync(metrics, c->metrics.p, 16, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_dt_backward(arp_dt* c) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_dt_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_BF16) { ARP_TRY(forward<bf16_t>(c, true)); ARP_TRY(backward<bf16_t>(c)); }
    else if (c->cfg.mode == ARP_MODE_F16) { ARP_TRY(forward<f16_t>(c, true)); ARP_TRY(backward<f16_t>(c)); }
    else { ARP_TRY(forward<float>(c, true)); ARP_TRY(backward<float>(c)); }
    ARP_TRY(l2_penalty(c));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_dt_train_step_async(arp_dt* c, float lr) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_dt_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_F16) return step_impl<f16_t>(c, lr, nullptr);
    return c->cfg.mode == ARP_MODE_BF16 ? step_impl<bf16_t>(c, lr, nullptr) : step_impl<float>(c, lr, nullptr);
}

int arp_dt_train_step(arp_dt* c, float lr, float* aux) {
    if (!c || !aux) return fail("null argument");
    if (c->B <= 0) return fail("no batch staged: call arp_dt_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_F16) return step_impl<f16_t>(c, lr, aux);
    return c->cfg.mode == ARP_MODE_BF16 ? step_impl<bf16_t>(c, lr, aux) : step_impl<float>(c, lr, aux);
}

// create_val_step's val_step_fn (main_procgen.py:144-169): forward only, then pmean over the ranks of the four metrics
// aux4 = {loss, trans_loss, return_loss, acc * 100} (the reference's dict order, :154-159).
int arp_dt_val_step(arp_dt* c, float* aux4) {
    if (!c || !aux4) return fail("null argument");
    if (c->B <= 0) return fail("no batch staged: call arp_dt_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(c->cfg.mode == ARP_MODE_BF16 ? fwd_bwd_graphed<bf16_t>(c, 3) : (c->cfg.mode == ARP_MODE_F16 ? fwd_bwd_graphed<f16_t>(c, 3) : fwd_bwd_graphed<float>(c, 3)));
    const bool comm = c->has_comm && (c->cfg.world > 1 || c->force_comm);
    if (comm && rccl_api()->AllReduce(c->metrics.p, c->metrics.p, 4, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess)
        return fail("ncclAllReduce(metrics) failed");
    ARP_TRY(mark_slot_read(c));
    float m[4];
    ARP_HIP_OK(hipMemcpyAsync(m, c->metrics.p, 16, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    const float inv = 1.0f / (float)std::max(c->cfg.world, 1);
    aux4[0] = m[0] * inv;          // loss (no L2 penalty: val_fn returns the model's loss)
    aux4[1] = m[2] * inv;          // trans_loss
    aux4[2] = m[3] * inv;          // return_loss
    aux4[3] = m[1] * inv * 100.f;  // acc * 100
    return 0;
}

// The flat-gradient ranges of the data-parallel step's two all-reduce buckets, from the configuration alone (no GPU):
// ranges8 = {lo, hi} x {bucket 1 range a, bucket 1 range b, bucket 2 range a, bucket 2 range b} in floats; *total = P.
int arp_dt_bucket_plan(const arp_dt_cfg* cfg, int64_t* ranges8, int64_t* total) {
    if (!cfg || !ranges8 || !total) return fail("null argument");
    arp_dt tmp;
    tmp.cfg = *cfg;
    build_layout(&tmp);
    const BucketPlan b = bucket_plan(&tmp);
    for (int i = 0; i < 4; ++i) {
        ranges8[2 * i] = (int64_t)b.lo[i];
        ranges8[2 * i + 1] = (int64_t)b.hi[i];
    }
    *total = (int64_t)tmp.P;
    return 0;
}

int arp_dt_sync(arp_dt* c) {
    if (!c) return fail("null handle");
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_dt_event_record(arp_dt* c, arp_event* e) {
    if (!c || !e) return fail("null argument");
    ARP_HIP_OK(hipEventRecord(e->e, c->stream));
    return 0;
}

int arp_dt_comm_unique_id(void* id128) {
    if (!id128) return fail("null argument");
    if (!rccl_api()) return fail("librccl.so.1 could not be loaded");
    ncclUniqueId id;
    if (ncclResult_t r = rccl_api()->GetUniqueId(&id); r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(ncclUniqueId) == 128, "unexpected ncclUniqueId size");
    memcpy(id128, &id, 128);
    return 0;
}

int arp_dt_comm_init(arp_dt* c, const void* id128, int world, int rank) {
    if (!c || !id128 || world <= 0 || rank < 0 || rank >= world) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    if (!rccl_api()) return fail("librccl.so.1 could not be loaded");
    // the communication stream exists BEFORE RCCL sets up its own queues: created lazily at the first staged step instead (behind them) the staged step ran 1.54 ms
    // where it runs 0.80 (profiles/r6_n1_flow.txt, last block) -- which hardware queue a stream lands on follows the order of creation (runtime.h::prime_runtime)
    if (!c->comm_stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (ncclResult_t r = rccl_api()->CommInitRank(&c->comm, world, id, rank); r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    c->has_comm = true;
    c->cfg.world = world;
    c->cfg.rank = rank;
    return 0;
}

int arp_dt_comm_info(arp_dt* c, int32_t* info5) {
    if (!c || !info5) return fail("null argument");
    return rccl_comm_info(c->comm, c->has_comm, c->cfg.device, info5);
}
int arp_dt_comm_selfcheck(arp_dt* c, double* sum) {
    if (!c || !sum) return fail("null argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(c->scal.ensure(4096 * 4));
    return rccl_selfcheck(c->comm, c->has_comm, c->stream, c->scal.as<float>(), c->cfg.rank, sum);
}

// Test hook (scripts/adapter_chain_probe.py, tests): the bytes of one of the step's intermediate device buffers after a forward, by name -- "Xc" / "H1c" the adapter's
// [hi | x4 | dx4] operand rows, "W1c" / "W2c" its packed weights, "wc_scal" their scales (16 ints), "A32" / "A" / "Adx" its output as the mix reads it, "Y" the mix,
// "img" the image embedding.  Copies min(bytes, the buffer's size) bytes; returns the buffer's size.
int64_t arp_dt_debug_read(arp_dt* c, const char* name, void* out, int64_t bytes) {
    if (!c || !name) { fail("null argument"); return -1; }
    const std::string n = name;
    const DevBuf* b = n == "Xc" ? &c->Xc : n == "H1c" ? &c->H1c : n == "W1c" ? &c->W1c : n == "W2c" ? &c->W2c : n == "wc_scal" ? &c->wc_scal : n == "A32" ? &c->A32 : n == "A" ? &c->A
                      : n == "Adx" ? &c->Adx : n == "Y" ? &c->Y : n == "img" ? &c->img : n == "H1" ? &c->H1 : n == "Xb" ? &c->Xb : nullptr;
    if (!b) { fail("unknown buffer name"); return -1; }
    if (hipSetDevice(c->cfg.device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { fail("device"); return -1; }
    if (out && bytes > 0 && b->p) {
        if (hipMemcpy(out, b->p, (size_t)std::min<int64_t>(bytes, (int64_t)b->bytes), hipMemcpyDeviceToHost) != hipSuccess) { fail("copy"); return -1; }
    }
    return (int64_t)b->bytes;
}

int arp_dt_set_adapter_corrections(arp_dt* c, int on) {
    if (!c) return fail("null handle");
    if (on && c->cfg.mode != ARP_MODE_F16) return fail("adapter corrections exist in ARP_MODE_F16 only (binary16 products corrected on the fp4 MFMA)");
    if (on && (c->cfg.enc_dim % 256 || c->cfg.enc_dim < 512)) return fail("adapter corrections need enc_dim to be a multiple of 256 and >= 512");
    c->adapter_c = on != 0;
    c->shadows_stale = true;  // the packed [W_hi | dW4 | W4] weights are built by refresh_shadows
    return 0;
}

// sync_state_fn (main_procgen.py:94-101): every rank takes rank 0's params and optimizer state
int arp_dt_broadcast_state(arp_dt* c) {
    if (!c) return fail("null handle");
    if (!c->has_comm) return 0;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    DevBuf* fb[] = {&c->params, &c->mu, &c->nu};
    for (auto* b : fb)
        if (ncclResult_t r = rccl_api()->Broadcast(b->p, b->p, c->P, ncclFloat, 0, c->comm, c->stream); r != ncclSuccess)
            return rccl_fail("ncclBroadcast", r);
    // ... and its step counter (rank 0's whole TrainState, main_procgen.py:94-101: optax's count drives the bias correction and
    // the learning-rate schedule): 8 bytes through the scratch buffer
    ARP_TRY(c->scal.ensure(4096 * 4));
    long long st = c->step;
    ARP_HIP_OK(hipMemcpyAsync(c->scal.p, &st, 8, hipMemcpyHostToDevice, c->stream));
    if (ncclResult_t r = rccl_api()->Broadcast(c->scal.p, c->scal.p, 8, ncclChar, 0, c->comm, c->stream); r != ncclSuccess) return rccl_fail("ncclBroadcast(step)", r);
    ARP_HIP_OK(hipMemcpyAsync(&st, c->scal.p, 8, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    c->step = st;
    c->shadows_stale = c->mirror_stale = true;
    return 0;
}

// ---- single-kernel entry points of the train step's own kernels (host buffers; tests/test_ops_gpu.py) ----------------------------
}  // extern "C"

namespace {
// host f32 -> device operand type T (rounded on the device by the step's own conversion kernel)
template <typename T> int upload_as(DevBuf& stage, DevBuf& dst, const float* src, size_t n) {
    ARP_TRY(stage.ensure(n * 4));
    ARP_TRY(dst.ensure(n * sizeof(T)));
    ARP_HIP_OK(hipMemcpy(stage.p, src, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((convert_kernel<T>), dim3(cdiv(n, 1024)), dim3(256), 0, nullptr, stage.as<float>(), dst.as<T>(), n);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}
template <typename T> __global__ void widen_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = Elem<T>::ld(in + i);
}
template <typename T> int download_from(DevBuf& stage, const DevBuf& src, float* dst, size_t n) {
    ARP_TRY(stage.ensure(n * 4));
    hipLaunchKernelGGL((widen_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, nullptr, src.as<T>(), stage.as<float>(), n);
    ARP_HIP_OK(hipGetLastError());
    ARP_HIP_OK(hipMemcpy(dst, stage.p, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

template <typename T> int op_gemm_tn(int tile256, int ksplit, const float* A, const float* B, float* out, int M, int N, int K, float alpha) {
    DevBuf st, dA, dB, dP, dO;
    auto body = [&]() -> int {
        ARP_TRY(upload_as<T>(st, dA, A, (size_t)K * M));
        ARP_TRY(upload_as<T>(st, dB, B, (size_t)K * N));
        const size_t MN = (size_t)M * N;
        ARP_TRY(dP.ensure((size_t)ksplit * MN * 4));
        ARP_TRY(dO.ensure(MN * 4));
        GemmTnArgs g;
        g.A = dA.p; g.B = dB.p; g.M = M; g.N = N; g.K = K; g.lda = M; g.ldb = N; g.ldo = N; g.ksplit = ksplit; g.tile256 = tile256;
        g.xcd_slices = tile256 && ksplit % 8 == 0;
        const int tcode = __is_same(T, bf16_t) ? 1 : 2;
        if (ksplit == 1) {
            g.out = dO.as<float>(); g.slice_stride = 0; g.alpha = alpha;
            ARP_TRY(launch_gemm_tn(tcode, g, nullptr));
        } else {
            g.out = dP.as<float>(); g.slice_stride = MN; g.alpha = 1.f;
            ARP_TRY(launch_gemm_tn(tcode, g, nullptr));
            launch_splitk_reduce<float>(nullptr, dP.as<float>(), ksplit, MN, N, nullptr, ACT_NONE, dO.as<float>(), nullptr, 0, alpha);
            ARP_HIP_OK(hipGetLastError());
        }
        ARP_HIP_OK(hipMemcpy(out, dO.p, MN * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    st.release(); dA.release(); dB.release(); dP.release(); dO.release();
    return rc;
}

template <typename T>
int op_gemm_relu_bwd(const float* A, const float* W, const float* mask, float* out, float* colsum, int M, int N, int K) {
    DevBuf st, dA, dW, dM, dO, dC, dS;
    auto body = [&]() -> int {
        ARP_TRY(upload_as<T>(st, dA, A, (size_t)M * K));
        ARP_TRY(upload_as<T>(st, dW, W, (size_t)N * K));
        ARP_TRY(upload_as<T>(st, dM, mask, (size_t)M * N));
        const int mt = cdiv(M, 256);
        ARP_TRY(dO.ensure((size_t)M * N * sizeof(T)));
        ARP_TRY(dC.ensure((size_t)mt * N * 4));
        ARP_TRY(dS.ensure((size_t)N * 4));
        GemmArgs g;
        g.A = dA.p; g.W = dW.p; g.out = dO.p; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N;
        g.mask = dM.p; g.ldm = N; g.colsum_part = dC.as<float>();
        ARP_TRY((launch_gemm256_nt<T, T, ACT_NONE, false, SITE_DT>(g, nullptr)));
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(256), 0, nullptr, dC.as<float>(), mt, N, dS.as<float>(), 1.f);
        ARP_HIP_OK(hipGetLastError());
        ARP_TRY(download_from<T>(st, dO, out, (size_t)M * N));
        ARP_HIP_OK(hipMemcpy(colsum, dS.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    st.release(); dA.release(); dW.release(); dM.release(); dO.release(); dC.release(); dS.release();
    return rc;
}

template <typename T>
int op_adapter_dy(const float* dz, const float* Wi, const float* A, const float* x, float rw, float* dApre, float* colsum, float* dres, int R, int E,
                  int tokens, int D) {
    const size_t Kin = (size_t)tokens * D;
    DevBuf st, ddz, dWi, dA, dx, drw, dO, dC, dP, dS;
    auto body = [&]() -> int {
        ARP_TRY(upload_as<T>(st, ddz, dz, (size_t)R * E));
        ARP_TRY(upload_as<T>(st, dWi, Wi, (size_t)E * Kin));
        ARP_TRY(upload_as<T>(st, dA, A, (size_t)R * Kin));
        ARP_TRY(dx.ensure((size_t)R * Kin * 4));
        ARP_HIP_OK(hipMemcpy(dx.p, x, (size_t)R * Kin * 4, hipMemcpyHostToDevice));
        ARP_TRY(drw.ensure(16));
        ARP_HIP_OK(hipMemcpy(drw.p, &rw, 4, hipMemcpyHostToDevice));
        const int nrb = adapter_dy_row_blocks(R), nct = (int)(Kin / 128);
        ARP_TRY(dO.ensure((size_t)R * Kin * sizeof(T)));
        ARP_TRY(dC.ensure((size_t)nrb * tokens * D * 4));
        ARP_TRY(dP.ensure((size_t)nrb * nct * 4));
        ARP_TRY(dS.ensure((size_t)(D + 4) * 4));
        AdapterDyArgs a;
        a.dz = ddz.p; a.Wi = dWi.p; a.A = dA.p; a.x32 = dx.as<float>(); a.rw = drw.as<float>(); a.dApre = dO.p; a.colpart = dC.as<float>();
        a.dres_part = dP.as<float>(); a.R = R; a.E = E; a.Kin = (int)Kin; a.D = D;
        ARP_TRY(launch_adapter_dy(__is_same(T, bf16_t) ? 1 : 2, a, nullptr));
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, nullptr, dC.as<float>(), nrb * tokens, D, dS.as<float>(), 1.f);
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, nullptr, dP.as<float>(), nrb * nct, 1.f, dS.as<float>() + D, 0);
        ARP_HIP_OK(hipGetLastError());
        ARP_TRY(download_from<T>(st, dO, dApre, (size_t)R * Kin));
        ARP_HIP_OK(hipMemcpy(colsum, dS.p, (size_t)D * 4, hipMemcpyDeviceToHost));
        ARP_HIP_OK(hipMemcpy(dres, dS.as<float>() + D, 4, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    for (DevBuf* b : {&st, &ddz, &dWi, &dA, &dx, &drw, &dO, &dC, &dP, &dS}) b->release();
    return rc;
}
}  // namespace

extern "C" {

// C[M,N] = alpha * sum_k A[k,m] B[k,n] on the TN weight-gradient kernels (gemm_tn.h): A [K,M], B [K,N] row-major, rounded to the
// 16-bit operand type of `mode`; tile256 = 0 (128 x 128 tiles: M, N % 128, K % 64) or 1 (256 x 256 tiles: M, N % 256, K % 64);
// ksplit > 1 runs the split-K slabs + fixed-order reduction of the train step.
int arp_op_gemm_tn(int mode, int tile256, int ksplit, const float* A, const float* B, float* out, int M, int N, int K, float alpha) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || ksplit < 1) return fail("bad argument");
    if (mode == ARP_MODE_F16) return op_gemm_tn<f16_t>(tile256, ksplit, A, B, out, M, N, K, alpha);
    if (mode == ARP_MODE_BF16) return op_gemm_tn<bf16_t>(tile256, ksplit, A, B, out, M, N, K, alpha);
    return fail("arp_op_gemm_tn: 16-bit modes only");
}
// out[M,N] = (A[M,K] . W[N,K]^T) * (mask[M,N] > 0) in the operand type (returned widened), colsum[N] = column sums of the stored
// values: the ReLU-backward epilogue of the 256 x 256 GEMM (gemm256.h, GemmArgs::mask).  N % 8 == 0, K % 64 == 0.
int arp_op_gemm_relu_bwd(int mode, const float* A, const float* W, const float* mask, float* out, float* colsum, int M, int N, int K) {
    if (!A || !W || !mask || !out || !colsum || M <= 0 || N <= 0 || K <= 0) return fail("bad argument");
    if (mode == ARP_MODE_F16) return op_gemm_relu_bwd<f16_t>(A, W, mask, out, colsum, M, N, K);
    if (mode == ARP_MODE_BF16) return op_gemm_relu_bwd<bf16_t>(A, W, mask, out, colsum, M, N, K);
    return fail("arp_op_gemm_relu_bwd: 16-bit modes only");
}
// The fused adapter-backward pass (adapter_bwd.h): dApre[R, tokens*D] = sigmoid(rw) * (dz[R,E] . Wi[E, tokens*D]) * (A > 0) in the
// operand type (returned widened), colsum[D] = its sums over rows and tokens, dres = sum (dz . Wi) * (A - x).
int arp_op_adapter_dy(int mode, const float* dz, const float* Wi, const float* A, const float* x, float rw, float* dApre, float* colsum, float* dres,
                      int R, int E, int tokens, int D) {
    if (!dz || !Wi || !A || !x || !dApre || !colsum || !dres || R <= 0 || tokens <= 0) return fail("bad argument");
    if (!adapter_dy_supported(E, D, (long long)tokens * D)) return fail("arp_op_adapter_dy: unsupported geometry (E in {32, 64, 128}, D % 128 == 0)");
    if (mode == ARP_MODE_F16) return op_adapter_dy<f16_t>(dz, Wi, A, x, rw, dApre, colsum, dres, R, E, tokens, D);
    if (mode == ARP_MODE_BF16) return op_adapter_dy<bf16_t>(dz, Wi, A, x, rw, dApre, colsum, dres, R, E, tokens, D);
    return fail("arp_op_adapter_dy: 16-bit modes only");
}

int arp_dt_profile_enable(arp_dt* c, int on) {
    if (!c) return fail("null handle");
    c->prof.on = on != 0;
    return 0;
}
int arp_dt_profile_reset(arp_dt* c) {
    if (!c) return fail("null handle");
    c->prof.reset();
    return 0;
}
int arp_dt_profile_json(arp_dt* c, char* buf, int buf_len) {
    if (!c || !buf) return fail("null argument");
    const std::string s = c->prof.json();
    if ((int)s.size() + 1 > buf_len) return fail("profile buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"

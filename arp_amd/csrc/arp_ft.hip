// the CLIP multi-scale adapter fine-tune step (BASELINE.json configs[4]) -- the trainable head on top of
// the frozen CLIP towers, as one HIP train step behind the C ABI (include/arp_hip.h, arp_ft_*).
//
// Reference: finetune_module/clip_multiscale_adapter.py (CLIPMultiscaleAdapter.encode_image :134-149, encode_text
// :151-175, forward :177-250), finetune_module/layers.py:6-60 (AdapterMLP), finetune_module/finetune.py:139-141
// (CLIP frozen; torch.optim.AdamW over everything else).  The frozen towers' outputs -- per-block CLS / EOT features
// (detached by the reference's hooks, utils.py:6-18) and the final CLIP features -- are this step's INPUTS, exactly as
// BASELINE.json configs[3] feeds the policy step with encodings.
//
// Per step (B samples, 3 frames each; Mi = 3B image rows, Mt = B text rows; F = layers*width_t + embed):
//   U  = X Wint^T (bias-free)            f = [U | clip feature]
//   A  = relu(f W1^T + b1) W2^T + b2     y = res f + (1 - res) A      a = y / ||y||
//   scores, VIP loss, C = [a1 | t | a2 | t] -> relu(C V1^T + c1) V2^T + c2 -> CE;  loss = vip + lambda_id * id
// Every contraction with a large weight is an NT MFMA GEMM (gemm.h / gemm256.h).  The row counts are tiny (<= 192), so
// forward / dX GEMMs are weight-streaming bound and run split-K to fill the chip, and the weight-gradient GEMMs
// (dW = dY^T X, contraction over <= 192 rows) are output-write bound.  Parameters use torch's own names and [out, in]
// layout, which is already the NT operand layout.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/arp_hip.h"
#include "common.h"
#include "dtops.h"
#include "ftops.h"
#include "gemm.h"
#include "gemm256.h"
#include "gemm_tn.h"
#include "rccl_dl.h"
#include "runtime.h"

using namespace arp;

namespace {
struct FtParam {
    std::string name;
    std::vector<int64_t> shape;
    size_t off = 0, size = 0;
};
inline int cdiv(size_t a, size_t b) { return (int)((a + b - 1) / b); }
enum { SITE_FT = 24 };
}  // namespace

struct arp_ft {
    arp_ft_cfg cfg;
    hipStream_t stream = nullptr;
    std::vector<FtParam> infos;
    std::map<std::string, int> index;
    size_t P = 0;
    DevBuf params, grads, mu, nu;
    long long step = 0;
    // AdamW inside the weight-gradient GEMMs (gemm.h, GEMM_SITE_ADAMW): in a single-process step the seven big dW products update their
    // weights from the epilogue instead of storing 1.9 GB of gradient for a separate pass to read back (ARP_FT_FUSE_ADAM=0 disables;
    // never with a communicator: the gradient has to be all-reduced first; never in arp_ft_backward: its caller reads the gradients)
    bool fuse_adam = true, fuse_now = false;
    float fuse_lr = 0.f, fuse_bc1 = 1.f, fuse_bc2 = 1.f;
    std::vector<std::pair<size_t, size_t>> fused;  // flat [lo, hi) ranges already updated this step
    std::vector<std::pair<size_t, size_t>> grads_gone;  // ... of the LAST step: their gradients were consumed inside the GEMMs, never stored
    bool shadows_stale = true;   // the host wrote parameters: rebuild the bf16 mirror from f32
    bool transposed_stale = true; // parameters moved (host write or AdamW): rebuild the transposed shadows
    int B = 0;
    // operand-type weight shadows: forward layout [out, in] (aliases the f32 parameters in f32 mode) and, where the
    // backward needs dX, the transposed layout [in, out]
    DevBuf dropped;  // one 32-bit device counter: non-finite gradient elements AdamW treated as missing (f16 mode only) since the last read
    uint64_t dropped_total = 0;  // ... drained into this 64-bit total at every arp_ft_dropped_gradients (the device word cannot wrap between two reads of a sane run)
    DevBuf mirror;  // bf16 mode: bf16 copy of the flat parameter vector (same offsets) = every forward-layout operand
    DevBuf sW1t[2], sW2t[2], sV1t;
    // 16-bit modes: dX = dY . W on the "NN" kernel (gemm_tn.h: W read AS STORED through the transposing LDS read), every dX ahead of its layer's
    // weight-gradient GEMM (whose fused AdamW epilogue moves the weight): no transposed shadows, no ft.refresh_shadows.  ARP_FT_NN=0: round 3's NT path.
    bool nn_dx = true;
    // inputs
    DevBuf x_in[2], x_fin[2], r, action;
    // per tower (0 = image rows Mi, 1 = text rows Mt)
    DevBuf X[2], XT[2], f[2], fT_[2], fTt[2], H[2], HT[2], A[2], a[2], nrm[2], da[2], dA[2], dAT_[2], dAt[2], dfd[2], dH[2], dHp[2], dHpt[2], df[2], dUt[2],
        dres_part[2];
    DevBuf scores, ds, C, CT_, Ct, Hinv, logits, dlogits, dHinv, dHinvT_, dHinvt, dC, metrics, scal, part;
    Profiler prof;
    // data parallelism (BASELINE configs[4]: DP = 8): one process per GPU, one all-reduce(sum) of the flat gradient per step
    ncclComm_t comm = nullptr;
    bool has_comm = false;
    int world = 1, rank = 0;
    bool grads_summed = false;  // the gradient buffer holds the all-reduced SUM over ranks (set by a data-parallel step)
    // The 1.9 GB gradient goes out in SEVEN buckets in the order the backward produces them (inverse model; per tower: second adapter
    // layer, first adapter layer, intermediate linear), each on the communication stream as soon as its last weight-gradient GEMM is
    // enqueued, so that all but the last bucket's all-reduce runs beside the remaining backward (ft_buckets, step_impl).
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_bucket[8] = {}, ev_comm = nullptr;
    bool overlap_comm = true;   // ARP_FT_OVERLAP=0: one all-reduce behind the whole backward
    bool force_comm = false;    // ARP_FT_FORCE_COMM=1: run the all-reduce path at world = 1 too (what a 1-GPU box can test)
    bool comm_live = false;     // set by step_impl around backward(): the bucket hooks fire

    size_t esz() const { return cfg.mode == ARP_MODE_F32 ? 4 : 2; }
    // f16 mode: every gradient is seeded with this power-of-two factor (ft_loss_kernel) and carries it to the f32 gradient buffer;
    // AdamW's gscale and arp_ft_get_tensor(which = 1) take it out.  bf16 / f32 have the range: 1.
    float grad_scale() const { return cfg.mode == ARP_MODE_F16 ? 1024.f : 1.f; }
    int groups() const { return cfg.goal_conditioned ? 4 : 3; }  // image groups per sample: image0..2 (+ the goal frame image3)
    DevBuf dist;  // goal_conditioned: ||a3 - a_k|| per score row
    int Dv() const { return cfg.layers * cfg.width_v; }
    int Dt() const { return cfg.layers * cfg.width_t; }
    int F() const { return cfg.layers * cfg.width_t + cfg.embed; }
    int Hd() const { return cfg.hidden * (cfg.layers + 1); }
    float* p(const std::string& n) { return params.as<float>() + infos[index.at(n)].off; }
    float* g(const std::string& n) { return grads.as<float>() + infos[index.at(n)].off; }
    size_t psize(const std::string& n) { return infos[index.at(n)].size; }
};

namespace {

void build_layout(arp_ft* c) {
    const int F = c->F(), Hd = c->Hd();
    std::vector<FtParam> v;
    auto add = [&](const std::string& name, std::vector<int64_t> shape) {
        FtParam pi;
        pi.name = name;
        pi.shape = shape;
        pi.size = 1;
        for (auto d : shape) pi.size *= (size_t)d;
        v.push_back(pi);
    };
    add("image_intermediate_linear.weight", {c->Dt(), c->Dv()});
    add("text_intermediate_linear.weight", {c->Dt(), c->Dt()});
    for (const char* a : {"image_adapter", "text_adapter"}) {
        add(std::string(a) + ".layers.0.weight", {Hd, F});
        add(std::string(a) + ".layers.0.bias", {Hd});
        add(std::string(a) + ".layers.3.weight", {F, Hd});
        add(std::string(a) + ".layers.3.bias", {F});
    }
    add("inverse_layer.layers.0.weight", {c->cfg.hidden, 4 * F});
    add("inverse_layer.layers.0.bias", {c->cfg.hidden});
    add("inverse_layer.layers.3.weight", {c->cfg.n_actions, c->cfg.hidden});
    add("inverse_layer.layers.3.bias", {c->cfg.n_actions});
    add("image_residual_weight", {});
    add("text_residual_weight", {});
    add("lambda_id", {});
    size_t off = 0;
    for (auto& pi : v) {
        pi.off = off;
        off += (pi.size + 3) & ~(size_t)3;
    }
    c->P = off;
    c->infos = v;
    for (size_t i = 0; i < v.size(); ++i) c->index[v[i].name] = (int)i;
}

template <typename TI, typename TM, typename TO>
int ft_transpose(arp_ft* c, const TI* in, int ldi, const TM* mask, TO* outN, int ldn, TO* outT, int ldt, int R, int Cc) {
    hipLaunchKernelGGL((transpose_mask_kernel<TI, TM, TO>), dim3(cdiv(Cc, 64), cdiv(R, 64)), dim3(256), 0, c->stream, in, ldi, mask, nullptr, 1.f, outN,
                       ldn, outT, ldt, R, Cc);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// Parameters that receive no gradient in this configuration (torch leaves a parameter whose .grad is None alone -- no decay, no moment
// update): ranges [lo, hi) of the flat parameter vector, padded offsets.  Used by apply_update AND by the fused AdamW epilogue's gate.
static FtSkip ft_skip_ranges(arp_ft* c) {
    FtSkip skip;
    for (int r = 0; r < FT_SKIP_RANGES; ++r) skip.lo[r] = skip.hi[r] = 0;
    int nskip = 0;
    auto skip_span = [&](const char* first, const char* last) {  // [first, end of last), padded offsets
        const FtParam& a = c->infos[c->index.at(first)];
        const FtParam& b = c->infos[c->index.at(last)];
        skip.lo[nskip] = a.off;
        skip.hi[nskip] = b.off + ((b.size + 3) & ~(size_t)3);
        ++nskip;
    };
    if (!c->cfg.use_id) {
        skip_span("inverse_layer.layers.0.weight", "inverse_layer.layers.3.bias");
        skip_span("lambda_id", "lambda_id");
    }
    if (c->cfg.goal_conditioned) {
        skip_span("text_intermediate_linear.weight", "text_intermediate_linear.weight");
        skip_span("text_adapter.layers.0.weight", "text_adapter.layers.3.bias");
        skip_span("text_residual_weight", "text_residual_weight");
    }
    return skip;
}

// out[M, N] (ldo) = act(A[M, K] . W[N, K]^T + bias) (+ resid, same view as out).  Split over K whenever the 128x128
// grid alone would leave most of the chip idle (the row counts of this step are <= 192).
template <typename T, typename OutT>
int ft_gemm(arp_ft* c, const char* site, const void* A, int lda, const void* W, int ldw, const float* bias, int act, const float* resid, OutT* out,
            int ldo, int M, int N, int K) {
    constexpr int EPB = 128 / (int)sizeof(T);
    const int nk = K / EPB;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    static const int wg_target = getenv("ARP_SPLITK_WGS") ? atoi(getenv("ARP_SPLITK_WGS")) : 512;  // one resident round (2 WG/CU x 256 CUs); measured best of 256..2048
    int S = std::max(1, std::min(nk, wg_target / std::max(tiles, 1)));
    const int per = (nk + S - 1) / S;
    S = (nk + per - 1) / per;
    ProfScope ps(c->prof, c->stream, site);
    GemmArgs g;
    g.A = A; g.W = W; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
    if (S == 1 && act == ACT_NONE && !resid && tiles >= 256) {  // big output, short contraction: the weight-gradient GEMMs
        g.bias = bias; g.resid = nullptr; g.out = out; g.ldr = ldo; g.ldo = ldo;
        if constexpr (sizeof(T) == 2 && sizeof(OutT) == 4) {
            const float* gb = c->grads.as<float>();
            const float* o = reinterpret_cast<const float*>(out);
            bool skipped = false;  // a weight that gets no gradient in this configuration (use_id = 0: the inverse model) must not be decayed:
            if (c->fuse_now && o >= gb) {  // ADVICE r3 -- the fused epilogue would apply p *= 1 - lr wd with g = 0 where AdamW leaves the tensor alone
                const FtSkip sk = ft_skip_ranges(c);
                const size_t lo = (size_t)(o - gb), hi = lo + (size_t)M * N;
                for (int r = 0; r < FT_SKIP_RANGES; ++r) skipped |= sk.lo[r] < hi && lo < sk.hi[r];
            }
            if (c->fuse_now && !skipped && !bias && ldo == N && (N & 7) == 0 && o >= gb && o + (size_t)M * N <= gb + c->P) {
                const size_t off = (size_t)(o - gb);
                g.adam_p = c->params.as<float>() + off; g.adam_m = c->mu.as<float>() + off; g.adam_v = c->nu.as<float>() + off;
                g.adam_mirror = c->mirror.p ? static_cast<void*>(c->mirror.as<T>() + off) : nullptr;
                g.adam_gscale = 1.0f / c->grad_scale(); g.adam_lr = c->fuse_lr; g.adam_wd = c->cfg.weight_decay; g.adam_b1 = c->cfg.b1; g.adam_b2 = c->cfg.b2;
                g.adam_eps = c->cfg.eps; g.adam_bc1 = c->fuse_bc1; g.adam_bc2 = c->fuse_bc2;
                g.adam_mask = c->cfg.mode == ARP_MODE_F16; g.adam_dropped = c->dropped.as<unsigned int>();
                c->fused.emplace_back(off, off + (size_t)M * N);
                return launch_gemm_nt<T, float, ACT_NONE, false, GEMM_SITE_ADAMW>(g, c->stream);
            }
        }
        return launch_gemm_auto<T, OutT, ACT_NONE, false, SITE_FT>(g, c->stream, 0);
    }
    ARP_TRY(c->part.ensure((size_t)S * M * N * 4));
    g.bias = nullptr; g.resid = nullptr; g.out = c->part.p; g.ldr = N; g.ldo = N;
    g.ksplit = S; g.slice_stride = (size_t)M * N;
    static const int mfast = getenv("ARP_FT_MFAST") ? atoi(getenv("ARP_FT_MFAST")) : 1;
    g.m_fast = (mfast && M <= 4 * GEMM_BM) ? 1 : 0;  // the image tower's 192 rows = two row tiles over the same weight stream
    ARP_TRY((launch_gemm_nt<T, float, ACT_NONE, false, SITE_FT + 1>(g, c->stream)));
    const size_t MN = (size_t)M * N;
    launch_splitk_reduce<OutT>(c->stream, c->part.as<float>(), S, MN, N, bias, act, out, resid, ldo == N ? 0 : ldo);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// dX[M, N] (ldo) = dY[M, K] . W[K, N] (+ resid) with W = the operand-type mirror of a weight [out = K, in = N] as it lies in memory: the NN kernel,
// split over K into f32 slabs (as ft_gemm splits its NT products), fixed-order reduction.
template <typename T, typename OutT>
int ft_gemm_nn(arp_ft* c, const char* site, const void* A, int lda, const void* W, int ldw, const float* resid, OutT* out, int ldo, int M, int N, int K) {
    static_assert(sizeof(T) == 2, "16-bit modes only");
    const int nk = K / 64;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    static const int wg_target = getenv("ARP_SPLITK_WGS") ? atoi(getenv("ARP_SPLITK_WGS")) : 512;
    int S = std::max(1, std::min(nk, wg_target / std::max(tiles, 1)));
    const int per = (nk + S - 1) / S;
    S = (nk + per - 1) / per;
    ProfScope ps(c->prof, c->stream, site);
    ARP_TRY(c->part.ensure((size_t)S * M * N * 4));
    GemmTnArgs g;
    g.A = A; g.B = W; g.out = c->part.as<float>(); g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldw; g.ldo = N; g.ksplit = S;
    g.slice_stride = (size_t)M * N; g.alpha = 1.f;
    ARP_TRY(launch_gemm_nn(__is_same(T, bf16_t) ? 1 : 2, g, c->stream));
    const size_t MN = (size_t)M * N;
    launch_splitk_reduce<OutT>(c->stream, c->part.as<float>(), S, MN, N, (const float*)nullptr, (int)ACT_NONE, out, resid, ldo == N ? 0 : ldo);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

int sgemm(arp_ft* c, const float* A, int ta, const float* Bm, int tb, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int act = ACT_NONE) {
    SmallGemm g{A, Bm, bias, nullptr, C, M, N, K, lda, ldb, N, ta, tb, act, 0};
    hipLaunchKernelGGL(small_gemm_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, c->stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

const char* TW[2] = {"image", "text"};

template <typename T> const T* fwd_w(arp_ft* c, const std::string& name) {
    if constexpr (sizeof(T) == 4) return reinterpret_cast<const T*>(c->p(name));
    return c->mirror.as<T>() + c->infos[c->index.at(name)].off;
}

// Forward-layout operands: the f32 parameters themselves (f32 mode) or their bf16 mirror, which AdamW keeps current; only a
// host-side parameter write forces a conversion pass.  Transposed operands ([in, out], for dX) are rebuilt every step
// from the operand-type copy.
template <typename T> int refresh_shadows(arp_ft* c) {
    const bool nn = sizeof(T) == 2 && c->nn_dx;
    if (!c->shadows_stale && (!c->transposed_stale || nn)) return 0;
    ProfScope ps(c->prof, c->stream, "ft.refresh_shadows");
    const int F = c->F(), Hd = c->Hd(), Hi = c->cfg.hidden;
    if (sizeof(T) == 2 && c->shadows_stale) {
        hipLaunchKernelGGL((convert_kernel<T>), dim3(cdiv(c->P, 1024)), dim3(256), 0, c->stream, c->params.as<float>(), c->mirror.as<T>(), c->P);
        ARP_HIP_OK(hipGetLastError());
    }
    if (nn) {  // the dX products read the mirror in place
        c->shadows_stale = false;
        return 0;
    }
    for (int w = 0; w < 2; ++w) {
        const std::string a = std::string(TW[w]) + "_adapter";
        ARP_TRY((ft_transpose<T, T, T>(c, fwd_w<T>(c, a + ".layers.0.weight"), F, nullptr, nullptr, 0, c->sW1t[w].as<T>(), Hd, Hd, F)));
        ARP_TRY((ft_transpose<T, T, T>(c, fwd_w<T>(c, a + ".layers.3.weight"), Hd, nullptr, nullptr, 0, c->sW2t[w].as<T>(), F, F, Hd)));
    }
    ARP_TRY((ft_transpose<T, T, T>(c, fwd_w<T>(c, "inverse_layer.layers.0.weight"), 4 * F, nullptr, nullptr, 0, c->sV1t.as<T>(), Hi, Hi, 4 * F)));
    c->shadows_stale = false;
    c->transposed_stale = false;
    return 0;
}

int ensure_buffers(arp_ft* c, int B) {
    if (B == c->B) return 0;
    const size_t e = c->esz();
    const int F = c->F(), Hd = c->Hd(), Hi = c->cfg.hidden, NA = c->cfg.n_actions, E = c->cfg.embed;
    auto f32 = [&](DevBuf& b, size_t n) { return b.ensure(std::max<size_t>(n, 4) * 4); };
    auto typ = [&](DevBuf& b, size_t n) { return b.ensure(std::max<size_t>(n, 8) * e); };
    auto typz = [&](DevBuf& b, size_t n) {  // transposed operands: the K padding (rows .. 64-multiple) must read as zeros
        ARP_TRY(b.ensure(std::max<size_t>(n, 8) * e));
        ARP_HIP_OK(hipMemsetAsync(b.p, 0, std::max<size_t>(n, 8) * e, c->stream));
        return 0;
    };
    for (int w = 0; w < 2; ++w) {
        const size_t M = w == 0 ? c->groups() * (size_t)B : (size_t)B, Mp = (M + 63) / 64 * 64;
        const size_t Din = w == 0 ? c->Dv() : c->Dt();
        ARP_TRY(f32(c->x_in[w], M * Din)); ARP_TRY(f32(c->x_fin[w], M * E));
        ARP_TRY(typ(c->X[w], M * Din)); ARP_TRY(typz(c->XT[w], Din * Mp));
        ARP_TRY(f32(c->f[w], M * F)); ARP_TRY(typ(c->fT_[w], M * F)); ARP_TRY(typz(c->fTt[w], F * Mp));
        ARP_TRY(typ(c->H[w], M * Hd)); ARP_TRY(typz(c->HT[w], Hd * Mp));
        ARP_TRY(f32(c->A[w], M * F)); ARP_TRY(f32(c->a[w], M * F)); ARP_TRY(f32(c->nrm[w], M)); ARP_TRY(f32(c->da[w], M * F));
        ARP_TRY(f32(c->dA[w], M * F)); ARP_TRY(typ(c->dAT_[w], M * F)); ARP_TRY(typz(c->dAt[w], F * Mp)); ARP_TRY(f32(c->dfd[w], M * F));
        ARP_TRY(typ(c->dH[w], M * Hd)); ARP_TRY(typ(c->dHp[w], M * Hd)); ARP_TRY(typz(c->dHpt[w], Hd * Mp));
        ARP_TRY(f32(c->df[w], M * F)); ARP_TRY(typz(c->dUt[w], (size_t)c->Dt() * Mp)); ARP_TRY(f32(c->dres_part[w], M));
    }
    const size_t Bp = ((size_t)B + 63) / 64 * 64;
    ARP_TRY(f32(c->r, B)); ARP_TRY(f32(c->action, B)); ARP_TRY(f32(c->scores, 3 * (size_t)B)); ARP_TRY(f32(c->ds, 3 * (size_t)B)); ARP_TRY(f32(c->dist, 3 * (size_t)B));
    ARP_TRY(f32(c->C, (size_t)B * 4 * F)); ARP_TRY(typ(c->CT_, (size_t)B * 4 * F)); ARP_TRY(typz(c->Ct, (size_t)4 * F * Bp));
    ARP_TRY(f32(c->Hinv, (size_t)B * Hi)); ARP_TRY(f32(c->logits, (size_t)B * NA)); ARP_TRY(f32(c->dlogits, (size_t)B * NA));
    ARP_TRY(f32(c->dHinv, (size_t)B * Hi)); ARP_TRY(typ(c->dHinvT_, (size_t)B * Hi)); ARP_TRY(typz(c->dHinvt, (size_t)Hi * Bp));
    ARP_TRY(f32(c->dC, (size_t)B * 4 * F)); ARP_TRY(f32(c->metrics, 8)); ARP_TRY(f32(c->scal, 64));
    c->B = B;
    return 0;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// One tower's head: tower features (c->x_in[w], c->x_fin[w], M rows) -> adapted, normalised features c->a[w]
// (clip_multiscale_adapter.py:134-149 / :151-175), leaving every activation the backward needs.
template <typename T> int encode_tower(arp_ft* c, int w, int M) {
    const arp_ft_cfg& k = c->cfg;
    const int F = c->F(), Hd = c->Hd(), E = k.embed, Dt = c->Dt();
    const int Mp = (M + 63) / 64 * 64, Din = w == 0 ? c->Dv() : Dt;
    const std::string a = std::string(TW[w]) + "_adapter", pre = std::string("ft.") + TW[w];
    // frozen-tower features -> operand type, both layouts (the transposed one feeds dWint)
    ARP_TRY((ft_transpose<float, float, T>(c, c->x_in[w].as<float>(), Din, nullptr, c->X[w].as<T>(), Din, c->XT[w].as<T>(), Mp, M, Din)));
    // f = [X Wint^T | final]   (:141-143 / :164-166)
    ARP_TRY((ft_gemm<T, float>(c, (pre + "_inter").c_str(), c->X[w].p, Din, fwd_w<T>(c, std::string(TW[w]) + "_intermediate_linear.weight"), Din,
                               nullptr, ACT_NONE, nullptr, c->f[w].as<float>(), F, M, Dt, Din)));
    hipLaunchKernelGGL(ft_copy_cols_kernel, dim3(cdiv((size_t)M * E, 256)), dim3(256), 0, c->stream, c->x_fin[w].as<float>(), E, c->f[w].as<float>(), F,
                       Dt, M);
    ARP_TRY((ft_transpose<float, float, T>(c, c->f[w].as<float>(), F, nullptr, c->fT_[w].as<T>(), F, c->fTt[w].as<T>(), Mp, M, F)));
    // AdapterMLP (layers.py:43-60 with num_layers = 2): Linear -> ReLU -> Linear
    ARP_TRY((ft_gemm<T, T>(c, (pre + "_fc1").c_str(), c->fT_[w].p, F, fwd_w<T>(c, a + ".layers.0.weight"), F, c->p(a + ".layers.0.bias"), ACT_RELU,
                           nullptr, c->H[w].as<T>(), Hd, M, Hd, F)));
    ARP_TRY((ft_gemm<T, float>(c, (pre + "_fc2").c_str(), c->H[w].p, Hd, fwd_w<T>(c, a + ".layers.3.weight"), Hd, c->p(a + ".layers.3.bias"),
                               ACT_NONE, nullptr, c->A[w].as<float>(), F, M, F, Hd)));
    ProfScope ps(c->prof, c->stream, "ft.rowops");
    hipLaunchKernelGGL(ft_mix_norm_fwd_kernel, dim3(M), dim3(256), 0, c->stream, c->f[w].as<float>(), c->A[w].as<float>(),
                       c->p(std::string(TW[w]) + "_residual_weight"), c->a[w].as<float>(), c->nrm[w].as<float>(), F);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename T> int forward(arp_ft* c) {
    const arp_ft_cfg& k = c->cfg;
    const int B = c->B, F = c->F(), Hi = k.hidden, NA = k.n_actions;
    ARP_TRY(refresh_shadows<T>(c));
    ARP_TRY(encode_tower<T>(c, 0, c->groups() * B));
    if (!k.goal_conditioned) ARP_TRY(encode_tower<T>(c, 1, B));
    {
        ProfScope ps(c->prof, c->stream, "ft.rowops");
        // the "prompt side" t of the scores and of the inverse-model input: the text head's output, or (goal_conditioned) image3's
        const float* t = k.goal_conditioned ? c->a[0].as<float>() + (size_t)3 * B * F : c->a[1].as<float>();
        if (k.goal_conditioned)
            hipLaunchKernelGGL(ft_goal_scores_kernel, dim3(3 * B), dim3(256), 0, c->stream, c->a[0].as<float>(), c->scores.as<float>(), c->dist.as<float>(), B, F);
        else
            hipLaunchKernelGGL(ft_scores_kernel, dim3(3 * B), dim3(256), 0, c->stream, c->a[0].as<float>(), t, expf(k.logit_scale), c->scores.as<float>(), B, F);
        hipLaunchKernelGGL(ft_build_c_kernel, dim3(cdiv((size_t)B * 4 * F, 256)), dim3(256), 0, c->stream, c->a[0].as<float>(), t, c->C.as<float>(), B, F);
        ARP_HIP_OK(hipGetLastError());
    }
    const int Bp = (B + 63) / 64 * 64;
    ARP_TRY((ft_transpose<float, float, T>(c, c->C.as<float>(), 4 * F, nullptr, c->CT_.as<T>(), 4 * F, c->Ct.as<T>(), Bp, B, 4 * F)));
    ARP_TRY((ft_gemm<T, float>(c, "ft.inverse_fc1", c->CT_.p, 4 * F, fwd_w<T>(c, "inverse_layer.layers.0.weight"), 4 * F,
                               c->p("inverse_layer.layers.0.bias"), ACT_RELU, nullptr, c->Hinv.as<float>(), Hi, B, Hi, 4 * F)));
    ProfScope ps(c->prof, c->stream, "ft.loss");
    if ((Hi & 3) == 0) {  // logits [B, 15] over K = hidden: one wave per logit
        hipLaunchKernelGGL(ft_rowdot_kernel, dim3(cdiv((size_t)B * NA, 4)), dim3(256), 0, c->stream, c->Hinv.as<float>(), c->p("inverse_layer.layers.3.weight"),
                           c->p("inverse_layer.layers.3.bias"), c->logits.as<float>(), B, NA, Hi, Hi, Hi);
        ARP_HIP_OK(hipGetLastError());
    } else
    ARP_TRY(sgemm(c, c->Hinv.as<float>(), 0, c->p("inverse_layer.layers.3.weight"), 1, c->p("inverse_layer.layers.3.bias"), c->logits.as<float>(), B, NA, Hi,
                  Hi, Hi));
    hipLaunchKernelGGL(ft_loss_kernel, dim3(1), dim3(256), 0, c->stream, c->scores.as<float>(), c->r.as<float>(), c->logits.as<float>(), c->action.as<int>(),
                       B, NA, k.gamma, c->p("lambda_id"), k.use_vip, k.use_id, c->metrics.as<float>(), c->ds.as<float>(), c->dlogits.as<float>(),
                       c->g("lambda_id"), c->grad_scale());
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// ---- backward: every entry of c->grads written exactly once ----------------------------------------------------------
// Flat-gradient ranges of the all-reduce buckets, in production order.  Bucket b = ranges [2b, 2b + 1] (the second one empty except
// for the last bucket, which also takes the three scalar parameters at the end of the flat vector).  They tile [0, P) exactly once.
constexpr int FT_BUCKETS = 7;
struct FtBuckets { size_t lo[2 * FT_BUCKETS], hi[2 * FT_BUCKETS]; };
FtBuckets ft_buckets(const arp_ft* c) {
    FtBuckets b;
    for (int i = 0; i < 2 * FT_BUCKETS; ++i) b.lo[i] = b.hi[i] = 0;
    auto off = [&](const char* n) { return c->infos[c->index.at(n)].off; };
    auto span = [&](int i, const char* first, const char* next) { b.lo[i] = off(first); b.hi[i] = next ? off(next) : c->P; };
    span(0, "inverse_layer.layers.0.weight", "image_residual_weight");                        // bucket 0: the inverse model
    span(2, "image_adapter.layers.3.weight", "text_adapter.layers.0.weight");                 // 1: image adapter, second layer
    span(4, "image_adapter.layers.0.weight", "image_adapter.layers.3.weight");                // 2: image adapter, first layer
    span(6, "image_intermediate_linear.weight", "text_intermediate_linear.weight");           // 3
    span(8, "text_adapter.layers.3.weight", "inverse_layer.layers.0.weight");                 // 4: text adapter, second layer
    span(10, "text_adapter.layers.0.weight", "text_adapter.layers.3.weight");                 // 5: text adapter, first layer
    span(12, "text_intermediate_linear.weight", "image_adapter.layers.0.weight");             // 6: ... and the scalars
    span(13, "image_residual_weight", nullptr);
    return b;
}
// bucket `b` is complete on the compute stream: all-reduce it on the communication stream (no-op outside a data-parallel step)
int ft_bucket_ready(arp_ft* c, int b) {
    if (!c->comm_live) return 0;
    const FtBuckets plan = ft_buckets(c);
    ARP_HIP_OK(hipEventRecord(c->ev_bucket[b], c->stream));
    ARP_HIP_OK(hipStreamWaitEvent(c->comm_stream, c->ev_bucket[b], 0));
    static const char* names[FT_BUCKETS] = {"ft.allreduce_b0", "ft.allreduce_b1", "ft.allreduce_b2", "ft.allreduce_b3", "ft.allreduce_b4", "ft.allreduce_b5", "ft.allreduce_b6"};
    ProfScope ps(c->prof, c->comm_stream, names[b]);
    RcclApi* r = rccl_api();
    const bool grp = r->GroupStart && r->GroupEnd;
    if (grp) r->GroupStart();
    int rc = 0;
    for (int i = 2 * b; i < 2 * b + 2; ++i)
        if (plan.hi[i] > plan.lo[i] && !(c->cfg.goal_conditioned && i >= 8 && i <= 12)) {  // goal_conditioned: the text head has no gradient
            float* p = c->grads.as<float>() + plan.lo[i];
            if (r->AllReduce(p, p, plan.hi[i] - plan.lo[i], ncclFloat, ncclSum, c->comm, c->comm_stream) != ncclSuccess) rc = -1;
        }
    if (b == FT_BUCKETS - 1 && r->AllReduce(c->metrics.p, c->metrics.p, 4, ncclFloat, ncclSum, c->comm, c->comm_stream) != ncclSuccess) rc = -1;
    if (grp) r->GroupEnd();
    return rc ? fail("ncclAllReduce(gradient bucket) failed") : 0;
}

template <typename T> int backward(arp_ft* c) {
    const arp_ft_cfg& k = c->cfg;
    c->grads_summed = false;  // this rank's own gradient from here on
    const int B = c->B, F = c->F(), Hd = c->Hd(), Hi = k.hidden, NA = k.n_actions, Dt = c->Dt();
    const int Bp = (B + 63) / 64 * 64;
    {
        // inverse model (:232-237): logits = relu(C V1^T + c1) V2^T + c2
        ProfScope ps(c->prof, c->stream, "ft.inverse_bwd_small");
        ARP_TRY(sgemm(c, c->dlogits.as<float>(), 1, c->Hinv.as<float>(), 0, nullptr, c->g("inverse_layer.layers.3.weight"), NA, Hi, B, NA, Hi));
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(NA, 64)), dim3(256), 0, c->stream, c->dlogits.as<float>(), B, NA, c->g("inverse_layer.layers.3.bias"));
        ARP_TRY(sgemm(c, c->dlogits.as<float>(), 0, c->p("inverse_layer.layers.3.weight"), 0, nullptr, c->dHinv.as<float>(), B, Hi, NA, NA, Hi));
        // relu mask (Hinv is the post-activation), both layouts
        ARP_TRY((ft_transpose<float, float, T>(c, c->dHinv.as<float>(), Hi, c->Hinv.as<float>(), c->dHinvT_.as<T>(), Hi, c->dHinvt.as<T>(), Bp, B, Hi)));
        hipLaunchKernelGGL((rowsum_kernel<T>), dim3(Hi), dim3(256), 0, c->stream, c->dHinvt.as<T>(), Bp, B, c->g("inverse_layer.layers.0.bias"), Hi);
        ARP_HIP_OK(hipGetLastError());
    }
    const bool nn = sizeof(T) == 2 && c->nn_dx;  // (nn_dx already says the geometry fits the kernel: arp_ft_create)
    // NN path: every dX runs BEFORE its layer's weight-gradient GEMM -- that GEMM's fused AdamW epilogue moves the weight (and its mirror) in place
    if constexpr (sizeof(T) == 2) {
        if (nn) ARP_TRY((ft_gemm_nn<T, float>(c, "ft.inverse_fc1_dX", c->dHinvT_.p, Hi, fwd_w<T>(c, "inverse_layer.layers.0.weight"), 4 * F, nullptr, c->dC.as<float>(),
                                              4 * F, B, 4 * F, Hi)));
    }
    ARP_TRY((ft_gemm<T, float>(c, "ft.inverse_fc1_dW", c->dHinvt.p, Bp, c->Ct.p, Bp, nullptr, ACT_NONE, nullptr, c->g("inverse_layer.layers.0.weight"), 4 * F, Hi,
                               4 * F, Bp)));
    ARP_TRY(ft_bucket_ready(c, 0));
    if (!nn) ARP_TRY((ft_gemm<T, float>(c, "ft.inverse_fc1_dX", c->dHinvT_.p, Hi, c->sV1t.p, Hi, nullptr, ACT_NONE, nullptr, c->dC.as<float>(), 4 * F, B, 4 * F, Hi)));
    {
        ProfScope ps(c->prof, c->stream, "ft.rowops");
        if (k.goal_conditioned)
            hipLaunchKernelGGL(ft_goal_feat_grad_kernel, dim3(cdiv((size_t)B * F, 256)), dim3(256), 0, c->stream, c->ds.as<float>(), c->a[0].as<float>(),
                               c->dist.as<float>(), c->dC.as<float>(), c->da[0].as<float>(), B, F);
        else
            hipLaunchKernelGGL(ft_feat_grad_kernel, dim3(cdiv((size_t)B * F, 256)), dim3(256), 0, c->stream, c->ds.as<float>(), c->a[0].as<float>(),
                               c->a[1].as<float>(), c->dC.as<float>(), expf(k.logit_scale), c->da[0].as<float>(), c->da[1].as<float>(), B, F);
        ARP_HIP_OK(hipGetLastError());
    }
    for (int w = 0; w < (k.goal_conditioned ? 1 : 2); ++w) {
        const int M = w == 0 ? c->groups() * B : B, Mp = (M + 63) / 64 * 64, Din = w == 0 ? c->Dv() : Dt;
        const std::string a = std::string(TW[w]) + "_adapter", pre = std::string("ft.") + TW[w], rwn = std::string(TW[w]) + "_residual_weight";
        {
            ProfScope ps(c->prof, c->stream, "ft.rowops");
            hipLaunchKernelGGL(ft_mix_norm_bwd_kernel, dim3(M), dim3(256), 0, c->stream, c->da[w].as<float>(), c->a[w].as<float>(), c->nrm[w].as<float>(),
                               c->f[w].as<float>(), c->A[w].as<float>(), c->p(rwn), c->dA[w].as<float>(), c->dfd[w].as<float>(),
                               c->dres_part[w].as<float>(), F);
            hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->dres_part[w].as<float>(), M, 1.0f, c->scal.as<float>() + 8 + w, 0);
            hipLaunchKernelGGL(dres_to_drw_kernel, dim3(1), dim3(1), 0, c->stream, c->scal.as<float>() + 8 + w, c->p(rwn), c->g(rwn));
            hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(F, 64)), dim3(256), 0, c->stream, c->dA[w].as<float>(), M, F, c->g(a + ".layers.3.bias"));
            ARP_HIP_OK(hipGetLastError());
        }
        ARP_TRY((ft_transpose<float, float, T>(c, c->dA[w].as<float>(), F, nullptr, c->dAT_[w].as<T>(), F, c->dAt[w].as<T>(), Mp, M, F)));
        ARP_TRY((ft_transpose<T, T, T>(c, c->H[w].as<T>(), Hd, nullptr, nullptr, 0, c->HT[w].as<T>(), Mp, M, Hd)));
        // A = H W2^T + b2
        if constexpr (sizeof(T) == 2) {
            if (nn) ARP_TRY((ft_gemm_nn<T, T>(c, (pre + "_fc2_dX").c_str(), c->dAT_[w].p, F, fwd_w<T>(c, a + ".layers.3.weight"), Hd, nullptr, c->dH[w].as<T>(), Hd, M, Hd, F)));
        }
        ARP_TRY((ft_gemm<T, float>(c, (pre + "_fc2_dW").c_str(), c->dAt[w].p, Mp, c->HT[w].p, Mp, nullptr, ACT_NONE, nullptr, c->g(a + ".layers.3.weight"), Hd, F,
                                   Hd, Mp)));
        ARP_TRY(ft_bucket_ready(c, 1 + 3 * w));
        if (!nn) ARP_TRY((ft_gemm<T, T>(c, (pre + "_fc2_dX").c_str(), c->dAT_[w].p, F, c->sW2t[w].p, F, nullptr, ACT_NONE, nullptr, c->dH[w].as<T>(), Hd, M, Hd, F)));
        // relu mask (H is the post-activation), both layouts; H = relu(f W1^T + b1)
        ARP_TRY((ft_transpose<T, T, T>(c, c->dH[w].as<T>(), Hd, c->H[w].as<T>(), c->dHp[w].as<T>(), Hd, c->dHpt[w].as<T>(), Mp, M, Hd)));
        hipLaunchKernelGGL((rowsum_kernel<T>), dim3(Hd), dim3(256), 0, c->stream, c->dHpt[w].as<T>(), Mp, M, c->g(a + ".layers.0.bias"), Hd);
        ARP_HIP_OK(hipGetLastError());
        // df = res*dy (direct path) + dHpre W1
        if constexpr (sizeof(T) == 2) {
            if (nn) ARP_TRY((ft_gemm_nn<T, float>(c, (pre + "_fc1_dX").c_str(), c->dHp[w].p, Hd, fwd_w<T>(c, a + ".layers.0.weight"), F, c->dfd[w].as<float>(),
                                                  c->df[w].as<float>(), F, M, F, Hd)));
        }
        ARP_TRY((ft_gemm<T, float>(c, (pre + "_fc1_dW").c_str(), c->dHpt[w].p, Mp, c->fTt[w].p, Mp, nullptr, ACT_NONE, nullptr, c->g(a + ".layers.0.weight"), F, Hd,
                                   F, Mp)));
        ARP_TRY(ft_bucket_ready(c, 2 + 3 * w));
        if (!nn) ARP_TRY((ft_gemm<T, float>(c, (pre + "_fc1_dX").c_str(), c->dHp[w].p, Hd, c->sW1t[w].p, Hd, nullptr, ACT_NONE, c->dfd[w].as<float>(),
                                            c->df[w].as<float>(), F, M, F, Hd)));
        // U = X Wint^T occupies the first Dt columns of f
        ARP_TRY((ft_transpose<float, float, T>(c, c->df[w].as<float>(), F, nullptr, nullptr, 0, c->dUt[w].as<T>(), Mp, M, Dt)));
        ARP_TRY((ft_gemm<T, float>(c, (pre + "_inter_dW").c_str(), c->dUt[w].p, Mp, c->XT[w].p, Mp, nullptr, ACT_NONE, nullptr,
                                   c->g(std::string(TW[w]) + "_intermediate_linear.weight"), Din, Dt, Din, Mp)));
        ARP_TRY(ft_bucket_ready(c, 3 + 3 * w));  // w = 1: the last bucket (+ the scalar parameters and the loss scalars)
    }
    if (k.goal_conditioned) ARP_TRY(ft_bucket_ready(c, FT_BUCKETS - 1));  // no text head ran: only the scalars and the loss scalars are left
    return 0;
}

int apply_update(arp_ft* c, float lr) {
    ProfScope ps(c->prof, c->stream, "ft.adamw");
    const double t = (double)(c->step + 1);
    const float bc1 = (float)(1.0 - std::pow((double)c->cfg.b1, t)), bc2 = (float)(1.0 - std::pow((double)c->cfg.b2, t));
    // parameters that received no gradient this step are left alone, as torch does for .grad is None
    const FtSkip skip = ft_skip_ranges(c);
    const float gscale = 1.0f / ((float)std::max(c->world, 1) * c->grad_scale());
    // the segments of the flat parameter vector that the weight-gradient GEMMs have NOT already updated (all of it when nothing was fused)
    std::sort(c->fused.begin(), c->fused.end());
    std::vector<std::pair<size_t, size_t>> todo;
    size_t at = 0;
    for (const auto& f : c->fused) {
        if (f.first > at) todo.emplace_back(at, f.first);
        at = std::max(at, f.second);
    }
    if (at < c->P) todo.emplace_back(at, c->P);
    c->grads_gone = c->fused;
    c->fused.clear();
    for (const auto& seg : todo) {
        const size_t lo = seg.first, n = seg.second - seg.first;  // lo and n are multiples of 4 (tensor offsets and padded sizes are)
        FtSkip sk = skip;  // the kernel indexes from its own base: shift the skip ranges
        for (int r = 0; r < FT_SKIP_RANGES; ++r) {
            sk.lo[r] = skip.lo[r] > lo ? skip.lo[r] - lo : 0;
            sk.hi[r] = skip.hi[r] > lo ? skip.hi[r] - lo : 0;
        }
#define ARP_FT_ADAMW(TM)                                                                                                                          \
    hipLaunchKernelGGL((ft_adamw_kernel<TM>), dim3(cdiv(n, 1024)), dim3(256), 0, c->stream, c->params.as<float>() + lo, c->grads.as<float>() + lo, \
                       c->mu.as<float>() + lo, c->nu.as<float>() + lo, gscale, lr, c->cfg.weight_decay, c->cfg.b1, c->cfg.b2, c->cfg.eps, bc1, bc2, n, \
                       c->mirror.p ? c->mirror.as<TM>() + lo : nullptr, sk, c->cfg.mode == ARP_MODE_F16 ? 1 : 0, c->dropped.as<unsigned int>())
        if (c->cfg.mode == ARP_MODE_BF16) ARP_FT_ADAMW(bf16_t);
        else if (c->cfg.mode == ARP_MODE_F16) ARP_FT_ADAMW(f16_t);
        else ARP_FT_ADAMW(float);
#undef ARP_FT_ADAMW
    }
    ARP_HIP_OK(hipGetLastError());
    c->step += 1;
    c->transposed_stale = true;
    return 0;
}

template <typename T> int step_impl(arp_ft* c, float lr, float* aux) {
    ARP_TRY(forward<T>(c));
    const bool comm = c->has_comm && (c->world > 1 || c->force_comm);
    if (comm && c->overlap_comm) {
        // data parallel, overlapped: the buckets leave from inside backward() (ft_bucket_ready); AdamW waits for the last one
        c->comm_live = true;
        const int rc = backward<T>(c);
        c->comm_live = false;
        ARP_TRY(rc);
        ARP_HIP_OK(hipEventRecord(c->ev_comm, c->comm_stream));
        ARP_HIP_OK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        c->grads_summed = true;
    } else {
        // single process: the big weight-gradient GEMMs apply AdamW themselves (the step's lr and bias corrections are known here)
        c->fused.clear();
        if (c->fuse_adam && !comm && sizeof(T) == 2) {
            const double t = (double)(c->step + 1);
            c->fuse_lr = lr;
            c->fuse_bc1 = (float)(1.0 - std::pow((double)c->cfg.b1, t));
            c->fuse_bc2 = (float)(1.0 - std::pow((double)c->cfg.b2, t));
            c->fuse_now = true;
        }
        const int rc = backward<T>(c);
        c->fuse_now = false;
        if (rc != 0) {
            c->fused.clear();
            return rc;
        }
    }
    if (comm && !c->overlap_comm) {
        // the serial form: every rank ran its shard; ONE all-reduce(sum) of the flat f32 gradient (1.9 GB at full size) plus one of
        // the 4 loss scalars -- the mean over ranks is taken by the 1/world factor inside the AdamW kernel
        ProfScope ps(c->prof, c->stream, "ft.allreduce");
        if (rccl_api()->AllReduce(c->grads.p, c->grads.p, c->P, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return fail("ncclAllReduce(grads) failed");
        if (rccl_api()->AllReduce(c->metrics.p, c->metrics.p, 4, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return fail("ncclAllReduce(metrics) failed");
        c->grads_summed = true;
    }
    ARP_TRY(apply_update(c, lr));
    if (aux) {
        ARP_HIP_OK(hipMemcpyAsync(aux, c->metrics.p, 16, hipMemcpyDeviceToHost, c->stream));
        ARP_HIP_OK(hipStreamSynchronize(c->stream));
        const float inv = 1.0f / (float)std::max(c->world, 1);
        for (int i = 0; i < 3; ++i) aux[i] *= inv;  // loss, vip_loss, id_loss: rank means; lambda_id is the same everywhere
        if (c->world > 1) aux[3] *= inv;
    }
    return 0;
}

}  // namespace

// =================================== C ABI ===============================================================
extern "C" {

int arp_ft_create(const arp_ft_cfg* cfg, arp_ft** out) {
    if (!cfg || !out) return fail("null argument");
    const arp_ft_cfg& k = *cfg;
    if (k.mode != ARP_MODE_F32 && k.mode != ARP_MODE_BF16 && k.mode != ARP_MODE_F16) return fail("bad mode");
    if (k.layers <= 0 || k.width_v <= 0 || k.width_t <= 0 || k.embed <= 0 || k.hidden <= 0 || k.n_actions <= 0) return fail("bad geometry");
    const int kq = k.mode == ARP_MODE_F32 ? 32 : 64;
    const int F = k.layers * k.width_t + k.embed;
    // every contraction length of an MFMA GEMM must be a whole number of K-tiles
    for (int d : {k.layers * k.width_v, k.layers * k.width_t, F, k.hidden * (k.layers + 1), k.hidden})
        if (d % kq) return fail("feature widths must be multiples of " + std::to_string(kq));
    int ndev = 0;
    ARP_HIP_OK(hipGetDeviceCount(&ndev));
    if (k.device < 0 || k.device >= ndev) return fail("no such HIP device: " + std::to_string(k.device));
    ARP_HIP_OK(hipSetDevice(k.device));
    ARP_TRY(prime_runtime(k.device));  // (runtime.h: one null-stream copy before the process's first stream exists)
    arp_ft* c = new arp_ft();
    c->cfg = k;
    build_layout(c);
    auto body = [&]() -> int {
        ARP_HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        // (the communication stream is created with the communicator: a process's streams share GPU_MAX_HW_QUEUES hardware queues, arp_dt.hip)
        for (auto& e : c->ev_bucket) ARP_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ARP_HIP_OK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
        if (const char* e = getenv("ARP_FT_OVERLAP")) c->overlap_comm = atoi(e) != 0;
        if (const char* e = getenv("ARP_FT_FORCE_COMM")) c->force_comm = atoi(e) != 0;
        if (const char* e = getenv("ARP_FT_FUSE_ADAM")) c->fuse_adam = atoi(e) != 0;
        if (const char* e = getenv("ARP_FT_NN")) c->nn_dx = atoi(e) != 0;
        {   // the NN dX kernel wants output widths in multiples of 128 and contraction lengths in multiples of 64 (16-bit modes only)
            const int Fq = c->F(), Hq = c->Hd();
            c->nn_dx = c->nn_dx && k.mode != ARP_MODE_F32 && Fq % 128 == 0 && Hq % 128 == 0 && k.hidden % 64 == 0;
        }
        DevBuf* fb[] = {&c->params, &c->grads, &c->mu, &c->nu};
        for (auto* b : fb) {
            ARP_TRY(b->ensure(c->P * 4));
            ARP_HIP_OK(hipMemset(b->p, 0, c->P * 4));
        }
        const size_t e = c->esz(), Fd = c->F(), Hd = c->Hd(), Hi = k.hidden;
        ARP_TRY(c->dropped.ensure(16));
        ARP_HIP_OK(hipMemset(c->dropped.p, 0, 16));
        if (k.mode != ARP_MODE_F32) ARP_TRY(c->mirror.ensure(c->P * e));
        if (!c->nn_dx) {  // the NT dX path's transposed weight shadows (817 MB at the real geometry in a 16-bit mode)
            for (int w = 0; w < 2; ++w) { ARP_TRY(c->sW1t[w].ensure(Hd * Fd * e)); ARP_TRY(c->sW2t[w].ensure(Hd * Fd * e)); }
            ARP_TRY(c->sV1t.ensure(Hi * 4 * Fd * e));
        }
        return 0;
    };
    if (body() != 0) { arp_ft_destroy(c); return -1; }
    *out = c;
    return 0;
}

int arp_ft_destroy(arp_ft* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->has_comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
    c->prof.destroy();
    DevBuf* all[] = {&c->params, &c->grads, &c->mu, &c->nu, &c->mirror, &c->sV1t, &c->r, &c->action, &c->scores, &c->ds, &c->dist, &c->C, &c->CT_, &c->Ct, &c->Hinv, &c->logits,
                     &c->dlogits, &c->dHinv, &c->dHinvT_, &c->dHinvt, &c->dist, &c->dC, &c->metrics, &c->scal, &c->part};
    for (auto* b : all) b->release();
    for (int w = 0; w < 2; ++w) {
        DevBuf* tw[] = {&c->sW1t[w], &c->sW2t[w], &c->x_in[w], &c->x_fin[w], &c->X[w], &c->XT[w], &c->f[w], &c->fT_[w],
                        &c->fTt[w], &c->H[w], &c->HT[w], &c->A[w], &c->a[w], &c->nrm[w], &c->da[w], &c->dA[w], &c->dAT_[w], &c->dAt[w], &c->dfd[w], &c->dH[w],
                        &c->dHp[w], &c->dHpt[w], &c->df[w], &c->dUt[w], &c->dres_part[w]};
        for (auto* b : tw) b->release();
    }
    if (c->comm_stream) {
        (void)hipStreamSynchronize(c->comm_stream);
        (void)hipStreamDestroy(c->comm_stream);
    }
    for (auto e : c->ev_bucket)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int arp_ft_num_params(arp_ft* c, int64_t* total, int32_t* n_tensors) {
    if (!c) return fail("null handle");
    size_t n = 0;
    for (auto& pi : c->infos) n += pi.size;
    if (total) *total = (int64_t)n;
    if (n_tensors) *n_tensors = (int32_t)c->infos.size();
    return 0;
}

int arp_ft_param_info(arp_ft* c, int i, char* name_buf, int name_len, int64_t* shape4, int32_t* ndim) {
    if (!c || i < 0 || i >= (int)c->infos.size() || !name_buf || !shape4 || !ndim) return fail("bad argument");
    const FtParam& pi = c->infos[i];
    if ((int)pi.name.size() + 1 > name_len) return fail("name buffer too small");
    memcpy(name_buf, pi.name.c_str(), pi.name.size() + 1);
    *ndim = (int32_t)pi.shape.size();
    for (size_t d = 0; d < pi.shape.size() && d < 4; ++d) shape4[d] = pi.shape[d];
    return 0;
}

// which: 0 = params, 1 = grads, 2 = adam exp_avg, 3 = adam exp_avg_sq.  torch layout ([out, in] weights) on both sides.
static int ft_tensor_io(arp_ft* c, const char* name, int which, float* host, int write) {
    if (!c || !name || !host) return fail("null argument");
    auto it = c->index.find(name);
    if (it == c->index.end()) return fail(std::string("unknown parameter: ") + name);
    const FtParam& pi = c->infos[it->second];
    DevBuf* bufs[] = {&c->params, &c->grads, &c->mu, &c->nu};
    if (which < 0 || which > 3) return fail("bad tensor selector");
    float* dev = bufs[which]->as<float>() + pi.off;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    if (write) {
        ARP_HIP_OK(hipMemcpy(dev, host, pi.size * 4, hipMemcpyHostToDevice));
        if (which == 0) c->shadows_stale = true;
    } else {
        if (which == 1)
            for (const auto& r : c->grads_gone)
                if (pi.off >= r.first && pi.off < r.second)
                    return fail(std::string(name) + ": the last step applied this weight's gradient inside its GEMM and never stored it "
                                "(arp_ft_backward materialises every gradient; ARP_FT_FUSE_ADAM=0 makes arp_ft_train_step do so too)");
        ARP_HIP_OK(hipMemcpy(host, dev, pi.size * 4, hipMemcpyDeviceToHost));
        // gradients carry the f16 mode's seed scale on the device and, after a data-parallel step, the SUM over ranks (the mean is
        // folded into AdamW): the getter returns what the optimizer consumed -- the un-scaled rank MEAN
        const float inv = 1.0f / (c->grad_scale() * (c->grads_summed ? (float)std::max(c->world, 1) : 1.f));
        if (which == 1 && inv != 1.f)
            for (size_t i = 0; i < pi.size; ++i) host[i] *= inv;
    }
    return 0;
}
int arp_ft_set_tensor(arp_ft* c, const char* name, int which, const float* data) { return ft_tensor_io(c, name, which, const_cast<float*>(data), 1); }
int arp_ft_get_tensor(arp_ft* c, const char* name, int which, float* out) { return ft_tensor_io(c, name, which, out, 0); }
int arp_ft_set_step(arp_ft* c, int64_t step) {
    if (!c || step < 0) return fail("bad argument");
    c->step = step;
    return 0;
}
int arp_ft_dropped_gradients(arp_ft* c, uint64_t* count) {
    if (!c || !count) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    unsigned int v = 0;
    ARP_HIP_OK(hipMemcpy(&v, c->dropped.p, 4, hipMemcpyDeviceToHost));
    if (v) ARP_HIP_OK(hipMemset(c->dropped.p, 0, 4));
    c->dropped_total += v;
    *count = c->dropped_total;
    return 0;
}
int arp_ft_get_step(arp_ft* c, int64_t* step) {
    if (!c || !step) return fail("bad argument");
    *step = c->step;
    return 0;
}

int arp_ft_set_batch(arp_ft* c, const float* img_inter, const float* img_final, const float* txt_inter, const float* txt_final, const float* r,
                     const int32_t* action, int B) {
    const bool goal = c && c->cfg.goal_conditioned;  // goal_conditioned: FOUR image groups, the prompt features are not read (may be null)
    if (!c || !img_inter || !img_final || (!goal && (!txt_inter || !txt_final)) || !r || !action || B <= 0) return fail("bad argument");
    for (int i = 0; i < B; ++i)
        if (action[i] < 0 || action[i] >= c->cfg.n_actions) return fail("action id out of range");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(ensure_buffers(c, B));
    const size_t E = c->cfg.embed;
    ARP_HIP_OK(hipMemcpyAsync(c->x_in[0].p, img_inter, (size_t)c->groups() * B * c->Dv() * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipMemcpyAsync(c->x_fin[0].p, img_final, (size_t)c->groups() * B * E * 4, hipMemcpyHostToDevice, c->stream));
    if (!goal) {
        ARP_HIP_OK(hipMemcpyAsync(c->x_in[1].p, txt_inter, (size_t)B * c->Dt() * 4, hipMemcpyHostToDevice, c->stream));
        ARP_HIP_OK(hipMemcpyAsync(c->x_fin[1].p, txt_final, (size_t)B * E * 4, hipMemcpyHostToDevice, c->stream));
    }
    ARP_HIP_OK(hipMemcpyAsync(c->r.p, r, (size_t)B * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipMemcpyAsync(c->action.p, action, (size_t)B * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

// Same batch with the four feature arrays already in device memory (outputs of arp_clip_encode_*_multiscale_dev).
int arp_ft_set_batch_dev(arp_ft* c, const float* img_inter_dev, const float* img_final_dev, const float* txt_inter_dev, const float* txt_final_dev,
                         const float* r, const int32_t* action, int B) {
    const bool goal = c && c->cfg.goal_conditioned;
    if (!c || !img_inter_dev || !img_final_dev || (!goal && (!txt_inter_dev || !txt_final_dev)) || !r || !action || B <= 0) return fail("bad argument");
    for (int i = 0; i < B; ++i)
        if (action[i] < 0 || action[i] >= c->cfg.n_actions) return fail("action id out of range");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(ensure_buffers(c, B));
    const size_t E = c->cfg.embed;
    ARP_HIP_OK(hipMemcpyAsync(c->x_in[0].p, img_inter_dev, (size_t)c->groups() * B * c->Dv() * 4, hipMemcpyDeviceToDevice, c->stream));
    ARP_HIP_OK(hipMemcpyAsync(c->x_fin[0].p, img_final_dev, (size_t)c->groups() * B * E * 4, hipMemcpyDeviceToDevice, c->stream));
    if (!goal) {
        ARP_HIP_OK(hipMemcpyAsync(c->x_in[1].p, txt_inter_dev, (size_t)B * c->Dt() * 4, hipMemcpyDeviceToDevice, c->stream));
        ARP_HIP_OK(hipMemcpyAsync(c->x_fin[1].p, txt_final_dev, (size_t)B * E * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    ARP_HIP_OK(hipMemcpyAsync(c->r.p, r, (size_t)B * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipMemcpyAsync(c->action.p, action, (size_t)B * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_ft_forward(arp_ft* c, float* metrics4, float* scores, float* logits) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_ft_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(c->cfg.mode == ARP_MODE_BF16 ? forward<bf16_t>(c) : (c->cfg.mode == ARP_MODE_F16 ? forward<f16_t>(c) : forward<float>(c)));
    if (metrics4) ARP_HIP_OK(hipMemcpyAsync(metrics4, c->metrics.p, 16, hipMemcpyDeviceToHost, c->stream));
    if (scores) ARP_HIP_OK(hipMemcpyAsync(scores, c->scores.p, (size_t)3 * c->B * 4, hipMemcpyDeviceToHost, c->stream));
    if (logits) ARP_HIP_OK(hipMemcpyAsync(logits, c->logits.p, (size_t)c->B * c->cfg.n_actions * 4, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

// Inference half of one tower's head (model.encode_image / model.encode_text of the clip_ft labelling branch,
// arp_dt/label_reward.py:165-230): tower features [n, .] -> adapted, L2-normalised features [n, F].  Reuses the step's
// workspaces, so a batch staged with arp_ft_set_batch must be staged again afterwards.
int arp_ft_encode(arp_ft* c, int which, const float* inter, const float* final_feat, int n, float* out) {
    if (!c || !inter || !final_feat || !out || n <= 0 || which < 0 || which > 1) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(ensure_buffers(c, which == 0 ? (n + c->groups() - 1) / c->groups() : n));
    c->B = 0;  // the staged batch (if any) is gone
    const size_t Din = which == 0 ? c->Dv() : c->Dt();
    ARP_HIP_OK(hipMemcpyAsync(c->x_in[which].p, inter, (size_t)n * Din * 4, hipMemcpyHostToDevice, c->stream));
    ARP_HIP_OK(hipMemcpyAsync(c->x_fin[which].p, final_feat, (size_t)n * c->cfg.embed * 4, hipMemcpyHostToDevice, c->stream));
    if (c->cfg.mode == ARP_MODE_BF16) { ARP_TRY(refresh_shadows<bf16_t>(c)); ARP_TRY(encode_tower<bf16_t>(c, which, n)); }
    else if (c->cfg.mode == ARP_MODE_F16) { ARP_TRY(refresh_shadows<f16_t>(c)); ARP_TRY(encode_tower<f16_t>(c, which, n)); }
    else { ARP_TRY(refresh_shadows<float>(c)); ARP_TRY(encode_tower<float>(c, which, n)); }
    ARP_HIP_OK(hipMemcpyAsync(out, c->a[which].p, (size_t)n * c->F() * 4, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_ft_backward(arp_ft* c) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_ft_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    c->grads_gone.clear();  // every gradient is stored by this call
    if (c->cfg.mode == ARP_MODE_BF16) { ARP_TRY(forward<bf16_t>(c)); ARP_TRY(backward<bf16_t>(c)); }
    else if (c->cfg.mode == ARP_MODE_F16) { ARP_TRY(forward<f16_t>(c)); ARP_TRY(backward<f16_t>(c)); }
    else { ARP_TRY(forward<float>(c)); ARP_TRY(backward<float>(c)); }
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int arp_ft_train_step_async(arp_ft* c, float lr) {
    if (!c) return fail("null handle");
    if (c->B <= 0) return fail("no batch staged: call arp_ft_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_F16) return step_impl<f16_t>(c, lr, nullptr);
    return c->cfg.mode == ARP_MODE_BF16 ? step_impl<bf16_t>(c, lr, nullptr) : step_impl<float>(c, lr, nullptr);
}
int arp_ft_train_step(arp_ft* c, float lr, float* aux4) {
    if (!c || !aux4) return fail("null argument");
    if (c->B <= 0) return fail("no batch staged: call arp_ft_set_batch first");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    if (c->cfg.mode == ARP_MODE_F16) return step_impl<f16_t>(c, lr, aux4);
    return c->cfg.mode == ARP_MODE_BF16 ? step_impl<bf16_t>(c, lr, aux4) : step_impl<float>(c, lr, aux4);
}
int arp_ft_sync(arp_ft* c) {
    if (!c) return fail("null handle");
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}
int arp_ft_event_record(arp_ft* c, arp_event* e) {
    if (!c || !e) return fail("null argument");
    ARP_HIP_OK(hipEventRecord(e->e, c->stream));
    return 0;
}
// Data parallelism (BASELINE configs[4]): the id comes from arp_dt_comm_unique_id (one RCCL id serves any handle type)
// Flat-gradient ranges of the data-parallel step's seven all-reduce buckets in production order, from the configuration alone (no
// GPU): ranges28 = {lo, hi} x 14 in floats (bucket b = ranges 2b, 2b + 1), *total = the flat parameter count.
int arp_ft_bucket_plan(const arp_ft_cfg* cfg, int64_t* ranges28, int64_t* total) {
    if (!cfg || !ranges28 || !total) return fail("null argument");
    arp_ft tmp;
    tmp.cfg = *cfg;
    build_layout(&tmp);
    const FtBuckets b = ft_buckets(&tmp);
    for (int i = 0; i < 2 * FT_BUCKETS; ++i) {
        ranges28[2 * i] = (int64_t)b.lo[i];
        ranges28[2 * i + 1] = (int64_t)b.hi[i];
    }
    *total = (int64_t)tmp.P;
    return 0;
}

int arp_ft_comm_init(arp_ft* c, const void* id128, int world, int rank) {
    if (!c || !id128 || world <= 0 || rank < 0 || rank >= world) return fail("bad argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "unexpected ncclUniqueId size");
    memcpy(&id, id128, 128);
    if (!rccl_api()) return fail("librccl.so.1 could not be loaded");
    if (!c->comm_stream) ARP_HIP_OK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));  // (before RCCL's own queues: arp_dt.hip::arp_dt_comm_init)
    if (ncclResult_t r = rccl_api()->CommInitRank(&c->comm, world, id, rank); r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    c->has_comm = true;
    c->world = world;
    c->rank = rank;
    return 0;
}

int arp_ft_comm_info(arp_ft* c, int32_t* info5) {
    if (!c || !info5) return fail("null argument");
    return rccl_comm_info(c->comm, c->has_comm, c->cfg.device, info5);
}
int arp_ft_comm_selfcheck(arp_ft* c, double* sum) {
    if (!c || !sum) return fail("null argument");
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    ARP_TRY(c->scal.ensure(64));
    return rccl_selfcheck(c->comm, c->has_comm, c->stream, c->scal.as<float>(), c->rank, sum);
}

// every rank takes rank 0's parameters, AdamW moments and step counter (what loading one checkpoint on every rank gives)
int arp_ft_broadcast_state(arp_ft* c) {
    if (!c) return fail("null handle");
    if (!c->has_comm) return 0;
    ARP_HIP_OK(hipSetDevice(c->cfg.device));
    DevBuf* fb[] = {&c->params, &c->mu, &c->nu};
    for (auto* b : fb)
        if (ncclResult_t r = rccl_api()->Broadcast(b->p, b->p, c->P, ncclFloat, 0, c->comm, c->stream); r != ncclSuccess) return rccl_fail("ncclBroadcast", r);
    ARP_TRY(c->scal.ensure(64));
    long long st = c->step;
    ARP_HIP_OK(hipMemcpyAsync(c->scal.p, &st, 8, hipMemcpyHostToDevice, c->stream));
    if (ncclResult_t r = rccl_api()->Broadcast(c->scal.p, c->scal.p, 8, ncclChar, 0, c->comm, c->stream); r != ncclSuccess) return rccl_fail("ncclBroadcast(step)", r);
    ARP_HIP_OK(hipMemcpyAsync(&st, c->scal.p, 8, hipMemcpyDeviceToHost, c->stream));
    ARP_HIP_OK(hipStreamSynchronize(c->stream));
    c->step = st;
    c->shadows_stale = true;
    c->transposed_stale = true;
    return 0;
}

int arp_ft_profile_enable(arp_ft* c, int on) {
    if (!c) return fail("null handle");
    c->prof.on = on != 0;
    return 0;
}
int arp_ft_profile_reset(arp_ft* c) {
    if (!c) return fail("null handle");
    c->prof.reset();
    return 0;
}
int arp_ft_profile_json(arp_ft* c, char* buf, int buf_len) {
    if (!c || !buf) return fail("null argument");
    const std::string s = c->prof.json();
    if ((int)s.size() + 1 > buf_len) return fail("profile buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"

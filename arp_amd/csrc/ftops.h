// Glue kernels of the CLIP multi-scale adapter fine-tune step (reference:
// finetune_module/clip_multiscale_adapter.py:134-250, finetune_module/finetune.py:139-141).  The contractions run on the
// MFMA GEMMs of gemm.h / gemm256.h; these are the row-wise pieces around them.  Fixed-order reductions only.
#pragma once
#include "common.h"

namespace arp {

// dst[m, col0 + c] = src[m, c]   (the CLIP feature behind the projected per-block features: torch.cat, :143 / :166)
static __global__ __launch_bounds__(256) void ft_copy_cols_kernel(const float* __restrict__ src, int w, float* __restrict__ dst, int ldd, int col0,
                                                                  int M) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * w) return;
    const int m = (int)(i / w), c = (int)(i - (size_t)m * w);
    dst[(size_t)m * ldd + col0 + c] = src[i];
}

__device__ __forceinline__ float ft_block_sum(float v, float* red) {  // 256 threads, result to every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = res*f + (1-res)*A ; a = y / max(||y||, 1e-12)   (res weights the ORIGINAL feature here, :145-148 / :168-170)
static __global__ __launch_bounds__(256) void ft_mix_norm_fwd_kernel(const float* __restrict__ f, const float* __restrict__ A,
                                                                     const float* __restrict__ rw, float* __restrict__ a_out,
                                                                     float* __restrict__ nrm, int F) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const float res = 1.0f / (1.0f + expf(-rw[0]));
    const float* fr = f + (size_t)m * F;
    const float* ar = A + (size_t)m * F;
    float ss = 0.f;
    for (int c = threadIdx.x; c < F; c += 256) {
        const float y = res * fr[c] + (1.f - res) * ar[c];
        ss += y * y;
    }
    ss = ft_block_sum(ss, red);
    const float n = fmaxf(sqrtf(ss), 1e-12f);
    if (threadIdx.x == 0) nrm[m] = n;
    for (int c = threadIdx.x; c < F; c += 256) a_out[(size_t)m * F + c] = (res * fr[c] + (1.f - res) * ar[c]) / n;
}

// s[k*B + b] = scale * <a[k*B + b], t[b]>   (the diagonal of logit_scale * a_k t^T, :203-207)
static __global__ __launch_bounds__(256) void ft_scores_kernel(const float* __restrict__ a, const float* __restrict__ t, float scale,
                                                               float* __restrict__ s, int B, int F) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row % B;
    float d = 0.f;
    for (int c = threadIdx.x; c < F; c += 256) d += a[(size_t)row * F + c] * t[(size_t)b * F + c];
    d = ft_block_sum(d, red);
    if (threadIdx.x == 0) s[row] = scale * d;
}

// goal_conditioned (:208-212): s[k*B + b] = -||a[3B + b] - a[k*B + b]||, k = 0..2; the distance is kept for the backward
static __global__ __launch_bounds__(256) void ft_goal_scores_kernel(const float* __restrict__ a, float* __restrict__ s, float* __restrict__ dist, int B,
                                                                    int F) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row % B;
    float q = 0.f;
    for (int c = threadIdx.x; c < F; c += 256) {
        const float d = a[((size_t)3 * B + b) * F + c] - a[(size_t)row * F + c];
        q += d * d;
    }
    q = ft_block_sum(q, red);
    if (threadIdx.x == 0) {
        const float n = sqrtf(q);
        s[row] = -n;
        dist[row] = n;
    }
}
// ... and its backward together with the inverse-model input's ([a1 | a3 | a2 | a3], :224-230):
//   d s_k / d a_k = (a3 - a_k) / ||.||,  d s_k / d a3 = -(a3 - a_k) / ||.||   (torch.linalg.norm's gradient)
static __global__ __launch_bounds__(256) void ft_goal_feat_grad_kernel(const float* __restrict__ ds, const float* __restrict__ a,
                                                                       const float* __restrict__ dist, const float* __restrict__ dC,
                                                                       float* __restrict__ da, int B, int F) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * F) return;
    const int b = (int)(i / F), c = (int)(i - (size_t)b * F);
    const float gv = a[((size_t)3 * B + b) * F + c];
    const float* dc = dC + (size_t)b * 4 * F;
    float acc = dc[F + c] + dc[3 * F + c];
    for (int k = 0; k < 3; ++k) {
        const size_t row = (size_t)k * B + b;
        const float u = ds[row] * (gv - a[row * F + c]) / fmaxf(dist[row], 1e-30f);
        da[row * F + c] = u + (k == 1 ? dc[c] : k == 2 ? dc[2 * F + c] : 0.f);
        acc -= u;
    }
    da[((size_t)3 * B + b) * F + c] = acc;
}

// VIP loss + inverse-dynamics cross entropy and their gradients w.r.t. the scores / logits (:214-246), single block.
// The exponent of the VIP term broadcasts r [B,1] against the scores [B] to a [B,B] matrix in the reference; its mean
// factorises as mean_i exp(-(r_i - 1)) * mean_j exp(-(gamma s2_j - s1_j)).
// metrics: [0] loss, [1] vip_loss, [2] id_loss, [3] lambda_id.   dlambda = d loss / d lambda_id.
static __global__ __launch_bounds__(256) void ft_loss_kernel(const float* __restrict__ s, const float* __restrict__ r, const float* __restrict__ logits,
                                                             const int* __restrict__ action, int B, int NA, float gamma,
                                                             const float* __restrict__ lambda_id, int use_vip, int use_id,
                                                             float* __restrict__ metrics, float* __restrict__ ds, float* __restrict__ dlogits,
                                                             float* __restrict__ dlambda, float gs = 1.f) {
    // gs: power-of-two scale on every gradient this kernel seeds (f16 mode: the 16-bit gradient activations downstream would sit
    // below binary16's normal range otherwise); everything after it is linear, AdamW / arp_ft_get_tensor take it out again
    __shared__ float red[4];
    const float lam = lambda_id[0];
    float s0 = 0.f, er = 0.f, ee = 0.f, ce = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        s0 += s[b];
        er += expf(-(r[b] - 1.0f));
        ee += expf(-(gamma * s[2 * B + b] - s[B + b]));
        const float* l = logits + (size_t)b * NA;
        float mx = l[0];
        for (int c = 1; c < NA; ++c) mx = fmaxf(mx, l[c]);
        float sum = 0.f;
        for (int c = 0; c < NA; ++c) sum += expf(l[c] - mx);
        const float lse = logf(sum) + mx;
        ce += lse - l[action[b]];
        for (int c = 0; c < NA; ++c)
            dlogits[(size_t)b * NA + c] = use_id ? gs * lam * (expf(l[c] - lse) - (c == action[b] ? 1.f : 0.f)) / (float)B : 0.f;
    }
    s0 = ft_block_sum(s0, red);
    er = ft_block_sum(er, red);
    ee = ft_block_sum(ee, red);
    ce = ft_block_sum(ce, red);
    const float Rm = er / (float)B, Z = Rm * ee / (float)B;
    const float vip = (1.f - gamma) * -(s0 / (float)B) + logf(1e-8f + Z);
    const float idl = ce / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float e = expf(-(gamma * s[2 * B + b] - s[B + b]));
        const float w = use_vip ? Rm * e / ((float)B * (1e-8f + Z)) : 0.f;
        ds[b] = use_vip ? gs * -(1.f - gamma) / (float)B : 0.f;
        ds[B + b] = gs * w;
        ds[2 * B + b] = gs * -gamma * w;
    }
    if (threadIdx.x == 0) {
        metrics[0] = (use_vip ? vip : 0.f) + (use_id ? lam * idl : 0.f);
        metrics[1] = vip;
        metrics[2] = idl;
        metrics[3] = lam;
        dlambda[0] = use_id ? gs * idl : 0.f;
    }
}

// out[m, n] = bias[n] + sum_k A[m, k] W[n, k]: ONE WAVE per output element, for products with a handful of outputs and a long contraction (the inverse
// model's logits: 64 x 15 outputs, K = 1024 -- on the 32 x 32-tile small GEMM that was two workgroups walking 1024 k one after the other, 70 us of a
// 3.4 ms step).  The k order is fixed (lane-strided float4 chunks, then the wave's butterfly): bit-reproducible.
static __global__ __launch_bounds__(256) void ft_rowdot_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ out, int M, int N, int K, int lda, int ldw) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= M * N) return;
    const int m = o / N, n = o - m * N;
    const float* a = A + (size_t)m * lda;
    const float* w = W + (size_t)n * ldw;
    float s = 0.f;
    int k = lane * 4;
    for (; k + 3 < K; k += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + k), y = *reinterpret_cast<const float4*>(w + k);
        s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    // (K is a multiple of 4: the caller checks)
    s = wave_sum(s);
    if (lane == 0) out[o] = s + (bias ? bias[n] : 0.f);
}

// C[b] = [a1, t, a2, t]   (:232-235)
static __global__ __launch_bounds__(256) void ft_build_c_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ C,
                                                                int B, int F) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * 4 * F) return;
    const int b = (int)(i / (4 * F)), c = (int)(i - (size_t)b * 4 * F);
    const int seg = c / F, cc = c - seg * F;
    C[i] = seg == 0 ? a[((size_t)B + b) * F + cc] : seg == 2 ? a[((size_t)2 * B + b) * F + cc] : t[(size_t)b * F + cc];
}

// da[k*B+b] = ds[k*B+b]*scale*t[b] + (dC part of a_k);   dt[b] = scale * sum_k ds[k*B+b]*a[k*B+b] + dC[b, F:2F] + dC[b, 3F:4F]
static __global__ __launch_bounds__(256) void ft_feat_grad_kernel(const float* __restrict__ ds, const float* __restrict__ a,
                                                                  const float* __restrict__ t, const float* __restrict__ dC, float scale,
                                                                  float* __restrict__ da, float* __restrict__ dt, int B, int F) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * F) return;
    const int b = (int)(i / F), c = (int)(i - (size_t)b * F);
    const float tv = t[i];
    const float* dc = dC + (size_t)b * 4 * F;
    float acc = 0.f;
    for (int k = 0; k < 3; ++k) {
        const size_t row = (size_t)k * B + b;
        const float d = ds[row] * scale;
        acc += d * a[row * F + c];
        da[row * F + c] = d * tv + (k == 1 ? dc[c] : k == 2 ? dc[2 * F + c] : 0.f);
    }
    dt[i] = acc + dc[F + c] + dc[3 * F + c];
}

// backward of normalise + mix for one row:  dy = (da - a <a, da>) / n ;  dA = (1-res) dy ;  dfd = res dy ;
// dres_part[m] = <dy, f - A>
static __global__ __launch_bounds__(256) void ft_mix_norm_bwd_kernel(const float* __restrict__ da, const float* __restrict__ a,
                                                                     const float* __restrict__ nrm, const float* __restrict__ f,
                                                                     const float* __restrict__ A, const float* __restrict__ rw,
                                                                     float* __restrict__ dA, float* __restrict__ dfd, float* __restrict__ dres_part,
                                                                     int F) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const float res = 1.0f / (1.0f + expf(-rw[0]));
    const size_t o = (size_t)m * F;
    float dot = 0.f;
    for (int c = threadIdx.x; c < F; c += 256) dot += a[o + c] * da[o + c];
    dot = ft_block_sum(dot, red);
    const float inv = 1.0f / nrm[m];
    float dr = 0.f;
    for (int c = threadIdx.x; c < F; c += 256) {
        const float dy = (da[o + c] - a[o + c] * dot) * inv;
        dA[o + c] = (1.f - res) * dy;
        dfd[o + c] = res * dy;
        dr += dy * (f[o + c] - A[o + c]);
    }
    dr = ft_block_sum(dr, red);
    if (threadIdx.x == 0) dres_part[m] = dr;
}

// torch.optim.AdamW: decoupled decay on EVERY parameter (finetune.py:141 passes model.parameters()), bias correction
// mirror (bf16 mode): the operand-type copy of the whole flat parameter vector, refreshed here so that the next step's
// forward GEMMs need no separate conversion pass over the f32 parameters.
constexpr int FT_SKIP_RANGES = 6;
struct FtSkip { size_t lo[FT_SKIP_RANGES], hi[FT_SKIP_RANGES]; };  // flat ranges of parameters without a gradient (empty: lo = hi = 0)
template <typename TM>
static __global__ __launch_bounds__(256) void ft_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu,
                                                              float* __restrict__ nu, float gscale, float lr, float wd, float b1, float b2,
                                                              float eps, float bc1, float bc2, size_t n, TM* __restrict__ mirror, FtSkip skip,
                                                              int mask_nonfinite, unsigned int* __restrict__ dropped) {
    // one float4 per thread (every tensor's offset and padded size are multiples of 4, so are the skip ranges): 14.3 GB per step at
    // full size is the fine-tune step's largest item
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    // torch.optim.AdamW skips parameters whose .grad is None -- no decay, no moment update (with use_id_loss off the inverse
    // model and lambda_id never receive a gradient, with goal_conditioned on the text head does not: finetune.py:141 +
    // clip_multiscale_adapter.py:177-250)
#pragma unroll
    for (int r = 0; r < FT_SKIP_RANGES; ++r)
        if (i >= skip.lo[r] && i < skip.hi[r]) return;
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(mu + i), vv = *reinterpret_cast<float4*>(nu + i);
    float gg[4] = {gv.x, gv.y, gv.z, gv.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, nn[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gi = gg[e] * gscale;
        // f16 mode seeds every gradient x 1024 (arp_ft.hip): an entry that overflowed binary16 on the way arrives as inf / NaN and would
        // poison this parameter and both moments for good -- it is treated as a missing (zero) gradient for this step AND COUNTED
        // (arp_ft_dropped_gradients).  In the other modes a non-finite gradient is a diverged step or a bug and stays visible as NaNs,
        // as in torch.
        if (mask_nonfinite) {  // counted once per wave instruction (ballot + popcount), not once per element
            const bool bad = !(fabsf(gi) < 3.0e38f);
            const unsigned long long bal = __ballot(bad);
            if (bad) gi = 0.f;
            if (bal && (int)(threadIdx.x & 63) == __ffsll((long long)bal) - 1) atomicAdd(dropped, (unsigned int)__popcll(bal));
        }
        mm[e] = b1 * mm[e] + (1.f - b1) * gi;
        nn[e] = b2 * nn[e] + (1.f - b2) * gi * gi;
        const float pd = pp[e] * (1.f - lr * wd);
        pp[e] = pd - (lr / bc1) * mm[e] / (sqrtf(nn[e]) / sqrtf(bc2) + eps);
    }
    *reinterpret_cast<float4*>(mu + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4*>(nu + i) = make_float4(nn[0], nn[1], nn[2], nn[3]);
    *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    if constexpr (sizeof(TM) == 2) {
        if (mirror) store4(mirror + i, pp[0], pp[1], pp[2], pp[3]);
    }
}

}  // namespace arp

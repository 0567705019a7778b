// Row-wise (HBM-bound) kernels of the CLIP towers: LayerNorm, token assembly + ln_pre, text token
// embedding, feature normalisation + image.text similarity.  One 64-lane wave per row, float4 /
// 8-byte-bf16 vector accesses, wavefront shuffle reductions, statistics in f32.
#pragma once
#include "common.h"

namespace arp {

constexpr int ROW_MAX_V4 = 8;  // up to 8 float4 per lane -> rows up to 2048 wide
// NV = float4s per lane actually needed, ceil(D / 256); kernels are instantiated for 1,2,3,4,8 so a
// 768-wide row costs 12 VGPRs of payload, not 32 (occupancy matters: these kernels are HBM-bound).
#define ARP_NV_DISPATCH(D, CALL)                   \
    do {                                           \
        const int _nv = ((D) + 255) / 256;         \
        if (_nv <= 1) { CALL(1); }                 \
        else if (_nv == 2) { CALL(2); }            \
        else if (_nv == 3) { CALL(3); }            \
        else if (_nv == 4) { CALL(4); }            \
        else { CALL(8); }                          \
    } while (0)

// LayerNorm of one row held as v[nv][4] (lane-strided float4s); two-pass statistics in f32, the
// same formula as the reference (arp_dt/models/openai/layers.py:9, eps 1e-5; flax nn.LayerNorm
// eps 1e-6 for the policy: arp_dt/layers.py:126).
template <typename OutT, int NV>
__device__ __forceinline__ void ln_row_store(float (&v)[NV][4], int D, int lane, const float* __restrict__ w,
                                             const float* __restrict__ b, float eps, OutT* __restrict__ orow) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float4 ww = *reinterpret_cast<const float4*>(w + c);
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            if constexpr (__is_same(OutT, f16c_t) || __is_same(OutT, f16c2_t))  // ARP_MODE_F16C: [hi | x4 (| dx4)] operand row of a gemm256 MIXC product (common.h)
                store_f16c<__is_same(OutT, f16c2_t)>(reinterpret_cast<f16_t*>(orow), c, D, (v[i][0] - mean) * rstd * ww.x + bb.x, (v[i][1] - mean) * rstd * ww.y + bb.y,
                                                      (v[i][2] - mean) * rstd * ww.z + bb.z, (v[i][3] - mean) * rstd * ww.w + bb.w);
            else if constexpr (__is_same(OutT, f16x3_t))  // ARP_MODE_F16X3: the next GEMM's K-concatenated operand, written here instead of by a split pass
                store_split3(reinterpret_cast<f16_t*>(orow) + c, (size_t)D, (v[i][0] - mean) * rstd * ww.x + bb.x, (v[i][1] - mean) * rstd * ww.y + bb.y,
                             (v[i][2] - mean) * rstd * ww.z + bb.z, (v[i][3] - mean) * rstd * ww.w + bb.w);
            else
            store4(orow + c, (v[i][0] - mean) * rstd * ww.x + bb.x, (v[i][1] - mean) * rstd * ww.y + bb.y,
                   (v[i][2] - mean) * rstd * ww.z + bb.z, (v[i][3] - mean) * rstd * ww.w + bb.w);
        }
    }
}

// out[r, :] = LN(in[r * in_stride .. + D])      (in f32; out T).  D % 4 == 0, D <= 2048.
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, size_t in_stride, OutT* __restrict__ out,
                                                        int out_stride, const float* __restrict__ w,
                                                        const float* __restrict__ b, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* irow = in + (size_t)row * in_stride;
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) load4(irow + c, v[i]);
    }
    ln_row_store<OutT, NV>(v, D, lane, w, b, eps, out + (size_t)row * out_stride);
}

// Gathered variant: row r reads in[row_idx[r] * in_stride ..] (text tower: EOT rows).
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void layernorm_gather_kernel(const float* __restrict__ in, size_t in_stride,
                                                               const int* __restrict__ row_idx, OutT* __restrict__ out,
                                                               int out_stride, const float* __restrict__ w,
                                                               const float* __restrict__ b, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* irow = in + (size_t)row_idx[row] * in_stride;
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) load4(irow + c, v[i]);
    }
    ln_row_store<OutT, NV>(v, D, lane, w, b, eps, out + (size_t)row * out_stride);
}

// ViT token assembly + ln_pre (arp_dt/models/openai/layers.py:301-322):
//   row (b, 0)   = class_embedding + pos[0]
//   row (b, 1+p) = patch_embed[b*GG + p] + pos[1+p]
//   x = ln_pre(row)                     -> f32 residual stream [B*ntok, D]
template <int NV>
__global__ __launch_bounds__(256) void vit_assemble_lnpre_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                                 const float* __restrict__ pos, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ x, int rows,
                                                                 int ntok, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int bi = row / ntok, t = row - bi * ntok;
    const float* src = (t == 0) ? cls : patch + ((size_t)bi * (ntok - 1) + (t - 1)) * D;
    const float* prow = pos + (size_t)t * D;
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            float p4[4];
            load4(src + c, v[i]);
            load4(prow + c, p4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += p4[j];
        }
    }
    ln_row_store<float, NV>(v, D, lane, w, b, eps, x + (size_t)row * D);
}

// Latency-path variant (one frame, or a few): the patch embedding arrives as S split-K slabs of the skinny GEMM, and the row that
// ln_pre produces is normalised once more with the first block's ln_1 into h -- one launch instead of three.
template <typename T, int NV>
__global__ __launch_bounds__(64) void vit_assemble_lat_kernel(const float* __restrict__ patch, int S, size_t slice_stride, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ x, T* __restrict__ h, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, int ntok, int D, float eps,
                                                              float* __restrict__ stats = nullptr, int parts = 0) {
    // stats != null (LayerNorm folded into the latency path's GEMMs): h receives the operand-type copy of x instead of ln_1(x), and
    // stats[row][0] = (sum, sum of squares) of the row, stats[row][1 .. parts) = 0 -- the layout the strip producers write
    const int lane = threadIdx.x, row = blockIdx.x;
    const int bi = row / ntok, t = row - bi * ntok;
    const float* src = (t == 0) ? cls : patch + ((size_t)bi * (ntok - 1) + (t - 1)) * D;
    const float* prow = pos + (size_t)t * D;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            float p4[4], sl[3][4];
            load4(src + c, v[i]);
            load4(prow + c, p4);
            const int nsl = t != 0 ? S - 1 : 0;  // slabs 1 .. S-1 on top of slab 0, all loads issued before the first add
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < nsl) load4(src + (size_t)(k + 1) * slice_stride + c, sl[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += p4[j];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < nsl) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] += sl[k][j];
                }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    // ln_pre in registers (the arithmetic of ln_row_store), then ln_1 of the first block on the rounded-to-f32 result
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float4 ww = *reinterpret_cast<const float4*>(w + c);
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            v[i][0] = (v[i][0] - mean) * rstd * ww.x + bb.x; v[i][1] = (v[i][1] - mean) * rstd * ww.y + bb.y;
            v[i][2] = (v[i][2] - mean) * rstd * ww.z + bb.z; v[i][3] = (v[i][3] - mean) * rstd * ww.w + bb.w;
            store4(x + (size_t)row * D + c, v[i][0], v[i][1], v[i][2], v[i][3]);
        }
    }
    if (stats) {
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                store4(h + (size_t)row * D + c, v[i][0], v[i][1], v[i][2], v[i][3]);
                su += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        }
        su = wave_sum(su);
        sq = wave_sum(sq);
        float* st = stats + (size_t)row * parts * 2;
        for (int p = lane; p < parts; p += 64) {
            st[2 * p] = p == 0 ? su : 0.f;
            st[2 * p + 1] = p == 0 ? sq : 0.f;
        }
        return;
    }
    ln_row_store<T, NV>(v, D, lane, w1, b1, eps, h + (size_t)row * D);
}

// Text token embedding (arp_dt/models/openai/layers.py:364-365): x[p*ctx + t] = tok_emb[tokens] + pos[t]
static __global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ emb,
                                                         const float* __restrict__ pos, float* __restrict__ x, int rows, int ctx,
                                                         int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int t = row % ctx;
    const float* e = emb + (size_t)tokens[row] * D;
    for (int c = lane * 4; c < D; c += 256) {
        float a[4], p4[4];
        load4(e + c, a);
        load4(pos + (size_t)t * D + c, p4);
        store4(x + (size_t)row * D + c, a[0] + p4[0], a[1] + p4[1], a[2] + p4[2], a[3] + p4[3]);
    }
}

// f[r, :] /= ||f[r, :]||   (arp_dt/models/openai/layers.py:429-439), in place, f32.
static __global__ __launch_bounds__(256) void l2_normalize_kernel(float* __restrict__ f, int rows, int E) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = f + (size_t)row * E;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += r[c] * r[c];
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < E; c += 64) r[c] *= inv;
}

// reward[i] = exp(logit_scale) * < img[i]/||img[i]||, txt_n >  -- the "image x text GEMV"
// (arp_dt/label_reward.py:140-146: logits_per_text[0]).  txt_n is already L2-normalised.
static __global__ __launch_bounds__(256) void reward_kernel(const float* __restrict__ img, const float* __restrict__ txt_n,
                                                     float scale, float* __restrict__ reward, int rows, int E) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = img + (size_t)row * E;
    float ss = 0.f, dt = 0.f;
    for (int c = lane; c < E; c += 64) {
        const float a = r[c];
        ss += a * a;
        dt += a * txt_n[c];
    }
    ss = wave_sum(ss);
    dt = wave_sum(dt);
    if (lane == 0) reward[row] = scale * (dt / sqrtf(ss));
}

}  // namespace arp

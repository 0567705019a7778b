// Fused frame preprocessing (HBM-bound integer/byte kernel):
//   uint8 NHWC frame -> [center-crop] -> Pillow-exact antialiased bicubic resize (two passes,
//   horizontal first, 22-bit fixed-point taps, uint8 round/clamp after EACH pass) -> x/255 ->
//   (x - mean)/std -> written directly in the ViT's patch-major (im2col) layout so that the
//   patch-embedding conv is a plain GEMM (or NCHW f32 for the parity-test entry point).
//
// Reference semantics: arp_dt/label_reward.py:109-121 (default) and :92-102 (use_crop); the resize
// recipe is SURVEY.md Appendix A (verified bit-exact against PIL).  The /255 and normalise steps
// are a 3x256-entry f32 lookup table computed on the host with the same IEEE f32 operations
// torchvision's ToTensor/Normalize perform, so the f32 output is bit-identical by construction.
//
// One workgroup = one (frame, tile of TR output rows).  The input rows the tile needs (<= ~42 rows
// of 768 B for 256->224, TR = 32) are fetched once with coalesced 16-byte loads into LDS, the
// horizontal pass writes a uint8 intermediate to LDS, the vertical pass reads it back and streams
// the normalised result out in 8/16-byte vectors.
#pragma once
#include "common.h"

namespace arp {

struct PreprocArgs {
    const uint8_t* frames;  // [n, H, W, 3]
    void* out;              // patch-major [n*G*G, 3*P*P] (T) or NCHW f32 [n,3,R,R]
    const int* h_tab;       // [R][1 + kmax_h]: (xmin | cnt << 16), then kmax_h int32 weights
    const int* v_tab;       // [R][1 + kmax_v]
    const float* lut;       // [3][256]
    int n, H, W;            // frame geometry
    int cy, cx, ch, cw;     // crop window (full frame when not cropping)
    int R, P;               // output resolution (224), patch size
    int kmax_h, kmax_v;
    int TR;                 // output rows per workgroup
    int max_rows;           // LDS capacity in input rows
};

enum { PRE_PATCH = 0, PRE_NCHW = 1 };

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void preprocess_kernel(PreprocArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tiles = (a.R + a.TR - 1) / a.TR;
    const int frame = blockIdx.x / tiles;
    const int oy0 = (blockIdx.x - frame * tiles) * a.TR;
    const int oy1 = min(oy0 + a.TR, a.R);
    const int hs = 1 + a.kmax_h, vs = 1 + a.kmax_v;
    const int in_row_bytes = a.cw * 3;
    const int mid_row_bytes = a.R * 3;

    // LDS carve (every offset a multiple of 16)
    int* hT = reinterpret_cast<int*>(smem);
    float* lut = reinterpret_cast<float*>(smem + ((a.R * hs * 4 + 15) & ~15));
    uint8_t* in_s = reinterpret_cast<uint8_t*>(lut + 768);
    uint8_t* mid_s = in_s + ((a.max_rows * in_row_bytes + 15) & ~15);

    for (int i = tid; i < a.R * hs; i += 256) hT[i] = a.h_tab[i];
    for (int i = tid; i < 768; i += 256) lut[i] = a.lut[i];

    const int v_first = a.v_tab[oy0 * vs];
    const int v_last = a.v_tab[(oy1 - 1) * vs];
    const int y_lo = v_first & 0xffff;
    const int y_hi = (v_last & 0xffff) + (v_last >> 16);
    const int rows = y_hi - y_lo;

    // ---- stage input rows [cy + y_lo, cy + y_hi) x [cx, cx + cw) ---------------------------------
    const uint8_t* fbase = a.frames + (size_t)frame * a.H * a.W * 3;
    if (a.cx == 0 && a.cw == a.W && (in_row_bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(fbase) & 15) == 0)) {
        const uint4* src = reinterpret_cast<const uint4*>(fbase + (size_t)(a.cy + y_lo) * in_row_bytes);
        uint4* dst = reinterpret_cast<uint4*>(in_s);
        const int nvec = rows * in_row_bytes / 16;
        for (int i = tid; i < nvec; i += 256) dst[i] = src[i];
    } else {
        for (int i = tid; i < rows * in_row_bytes; i += 256) {
            const int r = i / in_row_bytes, cb = i - r * in_row_bytes;
            in_s[i] = fbase[((size_t)(a.cy + y_lo + r) * a.W + a.cx) * 3 + cb];
        }
    }
    __syncthreads();

    // ---- horizontal pass: mid[r][ox][c] = clip8((2^21 + sum_k W[ox][k] * in[r][xmin+k][c]) >> 22) ----
    const int quads = mid_row_bytes / 4;
    for (int i = tid; i < rows * quads; i += 256) {
        const int r = i / quads, j0 = (i - r * quads) * 4;
        const uint8_t* irow = in_s + r * in_row_bytes;
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            const int ox = j / 3, c = j - ox * 3;
            const int* t = hT + ox * hs;
            const int xmin = t[0] & 0xffff, cnt = t[0] >> 16;
            int acc = 1 << 21;
            for (int k = 0; k < cnt; ++k) acc += t[1 + k] * (int)irow[(xmin + k) * 3 + c];
            acc >>= 22;
            acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
            packed |= (uint32_t)acc << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(mid_s + r * mid_row_bytes + j0) = packed;
    }
    __syncthreads();

    // ---- vertical pass + normalise + layout -------------------------------------------------------
    const int xg = a.R / 4;
    const int items = (oy1 - oy0) * 3 * xg;
    const int G = a.R / a.P;
    for (int i = tid; i < items; i += 256) {
        const int g4 = i % xg;
        const int c = (i / xg) % 3;
        const int oy = oy0 + i / (xg * 3);
        const int* t = a.v_tab + oy * vs;
        const int ymin = (t[0] & 0xffff) - y_lo, cnt = t[0] >> 16;
        int acc[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
        const int ox = g4 * 4;
        for (int k = 0; k < cnt; ++k) {
            const int w = t[1 + k];
            const uint8_t* mrow = mid_s + (ymin + k) * mid_row_bytes + ox * 3 + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * (int)mrow[e * 3];
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int q = acc[e] >> 22;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            v[e] = lut[c * 256 + q];
        }
        if constexpr (LAYOUT == PRE_PATCH) {
            const size_t prow = ((size_t)frame * G + oy / a.P) * G + ox / a.P;
            const int kidx = c * a.P * a.P + (oy % a.P) * a.P + (ox % a.P);
            store4(static_cast<T*>(a.out) + prow * (size_t)(3 * a.P * a.P) + kidx, v[0], v[1], v[2], v[3]);
        } else {
            store4(static_cast<T*>(a.out) + (((size_t)frame * 3 + c) * a.R + oy) * a.R + ox, v[0], v[1], v[2], v[3]);
        }
    }
}

// ---- fast instance for short filters (<= 8 taps per axis: every resize of this path, e.g. 256 -> 224 has 5) ---------------
// Same arithmetic as preprocess_kernel, bit for bit; what changes is how LDS is read.  The generic kernel reads one byte
// and one weight per tap per output (~19 LDS instructions per output byte over the two passes).  Here
//   * horizontal: one thread = one output pixel x 4 input rows; the pixel's window (taps x RGB <= 24 contiguous bytes) is
//     read as 7 aligned dwords per row, realigned with v_alignbyte and unpacked in registers; weights are read once per pixel;
//   * vertical: one thread = 4 output pixels x RGB = 12 contiguous bytes of the intermediate rows, 3 dword reads per tap,
//     tap weights staged in LDS; the three channels leave as three 4-wide stores.
// ~5 LDS instructions per output byte; the kernel becomes HBM-bound.
// KT = taps computed per output (>= both filters' longest; instances 5, 6, 8).  256 -> 224 needs 5: with the former fixed 8 the
// kernel multiplied 24 window bytes where 15 carry a non-zero tap and read 7 dwords where 5 hold the window (round 3).
template <typename T, int LAYOUT, int KT>
__global__ __launch_bounds__(256) void preprocess_fast_kernel(PreprocArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NE = (3 * KT + 3) / 4;  // dwords holding the window's 3 * KT bytes once realigned
    constexpr int ND = NE + 1;            // aligned dwords read (the window starts 0..3 bytes into the first)
    const int tid = threadIdx.x;
    const int tiles = (a.R + a.TR - 1) / a.TR;
    const int frame = blockIdx.x / tiles;
    const int oy0 = (blockIdx.x - frame * tiles) * a.TR;
    const int oy1 = min(oy0 + a.TR, a.R);
    const int hs = 1 + a.kmax_h, vs = 1 + a.kmax_v;
    const int in_row_bytes = a.cw * 3;
    const int mid_row_bytes = a.R * 3;

    int* hT = reinterpret_cast<int*>(smem);
    float* lut = reinterpret_cast<float*>(smem + ((a.R * hs * 4 + 15) & ~15));
    int* vT = reinterpret_cast<int*>(lut + 768);
    uint8_t* in_s = reinterpret_cast<uint8_t*>(vT) + ((a.TR * vs * 4 + 15) & ~15);
    uint8_t* mid_s = in_s + ((a.max_rows * in_row_bytes + 15) & ~15);

    // Everything the prologue needs from memory is requested at once -- the two row bounds first, then the tap tables and the normalisation LUT in batches of
    // eight unconditional loads (clamped indices) per thread.  The three plain copy loops that stood here waited one memory round trip per iteration, and the
    // row bounds another two behind them: ~11 dependent round trips before the first input row was requested, most of a workgroup's 12-15 us (round 5).
#ifdef ARP_PRE_OLD_PROLOGUE  // rounds 1-5, for A/B builds of scripts/preprocess_bench.hip
    for (int i = tid; i < a.R * hs; i += 256) hT[i] = a.h_tab[i];
    for (int i = tid; i < 768; i += 256) lut[i] = a.lut[i];
    for (int i = tid; i < (oy1 - oy0) * vs; i += 256) vT[i] = a.v_tab[oy0 * vs + i];
#endif
    const int v_first = a.v_tab[oy0 * vs];
    const int v_last = a.v_tab[(oy1 - 1) * vs];
#ifndef ARP_PRE_OLD_PROLOGUE
    {
        const int nh = a.R * hs, nv = (oy1 - oy0) * vs;
        for (int base = 0; base < nh; base += 8 * 256) {
            int hr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) hr[q] = a.h_tab[min(base + q * 256 + tid, nh - 1)];
            float lr[3];
            int vr = 0;
            if (base == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) lr[q] = a.lut[q * 256 + tid];
                vr = a.v_tab[oy0 * vs + min(tid, nv - 1)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (base + q * 256 + tid < nh) hT[base + q * 256 + tid] = hr[q];
            if (base == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) lut[q * 256 + tid] = lr[q];
                if (tid < nv) vT[tid] = vr;
            }
        }
        for (int i = tid + 256; i < nv; i += 256) vT[i] = a.v_tab[oy0 * vs + i];  // (tiles of more than 256 / vs rows: not the shipped plans)
    }
#endif
    const int y_lo = v_first & 0xffff;
    const int y_hi = (v_last & 0xffff) + (v_last >> 16);
    const int rows = y_hi - y_lo;

    const uint8_t* fbase = a.frames + (size_t)frame * a.H * a.W * 3;
    if (a.cx == 0 && a.cw == a.W && (in_row_bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(fbase) & 15) == 0)) {
        const uint4* src = reinterpret_cast<const uint4*>(fbase + (size_t)(a.cy + y_lo) * in_row_bytes);
        uint4* dst = reinterpret_cast<uint4*>(in_s);
        const int nvec = rows * in_row_bytes / 16;
        // four loads in flight per thread (a plain copy loop waits one memory round trip per iteration: six to ten of them per tile)
        int i = tid;
        for (; i + 768 < nvec; i += 1024) {
            const uint4 v0 = src[i], v1 = src[i + 256], v2 = src[i + 512], v3 = src[i + 768];
            dst[i] = v0; dst[i + 256] = v1; dst[i + 512] = v2; dst[i + 768] = v3;
        }
        for (; i < nvec; i += 256) dst[i] = src[i];
    } else {
        for (int i = tid; i < rows * in_row_bytes; i += 256) {
            const int r = i / in_row_bytes, cb = i - r * in_row_bytes;
            in_s[i] = fbase[((size_t)(a.cy + y_lo + r) * a.W + a.cx) * 3 + cb];
        }
    }
    __syncthreads();

    // ---- horizontal pass ------------------------------------------------------------------------------------------
    const int ngroups = (rows + 3) >> 2;
    for (int i = tid; i < ngroups * a.R; i += 256) {
        const int rg = i / a.R, ox = i - rg * a.R;
        const int* t = hT + ox * hs;
        const int start = (t[0] & 0xffff) * 3, a0 = start & ~3, sh = start & 3;
        int w[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) w[k] = k < a.kmax_h ? t[1 + k] : 0;  // zero beyond the filter's own length
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = min(rg * 4 + rr, rows - 1);
            const uint32_t* p = reinterpret_cast<const uint32_t*>(in_s + r * in_row_bytes + a0);
            uint32_t d[ND], e[NE];
#pragma unroll
            for (int q = 0; q < ND; ++q) d[q] = p[q];
#pragma unroll
            for (int q = 0; q < NE; ++q) e[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
            int acc[3] = {1 << 21, 1 << 21, 1 << 21};
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int b = 3 * k + c;
                    // taps are 22-bit fixed point (|w| < 2^23) and samples 8-bit: the full-rate 24-bit multiply is exact
                    acc[c] += __mul24(w[k], (int)((e[b >> 2] >> (8 * (b & 3))) & 0xffu));
                }
            if (rg * 4 + rr < rows) {
                uint8_t* m = mid_s + r * mid_row_bytes + ox * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    int q = acc[c] >> 22;
                    q = q < 0 ? 0 : (q > 255 ? 255 : q);
                    m[c] = (uint8_t)q;
                }
            }
        }
    }
    __syncthreads();

    // ---- vertical pass + normalise + layout --------------------------------------------------------------------------
    const int g12n = a.R / 4;  // groups of 4 pixels = 12 bytes
    const int G = a.R / a.P;
    for (int i = tid; i < (oy1 - oy0) * g12n; i += 256) {
        const int oyl = i / g12n, g12 = i - oyl * g12n;
        const int oy = oy0 + oyl;
        const int* t = vT + oyl * vs;
        const int ymin = (t[0] & 0xffff) - y_lo;
        int acc[12];
#pragma unroll
        for (int b = 0; b < 12; ++b) acc[b] = 1 << 21;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int wk = k < a.kmax_v ? t[1 + k] : 0;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(mid_s + min(ymin + k, rows - 1) * mid_row_bytes + g12 * 12);
            const uint32_t d[3] = {p[0], p[1], p[2]};
#pragma unroll
            for (int b = 0; b < 12; ++b) acc[b] += __mul24(wk, (int)((d[b >> 2] >> (8 * (b & 3))) & 0xffu));
        }
        const int ox = g12 * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int q = acc[e * 3 + c] >> 22;
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                v[e] = lut[c * 256 + q];
            }
            if constexpr (LAYOUT == PRE_PATCH) {
                const size_t prow = ((size_t)frame * G + oy / a.P) * G + ox / a.P;
                const int kidx = c * a.P * a.P + (oy % a.P) * a.P + (ox % a.P);
                store4(static_cast<T*>(a.out) + prow * (size_t)(3 * a.P * a.P) + kidx, v[0], v[1], v[2], v[3]);
            } else {
                store4(static_cast<T*>(a.out) + (((size_t)frame * 3 + c) * a.R + oy) * a.R + ox, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- the fine-tune path's transform (finetune_module/clip_multiscale_adapter.py:120-132) --------------------------------
// x.float() -> torchvision resize on a tensor = bilinear, align_corners = False, no antialias (only when BOTH sides
// differ from 224, :127) -> x / 255 -> (x - mean) / std.  Same arithmetic order as torch's upsample_bilinear2d:
// src = (dst + 0.5) * in/out - 0.5 clamped at 0;  v = h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11).
template <typename T>
__global__ __launch_bounds__(256) void preprocess_bilinear_kernel(const uint8_t* __restrict__ frames, T* __restrict__ out, int n, int H, int W, int R,
                                                                  int P, int resize) {
    const int G = R / P;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread = 4 consecutive ox
    const size_t total = (size_t)n * 3 * R * (R / 4);
    if (i >= total) return;
    const int ox4 = (int)(i % (R / 4));
    const int oy = (int)((i / (R / 4)) % R);
    const int c = (int)((i / ((size_t)(R / 4) * R)) % 3);
    const int frame = (int)(i / ((size_t)(R / 4) * R * 3));
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const uint8_t* f = frames + (size_t)frame * H * W * 3;
    float v[4];
    const float sy = (float)H / (float)R, sx = (float)W / (float)R;
    float fy = resize ? ((float)oy + 0.5f) * sy - 0.5f : (float)oy;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float h1 = fy - (float)y0, h0 = 1.f - h1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = ox4 * 4 + k;
        float fx = resize ? ((float)ox + 0.5f) * sx - 0.5f : (float)ox;
        fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float w1 = fx - (float)x0, w0 = 1.f - w1;
        const float v00 = f[((size_t)y0 * W + x0) * 3 + c], v01 = f[((size_t)y0 * W + x1) * 3 + c];
        const float v10 = f[((size_t)y1 * W + x0) * 3 + c], v11 = f[((size_t)y1 * W + x1) * 3 + c];
        const float px = resize ? h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11) : v00;
        v[k] = (px / 255.0f - mean[c]) / stdv[c];
    }
    const int ox = ox4 * 4;
    const size_t prow = ((size_t)frame * G + oy / P) * G + ox / P;
    const int kidx = c * P * P + (oy % P) * P + (ox % P);
    store4(out + prow * (size_t)(3 * P * P) + kidx, v[0], v[1], v[2], v[3]);
}

}  // namespace arp

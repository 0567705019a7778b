// Host side of the frame preprocessing (Pillow-exact resample tables, the per-geometry LDS plan, the normalisation lookup table and the
// launch): shared by arp_clip.hip and the standalone harness scripts/preprocess_bench.hip.  Kernels: preprocess.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "preprocess.h"
#include "runtime.h"

namespace arp {

// ---- Pillow resample table ------------------------------------------------------------------------
static inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

inline void build_bicubic_table(int in_size, int out_size, ResampleTable& t) {
    const double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    t.ksize = ksize;
    t.kmax = 0;
    t.xmin.assign(out_size, 0);
    t.cnt.assign(out_size, 0);
    t.w.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (double)(1 << 22);
            t.w[(size_t)xx * ksize + x] = (k[x] < 0) ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        t.xmin[xx] = xmin;
        t.cnt[xx] = xmax;
        t.kmax = std::max(t.kmax, xmax);
    }
}

// python round() / torchvision CenterCrop offset: round-half-to-even
static inline int round_half_even(double v) { return (int)std::nearbyint(v); }

struct ResizePlan {
    int H = 0, W = 0, use_crop = 0, R = 0;
    int cy = 0, cx = 0, ch = 0, cw = 0;
    int kmax_h = 0, kmax_v = 0;
    int TR = 32, max_rows = 0;
    size_t lds_bytes = 0;
    DevBuf h_tab, v_tab;
};

// Packs rows [first, first + R) of a table as [R][1 + kmax]: (xmin | cnt << 16), weights...
static void pack_table(const ResampleTable& t, int first, int R, std::vector<int>& out) {
    const int stride = 1 + t.kmax;
    out.assign((size_t)R * stride, 0);
    for (int o = 0; o < R; ++o) {
        const int src = first + o;
        out[(size_t)o * stride] = t.xmin[src] | (t.cnt[src] << 16);
        for (int k = 0; k < t.cnt[src]; ++k) out[(size_t)o * stride + 1 + k] = t.w[(size_t)src * t.ksize + k];
    }
}

// tr_start: the largest tile (output rows per workgroup) tried.  32 for batches (R / 32 = 7 workgroups per frame, thousands per launch);
// the single-frame pass asks for 4 (56 workgroups for its one frame: 17 -> ~5 us)
static int build_plan(int H, int W, int use_crop, int R, ResizePlan& p, int tr_start = 32) {
    if (H <= 0 || W <= 0 || R <= 0 || (R & 3)) return fail("preprocess: bad geometry");
    p.H = H; p.W = W; p.use_crop = use_crop; p.R = R;
    int top = 0, left = 0;  // CenterCrop(R) offsets after the resize (non-crop transform)
    int oh = R, ow = R;
    if (use_crop) {
        // label_reward.py:92-104: CenterCrop(image_size // 2), image_size = frame width, then Resize(R)
        const int crop = W / 2;
        if (crop <= 0 || crop > H) return fail("preprocess: use_crop needs H >= W/2");
        p.ch = p.cw = crop;
        p.cy = round_half_even((H - crop) / 2.0);
        p.cx = round_half_even((W - crop) / 2.0);
    } else {
        p.cy = p.cx = 0; p.ch = H; p.cw = W;
        // torchvision Resize(int): shorter side -> R, long side int(R * long / short); then CenterCrop(R)
        if (H <= W) { oh = R; ow = (int)((double)R * W / H); } else { ow = R; oh = (int)((double)R * H / W); }
        top = round_half_even((oh - R) / 2.0);
        left = round_half_even((ow - R) / 2.0);
    }
    ResampleTable th, tv;
    build_bicubic_table(p.cw, ow, th);
    build_bicubic_table(p.ch, oh, tv);
    std::vector<int> hp, vp;
    pack_table(th, left, R, hp);
    pack_table(tv, top, R, vp);
    p.kmax_h = th.kmax; p.kmax_v = tv.kmax;
    // rows of input needed by a tile of TR output rows; shrink TR until the LDS carve fits the budget
    const int row_bytes = p.cw * 3;
    for (p.TR = tr_start; p.TR >= 1; p.TR >>= 1) {
        int need = 0;
        for (int o0 = 0; o0 < R; o0 += p.TR) {
            const int o1 = std::min(o0 + p.TR, R) - 1;
            const int lo = vp[(size_t)o0 * (1 + tv.kmax)] & 0xffff;
            const int e = vp[(size_t)o1 * (1 + tv.kmax)];
            need = std::max(need, (e & 0xffff) + (e >> 16) - lo);
        }
        p.max_rows = need;
        // (the fast instance also stages the tile's vertical taps and over-reads up to 27 bytes past a window)
        p.lds_bytes = (size_t)((R * (1 + th.kmax) * 4 + 15) & ~15) + 768 * 4 + (size_t)((p.TR * (1 + tv.kmax) * 4 + 15) & ~15) +
                      (size_t)((need * row_bytes + 15) & ~15) + (size_t)need * R * 3 + 64;
        if (p.lds_bytes <= (size_t)(getenv("ARP_PRE_LDS_KB") ? atoi(getenv("ARP_PRE_LDS_KB")) : 52) * 1024) break;  // three workgroups per CU (measured: 52 KiB 0.30 ms, 78 KiB 0.38 ms per 1024 frames)
    }
    if (p.TR < 1) return fail("preprocess: frame too wide for the LDS tile");
    ARP_TRY(p.h_tab.ensure(hp.size() * 4));
    ARP_TRY(p.v_tab.ensure(vp.size() * 4));
    ARP_HIP_OK(hipMemcpy(p.h_tab.p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    ARP_HIP_OK(hipMemcpy(p.v_tab.p, vp.data(), vp.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

static void build_lut(float* lut /* [3][256] */) {
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};  // label_reward.py:117
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float x = (float)v / 255.0f;  // ToTensor: uint8 -> f32, div 255
            volatile float y = x - mean[c];        // Normalize: (x - mean) / std, f32
            lut[c * 256 + v] = y / stdv[c];
        }
}

template <typename T, int LAYOUT>
static int launch_preprocess(const ResizePlan& p, const uint8_t* frames, int n, int P, const float* lut, void* out,
                             hipStream_t stream) {
    PreprocArgs a;
    a.frames = frames; a.out = out;
    a.h_tab = p.h_tab.as<int>(); a.v_tab = p.v_tab.as<int>(); a.lut = lut;
    a.n = n; a.H = p.H; a.W = p.W; a.cy = p.cy; a.cx = p.cx; a.ch = p.ch; a.cw = p.cw;
    a.R = p.R; a.P = P; a.kmax_h = p.kmax_h; a.kmax_v = p.kmax_v; a.TR = p.TR; a.max_rows = p.max_rows;
    // short filters + dword-aligned rows: the register-unpacking instance; anything else: the generic one
    const bool fast = p.kmax_h <= 8 && p.kmax_v <= 8 && ((p.cw * 3) & 3) == 0 && !getenv("ARP_PREPROCESS_GENERIC");
    int kt = std::max(p.kmax_h, p.kmax_v);
    if (const char* e = getenv("ARP_PRE_KT")) kt = std::max(kt, atoi(e));  // A/B: force a longer instance (8 = the round-2 kernel)
    auto kern = !fast ? preprocess_kernel<T, LAYOUT> : (kt <= 5 ? preprocess_fast_kernel<T, LAYOUT, 5> : (kt <= 6 ? preprocess_fast_kernel<T, LAYOUT, 6> : preprocess_fast_kernel<T, LAYOUT, 8>));
    ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)p.lds_bytes));
    const int tiles = (p.R + p.TR - 1) / p.TR;
    hipLaunchKernelGGL(kern, dim3(n * tiles), dim3(256), p.lds_bytes, stream, a);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp

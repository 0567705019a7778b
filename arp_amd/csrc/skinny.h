// Skinny NT GEMM for the LATENCY path: a few hundred rows at most (one frame's 50 / 197 tokens, a handful of frames, class rows).
//
// The throughput kernels (gemm.h, gemm256.h) tile the OUTPUT: at M = 50 a [50, 768] x [768, 3072]^T product is 24 tiles of
// 128 x 128 on a 256-CU chip, and a K = 3072 one is 6 workgroups that each walk 48 K-tiles one memory round trip at a time
// (37 us per c_proj launch at one frame: profiles/r3_latency_sites_before.txt).  At this size the product is neither MFMA- nor
// HBM-bound but LATENCY-bound: what matters is that every byte of W is requested at once, from as many CUs as there are.
//
// So this kernel tiles W instead: a workgroup owns 64 rows x 16 NT columns of the output and ALL of K, split over its waves; the waves'
// partial tiles are summed through LDS in a fixed order (deterministic) and the epilogue (bias, activation, f32 residual, f32 / 16-bit
// store, or a raw split-K slab) runs on the sum.  grid.x = n-groups, grid.y = cross-workgroup K split, grid.z = m-groups of 64 rows.
// Two ways of bringing the operands in (skinny.hip):
//   * the coalesced kernel (K a multiple of 64 x waves -- every shape of the towers): each wave DMAs its K range into its own double-
//     buffered LDS image, 8 rows x 128 B per instruction, and reads the MFMA fragments back; no workgroup barrier in the loop.  One CU
//     fills from L2 at 65-69 B/clk with such whole-line instructions and at 18 B/clk with a 16-row x 64-B fragment gather
//     (scripts/fill_bench.hip), and above the launch floor the operand fill IS a small GEMM's time;
//   * the gather kernel (any K % 32 == 0): fragments straight into MFMA operand layout, all loads issued up front.
// LayerNorm can be folded in (tower.h): a producer's epilogue also writes the operand-type copy of its rows and per-strip (sum, sum of
// squares); a consumer runs on W diag(gamma) and un-normalised rows and applies rstd (acc - mu c) + d in its epilogue.
// skinny_reduce_ln_kernel sums split-K slabs, adds bias and residual and applies the next LayerNorm (the unfolded form).
#pragma once
#include "common.h"

namespace arp {

struct SkinnyArgs {
    const void* A = nullptr;       // [M, lda] T (f16 / bf16)
    const void* W = nullptr;       // [N, ldw] T
    const float* bias = nullptr;   // [N] or null
    const float* resid = nullptr;  // [M, ldr] f32 or null (out_f32 only; may alias out)
    void* out = nullptr;           // [M, ldo] T or f32;  ksplit > 1: f32 slabs out + s * slice_stride, raw products
    int M = 0, N = 0, K = 0;
    int lda = 0, ldw = 0, ldr = 0, ldo = 0;
    int act = 0;       // enum Act
    int out_f32 = 0;
    int ksplit = 1;
    size_t slice_stride = 0;
    // LayerNorm folded into the latency path (tower.h): a PRODUCER (direct residual epilogue on 16-column strips) also writes the operand-type
    // copy of the new row and, per row and strip, (sum, sum of squares) of the 16 new values; a CONSUMER's A operand is that copy and its
    // epilogue is  rstd_m * (acc - mu_m * ln_c[n]) + bias[n]  with W = W diag(gamma), ln_c = its row sums, bias = W beta + b.
    void* xb = nullptr;               // producer: [M, ldxb] T
    int ldxb = 0;
    float* stats_out = nullptr;       // producer: [M][N / 16][2]
    const float* ln_stats = nullptr;  // consumer: [M][ln_parts][2]
    const float* ln_c = nullptr;      // consumer: [N]
    int ln_parts = 0;
    float ln_inv_d = 0.f, ln_eps = 0.f;
    int strips = 0;    // 1: 16-column strips even above 64 rows (A/B switch of the harness; producers of LayerNorm statistics)
    int gather = 0;    // 1: the fragment-gather kernel even where the coalesced LDS-DMA kernel applies (A/B switch of the harness)
};

constexpr int SKINNY_MAX_M = 1024;  // hard cap of the kernel's domain (slab buffers are sized for it); the owner's row limit is lower

// true when launch_skinny_gemm accepts the geometry
inline bool skinny_supported(int M, int N, int K, int lda, int ldw, int ksplit = 1) {
    return M >= 1 && M <= SKINNY_MAX_M && N >= 16 && (N & 15) == 0 && K >= 32 && ksplit >= 1 && (K % (32 * ksplit)) == 0 && (lda & 7) == 0 &&
           (ldw & 7) == 0;
}

// tcode: 1 = bf16, 2 = f16
int launch_skinny_gemm(int tcode, const SkinnyArgs& g, hipStream_t stream);

// x[r] (f32, row stride x_stride) = x[r] + bias + sum_s part[s][r][:] ;  h[r] = LayerNorm(x[r]) as T when ln_w != null.
// One wave per row; `part` slabs are [rows, D] f32 with stride slice_stride between slabs.
int launch_skinny_reduce_ln(int tcode, const float* part, int S, size_t slice_stride, const float* bias, float* x, size_t x_stride, void* h,
                            int h_stride, const float* ln_w, const float* ln_b, int rows, int D, float eps, hipStream_t stream);

}  // namespace arp

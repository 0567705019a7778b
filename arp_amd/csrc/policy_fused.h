// The 12-token policy transformer of path (2), forward AND the activation half of backward, as ONE kernel:
// one 512-thread workgroup per sample keeps the sample's tokens in LDS from token assembly to the loss and
// back down to d(image embedding).  (Reference: arp_dt/ARPDT.py:159-222,238-261 tokens, heads, losses;
// arp_dt/layers.py:11-166 Transformer/Block/Attention/FeedForward.)
//
// Why: at B = 32 samples per GPU the transformer is 384 token rows x E = 128 -- about 1 GFLOP per step spread over
// ~85 dependent launches of a few microseconds each (0.9 ms of a 2.1 ms step).  Per sample nothing couples the
// samples except the parameter gradients, so:
//   * this kernel does everything per-sample and SAVES, for every linear layer, its input X and its output
//     gradient dY (and per-row LayerNorm scale/bias contributions);
//   * the parameter gradients dW = dY^T X, db = colsum(dY) are then produced for all layers at once by two
//     grouped launches (grouped_small_gemm_kernel / grouped_colsum_kernel): fixed-order, no atomics.
//
// Matmuls run on v_mfma_f32_16x16x4_f32 (true fp32) with the 16-row token block as one MFMA operand and the weight
// rows streamed from L2 as the other, in both the NT (forward, W[out][in] rows along the contraction) and NN
// (backward, dX = dY . W) forms; attention (<= 16 tokens, head_dim a multiple of 16) is MFMA too, one wave per head.
// Token rows beyond L are zero padding: rows never mix in a linear layer or LayerNorm, and attention only visits keys
// j <= i < L.  The kernel is instantiated per (E, H) so that every address inside a pipelined group is base + immediate
// (the runtime-shape version spilled to scratch); other geometries run the one-kernel-per-op path in arp_dt.hip.
//
// Measured (B = 32, E = 128, depth 2, s_memtime stamps per phase): 378 k cycles ~ 155 us, against ~910 us for the
// per-op path.  The eight big linears take 245 k of it; with the weight loads removed they take 145 k (the f32-MFMA
// floor at 32 cycles per 16x16x4 issue), so weight streaming still costs ~70 % on top of the MFMAs.
// Round 3: the weights are streamed from fragment-major copies (PfPackJob below): 154.6 -> 136.3 us per launch.
#pragma once
#include <utility>

#include "common.h"
#include "dtops.h"

namespace arp {

constexpr int PF_THREADS = 512, PF_NW = 8, PF_MAX_DEPTH = 4;

// The big linears stream their weights in FRAGMENT-MAJOR copies (pf_pack_kernel, once per launch of the fused kernel): block (tile, step) =
// 64 lanes x 16 B in exactly the order the lanes consume it, so one wave instruction reads 1 KiB of whole 128-byte lines.  Read in place
// from W[out][in] the same fragment is 16 rows x 64 B per instruction -- the gather that one CU pulls out of L2 at 18 B/clk where
// whole-line instructions get 65-69 (scripts/fill_bench.hip); the weight stream was 100 k of the kernel's 378 k cycles.
//   nt copy: block (t, s) of W[N][K], KS = K / 16 steps per 16-row tile t: lane (q, j) holds W[16 t + j][16 s + 4 q .. + 3]
//   nn copy: block (c, s), NS = N / 16 steps per 16-column tile c:          lane (q, j) holds W[16 s + 4 q + r][16 c + j], r = 0..3
// x3 copies (PfPackJob::x3, the 16-bit modes of the step): the same blocks for v_mfma_f32_16x16x32_f16 on (hi, lo) binary16 pairs -- block (t, s) is
// one 32-deep contraction step = 2 KiB: lane (q, j) holds the eight values k = 32 s + 8 q .. + 7 of its row (nt) / column (nn), times 2^8, as
// hi = f16(w') at byte 16 lane and lo = f16(w' - hi) at byte 1024 + 16 lane.  Same bytes as the f32 copy, whole 128-byte lines per instruction.
struct PfPackJob {
    const float* W;
    float *nt, *nn;  // nn may be null (forward-only weights)
    int N, K, x3;
};
struct PfBlk {
    const float *ln0w, *ln0b, *wqkv, *bqkv, *wo, *bo, *ln1w, *ln1b, *wfc1, *wfc2;  // weights, device layout [out, in]
    const float *wqkv_nt, *wo_nt, *wfc1_nt, *wfc2_nt, *wqkv_nn, *wo_nn, *wfc1_nn, *wfc2_nn;  // fragment-major copies
    float *x, *ln0, *qkv, *att, *hmid, *ln1, *u, *gl;                              // saved forward activations [B*L, .]
    float *d_x1, *d_u, *d_mid, *d_qkv, *dws0, *dbs0, *dws1, *dbs1;                 // saved output gradients / LN row terms
};
struct PfArgs {
    int T, L, E, H, heads, NA, depth, do_bwd, R;
    float lambda;
    float alibi[16];         // per-head slope of config.alibi_bias (layers.py:74-78), zeros when off: score += slope * key index
    const float *img, *rtg;  // [R, E] tanh'd image embedding, [R]
    const int* action;       // [R]
    const float *Wr, *emb;   // rtg_input/kernel [E], action_input/embedding [NA, E]
    PfBlk blk[PF_MAX_DEPTH];
    const float *lnfw, *lnfb, *wa0, *ba0, *wa2, *wr0, *br0, *wr2;
    const float *wa0_nt, *wr0_nt, *wa0_nn, *wr0_nn;  // fragment-major copies
    float *xf, *a_in, *r_in, *ha, *hr, *logits, *ret;           // saved forward (head stage)
    float *dlogits, *dret, *dha, *dhr, *dwsf, *dbsf, *dtok, *dz;  // saved backward
    float* loss_part;                                           // [B][4]: sum CE, hits, sum squared error
    void* dzb;       // optional: dz * dz_scale in the operand type (dzb_f16: binary16, saturated; else bfloat16) -- what transpose_mask<float, float, T> made of dz
    float dz_scale;  // in a launch of its own
    int dzb_f16;
};

inline size_t pf_lds_bytes(int E, int H, int heads, int depth) {
    const int WB = H > 3 * E ? H : 3 * E;
    return ((size_t)4 * 16 * (E + 4) + (size_t)3 * 16 * (WB + 4) + (size_t)2 * heads * 256 + 64 + (size_t)(8 * depth + 5) * E) * 4;  // (the NN form may read up to 127 finite floats past a row: always inside the allocation)
}

// ---- in-kernel building blocks ---------------------------------------------------------------------------
// Both linear forms stream their weights through a ring of R register slots, one slot per 16-wide contraction step:
// a slot is refilled (with the same step of the NEXT group) right after its 4 MFMAs, so about R loads -- 8 KB per wave,
// 64 KB per workgroup -- are in flight while the MFMAs run.  The loops are branch-free in their vector-memory traffic:
// the refill is unconditional (the last group is peeled instead of guarded), epilogues write LDS only and the activations
// are saved to global memory by separate passes.  Any conditional load or store inside such a loop makes hipcc's
// s_waitcnt vmcnt counts conservative, which silently serialises load and compute.  Shapes are template parameters so
// every address inside a group is base + immediate.
//
// out[i][n] = sum_k Xs[i][k] * W[n][k]   (NT).  Each wave owns the 16-column tiles n0 = 16*(wave + 8*t).
// epi(i, n, acc): this lane's token row i and 4 consecutive columns n..n+3.
// hipcc's own s_waitcnt insertion waits vmcnt(0) at the head of such a loop (every refill of the previous group), which
// serialises load and compute; so the weight loads are inline asm (invisible to that pass) and the waits are explicit,
// counted ones -- loads return in issue order, and nothing else touches vector memory inside the loop.
template <int OFF> __device__ __forceinline__ void pf_gload4(f32x4_v& d, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pf_gload1(float& d, const float* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void pf_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The weights are packed times 2^8: their lo halves (2^-12 of the value) stay binary16 normals down to |w| = 2^-10, and anything smaller is off by less
// than 2^-33 absolute; |w| >= 256 would overflow the hi half (a LayerNorm'd 128-wide transformer's weights are O(1)).
constexpr float PF_W_SCALE = 256.f;
// (hi, lo) of eight consecutive operands: x * s = hi + lo up to 2^-22 relative; s is a power of two (so x * s is exact and it does not matter whether
// the compiler rounds it to f32 before a use or fuses it into one -- see common.h::pin_f32 for what happens when it is not)
__device__ __forceinline__ void pf_split8(const float (&x)[8], float s, f16x8_v& hi, f16x8_v& lo) {
    // common.h::split2_f16: scale, hi and lo as four v_fma_mix instructions per pair (16 per step; convert / convert back / subtract / convert took 24)
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_f16(x[2 * e], x[2 * e + 1], s, h[e], l[e]);
    hi = __builtin_bit_cast(f16x8_v, u32x4_v{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(f16x8_v, u32x4_v{l[0], l[1], l[2], l[3]});
}

// block bx of gx of job by (a stand-alone launch's (blockIdx.x, gridDim.x, blockIdx.y), or a slice of a merged launch's grid: arp_dt.hip dt_prologue_kernel)
__device__ __forceinline__ void pf_pack_block(const PfPackJob* __restrict__ jobs, int bx, int gx, int by) {
    const PfPackJob jb = jobs[by];
    if (jb.x3) {
        const int total8 = jb.N * jb.K / 8, KS = jb.K / 32, NS = jb.N / 32;
        for (int idx = bx * 256 + threadIdx.x; idx < total8; idx += gx * 256) {
            const int lane = idx & 63, blk = idx >> 6, q = lane >> 4, j = lane & 15;
            float v[8];
            f16x8_v hi, lo;
            {
                const int t = blk / KS, st = blk - t * KS;
                const float* p = jb.W + (size_t)(t * 16 + j) * jb.K + st * 32 + 8 * q;
                load4(p, *reinterpret_cast<float(*)[4]>(v));
                load4(p + 4, *reinterpret_cast<float(*)[4]>(v + 4));
                pf_split8(v, PF_W_SCALE, hi, lo);
                char* o = reinterpret_cast<char*>(jb.nt) + (size_t)blk * 2048 + lane * 16;
                *reinterpret_cast<f16x8_v*>(o) = hi;
                *reinterpret_cast<f16x8_v*>(o + 1024) = lo;
            }
            if (jb.nn) {
                const int c = blk / NS, st = blk - c * NS;
                const float* p = jb.W + (size_t)(st * 32 + 8 * q) * jb.K + c * 16 + j;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = p[(size_t)e * jb.K];
                pf_split8(v, PF_W_SCALE, hi, lo);
                char* o = reinterpret_cast<char*>(jb.nn) + (size_t)blk * 2048 + lane * 16;
                *reinterpret_cast<f16x8_v*>(o) = hi;
                *reinterpret_cast<f16x8_v*>(o + 1024) = lo;
            }
        }
        return;
    }
    const int total4 = jb.N * jb.K / 4, KS = jb.K / 16, NS = jb.N / 16;
    for (int idx = bx * 256 + threadIdx.x; idx < total4; idx += gx * 256) {
        const int lane = idx & 63, blk = idx >> 6, q = lane >> 4, j = lane & 15;
        {
            const int t = blk / KS, st = blk - t * KS;
            reinterpret_cast<float4*>(jb.nt)[idx] = *reinterpret_cast<const float4*>(jb.W + (size_t)(t * 16 + j) * jb.K + st * 16 + 4 * q);
        }
        if (jb.nn) {
            const int c = blk / NS, st = blk - c * NS;
            const float* p = jb.W + (size_t)(st * 16 + 4 * q) * jb.K + c * 16 + j;
            reinterpret_cast<float4*>(jb.nn)[idx] = make_float4(p[0], p[jb.K], p[2 * (size_t)jb.K], p[3 * (size_t)jb.K]);
        }
    }
}
static __global__ __launch_bounds__(256) void pf_pack_kernel(const PfPackJob* __restrict__ jobs) { pf_pack_block(jobs, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y); }

// The same linear on 16-bit MFMA (the 16-bit modes of the step): out[i][tile 16 t + r] = sum_k Xs[i][k] * Wt[r][k] with BOTH operands as (hi, lo)
// binary16 pairs and the product as hi.hi + lo.hi + hi.lo -- three v_mfma_f32_16x16x32_f16 per 32-deep step (48 matrix-pipe cycles) where the f32
// form takes eight 16x16x4 (256), at 2^-22 relative per product instead of 2^-24: f32-level, nowhere near what one binary16 rounding costs (2^-11).
// W = the x3 fragment-major copy (nt copy: Wt = W rows; nn copy: Wt = W columns; the two forms differ only there).  The token operand is split in
// registers from the f32 LDS tile, eight values per lane and step (24 VALU instructions: this, not the matrix pipe, is what the loop is bound by),
// after a per-linear power-of-two scale s that puts the tile's largest magnitude at 2^14..2^15: backward operands are 1e-3 .. 1e-8 and would sit in
// binary16's subnormals unscaled.  Every wave reads the whole 16 x KC tile through its own fragments, so each wave finds the same s on its own: no
// barrier, no LDS scratch.  What falls below 2^-14 of the maximum loses relative precision but not absolute (block floating point): its error is
// below the f32 rounding of the largest terms of the same sum.  A wave with several output tiles and a short contraction (qkv, fc1, d fc2: KC = E)
// splits its fragments once and keeps them (8 registers per step).
template <int NTL, int KC, int LD, class Epi>
__device__ __forceinline__ void pf_lin_x3(const float* Xs, const float* __restrict__ W, int wave, int lane, Epi epi) {
    static_assert(KC % 32 == 0, "32-deep contraction steps");
    constexpr int KS = KC / 32, R = (KS % 8 == 0) ? 8 : (KS % 4 == 0 ? 4 : (KS % 2 == 0 ? 2 : 1)), GP = KS / R;
    constexpr bool KEEP = NTL > PF_NW && KS <= 4;  // more than one tile per wave: split once
    const int q = lane >> 4, j = lane & 15;
    const int tiles_w = wave < NTL ? (NTL - 1 - wave) / PF_NW + 1 : 0;
    const int NG = tiles_w * GP;
    if (NG == 0) return;
    const float* pl = W + ((size_t)wave * KS * 128 + lane) * 4;  // 512 floats per step: hi at 16 B x lane, lo 1 KiB further
    const float* xs = Xs + j * LD + 8 * q;
    f32x4_v wh[R], wl[R];
    f32x4_v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    int tile = wave, lg = 0, cg = 0;
    auto next_ptr = [&]() {
        if (GP > 1 && ++lg < GP) pl += R * 512;
        else { lg = 0; pl += ((size_t)PF_NW * KS - (GP - 1) * R) * 512; }
    };
    auto load_slot = [&]<int U>(std::integral_constant<int, U>) {
        pf_gload4<((2 * U) & 3) * 1024>(wh[U], pl + ((2 * U) >> 2) * 1024);
        pf_gload4<((2 * U + 1) & 3) * 1024>(wl[U], pl + ((2 * U + 1) >> 2) * 1024);
    };
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... U>(std::integer_sequence<int, U...>) { (load_slot(std::integral_constant<int, U>{}), ...); }(std::make_integer_sequence<int, R>{});
    __builtin_amdgcn_sched_barrier(0);
    next_ptr();
    // the tile's scale, while the first weights are on their way
    float amax = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        const float4 a = *reinterpret_cast<const float4*>(xs + 32 * st), b = *reinterpret_cast<const float4*>(xs + 32 * st + 4);
        // (fmaxf canonicalises each operand first: two instructions per element where v_max3_f32 with |.| source modifiers takes half of one)
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(a.x), "v"(a.y));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(a.z), "v"(a.w));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(b.x), "v"(b.y));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(b.z), "v"(b.w));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    // s = 2^(14 - floor(log2 amax)), clamped to a finite normal power of two; an all-zero (or non-finite) tile takes s = 1
    const int eb = (int)((__float_as_uint(amax) >> 23) & 0xff);
    const int sb = (eb == 0 || eb == 255) ? 127 : min(max(268 - eb, 1), 254);
    const float sc = __uint_as_float((unsigned)sb << 23);
    const float inv = __uint_as_float((unsigned)(254 - sb) << 23) * (1.0f / PF_W_SCALE);
    auto split = [&](const float* xp, f16x8_v& xh, f16x8_v& xl) {
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(xp);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(xp + 4);
        pf_split8(v, sc, xh, xl);
    };
    f16x8_v kh[KEEP ? KS : 1], kl[KEEP ? KS : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int st = 0; st < KS; ++st) split(xs + 32 * st, kh[st], kl[st]);
    }
    // the two cross terms go to a second accumulator: two independent MFMA chains, and the small terms are summed among themselves first
    auto step = [&](const f32x4_v& h, const f32x4_v& l, int st) {
        f16x8_v xh, xl;
        if constexpr (KEEP) { xh = kh[st]; xl = kl[st]; }
        else split(xs + 32 * st, xh, xl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v, h), xh, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v, l), xh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v, h), xl, acc2, 0, 0, 0);
    };
    auto result = [&]() { return (acc + acc2) * inv; };
    for (int gg = 0; gg + 1 < NG; ++gg) {
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<2 * R - 2>(), step(wh[U], wl[U], cg * R + U), __builtin_amdgcn_sched_barrier(0), load_slot(std::integral_constant<int, U>{}),
              __builtin_amdgcn_sched_barrier(0)),
             ...);
        }(std::make_integer_sequence<int, R>{});
        next_ptr();
        if (++cg == GP) {
            cg = 0;
            epi(j, tile * 16 + 4 * q, result());
            acc = f32x4_v{0.f, 0.f, 0.f, 0.f};
            acc2 = f32x4_v{0.f, 0.f, 0.f, 0.f};
            tile += PF_NW;
        }
    }
    {
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<2 * (R - 1 - U)>(), step(wh[U], wl[U], cg * R + U)), ...);
        }(std::make_integer_sequence<int, R>{});
        epi(j, tile * 16 + 4 * q, result());  // the last group always closes a tile
    }
}

template <int N, int K, int LDX, bool X3 = false, class Epi>
__device__ __forceinline__ void pf_lin_nt(const float* Xs, const float* __restrict__ W, int wave, int lane, Epi epi) {
    static_assert(N % 16 == 0 && K % 16 == 0, "tile multiples");
    if constexpr (X3) return pf_lin_x3<N / 16, K, LDX>(Xs, W, wave, lane, epi);
    constexpr int KS = K / 16, R = (KS % 8 == 0) ? 8 : (KS % 4 == 0 ? 4 : (KS % 2 == 0 ? 2 : 1)), GP = KS / R, NTL = N / 16;
    const int q = lane >> 4, j = lane & 15;
    const int tiles_w = wave < NTL ? (NTL - 1 - wave) / PF_NW + 1 : 0;
    const int NG = tiles_w * GP;
    if (NG == 0) return;
    const float* pl = W + ((size_t)wave * KS * 64 + lane) * 4;  // next group to load (W = the nt fragment-major copy: 256 floats per step)
    const float* xs = Xs + j * LDX + 4 * q;
    f32x4_v w[R];
    f32x4_v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    int tile = wave, lg = 0, cg = 0;
    auto next_ptr = [&]() {
        if (GP > 1 && ++lg < GP) pl += R * 256;
        else { lg = 0; pl += ((size_t)PF_NW * KS - (GP - 1) * R) * 256; }
    };
    auto mfma4 = [&](const f32x4_v& wv, const float* xp) {
        const float4 x = *reinterpret_cast<const float4*>(xp);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], x.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], x.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], x.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], x.w, acc2, 0, 0, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... U>(std::integer_sequence<int, U...>) { (pf_gload4<(U & 3) * 1024>(w[U], pl + (U >> 2) * 1024), ...); }(std::make_integer_sequence<int, R>{});
    __builtin_amdgcn_sched_barrier(0);
    next_ptr();
    for (int gg = 0; gg + 1 < NG; ++gg) {
        const float* xg = xs + cg * (R * 16);
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<R - 1>(), mfma4(w[U], xg + U * 16), __builtin_amdgcn_sched_barrier(0), pf_gload4<(U & 3) * 1024>(w[U], pl + (U >> 2) * 1024),
              __builtin_amdgcn_sched_barrier(0)),
             ...);
        }(std::make_integer_sequence<int, R>{});
        next_ptr();
        if (++cg == GP) {
            cg = 0;
            epi(j, tile * 16 + 4 * q, acc + acc2);
            acc = f32x4_v{0.f, 0.f, 0.f, 0.f};
            acc2 = f32x4_v{0.f, 0.f, 0.f, 0.f};
            tile += PF_NW;
        }
    }
    {
        const float* xg = xs + cg * (R * 16);
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<R - 1 - U>(), mfma4(w[U], xg + U * 16)), ...);
        }(std::make_integer_sequence<int, R>{});
        epi(j, tile * 16 + 4 * q, acc + acc2);  // the last group always closes a tile
    }
}

// out[i][c] = sum_n dYs[i][n] * W[n][c]   (NN; contraction over the N rows of W, output over its K columns).
template <int N, int K, int LDY, bool X3 = false, class Epi>
__device__ __forceinline__ void pf_lin_nn(const float* dYs, const float* __restrict__ W, int wave, int lane, Epi epi) {
    static_assert(N % 16 == 0 && K % 16 == 0, "tile multiples");
    if constexpr (X3) return pf_lin_x3<K / 16, N, LDY>(dYs, W, wave, lane, epi);
    constexpr int NS = N / 16, R = (NS % 8 == 0) ? 8 : (NS % 4 == 0 ? 4 : (NS % 2 == 0 ? 2 : 1)), GP = NS / R, KT = K / 16;
    const int q = lane >> 4, j = lane & 15;
    const int tiles_w = wave < KT ? (KT - 1 - wave) / PF_NW + 1 : 0;
    const int NG = tiles_w * GP;
    if (NG == 0) return;
    const float* pl = W + ((size_t)wave * NS * 64 + lane) * 4;  // W = the nn fragment-major copy: one float4 per lane and step
    const float* ys = dYs + j * LDY + 4 * q;
    f32x4_v w[R];  // [step][r] = contraction row 4q + r of the step, this lane's output column
    f32x4_v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    int tile = wave, lg = 0, cg = 0;
    auto next_ptr = [&]() {
        if (GP > 1 && ++lg < GP) pl += R * 256;
        else { lg = 0; pl += ((size_t)PF_NW * NS - (GP - 1) * R) * 256; }
    };
    auto mfma4 = [&](const f32x4_v& wv, const float* yp) {
        const float4 y = *reinterpret_cast<const float4*>(yp);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], y.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], y.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], y.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], y.w, acc2, 0, 0, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... U>(std::integer_sequence<int, U...>) { (pf_gload4<(U & 3) * 1024>(w[U], pl + (U >> 2) * 1024), ...); }(std::make_integer_sequence<int, R>{});
    __builtin_amdgcn_sched_barrier(0);
    next_ptr();
    for (int gg = 0; gg + 1 < NG; ++gg) {
        const float* yg = ys + cg * (R * 16);
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<R - 1>(), mfma4(w[U], yg + U * 16), __builtin_amdgcn_sched_barrier(0), pf_gload4<(U & 3) * 1024>(w[U], pl + (U >> 2) * 1024),
              __builtin_amdgcn_sched_barrier(0)),
             ...);
        }(std::make_integer_sequence<int, R>{});
        next_ptr();
        if (++cg == GP) {
            cg = 0;
            epi(j, tile * 16 + 4 * q, acc + acc2);
            acc = f32x4_v{0.f, 0.f, 0.f, 0.f};
            acc2 = f32x4_v{0.f, 0.f, 0.f, 0.f};
            tile += PF_NW;
        }
    }
    {
        const float* yg = ys + cg * (R * 16);
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((pf_wait_vm<R - 1 - U>(), mfma4(w[U], yg + U * 16)), ...);
        }(std::make_integer_sequence<int, R>{});
        epi(j, tile * 16 + 4 * q, acc + acc2);
    }
}

// The head layers with N <= 16 outputs (NT: the logits / the return prediction) or N <= 16 contraction rows (NN: their
// input gradients): one MFMA tile, every load issued up front.  Rows >= N are clamped (finite): the NT epilogue must
// skip columns >= N, and the NN operand dYs must hold zeros in columns N..15.
template <int K, int LDX, class Epi>
__device__ __forceinline__ void pf_lin_nt_small(const float* Xs, const float* __restrict__ W, int N, int wave, int lane, Epi epi) {
    if (wave != 0) return;
    constexpr int KS = K / 16;
    const int q = lane >> 4, j = lane & 15;
    const float* p = W + (size_t)(j < N ? j : N - 1) * K + 4 * q;
    float4 w[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) w[u] = *reinterpret_cast<const float4*>(p + u * 16);
    f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        const float4 x = *reinterpret_cast<const float4*>(Xs + j * LDX + u * 16 + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].w, x.w, acc, 0, 0, 0);
    }
    epi(j, 4 * q, acc);
}
template <int K, int LDY, class Epi>
__device__ __forceinline__ void pf_lin_nn_small(const float* dYs, const float* __restrict__ W, int N, int wave, int lane, Epi epi) {
    const int q = lane >> 4, j = lane & 15;
    for (int tile = wave; tile < K / 16; tile += PF_NW) {
        const float* p = W + tile * 16 + j;
        float wv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[r] = p[(size_t)(4 * q + r < N ? 4 * q + r : N - 1) * K];
        const float4 y = *reinterpret_cast<const float4*>(dYs + j * LDY + 4 * q);
        f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], y.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], y.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], y.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], y.w, acc, 0, 0, 0);
        epi(j, tile * 16 + 4 * q, acc);
    }
}

// rows [0, rows) of an LDS tile -> global rows (float4 runs; width % 4 == 0)
__device__ __forceinline__ void pf_save_rows(const float* src, int ld, int width, float* __restrict__ dst, int rows, int tid) {
    const int w4 = width >> 2;
    for (int idx = tid; idx < rows * w4; idx += PF_THREADS) {
        const int i = idx / w4, c = (idx - i * w4) * 4;
        *reinterpret_cast<float4*>(dst + (size_t)i * width + c) = *reinterpret_cast<const float4*>(src + i * ld + c);
    }
}

// LayerNorm of the 16 LDS rows (2 per wave), Flax eps 1e-6; same arithmetic as ln_fwd_f32_kernel
__device__ __forceinline__ void pf_ln_fwd(const float* src, float* dst, int ld, const float* w, const float* b, int E,
                                          float* __restrict__ save, int rows_valid, int wave, int lane) {
    for (int i = wave; i < 16; i += PF_NW) {
        const float* xr = src + i * ld;
        float s = 0.f;
        for (int c = lane; c < E; c += 64) s += xr[c];
        const float mean = wave_sum(s) / E;
        float qq = 0.f;
        for (int c = lane; c < E; c += 64) { const float d = xr[c] - mean; qq += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(qq) / E + 1e-6f);
        for (int c = lane; c < E; c += 64) {
            const float v = (xr[c] - mean) * rstd * w[c] + b[c];
            dst[i * ld + c] = v;
            if (save && i < rows_valid) save[(size_t)i * E + c] = v;
        }
    }
}
// LayerNorm backward on LDS rows: dxs[i] (+)= d/dx ; per-row dy*xhat and dy saved for the deferred column sums.
// Same arithmetic as ln_bwd_f32_kernel.
__device__ __forceinline__ void pf_ln_bwd(const float* xs, const float* dys, float* dxs, int ld, const float* w, int E, bool accumulate,
                                          float* __restrict__ dws, float* __restrict__ dbs, int rows_valid, int wave, int lane) {
    for (int i = wave; i < 16; i += PF_NW) {
        const float* xr = xs + i * ld;
        const float* dyr = dys + i * ld;
        float s = 0.f;
        for (int c = lane; c < E; c += 64) s += xr[c];
        const float mean = wave_sum(s) / E;
        float qq = 0.f;
        for (int c = lane; c < E; c += 64) { const float d = xr[c] - mean; qq += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(qq) / E + 1e-6f);
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < E; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * w[c];
            s1 += gg;
            s2 += gg * xh;
            if (i < rows_valid) {
                dws[(size_t)i * E + c] = dyr[c] * xh;
                dbs[(size_t)i * E + c] = dyr[c];
            }
        }
        s1 = wave_sum(s1) / E;
        s2 = wave_sum(s2) / E;
        for (int c = lane; c < E; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * w[c];
            const float v = rstd * (gg - s1 - xh * s2);
            dxs[i * ld + c] = accumulate ? dxs[i * ld + c] + v : v;
        }
    }
}


// ---- causal attention of one sample on MFMA (head_dim a multiple of 16, <= 16 tokens): one wave per head ------------
// S^T[key][i] = sum_d K[key][d] Q[i][d] lands as lane (q, j) <-> query i = j, keys 4q..4q+3, so the softmax is two
// cross-group shuffles and P^T is already the B operand of O^T = V^T . P^T.
__device__ __forceinline__ f32x4_v pf_attn_probs(const float* sQ, int ldW, int E, int hd, int h, int L, float scale, int q, int j, float slope = 0.f) {
    f32x4_v st = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < hd; d0 += 16) {
        const float4 kk = *reinterpret_cast<const float4*>(sQ + j * ldW + E + h * hd + d0 + 4 * q);
        const float4 qq = *reinterpret_cast<const float4*>(sQ + j * ldW + h * hd + d0 + 4 * q);
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.x, qq.x, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.y, qq.y, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.z, qq.z, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.w, qq.w, st, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = (4 * q + r) <= j && j < L;
        st[r] = ok ? st[r] * scale + slope * (float)(4 * q + r) : -INFINITY;  // + alibi bias on the key index (a constant: the backward's dS is unchanged)
        mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float e = (st[r] == -INFINITY) ? 0.f : expf(st[r] - mx);
        st[r] = e;
        sum += e;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) st[r] *= inv;
    return st;
}
__device__ __forceinline__ void pf_attn_fwd_mfma(const float* sQ, float* sA, int ldW, int ldE, int E, int hd, int heads, int L, float scale,
                                                 float* __restrict__ att_save, int wave, int lane, const float* alibi) {
    const int q = lane >> 4, j = lane & 15;
    for (int h = wave; h < heads; h += PF_NW) {
        const f32x4_v p = pf_attn_probs(sQ, ldW, E, hd, h, L, scale, q, j, alibi[h & 15]);
        for (int d0 = 0; d0 < hd; d0 += 16) {
            f32x4_v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[(4 * q + s) * ldW + 2 * E + h * hd + d0 + j], p[s], o, 0, 0, 0);
            const int c = h * hd + d0 + 4 * q;
            *reinterpret_cast<float4*>(sA + j * ldE + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (j < L) *reinterpret_cast<float4*>(att_save + (size_t)j * E + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
// pass a: P and dS (to LDS as [i][key]) and dQ;  pass b (after a barrier): dK and dV from the LDS copies
__device__ __forceinline__ void pf_attn_bwd_mfma_a(const float* sQ, const float* sA, float* sU, float* sP, float* sS, int ldW, int ldE, int E, int hd,
                                                   int heads, int L, float scale, float* __restrict__ dqkv_save, int wave, int lane, const float* alibi) {
    const int q = lane >> 4, j = lane & 15;
    for (int h = wave; h < heads; h += PF_NW) {
        const f32x4_v p = pf_attn_probs(sQ, ldW, E, hd, h, L, scale, q, j, alibi[h & 15]);
        f32x4_v dp = {0.f, 0.f, 0.f, 0.f};  // dP^T[key][i] = sum_d V[key][d] dO[i][d]
        for (int d0 = 0; d0 < hd; d0 += 16) {
            const float4 vv = *reinterpret_cast<const float4*>(sQ + j * ldW + 2 * E + h * hd + d0 + 4 * q);
            const float4 gg = *reinterpret_cast<const float4*>(sA + j * ldE + h * hd + d0 + 4 * q);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.x, gg.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.y, gg.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.z, gg.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.w, gg.w, dp, 0, 0, 0);
        }
        float dot = (p[0] * dp[0] + p[1] * dp[1]) + (p[2] * dp[2] + p[3] * dp[3]);
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        f32x4_v ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = p[r] * (dp[r] - dot) * scale;
        *reinterpret_cast<float4*>(sP + h * 256 + j * 16 + 4 * q) = make_float4(p[0], p[1], p[2], p[3]);
        *reinterpret_cast<float4*>(sS + h * 256 + j * 16 + 4 * q) = make_float4(ds[0], ds[1], ds[2], ds[3]);
        for (int d0 = 0; d0 < hd; d0 += 16) {  // dQ^T[d][i] = sum_key K[key][d] dS^T[key][i]
            f32x4_v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[(4 * q + s) * ldW + E + h * hd + d0 + j], ds[s], o, 0, 0, 0);
            const int c = h * hd + d0 + 4 * q;
            *reinterpret_cast<float4*>(sU + j * ldW + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (j < L) *reinterpret_cast<float4*>(dqkv_save + (size_t)j * 3 * E + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
__device__ __forceinline__ void pf_attn_bwd_mfma_b(const float* sQ, const float* sA, float* sU, const float* sP, const float* sS, int ldW, int ldE, int E,
                                                   int hd, int heads, int L, float* __restrict__ dqkv_save, int wave, int lane) {
    const int q = lane >> 4, j = lane & 15;
    for (int h = wave; h < heads; h += PF_NW) {
        for (int d0 = 0; d0 < hd; d0 += 16) {
            f32x4_v dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int i = 4 * q + s;  // contraction over queries
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[i * ldW + h * hd + d0 + j], sS[h * 256 + i * 16 + j], dk, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[i * ldE + h * hd + d0 + j], sP[h * 256 + i * 16 + j], dv, 0, 0, 0);
            }
            const int c = h * hd + d0 + 4 * q;  // lane: key j, head-dim columns c..c+3
            *reinterpret_cast<float4*>(sU + j * ldW + E + c) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            *reinterpret_cast<float4*>(sU + j * ldW + 2 * E + c) = make_float4(dv[0], dv[1], dv[2], dv[3]);
            if (j < L) {
                *reinterpret_cast<float4*>(dqkv_save + (size_t)j * 3 * E + E + c) = make_float4(dk[0], dk[1], dk[2], dk[3]);
                *reinterpret_cast<float4*>(dqkv_save + (size_t)j * 3 * E + 2 * E + c) = make_float4(dv[0], dv[1], dv[2], dv[3]);
            }
        }
    }
}

__device__ __forceinline__ float pf_gelu_grad(float r) {  // d/du of the tanh-approximate GELU (as ew_bwd_kernel)
    const float c = 0.7978845608028654f, a = 0.044715f;
    const float t = tanhf(c * (r + a * r * r * r));
    return 0.5f * (1.f + t) + 0.5f * r * (1.f - t * t) * c * (1.f + 3.f * a * r * r);
}

template <int E, int H, bool X3 = false>
static __global__ __launch_bounds__(PF_THREADS) void policy_fused_kernel(PfArgs a) {  // X3: the big linears on (hi, lo) binary16 pairs (pf_lin_x3)
    extern __shared__ __attribute__((aligned(16))) float pf_sm[];
    const int L = a.L, T = a.T, NA = a.NA, heads = a.heads;
    const int hd = E / heads;
    constexpr int WB = H > 3 * E ? H : 3 * E;
    constexpr int ldE = E + 4, ldW = WB + 4;
    float* sX = pf_sm;            // residual stream x
    float* sM = sX + 16 * ldE;    // hmid / scratch
    float* sY = sM + 16 * ldE;    // LayerNorm output (forward); gradient stream dh (backward)
    float* sA = sY + 16 * ldE;    // attention output / scratch
    float* sQ = sA + 16 * ldE;    // qkv
    float* sU = sQ + 16 * ldW;    // fc1 pre-activation / dqkv
    float* sG = sU + 16 * ldW;    // gelu output
    float* sP = sG + 16 * ldW;    // [heads][16][16] probabilities
    float* sS = sP + heads * 256; // [heads][16][16] dS
    const int total = 4 * 16 * ldE + 3 * 16 * ldW + 2 * heads * 256 + 64;
    // every bias / LayerNorm vector lives in LDS for the whole kernel: an epilogue or a LayerNorm row that loaded them
    // from global memory would wait a full memory latency per tile (and drain the weight prefetch with it)
    float* sV = pf_sm + total;  // per block [ln0w ln0b bqkv(3E) bo ln1w ln1b], then [lnfw lnfb ba0 br0 Wr]
    float* sVt = sV + 8 * a.depth * E;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const size_t t0 = (size_t)b * L, r0 = (size_t)b * T;
    const float scale = 1.0f / sqrtf((float)hd);

    for (int i = tid; i < total; i += PF_THREADS) pf_sm[i] = 0.f;
    for (int bi = 0; bi < a.depth; ++bi) {
        const PfBlk& k = a.blk[bi];
        float* v = sV + 8 * bi * E;
        for (int e = tid; e < E; e += PF_THREADS) {
            v[e] = k.ln0w[e]; v[E + e] = k.ln0b[e]; v[5 * E + e] = k.bo[e]; v[6 * E + e] = k.ln1w[e]; v[7 * E + e] = k.ln1b[e];
            v[2 * E + e] = k.bqkv[e]; v[3 * E + e] = k.bqkv[E + e]; v[4 * E + e] = k.bqkv[2 * E + e];
        }
    }
    for (int e = tid; e < E; e += PF_THREADS) {
        sVt[e] = a.lnfw[e]; sVt[E + e] = a.lnfb[e]; sVt[2 * E + e] = a.ba0[e]; sVt[3 * E + e] = a.br0[e]; sVt[4 * E + e] = a.Wr[e];
    }
    __syncthreads();
    // ---- token assembly: per time step [image, rtg, action] (ARPDT.py:159-172,278-293) ----------------------
    for (int idx = tid; idx < L * E; idx += PF_THREADS) {
        const int i = idx / E, e = idx - i * E;
        const int t = i / 3, m = i - 3 * t;
        float v;
        if (m == 0) v = a.img[(r0 + t) * E + e];
        else if (m == 1) v = a.rtg[r0 + t] * sVt[4 * E + e];
        else v = a.emb[(size_t)a.action[r0 + t] * E + e];
        sX[i * ldE + e] = v;
        a.blk[0].x[(t0 + i) * E + e] = v;
    }
    __syncthreads();

    auto store4g = [](float* p, const f32x4_v& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); };

    // ---- forward blocks ------------------------------------------------------------------------------------
    for (int bi = 0; bi < a.depth; ++bi) {
        const PfBlk& k = a.blk[bi];
        const float* vb = sV + 8 * bi * E;
        pf_ln_fwd(sX, sY, ldE, vb, vb + E, E, k.ln0 + t0 * E, L, wave, lane);
        __syncthreads();
        pf_lin_nt<3 * E, E, ldE, X3>(sY, k.wqkv_nt, wave, lane, [&](int i, int n, f32x4_v v) {
            const float4 bb = *reinterpret_cast<const float4*>(vb + 2 * E + n);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            *reinterpret_cast<float4*>(sQ + i * ldW + n) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_save_rows(sQ, ldW, 3 * E, k.qkv + t0 * 3 * E, L, tid);
        // causal attention (layers.py:70-90): scores * scale, masked, softmax, P.V
        pf_attn_fwd_mfma(sQ, sA, ldW, ldE, E, hd, heads, L, scale, k.att + t0 * E, wave, lane, a.alibi);
        __syncthreads();
        pf_lin_nt<E, E, ldE, X3>(sA, k.wo_nt, wave, lane, [&](int i, int n, f32x4_v v) {
            const float4 bb = *reinterpret_cast<const float4*>(vb + 5 * E + n);
            const float4 xr = *reinterpret_cast<const float4*>(sX + i * ldE + n);
            v[0] += bb.x + xr.x; v[1] += bb.y + xr.y; v[2] += bb.z + xr.z; v[3] += bb.w + xr.w;
            *reinterpret_cast<float4*>(sM + i * ldE + n) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_save_rows(sM, ldE, E, k.hmid + t0 * E, L, tid);
        pf_ln_fwd(sM, sY, ldE, vb + 6 * E, vb + 7 * E, E, k.ln1 + t0 * E, L, wave, lane);
        __syncthreads();
        pf_lin_nt<H, E, ldE, X3>(sY, k.wfc1_nt, wave, lane, [&](int i, int n, f32x4_v v) {
            f32x4_v gl;
#pragma unroll
            for (int r = 0; r < 4; ++r) gl[r] = apply_act<ACT_GELU_TANH>(v[r]);
            *reinterpret_cast<float4*>(sG + i * ldW + n) = make_float4(gl[0], gl[1], gl[2], gl[3]);
            *reinterpret_cast<float4*>(sU + i * ldW + n) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_save_rows(sU, ldW, H, k.u + t0 * H, L, tid);
        pf_save_rows(sG, ldW, H, k.gl + t0 * H, L, tid);
        float* xnext = bi + 1 < a.depth ? a.blk[bi + 1].x : a.xf;
        pf_lin_nt<E, H, ldW, X3>(sG, k.wfc2_nt, wave, lane, [&](int i, int n, f32x4_v v) {
            const float4 mr = *reinterpret_cast<const float4*>(sM + i * ldE + n);
            v[0] += mr.x; v[1] += mr.y; v[2] += mr.z; v[3] += mr.w;
            *reinterpret_cast<float4*>(sX + i * ldE + n) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_save_rows(sX, ldE, E, xnext + t0 * E, L, tid);
    }
    // ---- final LayerNorm, head inputs (ARPDT.py:203-205: action head <- rtg tokens 1::3, return head <- image tokens 0::3)
    pf_ln_fwd(sX, sY, ldE, sVt, sVt + E, E, nullptr, 0, wave, lane);
    float* hA = sU;               // a_in  [16][ldE]
    float* hHa = sU + 16 * ldE;   // relu(layers_0(a_in))
    float* hR = sG;               // r_in
    float* hHr = sG + 16 * ldE;   // relu(layers_0(r_in))
    __syncthreads();
    for (int idx = tid; idx < 16 * E; idx += PF_THREADS) {
        const int t = idx / E, c = idx - t * E;
        float va = 0.f, vr = 0.f;
        if (t < T) {
            va = sY[(3 * t + 1) * ldE + c];
            vr = sY[(3 * t) * ldE + c];
            a.a_in[(r0 + t) * E + c] = va;
            a.r_in[(r0 + t) * E + c] = vr;
        }
        hA[t * ldE + c] = va;
        hR[t * ldE + c] = vr;
    }
    __syncthreads();
    pf_lin_nt<E, E, ldE, X3>(hA, a.wa0_nt, wave, lane, [&](int i, int n, f32x4_v v) {
        const float4 bb = *reinterpret_cast<const float4*>(sVt + 2 * E + n);
        v[0] = fmaxf(v[0] + bb.x, 0.f); v[1] = fmaxf(v[1] + bb.y, 0.f); v[2] = fmaxf(v[2] + bb.z, 0.f); v[3] = fmaxf(v[3] + bb.w, 0.f);
        if (i >= T) v = f32x4_v{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(hHa + i * ldE + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (i < T) store4g(a.ha + (r0 + i) * E + n, v);
    });
    pf_lin_nt<E, E, ldE, X3>(hR, a.wr0_nt, wave, lane, [&](int i, int n, f32x4_v v) {
        const float4 bb = *reinterpret_cast<const float4*>(sVt + 3 * E + n);
        v[0] = fmaxf(v[0] + bb.x, 0.f); v[1] = fmaxf(v[1] + bb.y, 0.f); v[2] = fmaxf(v[2] + bb.z, 0.f); v[3] = fmaxf(v[3] + bb.w, 0.f);
        if (i >= T) v = f32x4_v{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(hHr + i * ldE + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (i < T) store4g(a.hr + (r0 + i) * E + n, v);
    });
    __syncthreads();
    // logits -> sQ[i][0..NA), return prediction -> sQ[i][16]
    pf_lin_nt_small<E, ldE>(hHa, a.wa2, NA, wave, lane, [&](int i, int n, f32x4_v v) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < NA) {
                sQ[i * ldW + n + r] = v[r];
                if (i < T) a.logits[(r0 + i) * NA + n + r] = v[r];
            }
    });
    pf_lin_nt_small<E, ldE>(hHr, a.wr2, 1, wave, lane, [&](int i, int n, f32x4_v v) {
        if (n == 0) {
            sQ[i * ldW + 16] = v[0];
            if (i < T) a.ret[r0 + i] = v[0];
        }
    });
    // sA <- dlogits, sM <- dret (both zero elsewhere)
    for (int i = tid; i < 16 * ldE; i += PF_THREADS) { sA[i] = 0.f; sM[i] = 0.f; }
    __syncthreads();
    // ---- losses (ARPDT.py:238-261): CE summed over all B*T*NA elements / (B*T*NA); MSE / (B*T) -------------
    if (tid < T) {
        const int t = tid;
        const float* l = sQ + t * ldW;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < NA; ++c)
            if (l[c] > mx) { mx = l[c]; am = c; }
        float sum = 0.f;
        for (int c = 0; c < NA; ++c) sum += expf(l[c] - mx);
        const float lse = logf(sum) + mx;
        const int lab = a.action[r0 + t];
        for (int c = 0; c < NA; ++c) {
            const float d = (expf(l[c] - lse) - (c == lab ? 1.f : 0.f)) / ((float)a.R * NA);
            sA[t * ldE + c] = d;
            a.dlogits[(r0 + t) * NA + c] = d;
        }
        const float dr = l[16] - a.rtg[r0 + t];
        const float dd = a.lambda * 2.f * dr / (float)a.R;
        sM[t * ldE] = dd;
        a.dret[r0 + t] = dd;
        // per-sample partials, combined over t in a fixed order by lane 0 below
        sS[t * 4 + 0] = lse - l[lab];
        sS[t * 4 + 1] = (am == lab) ? 1.f : 0.f;
        sS[t * 4 + 2] = dr * dr;
    }
    __syncthreads();
    if (tid == 0) {
        float ce = 0.f, hit = 0.f, se = 0.f;
        for (int t = 0; t < T; ++t) { ce += sS[t * 4]; hit += sS[t * 4 + 1]; se += sS[t * 4 + 2]; }
        a.loss_part[b * 4 + 0] = ce;
        a.loss_part[b * 4 + 1] = hit;
        a.loss_part[b * 4 + 2] = se;
        a.loss_part[b * 4 + 3] = 0.f;
    }
    if (!a.do_bwd) return;

    // ======================================= backward (activations) =========================================
    float* hDha = sQ;              // [16][ldE]
    float* hDhr = sQ + 16 * ldE;
    __syncthreads();
    pf_lin_nn_small<E, ldE>(sA, a.wa2, NA, wave, lane, [&](int i, int c, f32x4_v v) {
        const float4 hh = *reinterpret_cast<const float4*>(hHa + i * ldE + c);
        v[0] = hh.x > 0.f ? v[0] : 0.f; v[1] = hh.y > 0.f ? v[1] : 0.f; v[2] = hh.z > 0.f ? v[2] : 0.f; v[3] = hh.w > 0.f ? v[3] : 0.f;
        *reinterpret_cast<float4*>(hDha + i * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
        if (i < T) store4g(a.dha + (r0 + i) * E + c, v);
    });
    pf_lin_nn_small<E, ldE>(sM, a.wr2, 1, wave, lane, [&](int i, int c, f32x4_v v) {
        const float4 hh = *reinterpret_cast<const float4*>(hHr + i * ldE + c);
        v[0] = hh.x > 0.f ? v[0] : 0.f; v[1] = hh.y > 0.f ? v[1] : 0.f; v[2] = hh.z > 0.f ? v[2] : 0.f; v[3] = hh.w > 0.f ? v[3] : 0.f;
        *reinterpret_cast<float4*>(hDhr + i * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
        if (i < T) store4g(a.dhr + (r0 + i) * E + c, v);
    });
    __syncthreads();
    for (int i = tid; i < 16 * ldE; i += PF_THREADS) sA[i] = 0.f;  // sA <- d(hf): rows 3t+1 <- d a_in, rows 3t <- d r_in
    __syncthreads();
    pf_lin_nn<E, E, ldE, X3>(hDha, a.wa0_nn, wave, lane, [&](int i, int c, f32x4_v v) {
        if (i < T) *reinterpret_cast<float4*>(sA + (3 * i + 1) * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
    });
    pf_lin_nn<E, E, ldE, X3>(hDhr, a.wr0_nn, wave, lane, [&](int i, int c, f32x4_v v) {
        if (i < T) *reinterpret_cast<float4*>(sA + (3 * i) * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
    });
    __syncthreads();
    pf_ln_bwd(sX, sA, sY, ldE, sVt, E, false, a.dwsf + t0 * E, a.dbsf + t0 * E, L, wave, lane);  // sY = dh
    __syncthreads();

    for (int bi = a.depth - 1; bi >= 0; --bi) {
        const PfBlk& k = a.blk[bi];
        // x_{i+1} = hmid + gelu(ln1 Wfc1) Wfc2 : dY of fc2 is dh itself
        for (int idx = tid; idx < L * E; idx += PF_THREADS) {
            const int i = idx / E, c = idx - i * E;
            k.d_x1[(t0 + i) * E + c] = sY[i * ldE + c];
            sM[i * ldE + c] = k.hmid[(t0 + i) * E + c];  // for the LayerNorm_1 backward below
        }
        for (int idx = tid * 4; idx < L * H; idx += PF_THREADS * 4) {  // fc1 pre-activations for gelu' (H % 4 == 0)
            const int i = idx / H, c = idx - i * H;
            *reinterpret_cast<float4*>(sG + i * ldW + c) = *reinterpret_cast<const float4*>(k.u + (t0 + i) * H + c);
        }
        __syncthreads();
        pf_lin_nn<E, H, ldE, X3>(sY, k.wfc2_nn, wave, lane, [&](int i, int c, f32x4_v v) {
            if (i < L) {
                const float4 uu = *reinterpret_cast<const float4*>(sG + i * ldW + c);
                v[0] *= pf_gelu_grad(uu.x); v[1] *= pf_gelu_grad(uu.y); v[2] *= pf_gelu_grad(uu.z); v[3] *= pf_gelu_grad(uu.w);
            } else {
                v = f32x4_v{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<float4*>(sU + i * ldW + c) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_save_rows(sU, ldW, H, k.d_u + t0 * H, L, tid);
        pf_lin_nn<H, E, ldW, X3>(sU, k.wfc1_nn, wave, lane, [&](int i, int c, f32x4_v v) {
            *reinterpret_cast<float4*>(sA + i * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_ln_bwd(sM, sA, sY, ldE, sV + 8 * bi * E + 6 * E, E, true, k.dws1 + t0 * E, k.dbs1 + t0 * E, L, wave, lane);  // sY = d hmid
        __syncthreads();
        // hmid = x_i + att Wo + bo
        for (int idx = tid; idx < L * E; idx += PF_THREADS) {
            const int i = idx / E, c = idx - i * E;
            k.d_mid[(t0 + i) * E + c] = sY[i * ldE + c];
            sM[i * ldE + c] = k.x[(t0 + i) * E + c];  // for the LayerNorm_0 backward below
        }
        for (int idx = tid; idx < L * 3 * E; idx += PF_THREADS) {
            const int i = idx / (3 * E), c = idx - i * 3 * E;
            sQ[i * ldW + c] = k.qkv[(t0 + i) * 3 * E + c];
        }
        pf_lin_nn<E, E, ldE, X3>(sY, k.wo_nn, wave, lane, [&](int i, int c, f32x4_v v) {
            *reinterpret_cast<float4*>(sA + i * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);  // d att
        });
        __syncthreads();
        // attention backward: probabilities recomputed; dS = P * (dP - sum_j P dP) * scale
        pf_attn_bwd_mfma_a(sQ, sA, sU, sP, sS, ldW, ldE, E, hd, heads, L, scale, k.d_qkv + t0 * 3 * E, wave, lane, a.alibi);
        __syncthreads();
        pf_attn_bwd_mfma_b(sQ, sA, sU, sP, sS, ldW, ldE, E, hd, heads, L, k.d_qkv + t0 * 3 * E, wave, lane);
        __syncthreads();
        pf_lin_nn<3 * E, E, ldW, X3>(sU, k.wqkv_nn, wave, lane, [&](int i, int c, f32x4_v v) {
            *reinterpret_cast<float4*>(sA + i * ldE + c) = make_float4(v[0], v[1], v[2], v[3]);
        });
        __syncthreads();
        pf_ln_bwd(sM, sA, sY, ldE, sV + 8 * bi * E, E, true, k.dws0 + t0 * E, k.dbs0 + t0 * E, L, wave, lane);  // sY = d x_i
        __syncthreads();
    }
    // d tokens; d(pre-tanh image embedding) = d img * (1 - img^2)   (ARPDT.py:484)
    for (int idx = tid; idx < L * E; idx += PF_THREADS) {
        const int i = idx / E, e = idx - i * E;
        const float g = sY[i * ldE + e];
        a.dtok[(t0 + i) * E + e] = g;
        const int t = i / 3;
        if (i - 3 * t == 0) {
            const float y = a.img[(r0 + t) * E + e];
            const float dzv = pin_f32(g * (1.f - y * y));  // (ONE f32 value for both stores: common.h)
            a.dz[(r0 + t) * E + e] = dzv;
            if (a.dzb) {
                float v = dzv * a.dz_scale;
                if (a.dzb_f16) {
                    v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                    Elem<f16_t>::st(static_cast<f16_t*>(a.dzb) + (r0 + t) * E + e, v);
                } else {
                    Elem<bf16_t>::st(static_cast<bf16_t*>(a.dzb) + (r0 + t) * E + e, v);
                }
            }
        }
    }
}

// metrics: [0] loss = trans + lambda*ret, [1] acc (fraction), [2] trans_loss, [3] return_loss  (as loss_kernel)
static __global__ void loss_finish_kernel(const float* __restrict__ part, int B, int R, int NA, float lambda, float* __restrict__ metrics) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ce = 0.f, hit = 0.f, se = 0.f;
    for (int b = 0; b < B; ++b) { ce += part[b * 4]; hit += part[b * 4 + 1]; se += part[b * 4 + 2]; }
    const float trans = ce / ((float)R * NA), rl = se / (float)R;
    metrics[0] = trans + lambda * rl;
    metrics[1] = hit / (float)R;
    metrics[2] = trans;
    metrics[3] = rl;
}

// ---- grouped launches for the deferred parameter gradients ---------------------------------------------------
// tile_prefix[p] .. tile_prefix[p+1] are the 32x32 output tiles of problem p
// One 32 x 32 tile of C[M,N] = sum_k A[k,m] B[k,n] (both operands k-major: a weight gradient dY^T X) on the f32 MFMA: each of the four
// waves owns a 16 x 16 block and feeds v_mfma_f32_16x16x4_f32 STRAIGHT from global memory -- lane l supplies A[k0 + l/16][m0 + l%16] and
// B[k0 + l/16][n0 + l%16], 64-byte runs per 16 lanes -- with 48 steps of loads in flight.  Exact f32 (the instruction is an fmaf
// chain).  The VALU tile (LDS-staged, 4 FMAs per 4 LDS reads) took 33 us for the transformer's ~420 gradient tiles, this ~10.
__device__ __forceinline__ void small_gemm_tile_tn_mfma(const SmallGemm& g, int bx, int by) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = by * 32 + (wave >> 1) * 16, n0 = bx * 32 + (wave & 1) * 16;
    if (m0 >= g.M || n0 >= g.N) return;  // M, N are multiples of 16 here
    // explicitly GLOBAL pointers: the job table is read from memory, so the compiler only knows `generic` and emits flat loads, which it
    // then waits for one MFMA at a time (vmcnt(0) lgkmcnt(0) after every pair: 96 dependent memory round trips per tile)
    typedef const __attribute__((address_space(1))) float* gptr;
    gptr a = (gptr)(g.A + (size_t)(lane >> 4) * g.lda + m0 + (lane & 15));
    gptr b = (gptr)(g.B + (size_t)(lane >> 4) * g.ldb + n0 + (lane & 15));
    f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
    const size_t sa = (size_t)4 * g.lda, sb = (size_t)4 * g.ldb;
    int k = 0;
    constexpr int UB = 48;  // steps per batch: the whole batch's loads are issued before its first MFMA (K = 384 rows = two batches)
    for (; k + 4 * UB <= g.K; k += 4 * UB) {
        float av[UB], bv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            av[u] = a[(size_t)u * sa];
            bv[u] = b[(size_t)u * sb];
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the batch go out before its first MFMA (left alone, hipcc keeps two in flight)
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a += UB * sa;
        b += UB * sb;
    }
    for (; k < g.K; k += 4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(*a, *b, acc, 0, 0, 0);
        a += sa;
        b += sb;
    }
    // lane holds rows m0 + 4 (l / 16) + j, column n0 + l % 16
#pragma unroll
    for (int j = 0; j < 4; ++j) g.C[(size_t)(m0 + 4 * (lane >> 4) + j) * g.ldc + n0 + (lane & 15)] = acc[j];
}

// which job does this block belong to: the number of job boundaries at or below blockIdx.x, found with ONE load per lane and a ballot
// (a scan `while (blockIdx.x >= tile_prefix[p + 1]) ++p` is up to n dependent memory round trips in every block; n <= 64)
__device__ __forceinline__ int grouped_job_index(const int* __restrict__ tile_prefix, int n, int bid) {
    const int lane = threadIdx.x & 63;
    const int bound = lane + 1 < n ? tile_prefix[lane + 1] : 0x7fffffff;
    return __builtin_amdgcn_readfirstlane(__popcll(__ballot(bid >= bound)));
}
__device__ __forceinline__ int grouped_job_index(const int* __restrict__ tile_prefix, int n) { return grouped_job_index(tile_prefix, n, (int)blockIdx.x); }

__device__ __forceinline__ void grouped_small_gemm_block(const SmallGemm* __restrict__ tab, const int* __restrict__ tile_prefix, int n, int bid) {
    const int p = grouped_job_index(tile_prefix, n, bid);
    const SmallGemm g = tab[p];
    const int local = bid - tile_prefix[p];
    const int nbx = (g.N + 31) / 32;
    if (g.ta && !g.tb && !g.bias && !g.resid && g.act == ACT_NONE && !g.accumulate && !((g.M | g.N) & 15) && !(g.K & 3))
        small_gemm_tile_tn_mfma(g, local % nbx, local / nbx);
    else
        small_gemm_tile(g, local % nbx, local / nbx);
}
static __global__ __launch_bounds__(256) void grouped_small_gemm_kernel(const SmallGemm* __restrict__ tab, const int* __restrict__ tile_prefix, int n) {
    grouped_small_gemm_block(tab, tile_prefix, n, (int)blockIdx.x);
}
// embedding / rtg-projection gradients from d tokens: block a < NA sums the action-token rows whose action is a, block NA the
// rtg-token rows weighted by rtg (same sums as tokens_bwd_kernel, no read-modify-write chain).  The rows are split over 1024 / E thread
// groups and every row's value is LOADED unconditionally (selected afterwards), so the loads of an unrolled batch are in flight
// together: one thread walking all R rows with a load behind each `action[r] == a` test took 31 us of the 0.9 ms step at R = 128.
// Fixed summation order (group g: rows g, g + G, ...; then groups 0..G-1).
constexpr int TOKB_THREADS = 1024;
// NT = 1024: the stand-alone launch; NT = 256: one block of pf_param_grads_kernel (G = 256 / E row groups, deeper unroll: 32 rows' loads in flight per lane)
template <int NT>
__device__ __forceinline__ void tokens_bwd_block(int a, const float* __restrict__ dtok, const float* __restrict__ rtg, const int* __restrict__ action,
                                                 float* __restrict__ dWr, float* __restrict__ demb, int R, int E, int n_actions) {
    // The summation order is that of TOKB_THREADS threads whatever NT is: G = TOKB_THREADS / E row groups (group g: rows g, g + G, ...), then groups 0..G-1.
    // A smaller block gives each thread TOKB_THREADS / NT of the groups (their loads all independent), so both launch forms produce the same bits.
    __shared__ float red[TOKB_THREADS];
    const bool is_act = a < n_actions;
    const int tok = is_act ? 2 : 1;
    if (E <= NT && NT % E == 0) {
        constexpr int PER = TOKB_THREADS / NT;
        const int G = TOKB_THREADS / E, grp0 = (threadIdx.x / E) * PER, e = threadIdx.x % E;
        float s[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) s[q] = 0.f;
        // The block-uniform `is_act` is decided ONCE, outside the loops, and every load is unconditional: written as `is_act ? (action[r] == a ? x : 0) : rtg[r] * x`
        // per element, each of the two second loads sat in its own branch behind the first and was waited for alone -- ~130 dependent round trips per thread in the
        // 256-thread form (found in the ISA: it was what made the merged gradient launch 26 us instead of the 16 of its MFMA tiles).  Same sums, same order.
        auto rows = [&](auto ACT_) __attribute__((always_inline)) {
            constexpr int UB = 8;  // row steps per batch: 2 * UB * PER loads requested before the first add (hipcc alone keeps one step's in flight)
            for (int r0 = 0; r0 < R; r0 += G * UB) {
                float xv[UB][PER], wv[UB][PER];
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int rc = min(r0 + u * G + grp0 + q, R - 1);
                        xv[u][q] = dtok[((size_t)rc * 3 + tok) * E + e];
                        if constexpr (decltype(ACT_)::value) wv[u][q] = __int_as_float(action[rc]);
                        else wv[u][q] = rtg[rc];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int rr = r0 + u * G + grp0 + q;
                        float v;
                        if constexpr (decltype(ACT_)::value) v = __float_as_int(wv[u][q]) == a ? xv[u][q] : 0.f;
                        else v = wv[u][q] * xv[u][q];
                        s[q] += rr < R ? v : 0.f;  // (R is a multiple of G in every shipped geometry: nothing is ever added for a row past the end)
                    }
            }
        };
        if (is_act) rows(std::true_type{});
        else rows(std::false_type{});
#pragma unroll
        for (int q = 0; q < PER; ++q) red[(grp0 + q) * E + e] = s[q];
        __syncthreads();
        if (grp0 == 0) {
            float t = red[e];
            for (int k = 1; k < G; ++k) t += red[k * E + e];
            if (is_act) demb[(size_t)a * E + e] = t;
            else dWr[e] = t;
        }
        return;
    }
    for (int e = threadIdx.x; e < E; e += NT) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < R; ++r) {
            const float x = dtok[((size_t)r * 3 + tok) * E + e];
            s += is_act ? (action[r] == a ? x : 0.f) : rtg[r] * x;
        }
        if (is_act) demb[(size_t)a * E + e] = s;
        else dWr[e] = s;
    }
}
static __global__ __launch_bounds__(TOKB_THREADS) void tokens_bwd_par_kernel(const float* __restrict__ dtok, const float* __restrict__ rtg, const int* __restrict__ action,
                                                             float* __restrict__ dWr, float* __restrict__ demb, int R, int E, int n_actions) {
    tokens_bwd_block<TOKB_THREADS>((int)blockIdx.x, dtok, rtg, action, dWr, demb, R, E, n_actions);
}
struct ColSumJob { const float* in; float* out; int R, C; };
static __global__ __launch_bounds__(256) void grouped_colsum_kernel(const ColSumJob* __restrict__ tab, const int* __restrict__ tile_prefix, int n) {
    const int p = grouped_job_index(tile_prefix, n);
    const ColSumJob j = tab[p];
    colsum_tile(j.in, j.R, j.C, j.out, blockIdx.x - tile_prefix[p]);
}

// Everything that turns the fused kernel's saved activation gradients into parameter gradients and metrics, in ONE launch: the grouped weight-gradient
// tiles, the grouped column sums, the embedding / rtg-projection sums and the loss reduction were four dependent launches (16 + 6 + 8 + 4.5 us and three
// boundaries on the step's critical path); none of them reads another's output, so their blocks share a grid and the launch lasts as long as its longest part.
struct PfGradsArgs {
    const SmallGemm* gtab; const int* gprefix; int n_gemm, gemm_tiles;
    const ColSumJob* ctab; const int* cprefix; int n_cs, cs_tiles;
    const float *dtok, *rtg; const int* action; float *dWr, *demb; int R, E, NA;
    const float* loss_part; int B; float lambda; float* metrics;
};
static __global__ __launch_bounds__(256) void pf_param_grads_kernel(PfGradsArgs a) {
    // the blocks with the longest dependent chains first (they are dispatched first): embedding sums, column sums, then the MFMA tiles
    int b = (int)blockIdx.x;
    if (b <= a.NA) { tokens_bwd_block<256>(b, a.dtok, a.rtg, a.action, a.dWr, a.demb, a.R, a.E, a.NA); return; }
    b -= a.NA + 1;
    if (b >= 1 && b <= a.cs_tiles) {
        b -= 1;
        const int p = grouped_job_index(a.cprefix, a.n_cs, b);
        const ColSumJob j = a.ctab[p];
        colsum_tile(j.in, j.R, j.C, j.out, b - a.cprefix[p]);
        return;
    }
    if (b > a.cs_tiles) { grouped_small_gemm_block(a.gtab, a.gprefix, a.n_gemm, b - a.cs_tiles - 1); return; }
    if (threadIdx.x == 0) {  // (b == 0) loss_finish_kernel's arithmetic
        float ce = 0.f, hit = 0.f, se = 0.f;
        for (int i = 0; i < a.B; ++i) { ce += a.loss_part[i * 4]; hit += a.loss_part[i * 4 + 1]; se += a.loss_part[i * 4 + 2]; }
        const float trans = ce / ((float)a.R * a.NA), rl = se / (float)a.R;
        a.metrics[0] = trans + a.lambda * rl;
        a.metrics[1] = hit / (float)a.R;
        a.metrics[2] = trans;
        a.metrics[3] = rl;
    }
}

}  // namespace arp

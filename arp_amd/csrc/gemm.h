// NT GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// A (activations) and W (torch-Linear weight layout, [out, in]) are both K-contiguous, so both
// LDS tiles are [rows][128 bytes] and every MFMA fragment is one ds_read_b128.
//   * T = bf16_t : v_mfma_f32_16x16x32_bf16, BK = 64 elements  (throughput mode)
//   * T = float  : v_mfma_f32_16x16x4_f32,  BK = 32 elements  (parity mode; exact f32 FMA chains)
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA fragments),
// double-buffered LDS filled by global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), XOR
// swizzle on the 16-byte chunk index (applied to the per-lane SOURCE address, because the LDS
// destination of an LDS-DMA is lane-linear) so the ds_read_b128 fragment reads are
// bank-conflict-free, and an XCD-aware block->tile map so the 32 CUs behind one L2 share panels.
//
// The MFMA is issued "swapped" (W fragment as the A operand): D[n][m], so each lane ends up with
// 4 CONSECUTIVE n for one m -> the epilogue reads bias/residual and writes the output as 8/16-byte
// vectors.
//
// Epilogue: (+bias[n]) -> activation -> (+residual[m,n], f32) -> OutT.
#pragma once
#include "common.h"

#ifndef ARP_ADAMW_PIPE
#define ARP_ADAMW_PIPE 1
#endif
namespace arp {

struct GemmArgs {
    const void* A = nullptr;      // [M, lda] T
    const void* W = nullptr;      // [N, ldw] T
    const float* bias = nullptr;  // [N] or nullptr
    const float* resid = nullptr; // [M, ldr] f32 or nullptr (may alias out when OutT == float)
    void* out = nullptr;          // [M, ldo] OutT
    int M = 0, N = 0, K = 0;
    int lda = 0, ldw = 0, ldr = 0, ldo = 0;
    int flags = 0;      // debug/ablation: bit 0 = skip the epilogue stores (timing experiments only); bit 5 (32): the caller's promise that A holds
                        // ceil(M / 256) * 256 readable rows (gemm256 MIXC (ARP_G2_MIX_UNIFORM) reads whole row tiles without a per-lane clamp; the extra rows are never stored)
    unsigned long long* clock_acc = nullptr;  // gemm256's CLK diagnostic instance only: [0] += d(s_memtime), [1] += d(s_memrealtime), [2] += 1 per workgroup
    int ksplit = 1;     // split-K: gridDim.y slices of K; slice s writes to out + s * slice_stride (bias/resid ignored by callers)
    size_t slice_stride = 0;
    // LayerNorm folding.  Consumer side (a GEMM whose A operand is the UN-normalised row x):
    //   out = rstd_m * (x.W'^T - mu_m * c) + d,  W' = W diag(gamma), c[n] = sum_k W'[n,k], d = W beta + bias (passed as bias)
    const float* ln_stats = nullptr;  // [M][ln_parts][2]: (sum, sum of squares) of row m over each 128-column segment; gemm256's folded epilogue holds at most
                                      // 8 parts per row (width <= 1024: the launcher refuses more), gemm.h's loops over any count
    const float* ln_c = nullptr;      // [N]
    int ln_parts = 0;
    float ln_inv_d = 0.f, ln_eps = 0.f;
    // Producer side (the f32 residual epilogue): also emit the operand-type copy of the new row and its partial sums
    void* xb_out = nullptr;           // [M, ldxb] T
    int ldxb = 0;
    int m_fast = 0;                   // gemm_nt_kernel: walk the tiles m-fastest, so the m-tiles of one weight column panel are neighbours on one XCD and share the
                                      // panel through its L2 (a few rows against a long weight stream: the fine-tune head's M = 192 products read W once, not twice)
    int split3 = 0;                   // f32-output epilogues, T = f16: xb_out rows are [hi | lo | hi], N wide each (store_split3: the next GEMM's operand in
                                      // ARP_MODE_F16X3); `out` may then be null
    float* stats_out = nullptr;       // [M][N/128][2]
    int group_m = 0;          // gemm256: tile-rows per L2 group (0 = default)
    int stagger_groups = 0;   // >1: the first wave of workgroups starts in `stagger_groups` phase groups spread over
    int stagger_cycles = 0;   //     `stagger_cycles` shader cycles, so CUs do not all reach their store epilogue together
    float alpha = 1.f;        // out = act(alpha * (A.W^T) + bias) (+ resid); honoured by gemm_nt_kernel (launch_gemm_auto routes alpha != 1
                              // there: power-of-two un-scaling of f16 gradient GEMMs, arp_dt.hip) and by the fp8 instances of gemm256
    float out_scale = 1.f;    // fp8 output only: the stored value is out_scale * act(...) (the consumer's alpha takes it out again)
    // gemm256, 16-bit output, staged epilogue only (a ReLU backward fused into the producing GEMM: arp_dt.hip's adapter):
    const void* mask = nullptr;       // [M, ldm] OutT: out = (mask > 0) ? value : 0
    int ldm = 0;
    float* colsum_part = nullptr;     // [ceil(M / 256)][N] f32: column sums of the ROUNDED masked output over each tile's rows
    int ovl = 0;              // gemm256, 16-bit output: a workgroup with another tile to do drains this tile's stores under that tile's first phases
    // gemm_nt_kernel, SITE == GEMM_SITE_ADAMW only (the fine-tune head's weight-gradient GEMMs, arp_ft.hip): the product IS the gradient
    // of the [M, N] weight at adam_p; instead of storing it, the epilogue applies torch.optim.AdamW to that weight in place (reads p, m,
    // v, writes p, m, v and the operand-type mirror): 26 bytes per parameter instead of 4 (store dW) + 30 (a separate AdamW pass).
    float *adam_p = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    void* adam_mirror = nullptr;
    int adam_mask = 0;                     // f16 mode: a non-finite gradient entry counts as missing (and is counted into adam_dropped)
    unsigned int* adam_dropped = nullptr;
    float adam_gscale = 1.f, adam_lr = 0.f, adam_wd = 0.f, adam_b1 = 0.f, adam_b2 = 0.f, adam_eps = 0.f, adam_bc1 = 1.f, adam_bc2 = 1.f;
    // gemm256 MIXC instances (T = f16; row N1's ARP_MODE_F16C): a row of A / W is [hi: binary16 x Kc | e2m1 x Kc (| e2m1 x Kc)], i.e. its K-tiles
    // (128 bytes each) are mix_nk16 = Kc / 64 binary16 tiles, then mix_nkc_a = Kc / 256 fp4 tiles whose products are scaled by 2^-mix_sa, then fp4 tiles
    // scaled by 2^-mix_sb: out = A_hi.W_hi^T + 2^-sa A4.dW4^T (+ 2^-sb dA4.W4^T) -- the operand-rounding corrections of a binary16 GEMM on the scaled
    // fp4 MFMA, which moves four times the k per cycle.  K (in binary16 units) covers all of them: K = 64 * (nk16 + nkc_a + nkc_b).
    int mix_nk16 = 0, mix_nkc_a = 0, mix_sa = 0, mix_sb = 0;
    // gemm256, 16-bit staged epilogue: also store fp4(value * 2^x8_shift) of the same tile at xb_out + row * ldxb (bytes) + column / 2 -- the e2m1
    // segment of the NEXT GEMM's [hi | x4] operand row (c_fc's epilogue feeding c_proj in ARP_MODE_F16C)
    int x8_shift = -1;
    void* dx4_out = nullptr;          // ... and fp4((value - rn16(value)) * 2^F16C_DX_SHIFT) at dx4_out + row * ldxb + column / 2, straight from the accumulators
                                      // while the tile is staged (the rounded tile in LDS no longer has it): 2-byte stores, for the products that want both sides
    const int* mix_sptr = nullptr;    // MIXC: mix_sa / mix_sb are F16C_X_SHIFT + mix_sptr[0] / F16C_DX_SHIFT + mix_sptr[1], read on the device (weights whose
                                      // scales are chosen by a device kernel each step: the policy's adapter, arp_dt.hip)
};
constexpr int GEMM_SITE_ADAMW = 26;

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_THREADS = 256;
constexpr int GEMM_ROW_BYTES = 128;                                   // one LDS row = one K-tile of one row
constexpr int GEMM_STAGE_BYTES = (GEMM_BM + GEMM_BN) * GEMM_ROW_BYTES;  // 32 KiB
constexpr int GEMM_LDS_BYTES = 2 * GEMM_STAGE_BYTES;                  // 64 KiB -> 2 workgroups / CU

// bijective XCD remap (cdna_hip_programming.md T1): blocks b, b+8, ... share an XCD (L2); give
// each XCD a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// mean / rstd of row m from its 128-column partial sums
__device__ __forceinline__ void ln_row_stats(const GemmArgs& g, int m, float& mu, float& rs) {
    const float* st = g.ln_stats + (size_t)m * g.ln_parts * 2;
    float s = 0.f, q = 0.f;
    for (int p = 0; p < g.ln_parts; ++p) {
        s += st[2 * p];
        q += st[2 * p + 1];
    }
    mu = s * g.ln_inv_d;
    rs = 1.0f / sqrtf(fmaxf(q * g.ln_inv_d - mu * mu, 0.f) + g.ln_eps);
}
// sum over each 32-lane half of the wavefront (all lanes of the half receive it)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// STAGES = ring depth of the K loop.  2 (default): two workgroups per CU, a K-tile's loads land under ONE K-tile of MFMAs -- what a grid
// that fills the chip wants (a four-stage ring was 3-8 % slower on the train steps' split-K streams: two resident workgroups already
// keep as many bytes in flight).  4: three K-tiles of loads in flight behind counted vmcnt waits, for grids of at most ONE workgroup
// per CU -- the single-frame (online reward) tower, whose 24-96-tile GEMMs otherwise spend a full memory round trip per K-tile.
template <typename T, typename OutT, int ACT, bool RESID, int SITE, int STAGES = 2>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPB = GEMM_ROW_BYTES / (int)sizeof(T);  // elements per K-tile: 64 (bf16) / 32 (f32)
    constexpr int EPC = 16 / (int)sizeof(T);              // elements per 16-byte chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (g.N + GEMM_BN - 1) / GEMM_BN;
    const int m_tiles = (g.M + GEMM_BM - 1) / GEMM_BM;
    const int tile = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int m0 = (g.m_fast ? tile % m_tiles : tile / n_tiles) * GEMM_BM;
    const int n0 = (g.m_fast ? tile / m_tiles : tile % n_tiles) * GEMM_BN;

    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.W);

    // Two workgroups share a CU.  Launched together they stay in lockstep (equal tiles) and reach their store
    // epilogues together; delaying the second resident wave of workgroups by about half a tile puts one
    // workgroup's epilogue under the other's K loop.
    if (g.stagger_groups > 1 && blockIdx.x >= 256 && blockIdx.x < 512 && blockIdx.y == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)g.stagger_cycles) __builtin_amdgcn_s_sleep(64);
    }

    // ---- LDS-DMA staging: each wave-instruction fills 1 KiB = 8 rows x 128 B -----------------
    // lane -> (row-in-group = lane/8, physical chunk = lane%8); it fetches logical chunk
    // (lane%8) ^ (row&7) so that a reader of logical chunk c finds it at physical c ^ (row&7).
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;  // (row & 7) == srow because groups start at multiples of 8
    const T* a_src[4];
    const T* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i * 4 + wave) * 8 + srow;
        int am = m0 + r;
        am = am < g.M ? am : g.M - 1;  // clamp: rows past M are computed on valid memory and never stored
        a_src[i] = A + (size_t)am * g.lda + schunk * EPC;
        int wn = n0 + r;
        wn = wn < g.N ? wn : g.N - 1;
        w_src[i] = W + (size_t)wn * g.ldw + schunk * EPC;
    }
    int kt0 = 0;
    int nk = g.K / EPB;
    if (g.ksplit > 1) {  // this block's K-tile range
        const int per = (nk + g.ksplit - 1) / g.ksplit;
        kt0 = blockIdx.y * per;
        nk = min(per, nk - kt0);
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * GEMM_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            char* dst = base + (i * 4 + wave) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (size_t)(kt0 + kt) * EPB),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + (size_t)(kt0 + kt) * EPB),
                                             (__attribute__((address_space(3))) void*)(dst + GEMM_BM * GEMM_ROW_BYTES), 16, 0, 0);
        }
    };

    // ---- fragment addressing -------------------------------------------------------------------
    const int wr = wave >> 1, wc = wave & 1;  // wave tile: rows (m) wr*64.., cols (n) wc*64..
    const int fr = lane & 15, fg = lane >> 4;
    // byte offset of (row, logical chunk c) inside a tile: row*128 + ((c ^ (row&7)) << 4)
    int a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ar = wr * 64 + i * 16 + fr;
        const int wrow = wc * 64 + i * 16 + fr;
        a_off[i] = ar * GEMM_ROW_BYTES;
        w_off[i] = GEMM_BM * GEMM_ROW_BYTES + wrow * GEMM_ROW_BYTES;
    }
    const int sw = fr & 7;  // (row & 7) for every fragment row of this lane (tile bases are multiples of 16)

    f32x4_v acc[4][4];  // [ni][mi]
    const int mfrag_live = min(4, max(0, (g.M - m0 - wr * 64 + 15) >> 4));  // this wave's 16-row m-fragments that hold a real row
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    static_assert(STAGES == 2 || STAGES == 4, "ring depths with instantiated wait counts");
    if constexpr (STAGES == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        for (int s = 0; s < STAGES - 1 && s < nk; ++s) stage(s, s);
    }

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & (STAGES - 1);
        if constexpr (STAGES == 2) {
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        } else {
            // tile kt has landed once at most the younger tiles (8 LDS-DMA instructions per thread each) are still in flight; the
            // barrier also says every wave is done with tile kt-1, whose slot tile kt+3 goes to
            const int ahead = min(nk - 1 - kt, STAGES - 2);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) & (STAGES - 1), kt + STAGES - 1);
        }
        const char* base = smem + cur * GEMM_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fg) ^ sw) << 4;
            u32x4_v af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const u32x4_v*>(base + a_off[i] + coff);
                wf[i] = *reinterpret_cast<const u32x4_v*>(base + w_off[i] + coff);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (sizeof(T) == 2) {
                        acc[ni][mi] = mfma16<T>(wf[ni], af[mi], acc[ni][mi]);
                    } else {
                        // f32: a handful of rows against a long K (the policy's f32 image_text_input at batch 1: 4 rows x 197 376) would
                        // spend 16x its useful time in 16x16x4 MFMAs on clamped rows; m-fragments wholly past M are skipped (wave-uniform)
                        if (mi >= mfrag_live) continue;
                        // 16 floats of K per (ks): lane group fg holds k = 16*ks + 4*fg + j in element j
                        // of BOTH operands, so the four 16x16x4 MFMAs (j = 0..3) cover them all.
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(wf[ni][j]), __uint_as_float(af[mi][j]), acc[ni][mi], 0, 0, 0);
                    }
                }
        }
        if constexpr (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's fragment reads are behind the wave before the next barrier
        }
    }
    if constexpr (STAGES != 2) __syncthreads();  // the staged epilogue overlays the ring

    // ---- epilogue: lane holds, per (ni, mi): m = .. + fr, n = .. + 4*fg + {0,1,2,3} ----------------
    OutT* out = static_cast<OutT*>(g.out) + (size_t)blockIdx.y * g.slice_stride;  // may alias g.resid (in-place residual add)
    const bool vec_ok = ((g.N | g.ldo | g.ldr) & 3) == 0;
    // ---- staged epilogue (see gemm256.h): accumulators -> bias/activation -> LDS -> whole 256-B / 512-B rows.
    // With two workgroups per CU one workgroup's store phase runs under the other's K loop.
    const bool staged = vec_ok && ((g.N | g.ldo) & 7) == 0 && !(g.flags & 2);
    if (staged) {
        if constexpr (sizeof(OutT) == 2) {
            constexpr int RS = 128 * 2 + 16;
            // the thread's four bias (and folded-LayerNorm c) fragments once, unconditionally from clamped columns, ahead of the staging loop: behind a per-fragment
            // `n < N` branch each was a load + vmcnt(0) of its own (gemm256.h; N % 8 == 0 on this path, columns past N are never stored)
            float4 bq[4], cq[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int nc = min(n0 + wc * 64 + ni * 16 + fg * 4, g.N - 4);
                // (a pointer select, then an unconditional load: with no bias / no fold the 16 bytes come from the A operand and are not used)
                bq[ni] = *reinterpret_cast<const float4*>(g.bias ? g.bias + nc : reinterpret_cast<const float*>(g.A));
                cq[ni] = *reinterpret_cast<const float4*>(g.ln_stats ? g.ln_c + nc : reinterpret_cast<const float*>(g.A));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float mu = 0.f, rs = 1.f;
                if (g.ln_stats) ln_row_stats(g, min(m0 + wr * 64 + mi * 16 + fr, g.M - 1), mu, rs);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int row = wr * 64 + mi * 16 + fr, col = wc * 64 + ni * 16 + fg * 4;
                    float v[4] = {acc[ni][mi][0] * g.alpha, acc[ni][mi][1] * g.alpha, acc[ni][mi][2] * g.alpha, acc[ni][mi][3] * g.alpha};
                    if (g.ln_stats) {
                        const float4 c4 = cq[ni];
                        v[0] = rs * (v[0] - mu * c4.x); v[1] = rs * (v[1] - mu * c4.y);
                        v[2] = rs * (v[2] - mu * c4.z); v[3] = rs * (v[3] - mu * c4.w);
                    }
                    if (g.bias) {  // (uniform: no load inside)
                        const float4 b = bq[ni];
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) == 2>(v[j]);
                    *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int r = it * 16 + wave * 4 + (lane >> 4);
                const int m = m0 + r, n = n0 + (lane & 15) * 8;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<uint4*>(out + (size_t)m * g.ldo + n) = *reinterpret_cast<const uint4*>(smem + r * RS + (lane & 15) * 16);
            }
        } else {
            constexpr int RSF = 128 * 4 + 16;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p) __syncthreads();
                // the pass's 8 residual rows per thread are requested before the tile is staged (see gemm256.h)
                float4 rres[8];
                if constexpr (RESID) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int lr = it * 8 + wave * 2 + (lane >> 5);
                        const int m = m0 + (lr >> 5) * 64 + p * 32 + (lr & 31), n = n0 + (lane & 31) * 4;
                        // unconditional, clamped address (gemm256.h: a guarded load is a branch + vmcnt(0) per row; N % 8 == 0 on the staged path)
                        rres[it] = *reinterpret_cast<const float4*>(g.resid + (size_t)min(m, g.M - 1) * g.ldr + min(n, g.N - 4));
                    }
                }
                // Fused AdamW: the weight's p / m / v rows one iteration AHEAD (ARP_ADAMW_PIPE, default on) -- iteration it + 1's three loads are issued
                // before iteration it's four stores, and iteration 0's before the tile is staged.  A wave's loads and stores retire through one in-order
                // counter: issued behind the stores (round 3), every iteration's loads also waited for the previous iteration's writes to reach memory.
                // Addresses of rows / columns past the edge are clamped (finite garbage, never used) so that the loads are unconditional.
                // (-DARP_ADAMW_PIPE=2 / 3 request two / three iterations ahead: measured the same as one on one box -- 3.47, 3.46, 3.47 ms per step.)
                constexpr bool ADAMW_PIPE = (ARP_ADAMW_PIPE != 0) && SITE == GEMM_SITE_ADAMW && !RESID && sizeof(T) == 2;
                constexpr int AD = ARP_ADAMW_PIPE > 1 ? ARP_ADAMW_PIPE : 1, AS = AD + 1;  // iterations ahead, register slots
                float4 pq[AS], mq[AS], vq[AS];
                auto adam_idx = [&](int it) {
                    const int lr = it * 8 + wave * 2 + (lane >> 5);
                    const int m = min(m0 + (lr >> 5) * 64 + p * 32 + (lr & 31), g.M - 1), n = min(n0 + (lane & 31) * 4, g.N - 4);
                    return (size_t)m * g.ldo + n;
                };
                auto adam_load = [&](int it) {
                    const size_t idx = adam_idx(it);
                    pq[it % AS] = *reinterpret_cast<const float4*>(g.adam_p + idx);
                    mq[it % AS] = *reinterpret_cast<const float4*>(g.adam_m + idx);
                    vq[it % AS] = *reinterpret_cast<const float4*>(g.adam_v + idx);
                };
                if constexpr (ADAMW_PIPE) {
#pragma unroll
                    for (int a = 0; a < AD; ++a) adam_load(a);
                }
                float4 bqf[4];  // the four bias fragments once per pass, unconditionally (see the 16-bit path above)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    bqf[ni] = *reinterpret_cast<const float4*>(g.bias ? g.bias + min(n0 + wc * 64 + ni * 16 + fg * 4, g.N - 4) : reinterpret_cast<const float*>(g.A));
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const int mi = 2 * p + mh;
                        const int lrow = wr * 32 + mh * 16 + fr, col = wc * 64 + ni * 16 + fg * 4;
                        float v[4] = {acc[ni][mi][0] * g.alpha, acc[ni][mi][1] * g.alpha, acc[ni][mi][2] * g.alpha, acc[ni][mi][3] * g.alpha};
                        if (g.bias) {  // (uniform: no load inside)
                            const float4 b = bqf[ni];
                            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) == 2>(v[j]);
                        *reinterpret_cast<float4*>(smem + lrow * RSF + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int lr = it * 8 + wave * 2 + (lane >> 5);
                    const int m = m0 + (lr >> 5) * 64 + p * 32 + (lr & 31), n = n0 + (lane & 31) * 4;
                    const bool ok = m < g.M && n < g.N;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (ADAMW_PIPE) {
                        if (it + AD < 8) adam_load(it + AD);
                    }
                    if (ok) {
                        v = *reinterpret_cast<const float4*>(smem + lr * RSF + (lane & 31) * 16);
                        if constexpr (RESID) {
                            const float4 r = rres[it];
                            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                        }
                        if constexpr (SITE == GEMM_SITE_ADAMW && !RESID && sizeof(T) == 2) {
                            // v = this weight's raw gradient: AdamW in place (ftops.h::ft_adamw_kernel's arithmetic, element for element)
                            // (requesting the pass's 24 p / m / v rows ahead of the staging, as the residual epilogue does with its 8, measured
                            //  SLOWER: 4.4 instead of 5.5 TB/s -- 96 more live registers under the staging loop)
                            const size_t idx = (size_t)m * g.ldo + n;
                            float4 pv, mv, vv;
                            if constexpr (ADAMW_PIPE) { pv = pq[it % AS]; mv = mq[it % AS]; vv = vq[it % AS]; }
                            else {
                                pv = *reinterpret_cast<const float4*>(g.adam_p + idx);
                                mv = *reinterpret_cast<const float4*>(g.adam_m + idx);
                                vv = *reinterpret_cast<const float4*>(g.adam_v + idx);
                            }
                            float gg[4] = {v.x, v.y, v.z, v.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, nn[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float gi = gg[e] * g.adam_gscale;
                                if (g.adam_mask) {  // one atomic per wave instruction, not per element (a badly overflowing step would serialise millions of same-address atomics here)
                                    const bool bad = !(fabsf(gi) < 3.0e38f);
                                    const unsigned long long bal = __ballot(bad);
                                    if (bad) gi = 0.f;
                                    if (bal && lane == __ffsll((long long)bal) - 1) atomicAdd(g.adam_dropped, (unsigned int)__popcll(bal));
                                }
                                mm[e] = g.adam_b1 * mm[e] + (1.f - g.adam_b1) * gi;
                                nn[e] = g.adam_b2 * nn[e] + (1.f - g.adam_b2) * gi * gi;
                                const float pd = pp[e] * (1.f - g.adam_lr * g.adam_wd);
                                pp[e] = pd - (g.adam_lr / g.adam_bc1) * mm[e] / (sqrtf(nn[e]) / sqrtf(g.adam_bc2) + g.adam_eps);
                            }
                            *reinterpret_cast<float4*>(g.adam_m + idx) = make_float4(mm[0], mm[1], mm[2], mm[3]);
                            *reinterpret_cast<float4*>(g.adam_v + idx) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                            *reinterpret_cast<float4*>(g.adam_p + idx) = make_float4(pp[0], pp[1], pp[2], pp[3]);
                            if (g.adam_mirror) store4(static_cast<T*>(g.adam_mirror) + idx, pp[0], pp[1], pp[2], pp[3]);
                        } else {
                        if (g.out) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * g.ldo + n) = v;
                        if (g.xb_out) {
                            if (g.split3) store_split3(static_cast<T*>(g.xb_out) + (size_t)m * g.ldxb + n, (size_t)g.N, v.x, v.y, v.z, v.w);
                            else store4(static_cast<T*>(g.xb_out) + (size_t)m * g.ldxb + n, v.x, v.y, v.z, v.w);
                        }
                        }
                    }
                    if (g.stats_out) {  // one 128-column segment per 32-lane half
                        const float s = half_wave_sum((v.x + v.y) + (v.z + v.w));
                        const float q = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
                        if ((lane & 31) == 0 && ok) {
                            float* st = g.stats_out + ((size_t)m * (g.N >> 7) + (n0 >> 7)) * 2;
                            st[0] = s;
                            st[1] = q;
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fg * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[ni][mi][0] * g.alpha, acc[ni][mi][1] * g.alpha, acc[ni][mi][2] * g.alpha, acc[ni][mi][3] * g.alpha};
            if (vec_ok) {
                if (g.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) == 2>(v[j]);
                if constexpr (RESID) {
                    const float4 r = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                store4(out + (size_t)m * g.ldo + n, v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= g.N) break;
                    float x = v[j] + (g.bias ? g.bias[n + j] : 0.f);
                    x = apply_act<ACT, sizeof(T) == 2>(x);
                    if constexpr (RESID) x += g.resid[(size_t)m * g.ldr + n + j];
                    Elem<OutT>::st(out + (size_t)m * g.ldo + n + j, x);
                }
            }
        }
    }
}

// Host launcher.  Requirements: K % (128/sizeof(T)) == 0, lda/ldw multiples of 16/sizeof(T),
// 16-byte aligned bases.  M and N are arbitrary (guarded; vector epilogue when N, ldo, ldr % 4 == 0).
template <typename T, typename OutT, int ACT, bool RESID, int SITE, int STAGES = 2>
inline int launch_gemm_nt(const GemmArgs& g, hipStream_t stream) {
    constexpr int EPB = GEMM_ROW_BYTES / (int)sizeof(T);
    if (g.M <= 0) return 0;
    if constexpr (STAGES == 2) {
        // a grid that cannot even give every CU one workgroup, with a contraction long enough to pipeline: the four-stage instance
        static const bool deep_ok = [] { const char* e = getenv("ARP_GEMM_DEEP"); return !e || atoi(e) != 0; }();
        const long wgs = (long)((g.M + GEMM_BM - 1) / GEMM_BM) * ((g.N + GEMM_BN - 1) / GEMM_BN) * (g.ksplit > 1 ? g.ksplit : 1);
        if (deep_ok && wgs <= 128 && g.K / EPB >= 6) return launch_gemm_nt<T, OutT, ACT, RESID, SITE, 4>(g, stream);
    }
    if (g.N <= 0 || g.K % EPB != 0 || g.K <= 0 || g.lda % (16 / (int)sizeof(T)) != 0 || g.ldw % (16 / (int)sizeof(T)) != 0)
        return fail("gemm_nt: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) +
                    " K=" + std::to_string(g.K));
    // The staged epilogues (N, ldo multiples of 8) load their bias / folded-LayerNorm / residual fragments UNCONDITIONALLY from clamped addresses, and with no
    // bias or fold a dummy 16 bytes from A: that needs N >= 8 (implied), a 16-byte-aligned A and one whole readable row of it
    if (((g.N | g.ldo) & 7) == 0 && !(g.flags & 2) && ((reinterpret_cast<uintptr_t>(g.A) & 15) != 0 || (size_t)g.K * sizeof(T) < 16))
        return fail("gemm_nt: the staged epilogue needs a 16-byte-aligned A operand");
    auto kern = gemm_nt_kernel<T, OutT, ACT, RESID, SITE, STAGES>;
    constexpr int lds_bytes = STAGES * GEMM_STAGE_BYTES;
    static_assert(lds_bytes >= GEMM_LDS_BYTES, "the staged epilogue needs the two-stage ring's 64 KiB");
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    const int m_tiles = (g.M + GEMM_BM - 1) / GEMM_BM;
    const int n_tiles = (g.N + GEMM_BN - 1) / GEMM_BN;
    hipLaunchKernelGGL(kern, dim3(m_tiles * n_tiles, g.ksplit > 1 ? g.ksplit : 1), dim3(GEMM_THREADS), lds_bytes, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp

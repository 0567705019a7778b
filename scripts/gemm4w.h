// 256x256-tile NT GEMM on FOUR waves (one per SIMD, 512 registers each):  C[M,N] = epilogue(A[M,K] . W[N,K]^T), 16-bit operands.
//
// Why (round 3): the eight-wave kernel of gemm256.h spends ~2550 cycles per K-tile against 2048 of MFMA time (in-kernel stamps,
// profiles/r3_g256_fine.txt): its two waves per SIMD share that SIMD's vector issue port -- every ds_read_b128 / LDS-DMA of one wave
// comes out of the other's MFMA stream -- and each wave's 128x64 tile re-reads (128 + 64) x 128 B of LDS per K-tile, 192 KB per
// workgroup.  Here a wave owns 128x128 (the 256 accumulator registers live in AGPRs, the other 256 hold two fragment sets): 128 KB of
// LDS reads per K-tile instead of 192, no partner wave on the SIMD, one barrier per K-tile instead of four.
//
//   * 4 waves as 2 (M) x 2 (N), 8 x 8 MFMA fragments (v_mfma_f32_16x16x32) per wave, K-tile = 128 B per row, two k-steps of 64 MFMAs.
//   * LDS ring: 2 K-tile buffers x [A 256 rows | W 256 rows] x 128 B = 128 KiB, 16-byte chunks XOR-swizzled by the row as in
//     gemm256.h (on the LDS-DMA source address and on the ds_read_b128 address).
//   * step (t,0) computes on fragment set 0 and reads set 1 = k-step 1 of K-tile t; then lgkmcnt(0), vmcnt(0) (K-tile t+1 has landed),
//     ONE barrier; step (t,1) computes on set 1, reads set 0 = k-step 0 of K-tile t+1 and issues the 16 LDS-DMA instructions of
//     K-tile t+2 into the buffer K-tile t has just left.  Every read / DMA sits behind four MFMAs of the same step.
//   * per accumulator the MFMA sequence is K-tiles ascending, k-step 0 then 1 -- the order of gemm256.h / gemm.h, so the result is
//     bit-identical to those kernels (scripts/gemm4w_bench.hip compares FNV checksums).
//
// MEASURED, NOT USED BY THE LIBRARY (profiles/r3_gemm4w_ab.txt, one process, interleaved rounds, same operands): correct -- FNV
// checksums equal gemm256's on every shape -- and its K loop is shorter in cycles (qkv 33.4 k against 36.6 k per tile, c_fc 33.7 k
// against 35.5 k), but four waves take longer over the staged epilogue (4.3 k + 4.9 k cycles against 2.4 k + 4.0 k; with QuickGELU
// 9.4 k against 6.1 k: one wave per SIMD has nobody to overlap its VALU and LDS latencies with) and the wall time is 4-8 % LONGER
// (qkv 204 vs 188 us, c_fc 274 vs 261, c_proj 291 vs 269, 4096^3 1213 vs 1264 TF).  Putting the 16 LDS-DMA of a K-tile into the first
// half of their step was 7-15 % slower still (the texture-address path takes one 1-KiB instruction per 16 cycles per CU), and removing
// the vmcnt wait altogether (wrong results, timing only) changed nothing: the loop is bound by instruction issue -- per four MFMAs
// one ds_read_b128, one LDS-DMA, one M0 write and one 64-bit address add leave the in-order wave no slack -- not by load latency.
// Kept as the starting point for a kernel whose epilogue runs in the MFMA shadows of the next tile (scripts/gemm4w_bench.hip).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace arp {

constexpr int G4_BM = 256, G4_BN = 256, G4_THREADS = 256;
constexpr int G4_BUF = 64 * 1024, G4_W_OFF = 32 * 1024;
constexpr int G4_TILE_BYTES = 256 * (256 * 2 + 16);  // the padded 16-bit epilogue tile (the f32 one, 128 x 1040 B, is smaller)
constexpr int G4_BIAS_OFF = G4_TILE_BYTES;
constexpr int G4_LDS_BYTES = G4_BIAS_OFF + 1024;
static_assert(G4_TILE_BYTES >= 2 * G4_BUF && G4_TILE_BYTES >= 128 * (256 * 4 + 16) && G4_LDS_BYTES <= 160 * 1024, "LDS plan");

#ifdef ARP_G4_STAMPS
__device__ long long* arp_g4_stamps = nullptr;
#endif
#ifndef ARP_G4_FRONT
#define ARP_G4_FRONT 0
#endif
#ifndef ARP_G4_NOWAIT
#define ARP_G4_NOWAIT 0
#endif

// One 16x16x32 MFMA accumulating IN PLACE in the accumulator file ("+a").  Through the builtin, hipcc (ROCm 7.2) un-ties destination
// and addend at 512 registers and shuffles the 256 loop-carried accumulators between a[] and v[] -- 380 v_accvgpr copies per K-tile
// beside 128 MFMAs.  The statement reads only registers the compiler has waited for ("v" operands of tracked ds_reads) and an
// accumulate chain needs no wait states (cdna_hip_programming.md section 5.7 item 2).
template <typename T> __device__ __forceinline__ void mfma16_acc(f32x4_v& c, const u32x4_v& a, const u32x4_v& b) {
    if constexpr (__is_same(T, f16_t)) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <typename T, typename OutT, int ACT, bool RESID>
__global__ __launch_bounds__(G4_THREADS, 1) void gemm4w_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = 16 / (int)sizeof(T);
    static_assert(sizeof(T) == 2, "16-bit operands");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int n_tiles = (g.N + G4_BN - 1) / G4_BN;
    const int m_tiles = (g.M + G4_BM - 1) / G4_BM;
    const int total_tiles = m_tiles * n_tiles;
    const int group_m = g.group_m > 0 ? g.group_m : 8;
    int m0, n0;
    {
        int t = xcd_remap(blockIdx.x, total_tiles);
        const int per_group = group_m * n_tiles;
        const int grp = t / per_group;
        const int first_m = grp * group_m;
        const int gsize = min(m_tiles - first_m, group_m);
        t -= grp * per_group;
        m0 = (first_m + t % gsize) * G4_BM;
        n0 = (t / gsize) * G4_BN;
    }
    const char* __restrict__ Ab = static_cast<const char*>(g.A);
    const char* __restrict__ Wb = static_cast<const char*>(g.W);

    // ---- LDS-DMA plan: 64 wave-instructions of 8 rows x 128 B per K-tile; wave w takes row groups j*4 + w of A (j < 8) and of W ----
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    uint32_t off[16];  // byte offset of this lane's 16 bytes inside A (j < 8) / W (j >= 8), K-tile 0
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j * 4 + wave) * 8 + srow;
        int am = m0 + row;
        am = am < g.M ? am : g.M - 1;
        off[j] = (uint32_t)(((size_t)am * g.lda + schunk * EPC) * sizeof(T));
        int wn = n0 + row;
        wn = wn < g.N ? wn : g.N - 1;
        off[8 + j] = (uint32_t)(((size_t)wn * g.ldw + schunk * EPC) * sizeof(T));
    }
    const int nk = g.K / (128 / (int)sizeof(T));
    auto dma = [&](auto J, int tt) {
        constexpr int j = decltype(J)::value;
        const char* base = (j < 8 ? Ab : Wb) + (size_t)tt * 128;
        char* dst = smem + (tt & 1) * G4_BUF + (j < 8 ? 0 : G4_W_OFF) + ((j & 7) * 4 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[j]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma_tile = [&](int tt) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (dma(std::integral_constant<int, J>{}, tt), ...); }(std::make_integer_sequence<int, 16>{});
    };

    // ---- fragment addressing ----
    const int fr = lane & 15, fg = lane >> 4;
    const int a_rd = (wr * 128 + fr) * 128;
    const int b_rd = G4_W_OFF + (wc * 128 + fr) * 128;
    const int coff0 = ((0 * 4 + fg) ^ (fr & 7)) << 4;
    const int coff1 = ((1 * 4 + fg) ^ (fr & 7)) << 4;

    f32x4_v acc[8][8];  // [mi][ni]
    u32x4_v fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    // the tile's bias slice goes to LDS ahead of everything else (oldest in the vmcnt order)
    float* bias_s = reinterpret_cast<float*>(smem + G4_BIAS_OFF);
    if (g.bias && wave == 0) {
        int n = n0 + lane * 4;
        n = n + 4 <= g.N ? n : (g.N >= 4 ? g.N - 4 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.bias + n), (__attribute__((address_space(3))) void*)bias_s, 16, 0, 0);
    }
#ifdef ARP_G4_STAMPS
    long long st_[4];
    st_[0] = __builtin_amdgcn_s_memtime();
#endif
    // one k-step: 64 MFMAs on the CUR fragment set; behind every four of them one fragment read of the OTHER set (READ) and one LDS-DMA
    // instruction (DMA).  rbuf / rks: where the other set is read from; dt: the K-tile the DMA instructions fetch.
    auto step = [&](auto CUR, auto READ, auto DMA, const char* rbuf, int rco, int dt) {
        constexpr int cur = decltype(CUR)::value;
        u32x4_v(&ca)[8] = cur ? fa1 : fa0;
        u32x4_v(&cb)[8] = cur ? fb1 : fb0;
        u32x4_v(&na)[8] = cur ? fa0 : fa1;
        u32x4_v(&nb)[8] = cur ? fb0 : fb1;
        [&]<int... G>(std::integer_sequence<int, G...>) {
            (([&] {
                 constexpr int gq = G;
#pragma unroll
                 for (int q = 0; q < 4; ++q) {
                     constexpr int dummy = 0;
                     (void)dummy;
                     const int idx = gq * 4 + q;
                     const int ni = idx >> 3, mi = idx & 7;
                     mfma16_acc<T>(acc[mi][ni], cb[ni], ca[mi]);
                 }
                 if constexpr (decltype(READ)::value) {
                     if constexpr (gq < 8) na[gq] = *reinterpret_cast<const u32x4_v*>(rbuf + a_rd + gq * 2048 + rco);
                     else nb[gq - 8] = *reinterpret_cast<const u32x4_v*>(rbuf + b_rd + (gq - 8) * 2048 + rco);
                 }
#if ARP_G4_FRONT
                 // the 16 LDS-DMA instructions of a K-tile in the first half of the step (two per group): 512 cycles more to land
                 if constexpr (decltype(DMA)::value && gq < 8) {
                     dma(std::integral_constant<int, 2 * gq>{}, dt);
                     dma(std::integral_constant<int, 2 * gq + 1>{}, dt);
                 }
#else
                 if constexpr (decltype(DMA)::value) dma(std::integral_constant<int, gq>{}, dt);
#endif
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, 16>{});
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N_ = std::false_type;

    // ---- prologue: K-tiles 0 and 1 in flight, K-tile 0 landed, fragment set 0 = k-step 0 of K-tile 0 ----
    dma_tile(0);
    if (nk > 1) dma_tile(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa0[i] = *reinterpret_cast<const u32x4_v*>(smem + a_rd + i * 2048 + coff0);
        fb0[i] = *reinterpret_cast<const u32x4_v*>(smem + b_rd + i * 2048 + coff0);
    }
    auto sync_tile = [&]() {  // my reads of the current buffer are in registers, my share of the next K-tile has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !ARP_G4_NOWAIT  // (timing experiment only: results are garbage without the wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; t + 2 < nk; ++t) {
        const char* buf = smem + (t & 1) * G4_BUF;
        const char* nbuf = smem + ((t + 1) & 1) * G4_BUF;
        step(C0{}, Y{}, N_{}, buf, coff1, 0);
        sync_tile();
        step(C1{}, Y{}, Y{}, nbuf, coff0, t + 2);
    }
    if (t + 1 < nk) {  // second-to-last K-tile: nothing left to fetch
        const char* buf = smem + (t & 1) * G4_BUF;
        const char* nbuf = smem + ((t + 1) & 1) * G4_BUF;
        step(C0{}, Y{}, N_{}, buf, coff1, 0);
        sync_tile();
        step(C1{}, Y{}, N_{}, nbuf, coff0, 0);
        ++t;
    }
    {   // last K-tile
        const char* buf = smem + (t & 1) * G4_BUF;
        step(C0{}, Y{}, N_{}, buf, coff1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        step(C1{}, N_{}, N_{}, buf, coff0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");  // + the wait states between the last MFMA and the compiler's accumulator reads
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring: LDS becomes the epilogue tile
    __builtin_amdgcn_sched_barrier(0);
#ifdef ARP_G4_STAMPS
    st_[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: accumulators -> bias / activation -> LDS tile -> whole rows out (as gemm256.h) ----
    OutT* out = static_cast<OutT*>(g.out);
    float4 bq[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
        bq[ni] = g.bias ? *reinterpret_cast<const float4*>(bias_s + wc * 128 + ni * 16 + fg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.flags & 1) {  // ablation: keep the accumulators live, store nothing
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sacc == 12345.678f) Elem<OutT>::st(out, sacc);
        return;
    }
    if constexpr (sizeof(OutT) == 2) {
        constexpr int RS = 256 * 2 + 16;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int row = wr * 128 + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int col = wc * 128 + ni * 16 + fg * 4;
                const f32x4_v a4 = acc[mi][ni];
                const float4 b = bq[ni];
                float v[4] = {a4[0] + b.x, a4[1] + b.y, a4[2] + b.z, a4[3] + b.w};
                apply_act4<ACT, true>(v);
                *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
            }
        }
#ifdef ARP_G4_STAMPS
        st_[2] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#pragma unroll 8
        for (int it = 0; it < 32; ++it) {
            const int r = it * 8 + wave * 2 + (lane >> 5);
            const int m = m0 + r, n = n0 + (lane & 31) * 8;
            if (m < g.M && n < g.N)
                *reinterpret_cast<u32x4_v*>(out + (size_t)m * g.ldo + n) = *reinterpret_cast<const u32x4_v*>(smem + r * RS + (lane & 31) * 16);
        }
    } else {
        constexpr int RSF = 256 * 4 + 16;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p) __syncthreads();
            float4 rres[32];  // the pass's residual rows are requested before the tile is staged: all in flight at once
            if constexpr (RESID) {
#pragma unroll
                for (int it = 0; it < 32; ++it) {
                    const int lr = it * 4 + wave;
                    const int m = m0 + (lr >> 6) * 128 + p * 64 + (lr & 63), n = n0 + lane * 4;
                    rres[it] = (m < g.M && n < g.N) ? *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) {
                    const int lrow = wr * 64 + mi * 16 + fr;
                    const int col = wc * 128 + ni * 16 + fg * 4;
                    const f32x4_v a4 = acc[p * 4 + mi][ni];
                    const float4 b = bq[ni];
                    float v[4] = {a4[0] + b.x, a4[1] + b.y, a4[2] + b.z, a4[3] + b.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, true>(v[j]);
                    *reinterpret_cast<float4*>(smem + lrow * RSF + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int lr = it * 4 + wave;
                const int m = m0 + (lr >> 6) * 128 + p * 64 + (lr & 63), n = n0 + lane * 4;
                if (m < g.M && n < g.N) {
                    float4 v = *reinterpret_cast<const float4*>(smem + lr * RSF + lane * 16);
                    if constexpr (RESID) {
                        const float4 r = rres[it];
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * g.ldo + n) = v;
                }
            }
        }
    }
#ifdef ARP_G4_STAMPS
    st_[3] = __builtin_amdgcn_s_memtime();
    if (arp_g4_stamps && (threadIdx.x & 63) == 0) {
        long long* d = arp_g4_stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
        for (int i = 0; i < 4; ++i) d[i] = st_[i];
    }
#endif
}

template <typename T, typename OutT, int ACT, bool RESID>
inline bool gemm4w_supported(const GemmArgs& g) {
    constexpr int EPB = 128 / (int)sizeof(T);
    const size_t abytes = (size_t)(g.M > 0 ? g.M - 1 : 0) * g.lda * sizeof(T) + (size_t)g.K * sizeof(T);
    const size_t wbytes = (size_t)(g.N > 0 ? g.N - 1 : 0) * g.ldw * sizeof(T) + (size_t)g.K * sizeof(T);
    return g.M > 0 && g.N > 0 && g.K > 0 && g.K % EPB == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 && abytes < (1ull << 32) && wbytes < (1ull << 32) &&
           ((g.N | g.ldo) & 7) == 0 && !(RESID && (g.ldr & 3)) && !g.ln_stats && !g.stats_out && !g.xb_out && !g.mask && g.ksplit <= 1 && g.alpha == 1.f;
}

template <typename T, typename OutT, int ACT, bool RESID>
inline int launch_gemm4w(const GemmArgs& g, hipStream_t stream) {
    if (!gemm4w_supported<T, OutT, ACT, RESID>(g)) return fail("gemm4w: unsupported shape");
    auto kern = gemm4w_kernel<T, OutT, ACT, RESID>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES));
        attr_set = true;
    }
    const int grid = ((g.M + G4_BM - 1) / G4_BM) * ((g.N + G4_BN - 1) / G4_BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G4_THREADS), G4_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp

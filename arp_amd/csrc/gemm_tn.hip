// TN weight-gradient GEMM (design notes in gemm_tn.h); its own translation unit.
#include "gemm_tn.h"

#include <type_traits>

#include "attention.h"  // tr_b64_v
#include "gemm.h"       // xcd_remap

namespace arp {

constexpr int TN_BM = 128, TN_BN = 128, TN_BK = 64, TN_THREADS = 256;
constexpr int TN_OP_BYTES = TN_BK * 256;           // one operand tile: 64 rows x 256 B
constexpr int TN_STAGE_BYTES = 2 * TN_OP_BYTES;    // 32 KiB
constexpr int TN_LDS_BYTES = 2 * TN_STAGE_BYTES;   // 64 KiB -> two workgroups per CU

template <typename T>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(GemmTnArgs g) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int n_tiles = g.N / TN_BN, m_tiles = g.M / TN_BM;
    const int tile = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int m0 = (tile / n_tiles) * TN_BM, n0 = (tile % n_tiles) * TN_BN;

    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);

    int nk = g.K / TN_BK, kt0 = 0;
    if (g.ksplit > 1) {
        const int per = (nk + g.ksplit - 1) / g.ksplit;
        kt0 = blockIdx.y * per;
        nk = min(per, nk - kt0);
        if (nk < 0) nk = 0;
    }

    // ---- LDS-DMA: a piece = 4 rows x 256 B; wave w fills pieces 4w..4w+3 of each operand tile -------------------------------
    // lane -> (row in piece = lane / 16, physical 16-B chunk = lane % 16); physical 32-B slot s holds logical slot s ^ (row & 7)
    const int prow = lane >> 4, pc = lane & 15;
    const T* srcA[4];
    const T* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 4 + prow;
        const int lc = ((((pc >> 1) ^ (row & 7)) << 1) | (pc & 1)) * 8;  // first element of the logical chunk this lane fetches
        srcA[i] = A + (size_t)(kt0 * TN_BK + row) * g.lda + m0 + lc;
        srcB[i] = B + (size_t)(kt0 * TN_BK + row) * g.ldb + n0 + lc;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * TN_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            char* dst = base + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * TN_BK * g.lda),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (size_t)kt * TN_BK * g.ldb),
                                             (__attribute__((address_space(3))) void*)(dst + TN_OP_BYTES), 16, 0, 0);
        }
    };

    // ---- transposing fragment reads: lane 4q+p of a 16-lane group addresses row (k0 + q), bytes 8p..8p+7 of a 32-B slot;
    // lane i receives column i of the four rows.  k-slot (fg, j) of step st <-> k = 32 st + 16 (j >> 2) + 4 fg + (j & 3), the same
    // mapping for both operands.
    const int fr = lane & 15, fg = lane >> 4;
    const int trq = fr >> 2, trp = fr & 3;
    auto frag = [&](const char* op, int blk16, int st) {
        const int r0 = 32 * st + 4 * fg + trq, r1 = r0 + 16;
        const tr_b64_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tr_b64_v*)(op + r0 * 256 + ((blk16 ^ (r0 & 7)) << 5) + trp * 8));
        const tr_b64_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tr_b64_v*)(op + r1 * 256 + ((blk16 ^ (r1 & 7)) << 5) + trp * 8));
        const u32x2_v l2 = __builtin_bit_cast(u32x2_v, lo), h2 = __builtin_bit_cast(u32x2_v, hi);
        return u32x4_v{l2[0], l2[1], h2[0], h2[1]};
    };

    f32x4_v acc[4][4];  // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const char* a_op = smem + cur * TN_STAGE_BYTES;
            const char* b_op = a_op + TN_OP_BYTES;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4_v af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = frag(a_op, wr * 4 + i, st);
                    bf[i] = frag(b_op, wc * 4 + i, st);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16<T>(bf[ni], af[mi], acc[ni][mi]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // ---- epilogue: lane holds m = .. + fr, n = .. + 4 fg + {0..3}: a store instruction straight from the accumulators is sixteen 64-byte pieces in sixteen rows.
    // Each wave stages its 64 x 64 f32 block through its own 16 KiB of the (now idle) operand ring -- 256-B rows, 16-B chunk c of row r at chunk c ^ (r & 15),
    // conflict-free both ways -- and writes it back as four whole 256-byte row segments per instruction (dWi: 101 MB of f32 in 1542 tiles).
    float* out = g.out + (size_t)blockIdx.y * g.slice_stride;
#ifdef ARP_TN_DIRECT_STORE  // the stores straight from the accumulators (rounds 2-5), for A/B builds of scripts/gemm_tn_bench.hip
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int m = m0 + wr * 64 + mi * 16 + fr, n = n0 + wc * 64 + ni * 16 + fg * 4;
            const f32x4_v a4 = acc[ni][mi];
            *reinterpret_cast<float4*>(out + (size_t)m * g.ldo + n) = make_float4(a4[0] * g.alpha, a4[1] * g.alpha, a4[2] * g.alpha, a4[3] * g.alpha);
        }
    return;
#endif
    char* stg = smem + wave * 16384;  // (the K loop's last barrier is behind every wave's last fragment read; a wave reads back only what it wrote)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int row = mi * 16 + fr, chunk = ni * 4 + fg;
            const f32x4_v a4 = acc[ni][mi];
            *reinterpret_cast<float4*>(stg + row * 256 + ((chunk ^ (row & 15)) << 4)) = make_float4(a4[0] * g.alpha, a4[1] * g.alpha, a4[2] * g.alpha, a4[3] * g.alpha);
        }
    const int rl = lane >> 4, ch = lane & 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = 4 * i + rl;
        const float4 v = *reinterpret_cast<const float4*>(stg + row * 256 + ((ch ^ (row & 15)) << 4));
        *reinterpret_cast<float4*>(out + (size_t)(m0 + wr * 64 + row) * g.ldo + n0 + wc * 64 + ch * 4) = v;
    }
}

// ---- 256 x 256 tile variant (the adapter's 768 x 768 x 32 896 weight gradients) ------------------------------------------------
// The 128 x 128 kernel above re-reads each operand six times past L2 (36 tiles x K x 512 B = 606 MB for 100 MB of operands) and
// waits for a whole K-tile's loads at the end of every iteration.  Here: 512 threads (8 waves as 2 x 4, 128 x 64 per wave), K-tiles
// of 32 rows (two 16-KiB operand tiles with 512-B rows), FOUR stages (128 KiB) with three K-tiles of loads in flight behind a counted
// vmcnt, one barrier per K-tile, and the nine tiles of one K-slice placed on ONE XCD (blockIdx % 8 selects the XCD) so that the
// slice's operand rows come from memory once and are shared through that XCD's L2.
#ifdef ARP_TN_STAMPS
__device__ long long* arp_tn_stamps = nullptr;  // scripts/gemm_tn_bench.hip: K-loop shader cycles / 100 MHz ticks / K-tiles per workgroup
#endif
constexpr int TW_BM = 256, TW_BN = 256, TW_BK = 32, TW_THREADS = 512, TW_STAGES = 4;
constexpr int TW_OP_BYTES = TW_BK * 512;          // 32 rows x 512 B
constexpr int TW_STAGE_BYTES = 2 * TW_OP_BYTES;   // 32 KiB
constexpr int TW_LDS_BYTES = TW_STAGES * TW_STAGE_BYTES;

template <typename T>
__global__ __launch_bounds__(TW_THREADS) void gemm_tn256_kernel(GemmTnArgs g, int tiles, int slices_per_xcd) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // physical workgroup p runs on XCD p % 8; that XCD owns slices [xcd * spx, (xcd + 1) * spx) and all their tiles
    int slice, tile;
    if (slices_per_xcd > 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        if (idx >= slices_per_xcd * tiles) return;
        slice = xcd * slices_per_xcd + idx / tiles;
        tile = idx % tiles;
        if (slice >= g.ksplit) return;
    } else {
        slice = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    const int n_tiles = g.N / TW_BN;
    const int m0 = (tile / n_tiles) * TW_BM, n0 = (tile % n_tiles) * TW_BN;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);

    int nk = g.K / TW_BK, kt0 = 0;
    {
        const int per = (nk + g.ksplit - 1) / g.ksplit;
        kt0 = slice * per;
        nk = max(min(per, nk - kt0), 0);
    }
    // ---- LDS-DMA: a piece = 2 rows x 512 B; wave w fills pieces 2w, 2w+1 of each operand tile -------------------------------------
    // lane -> (row in piece = lane / 32, physical 16-B chunk = lane % 32); physical 32-B slot s holds logical slot s ^ (row & 15)
    const int prow = lane >> 5, pc = lane & 31;
    const T* srcA[2];
    const T* srcB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 2 + prow;
        const int lc = ((((pc >> 1) ^ (row & 15)) << 1) | (pc & 1)) * 8;
        srcA[i] = A + (size_t)(kt0 * TW_BK + row) * g.lda + m0 + lc;
        srcB[i] = B + (size_t)(kt0 * TW_BK + row) * g.ldb + n0 + lc;
    }
    auto stage = [&](int kt) {
#if defined(TW_ABL) && (TW_ABL & 2)
        if (kt >= TW_STAGES - 1) return;  // ablation: prologue loads only
#endif
        // ring layout: A slots at [0, 64 KiB), B slots at [64 KiB, 128 KiB), 16 KiB each -- every fragment read of either operand is
        // then one per-lane base register plus an instruction offset (slot * 16 KiB [+ 8 KiB] < 64 KiB)
        char* base = smem + (kt & (TW_STAGES - 1)) * TW_OP_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = base + (wave * 2 + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * TW_BK * g.lda),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (size_t)kt * TW_BK * g.ldb),
                                             (__attribute__((address_space(3))) void*)(dst + TW_STAGES * TW_OP_BYTES), 16, 0, 0);
        }
    };
    // Transposing fragment reads as in the 128-tile kernel (rows of 512 B, slot swizzle by (row & 15)), but issued as inline asm with
    // hand-counted lgkmcnt waits: behind the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows an
    // LDS-DMA issue (it cannot tell the ring slots apart), which drains the whole prefetch ring once per K-tile -- the 128-tile
    // kernel's 1.9 k cycles per K-tile.  A fragment = rows r0 and r0 + 16 (same swizzle: +8192 B as an instruction offset).
    const int fr = lane & 15, fg = lane >> 4;
    const int trq = fr >> 2, trp = fr & 3;
    const int r0 = 4 * fg + trq;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // per-lane byte addresses inside ring slot 0 of each operand; the slot and a fragment's second half are instruction offsets
    uint32_t a_lo[8], b_lo[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_lo[i] = lds0 + r0 * 512 + (((wr * 8 + i) ^ (r0 & 15)) << 5) + trp * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) b_lo[i] = lds0 + TW_STAGES * TW_OP_BYTES + r0 * 512 + (((wc * 4 + i) ^ (r0 & 15)) << 5) + trp * 8;
    // A fragment's two halves stay two 64-bit values until its wait has been passed: the 128-bit MFMA operand is assembled only
    // there, so that any register copy the compiler needs for the tuple reads data that has arrived.
    struct Frag { u32x2_v lo, hi; };
    auto frag = [&]<int OFF>(uint32_t addr, std::integral_constant<int, OFF>) {
        Frag f;
#if defined(TW_ABL) && (TW_ABL & 1)
        asm volatile("v_mov_b32 %0, %1" : "=&v"(f.lo[0]) : "v"(addr));  // ablation: no LDS reads
        f.lo[1] = f.lo[0]; f.hi = f.lo;
#else
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(f.lo), "=&v"(f.hi)
                     : "v"(addr), "n"(OFF), "n"(OFF + 8192));
#endif
        return f;
    };
    auto op = [](const Frag& f) { return u32x4_v{f.lo[0], f.lo[1], f.hi[0], f.hi[1]}; };
#define TW_WAIT4(cnt, x)                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                                    \
                 : "+v"(x[0].lo), "+v"(x[0].hi), "+v"(x[1].lo), "+v"(x[1].hi), "+v"(x[2].lo), "+v"(x[2].hi), "+v"(x[3].lo), "+v"(x[3].hi))

    f32x4_v acc[4][8];  // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    // Software pipeline across the per-tile barrier: a K-tile's MFMAs run as two halves of 16 (m-blocks 0-3, then 4-7).  The
    // fragments of the second half are fetched behind the first half's MFMAs; the NEXT tile's B and first-half A fragments are
    // fetched (after the wait + barrier that publish that tile) behind the second half's.  Every non-MFMA instruction of the step
    // sits in the gap behind an MFMA: the SIMD's vector issue port, shared by the two resident waves, is what the loop is bound by
    // (an MFMA holds it for 8 of its 16 cycles; an LDS read ~4, an LDS-DMA piece ~60), so the steady-state step (FULL) is branch-free
    // and carries the ring slot (SLOT = kt % 4) as a compile-time constant -- no address arithmetic, no loop-end tests.
    auto step = [&]<int SLOT, bool FULL>(int kt, Frag (&bfc)[4], Frag (&alc)[4], Frag (&bfn)[4], Frag (&aln)[4], std::integral_constant<int, SLOT>,
                                         std::bool_constant<FULL>) {
        constexpr int NS = (SLOT + 1) & 3;
        constexpr int coff = SLOT * TW_OP_BYTES, noff = NS * TW_OP_BYTES;
        const uint32_t(&a_c)[8] = a_lo;
        const uint32_t(&a_n)[8] = a_lo;
        const uint32_t(&b_n)[4] = b_lo;
        const bool next = FULL || kt + 1 < nk;
        Frag ah[4];
        TW_WAIT4(0, bfc);  // this step's B and first-half A fragments (fetched behind the previous step's second half)
        TW_WAIT4(0, alc);
        __builtin_amdgcn_sched_barrier(0);
        u32x4_v b4[4], a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { b4[i] = op(bfc[i]); a4[i] = op(alc[i]); }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[ni][mi] = mfma16<T>(b4[ni], a4[mi], acc[ni][mi]);
                if (ni == 0) ah[mi] = frag(a_c[4 + mi], std::integral_constant<int, coff>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        TW_WAIT4(0, ah);  // this wave's last reads of tile kt have returned (also what lets the slot be overwritten two barriers on)
        if (next) {
            // tile kt+1 must have landed; tile kt+2 (4 LDS-DMA instructions per wave) may stay in flight
            if (FULL || kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(TW_ABL) && (TW_ABL & 4))
            __builtin_amdgcn_s_barrier();
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = op(ah[i]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[ni][4 + mi] = mfma16<T>(b4[ni], a4[mi], acc[ni][4 + mi]);
                const int j = mi * 4 + ni;
                if (next) {
                    if (j == 1 && (FULL || kt + TW_STAGES - 1 < nk)) stage(kt + TW_STAGES - 1);  // into tile kt-1's slot: every wave finished it a barrier ago
                    if (j >= 2 && j < 6) bfn[j - 2] = frag(b_n[j - 2], std::integral_constant<int, noff>{});
                    if (j >= 6 && j < 10) aln[j - 6] = frag(a_n[j - 6], std::integral_constant<int, noff>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
#ifdef ARP_TN_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (nk > 0) {
        for (int s = 0; s < TW_STAGES - 1 && s < nk; ++s) stage(s);
        if (nk >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        Frag bf0[4], al0[4], bf1[4], al1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bf0[i] = frag(b_lo[i], std::integral_constant<int, 0>{});
#pragma unroll
        for (int i = 0; i < 4; ++i) al0[i] = frag(a_lo[i], std::integral_constant<int, 0>{});
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        int kt = 0;
        for (; kt + 3 + TW_STAGES - 1 < nk; kt += 4) {  // four steady-state tiles: each still has a tile to prefetch
            step(kt, bf0, al0, bf1, al1, I0{}, std::true_type{});
            step(kt + 1, bf1, al1, bf0, al0, I1{}, std::true_type{});
            step(kt + 2, bf0, al0, bf1, al1, I2{}, std::true_type{});
            step(kt + 3, bf1, al1, bf0, al0, I3{}, std::true_type{});
        }
        for (; kt < nk; kt += 4) {
            step(kt, bf0, al0, bf1, al1, I0{}, std::false_type{});
            if (kt + 1 < nk) step(kt + 1, bf1, al1, bf0, al0, I1{}, std::false_type{});
            if (kt + 2 < nk) step(kt + 2, bf0, al0, bf1, al1, I2{}, std::false_type{});
            if (kt + 3 < nk) step(kt + 3, bf1, al1, bf0, al0, I3{}, std::false_type{});
        }
    }
#undef TW_WAIT4
#ifdef ARP_TN_STAMPS
    if (tid == 0 && arp_tn_stamps) {
        arp_tn_stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - st0;
        arp_tn_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        arp_tn_stamps[blockIdx.x * 4 + 2] = nk;
    }
#endif
    float* out = g.out + (size_t)slice * g.slice_stride;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int m = m0 + wr * 128 + mi * 16 + fr, n = n0 + wc * 64 + ni * 16 + fg * 4;
            const f32x4_v a4 = acc[ni][mi];
            *reinterpret_cast<float4*>(out + (size_t)m * g.ldo + n) = make_float4(a4[0] * g.alpha, a4[1] * g.alpha, a4[2] * g.alpha, a4[3] * g.alpha);
        }
}

template <typename T> static int launch_impl(const GemmTnArgs& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % TN_BM || g.N % TN_BN || g.K % TN_BK || g.lda % 8 || g.ldb % 8 || g.ldo % 4 || g.ksplit < 1)
        return fail("gemm_tn: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K));
    if (g.tile256) {
        if (g.M % TW_BM || g.N % TW_BN || (g.ksplit > 1 && !g.slice_stride)) return fail("gemm_tn: the 256-tile kernel needs M, N multiples of 256");
        auto kern = gemm_tn256_kernel<T>;
        static bool attr256 = false;
        if (!attr256) {
            ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES));
            attr256 = true;
        }
        const int tiles = (g.M / TW_BM) * (g.N / TW_BN);
        // XCD placement when the slices divide over the eight XCDs and an XCD's share fits its 32 CUs
        const int spx = (g.ksplit % 8 == 0 && (g.ksplit / 8) * tiles <= 32 && g.xcd_slices) ? g.ksplit / 8 : 0;
        const int grid = spx ? 8 * spx * tiles : tiles * g.ksplit;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TW_THREADS), TW_LDS_BYTES, stream, g, tiles, spx);
        ARP_HIP_OK(hipGetLastError());
        return 0;
    }
    auto kern = gemm_tn_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((g.M / TN_BM) * (g.N / TN_BN), g.ksplit), dim3(TN_THREADS), TN_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}


// ---- "NN" GEMM: C[M,N] = sum_k A[m,k] . B[k,n] -- A K-contiguous (an activation gradient [rows, out]), B stored ROW-major with the contraction
// index as the row (a weight [out, in] AS IT LIES IN MEMORY): dX = dY . W without a transposed copy of W (the fine-tune head rebuilt 381 M
// elements of transposed weight shadows every step for its NT dX products: ft.refresh_shadows, 0.34 ms of 3.6).  A tiles as in gemm.h
// ([128 m][64 k] 128-B rows, 16-B chunk ^ (row & 7), ds_read_b128 fragments: lane group fg of k-step st holds k = 32 st + 8 fg + j), B tiles as in
// gemm_tn_kernel ([64 k][128 n] 256-B rows by LDS-DMA, fragments by the transposing read) -- with the k-slot mapping of A: the two reads of
// a B fragment take rows 32 st + 8 fg + {0..3} and + {4..7}, and the 32-B slot key is ((row >> 3) & 1) << 2 | (row & 3), so that the two
// 16-lane groups of a half-wave (rows 8 apart) still meet eight different slots.  Split-K over gridDim.y into f32 slabs; rows of A past M
// are clamped (never stored), N a multiple of 128, K a multiple of 64.
template <typename T>
__global__ __launch_bounds__(TN_THREADS) void gemm_nn_kernel(GemmTnArgs g) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int n_tiles = g.N / TN_BN, m_tiles = (g.M + TN_BM - 1) / TN_BM;
    const int tile = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    // tiles of one column block are neighbours (the m-tiles of a weight column panel share it through L2)
    const int m0 = (tile % m_tiles) * TN_BM, n0 = (tile / m_tiles) * TN_BN;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);

    int nk = g.K / TN_BK, kt0 = 0;
    if (g.ksplit > 1) {
        const int per = (nk + g.ksplit - 1) / g.ksplit;
        kt0 = blockIdx.y * per;
        nk = max(min(per, nk - kt0), 0);
    }
    // ---- LDS-DMA plans ------------------------------------------------------------------------------------------------------------
    // A: a piece = 8 rows x 128 B; wave w fills pieces w, w + 4, w + 8, w + 12
    const int arow = lane >> 3, achunk = (lane & 7) ^ arow;
    // B: a piece = 4 rows x 256 B; wave w fills pieces 4w .. 4w + 3
    const int prow = lane >> 4, pc = lane & 15;
    const T* srcA[4];
    const T* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int am = m0 + (i * 4 + wave) * 8 + arow;
        am = am < g.M ? am : g.M - 1;
        srcA[i] = A + (size_t)am * g.lda + (size_t)kt0 * TN_BK + achunk * 8;
        const int row = (wave * 4 + i) * 4 + prow;
        const int key = (((row >> 3) & 1) << 2) | (row & 3);
        const int lc = ((((pc >> 1) ^ key) << 1) | (pc & 1)) * 8;
        srcB[i] = B + (size_t)(kt0 * TN_BK + row) * g.ldb + n0 + lc;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * TN_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * TN_BK),
                                             (__attribute__((address_space(3))) void*)(base + (i * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (size_t)kt * TN_BK * g.ldb),
                                             (__attribute__((address_space(3))) void*)(base + TN_OP_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int trq = fr >> 2, trp = fr & 3;
    auto fragB = [&](const char* op, int blk16, int st) {
        const int r0 = 32 * st + 8 * fg + trq, r1 = r0 + 4;
        const int k0 = (((r0 >> 3) & 1) << 2) | (r0 & 3);  // = the key of r1 as well (bits 0, 1 and 3 agree)
        const tr_b64_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tr_b64_v*)(op + r0 * 256 + ((blk16 ^ k0) << 5) + trp * 8));
        const tr_b64_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tr_b64_v*)(op + r1 * 256 + ((blk16 ^ k0) << 5) + trp * 8));
        const u32x2_v l2 = __builtin_bit_cast(u32x2_v, lo), h2 = __builtin_bit_cast(u32x2_v, hi);
        return u32x4_v{l2[0], l2[1], h2[0], h2[1]};
    };
    f32x4_v acc[4][4];  // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    if (nk > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const char* a_op = smem + cur * TN_STAGE_BYTES;
            const char* b_op = a_op + TN_OP_BYTES;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4_v af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *reinterpret_cast<const u32x4_v*>(a_op + (wr * 64 + i * 16 + fr) * 128 + (((st * 4 + fg) ^ (fr & 7)) << 4));
                    bf[i] = fragB(b_op, wc * 4 + i, st);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16<T>(bf[ni], af[mi], acc[ni][mi]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float* out = g.out + (size_t)blockIdx.y * g.slice_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fg * 4;
            const f32x4_v a4 = acc[ni][mi];
            *reinterpret_cast<float4*>(out + (size_t)m * g.ldo + n) = make_float4(a4[0] * g.alpha, a4[1] * g.alpha, a4[2] * g.alpha, a4[3] * g.alpha);
        }
    }
}

template <typename T> static int launch_nn_impl(const GemmTnArgs& g, hipStream_t stream) {
    if (g.M <= 0) return 0;
    if (g.N <= 0 || g.N % TN_BN || g.K <= 0 || g.K % TN_BK || (g.lda & 7) || (g.ldb & 7) || (g.ldo & 3) || g.ksplit < 1)
        return fail("gemm_nn: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K));
    auto kern = gemm_nn_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
        attr_set = true;
    }
    const int tiles = ((g.M + TN_BM - 1) / TN_BM) * (g.N / TN_BN);
    hipLaunchKernelGGL(kern, dim3(tiles, g.ksplit), dim3(TN_THREADS), TN_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

int launch_gemm_nn(int tcode, const GemmTnArgs& g, hipStream_t stream) {
    if (tcode == 1) return launch_nn_impl<bf16_t>(g, stream);
    if (tcode == 2) return launch_nn_impl<f16_t>(g, stream);
    return fail("gemm_nn: 16-bit operand types only");
}

int launch_gemm_tn(int tcode, const GemmTnArgs& g, hipStream_t stream) {
    if (tcode == 1) return launch_impl<bf16_t>(g, stream);
    if (tcode == 2) return launch_impl<f16_t>(g, stream);
    return fail("gemm_tn: 16-bit operand types only");
}

}  // namespace arp

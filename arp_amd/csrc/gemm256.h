// 256x256-tile NT GEMM for the big ViT GEMMs:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Structure (written for CDNA4, one 512-thread workgroup per CU, 128 KiB LDS):
//   * 8 waves as 2 (M) x 4 (N); each wave owns a 128x64 output block = 2x2 "quadrants" of 64x32
//     (4x2 MFMA fragments of 16x16), 128 accumulator VGPRs.
//   * K is consumed in tiles of 128 bytes per row (64 bf16 / 32 f32).  A K-tile is processed in TWO PHASES: A = quadrants
//     (0,0) (0,1), B = quadrants (1,1) (1,0); the two W register sub-tiles are loaded once per K-tile.
//   * Every phase is [load segment | s_barrier | MFMA segment (32 MFMAs) | s_barrier].  The two wave groups (wr = 0 / 1; they
//     sit pairwise on the same SIMDs) run ONE BARRIER OUT OF STEP, so while one group issues its MFMAs the other issues
//     its LDS reads and LDS-DMA: the SIMD's matrix pipe always has a wave to serve.  (The first version had four phases
//     per K-tile; 30 % of the wave cycles were parked in barriers / waits, ARP_G2_TWO_PHASE=0 rebuilds it.)
//   * Operands reach LDS by global_load_lds_dwordx4 (LDS-DMA), 2 instructions per thread per "unit" (128 rows x 128 B =
//     16 KiB):  U0 = A rows of quadrant-row 0, U1 / U2 = W rows of quadrant-col 0 / 1, U3 = A rows of quadrant-row 1.
//     Units are issued in STEPS -- even step 2t = {U0,U1,U2} of K-tile t, odd step 2t+1 = {U3} of K-tile t -- three steps
//     ahead of their first use into a 2-deep ring of K-tile buffers; the only waits are COUNTED s_waitcnt vmcnt(8) (two
//     steps = 64 KiB stay in flight across every barrier).  Hazards (phase p reads step p, issues step p+3 over the region
//     of step p-1, waits for step p+1 before its first barrier) are argued in DESIGN.md section 5.
//   * 16-byte-chunk XOR swizzle applied on the LDS-DMA SOURCE address and on the ds_read_b128
//     address (the LDS-DMA destination is lane-linear), as in gemm.h.
//   * block -> tile map: XCD-contiguous ranges, then groups of GROUP_M tile-rows walked column by
//     column, so the ~32 tiles resident on one XCD form a compact 2-D patch that shares panels in
//     that XCD's L2.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace arp {

#ifndef ARP_G2_TWO_PHASE
#define ARP_G2_TWO_PHASE 1
#endif
// Overlapped drain of the 16-bit-output epilogue (see the kernel): MEASURED, NOT ENABLED.  With two tiles per workgroup
// (ARP_GEMM_TPW=2) the isolated c_fc launch went 286 -> 272 us (-5 %), one persistent workgroup per CU 286 -> 281 us, but the
// two-stream labelling pass went 91.7 k -> 90.0 k frames/s on the same box (workgroups twice as long interleave the two streams
// more coarsely), and the extra live state costs the one-tile path two scratch reloads per tile.  The window it opens (start-up +
// two phases, 2-3 us) is short of the ~8 us a round's 31 MB take to reach HBM when every CU drains at once: hiding that needs
// the stores on waves that issue no loads for the next tile's first K-tiles.  Build with
// -DARP_G2_OVERLAP_DRAIN=1 to get it back.
#ifndef ARP_G2_OVERLAP_DRAIN
#define ARP_G2_OVERLAP_DRAIN 0
#endif
// K-loop variants measured in round 2 with the standalone harness scripts/gemm256_bench.hip (10 s per build) and NOT kept: moving
// U1 / U2 of phase B's six LDS-DMA issues in between that phase's MFMAs (3-7 % slower on every shape, 4096^3 1250 -> 1172 TF),
// issuing the LDS-DMA ahead of the phase's fragment reads, dropping s_setprio around the MFMA segments (both within +-1.5 % noise).
// Round 3 (-DARP_G2_FINE stamps, profiles/r3_g256_fine.txt): per K-tile the loop takes ~2550 cycles against 2048 of MFMA time.  A load
// segment is 16 ds_read_b128 + 2 LDS-DMA (phase A, 410 cycles) or 8 + 6 (phase B, 402): a read costs ~21 issue cycles (all four waves
// of a group read at once: the LDS array serves 256 B/clk), an LDS-DMA ~40 -- the 2 + 6 split IS the balanced one.  Making it 4 + 4 (W's
// second unit in a three-slot ring of its own so that it can be issued a phase earlier) was built, bit-identical, and 5-10 % SLOWER on
// every shape (qkv 212 -> 233 us, 4096^3 1276 -> 1140 TF): phase A grew to 580 cycles.  The 32-MFMA segments themselves run 570-640
// cycles, not 512, beside the partner group's reads: the two waves of a SIMD share its vector issue port.
constexpr int G2_BM = 256, G2_BN = 256, G2_THREADS = 512;
constexpr int G2_BUF_BYTES = (G2_BM + G2_BN) * 128;  // one K-tile: 64 KiB
static_assert(256 * (256 * 2 + 16) >= 2 * G2_BUF_BYTES && 256 * (256 * 2 + 16) >= 128 * (256 * 4 + 16), "epilogue tile must cover the K-tile ring");
constexpr int G2_TILE_BYTES = 256 * (256 * 2 + 16);  // 132 KiB: 2 K-tile buffers (128 KiB) or the padded epilogue tile
// call sites >= 16 (the policy step's: arp_dt.hip SITE_DT) carry the masked epilogue (GemmArgs::mask)
constexpr bool G2_MASK_SITE(int site) { return site >= 16; }
constexpr int G2_RED_OFF = G2_TILE_BYTES + 1024;     // + this tile's 256 bias values, fetched while the K loop runs
constexpr int G2_LDS_BYTES = G2_RED_OFF + 8 * 256 * 4;  // + the eight waves' column sums of the masked epilogue (GemmArgs::colsum_part)
constexpr int G2_B_REGION = G2_BM * 128;             // W rows start here inside a buffer
constexpr int G2_GROUP_M = 8;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}
// units allowed to stay in flight -> counted wait (2 LDS-DMA instructions per unit per thread)
__device__ __forceinline__ void wait_units(int allow) {
    if (allow >= 3) wait_vmcnt<6>();
    else if (allow == 2) wait_vmcnt<4>();
    else if (allow == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}

#if defined(ARP_G2_STAMPS) || defined(ARP_G2_FINE) || defined(ARP_G2_CLOCK)
__device__ long long* arp_g2_stamps = nullptr;  // scripts/gemm256_bench.hip: per-tile, per-wave phase time stamps
#endif
// Register cap (round 3 experiment, OFF): at 2 x 240 registers a SIMD has 32 left, exactly one wave of the row-wise kernels
// (layernorm_kernel allocates 32), so the OTHER stream's HBM-bound LayerNorm could be resident on a CU beside a GEMM workgroup instead
// of waiting for the CU to come free.  amdgpu_num_vgpr counts per register FILE on gfx90a+ (the backend doubles it for the unified
// file): 120 = 240 registers in all, which the K loop fits without a spill (eight dwords of prologue / epilogue state spill in the
// 16-bit-output instances).  Measured with gemm256 and gemm2w both capped: 96.8 k frames/s against 98-101 k uncapped -- every GEMM site a
// few per cent slower (c_fc 5.50 vs 5.25 ms, c_proj 4.70 vs 4.45 ms of site time per step), LayerNorm's site time unchanged (0.92 ms): no
// co-residency gain to pay for it.  128 = uncapped.
#ifndef ARP_G2_MAX_VGPR
#define ARP_G2_MAX_VGPR 128
#endif
// MFMA shape of the 16-bit K loop (round 4 A/B, VERDICT r3 next #2a).  M32 = v_mfma_f32_32x32x16: the wave's 128x64 block is 2x2
// quadrants of 2x1 fragments of 32x32; a K-tile is four 16-deep steps; same LDS bytes per K-tile (a wave reads its 128 + 64 rows
// once either way) but 32 MFMAs of 32 cycles per K-tile instead of 64 of 16, each holding the SIMD's vector issue port 8 cycles:
// 256 instead of 512 port cycles per K-tile for the partner wave's reads and LDS-DMA to share.  The 16-byte chunk swizzle key is
// (row >> 1) & 7 there (with row & 7 the 32-row fragment read is two-way conflicted: rows r and r + 8 of a lane group meet).
// MEASURED AND REJECTED (profiles/r4_mfma32_ab.txt; both shapes in one process, interleaved rounds, random operands; results bit-identical
// to the 16x16x32 loop): 12-26 % SLOWER by wall on every shape (qkv 187 -> 224 us, c_fc 256 -> 303, c_proj 263 -> 302, 4096^3 1296 -> 1031 TF).
// Two separate effects.  (i) With fragment reads and LDS-DMA ablated (MFMAs + barriers only) the 32x32 K-tile takes FEWER cycles (1 992 -
// 2 030 against 2 115) but the launch is 7-9 % longer: the chip holds a lower clock on this shape (MI355X_MICROARCH, DVFS item 7).  (ii) In
// the full loop the phase-B load segment (8 reads + 6 LDS-DMA) takes 740-780 cycles beside the partner's 32x32 MFMAs where it takes 400
// beside 16x16 ones -- with the reads ablated and the DMA kept the gap is still there (4096^3 137 vs 99 us), with the DMA ablated it closes
// (103 vs 98): an LDS-DMA issue costs ~2.4x as much next to this shape; s_setprio off and the k-steps of a fragment back to back change nothing.
#ifndef ARP_G2_MFMA32
#define ARP_G2_MFMA32 0
#endif
#ifndef ARP_G2_MIXC_RT
#define ARP_G2_MIXC_RT 0
#endif
#ifndef ARP_G2_MIX_UNIFORM  // MIXC: per-lane DMA offsets collapsed into one register per operand (needs padded operand buffers); measured: MORE scratch traffic, off
#define ARP_G2_MIX_UNIFORM 0
#endif
#ifndef ARP_G2_ABL  // harness ablations: bit 0 = no fragment reads after the first K-tile, bit 1 = no LDS-DMA after the prologue, bit 2 = no residual read in the f32 epilogue,
                    // bit 3 = no bias / activation arithmetic in the 16-bit staged epilogue (conversion, staging and stores stay) (wrong results, timing only)
#define ARP_G2_ABL 0
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16_v;
template <typename T> __device__ __forceinline__ f32x16_v mfma32(u32x4_v a, u32x4_v b, f32x16_v c) {
    if constexpr (__is_same(T, f16_t))
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
}
// K-loop version KV (common.h: ARP_G2_KV, dma16_saddr): 1 = SADDR-form LDS-DMA statements + peeled steady state, bit-identical to 0.
// MIXC (T = f16 only): the trailing K-tiles of every row are e2m1 (fp4) and run on the scaled f8f6f4 MFMA (GemmArgs::mix_*): same LDS image, same
// fragment reads, same 4-register operand tuples, one 16x16x128 MFMA where the binary16 K-tile issues a 16x16x32 one -- the same matrix-pipe cycles
// for four times the k.  (An e4m3 variant -- twice the k on 8-register tuples -- was built first and lost to register allocation: DESIGN 10 #20.)
// CLK (round 6): the DIAGNOSTIC instance that measures the clock the chip holds under the pass (MI355X_MICROARCH 'DVFS give-back' item 6: in the real kernel no
// stamp executes).  One s_memtime / s_memrealtime pair at the head and one at the foot of the workgroup; thread 0 adds the two differences and 1 to
// GemmArgs::clock_acc[0..2] -- a buffer of its own that nothing else reads.  clock = sum d(s_memtime) / sum d(s_memrealtime) x 100 MHz, time-weighted over the
// workgroups.  bench.py runs the timed steps once more with c_fc on this instance (arp_clip_clock_probe) and prints whole_pass.clock_ghz.
template <typename T, typename OutT, int ACT, bool RESID, int SITE, bool M32 = (ARP_G2_MFMA32 != 0), int KV = ARP_G2_KV, bool MIXC = false, bool CLK = false>
__global__ __launch_bounds__(G2_THREADS, 2) __attribute__((amdgpu_num_vgpr(ARP_G2_MAX_VGPR))) void gemm256_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPB = 128 / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr bool W32 = M32 && sizeof(T) == 2;  // the 32x32x16 loop exists for the 16-bit operand types only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
#ifdef ARP_G2_CLOCK  // diagnostic build only: the clock this workgroup ran at = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH, DVFS item 6)
    const long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    [[maybe_unused]] unsigned long long dclk_c0 = 0, dclk_r0 = 0;
    if constexpr (CLK) {
        dclk_c0 = __builtin_amdgcn_s_memtime();
        dclk_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // (lgkmcnt(0) alone: a builtin-form stamp left pending ahead of the loop makes hipcc write lgkmcnt(0) for the loop's LDS waits)
    }

    // ---- persistent tile loop ---------------------------------------------------------------------------
    // The launcher starts at most one workgroup per CU; each walks tiles blockIdx.x, + gridDim.x, ...  Between
    // tiles the NEXT tile's first five units are put in flight as soon as the epilogue has released LDS, so the
    // first-load latency of a tile overlaps the drain of the previous tile's stores (a wave's loads and stores
    // share one in-order vmcnt, so nothing more than that can overlap inside one workgroup).
    const int n_tiles = (g.N + G2_BN - 1) / G2_BN;
    const int m_tiles = (g.M + G2_BM - 1) / G2_BM;
    const int total_tiles = m_tiles * n_tiles;
    const int group_m = g.group_m > 0 ? g.group_m : G2_GROUP_M;
    int m0 = 0, n0 = 0;
    auto tile_coords = [&](int tix) __attribute__((always_inline)) {
        int t = xcd_remap(tix, total_tiles);
        const int per_group = group_m * n_tiles;
        const int grp = t / per_group;
        const int first_m = grp * group_m;
        const int gsize = min(m_tiles - first_m, group_m);
        t -= grp * per_group;
        m0 = (first_m + t % gsize) * G2_BM;
        n0 = (t / gsize) * G2_BN;
    };

    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.W);

    // De-phase the CUs: every tile costs the same, so without this all 256 CUs hit their (HBM-write-bound)
    // epilogue in the same window and nobody computes meanwhile.  Only the first resident wave of workgroups
    // waits; later workgroups inherit their CU's phase.
    if (g.stagger_groups > 1 && blockIdx.x < 256) {
        const int grp = (blockIdx.x >> 3) % g.stagger_groups;
        if (grp) {
            const unsigned long long target = (unsigned long long)g.stagger_cycles * grp / g.stagger_groups;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < target) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- LDS-DMA plan: unit u in {0:A q-row 0, 1:W q-col 0, 2:W q-col 1, 3:A q-row 1}, 2 instr / thread
    const int srow = lane >> 3;
    const T* src[4][2];
    int dst[4][2];  // byte offset inside a K-tile buffer (wave-uniform)
    // KV = 1: global address = tile base (SGPR pair: first row of the tile, current K-tile) + off32 (per lane, bytes, fixed for the tile)
    uint32_t off32[4][2];
    const char* a_tile = nullptr;
    const char* w_tile = nullptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // MIXC: no per-lane row clamp -- the caller promises ceil(M / 256) * 256 readable A rows and N % 256 == 0 -- so the eight per-lane offsets collapse
    // into TWO (one per operand) plus wave-uniform row terms that go into the SGPR base (ARP_G2_MIX_UNIFORM: built for the e4m3 variant's 8-register tuples, off)
    uint32_t off_a = 0, off_w = 0;
    auto setup_src = [&]() __attribute__((always_inline)) {
        if constexpr (KV == 1) {
            a_tile = reinterpret_cast<const char*>(A + (size_t)m0 * g.lda);
            w_tile = reinterpret_cast<const char*>(W + (size_t)n0 * g.ldw);
        }
        if constexpr (MIXC && ARP_G2_MIX_UNIFORM) {
            off_a = (uint32_t)(((size_t)srow * g.lda + ((lane & 7) ^ srow) * EPC) * sizeof(T));
            off_w = (uint32_t)(((size_t)srow * g.ldw + ((lane & 7) ^ srow) * EPC) * sizeof(T));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int lr0 = (wave * 2 + i) * 8;
                    const bool isA = (u == 0 || u == 3);
                    const int q = (u == 0 || u == 1) ? 0 : 1;
                    const int row0 = isA ? ((lr0 >> 6) * 128 + q * 64 + (lr0 & 63)) : ((lr0 >> 5) * 64 + q * 32 + (lr0 & 31));
                    dst[u][i] = (isA ? 0 : G2_B_REGION) + row0 * 128;
                }
            return;
        }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr0 = (wave * 2 + i) * 8;  // first of the 8 unit-local rows this instruction fills
            const bool isA = (u == 0 || u == 3);
            const int q = (u == 0 || u == 1) ? 0 : 1;
            const int row0 = isA ? ((lr0 >> 6) * 128 + q * 64 + (lr0 & 63)) : ((lr0 >> 5) * 64 + q * 32 + (lr0 & 31));
            const int row = row0 + srow;
            const int schunk = (lane & 7) ^ (W32 ? ((row >> 1) & 7) : srow);
            dst[u][i] = (isA ? 0 : G2_B_REGION) + row0 * 128;
            if (isA) {
                int am = m0 + row;
                am = am < g.M ? am : g.M - 1;
                if constexpr (KV == 1) off32[u][i] = (uint32_t)(((size_t)(am - m0) * g.lda + schunk * EPC) * sizeof(T));
                else src[u][i] = A + (size_t)am * g.lda + schunk * EPC;
            } else {
                int wn = n0 + row;
                wn = wn < g.N ? wn : g.N - 1;
                if constexpr (KV == 1) off32[u][i] = (uint32_t)(((size_t)(wn - n0) * g.ldw + schunk * EPC) * sizeof(T));
                else src[u][i] = W + (size_t)wn * g.ldw + schunk * EPC;
            }
        }
    };
    // MIXC: unit U of K-tile tt; the unit's first row (wave-uniform) goes into the SGPR base, the lane offset is one register per operand
    auto issue_mix = [&](int tt, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        constexpr bool isA = (u == 0 || u == 3);
        constexpr int q = (u == 0 || u == 1) ? 0 : 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr0 = (wave * 2 + i) * 8;
            const int row0 = isA ? ((lr0 >> 6) * 128 + q * 64 + (lr0 & 63)) : ((lr0 >> 5) * 64 + q * 32 + (lr0 & 31));
            const char* sb = (isA ? a_tile + (size_t)row0 * g.lda * sizeof(T) : w_tile + (size_t)row0 * g.ldw * sizeof(T)) + (size_t)tt * 128;
            dma16_saddr(sb, isA ? off_a : off_w, lds0 + (tt & 1) * G2_BUF_BYTES + (isA ? 0 : G2_B_REGION) + row0 * 128);
        }
    };
    const int nk = g.K / EPB;
    const int G = 4 * nk;  // total units
    // issue unit index gi (tile gi>>2, unit U) if it exists; U is a compile-time constant per phase
    auto issue = [&](int gi, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        if ((ARP_G2_ABL & 2) && gi >= 12) return;
        if (gi < G) {
            const int tt = gi >> 2;
            if constexpr (MIXC && ARP_G2_MIX_UNIFORM) {
                issue_mix(tt, U);
            } else if constexpr (KV == 1) {
                const char* sb = ((u == 0 || u == 3) ? a_tile : w_tile) + (size_t)tt * 128;
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16_saddr(sb, off32[u][i], lds0 + (tt & 1) * G2_BUF_BYTES + dst[u][i]);
            } else {
            char* base = smem + (tt & 1) * G2_BUF_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[u][i] + (size_t)tt * EPB),
                                                 (__attribute__((address_space(3))) void*)(base + dst[u][i]), 16, 0, 0);
            }
        }
    };
    // KV = 1, steady state: unit U of K-tile tt, no existence test
    // (the last two K-tiles' bodies "issue" K-tiles nk and nk + 1, which do not exist: their source is clamped to the last real K-tile -- re-read, in bounds,
    //  never consumed -- while the ring slot stays the nominal one, so the steady state needs no existence tests and no tail)
    auto issue_ss = [&](int tt, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        if constexpr (MIXC && ARP_G2_MIX_UNIFORM) {
            issue_mix(tt, U);
            return;
        }
        if constexpr (MIXC) {
            // (MIXC runs EVERY K-tile in a steady-state body: the last two "issue" K-tiles nk and nk + 1, which do not exist.  Their LDS-DMAs are issued all the
            //  same -- the loop's counted waits count them -- but for lane 0 only, from the last real K-tile (in bounds), into the nominal, dead ring slot.
            //  Issued for all 64 lanes they cost the K = 768 products 2 of 15 K-tiles' worth of address-path time.)
            const char* sb = ((u == 0 || u == 3) ? a_tile : w_tile) + (size_t)min(tt, nk - 1) * 128;
            const unsigned long long lanes = tt < nk ? ~0ull : 1ull;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_saddr_masked(sb, off32[u][i], lds0 + (tt & 1) * G2_BUF_BYTES + dst[u][i], lanes);
            return;
        }
        const char* sb = ((u == 0 || u == 3) ? a_tile : w_tile) + (size_t)tt * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_saddr(sb, off32[u][i], lds0 + (tt & 1) * G2_BUF_BYTES + dst[u][i]);
    };
    // the tile's 256 bias values -> LDS (one LDS-DMA of wave 0)
    auto bias_dma = [&](int n_first) __attribute__((always_inline)) {
        int n = n_first + lane * 4;
        n = n + 4 <= g.N ? n : (g.N >= 4 ? g.N - 4 : 0);  // clamped at the ragged edge: those columns are never stored
        if constexpr (KV == 1) dma16_saddr(g.bias, (uint32_t)n * 4u, lds0 + G2_TILE_BYTES);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.bias + n),
                                              (__attribute__((address_space(3))) void*)(smem + G2_TILE_BYTES), 16, 0, 0);
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    using U3 = std::integral_constant<int, 3>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- fragment addressing ----------------------------------------------------------------------
    const int fr = W32 ? (lane & 31) : (lane & 15), fg = W32 ? (lane >> 5) : (lane >> 4);
    const int a_base = (wr * 128 + fr) * 128;
    const int b_base = G2_B_REGION + (wc * 64 + fr) * 128;
    const int skey = W32 ? ((lane >> 1) & 7) : (fr & 7);
    // 16x16x32: k-step ks reads chunk ks * 4 + fg; 32x32x16: step s reads chunk 2 s + fg
    const int coff0 = ((0 * 4 + fg) ^ skey) << 4;
    const int coff1 = ((W32 ? 2 + fg : 1 * 4 + fg) ^ skey) << 4;
    const int coff2 = ((4 + fg) ^ skey) << 4;  // W32 only
    const int coff3 = ((6 + fg) ^ skey) << 4;

    // accumulators: 16x16 fragments [mq][nq][ni][mi] (4 registers each) or 32x32 fragments [mq][nq][mi] (16 each); the epilogue
    // reads both through acc4(mq, nq, ni, mi) = four consecutive columns of one row:
    //   row = wr*128 + mq*64 + mi*RSTEP + rl,  col = wc*64 + nq*32 + ni*CSTEP + cl
    constexpr int NI = W32 ? 4 : 2, MI = W32 ? 2 : 4, RSTEP = 64 / MI, CSTEP = 32 / NI;
    const int rl = fr, cl = fg * 4;
    f32x4_v acc[W32 ? 1 : 2][2][2][4];  // [mq][nq][ni][mi]
    f32x16_v acc32[W32 ? 2 : 1][2][2];  // [mq][nq][mi]
    auto acc4 = [&](int mq, int nq, int ni, int mi) -> f32x4_v {
        if constexpr (W32) {
            const f32x16_v& a = acc32[mq][nq][mi];
            return f32x4_v{a[4 * ni], a[4 * ni + 1], a[4 * ni + 2], a[4 * ni + 3]};
        } else {
            return acc[mq][nq][ni][mi];
        }
    };
    u32x4_v areg[W32 ? 2 : 4][W32 ? 4 : 2];  // [mi][ks]   A sub-tile of the current quadrant-row
    u32x4_v breg[2][W32 ? 1 : 2][W32 ? 4 : 2];  // [nq][ni][ks]  both W sub-tiles of the K-tile stay in registers: quadrant (1,0) reuses
                            // sub-tile 0 without re-reading LDS, so every LDS region is last read >= 3 phases before
                            // the LDS-DMA that overwrites it is issued (WAR margin for the staggered wave groups)

    using F16T = std::false_type;  // K-tile kinds of a MIXC row: binary16 ...
    using F8T = std::true_type;    // ... or e2m1 (fp4).  Same fragment registers, same LDS reads; compile-time per K-tile BODY (no run-time choice inside the MFMA segments)
    int abl_kt = 0;
    auto load_a = [&](const char* buf, int mq, auto F8) __attribute__((always_inline)) {
        if ((ARP_G2_ABL & 1) && abl_kt > 0) return;

#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const char* p = buf + a_base + (mq * 64 + mi * RSTEP) * 128;
            areg[mi][0] = *reinterpret_cast<const u32x4_v*>(p + coff0);
            areg[mi][1] = *reinterpret_cast<const u32x4_v*>(p + coff1);
            if constexpr (W32) {
                areg[mi][2] = *reinterpret_cast<const u32x4_v*>(p + coff2);
                areg[mi][3] = *reinterpret_cast<const u32x4_v*>(p + coff3);
            }
        }
    };
    auto load_b = [&](const char* buf, auto NQ, auto F8) __attribute__((always_inline)) {
        constexpr int nq = decltype(NQ)::value;
        if ((ARP_G2_ABL & 1) && abl_kt > 0) return;
#pragma unroll
        for (int ni = 0; ni < (W32 ? 1 : 2); ++ni) {
            const char* p = buf + b_base + (nq * 32 + ni * 16) * 128;
            breg[nq][ni][0] = *reinterpret_cast<const u32x4_v*>(p + coff0);
            breg[nq][ni][1] = *reinterpret_cast<const u32x4_v*>(p + coff1);
            if constexpr (W32) {
                breg[nq][ni][2] = *reinterpret_cast<const u32x4_v*>(p + coff2);
                breg[nq][ni][3] = *reinterpret_cast<const u32x4_v*>(p + coff3);
            }
        }
    };
    int mix_scale = 0;  // MIXC: the e8m0 scale word of the current e2m1 K-tile (wave-uniform)
    int mix_sa = g.mix_sa, mix_sb = g.mix_sb;
    if constexpr (MIXC) {
        if (g.mix_sptr) {  // scales chosen on the device (uniform loads)
            mix_sa = F16C_X_SHIFT + g.mix_sptr[0];
            mix_sb = F16C_DX_SHIFT + g.mix_sptr[1];
        }
    }
    auto mfma_quadrant = [&](auto MQ, auto NQ, auto F8 = F16T{}) __attribute__((always_inline)) {
        constexpr int mq = decltype(MQ)::value, nq = decltype(NQ)::value;
        if constexpr (MIXC && sizeof(T) == 2 && !W32) {
            if constexpr (decltype(F8)::value) {  // an e2m1 K-tile: two 128-deep k-steps, the loop nest of the binary16 tile
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) acc[mq][nq][ni][mi] = mfma_fp4_scaled(breg[nq][ni][ks], areg[mi][ks], acc[mq][nq][ni][mi], mix_scale);
                return;
            }
        }
        if constexpr (W32) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc32[mq][nq][mi] = mfma32<T>(breg[nq][0][ks], areg[mi][ks], acc32[mq][nq][mi]);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (sizeof(T) == 2) {
                        acc[mq][nq][ni][mi] = mfma16<T>(breg[nq][ni][ks], areg[mi][ks], acc[mq][nq][ni][mi]);
                    } else if constexpr (sizeof(T) == 1) {
                        // fp8: the K-tile's 128 bytes per row are ONE 16x16x128 step; both register halves go into a single MFMA
                        if (ks == 0)
                            acc[mq][nq][ni][mi] = mfma_fp8(breg[nq][ni][0], breg[nq][ni][1], areg[mi][0], areg[mi][1], acc[mq][nq][ni][mi]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mq][nq][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(breg[nq][ni][ks][j]), __uint_as_float(areg[mi][ks][j]), acc[mq][nq][ni][mi], 0, 0, 0);
                    }
                }
        }
    };
    // one phase = [reads + LDS-DMA issue + counted wait] barrier [MFMAs] barrier
    auto phase_tail = [&](int ph) __attribute__((always_inline)) {
        int allow = G - 3 - ph;
        allow = allow > 3 ? 3 : allow;
        wait_units(allow);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](auto MQ, auto NQ) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        mfma_quadrant(MQ, NQ);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#if ARP_G2_TWO_PHASE
    // ---- two phases per K-tile (halves the barrier count: 30 % of the wave cycles sat in s_barrier / s_waitcnt with four) ----
    // Issue STEPS: even step 2t = units U0,U1,U2 of K-tile t (first read in phase A(t)), odd step 2t+1 = unit U3 of K-tile t
    // (first read in phase B(t)).  Phase p reads step p and issues step p+3 -- the buffer region of step p-1, read one phase
    // earlier by both wave groups -- then waits until step p+1 has landed (steps p+2, p+3 stay in flight: 8 LDS-DMA
    // instructions per thread) so that it is visible, after the barrier, to the reads of phase p+1.
    const int S2 = 2 * nk;
    auto issue_step = [&](int st) __attribute__((always_inline)) {
        const int t4 = (st >> 1) * 4;
        if (st & 1) {
            issue(t4 + 3, U3{});
        } else {
            issue(t4 + 0, U0{});
            issue(t4 + 1, U1{});
            issue(t4 + 2, U2{});
        }
    };
    auto step_cnt = [&](int st) __attribute__((always_inline)) { return st < S2 ? ((st & 1) ? 2 : 6) : 0; };
    auto wait_instr = [&](int n) __attribute__((always_inline)) {
        if (n >= 24) wait_vmcnt<24>();
        else if (n >= 8) wait_vmcnt<8>();
        else if (n >= 6) wait_vmcnt<6>();
        else if (n >= 2) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };
    // `young` = the previous tile's epilogue stores, issued AFTER this tile's steps 0..2 (overlapped epilogue, below): they are
    // younger than every step <= 2, so a wait for such a step may leave them in flight too (vmcnt counts in issue order)
    int young = 0;
    auto phase_tail2 = [&](int p) __attribute__((always_inline)) {
        wait_instr(step_cnt(p + 2) + step_cnt(p + 3) + (p + 1 <= 2 ? young : 0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this phase's fragment reads are done before any wave may overwrite them
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute2 = [&](auto MQ, auto NQA, auto NQB, auto F8 = F16T{}) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        mfma_quadrant(MQ, NQA, F8);
        mfma_quadrant(MQ, NQB, F8);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#endif
    bool pre_issued = false;
    bool pre_issued_bias = false;
    bool pre4 = false;  // this tile's steps 0..2 were issued by the previous tile's overlapped epilogue, AHEAD of that tile's 16 stores
    for (int tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
#ifdef ARP_G2_STAMPS
    long long st_[6];
    st_[0] = __builtin_amdgcn_s_memtime();
#define ARP_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define ARP_STAMP(i)
#endif
    tile_coords(tix);
    setup_src();
    // the tile's bias slice (256 floats) goes to LDS by one LDS-DMA of wave 0, issued ahead of the operand units (so every
    // counted wait covers it): read from global memory in the epilogue it would cost a memory latency with nothing to hide it
    float* bias_s = reinterpret_cast<float*>(smem + G2_TILE_BYTES);
    if (g.bias && wave == 0 && !pre_issued_bias) bias_dma(n0);
    if constexpr (W32) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 16; ++d) acc32[a][b][c][d] = 0.f;
    } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    }
#ifdef ARP_G2_FINE  // scripts/gemm256_bench.hip -DARP_G2_FINE: s_memtime around every segment of the middle K-tile
    long long f_[12];
    for (int i = 0; i < 12; ++i) f_[i] = 0;
#define ARP_FST(i) if (kt == (nk >> 1)) f_[i] = __builtin_amdgcn_s_memtime()
#else
#define ARP_FST(i)
#endif
#if ARP_G2_TWO_PHASE
    // ---- prologue: steps 0..2 in flight, step 0 landed and visible ---------------------------------------
    if (!pre_issued) {
        issue_step(0);
        issue_step(1);
        issue_step(2);
    }
    young = pre4 ? 16 : 0;
    wait_instr(step_cnt(1) + step_cnt(2) + young);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) {  // stagger: group 1 runs one barrier behind group 0
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);

    int kt_first = 0;
    if constexpr (KV == 1 && !ARP_G2_OVERLAP_DRAIN) {
        // ---- steady state: K-tiles 0 .. nk-3 issue steps 2kt+3 (U3 of K-tile kt+1) and 2kt+4 (U0,U1,U2 of K-tile kt+2), both of which
        // exist, and leave exactly 8 LDS-DMA instructions in flight at each counted wait
        auto tail_ss = [&]() __attribute__((always_inline)) {
            wait_vmcnt<8>();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) as the builtin: hipcc's own wait bookkeeping sees it (behind the asm form it re-waits inside the MFMA segment)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto ktile_ss = [&](int kt, auto BUF, auto F8) __attribute__((always_inline)) {  // one K-tile out of ring buffer BUF (compile-time: its fragment addresses are loop constants)
            const char* buf = smem + ((MIXC && RESID && ARP_G2_MIXC_RT) ? (kt & 1) : decltype(BUF)::value) * G2_BUF_BYTES;
            if constexpr (MIXC && decltype(F8)::value) mix_scale = 0x7f7f7f00 | (127 - (kt < g.mix_nk16 + g.mix_nkc_a ? mix_sa : mix_sb));
            load_a(buf, 0, F8);
            load_b(buf, I0{}, F8);
            load_b(buf, I1{}, F8);
            issue_ss(kt + 1, U3{});
            tail_ss();
            compute2(I0{}, I0{}, I1{}, F8);
            load_a(buf, 1, F8);
            issue_ss(kt + 2, U0{});
            issue_ss(kt + 2, U1{});
            issue_ss(kt + 2, U2{});
            tail_ss();
            compute2(I1{}, I1{}, I0{}, F8);
        };
        auto run_ss = [&](int kt_end, auto F8) __attribute__((always_inline)) {  // steady-state K-tiles [kt_first, kt_end), all of one kind
            if (kt_first < kt_end && (kt_first & 1)) {
                ktile_ss(kt_first, I1{}, F8);
                ++kt_first;
            }
            for (; kt_first + 1 < kt_end; kt_first += 2) {
                ktile_ss(kt_first, I0{}, F8);
                ktile_ss(kt_first + 1, I1{}, F8);
            }
            if (kt_first < kt_end) {
                ktile_ss(kt_first, I0{}, F8);
                ++kt_first;
            }
        };
        if constexpr (MIXC) {
            // ALL K-tiles run in the steady-state bodies (issue_ss clamps the two over-issued K-tiles' source) instead of a tail whose shrinking waits and
            // existence tests hipcc compiled into straight-line copies with accumulator spills between the MFMAs (3 of a K = 768 product's 15 K-tiles, each
            // behind a vmcnt(0)): 15.2 -> 12.5 ms per encoder-inside step
            run_ss(g.mix_nk16, F16T{});
            run_ss(nk, F8T{});
            kt_first = nk;
        } else {
            // (every K-tile in a steady-state body here too, the last two with over-issued LDS-DMAs, was measured and NOT kept: the tail of these instances
            //  compiles clean, and two more K-tiles' worth of LDS-DMA issue cost the K = 768 products the 2-3 % the KV = 1 loop had won: profiles/r5_kv_ab_harness.txt)
            run_ss(nk - 2, F16T{});
        }
    }
    // the last two K-tiles (and every K-tile of a KV = 0 build): issues may not exist, the counted waits shrink
    auto ktile_tail = [&](int kt, auto F8) __attribute__((always_inline)) {
        const char* buf = smem + (kt & 1) * G2_BUF_BYTES;
        const int p = 2 * kt;
        abl_kt = kt;
        if constexpr (MIXC && decltype(F8)::value) mix_scale = 0x7f7f7f00 | (127 - (kt < g.mix_nk16 + g.mix_nkc_a ? mix_sa : mix_sb));
        // phase A: quadrants (0,0) and (0,1)
        ARP_FST(0);
        load_a(buf, 0, F8);
        load_b(buf, I0{}, F8);
        load_b(buf, I1{}, F8);
        issue_step(p + 3);
        ARP_FST(1);
        phase_tail2(p);
        ARP_FST(2);
        compute2(I0{}, I0{}, I1{}, F8);
        // phase B: quadrants (1,1) and (1,0) -- the W sub-tiles are still in registers
        ARP_FST(4);
        load_a(buf, 1, F8);
        issue_step(p + 4);
        ARP_FST(5);
        phase_tail2(p + 1);
        ARP_FST(6);
        compute2(I1{}, I1{}, I0{}, F8);
        ARP_FST(8);
    };
    if constexpr (!MIXC) {
        for (int kt = kt_first; kt < nk; ++kt) ktile_tail(kt, F16T{});  // KV = 0 builds, and products of fewer than three K-tiles
    }
    if constexpr (MIXC) wait_vmcnt<0>();  // this wave's over-issued LDS-DMAs have landed ...
#else
    // ---- prologue: units 0..4 in flight, units 0 and 1 landed and visible ---------------------------
    if (!pre_issued) {
        issue(0, U0{});
        issue(1, U1{});
        issue(2, U2{});
        issue(3, U3{});
        issue(4, U0{});
    }
    {
        const int last = G - 1 < 4 ? G - 1 : 4;
        wait_units(last - 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) {  // stagger: group 1 runs one barrier behind group 0
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const char* buf = smem + (kt & 1) * G2_BUF_BYTES;
        const int ph = 4 * kt;
        // phase 1: quadrant (0,0)
        load_a(buf, 0, F16T{});
        load_b(buf, I0{}, F16T{});
        issue(ph + 5, U1{});
        phase_tail(ph);
        compute(I0{}, I0{});
        // phase 2: quadrant (0,1)
        load_b(buf, I1{}, F16T{});
        issue(ph + 6, U2{});
        phase_tail(ph + 1);
        compute(I0{}, I1{});
        // phase 3: quadrant (1,1)
        load_a(buf, 1, F16T{});
        issue(ph + 7, U3{});
        phase_tail(ph + 2);
        compute(I1{}, I1{});
        // phase 4: quadrant (1,0) -- operands already in registers
        issue(ph + 8, U0{});
        phase_tail(ph + 3);
        compute(I1{}, I0{});
    }
#endif
    if (wr == 0) {  // re-align the two groups
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MIXC) {
        // ... and, one barrier later, every wave's: the epilogue stages its tile over the ring, which an LDS-DMA still in flight would overwrite
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef ARP_G2_FINE
    if (arp_g2_stamps && (threadIdx.x & 63) == 0) {
        long long* d = arp_g2_stamps + ((size_t)tix * 8 + wave) * 16;
        for (int i = 0; i < 9; ++i) d[i] = f_[i];
    }
#endif

    ARP_STAMP(1);
    auto epilogue = [&]() {
    // ---- epilogue ------------------------------------------------------------------------------------
    OutT* out = static_cast<OutT*>(g.out);  // may alias g.resid (in-place residual add)
#ifdef ARP_G2_EXPERIMENT_SAME_TILE
    if (g.flags & 16) out -= (size_t)m0 * g.ldo + n0;  // every tile stores to tile (0,0): takes HBM write bandwidth out of the picture
#endif
    const bool vec_ok = ((g.N | g.ldo | g.ldr) & 3) == 0;
    if (g.flags & 1) {  // ablation: keep the accumulators live, store (almost) nothing
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < NI; ++c)
#pragma unroll
                    for (int d = 0; d < MI; ++d) {
                        const f32x4_v t = acc4(a, b, c, d);
                        sacc += t[0] + t[1] + t[2] + t[3];
                    }
        if (sacc == 12345.678f) Elem<OutT>::st(out, sacc);
        return;
    }
    // ---- staged epilogue: accumulators -> (bias, activation) -> LDS tile -> whole rows out --------------
    // After the K loop every LDS byte is dead, so the 256x256 output tile is transposed through LDS and
    // leaves the CU as 16-byte-per-lane stores of 512 B (bf16) / 1 KiB (f32) contiguous runs -- whole
    // 128-B lines -- instead of 32 eight-byte stores per lane into 32-B row fragments; the residual is
    // read the same way.  (Direct per-fragment stores measured 118 us of a 358 us c_fc launch.)
    const bool staged = vec_ok && ((g.N | g.ldo) & (sizeof(OutT) == 1 ? 15 : 7)) == 0 && !(g.flags & 2);
    if (staged) {
        // The thread's four bias fragments are read ONCE, ahead of the staging loop (they depend on the fragment column only).
        // Reading them per fragment inside the loop -- behind a per-fragment `n < N` branch -- put 32 dependent LDS round trips
        // and 64 exec-mask branches on the staging path: 6.0 k of a 45.8 k-cycle qkv tile (in-kernel s_memtime stamps,
        // scripts/gemm256_bench.hip -DARP_G2_STAMPS).  Columns past N hold the clamped load's finite values and are never stored.
        float4 bq[2][NI];
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bq[nq][ni] = g.bias ? *reinterpret_cast<const float4*>(bias_s + wc * 64 + nq * 32 + ni * CSTEP + cl) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (sizeof(OutT) == 1) {
            // fp8 output (the MLP's hidden activation): out_scale * act(alpha * acc + bias), 4 values per dword, 256-B rows out
            constexpr int RS8 = 256 + 16;
#pragma unroll
            for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wr * 128 + mq * 64 + mi * RSTEP + rl;
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int col = wc * 64 + nq * 32 + ni * CSTEP + cl;
                            const f32x4_v a4 = acc4(mq, nq, ni, mi);
                            const float4 b = bq[nq][ni];
                            float v[4] = {a4[0] * g.alpha + b.x, a4[1] * g.alpha + b.y, a4[2] * g.alpha + b.z, a4[3] * g.alpha + b.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, true>(v[j]) * g.out_scale;
                            *reinterpret_cast<uint32_t*>(smem + row * RS8 + col) = pack_fp8x4(v[0], v[1], v[2], v[3]);
                        }
                }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int r = it * 32 + wave * 4 + (lane >> 4);
                const int m = m0 + r, n = n0 + (lane & 15) * 16;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<u32x4_v*>(reinterpret_cast<char*>(out) + (size_t)m * g.ldo + n) = *reinterpret_cast<const u32x4_v*>(smem + r * RS8 + (lane & 15) * 16);
            }
        } else if constexpr (sizeof(OutT) == 2) {
            constexpr int RS = 256 * 2 + 16;  // +16 B pad: fragment rows land on different banks
            auto stage16 = [&](auto LN) {  // LN: the folded-LayerNorm correction is compiled out of the ordinary path
                // Folded LayerNorm (gemm.h): the tile's 256 (mean, rstd) pairs are formed ONCE, by 256 threads, each from its row's partial sums requested in one
                // unconditional batch, and handed over in LDS; the column vector c is read once per thread like the bias.  (Rounds 1-4 called ln_row_stats per
                // fragment row -- a runtime loop of dependent loads, eight times per thread -- and read c behind a per-fragment `n < N` branch: ~100 serialised
                // loads per thread and tile, the 5.5 ms of site time that made the fold lose 16 % in profiles/r4_lnfold_ab.txt.)
                float* lnst = reinterpret_cast<float*>(smem + G2_RED_OFF);
                float4 cq[LN.value ? 2 : 1][LN.value ? NI : 1];
                if constexpr (LN.value) {
                    if (tid < 256) {
                        const float2* st = reinterpret_cast<const float2*>(g.ln_stats + (size_t)min(m0 + tid, g.M - 1) * g.ln_parts * 2);
                        float2 pr[8];
#pragma unroll
                        for (int p = 0; p < 8; ++p) pr[p] = st[min(p, g.ln_parts - 1)];
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if (p < g.ln_parts) { s += pr[p].x; q += pr[p].y; }
                        const float mu = s * g.ln_inv_d;
                        lnst[2 * tid] = mu;
                        lnst[2 * tid + 1] = 1.0f / sqrtf(fmaxf(q * g.ln_inv_d - mu * mu, 0.f) + g.ln_eps);
                    }
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            cq[nq][ni] = *reinterpret_cast<const float4*>(g.ln_c + min(n0 + wc * 64 + nq * 32 + ni * CSTEP + cl, g.N - 4));
                    __syncthreads();
                }
#pragma unroll
                for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int row = wr * 128 + mq * 64 + mi * RSTEP + rl;
                        float mu = 0.f, rs = 1.f;
                        if constexpr (LN.value) { mu = lnst[2 * row]; rs = lnst[2 * row + 1]; }
#pragma unroll
                        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const int col = wc * 64 + nq * 32 + ni * CSTEP + cl;
                                const f32x4_v a4 = acc4(mq, nq, ni, mi);
                                float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                                if constexpr (sizeof(T) == 1) { v[0] *= g.alpha; v[1] *= g.alpha; v[2] *= g.alpha; v[3] *= g.alpha; }
                                if constexpr (LN.value) {
                                    const float4 c4 = cq[LN.value ? nq : 0][LN.value ? ni : 0];
                                    v[0] = rs * (v[0] - mu * c4.x); v[1] = rs * (v[1] - mu * c4.y);
                                    v[2] = rs * (v[2] - mu * c4.z); v[3] = rs * (v[3] - mu * c4.w);
                                }
                                if constexpr ((ARP_G2_ABL & 8) == 0) {
                                    const float4 b = bq[nq][ni];
                                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                                    apply_act4<ACT, sizeof(T) <= 2>(v);
                                }
                                *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
                                if constexpr (MIXC && __is_same(OutT, f16_t)) {
                                    if (g.dx4_out && m0 + row < g.M && n0 + col < g.N) {  // the dx4 segment of the next product's operand row (GemmArgs::dx4_out)
                                        constexpr float sdx = (float)(1 << F16C_DX_SHIFT);
                                        *reinterpret_cast<uint16_t*>(static_cast<char*>(g.dx4_out) + (size_t)(m0 + row) * g.ldxb + ((n0 + col) >> 1)) =
                                            pack_fp4x4((v[0] - h2f(f2h(v[0]))) * sdx, (v[1] - h2f(f2h(v[1]))) * sdx, (v[2] - h2f(f2h(v[2]))) * sdx, (v[3] - h2f(f2h(v[3]))) * sdx);
                                    }
                                }
                            }
                    }
            };
            if constexpr (G2_MASK_SITE(SITE)) {
                stage16(std::false_type{});  // (the policy step's instances carry the masked epilogue instead of the folded LayerNorm: the launcher refuses ln_stats there)
            } else {
                if (g.ln_stats) stage16(std::true_type{});
                else stage16(std::false_type{});
            }
#if ARP_G2_TWO_PHASE && ARP_G2_OVERLAP_DRAIN
            // ---- overlapped drain (a workgroup that has another tile to do; interior tiles only) --------------------------
            // The tile's bytes move LDS -> registers (64 VGPRs: the accumulators are dead), the barrier releases LDS, the NEXT
            // tile's steps 0..2 are put in flight, and only then are the 16 stores per thread issued.  vmcnt counts in issue
            // order, so the stores are YOUNGER than those steps: the next tile's first waits (prologue, phases 0 and 1) leave them
            // in flight (`young`), and the HBM-write-bound drain of this tile overlaps the next tile's start-up and first two phases
            // instead of standing between two K-loops.  Every thread must issue exactly 16 store instructions for that
            // count to be safe, hence interior tiles only (no lane, hence no wave, skips a store).
            if (g.ovl && tix + (int)gridDim.x < total_tiles && m0 + G2_BM <= g.M && n0 + G2_BN <= g.N && nk >= 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                u32x4_v regs[16];
#pragma unroll
                for (int it = 0; it < 16; ++it)
                    regs[it] = *reinterpret_cast<const u32x4_v*>(smem + (it * 16 + wave * 2 + (lane >> 5)) * RS + (lane & 31) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // every wave holds its share of the tile: LDS is free
                __builtin_amdgcn_sched_barrier(0);
                OutT* obase = out + (size_t)(m0 + wave * 2 + (lane >> 5)) * g.ldo + n0 + (lane & 31) * 8;
                tile_coords(tix + gridDim.x);
                setup_src();
                if (g.bias && wave == 0) bias_dma(n0);
                issue_step(0);
                issue_step(1);
                issue_step(2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 16; ++it) *reinterpret_cast<u32x4_v*>(obase + (size_t)it * 16 * g.ldo) = regs[it];
                __builtin_amdgcn_sched_barrier(0);
                pre_issued = pre_issued_bias = pre4 = true;
                return;
            }
#endif
            ARP_STAMP(2);
            __syncthreads();
            ARP_STAMP(3);
            // (Round 5 tried the mask's sixteen row segments as one or two unconditional batches -- inside the branch below each load is a branch + vmcnt(0) of its own.
            //  No gain by wall, and the batch's addresses, formed at the head of the kernel, cost these instances 39 spilled registers: reverted.)
            if (G2_MASK_SITE(SITE) && g.mask) {  // compiled into the policy step's instances only: the ViT kernels keep their register budget
                // out = (mask > 0) ? value : 0 on whole 16-byte row segments, plus the tile's column sums of what was stored (the bias
                // gradient of the layer whose ReLU this is): one more 16-byte read per store instead of a separate pass over both tensors
                const OutT* __restrict__ mk = static_cast<const OutT*>(g.mask);
                float colacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int r = it * 16 + wave * 2 + (lane >> 5);
                    const int m = m0 + r, n = n0 + (lane & 31) * 8;
                    if (m < g.M && n < g.N) {
                        u32x4_v v = *reinterpret_cast<const u32x4_v*>(smem + r * RS + (lane & 31) * 16);
                        const u32x4_v mv = *reinterpret_cast<const u32x4_v*>(mk + (size_t)m * g.ldm + n);
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            // per 16-bit half: keep = magnitude non-zero and sign clear (bf16 and f16 alike; a NaN mask keeps, as NaN > 0
                            // never occurs behind a ReLU)
                            const uint32_t mag = mv[w] & 0x7fff7fffu;
                            const uint32_t nz = ((mag & 0xffffu) ? 0xffffu : 0u) | ((mag >> 16) ? 0xffff0000u : 0u);
                            const uint32_t pos = ((mv[w] & 0x8000u) ? 0u : 0xffffu) | ((mv[w] & 0x80000000u) ? 0u : 0xffff0000u);
                            const uint32_t kept = v[w] & nz & pos;
                            v[w] = kept;
                            OutT pr[2];
                            memcpy(pr, &kept, 4);
                            colacc[2 * w] += Elem<OutT>::ld(&pr[0]);
                            colacc[2 * w + 1] += Elem<OutT>::ld(&pr[1]);
                        }
                        *reinterpret_cast<u32x4_v*>(out + (size_t)m * g.ldo + n) = v;
                    }
                }
                if (g.colsum_part) {
                    float* red = reinterpret_cast<float*>(smem + G2_RED_OFF);
#pragma unroll
                    for (int e = 0; e < 8; ++e) colacc[e] += __shfl_xor(colacc[e], 32, 64);
                    if (lane < 32) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) red[wave * 256 + lane * 8 + e] = colacc[e];
                    }
                    __syncthreads();
                    if (tid < 256 && n0 + tid < g.N) {
                        float t = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) t += red[w * 256 + tid];
                        g.colsum_part[(size_t)(m0 / G2_BM) * g.N + n0 + tid] = t;
                    }
                }
                return;
            }
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int r = it * 16 + wave * 2 + (lane >> 5);
                const int m = m0 + r, n = n0 + (lane & 31) * 8;
                if (m < g.M && n < g.N) {
                    const u32x4_v v = *reinterpret_cast<const u32x4_v*>(smem + r * RS + (lane & 31) * 16);
                    if (g.flags & 4) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_v*>(out + (size_t)m * g.ldo + n));
                    else *reinterpret_cast<u32x4_v*>(out + (size_t)m * g.ldo + n) = v;
                    if constexpr (MIXC && __is_same(OutT, f16_t)) {
                        // ARP_MODE_F16C: the e2m1 segment of the next GEMM's [hi | x4] operand row, from the rounded tile (GemmArgs::x8_shift)
                        if (g.x8_shift >= 0) {
                            // v_cvt_scalef32_pk_fp4_f16 converts a PAIR of binary16 values divided by its scale operand (scripts/fp4_cvt_probe.hip: scale 0.5 doubles)
                            // straight from the staged tile's packed words: four instructions per eight values where unpack + multiply + the f32 form took twenty
                            // (same nibbles: a power-of-two scale is exact, the value is rounded once either way)
                            const float inv = 1.0f / (float)(1 << g.x8_shift);
                            // The four words go through SCALARS first: __builtin_bit_cast(f16x2_v, v[i]) on an ELEMENT of an ext-vector compiles (hipcc 7.0, -O3) to
                            // a read of element 0 whatever i is -- rounds 5 and 6 stored the first pair's two codes four times over (every x4 segment written by an
                            // epilogue: c_fc -> c_proj in the f16c encoder, fc1 -> fc2 in the corrected adapter), so the x4 . dW4 correction of the product behind it
                            // added noise of the size it was meant to remove.  Found with scripts/adapter_chain_probe.py (arp_dt_debug_read), pinned by
                            // tests/test_policy_gpu.py::test_adapter_operand_rows_hold_what_the_products_assume.
                            const uint32_t e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                            uint32_t w4 = 0;
                            w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e0), inv, 0);
                            w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e1), inv, 1);
                            w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e2), inv, 2);
                            w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e3), inv, 3);
                            *reinterpret_cast<uint32_t*>(static_cast<char*>(g.xb_out) + (size_t)m * g.ldxb + (n >> 1)) = w4;
                        }
                    }
                }
            }
        } else {
            constexpr int RSF = 256 * 4 + 16;
            // A pass's 16 residual rows per thread are requested BEFORE its tile is staged: all in flight at once instead of four dependent rounds
            // of load -> add -> store (the operand registers are dead by now).  Round 4: pass 1's rows are requested before pass 0's STORES are
            // issued -- a wave's loads and stores retire through one in-order counter, so requested behind those 16 stores the rows could not be
            // used before every one of them had completed (the write latency of a round in which all CUs drain together); requested ahead of them
            // they have been in flight for the whole of pass 0's copy-out.  (ARP_G2_RES_EARLY=0: round 3's order.)
#ifndef ARP_G2_RES_EARLY
#define ARP_G2_RES_EARLY 1
#endif
            float4 rres[2][16];
            auto load_res = [&](int p) {
                if constexpr (RESID) {
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        const int lr = it * 8 + wave;
                        const int m = m0 + (lr >> 6) * 128 + p * 64 + (lr & 63), n = n0 + lane * 4;
                        // UNCONDITIONAL, from a clamped address (rows / columns outside the matrix re-read its last row / last four columns; their values are never
                        // stored): behind a `m < M && n < N ? load : 0` hipcc branches around every load and waits vmcnt(0) at each join -- the sixteen "in flight
                        // together" were sixteen dependent round trips (round 5, found in the ISA; N % 8 == 0 on this path)
                        if constexpr ((ARP_G2_ABL & 4) == 0) rres[p][it] = *reinterpret_cast<const float4*>(g.resid + (size_t)min(m, g.M - 1) * g.ldr + min(n, g.N - 4));
                        else rres[p][it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            };
            load_res(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p) __syncthreads();
                if (p && !ARP_G2_RES_EARLY) load_res(1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int lrow = wr * 64 + mi * RSTEP + rl;
                            const int col = wc * 64 + nq * 32 + ni * CSTEP + cl;
                            const f32x4_v a4 = acc4(p, nq, ni, mi);
                            float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                            if constexpr (sizeof(T) == 1) { v[0] *= g.alpha; v[1] *= g.alpha; v[2] *= g.alpha; v[3] *= g.alpha; }
                            const float4 b = bq[nq][ni];
                            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) <= 2>(v[j]);
                            *reinterpret_cast<float4*>(smem + lrow * RSF + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                __syncthreads();
                if (p == 0 && ARP_G2_RES_EARLY) load_res(1);
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int lr = it * 8 + wave;
                    const int m = m0 + (lr >> 6) * 128 + p * 64 + (lr & 63), n = n0 + lane * 4;
                    const bool ok = m < g.M && n < g.N;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) {
                        v = *reinterpret_cast<const float4*>(smem + lr * RSF + lane * 16);
                        if constexpr (RESID) {
                            const float4 r = rres[p][it];
                            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                        }
                        if (g.out) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * g.ldo + n) = v;
                        if (g.xb_out) {
                            if (g.split3) store_split3(static_cast<T*>(g.xb_out) + (size_t)m * g.ldxb + n, (size_t)g.N, v.x, v.y, v.z, v.w);
                            else store4(static_cast<T*>(g.xb_out) + (size_t)m * g.ldxb + n, v.x, v.y, v.z, v.w);
                        }
                    }
                    if (g.stats_out) {  // folded-LayerNorm producer: one 128-column segment per 32-lane half
                        const float s = half_wave_sum((v.x + v.y) + (v.z + v.w));
                        const float q = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
                        if ((lane & 31) == 0 && ok) {
                            float* st = g.stats_out + ((size_t)m * (g.N >> 7) + ((n0 >> 7) + (lane >> 5))) * 2;
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
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wr * 128 + mq * 64 + mi * RSTEP + rl;
            if (m >= g.M) continue;
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + wc * 64 + nq * 32 + ni * CSTEP + cl;
                    if (n >= g.N) continue;
                    const f32x4_v a4 = acc4(mq, nq, ni, mi);
                    float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                    if constexpr (sizeof(T) == 1) { v[0] *= g.alpha; v[1] *= g.alpha; v[2] *= g.alpha; v[3] *= g.alpha; }
                    if (vec_ok) {
                        if (g.bias) {
                            const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) <= 2>(v[j]);
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
                            x = apply_act<ACT, sizeof(T) <= 2>(x);
                            if constexpr (RESID) x += g.resid[(size_t)m * g.ldr + n + j];
                            Elem<OutT>::st(out + (size_t)m * g.ldo + n + j, x);
                        }
                    }
                }
        }
    };
    pre_issued = false;
    pre_issued_bias = false;
    pre4 = false;
    epilogue();
#ifdef ARP_G2_STAMPS
    ARP_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ARP_STAMP(5);
    if (arp_g2_stamps && (threadIdx.x & 63) == 0) {
        long long* d = arp_g2_stamps + ((size_t)tix * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) d[i] = st_[i];
    }
#endif
    if (!pre_issued && tix + (int)gridDim.x < total_tiles) {
        __syncthreads();  // every wave has finished reading the epilogue tile out of LDS
        tile_coords(tix + gridDim.x);
        setup_src();
        if (g.bias && wave == 0) bias_dma(n0);
        pre_issued_bias = true;
#if ARP_G2_TWO_PHASE
        issue_step(0);
        issue_step(1);
        issue_step(2);
#else
        issue(0, U0{});
        issue(1, U1{});
        issue(2, U2{});
        issue(3, U3{});
        issue(4, U0{});
#endif
        pre_issued = true;
    }
    if constexpr (MIXC) break;  // one tile per workgroup (the launcher never makes these instances persistent): nothing of the K loop's state is then live
                                // across the epilogue, where hipcc would spill it -- and reload it in front of the next tile's loop, behind a vmcnt(0)
    }  // tile loop
#ifdef ARP_G2_CLOCK
    if (arp_g2_stamps && threadIdx.x == 0) {
        arp_g2_stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_c0;
        arp_g2_stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
    if constexpr (CLK) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (g.clock_acc && threadIdx.x == 0) {
            atomicAdd(g.clock_acc, c1 - dclk_c0);
            atomicAdd(g.clock_acc + 1, r1 - dclk_r0);
            atomicAdd(g.clock_acc + 2, 1ull);
        }
    }
}

template <typename T, typename OutT, int ACT, bool RESID, int SITE, bool M32 = (ARP_G2_MFMA32 != 0), int KV = ARP_G2_KV, bool MIXC = false, bool CLK = false>
inline int launch_gemm256_nt(const GemmArgs& g, hipStream_t stream) {
    constexpr int EPB = 128 / (int)sizeof(T);
    if (g.M <= 0) return 0;
    if (g.N <= 0 || g.K % EPB != 0 || g.K <= 0 || g.lda % (16 / (int)sizeof(T)) != 0 || g.ldw % (16 / (int)sizeof(T)) != 0)
        return fail("gemm256_nt: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) +
                    " K=" + std::to_string(g.K));
    if (g.ln_stats && G2_MASK_SITE(SITE)) return fail("gemm256_nt: no folded-LayerNorm epilogue in the masked-epilogue instances");
    if (g.ln_stats && (g.ln_parts < 1 || g.ln_parts > 8)) return fail("gemm256_nt: the folded-LayerNorm epilogue holds at most 8 partial sums per row (width <= 1024)");
    if (g.mask && (!G2_MASK_SITE(SITE) || sizeof(OutT) != 2 || ((g.N | g.ldo | g.ldm | g.ldr) & 7) || (g.flags & 3)))
        return fail("gemm256_nt: the masked epilogue needs a 16-bit output and N, ldo, ldm multiples of 8");
    // KV = 1 addresses a tile's rows by 32-bit byte offsets from the tile's first row
    if (KV == 1 && ((size_t)G2_BM * g.lda * sizeof(T) >= (1ull << 32) || (size_t)G2_BN * g.ldw * sizeof(T) >= (1ull << 32)))
        return fail("gemm256_nt: row stride too large for the 32-bit tile offsets of the KV = 1 K loop");
    if (MIXC && (sizeof(T) != 2 || g.mix_nk16 <= 0 || 64 * (g.mix_nk16 + g.mix_nkc_a) > g.K || g.K / 64 - g.mix_nk16 < 2 || !KV || (ARP_G2_MIX_UNIFORM && (g.N % G2_BN || !(g.flags & 32))) || g.mix_sa < 0 || g.mix_sa > 120 || g.mix_sb < 0 || g.mix_sb > 120))
        return fail("gemm256_nt: bad mixed binary16 / e2m1 K-tile plan (needs >= 2 trailing fp4 K-tiles, i.e. Kc >= 512 with one correction segment)");
    auto kern = gemm256_nt_kernel<T, OutT, ACT, RESID, SITE, M32, KV, MIXC, CLK>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G2_LDS_BYTES));
        attr_set = true;
    }
    const int m_tiles = (g.M + G2_BM - 1) / G2_BM;
    const int n_tiles = (g.N + G2_BN - 1) / G2_BN;
    int grid = m_tiles * n_tiles;
    static int persist = -1, n_cu = 0;
    if (persist < 0) {
        const char* e = getenv("ARP_GEMM_PERSIST");
        persist = e ? atoi(e) : 0;  // measured equal to one workgroup per tile (c_fc +5 %, qkv -4 %, whole pass 79.0 k vs 79.5 k frames/s)
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0 || (n_cu & 7)) n_cu = 256;
    }
    // (the MIXC instances do ONE tile per workgroup -- `if constexpr (MIXC) break` at the foot of the kernel's tile loop -- so their grid is never cut down:
    //  a persistent or tpw > 1 grid would leave every tile past the first `grid` unwritten, silently)
    if (!MIXC && persist && grid > n_cu) grid = n_cu;  // one workgroup per CU walks the tiles
    // 16-bit-output GEMMs: each workgroup does `tpw` tiles (tix, tix + grid, ...) so that all but its last epilogue drain under
    // the next tile's first phases (overlapped drain in the kernel).  tpw = 2 adds no tile-quantisation (c_fc: 2400 tiles = 9.4 -> 10
    // rounds either way) and keeps two streams interleaving at a granularity of two tiles.
    static int tpw = -1;
    if (tpw < 0) {
        const char* e = getenv("ARP_GEMM_TPW");
        tpw = e ? atoi(e) : 1;
        if (tpw < 1) tpw = 1;
    }
    GemmArgs ga = g;
    static int group_env = -1;  // ARP_GEMM_GROUP_M: tile-rows per L2 group of the block -> tile walk (0 = the default, G2_GROUP_M)
    if (group_env < 0) {
        const char* e = getenv("ARP_GEMM_GROUP_M");
        group_env = e ? atoi(e) : 0;
    }
    if (group_env > 0 && ga.group_m == 0) ga.group_m = group_env;
#if ARP_G2_OVERLAP_DRAIN
    if (!MIXC && sizeof(OutT) == 2 && tpw > 1 && !persist && grid >= 2 * n_cu) {
        grid = (grid + tpw - 1) / tpw;
        ga.ovl = 1;
    }
    if (!MIXC && sizeof(OutT) == 2 && persist) ga.ovl = 1;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G2_THREADS), G2_LDS_BYTES, stream, ga);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp
#include "gemm2w.h"
namespace arp {

// force: 0 = auto, 1 = 128x128 (2 WG/CU, simple double buffer), 2 = 256x256 (1 WG/CU, two-phase pipelined),
// 3 = 128x192 two-workgroups-per-CU kernel (gemm2w.h; 16-bit operands, instantiated epilogues only -- anything else falls
// through to auto).  Auto: a big-tile kernel when the grid fills the chip.
template <typename T, typename OutT, int ACT, bool RESID, int SITE>
inline int launch_gemm_auto(const GemmArgs& g, hipStream_t stream, int force = 0) {
    const long tiles256 = (long)((g.M + G2_BM - 1) / G2_BM) * ((g.N + G2_BN - 1) / G2_BN);
    if constexpr (sizeof(T) == 2) {
        if (force == 3 && gemm2w_has(__is_same(T, bf16_t) ? 1 : 2, sizeof(OutT) == 4 ? 1 : 0, ACT, RESID ? 1 : 0) && g.K % 64 == 0 && !(g.N & 7) &&
            !(g.ldo & 7) && !(RESID && (g.ldr & 3)) && !g.ln_stats && !g.stats_out && !g.xb_out && g.ksplit <= 1 && g.alpha == 1.f)
            return launch_gemm2w<T, OutT, ACT, RESID>(g, stream);
    }
    if (g.alpha != 1.f) return launch_gemm_nt<T, OutT, ACT, RESID, SITE>(g, stream);  // the only kernel with the alpha epilogue
    if (force == 1) return launch_gemm_nt<T, OutT, ACT, RESID, SITE>(g, stream);
    if (force == 2 || tiles256 >= 192) {
        // Tile rounds (round 5).  A product of a FEW rounds of 256 x 256 tiles whose last round is mostly empty -- c_proj: 600 tiles on 256 CUs = 2.34 rounds,
        // 300 = 1.17 for a 512-frame part -- runs its full rounds here and the rest of its ROWS on the 128 x 128 kernel (two workgroups per CU, a quarter of
        // the tile: 180-360 small tiles fill the chip where 44-88 big ones leave two thirds of it idle).  Same MFMA, same k order, same epilogue arithmetic:
        // the two kernels agree bit for bit (scripts/splitm_bench.hip checks the checksum on every shape), so a row's result does not depend on which one
        // computed it.  Measured there: c_proj 293 -> 278 us per 1 024 frames, 154.5 -> 134.4 us per 512-frame part (what the two-stream pass launches),
        // out_proj on this kernel 113 -> 104; products of many rounds (c_fc: 9.4, qkv) LOSE 2-16 % -- the dispatcher already balances them -- hence the gate.
        static const bool splitm = [] { const char* e = getenv("ARP_GEMM_SPLITM"); return e && atoi(e) != 0; }();
        const int n_t = (g.N + G2_BN - 1) / G2_BN, m_t = (g.M + G2_BM - 1) / G2_BM;
        const long rounds = tiles256 / 256, rem_tiles = tiles256 - rounds * 256;
        const int full_m = (int)(rounds * 256 / n_t);  // tile rows of the full rounds
        if (splitm && rounds >= 1 && rounds <= 3 && rem_tiles > 8 && rem_tiles <= 128 && full_m < m_t && g.ksplit <= 1 && !g.ln_stats && !g.stats_out && !g.xb_out &&
            !g.mask && !g.colsum_part && !g.adam_p && !(g.flags & 3) && g.mix_nk16 == 0) {
            GemmArgs a = g, b = g;
            a.M = full_m * G2_BM;
            b.M = g.M - a.M;
            b.A = static_cast<const T*>(g.A) + (size_t)a.M * g.lda;
            b.out = static_cast<OutT*>(g.out) + (size_t)a.M * g.ldo;
            if (g.resid) b.resid = g.resid + (size_t)a.M * g.ldr;
            if (int rc = launch_gemm256_nt<T, OutT, ACT, RESID, SITE>(a, stream)) return rc;
            return launch_gemm_nt<T, OutT, ACT, RESID, SITE>(b, stream);
        }
        return launch_gemm256_nt<T, OutT, ACT, RESID, SITE>(g, stream);
    }
    return launch_gemm_nt<T, OutT, ACT, RESID, SITE>(g, stream);
}

}  // namespace arp

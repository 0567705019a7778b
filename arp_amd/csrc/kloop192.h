// The 256 x 192 tile K loop shared by the fused QKV + attention kernel (qkvattn.hip) and the 256 x 192 GEMM (gemm192.hip):
// the two-phase pipeline of gemm256.h (8 waves as 2 (M) x 4 (N), LDS-DMA steps three ahead, counted vmcnt, wave groups one
// barrier out of step) with a 128 x 48 wave tile.  Units per K-tile: U0 = A rows of phase A (16 KiB, 2 LDS-DMA per thread),
// UW = the 192 W rows (24 KiB, 3 per thread), U3 = A rows of phase B; even step = {U0, UW} (5 instructions per thread), odd step
// = {U3} (2).  Hazards as argued in DESIGN.md section 5: phase p reads step p, issues step p+3 over the region of step p-1, and
// waits (vmcnt(7): one even + one odd step stay in flight) until step p+1 has landed before its first barrier.
#pragma once
#include <type_traits>

#include "common.h"

namespace arp {

constexpr int K192_THREADS = 512;
constexpr int K192_BUF_BYTES = (256 + 192) * 128;  // one K-tile: 56 KiB
constexpr int K192_W_REGION = 256 * 128;
constexpr int K192_RING_BYTES = 2 * K192_BUF_BYTES;  // 112 KiB; the 192 bias floats follow

// Accumulates acc[mq][ni][mi] (+)= A[m0.., K] . W[n0.., K]^T for the workgroup's 256 x 192 tile; on return every LDS byte of the
// ring is dead and the tile's 192 bias values (if any) sit at smem + K192_RING_BYTES.  Rows are clamped to [0, m_rows) /
// [0, n_rows): clamped rows are computed on valid memory and must not be consumed.
template <typename T>
__device__ __forceinline__ void kloop_256x192(char* smem, const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ bias, int lda,
                                              int ldw, int m0, int n0, int m_rows, int n_rows, int K, f32x4_v (&acc)[2][3][4]) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    constexpr int EPB = 64, EPC = 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA plan ---------------------------------------------------------------------------------------------------
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    constexpr int KV = ARP_G2_KV;  // 1: SADDR-form LDS-DMA statements + peeled steady state (common.h), same bits
    const T* srcA[2][2];  // [q-row][instr]
    int dstA[2][2];
    const T* srcW[3];
    int dstW[3];
    uint32_t offA[2][2], offW[3];  // KV = 1: byte offsets from the tile's first A / W row
    const char* a_tile = reinterpret_cast<const char*>(A + (size_t)m0 * lda);
    const char* w_tile = reinterpret_cast<const char*>(W + (size_t)n0 * ldw);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr0 = (wave * 2 + i) * 8;
            const int row0 = (lr0 >> 6) * 128 + q * 64 + (lr0 & 63);
            int am = m0 + row0 + srow;
            am = am < m_rows ? am : m_rows - 1;  // rows past the tile's frames are computed on valid memory and never consumed
            if constexpr (KV == 1) offA[q][i] = (uint32_t)(((size_t)(am - m0) * lda + schunk * EPC) * sizeof(T));
            else srcA[q][i] = A + (size_t)am * lda + schunk * EPC;
            dstA[q][i] = row0 * 128;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row0 = (wave * 3 + i) * 8;
        {
            int wn = n0 + row0 + srow;
            wn = wn < n_rows ? wn : n_rows - 1;
            if constexpr (KV == 1) offW[i] = (uint32_t)(((size_t)(wn - n0) * ldw + schunk * EPC) * sizeof(T));
            else srcW[i] = W + (size_t)wn * ldw + schunk * EPC;
        }
        dstW[i] = K192_W_REGION + row0 * 128;
    }
    const int nk = K / EPB;
    const int S2 = 2 * nk;
    // KV = 1: one step, no existence test (the caller knows it exists)
    auto issue_odd_ss = [&](int tt) {
        const char* sa = a_tile + (size_t)tt * 128;
        const uint32_t base = lds0 + (tt & 1) * K192_BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_saddr(sa, offA[1][i], base + dstA[1][i]);
    };
    auto issue_even_ss = [&](int tt) {
        const char* sa = a_tile + (size_t)tt * 128;
        const char* sw = w_tile + (size_t)tt * 128;
        const uint32_t base = lds0 + (tt & 1) * K192_BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_saddr(sa, offA[0][i], base + dstA[0][i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma16_saddr(sw, offW[i], base + dstW[i]);
    };
    auto issue_step = [&](int st) {
        if (st >= S2) return;
        const int tt = st >> 1;
        if constexpr (KV == 1) {
            if (st & 1) issue_odd_ss(tt);
            else issue_even_ss(tt);
            return;
        }
        char* base = smem + (tt & 1) * K192_BUF_BYTES;
        if (st & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[1][i] + (size_t)tt * EPB),
                                                 (__attribute__((address_space(3))) void*)(base + dstA[1][i]), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[0][i] + (size_t)tt * EPB),
                                                 (__attribute__((address_space(3))) void*)(base + dstA[0][i]), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + (size_t)tt * EPB),
                                                 (__attribute__((address_space(3))) void*)(base + dstW[i]), 16, 0, 0);
        }
    };
    auto step_cnt = [&](int st) { return st < S2 ? ((st & 1) ? 2 : 5) : 0; };
    auto wait_instr = [&](int n) {
        if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- fragment addressing ----------------------------------------------------------------------------------------------
    const int fr = lane & 15, fg = lane >> 4;
    const int a_base = (wr * 128 + fr) * 128;
    const int b_base = K192_W_REGION + (wc * 48 + fr) * 128;
    const int coff0 = ((0 * 4 + fg) ^ (fr & 7)) << 4;
    const int coff1 = ((1 * 4 + fg) ^ (fr & 7)) << 4;

    u32x4_v areg[4][2];    // [mi][ks]
    u32x4_v breg[3][2];    // [ni][ks]  loaded in phase A, reused in phase B
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[a][b][d] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    auto load_a = [&](const char* buf, int mq) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const char* p = buf + a_base + (mq * 64 + mi * 16) * 128;
            areg[mi][0] = *reinterpret_cast<const u32x4_v*>(p + coff0);
            areg[mi][1] = *reinterpret_cast<const u32x4_v*>(p + coff1);
        }
    };
    auto load_b = [&](const char* buf) {
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const char* p = buf + b_base + ni * 16 * 128;
            breg[ni][0] = *reinterpret_cast<const u32x4_v*>(p + coff0);
            breg[ni][1] = *reinterpret_cast<const u32x4_v*>(p + coff1);
        }
    };
    auto phase_tail = [&](int p) {
        wait_instr(step_cnt(p + 2) + step_cnt(p + 3));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](auto MQ) {
        constexpr int mq = decltype(MQ)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mq][ni][mi] = mfma16<T>(breg[ni][ks], areg[mi][ks], acc[mq][ni][mi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // the head's 192 bias values: one LDS-DMA of wave 0, older than every operand unit (so every counted wait covers it)
    float* bias_s = reinterpret_cast<float*>(smem + K192_RING_BYTES);
    if (bias && wave == 0 && lane < 48) {
        if constexpr (KV == 1) dma16_saddr(bias, (uint32_t)min(n0 + lane * 4, n_rows - 4) * 4u, lds0 + K192_RING_BYTES);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias + min(n0 + lane * 4, n_rows - 4)),
                                              (__attribute__((address_space(3))) void*)bias_s, 16, 0, 0);
    }
    issue_step(0);
    issue_step(1);
    issue_step(2);
    wait_instr(step_cnt(1) + step_cnt(2));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);

    int kt_first = 0;
    if constexpr (KV == 1) {
        // steady state: K-tiles 0 .. nk-3 issue steps 2kt+3 and 2kt+4, both of which exist; 7 LDS-DMA instructions stay in flight at each wait
        auto tail_ss = [&]() {
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), as the builtin (hipcc's wait bookkeeping sees it)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto ktile_ss = [&](int kt, auto BUF) {
            const char* buf = smem + decltype(BUF)::value * K192_BUF_BYTES;
            load_a(buf, 0);
            load_b(buf);
            issue_odd_ss(kt + 1);
            tail_ss();
            compute(I0{});
            load_a(buf, 1);
            issue_even_ss(kt + 2);
            tail_ss();
            compute(I1{});
        };
        for (; kt_first + 3 < nk; kt_first += 2) {
            ktile_ss(kt_first, I0{});
            ktile_ss(kt_first + 1, I1{});
        }
        if (kt_first + 2 < nk) {
            ktile_ss(kt_first, I0{});
            ++kt_first;
        }
    }
    for (int kt = kt_first; kt < nk; ++kt) {
        const char* buf = smem + (kt & 1) * K192_BUF_BYTES;
        const int p = 2 * kt;
        load_a(buf, 0);
        load_b(buf);
        issue_step(p + 3);
        phase_tail(p);
        compute(I0{});
        load_a(buf, 1);
        issue_step(p + 4);
        phase_tail(p + 1);
        compute(I1{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
    __builtin_amdgcn_sched_barrier(0);

}

}  // namespace arp

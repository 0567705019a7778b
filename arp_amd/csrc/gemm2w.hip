// 128x192-tile NT GEMM, two workgroups per CU (design notes in gemm2w.h); its own translation unit.
#include "gemm2w.h"

#include <cstdlib>
#include <vector>

#include "../../include/arp_hip.h"
#include "runtime.h"

namespace arp {

template <typename T, typename OutT, int ACT, bool RESID>
// (register cap experiment of gemm256.h, OFF: 120 = 240 registers in all, 128 = uncapped)
#ifndef W2_MAX_VGPR
#define W2_MAX_VGPR 128
#endif
__global__ __launch_bounds__(W2_THREADS, 2) __attribute__((amdgpu_num_vgpr(W2_MAX_VGPR))) void gemm2w_kernel(GemmArgs g) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPB = 64, EPC = 8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int n_tiles = (g.N + W2_BN - 1) / W2_BN;
    const int m_tiles = (g.M + W2_BM - 1) / W2_BM;
    int m0, n0;
    {
        int t = xcd_remap(blockIdx.x, m_tiles * n_tiles);
        const int group_m = g.group_m > 0 ? g.group_m : W2_GROUP_M;
        const int per_group = group_m * n_tiles;
        const int grp = t / per_group;
        const int first_m = grp * group_m;
        const int gsize = min(m_tiles - first_m, group_m);
        t -= grp * per_group;
        m0 = (first_m + t % gsize) * W2_BM;
        n0 = (t / gsize) * W2_BN;
    }
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.W);

    // ---- LDS-DMA plan: a piece = 8 rows x 128 B; wave w fills A pieces 4w..4w+3 and W pieces 6w..6w+5 ----------------------
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    constexpr int KV = ARP_G2_KV;  // 1: every LDS-DMA is a SADDR-form asm statement (common.h::dma16_saddr): tile base + K offset in SGPRs, 32-bit lane offsets
    const T* srcA[4];
    const T* srcW[6];
    uint32_t offA[4], offW[6];
    const char* a_tile = reinterpret_cast<const char*>(A + (size_t)m0 * g.lda);
    const char* w_tile = reinterpret_cast<const char*>(W + (size_t)n0 * g.ldw);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int am = m0 + (wave * 4 + i) * 8 + srow;
        am = am < g.M ? am : g.M - 1;
        if constexpr (KV == 1) offA[i] = (uint32_t)(((size_t)(am - m0) * g.lda + schunk * EPC) * sizeof(T));
        else srcA[i] = A + (size_t)am * g.lda + schunk * EPC;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int wn = n0 + (wave * 6 + i) * 8 + srow;
        wn = wn < g.N ? wn : g.N - 1;
        if constexpr (KV == 1) offW[i] = (uint32_t)(((size_t)(wn - n0) * g.ldw + schunk * EPC) * sizeof(T));
        else srcW[i] = W + (size_t)wn * g.ldw + schunk * EPC;
    }
    const int nk = g.K / EPB;
    auto issue = [&](int kt) {
        if constexpr (KV == 1) {
            const char* sa = a_tile + (size_t)kt * 128;
            const char* sw = w_tile + (size_t)kt * 128;
            const uint32_t b = lds0 + (kt & 1) * W2_BUF_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16_saddr(sa, offA[i], b + (wave * 4 + i) * 1024);
#pragma unroll
            for (int i = 0; i < 6; ++i) dma16_saddr(sw, offW[i], b + W2_W_REGION + (wave * 6 + i) * 1024);
            return;
        }
        char* base = smem + (kt & 1) * W2_BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * EPB),
                                             (__attribute__((address_space(3))) void*)(base + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + (size_t)kt * EPB),
                                             (__attribute__((address_space(3))) void*)(base + W2_W_REGION + (wave * 6 + i) * 1024), 16, 0, 0);
    };

    // ---- fragments ---------------------------------------------------------------------------------------------------------
    const int fr = lane & 15, fg = lane >> 4;
    const int a_base = (wr * 64 + fr) * 128;
    const int b_base = W2_W_REGION + (wc * 96 + fr) * 128;
    const int coff[2] = {((0 * 4 + fg) ^ (fr & 7)) << 4, ((1 * 4 + fg) ^ (fr & 7)) << 4};

    f32x4_v acc[6][4];  // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 6; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    u32x4_v a0[4], b0[6], a1[4], b1[6];
    auto read_set = [&](const char* buf, int ks, u32x4_v (&a)[4], u32x4_v (&b)[6]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const u32x4_v*>(buf + a_base + mi * 16 * 128 + coff[ks]);
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) b[ni] = *reinterpret_cast<const u32x4_v*>(buf + b_base + ni * 16 * 128 + coff[ks]);
    };
    auto mfma_set = [&](const u32x4_v (&a)[4], const u32x4_v (&b)[6]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 6; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16<T>(b[ni], a[mi], acc[ni][mi]);
        __builtin_amdgcn_s_setprio(0);
    };

    issue(0);
    if (nk > 1) {
        issue(1);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_set(smem, 0, a0, b0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* buf = smem + (kt & 1) * W2_BUF_BYTES;
        read_set(buf, 1, a1, b1);  // in flight under the MFMAs of half 0
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's reads of tile kt are done and its share of tile kt+1 has landed; the barrier joins the four waves.
        // (The waits are builtins, not inline asm, so that the compiler's own wait insertion knows what has been retired.)
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_set(smem + ((kt + 1) & 1) * W2_BUF_BYTES, 0, a0, b0);  // in flight under the MFMAs of half 1
        if (kt + 2 < nk) issue(kt + 2);                              // into the buffer tile kt was read from
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        // half 0 of the next tile was read under the MFMAs above and has long landed: retire it here, so that the wait in
        // front of the next MFMA block does not also cover the reads issued right before it
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
    }
    {  // last K-tile
        read_set(smem + ((nk - 1) & 1) * W2_BUF_BYTES, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        mfma_set(a1, b1);
    }
    __syncthreads();  // every wave is done with the ring before it is reused as the epilogue tile

    // ---- epilogue -----------------------------------------------------------------------------------------------------------
    OutT* out = static_cast<OutT*>(g.out);  // may alias g.resid (in-place residual add)
    const bool full_n = n0 + W2_BN <= g.N;
    float4 bias4[6];
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
        const int n = n0 + wc * 96 + ni * 16 + fg * 4;
        // an unconditional load from a selected pointer and a clamped column (a guarded load is a branch + vmcnt(0): six dependent round trips at the head of
        // every tile's epilogue); without a bias the 16 bytes come from the A operand and are replaced by zeros, columns past N are never stored
        const float4 bl = *reinterpret_cast<const float4*>(g.bias ? g.bias + min(n, g.N - 4) : reinterpret_cast<const float*>(g.A));
        bias4[ni] = g.bias ? bl : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g.flags & 1) {  // ablation: keep the accumulators live, store (almost) nothing
        float sacc = 0.f;
#pragma unroll
        for (int ni = 0; ni < 6; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) sacc += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (sacc == 12345.678f) Elem<OutT>::st(out, sacc);
        return;
    }
    if constexpr (sizeof(OutT) == 2) {
        constexpr int RS = W2_BN * 2 + 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const int row = wr * 64 + mi * 16 + fr, col = wc * 96 + ni * 16 + fg * 4;
                const f32x4_v a4 = acc[ni][mi];
                float v[4] = {a4[0] + bias4[ni].x, a4[1] + bias4[ni].y, a4[2] + bias4[ni].z, a4[3] + bias4[ni].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, true>(v[j]);
                *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
            }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 12; ++it) {
            const int idx = it * W2_THREADS + tid;
            const int r = idx / 24, ch = idx - r * 24;
            const int m = m0 + r, n = n0 + ch * 8;
            if (m < g.M && (full_n || n + 8 <= g.N))
                *reinterpret_cast<u32x4_v*>(out + (size_t)m * g.ldo + n) = *reinterpret_cast<const u32x4_v*>(smem + r * RS + ch * 16);
        }
    } else {
        constexpr int RSF = W2_BN * 4 + 16;
        // a pass's residual rows are requested before its tile is staged (all in flight at once); pass 1's are requested AHEAD of pass 0's stores --
        // behind them they could not be used before all twelve stores had completed (one in-order counter per wave): gemm256.h, round 4
#ifndef W2_RES_EARLY
#define W2_RES_EARLY 1
#endif
        float4 rres[2][12];
        auto load_res = [&](int p) {
            if constexpr (RESID) {
#pragma unroll
                for (int it = 0; it < 12; ++it) {
                    const int idx = it * W2_THREADS + tid;
                    const int lr = idx / 48, ch = idx - lr * 48;
                    const int m = m0 + (lr >> 5) * 64 + p * 32 + (lr & 31), n = n0 + ch * 4;
                    // unconditional, clamped address (gemm256.h: a guarded load is a branch + vmcnt(0) per row)
                    rres[p][it] = *reinterpret_cast<const float4*>(g.resid + (size_t)min(m, g.M - 1) * g.ldr + min(n, g.N - 4));
                }
            }
        };
        load_res(0);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p) __syncthreads();
            if (p && !W2_RES_EARLY) load_res(1);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) {
                    const int mi = 2 * p + mh;
                    const int lrow = wr * 32 + mh * 16 + fr, col = wc * 96 + ni * 16 + fg * 4;
                    const f32x4_v a4 = acc[ni][mi];
                    float v[4] = {a4[0] + bias4[ni].x, a4[1] + bias4[ni].y, a4[2] + bias4[ni].z, a4[3] + bias4[ni].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, true>(v[j]);
                    *reinterpret_cast<float4*>(smem + lrow * RSF + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
            if (p == 0 && W2_RES_EARLY) load_res(1);
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int idx = it * W2_THREADS + tid;
                const int lr = idx / 48, ch = idx - lr * 48;
                const int m = m0 + (lr >> 5) * 64 + p * 32 + (lr & 31), n = n0 + ch * 4;
                if (m < g.M && (full_n || n + 4 <= g.N)) {
                    float4 v = *reinterpret_cast<const float4*>(smem + lr * RSF + ch * 16);
                    if constexpr (RESID) {
                        const float4 r = rres[p][it];
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * g.ldo + n) = v;
                }
            }
        }
    }
}

template <typename T, typename OutT, int ACT, bool RESID>
static int launch_gemm2w_impl(const GemmArgs& g, hipStream_t stream) {
    if (g.M <= 0) return 0;
    // whole-vector epilogue only: N, ldo (and ldr) multiples of 8 / 4; ragged N is handled at 8- / 4-column granularity
    if (g.N <= 0 || g.K <= 0 || g.K % 64 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0 || (g.N & 7) || (g.ldo & 7) || (RESID && (g.ldr & 3)))
        return fail("gemm2w: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K));
    auto kern = gemm2w_kernel<T, OutT, ACT, RESID>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES));
        attr_set = true;
    }
    const int m_tiles = (g.M + W2_BM - 1) / W2_BM, n_tiles = (g.N + W2_BN - 1) / W2_BN;
    hipLaunchKernelGGL(kern, dim3(m_tiles * n_tiles), dim3(W2_THREADS), W2_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

// instantiated combinations: (out_f32, act, resid)
#define ARP_W2_COMBOS(X)        \
    X(0, ACT_NONE, 0)           \
    X(0, ACT_QGELU, 0)          \
    X(0, ACT_GELU_TANH, 0)      \
    X(1, ACT_NONE, 0)           \
    X(1, ACT_NONE, 1)

bool gemm2w_has(int tcode, int out_f32, int act, int resid) {
    if (tcode != 1 && tcode != 2) return false;
#define X(o, a, r) \
    if (out_f32 == o && act == a && resid == r) return true;
    ARP_W2_COMBOS(X)
#undef X
    return false;
}

template <typename T>
static int dispatch(int out_f32, int act, int resid, const GemmArgs& g, hipStream_t stream) {
#define X(o, a, r)                                     \
    if (out_f32 == o && act == a && resid == r) {      \
        if constexpr (o == 1) return launch_gemm2w_impl<T, float, a, r != 0>(g, stream); \
        else return launch_gemm2w_impl<T, T, a, r != 0>(g, stream);                      \
    }
    ARP_W2_COMBOS(X)
#undef X
    return fail("gemm2w: combination not instantiated");
}

int launch_gemm2w_dyn(int tcode, int out_f32, int act, int resid, const GemmArgs& g, hipStream_t stream) {
    if (tcode == 1) return dispatch<bf16_t>(out_f32, act, resid, g, stream);
    if (tcode == 2) return dispatch<f16_t>(out_f32, act, resid, g, stream);
    return fail("gemm2w: 16-bit operand types only");
}

}  // namespace arp

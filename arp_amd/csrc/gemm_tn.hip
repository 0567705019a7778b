// TN weight-gradient GEMM (design notes in gemm_tn.h); its own translation unit.
#include "gemm_tn.h"

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
    // ---- epilogue: lane holds m = .. + fr, n = .. + 4 fg + {0..3} ---------------------------------------------------------------
    float* out = g.out + (size_t)blockIdx.y * g.slice_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int m = m0 + wr * 64 + mi * 16 + fr, n = n0 + wc * 64 + ni * 16 + fg * 4;
            const f32x4_v a4 = acc[ni][mi];
            *reinterpret_cast<float4*>(out + (size_t)m * g.ldo + n) = make_float4(a4[0] * g.alpha, a4[1] * g.alpha, a4[2] * g.alpha, a4[3] * g.alpha);
        }
}

template <typename T> static int launch_impl(const GemmTnArgs& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % TN_BM || g.N % TN_BN || g.K % TN_BK || g.lda % 8 || g.ldb % 8 || g.ldo % 4 || g.ksplit < 1)
        return fail("gemm_tn: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K));
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

int launch_gemm_tn(int tcode, const GemmTnArgs& g, hipStream_t stream) {
    if (tcode == 1) return launch_impl<bf16_t>(g, stream);
    if (tcode == 2) return launch_impl<f16_t>(g, stream);
    return fail("gemm_tn: 16-bit operand types only");
}

}  // namespace arp

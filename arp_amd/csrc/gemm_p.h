// "Paired" NT GEMM:  C[M,N] = epilogue(A[M,K] . W[N,K]^T), 256x128 tile per 256-thread workgroup, TWO workgroups
// resident per CU.
//
// Why a second big-tile kernel (measured on MI355X): with one 256x256 workgroup per CU
// (gemm256.h) the K loop runs at 1.3 PFLOP/s but the per-tile prologue (first-load latency) and the epilogue
// (a CU sinks only ~12 B/cycle of stores; 128 KiB per tile) are serialised behind it -- at K = 768 they cost as
// much as the K loop.  A wave's loads and stores share one in-order vmcnt, so a workgroup cannot run ahead of
// its own stores; ANOTHER workgroup on the same CU can.  Here two independent workgroups share each CU:
// while one drains its epilogue or waits for its first tiles, the other's waves keep the matrix pipes busy.
//
//   * 4 waves as 2 (M) x 2 (N), 128x64 per wave (128 accumulator VGPRs, same fragment layout as gemm256.h),
//     __launch_bounds__(256, 2): 2 waves per SIMD, one from each resident workgroup.
//   * K in tiles of 64 bytes per row (32 bf16 / 16 f32): 24 KiB per K-tile, 3-deep LDS ring = 72 KiB per
//     workgroup (2 x 72 <= 160 KiB).  One K-tile = 12 ds_read_b128 + 32 MFMA per wave; ONE barrier per K-tile.
//   * LDS-DMA (global_load_lds_dwordx4), 6 per thread per K-tile, issued two K-tiles ahead; the only wait is a
//     counted s_waitcnt vmcnt(6) (the next tile stays in flight across the barrier).
//       RAW: tile kt is waited for (each wave, its own DMA) and then the barrier of iteration kt precedes every read.
//       WAR: tile kt+2 lands in the buffer tile kt-1 was read from; those reads fed the MFMAs of iteration kt-1,
//            which every wave finished before arriving at the barrier of iteration kt; the DMA is issued after it.
//   * 64-byte rows: 16-byte chunk c of row r lives at physical chunk c ^ ((4 - (r >> 2)) & 3); with that the four
//     16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS) each touch 16 distinct 16-byte slots.
//   * epilogue staged through LDS (whole 256-B / 512-B rows out), as in gemm256.h.
#pragma once
#include "common.h"
#include "gemm.h"

namespace arp {

constexpr int GP_BM = 256, GP_BN = 128, GP_THREADS = 256;
constexpr int GP_TILE_BYTES = (GP_BM + GP_BN) * 64;  // 24 KiB per K-tile
constexpr int GP_RING = 3;
constexpr int GP_LDS_BYTES = GP_RING * GP_TILE_BYTES;  // 72 KiB
constexpr int GP_W_REGION = GP_BM * 64;
static_assert(256 * (128 * 2 + 16) <= GP_LDS_BYTES && 128 * (128 * 4 + 16) <= GP_LDS_BYTES, "epilogue tile must fit the ring");

__device__ __forceinline__ int gp_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <typename T, typename OutT, int ACT, bool RESID, int SITE>
__global__ __launch_bounds__(GP_THREADS, 2) void gemm_p_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPB = 64 / (int)sizeof(T);  // elements per K-tile: 32 (bf16) / 16 (f32)
    constexpr int EPC = 16 / (int)sizeof(T);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int n_tiles = (g.N + GP_BN - 1) / GP_BN;
    const int m_tiles = (g.M + GP_BM - 1) / GP_BM;
    int t = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int group_m = g.group_m > 0 ? g.group_m : 8;
    const int per_group = group_m * n_tiles;
    const int grp = t / per_group;
    const int first_m = grp * group_m;
    const int gsize = min(m_tiles - first_m, group_m);
    t -= grp * per_group;
    const int m0 = (first_m + t % gsize) * GP_BM;
    const int n0 = (t / gsize) * GP_BN;

    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.W);

    // ---- LDS-DMA plan: one wave-instruction = 16 rows x 64 B; wave w fills A groups w, w+4, w+8, w+12 and W groups w, w+4
    const int srow = lane >> 2;                      // row inside the 16-row group
    const int schunk = (lane & 3) ^ gp_swz(srow);    // logical chunk this lane fetches (group bases are multiples of 16)
    const T* src[6];
    int dst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool isA = i < 4;
        const int grp16 = isA ? (wave + 4 * i) : (wave + 4 * (i - 4));
        const int row = grp16 * 16 + srow;
        dst[i] = (isA ? 0 : GP_W_REGION) + grp16 * 1024;
        if (isA) {
            int am = m0 + row;
            am = am < g.M ? am : g.M - 1;
            src[i] = A + (size_t)am * g.lda + schunk * EPC;
        } else {
            int wn = n0 + row;
            wn = wn < g.N ? wn : g.N - 1;
            src[i] = W + (size_t)wn * g.ldw + schunk * EPC;
        }
    }
    const int nk = g.K / EPB;
    auto issue = [&](int kt) {
        char* base = smem + (kt % GP_RING) * GP_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * EPB),
                                             (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
    };

    // ---- fragment addressing ----------------------------------------------------------------------
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = (fg ^ gp_swz(fr)) << 4;  // fragment rows are (multiple of 16) + fr
    const int a_base = (wr * 128 + fr) * 64 + coff;
    const int b_base = GP_W_REGION + (wc * 64 + fr) * 64 + coff;

    f32x4_v acc[2][2][2][4];  // [mq][nq][ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4_v{0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the fragments of K-tile kt+1 are read from LDS while the MFMAs of K-tile kt run, so a wave's
    // MFMAs issue back to back (two operand register sets, named statically by the 2x unrolled loop).
    u32x4_v af0[8], wf0[4], af1[8], wf1[4];
    auto read_frags = [&](int kt, u32x4_v (&af)[8], u32x4_v (&wf)[4]) {
        const char* buf = smem + (kt % GP_RING) * GP_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const u32x4_v*>(buf + a_base + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const u32x4_v*>(buf + b_base + i * 16 * 64);
    };
    auto mfmas = [&](const u32x4_v (&af)[8], const u32x4_v (&wf)[4]) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        if constexpr (sizeof(T) == 2) {
                            acc[mq][nq][ni][mi] = mfma16<T>(wf[nq * 2 + ni], af[mq * 4 + mi], acc[mq][nq][ni][mi]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[mq][nq][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    __uint_as_float(wf[nq * 2 + ni][j]), __uint_as_float(af[mq * 4 + mi][j]), acc[mq][nq][ni][mi], 0, 0, 0);
                        }
                    }
    };
    // wait until tile `kt` has landed (this wave's share), make it visible, and retire every LDS read of the tile
    // whose buffer the next DMA overwrites
    auto tile_ready = [&](int kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0);
    if (nk > 1) issue(1);
    tile_ready(0);
    read_frags(0, af0, wf0);
    if (nk > 2) issue(2);
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile kt in set 0
        if (kt + 1 < nk) {
            tile_ready(kt + 1);
            read_frags(kt + 1, af1, wf1);           // in flight under the MFMAs below
            if (kt + 3 < nk) issue(kt + 3);         // buffer of tile kt: its reads were retired by tile_ready's lgkmcnt(0)
        }
        __builtin_amdgcn_s_setprio(1);
        mfmas(af0, wf0);
        __builtin_amdgcn_s_setprio(0);
        // odd tile kt+1 in set 1
        if (kt + 1 < nk) {
            if (kt + 2 < nk) {
                tile_ready(kt + 2);
                read_frags(kt + 2, af0, wf0);
                if (kt + 4 < nk) issue(kt + 4);
            }
            __builtin_amdgcn_s_setprio(1);
            mfmas(af1, wf1);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    __syncthreads();  // every wave is done with the ring before it is reused as the epilogue tile

    // ---- epilogue --------------------------------------------------------------------------------------
    OutT* out = static_cast<OutT*>(g.out);  // may alias g.resid (in-place residual add)
    const bool vec_ok = ((g.N | g.ldo | g.ldr) & 3) == 0;
    const bool staged = vec_ok && ((g.N | g.ldo) & 7) == 0 && !(g.flags & 2);
    if (staged) {
        if constexpr (sizeof(OutT) == 2) {
            constexpr int RS = 128 * 2 + 16;
#pragma unroll
            for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int row = wr * 128 + mq * 64 + mi * 16 + fr;
                            const int col = wc * 64 + nq * 32 + ni * 16 + fg * 4;
                            const f32x4_v a4 = acc[mq][nq][ni][mi];
                            float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                            if (g.bias && n0 + col < g.N) {
                                const float4 b = *reinterpret_cast<const float4*>(g.bias + n0 + col);
                                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) == 2>(v[j]);
                            *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
                        }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
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
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int lrow = wr * 64 + mi * 16 + fr;
                            const int col = wc * 64 + nq * 32 + ni * 16 + fg * 4;
                            const f32x4_v a4 = acc[p][nq][ni][mi];
                            float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                            if (g.bias && n0 + col < g.N) {
                                const float4 b = *reinterpret_cast<const float4*>(g.bias + n0 + col);
                                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, sizeof(T) == 2>(v[j]);
                            *reinterpret_cast<float4*>(smem + lrow * RSF + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                __syncthreads();
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int lr = it * 8 + wave * 2 + (lane >> 5);
                    const int m = m0 + (lr >> 6) * 128 + p * 64 + (lr & 63), n = n0 + (lane & 31) * 4;
                    if (m < g.M && n < g.N) {
                        float4 v = *reinterpret_cast<const float4*>(smem + lr * RSF + (lane & 31) * 16);
                        if constexpr (RESID) {
                            const float4 r = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n);
                            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                        }
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * g.ldo + n) = v;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 128 + mq * 64 + mi * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int n = n0 + wc * 64 + nq * 32 + ni * 16 + fg * 4;
                    if (n >= g.N) continue;
                    const f32x4_v a4 = acc[mq][nq][ni][mi];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (n + j >= g.N) break;
                        float x = a4[j] + (g.bias ? g.bias[n + j] : 0.f);
                        x = apply_act<ACT, sizeof(T) == 2>(x);
                        if constexpr (RESID) x += g.resid[(size_t)m * g.ldr + n + j];
                        Elem<OutT>::st(out + (size_t)m * g.ldo + n + j, x);
                    }
                }
        }
}

template <typename T, typename OutT, int ACT, bool RESID, int SITE>
inline int launch_gemm_p(const GemmArgs& g, hipStream_t stream) {
    constexpr int EPB = 64 / (int)sizeof(T);
    if (g.M <= 0) return 0;
    if (g.N <= 0 || g.K % EPB != 0 || g.K <= 0 || g.lda % (16 / (int)sizeof(T)) != 0 || g.ldw % (16 / (int)sizeof(T)) != 0)
        return fail("gemm_p: unsupported shape M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K));
    auto kern = gemm_p_kernel<T, OutT, ACT, RESID, SITE>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES));
        attr_set = true;
    }
    const int m_tiles = (g.M + GP_BM - 1) / GP_BM;
    const int n_tiles = (g.N + GP_BN - 1) / GP_BN;
    hipLaunchKernelGGL(kern, dim3(m_tiles * n_tiles), dim3(GP_THREADS), GP_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp

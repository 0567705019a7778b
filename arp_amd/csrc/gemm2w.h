// "Two workgroups per CU" NT GEMM:  C[M,N] = epilogue(A[M,K] . W[N,K]^T), 128x192 tile per 256-thread workgroup.
//
// Why (round 2 measurements, scripts/store_bench.hip + the per-site timings of bench.py): with one 256x256 workgroup per CU
// (gemm256.h) a tile's epilogue is serialised behind its K loop, and at K = 768 it is 15-20 % of the tile: QuickGELU is ~34 VALU
// issue cycles per element-row (8.7 k cycles per SIMD per tile with both of the SIMD's waves in it), all 256 CUs hit their
// store burst together (a 128-KiB tile leaves ONE CU in 1.3 k cycles but takes 4.9 k when every CU stores at once), and the
// f32 residual read-modify-write is latency-bound.  None of that needs the matrix pipe, so a SECOND workgroup on the same CU
// can run its K loop meanwhile.  That needs two workgroups to fit a CU: <= 80 KiB of LDS and <= 256 registers at one wave per
// SIMD each.
//
//   * 4 waves as 2 (M) x 2 (N), 64 x 96 per wave = 4 x 6 MFMA fragments of 16x16 (96 accumulator VGPRs);
//   * K in tiles of 128 B per row (whole cache lines per LDS-DMA row segment -- the 64-B rows of the earlier paired kernel
//     asked L1/L2 for every line twice); one K-tile = 40 KiB, two buffers = 80 KiB;
//   * software pipeline inside the wave: the fragments of the NEXT half K-tile (32 elements) are read from LDS while the MFMAs
//     of the current half run, two register sets named statically;
//   * ONE barrier per K-tile: after it every wave has finished reading tile kt (so its buffer takes tile kt+2) and tile kt+1
//     has landed (each wave waits for its own LDS-DMA, the barrier joins them).  The other workgroup on the CU is not
//     synchronised with this one: its waves fill the matrix pipe while this one waits;
//   * epilogue staged through LDS (whole 384-B / 768-B rows out), bias from global memory (no LDS to spare).
//
// Per accumulator the MFMA sequence (K-tiles ascending, halves 0 then 1) is the one of gemm256.h / gemm.h: results are
// bit-identical to those kernels.
#pragma once
#include "common.h"
#include "gemm.h"

namespace arp {

constexpr int W2_BM = 128, W2_BN = 192, W2_THREADS = 256;
constexpr int W2_BUF_BYTES = (W2_BM + W2_BN) * 128;  // 40 KiB
constexpr int W2_W_REGION = W2_BM * 128;
constexpr int W2_LDS_BYTES = 2 * W2_BUF_BYTES;       // 80 KiB: two workgroups per CU
constexpr int W2_GROUP_M = 16;

// defined in gemm2w.hip; tcode: 1 = bf16, 2 = f16 operands; out_f32: output type float (else the operand type)
int launch_gemm2w_dyn(int tcode, int out_f32, int act, int resid, const GemmArgs& g, hipStream_t stream);
bool gemm2w_has(int tcode, int out_f32, int act, int resid);

template <typename T, typename OutT, int ACT, bool RESID>
inline int launch_gemm2w(const GemmArgs& g, hipStream_t stream) {
    if constexpr (sizeof(T) != 2) return fail("gemm2w: 16-bit operand types only");
    else return launch_gemm2w_dyn(__is_same(T, bf16_t) ? 1 : 2, sizeof(OutT) == 4 ? 1 : 0, ACT, RESID ? 1 : 0, g, stream);
}

}  // namespace arp

// "TN" GEMM for weight gradients:  C[M,N] = alpha * sum_k A[k,m] . B[k,n]  -- both operands stored ROW-major with the contraction
// index k as the row (A = an output gradient [rows, M], B = the layer input [rows, N]; dW = dY^T X).
//
// Why: the NT kernels (gemm.h / gemm256.h) want k contiguous in both operands, so every weight-gradient GEMM of the policy step
// used to be preceded by LDS-tiled transposes of both operands into K-padded [M, rows] / [N, rows] copies (0.10 ms of a
// 1.18 ms step at B = 32, plus 100 MB of extra traffic each).  Here the tiles go to LDS as they lie in memory
// ([64 k][128 m] rows of 256 B, by LDS-DMA) and the MFMA operand fragments are fetched with the CDNA4 transposing read
// ds_read_b64_tr_b16 (4 k x 16 m per 16-lane group, delivered k-major per lane) -- the idiom of the attention kernel's V^T.
//
//   * 128 x 128 tile per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave), two workgroups per CU;
//   * K-tile = 64 rows; A and B tiles 16 KiB each, double buffered (64 KiB);
//   * 32-byte slots (16 columns) XOR-swizzled by (row & 7): the eight (row, slot) segments a half-wave's transposing read
//     touches land on eight distinct bank groups (rows are 256 B = one full sweep of the 64 banks);
//   * split-K over gridDim.y into f32 partial slabs (fixed-order reduction by splitk_reduce_kernel -- no atomics);
//   * 16-bit operands only (the transposing read is a 16-bit instruction); the f32 parity mode keeps the transposed-copy path.
//
// Tried on the policy step's 768 x 768 x 32 896 weight gradients (0.075 ms each as built) and not kept: 32-row K-tiles at higher
// occupancy (0.09 ms), four stages / one workgroup per CU (0.107 ms), every K-slice's tiles on one XCD for L2 sharing (0.10 ms with
// 16 slices = 72 workgroups per XCD, 0.08 ms with 32-row tiles).  The contraction is bound by operand re-reads: 36 tiles x K x
// 512 B = 606 MB per GEMM past L2; 256 x 256 tiles would halve that.
//
// Requirements: M, N multiples of 128; K a multiple of 64 with rows [K_valid, K) of BOTH operands zero (callers pad).
#pragma once
#include "common.h"

namespace arp {

struct GemmTnArgs {
    const void* A;  // [K, lda] T
    const void* B;  // [K, ldb] T
    float* out;     // ksplit == 1: [M, ldo];  else partial slabs [ksplit][M][N] (ldo = N)
    int M, N, K;
    int lda, ldb, ldo;
    int ksplit;
    size_t slice_stride;
    float alpha;
    int tile256 = 0;     // 1: the 256 x 256-tile kernel (M, N multiples of 256; K a multiple of 64; four-stage ring)
    int xcd_slices = 0;  // 1 (tile256 only): ksplit is a multiple of 8 and each XCD runs ksplit / 8 whole K-slices
};

// tcode: 1 = bf16, 2 = f16 (defined in gemm_tn.hip)
int launch_gemm_tn(int tcode, const GemmTnArgs& g, hipStream_t stream);
// "NN": C[M,N] = alpha * sum_k A[m,k] . B[k,n] -- A [M, lda] K-contiguous, B [K, ldb] row-major as a weight [out, in] lies in memory (dX = dY . W with no
// transposed copy of W).  Same argument struct; M arbitrary (rows past M are never stored), N a multiple of 128, K a multiple of 64; split-K slabs
// [ksplit][M][N] (ldo = N) or ksplit == 1 with out [M, ldo].
int launch_gemm_nn(int tcode, const GemmTnArgs& g, hipStream_t stream);

}  // namespace arp

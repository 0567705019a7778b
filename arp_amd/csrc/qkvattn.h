// QKV projection + softmax(QK^T)V in ONE kernel for short sequences (ViT-B/32: N = 50 tokens per frame).
//
// Reference math: nn.MultiheadAttention inside ResidualAttentionBlock.attention, arp_dt/models/openai/layers.py:235-250
// (in-proj = x W_in^T + b_in, split q|k|v, per-head softmax(q k^T / sqrt(hd)) v).
//
// Why: as separate kernels the projection writes qkv [B*N, 3D] (236 MB per layer at 1024 frames) and the attention kernel
// reads it back (another 236 MB) to produce 79 MB.  Here a workgroup owns FPT whole frames (FPT*N <= 256 rows: 5 frames = 250
// rows at N = 50) and ONE head: its GEMM tile is 256 rows x 192 columns = that head's q | k | v (the in-proj weight rows are
// permuted head-major at load time), so after the K loop the tile holds everything the attention of those (frame, head) pairs
// needs.  The tile goes to LDS as 16-bit q / k / v images (the layout attn_mfma_kernel stages from global memory), the
// attention runs from LDS, and only the 64 output columns of the head leave the CU: 472 MB of HBM traffic per layer and one
// launch less.
//
// K loop: the two-phase pipeline of gemm256.h (8 waves as 2 (M) x 4 (N), LDS-DMA steps three ahead, counted vmcnt, wave groups
// one barrier out of step) with a 128 x 48 wave tile: phase A = rows 0..63 of the wave's 128 x its three 16-column fragments,
// phase B = rows 64..127; the three W fragments of a K-tile stay in registers across both phases.  Units per K-tile:
// U0 = A rows of phase A (16 KiB, 2 LDS-DMA per thread), UW = the 192 W rows (24 KiB, 3 per thread), U3 = A rows of phase B;
// even step = {U0, UW} (5 instructions per thread), odd step = {U3} (2).  Hazards as argued in DESIGN.md section 5: phase p
// reads step p, issues step p+3 over the region of step p-1, and waits (vmcnt(7): one even + one odd step stay in flight)
// until step p+1 has landed before its first barrier.
//
// Every q/k/v value is the same MFMA chain, rounded to the operand type at the same point, as in the unfused path, and the
// attention code is attn_mfma_kernel's: the output is bit-identical to gemm256 + attn_mfma (tests/test_clip_gpu.py).
#pragma once
#include "common.h"

namespace arp {

struct QkvAttnArgs {
    const void* A;      // [B*N, lda] T   LayerNorm output
    const void* W;      // [heads*192, ldw] T   head-major in-proj weight: rows h*192 + {0..63 q, 64..127 k, 128..191 v}
    const float* bias;  // [heads*192] same order
    void* out;          // [B*N, ldo] T   attention output, head h at columns h*64..
    int B, N, K, heads;
    int lda, ldw, ldo;
    int fpt;            // frames per tile, fpt*N <= 256
    int nq;             // query rows produced per frame (N, or 1 when only the class token is consumed)
    int causal;
    float scale;
    int group_m = 0;    // frame-tiles per L2 group of the block -> tile walk (0 = default; ARP_QA_GROUP_M overrides)
};

// usable when: 16-bit operands, head_dim 64, N <= 64, K a multiple of 64
inline bool qkv_attn_supported(int N, int D, int heads, int elem_size) {
    return elem_size == 2 && heads > 0 && D == heads * 64 && N >= 1 && N <= 64 && D % 64 == 0 && (256 / N) * N >= 192;
}

// defined in qkvattn.hip (its own translation unit: the kernel is compiled once, not once per includer)
int launch_qkv_attn_f16(QkvAttnArgs g, hipStream_t stream);
int launch_qkv_attn_bf16(QkvAttnArgs g, hipStream_t stream);
template <typename T> inline int launch_qkv_attn(const QkvAttnArgs& g, hipStream_t stream) {
    if constexpr (__is_same(T, f16_t)) return launch_qkv_attn_f16(g, stream);
    else if constexpr (__is_same(T, bf16_t)) return launch_qkv_attn_bf16(g, stream);
    else return fail("qkv_attn: 16-bit operand types only");
}

}  // namespace arp

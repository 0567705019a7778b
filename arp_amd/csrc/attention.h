// Short-sequence multi-head self-attention (N = 50 / 197 / 257 vision tokens, 77 text tokens,
// 12 policy tokens).  K and V of one (sample, head) fit in LDS, so there is no flash-style
// KV tiling: one workgroup per (sample, head) stages K/V once and produces every query row.
//
//   qkv : [B*N, 3*D]  (q | k | v along the last axis, head h at columns h*HD .. +HD of each third;
//          torch nn.MultiheadAttention in-proj layout, arp_dt/models/openai/model.py:234-238;
//          the policy's flax split "b n (h d)" is the same layout, arp_dt/layers.py:59-68)
//   out : [B*N, D]
//
// Two kernels:
//   attn_valu_kernel<T,HD>  : exact-f32 VALU kernel, one query row per lane, online softmax.
//                             Parity mode (T=float), and the bit-simple fallback for T=bf16.
//   attn_mfma_kernel        : bf16, HD = 64, QK^T and PV on v_mfma_f32_16x16x32_bf16; K and V staged
//                             row-major (XOR-swizzled 16-B chunks), V^T fragments fetched with the
//                             transposing LDS read; S is computed transposed (S^T = K.Q^T) so that the
//                             softmax'd accumulator IS the B operand of O^T = V^T.P^T with no
//                             cross-lane movement (cdna_hip_programming.md section 3, "an accumulator
//                             tile as the next MFMA's operand").
#pragma once
#include "common.h"

namespace arp {

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_valu_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int D,
                                                        int heads, float scale, int causal, int nq, const float* __restrict__ alibi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + (size_t)N * HD;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t ld = 3 * (size_t)D;
    const T* base = qkv + (size_t)b * N * ld + h * HD;
    for (int i = threadIdx.x; i < N * HD; i += blockDim.x) {
        const int t = i / HD, d = i - t * HD;
        Ks[i] = Elem<T>::ld(base + t * ld + D + d);
        Vs[i] = Elem<T>::ld(base + t * ld + 2 * D + d);
    }
    __syncthreads();
    const float slope = alibi ? alibi[h] : 0.f;  // config.alibi_bias of the policy transformer (arp_dt/layers.py:74-78): + slope_h * key index
    for (int qi = threadIdx.x; qi < nq; qi += blockDim.x) {  // nq = N, or fewer when only the first rows are consumed
        float q[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[d] = Elem<T>::ld(base + qi * ld + d) * scale;
            acc[d] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        const int kend = causal ? qi + 1 : N;
        for (int k = 0; k < kend; ++k) {
            const float* kr = Ks + (size_t)k * HD;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(q[d], kr[d], s);
            s += slope * (float)k;
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn);  // exp(-inf) = 0 on the first key
            const float p = expf(s - mn);
            const float* vr = Vs + (size_t)k * HD;
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, vr[d], acc[d] * alpha);
            m = mn;
        }
        const float inv = 1.0f / l;
        T* o = out + ((size_t)b * N + qi) * D + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) store4(o + d, acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
    }
}


// ---- exact-f32 MFMA kernel (head_dim 64): the parity mode's attention, and the attention of the f16x3 encoder mode ---------------------------------
// v_mfma_f32_16x16x4_f32 is a k-ordered f32 fmaf chain (MI355X_MICROARCH: 64 FLOP/clk/SIMD, the f32 vector rate -- but it leaves the VALU to the softmax
// and needs one operand register per lane instead of a 64-deep row per lane).  Same plan as the 16-bit kernel: S^T = K.Q^T so that the softmax'd accumulator
// IS the B operand of O^T = V^T.P^T: register r of key tile kt holds P[key kt*16 + 4g + r][query j] in lane (g, j), and the MFMA that consumes it takes
// V[kt*16 + 4g + r][d] as its A operand -- the contraction order over the keys is free.  K and V of one (sample, head) in LDS as f32 rows of 68 floats
// (272-byte rows: the ds_read_b128 of both operands meets one doubly used bank group per sixteen lanes); one 16-query block per wave at a time, NT key tiles of accumulators
// (68 registers at 257 tokens).  The M3AE encoder's 257-token attention in f32: 22.7 ms per step on attn_valu_kernel (one query per lane, a 64-deep
// register row, 13 TFLOP/s) -- see DESIGN 6b for what this kernel takes.
// out3 != null: the output row is written as the binary16 triple [hi | lo | hi] (3 D wide), the next GEMM's operand in ARP_MODE_F16X3, instead of f32.
// (The hardware exp2 in place of libm's expf was measured on the encoder's 257-token attention: 7.09 against 7.03-7.47 ms per step -- the softmax is not
//  what this kernel waits for -- at 7x the error of the encoder output, 5.4e-5 against 8.1e-6; not kept.)
template <int NT>
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int D, int heads, float scale,
                                                            int causal, int nq, f16_t* __restrict__ out3 = nullptr) {
    constexpr int HD = 64, LS = 68;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + NT * 16 * LS;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t ld = 3 * (size_t)D;
    const float* base = qkv + (size_t)b * N * ld + h * HD;
    {   // K and V of this (sample, head) -> LDS; rows past N are zero keys (masked below) and zero values.  Every load of the thread is requested before the
        // first LDS store (2 NT float4 in registers): written as load -> store per row, the NT rounds paid a memory latency each -- half of the kernel's time
        float4 kreg[NT], vreg[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = it * 256 + threadIdx.x, t = i >> 4, c4 = (i & 15) * 4;
            kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            vreg[it] = kreg[it];
            if (t < N) {
                kreg[it] = *reinterpret_cast<const float4*>(base + t * ld + D + c4);
                vreg[it] = *reinterpret_cast<const float4*>(base + t * ld + 2 * D + c4);
            }
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = it * 256 + threadIdx.x, t = i >> 4, c4 = (i & 15) * 4;
            *reinterpret_cast<float4*>(Ks + t * LS + c4) = kreg[it];
            *reinterpret_cast<float4*>(Vs + t * LS + c4) = vreg[it];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int q0 = wave * 16; q0 < nq; q0 += 64) {
        // K and V in LDS are invariant over this loop, and left alone the compiler hoists all 2 x 16 NT operand values out of it (544 registers at NT = 17:
        // every VGPR and AGPR, 72 spilled, and a v_accvgpr_read in front of each MFMA).  The fence makes each query block read its operands again.
        asm volatile("" ::: "memory");
        const int qi = q0 + j;
        const float* qrow = base + (size_t)min(qi, N - 1) * ld;
        // The contraction order over d is free as long as both operands agree: MFMA (u, e) takes d = 16 u + 4 g + e from lane group g, so that a lane's
        // four e are ONE float4 of its query row (4 loads up front; read as d = 4 s + g the compiler fetched each of 16 dwords right before its use and
        // waited vmcnt(0) for it) and ONE ds_read_b128 of a key row (68 LDS reads per query block instead of 272).
        float4 q4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            q4[u] = *reinterpret_cast<const float4*>(qrow + 16 * u + 4 * g);
            q4[u].x *= scale; q4[u].y *= scale; q4[u].z *= scale; q4[u].w *= scale;
        }
        f32x4_v acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) acc[kt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 k4[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) k4[kt] = *reinterpret_cast<const float4*>(Ks + (kt * 16 + j) * LS + 16 * u + 4 * g);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kt].x, q4[u].x, acc[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kt].y, q4[u].y, acc[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kt].z, q4[u].z, acc[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kt].w, q4[u].w, acc[kt], 0, 0, 0);
        }
        // softmax over the keys of query j: this lane holds keys kt*16 + 4g + r; the other three lane groups hold the rest
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                if (key >= N || (causal && key > qi)) acc[kt][r] = -INFINITY;
                m = fmaxf(m, acc[kt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = expf(acc[kt][r] - m);  // exp(-inf) = 0 for the masked keys; key 0 is never masked, so m is finite
                acc[kt][r] = p;
                l += p;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        // O^T = V^T . P^T: MFMA (kt, r) contracts over the keys kt*16 + 4 g' + r (g' = the four k-slots); its A operand row i (lane j) is ONE d.  The four
        // output tiles take d = 4 j + dt, so that the lane's four A values are one float4 of the V row: o[dt][reg] <-> d = 16 g + 4 reg + dt for query j.
        f32x4_v o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 v4 = *reinterpret_cast<const float4*>(Vs + (kt * 16 + 4 * g + r) * LS + 4 * j);
                o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, acc[kt][r], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, acc[kt][r], o[1], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, acc[kt][r], o[2], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, acc[kt][r], o[3], 0, 0, 0);
            }
        if (qi < nq) {
            const float inv = 1.0f / l;
            if (out3) {
                f16_t* orow3 = out3 + ((size_t)b * N + qi) * 3 * D + h * HD + 16 * g;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) store_split3(orow3 + 4 * reg, (size_t)D, o[0][reg] * inv, o[1][reg] * inv, o[2][reg] * inv, o[3][reg] * inv);
            } else {
                float* orow = out + ((size_t)b * N + qi) * D + h * HD + 16 * g;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    *reinterpret_cast<float4*>(orow + 4 * reg) = make_float4(o[0][reg] * inv, o[1][reg] * inv, o[2][reg] * inv, o[3][reg] * inv);
            }
        }
    }
}

// ---- (hi, lo) binary16 MFMA kernel (head_dim 64): the attention of the f16x3 encoder mode ----------------------------------------------------------------
// Same plan as the f32 kernel above with every product on v_mfma_f32_16x16x32_f16 as hi.hi + lo.hi + hi.lo (2^-22 relative, three instructions of 16
// cycles per 32-deep step where the f32 form takes eight of 32): K of one (sample, head) in LDS as two binary16 arrays [key][64] (hi, lo), V as two
// TRANSPOSED arrays [d][key]; Q split in registers (softmax scale and log2 e folded in before the split, so the scores come out in the exp2 domain);
// the softmax'd accumulators split in registers into the B operand of O^T = V^T.P^T.  Which keys a 16-row S^T tile holds is free, so tile kt takes
// keys 32 (kt / 2) + 8 (i / 4) + 4 (kt % 2) + i % 4 for its rows i: lane (g, j) then holds, of tiles 2c and 2c+1 together, the EIGHT CONSECUTIVE keys
// 32 c + 8 g .. + 7 of query j -- exactly one lane's share of a 32-deep step of O^T = V^T.P^T, whose A operand becomes ONE ds_read_b128 of a transposed
// V row.  P never moves between lanes.  P is carried times 2^10 (exact; its lo halves stay out of binary16's subnormals down to p = 1e-4) and the row
// sum with it, so the factor cancels in O / l.
// LDS images: rows of 128 B (K) / 16 B x ceil8(4 NC) (V^T), 16-byte chunks XOR-swizzled with a key made of the row bits that tell one lane group's
// sixteen rows apart ((row >> 1 & 1) << 1 | (row >> 4 & 1) << 2 for the permuted K rows, vkey() below for V^T): every fragment read is a conflict-free
// ds_read_b128.  (First version: padded rows of 144 B / 584 B, two 8-byte reads per V fragment -- SQ_LDS_BANK_CONFLICT 29 % of the LDS cycles and more
// LDS-array cycles per query block than MFMA cycles, profiles/r4_x3_pmc.json.)
#define ARP_SPLIT1(X, HI, LO)                  \
    do {                                       \
        const float _x = pin_f32(X);           \
        const _Float16 _h = (_Float16)_x;      \
        HI = _h;                               \
        LO = (_Float16)(_x - (float)_h);       \
    } while (0)
// -DARP_ATTNX3_STAMPS (scripts/attn_x3_stamps.hip only): s_memtime at the phase boundaries of each wave's FIRST query block, 16 slots per wave
#ifdef ARP_ATTNX3_STAMPS
__device__ long long* arp_ax3_stamps = nullptr;
#define ARP_AX3_STAMP(k)                                                                                                              \
    do {                                                                                                                              \
        if (arp_ax3_stamps && (threadIdx.x & 63) == 0 && ((k) < 2 || (k) > 7 || ax3_first))                                             \
            arp_ax3_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define ARP_AX3_STAMP(k) do { } while (0)
#endif
constexpr int attn_x3_vrow_bytes(int nt) { return ((((nt + 1) / 2) * 4 + 7) / 8) * 8 * 16; }
// one partial of the cooperative tail: O (64), max, sum + 2 pad floats -- a 68-float stride keeps every row 16-byte aligned for its float4 stores
constexpr int ATTN_X3_SCR = 68;
constexpr int attn_x3_lds_bytes(int nt) { return 2 * ((nt + 1) / 2) * 32 * 128 + 2 * 64 * attn_x3_vrow_bytes(nt) + ((nt + 1) / 2) * 2 * ATTN_X3_SCR * 4; }  // + the tail block's partials
template <int NT>
__global__ __launch_bounds__(512) void attn_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int D, int heads, float scale,
                                                      int causal, int nq, f16_t* __restrict__ out3) {
    constexpr int HD = 64, NC = (NT + 1) / 2, KROWS = NC * 32, VROWB = attn_x3_vrow_bytes(NT);
    extern __shared__ __attribute__((aligned(16))) char attn_smem[];
    char* Kh = attn_smem;
    char* Kl = Kh + KROWS * 128;
    char* Vh = Kl + KROWS * 128;
    char* Vl = Vh + HD * VROWB;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * N * ld + h * HD;
    [[maybe_unused]] bool ax3_first = true;
    ARP_AX3_STAMP(0);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto kkey = [](int row) { return (((row >> 1) & 1) << 1) | (((row >> 4) & 1) << 2); };
    // V^T rows are 640 B = 0 modulo 32 banks, so the key alone spreads BOTH access patterns: the fragment reads (sixteen rows d = 16 dt + j, chunks g / g + 1 per
    // lane group) and the transposing 4-byte staging stores (rows d = 4 dq + e of eight consecutive dq, four consecutive key pairs each: 32 distinct banks)
    auto vkey = [](int d) { return ((d & 6) ^ (((d >> 3) & 3) << 1)) | ((d >> 2) & 1); };
    // Staging.  EVERY global load of the workgroup's K and V is requested before the first split (NC + 2 ceil(NC / 2) float4 per thread: 72 registers at
    // 257 tokens, nothing else is live yet): written as load -> split -> store loops the staging took ~25 us of a 35 us workgroup -- one memory round trip
    // per iteration, with a single workgroup per CU (153 KB of LDS) and nothing to overlap it with.  Rows beyond N are clamped, then zeroed.
    constexpr int VI = (NC * 256 + 511) / 512;
    float4 kreg[NC], v0reg[VI], v1reg[VI];
#pragma unroll
    for (int i = 0; i < NC; ++i) {  // K rows: 8-byte pieces, two per 16-byte chunk
        const int idx = threadIdx.x + 512 * i, key = idx >> 4, c4 = idx & 15;
        kreg[i] = *reinterpret_cast<const float4*>(base + (size_t)min(key, N - 1) * ld + D + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {  // V: one key PAIR per thread and d-quad, so that a transposed store is one dword (two keys of one d)
        const int idx = min((int)threadIdx.x + 512 * i, NC * 256 - 1), kp = (idx >> 6) * 4 + (idx & 3), dq = (idx & 63) >> 2;  // a wave: 4 key pairs x 16 d-quads
        v0reg[i] = *reinterpret_cast<const float4*>(base + (size_t)min(2 * kp, N - 1) * ld + 2 * D + 4 * dq);
        v1reg[i] = *reinterpret_cast<const float4*>(base + (size_t)min(2 * kp + 1, N - 1) * ld + 2 * D + 4 * dq);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int idx = threadIdx.x + 512 * i, key = idx >> 4, c4 = idx & 15;
        const float4 v = key < N ? kreg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        h4 hi, lo;
        ARP_SPLIT1(v.x, hi[0], lo[0]); ARP_SPLIT1(v.y, hi[1], lo[1]); ARP_SPLIT1(v.z, hi[2], lo[2]); ARP_SPLIT1(v.w, hi[3], lo[3]);
        const int off = key * 128 + (((c4 >> 1) ^ kkey(key)) << 4) + (c4 & 1) * 8;
        *reinterpret_cast<h4*>(Kh + off) = hi;
        *reinterpret_cast<h4*>(Kl + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int idx = threadIdx.x + 512 * i, kp = (idx >> 6) * 4 + (idx & 3), dq = (idx & 63) >> 2;
        if (idx >= NC * 256) break;
        const float4 v0 = 2 * kp < N ? v0reg[i] : make_float4(0.f, 0.f, 0.f, 0.f), v1 = 2 * kp + 1 < N ? v1reg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h2 hi, lo;
            ARP_SPLIT1(a0[e], hi[0], lo[0]);
            ARP_SPLIT1(a1[e], hi[1], lo[1]);
            const int d = 4 * dq + e;
            const int off = d * VROWB + (((kp >> 2) ^ vkey(d)) << 4) + (kp & 3) * 4;
            *reinterpret_cast<h2*>(Vh + off) = hi;
            *reinterpret_cast<h2*>(Vl + off) = lo;
        }
    }
    __syncthreads();
    ARP_AX3_STAMP(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float qs = scale * 1.44269504088896340736f;
    const int krow_j = 8 * (j >> 2) + (j & 3);  // this lane's K row inside a 32-key pair of tiles (+ 4 for the odd tile)
    // A last block of one or two queries that would open a round of its own (257 tokens: 16 full blocks on 8 waves, then ONE query) is not given to one
    // wave while seven idle: every wave takes one 32-key chunk of it (below, after the full blocks) and the partial softmaxes are merged through LDS.
    const int nfull = nq >> 4, tail = nq & 15;
    const bool coop = tail >= 1 && tail <= 2 && nfull > 0 && (nfull & 7) == 0;
    const int nq_blocks = coop ? nfull * 16 : nq;
    // (Requesting the next block's Q rows ahead of this block's sixteen output stores -- one in-order counter, see gemm.h's AdamW epilogue -- was built and
    //  measured: no gain, nine spilled registers at NT = 17.  scripts/attn_x3_stamps.hip says where a workgroup's time goes.)
    for (int q0 = wave * 16; q0 < nq_blocks; q0 += 128) {  // eight waves: two per SIMD, one in its MFMAs while the other is in its softmax / splits
        asm volatile("" ::: "memory");  // K and V are loop invariant: keep the compiler from hoisting the operand reads of every query block out of the loop
        ARP_AX3_STAMP(2);
        const int qi = q0 + j;
        const float* qrow = base + (size_t)min(qi, N - 1) * ld;
        f16x8_v qh[2], ql[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 32 * c + 8 * g), bq = *reinterpret_cast<const float4*>(qrow + 32 * c + 8 * g + 4);
            uint32_t h[4], l[4];
            split2_f16(a.x, a.y, qs, h[0], l[0]); split2_f16(a.z, a.w, qs, h[1], l[1]);
            split2_f16(bq.x, bq.y, qs, h[2], l[2]); split2_f16(bq.z, bq.w, qs, h[3], l[3]);
            qh[c] = __builtin_bit_cast(f16x8_v, u32x4_v{h[0], h[1], h[2], h[3]});
            ql[c] = __builtin_bit_cast(f16x8_v, u32x4_v{l[0], l[1], l[2], l[3]});
        }
        ARP_AX3_STAMP(3);
        f32x4_v acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) acc[kt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
        // K fragments in groups of at most six tiles (48 registers, not the 136 of all seventeen at once: the kernel sits at the 256-register limit of two
        // waves per SIMD); six independent accumulators between an accumulator's own three MFMAs are 96 issue cycles, more than the instruction's latency
        constexpr int KG = 6;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k0 = 0; k0 < NT; k0 += KG) {
                f16x8_v kh[KG], kl[KG];
#pragma unroll
                for (int u = 0; u < KG; ++u) {
                    if (k0 + u >= NT) continue;
                    const int kt = k0 + u, row = 32 * (kt >> 1) + 4 * (kt & 1) + krow_j;
                    const int off = row * 128 + (((4 * c + g) ^ kkey(row)) << 4);
                    kh[u] = *reinterpret_cast<const f16x8_v*>(Kh + off);
                    kl[u] = *reinterpret_cast<const f16x8_v*>(Kl + off);
                }
#pragma unroll
                for (int u = 0; u < KG; ++u)
                    if (k0 + u < NT) acc[k0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[u], qh[c], acc[k0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < KG; ++u)
                    if (k0 + u < NT) acc[k0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[u], qh[c], acc[k0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < KG; ++u)
                    if (k0 + u < NT) acc[k0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[u], ql[c], acc[k0 + u], 0, 0, 0);
            }
        ARP_AX3_STAMP(4);
        // register r of tile kt, lane (g, j): key 32 (kt / 2) + 8 g + 4 (kt % 2) + r of query j
        // (masks only in the tiles that can hold a key >= N -- a uniform branch per tile -- or under a causal mask: compare + select + mask bookkeeping
        //  for every score were 140 vector and 136 scalar instructions of a query block whose last tile alone needs them; scripts/attn_x3_stamps.hip)
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (causal || 32 * (kt >> 1) + 4 * (kt & 1) + 27 >= N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r;
                    if (key >= N || (causal && key > qi)) acc[kt][r] = -INFINITY;
                }
            }
            m = fmaxf(fmaxf(fmaxf(acc[kt][0], acc[kt][1]), fmaxf(acc[kt][2], acc[kt][3])), m);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float m10 = m - 10.f;  // P times 2^10
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(acc[kt][r] - m10);  // 2^-inf = 0 for the masked keys; key 0 is never masked, so m is finite
                acc[kt][r] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        ARP_AX3_STAMP(5);
        f32x4_v o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t phu[4] = {0u, 0u, 0u, 0u}, plu[4] = {0u, 0u, 0u, 0u};  // keys 32 c + 8 g .. + 7
            split2_f16(acc[2 * c][0], acc[2 * c][1], 1.0f, phu[0], plu[0]);
            split2_f16(acc[2 * c][2], acc[2 * c][3], 1.0f, phu[1], plu[1]);
            if (2 * c + 1 < NT) {
                split2_f16(acc[2 * c + 1][0], acc[2 * c + 1][1], 1.0f, phu[2], plu[2]);
                split2_f16(acc[2 * c + 1][2], acc[2 * c + 1][3], 1.0f, phu[3], plu[3]);
            }
            const f16x8_v ph = __builtin_bit_cast(f16x8_v, u32x4_v{phu[0], phu[1], phu[2], phu[3]});
            const f16x8_v pl = __builtin_bit_cast(f16x8_v, u32x4_v{plu[0], plu[1], plu[2], plu[3]});
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + j;
                const int off = d * VROWB + (((4 * c + g) ^ vkey(d)) << 4);
                const f16x8_v vh = *reinterpret_cast<const f16x8_v*>(Vh + off), vl = *reinterpret_cast<const f16x8_v*>(Vl + off);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[dt], 0, 0, 0);
            }
        }
        ARP_AX3_STAMP(6);
        if (qi < nq) {
            const float inv = 1.0f / l;
            // lane (g, j): query j, d = 16 dt + 4 g + r
            if (out3) {
                f16_t* orow3 = out3 + ((size_t)b * N + qi) * 3 * D + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store_split3(orow3 + 16 * dt, (size_t)D, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            } else {
                float* orow = out + ((size_t)b * N + qi) * D + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<float4*>(orow + 16 * dt) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            }
        }
        ARP_AX3_STAMP(7);
        ax3_first = false;
    }
    ARP_AX3_STAMP(8);
    if (!coop) return;
    // ---- the short last block, one 32-key chunk per wave: partial (max, sum, O) per chunk and query -> LDS -> fixed-order merge ----------------------
    float* scr = reinterpret_cast<float*>(Vl + HD * VROWB);  // [NC][2][ATTN_X3_SCR]: O (64), max, sum, pad
    {
        const int q0 = nfull * 16, qi = q0 + j;
        const float* qrow = base + (size_t)min(qi, N - 1) * ld;
        f16x8_v qh[2], ql[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 32 * c + 8 * g), bq = *reinterpret_cast<const float4*>(qrow + 32 * c + 8 * g + 4);
            const float v[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, bq.x * qs, bq.y * qs, bq.z * qs, bq.w * qs};
#pragma unroll
            for (int e = 0; e < 8; ++e) ARP_SPLIT1(v[e], qh[c][e], ql[c][e]);
        }
        for (int c = wave; c < NC; c += 8) {
            f32x4_v a2[2] = {f32x4_v{0.f, 0.f, 0.f, 0.f}, f32x4_v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (2 * c + t >= NT) continue;
                    const int row = 32 * c + 4 * t + krow_j;
                    const int off = row * 128 + (((4 * cc + g) ^ kkey(row)) << 4);
                    const f16x8_v kh = *reinterpret_cast<const f16x8_v*>(Kh + off), kl = *reinterpret_cast<const f16x8_v*>(Kl + off);
                    a2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[cc], a2[t], 0, 0, 0);
                    a2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[cc], a2[t], 0, 0, 0);
                    a2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[cc], a2[t], 0, 0, 0);
                }
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 32 * c + 8 * g + 4 * t + r;
                    if (2 * c + t >= NT || key >= N || (causal && key > qi)) a2[t][r] = -INFINITY;
                    m = fmaxf(m, a2[t][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float m10 = (m == -INFINITY) ? 0.f : m - 10.f;  // a chunk with no visible key contributes nothing: P = 2^-inf = 0
            float l = 0.f;
            f16x8_v ph, pl;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(a2[t][r] - m10);
                    l += pv;
                    ARP_SPLIT1(pv, ph[4 * t + r], pl[4 * t + r]);
                }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            float* dst = scr + (c * 2 + j) * ATTN_X3_SCR;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + j;
                const int off = d * VROWB + (((4 * c + g) ^ vkey(d)) << 4);
                const f16x8_v vh = *reinterpret_cast<const f16x8_v*>(Vh + off), vl = *reinterpret_cast<const f16x8_v*>(Vl + off);
                f32x4_v o = {0.f, 0.f, 0.f, 0.f};
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o, 0, 0, 0);
                if (j < tail) *reinterpret_cast<float4*>(dst + 16 * dt + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (j < tail && g == 0) { dst[64] = m; dst[65] = l; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < tail * 16) {
        const int jq = threadIdx.x >> 4, d4 = (threadIdx.x & 15) * 4;
        float M = -INFINITY;
        for (int c = 0; c < NC; ++c) M = fmaxf(M, scr[(c * 2 + jq) * ATTN_X3_SCR + 64]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < NC; ++c) {  // fixed order
            const float* src = scr + (c * 2 + jq) * ATTN_X3_SCR;
            const float w = __builtin_amdgcn_exp2f(src[64] - M);  // 0 for an empty chunk (its max is -inf; M is finite: key 0 is visible to every query)
            L += w * src[65];
#pragma unroll
            for (int e = 0; e < 4; ++e) O[e] += w * src[d4 + e];
        }
        const float inv = 1.0f / L;
        const int qi = nfull * 16 + jq;
        if (out3) store_split3(out3 + ((size_t)b * N + qi) * 3 * D + h * HD + d4, (size_t)D, O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv);
        else *reinterpret_cast<float4*>(out + ((size_t)b * N + qi) * D + h * HD + d4) = make_float4(O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv);
    }
    ARP_AX3_STAMP(9);
}
#undef ARP_SPLIT1

// ---- MFMA kernel (bf16, head_dim 64) -------------------------------------------------------------
// NT = number of 16-key tiles (keys padded to a multiple of 32, i.e. NT even).
// LDS: K [NT*16 keys][128 B] and V [NT*16 keys][128 B], both row-major with the 16-byte chunk index XOR-swizzled
// by (key & 7).  K fragments are plain ds_read_b128 row reads; the V^T fragments of O^T = V^T . P^T are fetched
// with the CDNA4 transposing read ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, delivered column-major),
// so V is staged exactly like K -- no 2-byte transposing stores.
typedef __attribute__((ext_vector_type(4))) short tr_b64_v;

#ifdef ARP_ATTN_STAMPS
__device__ long long* arp_attn_stamps = nullptr;  // scripts/attn_bench.hip: per-workgroup, per-wave cycles per phase
#define AT_T() __builtin_amdgcn_s_memtime()
#define AT_ACC(i) { const long long t_ = AT_T(); at_[i] += t_ - at_last; at_last = t_; }
#else
#define AT_ACC(i)
#endif

template <typename T, int NT>
__global__ __launch_bounds__(256, (NT <= 4 ? 5 : 2)) void attn_mfma_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int D,
                                                        int heads, float scale, int causal, int nq, float out8 = 0.f, int outc = 0) {
    // out8 != 0: `out` is an e4m3 buffer [B*N, D] bytes and receives out8 * value (operand of an fp8 out_proj, tower.h)
    // outc != 0 (T = f16, ARP_MODE_F16C): `out` rows are [hi | x4 | dx4] (3 D bytes each, common.h::store_f16c) -- out_proj's MIXC operand
    // outc & 4 (round 6): no dx4 segment (the consumer corrects its weight rounding only -- ARP_F16C_PLAN digit 2 < 2 -- and never reads it)
    // (outc & 3) == 2 (round 6): the caller has permuted V's columns inside every head, column d' of the V block = original column pi(d'), pi = the swap of bits [5:4]
    //   and [3:2] of d (arp_enc.hip::vperm64).  O^T = V^T.P^T leaves lane (fg, fr) with the d' = 16 dt + 4 fg + r of query fr -- a layout the MFMA fixes -- and
    //   those are then the ORIGINAL columns 16 fg + 4 dt + r: sixteen consecutive ones.  A query's head slice leaves as four 32-byte pieces (+ 8 + 8 bytes of
    //   e2m1) instead of sixteen 8-byte pieces (+ sixteen 2 + 2): 4 stores per lane and block where the unpermuted form issues 12.  Same values, same bits:
    //   every output element is the same chain over the keys, computed by another lane.
    constexpr int NP = NT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NP * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t ld = 3 * (size_t)D;
    const T* base = qkv + (size_t)b * N * ld + h * 64;
#ifdef ARP_ATTN_STAMPS
    long long at_[5] = {0, 0, 0, 0, 0};
    long long at_last = AT_T();
#endif

    // K and V images by LDS-DMA, every piece in flight at once: a wave instruction fills 8 rows (1 KiB); lane -> (row, PHYSICAL 16-byte
    // chunk), which holds logical chunk (physical ^ (key & 7)), so the swizzle sits on the source address.  (Staged through registers
    // -- load, then ds_write -- the 66 KB of an N = 257 head took 15 k of the workgroup's 48 k cycles: the loads went out a few at a
    // time.)  Pad keys (>= N) alias key N-1: they are masked out of the softmax, and their probabilities are exactly 0 in P.V.
    {
        const int prow = lane >> 3, pch = lane & 7;
        for (int p = wave; p < NP / 8; p += 4) {
            const int key = p * 8 + prow;
            const int kk = key < N ? key : N - 1;
            const T* src = base + (size_t)kk * ld + ((pch ^ (key & 7)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + D),
                                             (__attribute__((address_space(3))) void*)(Ks + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * D),
                                             (__attribute__((address_space(3))) void*)(Vs + p * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    AT_ACC(0);

    const int fr = lane & 15, fg = lane >> 4;
    const int nqb = (nq + 15) >> 4;  // query rows >= nq are not produced (last ViT block: only the class token is read)
    const float c2 = scale * 1.4426950408889634f;  // exp(x*scale) = exp2(x*c2)
    // transposing-read addressing: lane 4q+p of a 16-lane group points at row (key0 + q), columns 4p..4p+3 of the
    // 16-column block [16*dt, 16*dt+16); lane i receives column i of the four rows
    const int trq = fr >> 2, trp = fr & 3;
    // (Two query blocks per wave at a time, so that each K / V^T fragment read feeds two MFMAs: 245-256 VGPRs plus accumulation
    //  registers -- one workgroup per CU instead of two, 100 us against 51 us at N = 197.  The full score row of a block, NT x 4
    //  registers, is what makes this kernel register-bound.)
    // gridDim.y > 1 (the single-frame tower: 12 workgroups would otherwise walk 13 query blocks on 4 waves each): the query blocks are
    // dealt over gridDim.y workgroups, each of which stages the whole K / V image of its head
    for (int qb = wave + 4 * (int)blockIdx.y; qb < nqb; qb += 4 * (int)gridDim.y) {
        int qrow = qb * 16 + fr;
        const int qvalid = qrow < N;
        qrow = qvalid ? qrow : N - 1;
        // (round 5: the next block's Q fragment requested one block ahead was neutral at N = 197 and costs 8 registers that the N = 257 instance does not have)
        u32x4_v qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const u32x4_v*>(base + qrow * ld + ks * 32 + fg * 8);

        // S^T tile kt: rows = keys 16*kt + 4*fg + r, col = query fr
        f32x4_v s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
            const int krow = kt * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4_v kf = *reinterpret_cast<const u32x4_v*>(Ks + krow * 128 + (((ks * 4 + fg) ^ (krow & 7)) << 4));
                s[kt] = mfma16<T>(kf, qf[ks], s[kt]);
            }
            // at most six key tiles' fragment reads ahead of their MFMAs: left alone, hipcc issues all 2 NT of them first -- 144 registers at NT = 18 (N = 257, the
            // M3AE encoder), twelve of which went LDS -> scratch -> register on their way to the MFMA (65 spilled registers in that instance; round 5, ISA)
            if constexpr (NT >= 14) {
                if ((kt % 6) == 5) __builtin_amdgcn_sched_barrier(0);
            }
        }
        AT_ACC(1);
        const int qidx = qb * 16 + fr;
        const int klim = causal ? (qidx < N ? qidx + 1 : N) : N;
        float mx = -INFINITY;
        // Without the causal mask only the pad keys (>= N) are hidden, and those sit in the last two key tiles (NT = 2 ceil(N / 32)): the other tiles skip the
        // compare + select per score -- the softmax, not the MFMAs, is what this kernel is bound by (56 scores per lane at N = 197).  Same values either way.
        if (!causal) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt >= NT - 2) {
                        const int key = kt * 16 + fg * 4 + r;
                        s[kt][r] = (key < N) ? s[kt][r] : -INFINITY;
                    }
                    mx = fmaxf(mx, s[kt][r]);
                }
        } else {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fg * 4 + r;
                const float v = (key < klim) ? s[kt][r] : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mc));  // masked keys: 2^-inf = 0
                s[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        AT_ACC(2);

        // O^T[d][q] = sum_key V^T[d][key] * P^T[key][q].  k-slot (fg, j) of step st <-> key 32*st + 16*(j>>2) + 4*fg + (j&3):
        // the B operand comes straight from s[2st], s[2st+1]; the A operand is two transposing reads of V.
        f32x4_v o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NT / 2; ++st) {
            u32x4_v pb;
            pb[0] = pack2<T>(s[2 * st][0], s[2 * st][1]);
            pb[1] = pack2<T>(s[2 * st][2], s[2 * st][3]);
            pb[2] = pack2<T>(s[2 * st + 1][0], s[2 * st + 1][1]);
            pb[3] = pack2<T>(s[2 * st + 1][2], s[2 * st + 1][3]);
            const int k0 = 32 * st + 4 * fg + trq, k1 = k0 + 16;  // the row this lane addresses in each block
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int ch = 2 * dt + (trp >> 1);
                const tr_b64_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) tr_b64_v*)(Vs + k0 * 128 + ((ch ^ (k0 & 7)) << 4) + (trp & 1) * 8));
                const tr_b64_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) tr_b64_v*)(Vs + k1 * 128 + ((ch ^ (k1 & 7)) << 4) + (trp & 1) * 8));
                const u32x2_v l2 = __builtin_bit_cast(u32x2_v, lo), h2 = __builtin_bit_cast(u32x2_v, hi);
                const u32x4_v va = {l2[0], l2[1], h2[0], h2[1]};
                o[dt] = mfma16<T>(va, pb, o[dt]);
            }
        }
        AT_ACC(3);
        if (qvalid && qidx < nq) {
            if (out8 != 0.f) {
                fp8_t* orow = reinterpret_cast<fp8_t*>(out) + ((size_t)b * N + qidx) * D + h * 64;
                const float sc = inv * out8;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16 + fg * 4, o[dt][0] * sc, o[dt][1] * sc, o[dt][2] * sc, o[dt][3] * sc);
            } else if (NT > 4 && (outc & 3) == 2) {  // (compiled out of the 96-register instances, NT <= 4: the sixteen-value store made them spill 22-36 registers whatever outc is at run time)
                if constexpr (__is_same(T, f16_t) && NT > 4) {
                    f16_t* orow = out + ((size_t)b * N + qidx) * 3 * D / 2;
                    float v16[16];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v16[4 * dt + r] = o[dt][r] * inv;
                    store_f16c16(orow, h * 64 + fg * 16, D, v16, !(outc & 4));
                }
            } else if (outc) {
                if constexpr (__is_same(T, f16_t)) {
                    f16_t* orow = out + ((size_t)b * N + qidx) * 3 * D / 2;
                    if (NT > 4 && (outc & 4)) {  // (the 96-register instances keep the one form: the second loop cost them 20 spilled registers)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            store_f16c<false>(orow, h * 64 + dt * 16 + fg * 4, D, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
                    } else {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            store_f16c<true>(orow, h * 64 + dt * 16 + fg * 4, D, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
                    }
                }
            } else {
                T* orow = out + ((size_t)b * N + qidx) * D + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    store4(orow + dt * 16 + fg * 4, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            }
        }
        AT_ACC(4);
    }
#ifdef ARP_ATTN_STAMPS
    if (arp_attn_stamps && lane == 0) {
        long long* d = arp_attn_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) d[i] = at_[i];
    }
#endif
}

}  // namespace arp

// Fused QKV projection + attention kernel (see qkvattn.h), compiled as its own translation unit.
#include "qkvattn.h"

#include <cstdlib>

#include "attention.h"
#include "gemm.h"
#include "gemm256.h"
#include "kloop192.h"

namespace arp {

constexpr int QA_THREADS = 512;
constexpr int QA_BUF_BYTES = K192_BUF_BYTES;
constexpr int QA_KV_ROWS = 320;                  // k / v images: 256 tile rows + 64 zero rows (the last frame's padded keys)
constexpr int QA_K_OFF = 256 * 128;
constexpr int QA_V_OFF = QA_K_OFF + QA_KV_ROWS * 128;
static_assert(QA_V_OFF + QA_KV_ROWS * 128 <= 2 * QA_BUF_BYTES, "q/k/v images must fit the K-tile ring");
constexpr int QA_LDS_BYTES = 2 * QA_BUF_BYTES + 192 * 4;

#ifdef ARP_QA_STAMPS
__device__ long long* arp_qa_stamps = nullptr;  // scripts/qkvattn_bench.hip: per-workgroup, per-wave phase time stamps
#define QA_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define QA_STAMP(i)
#endif

template <typename T>
__global__ __launch_bounds__(QA_THREADS, 2) void qkv_attn_kernel(QkvAttnArgs g) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPB = 64, EPC = 8;
    constexpr int NT = 4;  // 16-key tiles: N <= 64

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- tile: FPT frames x one head; XCD-contiguous ranges, groups of 8 frame-tiles walked head by head ----------------
    const int m_tiles = (g.B + g.fpt - 1) / g.fpt;
    const int n_tiles = g.heads;
    int mt, head;
    {
        int t = xcd_remap(blockIdx.x, m_tiles * n_tiles);
        const int group_m = g.group_m > 0 ? g.group_m : G2_GROUP_M;
        const int per_group = group_m * n_tiles;
        const int grp = t / per_group;
        const int first_m = grp * group_m;
        const int gsize = min(m_tiles - first_m, group_m);
        t -= grp * per_group;
        mt = first_m + t % gsize;
        head = t / gsize;
    }
    const int f0 = mt * g.fpt;
    const int nf = min(g.fpt, g.B - f0);
    const int m0 = f0 * g.N;
    const int rows = nf * g.N;  // valid rows of the tile
    const int Mtot = g.B * g.N;
    const int n0 = head * 192;

    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.W);

#ifdef ARP_QA_STAMPS
    long long st_[5];
#endif
    QA_STAMP(0);
    f32x4_v acc[2][3][4];  // [mq][ni][mi]
    kloop_256x192<T>(smem, A, W, g.bias, g.lda, g.ldw, m0, n0, Mtot, g.heads * 192, g.K, acc);
    QA_STAMP(1);
    const int fr = lane & 15, fg = lane >> 4;
    float* bias_s = reinterpret_cast<float*>(smem + K192_RING_BYTES);

    // ---- q | k | v images -------------------------------------------------------------------------------------------------
    // every LDS byte of the ring is dead: rows [0,256) of three 128-B-per-row images, 16-B chunks XOR-swizzled by (row & 7)
    for (int i = tid; i < 2 * 64 * 8; i += QA_THREADS) {
        const int img = i >> 9, rr = (i >> 3) & 63, ch = i & 7;
        *reinterpret_cast<u32x4_v*>(smem + (img ? QA_V_OFF : QA_K_OFF) + (256 + rr) * 128 + ch * 16) = u32x4_v{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wr * 128 + mq * 64 + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                const int col = wc * 48 + ni * 16 + fg * 4;
                const int img = col >> 6, d = col & 63;
                const float4 b = *reinterpret_cast<const float4*>(bias_s + col);
                const f32x4_v a4 = acc[mq][ni][mi];
                const int off = (img == 0 ? 0 : (img == 1 ? QA_K_OFF : QA_V_OFF)) + row * 128 + (((d >> 3) ^ (row & 7)) << 4) + (d & 7) * 2;
                *reinterpret_cast<uint2*>(smem + off) = make_uint2(pack2<T>(a4[0] + b.x, a4[1] + b.y), pack2<T>(a4[2] + b.z, a4[3] + b.w));
            }
        }
    __syncthreads();
    QA_STAMP(2);

    // ---- attention from LDS: one (frame, 16-query block) per wave at a time (code of attn_mfma_kernel) ----------------------
    // (Three units per wave walked stage by stage, so that their dependent chains interleave, measured 7.4 k cycles for the phase
    //  against 6.5 k: the phase is bound by issue throughput -- LDS reads, exp, MFMAs of two resident waves -- not by one chain's latency.)
    const char* Ks = smem + QA_K_OFF;
    const char* Vs = smem + QA_V_OFF;
    const int N = g.N;
    const int nqb = (g.nq + 15) >> 4;
    const float c2 = g.scale * 1.4426950408889634f;
    const int trq = fr >> 2, trp = fr & 3;
    for (int u = wave; u < nf * nqb; u += 8) {
        const int f = u / nqb, qb = u - f * nqb;
        const int R0 = f * N;
        const int qidx = qb * 16 + fr;
        const bool qvalid = qidx < N;
        const int qabs = R0 + (qvalid ? qidx : N - 1);
        u32x4_v qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const u32x4_v*>(smem + qabs * 128 + (((ks * 4 + fg) ^ (qabs & 7)) << 4));
        f32x4_v s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4_v{0.f, 0.f, 0.f, 0.f};
            const int krow = R0 + kt * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4_v kf = *reinterpret_cast<const u32x4_v*>(Ks + krow * 128 + (((ks * 4 + fg) ^ (krow & 7)) << 4));
                s[kt] = mfma16<T>(kf, qf[ks], s[kt]);
            }
        }
        const int klim = g.causal ? (qidx < N ? qidx + 1 : N) : N;
        float mx = -INFINITY;
        // Without the causal mask only the pad keys (>= N) are hidden: a tile whose sixteen keys are all real skips the compare + select per score (a
        // wave-uniform test per tile) -- the softmax, not the MFMAs, is what this kernel is bound by.  Same values either way.  (NT is fixed at 4 here while
        // N may be anything in 1..64: the test is on the tile's last key, not on the tile index -- with N < 32 tiles 0 and 1 hold pad keys too, whose LDS
        // rows are the NEXT frame's keys.  ADVICE r5.)
        if (!g.causal) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt * 16 + 15 >= N) {
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
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mc));
                s[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

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
            const int k0 = R0 + 32 * st + 4 * fg + trq, k1 = k0 + 16;
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
        // the block's output rows replace its own query rows in the q image (no other wave reads them)
        if (qvalid && qidx < g.nq) {
            const int row = R0 + qidx;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + fg * 4;
                *reinterpret_cast<uint2*>(smem + row * 128 + (((d >> 3) ^ (row & 7)) << 4) + (d & 7) * 2) =
                    make_uint2(pack2<T>(o[dt][0] * inv, o[dt][1] * inv), pack2<T>(o[dt][2] * inv, o[dt][3] * inv));
            }
        }
    }
    QA_STAMP(3);
    __syncthreads();
    // ---- whole 128-B row segments out ---------------------------------------------------------------------------------------
    T* out = static_cast<T*>(g.out);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 64 + (tid >> 3), ch = tid & 7;
        if (row < rows && (row % N) < g.nq) {
            const u32x4_v v = *reinterpret_cast<const u32x4_v*>(smem + row * 128 + ((ch ^ (row & 7)) << 4));
            *reinterpret_cast<u32x4_v*>(out + (size_t)(m0 + row) * g.ldo + head * 64 + ch * 8) = v;
        }
    }
#ifdef ARP_QA_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QA_STAMP(4);
    if (arp_qa_stamps && lane == 0) {
        long long* d = arp_qa_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) d[i] = st_[i];
    }
#endif
}

template <typename T>
static int launch_qkv_attn_impl(QkvAttnArgs g, hipStream_t stream) {
    if (g.B <= 0) return 0;
    if (!qkv_attn_supported(g.N, g.heads * 64, g.heads, (int)sizeof(T)) || g.K % 64 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0 || g.ldo % 8 != 0)
        return fail("qkv_attn: unsupported shape N=" + std::to_string(g.N) + " heads=" + std::to_string(g.heads) + " K=" + std::to_string(g.K));
    g.fpt = 256 / g.N;
    if (g.nq <= 0 || g.nq > g.N) g.nq = g.N;
    g.scale = 1.0f / sqrtf(64.0f);
    static const int group_env = getenv("ARP_QA_GROUP_M") ? atoi(getenv("ARP_QA_GROUP_M")) : 0;
    if (group_env > 0) g.group_m = group_env;
    auto kern = qkv_attn_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS_BYTES));
        attr_set = true;
    }
    const int m_tiles = (g.B + g.fpt - 1) / g.fpt;
    hipLaunchKernelGGL(kern, dim3(m_tiles * g.heads), dim3(QA_THREADS), QA_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}


int launch_qkv_attn_f16(QkvAttnArgs g, hipStream_t stream) { return launch_qkv_attn_impl<f16_t>(g, stream); }
int launch_qkv_attn_bf16(QkvAttnArgs g, hipStream_t stream) { return launch_qkv_attn_impl<bf16_t>(g, stream); }

}  // namespace arp

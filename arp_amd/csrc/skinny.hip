// Latency-path kernels: the W-tiled skinny GEMM and the slab-reduce + residual + LayerNorm row kernel (skinny.h).
#include "skinny.h"

#include "rowops.h"

namespace arp {

namespace {

__device__ __forceinline__ float act_rt(int act, float x) {
    switch (act) {
        case ACT_QGELU: return apply_act<ACT_QGELU>(x);
        case ACT_RELU: return apply_act<ACT_RELU>(x);
        case ACT_TANH: return apply_act<ACT_TANH>(x);
        case ACT_GELU_TANH: return apply_act<ACT_GELU_TANH>(x);
        default: return x;
    }
}

// mean / rstd of the workgroup's 16 MT rows from the producers' per-strip sums, into LDS [rows][2] (consumer of a folded LayerNorm).
// Eight lanes per row, each with every eighth strip (independent loads: a one-thread-per-row loop over 48 strips walked them one L2
// round trip at a time), then a fixed-order butterfly over the eight lanes.
template <int MT>
__device__ __forceinline__ void skinny_row_stats(const SkinnyArgs& g, int mt0, float* lnst) {
    const int sub = threadIdx.x & 7;
    for (int r = threadIdx.x >> 3; r < MT * 16; r += blockDim.x >> 3) {
        int m = mt0 * 16 + r;
        m = m < g.M ? m : g.M - 1;
        const float2* st = reinterpret_cast<const float2*>(g.ln_stats) + (size_t)m * g.ln_parts;
        float2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (sub + 8 * i < g.ln_parts) ? st[sub + 8 * i] : make_float2(0.f, 0.f);
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { su += v[i].x; sq += v[i].y; }
        for (int p = sub + 64; p < g.ln_parts; p += 8) { su += st[p].x; sq += st[p].y; }  // more than 64 strips (width > 1024)
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { su += __shfl_xor(su, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (sub == 0) {
            const float mu = su * g.ln_inv_d;
            lnst[2 * r] = mu;
            lnst[2 * r + 1] = 1.0f / sqrtf(fmaxf(sq * g.ln_inv_d - mu * mu, 0.f) + g.ln_eps);
        }
    }
}

// One output fragment: row m (this lane), columns n .. n + 3.  EVERY lane of the wave must call it (the statistics of a producer are
// reduced across the four lane groups of a row); rows >= M store nothing.  lrow: the row's index inside the workgroup (for lnst).
template <typename T>
__device__ __forceinline__ void skinny_store(const SkinnyArgs& g, int slice, int m, int n, f32x4_v v, const float* lnst, int lrow) {
    const bool live = m < g.M;
    if (gridDim.y > 1) {
        if (live) *reinterpret_cast<f32x4_v*>(static_cast<float*>(g.out) + (size_t)slice * g.slice_stride + (size_t)m * g.ldo + n) = v;
        return;
    }
    float r[4] = {v[0], v[1], v[2], v[3]};
    if (g.ln_stats) {
        const float mu = lnst[2 * lrow], rs = lnst[2 * lrow + 1];
        const float4 c4 = *reinterpret_cast<const float4*>(g.ln_c + n);
        r[0] = rs * (r[0] - mu * c4.x); r[1] = rs * (r[1] - mu * c4.y); r[2] = rs * (r[2] - mu * c4.z); r[3] = rs * (r[3] - mu * c4.w);
    }
    if (g.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
        r[0] += bb.x; r[1] += bb.y; r[2] += bb.z; r[3] += bb.w;
    }
    if (g.act != ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = act_rt(g.act, r[i]);
    }
    if (g.out_f32) {
        if (g.resid && live) {
            const float4 q = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n);
            r[0] += q.x; r[1] += q.y; r[2] += q.z; r[3] += q.w;
        }
        if (live) store4(static_cast<float*>(g.out) + (size_t)m * g.ldo + n, r[0], r[1], r[2], r[3]);
        if (g.stats_out) {  // producer of the next LayerNorm: the operand copy and this strip's (sum, sum of squares) per row
            if (live) store4(static_cast<T*>(g.xb) + (size_t)m * g.ldxb + n, r[0], r[1], r[2], r[3]);
            float su = live ? (r[0] + r[1]) + (r[2] + r[3]) : 0.f;
            float sq = live ? (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]) : 0.f;
            su += __shfl_xor(su, 16, 64); sq += __shfl_xor(sq, 16, 64);
            su += __shfl_xor(su, 32, 64); sq += __shfl_xor(sq, 32, 64);
            if (live && (threadIdx.x & 48) == 0) {
                float* st = g.stats_out + ((size_t)m * (g.N >> 4) + (n >> 4)) * 2;
                st[0] = su;
                st[1] = sq;
            }
        }
    } else if (live) {
        store4(static_cast<T*>(g.out) + (size_t)m * g.ldo + n, r[0], r[1], r[2], r[3]);
    }
}

// MT = 16-row m-tiles and NT = 16-column n-tiles per workgroup (every wave computes all MT x NT of them over its own K range),
// KS = MFMA k-steps (32 elements) per chunk; a wave walks chunks of 32 * KS elements of its K range.
// blockIdx.x: n-group (16 NT columns), blockIdx.y: cross-workgroup K slice, blockIdx.z: m-group (16 MT rows).
//   NT = 1: the one-frame shapes (<= 64 rows): (64 + 16) x K operand bytes per workgroup, N / 16 workgroups.
//   NT = 4, MT = 4: 65 .. 256 rows (ViT-B/16's 197): 64 x 64 tiles, (64 + 64) x K bytes per workgroup instead of the (208 + 16) x K
//     of one 16-column strip over all rows -- a CU fills its L1 at ~45 KB/us whatever the access pattern (scripts/skinny_bench.hip),
//     so operand bytes per workgroup ARE the kernel's time above the launch floor.
template <typename T, int MT, int NT, int KS>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(SkinnyArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lnst = reinterpret_cast<float*>(smem);                     // [MT * 16][2]: mean, rstd (folded-LayerNorm consumer)
    f32x4_v* red = reinterpret_cast<f32x4_v*>(smem + 1024);  // [NW][MT * NT][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int n0 = blockIdx.x * 16 * NT;
    const int mt0 = blockIdx.z * MT;
    const int slice = blockIdx.y;
    const int kslice = g.K / (int)gridDim.y;
    const int kw = kslice / NW;
    const int fr = lane & 15, fg = lane >> 4;
    const int kbase = slice * kslice + wave * kw + fg * 8;
    if (g.ln_stats) skinny_row_stats<MT>(g, mt0, lnst);
    const T* __restrict__ wp[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) wp[u] = static_cast<const T*>(g.W) + (size_t)(n0 + u * 16 + fr) * g.ldw + kbase;
    const T* __restrict__ ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = (mt0 + t) * 16 + fr;
        m = m < g.M ? m : g.M - 1;  // rows past M: computed on valid memory, never stored
        ap[t] = static_cast<const T*>(g.A) + (size_t)m * g.lda + kbase;
    }
    f32x4_v acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < kw; c += 32 * KS) {
        // every load of the chunk is issued before the first MFMA (left to itself the compiler keeps ~7 in flight and interleaves the
        // rest behind waits: two or three memory round trips instead of one)
        u32x4_v wf[NT][KS], af[MT][KS];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < KS; ++j) wf[u][j] = *reinterpret_cast<const u32x4_v*>(wp[u] + c + 32 * j);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < KS; ++j) af[t][j] = *reinterpret_cast<const u32x4_v*>(ap[t] + c + 32 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = mfma16<T>(wf[u][j], af[t][j], acc[t][u]);  // swapped: D[n = 4 fg + i][m = fr]
    }
    // per (m-tile t, n-tile u) a lane holds 4 consecutive output columns n0 + 16 u + 4 fg .. + 3 of row 16 (mt0 + t) + fr
    auto epilogue = [&](int f, f32x4_v v) {
        const int t = f / NT, u = f - t * NT;
        skinny_store<T>(g, slice, (mt0 + t) * 16 + fr, n0 + u * 16 + fg * 4, v, lnst, t * 16 + fr);
    };
    if (NW == 1) {
        if (g.ln_stats) __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) epilogue(t * NT + u, acc[t][u]);
        return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) red[(wave * (MT * NT) + t * NT + u) * 64 + lane] = acc[t][u];
    __syncthreads();
    for (int f = wave; f < MT * NT; f += NW) {
        f32x4_v s = red[f * 64 + lane];
        for (int v = 1; v < NW; ++v) s += red[(v * (MT * NT) + f) * 64 + lane];  // fixed order: waves 0 .. NW-1
        epilogue(f, s);
    }
}

// ---- the same product with COALESCED operand loads --------------------------------------------------------------------------------
// scripts/fill_bench.hip: one CU fills from L2 at 18 B/clk with the fragment gather above (16 rows x 64 B per wave instruction) and at
// 69 B/clk with 1 KiB-contiguous wave instructions -- so above the launch floor the gather IS the kernel's time.  Here every wave
// brings its K range in as 64-element blocks by LDS-DMA (8 rows x 128 B per instruction, the 16-byte chunk index XOR-swizzled on the
// source side like gemm.h) into its OWN double-buffered LDS image -- no workgroup barrier in the loop, only the wave's counted vmcnt --
// and reads the MFMA fragments back with ds_read_b128.  K % (64 x waves) == 0; other shapes keep the gather kernel.
template <typename T, int MT, int NT>
__global__ __launch_bounds__(512) void skinny_lds_kernel(SkinnyArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = (MT + NT) * 16, GROUPS = ROWS / 8, REGION = ROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int n0 = blockIdx.x * 16 * NT;
    const int mt0 = blockIdx.z * MT;
    const int slice = blockIdx.y;
    const int kslice = g.K / (int)gridDim.y;
    const int kw = kslice / NW, nblk = kw >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int kbase = slice * kslice + wave * kw;
    float* lnst = reinterpret_cast<float*>(smem);  // [MT * 16][2]: mean, rstd (folded-LayerNorm consumer); the images start 1 KiB in
    char* my = smem + 1024 + (size_t)wave * 2 * REGION;
    // LDS-DMA sources: lane -> (row-in-group = lane / 8, PHYSICAL chunk lane % 8 holding logical chunk (lane % 8) ^ (row & 7))
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const T* __restrict__ src[GROUPS];
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) {
        const int r = q * 8 + srow;
        if (r < MT * 16) {
            int m = mt0 * 16 + r;
            m = m < g.M ? m : g.M - 1;  // rows past M: computed on valid memory, never stored
            src[q] = static_cast<const T*>(g.A) + (size_t)m * g.lda + kbase + schunk * 8;
        } else {
            src[q] = static_cast<const T*>(g.W) + (size_t)(n0 + r - MT * 16) * g.ldw + kbase + schunk * 8;
        }
    }
    auto stage = [&](int buf, int b) {
#pragma unroll
        for (int q = 0; q < GROUPS; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + b * 64),
                                             (__attribute__((address_space(3))) void*)(my + buf * REGION + q * 1024), 16, 0, 0);
    };
    f32x4_v acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    if (nblk > 1) stage(1, 1);
    // (behind the first operand blocks in the memory queue, not in front of them: the row statistics are needed at the epilogue only)
    if (g.ln_stats) skinny_row_stats<MT>(g, mt0, lnst);
    const int sw = fr & 7;
    for (int b = 0; b < nblk; ++b) {
        // block b has landed once at most block b+1's GROUPS instructions are still in flight (loads return in issue order)
        if (b + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GROUPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* base = my + (b & 1) * REGION;
        u32x4_v af[2][MT], wf[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fg) ^ sw) << 4;
#pragma unroll
            for (int t = 0; t < MT; ++t) af[ks][t] = *reinterpret_cast<const u32x4_v*>(base + (t * 16 + fr) * 128 + coff);
#pragma unroll
            for (int u = 0; u < NT; ++u) wf[ks][u] = *reinterpret_cast<const u32x4_v*>(base + (MT * 16 + u * 16 + fr) * 128 + coff);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is in registers: its slot may take block b+2
        if (b + 2 < nblk) stage(b & 1, b + 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = mfma16<T>(wf[ks][u], af[ks][t], acc[t][u]);  // swapped: D[n = 4 fg + i][m = fr]
    }
    auto epilogue = [&](int f, f32x4_v v) {
        const int t = f / NT, u = f - t * NT;
        skinny_store<T>(g, slice, (mt0 + t) * 16 + fr, n0 + u * 16 + fg * 4, v, lnst, t * 16 + fr);
    };
    if (NW == 1) {
        if (g.ln_stats) __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) epilogue(t * NT + u, acc[t][u]);
        return;
    }
    __syncthreads();  // every wave is done with its operand image: the partial tiles overlay them
    f32x4_v* red = reinterpret_cast<f32x4_v*>(smem + 1024);  // [NW][MT * NT][64]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) red[(wave * (MT * NT) + t * NT + u) * 64 + lane] = acc[t][u];
    __syncthreads();
    for (int f = wave; f < MT * NT; f += NW) {
        f32x4_v sum = red[f * 64 + lane];
        for (int v = 1; v < NW; ++v) sum += red[(v * (MT * NT) + f) * 64 + lane];  // fixed order: waves 0 .. NW-1
        epilogue(f, sum);
    }
}

// waves for the coalesced kernel (0: the geometry needs the gather kernel)
template <int MT, int NT>
static int lds_waves(int kslice) {
    constexpr int REGION = (MT + NT) * 16 * 128;
    if (kslice % 64) return 0;
    for (int nw : {8, 6, 4, 3, 2, 1})
        if (kslice % (64 * nw) == 0 && (size_t)nw * 2 * REGION <= 149 * 1024 && (kslice / nw >= 128 || nw == 1)) return nw;
    return 0;
}
template <typename T, int MT, int NT>
int launch_lds(const SkinnyArgs& g, int nw, hipStream_t stream) {
    constexpr int REGION = (MT + NT) * 16 * 128;
    auto kern = skinny_lds_kernel<T, MT, NT>;
    const int lds = 1024 + std::max(nw * 2 * REGION, nw > 1 ? nw * MT * NT * 1024 : 0);
    if (lds > 48 * 1024) ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int mtd = (g.M + 15) / 16;
    hipLaunchKernelGGL(kern, dim3(g.N / (16 * NT), g.ksplit, (mtd + MT - 1) / MT), dim3(nw * 64), lds, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename T, int NV>
__global__ __launch_bounds__(64) void skinny_reduce_ln_kernel(const float* __restrict__ part, int S, size_t slice_stride, const float* __restrict__ bias,
                                                              float* __restrict__ x, size_t x_stride, T* __restrict__ h, int h_stride,
                                                              const float* __restrict__ ln_w, const float* __restrict__ ln_b, int D, float eps) {
    const int lane = threadIdx.x, row = blockIdx.x;
    float* xr = x + (size_t)row * x_stride;
    const float* pr = part + (size_t)row * D;
    float v[NV][4];
    // every load of the row is issued before the first add (a runtime slab loop would walk the slabs one L2 round trip at a time:
    // 6.0 us per launch against the 4.5 us of a plain LayerNorm kernel, profiles/r3_latency_trace_vitb32.txt); the adds keep slab order
    constexpr int SU = 4;  // slabs held in registers at once
    float4 pb[NV], ps[SU][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            load4(xr + c, v[i]);
            pb[i] = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < SU; ++s)
                ps[s][i] = s < S ? *reinterpret_cast<const float4*>(pr + (size_t)s * slice_stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            v[i][0] += pb[i].x; v[i][1] += pb[i].y; v[i][2] += pb[i].z; v[i][3] += pb[i].w;
#pragma unroll
            for (int s = 0; s < SU; ++s)
                if (s < S) { v[i][0] += ps[s][i].x; v[i][1] += ps[s][i].y; v[i][2] += ps[s][i].z; v[i][3] += ps[s][i].w; }
            for (int s = SU; s < S; ++s) {  // more than SU slabs (not used by the tower): the rest one by one
                const float4 p = *reinterpret_cast<const float4*>(pr + (size_t)s * slice_stride + c);
                v[i][0] += p.x; v[i][1] += p.y; v[i][2] += p.z; v[i][3] += p.w;
            }
            store4(xr + c, v[i][0], v[i][1], v[i][2], v[i][3]);
        }
    }
    if (ln_w) ln_row_store<T, NV>(v, D, lane, ln_w, ln_b, eps, h + (size_t)row * h_stride);
}

template <typename T, int MT, int NT, int KS>
int launch_one(const SkinnyArgs& g, int nw, hipStream_t stream) {
    auto kern = skinny_gemm_kernel<T, MT, NT, KS>;
    const int lds = 1024 + (nw > 1 ? nw * MT * NT * 1024 : 0);
    if (lds > 48 * 1024) ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int mtd = (g.M + 15) / 16;
    hipLaunchKernelGGL(kern, dim3(g.N / (16 * NT), g.ksplit, (mtd + MT - 1) / MT), dim3(nw * 64), lds, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

template <typename T, int MT, int NT>
int launch_ks(const SkinnyArgs& g, hipStream_t stream) {
    const int kslice = g.K / g.ksplit;
    if (!g.gather) {
        const int nwl = lds_waves<MT, NT>(kslice);
        if (nwl) return launch_lds<T, MT, NT>(g, nwl, stream);
    }
    int nw = 1;
    for (int c = 8; c >= 2; c >>= 1)
        if (kslice % (c * 32) == 0 && kslice / c >= 96) {
            nw = c;
            break;
        }
    const int kw = kslice / nw;
    if (kw % 96 == 0) return launch_one<T, MT, NT, 3>(g, nw, stream);
    if (kw % 64 == 0) return launch_one<T, MT, NT, 2>(g, nw, stream);
    return launch_one<T, MT, NT, 1>(g, nw, stream);
}

template <typename T>
int launch_mt(const SkinnyArgs& g, hipStream_t stream) {
    const int mt = (g.M + 15) / 16;
    if (mt <= 1) return launch_ks<T, 1, 1>(g, stream);
    if (mt <= 2) return launch_ks<T, 2, 1>(g, stream);
    if (mt <= 4) return launch_ks<T, 4, 1>(g, stream);
    // More than 64 rows: m-groups of 64 rows (blockIdx.z), n-groups of 16 NT columns.  With the coalesced kernel a launch costs
    // ~1.7 us + its workgroups' operand bytes at ~40 KB/us as long as every workgroup has a CU of its own (profiles/r3_skinny_variants.txt:
    // 120 KB 3.9 us, 144 KB 4.5, 168 KB 5.5, 192 KB 6.1 at 197 rows; 288-768 workgroups: 7.4-9.5 us), so: the NARROWEST n-group that
    // keeps the grid within 256 workgroups.
    const int mg = (mt + 3) / 4;
    const int kslice = g.K / g.ksplit;
    if (!g.strips && !g.gather && lds_waves<4, 1>(kslice)) {
        for (int nt = 1; nt <= 4; ++nt) {
            if (g.N % (16 * nt)) continue;
            if ((long)(g.N / (16 * nt)) * mg * g.ksplit > 256 && nt < 4) continue;
            if (nt == 1) return launch_ks<T, 4, 1>(g, stream);
            if (nt == 2) return launch_ks<T, 4, 2>(g, stream);
            if (nt == 3) return launch_ks<T, 4, 3>(g, stream);
            return launch_ks<T, 4, 4>(g, stream);
        }
    }
    // the gather kernel (K not a multiple of 64 x waves): wide outputs and 4-way split products on 64 x 64 tiles, the rest on strips
    if ((g.N & 63) == 0 && !g.strips && (g.N >= 1536 || g.ksplit >= 4)) return launch_ks<T, 4, 4>(g, stream);
    return launch_ks<T, 4, 1>(g, stream);
}

}  // namespace

int launch_skinny_gemm(int tcode, const SkinnyArgs& g, hipStream_t stream) {
    if (!skinny_supported(g.M, g.N, g.K, g.lda, g.ldw, g.ksplit)) return fail("skinny gemm: unsupported geometry");
    if (g.ksplit > 1 && (!g.slice_stride || (g.ldo & 3))) return fail("skinny gemm: split-K needs f32 slabs");
    if (g.resid && !g.out_f32) return fail("skinny gemm: the residual epilogue writes f32");
    if ((g.ldo & 3) || (g.resid && (g.ldr & 3))) return fail("skinny gemm: unaligned output");
    if (g.ln_stats && (!g.ln_c || g.ln_parts < 1 || g.ksplit > 1)) return fail("skinny gemm: folded LayerNorm needs ln_c, ln_parts and an unsplit product");
    if (g.stats_out && (!g.xb || (g.ldxb & 3) || !g.out_f32 || g.ksplit > 1 || !g.strips)) return fail("skinny gemm: a statistics producer is an unsplit f32 product on 16-column strips");
    if (tcode == 2) return launch_mt<f16_t>(g, stream);
    if (tcode == 1) return launch_mt<bf16_t>(g, stream);
    return fail("skinny gemm: 16-bit operands only");
}

int launch_skinny_reduce_ln(int tcode, const float* part, int S, size_t slice_stride, const float* bias, float* x, size_t x_stride, void* h,
                            int h_stride, const float* ln_w, const float* ln_b, int rows, int D, float eps, hipStream_t stream) {
    if (D % 4 || D > ROW_MAX_V4 * 256 || rows < 1 || S < 1) return fail("skinny reduce: unsupported geometry");
#define ARP_SR_CALL(NV)                                                                                                                          \
    do {                                                                                                                                         \
        if (tcode == 2)                                                                                                                          \
            hipLaunchKernelGGL((skinny_reduce_ln_kernel<f16_t, NV>), dim3(rows), dim3(64), 0, stream, part, S, slice_stride, bias, x, x_stride,   \
                               static_cast<f16_t*>(h), h_stride, ln_w, ln_b, D, eps);                                                            \
        else                                                                                                                                     \
            hipLaunchKernelGGL((skinny_reduce_ln_kernel<bf16_t, NV>), dim3(rows), dim3(64), 0, stream, part, S, slice_stride, bias, x, x_stride,  \
                               static_cast<bf16_t*>(h), h_stride, ln_w, ln_b, D, eps);                                                           \
    } while (0)
    if (tcode != 1 && tcode != 2) return fail("skinny reduce: 16-bit operands only");
    ARP_NV_DISPATCH(D, ARP_SR_CALL);
#undef ARP_SR_CALL
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace arp

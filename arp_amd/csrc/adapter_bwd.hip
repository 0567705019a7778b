// Fused dY / mask / bias-sum / d-res kernel of the policy step's adapter backward (design notes in adapter_bwd.h).
#include "adapter_bwd.h"

#include "attention.h"  // tr_b64_v

namespace arp {

constexpr int AD_BM = 128, AD_BN = 128, AD_THREADS = 512;
constexpr int AD_STAGE_STRIDE = (AD_BN + 4) * 4;            // f32 staging rows of 132 floats: the float4 fragment writes spread over the banks
constexpr int AD_STAGE_BYTES = AD_BM * AD_STAGE_STRIDE;     // 67 584
constexpr int AD_RED_BYTES = 8 * AD_BN * 4 + 64;            // per-wave column sums + the eight d-res partials
constexpr int AD_LDS_BYTES = AD_STAGE_BYTES + AD_RED_BYTES;  // 71.7 KiB: two workgroups per CU

template <typename T>
__global__ __launch_bounds__(AD_THREADS) void adapter_dy_kernel(AdapterDyArgs g) {
    static_assert(sizeof(T) == 2, "16-bit operand types only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves, 64 rows x 32 columns each
    const int n0 = blockIdx.x * AD_BN, rb = blockIdx.y, r0 = rb * AD_BM;
    const int E = g.E;
    const int nch = E >> 3;  // 16-byte chunks per dz row
    const T* __restrict__ dz = static_cast<const T*>(g.dz);
    const T* __restrict__ Wi = static_cast<const T*>(g.Wi);
    char* a_op = smem;                   // [128 rows][E * 2 B], chunk c of row r at physical chunk c ^ (r & (nch - 1))
    char* b_op = smem + AD_BM * E * 2;   // [E k-rows][256 B], 32-B slot s of row k at physical slot s ^ (k & 7)  (gemm_tn.h's image)

    // ---- operands to LDS by LDS-DMA (one shot: the contraction is only E long) ---------------------------------------------------
    {
        const int a_pieces = AD_BM * E * 2 / 1024;  // 1 KiB per wave-instruction
        for (int p = wave; p < a_pieces; p += 8) {
            const int idx = p * 64 + lane;
            const int row = idx / nch, pc = idx % nch;
            const int lc = pc ^ (row & (nch - 1));
            const int r = min(r0 + row, g.R - 1);  // rows past R: finite stand-ins, never stored or summed
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dz + (size_t)r * E + lc * 8),
                                             (__attribute__((address_space(3))) void*)(a_op + p * 1024), 16, 0, 0);
        }
        const int prow = lane >> 4, pc = lane & 15;
        for (int p = wave; p < E / 4; p += 8) {
            const int row = p * 4 + prow;
            const int lc = ((((pc >> 1) ^ (row & 7)) << 1) | (pc & 1)) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wi + (size_t)row * g.Kin + n0 + lc),
                                             (__attribute__((address_space(3))) void*)(b_op + p * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- dY tile = dz block . Wi tile on the 16x16x32 MFMA ----------------------------------------------------------------------
    // B fragments come through the transposing read (k-slot (fg, j) of step st <-> k = 32 st + 16 (j >> 2) + 4 fg + (j & 3)); the A
    // fragment of a lane is therefore the two 8-byte runs k = 32 st + 4 fg + {0..3} and + 16 of its row.
    const int fr = lane & 15, fg = lane >> 4;
    const int trq = fr >> 2, trp = fr & 3;
    f32x4_v acc[2][4];  // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < E / 32; ++st) {
        u32x4_v af[4], bf[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wr * 64 + mi * 16 + fr;
            const char* rp = a_op + row * E * 2;
            const int c0 = 4 * st + (fg >> 1), sw = row & (nch - 1);
            const u32x2_v lo = *reinterpret_cast<const u32x2_v*>(rp + ((c0 ^ sw) << 4) + (fg & 1) * 8);
            const u32x2_v hi = *reinterpret_cast<const u32x2_v*>(rp + (((c0 + 2) ^ sw) << 4) + (fg & 1) * 8);
            af[mi] = u32x4_v{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int blk16 = wc * 2 + ni;
            const int k0 = 32 * st + 4 * fg + trq, k1 = k0 + 16;
            const tr_b64_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) tr_b64_v*)(b_op + k0 * 256 + ((blk16 ^ (k0 & 7)) << 5) + trp * 8));
            const tr_b64_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) tr_b64_v*)(b_op + k1 * 256 + ((blk16 ^ (k1 & 7)) << 5) + trp * 8));
            const u32x2_v l2 = __builtin_bit_cast(u32x2_v, lo), h2 = __builtin_bit_cast(u32x2_v, hi);
            bf[ni] = u32x4_v{l2[0], l2[1], h2[0], h2[1]};
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16<T>(bf[ni], af[mi], acc[ni][mi]);
    }
    __syncthreads();  // the operand images are dead: the staging tile overlays them

    // The row pass's operands -- A and (with x16) the encodings, 16 bytes each for four rows per thread -- are requested HERE, all eight at once, from clamped
    // rows and unconditionally: inside the pass's `r < R` branch each row's loads were followed by a vmcnt(0) of their own (hipcc at the join of a branch
    // around a load), four dependent memory round trips behind the MFMAs (round 5, found in the ISA).
    const int chunk_ = (int)threadIdx.x & 15, rowl_ = (int)threadIdx.x >> 4;
    u32x4_v araw4[4], xraw4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const size_t flat = (size_t)min(r0 + p * 32 + rowl_, g.R - 1) * g.Kin + n0 + chunk_ * 8;
        araw4[p] = *reinterpret_cast<const u32x4_v*>(static_cast<const T*>(g.A) + flat);
        if (g.x16) xraw4[p] = *reinterpret_cast<const u32x4_v*>(static_cast<const T*>(g.x16) + flat);
    }

    // ---- stage the f32 tile (lane holds row .. + fr, columns .. + 4 fg + {0..3}) ------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int m = wr * 64 + mi * 16 + fr, n = wc * 32 + ni * 16 + fg * 4;
            *reinterpret_cast<f32x4_v*>(smem + m * AD_STAGE_STRIDE + n * 4) = acc[ni][mi];
        }
    __syncthreads();

    // ---- row-contiguous pass: 16-byte loads of A, 32-byte loads of x, 16-byte stores of dApre ----------------------------------------
    T* __restrict__ out = static_cast<T*>(g.dApre);
    const float s = 1.0f / (1.0f + expf(-g.rw[0]));  // res = sigmoid(residual_weight), as adapter_mix computes it
    const int chunk = tid & 15, rowl = tid >> 4;
    float colacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 32 + rowl, r = r0 + row;
        if (r < g.R) {
            const size_t flat = (size_t)r * g.Kin + n0 + chunk * 8;
            const u32x4_v araw = araw4[p];
            float x[8];
            if (g.x16) {  // the operand-type copy of the encodings (16 B instead of 32): only d loss / d res reads x, and the dY beside it is a 16-bit product already
                const u32x4_v xraw = xraw4[p];
                T xt[8];
                memcpy(xt, &xraw, 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = Elem<T>::ld(&xt[e]);
            } else {
                const float4 x0 = *reinterpret_cast<const float4*>(g.x32 + flat), x1 = *reinterpret_cast<const float4*>(g.x32 + flat + 4);
                x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
            }
            const float4 v0 = *reinterpret_cast<const float4*>(smem + row * AD_STAGE_STRIDE + chunk * 32);
            const float4 v1 = *reinterpret_cast<const float4*>(smem + row * AD_STAGE_STRIDE + chunk * 32 + 16);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            T a[8], o[8];
            memcpy(a, &araw, 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float af = Elem<T>::ld(&a[e]);
                dsum += v[e] * (af - x[e]);
                Elem<T>::st(&o[e], af > 0.f ? v[e] * s : 0.f);
                colacc[e] += Elem<T>::ld(&o[e]);  // sums of the ROUNDED values, as the unfused masked copy's were
            }
            u32x4_v oraw;
            memcpy(&oraw, o, 16);
            *reinterpret_cast<u32x4_v*>(out + flat) = oraw;
        }
    }
    // column sums: the four row groups of a wave by shuffles, the eight waves through LDS in a fixed order
    float* red = reinterpret_cast<float*>(smem + AD_STAGE_BYTES);
    float* dred = red + 8 * AD_BN;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        colacc[e] += __shfl_xor(colacc[e], 16, 64);
        colacc[e] += __shfl_xor(colacc[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * AD_BN + chunk * 8 + e] = colacc[e];
    }
    dsum = wave_sum(dsum);
    if (lane == 0) dred[wave] = dsum;
    __syncthreads();
    if (tid < AD_BN) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w * AD_BN + tid];
        const int token = n0 / g.D, dcol = n0 % g.D, tokens = g.Kin / g.D;
        g.colpart[((size_t)rb * tokens + token) * g.D + dcol + tid] = t;
    }
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += dred[w];
        g.dres_part[rb * gridDim.x + blockIdx.x] = t;
    }
}

bool adapter_dy_supported(int E, int D, long long Kin) {
    return E >= 32 && E <= 128 && E % 32 == 0 && (E & (E - 1)) == 0 && D > 0 && D % AD_BN == 0 && Kin % D == 0;
}

template <typename T> static int launch_impl(const AdapterDyArgs& g, hipStream_t stream) {
    if (g.R <= 0 || !adapter_dy_supported(g.E, g.D, g.Kin)) return fail("adapter_dy: unsupported geometry");
    auto kern = adapter_dy_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        ARP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AD_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.Kin / AD_BN, adapter_dy_row_blocks(g.R)), dim3(AD_THREADS), AD_LDS_BYTES, stream, g);
    ARP_HIP_OK(hipGetLastError());
    return 0;
}

int launch_adapter_dy(int tcode, const AdapterDyArgs& g, hipStream_t stream) {
    if (tcode == 1) return launch_impl<bf16_t>(g, stream);
    if (tcode == 2) return launch_impl<f16_t>(g, stream);
    return fail("adapter_dy: 16-bit operand types only");
}

}  // namespace arp

// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of libarp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstring>
#include <string>

namespace arp {

typedef uint16_t bf16_t;  // raw bf16 bits; arithmetic always happens in f32
// raw IEEE binary16 bits as a distinct type (ARP_MODE_F16: same MFMA rate as bf16, 11 significand bits instead of 8)
struct f16_t { uint16_t b; };
// output tag of row kernels in ARP_MODE_F16X3: a row of D values is stored as the binary16 triple [hi | lo | hi] (3 D wide; store_split3 below)
struct f16x3_t { uint16_t b; };
// raw OCP e4m3fn bits (no infinities, max 448, NaN = 0x7f / 0xff): operand type of the scaled fp8 MFMA (BASELINE configs[4])
struct fp8_t { uint8_t b; };
typedef __attribute__((ext_vector_type(8))) int i32x8_v;
typedef __attribute__((ext_vector_type(4))) int i32x4_v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_v;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_v;
typedef __attribute__((ext_vector_type(4))) float f32x4_v;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_v;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_v;

// Thread-local error string behind arp_last_error().
void set_error(const std::string& msg);
int fail(const std::string& msg);  // sets the error, returns -1

#define ARP_HIP_OK(expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::arp::fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                               ":" + std::to_string(__LINE__) + ")");                           \
    } while (0)

#define ARP_TRY(expr)            \
    do {                         \
        int _r = (expr);         \
        if (_r != 0) return _r;  \
    } while (0)

// ---- device-side scalar helpers -------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// f32 -> binary16, round-to-nearest-even (v_cvt_f16_f32); overflow goes to inf as IEEE says
__device__ __forceinline__ float h2f(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.b); }
__device__ __forceinline__ f16_t f2h(float f) { return f16_t{__builtin_bit_cast(uint16_t, (_Float16)f)}; }
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const f16x2_v v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// two f32 -> one 32-bit word of the 16-bit operand type T
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) return pack_h2(lo, hi);
    else return pack_bf2(lo, hi);
}
// one 16x16x32 MFMA on 16-bit operands of type T (8 elements per lane in a 128-bit register group)
template <typename T> __device__ __forceinline__ f32x4_v mfma16(u32x4_v a, u32x4_v b, f32x4_v c) {
    if constexpr (__is_same(T, f16_t))
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
}

// ---- fp8 (e4m3fn) helpers: conversions saturate to +-448 (the format has no infinity; an overflow would read back as NaN) -------
constexpr float FP8_MAX = 448.f;
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -FP8_MAX, FP8_MAX); b = __builtin_amdgcn_fmed3f(b, -FP8_MAX, FP8_MAX);
    c = __builtin_amdgcn_fmed3f(c, -FP8_MAX, FP8_MAX); d = __builtin_amdgcn_fmed3f(d, -FP8_MAX, FP8_MAX);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
// one v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands with unit block scales: 32 bytes of K per lane and operand, twice the
// FLOPs per cycle of the 16-bit 16x16x32 form.  The two 16-byte halves may hold ANY 32 of the 128 k-values as long as both
// operands use the same assignment (the product is a sum over k).
__device__ __forceinline__ f32x4_v mfma_fp8(u32x4_v a0, u32x4_v a1, u32x4_v b0, u32x4_v b1, f32x4_v c) {
    const i32x8_v a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const i32x8_v b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  // cbsz = blgp = 0: e4m3; scale 2^0
}

// one v_mfma_scale_f32_16x16x128_f8f6f4 on e2m1 (fp4) operands: 16 bytes = 32 k-values per lane and operand (a 4-register tuple, like the binary16 MFMA's),
// FOUR times the k per cycle of the 16-bit 16x16x32 form.  scale: byte 0 = the e8m0 block scale of operand A (127 - s multiplies the product by 2^-s),
// byte 1 = operand B's (127: 2^0) -- one register, picked apart by op_sel.
__device__ __forceinline__ f32x4_v mfma_fp4_scaled(u32x4_v a, u32x4_v b, f32x4_v c, int scale) {
    const i32x4_v a4 = __builtin_bit_cast(i32x4_v, a), b4 = __builtin_bit_cast(i32x4_v, b);
    // the fp4 form reads 4 registers per operand: the upper half of the builtin's 8-wide type is left undefined so that hipcc allocates 4-tuples
    const i32x8_v a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1), b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, scale, 1, scale);
}

// The same under a lane mask (wave-uniform, SGPR pair): an LDS-DMA that has to be ISSUED -- the counted vmcnt waits of a steady-state loop count it --
// but whose bytes nobody wants is issued for lane 0 only (mask = 1): 16 bytes through the address path instead of 1 KiB.
__device__ __forceinline__ void dma16_saddr_masked(const void* sbase, uint32_t voff, uint32_t lds_dst, unsigned long long mask) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(mask) : "memory");
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return h2f(*p); }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = f2h(v); }
};

template <> struct Elem<fp8_t> {
    static __device__ __forceinline__ float ld(const fp8_t* p) { return __builtin_amdgcn_cvt_f32_fp8((int)p->b, 0); }
    static __device__ __forceinline__ void st(fp8_t* p, float v) { p->b = (uint8_t)(pack_fp8x4(v, 0.f, 0.f, 0.f) & 0xffu); }
};
__device__ __forceinline__ void store4(fp8_t* p, float a, float b, float c, float d) { *reinterpret_cast<uint32_t*>(p) = pack_fp8x4(a, b, c, d); }
__device__ __forceinline__ void load4(const fp8_t* p, float (&v)[4]) {
    const int w = *reinterpret_cast<const int*>(p);
    v[0] = __builtin_amdgcn_cvt_f32_fp8(w, 0); v[1] = __builtin_amdgcn_cvt_f32_fp8(w, 1);
    v[2] = __builtin_amdgcn_cvt_f32_fp8(w, 2); v[3] = __builtin_amdgcn_cvt_f32_fp8(w, 3);
}

// store 4 consecutive values (16-B aligned for float, 8-B aligned for bf16)
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(a, b), pack_bf2(c, d));
}
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_h2(a, b), pack_h2(c, d));
}
// (hi, lo) binary16 pair of four f32 values in the K-concatenated operand layout of ARP_MODE_F16X3: hi at p, lo at p + n, hi again at p + 2 n
// ([x_hi | x_lo | x_hi] against [W_hi | W_hi | W_lo]: x.W on three 16-bit MFMAs to ~2^-22).  8-byte aligned like store4.
// A value about to be split into (hi, lo) must be ONE f32 number.  With fp-contract on, hipcc may evaluate `x = m * n` twice: rounded to f32 in front of
// one use and fused into another (v_fma_mixlo_f16 rounds m * n straight to binary16; `x - hi` becomes fma(m, n, -hi)) -- the two "hi" then differ by one
// binary16 ulp once in ~2^13 values and that element is off by 2^-11 instead of 2^-22 (found on the x3 attention's prescaled Q: 2 query rows of 514).
__device__ __forceinline__ float pin_f32(float x) {
    asm("" : "+v"(x));
    return x;
}
// (hi, lo) binary16 halves of x0 * s and x1 * s, packed pairwise: hi = rn16(x * s) and lo = rn16(x * s - hi), each ONE v_fma_mix instruction that forms the
// product (and the difference) exactly and rounds once -- four instructions for two values where convert / convert back / subtract / convert take five to six
// plus the scaling, and there is a single "hi" by construction (pin_f32's hazard cannot arise).  s may be any f32 (1.0f for a plain split).
__device__ __forceinline__ void split2_f16(float x0, float x1, float s, uint32_t& hi, uint32_t& lo) {
#ifdef ARP_NO_MIX_SPLIT  // A/B builds: the compiler's own conversion sequence on the pinned products
    const float a = pin_f32(x0 * s), b = pin_f32(x1 * s);
    const float ha = h2f(f2h(a)), hb = h2f(f2h(b));
    hi = pack_h2(ha, hb);
    lo = pack_h2(a - ha, b - hb);
    return;
#endif
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(s), "v"(hi));
}
__device__ __forceinline__ void store_split3(f16_t* p, size_t n, float a, float b, float c, float d) {
    uint2 hi, lo;
    split2_f16(a, b, 1.0f, hi.x, lo.x);
    split2_f16(c, d, 1.0f, hi.y, lo.y);
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + n) = lo;
    *reinterpret_cast<uint2*>(p + 2 * n) = hi;
}
template <typename T> __device__ __forceinline__ void store_split3(T* p, size_t n, float a, float b, float c, float d) { store4(p, a, b, c, d); }  // f16 operands only

// ---- ARP_MODE_F16C (row N1): binary16 GEMMs whose operand roundings are corrected on the scaled fp4 MFMA ---------------------------------------
// An operand row of K values is stored as [hi: binary16 x K | x4: e2m1 x K | dx4: e2m1 x K] (3 K bytes; two values per byte, value 2j in the low nibble):
// hi = rn16(x), x4 = fp4(hi * 2^F16C_X_SHIFT), dx4 = fp4((x - hi) * 2^F16C_DX_SHIFT), both saturating at +-6 (a clipped outlier only loses part of ITS
// correction term).  The consumer GEMM (gemm256 MIXC) computes hi.W_hi + 2^-s x4.dW4 (+ 2^-s' dx4.W4): the correction terms are 2^-12 of the product, so
// the 1-2 significant bits of e2m1 take the binary16 operand roundings out to ~1/4 of their size -- on an MFMA (v_mfma_scale_f32_16x16x128_f8f6f4 with
// fp4 operands) that moves FOUR times the k per cycle of the binary16 one and takes the same 4-register operand tuples: a K-tile of 128 bytes per row is
// 256 k-values, so both corrections cost K/256 + K/256 extra K-tiles on top of K/64: 1.5x the binary16 product where (hi, lo) binary16 pairs cost 3x.
// (An e4m3 variant of the same loop -- 2x the k per cycle, 8-register operand tuples -- was built first: hipcc could not hold its 8-tuples beside 128
// accumulators without scratch traffic inside the K loop, and a scratch reload is a vmcnt(0), i.e. a drained LDS-DMA ring: 47 ms per step instead of 10.)
struct f16c_t { uint16_t b; };   // output tag of row kernels: [hi | x4]           (the dx4 segment of the row is left unwritten)
struct f16c2_t { uint16_t b; };  // [hi | x4 | dx4]
constexpr int F16C_X_SHIFT = 1, F16C_DX_SHIFT = 13;  // (numpy sweep of the residual error: x 2^1 / dx 2^13 and weight scales one binade into saturation are the flat optimum)
// eight values -> eight e2m1 nibbles (value 0 in the low nibble of byte 0), round-to-nearest-even, saturating at +-6
__device__ __forceinline__ uint32_t pack_fp4x8(const float (&v)[8]) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[0], v[1], 1.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[2], v[3], 1.0f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[4], v[5], 1.0f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[6], v[7], 1.0f, 3);
    return w;
}
__device__ __forceinline__ uint16_t pack_fp4x4(float a, float b, float c, float d) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, a, b, 1.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, c, d, 1.0f, 1);
    return (uint16_t)w;
}
// four consecutive values at column c (a multiple of 4) of a row whose binary16 segment starts at `row` (K = row width)
template <bool WITH_DX> __device__ __forceinline__ void store_f16c(f16_t* row, int c, int K, float a, float b, float cc, float d) {
    const float v[4] = {pin_f32(a), pin_f32(b), pin_f32(cc), pin_f32(d)};
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = h2f(f2h(v[j]));
    *reinterpret_cast<uint2*>(row + c) = make_uint2(pack_h2(h[0], h[1]), pack_h2(h[2], h[3]));
    uint8_t* seg = reinterpret_cast<uint8_t*>(row + K);
    constexpr float sx = (float)(1 << F16C_X_SHIFT), sd = (float)(1 << F16C_DX_SHIFT);
    *reinterpret_cast<uint16_t*>(seg + (c >> 1)) = pack_fp4x4(h[0] * sx, h[1] * sx, h[2] * sx, h[3] * sx);
    if constexpr (WITH_DX) *reinterpret_cast<uint16_t*>(seg + (K >> 1) + (c >> 1)) = pack_fp4x4((v[0] - h[0]) * sd, (v[1] - h[1]) * sd, (v[2] - h[2]) * sd, (v[3] - h[3]) * sd);
}
// sixteen consecutive values at column c (a multiple of 16): the same three segments as whole 32- / 8- / 8-byte pieces (two 16-byte stores and two 8-byte
// stores per lane where four calls of the form above issue twelve stores of 8, 2 and 2 bytes)
__device__ __forceinline__ void store_f16c16(f16_t* row, int c, int K, const float (&a)[16], bool with_dx = true) {
    float v[16], h[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        v[j] = pin_f32(a[j]);
        h[j] = h2f(f2h(v[j]));
    }
    uint4* hi = reinterpret_cast<uint4*>(row + c);
    hi[0] = make_uint4(pack_h2(h[0], h[1]), pack_h2(h[2], h[3]), pack_h2(h[4], h[5]), pack_h2(h[6], h[7]));
    hi[1] = make_uint4(pack_h2(h[8], h[9]), pack_h2(h[10], h[11]), pack_h2(h[12], h[13]), pack_h2(h[14], h[15]));
    uint8_t* seg = reinterpret_cast<uint8_t*>(row + K);
    constexpr float sx = (float)(1 << F16C_X_SHIFT), sd = (float)(1 << F16C_DX_SHIFT);
    float x0[8], x1[8], d0[8], d1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x0[j] = h[j] * sx; x1[j] = h[8 + j] * sx;
        d0[j] = (v[j] - h[j]) * sd; d1[j] = (v[8 + j] - h[8 + j]) * sd;
    }
    *reinterpret_cast<uint2*>(seg + (c >> 1)) = make_uint2(pack_fp4x8(x0), pack_fp4x8(x1));
    if (with_dx) *reinterpret_cast<uint2*>(seg + (K >> 1) + (c >> 1)) = make_uint2(pack_fp4x8(d0), pack_fp4x8(d1));  // (uniform: the consumer's plan)
}

__device__ __forceinline__ void load4(const f16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const f16x2_v a = __builtin_bit_cast(f16x2_v, t.x), b = __builtin_bit_cast(f16x2_v, t.y);
    v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)b[0]; v[3] = (float)b[1];
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// 64-lane wavefront reductions (all lanes receive the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

enum Act { ACT_NONE = 0, ACT_QGELU = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_GELU_TANH = 4 };

// FAST selects hardware-approximate exp/rcp (bf16 throughput mode); otherwise libm-accurate f32.
template <int ACT, bool FAST = false> __device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == ACT_QGELU) {
        // QuickGELU x*sigmoid(1.702x)  (reference: arp_dt/models/openai/layers.py:12-13)
        // one multiply by the folded constant -1.702 * log2(e), v_exp_f32, v_rcp_f32: the 16-bit GEMM epilogues are VALU-bound on
        // this (34 issue cycles per element-row with the two separate multiplies of __expf(-1.702f * x); round 2 measurements)
        if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669595930157f));
        return x / (1.0f + expf(-1.702f * x));
    } else if constexpr (ACT == ACT_RELU) {
        return fmaxf(x, 0.0f);
    } else if constexpr (ACT == ACT_TANH) {
        return tanhf(x);
    } else if constexpr (ACT == ACT_GELU_TANH) {
        // flax nn.gelu default = tanh approximation (reference: arp_dt/layers.py:31)
        const float c = 0.7978845608028654f;
        const float u = c * (x + 0.044715f * x * x * x);
        // 0.5*x*(1+tanh(u)) == x * sigmoid(2u): one hardware exp + one rcp in the fast (bf16) mode
        if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -2.8853900817779268f));
        return 0.5f * x * (1.0f + tanhf(u));
    } else {
        return x;
    }
}

// Four fragment values at once.  In the 16-bit GEMM epilogues the activation is VALU-issue-bound (two transcendental and five
// plain instructions per element); written on float2 pairs the scale, the +1 and the final product become v_pk_mul_f32 /
// v_pk_add_f32 -- two elements per issue slot -- and the arithmetic per element is exactly apply_act<ACT, true>'s.
typedef __attribute__((ext_vector_type(2))) float f32x2_v;
template <int ACT, bool FAST> __device__ __forceinline__ void apply_act4(float (&v)[4]) {
    if constexpr (FAST && (ACT == ACT_QGELU || ACT == ACT_GELU_TANH)) {
        f32x2_v a = {v[0], v[1]}, b = {v[2], v[3]};
        f32x2_v ta, tb;
        if constexpr (ACT == ACT_QGELU) {
            ta = a * -2.4554669595930157f;
            tb = b * -2.4554669595930157f;
        } else {
            const float c = 0.7978845608028654f;
            ta = (c * (a + 0.044715f * a * a * a)) * -2.8853900817779268f;
            tb = (c * (b + 0.044715f * b * b * b)) * -2.8853900817779268f;
        }
        f32x2_v ea = {__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1])};
        f32x2_v eb = {__builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
        ea = ea + 1.0f;
        eb = eb + 1.0f;
        const f32x2_v ra = {__builtin_amdgcn_rcpf(ea[0]), __builtin_amdgcn_rcpf(ea[1])};
        const f32x2_v rb = {__builtin_amdgcn_rcpf(eb[0]), __builtin_amdgcn_rcpf(eb[1])};
        a = a * ra;
        b = b * rb;
        v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, FAST>(v[j]);
    }
}

// host-side f32 -> bf16 (RNE)
static inline bf16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// host-side f32 -> binary16 (RNE, overflow to inf, subnormals kept)
static inline f16_t host_f2h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return f16_t{(uint16_t)(sign | 0x7e00u)};        // NaN
    if (a >= 0x47800000u) return f16_t{(uint16_t)(sign | 0x7c00u)};       // >= 65536 (and inf) -> inf; 65520..65536 handled by rounding below
    if (a < 0x33000000u) return f16_t{sign};                               // < 2^-25 -> 0
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                              // 24-bit significand
    int shift = e >= -14 ? 13 : (13 + (-14 - e));                          // bits dropped
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    uint32_t h = e >= -14 ? (((uint32_t)(e + 15) << 10) + (q - 0x400u)) : q;  // carries propagate into the exponent
    return f16_t{(uint16_t)(sign | h)};
}

// host-side f32 -> e2m1 nibble (OCP fp4: 0, 0.5, 1, 1.5, 2, 3, 4, 6 and their negatives; round-to-nearest-even on the 1-bit significand, saturating at +-6)
static inline uint8_t host_f2fp4(float f) {
    const uint8_t sign = std::signbit(f) ? 8 : 0;
    const float a = std::fabs(f);
    uint8_t m;
    if (!(a == a)) m = 7;             // NaN -> saturate (never produced by the callers)
    else if (a <= 0.25f) m = 0;       // tie 0.25 -> 0 (even)
    else if (a < 0.75f) m = 1;        // 0.5
    else if (a <= 1.25f) m = 2;       // ties 0.75 -> 1.0, 1.25 -> 1.0
    else if (a < 1.75f) m = 3;        // 1.5
    else if (a <= 2.5f) m = 4;        // ties 1.75 -> 2.0, 2.5 -> 2.0
    else if (a < 3.5f) m = 5;         // 3.0
    else if (a <= 5.0f) m = 6;        // ties 3.5 -> 4.0, 5.0 -> 4.0
    else m = 7;                       // 6.0 (saturating)
    return sign | m;
}
// host-side f32 -> e4m3fn (RNE, saturating at +-448, subnormals down to 2^-9)
static inline fp8_t host_f2fp8(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    float a = fabsf(f);
    if (!(a == a)) return fp8_t{(uint8_t)(sign | 0x7f)};  // NaN
    if (a >= 448.f) return fp8_t{(uint8_t)(sign | 0x7e)};   // saturate: 0x7e = 448
    if (a < 0.0009765625f) return fp8_t{sign};              // < 2^-10: rounds to zero (half the smallest subnormal 2^-9)
    int e;
    const float m = frexpf(a, &e);                          // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                          // a = (2m) * 2^E, 2m in [1, 2)
    if (E < -6) {                                           // subnormal: units of 2^-9
        const float q = nearbyintf(a * 512.f);              // RNE (default rounding mode)
        const int qi = (int)q;
        return fp8_t{(uint8_t)(sign | (qi >= 8 ? 0x08 : qi))};  // 8 units = the smallest normal
    }
    float q = nearbyintf((2.f * m - 1.f) * 8.f);            // 3 mantissa bits
    if (q >= 8.f) { q = 0.f; E += 1; }
    if (E > 8 || (E == 8 && q > 6.f)) return fp8_t{(uint8_t)(sign | 0x7e)};
    return fp8_t{(uint8_t)(sign | ((E + 7) << 3) | (int)q)};
}

// K-loop version (round 5).  KV = 1: (i) every LDS-DMA is ONE inline-asm statement in the SADDR form --
// `s_mov_b32 m0, dst; s_nop 0; global_load_lds_dwordx4 v_off32, s[base:base+1]` -- with the tile's row base + K-tile offset in an SGPR
// pair bumped by SALU and a 32-bit per-lane offset that never changes (hipcc selects the VGPR-pair form for the builtin: one 64-bit
// `v_lshl_add_u64` per DMA on the SIMD's vector issue port, the resource section 5d prices the loop by; 16 address VGPRs -> 8);
// (ii) the steady state (K-tiles whose issues all exist) is peeled from the tails, so the loop body carries no `gi < G` branches and
// its counted waits are the literal vmcnt(8).  Same MFMA order, same LDS image: results bit-identical to KV = 0.
#ifndef ARP_G2_KV
#define ARP_G2_KV 1
#endif
// One LDS-DMA of 16 bytes per lane: global address = sbase (wave-uniform, SGPR pair) + voff (per lane, 32-bit), LDS address = lds_dst
// (wave-uniform byte address) + lane * 16.  M0 is written in the statement that reads it (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma16_saddr(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

}  // namespace arp

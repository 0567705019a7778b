// Reproduction of the hipcc 7.0 (-O3, gfx950) defect behind DESIGN.md section 0 (round 6): __builtin_bit_cast(f16x2_v, v[i]) on an ELEMENT of an ext-vector reads element 0.
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only scripts/bitcast_elem_repro.hip -o - | grep -E 'global_load|v_cvt_scalef32'
// ka (the round-6 epilogue form) and kb (the round-5 form) load ONE dword and convert it four times; kc (separate scalars) and kd (element copied to a scalar first: the fix) are right.
#include <hip/hip_runtime.h>
typedef _Float16 f16x2_v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_v __attribute__((ext_vector_type(4)));
// (a) round-6 form
__global__ void ka(const u32x4_v* in, unsigned* out, float inv) {
    const u32x4_v v = in[threadIdx.x];
    unsigned w4 = 0;
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, v[0]), inv, 0);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, v[1]), inv, 1);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, v[2]), inv, 2);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, v[3]), inv, 3);
    out[threadIdx.x] = w4;
}
// (b) round-5 form
__global__ void kb(const u32x4_v* in, unsigned* out, float sc) {
    const u32x4_v v = in[threadIdx.x];
    float f[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f16x2_v p2 = __builtin_bit_cast(f16x2_v, v[h]);
        f[2 * h] = (float)p2[0] * sc;
        f[2 * h + 1] = (float)p2[1] * sc;
    }
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, f[0], f[1], 1.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, f[2], f[3], 1.0f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, f[4], f[5], 1.0f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, f[6], f[7], 1.0f, 3);
    out[threadIdx.x] = w;
}
// (c) f16 builtin with sources in separate scalars (no vector element bitcast)
__global__ void kc(const unsigned* in, unsigned* out, float inv) {
    const unsigned a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128], d = in[threadIdx.x + 192];
    unsigned w4 = 0;
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, a), inv, 0);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, b), inv, 1);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, c), inv, 2);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, d), inv, 3);
    out[threadIdx.x] = w4;
}
// (d) element copied to a scalar first
__global__ void kd(const u32x4_v* in, unsigned* out, float inv) {
    const u32x4_v v = in[threadIdx.x];
    const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
    unsigned w4 = 0;
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e0), inv, 0);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e1), inv, 1);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e2), inv, 2);
    w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w4, __builtin_bit_cast(f16x2_v, e3), inv, 3);
    out[threadIdx.x] = w4;
}

// v_cvt_scalef32_pk_f32_fp4: which way does the scale go on the way UP?  hipcc --offload-arch=gfx950 -O2 scripts/fp4_cvt_probe2.hip -o scripts/fp4_cvt_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(float* out) {
    const unsigned w = 0x7654321fu;  // nibbles (low first): f,1,2,3,4,5,6,7 -> -6, .5, 1, 1.5, 2, 3, 4, 6
    const float scales[3] = {1.0f, 0.5f, 2.0f};
    for (int s = 0; s < 3; ++s) {
        f2 v;
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scales[s], 0); out[s * 8 + 0] = v[0]; out[s * 8 + 1] = v[1];
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scales[s], 1); out[s * 8 + 2] = v[0]; out[s * 8 + 3] = v[1];
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scales[s], 2); out[s * 8 + 4] = v[0]; out[s * 8 + 5] = v[1];
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scales[s], 3); out[s * 8 + 6] = v[0]; out[s * 8 + 7] = v[1];
    }
}
int main() {
    float* d; hipMalloc(&d, 24 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d);
    float h[24]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; ++s) { printf("scale %s:", s == 0 ? "1" : s == 1 ? "0.5" : "2"); for (int i = 0; i < 8; ++i) printf(" %g", h[s * 8 + i]); printf("\n"); }
    return 0;
}

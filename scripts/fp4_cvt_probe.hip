// What does the SCALE operand of v_cvt_scalef32_pk_fp4_{f32,f16} do?  (round 6: folding the 2^1 / 2^13 pre-multiplies of common.h::pack_fp4x8 into the
// instruction.)  hipcc --offload-arch=gfx950 -O2 scripts/fp4_cvt_probe.hip -o scripts/fp4_cvt_probe.bin && scripts/fp4_cvt_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* v, int n, unsigned* out) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = v[i], b = -v[i];
    const float scales[4] = {1.0f, 0.5f, 2.0f, 0.25f};
    for (int s = 0; s < 4; ++s) {
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, a, b, scales[s], 0);
        out[(i * 4 + s) * 2] = w & 0xff;
        unsigned w2 = 0;
        h2 hv = {(_Float16)a, (_Float16)b};
        w2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w2, hv, scales[s], 0);
        out[(i * 4 + s) * 2 + 1] = w2 & 0xff;
    }
}
int main() {
    const float hv[] = {0.f, 0.2f, 0.25f, 0.3f, 0.5f, 0.74f, 0.75f, 0.76f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 12.f, 100.f};
    const int n = sizeof(hv) / 4;
    float* dv; unsigned* dout;
    hipMalloc(&dv, sizeof(hv)); hipMalloc(&dout, n * 8 * 4);
    hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, n, dout);
    unsigned ho[22 * 8];
    hipMemcpy(ho, dout, n * 8 * 4, hipMemcpyDeviceToHost);
    const float grid[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    printf("value | f32 src: scale 1, 0.5, 2, 0.25 (decoded +v) | f16 src: same\n");
    for (int i = 0; i < n; ++i) {
        printf("%7.3f |", hv[i]);
        for (int src = 0; src < 2; ++src) {
            for (int s = 0; s < 4; ++s) {
                const unsigned byte = ho[(i * 4 + s) * 2 + src];
                printf(" %4.1f/%4.1f", grid[byte & 7] * ((byte & 8) ? -1 : 1), grid[(byte >> 4) & 7] * ((byte & 0x80) ? -1 : 1));
            }
            printf(" |");
        }
        printf("\n");
    }
    return 0;
}

// Standalone timing of layernorm_kernel (rowops.h) at the labelling shapes: f32 [rows, 768] in, f16 out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iarp_amd/csrc scripts/ln_bench.hip -o scripts/ln_bench.bin
#include <cstdio>
#include <vector>

#include "rowops.h"

namespace arp {
int fail(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); return -1; }
void set_error(const std::string&) {}
}  // namespace arp
using namespace arp;

static void run(int rows) {
    const int D = 768;
    float *x, *w, *b;
    f16_t* y;
    hipMalloc(&x, (size_t)rows * D * 4); hipMalloc(&y, (size_t)rows * D * 2); hipMalloc(&w, D * 4); hipMalloc(&b, D * 4);
    hipMemset(x, 0x3c, (size_t)rows * D * 4); hipMemset(w, 0, D * 4); hipMemset(b, 0, D * 4);
    auto go = [&]() { hipLaunchKernelGGL((layernorm_kernel<f16_t, 3>), dim3((rows + 3) / 4), dim3(256), 0, nullptr, x, (size_t)D, y, D, w, b, rows, D, 1e-5f); };
    for (int i = 0; i < 3; ++i) go();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 50;
    printf("layernorm rows=%d: %.1f us  %.2f TB/s (f32 read + f16 write)\n", rows, ms * 1e3, (double)rows * D * 6 / ms / 1e9);
    hipFree(x); hipFree(y); hipFree(w); hipFree(b);
}

int main() {
    run(25600);
    run(51200);
    run(204800);
    return 0;
}

// Store-path micro-benchmark (round 2): how fast can ONE workgroup per CU push a 256x256 output tile to memory,
// alone on the chip and with every CU doing it at once?  Decides whether the GEMM epilogue is bound by a per-CU
// store sink (then only overlap inside the CU helps) or by chip-wide write bandwidth (then de-phasing helps).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/store_bench.hip -o scripts/store_bench.bin && scripts/store_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u4;

// each workgroup (512 threads) writes `tiles` tiles of 256 rows x row_bytes (512 = f16 tile, 1024 = f32 tile);
// mode 0: registers -> global, 16-B per lane, whole rows contiguous; mode 1: the same through an LDS read first;
// mode 2: read-modify-write (residual add) of f32 rows
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(u4* out, int row_bytes, int ld_bytes, int tiles, long long* cycles) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    const int lanes_per_row = row_bytes / 16;
    const int rows_per_pass = 512 / lanes_per_row;
    const int passes = 256 / rows_per_pass;
    u4 v = {(unsigned)tid, 1u, 2u, 3u};
    if (MODE == 1) {
        for (int i = tid; i < 32768 / 16; i += 512) reinterpret_cast<u4*>(smem)[i] = v;
        __syncthreads();
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        char* base = reinterpret_cast<char*>(out) + ((size_t)blockIdx.x * tiles + t) * 256 * (size_t)ld_bytes;
        for (int p = 0; p < passes; ++p) {
            const int r = p * rows_per_pass + tid / lanes_per_row;
            u4* dst = reinterpret_cast<u4*>(base + (size_t)r * ld_bytes + (tid % lanes_per_row) * 16);
            if (MODE == 1) v = reinterpret_cast<u4*>(smem)[(tid * 7 + p) & 2047];
            if (MODE == 2) {
                u4 o = *dst;
                v = o + v;
            }
            *dst = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t cap = (size_t)14 << 30;
    u4* out;
    long long* cyc;
    hipMalloc(&out, cap);
    hipMemset(out, 0, cap);
    hipMalloc(&cyc, 4096 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    const int grids[] = {1, 32, 256};
    // ld_mul = 1: the tile's rows are contiguous (a 128-KiB block); 12: rows 6 KiB apart, as the 256 x 256 tile of a [M, 3072] f16 output
    for (int ld_mul : {1, 12})
    for (int mode = 0; mode < 3; ++mode)
        for (int rb : {512, 1024}) {
            if (mode == 2 && rb == 512) continue;
            for (int grid : grids) {
                const int tiles = 8;
                const size_t need = (size_t)grid * tiles * 256 * rb * ld_mul;
                if (need > cap) continue;
                float best = 1e9f;
                std::vector<long long> h(grid);
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 140 * 1024, 0, out, rb, rb * ld_mul, tiles, cyc);
                    if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 140 * 1024, 0, out, rb, rb * ld_mul, tiles, cyc);
                    if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(512), 140 * 1024, 0, out, rb, rb * ld_mul, tiles, cyc);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
                double avg = 0;
                for (auto c : h) avg += (double)c;
                avg /= grid;
                const double bytes_wg = (double)tiles * 256 * rb * (mode == 2 ? 2 : 1);
                printf("ld x%2d mode %d row_bytes %4d grid %3d: %8.1f us  %7.2f TB/s chip  %6.1f B/clk per WG (in-kernel %0.0f cyc per tile)\n", ld_mul, mode, rb, grid, best * 1e3,
                       bytes_wg * grid / (best * 1e-3) / 1e12, bytes_wg / avg, avg / tiles);
            }
        }
    return 0;
}

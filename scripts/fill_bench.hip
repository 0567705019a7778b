// How fast does ONE CU fill from L2, by access pattern?  Every workgroup (8 waves, one per CU) reads the same L2-resident buffer of
// `kb` KB, all loads issued before the first use, as
//   pattern 0: the skinny GEMM's fragment gather -- 16 B per lane, 16 rows x 64 B per wave instruction, rows `ld` bytes apart;
//   pattern 1: coalesced -- 16 B per lane, 1 KiB contiguous per wave instruction (8 whole 128-B lines);
//   pattern 2: coalesced LDS-DMA (global_load_lds_dwordx4), same addresses as pattern 1.
// Reported: microseconds per launch (graph replay, dependent launches) and bytes per clock per CU above the 3.0 us launch floor.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 scripts/fill_bench.hip -o scripts/fill_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PATTERN, int NLOAD>  // NLOAD wave-instructions of 1 KiB per wave
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ buf, int ld, unsigned int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int acc = 0;
    if constexpr (PATTERN == 0) {
        // wave w owns a 192-byte K range of every row (as skinny: K split over the waves); lane (fr, fg) reads row fr of a 16-row tile
        const int fr = lane & 15, fg = lane >> 4;
        u32x4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int tile = i / 3, j = i % 3;  // 3 k-steps per m-tile
            v[i] = *reinterpret_cast<const u32x4*>(buf + (size_t)(tile * 16 + fr) * ld + wave * 192 + j * 64 + fg * 16);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    } else if constexpr (PATTERN == 1) {
        u32x4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) v[i] = *reinterpret_cast<const u32x4*>(buf + ((size_t)(wave * NLOAD + i) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    } else if constexpr (PATTERN == 3 || PATTERN == 4) {
        // the coalesced skinny kernel's DMA: 8 rows x 128 B per wave instruction, rows `ld` bytes apart (3) or packed back to back (4:
        // what a K-blocked operand layout would give), wave w owns 128-byte column block(s) w, w + 8, ...
        const int srow = lane >> 3, ch = lane & 7;
        u32x4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int grp = i % 10, blk = i / 10;  // 80 rows per block
            const size_t row = grp * 8 + srow;
            const size_t off = PATTERN == 3 ? row * ld + (size_t)(wave + 8 * blk) * 128 + ch * 16 : ((size_t)(wave + 8 * blk) * 80 + row) * 128 + ch * 16;
            v[i] = *reinterpret_cast<const u32x4*>(buf + off);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    } else {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + ((size_t)(wave * NLOAD + i) * 64 + lane) * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (wave * NLOAD + i) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32x4 t = *reinterpret_cast<const u32x4*>(smem + (wave * NLOAD) * 1024 + lane * 16);
        acc += t[0] ^ t[1] ^ t[2] ^ t[3];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int PATTERN, int NLOAD>
static void run(const char* dbuf, unsigned int* sink, int wgs) {
    hipStream_t st;
    hipStreamCreate(&st);
    const int lds = PATTERN == 2 ? 8 * NLOAD * 1024 : 0;
    auto kern = fill_kernel<PATTERN, NLOAD>;
    if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    auto go = [&]() { hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, dbuf, 1536, sink); };
    for (int i = 0; i < 5; ++i) go();
    hipStreamSynchronize(st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 50; ++i) go();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < 8; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / 400.0, kb = 8.0 * NLOAD;
    printf("pattern %d, %3.0f KB per workgroup, %3d workgroups: %5.2f us per launch  -> %5.1f KB/us/CU above a 3.0 us floor (%4.1f B/clk at 2.1 GHz)\n", PATTERN, kb, wgs, us,
           kb / (us - 3.0), kb * 1024 / ((us - 3.0) * 2100.0));
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
}

int main() {
    char* dbuf;
    unsigned int* sink;
    hipMalloc(&dbuf, 4 << 20);
    hipMemset(dbuf, 1, 4 << 20);
    hipMalloc(&sink, 4096);
    for (int wgs : {192}) {
        run<0, 12>(dbuf, sink, wgs); run<1, 12>(dbuf, sink, wgs); run<2, 12>(dbuf, sink, wgs);
        run<0, 24>(dbuf, sink, wgs); run<1, 24>(dbuf, sink, wgs); run<2, 24>(dbuf, sink, wgs);
        run<0, 39>(dbuf, sink, wgs); run<1, 39>(dbuf, sink, wgs);
        run<3, 10>(dbuf, sink, wgs); run<4, 10>(dbuf, sink, wgs); run<3, 20>(dbuf, sink, wgs); run<4, 20>(dbuf, sink, wgs); run<3, 30>(dbuf, sink, wgs); run<4, 30>(dbuf, sink, wgs);
    }
    return 0;
}

// Microbenchmark: how fast can ONE CU move data global -> LDS, and does the path matter?
// Every conv / weight-gradient kernel of this repository stages its operands with global_load_lds_dwordx4 (1 KiB per wave
// instruction) and all of them plateau at about one piece per 42-60 cycles per CU ("DMA alone 1670 cycles
// per 40 pieces").  This tool isolates that rate: 256 workgroups of 512 threads (one per CU), each wave issues PIECES 1-KiB
// pieces per round from a source window of WINDOW bytes per workgroup (small window: L2-resident; large: HBM), waits, repeats.
//   mode 0: global_load_lds_dwordx4                  (LDS-DMA, what the kernels use)
//   mode 1: global_load_dwordx4 into VGPRs, dropped  (the plain vector-memory path)
// Reports bytes per shader clock per CU (clock from wall_clock64 = 100 MHz reference scaled by the measured ratio is avoided:
// we print GB/s per CU and the chip total; divide by the sustained clock yourself).
// build: hipcc -O3 --offload-arch=gfx950 glds_rate.hip -o glds_rate ; run: ./glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void stream(const char *__restrict__ src, size_t window, size_t stride, int rounds, unsigned *sink)
{
    __shared__ __attribute__((aligned(16))) char lds[8 * PIECES * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const char *base = src + (size_t)blockIdx.x * stride;   // stride 0: every workgroup streams the same window (L2 hits)
    const size_t per_round = (size_t)8 * PIECES * 1024;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        const size_t off = ((size_t)r * per_round) % window;
        const char *g = base + off + (size_t)wv * PIECES * 1024 + lane * 16;
        char *l = lds + wv * PIECES * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16(g + j * 1024, l + j * 1024);
            // the previous round's pieces must have landed before the next round overwrites them; this round's stay in flight
            if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            u32x4 v[PIECES];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) v[j] = *(const u32x4 *)(g + j * 1024);
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < PIECES; ++j) *(u32x4 *)(l + j * 1024 + lane * 16) = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < PIECES; ++j) { acc.x += v[j].x; acc.y ^= v[j].y; acc.z += v[j].z; acc.w ^= v[j].w; }
            }
        }
    }
    __syncthreads();
    if (MODE != 1) acc.x = ((const unsigned *)lds)[tid];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

template <int MODE, int PIECES> void run(const char *src, size_t window, size_t stride, unsigned *sink, const char *what)
{
    const int rounds = 4096 / PIECES;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, PIECES>), dim3(256), dim3(512), 0, 0, src, window, stride, rounds, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<MODE, PIECES>), dim3(256), dim3(512), 0, 0, src, window, stride, rounds, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double bytes = 256.0 * rounds * 8 * PIECES * 1024;
    printf("%-34s pieces/wave/round %d window %6zu KiB (%s): %7.3f ms  %7.1f GB/s chip  %6.2f GB/s per CU  (= %5.1f B/clk at 2.0 GHz)\n", what, PIECES,
           window >> 10, stride == 0 ? "shared" : stride <= (64u << 10) ? "private, L2" : stride <= (512u << 10) ? "private, MALL" : "private, HBM", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.0);
}

int main()
{
    const size_t big = (size_t)256 * (16 << 20);   // 16 MiB per workgroup: 4 GiB, streams from HBM
    char *src; unsigned *sink;
    if (hipMalloc(&src, big) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, big);
    for (int pass = 0; pass < 4; ++pass) {
        // 0: one 256-KiB window shared by every workgroup; 1 / 2: private 64-KiB / 512-KiB windows (2 / 16 MiB per XCD: inside /
        // beyond its 4-MiB L2, the latter served by the 256-MiB infinity cache); 3: private 16-MiB windows (HBM)
        const size_t window = pass == 0 ? (size_t)256 << 10 : pass == 1 ? (size_t)64 << 10 : pass == 2 ? (size_t)512 << 10 : (size_t)16 << 20;
        const size_t stride = pass ? window : 0;
        run<0, 4>(src, window, stride, sink, "LDS-DMA global_load_lds_dwordx4");
        run<0, 8>(src, window, stride, sink, "LDS-DMA global_load_lds_dwordx4");
        run<1, 4>(src, window, stride, sink, "global_load_dwordx4 -> VGPR");
        run<1, 8>(src, window, stride, sink, "global_load_dwordx4 -> VGPR");
        run<0, 16>(src, window, stride, sink, "LDS-DMA global_load_lds_dwordx4");
        run<1, 16>(src, window, stride, sink, "global_load_dwordx4 -> VGPR");
    }
    return 0;
}

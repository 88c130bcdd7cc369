// Microbenchmark: how long does a chip-wide burst of conv-epilogue-shaped stores take to be acknowledged?
// 256 workgroups x 8 waves each write a 256 x 256 bf16 tile (rows of 512 B inside a row-major [M][ld] tensor) as 16 passes of
// 16 B per lane, then wait for vmcnt(0); between bursts the workgroup idles for a "main loop" delay.
// build: hipcc -O3 --offload-arch=gfx950 store_burst.hip -o store_burst ; run: ./store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int POL>
__device__ __forceinline__ void st16(void *p, u32x4 v)
{
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int POL>
__global__ __launch_bounds__(512) void burst(char *out, int ld_bytes, int tiles_n, int rounds, int delay_us, int npass,
                                             unsigned long long *tlog)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 2, wn = wv & 3;
    for (int r = 0; r < rounds; ++r) {
        // idle "main loop"
        const unsigned long long t_start = wall_clock64();
        while (wall_clock64() - t_start < (unsigned long long)delay_us * 100ull) __builtin_amdgcn_s_sleep(32);
        __syncthreads();
        const int tile = r * gridDim.x + blockIdx.x;
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        char *base = out + ((size_t)(tm * 256 + wm * 128) * ld_bytes) + (size_t)(tn * 256 + wn * 64) * 2 + (lane & 7) * 16;
        const u32x4 v = {(unsigned)r, (unsigned)lane, (unsigned)wv, (unsigned)tile};
        const unsigned long long t0 = wall_clock64();
        for (int p = 0; p < npass; ++p) st16<POL>(base + (size_t)(p * 8 + (lane >> 3)) * ld_bytes, v);
        const unsigned long long t1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = wall_clock64();
        if (threadIdx.x == 0) {
            tlog[(size_t)(r * gridDim.x + blockIdx.x) * 2 + 0] = t1 - t0;
            tlog[(size_t)(r * gridDim.x + blockIdx.x) * 2 + 1] = t2 - t0;
        }
    }
}

template <int POL> void run(const char *name, int delay_us, int npass, int nwg)
{
    const int M = 131072, C = 512, rounds = 8;
    const size_t bytes = (size_t)M * C * 2 * 4;   // room for rounds * 256 tiles
    static char *out = nullptr;
    static unsigned long long *tlog = nullptr;
    if (!out) { hipMalloc(&out, bytes); hipMalloc(&tlog, sizeof(unsigned long long) * 2 * 256 * 64); }
    hipLaunchKernelGGL(burst<POL>, dim3(nwg), dim3(512), 0, 0, out, C * 2, C / 256, rounds, delay_us, npass, tlog);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * nwg * rounds);
    hipMemcpy(h.data(), tlog, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-10s delay %3d us, %2d passes (%3d KiB/WG), %3d WGs:", name, delay_us, npass, npass * 8, nwg);
    for (int r = 1; r < rounds; r += 3) {
        double iss = 0, tot = 0, mx = 0;
        for (int b = 0; b < nwg; ++b) {
            iss += h[(size_t)(r * nwg + b) * 2]; tot += h[(size_t)(r * nwg + b) * 2 + 1];
            mx = std::max(mx, (double)h[(size_t)(r * nwg + b) * 2 + 1]);
        }
        printf("  r%d issue %.2f ack mean %.2f max %.2f us", r, iss / nwg / 100, tot / nwg / 100, mx / 100);
    }
    printf("\n");
}

int main()
{
    for (int nwg : {256, 32, 8}) {
        for (int npass : {16, 4}) {
            run<0>("plain", 50, npass, nwg);
            run<1>("nt", 50, npass, nwg);
            run<2>("sc1", 50, npass, nwg);
            run<3>("sc0 sc1", 50, npass, nwg);
        }
    }
    run<0>("plain", 5, 16, 256);
    run<0>("plain", 200, 16, 256);
    return 0;
}

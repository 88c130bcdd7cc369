// Microbenchmark: MFMA issue rate of one wave per SIMD, alone and next to a sibling wave that (a) also issues MFMAs,
// (b) issues LDS-DMA pieces, (c) issues ds_read_b128.  One workgroup per CU; cycles per MFMA of wave 0.
// build: hipcc -O3 --offload-arch=gfx950 mfma_issue.hip -o mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// mode: sibling role 0 = none (4 waves/CU), 1 = MFMA too, 2 = LDS-DMA stream, 3 = ds_read_b128 stream
template <int SHAPE32>
__global__ __launch_bounds__(512) void k(int mode, int iters, const char *src, unsigned long long *out, float *sink)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool primary = wv < 4;            // waves 0-3: one per SIMD
    if (!primary && mode == 0) return;
    f32x4_t acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (short)(lane + i + e);
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (short)(lane * 3 + i + e);
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (primary || mode == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i * 4 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode == 2) {
        const char *g = src + ((size_t)blockIdx.x * 8 + wv) * 65536 + lane * 16;
        const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (wv - 4) * 8192;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int pz = 0; pz < 6; ++pz) glds16(g + ((it * 6 + pz) & 31) * 1024, __builtin_amdgcn_readfirstlane(la + pz * 1024));
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        uint4 s = make_uint4(0, 0, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int pz = 0; pz < 12; ++pz) {
                const uint4 v = *(const uint4 *)(lds + ((pz * 1024 + lane * 16 + it * 64) & 65535));
                s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w;
            }
        }
        if (s.x == 0x12345) sink[0] = 1.f;
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][3];
    if (r == 1.2345f) sink[1] = r;
}

int main()
{
    const int iters = 200, nwg = 256;
    char *src; unsigned long long *out; float *sink;
    hipMalloc(&src, (size_t)nwg * 8 * 65536); hipMemset(src, 1, (size_t)nwg * 8 * 65536);
    hipMalloc(&out, nwg * 8 * 8); hipMalloc(&sink, 16);
    const char *names[4] = {"alone (1 wave/SIMD)", "sibling also MFMA", "sibling LDS-DMA stream", "sibling ds_read_b128 stream"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(out, 0, nwg * 8 * 8);
        hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(512), 0, 0, mode, iters, src, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[256 * 8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double p = 0, s = 0;
        for (int b = 0; b < nwg; ++b) { p += h[b * 8 + 0]; s += h[b * 8 + 4]; }
        printf("%-30s primary wave: %6.2f cycles/MFMA   sibling wave total %8.0f cycles (%0.1f per primary MFMA)\n", names[mode],
               p / nwg / (iters * 32.0), s / nwg, s / nwg / (iters * 32.0));
    }
    return 0;
}

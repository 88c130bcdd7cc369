// Micro-benchmark: sustained rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (register operands only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[16];
    v2f y[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (v2f){x[2 * i], x[2 * i + 1]};
    const v2f aa = (v2f){a, a * 1.01f}, bb = (v2f){b, b * 0.99f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(aa), "v"(bb));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256 * waves;   // 256 CUs x (waves) blocks of 4 waves -> `waves` waves per SIMD
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)blocks * 256 * iters * 64.0;   // FMAs: 64 scalar fma or 32 pk (x2) per iteration
            printf("%s waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", waves, ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}

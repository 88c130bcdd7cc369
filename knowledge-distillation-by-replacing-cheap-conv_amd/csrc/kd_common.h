// Shared device/host helpers for the gfx950 kernels behind include/kdcc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kdcc.h"

// ---- error plumbing (never throws across the C boundary) --------------------
void kd_set_error(const char *fmt, ...);

#define KD_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            kd_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

#define KD_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            kd_set_error("%s: launch failed: %s", (name), hipGetErrorString(e__)); \
            return KD_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

// ---- result-altering diagnostics (skip-the-epilogue timing ablations, dropped kernel phases, in-kernel timestamps) ----
// exist only in a tuning build (`make TUNING=1` -> libkdcc_hip_tuning.so, loaded with KDCC_LIB=tuning by tools/): the default
// library never reads them, so a stray environment variable cannot change what training computes.
#ifdef KDCC_TUNING
#include <stdlib.h>
static inline int kd_tuning_env_int(const char *name)
{
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}
#define KD_TUNING_ENV_INT(name) kd_tuning_env_int(name)
#else
#define KD_TUNING_ENV_INT(name) 0
#endif

// ---- kernel-selection log (kd_debug_kernel_log_*: which device kernel each dispatcher picked) ------------------------
// Host-side counters only; nothing about a launch changes.  `name` must be a string literal (entries are keyed by pointer
// first, by text second).
void kd_note_kernel(const char *name);
#define KD_NOTE_KERNEL(name) kd_note_kernel(name)

static inline bool kd_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int kd_elem_size(int dtype) { return dtype == KD_BF16 ? 2 : 4; }

// ---- element types ------------------------------------------------------------
typedef uint16_t bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, keeps NaN a NaN) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int DT = KD_F32;
    __device__ static __forceinline__ float ld(const float *p) { return *p; }
    __device__ static __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int DT = KD_BF16;
    __device__ static __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

// dtype-erased scalar load/store (used by the strided loss / copy kernels)
__device__ __forceinline__ float kd_ld(const void *p, int dtype, int64_t i)
{
    return dtype == KD_BF16 ? bf16_to_f32(((const bf16_t *)p)[i]) : ((const float *)p)[i];
}
__device__ __forceinline__ void kd_st(void *p, int dtype, int64_t i, float v)
{
    if (dtype == KD_BF16) ((bf16_t *)p)[i] = f32_to_bf16(v);
    else ((float *)p)[i] = v;
}

// 8 consecutive elements <-> 8 floats (16 B of bf16 / 32 B of f32); p must be 16-B aligned
__device__ __forceinline__ void ld8(const float *p, float (&v)[8])
{
    const float4 a = ((const float4 *)p)[0], b = ((const float4 *)p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t *p, float (&v)[8])
{
    const uint4 u = *(const uint4 *)p;
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float *p, const float (&v)[8])
{
    ((float4 *)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    ((float4 *)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ void st8(bf16_t *p, const float (&v)[8])
{
    *(uint4 *)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                             pack_bf16x2(v[6], v[7]));
}

// ---- reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): gives each XCD a
// contiguous range of logical tile ids so neighbouring tiles share one L2.
// Bijective for any grid size (cdna_hip_programming.md, 8-phase template notes).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// MFMA fragment math shared by the implicit-GEMM conv and the 1x1 weight-gradient kernels.
// LDS tile format (both operands): 128 rows x 128 bytes of K, 16-B chunk c of row r stored at
// slot c ^ (r & 7) -- conflict-free for the ds_read_b128 fragment reads below.
#pragma once
#include "kd_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int IG_ROWB = 128;  // bytes of K per LDS row per stage

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS at (wave-uniform base + lane*16).
// Issued through inline asm on purpose: hipcc then does not track the DMA, so it inserts no vmcnt waits of its own
// in front of the fragment reads (at a loop header it falls back to vmcnt(0), which would drain the prefetch); every
// wait for these pieces is the hand-counted wait_stage_barrier / wait_vm_barrier.  M0 carries the LDS base and is restored
// (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    const uint32_t lds_addr =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}


// Several LDS-DMA pieces under ONE M0 value: the instruction's immediate offset applies to both the global and the LDS address,
// so piece j goes to (lds_base + OFFj + lane*16) from (gsrc_j + OFFj) -- callers pass gsrc_j already reduced by OFFj bytes.
// Rewriting M0 per piece serialises a wave's DMA issue at ~110 cycles per piece (measured: the next s_mov m0 waits until the
// previous global_load_lds has read it); with one M0 per group the pieces issue back to back.
__device__ __forceinline__ void glds16_x4(const void *g0, const void *g1, const void *g2, const void *g3, void *lds_base)
{
    const uint32_t lds_addr =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_addr)
                 : "memory");
}
// five pieces: lds_base is the address of piece 2 (offsets -2048 .. +2048)
__device__ __forceinline__ void glds16_x5(const void *g0, const void *g1, const void *g2, const void *g3, const void *g4,
                                          void *lds_base)
{
    const uint32_t lds_addr =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:-2048\n\t"
                 "global_load_lds_dwordx4 %2, off offset:-1024\n\t"
                 "global_load_lds_dwordx4 %3, off\n\t"
                 "global_load_lds_dwordx4 %4, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %5, off offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "s"(lds_addr)
                 : "memory");
}

// Wait until at most N of this wave's vector-memory operations (here: LDS-DMA pieces) are outstanding and all LDS
// reads have returned; then a bare barrier.  Unlike __syncthreads() this does not drain the DMA of the stages still
// in flight, which is what lets the prefetch run NST-1 K stages ahead.
template <int N> __device__ __forceinline__ void wait_vm_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
}
// r = number of stages issued after the one that must have landed (0 .. NST-2); G = pieces per wave per stage
template <int G> __device__ __forceinline__ void wait_stage_barrier(int r)
{
    if (r <= 0) wait_vm_barrier<0>();
    else if (r == 1) wait_vm_barrier<G>();
    else wait_vm_barrier<2 * G>();
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4 &a, const uint4 &b, f32x4_t &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // the 16-B chunk holds 4 consecutive k; MFMA e consumes element e of every lane's chunk:
    // lane (row r, quad q) supplies k = 4*chunk(q) + e on both operands, so the sum over q is consistent.
    __device__ static __forceinline__ void run(const uint4 &a, const uint4 &b, f32x4_t &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// One K stage of a wave's (16*MI) x 64 sub-tile:  acc[i][j] += A(rows wm*16*MI + 16i ..) . B(rows wn*64 + 16j ..)^T
// RB = bytes of K per LDS row per stage: 128 (two MFMA k-steps; 16-B chunk c of row r at slot c ^ (r & 7)) or
//      64 (one k-step; chunk c of row r at slot c ^ ((r >> 1) & 3)).  Both are conflict-free for ds_read_b128.
// SWAP: issue the MFMAs with the operands exchanged, i.e. accumulate the transposed 16x16 tiles (lane = pixel, registers =
// four consecutive output channels), the layout conv_igemm.hip's epilogue consumes.
template <typename T, int MI = 4, int RB = 128, bool SWAP = false>
__device__ __forceinline__ void ig_compute_stage(const char *sA, const char *sB, int wm, int wn, int lane,
                                                 f32x4_t (&acc)[MI][4])
{
    const int frow = lane & 15, fq = lane >> 4;
    const char *A = sA + (wm * 16 * MI + frow) * RB;
    const char *B = sB + (wn * 64 + frow) * RB;
#pragma unroll
    for (int ks = 0; ks < RB / 64; ++ks) {
        const int sw = RB == 128 ? (((fq + 4 * ks) ^ (lane & 7)) << 4) : ((fq ^ ((frow >> 1) & 3)) << 4);
        uint4 a[MI], b[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB + sw);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB + sw);
        if constexpr (sizeof(T) == 4) {
            // fp32 parity path: the f32 MFMA is an exact k-ordered fmaf chain with no wider internal accumulation, so a
            // single chain over K = 4608..9216 products drifts ~2e-5 relative per conv on zero-mean (backward) operands.
            // Accumulate each k-step (16 products) in a fresh register tile and add it to the running sum: short chains +
            // ~K/16 fp32 adds, i.e. blocked summation like the CPU reference's GEMM.
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                f32x4_t part[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    part[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    if (SWAP) Mma<T>::run(b[j], a[i], part[j]);
                    else Mma<T>::run(a[i], b[j], part[j]);
                    acc[i][j] += part[j];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) Mma<T>::run(b[j], a[i], acc[i][j]);
                    else Mma<T>::run(a[i], b[j], acc[i][j]);
                }
        }
        // issue the k-step's fragment reads back to back, then its MFMAs: one exposed LDS latency per k-step instead
        // of one per pair of reads (hipcc otherwise interleaves read-wait-4 MFMAs to save registers)
        __builtin_amdgcn_sched_group_barrier(0x100, MI + 4, 0);
        if constexpr (sizeof(T) != 4) __builtin_amdgcn_sched_group_barrier(0x008, MI * 4, 0);
    }
}

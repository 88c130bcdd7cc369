// Pieces shared by the dense-conv translation units (conv_igemm.hip, conv_lw.hip): launch parameters, the persistent kernels'
// epilogue (accumulators -> wave-private LDS patch -> 16-B rows) and the per-XCD tile walk.
#pragma once
#include "igemm_core.h"

namespace kdconv {
struct ConvParams {
    const void *x;
    const void *w;
    int M, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, ldx;
    int HoWo, nkc, nk, Ktot, tiles_n, ntiles, tiles_m, tn_group;
    int vec_ok;  // every epilogue pointer/stride is 16-B friendly
    int epi_batch;  // A/B hook: 0 = one pass at a time (KDCC_EPI_BATCH=0)
    int tune;       // A/B hook (KDCC_CONV_TUNE)
    int stagger_us; // A/B hook (KDCC_CONV_STAGGER, with tune & 16384)
    // second A source of a K-concatenated 1x1 conv (kd_conv1x1_dual_fwd): K stages [0, nk1) read x, [nk1, nk) read x2 (its own
    // pixel stride); w is [Cout][Cin + Cin2].  nk1 == nk when there is one source.
    const void *x2;
    int ldx2, nk1;
    kd_conv_epilogue ep;
};
}  // namespace kdconv
using kdconv::ConvParams;

namespace {

__device__ __forceinline__ void wait_vm_stores(int nst)
{
    // at most nst (= 16 * outputs) younger stores may stay outstanding; everything older -- the prologue DMAs -- has landed
    if (nst == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (nst == 32) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
    else if (nst == 48) asm volatile("s_waitcnt vmcnt(48) lgkmcnt(0)" ::: "memory");
    else if (nst == 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");   // + the four stores of the eval-BN sums
    else if (nst == 36) asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// the same without the barrier
__device__ __forceinline__ void wait_vm_only(int nst)
{
    if (nst == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (nst == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (nst == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (nst == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (nst == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Accumulators -> memory through a wave-private 2-KiB LDS patch OUTSIDE the stage buffers (which the next tile's prologue is
// already filling): one 16-pixel x 64-channel row tile at a time is written as bf16 (8-B chunk c of row r at c ^ r), then
// read back as two passes of 8 rows x 128 B, 16 B per lane -- whole 128-B lines per pixel for the operand loads and the
// stores (8-B accumulator-layout accesses straight to memory, 32-B pieces of 32 lines per instruction, measured 7-40 %
// slower per layer).  Arithmetic and rounding are those of ig_epilogue's bf16 fast path.
// Operand loads (residuals, mask): vmcnt retires in issue order, so a load issued after a batch of stores returns only once
// those stores are acknowledged -- with load -> store -> load per 32 rows the [res_pre, raw, act] epilogue of a 512-channel
// 3x3 tile took 21 us against 5.6 us without operands (tools/conv_timeline.py).  The host passes at most two operands
// (slot 0 / slot 1 in the order res_pre, mask, res_post): a single operand is loaded for the whole 128-row sub-tile up
// front, two operands for 64 rows at a time (one store -> load hand-over instead of three).  Issues exactly 16 stores per
// output and lane.
// PT: row tiles (2 KiB each) per LDS round trip -- the patch is PT * 2 KiB per wave where the stage buffers leave room
__device__ __forceinline__ uint2 ig_pack_acc(const f32x4_t &v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
__device__ __forceinline__ uint2 ig_pack_acc(const uint2 &v) { return v; }   // (conv_lw.hip hands over packed accumulators)

// one v_cvt_pk_bf16_f32 per pair (a vector cast; pack_bf16x2's two scalar casts cost three instructions -- kept in the
// ping-pong kernels, where the vector form once moved a compiler wait into a main loop: DESIGN.md section 5)
__device__ __forceinline__ uint32_t pack_bf16x2_v(float lo, float hi)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ void st8_v(bf16_t *p, const float (&v)[8])
{
    *(uint4 *)p = make_uint4(pack_bf16x2_v(v[0], v[1]), pack_bf16x2_v(v[2], v[3]), pack_bf16x2_v(v[4], v[5]), pack_bf16x2_v(v[6], v[7]));
}

// The lone-wave kernels' epilogue (OUTS != 0) is bound by its own instruction count -- one wave per SIMD, 4+ cycles per VALU
// instruction, ~1700 of them per 128 x 128 wave tile (tools/duo_timeline.py --tall: 6.5 us of a 40-us 512 x 128 tile with the stores
// removed).  Three things take instructions out without changing a bit of the result:
//   * rows are addressed through a buffer resource based at the wave's first row: per-lane byte offset computed once, the row
//     step is a scalar (the generic path spends ~6 VALU instructions of 64-bit address arithmetic per 16-B access).  Loads take
//     it as the instruction's scalar offset; STORES add it to the lane offset (one VALU add) and leave the scalar-offset field 0:
//     with an SGPR there, hipcc assumes (as the gfx9 hazard table says) that the store's four data registers may be rewritten by
//     the very next VALU instruction -- on gfx950 the second data dword of raw stores then came out corrupted in 0.4 % of the
//     elements whenever arithmetic on the same registers followed (the raw + act epilogues; tools/lw_check.py);
//   * act = v * scale + shift on pairs (v_pk_fma_f32: the same fused arithmetic per element);
//   * ReLU after the rounding, on the packed bf16 pair as v_pk_max_i16(x, 0) -- rounding is monotonic and keeps the sign, a negative
//     bf16 is a negative int16 (-0 included: max(-32768, 0) = +0 as fmaxf(-0, 0)); without ReLU the bound is -32768 (identity).
typedef float f32x2_v __attribute__((ext_vector_type(2)));
typedef short i16x2_v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);   // raw buffer, offsets stay far below 2 GiB
}
__device__ __forceinline__ uint32_t act_pair(float lo, float hi, f32x2_v sc, f32x2_v sh, i16x2_v relu_lo)
{
    const f32x2_v r = __builtin_elementwise_fma((f32x2_v){lo, hi}, sc, sh);
    const i16x2_v q = __builtin_bit_cast(i16x2_v, pack_bf16x2_v(r[0], r[1]));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(q, relu_lo));
}

// OUTS: 0 = which outputs exist is read from the epilogue descriptor per row (the ping-pong kernels); 1 raw, 2 act, 3 both at
// compile time + single-instruction packing (conv_lw.hip: a lone wave's epilogue is bound by its own instruction count)
template <int MI, int NOPS_, int PT = 1, typename ACC = f32x4_t, int OUTS = 0>   // NOPS_ = operands | 4 when the eval-BN sums are taken (kernels' NOPS parameter)
__device__ __forceinline__ void ig_epilogue_rows16(const ConvParams &p, char *patch, ACC (&acc)[MI][4], int mw, int nw,
                                                   int lane)
{
    typedef bf16_t T;
    constexpr int NOPS = NOPS_ & 3;
    constexpr bool SUMS = (NOPS_ & 4) != 0;
    // NOPS_ == 8 (round 6): no operands, and the per-channel sums of the OUTPUT over the wave's 128 rows go to the same partial rows
    // (S1 = sum of the stored values, S2 = 0) -- the global average pool of models/deeplabv3/deeplabv3.py:59-62 taken where the
    // tensor it pools is produced, instead of another pass over 2 GB
    constexpr bool OSUMS = NOPS_ == 8;
    static_assert((MI == 8 || (MI == 4 && !(NOPS_ & 12))) && NOPS_ >= 0 && (NOPS_ <= 7 || NOPS_ == 8) && NOPS_ != 4,
                  "128-row wave sub-tiles (64 rows at a time: conv_row_duo_kernel, no sums); eval-BN sums need the mask operand, output sums no operand");
    const kd_conv_epilogue &e = p.ep;
    // every per-lane address below derives from this copy: the compiler cannot hoist them out of the tile loop into
    // registers that would stay live through the main loop (which runs at the 256-VGPR limit)
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
    const int c0 = nw + (lane & 7) * 8, lrow = lane >> 3, c2 = (lane & 7) * 2;
    float mscale[8], ascale[8], ashift[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { mscale[q] = 1.f; ascale[q] = 1.f; ashift[q] = 0.f; }
    // operand slots
    // (NOPS == 0: compile-time false -- as run-time flags the three blocks below become 24 selects per row)
    const bool has_p = NOPS == 3 || (NOPS > 0 && e.res_pre != nullptr), has_m = NOPS == 3 || (NOPS > 0 && e.mask != nullptr),
               has_q = NOPS == 3 || (NOPS > 0 && e.res_post != nullptr);   // (three operands: all of them, compile-time)
    const float relu_lo = e.act_relu ? 0.f : -INFINITY;   // max(a, -inf) = a: no select per element
    if (has_m && e.mask_scale) ld8(e.mask_scale + c0, mscale);
    if (e.act_scale) ld8(e.act_scale + c0, ascale);
    if (e.act_shift) ld8(e.act_shift + c0, ashift);
    // lone-wave path: the outputs' rows through buffer resources (see above)
    const int mwu = __builtin_amdgcn_readfirstlane(mw);
    const __amdgpu_buffer_rsrc_t rs_raw = rows_rsrc((OUTS & 1) ? (const T *)e.out_raw + (size_t)mwu * e.ld_raw : nullptr);
    const __amdgpu_buffer_rsrc_t rs_act = rows_rsrc((OUTS & 2) ? (const T *)e.out_act + (size_t)mwu * e.ld_act : nullptr);
    const uint32_t vo_raw = (uint32_t)(lrow * e.ld_raw + c0) * 2u, vo_act = (uint32_t)(lrow * e.ld_act + c0) * 2u;
    const short relu_w = e.act_relu ? (short)0 : (short)-32768;
    const i16x2_v relu_i16 = {relu_w, relu_w};
    const T *s0 = (const T *)(has_p ? e.res_pre : has_m ? e.mask : e.res_post);
    const int ld0 = has_p ? e.ld_res_pre : has_m ? e.ld_mask : e.ld_res_post;
    const T *s1 = (const T *)(has_p && has_m ? e.mask : e.res_post);
    const int ld1 = has_p && has_m ? e.ld_mask : e.ld_res_post;
    constexpr int nops = NOPS;   // == operands present (host)
    const bool m_in1 = has_p && has_m, q_in1 = has_q && nops == 2;
    // 64 rows (8 passes) of up to two operands in flight: one operand -> ra = rows 0..63, rb = rows 64..127, both loaded up
    // front; two operands -> ra / rb = slot 0 / slot 1 of the current 64 rows.  The accumulators are rounded to packed bf16
    // in place right after the first loads are issued (the patch holds bf16 anyway), which halves their registers.
    uint4 ra[8], rb[8], rc[NOPS == 3 ? 8 : 1];   // three operands: 64 rows of res_pre / mask / res_post
    auto rowof = [&](int pass) { return (size_t)(mw + pass * 8 + lrow); };
    auto load64 = [&](const T *src, int ld, int hb, uint4 (&r)[8]) __attribute__((always_inline)) {
        if constexpr (OUTS != 0) {
            // (mw is wave-uniform: the resource and the row step live in SGPRs)
            const __amdgpu_buffer_rsrc_t rs = rows_rsrc(src + (size_t)__builtin_amdgcn_readfirstlane(mw) * ld);
            const uint32_t vo = (uint32_t)(lrow * ld + c0) * 2u;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const u32x4_v q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (hb * 8 + ps) * 16 * ld, 0);
                r[ps] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) r[ps] = *(const uint4 *)(src + rowof(hb * 8 + ps) * ld + c0);
        }
    };
    // eval-BN parameter sums of the masked gradient (backward: v = gradient w.r.t. the BN output where the activation is on):
    // S1[c] = sum_m v * mask_scale, S2[c] = sum_m v * mask_scale * act -- what kd_channel_sums would read back from memory
    float bs1[8], bs2[8];
    // (SUMS -- host: only with a mask, and never without the pointer)
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs1[q] = 0.f; bs2[q] = 0.f; }
    // pack first (frees half the accumulator registers), then the loads: no spills
    uint2 pk[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[i][j] = ig_pack_acc(acc[i][j]);
    if (nops >= 1) load64(s0, ld0, 0, ra);
    if (nops >= 2) load64(s1, ld1, 0, rb);
    if (nops == 1 && MI == 8) load64(s0, ld0, 1, rb);
    if constexpr (NOPS == 3) load64((const T *)e.res_post, e.ld_res_post, 0, rc);
    // hipcc's wait-count insertion is path-insensitive: a load whose uses sit under run-time conditions counts as pending on
    // the paths that skip them, and the wait it then needs lands in front of the main loop's first ds_read (which reuses the
    // registers) -- inside the hand-scheduled loop, draining the LDS-DMA every stage (tools/check_loop_waits.py).  The empty
    // asm statements below and in the row loop are unconditional uses: the waits are inserted here, where the first row's
    // arithmetic would have waited anyway.
    asm volatile("" ::"v"(ascale[0]), "v"(ascale[1]), "v"(ascale[2]), "v"(ascale[3]), "v"(ascale[4]), "v"(ascale[5]), "v"(ascale[6]),
                 "v"(ascale[7]), "v"(ashift[0]), "v"(ashift[1]), "v"(ashift[2]), "v"(ashift[3]), "v"(ashift[4]), "v"(ashift[5]),
                 "v"(ashift[6]), "v"(ashift[7]));
    if (NOPS > 0)
        asm volatile("" ::"v"(mscale[0]), "v"(mscale[1]), "v"(mscale[2]), "v"(mscale[3]), "v"(mscale[4]), "v"(mscale[5]), "v"(mscale[6]),
                     "v"(mscale[7]));
    const f32x2_v asc2[4] = {{ascale[0], ascale[1]}, {ascale[2], ascale[3]}, {ascale[4], ascale[5]}, {ascale[6], ascale[7]}};
    const f32x2_v ash2[4] = {{ashift[0], ashift[1]}, {ashift[2], ashift[3]}, {ashift[4], ashift[5]}, {ashift[6], ashift[7]}};
#pragma unroll
    for (int hb = 0; hb < MI / 4; ++hb) {   // 64 rows each
        if (nops >= 2 && hb == 1) { load64(s0, ld0, 1, ra); load64(s1, ld1, 1, rb); }
        if constexpr (NOPS == 3) { if (hb == 1) load64((const T *)e.res_post, e.ld_res_post, 1, rc); }
#pragma unroll
        for (int ig = 0; ig < 4; ig += PT) {
#pragma unroll
            for (int it = 0; it < PT; ++it) {
                const int i = 4 * hb + ig + it;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(uint2 *)(patch + it * 2048 + frow * 128 + (((j * 4 + fq) ^ frow) << 3)) = pk[i][j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < PT; ++it) {
            const int ii = ig + it;
            const char *pt = patch + it * 2048;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = h * 8 + lrow, ps = 2 * ii + h;       // ps: 8-row group inside the 64 rows
                const size_t m = rowof(hb * 8 + ps);
                const uint2 lo = *(const uint2 *)(pt + row * 128 + ((c2 ^ row) << 3));
                const uint2 hi = *(const uint2 *)(pt + row * 128 + (((c2 + 1) ^ row) << 3));
                const uint4 rawv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                float v[8], t[8];
                ld8((const bf16_t *)&rawv, v);
                const uint4 o0 = (hb == 1 && nops == 1) ? rb[ps] : ra[ps], o1 = rb[ps];
                if (NOPS >= 1) asm volatile("" ::"v"(o0.x), "v"(o0.y), "v"(o0.z), "v"(o0.w));
                if (NOPS >= 2) asm volatile("" ::"v"(o1.x), "v"(o1.y), "v"(o1.z), "v"(o1.w));
                if constexpr (NOPS == 3) asm volatile("" ::"v"(rc[ps].x), "v"(rc[ps].y), "v"(rc[ps].z), "v"(rc[ps].w));
                if (has_p) {
                    ld8((const T *)&o0, t);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += t[q];
                }
                if (has_m) {
                    const uint4 om = m_in1 ? o1 : o0;
                    ld8((const T *)&om, t);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = t[q] > 0.f ? v[q] * mscale[q] : 0.f;
                    if constexpr (SUMS) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { bs1[q] += v[q]; bs2[q] = fmaf(v[q], t[q], bs2[q]); }
                    }
                }
                if (has_q) {
                    const uint4 oq = NOPS == 3 ? rc[NOPS == 3 ? ps : 0] : (q_in1 ? o1 : o0);
                    ld8((const T *)&oq, t);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += t[q];
                }
                if constexpr (OSUMS) {     // (no operands: v holds the bf16 values that are stored)
#pragma unroll
                    for (int q = 0; q < 8; ++q) bs1[q] += v[q];
                }
                if constexpr (OUTS != 0) {
                    const int srow = (hb * 8 + ps) * 16;       // bytes per ld: row (hb * 8 + ps) * 8 of the wave's 128
                    if constexpr ((OUTS & 1) != 0) {
                        const u32x4_v o = nops == 0 ? (u32x4_v){rawv.x, rawv.y, rawv.z, rawv.w}
                                                    : (u32x4_v){pack_bf16x2_v(v[0], v[1]), pack_bf16x2_v(v[2], v[3]), pack_bf16x2_v(v[4], v[5]), pack_bf16x2_v(v[6], v[7])};
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_raw, vo_raw + (uint32_t)(srow * e.ld_raw), 0, 0);
                    }
                    if constexpr ((OUTS & 2) != 0) {
                        const u32x4_v o = {act_pair(v[0], v[1], asc2[0], ash2[0], relu_i16), act_pair(v[2], v[3], asc2[1], ash2[1], relu_i16),
                                           act_pair(v[4], v[5], asc2[2], ash2[2], relu_i16), act_pair(v[6], v[7], asc2[3], ash2[3], relu_i16)};
#ifdef KDCC_TUNING
                        if (p.tune & 128) { asm volatile("" ::"v"(o)); continue; }      // timing ablation: everything but the store itself
                        if (p.tune & 256) { *(u32x4_v *)((T *)e.out_act + (size_t)(((m & 15) + 16 * (blockIdx.x * 8 + (threadIdx.x >> 6))) % 4096) * e.ld_act + c0) = o; continue; }
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_act, vo_act + (uint32_t)(srow * e.ld_act), 0, 0);
                    }
                    continue;
                }
                if (OUTS ? (OUTS & 1) != 0 : e.out_raw != nullptr) {
                    if (nops == 0) *(uint4 *)((T *)e.out_raw + m * e.ld_raw + c0) = rawv;
                    else if (OUTS) st8_v((T *)e.out_raw + m * e.ld_raw + c0, v);
                    else st8((T *)e.out_raw + m * e.ld_raw + c0, v);
                }
                if (OUTS ? (OUTS & 2) != 0 : e.out_act != nullptr) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        t[q] = fmaxf(v[q] * ascale[q] + ashift[q], relu_lo);
                    }
#ifdef KDCC_TUNING
                    // timing ablations (tuning build): 128 = everything but the store itself; 256 = the store lands in a 16-KiB
                    // window per wave (same instruction stream, the bytes stay in L2)
                    if (p.tune & 128) { asm volatile("" ::"v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7])); continue; }
                    if (p.tune & 256) { st8((T *)e.out_act + (size_t)(((m & 15) + 16 * (blockIdx.x * 8 + (threadIdx.x >> 6))) % 4096) * e.ld_act + c0, t); continue; }
#endif
                    if (OUTS) st8_v((T *)e.out_act + m * e.ld_act + c0, t);
                    else st8((T *)e.out_act + m * e.ld_act + c0, t);
                }
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if constexpr (SUMS || OSUMS) {
        // lanes l, l + 8, .., l + 56 hold the same 8 channels (rows lrow + 8 k): butterfly over lane bits 3-5 in a fixed order,
        // then lane l < 8 writes partial row mw / 128 of [M / 128][2][Cout] (summed in row order by kd_bn_sums_finish)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int sft = 8; sft < 64; sft <<= 1) {
                bs1[q] += __shfl_xor(bs1[q], sft, 64);
                bs2[q] += __shfl_xor(bs2[q], sft, 64);
            }
        }
        if (lane < 8) {
            float *dst = e.bn_sums + (size_t)(mw >> 7) * 2 * p.Cout + c0;
            *(float4 *)dst = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
            *(float4 *)(dst + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
            *(float4 *)(dst + p.Cout) = make_float4(bs2[0], bs2[1], bs2[2], bs2[3]);
            *(float4 *)(dst + p.Cout + 4) = make_float4(bs2[4], bs2[5], bs2[6], bs2[7]);
        }
    }
}

// ---- the lone-wave kernels' epilogue (conv_lw.hip: 128-row x 64-channel halves of a 128 x 128 wave tile, packed accumulators) ----------
// Same values, same arithmetic and rounding as ig_epilogue_rows16<8, NOPS_, 2, uint2, OUTS> (bit-identical outputs), rearranged for a
// wave that has the SIMD to itself and therefore nobody to hide its latencies behind (tools/duo_timeline.py --tall: of the 2.6 us a
// half took without its stores, more than half was waiting -- for the per-channel constants' loads and for four serial LDS round trips):
//   * the per-channel constants are loaded by lw_epilogue_consts() ahead of time (the caller issues half 0's before it reads the
//     accumulators out of the AGPRs and half 1's before half 0's epilogue);
//   * the LDS transposes are software-pipelined over the two 2-KiB halves of the wave's patch, one 16-row tile each: while tile i is
//     unpacked, combined with its operands and stored, tile i + 1's read and tile i + 2's write are already in the LDS queue
//     (in-order per wave: a write issued behind a read of the same patch cannot overtake it).
struct LwEpiConsts {
    float mscale[8], ascale[8], ashift[8];
};
template <int NOPS_>
__device__ __forceinline__ void lw_epilogue_consts(const ConvParams &p, int nw, int lane, LwEpiConsts &k)
{
    constexpr int NOPS = NOPS_ & 3;
    const kd_conv_epilogue &e = p.ep;
    asm volatile("" : "+v"(lane));
    const int c0 = nw + (lane & 7) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) { k.mscale[q] = 1.f; k.ascale[q] = 1.f; k.ashift[q] = 0.f; }
    const bool has_m = NOPS == 3 || (NOPS > 0 && e.mask != nullptr);
    if (has_m && e.mask_scale) ld8(e.mask_scale + c0, k.mscale);
    if (e.act_scale) ld8(e.act_scale + c0, k.ascale);
    if (e.act_shift) ld8(e.act_shift + c0, k.ashift);
}

template <int NOPS_, int OUTS>
__device__ __forceinline__ void lw_epilogue_rows16(const ConvParams &p, char *patch, uint2 (&acc)[8][4], int mw, int nw, int lane, const LwEpiConsts &kc)
{
    typedef bf16_t T;
    constexpr int NOPS = NOPS_ & 3;
    constexpr bool SUMS = (NOPS_ & 4) != 0;
    static_assert(NOPS_ >= 0 && NOPS_ <= 7 && NOPS_ != 4 && OUTS >= 1 && OUTS <= 3, "operands | 4 (eval-BN sums, with the mask operand); raw / act / both");
    const kd_conv_epilogue &e = p.ep;
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
    const int c0 = nw + (lane & 7) * 8, lrow = lane >> 3, c2 = (lane & 7) * 2;
    const bool has_p = NOPS == 3 || (NOPS > 0 && e.res_pre != nullptr), has_m = NOPS == 3 || (NOPS > 0 && e.mask != nullptr),
               has_q = NOPS == 3 || (NOPS > 0 && e.res_post != nullptr);
    const int mwu = __builtin_amdgcn_readfirstlane(mw);
    const __amdgpu_buffer_rsrc_t rs_raw = rows_rsrc((OUTS & 1) ? (const T *)e.out_raw + (size_t)mwu * e.ld_raw : nullptr);
    const __amdgpu_buffer_rsrc_t rs_act = rows_rsrc((OUTS & 2) ? (const T *)e.out_act + (size_t)mwu * e.ld_act : nullptr);
    const uint32_t vo_raw = (uint32_t)(lrow * e.ld_raw + c0) * 2u, vo_act = (uint32_t)(lrow * e.ld_act + c0) * 2u;
    const short relu_w = e.act_relu ? (short)0 : (short)-32768;
    const i16x2_v relu_i16 = {relu_w, relu_w};
    const T *s0 = (const T *)(has_p ? e.res_pre : has_m ? e.mask : e.res_post);
    const int ld0 = has_p ? e.ld_res_pre : has_m ? e.ld_mask : e.ld_res_post;
    const T *s1 = (const T *)(has_p && has_m ? e.mask : e.res_post);
    const int ld1 = has_p && has_m ? e.ld_mask : e.ld_res_post;
    constexpr int nops = NOPS;
    const bool m_in1 = has_p && has_m, q_in1 = has_q && nops == 2;
    // operands: one -> ra = rows 0..63, rb = rows 64..127, both up front; two / three -> slots of the current 64 rows
    uint4 ra[8], rb[8], rc[NOPS == 3 ? 8 : 1];
    auto load64 = [&](const T *src, int ld, int hb, uint4 (&r)[8]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = rows_rsrc(src + (size_t)mwu * ld);
        const uint32_t vo = (uint32_t)(lrow * ld + c0) * 2u;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const u32x4_v q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (hb * 8 + ps) * 16 * ld, 0);
            r[ps] = make_uint4(q[0], q[1], q[2], q[3]);
        }
    };
    float bs1[8], bs2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs1[q] = 0.f; bs2[q] = 0.f; }
    if (nops >= 1) load64(s0, ld0, 0, ra);
    if (nops >= 2) load64(s1, ld1, 0, rb);
    if (nops == 1) load64(s0, ld0, 1, rb);
    if constexpr (NOPS == 3) load64((const T *)e.res_post, e.ld_res_post, 0, rc);
    const f32x2_v asc2[4] = {{kc.ascale[0], kc.ascale[1]}, {kc.ascale[2], kc.ascale[3]}, {kc.ascale[4], kc.ascale[5]}, {kc.ascale[6], kc.ascale[7]}};
    const f32x2_v ash2[4] = {{kc.ashift[0], kc.ashift[1]}, {kc.ashift[2], kc.ashift[3]}, {kc.ashift[4], kc.ashift[5]}, {kc.ashift[6], kc.ashift[7]}};

    auto write_tile = [&](int i) __attribute__((always_inline)) {        // 16 rows x 64 channels, 8-B chunk c of row r at c ^ r
        char *pt = patch + (i & 1) * 2048;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint2 *)(pt + frow * 128 + (((j * 4 + fq) ^ frow) << 3)) = acc[i][j];
    };
    auto read_tile = [&](int i, uint4 (&r)[2]) __attribute__((always_inline)) {   // two passes of 8 rows x 128 B, 16 B per lane
        const char *pt = patch + (i & 1) * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 8 + lrow;
            const uint2 lo = *(const uint2 *)(pt + row * 128 + ((c2 ^ row) << 3));
            const uint2 hi = *(const uint2 *)(pt + row * 128 + (((c2 + 1) ^ row) << 3));
            r[h] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    auto lds_order = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint4 cur[2], nxt[2];
    write_tile(0);
    write_tile(1);
    lds_order();
    read_tile(0, cur);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int hb = i >> 2;
        if ((i & 3) == 0 && hb == 1) {
            if (nops >= 2) { load64(s0, ld0, 1, ra); load64(s1, ld1, 1, rb); }
            if constexpr (NOPS == 3) load64((const T *)e.res_post, e.ld_res_post, 1, rc);
        }
        lds_order();
        if (i + 2 < 8) write_tile(i + 2);      // into the patch half tile i was read from (queued behind that read)
        lds_order();
        if (i + 1 < 8) read_tile(i + 1, nxt);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ps = 2 * (i & 3) + h;    // 8-row group inside the 64 rows
            const uint4 rawv = cur[h];
            float v[8], t[8];
            ld8((const bf16_t *)&rawv, v);
            const uint4 o0 = (hb == 1 && nops == 1) ? rb[ps] : ra[ps], o1 = rb[ps];
            if (has_p) {
                ld8((const T *)&o0, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += t[q];
            }
            if (has_m) {
                const uint4 om = m_in1 ? o1 : o0;
                ld8((const T *)&om, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = t[q] > 0.f ? v[q] * kc.mscale[q] : 0.f;
                if constexpr (SUMS) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { bs1[q] += v[q]; bs2[q] = fmaf(v[q], t[q], bs2[q]); }
                }
            }
            if (has_q) {
                const uint4 oq = NOPS == 3 ? rc[NOPS == 3 ? ps : 0] : (q_in1 ? o1 : o0);
                ld8((const T *)&oq, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += t[q];
            }
            const int srow = (hb * 8 + ps) * 16;       // x ld = byte offset of row (hb * 8 + ps) * 8 of the wave's 128
            if constexpr ((OUTS & 1) != 0) {
                const u32x4_v o = nops == 0 ? (u32x4_v){rawv.x, rawv.y, rawv.z, rawv.w}
                                            : (u32x4_v){pack_bf16x2_v(v[0], v[1]), pack_bf16x2_v(v[2], v[3]), pack_bf16x2_v(v[4], v[5]), pack_bf16x2_v(v[6], v[7])};
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_raw, vo_raw + (uint32_t)(srow * e.ld_raw), 0, 0);
            }
            if constexpr ((OUTS & 2) != 0) {
                const u32x4_v o = {act_pair(v[0], v[1], asc2[0], ash2[0], relu_i16), act_pair(v[2], v[3], asc2[1], ash2[1], relu_i16),
                                   act_pair(v[4], v[5], asc2[2], ash2[2], relu_i16), act_pair(v[6], v[7], asc2[3], ash2[3], relu_i16)};
#ifdef KDCC_TUNING
                if (p.tune & 128) { asm volatile("" ::"v"(o)); continue; }      // timing ablation: everything but the store itself
                if (p.tune & 256) { *(u32x4_v *)((T *)e.out_act + (size_t)((((mw + (hb * 8 + ps) * 8 + lrow) & 15) + 16 * (blockIdx.x * 8 + (threadIdx.x >> 6))) % 4096) * e.ld_act + c0) = o; continue; }
#endif
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_act, vo_act + (uint32_t)(srow * e.ld_act), 0, 0);
            }
        }
        cur[0] = nxt[0];
        cur[1] = nxt[1];
    }
    lds_order();
    if constexpr (SUMS) {
        // (ig_epilogue_rows16: lanes l, l + 8, .., l + 56 hold the same 8 channels; fixed butterfly order; partial row mw / 128)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int sft = 8; sft < 64; sft <<= 1) {
                bs1[q] += __shfl_xor(bs1[q], sft, 64);
                bs2[q] += __shfl_xor(bs2[q], sft, 64);
            }
        }
        if (lane < 8) {
            float *dst = e.bn_sums + (size_t)(mw >> 7) * 2 * p.Cout + c0;
            *(float4 *)dst = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
            *(float4 *)(dst + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
            *(float4 *)(dst + p.Cout) = make_float4(bs2[0], bs2[1], bs2[2], bs2[3]);
            *(float4 *)(dst + p.Cout + 4) = make_float4(bs2[4], bs2[5], bs2[6], bs2[7]);
        }
    }
}

// ---- the decoder's classifier in the epilogue of its last 3x3 (round 6; conv_row_lw_kernel<16>) ------------------------------------
// final[3] -> BN -> ReLU -> final[6] (a 1x1 onto the classes; models/deeplabv3/deeplabv3.py:127-139 of the reference): the 256-channel
// activation between them is written and read back for nothing else when no gradient flows through it (the frozen teacher; the student
// when the loss is the hints alone).  Cout == 256 is ONE N tile, so a workgroup holds every channel of its 256 pixels: the classifier
// runs on the accumulators instead.  Per 16-pixel tile and 64-channel half the packed raw accumulators pass through the wave's LDS patch
// exactly as in lw_epilogue_rows16 (same bf16 rounding of the raw value, same act_pair arithmetic: the activation that feeds the MFMAs is
// bit for bit the tensor the two-launch form stores), are read back as MFMA B fragments (lane = pixel row, 8 consecutive channels), and
// multiply the classifier rows (A fragments, loaded once per tile from the [32][Cout] bf16 matrix, rows >= ncls zero): D[class][pixel]
// in fp32.  The two waves that own the channel halves of a pixel row exchange their partial logits through their patches (two pixel
// tiles = 4 KiB per round, fixed order: low half + high half) and the low-half wave stores.  Differences to the two-launch form: the
// fp32 summation order over the 256 channels only.
struct LwClsState {
    uint4 w[2][2][2];        // [64-channel half][32-channel k-step][class tile]
    float sc[2][2][8], sh[2][2][8];   // BN scale / shift of the channels this lane feeds to the MFMAs: [half][k-step][8 channels]
    f32x4_t acc[8][2];       // [pixel tile][class tile]
};
__device__ __forceinline__ void lw_cls_begin(const ConvParams &p, int nw, int lane, LwClsState &s)
{
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, kq = lane >> 4;
    const bf16_t *w = (const bf16_t *)p.ep.cls_w;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                s.w[h][kc][ct] = *(const uint4 *)(w + (size_t)(ct * 16 + li) * p.Cout + nw + h * 64 + kc * 32 + kq * 8);
    // (loaded here, ahead of the accumulator read-out: a lone wave has nobody to hide their latency behind)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { s.sc[h][kc][q] = 1.f; s.sh[h][kc][q] = 0.f; }
            if (p.ep.act_scale) ld8(p.ep.act_scale + nw + h * 64 + kc * 32 + kq * 8, s.sc[h][kc]);
            if (p.ep.act_shift) ld8(p.ep.act_shift + nw + h * 64 + kc * 32 + kq * 8, s.sh[h][kc]);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) s.acc[i][ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
}
// one 64-channel half (channels nwh .. nwh + 63 of the tile) of the wave's 128 pixels
__device__ __forceinline__ void lw_epilogue_cls16(const ConvParams &p, char *patch, uint2 (&acc)[8][4], int lane, const uint4 (&w)[2][2],
                                                  const float (&sc)[2][8], const float (&sh)[2][8], f32x4_t (&cacc)[8][2])
{
    typedef bf16_t T;
    const kd_conv_epilogue &e = p.ep;
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
    const short relu_w = e.act_relu ? (short)0 : (short)-32768;
    const i16x2_v relu_i16 = {relu_w, relu_w};
    auto write_tile = [&](int i) __attribute__((always_inline)) {        // 16 rows x 64 channels, 8-B chunk c of row r at c ^ r (lw_epilogue_rows16's image)
        char *pt = patch + (i & 1) * 2048;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint2 *)(pt + frow * 128 + (((j * 4 + fq) ^ frow) << 3)) = acc[i][j];
    };
    auto lds_order = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    write_tile(0);
    write_tile(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lds_order();
        const char *pt = patch + (i & 1) * 2048;
        uint4 raw[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {      // row frow, channels kc * 32 + fq * 8 .. + 7 = 8-B chunks c, c + 1
            const int c = kc * 8 + fq * 2;
            const uint2 lo = *(const uint2 *)(pt + frow * 128 + ((c ^ frow) << 3));
            const uint2 hi = *(const uint2 *)(pt + frow * 128 + (((c + 1) ^ frow) << 3));
            raw[kc] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        lds_order();
        if (i + 2 < 8) write_tile(i + 2);      // into the patch half tile i was read from (queued behind that read)
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4c_t;
        u32x4c_t fb[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            float v[8];
            ld8((const T *)&raw[kc], v);
            const f32x2_v s0 = {sc[kc][0], sc[kc][1]}, s1 = {sc[kc][2], sc[kc][3]}, s2 = {sc[kc][4], sc[kc][5]}, s3 = {sc[kc][6], sc[kc][7]};
            const f32x2_v h0 = {sh[kc][0], sh[kc][1]}, h1 = {sh[kc][2], sh[kc][3]}, h2 = {sh[kc][4], sh[kc][5]}, h3 = {sh[kc][6], sh[kc][7]};
            fb[kc] = (u32x4c_t){act_pair(v[0], v[1], s0, h0, relu_i16), act_pair(v[2], v[3], s1, h1, relu_i16),
                                act_pair(v[4], v[5], s2, h2, relu_i16), act_pair(v[6], v[7], s3, h3, relu_i16)};
        }
        // The tile's four MFMAs as ONE asm statement on vector registers: an intrinsic's accumulator would be placed in the accumulation file,
        // which the tile statement owns; and, being asm, the statement has to bring its own wait states -- hipcc does not know that what it
        // reads next (copies of the sums, the LDS stores of lw_cls_finish) comes out of the matrix pipe (the first version, one MFMA per
        // statement and no s_nop, returned garbage in 0.3 % of the logits)
        const u32x4c_t w00 = {w[0][0].x, w[0][0].y, w[0][0].z, w[0][0].w}, w01 = {w[0][1].x, w[0][1].y, w[0][1].z, w[0][1].w},
                       w10 = {w[1][0].x, w[1][0].y, w[1][0].z, w[1][0].w}, w11 = {w[1][1].x, w[1][1].y, w[1][1].z, w[1][1].w};
        asm volatile("s_nop 3\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %2, %6, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %1, %3, %6, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %1, %5, %7, %1\n\t"
                     "s_nop 15\n\ts_nop 3"
                     : "+v"(cacc[i][0]), "+v"(cacc[i][1])
                     : "v"(w00), "v"(w01), "v"(w10), "v"(w11), "v"(fb[0]), "v"(fb[1]));
    }
    lds_order();
}
// the waves (wm, 0) and (wm, 1) hold the two channel halves of pixels mw .. mw + 127: low + high through the patches, (wm, 0) stores
__device__ __forceinline__ void lw_cls_finish(const ConvParams &p, char *patches, int wv, int mw, int lane, f32x4_t (&cacc)[8][2])
{
    const kd_conv_epilogue &e = p.ep;
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, q = lane >> 4, wn = wv & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (wn == 1) {
            char *own = patches + wv * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) *(f32x4_t *)(own + ((t * 2 + ct) * 64 + lane) * 16) = cacc[2 * r + t][ct];
        }
        __syncthreads();
        if (wn == 0) {
            const char *other = patches + (wv + 1) * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float *dst = e.cls_out + (size_t)(mw + (2 * r + t) * 16 + li) * e.ld_cls;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const f32x4_t hi = *(const f32x4_t *)(other + ((t * 2 + ct) * 64 + lane) * 16);
                    const f32x4_t v = cacc[2 * r + t][ct] + hi;
                    const int c0 = ct * 16 + 4 * q;
                    typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
                    if (c0 + 4 <= e.ncls) {
                        *(f32x4u_t *)(dst + c0) = v;      // one 16-B store (dword-aligned: 19 classes = 76 B per pixel); four dword stores made the epilogue store-issue-bound
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (c0 + k < e.ncls) dst[c0 + k] = v[k];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// the tiles of this workgroup: XCD x owns a contiguous range of tile ids (as xcd_remap deals them), its workgroups take
// them round-robin, so the tiles in flight on one XCD at any time are neighbours (shared image rows / weight slabs in L2)
// A/B experiment (tuning build, KDCC_CONV_TUNE & 16384): every second workgroup of an XCD starts p.stagger_us microseconds late, so
// that the store phases (epilogues) of one half of the chip fall into the main loops of the other half.
__device__ __forceinline__ void stagger_start(const ConvParams &p)
{
    if ((p.tune & 16384) && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_us * 100ull) __builtin_amdgcn_s_sleep(8);
    }
}

struct TileWalk {
    int t, t_end, step;
    __device__ __forceinline__ TileWalk(int nt)
    {
        const int xcd = blockIdx.x & 7, q = nt >> 3, r = nt & 7;
        t = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_end = t + q + (xcd < r ? 1 : 0);
        t += blockIdx.x >> 3;
        step = gridDim.x >> 3;
    }
};

}  // namespace

// conv_lw.hip: the one-wave-per-SIMD row kernel (128 x 128 wave tiles, hand-scheduled main loop).  nops_sums = NOPS | 4 when the
// eval-BN sums are taken.  Returns false when the instantiation does not exist.
bool kd_launch_conv_row_lw(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s);
bool kd_launch_conv_row_tall(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s);   // 512 x 128 tiles (Cout % 128 == 0, Cin % 64 == 0, W % 512 == 0)
int kd_lw_tlog_copy(unsigned long long *dst, size_t bytes);   // conv_row_duo_kernel's per-tile stamps (KDCC_CONV_TUNE & 1024)
bool kd_launch_conv_row_duo(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s);   // 256 x 128 tiles, two workgroups per CU
bool kd_launch_conv_pw_lw(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s);   // 1x1 / stride 1, Cin % 128 == 0

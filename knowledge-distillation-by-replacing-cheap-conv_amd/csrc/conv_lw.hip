// conv_row_lw_kernel: the 3x3 / stride 1 / 'same' row-buffer convolution (kd_conv2d_fwd; forward and input gradient of
// reference models/encoders/wider_resnet.py:124-167) with ONE wave per SIMD and a hand-scheduled main loop.
//
// Why: conv_row_persist_kernel<pp> (8 waves of 128 x 64, two per SIMD) reads 0.0234 B of LDS fragments per FLOP, spends one of
// its two waves per SIMD in the hand-over (LDS-DMA issue, fragment reads) while the other multiplies, and ends at 0.65 MFMA-busy.
// A 128 x 128 wave tile reads a third less per FLOP but needs 256 accumulator registers, i.e. the 512-register file of a lone
// wave -- and a lone wave has nobody to hide its reads and its DMA issue behind, which hipcc's schedule never overlaps with the
// MFMAs (r03: 430 TFLOP/s).  Here the loop is written by hand (tools/gen_conv_lw.py -> conv_lw_body.inc): accumulators in
// a[0:255] for the whole kernel, fragments in v[128:255], every ds_read and every 1-KiB LDS-DMA piece dealt into the shadow of
// the MFMAs (tools/ubench/lone_wave: 1128 cycles per 64-MFMA k-step against 1117 for the bare MFMAs; the ping-pong kernel's
// stage is 2430 cycles per 128), ONE barrier per k-step, B in four 16-KiB slots of 32 channels of K so that a slot has >= 2
// k-steps to land, the row buffer of period P + 1 issued under period P.  The staging runs ahead ACROSS tiles (persistent
// workgroups, conv_common.h TileWalk), so a tile boundary is the epilogue between two periods and nothing else.  All periods of a
// tile are ONE asm statement (LW_TILE_ASM: the fragment registers never live across compiler-generated code; the accumulators
// do, tools/check_lw_asm.py audits that hipcc never touches them).
//
// Work split: tile 256 pixels (a segment of one image row) x 256 output channels; wave (wm, wn) = (wv >> 1, wv & 1) owns pixels
// wm * 128 .. + 127, channels wn * 128 .. + 127.  A "period" = one (64-channel block, kernel row inside the image) = three taps =
// six k-steps.  LDS: two row buffers of 320 rows x 128 B (pixel x0 - dil + r, swizzled 16-B chunks; dil <= 32), four B slots of
// 256 rows x 64 B, four 4-KiB epilogue patches = 160 KiB.
// Results are bit-identical to conv_row_persist_kernel: the same k order into the same fp32 accumulation chains, the same epilogue.
#include "conv_common.h"
#include "conv_lw_body.inc"

namespace {

__device__ __attribute__((aligned(256))) uint32_t lw_zero_page[64];   // zero-initialised
__device__ unsigned long long lw_tlog[512 * 32 * 8];   // KDCC_CONV_TUNE & 1024 (tuning build): per workgroup and tile, 100-MHz stamps: tile start / main loop end / epilogue end / tile id; conv_row_tall_kernel also [4..7]: accumulators read (half 0) / staged loads landed / half 0 stored / accumulators read (half 1)

constexpr int LW_ABUF = 320 * 128, LW_BSLOT = 256 * 64, LW_NEED = 2 * LW_ABUF + 4 * LW_BSLOT;

template <int NOPS_>
__global__ __launch_bounds__(256, 1) void conv_row_lw_kernel(const ConvParams p)
{
    typedef bf16_t T;
    typedef unsigned long long u64;
    __shared__ __attribute__((aligned(1024))) char lds[LW_NEED + 4 * 4096];   // 160 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int d = p.dil;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    // ---- per-lane addresses handed to the asm statements.  They are loop-invariant, but 27 registers that stay live through the
    // epilogue push it into spills (and hipcc then parks values in the accumulator file: tools/check_lw_asm.py): they are
    // recomputed per tile from a laundered copy of the lane id instead.
    // fragment reads: A row (wm * 128 + 16 i + frow + kx * dil) of the row buffer, 16-B chunk (fq + 4 ks) at slot chunk ^ (row & 7);
    // B row (wn * 128 + 16 j + frow) of a slot, chunk fq at slot fq ^ ((row >> 1) & 3); i / j / buffer / slot are instruction offsets.
    // LDS-DMA sources (byte offsets from a wave-uniform base; the swizzle is applied on the source side): this wave's ten 8-row
    // pieces of a row buffer, its four 16-row pieces of a B slot.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    uint32_t va[6], vb, voa[10], vob[4], vz0, vz1, vr0;
    u32x4_t vzero;
    auto lane_addresses = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int frow = l & 15, fq = l >> 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int rsh = frow + kx * d;
                va[kx * 2 + ks] = lbase + (wm * 128 + rsh) * 128 + (((fq + 4 * ks) ^ (rsh & 7)) << 4);
            }
        vb = lbase + 2 * LW_ABUF + (wn * 128 + frow) * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);
#pragma unroll
        for (int j = 0; j < 10; ++j) voa[j] = (uint32_t)(((wv * 10 + j) * 8 + (l >> 3)) * (p.ldx * 2) + (((l & 7) ^ (l >> 3)) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) vob[j] = (uint32_t)(((wv * 4 + j) * 16 + (l >> 2)) * (p.Ktot * 2) + (((l & 3) ^ ((l >> 3) & 3)) << 4));
        vz0 = lbase + wv * 10240 + l * 16;
        vz1 = vz0 + LW_ABUF;
        vr0 = (uint32_t)(wv * 80 + (l >> 3));   // buffer row of this lane in piece 0 (piece j: + 8 j)
        vzero = (u32x4_t){0u, 0u, 0u, 0u};
    };
    lane_addresses();
    const uint32_t sldsA = __builtin_amdgcn_readfirstlane(lbase + wv * 10240);
    const uint32_t sldsB = __builtin_amdgcn_readfirstlane(lbase + 2 * LW_ABUF + wv * 4096);
    const uint32_t s2c = (uint32_t)(2 * p.Cin - 64);
    const long long dWl2 = 2ll * d * p.W * p.ldx;   // bytes from one kernel row's image row to the next

    // ---- tiles -----------------------------------------------------------------------------------------------------------------
    struct Tile {
        int m0, n0, kylo, nky;
        uint32_t lo, span;   // buffer rows [lo, lo + span) hold pixels of the image row (and are read)
        u64 abase, bbase;    // (n, ho, x0 - dil) of the input / the tile's first weight row
    };
    auto decode = [&](int tile, Tile &t) {
        int tn, tm;
        if (p.tn_group > 0) {
            const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
            tm = r / p.tn_group;
            tn = blk * p.tn_group + (r - tm * p.tn_group);
        } else {
            tn = tile % p.tiles_n;
            tm = tile / p.tiles_n;
        }
        t.m0 = tm * 256;
        t.n0 = tn * 256;
        const int n = t.m0 / p.HoWo, rem = t.m0 - n * p.HoWo;
        const int ho = rem / p.W, x0 = rem - ho * p.W;
        t.kylo = ho - d < 0 ? 1 : 0;
        t.nky = (ho + d >= p.H ? 1 : 2) - t.kylo + 1;
        // pixel x0 - d + r: outside the row for r < d in the row's first tile and for r >= 256 + d in its last
        t.lo = x0 == 0 ? (uint32_t)d : 0u;
        t.span = (x0 + 256 == p.W ? 256u + d : 320u) - t.lo;
        t.abase = (u64)p.x + (u64)(2ll * ((long long)((n * p.H + ho) * p.W + (x0 - d)) * p.ldx));
#ifdef KDCC_TUNING
        // timing ablation 2048 (tools/power_sweep.py): every tile stages image 0's rows 0 / d / 2d -- the input stays in the XCD's L2
        if (p.tune & 2048) t.abase = (u64)p.x + (u64)(2ll * ((long long)(d * p.W + (x0 - d)) * p.ldx));
#endif
        t.bbase = (u64)p.w + (u64)(2ll * (long long)t.n0 * p.Ktot);
    };
    // period q of a tile = (channel block q / nky, kernel row kylo + q % nky)
    auto a_of = [&](const Tile &t, int cb, int kyi) { return t.abase + (u64)((long long)(t.kylo + kyi - 1) * dWl2 + cb * 128); };
    auto b_of = [&](const Tile &t, int cb, int kyi) { return t.bbase + (u64)(2ll * ((long long)(t.kylo + kyi) * 3 * p.Cin + cb * 64)); };

    Tile cur, nxt;
    int c_tile = walk.t;
    decode(c_tile, cur);
#ifdef KDCC_TUNING
    // 1024 (tools/power_sweep.py): shader-clock and 100-MHz stamps around the whole workgroup -> the clock this kernel ran at
    if ((p.tune & 1024) && tid == 0 && blockIdx.x < 512) {
        lw_tlog[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
        lw_tlog[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // ---- prologue: row buffer of period 0 and the B slots of its k-steps 0 .. 3 (generic pieces: 64-bit lane addresses, lanes
    // outside the image read the zero page) ---------------------------------------------------------------------------------------
    {
        const char *ab = (const char *)a_of(cur, 0, 0);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint32_t r = vr0 + 8 * j - cur.lo;
            glds16(r < cur.span ? (const void *)(ab + voa[j]) : (const void *)lw_zero_page, lds + (wv * 10 + j) * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char *bb = (const char *)b_of(cur, 0, 0) + 2ll * (q >> 1) * p.Cin + (q & 1) * 64;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(bb + vob[j], lds + 2 * LW_ABUF + q * LW_BSLOT + (wv * 4 + j) * 1024);
        }
    }
    u64 sBp = b_of(cur, 0, 0) + (u64)(4ll * p.Cin);   // k-step 4 of period 0 = tap kx 2, first k-half
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // The accumulator file belongs to the asm statements (they name a[0:255] literally).  hipcc must know it is occupied, or it
    // parks its own values there under register pressure (seen in the three-operand instantiation: v_accvgpr_write a0 .. a21 in
    // the epilogue = silent corruption): eight 32-register values constrained to AGPRs are outputs of the zeroing statement and
    // read-write operands of every statement that touches an accumulator, so all 256 registers are live for the whole kernel.
    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;
    asm volatile(LW_ZERO_ACC_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define LW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)

    uint32_t flag = 0, par = 0;
#pragma unroll 1
    for (;;) {
        // the first k-step's fragments (the epilogue below runs in the fragment registers), then every period of the tile
        lane_addresses();
        asm volatile(LW_REFILL_ASM : : [vaf] "v"(va[0] + par * LW_ABUF), [vbf] "v"(vb + par * 2 * LW_BSLOT) : "memory", LW_CLOBBER_FRAG);
        const int n_tile = c_tile + walk.step;
        const bool more = n_tile < walk.t_end;
        if (more) decode(n_tile, nxt);
        else nxt = cur;                       // nothing left to stage: the last body re-stages this tile's period 0 (valid memory, unread)
        const uint32_t nper = (uint32_t)(p.nkc * cur.nky);
        const u64 sAn = a_of(cur, 0, 1), sBn = b_of(cur, 0, 1), sAnT = a_of(nxt, 0, 0), sBnT = b_of(nxt, 0, 0);
        const int dAs = (int)dWl2, dAw = (int)(128 - (cur.nky - 1) * dWl2), dBs = 6 * p.Cin, dBw = 128 - (cur.nky - 1) * 6 * p.Cin;
#define LW_TILE_OPERANDS \
: [sBp] "+s"(sBp), LW_ACC_RW \
                     : [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]), [va4] "v"(va[4]), [va5] "v"(va[5]), [vb] "v"(vb), \
                       [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [voa5] "v"(voa[5]), \
                       [voa6] "v"(voa[6]), [voa7] "v"(voa[7]), [voa8] "v"(voa[8]), [voa9] "v"(voa[9]), [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), \
                       [vob2] "v"(vob[2]), [vob3] "v"(vob[3]), [vz0] "v"(vz0), [vz1] "v"(vz1), [vzero] "v"(vzero), [vr0] "v"(vr0), \
                       [sAn] "s"(sAn), [sBn] "s"(sBn), [sAnT] "s"(sAnT), [sBnT] "s"(sBnT), [slo] "s"(cur.lo), [ssp] "s"(cur.span), \
                       [sloT] "s"(nxt.lo), [sspT] "s"(nxt.span), [sdAs] "s"(dAs), [sdAw] "s"(dAw), [sdBs] "s"(dBs), [sdBw] "s"(dBw), \
                       [snky] "s"((uint32_t)cur.nky), [snper] "s"(nper), [s2c] "s"(s2c), [sflag] "s"(flag), [spar] "s"(par), \
                       [sldsA] "s"(sldsA), [sldsB] "s"(sldsB)
#ifdef KDCC_TUNING
        // KDCC_CONV_TUNE & 32768: the deliberately broken schedule (one barrier removed, wave 0 delayed: tools/gen_conv_lw.py BROKEN) --
        // the defect tests/test_lw_bitwise_gpu.py's A/B has to find; results are wrong by construction
        // (the diagnostics build always runs the variant; brk = 0 takes the regular path through it)
        const uint32_t brk = (p.tune & 32768) ? 1u : 0u;
        asm volatile(LW_TILE_BROKEN_ASM LW_TILE_OPERANDS, [swv] "s"((uint32_t)wv), [sbrk] "s"(brk) : "memory", "scc", "vcc", LW_CLOBBER_S, LW_CLOBBER_FRAG);
#else
        asm volatile(LW_TILE_ASM LW_TILE_OPERANDS : "memory", "scc", "vcc", LW_CLOBBER_S, LW_CLOBBER_FRAG);
#endif
#undef LW_TILE_OPERANDS
        par = (par + nper) & 1u;
        // ---- the tile is complete: accumulators -> memory (conv_common.h ig_epilogue_rows16, 128 x 64 at a time) ----------------
        asm volatile("s_nop 15\n\ts_nop 15" : LW_ACC_RW : : "memory");   // the last MFMAs' results have reached the accumulator file
        if (!(p.tune & 64)) {   // (64: timing ablation without the epilogue, tuning build only)
        const int mw = cur.m0 + wm * 128, nw = cur.n0 + wn * 128;
        char *patch = lds + LW_NEED + wv * 4096;
#define LW_RD(I, JG)                                                                                                                  \
    {                                                                                                                                 \
        float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;                                                   \
        asm volatile(LW_READ_ACC_##I##_##JG##_ASM                                                                                     \
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7), "=v"(t8), "=v"(t9), "=v"(t10), \
                       "=v"(t11), "=v"(t12), "=v"(t13), "=v"(t14), "=v"(t15), LW_ACC_RW);                                             \
        acc[I][0] = make_uint2(pack_bf16x2_v(t0, t1), pack_bf16x2_v(t2, t3));      /* rounded to the stored type at once: half the registers */ \
        acc[I][1] = make_uint2(pack_bf16x2_v(t4, t5), pack_bf16x2_v(t6, t7));                                                             \
        acc[I][2] = make_uint2(pack_bf16x2_v(t8, t9), pack_bf16x2_v(t10, t11));                                                           \
        acc[I][3] = make_uint2(pack_bf16x2_v(t12, t13), pack_bf16x2_v(t14, t15));                                                         \
    }
        if constexpr (NOPS_ == 16) {     // the classifier on the accumulators instead of an activation store (conv_common.h lw_epilogue_cls16)
            LwClsState cs;
            lw_cls_begin(p, nw, lane, cs);
            {
                uint2 acc[8][4];
                LW_RD(0, 0) LW_RD(1, 0) LW_RD(2, 0) LW_RD(3, 0) LW_RD(4, 0) LW_RD(5, 0) LW_RD(6, 0) LW_RD(7, 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (as below: nothing older than the stores may be pending)
                lw_epilogue_cls16(p, patch, acc, lane, cs.w[0], cs.sc[0], cs.sh[0], cs.acc);
            }
            {
                uint2 acc[8][4];
                LW_RD(0, 1) LW_RD(1, 1) LW_RD(2, 1) LW_RD(3, 1) LW_RD(4, 1) LW_RD(5, 1) LW_RD(6, 1) LW_RD(7, 1)
                lw_epilogue_cls16(p, patch, acc, lane, cs.w[1], cs.sc[1], cs.sh[1], cs.acc);
            }
            lw_cls_finish(p, lds + LW_NEED, wv, mw, lane, cs.acc);
        } else {
        LwEpiConsts k0, k1;
        lw_epilogue_consts<NOPS_>(p, nw, lane, k0);      // (their latency passes under the accumulator read-out)
        {
            uint2 acc[8][4];
            LW_RD(0, 0) LW_RD(1, 0) LW_RD(2, 0) LW_RD(3, 0) LW_RD(4, 0) LW_RD(5, 0) LW_RD(6, 0) LW_RD(7, 0)
            // what ran ahead into the next tile's buffers has landed before the first store is issued: the next tile's first two
            // waits leave every store outstanding (vmcnt is one in-order counter), which is only sound if nothing older is pending
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lw_epilogue_consts<NOPS_>(p, nw + 64, lane, k1);
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw, lane, k0);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw, lane, k0);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw, lane, k0);
        }
        {
            uint2 acc[8][4];
            LW_RD(0, 1) LW_RD(1, 1) LW_RD(2, 1) LW_RD(3, 1) LW_RD(4, 1) LW_RD(5, 1) LW_RD(6, 1) LW_RD(7, 1)
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw + 64, lane, k1);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw + 64, lane, k1);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw + 64, lane, k1);
        }
        }
#undef LW_RD
        }
        if (!more) break;
        c_tile = n_tile;
        cur = nxt;
        flag = 1;
    }
#undef LW_ACC_RW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staging ran ahead of the last tile: nothing may land after the wave ends
#ifdef KDCC_TUNING
    if ((p.tune & 1024) && tid == 0 && blockIdx.x < 512) {
        lw_tlog[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
        lw_tlog[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


// ---- the same loop for 1x1 / stride 1 convolutions (kd_conv2d_fwd: the pointwise layers and their input gradients) -----------------
// No row buffer: the A operand (256 pixels x 32 channels per k-step) is staged per k-step like B, four 16-KiB slots each, 8 DMA
// pieces per wave and k-step (conv_igemm_persist_kernel<pp> stages 64 pieces per 64-deep stage through waves that also have to
// multiply; it sits at 0.48 of peak).  Cin % 128 == 0 (whole passes over the four slots).
template <int NOPS_>
__global__ __launch_bounds__(256, 1) void conv_pw_lw_kernel(const ConvParams p)
{
    typedef unsigned long long u64;
    constexpr int SLOT = LW_BSLOT, NEED = 8 * SLOT;
    __shared__ __attribute__((aligned(1024))) char lds[NEED + 4 * 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    uint32_t va, vb, voa[4], vob[4];
    auto lane_addresses = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int frow = l & 15, fq = l >> 4, sw = (fq ^ ((frow >> 1) & 3)) << 4;
        va = lbase + (wm * 128 + frow) * 64 + sw;
        vb = lbase + 4 * SLOT + (wn * 128 + frow) * 64 + sw;
        const int srow = l >> 2, chunk = ((l & 3) ^ ((l >> 3) & 3)) << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            voa[j] = (uint32_t)(((wv * 4 + j) * 16 + srow) * (p.ldx * 2) + chunk);
            vob[j] = (uint32_t)(((wv * 4 + j) * 16 + srow) * (p.Ktot * 2) + chunk);
        }
    };
    lane_addresses();
    const uint32_t sldsA = __builtin_amdgcn_readfirstlane(lbase + wv * 4096);
    const uint32_t sldsB = __builtin_amdgcn_readfirstlane(lbase + 4 * SLOT + wv * 4096);
    auto bases = [&](int tile, int &m0, int &n0, u64 &ab, u64 &bb) {
        int tn, tm;
        if (p.tn_group > 0) {
            const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
            tm = r / p.tn_group;
            tn = blk * p.tn_group + (r - tm * p.tn_group);
        } else {
            tn = tile % p.tiles_n;
            tm = tile / p.tiles_n;
        }
        m0 = tm * 256;
        n0 = tn * 256;
        ab = (u64)p.x + (u64)(2ll * (long long)m0 * p.ldx);
        bb = (u64)p.w + (u64)(2ll * (long long)n0 * p.Ktot);
    };
    int c_tile = walk.t, m0, n0, m0n, n0n;
    u64 ab, bb, abn, bbn;
    bases(c_tile, m0, n0, ab, bb);
    // prologue: k-steps 0 .. 3 of the first tile
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16((const char *)ab + q * 64 + voa[j], lds + q * SLOT + (wv * 4 + j) * 1024);
            glds16((const char *)bb + q * 64 + vob[j], lds + (4 + q) * SLOT + (wv * 4 + j) * 1024);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;   // the accumulator file is occupied (see conv_row_lw_kernel)
    asm volatile(LW_ZERO_ACC_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define LW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)
    const uint32_t nit = (uint32_t)(p.nk / 2);   // passes of four 32-channel k-steps (p.nk counts 64-channel stages)
    uint32_t flag = 0;
#pragma unroll 1
    for (;;) {
        lane_addresses();
        asm volatile(LW1_REFILL_ASM : : [va] "v"(va), [vb] "v"(vb) : "memory", LW_CLOBBER_FRAG);
        const int n_tile = c_tile + walk.step;
        const bool more = n_tile < walk.t_end;
        if (more) bases(n_tile, m0n, n0n, abn, bbn);
        else { abn = ab; bbn = bb; }             // nothing left to stage: re-stage this tile's first k-steps (valid memory, unread)
        const u64 sA = ab + 256, sB = bb + 256;
        asm volatile(LW1_TILE_ASM
                     : LW_ACC_RW
                     : [va] "v"(va), [vb] "v"(vb), [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]),
                       [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]), [sA] "s"(sA), [sB] "s"(sB),
                       [sAT] "s"(abn), [sBT] "s"(bbn), [snit] "s"(nit), [sflag] "s"(flag), [sldsA] "s"(sldsA), [sldsB] "s"(sldsB)
                     : "memory", "scc", LW_CLOBBER_S, LW_CLOBBER_FRAG);
        asm volatile("s_nop 15\n\ts_nop 15" : LW_ACC_RW : : "memory");
        if (!(p.tune & 64)) {
        const int mw = m0 + wm * 128, nw = n0 + wn * 128;
        char *patch = lds + NEED + wv * 8192;
#define LW_RD(I, JG)                                                                                                                  \
    {                                                                                                                                 \
        float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;                                                   \
        asm volatile(LW_READ_ACC_##I##_##JG##_ASM                                                                                     \
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7), "=v"(t8), "=v"(t9), "=v"(t10), \
                       "=v"(t11), "=v"(t12), "=v"(t13), "=v"(t14), "=v"(t15), LW_ACC_RW);                                             \
        acc[I][0] = make_uint2(pack_bf16x2_v(t0, t1), pack_bf16x2_v(t2, t3));                                                             \
        acc[I][1] = make_uint2(pack_bf16x2_v(t4, t5), pack_bf16x2_v(t6, t7));                                                             \
        acc[I][2] = make_uint2(pack_bf16x2_v(t8, t9), pack_bf16x2_v(t10, t11));                                                           \
        acc[I][3] = make_uint2(pack_bf16x2_v(t12, t13), pack_bf16x2_v(t14, t15));                                                         \
    }
        LwEpiConsts k0, k1;
        lw_epilogue_consts<NOPS_>(p, nw, lane, k0);      // (their latency passes under the accumulator read-out)
        {
            uint2 acc[8][4];
            LW_RD(0, 0) LW_RD(1, 0) LW_RD(2, 0) LW_RD(3, 0) LW_RD(4, 0) LW_RD(5, 0) LW_RD(6, 0) LW_RD(7, 0)
            // what ran ahead into the next tile's buffers has landed before the first store is issued: the next tile's first two
            // waits leave every store outstanding (vmcnt is one in-order counter), which is only sound if nothing older is pending
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lw_epilogue_consts<NOPS_>(p, nw + 64, lane, k1);
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw, lane, k0);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw, lane, k0);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw, lane, k0);
        }
        {
            uint2 acc[8][4];
            LW_RD(0, 1) LW_RD(1, 1) LW_RD(2, 1) LW_RD(3, 1) LW_RD(4, 1) LW_RD(5, 1) LW_RD(6, 1) LW_RD(7, 1)
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw + 64, lane, k1);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw + 64, lane, k1);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw + 64, lane, k1);
        }
#undef LW_RD
        }
        if (!more) break;
        c_tile = n_tile;
        m0 = m0n; n0 = n0n; ab = abn; bb = bbn;
        flag = 1;
    }
#undef LW_ACC_RW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- two workgroups per CU: the epilogue of one under the main loop of the other ---------------------------------------------------------
// conv_row_duo_kernel: the same hand-dealt stream on a 256 x 128 tile -- 4 waves of 128 x 64, 128 accumulators in a[0:127], 256 registers
// per wave, 72 KiB of LDS -- so that two independent workgroups are resident per CU (tools/gen_conv_lw.py, "duo").  K is staged in
// 32-channel periods (64-B row buffers / B slots, one slot per tap): the k order is (32-channel block, kernel row, tap) where the other
// row kernels walk (64-channel block, kernel row, tap, k-half), so results agree to fp32 summation order, not bit for bit.
// Cout % 128 == 0, Cin % 64 == 0.
constexpr int DUO_ABUF = 320 * 64, DUO_BSLOT = 128 * 64, DUO_NEED = 2 * DUO_ABUF + 3 * DUO_BSLOT;

template <int NOPS_>
__global__ __launch_bounds__(256, 2) void conv_row_duo_kernel(const ConvParams p)
{
    typedef unsigned long long u64;
    __shared__ __attribute__((aligned(1024))) char lds[DUO_NEED + 4 * 2048];   // 72 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int d = p.dil;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    uint32_t va[3], vb, voa[5], vob[2], vz0, vz1, vr0;
    u32x4_t vzero;
    auto lane_addresses = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int frow = l & 15, fq = l >> 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int rsh = frow + kx * d;
            va[kx] = lbase + (wm * 128 + rsh) * 64 + ((fq ^ ((rsh >> 1) & 3)) << 4);
        }
        vb = lbase + 2 * DUO_ABUF + (wn * 64 + frow) * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);
        const int srow = l >> 2, chunk = ((l & 3) ^ ((l >> 3) & 3)) << 4;
#pragma unroll
        for (int j = 0; j < 5; ++j) voa[j] = (uint32_t)(((wv * 5 + j) * 16 + srow) * (p.ldx * 2) + chunk);
#pragma unroll
        for (int j = 0; j < 2; ++j) vob[j] = (uint32_t)(((wv * 2 + j) * 16 + srow) * (p.Ktot * 2) + chunk);
        vz0 = lbase + wv * 5120 + l * 16;
        vz1 = vz0 + DUO_ABUF;
        vr0 = (uint32_t)(wv * 80 + srow);      // buffer row of this lane in piece 0 (piece j: + 16 j)
        vzero = (u32x4_t){0u, 0u, 0u, 0u};
    };
    lane_addresses();
    const uint32_t sldsA = __builtin_amdgcn_readfirstlane(lbase + wv * 5120);
    const uint32_t sldsB = __builtin_amdgcn_readfirstlane(lbase + 2 * DUO_ABUF + wv * 2048);
    const uint32_t s2cin = (uint32_t)(2 * p.Cin);
    const long long dWl2 = 2ll * d * p.W * p.ldx;
    const int nkc = p.Cin / 32;

    struct Tile {
        int m0, n0, kylo, nky;
        uint32_t lo, span;
        u64 abase, bbase;
    };
    auto decode = [&](int tile, Tile &t) {
        int tn, tm;
        if (p.tn_group > 0) {
            const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
            tm = r / p.tn_group;
            tn = blk * p.tn_group + (r - tm * p.tn_group);
        } else {
            tn = tile % p.tiles_n;
            tm = tile / p.tiles_n;
        }
        t.m0 = tm * 256;
        t.n0 = tn * 128;
        const int n = t.m0 / p.HoWo, rem = t.m0 - n * p.HoWo;
        const int ho = rem / p.W, x0 = rem - ho * p.W;
        t.kylo = ho - d < 0 ? 1 : 0;
        t.nky = (ho + d >= p.H ? 1 : 2) - t.kylo + 1;
        t.lo = x0 == 0 ? (uint32_t)d : 0u;
        t.span = (x0 + 256 == p.W ? 256u + d : 320u) - t.lo;
        t.abase = (u64)p.x + (u64)(2ll * ((long long)((n * p.H + ho) * p.W + (x0 - d)) * p.ldx));
        t.bbase = (u64)p.w + (u64)(2ll * (long long)t.n0 * p.Ktot);
    };
    // period q of a tile = (32-channel block q / nky, kernel row kylo + q % nky)
    auto a_of = [&](const Tile &t, int q) { return t.abase + (u64)((long long)(t.kylo + q % t.nky - 1) * dWl2 + (q / t.nky) * 64); };
    auto b_of = [&](const Tile &t, int q) { return t.bbase + (u64)(2ll * ((long long)(t.kylo + q % t.nky) * 3 * p.Cin + (q / t.nky) * 32)); };

    Tile cur, nxt;
    int c_tile = walk.t;
    decode(c_tile, cur);
    // prologue (generic pieces): row buffer of period 0, pieces 0-2 of period 1, B of period 0's three taps
    {
        const char *a0 = (const char *)a_of(cur, 0), *a1 = (const char *)a_of(cur, 1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const bool ok = vr0 + 16 * j - cur.lo < cur.span;
            glds16(ok ? (const void *)(a0 + voa[j]) : (const void *)lw_zero_page, lds + (wv * 5 + j) * 1024);
            if (j < 3) glds16(ok ? (const void *)(a1 + voa[j]) : (const void *)lw_zero_page, lds + DUO_ABUF + (wv * 5 + j) * 1024);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                glds16((const char *)b_of(cur, 0) + 2ll * kx * p.Cin + vob[j], lds + 2 * DUO_ABUF + kx * DUO_BSLOT + (wv * 2 + j) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The two workgroups of a CU do identical work: started together they would reach their epilogues together.  The second half
    // of the grid (dealt to the CUs' second slots) starts half a tile late -- a tile takes about 2 x nper x 3 k-steps x 1117 cycles
    // while both workgroups share the matrix pipes -- once per launch; the offset then persists (equal tiles).
    if (blockIdx.x * 2 >= gridDim.x && !(p.tune & 16384)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
        const unsigned long long ticks = (unsigned long long)(nkc * cur.nky) * (unsigned long long)(p.stagger_us > 0 ? p.stagger_us : 160);
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_s_barrier();
    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3;      // the accumulator file is occupied (see conv_row_lw_kernel)
    asm volatile(DUO_ZERO_ACC_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3));
#define DUO_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3)
    uint32_t flag = 0;
    int tcount = 0;
#pragma unroll 1
    for (;;) {
        const bool tl = (p.tune & 1024) && tid == 0 && tcount < 32 && blockIdx.x < 512;
        unsigned long long *tlp = lw_tlog + ((size_t)blockIdx.x * 32 + (tcount & 31)) * 8;
        if (tl) tlp[0] = __builtin_amdgcn_s_memrealtime();
        lane_addresses();
        asm volatile(DUO_REFILL_ASM : : [va0] "v"(va[0]), [vb] "v"(vb) : "memory", DUO_CLOBBER_FRAG);
        const int n_tile = c_tile + walk.step;
        const bool more = n_tile < walk.t_end;
        if (more) decode(n_tile, nxt);
        else nxt = cur;                       // nothing left to stage: re-stage this tile's first periods (valid memory, unread)
        const uint32_t nper = (uint32_t)(nkc * cur.nky);
        const u64 sAn1 = a_of(cur, 1), sBn1 = b_of(cur, 1), sAn2 = a_of(cur, 2), sBn2 = b_of(cur, 2);
        const u64 sAnT0 = a_of(nxt, 0), sBnT0 = b_of(nxt, 0), sAnT1 = a_of(nxt, 1), sBnT1 = b_of(nxt, 1);
        const int dAs = (int)dWl2, dAw = (int)(64 - (cur.nky - 1) * dWl2), dBs = 6 * p.Cin, dBw = 64 - (cur.nky - 1) * 6 * p.Cin;
        asm volatile(DUO_TILE_ASM
                     : DUO_ACC_RW
                     : [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [vb] "v"(vb), [voa0] "v"(voa[0]), [voa1] "v"(voa[1]),
                       [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vz0] "v"(vz0),
                       [vz1] "v"(vz1), [vzero] "v"(vzero), [vr0] "v"(vr0), [sAn1] "s"(sAn1), [sBn1] "s"(sBn1), [sAn2] "s"(sAn2),
                       [sBn2] "s"(sBn2), [sAnT0] "s"(sAnT0), [sBnT0] "s"(sBnT0), [sAnT1] "s"(sAnT1), [sBnT1] "s"(sBnT1), [slo] "s"(cur.lo),
                       [ssp] "s"(cur.span), [sloT] "s"(nxt.lo), [sspT] "s"(nxt.span), [sdAs] "s"(dAs), [sdAw] "s"(dAw), [sdBs] "s"(dBs),
                       [sdBw] "s"(dBw), [snky] "s"((uint32_t)cur.nky), [sky2] "s"((uint32_t)(2 % cur.nky)), [snper] "s"(nper),
                       [s2cin] "s"(s2cin), [sflag] "s"(flag), [sldsA] "s"(sldsA), [sldsB] "s"(sldsB)
                     : "memory", "scc", "vcc", DUO_CLOBBER_S, DUO_CLOBBER_FRAG);
        asm volatile("s_nop 15\n\ts_nop 15" : DUO_ACC_RW : : "memory");
        if (tl) tlp[1] = __builtin_amdgcn_s_memrealtime();
        if (!(p.tune & 64)) {
            const int mw = cur.m0 + wm * 128, nw = cur.n0 + wn * 64;
            char *patch = lds + DUO_NEED + wv * 2048;
#define DUO_RD(I, K)                                                                                                                  \
    {                                                                                                                                 \
        float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;                                                   \
        asm volatile(DUO_READ_ACC_##I##_ASM                                                                                           \
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7), "=v"(t8), "=v"(t9), "=v"(t10), \
                       "=v"(t11), "=v"(t12), "=v"(t13), "=v"(t14), "=v"(t15), DUO_ACC_RW);                                            \
        acc[K][0] = make_uint2(pack_bf16x2_v(t0, t1), pack_bf16x2_v(t2, t3));                                                         \
        acc[K][1] = make_uint2(pack_bf16x2_v(t4, t5), pack_bf16x2_v(t6, t7));                                                         \
        acc[K][2] = make_uint2(pack_bf16x2_v(t8, t9), pack_bf16x2_v(t10, t11));                                                       \
        acc[K][3] = make_uint2(pack_bf16x2_v(t12, t13), pack_bf16x2_v(t14, t15));                                                     \
    }
#define DUO_EPI(MW)                                                                                                                   \
    if (p.ep.out_raw && p.ep.out_act) ig_epilogue_rows16<4, NOPS_, 1, uint2, 3>(p, patch, acc, MW, nw, lane);                         \
    else if (p.ep.out_act) ig_epilogue_rows16<4, NOPS_, 1, uint2, 2>(p, patch, acc, MW, nw, lane);                                    \
    else ig_epilogue_rows16<4, NOPS_, 1, uint2, 1>(p, patch, acc, MW, nw, lane);
            // 64 rows at a time: the epilogue has 128 registers (the other 128 are the accumulator file)
            {
                uint2 acc[4][4];
                DUO_RD(0, 0) DUO_RD(1, 1) DUO_RD(2, 2) DUO_RD(3, 3)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (see conv_row_lw_kernel)
                DUO_EPI(mw)
            }
            {
                uint2 acc[4][4];
                DUO_RD(4, 0) DUO_RD(5, 1) DUO_RD(6, 2) DUO_RD(7, 3)
                DUO_EPI(mw + 64)
            }
#undef DUO_RD
#undef DUO_EPI
        }
        if (tl) { tlp[2] = __builtin_amdgcn_s_memrealtime(); tlp[3] = (unsigned long long)c_tile; }
        ++tcount;
        if (!more) break;
        c_tile = n_tile;
        cur = nxt;
        flag = 1;
    }
#undef DUO_ACC_RW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the lone-wave loop for the 128-output-channel layers (mod2, the GSCNN shape stream) --------------------------------------------------
// conv_row_tall_kernel: tile 512 pixels x 128 channels, wave w = pixels 128 w .. + 127 x all 128 channels, i.e. conv_row_lw_kernel's wave
// tile, accumulator layout and epilogue; K staged in 32-channel periods (tools/gen_conv_lw.py, "tall": two row buffers of 576 rows x 64 B,
// four 8-KiB B slots, 16-KiB epilogue patches = 120 KiB).  Replaces conv_row_pp128_kernel (8 waves of 128 x 64, hipcc-scheduled ping-pong:
// 0.38-0.40 of peak) where Cin % 64 == 0; same k order (32-channel block, kernel row, tap), same epilogue: bit-identical results.
constexpr int TALL_ABUF = 576 * 64, TALL_BSLOT = 128 * 64, TALL_NEED = 2 * TALL_ABUF + 4 * TALL_BSLOT;

template <int NOPS_>
__global__ __launch_bounds__(256, 1) void conv_row_tall_kernel(const ConvParams p)
{
    typedef unsigned long long u64;
    __shared__ __attribute__((aligned(1024))) char lds[TALL_NEED + 4 * 4096];   // 120 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.dil;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    uint32_t va[3], vb, voa[9], vob[2], vz0, vz1, vr0;
    u32x4_t vzero;
    auto lane_addresses = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int frow = l & 15, fq = l >> 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int rsh = frow + kx * d;
            va[kx] = lbase + (wv * 128 + rsh) * 64 + ((fq ^ ((rsh >> 1) & 3)) << 4);
        }
        vb = lbase + 2 * TALL_ABUF + frow * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);
        const int srow = l >> 2, chunk = ((l & 3) ^ ((l >> 3) & 3)) << 4;
#pragma unroll
        for (int j = 0; j < 9; ++j) voa[j] = (uint32_t)(((wv * 9 + j) * 16 + srow) * (p.ldx * 2) + chunk);
#pragma unroll
        for (int j = 0; j < 2; ++j) vob[j] = (uint32_t)(((wv * 2 + j) * 16 + srow) * (p.Ktot * 2) + chunk);
        vz0 = lbase + wv * 9216 + l * 16;
        vz1 = vz0 + TALL_ABUF;
        vr0 = (uint32_t)(wv * 144 + srow);     // buffer row of this lane in piece 0 (piece j: + 16 j)
        vzero = (u32x4_t){0u, 0u, 0u, 0u};
    };
    lane_addresses();
    const uint32_t sldsA = __builtin_amdgcn_readfirstlane(lbase + wv * 9216);
    const uint32_t sldsB = __builtin_amdgcn_readfirstlane(lbase + 2 * TALL_ABUF + wv * 2048);
    const uint32_t s2cin = (uint32_t)(2 * p.Cin);
    const long long dWl2 = 2ll * d * p.W * p.ldx;
    const int nkc = p.Cin / 32;

    struct Tile {
        int m0, n0, kylo, nky;
        uint32_t lo, span;
        u64 abase, bbase;
    };
    auto decode = [&](int tile, Tile &t) {
        const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
        t.m0 = tm * 512;
        t.n0 = tn * 128;
        const int n = t.m0 / p.HoWo, rem = t.m0 - n * p.HoWo;
        const int ho = rem / p.W, x0 = rem - ho * p.W;
        t.kylo = ho - d < 0 ? 1 : 0;
        t.nky = (ho + d >= p.H ? 1 : 2) - t.kylo + 1;
        // pixel x0 - d + r: outside the row for r < d in the row's first tile and for r >= 512 + d in its last
        t.lo = x0 == 0 ? (uint32_t)d : 0u;
        t.span = (x0 + 512 == p.W ? 512u + d : 576u) - t.lo;
        t.abase = (u64)p.x + (u64)(2ll * ((long long)((n * p.H + ho) * p.W + (x0 - d)) * p.ldx));
        if (p.tune & 2048) t.abase = (u64)p.x + (u64)(2ll * ((long long)((d + (tile & 7)) * p.W + (x0 - d)) * p.ldx));   // timing ablation (tuning build): every tile reads the same few rows (L2-resident input)
        t.bbase = (u64)p.w + (u64)(2ll * (long long)t.n0 * p.Ktot);
    };
    // period q of a tile = (32-channel block q / nky, kernel row kylo + q % nky)
    auto a_of = [&](const Tile &t, int q) { return t.abase + (u64)((long long)(t.kylo + q % t.nky - 1) * dWl2 + (q / t.nky) * 64); };
    auto b_of = [&](const Tile &t, int q) { return t.bbase + (u64)(2ll * ((long long)(t.kylo + q % t.nky) * 3 * p.Cin + (q / t.nky) * 32)); };

    Tile cur, nxt;
    int c_tile = walk.t;
    decode(c_tile, cur);
    // prologue (generic pieces): the row buffers of periods 0 and 1, B of k-steps 0 .. 3 = period 0's three taps and period 1's first
    {
        const char *a0 = (const char *)a_of(cur, 0), *a1 = (const char *)a_of(cur, 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const bool ok = vr0 + 16 * j - cur.lo < cur.span;
            glds16(ok ? (const void *)(a0 + voa[j]) : (const void *)lw_zero_page, lds + (wv * 9 + j) * 1024);
            glds16(ok ? (const void *)(a1 + voa[j]) : (const void *)lw_zero_page, lds + TALL_ABUF + (wv * 9 + j) * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char *bb = q < 3 ? (const char *)b_of(cur, 0) + 2ll * q * p.Cin : (const char *)b_of(cur, 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(bb + vob[j], lds + 2 * TALL_ABUF + q * TALL_BSLOT + (wv * 2 + j) * 1024);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;   // the accumulator file is occupied (see conv_row_lw_kernel)
    asm volatile(LW_ZERO_ACC_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define LW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)
    uint32_t flag = 0, par = 0;
    int tcount = 0;
#pragma unroll 1
    for (;;) {
        const bool tl = (p.tune & 1024) && tid == 0 && tcount < 32 && blockIdx.x < 512;     // tools/duo_timeline.py --tall
        unsigned long long *tlp = lw_tlog + ((size_t)blockIdx.x * 32 + (tcount & 31)) * 8;
        if (tl) tlp[0] = __builtin_amdgcn_s_memrealtime();
        lane_addresses();
        asm volatile(TALL_REFILL_ASM : : [va0] "v"(va[0]), [vbf] "v"(vb + par * 2 * TALL_BSLOT) : "memory", LW_CLOBBER_FRAG);
        const int n_tile = c_tile + walk.step;
        const bool more = n_tile < walk.t_end;
        if (more) decode(n_tile, nxt);
        else nxt = cur;                       // nothing left to stage: re-stage this tile's first periods (valid memory, unread)
        const uint32_t nper = (uint32_t)(nkc * cur.nky);
        const u64 sBn1 = b_of(cur, 1), sAn2 = a_of(cur, 2), sBn2 = b_of(cur, 2);
        const u64 sAnT0 = a_of(nxt, 0), sBnT0 = b_of(nxt, 0), sAnT1 = a_of(nxt, 1), sBnT1 = b_of(nxt, 1);
        const int dAs = (int)dWl2, dAw = (int)(64 - (cur.nky - 1) * dWl2), dBs = 6 * p.Cin, dBw = 64 - (cur.nky - 1) * 6 * p.Cin;
        asm volatile(TALL_TILE_ASM
                     : LW_ACC_RW
                     : [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [vb] "v"(vb), [voa0] "v"(voa[0]), [voa1] "v"(voa[1]),
                       [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [voa5] "v"(voa[5]), [voa6] "v"(voa[6]), [voa7] "v"(voa[7]),
                       [voa8] "v"(voa[8]), [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vz0] "v"(vz0), [vz1] "v"(vz1), [vzero] "v"(vzero),
                       [vr0] "v"(vr0), [sBn1] "s"(sBn1), [sAn2] "s"(sAn2), [sBn2] "s"(sBn2), [sAnT0] "s"(sAnT0), [sBnT0] "s"(sBnT0),
                       [sAnT1] "s"(sAnT1), [sBnT1] "s"(sBnT1), [slo] "s"(cur.lo), [ssp] "s"(cur.span), [sloT] "s"(nxt.lo), [sspT] "s"(nxt.span),
                       [sdAs] "s"(dAs), [sdAw] "s"(dAw), [sdBs] "s"(dBs), [sdBw] "s"(dBw), [snky] "s"((uint32_t)cur.nky),
                       [sky2] "s"((uint32_t)(2 % cur.nky)), [snper] "s"(nper), [s2cin] "s"(s2cin), [sflag] "s"(flag), [spar] "s"(par),
                       [sldsA] "s"(sldsA), [sldsB] "s"(sldsB)
                     : "memory", "scc", "vcc", DUO_CLOBBER_S, LW_CLOBBER_FRAG);
        par = (par + (nper >> 1)) & 1u;
        asm volatile("s_nop 15\n\ts_nop 15" : LW_ACC_RW : : "memory");
        if (tl) tlp[1] = __builtin_amdgcn_s_memrealtime();
        if (!(p.tune & 64)) {
        const int mw = cur.m0 + wv * 128, nw = cur.n0;
        char *patch = lds + TALL_NEED + wv * 4096;
#define LW_RD(I, JG)                                                                                                                  \
    {                                                                                                                                 \
        float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;                                                   \
        asm volatile(LW_READ_ACC_##I##_##JG##_ASM                                                                                     \
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7), "=v"(t8), "=v"(t9), "=v"(t10), \
                       "=v"(t11), "=v"(t12), "=v"(t13), "=v"(t14), "=v"(t15), LW_ACC_RW);                                             \
        acc[I][0] = make_uint2(pack_bf16x2_v(t0, t1), pack_bf16x2_v(t2, t3));                                                         \
        acc[I][1] = make_uint2(pack_bf16x2_v(t4, t5), pack_bf16x2_v(t6, t7));                                                         \
        acc[I][2] = make_uint2(pack_bf16x2_v(t8, t9), pack_bf16x2_v(t10, t11));                                                       \
        acc[I][3] = make_uint2(pack_bf16x2_v(t12, t13), pack_bf16x2_v(t14, t15));                                                     \
    }
        LwEpiConsts k0, k1;
        lw_epilogue_consts<NOPS_>(p, nw, lane, k0);      // (their latency passes under the accumulator read-out)
        {
            uint2 acc[8][4];
            LW_RD(0, 0) LW_RD(1, 0) LW_RD(2, 0) LW_RD(3, 0) LW_RD(4, 0) LW_RD(5, 0) LW_RD(6, 0) LW_RD(7, 0)
            if (tl) tlp[4] = __builtin_amdgcn_s_memrealtime();
            // what ran ahead into the next tile's buffers has landed before the first store: the first two waits of that tile may then
            // leave every store outstanding
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tl) tlp[5] = __builtin_amdgcn_s_memrealtime();
            lw_epilogue_consts<NOPS_>(p, nw + 64, lane, k1);
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw, lane, k0);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw, lane, k0);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw, lane, k0);
            if (tl) tlp[6] = __builtin_amdgcn_s_memrealtime();
        }
        {
            uint2 acc[8][4];
            LW_RD(0, 1) LW_RD(1, 1) LW_RD(2, 1) LW_RD(3, 1) LW_RD(4, 1) LW_RD(5, 1) LW_RD(6, 1) LW_RD(7, 1)
            if (tl) tlp[7] = __builtin_amdgcn_s_memrealtime();
            if (p.ep.out_raw && p.ep.out_act) lw_epilogue_rows16<NOPS_, 3>(p, patch, acc, mw, nw + 64, lane, k1);
            else if (p.ep.out_act) lw_epilogue_rows16<NOPS_, 2>(p, patch, acc, mw, nw + 64, lane, k1);
            else lw_epilogue_rows16<NOPS_, 1>(p, patch, acc, mw, nw + 64, lane, k1);
        }
#undef LW_RD
        }
        if (tl) { tlp[2] = __builtin_amdgcn_s_memrealtime(); tlp[3] = (unsigned long long)c_tile; }
        ++tcount;
        if (!more) break;
        c_tile = n_tile;
        cur = nxt;
        flag = 1;
    }
#undef LW_ACC_RW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

bool kd_launch_conv_row_lw(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s)
{
    const dim3 g(grid), b(256);
    switch (nops_sums) {
    case 0: hipLaunchKernelGGL((conv_row_lw_kernel<0>), g, b, 0, s, p); return true;
    case 1: hipLaunchKernelGGL((conv_row_lw_kernel<1>), g, b, 0, s, p); return true;
    case 2: hipLaunchKernelGGL((conv_row_lw_kernel<2>), g, b, 0, s, p); return true;
    case 3: hipLaunchKernelGGL((conv_row_lw_kernel<3>), g, b, 0, s, p); return true;
    case 5: hipLaunchKernelGGL((conv_row_lw_kernel<5>), g, b, 0, s, p); return true;
    case 6: hipLaunchKernelGGL((conv_row_lw_kernel<6>), g, b, 0, s, p); return true;
    case 16: hipLaunchKernelGGL((conv_row_lw_kernel<16>), g, b, 0, s, p); return true;     // classifier epilogue (no operand)
    default: return false;
    }
}

bool kd_launch_conv_pw_lw(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s)
{
    const dim3 g(grid), b(256);
    switch (nops_sums) {
    case 0: hipLaunchKernelGGL((conv_pw_lw_kernel<0>), g, b, 0, s, p); return true;
    case 1: hipLaunchKernelGGL((conv_pw_lw_kernel<1>), g, b, 0, s, p); return true;
    case 2: hipLaunchKernelGGL((conv_pw_lw_kernel<2>), g, b, 0, s, p); return true;
    case 3: hipLaunchKernelGGL((conv_pw_lw_kernel<3>), g, b, 0, s, p); return true;
    case 5: hipLaunchKernelGGL((conv_pw_lw_kernel<5>), g, b, 0, s, p); return true;
    case 6: hipLaunchKernelGGL((conv_pw_lw_kernel<6>), g, b, 0, s, p); return true;
    default: return false;
    }
}

bool kd_launch_conv_row_duo(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s)
{
    const dim3 g(grid), b(256);
    switch (nops_sums) {
    case 0: hipLaunchKernelGGL((conv_row_duo_kernel<0>), g, b, 0, s, p); return true;
    case 1: hipLaunchKernelGGL((conv_row_duo_kernel<1>), g, b, 0, s, p); return true;
    default: return false;      // two operands: the epilogue does not fit the 128 registers beside the accumulator file (hipcc then moves
                                // accumulators around: tools/check_lw_asm.py); eval-BN sums: one partial row per 128 pixels in one pass
    }
}

bool kd_launch_conv_row_tall(const ConvParams &p, int nops_sums, unsigned grid, hipStream_t s)
{
    const dim3 g(grid), b(256);
    switch (nops_sums) {
    case 0: hipLaunchKernelGGL((conv_row_tall_kernel<0>), g, b, 0, s, p); return true;
    case 1: hipLaunchKernelGGL((conv_row_tall_kernel<1>), g, b, 0, s, p); return true;
    case 2: hipLaunchKernelGGL((conv_row_tall_kernel<2>), g, b, 0, s, p); return true;
    case 5: hipLaunchKernelGGL((conv_row_tall_kernel<5>), g, b, 0, s, p); return true;
    case 6: hipLaunchKernelGGL((conv_row_tall_kernel<6>), g, b, 0, s, p); return true;
    default: return false;
    }
}

int kd_lw_tlog_copy(unsigned long long *dst, size_t bytes)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(lw_tlog), bytes < sizeof(lw_tlog) ? bytes : sizeof(lw_tlog), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}

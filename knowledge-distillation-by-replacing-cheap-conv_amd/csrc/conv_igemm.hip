// Implicit-GEMM NHWC convolution on gfx950 MFMA (kd_conv2d_fwd; also the dgrad
// engine when fed KD_PACK_DGRAD weights).  See include/kdcc.h for semantics.
//
// GEMM view:  Y[m][co] = sum_{tap,ci} X[pixel(m) + tap][ci] * Wp[co][tap][ci]
//   M = N*Ho*Wo pixels, N = Cout, K = kh*kw*Cin.
// Block tile 256(M) x 256(N) (waves 2x4 of 128x64, two K stages) or 256 x 128 (waves 4x2 of 64x64, three K
// stages, counted vmcnt + raw s_barrier so the LDS-DMA of stage k+2 spans the barrier of stage k); MFMA 16x16 tiles.
// One K stage = 128 bytes of K per row (64 bf16 / 32 f32): the A tile is gathered
// (im2col) and the B tile streamed straight into LDS with global_load_lds (16 B per
// lane), rows XOR-swizzled on the SOURCE side so that the ds_read_b128 fragment
// reads are bank-conflict free.  Out-of-image taps and tile tails read a zero page.
// Epilogue: accumulators -> per-wave LDS patch -> 8-channel (16/32 B) vector rows:
// residual add, BN(eval)+ReLU of the next layer, ReLU-mask backward, dual outputs.
#include <stdlib.h>
#include <atomic>
#include <string.h>

#include "conv_common.h"

namespace {

constexpr int EP_LD = 68;                   // floats per epilogue row (64 + pad)

// Tile configurations (8 waves = 512 threads, one workgroup per CU, two waves per SIMD):
//   CfgWide   256(M) x 256(N), waves 2x4 of 128x64, K stage = 128 B per row, two stages (128 KiB LDS)  -- Cout > 128
//   CfgNarrow2 256(M) x 128(N), waves 4x2 of 64x64, K stage = 64 B per row, three stages (72 KiB), two workgroups per
//             CU so one tile's epilogue / pipeline fill overlaps the other's main loop                  -- Cout <= 128
//             (short-K layers: mod2 3x3 128->128 625 -> 739 TFLOP/s, bot_fine / classifier 1x1 +36..44 % over CfgNarrow)
//   CfgNarrow 256 x 128, K stage = 128 B, three stages (144 KiB), one workgroup per CU: A/B only (KDCC_CONV_CFG=narrow1)
//   CfgDeep   256 x 256 with 64-B K stages, four resident (three in flight): measured SLOWER than CfgWide (878 vs 944
//             TFLOP/s over the student's shapes) -- prefetch depth is not the limiter; kept for A/B runs only.
// Measured anatomy of CfgWide on the 512/1024-channel 3x3 layers (tools/bench_conv.py ablations): MFMA + LDS fragment
// reads alone 1.5-1.7 PFLOP/s; + the LDS-DMA instruction stream hitting one cached page 1.4; + real L2 traffic 1.1.
// So the staging traffic (both operands) costs ~22 %, its issue ~9 %; wide tiles cut that traffic 1.5x vs 128-wide.
template <int MI_, int WM_, int WN_, int NST_, int RB_, int WPE_ = 2, int PIPE_ = 0> struct Cfg {
    static constexpr int MI = MI_, WM = WM_, WN = WN_, NST = NST_, RB = RB_, WPE = WPE_, PIPE = PIPE_;   // WPE: waves per SIMD to compile for
    static constexpr int BM = WM * MI * 16, BN = WN * 64;
    static constexpr int STAGE_A = BM * RB, STAGE_B = BN * RB, STAGE = STAGE_A + STAGE_B;
    static constexpr int PR = 1024 / RB;                        // rows per 1-KiB LDS-DMA piece
    static constexpr int NW = WM * WN;                           // waves per workgroup
    static constexpr int GA = BM / PR / NW, GB = BN / PR / NW;  // pieces per wave per stage (A, B)
    static constexpr int LDS_BYTES = NST * STAGE;
    static_assert(NW * 32 * EP_LD * 4 <= LDS_BYTES && NW * 16 * MI * 128 <= LDS_BYTES, "epilogue patches must fit the stage buffers");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
typedef Cfg<4, 4, 2, 3, 128> CfgNarrow;
typedef Cfg<8, 2, 4, 2, 128, 2, 1> CfgWide;   // bf16: software-pipelined main loop (see conv_igemm_row_kernel)
typedef Cfg<8, 2, 4, 2, 128> CfgWideF;        // fp32 parity path
typedef Cfg<8, 2, 4, 4, 64> CfgDeep;     // A/B runs only: KDCC_CONV_CFG=deep
typedef Cfg<8, 2, 2, 3, 64, 2> CfgHalf;      // 256 x 128, FOUR waves of 128x64, 72 KiB: two workgroups per CU (A/B: KDCC_CONV_CFG=half)
typedef Cfg<4, 4, 2, 3, 64, 4> CfgNarrow2;   // 256 x 128 with 64-B K stages, 72 KiB, <= 128 VGPRs: two workgroups per CU

__device__ __attribute__((aligned(256))) uint32_t kd_zero_page[64];  // zero-initialised
__device__ unsigned long long kd_conv_tlog[256 * 32 * 8 + 256 * 64];   // KDCC_CONV_TUNE=512: per-workgroup, per-tile timestamps (debug)


// load/store 8 channels with tail + alignment handling
template <typename U>
__device__ __forceinline__ void ld8_guard(const U *p, int valid, bool vec, float (&v)[8])
{
    if (vec && valid == 8) {
        ld8(p, v);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < valid ? Elem<U>::ld(p + e) : 0.f;
    }
}
template <typename U>
__device__ __forceinline__ void st8_guard(U *p, int valid, bool vec, const float (&v)[8])
{
    if (vec && valid == 8) {
        st8(p, v);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < valid) Elem<U>::st(p + e, v[e]);
    }
}

// ---- epilogue: accumulators -> per-wave LDS patch -> 8-channel vector rows (shared by both main loops) -----------------
template <typename T, int MI, int EB, int NJ = 4>   // EB: passes (8 rows each) whose epilogue operands are loaded together
__device__ __forceinline__ void ig_epilogue(const ConvParams &p, char *lds, f32x4_t (&acc)[MI][NJ], int m0, int n0, int wm,
                                            int wn, int wv, int lane)
{
    static_assert(NJ % 4 == 0, "64-column groups");
#pragma unroll
    for (int jg = 0; jg < NJ / 4; ++jg) {   // the wave's columns in groups of 64
    const int frow = lane & 15, fq = lane >> 4;
    const kd_conv_epilogue &e = p.ep;
    const int cg = (lane & 7) * 8;
    const int c0 = n0 + wn * (16 * NJ) + jg * 64 + cg;
    const int valid = p.Cout - c0 >= 8 ? 8 : (p.Cout - c0 > 0 ? p.Cout - c0 : 0);
    const bool vec = p.vec_ok != 0;

    // per-channel epilogue parameters: whole 32-B vectors, all in flight together (a guarded scalar load per element costs
    // one exposed memory latency each -- 24 of them per tile -- before the first output row can leave)
    float mscale[8], ascale[8], ashift[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { mscale[q] = 1.f; ascale[q] = 1.f; ashift[q] = 0.f; }
    if (p.tune & 256) {
    } else if (!(p.tune & 32) && __all(valid == 8)) {
        if (e.mask_scale) ld8(e.mask_scale + c0, mscale);
        if (e.act_scale) ld8(e.act_scale + c0, ascale);
        if (e.act_shift) ld8(e.act_shift + c0, ashift);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool ok = q < valid;
            if (e.mask_scale && ok) mscale[q] = e.mask_scale[c0 + q];
            if (e.act_scale && ok) ascale[q] = e.act_scale[c0 + q];
            if (e.act_shift && ok) ashift[q] = e.act_shift[c0 + q];
        }
    }

    // Accumulator layout (the main loops issue the MFMAs with swapped operands, i.e. they compute the transposed tile):
    // acc[i][j][r] = output(pixel 16*i + (lane & 15), channel 16*j + 4*(lane >> 4) + r) -- four consecutive channels of
    // one pixel per lane, so the patch is written with one vector store per MFMA tile instead of four scalar ones.
    // Each wave owns a private 32-pixel x 64-channel patch inside the (now idle) stage buffers, 32 rows of its sub-tile
    // in turn: fp32 (272-B rows) when T is fp32 or the raw output is fp32; else bf16, the wave's whole sub-tile at once
    // (swizzled 128-B rows; the raw output is exactly the patch contents).
    constexpr int PATCH = 32 * EP_LD * 4;
    constexpr int ROWB16 = 128;                      // bf16 patch row: 64 channels, 8-B chunk c of row r stored at c ^ (r & 15)
    const bool p16 = sizeof(T) == 2 && !e.raw_f32;   // block-uniform
    char *epb = lds + wv * (p16 ? 16 * MI * ROWB16 : PATCH);
    float *ep = (float *)epb;
    if (p16) {
        // the bf16 patch holds the wave's whole 16*MI x 64 sub-tile: one write phase, one wave barrier, then every
        // 8-row pass reads independently (no write -> barrier -> read round trip per 32 rows)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (a previous 64-column group's reads are done)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t v = acc[i][jg * 4 + j];
                *(uint2 *)(epb + (i * 16 + frow) * ROWB16 + (((j * 4 + fq) ^ frow) << 3)) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int half = 0; half < MI / 2; ++half) {
        if (!p16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t v = acc[half * 2 + i][jg * 4 + j];
                    *(float4 *)(ep + (i * 16 + frow) * EP_LD + j * 16 + fq * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
        if (!p16) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        bool done = false;
        if constexpr (sizeof(T) == 2) {
            // fast path (whole 8-channel vectors, 16-B aligned operands): the residual / mask operands of the four passes
            // are loaded together, ahead of the stores they could alias, so their latency is paid once per 32 rows
            if (p.epi_batch && __all(vec && valid == 8)) {
                const int mb = m0 + wm * (16 * MI) + half * 32 + (lane >> 3);
#pragma unroll
                for (int pb = 0; pb < 4; pb += EB) {
                uint4 rp[EB], rm[EB], rq[EB];
#pragma unroll
                for (int pass = pb; pass < pb + EB; ++pass) {
                    const size_t m = (size_t)min(mb + pass * 8, p.M - 1);
                    if (e.res_pre) rp[pass - pb] = *(const uint4 *)((const T *)e.res_pre + m * e.ld_res_pre + c0);
                    if (e.mask) rm[pass - pb] = *(const uint4 *)((const T *)e.mask + m * e.ld_mask + c0);
                    if (e.res_post) rq[pass - pb] = *(const uint4 *)((const T *)e.res_post + m * e.ld_res_post + c0);
                }
#pragma unroll
                for (int pass = pb; pass < pb + EB; ++pass) {
                    const int row = pass * 8 + (lane >> 3);
                    const int m = mb + pass * 8;
                    if (m >= p.M) continue;
                    float v[8], t[8];
                    uint4 rawv = make_uint4(0u, 0u, 0u, 0u);   // the row's 8 channels as stored bf16 (p16 only)
                    if (p16) {
                        const int prow = half * 32 + row, c2 = (lane & 7) * 2;
                        const uint2 lo = *(const uint2 *)(epb + prow * ROWB16 + ((c2 ^ (prow & 15)) << 3));
                        const uint2 hi = *(const uint2 *)(epb + prow * ROWB16 + (((c2 + 1) ^ (prow & 15)) << 3));
                        rawv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        ld8((const bf16_t *)&rawv, v);
                    } else {
                        const float4 lo = *(const float4 *)(ep + row * EP_LD + cg);
                        const float4 hi = *(const float4 *)(ep + row * EP_LD + cg + 4);
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    }
                    if (e.res_pre) {
                        ld8((const T *)&rp[pass - pb], t);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += t[q];
                    }
                    if (e.mask) {
                        ld8((const T *)&rm[pass - pb], t);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = t[q] > 0.f ? v[q] * mscale[q] : 0.f;
                    }
                    if (e.res_post) {
                        ld8((const T *)&rq[pass - pb], t);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += t[q];
                    }
                    if (e.out_raw && !(p.tune & 128)) {
                        if (e.raw_f32) st8((float *)e.out_raw + (size_t)m * e.ld_raw + c0, v);
                        else if (p16 && !e.res_pre && !e.mask && !e.res_post) *(uint4 *)((T *)e.out_raw + (size_t)m * e.ld_raw + c0) = rawv;
                        else st8((T *)e.out_raw + (size_t)m * e.ld_raw + c0, v);
                    }
                    if (e.out_act) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float a = v[q] * ascale[q] + ashift[q];
                            t[q] = e.act_relu ? fmaxf(a, 0.f) : a;
                        }
                        if (!(p.tune & 128) || t[0] == 1.2345f) st8((T *)e.out_act + (size_t)m * e.ld_act + c0, t);   // 128: timing ablation
                    }
                }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3);
            const int m = m0 + wm * (16 * MI) + half * 32 + row;
            if (m >= p.M || valid == 0) continue;
            float v[8];
            if (p16) {
                const int prow = half * 32 + row, c2 = (lane & 7) * 2;
                const uint2 lo = *(const uint2 *)(epb + prow * ROWB16 + ((c2 ^ (prow & 15)) << 3));
                const uint2 hi = *(const uint2 *)(epb + prow * ROWB16 + (((c2 + 1) ^ (prow & 15)) << 3));
                const uint4 rawv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                ld8((const bf16_t *)&rawv, v);
            } else {
                const float4 lo = *(const float4 *)(ep + row * EP_LD + cg);
                const float4 hi = *(const float4 *)(ep + row * EP_LD + cg + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            }
            float t[8];
            if (e.res_pre) {
                ld8_guard((const T *)e.res_pre + (size_t)m * e.ld_res_pre + c0, valid, vec, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += t[q];
            }
            if (e.mask) {
                ld8_guard((const T *)e.mask + (size_t)m * e.ld_mask + c0, valid, vec, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = t[q] > 0.f ? v[q] * mscale[q] : 0.f;
            }
            if (e.res_post) {
                ld8_guard((const T *)e.res_post + (size_t)m * e.ld_res_post + c0, valid, vec, t);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += t[q];
            }
            if (e.out_raw) {
                if (e.raw_f32) st8_guard((float *)e.out_raw + (size_t)m * e.ld_raw + c0, valid, vec, v);
                else st8_guard((T *)e.out_raw + (size_t)m * e.ld_raw + c0, valid, vec, v);
            }
            if (e.out_act) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float a = v[q] * ascale[q] + ashift[q];
                    t[q] = e.act_relu ? fmaxf(a, 0.f) : a;
                }
                st8_guard((T *)e.out_act + (size_t)m * e.ld_act + c0, valid, vec, t);
            }
        }
        }
        if (!p16) {   // the next half overwrites the fp32 patch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    }
}

template <typename T, typename CF>
__global__ __launch_bounds__(64 * CF::NW, CF::WPE) void conv_igemm_kernel(const ConvParams p)
{
    // One LDS array: the DMA is issued through inline asm, so hipcc sees only the fragment reads and inserts no waits.
    __shared__ __attribute__((aligned(16))) char lds[CF::LDS_BYTES];
    constexpr int ES = sizeof(T);
    constexpr int RB = CF::RB;
    constexpr int BK = RB / ES;     // K elements per stage
    constexpr int EPC = 16 / ES;    // elements per 16-B chunk
    constexpr int MI = CF::MI, GA = CF::GA, GB = CF::GB, NST = CF::NST, PR = CF::PR;
    constexpr int CPR = RB / 16;    // 16-B chunks per row

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / CF::WN, wn = wv % CF::WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * CF::BM, n0 = tn * CF::BN;

    const T *__restrict__ xg = (const T *)p.x;
    const T *__restrict__ wg = (const T *)p.w;
    const T *zero = (const T *)kd_zero_page;

    // ---- per-lane staging state: GA A rows + GB B rows, fixed over the K loop -------------
    // a piece (64 lanes x 16 B, LDS-linear) covers PR rows; lane -> (row in piece, slot); the swizzle is applied on the
    // SOURCE side: LDS slot s of row r holds chunk s ^ (r & 7) (128-B rows) / s ^ ((r >> 1) & 3) (64-B rows)
    const int srow = lane / CPR;
    const int chunk = RB == 128 ? ((lane & 7) ^ srow) : ((lane & 3) ^ ((srow >> 1) & 3));
    int a_off[GA];
    uint32_t a_mask[GA];
    int b_off[GB];
    const int ntaps = p.kh * p.kw;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int r = (wv * GA + j) * PR + srow;
        const int m = m0 + r;
        a_off[j] = 0;
        a_mask[j] = 0;
        if (m < p.M && ntaps == 1 && p.stride == 1 && p.pad == 0) {
            // 1x1 / stride 1: output pixel m is input pixel m (no divisions, no border)
            a_off[j] = m * p.ldx + chunk * EPC;
            a_mask[j] = 1u;
        } else if (m < p.M) {
            const int n = m / p.HoWo, rem = m - n * p.HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            a_off[j] = ((n * p.H + hi0) * p.W + wi0) * p.ldx + chunk * EPC;
            uint32_t mk = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int ky = t / p.kw, kx = t - ky * p.kw;
                const int hi = hi0 + ky * p.dil, wi = wi0 + kx * p.dil;
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) mk |= 1u << t;
            }
            a_mask[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int nn = n0 + (wv * GB + j) * PR + srow;
        b_off[j] = nn < p.Cout ? nn * p.Ktot + chunk * EPC : -1;
    }

    // next stage to issue.  K order is channel-block outer, tap inner: the nine taps of one 64-channel block re-read
    // (shifted) the same few image rows and one thin weight slab back to back, so they stay L2-resident; with taps outer
    // the XCD's working set (every channel of ~18 image rows + all weights) overflowed its 4 MiB L2 between re-uses.
    int s_kt = 0, s_tap = 0, s_cb = 0;
    auto stage = [&]() {
        const int ky = s_tap / p.kw, kx = s_tap - ky * p.kw;
        const int tap_off = (ky * p.dil * p.W + kx * p.dil) * p.ldx + s_cb * BK;
        const int w_off = s_tap * p.Cin + s_cb * BK;
        char *la = lds + (s_kt % NST) * CF::STAGE + wv * (GA * 1024);
        char *lb = lds + (s_kt % NST) * CF::STAGE + CF::STAGE_A + wv * (GB * 1024);
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            const T *src = ((a_mask[j] >> s_tap) & 1u) ? xg + (a_off[j] + tap_off) : zero;
            glds16(src, la + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const T *src = b_off[j] >= 0 ? wg + (b_off[j] + w_off) : zero;
            glds16(src, lb + j * 1024);
        }
        ++s_kt;
        if (++s_tap == ntaps) { s_tap = 0; ++s_cb; }
        // keep the DMA issue ahead of the MFMA block it is meant to overlap with
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;

    // ---- K loop: NST stages resident, NST-1 in flight ahead of the one being consumed; one barrier per stage ----
    // A wave's vmcnt counts its own DMA pieces in issue order: to know stage k+1 has landed while stage k+2 is still
    // in flight, wait until at most (GA+GB) pieces are outstanding (three-stage config), else drain to zero.
    const int nk = p.nk;
    constexpr int D = NST - 1, G = GA + GB;
    static_assert(D <= 3, "wait_stage_barrier handles up to two stages in flight behind the awaited one");
    if constexpr (CF::PIPE) {
        // two 128-B stages, fragments read one k-step ahead; the stage hand-over sits between the two MFMA blocks
        static_assert(RB == 128 && NST == 2 && sizeof(T) == 2, "pipelined loop: bf16, two 128-B stages");
        uint4 a0[MI], b0[4], a1[MI], b1[4];
        auto read_frags = [&](int kt, int ks, uint4 (&a)[MI], uint4 (&b)[4]) __attribute__((always_inline)) {
            const char *sA = lds + (kt & 1) * CF::STAGE;
            const int sw = ((fq + 4 * ks) ^ (lane & 7)) << 4;
            const char *A = sA + (wm * 16 * MI + frow) * RB + sw, *B = sA + CF::STAGE_A + (wn * 64 + frow) * RB + sw;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfmas = [&](const uint4 (&a)[MI], const uint4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);   // transposed tile: see ig_epilogue
            __builtin_amdgcn_sched_barrier(0);
        };
        stage();
        wait_vm_barrier<0>();
        if (nk > 1) stage();
        read_frags(0, 0, a0, b0);
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            read_frags(kt, 1, a1, b1);
            mfmas(a0, b0);
            if (kt + 1 < nk) {
                wait_vm_barrier<0>();        // this stage's fragments are in registers, stage kt+1 has landed
                if (kt + 2 < nk) stage();    // into the stage just released
                read_frags(kt + 1, 0, a0, b0);
            }
            mfmas(a1, b1);
        }
    } else {
    int issued = 0;
    for (; issued < D && issued < nk; ++issued) stage();
    wait_stage_barrier<G>(issued - 1);   // stage 0 has landed (for every wave, after the barrier)
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        if (issued < nk) { stage(); ++issued; }
        const char *sA = lds + (kt % NST) * CF::STAGE;
        ig_compute_stage<T, MI, RB, true>(sA, sA + CF::STAGE_A, wm, wn, lane, acc);
        if (kt + 1 < nk) wait_stage_barrier<G>(issued - (kt + 2));   // stage kt+1 landed; later ones may still fly
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done reading the stage buffers

    if (!(p.tune & 64)) ig_epilogue<T, MI, (CF::WPE > 2 ? 1 : 4)>(p, lds, acc, m0, n0, wm, wn, wv, lane);   // 64: timing ablation
}

// ---- 3x3 / stride 1 / 'same' convolutions whose 256-pixel tiles are segments of one image row -----------------------------
// The im2col gather stages the A tile once per tap: the three kx taps of a kernel row re-read the same image row shifted by
// +-dil pixels, so two thirds of the A traffic (L2 -> LDS bytes, DMA instructions, LDS writes) is redundant.  Here the A
// operand is a ROW BUFFER: per (channel block, ky) one segment of 256 + 2*dil pixels (halo included, zero outside the
// image) is staged once, and the three kx stages read their fragments from it at row offsets kx*dil; the XOR swizzle is
// keyed on the buffer row, so the shifted reads stay bank-conflict free.  The B tile is staged per tap as before.
// Kernel rows that fall outside the image contribute nothing and are skipped for the whole tile.
// Wide: 2 x 40 KiB row buffers (320 rows: dil <= 32) + 2 x 32 KiB B stages = 144 KiB.  Narrow (Cout <= 128, bf16): 64-B rows,
// 2 x 24 KiB + 2 x 8 KiB, two workgroups per CU.
template <int MI_, int WM_, int WN_, int RB_, int AROWS_, int NBS_, int WPE_, int NJ_ = 4, int PIPE_ = 0> struct CfgRowT {
    static constexpr int MI = MI_, WM = WM_, WN = WN_, RB = RB_, NBS = NBS_, WPE = WPE_, NJ = NJ_, PIPE = PIPE_;   // NBS: B stages resident; NJ: 16-column MFMA tiles per wave
    static constexpr int BM = WM * MI * 16, BN = WN * NJ * 16;
    static constexpr int PR = 1024 / RB;           // rows per 1-KiB LDS-DMA piece
    static constexpr int AROWS = AROWS_, ABUF = AROWS * RB, BSTAGE = BN * RB;
    static constexpr int NW = WM * WN;             // waves per workgroup
    static constexpr int GAR = AROWS / PR / NW;    // pieces per wave per row buffer
    static constexpr int GB = BN / PR / NW;        // ... per B stage
    static constexpr int MAXDIL = (AROWS - BM) / 2;
    static constexpr int NEED = 2 * ABUF + NBS * BSTAGE, EPI = NW * 32 * EP_LD * 4;
    static_assert(NW * 16 * MI * 128 <= NEED || NW * 16 * MI * 128 <= EPI, "bf16 epilogue patches must fit");
    static constexpr int LDS_BYTES = NEED > EPI ? NEED : EPI;   // the epilogue patches reuse the buffers
    static_assert(BM == 256, "256-pixel tiles");
    static_assert(AROWS % (PR * NW) == 0 && BN % (PR * NW) == 0, "whole pieces per wave");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
typedef CfgRowT<8, 2, 4, 128, 320, 2, 2, 4, 1> CfgRow;    // 256 x 256, 144 KiB: dil <= 32; software-pipelined main loop (bf16)
typedef CfgRowT<8, 2, 4, 128, 384, 2, 2, 4, 1> CfgRowX;   // 256 x 256, 160 KiB: dil <= 64 (ASPP rate 36)
typedef CfgRowT<8, 2, 4, 128, 320, 2, 2> CfgRowF;         // fp32 parity path: plain loop (blocked accumulation)
typedef CfgRowT<8, 2, 4, 128, 384, 2, 2> CfgRowXF;
typedef CfgRowT<8, 2, 2, 64, 320, 3, 2> CfgRowH;    // 256 x 128, FOUR waves of 128x64, 64-B K stages, 64 KiB: two workgroups per CU
typedef CfgRowT<8, 2, 2, 64, 384, 3, 2> CfgRowHX;   // ... dil <= 64, 72 KiB
typedef CfgRowT<4, 4, 2, 64, 384, 3, 4> CfgRowN;    // 256 x 128 with 64-B K stages, 72 KiB, <= 128 VGPRs: two workgroups per CU

template <typename T, typename CF>
__global__ __launch_bounds__(64 * CF::NW, CF::WPE) void conv_igemm_row_kernel(const ConvParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[CF::LDS_BYTES];
    constexpr int ES = sizeof(T), RB = CF::RB, BK = RB / ES, EPC = 16 / ES;
    constexpr int MI = CF::MI, NJ = CF::NJ, GAR = CF::GAR, GB = CF::GB, PR = CF::PR, CPR = RB / 16;
    char *const ldsB = lds + 2 * CF::ABUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / CF::WN, wn = wv % CF::WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * CF::BM, n0 = tn * CF::BN;
    const T *__restrict__ xg = (const T *)p.x;
    const T *__restrict__ wg = (const T *)p.w;
    const T *zero = (const T *)kd_zero_page;
    const int d = p.dil;

    // the tile: image n, output row ho, columns x0 .. x0 + 255 (W % 256 == 0, Ho == H, Wo == W)
    const int n = m0 / p.HoWo, rem = m0 - n * p.HoWo;
    const int ho = rem / p.W, x0 = rem - ho * p.W;
    // kernel rows inside the image (block-uniform)
    const int ky_lo = ho - d < 0 ? 1 : 0, ky_hi = ho + d >= p.H ? 1 : 2;
    const int nky = ky_hi - ky_lo + 1;

    // ---- per-lane staging state -------------------------------------------------------------------------------------
    // source-side swizzle: LDS slot s of buffer row r holds chunk s ^ (r & 7) (128-B rows) / s ^ ((r >> 1) & 3) (64-B rows);
    // pieces start at multiples of PR rows, so the key is a function of the lane's row inside the piece
    const int srow = lane / CPR;
    const int chunk = RB == 128 ? ((lane & 7) ^ srow) : ((lane & 3) ^ ((srow >> 1) & 3));
    int a_off[GAR];                                  // element offset of (n, ho, x) + chunk for the lane's row of piece j
    uint32_t a_ok = 0;                               // bit j: that pixel is inside the image row and the buffer
#pragma unroll
    for (int j = 0; j < GAR; ++j) {
        const int r = (wv * GAR + j) * PR + srow;    // buffer row: pixel x0 - d + r
        const int x = x0 - d + r;
        const bool ok = r < CF::BM + 2 * d && x >= 0 && x < p.W;
        a_off[j] = ((n * p.H + ho) * p.W + (ok ? x : 0)) * p.ldx + chunk * EPC;
        a_ok |= ok ? (1u << j) : 0u;
    }
    int b_off[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int nn = n0 + (wv * GB + j) * PR + srow;
        b_off[j] = nn < p.Cout ? nn * p.Ktot + chunk * EPC : -1;
    }

    // stage order: channel block outer, ky, kx inner (taps of one channel block back to back: L2-resident)
    int u_cb = 0, u_ky = ky_lo, u_idx = 0;           // next row buffer to issue
    auto stage_a = [&]() {
        const int row_off = ((u_ky - 1) * d * p.W) * p.ldx + u_cb * BK;
        char *la = lds + (u_idx & 1) * CF::ABUF + wv * (GAR * 1024);
#pragma unroll
        for (int j = 0; j < GAR; ++j) {
            const T *src = ((a_ok >> j) & 1u) ? xg + (a_off[j] + row_off) : zero;
            glds16(src, la + j * 1024);
        }
        ++u_idx;
        if (++u_ky > ky_hi) { u_ky = ky_lo; ++u_cb; }
        __builtin_amdgcn_sched_barrier(0);
    };
    int s_cb = 0, s_ky = ky_lo, s_kx = 0, s_idx = 0; // next B stage to issue
    auto stage_b = [&]() {
        const int w_off = (s_ky * 3 + s_kx) * p.Cin + s_cb * BK;
        char *lb = ldsB + (s_idx % CF::NBS) * CF::BSTAGE + wv * (GB * 1024);
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const T *src = b_off[j] >= 0 ? wg + (b_off[j] + w_off) : zero;
            glds16(src, lb + j * 1024);
        }
        ++s_idx;
        if (++s_kx == 3) {
            s_kx = 0;
            if (++s_ky > ky_hi) { s_ky = ky_lo; ++s_cb; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // B stages run NBS - 1 ahead of the one being consumed.  vmcnt completes in issue order; with three B stages the
    // operations younger than B(s+1) at the end of stage s are the ones issued in stage s itself (the next row buffer on
    // kx == 0, then B(s+2)), so "all but those" == B(s+1) -- and every older row buffer -- has landed.
    const int nu = p.nkc * nky, ns = nu * 3;
    constexpr int NBS = CF::NBS;
    stage_a();
    stage_b();
    if (NBS == 3 && ns > 1) stage_b();
    wait_vm_barrier<0>();
    if constexpr (CF::PIPE) {
        // Software-pipelined main loop (128-B stages = two k-steps, two B stages): the fragments of a k-step are read
        // one k-step ahead into a second register set, and the stage hand-over (wait + barrier + the DMA of stage s+2 +
        // the first fragment reads of stage s+1) sits between the two k-steps' MFMA blocks -- the second block's operands
        // are already in registers, so its 16*MI*NJ/16 MFMAs cover the barrier and the LDS latency of the next reads.
        static_assert(RB == 128 && NBS == 2 && sizeof(T) == 2, "pipelined loop: bf16, 128-B stages, two B stages");
        uint4 a0[MI], b0[NJ], a1[MI], b1[NJ];
        auto read_frags = [&](int s, int ks, uint4 (&a)[MI], uint4 (&b)[NJ]) __attribute__((always_inline)) {
            const int u = s / 3, kx = s - u * 3;
            const int rsh = frow + kx * d;
            const char *A = lds + (u & 1) * CF::ABUF + (wm * (16 * MI) + rsh) * RB + (((fq + 4 * ks) ^ (rsh & 7)) << 4);
            const char *B = ldsB + (s & 1) * CF::BSTAGE + (wn * (16 * NJ) + frow) * RB + (((fq + 4 * ks) ^ (lane & 7)) << 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfmas = [&](const uint4 (&a)[MI], const uint4 (&b)[NJ]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);   // transposed tile: see ig_epilogue
            __builtin_amdgcn_sched_barrier(0);
        };
        if (ns > 1) stage_b();   // B(1)
        if (nu > 1) stage_a();   // row buffer 1
        read_frags(0, 0, a0, b0);
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            read_frags(s, 1, a1, b1);
            mfmas(a0, b0);
            if (s + 1 < ns) {
                wait_vm_barrier<0>();   // this stage's reads are in registers, stage s+1 (and its row buffer) has landed
                if (s + 2 < ns) stage_b();   // into the B stage just released
                if ((s + 1) % 3 == 0 && (s + 1) / 3 + 1 < nu) stage_a();   // into the row buffer just released
                read_frags(s + 1, 0, a0, b0);
            }
            mfmas(a1, b1);
        }
    } else {
#pragma unroll 1
    for (int u = 0; u < nu; ++u) {
        const char *Au = lds + (u & 1) * CF::ABUF;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            const int s = u * 3 + kx;
            // the other row buffer was last read in the previous super-stage, which every wave left at a barrier
            const bool new_a = kx == 0 && u + 1 < nu, new_b = s + NBS - 1 < ns;
            if (NBS == 2 && !(p.tune & 1)) {
                // two B stages: B(s+1) first, then the next row buffer, which may then stay in flight across this stage's
                // wait (it is first read two stages later; the wait of stage s+1 covers it)
                if (new_b) stage_b();
                if (new_a) stage_a();
            } else {
                if (new_a) stage_a();
                if (new_b) stage_b();
            }
            // fragments: A rows shifted by kx*dil inside the row buffer, B from the per-tap stage
            const int rsh = frow + kx * d;
            const char *A = Au + (wm * (16 * MI) + rsh) * RB;
            const char *B = ldsB + (s % NBS) * CF::BSTAGE + (wn * (16 * NJ) + frow) * RB;
#pragma unroll
            for (int ks = 0; ks < RB / 64; ++ks) {
                const int swa = RB == 128 ? (((fq + 4 * ks) ^ (rsh & 7)) << 4) : ((fq ^ ((rsh >> 1) & 3)) << 4);
                const int swb = RB == 128 ? (((fq + 4 * ks) ^ (lane & 7)) << 4) : ((fq ^ ((frow >> 1) & 3)) << 4);
                uint4 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB + swa);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB + swb);
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        f32x4_t part[NJ];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            part[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                            Mma<T>::run(b[j], a[i], part[j]);
                            acc[i][j] += part[j];
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);   // transposed tile: see ig_epilogue
                }
                __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ, 0);
                if constexpr (sizeof(T) != 4) __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
            }
            if (s + 1 < ns) {
                if (NBS == 3 && new_b) {
                    if (new_a) wait_vm_barrier<GAR + GB>();
                    else wait_vm_barrier<GB>();
                } else if (NBS == 2 && new_a && new_b && !(p.tune & 1)) {
                    wait_vm_barrier<GAR>();
                } else {
                    wait_vm_barrier<0>();
                }
            }
        }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done reading the buffers
    if (!(p.tune & 64)) ig_epilogue<T, MI, (CF::WPE > 2 ? 1 : 4), NJ>(p, lds, acc, m0, n0, wm, wn, wv, lane);
}

// ---- persistent bf16 kernels: one workgroup per CU walks its XCD's tiles -----------------------------------------------------
// Measured on the one-tile-per-workgroup kernels above (KDCC_CONV_TUNE=64/128 ablations, r02): the epilogue is 21 % of the
// conv time of a train step, more than half of that the output stores themselves -- every CU reaches its epilogue at the
// same moment, the burst drains at the fabric's ~5 TB/s with nothing else running, and a workgroup cannot retire (nor its
// successor start: LDS) before its stores are acknowledged.  Two workgroups per CU (256 x 128 tiles) hide that but lose more
// in the main loop (1.5x the staging traffic): -10 %.  Here instead the workgroup stays resident and walks tiles:
//   * the epilogue transposes through a small wave-private LDS patch that lies outside the stage buffers (16 KiB in all);
//   * the stage buffers are therefore free when the main loop ends: the first stages of the NEXT tile are issued before the epilogue, so
//     its pipeline fill overlaps the epilogue, and -- vmcnt retiring in issue order -- those DMAs are OLDER than the
//     epilogue's stores: the next main loop starts after a counted wait that leaves exactly the stores outstanding, which
//     then drain under its first two K stages instead of in front of an idle CU.
// Conditions (host side): bf16 in and out, M % 256 == 0, Cout % BN == 0, 16-B friendly epilogue operands, dil <= 32.

template <typename CF, int NOPS, bool DBG = false, bool PP = false>   // PP: see conv_igemm_persist_kernel
__global__ __launch_bounds__(64 * CF::NW, CF::WPE) void conv_row_persist_kernel(const ConvParams p)
{
    typedef bf16_t T;
    static_assert(CF::PIPE && CF::RB == 128 && CF::NBS == 2, "bf16 pipelined configuration");
    static_assert(CF::NEED + CF::NW * 2048 <= 160 * 1024, "stage buffers + epilogue patches");
    __shared__ __attribute__((aligned(16))) char lds[CF::NEED + CF::NW * 2048];
    constexpr int RB = CF::RB, BK = RB / 2, EPC = 8;
    constexpr int MI = CF::MI, NJ = CF::NJ, GAR = CF::GAR, GB = CF::GB, PR = CF::PR, CPR = RB / 16;
    char *const ldsB = lds + 2 * CF::ABUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / CF::WN, wn = wv % CF::WN;
    const T *__restrict__ xg = (const T *)p.x;
    const T *__restrict__ wg = (const T *)p.w;
    const T *zero = (const T *)kd_zero_page;
    const int d = p.dil;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);

    const int srow = lane / CPR;
    const int chunk = (lane & 7) ^ srow;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- per-tile staging state (see conv_igemm_row_kernel) ------------------------------------------------------------------
    int m0 = 0, n0 = 0, ky_lo = 0, ky_hi = 0, nu = 0, ns = 0;
    // Every wave stages its own GAR / GB pieces of a stage, but the two waves of a SIMD (wv and wv + NW/2: a workgroup's waves
    // go to the SIMDs cyclically) do so at different points of the stage.  Phase clocks (KDCC_CONV_TUNE=512,
    // tools/conv_timeline.py) show the hand-over as the main loop's loss: the ~45 KiB of a stage are 45 LDS-DMA instructions
    // on the CU's one vector-memory path, each holds its wave for 100-160 cycles while the others queue, and with all eight
    // waves issuing right after the barrier no MFMA is issued for 620-820 cycles of a 3150-cycle stage (its MFMAs need 2048).
    // The "early" waves issue after the barrier, the "late" ones after their second MFMA block, so one wave of each SIMD has
    // MFMAs to issue meanwhile: -1.4 % step time.  (Pieces placed between the MFMAs of the block cost 200+ cycles each:
    // 3150 -> 4000 cycles per stage; one wave per SIMD issuing everything: 3060.)
    // Piece j of a wave lies j * PR rows after its piece 0: no offset arrays; the pieces of a group share one M0 value.
    constexpr int GAR2 = GAR, GB2 = GB;
    const bool early = (p.tune & 2048) ? true : wv < CF::NW / 2;   // 2048: A/B, every wave issues right after the barrier
    int a_off0 = 0, b_off0 = 0;
    uint32_t a_ok = 0;
    int u_cb = 0, u_ky = 0, u_idx = 0, s_cb = 0, s_ky = 0, s_kx = 0, s_idx = 0;
    auto setup = [&](int tile) {
        int tn, tm;
        if (p.tn_group > 0) {   // tn-blocked order: p.tn_group N tiles x all M tiles, then the next N block
            const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
            tm = r / p.tn_group;
            tn = blk * p.tn_group + (r - tm * p.tn_group);
        } else {
            tn = tile % p.tiles_n;
            tm = tile / p.tiles_n;
        }
        m0 = tm * CF::BM;
        n0 = tn * CF::BN;
        const int n = m0 / p.HoWo, rem = m0 - n * p.HoWo;
        const int ho = rem / p.W, x0 = rem - ho * p.W;
        ky_lo = ho - d < 0 ? 1 : 0;
        ky_hi = ho + d >= p.H ? 1 : 2;
        nu = p.nkc * (ky_hi - ky_lo + 1);
        ns = nu * 3;
        a_ok = 0;
        // PP: group 0 (waves 0-3) stages the row-buffer rows it reads, 0 .. 191 (128 + 2 * dil <= 192), 6 pieces each, and all of
        // B, 8 pieces each; group 1 stages rows 192 .. 319, 4 pieces each
        const int pbase = PP ? (wv < 4 ? wv * 6 : 24 + (wv - 4) * 4) : wv * GAR2;
        const int r0 = pbase * PR + srow;                // buffer row of piece 0: pixel x0 - d + r0
#pragma unroll
        for (int j = 0; j < (PP ? 6 : GAR2); ++j) {
            const int r = r0 + j * PR, x = x0 - d + r;
            a_ok |= (r < CF::BM + 2 * d && x >= 0 && x < p.W) ? (1u << j) : 0u;
        }
        a_off0 = ((n * p.H + ho) * p.W + (x0 - d + r0)) * p.ldx + chunk * EPC;   // may point before the row: masked by a_ok
        b_off0 = (n0 + wv * (PP ? 2 : 1) * GB2 * PR + srow) * p.Ktot + chunk * EPC;
        u_cb = 0; u_ky = ky_lo; u_idx = 0;
        s_cb = 0; s_ky = ky_lo; s_kx = 0; s_idx = 0;
    };
    auto stage_a = [&]() {
        const int row_off = ((u_ky - 1) * d * p.W) * p.ldx + u_cb * BK;
        static_assert(GAR2 == 5 && GB2 == 4, "piece groups below");
        if constexpr (PP) {
            // (pieces that lie wholly behind the BM + 2 dil rows the fragments read are not issued: 7 of the 40 at dil = 1)
            const int pbase = wv < 4 ? wv * 6 : 24 + (wv - 4) * 4,
                      np = (p.tune & 4096) ? (wv < 4 ? 6 : 4) : min(wv < 4 ? 6 : 4, (CF::BM + 2 * d + PR - 1) / PR - pbase);   // 4096: A/B, all 40
            char *la = lds + (u_idx & 1) * CF::ABUF + pbase * 1024;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < np) glds16(((a_ok >> j) & 1u) ? xg + (a_off0 + j * PR * p.ldx + row_off) : zero, la + j * 1024);
        } else {
            char *la = lds + (u_idx & 1) * CF::ABUF + wv * (GAR2 * 1024);
            const char *ga[GAR2];
#pragma unroll
            for (int j = 0; j < GAR2; ++j)   // source of piece j, reduced by the instruction offset of that piece
                ga[j] = (const char *)(((a_ok >> j) & 1u) ? xg + (a_off0 + j * PR * p.ldx + row_off) : zero) - (j - 2) * 1024;
            glds16_x5(ga[0], ga[1], ga[2], ga[3], ga[4], la + 2048);
        }
        ++u_idx;
        if (++u_ky > ky_hi) { u_ky = ky_lo; ++u_cb; }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_b = [&]() {
        const int w_off = (s_ky * 3 + s_kx) * p.Cin + s_cb * BK;
        const char *gb[GB2];
        if constexpr (PP) {
            if (wv < 4) {
                char *lb = ldsB + (s_idx & 1) * CF::BSTAGE + wv * (2 * GB2 * 1024);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int j = 0; j < GB2; ++j)
                        gb[j] = (const char *)(wg + (b_off0 + (h2 * GB2 + j) * PR * p.Ktot + w_off)) - j * 1024;
                    glds16_x4(gb[0], gb[1], gb[2], gb[3], lb + h2 * GB2 * 1024);
                }
            }
        } else {
            char *lb = ldsB + (s_idx & 1) * CF::BSTAGE + wv * (GB2 * 1024);
#pragma unroll
            for (int j = 0; j < GB2; ++j) gb[j] = (const char *)(wg + (b_off0 + j * PR * p.Ktot + w_off)) - j * 1024;
            glds16_x4(gb[0], gb[1], gb[2], gb[3], lb);
        }
        ++s_idx;
        if (++s_kx == 3) {
            s_kx = 0;
            if (++s_ky > ky_hi) { s_ky = ky_lo; ++s_cb; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prologue = [&]() {   // both row buffers and both B stages are free
        stage_a();
        stage_b();
        stage_b();            // ns >= 6
        if (nu > 1) stage_a();
    };

    f32x4_t acc[MI][NJ];
    uint4 a0[MI], b0[NJ], a1[MI], b1[NJ];
    auto read_frags = [&](int s, int ks, uint4 (&a)[MI], uint4 (&b)[NJ]) __attribute__((always_inline)) {
        const int u = s / 3, kx = s - u * 3;
        const int rsh = frow + kx * d;
        const char *A = lds + (u & 1) * CF::ABUF + (wm * (16 * MI) + rsh) * RB + (((fq + 4 * ks) ^ (rsh & 7)) << 4);
        const char *B = ldsB + (s & 1) * CF::BSTAGE + (wn * (16 * NJ) + frow) * RB + (((fq + 4 * ks) ^ (lane & 7)) << 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](const uint4 (&a)[MI], const uint4 (&b)[NJ]) __attribute__((always_inline)) {
        // the multiplying wave issues ahead of its SIMD partner (which stages / reads / stores): 3x3 class 101.0 -> 100.0 ms,
        // 1x1 class 38.2 -> 38.0 ms per step, alternating runs; KDCC_CONV_TUNE=8192 (tuning build) leaves the priorities alone
        if (!(p.tune & 8192)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);
        if (!(p.tune & 8192)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nst_epi = 2 * MI * ((p.ep.out_raw ? 1 : 0) + (p.ep.out_act ? 1 : 0)) + ((NOPS & 4) ? 4 : 0);
    int nst = 0;   // stores issued after the pending prologue
    int tcount = 0;
    setup(walk.t);
    prologue();
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bool tl = (p.tune & 512) && tid == 0 && tcount < 32;
        unsigned long long *tlp = kd_conv_tlog + ((size_t)blockIdx.x * 32 + (tcount & 31)) * 8;
        if (tl) tlp[0] = wall_clock64();
        wait_vm_stores(nst);
        if (tl) tlp[1] = wall_clock64();   // the prologue has landed for every wave; the previous tile's stores may still drain
        unsigned long long cprev = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, ph5 = 0;
        if constexpr (PP) {
            // roles and barriers as in conv_igemm_persist_kernel<.., PP>: group 0 computes stage sc in half-period 2*sc, group 1
            // in 2*sc + 1; row buffer u + 2 replaces u once group 1 has read stage 3*u + 2
            // (DBG: phase clocks per wave -- group 0: reads | barrier | MFMAs | DMA wait | barrier | DMA issue;
            //  group 1: DMA wait | barrier | DMA issue | reads | barrier | MFMAs)
            if (wv < 4) {
#pragma unroll 1
                for (int sc = 0; sc < ns; ++sc) {
                    if (DBG) cprev = clock64();
                    read_frags(sc, 0, a0, b0);
                    read_frags(sc, 1, a1, b1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (DBG) c1 = clock64();
                    __builtin_amdgcn_s_barrier();          // half-period 2*sc
                    if (DBG) c2 = clock64();
                    mfmas(a0, b0);
                    mfmas(a1, b1);
                    if (DBG) c3 = clock64();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage sc + 1 (issued a stage ago)
                    if (DBG) c4 = clock64();
                    __builtin_amdgcn_s_barrier();          // half-period 2*sc + 1: group 1 has read stage sc
                    if (DBG) c5 = clock64();
                    if (sc + 2 < ns) stage_b();
                    if (sc % 3 == 2 && sc / 3 + 2 < nu) stage_a();
                    if (DBG) { const unsigned long long c6 = clock64(); ph0 += c1 - cprev; ph1 += c2 - c1; ph2 += c3 - c2; ph3 += c4 - c3; ph4 += c5 - c4; ph5 += c6 - c5; }
                }
            } else {
#pragma unroll 1
                for (int sc = 0; sc < ns; ++sc) {
                    if (DBG) cprev = clock64();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's row-buffer rows (issued >= a stage ago)
                    if (DBG) c1 = clock64();
                    __builtin_amdgcn_s_barrier();          // half-period 2*sc
                    if (DBG) c2 = clock64();
                    if (sc % 3 == 0 && sc >= 3 && sc / 3 + 1 < nu) stage_a();   // rows 192.. of row buffer sc/3 + 1
                    if (sc + 2 < ns) stage_b();            // bookkeeping only (group 0 stages B)
                    if (DBG) c3 = clock64();
                    read_frags(sc, 0, a0, b0);
                    read_frags(sc, 1, a1, b1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (DBG) c4 = clock64();
                    __builtin_amdgcn_s_barrier();          // half-period 2*sc + 1
                    if (DBG) c5 = clock64();
                    mfmas(a0, b0);
                    mfmas(a1, b1);
                    if (DBG) { const unsigned long long c6 = clock64(); ph0 += c1 - cprev; ph1 += c2 - c1; ph2 += c3 - c2; ph3 += c4 - c3; ph4 += c5 - c4; ph5 += c6 - c5; }
                }
            }
        } else {
        read_frags(0, 0, a0, b0);
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            if (DBG && s == 1) cprev = clock64();
            read_frags(s, 1, a1, b1);
            mfmas(a0, b0);
            if (s + 1 < ns) {
                // stage s+1 (and its row buffer) has landed: for s == 0 the prologue wait covered it, and a vmcnt(0) here
                // would wait for the previous tile's stores
                if (s == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                else if (DBG) {   // debug: phase clocks (shader cycles) of this wave, see tools/conv_timeline.py
                    c1 = clock64();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    c2 = clock64();
                    __builtin_amdgcn_s_barrier();
                    c3 = clock64();
                }
                else wait_vm_barrier<0>();
                if (early) {
                    if (s + 2 < ns) stage_b();
                    if ((s + 1) % 3 == 0 && (s + 1) / 3 + 1 < nu) stage_a();
                }
                if (DBG) c4 = clock64();
                read_frags(s + 1, 0, a0, b0);
                if (DBG) c5 = clock64();
            }
            mfmas(a1, b1);
            if (!early && s + 1 < ns) {
                if (s + 2 < ns) stage_b();
                if ((s + 1) % 3 == 0 && (s + 1) / 3 + 1 < nu) stage_a();
            }
            if (DBG && s > 0 && s + 1 < ns) {
                const unsigned long long c6 = clock64();
                ph0 += c1 - cprev; ph1 += c2 - c1; ph2 += c3 - c2; ph3 += c4 - c3; ph4 += c5 - c4; ph5 += c6 - c5;
                cprev = c6;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave is done reading the buffers
        }
        const int mw = m0 + wm * (16 * MI), nw = n0 + wn * (16 * NJ);
        walk.t += walk.step;
        const bool more = walk.t < walk.t_end;
        if (tl) tlp[4] = wall_clock64();
        if (DBG && lane == 0 && tcount == 1) {   // per-wave phase totals of the second tile
            unsigned long long *o = kd_conv_tlog + 256 * 32 * 8 + (blockIdx.x * 8 + wv) * 8;
            o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = ph3; o[4] = ph4; o[5] = ph5;
        }
        if (more) { setup(walk.t); if (tl) tlp[7] = wall_clock64(); prologue(); }
        if (tl) tlp[5] = wall_clock64();
        if (!(p.tune & 64)) ig_epilogue_rows16<MI, NOPS>(p, lds + CF::NEED + wv * 2048, acc, mw, nw, lane);
        if (tl) tlp[6] = wall_clock64();
        ++tcount;
        if (!more) break;
        nst = (p.tune & 64) ? 0 : nst_epi;
    }
}

// PP ("ping-pong"): the two waves of a SIMD run half a stage out of phase -- while one issues its 64 MFMAs of a stage, the other
// does everything else (its share of the next stage's DMA, then the 24 fragment reads of the stage it computes next), and they
// swap behind ONE barrier per half stage.  Measured (tools/ubench/mfma_issue.hip): a wave alone on its SIMD issues MFMAs at
// 17.1-17.3 cycles each -- one wave saturates the matrix pipe -- whether or not its sibling streams LDS-DMA or ds_reads; with
// both waves in their MFMA blocks the older one takes the pipe (17 vs 33 cycles per MFMA).  In lock-step both waves of a SIMD
// leave the pipe idle during the hand-over (barrier, ~45 DMA pieces through the CU's one vector-memory path, fragment reads:
// ~1100 of a 3100-cycle stage).
template <typename CF, int NOPS, bool PP = false>
__global__ __launch_bounds__(64 * CF::NW, CF::WPE) void conv_igemm_persist_kernel(const ConvParams p)
{
    typedef bf16_t T;
    static_assert(CF::PIPE && CF::RB == 128 && CF::NST == 2, "bf16 pipelined configuration");
    static_assert(CF::LDS_BYTES + CF::NW * 4096 <= 160 * 1024, "stage buffers + epilogue patches");
    __shared__ __attribute__((aligned(16))) char lds[CF::LDS_BYTES + CF::NW * 4096];
    constexpr int RB = CF::RB, BK = RB / 2, EPC = 8;
    constexpr int MI = CF::MI, GA = CF::GA, GB = CF::GB, PR = CF::PR, CPR = RB / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / CF::WN, wn = wv % CF::WN;
    const T *__restrict__ xg = (const T *)p.x;
    const T *__restrict__ wg = (const T *)p.w;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);

    const int srow = lane / CPR;
    const int chunk = (lane & 7) ^ srow;
    const int frow = lane & 15, fq = lane >> 4;
    const bool early = (p.tune & 2048) ? true : wv < CF::NW / 2;   // see conv_row_persist_kernel

    // 1x1 / stride 1 / no padding only (host): output pixel m is input pixel m, one tap; piece j of a wave lies j * PR rows
    // after its piece 0
    int m0 = 0, n0 = 0, a_off0 = 0, b_off0 = 0, s_kt = 0;
    auto setup = [&](int tile) {
        int tn, tm;
        if (p.tn_group > 0) {   // tn-blocked order: p.tn_group N tiles x all M tiles, then the next N block
            const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
            tm = r / p.tn_group;
            tn = blk * p.tn_group + (r - tm * p.tn_group);
        } else {
            tn = tile % p.tiles_n;
            tm = tile / p.tiles_n;
        }
        m0 = tm * CF::BM;
        n0 = tn * CF::BN;
        a_off0 = (m0 + wv * GA * PR + srow) * p.ldx + chunk * EPC;
        b_off0 = (n0 + wv * (PP ? 2 : 1) * GB * PR + srow) * p.Ktot + chunk * EPC;   // PP: waves 0-3 stage all of B, 64 rows each
        s_kt = 0;
    };
    auto stage = [&]() {
        char *la = lds + (s_kt & 1) * CF::STAGE + wv * (GA * 1024);
        static_assert(GA == 4 && GB == 4, "piece groups below");
        const char *ga[GA], *gb[GB];   // sources reduced by the instruction offset of the piece (one M0 value per group)
#pragma unroll
        for (int j = 0; j < GA; ++j) ga[j] = (const char *)(xg + (a_off0 + j * PR * p.ldx + s_kt * BK)) - j * 1024;
        glds16_x4(ga[0], ga[1], ga[2], ga[3], la);
        if constexpr (PP) {
            // the rows only this wave's group reads (A: rows wv*32 ..) come from every wave; B, which both groups read, from
            // group 0 alone, whose pieces have three half-periods to land and are published by its own counted wait
            if (wv < 4) {
                char *lb = lds + (s_kt & 1) * CF::STAGE + CF::STAGE_A + wv * (2 * GB * 1024);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int j = 0; j < GB; ++j)
                        gb[j] = (const char *)(wg + (b_off0 + (h2 * GB + j) * PR * p.Ktot + s_kt * BK)) - j * 1024;
                    glds16_x4(gb[0], gb[1], gb[2], gb[3], lb + h2 * GB * 1024);
                }
            }
        } else {
            char *lb = lds + (s_kt & 1) * CF::STAGE + CF::STAGE_A + wv * (GB * 1024);
#pragma unroll
            for (int j = 0; j < GB; ++j) gb[j] = (const char *)(wg + (b_off0 + j * PR * p.Ktot + s_kt * BK)) - j * 1024;
            glds16_x4(gb[0], gb[1], gb[2], gb[3], lb);
        }
        ++s_kt;
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nk = p.nk;
    auto prologue = [&]() {
        stage();
        if (nk > 1) stage();
    };

    f32x4_t acc[MI][4];
    uint4 a0[MI], b0[4], a1[MI], b1[4];
    auto read_frags = [&](int kt, int ks, uint4 (&a)[MI], uint4 (&b)[4]) __attribute__((always_inline)) {
        const char *sA = lds + (kt & 1) * CF::STAGE;
        const int sw = ((fq + 4 * ks) ^ (lane & 7)) << 4;
        const char *A = sA + (wm * 16 * MI + frow) * RB + sw, *B = sA + CF::STAGE_A + (wn * 64 + frow) * RB + sw;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *(const uint4 *)(A + i * 16 * RB);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const uint4 *)(B + j * 16 * RB);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](const uint4 (&a)[MI], const uint4 (&b)[4]) __attribute__((always_inline)) {
        if (!(p.tune & 8192)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);
        if (!(p.tune & 8192)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nst_epi = 2 * MI * ((p.ep.out_raw ? 1 : 0) + (p.ep.out_act ? 1 : 0)) + ((NOPS & 12) ? 4 : 0);
    if constexpr (PP) {
        // Ping-pong with seamless tile transitions: the staging runs ahead of the compute ACROSS tiles -- group 0 always issues
        // the stage two ahead of the one it computes, group 1 its rows of the stage one ahead, whichever tile that stage
        // belongs to -- so there is no per-tile pipeline fill and no barrier at the top of a tile; a tile boundary is just the
        // epilogue between two stages.  Stage buffers alternate over the global stage count.
        auto offsets = [&](int tile, int &tm0, int &tn0, int &ao, int &bo) {
            int tn, tm;
            if (p.tn_group > 0) {
                const int per = p.tiles_m * p.tn_group, blk = tile / per, r = tile - blk * per;
                tm = r / p.tn_group;
                tn = blk * p.tn_group + (r - tm * p.tn_group);
            } else {
                tn = tile % p.tiles_n;
                tm = tile / p.tiles_n;
            }
            tm0 = tm * CF::BM;
            tn0 = tn * CF::BN;
            ao = tm0 + wv * GA * PR + srow;          // this lane's row of piece 0 (times the source's pixel stride in issue())
            bo = (tn0 + wv * 2 * GB * PR + srow) * p.Ktot + chunk * EPC;
        };
        // issue iterator: (tile, k stage) of the next stage this wave stages, and the global count of stages it has staged
        int it_tile = walk.t, it_k = 0, it_g = 0, it_ao = 0, it_bo = 0, it_m0, it_n0;
        offsets(it_tile, it_m0, it_n0, it_ao, it_bo);
        const T *__restrict__ xg2 = (const T *)p.x2;
        auto issue = [&]() {
            if (it_tile >= walk.t_end) return;
            char *la = lds + (it_g & 1) * CF::STAGE + wv * (GA * 1024);
            const char *ga[GA], *gb[GB];
            // K-concatenated 1x1 (two A sources, kd_conv1x1_dual_fwd): stages >= nk1 come from x2; the choice is wave-uniform
            const bool second = it_k >= p.nk1;
            const T *src = second ? xg2 : xg;
            const int ldr = second ? p.ldx2 : p.ldx, ks = second ? it_k - p.nk1 : it_k;
            const int a0 = it_ao * ldr + chunk * EPC + ks * BK;
#pragma unroll
            for (int j = 0; j < GA; ++j) ga[j] = (const char *)(src + (a0 + j * PR * ldr)) - j * 1024;
            glds16_x4(ga[0], ga[1], ga[2], ga[3], la);
            if (wv < 4) {   // B, which both groups read, comes from group 0 alone (published by its wait a full stage early)
                char *lb = lds + (it_g & 1) * CF::STAGE + CF::STAGE_A + wv * (2 * GB * 1024);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int j = 0; j < GB; ++j)
                        gb[j] = (const char *)(wg + (it_bo + (h2 * GB + j) * PR * p.Ktot + it_k * BK)) - j * 1024;
                    glds16_x4(gb[0], gb[1], gb[2], gb[3], lb + h2 * GB * 1024);
                }
            }
            ++it_g;
            if (++it_k == nk) {
                it_k = 0;
                it_tile += walk.step;
                if (it_tile < walk.t_end) offsets(it_tile, it_m0, it_n0, it_ao, it_bo);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_all = [&](int g) __attribute__((always_inline)) {
            read_frags(g, 0, a0, b0);
            read_frags(g, 1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int g = 0;                       // global index of the stage being computed
        bool first = true;
        if (wv < 4) { issue(); issue(); } else { issue(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // the first stages have landed for every wave
#pragma unroll 1
        for (int tile = walk.t; tile < walk.t_end; tile += walk.step) {
            int a_dummy, b_dummy;
            offsets(tile, m0, n0, a_dummy, b_dummy);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (wv < 4) {
#pragma unroll 1
                for (int sc = 0; sc < nk; ++sc, ++g) {
                    read_all(g);                           // (stage g was published behind the previous barrier)
                    __builtin_amdgcn_s_barrier();          // half-period 2*g: group 1 may stage its rows of stage g + 1
                    mfmas(a0, b0);
                    mfmas(a1, b1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage g + 1 (issued a stage ago)
                    __builtin_amdgcn_s_barrier();          // half-period 2*g + 1: group 1 has read stage g
                    issue();                               // stage g + 2
                }
            } else {
#pragma unroll 1
                for (int sc = 0; sc < nk; ++sc, ++g) {
                    // this wave's A rows of stage g (issued a stage ago); right after an epilogue its stores may still drain
                    if (sc == 0 && !first) wait_vm_only((p.tune & 64) ? 0 : nst_epi);
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();          // half-period 2*g
                    issue();                               // A rows of stage g + 1
                    read_all(g);
                    __builtin_amdgcn_s_barrier();          // half-period 2*g + 1
                    mfmas(a0, b0);
                    mfmas(a1, b1);
                }
            }
            first = false;
            if (!(p.tune & 64)) ig_epilogue_rows16<MI, NOPS, 2>(p, lds + CF::LDS_BYTES + wv * 4096, acc, m0 + wm * (16 * MI), n0 + wn * 64, lane);
        }
        return;
    }
    int nst = 0, tcount = 0;
    setup(walk.t);
    prologue();
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bool tl = (p.tune & 512) && tid == 0 && tcount < 32;
        unsigned long long *tlp = kd_conv_tlog + ((size_t)blockIdx.x * 32 + (tcount & 31)) * 8;
        if (tl) tlp[0] = wall_clock64();
        wait_vm_stores(nst);
        if (tl) tlp[1] = wall_clock64();
        {
        read_frags(0, 0, a0, b0);
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            read_frags(kt, 1, a1, b1);
            mfmas(a0, b0);
            if (kt + 1 < nk) {
                if (kt == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                else wait_vm_barrier<0>();
                if (early && kt + 2 < nk) stage();
                read_frags(kt + 1, 0, a0, b0);
            }
            mfmas(a1, b1);
            if (!early && kt + 2 < nk) stage();   // the sibling wave of the SIMD issued its pieces before this block
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        }
        const int mw = m0 + wm * (16 * MI), nw = n0 + wn * 64;
        walk.t += walk.step;
        const bool more = walk.t < walk.t_end;
        if (tl) tlp[4] = wall_clock64();
        if (more) { setup(walk.t); if (tl) tlp[7] = wall_clock64(); prologue(); }
        if (tl) tlp[5] = wall_clock64();
        if (!(p.tune & 64)) ig_epilogue_rows16<MI, NOPS, 2>(p, lds + CF::LDS_BYTES + wv * 4096, acc, mw, nw, lane);
        if (tl) tlp[6] = wall_clock64();
        ++tcount;
        if (!more) break;
        nst = (p.tune & 64) ? 0 : nst_epi;
    }
}

// ---- ping-pong row kernel for Cout = 128 layers (mod2, the GSCNN shape stream): 512 pixels x 128 channels per workgroup ---------
// The 256 x 128 two-workgroups-per-CU config runs these layers at 0.36-0.39 of the MFMA peak (short K, 16 MFMAs per wave between
// barriers).  Here the tile is twice as long in M so that a wave again owns 128 x 64 outputs (the wide kernels' sub-tile and
// epilogue), waves 4 x 2, and the ping-pong schedule of conv_row_persist_kernel<.., PP> applies: group 0 = waves 0-3 (output
// pixels 0-255), group 1 = waves 4-7 (pixels 256-511).  64-B K stages (one MFMA k-step, 32 MFMAs per wave and half-period):
// two row buffers of 576 pixels x 64 B (dil <= 16) + two B stages of 128 x 64 B + the epilogue patches = 104 KiB.
// Group 0 stages row-buffer rows 0-319 (it reads 0 .. 255 + 2*dil) and B; group 1 rows 320-575.
template <int NOPS, bool DBG = false>
__global__ __launch_bounds__(512, 2) void conv_row_pp128_kernel(const ConvParams p)
{
    typedef bf16_t T;
    constexpr int RB = 64, BK = 32, EPC = 8, PR = 16, MI = 8;
    constexpr int BM = 512, BN = 128, AROWS = 576, ABUF = AROWS * RB, BSTAGE = BN * RB, NEED = 2 * ABUF + 2 * BSTAGE;
    static_assert(NEED + 8 * 8192 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char lds[NEED + 8 * 8192];
    char *const ldsB = lds + 2 * ABUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const T *__restrict__ xg = (const T *)p.x;
    const T *__restrict__ wg = (const T *)p.w;
    const T *zero = (const T *)kd_zero_page;
    const int d = p.dil;
    TileWalk walk(p.ntiles);
    if (walk.t >= walk.t_end) return;
    stagger_start(p);

    // 64-B rows: LDS slot s of row r holds chunk s ^ ((r >> 1) & 3); a piece = 16 rows, lane -> (row lane / 4, slot lane & 3)
    const int srow = lane >> 2;
    const int chunk = (lane & 3) ^ ((srow >> 1) & 3);
    const int frow = lane & 15, fq = lane >> 4;
    // row-buffer pieces of this wave (16 rows each); those wholly behind the BM + 2 dil rows that are read are not issued
    const int pbase = wv < 4 ? wv * 5 : 20 + (wv - 4) * 4, np = (p.tune & 4096) ? (wv < 4 ? 5 : 4) : min(wv < 4 ? 5 : 4, (BM + 2 * d + 15) / 16 - pbase);

    int m0 = 0, n0 = 0, ky_lo = 0, ky_hi = 0, nu = 0, ns = 0, a_off0 = 0, b_off0 = 0;
    uint32_t a_ok = 0;
    int u_cb = 0, u_ky = 0, u_idx = 0, s_cb = 0, s_ky = 0, s_kx = 0, s_idx = 0;
    auto setup = [&](int tile) {
        const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
        m0 = tm * BM;
        n0 = tn * BN;
        const int n = m0 / p.HoWo, rem = m0 - n * p.HoWo;
        const int ho = rem / p.W, x0 = rem - ho * p.W;
        ky_lo = ho - d < 0 ? 1 : 0;
        ky_hi = ho + d >= p.H ? 1 : 2;
        nu = p.nkc * (ky_hi - ky_lo + 1);
        ns = nu * 3;
        a_ok = 0;
        const int r0 = pbase * PR + srow;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int r = r0 + j * PR, x = x0 - d + r;
            a_ok |= (r < BM + 2 * d && x >= 0 && x < p.W) ? (1u << j) : 0u;
        }
        a_off0 = ((n * p.H + ho) * p.W + (x0 - d + r0)) * p.ldx + chunk * EPC;   // may point before the row: masked by a_ok
        b_off0 = (n0 + wv * 2 * PR + srow) * p.Ktot + chunk * EPC;               // group 0: 32 rows of B per wave
        u_cb = 0; u_ky = ky_lo; u_idx = 0;
        s_cb = 0; s_ky = ky_lo; s_kx = 0; s_idx = 0;
    };
    auto stage_a = [&]() {
        const int row_off = ((u_ky - 1) * d * p.W) * p.ldx + u_cb * BK;
        char *la = lds + (u_idx & 1) * ABUF + pbase * 1024;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j < np) glds16(((a_ok >> j) & 1u) ? xg + (a_off0 + j * PR * p.ldx + row_off) : zero, la + j * 1024);
        ++u_idx;
        if (++u_ky > ky_hi) { u_ky = ky_lo; ++u_cb; }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_b = [&]() {
        const int w_off = (s_ky * 3 + s_kx) * p.Cin + s_cb * BK;
        if (wv < 4) {
            char *lb = ldsB + (s_idx & 1) * BSTAGE + wv * 2048;
            glds16(wg + (b_off0 + w_off), lb);
            glds16(wg + (b_off0 + PR * p.Ktot + w_off), lb + 1024);
        }
        ++s_idx;
        if (++s_kx == 3) {
            s_kx = 0;
            if (++s_ky > ky_hi) { s_ky = ky_lo; ++s_cb; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prologue = [&]() {
        stage_a();
        stage_b();
        stage_b();            // ns >= 6
        if (nu > 1) stage_a();
    };

    f32x4_t acc[MI][4];
    uint4 a0[MI], b0[4];
    auto read_frags = [&](int s) __attribute__((always_inline)) {
        const int u = s / 3, kx = s - u * 3;
        const int rsh = frow + kx * d;
        const char *A = lds + (u & 1) * ABUF + (wm * 128 + rsh) * RB + ((fq ^ ((rsh >> 1) & 3)) << 4);
        const char *B = ldsB + (s & 1) * BSTAGE + (wn * 64 + frow) * RB + ((fq ^ ((frow >> 1) & 3)) << 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) a0[i] = *(const uint4 *)(A + i * 16 * RB);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *(const uint4 *)(B + j * 16 * RB);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(b0[j], a0[i], acc[i][j]);   // transposed tile: see ig_epilogue
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nst_epi = 2 * MI * ((p.ep.out_raw ? 1 : 0) + (p.ep.out_act ? 1 : 0)) + ((NOPS & 4) ? 4 : 0);
    int nst = 0, tcount = 0;
    setup(walk.t);
    prologue();
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        wait_vm_stores(nst);
        unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
        if (wv < 4) {
#pragma unroll 1
            for (int sc = 0; sc < ns; ++sc) {
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
                if (DBG) c0 = clock64();
                read_frags(sc);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (DBG) c1 = clock64();
                __builtin_amdgcn_s_barrier();          // half-period 2*sc
                if (DBG) c2 = clock64();
                mfmas();
                if (DBG) c3 = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage sc + 1 (issued a stage ago)
                if (DBG) c4 = clock64();
                __builtin_amdgcn_s_barrier();          // half-period 2*sc + 1: group 1 has read stage sc
                if (DBG) c5 = clock64();
                if (sc + 2 < ns) stage_b();
                if (sc % 3 == 2 && sc / 3 + 2 < nu) stage_a();
                if (DBG) { const unsigned long long c6 = clock64(); ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4; ph[5] += c6 - c5; }
            }
        } else {
#pragma unroll 1
            for (int sc = 0; sc < ns; ++sc) {
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
                if (DBG) c0 = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's row-buffer rows (issued >= a stage ago)
                if (DBG) c1 = clock64();
                __builtin_amdgcn_s_barrier();          // half-period 2*sc
                if (DBG) c2 = clock64();
                if (sc % 3 == 0 && sc >= 3 && sc / 3 + 1 < nu) stage_a();
                if (sc + 2 < ns) stage_b();            // bookkeeping only (group 0 stages B)
                if (DBG) c3 = clock64();
                read_frags(sc);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (DBG) c4 = clock64();
                __builtin_amdgcn_s_barrier();          // half-period 2*sc + 1
                if (DBG) c5 = clock64();
                mfmas();
                if (DBG) { const unsigned long long c6 = clock64(); ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4; ph[5] += c6 - c5; }
            }
        }
        if (DBG && lane == 0 && tcount == 1) {   // per-wave phase totals of the second tile (tools/conv_timeline.py)
            unsigned long long *o = kd_conv_tlog + 256 * 32 * 8 + (blockIdx.x * 8 + wv) * 8;
            for (int q = 0; q < 6; ++q) o[q] = ph[q];
            o[6] = ns;
        }
        ++tcount;
        const int mw = m0 + wm * 128, nw = n0 + wn * 64;
        walk.t += walk.step;
        const bool more = walk.t < walk.t_end;
        if (more) { setup(walk.t); prologue(); }
        if (!(p.tune & 64)) ig_epilogue_rows16<MI, NOPS, 4>(p, lds + NEED + wv * 8192, acc, mw, nw, lane);
        if (!more) break;
        nst = (p.tune & 64) ? 0 : nst_epi;
    }
}

// ---- weight packing ------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_weight_kernel(const float *__restrict__ src, T *__restrict__ dst, int mode, int Cout,
                                        int Cin, int kh, int kw, int cin_pad)
{
    // one thread per destination element
    const int taps = kh * kw;
    const size_t total = mode == KD_PACK_FWD ? (size_t)Cout * taps * cin_pad : (size_t)Cin * taps * Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (mode == KD_PACK_FWD) {
            const int ci = (int)(i % cin_pad);
            const size_t r = i / cin_pad;
            const int t = (int)(r % taps), co = (int)(r / taps);
            if (ci < Cin) v = src[((size_t)co * Cin + ci) * taps + t];
        } else {
            const int co = (int)(i % Cout);
            const size_t r = i / Cout;
            const int tf = (int)(r % taps), ci = (int)(r / taps);
            const int t = taps - 1 - tf;  // flip both spatial axes
            v = src[((size_t)co * Cin + ci) * taps + t];
        }
        Elem<T>::st(dst + i, v);
    }
}

}  // namespace

// conv_row_lw_kernel (conv_lw.hip: one wave per SIMD, hand-scheduled loop) instead of conv_row_persist_kernel<pp>; A/B: KDCC_CONV_LW=0
static bool lw_row()
{
    static int lw = -1;
    if (lw < 0) { const char *v = getenv("KDCC_CONV_LW"); lw = !(v && v[0] == '0'); }
    return lw != 0;
}

static bool pp_row()
{
    static int pp = -1;
    if (pp < 0) { const char *v = getenv("KDCC_CONV_PP"); pp = !(v && v[0] == '0'); }   // A/B: KDCC_CONV_PP=0 = lock-step waves
    return pp != 0;
}

// Which kernel kd_conv2d_fwd runs for a problem (shared with kd_conv2d_bn_sums_rows, which must predict it).
struct ConvSel {
    int cfg;   // 0 narrow2, 1 wide, 2 deep, 3 narrow (one workgroup per CU)
    bool norow, half, row_wide, row_x, row_narrow, vec_ok;
    int nops, ncu;
    bool use_row_persist, use_igemm_persist, use_pp128;
};
static ConvSel conv_select(const kd_conv_desc *d, const kd_conv_epilogue *ep, int tune)
{
    ConvSel c;
    const int es = kd_elem_size(d->dtype);
    const int M = d->N * d->Ho * d->Wo;
    auto ok = [&](const void *ptr, int ld, int esz) { return !ptr || (kd_aligned16(ptr) && (ld * esz) % 16 == 0); };
    c.vec_ok = ok(ep->res_pre, ep->ld_res_pre, es) && ok(ep->mask, ep->ld_mask, es) &&
               ok(ep->res_post, ep->ld_res_post, es) && ok(ep->out_raw, ep->ld_raw, ep->raw_f32 ? 4 : es) &&
               ok(ep->out_act, ep->ld_act, es);
    // wide tiles only when they still fill the chip (one workgroup per CU, 256 CUs); e.g. the ASPP 4096->256 1x1 at
    // 128x256 pixels would give 128 wide tiles, so it runs on the narrow config (256 tiles)
    const long long wide_tiles = (long long)((M + CfgWide::BM - 1) / CfgWide::BM) * ((d->Cout + CfgWide::BN - 1) / CfgWide::BN);
    c.norow = false; c.half = false;
    c.cfg = (d->Cout > 128 && wide_tiles >= 224) ? 1 : 0;
    if (const char *e = getenv("KDCC_CONV_CFG")) {              // tuning hook
        if (!strcmp(e, "narrow")) c.cfg = 0;
        else if (!strcmp(e, "norow")) c.norow = true;
        else if (!strcmp(e, "half")) c.half = true;
        else if (!strcmp(e, "deep") && c.cfg == 1) c.cfg = 2;
        else if (!strcmp(e, "narrow1") && c.cfg == 0) c.cfg = 3;
    }
    // 256-pixel tiles that are segments of one image row, 3x3 / stride 1 / 'same': row-buffer kernels (the narrow one is
    // compiled for <= 128 VGPRs, which the fp32 parity path's blocked accumulation does not fit)
    const bool row_geom = !c.norow && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == d->dil && d->W % 256 == 0;
    c.row_wide = row_geom && c.cfg == 1 && d->dil <= CfgRowX::MAXDIL;
    c.row_x = c.row_wide && d->dil > CfgRow::MAXDIL;
    c.row_narrow = row_geom && c.cfg == 0 && d->dtype == KD_BF16 && d->dil <= CfgRowN::MAXDIL;
    // persistent kernels (bf16 wide tiles, whole tiles, vector-friendly epilogue): one workgroup per CU walks the tiles
    static int persist = -1, ncu = 0;
    if (persist < 0) {
        const char *v = getenv("KDCC_CONV_PERSIST");
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
            ncu = 256;
        persist = !(v && v[0] == '0');
    }
    c.ncu = ncu;
    c.nops = (ep->res_pre ? 1 : 0) + (ep->mask ? 1 : 0) + (ep->res_post ? 1 : 0);
    const bool persist_ok = persist && !c.half && d->dtype == KD_BF16 && c.cfg == 1 && c.vec_ok && !ep->raw_f32 && M % 256 == 0 && d->Cout % 256 == 0 &&
                            (c.nops <= 2 || (pp_row() && !(tune & 512)));   // three operands: the default (ping-pong) instantiations only
    c.use_row_persist = persist_ok && c.row_wide && !c.row_x;
    c.use_igemm_persist = !c.use_row_persist && persist_ok && d->kh == 1 && d->stride == 1 && d->pad == 0;
    c.use_pp128 = !c.use_row_persist && !c.use_igemm_persist && !(c.row_wide && c.half && d->dtype == KD_BF16) && c.row_narrow && pp_row() && persist &&
                  c.vec_ok && !ep->raw_f32 && c.nops <= 2 && d->Cout % 128 == 0 && d->W % 512 == 0 && d->dil <= 16 && d->Cin % 32 == 0;
    return c;
}

/* Rows of per-channel partial sums the kernel selected for (d, ep) writes to ep->bn_sums (one per 128 output pixels), or 0 when
 * that kernel does not produce them (the caller then runs kd_channel_sums on the result). */
extern "C" int32_t kd_conv2d_bn_sums_rows(const kd_conv_desc *d, const kd_conv_epilogue *ep)
{
    if (!d || !ep || d->dtype != KD_BF16 || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
    static int tn = -1;
    if (tn < 0) tn = KD_TUNING_ENV_INT("KDCC_CONV_TUNE");
    const ConvSel c = conv_select(d, ep, tn);
    if (!ep->mask) {
        // no mask: the sums of the OUTPUT (S1 = sum of the stored values, S2 = 0), which only the ping-pong 1x1 kernel without
        // epilogue operands and with the raw output alone takes (the tensor the ASPP image pooling averages)
        static int lw_pw = -1;
        if (lw_pw < 0) { const char *v = getenv("KDCC_CONV_LW_PW"); lw_pw = (v && v[0] == '1') ? 1 : 0; }
        if (!c.use_igemm_persist || c.nops != 0 || !ep->out_raw || ep->out_act || ep->raw_f32 || (tn & 512) || !pp_row() || lw_pw) return 0;
        return (int32_t)((long long)d->N * d->Ho * d->Wo / 128);
    }
    // (the ping-pong instantiations with one or two epilogue operands; with three the sums' registers spill 150 values)
    if (!(c.use_row_persist || c.use_igemm_persist || c.use_pp128) || (tn & 512) || !pp_row() || (c.use_row_persist && d->dil > 32) || c.nops > 2) return 0;
    return (int32_t)((long long)d->N * d->Ho * d->Wo / 128);
}

/* The classifier epilogue (kd_conv_epilogue.cls_w): conv_row_lw_kernel only, one N tile (Cout == 256), nothing else in the epilogue. */
extern "C" int32_t kd_conv2d_cls_supported(const kd_conv_desc *d, const kd_conv_epilogue *ep)
{
    if (!d || !ep || !ep->cls_w || !ep->cls_out || d->dtype != KD_BF16 || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
    if (d->Cout != 256 || ep->ncls < 1 || ep->ncls > 32 || ep->ld_cls < ep->ncls || !kd_aligned16(ep->cls_w)) return 0;
    if (ep->out_raw || ep->out_act || ep->bn_sums || ep->raw_f32) return 0;
    static int tn = -1;
    if (tn < 0) tn = KD_TUNING_ENV_INT("KDCC_CONV_TUNE");
    static int duo = -1;
    if (duo < 0) { const char *v = getenv("KDCC_CONV_DUO"); duo = v ? atoi(v) : 0; }
    const ConvSel c = conv_select(d, ep, tn);
    return c.use_row_persist && c.nops == 0 && duo < 2 && lw_row() && pp_row() && d->dil <= 32 && d->H >= 2 * d->dil && !(tn & 512);
}

// Workgroups of the persistent conv grids: one per CU, or fewer (a multiple of 8, one XCD round) when KDCC_PERSIST_CUS / kd_conv_set_persist_cus
// says so -- that leaves CUs to a kernel on another stream (the RCCL all-reduce the reducer launches from inside backward, which
// otherwise only gets a CU between two conv launches).  Results do not depend on it: a tile's arithmetic is the same whichever
// workgroup computes it (tests/test_ddp_gpu.py).
static std::atomic<int> g_persist_cus{-1};   // -1: not read yet; 0: every CU (process-wide; launches on any thread / device read it)
static int persist_cus_value(int ncu)
{
    int n = g_persist_cus.load(std::memory_order_relaxed);
    if (n < 0) {
        const char *v = getenv("KDCC_PERSIST_CUS");
        n = v ? atoi(v) : 0;
        if (n < 0) n = 0;
        int expect = -1;
        if (!g_persist_cus.compare_exchange_strong(expect, n)) n = expect;   // (a concurrent kd_conv_set_persist_cus wins)
    }
    if (n == 0 || n > ncu) n = ncu;        // 0 = every CU; a cap above the CU count is the full chip
    if (n < 8) n = 8;                      // a cap below one XCD round is one XCD round (never "the full chip")
    return n - n % 8;
}
extern "C" int kd_conv_set_persist_cus(int32_t n)
{
    KD_REQUIRE(n >= 0, KD_ERR_INVALID, "kd_conv_set_persist_cus: n must be >= 0 (0 = one workgroup per CU)");
    g_persist_cus.store(n);
    return KD_OK;
}

// x2 != nullptr: K-concatenated 1x1 conv (kd_conv1x1_dual_fwd) -- d->Cin is the TOTAL reduction depth, channels [0, cin1) come from
// x (pixel stride d->ldx), [cin1, d->Cin) from x2 (pixel stride ldx2); only the persistent ping-pong 1x1 kernel takes it.
static int conv2d_fwd_impl(const kd_conv_desc *d, const void *x, const void *w_packed, const kd_conv_epilogue *ep,
                           kd_stream_t stream, const void *x2, int cin1, int ldx2)
{
    KD_REQUIRE(d && x && w_packed && ep, KD_ERR_INVALID, "kd_conv2d_fwd: null argument");
    KD_REQUIRE(d->dtype == KD_F32 || d->dtype == KD_BF16, KD_ERR_INVALID, "kd_conv2d_fwd: bad dtype %d", d->dtype);
    const int es = kd_elem_size(d->dtype);
    const int bk = 128 / es;   // channel granule (both K-stage widths divide it)
    KD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0,
               KD_ERR_INVALID, "kd_conv2d_fwd: non-positive dimension");
    KD_REQUIRE((d->kh == 1 && d->kw == 1) || (d->kh == 3 && d->kw == 3), KD_ERR_UNSUPPORTED,
               "kd_conv2d_fwd: kernel %dx%d not supported (1x1, 3x3)", d->kh, d->kw);
    KD_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, KD_ERR_INVALID, "kd_conv2d_fwd: bad stride/dil/pad");
    KD_REQUIRE(d->Cin % bk == 0, KD_ERR_UNSUPPORTED, "kd_conv2d_fwd: Cin=%d must be a multiple of %d", d->Cin, bk);
    KD_REQUIRE(d->ldx >= (x2 ? cin1 : d->Cin) && (d->ldx * es) % 16 == 0, KD_ERR_INVALID, "kd_conv2d_fwd: bad ldx=%d", d->ldx);
    KD_REQUIRE(kd_aligned16(x) && kd_aligned16(w_packed), KD_ERR_INVALID, "kd_conv2d_fwd: x/w must be 16-B aligned");
    const int ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    KD_REQUIRE(ho == d->Ho && wo == d->Wo, KD_ERR_INVALID, "kd_conv2d_fwd: Ho/Wo (%d,%d) inconsistent, expect (%d,%d)",
               d->Ho, d->Wo, ho, wo);
    const long long in_elems = (long long)d->N * d->H * d->W * d->ldx;
    const long long w_elems = (long long)d->Cout * d->kh * d->kw * d->Cin;
    KD_REQUIRE(in_elems < (1ll << 31) && w_elems < (1ll << 31), KD_ERR_UNSUPPORTED,
               "kd_conv2d_fwd: tensor exceeds 2^31 elements");
    const bool cls = ep->cls_w != nullptr || ep->cls_out != nullptr;
    KD_REQUIRE(!cls || (!x2 && kd_conv2d_cls_supported(d, ep)), KD_ERR_UNSUPPORTED,
               "kd_conv2d_fwd: the classifier epilogue needs conv_row_lw_kernel with one N tile and nothing else in the epilogue (ask kd_conv2d_cls_supported first)");
    KD_REQUIRE(cls || ep->out_raw || ep->out_act, KD_ERR_INVALID, "kd_conv2d_fwd: no output requested");

    ConvParams p;
    p.x = x; p.w = w_packed;
    p.M = d->N * d->Ho * d->Wo;
    p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.ldx = d->ldx;
    p.HoWo = d->Ho * d->Wo;
    p.Ktot = d->kh * d->kw * d->Cin;
    p.x2 = x2; p.ldx2 = ldx2; p.nk1 = 0x7fffffff;
    p.ep = *ep;
    {
        static int eb = -1;
        if (eb < 0) { const char *v = getenv("KDCC_EPI_BATCH"); eb = !(v && v[0] == '0'); }
        p.epi_batch = eb;
        static int tn = -1;
        if (tn < 0) tn = KD_TUNING_ENV_INT("KDCC_CONV_TUNE");   // timing ablations / timestamps: tuning build only (kd_common.h)
        p.tune = tn;
        static int sg = -1;
        if (sg < 0) sg = KD_TUNING_ENV_INT("KDCC_CONV_STAGGER");
        p.stagger_us = sg;
    }
    const ConvSel sel = conv_select(d, ep, p.tune);
    p.vec_ok = sel.vec_ok;
    const int cfg = sel.cfg, nops = sel.nops, ncu = sel.ncu;
    const bool half = sel.half, row_wide = sel.row_wide, row_x = sel.row_x, row_narrow = sel.row_narrow;
    KD_REQUIRE(!ep->bn_sums || kd_conv2d_bn_sums_rows(d, ep) > 0, KD_ERR_UNSUPPORTED,
               "kd_conv2d_fwd: bn_sums is not produced by the kernel this problem selects (ask kd_conv2d_bn_sums_rows first)");
    if (ep->bn_sums) KD_NOTE_KERNEL(ep->mask ? "bn_sums_epilogue" : "out_sums_epilogue");   // (kernel-selection log: counted next to the kernel that carries it)
    if (cls) KD_NOTE_KERNEL("cls_epilogue");
    hipStream_t s = (hipStream_t)stream;
    // Workgroups of the persistent kernels (one per CU, each walks tiles for 1-6 ms).  KDCC_PERSIST_CUS=n (a multiple of 8,
    // e.g. 248) leaves CUs free for a concurrent kernel -- the RCCL all-reduce the gradient reducer launches on its side stream
    // from inside backward -- which otherwise only gets a CU between two conv launches.  Results do not depend on it: a tile's
    // arithmetic is the same whichever workgroup computes it (tests/test_ddp_gpu.py).
    auto persist_cus = [&]() { return persist_cus_value(ncu); };
    auto launch = [&](auto cf, auto tag) {
        using CF = decltype(cf);
        using T = decltype(tag);
        const int bkk = CF::RB / (int)sizeof(T);
        p.nkc = d->Cin / bkk;
        p.nk = d->kh * d->kw * p.nkc;
        p.tiles_n = (d->Cout + CF::BN - 1) / CF::BN;
        const int tiles_m = (p.M + CF::BM - 1) / CF::BM;
        hipLaunchKernelGGL((conv_igemm_kernel<T, CF>), dim3((unsigned)(tiles_m * p.tiles_n)), dim3(64 * CF::NW), 0, s, p);
    };
    const bool f32 = d->dtype == KD_F32;
    auto persist_grid = [&]() {
        p.tiles_n = d->Cout / 256;
        p.tiles_m = p.M / 256;
        p.ntiles = p.tiles_m * p.tiles_n;
        static int tng = -1;
        // N tiles walked four at a time over all M tiles (Cout >= 2048): an XCD then keeps 4 weight slabs (K x 256) in its L2 for
        // the whole launch instead of cycling all 8-16 of them per round of tiles; +3-4 % on the 4096-wide 1x1 layers
        if (tng < 0) { const char *v = getenv("KDCC_CONV_TNGROUP"); tng = v ? atoi(v) : 4; }
        p.tn_group = (tng > 0 && p.tiles_n > tng && p.tiles_n % tng == 0) ? tng : 0;
        const int nwg = p.ntiles < persist_cus() ? p.ntiles : persist_cus();
        return dim3((unsigned)((nwg + 7) / 8 * 8));
    };
    // conv_row_duo_kernel (conv_lw.hip): 256 x 128 tiles, two workgroups per CU -- one's epilogue under the other's main loop.
    // KDCC_CONV_DUO: 0 off, 1 the Cout = 128 layers (instead of the 512 x 128 ping-pong kernel), 2 every row-buffer layer it fits
    static int duo = -1;
    if (duo < 0) { const char *v = getenv("KDCC_CONV_DUO"); duo = v ? atoi(v) : 0; }
    const bool duo_ok = duo > 0 && lw_row() && d->dtype == KD_BF16 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == d->dil && d->dil <= 32 &&
                        d->W % 256 == 0 && p.M % 256 == 0 && d->H >= 2 * d->dil && d->Cout % 128 == 0 && d->Cin % 64 == 0 && sel.vec_ok && !ep->raw_f32 && nops <= 1 &&
                        !ep->bn_sums && !(p.tune & 512) && (long long)(p.M / 256) * (d->Cout / 128) >= 2 * ncu &&
                        (duo >= 2 || d->Cout == 128);
    if (x2)   // two A sources: the persistent ping-pong 1x1 kernel only (kd_conv1x1_dual_supported)
        KD_REQUIRE(!duo_ok && !sel.use_row_persist && sel.use_igemm_persist && pp_row() && cin1 % (CfgWide::RB / es) == 0, KD_ERR_UNSUPPORTED,
                   "kd_conv1x1_dual_fwd: this shape does not select conv_igemm_persist_kernel<pp> (ask kd_conv1x1_dual_supported first)");
    if (duo_ok) {
        p.tiles_n = d->Cout / 128;
        p.tiles_m = p.M / 256;
        p.ntiles = p.tiles_m * p.tiles_n;
        {   // N tiles walked in groups over all M tiles for wide layers (see persist_grid): 8 tiles of 128 channels = 4 of 256
            static int tng = -1;
            if (tng < 0) { const char *v = getenv("KDCC_CONV_TNGROUP"); tng = v ? atoi(v) : 4; }
            p.tn_group = (tng > 0 && p.tiles_n > 2 * tng && p.tiles_n % (2 * tng) == 0) ? 2 * tng : 0;
        }
        p.nkc = d->Cin / 32;
        p.nk = 9 * p.nkc;
        const int cap = 2 * persist_cus();
        const int nwg = p.ntiles < cap ? p.ntiles : cap;
        KD_NOTE_KERNEL("conv_row_duo_kernel");
        KD_REQUIRE(kd_launch_conv_row_duo(p, nops, (unsigned)((nwg + 7) / 8 * 8), s), KD_ERR_UNSUPPORTED, "kd_conv2d_fwd: no conv_row_duo_kernel instantiation");
    } else if (sel.use_row_persist) {
        p.nkc = d->Cin / (CfgRow::RB / es);
        p.nk = 9 * p.nkc;
        const dim3 grid = persist_grid();
        // H >= 2 dil: every output row has >= 2 kernel rows inside the image, which the loop's period hand-over assumes (with dil < H < 2 dil the
        // rows H - dil <= ho < dil have one; they go to the ping-pong kernel, which counts kernel rows per tile)
        const bool lw = lw_row() && pp_row() && d->dil <= 32 && d->H >= 2 * d->dil && !(p.tune & 512);
        KD_NOTE_KERNEL((p.tune & 512) ? "conv_row_persist_kernel<dbg>" : lw ? "conv_row_lw_kernel" : (pp_row() && d->dil <= 32) ? "conv_row_persist_kernel<pp>" : "conv_row_persist_kernel<lockstep>");
        if (lw) {
            KD_REQUIRE(kd_launch_conv_row_lw(p, cls ? 16 : (nops | (ep->bn_sums ? 4 : 0)), grid.x, s), KD_ERR_UNSUPPORTED, "kd_conv2d_fwd: no conv_row_lw_kernel instantiation for %d epilogue operands", nops);
        } else if (p.tune & 512) {   // phase clocks (tools/conv_timeline.py)
            if (nops == 0) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 0, true, true>), grid, dim3(512), 0, s, p);
            else if (nops == 1) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 1, true, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 2, true, true>), grid, dim3(512), 0, s, p);
        } else if (pp_row() && d->dil <= 32) {
            if (nops == 0) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 0, false, true>), grid, dim3(512), 0, s, p);
            else if (ep->bn_sums && nops == 1) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 5, false, true>), grid, dim3(512), 0, s, p);
            else if (ep->bn_sums && nops == 2) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 6, false, true>), grid, dim3(512), 0, s, p);
            else if (nops == 1) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 1, false, true>), grid, dim3(512), 0, s, p);
            else if (nops == 2) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 2, false, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 3, false, true>), grid, dim3(512), 0, s, p);
        } else if (nops == 0) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 0>), grid, dim3(512), 0, s, p);
        else if (nops == 1) hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_row_persist_kernel<CfgRow, 2>), grid, dim3(512), 0, s, p);
    } else if (sel.use_igemm_persist) {
        p.nkc = d->Cin / (CfgWide::RB / es);
        p.nk = d->kh * d->kw * p.nkc;
        const dim3 grid = persist_grid();
        // conv_pw_lw_kernel (the lone-wave loop for 1x1 layers) is bit-identical but NOT faster here: its main loop ties the
        // ping-pong kernel's (8 DMA pieces per k-step keep both near the staging rate) and its serial epilogue -- four waves with
        // twice the instructions each, nothing to overlap them -- costs 10-60 % more on these short-K layers (tools/lw_ablate.sh).
        // Opt-in for A/B: KDCC_CONV_LW_PW=1.
        static int lw_pw = -1;
        if (lw_pw < 0) { const char *v = getenv("KDCC_CONV_LW_PW"); lw_pw = (v && v[0] == '1') ? 1 : 0; }
        if (x2) p.nk1 = cin1 / (CfgWide::RB / es);
        const bool lw = lw_pw && lw_row() && pp_row() && d->Cin % 128 == 0 && !(p.tune & 512) && !x2;
        KD_NOTE_KERNEL(lw ? "conv_pw_lw_kernel" : x2 ? "conv_igemm_persist_kernel<pp,dual>" : pp_row() ? "conv_igemm_persist_kernel<pp>" : "conv_igemm_persist_kernel<lockstep>");
        if (lw) {
            KD_REQUIRE(kd_launch_conv_pw_lw(p, nops | (ep->bn_sums ? 4 : 0), grid.x, s), KD_ERR_UNSUPPORTED, "kd_conv2d_fwd: no conv_pw_lw_kernel instantiation for %d epilogue operands", nops);
        } else if (pp_row()) {
            if (nops == 0 && ep->bn_sums) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 8, true>), grid, dim3(512), 0, s, p);
            else if (nops == 0) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 0, true>), grid, dim3(512), 0, s, p);
            else if (ep->bn_sums && nops == 1) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 5, true>), grid, dim3(512), 0, s, p);
            else if (ep->bn_sums && nops == 2) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 6, true>), grid, dim3(512), 0, s, p);
            else if (nops == 1) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 1, true>), grid, dim3(512), 0, s, p);
            else if (nops == 2) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 2, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 3, true>), grid, dim3(512), 0, s, p);
        } else if (nops == 0) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 0>), grid, dim3(512), 0, s, p);
        else if (nops == 1) hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_persist_kernel<CfgWide, 2>), grid, dim3(512), 0, s, p);
    } else if (row_wide && half && d->dtype == KD_BF16) {
        p.nkc = d->Cin / (CfgRowH::RB / es);
        p.nk = 9 * p.nkc;
        p.tiles_n = (d->Cout + CfgRowH::BN - 1) / CfgRowH::BN;
        const dim3 grid((unsigned)((p.M / 256) * p.tiles_n));
        KD_NOTE_KERNEL("conv_igemm_row_kernel<half>");
        if (row_x) hipLaunchKernelGGL((conv_igemm_row_kernel<bf16_t, CfgRowHX>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_row_kernel<bf16_t, CfgRowH>), grid, dim3(256), 0, s, p);
    } else if (sel.use_pp128) {
        // Cout = 128 layers: 512 x 128 ping-pong tiles
        p.nkc = d->Cin / 32;
        p.nk = 9 * p.nkc;
        p.tiles_n = d->Cout / 128;
        p.tiles_m = p.M / 512;
        p.ntiles = p.tiles_m * p.tiles_n;
        p.tn_group = 0;
        const int nwg = p.ntiles < persist_cus() ? p.ntiles : persist_cus();
        const dim3 grid((unsigned)((nwg + 7) / 8 * 8));
        // conv_row_tall_kernel (conv_lw.hip): the same tiles with one wave per SIMD and the hand-scheduled loop; A/B: KDCC_CONV_LW=0
        const bool tall = lw_row() && d->Cin % 64 == 0 && d->H > d->dil && !(p.tune & 512) && !(ep->bn_sums && nops == 0);
        KD_NOTE_KERNEL(tall ? "conv_row_tall_kernel" : "conv_row_pp128_kernel");
        if (tall) {
            KD_REQUIRE(kd_launch_conv_row_tall(p, nops | (ep->bn_sums ? 4 : 0), grid.x, s), KD_ERR_UNSUPPORTED, "kd_conv2d_fwd: no conv_row_tall_kernel instantiation for %d epilogue operands", nops);
        } else
        if (p.tune & 512) hipLaunchKernelGGL((conv_row_pp128_kernel<0, true>), grid, dim3(512), 0, s, p);   // phase clocks (no-operand form only)
        else if (nops == 0) hipLaunchKernelGGL((conv_row_pp128_kernel<0>), grid, dim3(512), 0, s, p);
        else if (ep->bn_sums && nops == 1) hipLaunchKernelGGL((conv_row_pp128_kernel<5>), grid, dim3(512), 0, s, p);
        else if (ep->bn_sums) hipLaunchKernelGGL((conv_row_pp128_kernel<6>), grid, dim3(512), 0, s, p);
        else if (nops == 1) hipLaunchKernelGGL((conv_row_pp128_kernel<1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_row_pp128_kernel<2>), grid, dim3(512), 0, s, p);
    } else if (row_wide || row_narrow) {
        p.nkc = d->Cin / ((row_wide ? CfgRow::RB : CfgRowN::RB) / es);
        p.nk = 9 * p.nkc;
        p.tiles_n = (d->Cout + (row_wide ? CfgRow::BN : CfgRowN::BN) - 1) / (row_wide ? CfgRow::BN : CfgRowN::BN);
        const dim3 grid((unsigned)((p.M / 256) * p.tiles_n));
        KD_NOTE_KERNEL(row_narrow ? "conv_igemm_row_kernel<narrow>" : row_x ? (f32 ? "conv_igemm_row_kernel<f32,x>" : "conv_igemm_row_kernel<x>")
                                  : (f32 ? "conv_igemm_row_kernel<f32,wide>" : "conv_igemm_row_kernel<wide>"));
        if (row_narrow) hipLaunchKernelGGL((conv_igemm_row_kernel<bf16_t, CfgRowN>), grid, dim3(512), 0, s, p);
        else if (d->dtype == KD_BF16 && row_x) hipLaunchKernelGGL((conv_igemm_row_kernel<bf16_t, CfgRowX>), grid, dim3(512), 0, s, p);
        else if (d->dtype == KD_BF16) hipLaunchKernelGGL((conv_igemm_row_kernel<bf16_t, CfgRow>), grid, dim3(512), 0, s, p);
        else if (row_x) hipLaunchKernelGGL((conv_igemm_row_kernel<float, CfgRowXF>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_row_kernel<float, CfgRowF>), grid, dim3(512), 0, s, p);
    } else if (d->dtype == KD_BF16) {
        KD_NOTE_KERNEL(cfg == 1 ? (half ? "conv_igemm_kernel<half>" : "conv_igemm_kernel<wide>") : cfg == 2 ? "conv_igemm_kernel<deep>"
                                : cfg == 3 ? "conv_igemm_kernel<narrow>" : "conv_igemm_kernel<narrow2>");
        if (cfg == 1 && (p.tune & 8)) launch(CfgWideF{}, bf16_t{});   // A/B: plain main loop
        else if (cfg == 1 && half) launch(CfgHalf{}, bf16_t{});
        else if (cfg == 1) launch(CfgWide{}, bf16_t{});
        else if (cfg == 2) launch(CfgDeep{}, bf16_t{});
        else if (cfg == 3) launch(CfgNarrow{}, bf16_t{});
        else launch(CfgNarrow2{}, bf16_t{});
    } else {
        KD_NOTE_KERNEL(cfg == 1 ? "conv_igemm_kernel<f32,wide>" : cfg == 2 ? "conv_igemm_kernel<f32,deep>" : "conv_igemm_kernel<f32,narrow>");
        if (cfg == 1) launch(CfgWideF{}, float{});
        else if (cfg == 2) launch(CfgDeep{}, float{});
        else launch(CfgNarrow{}, float{});   // fp32 parity path: its blocked accumulation does not fit 128 VGPRs
    }
    KD_CHECK_LAUNCH("kd_conv2d_fwd");
    return KD_OK;
}

extern "C" int kd_conv2d_fwd(const kd_conv_desc *d, const void *x, const void *w_packed, const kd_conv_epilogue *ep,
                             kd_stream_t stream)
{
    return conv2d_fwd_impl(d, x, w_packed, ep, stream, nullptr, 0, 0);
}

// ---- K-concatenated 1x1 conv: y = [x | x2] . [w | w2]^T in ONE accumulator chain ------------------------------------------------
// The bottleneck blocks of WRN-38 (mod6 / mod7, wider_resnet.py:143-182) end in `out = conv3(...); out.add_(shortcut)` with
// shortcut = proj_conv(bn1) -- two 1x1 convs onto the same output.  As one GEMM over the concatenated K (mod6: 1024 + 1024, mod7:
// 2048 + 2048) the 2048- / 4096-channel shortcut tensor is never written and read back and one wide epilogue disappears; the same
// for the input gradient of the block (conv1's and proj_conv's dgrads both land on bn1's output: K = Cmid + Cout).
static bool dual_shape_ok(const kd_conv_desc *d, int Cin2, int ldx2, const kd_conv_epilogue *ep, kd_conv_desc *tot)
{
    if (!d || !ep || d->dtype != KD_BF16 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->H != d->Ho || d->W != d->Wo) return false;
    if (Cin2 <= 0 || d->Cin % 64 != 0 || Cin2 % 64 != 0 || ldx2 < Cin2 || (ldx2 * 2) % 16 != 0) return false;
    *tot = *d;
    tot->Cin = d->Cin + Cin2;
    static int tn = -1;
    if (tn < 0) tn = KD_TUNING_ENV_INT("KDCC_CONV_TUNE");
    const ConvSel c = conv_select(tot, ep, tn);
    static int dual = -1;
    if (dual < 0) { const char *v = getenv("KDCC_CONV_DUAL"); dual = !(v && v[0] == '0'); }   // A/B: 0 = two launches (engine falls back)
    return dual && c.use_igemm_persist && !c.use_row_persist && pp_row();
}

extern "C" int32_t kd_conv1x1_dual_supported(const kd_conv_desc *d, int32_t Cin2, int32_t ldx2, const kd_conv_epilogue *ep)
{
    kd_conv_desc tot;
    return dual_shape_ok(d, Cin2, ldx2, ep, &tot) ? 1 : 0;
}

extern "C" int kd_conv1x1_dual_fwd(const kd_conv_desc *d, const void *x, const void *x2, int32_t Cin2, int32_t ldx2, const void *w_cat,
                                   const kd_conv_epilogue *ep, kd_stream_t stream)
{
    KD_REQUIRE(d && x && x2 && w_cat && ep, KD_ERR_INVALID, "kd_conv1x1_dual_fwd: null argument");
    KD_REQUIRE(kd_aligned16(x2), KD_ERR_INVALID, "kd_conv1x1_dual_fwd: x2 must be 16-B aligned");
    kd_conv_desc tot;
    KD_REQUIRE(dual_shape_ok(d, Cin2, ldx2, ep, &tot), KD_ERR_UNSUPPORTED,
               "kd_conv1x1_dual_fwd: needs a bf16 1x1 / stride-1 conv on the persistent 256 x 256 tiles (kd_conv1x1_dual_supported)");
    KD_REQUIRE((long long)d->N * d->H * d->W * ldx2 < (1ll << 31), KD_ERR_UNSUPPORTED, "kd_conv1x1_dual_fwd: x2 exceeds 2^31 elements");
    return conv2d_fwd_impl(&tot, x, w_cat, ep, stream, x2, d->Cin, ldx2);
}

extern "C" int kd_debug_conv_tlog(unsigned long long *dst, size_t bytes)
{
    if (KD_TUNING_ENV_INT("KDCC_CONV_TUNE") & 1024) return kd_lw_tlog_copy(dst, bytes) == 0 ? KD_OK : KD_ERR_HIP;   // conv_lw.hip's log
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(kd_conv_tlog), bytes < sizeof(kd_conv_tlog) ? bytes : sizeof(kd_conv_tlog), 0,
                               hipMemcpyDeviceToHost) == hipSuccess ? KD_OK : KD_ERR_HIP;
}

extern "C" int kd_pack_conv_weight(const float *src, void *dst, int32_t dtype, int32_t mode, int32_t Cout, int32_t Cin,
                                   int32_t kh, int32_t kw, int32_t cin_pad, kd_stream_t stream)
{
    KD_REQUIRE(src && dst, KD_ERR_INVALID, "kd_pack_conv_weight: null argument");
    KD_REQUIRE(dtype == KD_F32 || dtype == KD_BF16, KD_ERR_INVALID, "kd_pack_conv_weight: bad dtype");
    KD_REQUIRE(mode == KD_PACK_FWD || mode == KD_PACK_DGRAD, KD_ERR_INVALID, "kd_pack_conv_weight: bad mode");
    KD_REQUIRE(Cout > 0 && Cin > 0 && kh > 0 && kw > 0, KD_ERR_INVALID, "kd_pack_conv_weight: bad shape");
    if (mode == KD_PACK_FWD) KD_REQUIRE(cin_pad >= Cin, KD_ERR_INVALID, "kd_pack_conv_weight: cin_pad < Cin");
    else KD_REQUIRE(cin_pad == Cin, KD_ERR_INVALID, "kd_pack_conv_weight: dgrad pack takes cin_pad == Cin");
    const size_t total = mode == KD_PACK_FWD ? (size_t)Cout * kh * kw * cin_pad : (size_t)Cin * kh * kw * Cout;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16)
        hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t *)dst, mode, Cout,
                           Cin, kh, kw, cin_pad);
    else
        hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float *)dst, mode, Cout,
                           Cin, kh, kw, cin_pad);
    KD_CHECK_LAUNCH("kd_pack_conv_weight");
    return KD_OK;
}

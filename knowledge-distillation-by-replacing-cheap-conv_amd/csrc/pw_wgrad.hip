// Weight gradient of a 1x1 convolution (kd_pw_wgrad):
//   dw[co][ci] = sum_m dy[m][co] * a[m][ci]            (TN GEMM, reduction over pixels)
// Both operands are pixel-major (NHWC), i.e. the reduction index is the strided one, so
// each K stage (64 pixels bf16 / 32 pixels f32) is transposed on its way into LDS
// (global 16-B row chunks -> registers -> per-element LDS writes) into the same
// [channel][128 B of K] swizzled image the conv kernel uses; the MFMA core is shared.
// Split-K over pixels with fp32 partial slabs in the caller's workspace and a
// fixed-order reduction (deterministic, unlike float atomics).
#include <stdlib.h>

#include "igemm_core.h"
#include "wgrad_lw_body.inc"
#include "wgrad_pw_lw_body.inc"

namespace {

constexpr int TM = 128, TN = 128;
__device__ __attribute__((aligned(256))) uint32_t kd_zero_page_w[64];  // zero-initialised: source of out-of-range DMA lanes
constexpr int STAGE_BYTES = (TM + TN) * IG_ROWB;  // 32 KiB

struct WgradParams {
    const void *a;
    const void *dy;
    float *part;  // [splits][taps][Cout][Cin]
    int M, Cin, Cout, lda, ldy;
    int tiles_ci, rows_per_split, splits, tiles, dbg;
    // conv geometry of the A operand (kd_conv2d_wgrad): GEMM row m = output pixel (n, ho, wo) reads input pixel
    // (n, ho*stride - pad + ky*dil, wo*stride - pad + kx*dil) for the tap blockIdx.z = ky*kw + kx, zeros outside the image
    int geom, H, W, Ho, Wo, kw, stride, pad, dil;
    uint32_t mg_howo, sh_howo, mg_wo, sh_wo;   // magic numbers: m / (Ho*Wo), rem / Wo without integer division
};

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t magic, uint32_t shift)
{
    return (__umulhi(n, magic) + n) >> shift;   // exact for n < 2^31 (host-side magic: see fastdiv_magic)
}

// input-pixel index (in pixels, not elements) of GEMM row m for tap (ky, kx), or -1 when the tap falls outside the image
__device__ __forceinline__ int a_row(const WgradParams &p, int m, int ky, int kx)
{
    if (!p.geom) return m;
    const uint32_t n = fastdiv((uint32_t)m, p.mg_howo, p.sh_howo);
    const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Ho * p.Wo);
    const uint32_t ho = fastdiv(rem, p.mg_wo, p.sh_wo);
    const uint32_t wo = rem - ho * (uint32_t)p.Wo;
    const int hi = (int)ho * p.stride - p.pad + ky * p.dil, wi = (int)wo * p.stride - p.pad + kx * p.dil;
    if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) return -1;
    return ((int)n * p.H + hi) * p.W + wi;
}

// transposing stage loader: rows = pixels, 16-B chunks of channels -> LDS [channel][k]
template <typename T> struct Stager {
    static constexpr int ES = sizeof(T);
    static constexpr int KROWS = IG_ROWB / ES;  // pixels per stage: 64 (bf16) / 32 (f32)
    static constexpr int EPC = 16 / ES;         // channels per 16-B chunk: 8 / 4
    static constexpr int SUBS = 64 / KROWS;     // chunk columns covered by one wave instruction: 1 / 2

    __device__ static __forceinline__ void load(const T *g, int ld, int m_base, int M, int c_base, int C, int wv,
                                                int lane, uint4 (&r)[4])
    {
        const int mrow = lane % KROWS, sub = lane / KROWS;
        const int m = m_base + mrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cch = (wv * 4 + i) * SUBS + sub;
            const int c = c_base + cch * EPC;
            if (m < M && c + EPC <= C) {
                r[i] = *(const uint4 *)(g + (size_t)m * ld + c);
            } else {
                // tails: element-wise (channels beyond C or pixels beyond M contribute zeros)
                __attribute__((aligned(16))) T tmp[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) tmp[e] = (m < M && c + e < C) ? g[(size_t)m * ld + c + e] : (T)0;
                r[i] = *(const uint4 *)tmp;
            }
        }
    }
    // A operand of a general conv: the row's source pixel comes from a_row()
    __device__ static __forceinline__ void load_a(const WgradParams &p, const T *g, int m_base, int M, int c_base, int wv,
                                                  int lane, int ky, int kx, uint4 (&r)[4])
    {
        const int mrow = lane % KROWS, sub = lane / KROWS;
        const int m = m_base + mrow;
        const int px = m < M ? a_row(p, m, ky, kx) : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cch = (wv * 4 + i) * SUBS + sub;
            const int c = c_base + cch * EPC;
            if (px >= 0 && c + EPC <= p.Cin) {
                r[i] = *(const uint4 *)(g + (size_t)px * p.lda + c);
            } else {
                __attribute__((aligned(16))) T tmp[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) tmp[e] = (px >= 0 && c + e < p.Cin) ? g[(size_t)px * p.lda + c + e] : (T)0;
                r[i] = *(const uint4 *)tmp;
            }
        }
    }
    __device__ static __forceinline__ void store(char *s, int wv, int lane, const uint4 (&r)[4])
    {
        const int k = lane % KROWS, sub = lane / KROWS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cch = (wv * 4 + i) * SUBS + sub;
            const T *v = (const T *)&r[i];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int row = cch * EPC + e;  // channel within the tile
                const int kchunk = (k * ES) >> 4;
                *(T *)(s + row * IG_ROWB + ((kchunk ^ (row & 7)) << 4) + ((k * ES) & 15)) = v[e];
            }
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];
    using S = Stager<T>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tile = blockIdx.x, split = blockIdx.y, tap = blockIdx.z;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int t_ci = tile % p.tiles_ci, t_co = tile / p.tiles_ci;
    const int co0 = t_co * TM, ci0 = t_ci * TN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = (m_end - m_begin + S::KROWS - 1) / S::KROWS;

    const T *dy = (const T *)p.dy;
    const T *a = (const T *)p.a;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    if (nst > 0) {
        S::load(dy, p.ldy, m_begin, m_end, co0, p.Cout, wv, lane, ra);
        S::load_a(p, a, m_begin, m_end, ci0, wv, lane, ky, kx, rb);
        S::store(lds, wv, lane, ra);
        S::store(lds + TM * IG_ROWB, wv, lane, rb);
    }
    __syncthreads();
    int cur = 0;
    for (int st = 0; st < nst; ++st) {
        const bool more = st + 1 < nst;
        if (more) {
            const int mb = m_begin + (st + 1) * S::KROWS;
            S::load(dy, p.ldy, mb, m_end, co0, p.Cout, wv, lane, ra);
            S::load_a(p, a, mb, m_end, ci0, wv, lane, ky, kx, rb);
        }
        const char *sA = lds + cur * STAGE_BYTES;
        ig_compute_stage<T>(sA, sA + TM * IG_ROWB, wm, wn, lane, acc);
        if (more) {
            char *nx = lds + (cur ^ 1) * STAGE_BYTES;
            S::store(nx, wv, lane, ra);
            S::store(nx + TM * IG_ROWB, wv, lane, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // partial tile -> slab [split][tap][Cout][Cin]
    float *out = p.part + ((size_t)split * gridDim.z + tap) * p.Cout * p.Cin;
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + i * 16 + fq * 4 + r;
                const int ci = ci0 + wn * 64 + j * 16 + frow;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
}

// ---- bf16 fast path: LDS-DMA staging in the natural [pixel][channel] layout + hardware-transposed fragment reads ----
// Stage = 64 pixels x 128 channels per operand (16 KiB each), double-buffered.  The MFMA operands need 8 consecutive
// k (= pixels) per lane for one channel: ds_read_b64_tr_b16 delivers a 4-pixel x 16-channel block column-major, so two
// of them per 16x32 fragment replace the per-element transposing LDS writes of the generic kernel.  LDS rows are
// 256 B; the 32-B channel block b of row m is stored at block b ^ f(m), f(m) = (m & 3) | ((m >> 3) & 1) << 2, which
// makes the eight rows a 32-lane half touches (m, m+1, m+2, m+3, m+8 ...) hit distinct bank groups.
typedef short v4i16_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tr_f(int m) { return (m & 3) | (((m >> 3) & 1) << 2); }

// NST stages resident (32 KiB each), NST-1 in flight ahead of the one being consumed (counted vmcnt: a wave's DMA pieces retire
// in issue order, 8 per stage).  NST = 2: two workgroups per CU, one stage in flight each; NST = 4: one workgroup per CU with
// three stages (96 KiB) in flight -- the kernel is bound by the latency of its gathered stage, not by MFMA or LDS.
template <int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void pw_wgrad_tr_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[NST * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tile = blockIdx.x, split = blockIdx.y, tap = blockIdx.z;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int t_ci = tile % p.tiles_ci, t_co = tile / p.tiles_ci;
    const int co0 = t_co * TM, ci0 = t_ci * TN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = (m_end - m_begin + 63) / 64;
    const bf16_t *dy = (const bf16_t *)p.dy;
    const bf16_t *a = (const bf16_t *)p.a;
    const bf16_t *zero = (const bf16_t *)kd_zero_page_w;

    // staging: a piece = 4 pixel rows x 256 B; lane -> (row l>>4, slot l&15); 4 pieces per operand per wave per stage
    const int prow = lane >> 4, slot = lane & 15;
    auto stage = [&](int st, int buf) {
        char *base = lds + buf * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wv * 4 + j) * 4 + prow;          // row within the stage (0..63)
            const int m = m_begin + st * 64 + r;
            const int c = (slot ^ (tr_f(r) << 1)) * 8;      // source-side swizzle (8 channels per 16-B chunk)
            const bool mok = m < m_end;
            const int px = mok ? a_row(p, m, ky, kx) : -1;
            const bf16_t *s0 = (mok && co0 + c < p.Cout) ? dy + (size_t)m * p.ldy + co0 + c : zero;
            const bf16_t *s1 = (px >= 0 && ci0 + c < p.Cin) ? a + (size_t)px * p.lda + ci0 + c : zero;
            glds16(s0, base + (wv * 4 + j) * 1024);
            glds16(s1, base + 16384 + (wv * 4 + j) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, li = lane & 15;
    auto frag = [&](const char *img, int ks, int t) {
        // 16 channels (block t) x 32 pixels (k-step ks): this lane's 8 k values for channel 16t + li
        const int m0 = ks * 32 + 8 * q + (li >> 2);
        const int m1 = m0 + 4;
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m0 * 256 + ((t ^ tr_f(m0)) << 5) + (li & 3) * 8));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m1 * 256 + ((t ^ tr_f(m1)) << 5) + (li & 3) * 8));
        return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    if (nst > 0) stage(0, 0);
    wait_vm_barrier<0>();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) stage(st + 1, cur ^ 1);
        const char *imgY = lds + cur * 32768, *imgA = imgY + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag(imgY, ks, wm * 4 + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag(imgA, ks, wn * 4 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        wait_vm_barrier<0>();   // next stage landed everywhere; everybody is done reading this one
    }

    float *out = p.part + ((size_t)split * gridDim.z + tap) * p.Cout * p.Cin;
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + i * 16 + fq * 4 + r;
                const int ci = ci0 + wn * 64 + j * 16 + frow;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
}

// slabs [split][tap][Cout][Cin] -> dw (Cout, Cin, kh, kw) like nn.Conv2d.weight.grad; fixed summation order
__global__ void slab_reduce_taps_kernel(const float *__restrict__ part, float *__restrict__ dw, int Cout, int Cin, int taps,
                                        int splits, int accumulate)
{
    const size_t plane = (size_t)Cout * Cin, n = plane * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const size_t cc = i / taps;                  // co * Cin + ci
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[((size_t)k * taps + t) * plane + cc];
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

// the same with one thread per (co, ci) and the taps in registers: the slab reads are coalesced along ci for every tap (the kernel
// above walks the output index, so neighbouring threads read nine different planes), the output is nine consecutive floats
template <int TAPS>
__global__ void slab_reduce_taps_cc_kernel(const float *__restrict__ part, float *__restrict__ dw, size_t plane, int splits,
                                           int accumulate)
{
    for (size_t cc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; cc < plane; cc += (size_t)gridDim.x * blockDim.x) {
        float s[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
        for (int k = 0; k < splits; ++k)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s[t] += part[((size_t)k * TAPS + t) * plane + cc];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) dw[cc * TAPS + t] = accumulate ? dw[cc * TAPS + t] + s[t] : s[t];
    }
}

// ---- wide tile: 256 (Cout) x 256 (Cin) per workgroup, 8 waves (2 x 4 of 128 x 64), one workgroup per CU --------------------------
// The 128 x 128 kernel above keeps one 32-KiB stage in flight per workgroup (two per CU): at the latency of a gathered L2 / HBM
// read that is ~0.2 of the MFMA peak whatever the shape.  Doubling both tile edges halves the bytes staged per FLOP, so the
// same bytes in flight feed twice the matrix work -- the trade the forward kernel's 256 x 256 configs make.  Stage = 64 pixels
// x (256 + 256) channels = four 16-KiB images in the [pixel][128 channels] layout of pw_wgrad_tr_kernel (same source-side
// swizzle, same transposing fragment reads); double-buffered: 128 KiB.
template <bool PW>   // PW: 1x1 / stride 1 / no padding (GEMM row m IS input pixel m): staging addresses are per-lane constants
__global__ __launch_bounds__(512, 2) void conv_wgrad_wide_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * 65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    const int tile = blockIdx.x, split = blockIdx.y, tap = blockIdx.z;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int t_ci = tile % p.tiles_ci, t_co = tile / p.tiles_ci;
    const int co0 = t_co * 256, ci0 = t_ci * 256;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = (m_end - m_begin + 63) / 64;
    const bf16_t *dy = (const bf16_t *)p.dy;
    const bf16_t *a = (const bf16_t *)p.a;
    const bf16_t *zero = (const bf16_t *)kd_zero_page_w;

    // staging: a piece = 4 pixel rows x 256 B of one image; image q of a stage: 0/1 = dy channels 0-127 / 128-255, 2/3 = a.
    // 16 pieces per image, 64 per stage, 8 per wave: wave w stages pieces w*2, w*2+1 of every image (rows (w*2+j)*4 ..)
    const int prow = lane >> 4, slot = lane & 15;
    // pointwise form: element offsets of this lane's eight pieces relative to the stage's first pixel, and which of them exist
    // (the general form below spends ~200 scalar + vector instructions per stage on pixel maps, bounds and 64-bit addresses --
    // as many issue cycles as the stage's 64 MFMAs)
    int poy[4], pox[4];
    uint32_t pvy = 0, pvx = 0;
    if (PW) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wv * 2 + j) * 4 + prow;
            const int c = (slot ^ (tr_f(r) << 1)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = h * 128 + c;
                poy[j * 2 + h] = r * p.ldy + co0 + cc;
                pox[j * 2 + h] = r * p.lda + ci0 + cc;
                pvy |= co0 + cc < p.Cout ? (1u << (j * 2 + h)) : 0u;
                pvx |= ci0 + cc < p.Cin ? (1u << (j * 2 + h)) : 0u;
            }
        }
    }
    auto stage = [&](int st, int buf) {
        char *base = lds + buf * 65536;
        if (PW && m_begin + (st + 1) * 64 <= m_end) {   // whole stage inside the split (always, when M % 64 == 0)
            const bf16_t *yb = dy + (size_t)(m_begin + st * 64) * p.ldy, *xb = a + (size_t)(m_begin + st * 64) * p.lda;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = j * 2 + h, pc = wv * 2 + j;
                    glds16(((pvy >> q) & 1u) ? yb + poy[q] : zero, base + h * 16384 + pc * 1024);
                    glds16(((pvx >> q) & 1u) ? xb + pox[q] : zero, base + (2 + h) * 16384 + pc * 1024);
                }
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pc = wv * 2 + j;
            const int r = pc * 4 + prow;                    // row within the stage (0..63)
            const int m = m_begin + st * 64 + r;
            const int c = (slot ^ (tr_f(r) << 1)) * 8;      // source-side swizzle (8 channels per 16-B chunk)
            const bool mok = m < m_end;
            const int px = mok ? a_row(p, m, ky, kx) : -1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = h * 128 + c;
                const bf16_t *s0 = (mok && co0 + cc < p.Cout) ? dy + (size_t)m * p.ldy + co0 + cc : zero;
                const bf16_t *s1 = (px >= 0 && ci0 + cc < p.Cin) ? a + (size_t)px * p.lda + ci0 + cc : zero;
                glds16(s0, base + h * 16384 + pc * 1024);
                glds16(s1, base + (2 + h) * 16384 + pc * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, li = lane & 15;
    auto frag = [&](const char *img, int ks, int t) {
        const int m0 = ks * 32 + 8 * q + (li >> 2);
        const int m1 = m0 + 4;
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m0 * 256 + ((t ^ tr_f(m0)) << 5) + (li & 3) * 8));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m1 * 256 + ((t ^ tr_f(m1)) << 5) + (li & 3) * 8));
        return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    if (nst > 0) stage(0, 0);
    wait_vm_barrier<0>();
    // (the row kernel's early / late DMA issue -- the two waves of a SIMD staging at different points of the stage -- measured
    // here: 2048->4096 +0.7 %, 4096->256 -2 %; not kept)
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) stage(st + 1, cur ^ 1);
        const char *imgY = lds + cur * 65536 + wm * 16384;                  // this wave's 128 output channels
        const char *imgA = lds + cur * 65536 + (2 + (wn >> 1)) * 16384;     // ... and its 64 input channels
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[8], fb[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = frag(imgY, ks, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag(imgA, ks, (wn & 1) * 4 + j);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        wait_vm_barrier<0>();   // next stage landed everywhere; everybody is done reading this one
    }

    float *out = p.part + ((size_t)split * gridDim.z + tap) * p.Cout * p.Cin;
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 128 + i * 16 + fq * 4 + r;
                const int ci = ci0 + wn * 64 + j * 16 + frow;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
}

// ---- the wide tile of the 1x1 / stride-1 layers on ONE wave per SIMD (round 6; stage loop generated by tools/gen_wgrad_pw_lw.py) ----
// conv_wgrad_wide_kernel<true> keeps one 64-KiB stage in flight and runs its fragment reads and MFMAs as bulk phases of 8 waves
// (MFMA busy 0.50).  Here 4 waves of 128 Cout x 128 Cin (64 accumulator tiles = a[0:255], two fragment sets in v[128:255]) walk the
// same pixels 32 at a time: four 32-KiB stages in a ring (three in flight), per stage 64 MFMAs with the wave's 8 LDS-DMA pieces of
// stage st + 3 dealt into the first 32 gaps, one barrier, and the 32 transposing reads of stage st + 1 dealt into the last 32.
// Same tiles, splits and K order: bit-identical slabs.  Cin % 256 == 0, Cout % 256 == 0, M % 64 == 0.
__global__ __launch_bounds__(256, 1) void conv_wgrad_pw_lw_kernel(const WgradParams p);

// ---- row-buffer tile for 3x3 / stride 1 / 'same' convs: 128 (Cout) x [3 taps kx] x 128 (Cin) per workgroup -------------------------
// Per tap the kernels above stage dy and the (shifted) activations again: nine passes over both tensors.  Here a K stage is 64
// consecutive pixels of ONE image row (W % 64 == 0): dy is staged once and serves the three kx taps of a kernel row, and the
// activations go to LDS as a ROW BUFFER of 64 + 2*dil pixels (halo included, zero outside the image) that the three taps read
// at row offsets kx*dil -- the trick of conv_igemm_row_kernel; the transposing fragment reads' swizzle is keyed on the buffer
// row, so the shifted reads stay conflict-free.  Bytes staged per MAC: (64*128 + 96*128) * 2 B per 128 x 384 x 64 MACs = 1/77
// against 1/32 for the 128 x 128 kernel and 1/64 for the 256 x 256 one.  8 waves (2 x 4 of 64 x 96), grid.z = ky; a
// four-stage ring of 40 KiB stages (160 KiB, one workgroup per CU): three stages in flight behind a counted vmcnt.
// s_waitcnt immediate (gfx9 layout): vmcnt(vm) lgkmcnt(0), expcnt untouched
constexpr int waitcnt_vm_lgkm0(int vm) { return (vm & 15) | (7 << 4) | ((vm >> 4) << 14); }
constexpr int WR_XROWS = 96;                        // row-buffer rows: 64 + 2 * dil <= 96
constexpr int WR_STAGE = 16384 + WR_XROWS * 256;    // dy image + row buffer
constexpr int WR_NST = 4;
__device__ unsigned long long kd_wgrad_tlog[256 * 8 * 8];   // KDCC_WGRAD_DBG=1: per-wave phase clocks of conv_wgrad_row_kernel (debug)
// ABL (tuning build only, KDCC_WGRAD_DBG bits; timing ablations, results wrong): 2 = no MFMAs, 4 = no fragment reads, 8 = no DMA
template <bool ABL, int MODE>   // MODE 0 / 1: half-stage software pipeline, fragment reads in front of / dealt between the MFMAs; 2: ping-pong
__global__ __launch_bounds__(512, 2) void conv_wgrad_row_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[WR_NST * WR_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    // 1-D grid, XCD-aware: the three kernel rows of one (pixel split, tile) are neighbours on ONE XCD -- they stream the same dy
    // stages and (shifted by +-dil rows) the same activations at the same time, so two of the three reads hit that XCD's L2
    // (with ky as the slowest grid dimension the three passes over the tensors ran minutes apart in GPU terms: 3x the HBM reads)
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ky = lin % 3; lin /= 3;
    const int tile = lin % p.tiles, split = lin / p.tiles;
    const int t_ci = tile % p.tiles_ci, t_co = tile / p.tiles_ci;
    const int co0 = t_co * 128, ci0 = t_ci * 128;
    const int m_begin = split * p.rows_per_split;           // multiples of 64; M % 64 == 0 (host)
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = (m_end - m_begin) / 64;
    const int d = p.dil;
    const bf16_t *dy = (const bf16_t *)p.dy;
    const bf16_t *a = (const bf16_t *)p.a;
    const bf16_t *zero = (const bf16_t *)kd_zero_page_w;

    const int prow = lane >> 4, slot = lane & 15;
    // per-lane invariants of this wave's pieces (phase clocks, tools/wgrad_timeline.py: with the whole address arithmetic per
    // stage the DMA issue took 1130-1625 of a 3740-cycle stage): element offsets relative to the stage's first dy pixel / first
    // row-buffer pixel, and the lane-invariant halves of the validity tests
    int voy[2], vox[3];
    uint32_t vy = 0, vx = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                           // dy: 16 pieces of 4 pixel rows x 256 B
        const int r = (wv * 2 + j) * 4 + prow;
        const int c = (slot ^ (tr_f(r) << 1)) * 8;          // source-side swizzle (8 channels per 16-B chunk)
        voy[j] = r * p.ldy + co0 + c;
        vy |= co0 + c < p.Cout ? (1u << j) : 0u;
    }
    // row buffer: up to 24 pieces, row r = pixel x0 - dil + r; wave w stages pieces w, 8 + w, 16 + w -- only those that hold
    // one of the 64 + 2 dil rows the fragments read (dil = 1: 17 pieces, two per wave and a third for wave 0)
    const int nxp = __builtin_amdgcn_readfirstlane(min(3, (64 + 2 * d - 4 * wv + 31) / 32));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = (j * 8 + wv) * 4 + prow;
        const int c = (slot ^ (tr_f(r) << 1)) * 8;
        vox[j] = r * p.lda + ci0 + c;
        vx |= (r < 64 + 2 * d && ci0 + c < p.Cin) ? (1u << j) : 0u;
    }
    auto stage = [&](int st) {
        if (ABL && (p.dbg & 8)) return;
        char *base = lds + (st % WR_NST) * WR_STAGE;
        const int m0 = m_begin + st * 64;
        const uint32_t n = fastdiv((uint32_t)m0, p.mg_howo, p.sh_howo);
        const uint32_t rem = (uint32_t)m0 - n * (uint32_t)(p.H * p.W);
        const int ho = (int)fastdiv(rem, p.mg_wo, p.sh_wo);
        const int x0 = (int)rem - ho * p.W;
        const int hi = ho + (ky - 1) * d;
        const bool rowok = hi >= 0 && hi < p.H;
        const bf16_t *yb = dy + (size_t)m0 * p.ldy;                                                   // wave-uniform bases
        const bf16_t *xb = a + ((long long)((int)n * p.H + (rowok ? hi : 0)) * p.W + (x0 - d)) * p.lda;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(((vy >> j) & 1u) ? yb + voy[j] : zero, base + (wv * 2 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nxp) {
                const int x = x0 - d + (j * 8 + wv) * 4 + prow;
                const bool ok = rowok && ((vx >> j) & 1u) && x >= 0 && x < p.W;
                glds16(ok ? xb + vox[j] : zero, base + 16384 + (j * 8 + wv) * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4_t acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, li = lane & 15;
    // this wave's six 16-column tiles of the 3 x 128 (kx, ci) columns
    int jkx[6], jct[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { const int jt = wn * 6 + j; jkx[j] = jt >> 3; jct[j] = jt & 7; }

    // Software pipeline over HALF stages (k = 32 pixels each): the fragments of the next half are in flight while the 24 MFMAs
    // of this one run.  With the reads of a whole stage in front of its MFMAs every wave of the workgroup read at the same
    // time and multiplied at the same time -- LDS port (1600 cycles per stage) and matrix pipe (1536) took turns: 3600 cycles
    // per stage.  One barrier per stage, between the halves: it publishes stage st + 1 (whose first half is read right after
    // it) and retires stage st - 1, whose buffer the DMA of stage st + 3 then overwrites; two stages stay in flight behind it.
    // (fragments live in integer vectors: a bf16 vector that is live across a branch is taken apart element by element)
    typedef int v2i32_t __attribute__((ext_vector_type(2)));
    uint4 fa0[4], fb0[6], fa1[4], fb1[6];
    if (ABL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa0[i] = fa1[i] = make_uint4(tid, 1, 2, 3);
#pragma unroll
        for (int j = 0; j < 6; ++j) fb0[j] = fb1[j] = make_uint4(tid, 1, 2, 3);
    }
    auto frag4 = [&](const char *img, int mrow, int t) __attribute__((always_inline)) {
        const int m0 = mrow + 8 * q + (li >> 2);
        const int m1 = m0 + 4;
        const v2i32_t lo = __builtin_bit_cast(v2i32_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m0 * 256 + ((t ^ tr_f(m0)) << 5) + (li & 3) * 8)));
        const v2i32_t hi = __builtin_bit_cast(v2i32_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t *)(img + m1 * 256 + ((t ^ tr_f(m1)) << 5) + (li & 3) * 8)));
        return make_uint4((uint32_t)lo.x, (uint32_t)lo.y, (uint32_t)hi.x, (uint32_t)hi.y);
    };
    auto read_frags = [&](int st, int ks, uint4 (&fa)[4], uint4 (&fb)[6]) __attribute__((always_inline)) {
        if (ABL && (p.dbg & 4)) return;
        const char *imgY = lds + (st % WR_NST) * WR_STAGE, *imgX = imgY + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = frag4(imgY, ks * 32, wm * 4 + i);
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[j] = frag4(imgX, ks * 32 + jkx[j] * d, jct[j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](const uint4 (&fa)[4], const uint4 (&fb)[6]) __attribute__((always_inline)) {
        if (ABL && (p.dbg & 2)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) Mma<bf16_t>::run(fa[i], fb[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // one half stage: the 24 MFMAs on (faM, fbM) with the 20 reads of the NEXT half's fragments (faR, fbR) dealt between them,
    // five per row of six MFMAs -- a burst of 8 x 20 reads in front of the MFMAs keeps the LDS port busy for 640 cycles while the
    // matrix pipe waits for the first fragments, then the pipe runs with the port idle
    auto half = [&](int st, int ks, uint4 (&faR)[4], uint4 (&fbR)[6], const uint4 (&faM)[4], const uint4 (&fbM)[6]) __attribute__((always_inline)) {
        const char *imgY = lds + (st % WR_NST) * WR_STAGE, *imgX = imgY + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!(ABL && (p.dbg & 4))) {
#pragma unroll
                for (int k = 5 * i; k < 5 * i + 5; ++k) {
                    const int hf = k & 1, f = (k < 8 ? k : k - 8) >> 1;
                    const char *img = k < 8 ? imgY : imgX;
                    const int m = ks * 32 + (k < 8 ? 0 : jkx[f] * d) + 8 * q + (li >> 2) + 4 * hf;
                    const int t = k < 8 ? wm * 4 + f : jct[f];
                    const v2i32_t v = __builtin_bit_cast(v2i32_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) v4i16_t *)(img + m * 256 + ((t ^ tr_f(m)) << 5) + (li & 3) * 8)));
                    uint4 &dst = k < 8 ? faR[f] : fbR[f];
                    if (hf == 0) { dst.x = (uint32_t)v.x; dst.y = (uint32_t)v.y; }
                    else { dst.z = (uint32_t)v.x; dst.w = (uint32_t)v.y; }
                }
            }
            if (!(ABL && (p.dbg & 2))) {
#pragma unroll
                for (int j = 0; j < 6; ++j) Mma<bf16_t>::run(faM[i], fbM[j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool dbg = ABL && p.dbg != 0;
    unsigned long long c0 = 0;
    if (dbg) c0 = clock64();
    // vmcnt(n) lgkmcnt(0) for a wave-uniform n: this wave issues gw = 2 + nxp pieces per stage (an s_waitcnt immediate per case)
    const int gw = 2 + nxp;
    auto wait_pieces = [&](int n) __attribute__((always_inline)) {
        switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0)); break;
        case 2: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(2)); break;
        case 3: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(3)); break;
        case 4: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(4)); break;
        case 5: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(5)); break;
        case 6: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(6)); break;
        case 8: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(8)); break;
        case 9: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(9)); break;
        case 10: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(10)); break;
        case 12: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(12)); break;
        default: __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(15)); break;
        }
    };
    if constexpr (MODE == 2) {
        // Ping-pong (the schedule of conv_row_persist_kernel<.., PP>): waves 0-3 (group 0, one per SIMD) multiply stage st while
        // waves 4-7 (group 1) issue their DMA pieces and read their fragments of stage st, then the groups swap -- the matrix
        // pipe of every SIMD always has one wave that does nothing but MFMAs.  Two barriers per stage (the half-periods); a wave
        // holds the fragments of a whole stage (both k halves).  All four stage buffers are in flight: stage st + 4 goes into
        // the buffer of stage st once both groups have read it (behind barrier 2 st + 1).
        for (int st = 0; st < WR_NST && st < nst; ++st) stage(st);
        wait_pieces(min(nst - 1, WR_NST - 1) * gw);
        __builtin_amdgcn_s_barrier();    // stage 0 has landed for every wave
        auto landed = [&](int st) __attribute__((always_inline)) {   // before barrier 2 st + 1: this wave's pieces of stage st + 1
            wait_pieces(max(min(st + WR_NST - 1, nst - 1) - (st + 1), 0) * gw);
        };
        if (wv < 4) {
#pragma unroll 1
            for (int st = 0; st < nst; ++st) {
                read_frags(st, 0, fa0, fb0);
                read_frags(st, 1, fa1, fb1);
                __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));   // lgkmcnt(0) only
                __builtin_amdgcn_s_barrier();          // half-period 2 st: group 1 reads stage st
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa0, fb0);
                mfmas(fa1, fb1);
                landed(st);
                __builtin_amdgcn_s_barrier();          // half-period 2 st + 1: group 1 has read stage st
                if (st + WR_NST < nst) stage(st + WR_NST);
            }
        } else {
#pragma unroll 1
            for (int st = 0; st < nst; ++st) {
                __builtin_amdgcn_s_barrier();          // half-period 2 st
                if (st > 0 && st - 1 + WR_NST < nst) stage(st - 1 + WR_NST);   // into the buffer of stage st - 1
                read_frags(st, 0, fa0, fb0);
                read_frags(st, 1, fa1, fb1);
                landed(st);
                __builtin_amdgcn_s_barrier();          // half-period 2 st + 1
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa0, fb0);
                mfmas(fa1, fb1);
            }
        }
    } else {
    constexpr bool IL = MODE == 1;
    for (int st = 0; st < WR_NST - 1 && st < nst; ++st) stage(st);
    wait_pieces(min(nst - 1, WR_NST - 2) * gw);   // stage 0 has landed once at most the stages issued after it are outstanding
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, fa0, fb0);
    // The two waves of a SIMD (w, w + 4) issue their pieces at different points of the stage: while the vector-memory queue
    // holds one of them in its five global_load_lds (1670 cycles per stage for the whole workgroup when nothing else runs),
    // the other has MFMAs to issue.  "late" waves issue stage st + 3 after the second half's MFMAs instead of before them.
    const bool early = wv < 4;
#pragma unroll 1
    for (int st = 0; st + 1 < nst; ++st) {
        if (IL) half(st, 1, fa1, fb1, fa0, fb0);
        else { read_frags(st, 1, fa1, fb1); mfmas(fa0, fb0); }
        // the waits are the BUILTIN, which hipcc's wait-count pass sees: it knows the second half's fragments have arrived and
        // puts no lgkmcnt wait between the reads below and the MFMAs that do not use them
        wait_pieces(st + 2 < nst ? gw : 0);   // stage st + 1 has landed; st + 2 may be outstanding
        __builtin_amdgcn_s_barrier();
        if (early && st + WR_NST - 1 < nst) stage(st + WR_NST - 1);   // into the buffer of stage st - 1
        if (IL) half(st + 1, 0, fa0, fb0, fa1, fb1);
        else { read_frags(st + 1, 0, fa0, fb0); mfmas(fa1, fb1); }
        if (!early && st + WR_NST - 1 < nst) stage(st + WR_NST - 1);
    }
    if (IL) half(nst - 1, 1, fa1, fb1, fa0, fb0);
    else { read_frags(nst - 1, 1, fa1, fb1); mfmas(fa0, fb0); }
    mfmas(fa1, fb1);
    }
    if (dbg && lane == 0 && blockIdx.x < 256) {
        unsigned long long *o = kd_wgrad_tlog + (blockIdx.x * 8 + wv) * 8;
        o[0] = clock64() - c0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = nst;
    }

    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float *out = p.part + ((size_t)split * 9 + ky * 3 + jkx[j]) * p.Cout * p.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + i * 16 + fq * 4 + r;
                const int ci = ci0 + jct[j] * 16 + frow;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
    }
}

// ---- the same tile on ONE wave per SIMD (round 6; stage loop generated by tools/gen_wgrad_lw.py -> wgrad_lw_body.inc) ----------------
// conv_wgrad_row_kernel's stage takes 3300 cycles for 1536 cycles of MFMAs: its phases (LDS-DMA issue 1670, transposing fragment reads
// 1340, MFMAs 1800 when run alone, tools/wgrad_timeline.py) barely overlap in 8 lock-step waves.  Here 4 waves (wave (wm, wn) = 64 Cout x
// [3 kx x 64 Cin]: 48 accumulator tiles in a[0:191], fragments in v[128:255]) run a hand-dealt stream: per 32-pixel k-step 48 MFMAs with
// the 32 ds_read_b64_tr_b16 of the next k-step and 4-5 LDS-DMA pieces between them, one barrier per stage; the loop is unrolled over the
// four ring slots, so a stage's slot is an immediate of its reads and pieces.  Same decomposition (grid, splits, kernel row per workgroup), the same LDS images and the same K order: bit-identical partial
// slabs.  Cout % 128 == 0, Cin % 8 == 0 (the lanes of a ragged last Cin tile neither load nor store), dil <= 8 (row buffer of 80 rows:
// every wave stages 4 dy + 5 row-buffer pieces per stage).
__global__ __launch_bounds__(256, 1) void conv_wgrad_lw_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(1024))) char lds[WR_NST * WR_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ky = lin % 3; lin /= 3;
    const int tile = lin % p.tiles, split = lin / p.tiles;
    // Tile order: strips of TWO Cin tiles, Cout tiles running inside a strip.  The ~32 workgroups an XCD runs at a time are ~11 consecutive
    // tiles x 3 kernel rows of one split; per stage they pull (Cout tiles) x 16 KB of dy and (Cin tiles) x 3 rows x 20 KB of activations
    // through that XCD's L2.  Cin-fastest order (8 Cin tiles x 1.3 Cout tiles) made that 500 KB per stage, 38 % of all L2 requests missed
    // (FETCH_SIZE 22 GB for 1.6 GB of operands on the 1024 -> 2048 layer); 2 x 5.3 makes it ~205 KB.
    int t_ci, t_co;
    {
        const int tiles_co = p.tiles / p.tiles_ci, strip = tile / (2 * tiles_co), wdt = min(2, p.tiles_ci - 2 * strip), loc = tile - strip * 2 * tiles_co;
        t_co = wdt == 2 ? loc >> 1 : loc;
        t_ci = 2 * strip + (wdt == 2 ? loc & 1 : 0);
    }
    const int co0 = t_co * 128, ci0 = t_ci * 128;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = max(0, (m_end - m_begin) / 64);   // (an empty split writes its zero slab: the generated loop skips itself)
    const int d = p.dil;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    const int q = lane >> 4, li = lane & 15;

    // fragment addresses (ring slot 0): A = dy^T tile wm * 4 + i, B = row-buffer tile (kx, wn * 4 + c) read at row offset kx * dil; the
    // swizzle is keyed on the LDS row (tr_f), which for the shifted row-buffer reads depends on kx * dil and on the half
    uint32_t va[4], vb[12][2];
    {
        const int mA = 8 * q + (li >> 2), fA = tr_f(mA);
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = lbase + mA * 256 + (((wm * 4 + i) ^ fA) << 5) + (li & 3) * 8;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int kx = j >> 2, ct = wn * 4 + (j & 3);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int m = kx * d + 8 * q + (li >> 2) + 4 * hf;
                vb[j][hf] = lbase + 16384 + m * 256 + ((ct ^ tr_f(m)) << 5) + (li & 3) * 8;
            }
        }
    }
    // LDS-DMA sources: byte offsets from the stage's first dy pixel / first row-buffer pixel; wave w stages pieces w, w + 4, ...
    const int prow = lane >> 4, slot = lane & 15;
    uint32_t voy[4], vox[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = (wv + 4 * k) * 4 + prow;
        voy[k] = (uint32_t)(r * p.ldy + co0 + (slot ^ (tr_f(r) << 1)) * 8) * 2u;
    }
    unsigned long long chm[5], stm[4];   // lanes of a row-buffer piece / of a column tile of the store whose input channel exists (ragged last Cin tile)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int r = (wv + 4 * k) * 4 + prow, c = (slot ^ (tr_f(r) << 1)) * 8;
        vox[k] = (uint32_t)(r * p.lda + ci0 + c) * 2u;
        chm[k] = __ballot(ci0 + c < p.Cin);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) stm[c] = __ballot(ci0 + wn * 64 + c * 16 + (lane & 15) < p.Cin);
    const uint32_t vr0 = (uint32_t)(wv * 4 + prow);                    // this lane's buffer row in piece 0 (piece k: + 16 k)
    const uint32_t vzl = lbase + 16384 + wv * 1024 + lane * 16, vzh = vzl + 2 * WR_STAGE;   // its 16 bytes of that piece in ring slot 0 / 2
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4w_t;
    const u32x4w_t vzero = {0u, 0u, 0u, 0u};
    // the fragment addresses a second time, for ring slots 2 and 3 (a ds offset field holds 16 bits: slot 3 does not fit behind slot 0)
    uint32_t wa[4], wb[12][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = va[i] + 2 * WR_STAGE;
#pragma unroll
    for (int j = 0; j < 12; ++j) { wb[j][0] = vb[j][0] + 2 * WR_STAGE; wb[j][1] = vb[j][1] + 2 * WR_STAGE; }

    // stage 0 of the split
    const uint32_t n = fastdiv((uint32_t)m_begin, p.mg_howo, p.sh_howo);
    const uint32_t rem = (uint32_t)m_begin - n * (uint32_t)(p.H * p.W);
    const uint32_t ho = fastdiv(rem, p.mg_wo, p.sh_wo);
    const uint32_t x0 = rem - ho * (uint32_t)p.W;
    const bf16_t *syb = (const bf16_t *)p.dy + (size_t)m_begin * p.ldy;
    const bf16_t *sxb = (const bf16_t *)p.a + ((long long)m_begin + (long long)(ky - 1) * d * p.W - d) * p.lda;   // (only dereferenced for lanes inside the image)
    const uint32_t sdy = 64u * p.ldy * 2u, sdx = 64u * p.lda * 2u;
    const int skyd = (ky - 1) * d;
    const uint32_t send1 = 64u + d, send2 = 64u + 2u * d, sldsw = lbase + wv * 1024;

    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;
    asm volatile(WGRAD_LW_ZERO_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define WGLW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)
#define WGLW_ADDR(P, A, B)                                                                                                                       \
    [P##a0] "v"(A[0]), [P##a1] "v"(A[1]), [P##a2] "v"(A[2]), [P##a3] "v"(A[3]), [P##b0a] "v"(B[0][0]), [P##b0b] "v"(B[0][1]),                    \
    [P##b1a] "v"(B[1][0]), [P##b1b] "v"(B[1][1]), [P##b2a] "v"(B[2][0]), [P##b2b] "v"(B[2][1]), [P##b3a] "v"(B[3][0]), [P##b3b] "v"(B[3][1]),    \
    [P##b4a] "v"(B[4][0]), [P##b4b] "v"(B[4][1]), [P##b5a] "v"(B[5][0]), [P##b5b] "v"(B[5][1]), [P##b6a] "v"(B[6][0]), [P##b6b] "v"(B[6][1]),    \
    [P##b7a] "v"(B[7][0]), [P##b7b] "v"(B[7][1]), [P##b8a] "v"(B[8][0]), [P##b8b] "v"(B[8][1]), [P##b9a] "v"(B[9][0]), [P##b9b] "v"(B[9][1]),    \
    [P##b10a] "v"(B[10][0]), [P##b10b] "v"(B[10][1]), [P##b11a] "v"(B[11][0]), [P##b11b] "v"(B[11][1])
    asm volatile(WGRAD_LW_LOOP_ASM
                 : WGLW_ACC_RW
                 : WGLW_ADDR(v, va, vb), WGLW_ADDR(w, wa, wb),
                   [voy0] "v"(voy[0]), [voy1] "v"(voy[1]), [voy2] "v"(voy[2]), [voy3] "v"(voy[3]),
                   [vox0] "v"(vox[0]), [vox1] "v"(vox[1]), [vox2] "v"(vox[2]), [vox3] "v"(vox[3]), [vox4] "v"(vox[4]),
                   [vr0] "v"(vr0), [vzl] "v"(vzl), [vzh] "v"(vzh), [vzero] "v"(vzero),
                   [syb] "s"(syb), [sxb] "s"(sxb), [sx0] "s"(x0), [sho] "s"(ho), [snst] "s"((uint32_t)nst), [sdy] "s"(sdy), [sdx] "s"(sdx),
                   [sW] "s"((uint32_t)p.W), [sH] "s"((uint32_t)p.H), [sd] "s"((uint32_t)d), [send1] "s"(send1), [send2] "s"(send2), [skyd] "s"(skyd),
                   [sldsw] "s"(sldsw), [schm0] "s"(chm[0]), [schm1] "s"(chm[1]), [schm2] "s"(chm[2]), [schm3] "s"(chm[3]), [schm4] "s"(chm[4])
                 : "memory", "scc", "vcc", WGRAD_LW_CLOBBER_S, WGRAD_LW_CLOBBER_V);
#undef WGLW_ADDR
    // accumulators -> the split's fp32 slab: D[co = 16 i + 4 q + r][ci = li] of tile (i, j = kx * 4 + c)
    float *out0 = p.part + ((size_t)split * 9 + ky * 3) * p.Cout * p.Cin + (size_t)co0 * p.Cin + ci0;
    const float *out1 = out0 + (size_t)p.Cout * p.Cin, *out2 = out1 + (size_t)p.Cout * p.Cin;
    const uint32_t vob = (uint32_t)((wm * 64 + q * 4) * p.Cin + wn * 64 + li) * 4u, cin4 = (uint32_t)p.Cin * 4u;
    asm volatile(WGRAD_LW_STORE_ASM
                 : WGLW_ACC_RW
                 : [vob] "v"(vob), [vcin4] "s"(cin4), [sout0] "s"(out0), [sout1] "s"(out1), [sout2] "s"(out2),
                   [sstm0] "s"(stm[0]), [sstm1] "s"(stm[1]), [sstm2] "s"(stm[2]), [sstm3] "s"(stm[3])
                 : "memory", "v127");
#undef WGLW_ACC_RW
}

__global__ __launch_bounds__(256, 1) void conv_wgrad_pw_lw_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(1024))) char lds[4 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    // 1-D grid, XCD-aware: the tiles of one split are neighbours on ONE XCD (they stream the same dy rows / the same activation rows at
    // the same time); dealt round-robin, the 16 Cin tiles of the 4096 -> 256 layer fetched their split's dy into eight L2s
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lin % p.tiles, split = lin / p.tiles;
    const int t_ci = tile % p.tiles_ci, t_co = tile / p.tiles_ci;
    const int co0 = t_co * 256, ci0 = t_ci * 256;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nst = max(0, (m_end - m_begin) / 32);
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    const int q = lane >> 4, li = lane & 15;

    // fragment addresses in ring slot 0 (and, second copy, slot 2): image wm holds this wave's 128 output channels, image 2 + wn its input channels
    uint32_t va[8], vb[8], wa[8], wb[8];
    {
        const int m0 = 8 * q + (li >> 2), f = tr_f(m0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t o = (uint32_t)(m0 * 256 + ((t ^ f) << 5) + (li & 3) * 8);
            va[t] = lbase + wm * 8192 + o; vb[t] = lbase + (2 + wn) * 8192 + o;
            wa[t] = va[t] + 65536; wb[t] = vb[t] + 65536;
        }
    }
    // LDS-DMA sources: wave w stages pieces 2 w + j (rows (2 w + j) * 4 ..) of every image; byte offsets from the stage's first pixel
    const int prow = lane >> 4, slot = lane & 15;
    uint32_t voy[4], vox[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wv * 2 + j) * 4 + prow, c = (slot ^ (tr_f(r) << 1)) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            voy[j * 2 + h] = (uint32_t)(r * p.ldy + co0 + h * 128 + c) * 2u;
            vox[j * 2 + h] = (uint32_t)(r * p.lda + ci0 + h * 128 + c) * 2u;
        }
    }
    const bf16_t *syb = (const bf16_t *)p.dy + (size_t)m_begin * p.ldy, *sxb = (const bf16_t *)p.a + (size_t)m_begin * p.lda;
    const uint32_t sdy = 32u * p.ldy * 2u, sdx = 32u * p.lda * 2u, sldsw = lbase + wv * 2048;

    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;
    asm volatile(WGRAD_LW_ZERO_ASM : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define WGPW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)
#define WGPW_ADDR(P, A, B)                                                                                                            \
    [P##a0] "v"(A[0]), [P##a1] "v"(A[1]), [P##a2] "v"(A[2]), [P##a3] "v"(A[3]), [P##a4] "v"(A[4]), [P##a5] "v"(A[5]), [P##a6] "v"(A[6]), \
    [P##a7] "v"(A[7]), [P##b0] "v"(B[0]), [P##b1] "v"(B[1]), [P##b2] "v"(B[2]), [P##b3] "v"(B[3]), [P##b4] "v"(B[4]), [P##b5] "v"(B[5]), \
    [P##b6] "v"(B[6]), [P##b7] "v"(B[7])
    asm volatile(WGRAD_PW_LW_LOOP_ASM
                 : WGPW_ACC_RW
                 : WGPW_ADDR(v, va, vb), WGPW_ADDR(w, wa, wb),
                   [voy0] "v"(voy[0]), [voy1] "v"(voy[1]), [voy2] "v"(voy[2]), [voy3] "v"(voy[3]),
                   [vox0] "v"(vox[0]), [vox1] "v"(vox[1]), [vox2] "v"(vox[2]), [vox3] "v"(vox[3]),
                   [syb] "s"(syb), [sxb] "s"(sxb), [snst] "s"((uint32_t)nst), [sdy] "s"(sdy), [sdx] "s"(sdx), [sldsw] "s"(sldsw)
                 : "memory", "scc", WGRAD_PW_LW_CLOBBER_S, WGRAD_PW_LW_CLOBBER_V);
#undef WGPW_ADDR
    float *out = p.part + (size_t)split * p.Cout * p.Cin + (size_t)co0 * p.Cin + ci0;
    const uint32_t vob = (uint32_t)((wm * 128 + q * 4) * p.Cin + wn * 128 + li) * 4u, cin4 = (uint32_t)p.Cin * 4u;
    asm volatile(WGRAD_PW_LW_STORE_ASM
                 : WGPW_ACC_RW
                 : [vob] "v"(vob), [vcin4] "s"(cin4), [sout] "s"(out)
                 : "memory", "v127");
#undef WGPW_ACC_RW
}

// plan of the row-buffer kernel: ~2.5 one-per-CU workgroups per CU, >= 8 stages per split
int fill_splits(int per, int max_splits);   // below, next to wide_plan
void row_plan(long long M, int Cin, int Cout, int &tiles, int &tiles_ci, int &splits, int &rps)
{
    tiles_ci = (Cin + 127) / 128;
    tiles = tiles_ci * ((Cout + 127) / 128);
    const int stages = (int)(M / 64);
    static int target = -1;
    if (target < 0) { const char *e = getenv("KDCC_WGRAD_TARGET"); target = e ? atoi(e) : 0; }   // A/B: floor(n / workgroups per split)
    const int per = tiles * 3, max_splits = (stages + 7) / 8;
    splits = target > 0 ? target / per : fill_splits(per, max_splits);
    splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
    rps = ((stages + splits - 1) / splits) * 64;
    splits = (int)((M + rps - 1) / rps);
}
bool row_eligible(const kd_conv_desc *d)
{
    return d->dtype == KD_BF16 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == d->dil && 2 * d->dil + 64 <= WR_XROWS &&
           d->W % 64 == 0 && d->Cin % 8 == 0 && d->Cout % 8 == 0;
}

__global__ void slab_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, size_t n, int splits,
                                   int accumulate)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

// n % 4 == 0, 16-B aligned slabs: four sums per thread, 16-B loads
__global__ void slab_reduce4_kernel(const float4 *__restrict__ part, float4 *__restrict__ dw, size_t n4, int splits,
                                    int accumulate)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < splits; ++k) {
            const float4 v = part[(size_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (accumulate) { const float4 o = dw[i]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        dw[i] = s;
    }
}

void launch_slab_reduce(const float *part, float *dw, size_t n, int splits, int accumulate, hipStream_t s)
{
    if (n % 4 == 0 && kd_aligned16(part) && kd_aligned16(dw)) {
        const size_t n4 = n / 4;
        const int rb = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3(rb), dim3(256), 0, s, (const float4 *)part, (float4 *)dw, n4, splits, accumulate);
    } else {
        const int rb = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, part, dw, n, splits, accumulate);
    }
}

void launch_tr(dim3 grid, hipStream_t s, const WgradParams &p)
{
    static int nst = -1;
    if (nst < 0) { const char *e = getenv("KDCC_WGRAD_NST"); nst = e ? atoi(e) : 2; }   // A/B hook: 2 | 3 | 4 (measured: 128->128 3x3 at 512x1024 1.70 ms with 2, 2.44 with 3 or 4: occupancy beats depth)
    KD_NOTE_KERNEL("pw_wgrad_tr_kernel");
    if (nst == 2) hipLaunchKernelGGL(pw_wgrad_tr_kernel<2>, grid, dim3(256), 0, s, p);
    else if (nst == 3) hipLaunchKernelGGL(pw_wgrad_tr_kernel<3>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pw_wgrad_tr_kernel<4>, grid, dim3(256), 0, s, p);
}

void plan(int dtype, int M, int Cin, int Cout, int &tiles, int &tiles_ci, int &splits, int &rows_per_split, int taps = 1)
{
    const int krows = IG_ROWB / kd_elem_size(dtype);
    tiles_ci = (Cin + TN - 1) / TN;
    tiles = tiles_ci * ((Cout + TM - 1) / TM);
    const int stages = (M + krows - 1) / krows;
    int want = (1024 + tiles * taps - 1) / (tiles * taps);   // aim for ~1024 workgroups (4 per CU)
    const int max_splits = (stages + 3) / 4;        // keep >= 4 stages per split
    splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    const int st_per = (stages + splits - 1) / splits;
    rows_per_split = st_per * krows;
    splits = (M + rows_per_split - 1) / rows_per_split;
}

// plan of the 256 x 256 tile (bf16 LDS-DMA path): ~3 waves of one-per-CU workgroups, >= 8 stages per split
bool wide_tile_pays(int dtype, int Cin, int Cout)
{
    const long long pad256 = (long long)((Cout + 255) / 256) * ((Cin + 255) / 256) * 65536;
    const long long pad128 = (long long)((Cout + 127) / 128) * ((Cin + 127) / 128) * 16384;
    bool wide = dtype == KD_BF16 && Cin % 8 == 0 && Cout % 8 == 0 && Cout >= 256 && Cin >= 256 &&
                pad256 * 100 <= pad128 * 135;   // (the decoder's 304-channel conv: 2 x 256 vs 3 x 128 columns)
    if (const char *e = getenv("KDCC_WGRAD_WIDE")) wide = wide && e[0] != '0';
    return wide;
}
// Split-K factor for kernels that run ONE workgroup per CU, all of equal length (the 256 x 256 and the row-buffer weight-gradient
// tiles): the launch takes ceil(W / 256) rounds of the 256 CUs with W = splits * per workgroups, so W should sit just BELOW a
// multiple of 256, and the fewest rounds that fill the chip win (fewer partial slabs to write and reduce, fewer pipeline fills).
// Returns the first floor(256 k / per), k = 1..4, that fills >= 95 % of its k rounds, else the best fill.  Round 2 aimed at "about
// 640 / 768 workgroups" rounded UP: 2.5 rounds with the last half empty, or one workgroup into a fourth round.  Measured
// (tools/bench_wgrad.py, 4 images): 3x3 128->128 at 512x1024 1.01 -> 0.85 ms, 256->256 at 512x1024 2.99 -> 2.50, 512->512 0.78 -> 0.73,
// 1024->512 1.44 -> 1.27, 304->256 4.45 -> 3.87; 1x1 512->512 0.220 -> 0.114, 4096->256 0.371 -> 0.313; mode B 24.5 -> 25.8 img/s.
int fill_splits(int per, int max_splits)
{
    int splits = 1;
    double best = -1.0;
    for (int k = 1; k <= 4; ++k) {
        int sp = (256 * k) / per;
        sp = sp < 1 ? 1 : (sp > max_splits ? max_splits : sp);
        const long long w = (long long)sp * per;
        const double fill = (double)w / (double)(((w + 255) / 256) * 256);
        if (fill >= 0.95) return sp;
        if (fill > best) { best = fill; splits = sp; }
    }
    return splits;
}

// conv_wgrad_pw_lw_kernel instead of conv_wgrad_wide_kernel<true>: whole 256 x 256 tiles and whole 32-pixel stages.  Measured at 8 images
// (tools/wgrad_lw_check.py, ms incl. the reduce): 2048->4096 4.49 -> 3.28, 1024->2048 1.19 -> 0.80, 2048->1024 1.17 -> 0.78, 4096->256 0.85 -> 0.66,
// 512->1024 0.34 -> 0.22, 1280->256 0.24 -> 0.18, 512->512 0.20 -> 0.15.  (With the workgroups dealt round-robin over the XCDs, as the 8-wave
// kernel's 2-D grid is, the short splits were SLOWER than the 8-wave kernel, x 0.90-0.92: the tiles of a split fetched their shared rows into
// eight L2s.)  mode: KDCC_WGRAD_PW_LW, 0 = never.
bool pw_lw_pays(long long M, int Cin, int Cout, int rps, int mode)
{
    return mode && Cin % 256 == 0 && Cout % 256 == 0 && M % 64 == 0 && rps % 32 == 0;
}

void wide_plan(long long M, int Cin, int Cout, int taps, int &tiles, int &tiles_ci, int &splits, int &rps)
{
    tiles_ci = (Cin + 255) / 256;
    tiles = tiles_ci * ((Cout + 255) / 256);
    const int stages = (int)((M + 63) / 64);
    static int target = -1;
    if (target < 0) { const char *e = getenv("KDCC_WGRAD_WIDE_TARGET"); target = e ? atoi(e) : 0; }   // A/B: floor(n / workgroups per split)
    const int max_splits = (stages + 7) / 8;
    const int want = target > 0 ? target / (tiles * taps) : fill_splits(tiles * taps, max_splits);
    splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    rps = ((stages + splits - 1) / splits) * 64;
    splits = (int)((M + rps - 1) / rps);
}

}  // namespace

extern "C" size_t kd_pw_wgrad_workspace(int32_t M, int32_t Cin, int32_t Cout)
{
    // dtype-independent upper bound: plan with the smaller K stage (f32: 32 rows) gives the larger split count
    int tiles, tiles_ci, s0, s1, rps;
    plan(KD_F32, M, Cin, Cout, tiles, tiles_ci, s0, rps);
    plan(KD_BF16, M, Cin, Cout, tiles, tiles_ci, s1, rps);
    int splits = s0 > s1 ? s0 : s1;
    const int stages = (M + 63) / 64;
    const int wsplits = (stages + 7) / 8 < 768 ? (stages + 7) / 8 : 768;   // the wide-tile plan never splits finer
    if (wsplits > splits) splits = wsplits;
    return (size_t)splits * Cout * Cin * sizeof(float);
}

extern "C" int kd_pw_wgrad(int32_t dtype, int32_t M, int32_t Cin, int32_t Cout, const void *a, int32_t lda,
                           const void *dy, int32_t ldy, float *dw, int32_t accumulate, void *workspace,
                           size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(a && dy && dw && workspace, KD_ERR_INVALID, "kd_pw_wgrad: null argument");
    KD_REQUIRE(dtype == KD_F32 || dtype == KD_BF16, KD_ERR_INVALID, "kd_pw_wgrad: bad dtype");
    KD_REQUIRE(M > 0 && Cin > 0 && Cout > 0, KD_ERR_INVALID, "kd_pw_wgrad: bad shape");
    const int es = kd_elem_size(dtype);
    KD_REQUIRE(lda >= Cin && ldy >= Cout && (lda * es) % 16 == 0 && (ldy * es) % 16 == 0 && kd_aligned16(a) &&
                   kd_aligned16(dy),
               KD_ERR_INVALID, "kd_pw_wgrad: operands must be 16-B aligned with 16-B multiple row strides");
    int tiles, tiles_ci, splits, rps;
    plan(dtype, M, Cin, Cout, tiles, tiles_ci, splits, rps);
    const bool wide = wide_tile_pays(dtype, Cin, Cout);   // the 256 x 256 tile of kd_conv2d_wgrad (444 -> 624 TFLOP/s on large layers)
    if (wide) wide_plan(M, Cin, Cout, 1, tiles, tiles_ci, splits, rps);
    KD_REQUIRE(workspace_bytes >= (size_t)splits * Cout * Cin * sizeof(float), KD_ERR_WORKSPACE,
               "kd_pw_wgrad: workspace %zu < %zu", workspace_bytes, (size_t)splits * Cout * Cin * sizeof(float));
    WgradParams p;
    p.a = a; p.dy = dy; p.part = (float *)workspace;
    p.M = M; p.Cin = Cin; p.Cout = Cout; p.lda = lda; p.ldy = ldy;
    p.tiles_ci = tiles_ci; p.rows_per_split = rps; p.splits = splits; p.tiles = tiles;
    { static int wd = -1; if (wd < 0) wd = KD_TUNING_ENV_INT("KDCC_WGRAD_DBG"); p.dbg = wd; }   // phase clocks: tuning build only
    p.geom = 0; p.kw = 1; p.H = p.W = p.Ho = p.Wo = 0; p.stride = 1; p.pad = 0; p.dil = 1;
    p.mg_howo = p.sh_howo = p.mg_wo = p.sh_wo = 0;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    static int pwlw = -1;
    if (pwlw < 0) { const char *e = getenv("KDCC_WGRAD_PW_LW"); pwlw = e ? atoi(e) : 1; }   // A/B: 0 = conv_wgrad_wide_kernel (8 waves); bit-identical
    if (wide && !p.dbg && pw_lw_pays(M, Cin, Cout, rps, pwlw)) {
        KD_NOTE_KERNEL("conv_wgrad_pw_lw_kernel");
        hipLaunchKernelGGL(conv_wgrad_pw_lw_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, p);
    }
    else if (wide) { KD_NOTE_KERNEL("conv_wgrad_wide_kernel"); hipLaunchKernelGGL(conv_wgrad_wide_kernel<true>, grid, dim3(512), 0, s, p); }
    else if (dtype == KD_BF16 && Cin % 8 == 0 && Cout % 8 == 0) launch_tr(grid, s, p);
    else if (dtype == KD_BF16) { KD_NOTE_KERNEL("pw_wgrad_kernel<bf16>"); hipLaunchKernelGGL(pw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, p); }
    else { KD_NOTE_KERNEL("pw_wgrad_kernel<f32>"); hipLaunchKernelGGL(pw_wgrad_kernel<float>, grid, dim3(256), 0, s, p); }
    KD_CHECK_LAUNCH("kd_pw_wgrad");
    launch_slab_reduce((const float *)workspace, dw, (size_t)Cout * Cin, splits, accumulate, s);
    KD_CHECK_LAUNCH("kd_pw_wgrad(reduce)");
    return KD_OK;
}

// ---- general convolution weight gradient (kd_conv2d_wgrad): the same TN GEMM per tap, A rows gathered through the conv's
// pixel map (stride, padding, dilation; zeros outside the image).  grid = (Cout x Cin tiles, pixel splits, taps).
namespace {
void fastdiv_magic(uint32_t d, uint32_t &magic, uint32_t &shift)
{
    // n / d == (umulhi(n, magic) + n) >> shift for 0 <= n < 2^31 (round-up method, 33-bit magic with the top bit implicit)
    shift = 0;
    while ((1ull << shift) < d) ++shift;
    magic = (uint32_t)(((1ull << 32) * ((1ull << shift) - d)) / d + 1);
}
}  // namespace

extern "C" int kd_debug_wgrad_tlog(unsigned long long *dst, size_t bytes)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(kd_wgrad_tlog), bytes < sizeof(kd_wgrad_tlog) ? bytes : sizeof(kd_wgrad_tlog), 0,
                               hipMemcpyDeviceToHost) == hipSuccess ? KD_OK : KD_ERR_HIP;
}

extern "C" size_t kd_conv2d_wgrad_workspace(const kd_conv_desc *d)
{
    if (!d) return 0;
    const int M = d->N * d->Ho * d->Wo, taps = d->kh * d->kw;
    int tiles, tiles_ci, s0, s1, rps;
    plan(KD_F32, M, d->Cin, d->Cout, tiles, tiles_ci, s0, rps, taps);
    plan(KD_BF16, M, d->Cin, d->Cout, tiles, tiles_ci, s1, rps, taps);
    int splits = s0 > s1 ? s0 : s1;
    const int stages = (M + 63) / 64;
    const int wsplits = (stages + 7) / 8 < 768 ? (stages + 7) / 8 : 768;   // the wide-tile plan never splits finer
    if (wsplits > splits) splits = wsplits;
    if (row_eligible(d)) {
        int rt, rtc, rs, rr;
        row_plan(M, d->Cin, d->Cout, rt, rtc, rs, rr);
        if (rs > splits) splits = rs;
    }
    return (size_t)splits * taps * d->Cout * d->Cin * sizeof(float);
}

extern "C" int kd_conv2d_wgrad(const kd_conv_desc *d, const void *x, const void *dy, int32_t ld_dy, float *dw,
                               int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(d && x && dy && dw && workspace, KD_ERR_INVALID, "kd_conv2d_wgrad: null argument");
    KD_REQUIRE(d->dtype == KD_F32 || d->dtype == KD_BF16, KD_ERR_INVALID, "kd_conv2d_wgrad: bad dtype");
    KD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 && d->stride >= 1 &&
                   d->dil >= 1 && d->pad >= 0,
               KD_ERR_INVALID, "kd_conv2d_wgrad: bad descriptor");
    const int ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    KD_REQUIRE(ho == d->Ho && wo == d->Wo && ho > 0 && wo > 0, KD_ERR_INVALID, "kd_conv2d_wgrad: Ho/Wo (%d,%d) inconsistent, expect (%d,%d)",
               d->Ho, d->Wo, ho, wo);
    const int es = kd_elem_size(d->dtype);
    KD_REQUIRE(d->ldx >= d->Cin && ld_dy >= d->Cout && (d->ldx * es) % 16 == 0 && (ld_dy * es) % 16 == 0 && kd_aligned16(x) &&
                   kd_aligned16(dy),
               KD_ERR_INVALID, "kd_conv2d_wgrad: operands must be 16-B aligned with 16-B multiple pixel strides");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    KD_REQUIRE(M < (1ll << 31) && (long long)d->N * d->H * d->W < (1ll << 31), KD_ERR_UNSUPPORTED, "kd_conv2d_wgrad: too many pixels");
    const int taps = d->kh * d->kw;
    KD_REQUIRE(taps <= 65535, KD_ERR_UNSUPPORTED, "kd_conv2d_wgrad: kernel too large");
    int tiles, tiles_ci, splits, rps;
    plan(d->dtype, (int)M, d->Cin, d->Cout, tiles, tiles_ci, splits, rps, taps);
    bool wide = wide_tile_pays(d->dtype, d->Cin, d->Cout);
    // A/B hook KDCC_WGRAD_ROW: 0 = never, 1 = only where the 256 x 256 tile is not chosen, 2 = wherever eligible (default: it is
    // faster on every 3x3 layer of the net, tools/bench_wgrad.py at 4 images: 128->128 1.70 -> 1.06 ms, 304->256 7.25 -> 4.69,
    // 256->256 1.04 -> 0.88, 512->512 0.87 -> 0.78, 512->1024 dil 2 1.60 -> 1.43, 64->128 1.51 -> 0.83)
    static int rowmode = -1;
    if (rowmode < 0) { const char *e = getenv("KDCC_WGRAD_ROW"); rowmode = e ? atoi(e) : 2; }
    const bool row = row_eligible(d) && (rowmode == 2 || (rowmode == 1 && !wide));
    if (row) {
        wide = false;
        row_plan(M, d->Cin, d->Cout, tiles, tiles_ci, splits, rps);
    } else if (wide) {
        wide_plan(M, d->Cin, d->Cout, taps, tiles, tiles_ci, splits, rps);
    }
    const size_t need = (size_t)splits * taps * d->Cout * d->Cin * sizeof(float);
    KD_REQUIRE(workspace_bytes >= need, KD_ERR_WORKSPACE, "kd_conv2d_wgrad: workspace %zu < %zu", workspace_bytes, need);
    WgradParams p;
    p.a = x; p.dy = dy; p.part = (float *)workspace;
    p.M = (int)M; p.Cin = d->Cin; p.Cout = d->Cout; p.lda = d->ldx; p.ldy = ld_dy;
    p.tiles_ci = tiles_ci; p.rows_per_split = rps; p.splits = splits; p.tiles = tiles;
    { static int wd = -1; if (wd < 0) wd = KD_TUNING_ENV_INT("KDCC_WGRAD_DBG"); p.dbg = wd; }   // phase clocks: tuning build only
    p.geom = !(taps == 1 && d->stride == 1 && d->pad == 0);
    static int wide_general = -1;
    if (wide_general < 0) { const char *e = getenv("KDCC_WGRAD_WIDE_GENERAL"); wide_general = e && e[0] == '1'; }   // A/B: the general staging on 1x1 too
    p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    fastdiv_magic((uint32_t)(d->Ho * d->Wo), p.mg_howo, p.sh_howo);
    fastdiv_magic((uint32_t)d->Wo, p.mg_wo, p.sh_wo);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)tiles, (unsigned)splits, (unsigned)taps);
    if (row) {
        fastdiv_magic((uint32_t)(d->H * d->W), p.mg_howo, p.sh_howo);
        fastdiv_magic((uint32_t)d->W, p.mg_wo, p.sh_wo);
        p.tiles = tiles;
        static int lwmode = -1;
        if (lwmode < 0) { const char *e = getenv("KDCC_WGRAD_LW"); lwmode = (e && e[0] == '0') ? 0 : 1; }   // A/B: 0 = conv_wgrad_row_kernel (8 waves); bit-identical
        if (lwmode && !p.dbg && d->Cout % 128 == 0 && d->dil <= 8) {   // (Cin % 8 == 0: row_eligible; a ragged last Cin tile is masked)
            KD_NOTE_KERNEL("conv_wgrad_lw_kernel");
            hipLaunchKernelGGL(conv_wgrad_lw_kernel, dim3((unsigned)(tiles * splits * 3)), dim3(256), 0, s, p);
        } else {
        KD_NOTE_KERNEL("conv_wgrad_row_kernel");
#ifdef KDCC_TUNING
        static int il = -1;
        if (il < 0) { const char *e = getenv("KDCC_WGRAD_IL"); il = e ? atoi(e) : 1; }   // A/B: 0 = reads in front of the MFMAs, 1 = interleaved (shipped), 2 = ping-pong
        const dim3 grid((unsigned)(tiles * splits * 3));
        if (p.dbg && il == 2) hipLaunchKernelGGL((conv_wgrad_row_kernel<true, 2>), grid, dim3(512), 0, s, p);
        else if (p.dbg && il == 1) hipLaunchKernelGGL((conv_wgrad_row_kernel<true, 1>), grid, dim3(512), 0, s, p);
        else if (p.dbg) hipLaunchKernelGGL((conv_wgrad_row_kernel<true, 0>), grid, dim3(512), 0, s, p);
        else if (il == 2) hipLaunchKernelGGL((conv_wgrad_row_kernel<false, 2>), grid, dim3(512), 0, s, p);
        else if (il == 0) hipLaunchKernelGGL((conv_wgrad_row_kernel<false, 0>), grid, dim3(512), 0, s, p);
        else
#endif
        hipLaunchKernelGGL((conv_wgrad_row_kernel<false, 1>), dim3((unsigned)(tiles * splits * 3)), dim3(512), 0, s, p);
        }
    } else if (wide) {
        static int pwlw = -1;
        if (pwlw < 0) { const char *e = getenv("KDCC_WGRAD_PW_LW"); pwlw = e ? atoi(e) : 1; }   // A/B: 0 = conv_wgrad_wide_kernel (8 waves); bit-identical
        if (!p.dbg && !p.geom && !wide_general && taps == 1 && pw_lw_pays(M, d->Cin, d->Cout, rps, pwlw)) {
            KD_NOTE_KERNEL("conv_wgrad_pw_lw_kernel");
            hipLaunchKernelGGL(conv_wgrad_pw_lw_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, p);
        } else {
        KD_NOTE_KERNEL("conv_wgrad_wide_kernel");
        if (!p.geom && !wide_general) hipLaunchKernelGGL(conv_wgrad_wide_kernel<true>, grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_wide_kernel<false>, grid, dim3(512), 0, s, p);
        }
    }
    else if (d->dtype == KD_BF16 && d->Cin % 8 == 0 && d->Cout % 8 == 0) launch_tr(grid, s, p);
    else if (d->dtype == KD_BF16) { KD_NOTE_KERNEL("pw_wgrad_kernel<bf16>"); hipLaunchKernelGGL(pw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, p); }
    else { KD_NOTE_KERNEL("pw_wgrad_kernel<f32>"); hipLaunchKernelGGL(pw_wgrad_kernel<float>, grid, dim3(256), 0, s, p); }
    KD_CHECK_LAUNCH("kd_conv2d_wgrad");
    const size_t n = (size_t)d->Cout * d->Cin * taps;
    const int rb = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (taps == 1) {
        launch_slab_reduce((const float *)workspace, dw, (size_t)d->Cout * d->Cin, splits, accumulate, s);
    } else if (taps == 9) {
        const size_t plane = (size_t)d->Cout * d->Cin;
        const int rb9 = (int)((plane + 255) / 256 > 4096 ? 4096 : (plane + 255) / 256);
        hipLaunchKernelGGL(slab_reduce_taps_cc_kernel<9>, dim3(rb9), dim3(256), 0, s, (const float *)workspace, dw, plane, splits,
                           accumulate);
    } else
        hipLaunchKernelGGL(slab_reduce_taps_kernel, dim3(rb), dim3(256), 0, s, (const float *)workspace, dw, d->Cout, d->Cin, taps,
                           splits, accumulate);
    KD_CHECK_LAUNCH("kd_conv2d_wgrad(reduce)");
    return KD_OK;
}

// Depthwise k x k dilated convolution, NHWC (kd_dwconv_fwd / kd_dwconv_wgrad).
// The shipped cheap conv is 9x9, dilation 5, padding 20 (effective 41x41 field): the
// image splits into dil*dil residue classes, each an independent dense k x k stencil on
// the lattice {r + dil*l}.  A thread owns 4 channels and an S x R patch of lattice
// outputs, so each input vector (4 channels, 8/16 B) it loads feeds up to k*S FMAs and
// every weight quad k*R of them; 16 consecutive lanes cover 64 contiguous channels
// (128/256 B per pixel per load instruction).  Weights sit in LDS tap-major.
// VALU/L1-bound (81 FMA per output element).  These register kernels are the fp32 parity path, the 3x3 / odd-channel
// path and the input gradient that carries a BN/ReLU-mask + residual epilogue; bf16 9x9 forward, plain input gradient
// and weight gradient go to the matrix-core kernels of dwconv_mfma.hip (1.6-1.8x faster), which kd_dwconv_fwd /
// kd_dwconv_wgrad try first.
// Measured (MI355X, 4096 ch @128x256, bf16): 46 TFLOP/s fwd, 30 wgrad; sustained v_pk_fma_f32 rate of the chip is 127-138
// TFLOP/s (tools/ubench/valu_rate.hip), the kernel's own VALU mix (62 % pk_fma) bounds it at ~75.  An LDS-DMA row-ring
// variant (3x instead of 10x input over-fetch, no masks) was built and measured slower (31-39 TFLOP/s: idle lanes at the
// lattice/tile edges outweigh the cleaner load path), so the register-only form stays.
#include <stdlib.h>
#include <string.h>

#include "kd_common.h"

namespace {

constexpr int CB = 64;    // channels per block
constexpr int CQ = 16;    // channel quads per block (threads along channels)
constexpr int WTR = 8;    // lattice cols per thread in the weight-gradient kernel

struct DwParams {
    const void *x;
    const float *w;      // [k*k][C]
    const float *bias;   // (C) or null
    void *y;
    int N, H, W, C, pad, dil, ldx, ldy;
    int LH, LW;          // lattice extent upper bounds: ceil(H/dil), ceil(W/dil)
    int tiles_h, tiles_w;
    int gx, gz;          // logical grid extents: tile groups, N*dil*dil (channel blocks are the slowest index)
    kd_dw_epilogue ep;
};

template <typename T> __device__ __forceinline__ void ld4(const T *p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float *p, float (&v)[4])
{
    const float4 a = *(const float4 *)p;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t *p, float (&v)[4])
{
    const uint2 u = *(const uint2 *)p;
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
// Explicit 2-wide float vectors: the FMAs are written on register PAIRS from the start, so hipcc emits v_pk_fma_f32
// on values that already sit in adjacent VGPRs (the scalar formulation was SLP-packed after the fact, at the price of
// ~0.7 v_mov per v_pk_fma to shuffle operands into pairs).  A loaded 4-channel vector becomes two pairs; out-of-image
// samples are zeroed by AND-ing the still-packed words with an all-ones / zero mask (2 ops instead of 4 selects).
typedef float v2f __attribute__((ext_vector_type(2)));
struct Quad { v2f lo, hi; };   // channels (0,1) and (2,3)
__device__ __forceinline__ Quad ldq(const float *p, uint32_t m)
{
    const float4 a = *(const float4 *)p;
    Quad q;
    q.lo = (v2f){__uint_as_float(__float_as_uint(a.x) & m), __uint_as_float(__float_as_uint(a.y) & m)};
    q.hi = (v2f){__uint_as_float(__float_as_uint(a.z) & m), __uint_as_float(__float_as_uint(a.w) & m)};
    return q;
}
__device__ __forceinline__ Quad ldq(const bf16_t *p, uint32_t m)
{
    uint2 u = *(const uint2 *)p;
    u.x &= m; u.y &= m;
    Quad q;
    q.lo = (v2f){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
    q.hi = (v2f){__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    return q;
}
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ Quad unpackq(const float4 &a, uint32_t m)
{
    Quad q;
    q.lo = (v2f){__uint_as_float(__float_as_uint(a.x) & m), __uint_as_float(__float_as_uint(a.y) & m)};
    q.hi = (v2f){__uint_as_float(__float_as_uint(a.z) & m), __uint_as_float(__float_as_uint(a.w) & m)};
    return q;
}
__device__ __forceinline__ Quad unpackq(uint2 u, uint32_t m)
{
    u.x &= m; u.y &= m;
    Quad q;
    q.lo = (v2f){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
    q.hi = (v2f){__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    return q;
}
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void st4(float *p, const float (&v)[4]) { *(float4 *)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t *p, const float (&v)[4])
{
    *(uint2 *)p = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// grid: x = ceil(tiles_h*tiles_w / 16), y = ceil(C / 64), z = N * dil * dil
template <typename T, int K, int TS, int TR>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwParams p)
{
    __shared__ __attribute__((aligned(16))) float wl[K * K * CB];
    const int tid = threadIdx.x;
    const int cq = tid & (CQ - 1), tt = tid >> 4;
    // 1-D grid, XCD-aware: each XCD gets a contiguous range of (tile group fastest, then residue class, then channel
    // block), so blocks sharing halo rows / the same 64-channel slab run on one L2
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lin % p.gx; lin /= p.gx;
    const int bz = lin % p.gz;
    const int by = lin / p.gz;
    const int c0 = by * CB;
    for (int i = tid; i < K * K * CB; i += 256) {
        const int t = i / CB, c = c0 + (i - t * CB);
        wl[i] = c < p.C ? p.w[(size_t)t * p.C + c] : 0.f;
    }
    __syncthreads();

    const int dd = p.dil * p.dil;
    const int n = bz / dd, rc = bz - n * dd;
    const int rh = rc / p.dil, rw = rc - rh * p.dil;
    const int tile = bx * 16 + tt;
    const int c = c0 + cq * 4;
    if (tile >= p.tiles_h * p.tiles_w || c >= p.C) return;
    const int th = tile / p.tiles_w, tw = tile - th * p.tiles_w;
    const int h0 = rh + p.dil * (th * TS), w0 = rw + p.dil * (tw * TR);
    if (h0 >= p.H || w0 >= p.W) return;

    Quad acc2[TS][TR];
    {
        Quad b0;
        b0.lo = (v2f){0.f, 0.f}; b0.hi = (v2f){0.f, 0.f};
        if (p.bias) { b0.lo = (v2f){p.bias[c], p.bias[c + 1]}; b0.hi = (v2f){p.bias[c + 2], p.bias[c + 3]}; }
#pragma unroll
        for (int s = 0; s < TS; ++s)
#pragma unroll
            for (int r = 0; r < TR; ++r) acc2[s][r] = b0;
    }

    // per-thread column table (same for every input row): clamped pixel offset and all-ones / zero validity mask
    constexpr int NX = TR + K - 1;
    int coff[NX];
    uint32_t cmask = 0;
#pragma unroll
    for (int idx = 0; idx < NX; ++idx) {
        const int win = w0 - p.pad + idx * p.dil;
        coff[idx] = (win < 0 ? 0 : (win >= p.W ? p.W - 1 : win)) * p.ldx;
        cmask |= (win >= 0 && win < p.W) ? (1u << idx) : 0u;
    }
    const T *xb = (const T *)p.x + (size_t)n * p.H * p.W * p.ldx + c;
    // Software prefetch: the next input row's (still packed) loads are issued as soon as the current row has been
    // unpacked, so they fly under this row's 2*K*TR packed FMAs.  Row addresses are clamped into the image (a branch
    // per load would serialise them behind vmcnt(0) waits); rows outside the image are skipped wave-uniformly or zeroed
    // through the mask.
    typedef typename Raw4<T>::type raw_t;
    auto issue = [&](int rho, raw_t (&dst)[NX]) {
        const int hin = h0 - p.pad + rho * p.dil;
        const int hc = hin < 0 ? 0 : (hin >= p.H ? p.H - 1 : hin);
        const T *xr = xb + (size_t)hc * p.W * p.ldx;
#pragma unroll
        for (int idx = 0; idx < NX; ++idx) dst[idx] = *(const raw_t *)(xr + coff[idx]);
    };
    raw_t raw[NX];
    issue(0, raw);
#pragma unroll 1
    for (int rho = 0; rho < TS + K - 1; ++rho) {
        const int hin = h0 - p.pad + rho * p.dil;
        const bool rok = hin >= 0 && hin < p.H;
        const uint32_t rmask = rok ? 0xffffffffu : 0u;
        Quad xv[NX];
#pragma unroll
        for (int idx = 0; idx < NX; ++idx) xv[idx] = unpackq(raw[idx], rmask & (uint32_t)(-(int)((cmask >> idx) & 1u)));
        if (rho + 1 < TS + K - 1) issue(rho + 1, raw);
        if (__ballot(rok) == 0ull) continue;   // the whole wave's row is outside the image
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const int i = rho - s;  // tap row feeding output row s from input row rho
            if (i < 0 || i >= K) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float4 w4 = *(const float4 *)&wl[(i * K + j) * CB + cq * 4];
                const v2f wlo = (v2f){w4.x, w4.y}, whi = (v2f){w4.z, w4.w};
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    acc2[s][r].lo = fma2(xv[r + j].lo, wlo, acc2[s][r].lo);
                    acc2[s][r].hi = fma2(xv[r + j].hi, whi, acc2[s][r].hi);
                }
            }
        }
    }
    float acc[TS][TR][4];
#pragma unroll
    for (int s = 0; s < TS; ++s)
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            acc[s][r][0] = acc2[s][r].lo[0]; acc[s][r][1] = acc2[s][r].lo[1];
            acc[s][r][2] = acc2[s][r].hi[0]; acc[s][r][3] = acc2[s][r].hi[1];
        }

    T *yb = (T *)p.y + (size_t)n * p.H * p.W * p.ldy + c;
    const kd_dw_epilogue &e = p.ep;
    float ms[4] = {1.f, 1.f, 1.f, 1.f};
    if (e.mask && e.mask_scale) { ms[0] = e.mask_scale[c]; ms[1] = e.mask_scale[c + 1]; ms[2] = e.mask_scale[c + 2]; ms[3] = e.mask_scale[c + 3]; }
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int h = h0 + s * p.dil;
        if (h >= p.H) continue;
        // epilogue operands for the whole output row are loaded first (independent loads in flight together)
        float e_pre[TR][4], e_msk[TR][4], e_post[TR][4];
        size_t pix[TR];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const int w = w0 + r * p.dil;
            pix[r] = ((size_t)n * p.H + h) * p.W + (w < p.W ? w : p.W - 1);
        }
        if (e.res_pre) {
#pragma unroll
            for (int r = 0; r < TR; ++r) ld4<T>((const T *)e.res_pre + pix[r] * e.ld_res_pre + c, e_pre[r]);
        }
        if (e.mask) {
#pragma unroll
            for (int r = 0; r < TR; ++r) ld4<T>((const T *)e.mask + pix[r] * e.ld_mask + c, e_msk[r]);
        }
        if (e.res_post) {
#pragma unroll
            for (int r = 0; r < TR; ++r) ld4<T>((const T *)e.res_post + pix[r] * e.ld_res_post + c, e_post[r]);
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const int w = w0 + r * p.dil;
            if (e.res_pre) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s][r][q] += e_pre[r][q];
            }
            if (e.mask) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s][r][q] = e_msk[r][q] > 0.f ? acc[s][r][q] * ms[q] : 0.f;
            }
            if (e.res_post) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s][r][q] += e_post[r][q];
            }
            if (w < p.W) st4(yb + ((size_t)h * p.W + w) * p.ldy, acc[s][r]);
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------
// thread = (channel quad, lattice row strip); block = 16 quads x 16 strips, blockIdx.z = tap row i.
// Each thread keeps K taps x 4 channels of partial sums over its strip, the block reduces the 16
// strips through LDS and writes one partial [K][64] slab; a second kernel sums slabs in order.
struct DwWgradParams {
    const void *x;
    const void *dy;
    float *part;  // [gridDim.x][K*K][C]
    int N, H, W, C, pad, dil, ldx, ld_dy;
    int LH, LW, nstrips, nslabs;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const DwWgradParams p)
{
    __shared__ float red[16][K][CB + 4];
    const int tid = threadIdx.x;
    const int cq = tid & (CQ - 1), tt = tid >> 4;
    // 1-D grid, XCD-aware, tap row fastest: the K blocks that read the same dy rows (and x rows one lattice row apart)
    // are neighbours on one XCD instead of K far-apart sweeps over both tensors (measured 9x HBM over-fetch before)
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int i = lin % K; lin /= K;       // tap row
    const int bx = lin % p.nslabs;
    const int by = lin / p.nslabs;
    const int c = by * CB + cq * 4;
    const int strip = bx * 16 + tt;

    Quad acc2[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { acc2[j].lo = (v2f){0.f, 0.f}; acc2[j].hi = (v2f){0.f, 0.f}; }

    if (strip < p.nstrips && c < p.C) {
        // strip -> (n, rh, rw, lh)
        const int dd = p.dil * p.dil;
        int t = strip;
        const int lh = t % p.LH; t /= p.LH;
        const int rc = t % dd;
        const int n = t / dd;
        const int rh = rc / p.dil, rw = rc - rh * p.dil;
        const int h = rh + p.dil * lh;
        const int hin = h - p.pad + i * p.dil;
        if (h < p.H && hin >= 0 && hin < p.H) {
            const T *dyr = (const T *)p.dy + ((size_t)n * p.H + h) * p.W * p.ld_dy + c;
            const T *xr = (const T *)p.x + ((size_t)n * p.H + hin) * p.W * p.ldx + c;
            typedef typename Raw4<T>::type raw_t;
            constexpr int NX = WTR + K - 1;
            // chunks of WTR lattice columns; the next chunk's (packed) loads are issued as soon as this chunk is unpacked
            auto issue = [&](int lw0, raw_t (&gd)[WTR], raw_t (&xd)[NX]) {
                const int w0 = rw + p.dil * lw0;
#pragma unroll
                for (int r = 0; r < WTR; ++r) {
                    const int w = w0 + r * p.dil;
                    gd[r] = *(const raw_t *)(dyr + (size_t)(w >= p.W ? p.W - 1 : w) * p.ld_dy);
                }
#pragma unroll
                for (int idx = 0; idx < NX; ++idx) {
                    const int win = w0 - p.pad + idx * p.dil;
                    xd[idx] = *(const raw_t *)(xr + (size_t)(win < 0 ? 0 : (win >= p.W ? p.W - 1 : win)) * p.ldx);
                }
            };
            int nchunks = 0;
            for (int lw0 = 0; lw0 < p.LW && rw + p.dil * lw0 < p.W; lw0 += WTR) ++nchunks;
            raw_t gr[WTR], xr4[NX];
            if (nchunks > 0) issue(0, gr, xr4);
#pragma unroll 1
            for (int ch = 0; ch < nchunks; ++ch) {
                const int w0 = rw + p.dil * ch * WTR;
                Quad g[WTR], xv[NX];
#pragma unroll
                for (int r = 0; r < WTR; ++r) g[r] = unpackq(gr[r], (w0 + r * p.dil < p.W) ? 0xffffffffu : 0u);
#pragma unroll
                for (int idx = 0; idx < NX; ++idx) {
                    const int win = w0 - p.pad + idx * p.dil;
                    xv[idx] = unpackq(xr4[idx], (win >= 0 && win < p.W) ? 0xffffffffu : 0u);
                }
                if (ch + 1 < nchunks) issue((ch + 1) * WTR, gr, xr4);
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int r = 0; r < WTR; ++r) {
                        acc2[j].lo = fma2(g[r].lo, xv[r + j].lo, acc2[j].lo);
                        acc2[j].hi = fma2(g[r].hi, xv[r + j].hi, acc2[j].hi);
                    }
            }
        }
    }
    float acc[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        acc[j][0] = acc2[j].lo[0]; acc[j][1] = acc2[j].lo[1]; acc[j][2] = acc2[j].hi[0]; acc[j][3] = acc2[j].hi[1];
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[tt][j][cq * 4 + q] = acc[j][q];
    __syncthreads();
    // 16 strips -> 1, fixed order
    for (int o = tid; o < K * CB; o += 256) {
        const int j = o / CB, cc = o - j * CB;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][j][cc];
        const int cg = by * CB + cc;
        if (cg < p.C) p.part[((size_t)bx * K * K + (i * K + j)) * p.C + cg] = s;
    }
}

// dw (C,1,K,K) = sum over slabs of part[slab][tap][c]
__global__ void dw_slab_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, int nslabs, int taps,
                                      int C, int accumulate)
{
    const int total = taps * C;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int t = o / C, c = o - t * C;
        float s = 0.f;
        for (int k = 0; k < nslabs; ++k) s += part[((size_t)k * taps + t) * C + c];
        float *d = dw + (size_t)c * taps + t;
        *d = accumulate ? *d + s : s;
    }
}

__global__ void pack_dw_weight_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int taps, int flip)
{
    const int total = C * taps;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int t = o / C, c = o - t * C;
        dst[o] = src[(size_t)c * taps + (flip ? taps - 1 - t : t)];
    }
}

int check_desc(const kd_dw_desc *d, const char *who)
{
    KD_REQUIRE(d, KD_ERR_INVALID, "%s: null descriptor", who);
    KD_REQUIRE(d->dtype == KD_F32 || d->dtype == KD_BF16, KD_ERR_INVALID, "%s: bad dtype", who);
    KD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->dil >= 1 && d->pad >= 0, KD_ERR_INVALID,
               "%s: bad shape", who);
    KD_REQUIRE(d->k == 3 || d->k == 9, KD_ERR_UNSUPPORTED, "%s: kernel size %d not supported (3, 9)", who, d->k);
    KD_REQUIRE(2 * d->pad == d->dil * (d->k - 1), KD_ERR_UNSUPPORTED,
               "%s: only 'same' geometry (2*pad == dil*(k-1)) is supported, got pad=%d dil=%d k=%d", who, d->pad,
               d->dil, d->k);
    KD_REQUIRE(d->C % 4 == 0, KD_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 4", who, d->C);
    const int es = kd_elem_size(d->dtype);
    KD_REQUIRE(d->ldx >= d->C && (d->ldx * es) % (4 * es) == 0, KD_ERR_INVALID, "%s: bad ldx", who);
    return KD_OK;
}

}  // namespace

// dwconv_mfma.hip: matrix-core path for bf16 / 9x9 (1 = launched, 0 = not eligible, < 0 = error)
int kd_internal_dw_mfma_fwd(const kd_dw_desc *d, const void *x, const float *w_taps, const float *bias,
                            const kd_dw_epilogue *ep, void *y, hipStream_t s);

long long kd_internal_lattice_rows(int N, int H, int W, int dil);
int kd_internal_dw_lattice_ok(const kd_dw_desc *d, int nb);
int kd_internal_dw_mfma_fwd_n(const kd_dw_desc *d, int nb, int fan, const void *const *xs, const float *const *ws, void *const *ys,
                              const float *bias, const kd_dw_epilogue *ep, hipStream_t s, int lp = 0);
int kd_internal_dw_mfma_wgrad_slabs(const kd_dw_desc *d);
int kd_internal_dw_mfma_wgrad(const kd_dw_desc *d, const void *x, const void *dy, int ld_dy, float *part, hipStream_t s);
int kd_internal_dw_mfma_wgrad_multi_slabs(const kd_dw_desc *d, int n);
int kd_internal_dw_mfma_wgrad_multi(const kd_dw_desc *d, int n, const void *x, const void *const *dys, int ld_dy, float *part, hipStream_t s, int lp = 0);

extern "C" int kd_pack_dw_weight(const float *src, float *dst, int32_t C, int32_t k, int32_t flip, kd_stream_t stream)
{
    KD_REQUIRE(src && dst && C > 0 && k > 0, KD_ERR_INVALID, "kd_pack_dw_weight: bad argument");
    const int total = C * k * k;
    hipLaunchKernelGGL(pack_dw_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       k * k, flip);
    KD_CHECK_LAUNCH("kd_pack_dw_weight");
    return KD_OK;
}

extern "C" int kd_dwconv_fwd(const kd_dw_desc *d, const void *x, const float *w_taps, const float *bias,
                             const kd_dw_epilogue *ep, void *y, kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_fwd");
    if (rc) return rc;
    KD_REQUIRE(x && w_taps && y, KD_ERR_INVALID, "kd_dwconv_fwd: null argument");
    const int es = kd_elem_size(d->dtype);
    KD_REQUIRE(d->ldy >= d->C && d->ldy % 4 == 0 && d->ldx % 4 == 0, KD_ERR_INVALID, "kd_dwconv_fwd: ld must be a multiple of 4");
    KD_REQUIRE(((uintptr_t)x % (4 * es)) == 0 && ((uintptr_t)y % (4 * es)) == 0, KD_ERR_INVALID,
               "kd_dwconv_fwd: x/y must be aligned to 4 elements");
    {
        const int took = kd_internal_dw_mfma_fwd(d, x, w_taps, bias, ep, y, (hipStream_t)stream);
        if (took < 0) return took;
        if (took) return KD_OK;
    }
    DwParams p;
    memset(&p.ep, 0, sizeof(p.ep));
    if (ep) {
        p.ep = *ep;
        auto okp = [&](const void *q, int ld) { return !q || (((uintptr_t)q % (4 * es)) == 0 && ld >= d->C && ld % 4 == 0); };
        KD_REQUIRE(okp(ep->res_pre, ep->ld_res_pre) && okp(ep->mask, ep->ld_mask) && okp(ep->res_post, ep->ld_res_post),
                   KD_ERR_INVALID, "kd_dwconv_fwd: epilogue operands must be aligned to 4 elements");
    }
    p.x = x; p.w = w_taps; p.bias = bias; p.y = y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.pad = d->pad; p.dil = d->dil; p.ldx = d->ldx; p.ldy = d->ldy;
    p.LH = (d->H + d->dil - 1) / d->dil;
    p.LW = (d->W + d->dil - 1) / d->dil;
    // thread tile (lattice rows x cols per thread): 2x8 measured fastest at the student's shapes
    int ts = 2, tr = 8;
    if (const char *e = getenv("KDCC_DW_TILE")) sscanf(e, "%dx%d", &ts, &tr);   // tuning hook (2x8 | 2x4 | 4x4)
    p.tiles_h = (p.LH + ts - 1) / ts;
    p.tiles_w = (p.LW + tr - 1) / tr;
    p.gx = (p.tiles_h * p.tiles_w + 15) / 16;
    p.gz = d->N * d->dil * d->dil;
    const dim3 grid((unsigned)(p.gx * p.gz * ((d->C + CB - 1) / CB)));
    hipStream_t s = (hipStream_t)stream;
#define KD_DW_LAUNCH(TT, KK, A, B) hipLaunchKernelGGL((dwconv_fwd_kernel<TT, KK, A, B>), grid, dim3(256), 0, s, p)
#define KD_DW_TILES(TT, KK)                                   \
    do {                                                      \
        if (ts == 2 && tr == 4) KD_DW_LAUNCH(TT, KK, 2, 4);   \
        else if (ts == 4 && tr == 4) KD_DW_LAUNCH(TT, KK, 4, 4); \
        else KD_DW_LAUNCH(TT, KK, 2, 8);                      \
    } while (0)
    KD_NOTE_KERNEL(d->dtype == KD_BF16 ? "dwconv_fwd_kernel<bf16>" : "dwconv_fwd_kernel<f32>");
    if (d->dtype == KD_BF16) {
        if (d->k == 9) KD_DW_TILES(bf16_t, 9);
        else KD_DW_TILES(bf16_t, 3);
    } else {
        if (d->k == 9) KD_DW_TILES(float, 9);
        else KD_DW_TILES(float, 3);
    }
    KD_CHECK_LAUNCH("kd_dwconv_fwd");
    return KD_OK;
}

extern "C" int kd_dwconv_fwd_sum(const kd_dw_desc *d, int32_t n, const void *const *xs, const float *const *w_taps, void *y,
                                 kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_fwd_sum");
    if (rc) return rc;
    KD_REQUIRE(n >= 1 && xs && w_taps && y, KD_ERR_INVALID, "kd_dwconv_fwd_sum: null argument or n < 1");
    for (int i = 0; i < n; ++i) {
        KD_REQUIRE(xs[i] && w_taps[i], KD_ERR_INVALID, "kd_dwconv_fwd_sum: null input / tap table %d", i);
        KD_REQUIRE(xs[i] != y, KD_ERR_INVALID, "kd_dwconv_fwd_sum: y must not alias an input");
    }
    // up to three inputs per launch on the matrix cores; longer lists and every other shape chain through res_post
    int done = 0;
    if (n <= 3) {
        const int took = kd_internal_dw_mfma_fwd_n(d, n, 0, xs, w_taps, &y, nullptr, nullptr, (hipStream_t)stream);
        if (took < 0) return took;
        if (took) return KD_OK;
    }
    for (; done < n; ++done) {
        kd_dw_epilogue ep;
        memset(&ep, 0, sizeof(ep));
        ep.res_post = y;
        ep.ld_res_post = d->ldy;
        rc = kd_dwconv_fwd(d, xs[done], w_taps[done], nullptr, done ? &ep : nullptr, y, stream);
        if (rc) return rc;
    }
    return KD_OK;
}

extern "C" int kd_dwconv_fwd_fanout(const kd_dw_desc *d, int32_t n, const void *x, const float *const *w_taps, void *const *ys,
                                    kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_fwd_fanout");
    if (rc) return rc;
    KD_REQUIRE(n >= 1 && x && w_taps && ys, KD_ERR_INVALID, "kd_dwconv_fwd_fanout: null argument or n < 1");
    for (int i = 0; i < n; ++i) {
        KD_REQUIRE(ys[i] && w_taps[i], KD_ERR_INVALID, "kd_dwconv_fwd_fanout: null output / tap table %d", i);
        KD_REQUIRE(ys[i] != x, KD_ERR_INVALID, "kd_dwconv_fwd_fanout: an output must not alias the input");
        for (int j = 0; j < i; ++j) KD_REQUIRE(ys[i] != ys[j], KD_ERR_INVALID, "kd_dwconv_fwd_fanout: outputs %d and %d alias", j, i);
    }
    // up to three outputs per launch on the matrix cores (the tile of x staged once); everything else one launch per output
    for (int done = 0; done < n;) {
        const int m = n - done < 3 ? n - done : 3;
        const int took = kd_internal_dw_mfma_fwd_n(d, m, 1, &x, w_taps + done, ys + done, nullptr, nullptr, (hipStream_t)stream);
        if (took < 0) return took;
        if (took) { done += m; continue; }
        for (int i = 0; i < m; ++i) {
            rc = kd_dwconv_fwd(d, x, w_taps[done + i], nullptr, nullptr, ys[done + i], stream);
            if (rc) return rc;
        }
        done += m;
    }
    return KD_OK;
}

static int wgrad_slabs(const kd_dw_desc *d)
{
    const int LH = (d->H + d->dil - 1) / d->dil;
    const int nstrips = d->N * d->dil * d->dil * LH;
    return (nstrips + 15) / 16;
}

extern "C" size_t kd_dwconv_wgrad_workspace(const kd_dw_desc *d)
{
    if (!d || d->dil < 1) return 0;
    const int slabs = wgrad_slabs(d), mslabs = kd_internal_dw_mfma_wgrad_slabs(d);
    return (size_t)(slabs > mslabs ? slabs : mslabs) * d->k * d->k * d->C * sizeof(float);
}

extern "C" int kd_dwconv_wgrad(const kd_dw_desc *d, const void *x, const void *dy, int32_t ld_dy, float *dw,
                               int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_wgrad");
    if (rc) return rc;
    KD_REQUIRE(x && dy && dw && workspace, KD_ERR_INVALID, "kd_dwconv_wgrad: null argument");
    const int es = kd_elem_size(d->dtype);
    KD_REQUIRE(ld_dy >= d->C && ld_dy % 4 == 0 && d->ldx % 4 == 0, KD_ERR_INVALID, "kd_dwconv_wgrad: ld must be a multiple of 4");
    KD_REQUIRE(((uintptr_t)x % (4 * es)) == 0 && ((uintptr_t)dy % (4 * es)) == 0, KD_ERR_INVALID,
               "kd_dwconv_wgrad: x/dy must be aligned to 4 elements");
    KD_REQUIRE(workspace_bytes >= kd_dwconv_wgrad_workspace(d), KD_ERR_WORKSPACE, "kd_dwconv_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    {
        const int took = kd_internal_dw_mfma_wgrad(d, x, dy, ld_dy, (float *)workspace, s);
        if (took < 0) return took;
        if (took) {
            const int total = d->k * d->k * d->C;
            hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const float *)workspace, dw,
                               kd_internal_dw_mfma_wgrad_slabs(d), d->k * d->k, d->C, accumulate);
            KD_CHECK_LAUNCH("kd_dwconv_wgrad(reduce)");
            return KD_OK;
        }
    }
    DwWgradParams p;
    p.x = x; p.dy = dy; p.part = (float *)workspace;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.pad = d->pad; p.dil = d->dil; p.ldx = d->ldx; p.ld_dy = ld_dy;
    p.LH = (d->H + d->dil - 1) / d->dil;
    p.LW = (d->W + d->dil - 1) / d->dil;
    p.nstrips = d->N * d->dil * d->dil * p.LH;
    const int slabs = wgrad_slabs(d);
    p.nslabs = slabs;
    const dim3 grid((unsigned)(slabs * ((d->C + CB - 1) / CB) * d->k));
    KD_NOTE_KERNEL(d->dtype == KD_BF16 ? "dwconv_wgrad_kernel<bf16>" : "dwconv_wgrad_kernel<f32>");
    if (d->dtype == KD_BF16) {
        if (d->k == 9) hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16_t, 9>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16_t, 3>), grid, dim3(256), 0, s, p);
    } else {
        if (d->k == 9) hipLaunchKernelGGL((dwconv_wgrad_kernel<float, 9>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<float, 3>), grid, dim3(256), 0, s, p);
    }
    KD_CHECK_LAUNCH("kd_dwconv_wgrad");
    const int total = d->k * d->k * d->C;
    hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const float *)workspace, dw,
                       slabs, d->k * d->k, d->C, accumulate);
    KD_CHECK_LAUNCH("kd_dwconv_wgrad(reduce)");
    return KD_OK;
}

// Weight gradients of n depthwise convs of one geometry that read ONE input (the replaced ASPP branches): fused on the matrix
// cores for n = 2, 3 where the shape allows (x staged once, its Hankel windows shared by the branches), else one launch each.
extern "C" size_t kd_dwconv_wgrad_multi_workspace(const kd_dw_desc *d, int32_t n)
{
    if (!d || d->dil < 1 || n < 1) return 0;
    const size_t single = kd_dwconv_wgrad_workspace(d);
    // branches are fused three (or two) at a time; the largest chunk sets the size (slabs per branch do not depend on the count)
    const int m = n < 3 ? n : 3;
    const size_t multi = m >= 2 ? (size_t)kd_internal_dw_mfma_wgrad_multi_slabs(d, m) * m * d->k * d->k * d->C * sizeof(float) : 0;
    return single > multi ? single : multi;
}

extern "C" int kd_dwconv_wgrad_multi(const kd_dw_desc *d, int32_t n, const void *x, const void *const *dys, int32_t ld_dy,
                                     float *const *dws, int32_t accumulate, void *workspace, size_t workspace_bytes,
                                     kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_wgrad_multi");
    if (rc) return rc;
    KD_REQUIRE(n >= 1 && x && dys && dws && workspace, KD_ERR_INVALID, "kd_dwconv_wgrad_multi: null argument or n < 1");
    for (int i = 0; i < n; ++i) KD_REQUIRE(dys[i] && dws[i], KD_ERR_INVALID, "kd_dwconv_wgrad_multi: null gradient %d", i);
    KD_REQUIRE(workspace_bytes >= kd_dwconv_wgrad_multi_workspace(d, n), KD_ERR_WORKSPACE, "kd_dwconv_wgrad_multi: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    for (int done = 0; done < n;) {
        const int m = n - done < 3 ? n - done : 3;
        const int slabs = kd_internal_dw_mfma_wgrad_multi_slabs(d, m);
        int took = 0;
        if (slabs > 0) {
            const int es = kd_elem_size(d->dtype);
            bool ok = ld_dy >= d->C && ld_dy % 4 == 0 && d->ldx % 4 == 0 && ((uintptr_t)x % (4 * es)) == 0;
            for (int i = 0; i < m; ++i) ok = ok && ((uintptr_t)dys[done + i] % (4 * es)) == 0;
            if (ok) took = kd_internal_dw_mfma_wgrad_multi(d, m, x, dys + done, ld_dy, (float *)workspace, s);
            if (took < 0) return took;
        }
        if (took) {
            const int total = d->k * d->k * d->C;
            for (int i = 0; i < m; ++i) {
                hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s,
                                   (const float *)workspace + (size_t)i * slabs * total, dws[done + i], slabs, d->k * d->k, d->C, accumulate);
                KD_CHECK_LAUNCH("kd_dwconv_wgrad_multi(reduce)");
            }
        } else {
            for (int i = 0; i < m; ++i) {
                rc = kd_dwconv_wgrad(d, x, dys[done + i], ld_dy, dws[done + i], accumulate, workspace, workspace_bytes, stream);
                if (rc) return rc;
            }
        }
        done += m;
    }
    return KD_OK;
}


// ---- lattice-planar intermediates (include/kdcc.h "lattice-planar layout"; kernels in dwconv_mfma.hip) ----------------------
namespace {
// rows of an NHWC tensor <-> rows in lattice order (dense [rows][C] with stride ld_rows): the 1x1 convs next to the depthwise
// kernels run over rows in lattice order, their other operand / result (the branch's 256-channel output, hint gradient) lives
// in image order.  A thread moves 16 B; gather zero-fills the rows of padded cells and of the plane's tail.
template <bool GATHER>   // GATHER: img -> rows; else rows -> img (the two never alias: checked by the caller's shapes, not assumed by the compiler)
__global__ void lattice_rows_kernel(uint4 *img, uint4 *rows, long long nrows, int N, int H, int W, int d,
                                    int Ly, int Lx, int c16, int ld_img16, int ld_rows16)
{
    const long long total = nrows * c16;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / c16;
        const int c = (int)(e - row * c16);
        long long r = row;
        const int lx = (int)(r % Lx); r /= Lx;
        const int ly = (int)(r % Ly); r /= Ly;
        const int cls = (int)(r % (d * d)); r /= d * d;
        const int yy = cls / d + d * ly, xx = cls % d + d * lx;
        const bool in = r < N && yy < H && xx < W;
        if (GATHER) rows[row * ld_rows16 + c] = in ? img[((r * H + yy) * W + xx) * ld_img16 + c] : make_uint4(0u, 0u, 0u, 0u);
        else if (in) img[((r * H + yy) * W + xx) * ld_img16 + c] = rows[row * ld_rows16 + c];
    }
}
}  // namespace

extern "C" int64_t kd_lattice_rows(int32_t N, int32_t H, int32_t W, int32_t dil)
{
    return (int64_t)kd_internal_lattice_rows(N, H, W, dil);
}

extern "C" int32_t kd_dwconv_lattice_ok(const kd_dw_desc *d, int32_t n)
{
    if (!d || check_desc(d, "kd_dwconv_lattice_ok")) return 0;
    return kd_internal_dw_lattice_ok(d, n);
}

extern "C" int kd_lattice_rows_move(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dil, void *img, int32_t ld_img,
                                    void *rows, int32_t ld_rows, int32_t to_rows, kd_stream_t stream)
{
    KD_REQUIRE(img && rows && N > 0 && H > 0 && W > 0 && C > 0 && dil >= 1, KD_ERR_INVALID, "kd_lattice_rows_move: bad argument");
    KD_REQUIRE(dtype == KD_F32 || dtype == KD_BF16, KD_ERR_INVALID, "kd_lattice_rows_move: bad dtype");
    const int es = kd_elem_size(dtype), v = 16 / es;
    KD_REQUIRE(C % v == 0 && ld_img % v == 0 && ld_rows % v == 0 && ld_img >= C && ld_rows >= C && kd_aligned16(img) && kd_aligned16(rows),
               KD_ERR_INVALID, "kd_lattice_rows_move: channels / strides must be multiples of 16 B and the tensors 16-B aligned");
    const long long nrows = kd_internal_lattice_rows(N, H, W, dil);
    const int Ly = (H + dil - 1) / dil, Lx = (W + dil - 1) / dil;
    const long long total = nrows * (C / v);
    const unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    if (to_rows)
        hipLaunchKernelGGL(lattice_rows_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4 *)img, (uint4 *)rows, nrows, N, H, W,
                           dil, Ly, Lx, C / v, ld_img / v, ld_rows / v);
    else
        hipLaunchKernelGGL(lattice_rows_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4 *)img, (uint4 *)rows, nrows, N, H, W,
                           dil, Ly, Lx, C / v, ld_img / v, ld_rows / v);
    KD_CHECK_LAUNCH("kd_lattice_rows_move");
    return KD_OK;
}

extern "C" int kd_dwconv_fwd_fanout_lattice(const kd_dw_desc *d, int32_t n, const void *x, const float *const *w_taps, void *const *ys,
                                            kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_fwd_fanout_lattice");
    if (rc) return rc;
    KD_REQUIRE(n >= 2 && n <= 3 && x && w_taps && ys, KD_ERR_INVALID, "kd_dwconv_fwd_fanout_lattice: null argument or n not 2, 3");
    for (int i = 0; i < n; ++i) {
        KD_REQUIRE(ys[i] && w_taps[i] && ys[i] != x, KD_ERR_INVALID, "kd_dwconv_fwd_fanout_lattice: null / aliasing output %d", i);
        for (int j = 0; j < i; ++j) KD_REQUIRE(ys[i] != ys[j], KD_ERR_INVALID, "kd_dwconv_fwd_fanout_lattice: outputs %d and %d alias", j, i);
    }
    KD_REQUIRE(kd_internal_dw_lattice_ok(d, n), KD_ERR_UNSUPPORTED, "kd_dwconv_fwd_fanout_lattice: shape not eligible (kd_dwconv_lattice_ok)");
    kd_dw_desc dd = *d;
    dd.ldy = d->C;   // (unused by the lattice kernels; the eligibility test of the shared launcher reads it)
    const int took = kd_internal_dw_mfma_fwd_n(&dd, n, 1, &x, w_taps, ys, nullptr, nullptr, (hipStream_t)stream, 1);
    if (took < 0) return took;
    KD_REQUIRE(took, KD_ERR_UNSUPPORTED, "kd_dwconv_fwd_fanout_lattice: the matrix-core kernel refused the call");
    return KD_OK;
}

extern "C" int kd_dwconv_fwd_sum_lattice(const kd_dw_desc *d, int32_t n, const void *const *xs, const float *const *w_taps, void *y,
                                         kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_fwd_sum_lattice");
    if (rc) return rc;
    KD_REQUIRE(n >= 2 && n <= 3 && xs && w_taps && y, KD_ERR_INVALID, "kd_dwconv_fwd_sum_lattice: null argument or n not 2, 3");
    for (int i = 0; i < n; ++i) KD_REQUIRE(xs[i] && w_taps[i] && xs[i] != y, KD_ERR_INVALID, "kd_dwconv_fwd_sum_lattice: null / aliasing input %d", i);
    KD_REQUIRE(d->ldy >= d->C && d->ldy % 8 == 0, KD_ERR_INVALID, "kd_dwconv_fwd_sum_lattice: bad ldy");
    KD_REQUIRE(kd_internal_dw_lattice_ok(d, n), KD_ERR_UNSUPPORTED, "kd_dwconv_fwd_sum_lattice: shape not eligible (kd_dwconv_lattice_ok)");
    const int took = kd_internal_dw_mfma_fwd_n(d, n, 0, xs, w_taps, &y, nullptr, nullptr, (hipStream_t)stream, 1);
    if (took < 0) return took;
    KD_REQUIRE(took, KD_ERR_UNSUPPORTED, "kd_dwconv_fwd_sum_lattice: the matrix-core kernel refused the call");
    return KD_OK;
}

extern "C" int kd_dwconv_wgrad_multi_lattice(const kd_dw_desc *d, int32_t n, const void *x, const void *const *dys, float *const *dws,
                                             int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    int rc = check_desc(d, "kd_dwconv_wgrad_multi_lattice");
    if (rc) return rc;
    KD_REQUIRE(n >= 2 && n <= 3 && x && dys && dws && workspace, KD_ERR_INVALID, "kd_dwconv_wgrad_multi_lattice: null argument or n not 2, 3");
    for (int i = 0; i < n; ++i) KD_REQUIRE(dys[i] && dws[i], KD_ERR_INVALID, "kd_dwconv_wgrad_multi_lattice: null gradient %d", i);
    KD_REQUIRE(workspace_bytes >= kd_dwconv_wgrad_multi_workspace(d, n), KD_ERR_WORKSPACE, "kd_dwconv_wgrad_multi_lattice: workspace too small");
    KD_REQUIRE(kd_internal_dw_lattice_ok(d, n), KD_ERR_UNSUPPORTED, "kd_dwconv_wgrad_multi_lattice: shape not eligible (kd_dwconv_lattice_ok)");
    hipStream_t s = (hipStream_t)stream;
    const int slabs = kd_internal_dw_mfma_wgrad_multi_slabs(d, n);
    KD_REQUIRE(slabs > 0, KD_ERR_UNSUPPORTED, "kd_dwconv_wgrad_multi_lattice: shape not eligible");
    const int took = kd_internal_dw_mfma_wgrad_multi(d, n, x, dys, d->C, (float *)workspace, s, 1);
    if (took < 0) return took;
    KD_REQUIRE(took, KD_ERR_UNSUPPORTED, "kd_dwconv_wgrad_multi_lattice: the matrix-core kernel refused the call");
    const int total = d->k * d->k * d->C;
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const float *)workspace + (size_t)i * slabs * total,
                           dws[i], slabs, d->k * d->k, d->C, accumulate);
        KD_CHECK_LAUNCH("kd_dwconv_wgrad_multi_lattice(reduce)");
    }
    return KD_OK;
}

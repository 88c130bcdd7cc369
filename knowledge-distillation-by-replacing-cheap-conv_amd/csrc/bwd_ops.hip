// Backward plumbing of the student graph (everything autograd would run between the conv dgrad / wgrad kernels when
// gradients flow through the whole network: loss = kd + hint with every student parameter trainable, SURVEY 8(d) mode B,
// and hints taken after a BN+ReLU such as the `aspp` module output of cfg/cityscapes/51M_deeplab_incremental.json).
// All kernels are HBM-bound streams over NHWC tensors; every reduction has a fixed order (two stages, no float atomics),
// so results are bit-reproducible run to run.
#include "kd_common.h"

namespace {

inline int grid_for(long long total, int cap = 16384)
{
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool ok_dt(int d) { return d == KD_F32 || d == KD_BF16; }
inline bool vec_ok(const void *p, int ld, int es) { return !p || (kd_aligned16(p) && (ld * es) % 16 == 0); }

// ---- d relu(bn(x)) / dx on a gradient ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void relu_bn_bwd_kernel(const T *__restrict__ g, int ldg, const T *__restrict__ mask, int ldm,
                                                          const float *__restrict__ scale, const T *__restrict__ res, int ldres,
                                                          T *__restrict__ y, int ldy, long long M, int C8)
{
    const long long total = M * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / C8;
        const int c = (int)(i - m * C8) * 8;
        float v[8], k[8], s[8];
        ld8(g + m * ldg + c, v);
        ld8(mask + m * ldm + c, k);
        ld8(scale + c, s);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = k[q] > 0.f ? v[q] * s[q] : 0.f;
        if (res) {
            ld8(res + m * ldres + c, k);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += k[q];
        }
        st8(y + m * ldy + c, v);
    }
}

// ---- per-group channel sums ------------------------------------------------------------------------------------------------
// stage 1: block = 32 channel octets (256 channels, 512 B of a bf16 row: fully coalesced 16-B loads) x 8 row lanes; a block
// owns one chunk of rows of one group, every thread sums its rows (every 8th) in fp32, the 8 row lanes are combined in LDS in
// fixed order; stage 2: one thread per (group, channel) adds the chunk partials in order, in fp64.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void channel_sums_partial_kernel(const T *__restrict__ g, int ldg, const T *__restrict__ sub,
                                                                   int ldsub, const T *__restrict__ a, int lda, long long rows,
                                                                   int C, int chunks, float *__restrict__ part)
{
    // lanes along channels: 32 vectors (256 channels with VEC = 8), or fewer for narrow tensors so that the other lanes take
    // more rows instead of idling (C = 64 / 128 at full resolution are the longest calls of a mode-B step)
    __shared__ float red[2][256 * VEC];
    const int nvec = (C + VEC - 1) / VEC;
    const int lc = nvec >= 32 ? 32 : (nvec > 16 ? 32 : (nvec > 8 ? 16 : (nvec > 4 ? 8 : 4)));
    const int nrl = 256 / lc;
    const int cq = threadIdx.x % lc, rl = threadIdx.x / lc;
    const int c = (blockIdx.y * lc + cq) * VEC;
    const int chunk = blockIdx.x, grp = blockIdx.z;
    const long long per = (rows + chunks - 1) / chunks;
    const long long r0 = (long long)chunk * per, r1 = min(rows, r0 + per);
    const long long base = (long long)grp * rows;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) s1[q] = s2[q] = 0.f;
    if (c < C) {
        // four rows per trip: their (up to 12) loads are issued together, then accumulated in row order -- the sums are the
        // same fp32 chains as a row-at-a-time loop, but a thread keeps 8-12 loads in flight instead of 2-3 (1.7 -> HBM rate)
        auto load_row = [&](long long r, float (&v)[VEC], float (&t)[VEC]) __attribute__((always_inline)) {
            const long long m = base + r;
            if constexpr (VEC == 8) ld8(g + m * ldg + c, v);
            else v[0] = Elem<T>::ld(g + m * ldg + c);
            if (sub) {
                float u[VEC];
                if constexpr (VEC == 8) ld8(sub + m * ldsub + c, u);
                else u[0] = Elem<T>::ld(sub + m * ldsub + c);
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] -= u[q];
            }
            if (a) {
                if constexpr (VEC == 8) ld8(a + m * lda + c, t);
                else t[0] = Elem<T>::ld(a + m * lda + c);
            }
        };
        auto add_row = [&](const float (&v)[VEC], const float (&t)[VEC]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) s1[q] += v[q];
            if (a) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) s2[q] += v[q] * t[q];
            }
        };
        long long r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC], t0[VEC], t1[VEC], t2[VEC], t3[VEC];
            load_row(r, v0, t0); load_row(r + nrl, v1, t1); load_row(r + 2 * nrl, v2, t2); load_row(r + 3 * nrl, v3, t3);
            add_row(v0, t0); add_row(v1, t1); add_row(v2, t2); add_row(v3, t3);
        }
        for (; r < r1; r += nrl) {
            float v[VEC], t[VEC];
            load_row(r, v, t);
            add_row(v, t);
        }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) { red[0][(rl * lc + cq) * VEC + q] = s1[q]; red[1][(rl * lc + cq) * VEC + q] = s2[q]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * lc * VEC; e += 256) {
        const int which = e / (lc * VEC), col = e - which * (lc * VEC);
        const int cc = blockIdx.y * lc * VEC + col;
        if (cc >= C) continue;
        float t = 0.f;
        for (int k = 0; k < nrl; ++k) t += red[which][k * lc * VEC + col];   // fixed order
        part[(((size_t)grp * chunks + chunk) * 2 + which) * C + cc] = t;
    }
}

// block = 16 channels x 16 chunk lanes: lane l adds chunks l, l+16, ... in order (fp64), the sixteen lane sums are combined
// as a fixed binary tree -- bit-reproducible.  (64 channels x 4 lanes left a 256-channel tensor with four blocks walking
// 1024 partials each: 37 us per call, 44 calls per mode-B step.)
__global__ __launch_bounds__(256) void channel_sums_finish_kernel(const float *__restrict__ part, int groups, int chunks, int C,
                                                                  float *__restrict__ s1, float *__restrict__ s2)
{
    __shared__ double sh[2][16][16];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int cblocks = (C + 15) / 16;
    const int grp = blockIdx.x / cblocks, c = (blockIdx.x - grp * cblocks) * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int k = lane; k < chunks; k += 16) {
            const float *o = part + (((size_t)grp * chunks + k) * 2) * C;
            a += o[c];
            if (s2) b += o[C + c];
        }
    sh[0][lane][cl] = a;
    sh[1][lane][cl] = b;
    __syncthreads();
    for (int w = 8; w >= 1; w >>= 1) {
        if (lane < w) {
            sh[0][lane][cl] += sh[0][lane + w][cl];
            sh[1][lane][cl] += sh[1][lane + w][cl];
        }
        __syncthreads();
    }
    if (lane == 0 && c < C) {
        s1[(size_t)grp * C + c] = (float)sh[0][0][cl];
        if (s2) s2[(size_t)grp * C + c] = (float)sh[1][0][cl];
    }
}

// eval-mode BN parameter gradients from the two channel sums of the gradient w.r.t. the BN INPUT (post ReLU mask):
//   g_y = g_x / scale;  dbeta = sum g_y = s1 / scale;  dgamma = sum g_y * xhat = (s2 - beta * s1) / (scale * gamma)
// with s2 = sum g_x * relu(bn(x)) (masked elements carry g_x = 0).  A channel whose gamma (hence scale) is 0 gets 0.
__global__ void bn_eval_param_grads_kernel(const float *s1, const float *s2, const float *scale, const float *gamma,
                                           const float *beta, float *dgamma, float *dbeta, int C, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = scale[c], gm = gamma[c];
    const float db = sc != 0.f ? s1[c] / sc : 0.f;
    const float dg = (sc != 0.f && gm != 0.f) ? (s2[c] - beta[c] * s1[c]) / (sc * gm) : 0.f;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
}

// ---- max-pool 3x3 / stride 2 / pad 1 backward ------------------------------------------------------------------------------
// gather form: input pixel (h, w) belongs to at most 2 x 2 windows; it receives a window's gradient when it is that window's
// arg-max -- the FIRST maximum in (ky, kx) scan order, as nn.MaxPool2d's backward (a NaN counts as the maximum).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ gy, int ldgy,
                                                          T *__restrict__ gx, int ldgx, int N, int H, int W, int C, int Ho, int Wo)
{
    const int CV = C / VEC;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * CV) return;
    const int w = i / CV, c = (i - w * CV) * VEC;
    const int h = blockIdx.y, n = blockIdx.z;
    const T *xn = x + (size_t)n * H * W * ldx;
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
    float mine[VEC];
    if constexpr (VEC == 8) ld8(xn + ((size_t)h * W + w) * ldx + c, mine);
    else mine[0] = Elem<T>::ld(xn + ((size_t)h * W + w) * ldx + c);
    // windows (ho, wo) with 2*ho - 1 <= h <= 2*ho + 1
    const int ho_lo = max(0, h / 2), ho_hi = min(Ho - 1, (h + 1) / 2);
    const int wo_lo = max(0, w / 2), wo_hi = min(Wo - 1, (w + 1) / 2);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            // is (h, w) the first maximum of window (ho, wo)?  An element earlier in scan order wins on >=, a later one on >.
            bool win[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) win[q] = true;
            for (int ky = 0; ky < 3; ++ky) {
                const int hh = 2 * ho - 1 + ky;
                if (hh < 0 || hh >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int ww = 2 * wo - 1 + kx;
                    if (ww < 0 || ww >= W || (hh == h && ww == w)) continue;
                    const bool earlier = hh < h || (hh == h && ww < w);
                    float o[VEC];
                    if constexpr (VEC == 8) ld8(xn + ((size_t)hh * W + ww) * ldx + c, o);
                    else o[0] = Elem<T>::ld(xn + ((size_t)hh * W + ww) * ldx + c);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const bool onan = o[q] != o[q], mnan = mine[q] != mine[q];
                        const bool beats = earlier ? (onan || (!mnan && o[q] >= mine[q])) : (!mnan && (onan || o[q] > mine[q]));
                        if (beats) win[q] = false;
                    }
                }
            }
            float gv[VEC];
            const T *gp = gy + (((size_t)n * Ho + ho) * Wo + wo) * ldgy + c;
            if constexpr (VEC == 8) ld8(gp, gv);
            else gv[0] = Elem<T>::ld(gp);
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                if (win[q]) acc[q] += gv[q];
        }
    T *op = gx + (((size_t)n * H + h) * W + w) * ldgx + c;
    if constexpr (VEC == 8) st8(op, acc);
    else Elem<T>::st(op, acc[0]);
}

// fast path (C % 8 == 0): (1) per window, the position 0..8 of its first maximum per channel -> one byte per (window, channel);
// (2) per input pixel, the <= 4 windows covering it: add the window's gradient where the stored position is this pixel's.
// ~2.6 GB of traffic for the 64-channel 1024x2048 pool at 4 images instead of 32 dependent 16-B loads per pixel.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(const T *__restrict__ x, int ldx, uint2 *__restrict__ idx, int N, int H, int W,
                                                             int C8, int Ho, int Wo)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Wo * C8) return;
    const int wo = i / C8, cq = i - wo * C8;
    const int ho = blockIdx.y, n = blockIdx.z;
    float m[8];
    uint32_t pos[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { m[q] = -INFINITY; pos[q] = 0xffu; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hh = 2 * ho - 1 + ky;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ww = 2 * wo - 1 + kx;
            if (ww < 0 || ww >= W) continue;
            float v[8];
            ld8(x + (((size_t)n * H + hh) * W + ww) * ldx + cq * 8, v);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (v[q] > m[q] || v[q] != v[q] || pos[q] == 0xffu) { m[q] = v[q]; pos[q] = ky * 3 + kx; }   // nn.MaxPool2d's rule
        }
    }
    idx[((size_t)n * Ho + ho) * Wo * C8 + (size_t)wo * C8 + cq] =
        make_uint2(pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24), pos[4] | (pos[5] << 8) | (pos[6] << 16) | (pos[7] << 24));
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_gather_kernel(const uint2 *__restrict__ idx, const T *__restrict__ gy, int ldgy,
                                                                 T *__restrict__ gx, int ldgx, int N, int H, int W, int C8, int Ho, int Wo)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * C8) return;
    const int w = i / C8, cq = i - w * C8;
    const int h = blockIdx.y, n = blockIdx.z;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    const int ho_lo = h / 2, ho_hi = min(Ho - 1, (h + 1) / 2);
    const int wo_lo = w / 2, wo_hi = min(Wo - 1, (w + 1) / 2);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            const uint32_t mine = (uint32_t)((h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1)));
            const size_t o = ((size_t)n * Ho + ho) * Wo + wo;
            const uint2 p = idx[o * C8 + cq];
            float g[8];
            ld8(gy + o * ldgy + cq * 8, g);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t pq = ((q < 4 ? p.x : p.y) >> (8 * (q & 3))) & 0xffu;
                if (pq == mine) acc[q] += g[q];
            }
        }
    st8(gx + (((size_t)n * H + h) * W + w) * ldgx + cq * 8, acc);
}

// ---- bilinear (align_corners=True) upsample backward, separable gather ---------------------------------------------------------
// forward: src = o * (I - 1) / (O - 1); i0 = floor(src), i1 = min(i0 + 1, I - 1), f = src - i0; y[o] = (1-f) x[i0] + f x[i1].
// backward along one axis: gx[i] = sum_o [i0(o) == i] (1 - f(o)) gy[o] + [i1(o) == i] f(o) gy[o]; the candidate range of o is
// bracketed generously and every candidate re-evaluates the forward's own (i0, i1, f), so the two sides cannot disagree.
__device__ __forceinline__ void up_src(int o, float sc, float off, int I, int &i0, int &i1, float &f)
{
    // align_corners=True: off = 0, sc = (I-1)/(O-1); align_corners=False: sc = I/O, off = 0.5 sc - 0.5, clamped at 0 (the
    // forward kernel's own expression, trunk_ops.hip upsample_kernel)
    const float src = fmaxf(o * sc + off, 0.f);
    i0 = min((int)src, I - 1);
    i1 = min(i0 + 1, I - 1);
    f = src - (float)i0;
}

// pass 1: along W.  gy (N, Ho, Wo, C) -> tmp (N, Ho, W, C) float.  VEC = 8: a thread owns 8 channels (16-B loads; the 256-channel
// decoder gradient took 1.33 ms one element per thread), VEC = 1: any C (the 19-class logits).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample_bwd_w_kernel(const T *__restrict__ gy, int ldgy, float *__restrict__ tmp, int N,
                                                             int Ho, int Wo, int W, int C, float sw, float ow)
{
    const int CV = C / VEC;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * CV) return;
    const int w = i / CV, c = (i - w * CV) * VEC;
    const int ho = blockIdx.y, n = blockIdx.z;
    const float inv = sw > 0.f ? 1.f / sw : 0.f;
    int lo = sw > 0.f ? (int)((w - 1 - ow) * inv) - 1 : 0, hi = sw > 0.f ? (int)((w + 1 - ow) * inv) + 2 : Wo - 1;
    lo = max(lo, 0); hi = min(hi, Wo - 1);
    const T *row = gy + (((size_t)n * Ho + ho) * Wo) * ldgy + c;
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
    for (int o = lo; o <= hi; ++o) {
        int i0, i1; float f;
        up_src(o, sw, ow, W, i0, i1, f);
        float wgt = 0.f;
        if (i0 == w) wgt += 1.f - f;
        if (i1 == w) wgt += f;
        if (wgt != 0.f) {
            if constexpr (VEC == 8) {
                float v[8];
                ld8(row + (size_t)o * ldgy, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += wgt * v[q];
            } else {
                acc[0] += wgt * Elem<T>::ld(row + (size_t)o * ldgy);
            }
        }
    }
    float *dst = tmp + (((size_t)n * Ho + ho) * W + w) * C + c;
    if constexpr (VEC == 8) st8(dst, acc);
    else dst[0] = acc[0];
}

// pass 2: along H.  tmp (N, Ho, W, C) float -> gx (N, H, W, C)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample_bwd_h_kernel(const float *__restrict__ tmp, T *__restrict__ gx, int ldgx, int N, int Ho,
                                                             int H, int W, int C, float sh, float oh)
{
    const int CV = C / VEC;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * CV) return;
    const int w = i / CV, c = (i - w * CV) * VEC;
    const int h = blockIdx.y, n = blockIdx.z;
    const float inv = sh > 0.f ? 1.f / sh : 0.f;
    int lo = sh > 0.f ? (int)((h - 1 - oh) * inv) - 1 : 0, hi = sh > 0.f ? (int)((h + 1 - oh) * inv) + 2 : Ho - 1;
    lo = max(lo, 0); hi = min(hi, Ho - 1);
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
    for (int o = lo; o <= hi; ++o) {
        int i0, i1; float f;
        up_src(o, sh, oh, H, i0, i1, f);
        float wgt = 0.f;
        if (i0 == h) wgt += 1.f - f;
        if (i1 == h) wgt += f;
        if (wgt != 0.f) {
            const float *src = tmp + (((size_t)n * Ho + o) * W + w) * C + c;
            if constexpr (VEC == 8) {
                float v[8];
                ld8(src, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += wgt * v[q];
            } else {
                acc[0] += wgt * src[0];
            }
        }
    }
    T *dst = gx + (((size_t)n * H + h) * W + w) * ldgx + c;
    if constexpr (VEC == 8) st8(dst, acc);
    else Elem<T>::st(dst, acc[0]);
}

// ---- zero insertion (the transposed view of a strided conv's output gradient) -----------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void zero_insert_kernel(const T *__restrict__ x, int ldx, T *__restrict__ y, int ldy, int H, int W,
                                                          int C8, int stride, int Hy, int Wy)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Wy * C8) return;
    const int w = i / C8, c = (i - w * C8) * 8;
    const int h = blockIdx.y, n = blockIdx.z;
    uint4 z[sizeof(T) / 2];
#pragma unroll
    for (unsigned q = 0; q < sizeof(T) / 2; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
    const bool src = h % stride == 0 && w % stride == 0 && h / stride < H && w / stride < W;
    T *op = y + (((size_t)n * Hy + h) * Wy + w) * ldy + c;
    if (src) {
        const T *ip = x + (((size_t)n * H + h / stride) * W + w / stride) * ldx + c;
#pragma unroll
        for (unsigned q = 0; q < sizeof(T) / 2; ++q) z[q] = ((const uint4 *)ip)[q];
    }
#pragma unroll
    for (unsigned q = 0; q < sizeof(T) / 2; ++q) ((uint4 *)op)[q] = z[q];
}

// ---- y[n, p, c] (+)= v[n, c] * alpha  (broadcast of a per-image vector over the pixels) --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void broadcast_add_kernel(const float *__restrict__ v, T *__restrict__ y, int ldy, int N,
                                                            long long HW, int C8, float alpha, int accumulate)
{
    const long long total = (long long)N * HW * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % C8);
        const long long pix = i / C8;
        const int n = (int)(pix / HW);
        float t[8], a[8];
        ld8(v + (size_t)n * C8 * 8 + cq * 8, t);
        T *op = y + (size_t)pix * ldy + cq * 8;
        if (accumulate) {
            ld8(op, a);
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = a[q] + alpha * t[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] *= alpha;
        }
        st8(op, t);
    }
}

// ---- stem conv weight gradient: dW[64][3][3][3] = sum_pixels dy[p][64] x x_nchw[p + tap][3] -------------------------------
// persistent blocks over (image row, 256-column segment) work items; the 3 x 258 x 3 input patch goes to LDS, thread =
// (output channel, pixel phase): 27 running sums each; block partials -> fixed-order reduce.
constexpr int SW_SEG = 256;

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float *__restrict__ x, const T *__restrict__ dy, int lddy, int N, int H,
                                                         int W, float *__restrict__ part)
{
    __shared__ float patch[3][3][SW_SEG + 2];   // [ci][ky][col]
    __shared__ float red[4][64 * 27];
    const int co = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int nseg = (W + SW_SEG - 1) / SW_SEG;
    const long long nitems = (long long)N * H * nseg;
    float acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int seg = (int)(it % nseg);
        const long long r = it / nseg;
        const int h = (int)(r % H), n = (int)(r / H);
        const int w0 = seg * SW_SEG, cols = min(SW_SEG, W - w0);
        __syncthreads();
        for (int e = threadIdx.x; e < 9 * (SW_SEG + 2); e += 256) {
            const int col = e % (SW_SEG + 2), rk = e / (SW_SEG + 2);
            const int ci = rk / 3, ky = rk - ci * 3;
            const int hh = h - 1 + ky, ww = w0 - 1 + col;
            patch[ci][ky][col] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? x[(((size_t)n * 3 + ci) * H + hh) * W + ww] : 0.f;
        }
        __syncthreads();
        const T *dyr = dy + (((size_t)n * H + h) * W + w0) * lddy + co;
        for (int c = ph; c < cols; c += 4) {
            const float g = Elem<T>::ld(dyr + (size_t)c * lddy);
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[(ci * 3 + ky) * 3 + kx] += g * patch[ci][ky][c + kx];
        }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) red[ph][co * 27 + t] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 27; e += 256)
        part[(size_t)blockIdx.x * 64 * 27 + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// The same reduction on the matrix cores (bf16 dy): the VALU kernel above streams 2.1 GB of dy at 8 images in 2.4 ms -- 27 FMAs
// and as many broadcast LDS reads per pixel and thread -- where 58 GFLOP are nothing for the fp32 MFMA.  Per work item (256
// pixels of one image row): dy goes to LDS as [pixel][64 ch] bf16 (coalesced 16-B loads), the 3 x 3 x 3 input patch as fp32
// rows like above; wave w owns output channels 16 w .. 16 w + 15 and two 16 x 16 accumulators over the 27 (+5 zero) columns
// (ci, ky, kx).  One v_mfma_f32_16x16x4_f32 step contracts FOUR pixels: A[i][k] = dy[p0 + k][16 w + i] (a 2-B LDS read,
// widened), B[k][j] = x[ci_j][h - 1 + ky_j][w0 + p0 + k - 1 + kx_j] (a 4-B LDS read at a per-lane offset): x stays fp32,
// products and sums are fp32 as in the kernel above.  Same partial layout and finish kernel.
typedef float stem_f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float *__restrict__ x, const bf16_t *__restrict__ dy, int lddy, int N,
                                                              int H, int W, float *__restrict__ part)
{
    constexpr int PROW = SW_SEG + 4;
    __shared__ __attribute__((aligned(16))) float patch[9][PROW];          // [(ci, ky)][col], zero outside the image
    __shared__ __attribute__((aligned(16))) bf16_t dys[SW_SEG][64];        // [pixel][channel], zero behind the row's end
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, k = lane >> 4;
    // this lane's two B columns: col = (ci * 3 + ky) * 3 + kx; columns 27..31 read a zeroed LDS word
    int boff[2];
    bool bok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = t * 16 + i;
        bok[t] = col < 27;
        const int rk = bok[t] ? col / 3 : 0, kx = bok[t] ? col - rk * 3 : 0;
        boff[t] = rk * PROW + k + kx;
    }
    const int nseg = (W + SW_SEG - 1) / SW_SEG;
    const long long nitems = (long long)N * H * nseg;
    stem_f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // an item's loads are all issued together and land in registers under the PREVIOUS item's MFMAs (one memory latency per
    // item exposed otherwise -- nine with a load in front of each LDS store: 1.9 ms)
    constexpr int NPF = (9 * PROW + 255) / 256;
    float pv[NPF];
    uint4 dv[8];
    auto fetch = [&](long long it) __attribute__((always_inline)) {
        const int seg = (int)(it % nseg);
        const long long r = it / nseg;
        const int h = (int)(r % H), n = (int)(r / H);
        const int w0 = seg * SW_SEG, cols = min(SW_SEG, W - w0);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = tid + q * 256;
            const int rk = e / PROW, col = e - rk * PROW;
            const int ci = rk / 3, ky = rk - ci * 3;
            const int hh = h - 1 + ky, ww = w0 - 1 + col;
            pv[q] = (e < 9 * PROW && hh >= 0 && hh < H && ww >= 0 && ww < W && col < cols + 2) ? x[(((size_t)n * 3 + ci) * H + hh) * W + ww] : 0.f;
        }
        // dy: 256 pixels x 128 B, eight 16-B pieces per thread (pixel = piece / 8)
        const bf16_t *src = dy + (((size_t)n * H + h) * W + w0) * lddy;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pc = tid + q * 256, px = pc >> 3, ch = (pc & 7) * 8;
            dv[q] = make_uint4(0u, 0u, 0u, 0u);
            if (px < cols) dv[q] = *(const uint4 *)(src + (size_t)px * lddy + ch);
        }
    };
    if ((long long)blockIdx.x < nitems) fetch(blockIdx.x);
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NPF; ++q)
            if (tid + q * 256 < 9 * PROW) (&patch[0][0])[tid + q * 256] = pv[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pc = tid + q * 256;
            *(uint4 *)&dys[pc >> 3][(pc & 7) * 8] = dv[q];
        }
        __syncthreads();
        if (it + gridDim.x < nitems) fetch(it + gridDim.x);
        const float *pf = &patch[0][0];
        const bf16_t *da = &dys[k][wv * 16 + i];
#pragma unroll 8
        for (int p0 = 0; p0 < SW_SEG; p0 += 4) {
            const float a = Elem<bf16_t>::ld(da + (size_t)p0 * 64);
            const float b0 = pf[boff[0] + p0], b1 = bok[1] ? pf[boff[1] + p0] : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
        }
    }
    // D[row = channel][col]: lane (j = lane & 15, q = lane >> 4) holds rows 4 q .. 4 q + 3 of column j
    float *out = part + (size_t)blockIdx.x * 64 * 27;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int co = wv * 16 + k * 4 + rr;
        out[co * 27 + i] = acc0[rr];
        if (16 + i < 27) out[co * 27 + 16 + i] = acc1[rr];
    }
}

__global__ void stem_wgrad_finish_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ dw, int accumulate)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 64 * 27) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * 64 * 27 + e];
    dw[e] = accumulate ? dw[e] + (float)s : (float)s;
}

}  // namespace

extern "C" int kd_relu_bn_bwd(int32_t dtype, const void *g, int32_t ldg, const void *mask, int32_t ldm, const float *scale,
                              const void *res, int32_t ldres, void *y, int32_t ldy, int64_t M, int32_t C, kd_stream_t stream)
{
    KD_REQUIRE(g && mask && scale && y && M > 0 && C > 0, KD_ERR_INVALID, "kd_relu_bn_bwd: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_relu_bn_bwd: bad dtype");
    const int es = kd_elem_size(dtype);
    KD_REQUIRE(C % 8 == 0 && vec_ok(g, ldg, es) && vec_ok(mask, ldm, es) && vec_ok(res, ldres, es) && vec_ok(y, ldy, es) &&
                   kd_aligned16(scale),
               KD_ERR_INVALID, "kd_relu_bn_bwd: C %% 8 and 16-B aligned views required");
    const int nb = grid_for((long long)M * (C / 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16)
        hipLaunchKernelGGL(relu_bn_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t *)g, ldg, (const bf16_t *)mask, ldm,
                           scale, (const bf16_t *)res, ldres, (bf16_t *)y, ldy, (long long)M, C / 8);
    else
        hipLaunchKernelGGL(relu_bn_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float *)g, ldg, (const float *)mask, ldm, scale,
                           (const float *)res, ldres, (float *)y, ldy, (long long)M, C / 8);
    KD_CHECK_LAUNCH("kd_relu_bn_bwd");
    return KD_OK;
}

static int cs_chunks(long long rows)
{
    long long c = (rows + 511) / 512;        // >= 512 rows (64 per thread) per block, at most 1024 chunks
    return (int)(c < 1 ? 1 : (c > 1024 ? 1024 : c));
}

extern "C" size_t kd_channel_sums_workspace(int32_t groups, int64_t rows_per_group, int32_t C)
{
    return (size_t)groups * cs_chunks(rows_per_group) * 2 * (size_t)C * sizeof(float);
}

extern "C" int kd_channel_sums(int32_t dtype, const void *g, int32_t ldg, const void *sub, int32_t ldsub, const void *a, int32_t lda,
                               int32_t groups, int64_t rows_per_group, int32_t C, float *s1, float *s2, void *workspace,
                               size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(g && s1 && workspace && groups > 0 && rows_per_group > 0 && C > 0, KD_ERR_INVALID, "kd_channel_sums: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_channel_sums: bad dtype");
    KD_REQUIRE(!a || s2, KD_ERR_INVALID, "kd_channel_sums: `a` given without s2");
    KD_REQUIRE(workspace_bytes >= kd_channel_sums_workspace(groups, rows_per_group, C), KD_ERR_WORKSPACE,
               "kd_channel_sums: workspace too small");
    const int es = kd_elem_size(dtype);
    const bool vec = C % 8 == 0 && vec_ok(g, ldg, es) && vec_ok(sub, ldsub, es) && vec_ok(a, lda, es);
    const int nvec = vec ? (C + 7) / 8 : C;
    const int lc = nvec > 16 ? 32 : (nvec > 8 ? 16 : (nvec > 4 ? 8 : 4));   // channel lanes of a block, as in the kernel
    // row chunks: ~2048 blocks over (groups, channel blocks, chunks) fill the chip; more only lengthens the second stage
    // (never more than the workspace was sized for)
    int chunks = cs_chunks(rows_per_group);
    {
        const long long other = (long long)groups * ((nvec + lc - 1) / lc);
        const long long want = (2048 + other - 1) / other;
        if (want < chunks) chunks = (int)(want < 16 ? 16 : want);
        if (chunks > cs_chunks(rows_per_group)) chunks = cs_chunks(rows_per_group);
    }
    const dim3 grid((unsigned)chunks, (unsigned)((nvec + lc - 1) / lc), (unsigned)groups);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) {
        if (vec) hipLaunchKernelGGL((channel_sums_partial_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t *)g, ldg, (const bf16_t *)sub, ldsub, (const bf16_t *)a, lda, (long long)rows_per_group, C, chunks, (float *)workspace);
        else hipLaunchKernelGGL((channel_sums_partial_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t *)g, ldg, (const bf16_t *)sub, ldsub, (const bf16_t *)a, lda, (long long)rows_per_group, C, chunks, (float *)workspace);
    } else {
        if (vec) hipLaunchKernelGGL((channel_sums_partial_kernel<float, 8>), grid, dim3(256), 0, s, (const float *)g, ldg, (const float *)sub, ldsub, (const float *)a, lda, (long long)rows_per_group, C, chunks, (float *)workspace);
        else hipLaunchKernelGGL((channel_sums_partial_kernel<float, 1>), grid, dim3(256), 0, s, (const float *)g, ldg, (const float *)sub, ldsub, (const float *)a, lda, (long long)rows_per_group, C, chunks, (float *)workspace);
    }
    KD_CHECK_LAUNCH("kd_channel_sums");
    hipLaunchKernelGGL(channel_sums_finish_kernel, dim3((unsigned)(groups * ((C + 15) / 16))), dim3(256), 0, s, (const float *)workspace,
                       groups, chunks, C, s1, a ? s2 : (float *)nullptr);
    KD_CHECK_LAUNCH("kd_channel_sums(finish)");
    return KD_OK;
}

namespace {
// rows [chunk * per, ...) of part[R][2][C] -> out[chunk][2][C] (fp64 accumulators, fixed order): first stage of
// kd_bn_sums_finish for the full-resolution layers (32768 partial rows and more)
__global__ __launch_bounds__(256) void bn_sums_stage_kernel(const float *__restrict__ part, int R, int C, int per, float *__restrict__ out)
{
    __shared__ double sh[2][16][16];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, chunk = blockIdx.y;
    const int r0 = chunk * per, r1 = min(R, r0 + per);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int k = r0 + lane; k < r1; k += 16) {
            const float *o = part + (size_t)k * 2 * C;
            a += o[c];
            b += o[C + c];
        }
    sh[0][lane][cl] = a;
    sh[1][lane][cl] = b;
    __syncthreads();
    for (int w = 8; w >= 1; w >>= 1) {
        if (lane < w) {
            sh[0][lane][cl] += sh[0][lane + w][cl];
            sh[1][lane][cl] += sh[1][lane + w][cl];
        }
        __syncthreads();
    }
    if (lane == 0 && c < C) {
        out[(size_t)chunk * 2 * C + c] = (float)sh[0][0][cl];
        out[(size_t)chunk * 2 * C + C + c] = (float)sh[1][0][cl];
    }
}
}  // namespace

static int bn_sums_chunk(int rows) { return rows > 16384 ? 256 : 64; }   // rows per first-stage block

extern "C" size_t kd_bn_sums_finish_workspace(int32_t rows, int32_t C)
{
    return rows > 256 ? (size_t)((rows + bn_sums_chunk(rows) - 1) / bn_sums_chunk(rows)) * 2 * (size_t)C * sizeof(float) : 0;
}

extern "C" int kd_bn_sums_finish(const float *part, int32_t rows, int32_t C, float *s1, float *s2, void *workspace, size_t workspace_bytes,
                                 kd_stream_t stream)
{
    KD_REQUIRE(part && s1 && s2 && rows > 0 && C > 0, KD_ERR_INVALID, "kd_bn_sums_finish: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (rows > 256) {
        // (C / 16 blocks walking 2048 rows each took 36 us per call, 41 calls per mode-B step: the rows are split first)
        KD_REQUIRE(workspace && workspace_bytes >= kd_bn_sums_finish_workspace(rows, C), KD_ERR_WORKSPACE, "kd_bn_sums_finish: workspace too small");
        const int per = bn_sums_chunk(rows), chunks = (rows + per - 1) / per;
        hipLaunchKernelGGL(bn_sums_stage_kernel, dim3((unsigned)((C + 15) / 16), (unsigned)chunks), dim3(256), 0, s, part, rows, C, per, (float *)workspace);
        KD_CHECK_LAUNCH("kd_bn_sums_finish(stage)");
        part = (const float *)workspace;
        rows = chunks;
    }
    hipLaunchKernelGGL(channel_sums_finish_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, part, 1, rows, C, s1, s2);
    KD_CHECK_LAUNCH("kd_bn_sums_finish");
    return KD_OK;
}

extern "C" int kd_bn_eval_param_grads(const float *s1, const float *s2, const float *scale, const float *gamma, const float *beta,
                                      float *dgamma, float *dbeta, int32_t C, int32_t accumulate, kd_stream_t stream)
{
    KD_REQUIRE(s1 && s2 && scale && gamma && beta && dgamma && dbeta && C > 0, KD_ERR_INVALID, "kd_bn_eval_param_grads: bad argument");
    hipLaunchKernelGGL(bn_eval_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, s1, s2, scale, gamma, beta,
                       dgamma, dbeta, C, accumulate);
    KD_CHECK_LAUNCH("kd_bn_eval_param_grads");
    return KD_OK;
}

extern "C" size_t kd_maxpool3x3s2_bwd_workspace(int32_t N, int32_t H, int32_t W, int32_t C)
{
    return (size_t)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (size_t)((C + 7) / 8) * 8;   // one byte per window and channel
}

extern "C" int kd_maxpool3x3s2_bwd(int32_t dtype, const void *x, int32_t ldx, const void *gy, int32_t ldgy, void *gx, int32_t ldgx,
                                   int32_t N, int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes,
                                   kd_stream_t stream)
{
    KD_REQUIRE(x && gy && gx && N > 0 && H > 0 && W > 0 && C > 0, KD_ERR_INVALID, "kd_maxpool3x3s2_bwd: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_maxpool3x3s2_bwd: bad dtype");
    const int es = kd_elem_size(dtype);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const bool vec = C % 8 == 0 && vec_ok(x, ldx, es) && vec_ok(gy, ldgy, es) && vec_ok(gx, ldgx, es);
    hipStream_t s = (hipStream_t)stream;
    if (vec && workspace && workspace_bytes >= kd_maxpool3x3s2_bwd_workspace(N, H, W, C) && ((uintptr_t)workspace & 7) == 0) {
        const int C8 = C / 8;
        const dim3 g1((unsigned)((Wo * C8 + 255) / 256), (unsigned)Ho, (unsigned)N), g2((unsigned)((W * C8 + 255) / 256), (unsigned)H, (unsigned)N);
        if (dtype == KD_BF16) {
            hipLaunchKernelGGL(maxpool_argmax_kernel<bf16_t>, g1, dim3(256), 0, s, (const bf16_t *)x, ldx, (uint2 *)workspace, N, H, W, C8, Ho, Wo);
            hipLaunchKernelGGL(maxpool_bwd_gather_kernel<bf16_t>, g2, dim3(256), 0, s, (const uint2 *)workspace, (const bf16_t *)gy, ldgy, (bf16_t *)gx, ldgx, N, H, W, C8, Ho, Wo);
        } else {
            hipLaunchKernelGGL(maxpool_argmax_kernel<float>, g1, dim3(256), 0, s, (const float *)x, ldx, (uint2 *)workspace, N, H, W, C8, Ho, Wo);
            hipLaunchKernelGGL(maxpool_bwd_gather_kernel<float>, g2, dim3(256), 0, s, (const uint2 *)workspace, (const float *)gy, ldgy, (float *)gx, ldgx, N, H, W, C8, Ho, Wo);
        }
        KD_CHECK_LAUNCH("kd_maxpool3x3s2_bwd");
        return KD_OK;
    }
    const dim3 g((unsigned)((W * (vec ? C / 8 : C) + 255) / 256), (unsigned)H, (unsigned)N);
    if (dtype == KD_BF16) {
        if (vec) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8>), g, dim3(256), 0, s, (const bf16_t *)x, ldx, (const bf16_t *)gy, ldgy, (bf16_t *)gx, ldgx, N, H, W, C, Ho, Wo);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 1>), g, dim3(256), 0, s, (const bf16_t *)x, ldx, (const bf16_t *)gy, ldgy, (bf16_t *)gx, ldgx, N, H, W, C, Ho, Wo);
    } else {
        if (vec) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 8>), g, dim3(256), 0, s, (const float *)x, ldx, (const float *)gy, ldgy, (float *)gx, ldgx, N, H, W, C, Ho, Wo);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<float, 1>), g, dim3(256), 0, s, (const float *)x, ldx, (const float *)gy, ldgy, (float *)gx, ldgx, N, H, W, C, Ho, Wo);
    }
    KD_CHECK_LAUNCH("kd_maxpool3x3s2_bwd");
    return KD_OK;
}

extern "C" size_t kd_upsample_bilinear_ac_bwd_workspace(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo)
{
    (void)H; (void)Wo;
    return (size_t)N * Ho * W * C * sizeof(float);
}

extern "C" int kd_upsample_bilinear_bwd(const void *gy, int32_t gy_dtype, int32_t ldgy, void *gx, int32_t gx_dtype, int32_t ldgx,
                                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t align_corners,
                                        void *workspace, size_t workspace_bytes, kd_stream_t stream);

extern "C" int kd_upsample_bilinear_ac_bwd(const void *gy, int32_t gy_dtype, int32_t ldgy, void *gx, int32_t gx_dtype, int32_t ldgx,
                                           int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *workspace,
                                           size_t workspace_bytes, kd_stream_t stream)
{
    return kd_upsample_bilinear_bwd(gy, gy_dtype, ldgy, gx, gx_dtype, ldgx, N, H, W, C, Ho, Wo, 1, workspace, workspace_bytes, stream);
}

extern "C" int kd_upsample_bilinear_bwd(const void *gy, int32_t gy_dtype, int32_t ldgy, void *gx, int32_t gx_dtype, int32_t ldgx,
                                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t align_corners,
                                        void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(gy && gx && workspace && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, KD_ERR_INVALID,
               "kd_upsample_bilinear_ac_bwd: bad argument");
    KD_REQUIRE(ok_dt(gy_dtype) && ok_dt(gx_dtype), KD_ERR_INVALID, "kd_upsample_bilinear_ac_bwd: bad dtype");
    KD_REQUIRE(workspace_bytes >= kd_upsample_bilinear_ac_bwd_workspace(N, H, W, C, Ho, Wo), KD_ERR_WORKSPACE,
               "kd_upsample_bilinear_ac_bwd: workspace too small");
    float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;   // the forward kernel's scales and offsets (kd_upsample_bilinear)
    float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    float oh = 0.f, ow = 0.f;
    if (!align_corners) {
        sh = (float)H / (float)Ho; sw = (float)W / (float)Wo;
        oh = 0.5f * sh - 0.5f; ow = 0.5f * sw - 0.5f;
    }
    hipStream_t s = (hipStream_t)stream;
    float *tmp = (float *)workspace;
    const bool vec = C % 8 == 0 && vec_ok(gy, ldgy, kd_elem_size(gy_dtype)) && vec_ok(gx, ldgx, kd_elem_size(gx_dtype)) && kd_aligned16(tmp);
    const int cv = vec ? C / 8 : C;
    const dim3 g1((unsigned)((W * cv + 255) / 256), (unsigned)Ho, (unsigned)N);
    if (vec) {
        if (gy_dtype == KD_BF16) hipLaunchKernelGGL((upsample_bwd_w_kernel<bf16_t, 8>), g1, dim3(256), 0, s, (const bf16_t *)gy, ldgy, tmp, N, Ho, Wo, W, C, sw, ow);
        else hipLaunchKernelGGL((upsample_bwd_w_kernel<float, 8>), g1, dim3(256), 0, s, (const float *)gy, ldgy, tmp, N, Ho, Wo, W, C, sw, ow);
    } else {
        if (gy_dtype == KD_BF16) hipLaunchKernelGGL((upsample_bwd_w_kernel<bf16_t, 1>), g1, dim3(256), 0, s, (const bf16_t *)gy, ldgy, tmp, N, Ho, Wo, W, C, sw, ow);
        else hipLaunchKernelGGL((upsample_bwd_w_kernel<float, 1>), g1, dim3(256), 0, s, (const float *)gy, ldgy, tmp, N, Ho, Wo, W, C, sw, ow);
    }
    KD_CHECK_LAUNCH("kd_upsample_bilinear_ac_bwd(w)");
    const dim3 g2((unsigned)((W * cv + 255) / 256), (unsigned)H, (unsigned)N);
    if (vec) {
        if (gx_dtype == KD_BF16) hipLaunchKernelGGL((upsample_bwd_h_kernel<bf16_t, 8>), g2, dim3(256), 0, s, (const float *)tmp, (bf16_t *)gx, ldgx, N, Ho, H, W, C, sh, oh);
        else hipLaunchKernelGGL((upsample_bwd_h_kernel<float, 8>), g2, dim3(256), 0, s, (const float *)tmp, (float *)gx, ldgx, N, Ho, H, W, C, sh, oh);
    } else {
        if (gx_dtype == KD_BF16) hipLaunchKernelGGL((upsample_bwd_h_kernel<bf16_t, 1>), g2, dim3(256), 0, s, (const float *)tmp, (bf16_t *)gx, ldgx, N, Ho, H, W, C, sh, oh);
        else hipLaunchKernelGGL((upsample_bwd_h_kernel<float, 1>), g2, dim3(256), 0, s, (const float *)tmp, (float *)gx, ldgx, N, Ho, H, W, C, sh, oh);
    }
    KD_CHECK_LAUNCH("kd_upsample_bilinear_ac_bwd(h)");
    return KD_OK;
}

extern "C" int kd_zero_insert(int32_t dtype, const void *x, int32_t ldx, void *y, int32_t ldy, int32_t N, int32_t H, int32_t W,
                              int32_t C, int32_t stride, int32_t Hy, int32_t Wy, kd_stream_t stream)
{
    KD_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && stride >= 1, KD_ERR_INVALID, "kd_zero_insert: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_zero_insert: bad dtype");
    KD_REQUIRE(Hy >= (H - 1) * stride + 1 && Wy >= (W - 1) * stride + 1, KD_ERR_INVALID, "kd_zero_insert: output too small");
    const int es = kd_elem_size(dtype);
    KD_REQUIRE(C % 8 == 0 && vec_ok(x, ldx, es) && vec_ok(y, ldy, es), KD_ERR_INVALID, "kd_zero_insert: C %% 8 and 16-B alignment required");
    const dim3 g((unsigned)((Wy * (C / 8) + 255) / 256), (unsigned)Hy, (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) hipLaunchKernelGGL(zero_insert_kernel<bf16_t>, g, dim3(256), 0, s, (const bf16_t *)x, ldx, (bf16_t *)y, ldy, H, W, C / 8, stride, Hy, Wy);
    else hipLaunchKernelGGL(zero_insert_kernel<float>, g, dim3(256), 0, s, (const float *)x, ldx, (float *)y, ldy, H, W, C / 8, stride, Hy, Wy);
    KD_CHECK_LAUNCH("kd_zero_insert");
    return KD_OK;
}

extern "C" int kd_broadcast_add(int32_t dtype, const float *v, void *y, int32_t ldy, int32_t N, int64_t HW, int32_t C, float alpha,
                                int32_t accumulate, kd_stream_t stream)
{
    KD_REQUIRE(v && y && N > 0 && HW > 0 && C > 0, KD_ERR_INVALID, "kd_broadcast_add: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_broadcast_add: bad dtype");
    KD_REQUIRE(C % 8 == 0 && kd_aligned16(v) && vec_ok(y, ldy, kd_elem_size(dtype)), KD_ERR_INVALID,
               "kd_broadcast_add: C %% 8 and 16-B alignment required");
    const int nb = grid_for((long long)N * HW * (C / 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) hipLaunchKernelGGL(broadcast_add_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, v, (bf16_t *)y, ldy, N, (long long)HW, C / 8, alpha, accumulate);
    else hipLaunchKernelGGL(broadcast_add_kernel<float>, dim3(nb), dim3(256), 0, s, v, (float *)y, ldy, N, (long long)HW, C / 8, alpha, accumulate);
    KD_CHECK_LAUNCH("kd_broadcast_add");
    return KD_OK;
}

static int stem_blocks(int N, int H, int W)
{
    const long long items = (long long)N * H * ((W + SW_SEG - 1) / SW_SEG);
    return (int)(items < 768 ? items : 768);   // three 256-thread blocks per CU (41 KiB of LDS each): one round
}

extern "C" size_t kd_stem_wgrad_workspace(int32_t N, int32_t H, int32_t W)
{
    return (size_t)stem_blocks(N, H, W) * 64 * 27 * sizeof(float);
}

extern "C" int kd_stem_wgrad(int32_t dtype, const float *x_nchw, const void *dy, int32_t ld_dy, float *dw, int32_t N, int32_t H,
                             int32_t W, int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(x_nchw && dy && dw && workspace && N > 0 && H > 0 && W > 0 && ld_dy >= 64, KD_ERR_INVALID, "kd_stem_wgrad: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_stem_wgrad: bad dtype");
    KD_REQUIRE(workspace_bytes >= kd_stem_wgrad_workspace(N, H, W), KD_ERR_WORKSPACE, "kd_stem_wgrad: workspace too small");
    const int nb = stem_blocks(N, H, W);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16 && ld_dy % 8 == 0 && kd_aligned16(dy)) {
        KD_NOTE_KERNEL("stem_wgrad_mfma_kernel");
        hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(nb), dim3(256), 0, s, x_nchw, (const bf16_t *)dy, ld_dy, N, H, W, (float *)workspace);
    } else if (dtype == KD_BF16) hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, x_nchw, (const bf16_t *)dy, ld_dy, N, H, W, (float *)workspace);
    else hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(nb), dim3(256), 0, s, x_nchw, (const float *)dy, ld_dy, N, H, W, (float *)workspace);
    KD_CHECK_LAUNCH("kd_stem_wgrad");
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3((64 * 27 + 255) / 256), dim3(256), 0, s, (const float *)workspace, nb, dw, accumulate);
    KD_CHECK_LAUNCH("kd_stem_wgrad(finish)");
    return KD_OK;
}

// Gated-SCNN shape stream (BASELINE config 5; reference models/gscnn/gscnn.py:183-325): the full-resolution pieces that are
// not convolutions the MFMA kernels can take -- 8/16/32-channel per-pixel algebra at 2 M pixels per image, all HBM-bound:
//   kd_gated_conv      GatedSpatialConv2d.forward (models/gscnn/gate_spatial_conv.py:50-60), one fused pass per pixel
//   kd_edge_attention  fuse (8 -> 1) -> sigmoid -> cat with the Canny map -> cw (2 -> 1) -> sigmoid   (gscnn.py:308-314)
//   kd_edge_aspp       edge branch of the ASPP module: bilinear resample + 1x1 (1 -> 256) + BN + ReLU    (gscnn.py:168-171)
//   kd_canny           the Canny edge map the reference computes on the host with cv2.Canny(uint8 image, 10, 100)
//                      (gscnn.py:284-288): Sobel 3x3, L1 magnitude, non-maximum suppression, hysteresis.  Parity of this
//                      one operator is UNPINNED (opencv-python is not vendored); it follows the published algorithm.
#include <stdlib.h>

#include "kd_common.h"

namespace {

inline bool ok_dt(int d) { return d == KD_F32 || d == KD_BF16; }
inline int blocks_for(long long total, int cap = 1 << 20)
{
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- gated spatial conv -----------------------------------------------------------------------------------------------------
// per pixel, with u = [feat(C); gate(1)]:
//   z = relu(W1 u + b1)        (C+1 hidden units; the leading eval-BN is folded into W1 / b1 by the caller)
//   alpha = sigmoid(w2 . z + b2)   (trailing BN(1) folded into w2 / b2)
//   out = Wg (feat * (alpha + 1))
// params (fp32): W1 [(C+1)][(C+1)], b1 [C+1], w2 [C+1], b2 [1], Wg [C][C], in this order.
// Weights live in LDS with rows padded to a multiple of four floats; a thread owns PIX = 4 consecutive pixels, so one
// broadcast ds_read_b128 (four weights) feeds 16 FMAs.  (One LDS read per FMA made the first version LDS-issue-bound at 4.4 ms
// for the 32-channel gate at 4 x 1024 x 2048 pixels; streaming the weights through scalar loads was slower still, 5.4 ms.)
template <typename T, int C>
__global__ __launch_bounds__(256) void gated_conv_kernel(const T *__restrict__ feat, int ldf, const T *__restrict__ gate, int ldg,
                                                         const float *__restrict__ prm, T *__restrict__ out, int ldo, long long npix)
{
    constexpr int H = C + 1, HP = (H + 3) & ~3, PIX = C >= 32 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) float sW1[H * HP];
    __shared__ __attribute__((aligned(16))) float sWg[C * C];
    __shared__ float sb1[H], sw2[H];
    __shared__ float sb2;
    for (int i = threadIdx.x; i < H * HP; i += 256) {
        const int j = i / HP, k = i - j * HP;
        sW1[i] = k < H ? prm[j * H + k] : 0.f;
    }
    for (int i = threadIdx.x; i < H; i += 256) { sb1[i] = prm[H * H + i]; sw2[i] = prm[H * H + H + i]; }
    if (threadIdx.x == 0) sb2 = prm[H * H + 2 * H];
    for (int i = threadIdx.x; i < C * C; i += 256) sWg[i] = prm[H * H + 2 * H + 1 + i];
    __syncthreads();
    const long long ngroups = (npix + PIX - 1) / PIX;
    for (long long gidx = (long long)blockIdx.x * 256 + threadIdx.x; gidx < ngroups; gidx += (long long)gridDim.x * 256) {
        const long long p0 = gidx * PIX;
        float u[PIX][HP];
#pragma unroll
        for (int px = 0; px < PIX; ++px) {
            const long long p = min(p0 + px, npix - 1);
#pragma unroll
            for (int q = 0; q < C / 8; ++q) {
                float v[8];
                ld8(feat + p * ldf + q * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) u[px][q * 8 + e] = v[e];
            }
            u[px][C] = Elem<T>::ld(gate + p * ldg);
#pragma unroll
            for (int k = H; k < HP; ++k) u[px][k] = 0.f;
        }
        float a[PIX];
#pragma unroll
        for (int px = 0; px < PIX; ++px) a[px] = sb2;
#pragma unroll 1
        for (int j = 0; j < H; ++j) {
            asm volatile("" ::: "memory");   // keep the (loop-invariant) LDS weight reads inside the loop: hoisted, they spill
            float z[PIX];
#pragma unroll
            for (int px = 0; px < PIX; ++px) z[px] = sb1[j];
#pragma unroll
            for (int k = 0; k < HP; k += 4) {
                const float4 w = *(const float4 *)&sW1[j * HP + k];
#pragma unroll
                for (int px = 0; px < PIX; ++px)
                    z[px] = fmaf(w.w, u[px][k + 3], fmaf(w.z, u[px][k + 2], fmaf(w.y, u[px][k + 1], fmaf(w.x, u[px][k], z[px]))));
            }
            const float wj = sw2[j];
#pragma unroll
            for (int px = 0; px < PIX; ++px) a[px] = fmaf(wj, fmaxf(z[px], 0.f), a[px]);
        }
#pragma unroll
        for (int px = 0; px < PIX; ++px) {
            const float k = 1.f / (1.f + __expf(-a[px])) + 1.f;
#pragma unroll
            for (int i = 0; i < C; ++i) u[px][i] *= k;
        }
#pragma unroll 1
        for (int q = 0; q < C / 8; ++q) {
            float v[PIX][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                asm volatile("" ::: "memory");
                float s[PIX];
#pragma unroll
                for (int px = 0; px < PIX; ++px) s[px] = 0.f;
#pragma unroll
                for (int i = 0; i < C; i += 4) {
                    const float4 w = *(const float4 *)&sWg[(q * 8 + e) * C + i];
#pragma unroll
                    for (int px = 0; px < PIX; ++px)
                        s[px] = fmaf(w.w, u[px][i + 3], fmaf(w.z, u[px][i + 2], fmaf(w.y, u[px][i + 1], fmaf(w.x, u[px][i], s[px]))));
                }
#pragma unroll
                for (int px = 0; px < PIX; ++px) v[px][e] = s[px];
            }
#pragma unroll
            for (int px = 0; px < PIX; ++px)
                if (p0 + px < npix) st8(out + (p0 + px) * ldo + q * 8, v[px]);
        }
    }
}

// ---- edge attention -----------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void edge_attention_kernel(const T *__restrict__ cs, int ldc, const float *__restrict__ canny,
                                                             const float *__restrict__ w /* fuse[8], cw[2] */, float *__restrict__ acts,
                                                             long long npix)
{
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float v[8];
        ld8(cs + p * ldc, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(w[e], v[e], s);
        const float edge = 1.f / (1.f + __expf(-s));
        const float a = w[8] * edge + w[9] * canny[p];
        acts[p] = 1.f / (1.f + __expf(-a));
    }
}

// ---- edge branch of the ASPP module -----------------------------------------------------------------------------------------
// y[n,ho,wo,c] = relu(bilinear_ac(acts)[n,ho,wo] * w[c] * scale[c] + shift[c]); acts (N,H,W) float, y a channel slice (ld)
template <typename T>
__global__ __launch_bounds__(256) void edge_aspp_kernel(const float *__restrict__ acts, int H, int W, const float *__restrict__ w,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        T *__restrict__ y, int ldy, int N, int Ho, int Wo, int C8, float sh, float sw)
{
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % C8);
        long long r = i / C8;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho), n = (int)(r / Ho);
        const float fh = ho * sh, fw = wo * sw;
        int h0 = (int)fh; h0 = h0 > H - 1 ? H - 1 : h0;
        int w0 = (int)fw; w0 = w0 > W - 1 ? W - 1 : w0;
        const int h1 = h0 + 1 < H ? h0 + 1 : H - 1, w1 = w0 + 1 < W ? w0 + 1 : W - 1;
        const float ah = fh - h0, aw = fw - w0;
        const float *b = acts + (size_t)n * H * W;
        const float e = (1.f - ah) * ((1.f - aw) * b[(size_t)h0 * W + w0] + aw * b[(size_t)h0 * W + w1]) +
                        ah * ((1.f - aw) * b[(size_t)h1 * W + w0] + aw * b[(size_t)h1 * W + w1]);
        float wv[8], sc[8], sf[8], o[8];
        ld8(w + cq * 8, wv); ld8(scale + cq * 8, sc); ld8(shift + cq * 8, sf);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = fmaxf(fmaf(e * wv[q], sc[q], sf[q]), 0.f);
        st8(y + (((size_t)n * Ho + ho) * Wo + wo) * ldy + cq * 8, o);
    }
}

// ---- Canny ----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int u8_of(float v)
{
    // numpy's float -> uint8 cast as the reference applies it to the normalised image (gscnn.py:284): truncation toward zero,
    // modulo 256 for out-of-range values
    return (int)((unsigned int)(int)v & 0xffu);
}
__device__ __forceinline__ int px(const float *img, int H, int W, int y, int x)
{
    y = y < 0 ? 0 : (y >= H ? H - 1 : y);   // BORDER_REPLICATE
    x = x < 0 ? 0 : (x >= W ? W - 1 : x);
    return u8_of(img[(size_t)y * W + x]);
}

// gradient (Sobel 3x3 on every colour channel, keep the channel with the largest L1 magnitude): dx, dy as shorts, mag as int
__global__ __launch_bounds__(256) void canny_grad_kernel(const float *__restrict__ x, int N, int H, int W, short2 *__restrict__ grad,
                                                         int *__restrict__ mag)
{
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const long long r = i / W;
        const int yy = (int)(r % H), n = (int)(r / H);
        int bx = 0, by = 0, bm = -1;
        for (int c = 0; c < 3; ++c) {
            const float *im = x + ((size_t)n * 3 + c) * H * W;
            const int a00 = px(im, H, W, yy - 1, xx - 1), a01 = px(im, H, W, yy - 1, xx), a02 = px(im, H, W, yy - 1, xx + 1);
            const int a10 = px(im, H, W, yy, xx - 1), a12 = px(im, H, W, yy, xx + 1);
            const int a20 = px(im, H, W, yy + 1, xx - 1), a21 = px(im, H, W, yy + 1, xx), a22 = px(im, H, W, yy + 1, xx + 1);
            const int gx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
            const int gy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
            const int m = abs(gx) + abs(gy);
            if (m > bm) { bm = m; bx = gx; by = gy; }
        }
        grad[i] = make_short2((short)bx, (short)by);
        mag[i] = bm;
    }
}

// non-maximum suppression + double threshold: state 0 = no edge, 1 = weak (candidate), 2 = strong
__global__ __launch_bounds__(256) void canny_nms_kernel(const short2 *__restrict__ grad, const int *__restrict__ mag, int N, int H, int W,
                                                        int low, int high, unsigned char *__restrict__ state)
{
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const long long r = i / W;
        const int yy = (int)(r % H);
        const int *mg = mag + (i - (long long)yy * W - xx);
        auto M = [&](int y, int x) { return (y < 0 || y >= H || x < 0 || x >= W) ? 0 : mg[(size_t)y * W + x]; };
        const int m = M(yy, xx);
        unsigned char st = 0;
        if (m > low) {
            const int xs = grad[i].x, ys = grad[i].y;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;                       // tan(22.5 deg) * 2^15
            bool keep;
            if (ay < tg22x) keep = m > M(yy, xx - 1) && m >= M(yy, xx + 1);
            else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) keep = m > M(yy - 1, xx) && m >= M(yy + 1, xx);
                else {
                    const int s = (xs ^ ys) < 0 ? -1 : 1;
                    keep = m > M(yy - 1, xx - s) && m > M(yy + 1, xx + s);
                }
            }
            if (keep) st = m > high ? 2 : 1;
        }
        state[i] = st;
    }
}

// one hysteresis sweep: a weak pixel with a strong 8-neighbour becomes strong; *changed counts promotions
__global__ __launch_bounds__(256) void canny_hyst_kernel(unsigned char *__restrict__ state, int N, int H, int W, int *__restrict__ changed)
{
    const long long total = (long long)N * H * W;
    int local = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (state[i] != 1) continue;
        const int xx = (int)(i % W);
        const long long r = i / W;
        const int yy = (int)(r % H);
        unsigned char *s = state + (i - (long long)yy * W - xx);
        bool strong = false;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int y = yy + dy, x = xx + dx;
                if ((dy || dx) && y >= 0 && y < H && x >= 0 && x < W && s[(size_t)y * W + x] == 2) strong = true;
            }
        if (strong) { state[i] = 2; local = 1; }   // racing promotions only speed the fixed point up; the result is unique
    }
    if (local) atomicAdd(changed, 1);
}

__global__ __launch_bounds__(256) void canny_finish_kernel(const unsigned char *__restrict__ state, float *__restrict__ out, long long total)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        out[i] = state[i] == 2 ? 255.f : 0.f;
}

// bf16 form on the matrix cores.  Per 16 pixels the two per-pixel matrix products become MFMAs that share ONE operand: with
// F = the 16 pixels' features as the B operand (lane (pixel, k-group) = 8 consecutive channels = one 16-B global load, no LDS),
//   Z = W1[:, :C] F  (+ W1[:, C] gate + b1 as a rank-1 update in registers),   alpha = sigmoid(w2 . relu(Z) + b2),
//   out = Wg (F (alpha + 1)) = (alpha + 1) (Wg F)        -- alpha is a per-pixel scalar, so the second product does not wait for it
// and both sets of weights sit in registers as A operands (a lane ends up with four consecutive hidden units / output channels
// of its pixel; the dot with w2 is finished across the four k-group lanes of a pixel by two lane swaps).  ~80 VALU
// instructions per 16 pixels instead of ~2100 FMAs per pixel: the kernel becomes HBM-bound like its neighbours.
typedef __attribute__((ext_vector_type(8))) short gc_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float gc_f32x4_t;
template <int C>
__global__ __launch_bounds__(256) void gated_conv_mfma_kernel(const bf16_t *__restrict__ feat, int ldf, const bf16_t *__restrict__ gate,
                                                              int ldg, const float *__restrict__ prm, bf16_t *__restrict__ out, int ldo,
                                                              long long npix)
{
    constexpr int H = C + 1;
    constexpr int NT1 = (H + 15) / 16;   // 16-row tiles of hidden units
    constexpr int NT2 = (C + 15) / 16;   // 16-row tiles of output channels
    const int lane = threadIdx.x & 63, fi = lane & 15, kg = lane >> 4;
    const float *W1 = prm, *b1p = prm + H * H, *w2p = b1p + H, *b2p = w2p + H, *Wgp = b2p + 1;
    // A operands: lane (row fi of tile t, k-group kg) holds input channels 8kg .. 8kg+7 (zero beyond C)
    uint4 a1[NT1], a2[NT2];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int h = t * 16 + fi, k = kg * 8 + q;
            v[q] = (h < H && k < C) ? W1[h * H + k] : 0.f;
        }
        a1[t] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = t * 16 + fi, k = kg * 8 + q;
            v[q] = (c < C && k < C) ? Wgp[c * C + k] : 0.f;
        }
        a2[t] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
    // per-lane constants of its D rows h = 16t + 4kg + r: gate column of W1, b1, w2 (zero for padded rows: relu(0) * 0)
    float wg1[NT1][4], bb1[NT1][4], ww2[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = t * 16 + kg * 4 + r;
            wg1[t][r] = h < H ? W1[h * H + C] : 0.f;
            bb1[t][r] = h < H ? b1p[h] : 0.f;
            ww2[t][r] = h < H ? w2p[h] : 0.f;
        }
    const float b2 = *b2p;
    const long long ngroups = (npix + 15) / 16;
    const long long wave_id = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
    for (long long g = wave_id; g < ngroups; g += nwaves) {
        const long long pix = g * 16 + fi;
        const bool ok = pix < npix;
        const long long pc = ok ? pix : npix - 1;
        uint4 bv = make_uint4(0u, 0u, 0u, 0u);
        if (kg * 8 < C) bv = *(const uint4 *)(feat + pc * ldf + kg * 8);
        const float gv = bf16_to_f32(gate[pc * ldg]);
        gc_f32x4_t z[NT1], o[NT2];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
            z[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gc_bf16x8_t, a1[t]), __builtin_bit_cast(gc_bf16x8_t, bv),
                                                           (gc_f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t)
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gc_bf16x8_t, a2[t]), __builtin_bit_cast(gc_bf16x8_t, bv),
                                                           (gc_f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(ww2[t][r], fmaxf(fmaf(wg1[t][r], gv, z[t][r] + bb1[t][r]), 0.f), dot);
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        const float k1 = 1.f / (1.f + __expf(-(dot + b2))) + 1.f;
        if (ok) {
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const int c = t * 16 + kg * 4;
                if (c < C)
                    *(uint2 *)(out + pix * ldo + c) = make_uint2(pack_bf16x2(o[t][0] * k1, o[t][1] * k1), pack_bf16x2(o[t][2] * k1, o[t][3] * k1));
            }
        }
    }
}

// 1x1 conv + bias on few channels (the squeezes d1 / d2 / d3 of the shape stream, gscnn.py:232-235: 64 -> 32, 32 -> 16, 16 -> 8 at full
// resolution), bf16: like the gated conv above, the pixels' channels are the B operand straight from global memory and the
// weights sit in registers as the A operand; a lane ends up with four consecutive output channels of its pixel.  Reads the
// CIN channels that exist instead of a tensor padded to the GEMM kernels' 64-channel granule.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void pointwise_small_kernel(const bf16_t *__restrict__ x, int ldx, const float *__restrict__ w,
                                                              const float *__restrict__ bias, bf16_t *__restrict__ y, int ldy, long long npix)
{
    constexpr int NKS = (CIN + 31) / 32, NT = (COUT + 15) / 16;
    const int lane = threadIdx.x & 63, fi = lane & 15, kg = lane >> 4;
    uint4 a[NT][NKS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = t * 16 + fi, k = ks * 32 + kg * 8 + q;
                v[q] = (c < COUT && k < CIN) ? w[c * CIN + k] : 0.f;
            }
            a[t][ks] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
    float bb[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = t * 16 + kg * 4 + r;
            bb[t][r] = (bias && c < COUT) ? bias[c] : 0.f;
        }
    const long long ngroups = (npix + 15) / 16;
    const long long wave_id = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
    for (long long g = wave_id; g < ngroups; g += nwaves) {
        const long long pix = g * 16 + fi;
        const bool ok = pix < npix;
        const long long pc = ok ? pix : npix - 1;
        uint4 bv[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bv[ks] = make_uint4(0u, 0u, 0u, 0u);
            if (ks * 32 + kg * 8 < CIN) bv[ks] = *(const uint4 *)(x + pc * ldx + ks * 32 + kg * 8);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gc_f32x4_t o = {bb[t][0], bb[t][1], bb[t][2], bb[t][3]};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gc_bf16x8_t, a[t][ks]), __builtin_bit_cast(gc_bf16x8_t, bv[ks]), o, 0, 0, 0);
            const int c = t * 16 + kg * 4;
            if (ok && c < COUT) *(uint2 *)(y + pix * ldy + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
    }
}

// ---- 3x3 conv on 16 / 32 channels (the BasicBlocks of the shape stream at full resolution) ------------------------------------
// res2 / res3 (encoders/Resnet.py:64-99 via gscnn.py:237-243) are 32- and 16-channel 3x3 convs on 2 M pixels per image: 0.16 / 0.04
// TFLOP per launch at 4 images against 0.5-1.1 GB of activations -- HBM-bound by a wide margin.  Zero-padded to the 64-channel
// granule of the implicit-GEMM kernels they cost 4x / 16x the MACs and the padded bytes (1.4 ms each).  Here a workgroup owns a
// 256-pixel row segment and walks down SC_ROWS output rows with a four-slot ring of input rows in LDS (every input row is
// fetched once per workgroup, fully coalesced); per 16 pixels one v_mfma_f32_16x16x32_bf16 per tap (C = 32) or per pair of
// taps (C = 16: K = 2 taps x 16 channels) with the weights as the A operand, so a lane ends up with four consecutive output
// channels of one pixel; bias (the folded BatchNorm shift), residual and ReLU in fp32 before the single rounding to bf16.
constexpr int SC_SEG = 256, SC_ROWS = 16;
typedef __attribute__((ext_vector_type(8))) short sc_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float sc_f32x4_t;
struct SmallConvParams {
    const bf16_t *x; const bf16_t *w; const float *bias; const bf16_t *res; bf16_t *y;
    int ldx, ldres, ldy, N, H, W, relu, nsx, nsy;
};
template <int C>
__global__ __launch_bounds__(C == 64 ? 512 : 256) void conv3x3_small_kernel(const SmallConvParams p)
{
    // C = 64 (res1): eight waves -- a wave owns two of the four 16-channel output tiles and a quarter of the segment's pixels,
    // so its weight fragments (9 taps x 2 k-steps x 2 tiles) fit in registers; C = 32 / 16: four waves, all output tiles each
    constexpr int NTHR = C == 64 ? 512 : 256;
    constexpr int PXB = C * 2;                      // bytes per pixel
    constexpr int PXL = C == 64 ? PXB + 16 : PXB;   // pixel stride in LDS.  C = 64: 144 B = 16 B x an odd number, so the 16 pixels
                                                    // of a fragment read land in 16 different 16-B bank groups (a 128-B stride puts them
                                                    // all in one: 1.00 -> 0.85 ms).  C = 32 / 16 stay dense: the padding costs them a
                                                    // workgroup per CU (66 -> 82 KiB), which is worth more than their 2- / 4-way conflicts
    constexpr int ROWB = (SC_SEG + 2) * PXL;        // one ring slot: the segment + one halo pixel each side
    constexpr int CPP = PXB / 16;                   // 16-B chunks per pixel
    constexpr int NCH = (SC_SEG + 2) * CPP;         // chunks per row
    constexpr int NLD = (NCH + NTHR - 1) / NTHR;
    constexpr int NT = C == 16 ? 5 : 9;             // taps (C = 16: tap pairs) per 16 pixels
    constexpr int NKS = C == 64 ? 2 : 1;            // 32-deep k-steps per tap
    constexpr int NCT = C == 16 ? 1 : 2;            // 16-channel output tiles per wave
    extern __shared__ __attribute__((aligned(16))) char ring[];   // 4 * ROWB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, kg = lane >> 4;
    const int ct0 = C == 64 ? (wave & 1) * 2 : 0;   // first output tile of this wave
    const int pq = C == 64 ? wave >> 1 : wave;      // pixel quarter of the segment
    int b = blockIdx.x;
    const int sx = b % p.nsx; b /= p.nsx;
    const int sy = b % p.nsy;
    const int n = b / p.nsy;
    const int x0 = sx * SC_SEG, r0 = sy * SC_ROWS, r1 = min(r0 + SC_ROWS, p.H);
    const bf16_t *xn = p.x + (size_t)n * p.H * p.W * p.ldx;

    // weight fragments (A operand): lane (output channel fi of tile ct, k-group kg)
    uint4 wf[NT][NKS][NCT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                int tap, cin;
                if (C == 16) { tap = 2 * t + (kg >> 1); cin = (kg & 1) * 8; }
                else { tap = t; cin = ks * 32 + kg * 8; }
                wf[t][ks][ct] = tap < 9 ? *(const uint4 *)(p.w + ((size_t)((ct0 + ct) * 16 + fi) * 9 + tap) * C + cin) : make_uint4(0u, 0u, 0u, 0u);
            }
    // per-lane LDS offsets of the pixel fragments (B operand) relative to (slot of the row above, first pixel of a group)
    int boff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int tap, cin;
        if (C == 16) { tap = min(2 * t + (kg >> 1), 8); cin = (kg & 1) * 8; }
        else { tap = t; cin = kg * 8; }
        boff[t] = ((tap / 3) << 16) | ((fi + tap % 3) * PXL + cin * 2);   // high half: ring row 0..2, low half: byte offset
    }
    float bias[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[ct][r] = p.bias ? p.bias[(ct0 + ct) * 16 + kg * 4 + r] : 0.f;
    float bias8[8];   // the 8 channels this lane owns after the k-group pair swap (two-tile instantiations)
#pragma unroll
    for (int q = 0; q < 8; ++q) bias8[q] = (NCT == 2 && p.bias) ? p.bias[(ct0 + (kg & 1)) * 16 + (kg & ~1) * 4 + q] : 0.f;

    uint4 st[NLD];
    auto fetch_row = [&](int hr) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NTHR;
            const int px = c / CPP, part = c - px * CPP, gx = x0 - 1 + px;
            const bool ok = c < NCH && hr >= 0 && hr < p.H && gx >= 0 && gx < p.W;
            st[i] = make_uint4(0u, 0u, 0u, 0u);
            if (ok) st[i] = *(const uint4 *)(xn + ((size_t)hr * p.W + gx) * p.ldx + part * 8);
        }
    };
    auto put_row = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NTHR;
            if (c < NCH) *(uint4 *)(ring + slot * ROWB + (c / CPP) * PXL + (c % CPP) * 16) = st[i];
        }
    };
    // slot of input row hr: (hr - r0 + 1) & 3
    fetch_row(r0 - 1); put_row(0);
    fetch_row(r0); put_row(1);
    fetch_row(r0 + 1); put_row(2);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const bool more = r + 1 < r1;
        if (more) fetch_row(r + 2);
        const int s0 = (r - r0) & 3;   // slot of row r - 1
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gx = pq * 64 + g * 16;   // first pixel of the group within the segment
            sc_f32x4_t acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = (sc_f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int slot = (s0 + (boff[t] >> 16)) & 3;
                const char *bp = ring + slot * ROWB + gx * PXL + (boff[t] & 0xffff);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const uint4 bv = *(const uint4 *)(bp + ks * 64);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sc_bf16x8_t, wf[t][ks][ct]),
                                                                          __builtin_bit_cast(sc_bf16x8_t, bv), acc[ct], 0, 0, 0);
                }
            }
            const int xx = x0 + gx + fi;
            if constexpr (NCT == 2) {
                // a lane holds channels 4kg..4kg+3 of BOTH tiles; the k-group pair (kg, kg ^ 1) swaps one tile's quad so that the
                // even lane owns 8 consecutive channels of the first tile and the odd lane 8 of the second: 16-B residual loads
                // and stores instead of two 8-B ones each (the epilogue was store-issue-bound: 8.2 us per 256-pixel row at C = 64)
                const bool odd = kg & 1;
                float lo[4], hi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float send = odd ? acc[0][q] : acc[1][q];
                    const float recv = __shfl_xor(send, 16);
                    lo[q] = odd ? recv : acc[0][q];
                    hi[q] = odd ? acc[1][q] : recv;
                }
                if (xx < p.W) {
                    const size_t pix = ((size_t)n * p.H + r) * p.W + xx;
                    const int ch = (ct0 + (odd ? 1 : 0)) * 16 + (kg & ~1) * 4;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = lo[q] + bias8[q]; v[4 + q] = hi[q] + bias8[4 + q]; }
                    if (p.res) {
                        float rv[8];
                        ld8(p.res + pix * p.ldres + ch, rv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rv[q];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                    }
                    st8(p.y + pix * p.ldy + ch, v);
                }
            } else if (xx < p.W) {
                const size_t pix = ((size_t)n * p.H + r) * p.W + xx;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int ch = (ct0 + ct) * 16 + kg * 4;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[ct][q] + bias[ct][q];
                    if (p.res) {
                        const uint2 rv = *(const uint2 *)(p.res + pix * p.ldres + ch);
                        v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                        v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    if (p.relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                    }
                    *(uint2 *)(p.y + pix * p.ldy + ch) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            }
        }
        if (more) put_row((r - r0 + 3) & 3);   // row r + 2 -> the slot row r - 2 left free
        __syncthreads();
    }
}

}  // namespace

extern "C" int kd_gated_conv(int32_t dtype, const void *feat, int32_t ldf, const void *gate, int32_t ldg, const float *params,
                             void *out, int32_t ldo, int64_t npix, int32_t C, kd_stream_t stream)
{
    KD_REQUIRE(feat && gate && params && out && npix > 0, KD_ERR_INVALID, "kd_gated_conv: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_gated_conv: bad dtype");
    KD_REQUIRE(C == 8 || C == 16 || C == 32, KD_ERR_UNSUPPORTED, "kd_gated_conv: C must be 8, 16 or 32 (got %d)", C);
    const int es = kd_elem_size(dtype);
    KD_REQUIRE(kd_aligned16(feat) && kd_aligned16(out) && (ldf * es) % 16 == 0 && (ldo * es) % 16 == 0, KD_ERR_INVALID,
               "kd_gated_conv: 16-B aligned feature views required");
    hipStream_t s = (hipStream_t)stream;
    {
        static int mfma = -1;
        if (mfma < 0) { const char *e = getenv("KDCC_GATED_MFMA"); mfma = !(e && e[0] == '0'); }   // A/B: 0 = the VALU kernel
        if (mfma && dtype == KD_BF16 && (ldo * es) % 8 == 0) {
            const long long waves = (npix + 15) / 16;
            const int nbm = blocks_for((waves + 3) / 4 * 256, 256 * 32);   // <= 32 workgroups per CU, grid-stride over the rest
#define KD_GCM(CC) hipLaunchKernelGGL((gated_conv_mfma_kernel<CC>), dim3(nbm), dim3(256), 0, s, (const bf16_t *)feat, ldf, (const bf16_t *)gate, ldg, params, (bf16_t *)out, ldo, (long long)npix)
            KD_NOTE_KERNEL("gated_conv_mfma_kernel");
            if (C == 8) KD_GCM(8); else if (C == 16) KD_GCM(16); else KD_GCM(32);
#undef KD_GCM
            KD_CHECK_LAUNCH("kd_gated_conv(mfma)");
            return KD_OK;
        }
    }
    const int nb = blocks_for((npix + (C >= 32 ? 1 : 3)) / (C >= 32 ? 2 : 4), 65536);
#define KD_GC(T, CC) hipLaunchKernelGGL((gated_conv_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T *)feat, ldf, (const T *)gate, ldg, params, (T *)out, ldo, (long long)npix)
    KD_NOTE_KERNEL("gated_conv_kernel");
    if (dtype == KD_BF16) { if (C == 8) KD_GC(bf16_t, 8); else if (C == 16) KD_GC(bf16_t, 16); else KD_GC(bf16_t, 32); }
    else { if (C == 8) KD_GC(float, 8); else if (C == 16) KD_GC(float, 16); else KD_GC(float, 32); }
#undef KD_GC
    KD_CHECK_LAUNCH("kd_gated_conv");
    return KD_OK;
}

extern "C" int kd_edge_attention(int32_t dtype, const void *cs, int32_t ldc, const float *canny, const float *weights, float *acts,
                                 int64_t npix, kd_stream_t stream)
{
    KD_REQUIRE(cs && canny && weights && acts && npix > 0, KD_ERR_INVALID, "kd_edge_attention: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_edge_attention: bad dtype");
    KD_REQUIRE(kd_aligned16(cs) && (ldc * kd_elem_size(dtype)) % 16 == 0, KD_ERR_INVALID, "kd_edge_attention: 16-B aligned view required");
    const int nb = blocks_for(npix, 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) hipLaunchKernelGGL(edge_attention_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t *)cs, ldc, canny, weights, acts, (long long)npix);
    else hipLaunchKernelGGL(edge_attention_kernel<float>, dim3(nb), dim3(256), 0, s, (const float *)cs, ldc, canny, weights, acts, (long long)npix);
    KD_CHECK_LAUNCH("kd_edge_attention");
    return KD_OK;
}

extern "C" int kd_edge_aspp(int32_t dtype, const float *acts, int32_t H, int32_t W, const float *w, const float *scale, const float *shift,
                            void *y, int32_t ldy, int32_t N, int32_t Ho, int32_t Wo, int32_t C, kd_stream_t stream)
{
    KD_REQUIRE(acts && w && scale && shift && y && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0, KD_ERR_INVALID, "kd_edge_aspp: bad argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_edge_aspp: bad dtype");
    KD_REQUIRE(C % 8 == 0 && kd_aligned16(y) && (ldy * kd_elem_size(dtype)) % 16 == 0 && kd_aligned16(w) && kd_aligned16(scale) && kd_aligned16(shift),
               KD_ERR_INVALID, "kd_edge_aspp: C %% 8 and 16-B alignment required");
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int nb = blocks_for((long long)N * Ho * Wo * (C / 8), 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) hipLaunchKernelGGL(edge_aspp_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, acts, H, W, w, scale, shift, (bf16_t *)y, ldy, N, Ho, Wo, C / 8, sh, sw);
    else hipLaunchKernelGGL(edge_aspp_kernel<float>, dim3(nb), dim3(256), 0, s, acts, H, W, w, scale, shift, (float *)y, ldy, N, Ho, Wo, C / 8, sh, sw);
    KD_CHECK_LAUNCH("kd_edge_aspp");
    return KD_OK;
}

extern "C" size_t kd_canny_workspace(int32_t N, int32_t H, int32_t W)
{
    const size_t n = (size_t)N * H * W;
    return n * (sizeof(short2) + sizeof(int) + 1) + 256;
}

// x: (N,3,H,W) float NCHW (the trainer's batch); out: (N,H,W) float, 0 / 255.  `changed` is a device int the caller reads
// back between sweeps: kd_canny runs the gradient / NMS stages and `sweeps` hysteresis sweeps, adding the number of blocks
// that promoted a pixel in the LAST sweep to *changed (zeroed first); call kd_canny_continue until it reads 0.
extern "C" int kd_canny(const float *x, int32_t N, int32_t H, int32_t W, int32_t low, int32_t high, int32_t sweeps, float *out,
                        int32_t *changed, void *workspace, size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(x && out && changed && workspace && N > 0 && H > 0 && W > 0 && sweeps >= 0, KD_ERR_INVALID, "kd_canny: bad argument");
    KD_REQUIRE(workspace_bytes >= kd_canny_workspace(N, H, W), KD_ERR_WORKSPACE, "kd_canny: workspace too small");
    const long long n = (long long)N * H * W;
    short2 *grad = (short2 *)workspace;
    int *mag = (int *)(grad + n);
    unsigned char *state = (unsigned char *)(mag + n);
    const int nb = blocks_for(n, 65536);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(canny_grad_kernel, dim3(nb), dim3(256), 0, s, x, N, H, W, grad, mag);
    hipLaunchKernelGGL(canny_nms_kernel, dim3(nb), dim3(256), 0, s, (const short2 *)grad, (const int *)mag, N, H, W, low, high, state);
    KD_CHECK_LAUNCH("kd_canny");
    for (int i = 0; i < sweeps; ++i) {
        if (hipMemsetAsync(changed, 0, sizeof(int), s) != hipSuccess) { kd_set_error("kd_canny: memset failed"); return KD_ERR_HIP; }
        hipLaunchKernelGGL(canny_hyst_kernel, dim3(nb), dim3(256), 0, s, state, N, H, W, changed);
    }
    hipLaunchKernelGGL(canny_finish_kernel, dim3(nb), dim3(256), 0, s, (const unsigned char *)state, out, n);
    KD_CHECK_LAUNCH("kd_canny(hysteresis)");
    return KD_OK;
}

extern "C" int kd_canny_continue(int32_t N, int32_t H, int32_t W, int32_t sweeps, float *out, int32_t *changed, void *workspace,
                                 size_t workspace_bytes, kd_stream_t stream)
{
    KD_REQUIRE(out && changed && workspace && N > 0 && H > 0 && W > 0 && sweeps > 0, KD_ERR_INVALID, "kd_canny_continue: bad argument");
    KD_REQUIRE(workspace_bytes >= kd_canny_workspace(N, H, W), KD_ERR_WORKSPACE, "kd_canny_continue: workspace too small");
    const long long n = (long long)N * H * W;
    unsigned char *state = (unsigned char *)workspace + n * (sizeof(short2) + sizeof(int));
    const int nb = blocks_for(n, 65536);
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < sweeps; ++i) {
        if (hipMemsetAsync(changed, 0, sizeof(int), s) != hipSuccess) { kd_set_error("kd_canny_continue: memset failed"); return KD_ERR_HIP; }
        hipLaunchKernelGGL(canny_hyst_kernel, dim3(nb), dim3(256), 0, s, state, N, H, W, changed);
    }
    hipLaunchKernelGGL(canny_finish_kernel, dim3(nb), dim3(256), 0, s, (const unsigned char *)state, out, n);
    KD_CHECK_LAUNCH("kd_canny_continue");
    return KD_OK;
}

extern "C" int kd_conv3x3_small(const void *x, int32_t ldx, const void *w, const float *bias, const void *res, int32_t ldres, void *y,
                                int32_t ldy, int32_t N, int32_t H, int32_t W, int32_t C, int32_t relu, kd_stream_t stream)
{
    KD_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, KD_ERR_INVALID, "kd_conv3x3_small: bad argument");
    KD_REQUIRE(C == 16 || C == 32 || C == 64, KD_ERR_UNSUPPORTED, "kd_conv3x3_small: C must be 16, 32 or 64 (got %d)", C);
    KD_REQUIRE(ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0 && kd_aligned16(x) && kd_aligned16(w) && kd_aligned16(y),
               KD_ERR_INVALID, "kd_conv3x3_small: x / y need 16-B aligned pixels (ld %% 8)");
    KD_REQUIRE(!res || (ldres >= C && ldres % 8 == 0 && kd_aligned16(res)), KD_ERR_INVALID,
               "kd_conv3x3_small: res needs 16-B aligned pixels (ld %% 8)");
    KD_REQUIRE(x != y, KD_ERR_INVALID, "kd_conv3x3_small: y must not alias x");
    SmallConvParams p;
    p.x = (const bf16_t *)x; p.w = (const bf16_t *)w; p.bias = bias; p.res = (const bf16_t *)res; p.y = (bf16_t *)y;
    p.ldx = ldx; p.ldres = ldres; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.relu = relu;
    p.nsx = (W + SC_SEG - 1) / SC_SEG;
    p.nsy = (H + SC_ROWS - 1) / SC_ROWS;
    const long long blocks = (long long)N * p.nsx * p.nsy;
    KD_REQUIRE(blocks <= 0x7fffffffLL, KD_ERR_UNSUPPORTED, "kd_conv3x3_small: image too large");
    hipStream_t s = (hipStream_t)stream;
    const int lds = 4 * (SC_SEG + 2) * (C == 64 ? C * 2 + 16 : C * 2);
    if (C == 64) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void *)conv3x3_small_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                kd_set_error("kd_conv3x3_small: cannot reserve %d B of LDS", lds);
                return KD_ERR_HIP;
            }
            attr_set = true;
        }
        KD_NOTE_KERNEL("conv3x3_small_kernel<64>");
        hipLaunchKernelGGL(conv3x3_small_kernel<64>, dim3((unsigned)blocks), dim3(512), lds, s, p);
    } else if (C == 32) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void *)conv3x3_small_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                kd_set_error("kd_conv3x3_small: cannot reserve %d B of LDS", lds);
                return KD_ERR_HIP;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL(conv3x3_small_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, s, p);
    } else {
        hipLaunchKernelGGL(conv3x3_small_kernel<16>, dim3((unsigned)blocks), dim3(256), lds, s, p);
    }
    KD_CHECK_LAUNCH("kd_conv3x3_small");
    return KD_OK;
}

extern "C" int kd_pointwise_small(const void *x, int32_t ldx, const float *w, const float *bias, void *y, int32_t ldy, int64_t npix,
                                  int32_t Cin, int32_t Cout, kd_stream_t stream)
{
    KD_REQUIRE(x && w && y && npix > 0, KD_ERR_INVALID, "kd_pointwise_small: bad argument");
    KD_REQUIRE((Cin == 64 && Cout == 32) || (Cin == 32 && Cout == 16) || (Cin == 16 && Cout == 8), KD_ERR_UNSUPPORTED,
               "kd_pointwise_small: (Cin, Cout) must be (64, 32), (32, 16) or (16, 8) (got %d, %d)", Cin, Cout);
    KD_REQUIRE(ldx >= Cin && ldy >= Cout && ldx % 8 == 0 && ldy % 4 == 0 && kd_aligned16(x) && ((uintptr_t)y % 8) == 0, KD_ERR_INVALID,
               "kd_pointwise_small: x needs 16-B aligned pixels (ldx %% 8), y 8-B aligned channel quads (ldy %% 4)");
    const long long waves = (npix + 15) / 16;
    const int nb = blocks_for((waves + 3) / 4 * 256, 256 * 32);
    hipStream_t s = (hipStream_t)stream;
#define KD_PWS(CI, CO) hipLaunchKernelGGL((pointwise_small_kernel<CI, CO>), dim3(nb), dim3(256), 0, s, (const bf16_t *)x, ldx, w, bias, (bf16_t *)y, ldy, (long long)npix)
    if (Cin == 64) KD_PWS(64, 32); else if (Cin == 32) KD_PWS(32, 16); else KD_PWS(16, 8);
#undef KD_PWS
    KD_CHECK_LAUNCH("kd_pointwise_small");
    return KD_OK;
}

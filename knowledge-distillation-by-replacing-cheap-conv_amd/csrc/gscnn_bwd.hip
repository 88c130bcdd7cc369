// Backward of the Gated-SCNN shape stream (reference models/gscnn/gscnn.py:269-314 under autograd): the per-pixel, small-channel
// algebra the forward kernels of gscnn_ops.hip fuse, decomposed into a few general pieces so that every layer of the stream --
// the 1x1 squeezes d1..d3, the three GatedSpatialConv2d (gate_spatial_conv.py:50-60), fuse / cw / the two sigmoids, the dsn side
// outputs and the edge branch of the ASPP module -- gets its input gradient and its weight gradients:
//   kd_small_linear        y[p][co] (+)= bias[co] + sum_ci w[co][ci] x[p][ci]  (optionally ReLU): any 1x1 map with <= 72 channels on
//                          either side -- used forward (recomputing a gate's hidden activations) and, with the transposed matrix,
//                          as the input gradient of every such map
//   kd_small_wgrad         dW[cb][ca] = sum_p b[p][cb] a[p][ca], db[cb] = sum_p b[p][cb]: the weight / bias gradient of the same maps
//                          (two-stage, fixed-order: deterministic)
//   kd_gate_mix_bwd        the gate's multiplicative tail  v = feat * (sigmoid(a) + 1)  and its backward
//   kd_edge_attention_bwd  acts = sigmoid(cw0 * sigmoid(fuse . cs) + cw1 * canny): gradients w.r.t. both pre-activations
//   kd_rank1_add           y[p][c] += g[p] * w[c]: the input gradient of a C -> 1 side output (dsn3 / dsn4 / dsn7)
// This is the training-mode path of BASELINE config 5 beyond its shipped plan (loss terms that reach the logits, `aspp` hints,
// trainable shape-stream parameters); it is built for correctness and determinism, not tuned: operands are read with scalar
// loads of either storage type, the arithmetic is fp32.
#include "kd_common.h"

namespace {

inline bool ok_dt(int d) { return d == KD_F32 || d == KD_BF16; }
inline int blocks_for(long long total, int cap = 1 << 16)
{
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

constexpr int SL_MAXC = 72;

// ---- y = x W^T + b ------------------------------------------------------------------------------------------------------------
// weights transposed into LDS as [ci][COP] so that one pixel's multiply-accumulates read consecutive, lane-uniform words
template <int COP>
__global__ __launch_bounds__(256) void small_linear_kernel(const void *__restrict__ x, int xdt, int ldx, int Cin, const float *__restrict__ w,
                                                           const float *__restrict__ bias, void *__restrict__ y, int ydt, int ldy, int Cout,
                                                           long long npix, int accumulate, int relu, const float *__restrict__ mask, int ldm)
{
    __shared__ float sw[SL_MAXC * COP];
    __shared__ float sb[COP];
    for (int i = threadIdx.x; i < Cin * COP; i += 256) {
        const int ci = i / COP, co = i - ci * COP;
        sw[i] = co < Cout ? w[co * Cin + ci] : 0.f;
    }
    for (int i = threadIdx.x; i < COP; i += 256) sb[i] = (bias && i < Cout) ? bias[i] : 0.f;
    __syncthreads();
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float acc[COP];
#pragma unroll
        for (int co = 0; co < COP; ++co) acc[co] = sb[co];
        for (int ci = 0; ci < Cin; ++ci) {
            const float v = kd_ld(x, xdt, p * ldx + ci);
            const float *wr = &sw[ci * COP];
#pragma unroll
            for (int co = 0; co < COP; ++co) acc[co] = fmaf(wr[co], v, acc[co]);
        }
#pragma unroll
        for (int co = 0; co < COP; ++co)
            if (co < Cout) {
                float v = acc[co];
                if (relu) v = fmaxf(v, 0.f);
                if (mask && !(mask[p * ldm + co] > 0.f)) v = 0.f;     // backward through a ReLU whose output is `mask`
                if (accumulate) v += kd_ld(y, ydt, p * ldy + co);
                kd_st(y, ydt, p * ldy + co, v);
            }
    }
}

// ---- dW = b^T a, db = sum b ------------------------------------------------------------------------------------------------------
// a block walks its pixel range in chunks of 64 staged in LDS as fp32; thread t owns the (cb, ca) pairs t, t + 256, ...
constexpr int SW_CH = 64, SW_MAXP = (SL_MAXC * SL_MAXC + 255) / 256;
__global__ __launch_bounds__(256) void small_wgrad_partial_kernel(const void *__restrict__ a, int adt, int lda, int Ca, const void *__restrict__ b,
                                                                  int bdt, int ldb, int Cb, long long npix, long long per_block,
                                                                  float *__restrict__ part /* [blocks][Cb*Ca + Cb] */)
{
    __shared__ float sa[SW_CH * SL_MAXC], sbm[SW_CH * SL_MAXC];
    const int tid = threadIdx.x, total = Ca * Cb;
    float acc[SW_MAXP], accb = 0.f;
#pragma unroll
    for (int k = 0; k < SW_MAXP; ++k) acc[k] = 0.f;
    const long long p_begin = (long long)blockIdx.x * per_block, p_end = min(npix, p_begin + per_block);
    for (long long p0 = p_begin; p0 < p_end; p0 += SW_CH) {
        const int n = (int)min((long long)SW_CH, p_end - p0);
        __syncthreads();
        for (int i = tid; i < SW_CH * Ca; i += 256) {
            const int r = i / Ca, c = i - r * Ca;
            sa[r * Ca + c] = r < n ? kd_ld(a, adt, (p0 + r) * lda + c) : 0.f;
        }
        for (int i = tid; i < SW_CH * Cb; i += 256) {
            const int r = i / Cb, c = i - r * Cb;
            sbm[r * Cb + c] = r < n ? kd_ld(b, bdt, (p0 + r) * ldb + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SW_MAXP; ++k) {
            const int idx = tid + k * 256;
            if (idx < total) {
                const int cb = idx / Ca, ca = idx - cb * Ca;
                float s = acc[k];
                for (int r = 0; r < SW_CH; ++r) s = fmaf(sbm[r * Cb + cb], sa[r * Ca + ca], s);
                acc[k] = s;
            }
        }
        if (tid < Cb)
            for (int r = 0; r < SW_CH; ++r) accb += sbm[r * Cb + tid];
    }
    float *o = part + (size_t)blockIdx.x * (total + Cb);
#pragma unroll
    for (int k = 0; k < SW_MAXP; ++k) {
        const int idx = tid + k * 256;
        if (idx < total) o[idx] = acc[k];
    }
    if (tid < Cb) o[total + tid] = accb;
}

__global__ __launch_bounds__(256) void small_wgrad_finish_kernel(const float *__restrict__ part, int nblocks, int total, int Cb, float *__restrict__ dw,
                                                                 float *__restrict__ db, int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total + Cb) return;
    double s = 0.0;
    for (int k = 0; k < nblocks; ++k) s += (double)part[(size_t)k * (total + Cb) + i];   // fixed order
    if (i < total) dw[i] = (accumulate ? dw[i] : 0.f) + (float)s;
    else if (db) db[i - total] = (accumulate ? db[i - total] : 0.f) + (float)s;
}

// ---- v = feat * (sigmoid(a) + 1) and its backward ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const void *__restrict__ feat, int fdt, int ldf, const float *__restrict__ a,
                                                           const float *__restrict__ gv, int ldgv, float *__restrict__ gfeat, int ldgf,
                                                           float *__restrict__ ga, float *__restrict__ v, int ldv, int C, long long npix)
{
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const float al = 1.f / (1.f + __expf(-a[p]));
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float f = kd_ld(feat, fdt, p * ldf + c);
            const float g = gv ? gv[p * ldgv + c] : 0.f;
            dot = fmaf(g, f, dot);
            if (gfeat) gfeat[p * ldgf + c] = g * (al + 1.f);
            if (v) v[p * ldv + c] = f * (al + 1.f);
        }
        if (ga) ga[p] = dot * al * (1.f - al);
    }
}

// ---- edge attention backward ---------------------------------------------------------------------------------------------------------
// s = fuse . cs; eo = sigmoid(s); t = cw0 eo + cw1 canny; acts = sigmoid(t).  Given g = dL/dacts:
//   g_t = g acts (1 - acts)   (-> dcw = sum g_t [eo, canny]);   g_s = g_t cw0 eo (1 - eo)   (-> dfuse = sum g_s cs, g_cs = g_s fuse)
__global__ __launch_bounds__(256) void edge_attention_bwd_kernel(const void *__restrict__ cs, int cdt, int ldc, const float *__restrict__ canny,
                                                                 const float *__restrict__ w, const float *__restrict__ g, float *__restrict__ gt,
                                                                 float *__restrict__ gs, float *__restrict__ eo_canny /* [p][2] */, long long npix)
{
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(w[e], kd_ld(cs, cdt, p * ldc + e), s);
        const float eo = 1.f / (1.f + __expf(-s));
        const float t = w[8] * eo + w[9] * canny[p];
        const float ac = 1.f / (1.f + __expf(-t));
        const float g_t = g[p] * ac * (1.f - ac);
        gt[p] = g_t;
        gs[p] = g_t * w[8] * eo * (1.f - eo);
        eo_canny[2 * p] = eo;
        eo_canny[2 * p + 1] = canny[p];
    }
}

// ---- y[p][c] += g[p] * w[c] --------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rank1_add_kernel(T *__restrict__ y, int ldy, const float *__restrict__ g, const float *__restrict__ w, int C8,
                                                        long long npix, int accumulate)
{
    const long long total = npix * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / C8;
        const int cq = (int)(i - p * C8);
        float wv[8], v[8];
        ld8(w + cq * 8, wv);
        if (accumulate) ld8(y + p * ldy + cq * 8, v);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = 0.f;
        }
        const float gp = g[p];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaf(gp, wv[q], v[q]);
        st8(y + p * ldy + cq * 8, v);
    }
}

}  // namespace

extern "C" int kd_small_linear(const void *x, int32_t x_dtype, int32_t ldx, int32_t Cin, const float *w, const float *bias, void *y,
                               int32_t y_dtype, int32_t ldy, int32_t Cout, int64_t npix, int32_t accumulate, int32_t relu,
                               const float *mask, int32_t ldm, kd_stream_t stream)
{
    KD_REQUIRE(x && w && y && npix > 0, KD_ERR_INVALID, "kd_small_linear: null argument");
    KD_REQUIRE(ok_dt(x_dtype) && ok_dt(y_dtype), KD_ERR_INVALID, "kd_small_linear: bad dtype");
    KD_REQUIRE(!mask || ldm >= Cout, KD_ERR_INVALID, "kd_small_linear: bad mask stride");
    KD_REQUIRE(Cin >= 1 && Cin <= SL_MAXC && Cout >= 1 && Cout <= SL_MAXC && ldx >= Cin && ldy >= Cout, KD_ERR_UNSUPPORTED,
               "kd_small_linear: 1..%d channels on each side (got %d -> %d)", SL_MAXC, Cin, Cout);
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for(npix);
#define KD_SL(COP) hipLaunchKernelGGL(small_linear_kernel<COP>, dim3(nb), dim3(256), 0, s, x, x_dtype, ldx, Cin, w, bias, y, y_dtype, ldy, Cout, \
                                      (long long)npix, accumulate, relu, mask, ldm)
    if (Cout <= 8) KD_SL(8);
    else if (Cout <= 16) KD_SL(16);
    else if (Cout <= 24) KD_SL(24);
    else if (Cout <= 40) KD_SL(40);
    else KD_SL(72);
#undef KD_SL
    KD_CHECK_LAUNCH("kd_small_linear");
    return KD_OK;
}

static int small_wgrad_blocks(int64_t npix, long long *per_block)
{
    long long nb = (npix + 4095) / 4096;      // >= 64 chunks of 64 pixels per block
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    long long pb = (npix + nb - 1) / nb;
    pb = (pb + SW_CH - 1) / SW_CH * SW_CH;
    nb = (npix + pb - 1) / pb;
    *per_block = pb;
    return (int)nb;
}

extern "C" size_t kd_small_wgrad_workspace(int32_t Ca, int32_t Cb, int64_t npix)
{
    if (Ca < 1 || Cb < 1 || npix < 1) return 0;
    long long pb;
    return (size_t)small_wgrad_blocks(npix, &pb) * ((size_t)Ca * Cb + Cb) * sizeof(float);
}

extern "C" int kd_small_wgrad(const void *a, int32_t a_dtype, int32_t lda, int32_t Ca, const void *b, int32_t b_dtype, int32_t ldb, int32_t Cb,
                              int64_t npix, float *dw, float *db, int32_t accumulate, void *workspace, size_t workspace_bytes,
                              kd_stream_t stream)
{
    KD_REQUIRE(a && b && dw && workspace && npix > 0, KD_ERR_INVALID, "kd_small_wgrad: null argument");
    KD_REQUIRE(ok_dt(a_dtype) && ok_dt(b_dtype), KD_ERR_INVALID, "kd_small_wgrad: bad dtype");
    KD_REQUIRE(Ca >= 1 && Ca <= SL_MAXC && Cb >= 1 && Cb <= SL_MAXC && lda >= Ca && ldb >= Cb, KD_ERR_UNSUPPORTED,
               "kd_small_wgrad: 1..%d channels on each side (got %d x %d)", SL_MAXC, Ca, Cb);
    KD_REQUIRE(workspace_bytes >= kd_small_wgrad_workspace(Ca, Cb, npix), KD_ERR_WORKSPACE, "kd_small_wgrad: workspace too small");
    long long pb;
    const int nb = small_wgrad_blocks(npix, &pb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(small_wgrad_partial_kernel, dim3(nb), dim3(256), 0, s, a, a_dtype, lda, Ca, b, b_dtype, ldb, Cb, (long long)npix, pb,
                       (float *)workspace);
    KD_CHECK_LAUNCH("kd_small_wgrad(partial)");
    const int total = Ca * Cb;
    hipLaunchKernelGGL(small_wgrad_finish_kernel, dim3((total + Cb + 255) / 256), dim3(256), 0, s, (const float *)workspace, nb, total, Cb, dw, db,
                       accumulate);
    KD_CHECK_LAUNCH("kd_small_wgrad(finish)");
    return KD_OK;
}

extern "C" int kd_gate_mix_bwd(const void *feat, int32_t feat_dtype, int32_t ldf, const float *a, const float *gv, int32_t ldgv, float *gfeat,
                               int32_t ldgf, float *ga, float *v, int32_t ldv, int32_t C, int64_t npix, kd_stream_t stream)
{
    KD_REQUIRE(feat && a && npix > 0 && C >= 1, KD_ERR_INVALID, "kd_gate_mix_bwd: null argument");
    KD_REQUIRE(ok_dt(feat_dtype), KD_ERR_INVALID, "kd_gate_mix_bwd: bad dtype");
    KD_REQUIRE((gfeat == nullptr && ga == nullptr) || gv, KD_ERR_INVALID, "kd_gate_mix_bwd: gradients requested without gv");
    hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, feat, feat_dtype, ldf, a, gv, ldgv, gfeat,
                       ldgf, ga, v, ldv, C, (long long)npix);
    KD_CHECK_LAUNCH("kd_gate_mix_bwd");
    return KD_OK;
}

extern "C" int kd_edge_attention_bwd(int32_t dtype, const void *cs, int32_t ldc, const float *canny, const float *weights, const float *g_acts,
                                     float *g_t, float *g_s, float *eo_canny, int64_t npix, kd_stream_t stream)
{
    KD_REQUIRE(cs && canny && weights && g_acts && g_t && g_s && eo_canny && npix > 0, KD_ERR_INVALID, "kd_edge_attention_bwd: null argument");
    KD_REQUIRE(ok_dt(dtype) && ldc >= 8, KD_ERR_INVALID, "kd_edge_attention_bwd: bad dtype / ldc");
    hipLaunchKernelGGL(edge_attention_bwd_kernel, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, cs, dtype, ldc, canny, weights, g_acts,
                       g_t, g_s, eo_canny, (long long)npix);
    KD_CHECK_LAUNCH("kd_edge_attention_bwd");
    return KD_OK;
}

extern "C" int kd_rank1_add(int32_t dtype, void *y, int32_t ldy, const float *g, const float *w, int32_t C, int64_t npix, int32_t accumulate,
                            kd_stream_t stream)
{
    KD_REQUIRE(y && g && w && npix > 0 && C > 0, KD_ERR_INVALID, "kd_rank1_add: null argument");
    KD_REQUIRE(ok_dt(dtype), KD_ERR_INVALID, "kd_rank1_add: bad dtype");
    const int es = kd_elem_size(dtype);
    KD_REQUIRE(C % 8 == 0 && kd_aligned16(y) && (ldy * es) % 16 == 0 && kd_aligned16(w), KD_ERR_INVALID,
               "kd_rank1_add: C %% 8 and 16-B aligned rows required");
    const int nb = blocks_for(npix * (C / 8), 1 << 20);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KD_BF16) hipLaunchKernelGGL(rank1_add_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (bf16_t *)y, ldy, g, w, C / 8, (long long)npix, accumulate);
    else hipLaunchKernelGGL(rank1_add_kernel<float>, dim3(nb), dim3(256), 0, s, (float *)y, ldy, g, w, C / 8, (long long)npix, accumulate);
    KD_CHECK_LAUNCH("kd_rank1_add");
    return KD_OK;
}

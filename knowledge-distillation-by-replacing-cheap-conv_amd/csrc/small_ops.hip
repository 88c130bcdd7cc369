// Small-shape path of the CIFAR plumbing config (BASELINE config 1: ResNet-20 teacher -> student, ClassificationTrainer,
// trainer/classification_trainer.py:18-95; models/cifar_models/resnet.py): 3 / 16 / 32 / 64 channels at 32x32 .. 8x8, which the
// MFMA implicit-GEMM kernels do not take (their K granule is 32 / 64 channels), and BatchNorm2d in TRAINING mode
// (classification_trainer.py:21 calls model.train()).  NCHW fp32 like the reference's tensors: the whole
// 0.27 M-parameter network is launch-bound, so these are plain, exact, deterministic kernels (direct convolution, per-channel
// block reductions in a fixed order) rather than tiled ones.
#include "kd_common.h"

namespace {

struct DConv {
    int N, C, H, W, K, kh, kw, stride, pad, dil, groups, Ho, Wo, Cg, Kg;
};

// y[n][k][ho][wo] = bias[k] + sum_{c in group, ky, kx} x[n][c][ho*s-p+ky*d][wo*s-p+kx*d] * w[k][c][ky][kx]
__global__ __launch_bounds__(256) void dconv_fwd_kernel(DConv d, const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y)
{
    const long long total = (long long)d.N * d.K * d.Ho * d.Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int wo = (int)(i % d.Wo);
        long long r = i / d.Wo;
        const int ho = (int)(r % d.Ho); r /= d.Ho;
        const int k = (int)(r % d.K), n = (int)(r / d.K);
        const int g = k / d.Kg;
        float acc = bias ? bias[k] : 0.f;
        for (int c = 0; c < d.Cg; ++c) {
            const float *xp = x + ((size_t)n * d.C + g * d.Cg + c) * d.H * d.W;
            const float *wp = w + ((size_t)k * d.Cg + c) * d.kh * d.kw;
            for (int ky = 0; ky < d.kh; ++ky) {
                const int hi = ho * d.stride - d.pad + ky * d.dil;
                if (hi < 0 || hi >= d.H) continue;
                for (int kx = 0; kx < d.kw; ++kx) {
                    const int wi = wo * d.stride - d.pad + kx * d.dil;
                    if (wi < 0 || wi >= d.W) continue;
                    acc = fmaf(xp[(size_t)hi * d.W + wi], wp[ky * d.kw + kx], acc);
                }
            }
        }
        y[i] = acc;
    }
}

// dx[n][c][hi][wi] = sum_{k in group, ky, kx : (hi+p-ky*d) % s == 0} dy[n][k][(hi+p-ky*d)/s][..] * w[k][c][ky][kx]
__global__ __launch_bounds__(256) void dconv_dgrad_kernel(DConv d, const float *__restrict__ dy, const float *__restrict__ w,
                                                          float *__restrict__ dx)
{
    const long long total = (long long)d.N * d.C * d.H * d.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int wi = (int)(i % d.W);
        long long r = i / d.W;
        const int hi = (int)(r % d.H); r /= d.H;
        const int c = (int)(r % d.C), n = (int)(r / d.C);
        const int g = c / d.Cg, cl = c - g * d.Cg;
        float acc = 0.f;
        for (int kk = 0; kk < d.Kg; ++kk) {
            const int k = g * d.Kg + kk;
            const float *gp = dy + ((size_t)n * d.K + k) * d.Ho * d.Wo;
            const float *wp = w + ((size_t)k * d.Cg + cl) * d.kh * d.kw;
            for (int ky = 0; ky < d.kh; ++ky) {
                const int t = hi + d.pad - ky * d.dil;
                if (t < 0 || t % d.stride) continue;
                const int ho = t / d.stride;
                if (ho >= d.Ho) continue;
                for (int kx = 0; kx < d.kw; ++kx) {
                    const int u = wi + d.pad - kx * d.dil;
                    if (u < 0 || u % d.stride) continue;
                    const int wo = u / d.stride;
                    if (wo >= d.Wo) continue;
                    acc = fmaf(gp[(size_t)ho * d.Wo + wo], wp[ky * d.kw + kx], acc);
                }
            }
        }
        dx[i] = acc;
    }
}

// fixed-order block sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum256(float v, float *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// dw[k][c][ky][kx] = sum_{n,ho,wo} dy[n][k][ho][wo] * x[n][g*Cg+c][ho*s-p+ky*d][wo*s-p+kx*d]; one block per (k, c) pair
__global__ __launch_bounds__(256) void dconv_wgrad_kernel(DConv d, const float *__restrict__ x, const float *__restrict__ dy,
                                                          float *__restrict__ dw, int accumulate)
{
    __shared__ float sh[4];
    const int k = blockIdx.x / d.Cg, cl = blockIdx.x - k * d.Cg;
    const int g = k / d.Kg;
    const int npix = d.N * d.Ho * d.Wo;
    for (int ky = 0; ky < d.kh; ++ky)
        for (int kx = 0; kx < d.kw; ++kx) {
            float acc = 0.f;
            for (int p = threadIdx.x; p < npix; p += 256) {
                const int wo = p % d.Wo;
                const int r = p / d.Wo;
                const int ho = r % d.Ho, n = r / d.Ho;
                const int hi = ho * d.stride - d.pad + ky * d.dil, wi = wo * d.stride - d.pad + kx * d.dil;
                if (hi < 0 || hi >= d.H || wi < 0 || wi >= d.W) continue;
                acc = fmaf(dy[(((size_t)n * d.K + k) * d.Ho + ho) * d.Wo + wo],
                           x[(((size_t)n * d.C + g * d.Cg + cl) * d.H + hi) * d.W + wi], acc);
            }
            const float s = block_sum256(acc, sh);
            if (threadIdx.x == 0) {
                float *o = dw + (((size_t)k * d.Cg + cl) * d.kh + ky) * d.kw + kx;
                *o = accumulate ? *o + s : s;
            }
        }
}

// dbias[k] = sum_{n,ho,wo} dy[n][k][ho][wo]
__global__ __launch_bounds__(256) void dconv_bias_grad_kernel(DConv d, const float *__restrict__ dy, float *__restrict__ db,
                                                              int accumulate)
{
    __shared__ float sh[4];
    const int k = blockIdx.x, hw = d.Ho * d.Wo;
    float acc = 0.f;
    for (int p = threadIdx.x; p < d.N * hw; p += 256) acc += dy[((size_t)(p / hw) * d.K + k) * hw + p % hw];
    const float s = block_sum256(acc, sh);
    if (threadIdx.x == 0) db[k] = accumulate ? db[k] + s : s;
}

// ---- BatchNorm2d, NCHW fp32, one block per channel -------------------------------------------------------------------------
// training: batch mean / biased variance (two passes: mean, then centred sum of squares), running statistics updated with the
// unbiased variance like nn.BatchNorm2d; eval: the running statistics.  Optional fused ReLU.
__global__ __launch_bounds__(256) void bn2d_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ y,
                                                       float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                       float *__restrict__ run_mean, float *__restrict__ run_var, float momentum,
                                                       float eps, int training, int relu, int N, int C, int HW)
{
    __shared__ float sh[4];
    __shared__ float stat[2];
    const int c = blockIdx.x, M = N * HW;
    if (training) {
        float s = 0.f;
        for (int p = threadIdx.x; p < M; p += 256) s += x[((size_t)(p / HW) * C + c) * HW + p % HW];
        s = block_sum256(s, sh);
        if (threadIdx.x == 0) stat[0] = s / (float)M;
        __syncthreads();
        const float mean = stat[0];
        float q = 0.f;
        for (int p = threadIdx.x; p < M; p += 256) {
            const float v = x[((size_t)(p / HW) * C + c) * HW + p % HW] - mean;
            q = fmaf(v, v, q);
        }
        q = block_sum256(q, sh);
        if (threadIdx.x == 0) {
            const float var = q / (float)M;
            stat[1] = rsqrtf(var + eps);
            if (save_mean) save_mean[c] = mean;
            if (save_invstd) save_invstd[c] = stat[1];
            if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (M > 1 ? q / (float)(M - 1) : var);
        }
        __syncthreads();
    } else {
        if (threadIdx.x == 0) {
            stat[0] = run_mean[c];
            stat[1] = rsqrtf(run_var[c] + eps);
            if (save_mean) save_mean[c] = stat[0];
            if (save_invstd) save_invstd[c] = stat[1];
        }
        __syncthreads();
    }
    const float sc = gamma[c] * stat[1], shf = beta[c] - stat[0] * sc;
    for (int p = threadIdx.x; p < M; p += 256) {
        const size_t i = ((size_t)(p / HW) * C + c) * HW + p % HW;
        const float v = fmaf(x[i], sc, shf);
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

// backward.  dy' = relu ? dy * [y > 0] : dy;  xhat = (x - mean) * invstd
//   dbeta = sum dy',  dgamma = sum dy' * xhat
//   training: dx = gamma * invstd * (dy' - dbeta / M - xhat * dgamma / M);   eval: dx = gamma * invstd * dy'
__global__ __launch_bounds__(256) void bn2d_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                       const float *__restrict__ y, const float *__restrict__ gamma,
                                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                                       float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                       int training, int relu, int accumulate, int N, int C, int HW)
{
    __shared__ float sh[4];
    __shared__ float stat[2];
    const int c = blockIdx.x, M = N * HW;
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int p = threadIdx.x; p < M; p += 256) {
        const size_t i = ((size_t)(p / HW) * C + c) * HW + p % HW;
        const float g = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        s1 += g;
        s2 = fmaf(g, (x[i] - mu) * is, s2);
    }
    s1 = block_sum256(s1, sh);
    s2 = block_sum256(s2, sh);
    if (threadIdx.x == 0) {
        stat[0] = s1;
        stat[1] = s2;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
    }
    __syncthreads();
    if (!dx) return;
    const float k = gamma[c] * is, m1 = stat[0] / (float)M, m2 = stat[1] / (float)M;
    for (int p = threadIdx.x; p < M; p += 256) {
        const size_t i = ((size_t)(p / HW) * C + c) * HW + p % HW;
        const float g = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        dx[i] = training ? k * (g - m1 - (x[i] - mu) * is * m2) : k * g;
    }
}

inline int blocks_for(long long total)
{
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

int fill(const kd_dconv_desc *s, DConv &d, const char *who)
{
    KD_REQUIRE(s, KD_ERR_INVALID, "%s: null descriptor", who);
    KD_REQUIRE(s->N > 0 && s->C > 0 && s->H > 0 && s->W > 0 && s->K > 0 && s->kh > 0 && s->kw > 0 && s->stride >= 1 && s->pad >= 0 &&
                   s->dil >= 1 && s->groups >= 1,
               KD_ERR_INVALID, "%s: bad descriptor", who);
    KD_REQUIRE(s->C % s->groups == 0 && s->K % s->groups == 0, KD_ERR_INVALID, "%s: channels not divisible by groups", who);
    d.N = s->N; d.C = s->C; d.H = s->H; d.W = s->W; d.K = s->K; d.kh = s->kh; d.kw = s->kw; d.stride = s->stride; d.pad = s->pad;
    d.dil = s->dil; d.groups = s->groups;
    d.Ho = (s->H + 2 * s->pad - s->dil * (s->kh - 1) - 1) / s->stride + 1;
    d.Wo = (s->W + 2 * s->pad - s->dil * (s->kw - 1) - 1) / s->stride + 1;
    KD_REQUIRE(d.Ho > 0 && d.Wo > 0, KD_ERR_INVALID, "%s: empty output", who);
    d.Cg = s->C / s->groups; d.Kg = s->K / s->groups;
    KD_REQUIRE((long long)s->N * s->C * s->H * s->W < (1ll << 31) && (long long)s->N * s->K * d.Ho * d.Wo < (1ll << 31), KD_ERR_UNSUPPORTED,
               "%s: tensor exceeds 2^31 elements", who);
    return KD_OK;
}

}  // namespace

extern "C" int kd_conv2d_direct_fwd(const kd_dconv_desc *s, const float *x, const float *w, const float *bias, float *y,
                                    kd_stream_t stream)
{
    DConv d;
    if (int rc = fill(s, d, "kd_conv2d_direct_fwd")) return rc;
    KD_REQUIRE(x && w && y, KD_ERR_INVALID, "kd_conv2d_direct_fwd: null argument");
    hipLaunchKernelGGL(dconv_fwd_kernel, dim3(blocks_for((long long)d.N * d.K * d.Ho * d.Wo)), dim3(256), 0, (hipStream_t)stream, d, x, w,
                       bias, y);
    KD_CHECK_LAUNCH("kd_conv2d_direct_fwd");
    return KD_OK;
}

extern "C" int kd_conv2d_direct_dgrad(const kd_dconv_desc *s, const float *dy, const float *w, float *dx, kd_stream_t stream)
{
    DConv d;
    if (int rc = fill(s, d, "kd_conv2d_direct_dgrad")) return rc;
    KD_REQUIRE(dy && w && dx, KD_ERR_INVALID, "kd_conv2d_direct_dgrad: null argument");
    hipLaunchKernelGGL(dconv_dgrad_kernel, dim3(blocks_for((long long)d.N * d.C * d.H * d.W)), dim3(256), 0, (hipStream_t)stream, d, dy, w, dx);
    KD_CHECK_LAUNCH("kd_conv2d_direct_dgrad");
    return KD_OK;
}

extern "C" int kd_conv2d_direct_wgrad(const kd_dconv_desc *s, const float *x, const float *dy, float *dw, float *dbias,
                                      int32_t accumulate, kd_stream_t stream)
{
    DConv d;
    if (int rc = fill(s, d, "kd_conv2d_direct_wgrad")) return rc;
    KD_REQUIRE(x && dy && (dw || dbias), KD_ERR_INVALID, "kd_conv2d_direct_wgrad: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (dw) {
        hipLaunchKernelGGL(dconv_wgrad_kernel, dim3((unsigned)(d.K * d.Cg)), dim3(256), 0, st, d, x, dy, dw, accumulate);
        KD_CHECK_LAUNCH("kd_conv2d_direct_wgrad");
    }
    if (dbias) {
        hipLaunchKernelGGL(dconv_bias_grad_kernel, dim3((unsigned)d.K), dim3(256), 0, st, d, dy, dbias, accumulate);
        KD_CHECK_LAUNCH("kd_conv2d_direct_wgrad(bias)");
    }
    return KD_OK;
}

extern "C" int kd_bn2d_fwd(const float *x, const float *gamma, const float *beta, float *y, float *save_mean, float *save_invstd,
                           float *running_mean, float *running_var, float momentum, float eps, int32_t training, int32_t relu,
                           int32_t N, int32_t C, int32_t HW, kd_stream_t stream)
{
    KD_REQUIRE(x && gamma && beta && y && N > 0 && C > 0 && HW > 0, KD_ERR_INVALID, "kd_bn2d_fwd: bad argument");
    KD_REQUIRE(training || (running_mean && running_var), KD_ERR_INVALID, "kd_bn2d_fwd: eval mode needs the running statistics");
    KD_REQUIRE((long long)N * C * HW < (1ll << 31), KD_ERR_UNSUPPORTED, "kd_bn2d_fwd: tensor exceeds 2^31 elements");
    hipLaunchKernelGGL(bn2d_fwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, save_mean, save_invstd,
                       running_mean, running_var, momentum, eps, training, relu, N, C, HW);
    KD_CHECK_LAUNCH("kd_bn2d_fwd");
    return KD_OK;
}

extern "C" int kd_bn2d_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *save_mean,
                           const float *save_invstd, float *dx, float *dgamma, float *dbeta, int32_t training, int32_t relu,
                           int32_t accumulate, int32_t N, int32_t C, int32_t HW, kd_stream_t stream)
{
    KD_REQUIRE(dy && x && gamma && save_mean && save_invstd && N > 0 && C > 0 && HW > 0, KD_ERR_INVALID, "kd_bn2d_bwd: bad argument");
    KD_REQUIRE(!relu || y, KD_ERR_INVALID, "kd_bn2d_bwd: the fused-ReLU backward needs the forward output");
    hipLaunchKernelGGL(bn2d_bwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, dy, x, y, gamma, save_mean, save_invstd, dx,
                       dgamma, dbeta, training, relu, accumulate, N, C, HW);
    KD_CHECK_LAUNCH("kd_bn2d_bwd");
    return KD_OK;
}

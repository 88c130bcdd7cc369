// Weight gradient of a 1x1 convolution (kd_pw_wgrad):
//   dw[co][ci] = sum_m dy[m][co] * a[m][ci]            (TN GEMM, reduction over pixels)
// Both operands are pixel-major (NHWC), i.e. the reduction index is the strided one, so
// each K stage (64 pixels bf16 / 32 pixels f32) is transposed on its way into LDS
// (global 16-B row chunks -> registers -> per-element LDS writes) into the same
// [channel][128 B of K] swizzled image the conv kernel uses; the MFMA core is shared.
// Split-K over pixels with fp32 partial slabs in the caller's workspace and a
// fixed-order reduction (deterministic, unlike float atomics).
#include "igemm_core.h"

namespace {

constexpr int TM = 128, TN = 128;
__device__ __attribute__((aligned(256))) uint32_t kd_zero_page_w[64];  // zero-initialised: source of out-of-range DMA lanes
constexpr int STAGE_BYTES = (TM + TN) * IG_ROWB;  // 32 KiB

struct WgradParams {
    const void *a;
    const void *dy;
    float *part;  // [splits][Cout][Cin]
    int M, Cin, Cout, lda, ldy;
    int tiles_ci, rows_per_split, splits;
};

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
    const int tile = blockIdx.x, split = blockIdx.y;
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
        S::load(a, p.lda, m_begin, m_end, ci0, p.Cin, wv, lane, rb);
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
            S::load(a, p.lda, mb, m_end, ci0, p.Cin, wv, lane, rb);
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

    // partial tile -> slab [split][Cout][Cin]
    float *out = p.part + (size_t)split * p.Cout * p.Cin;
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

__global__ __launch_bounds__(256, 2) void pw_wgrad_tr_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tile = blockIdx.x, split = blockIdx.y;
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
            const bf16_t *s0 = (mok && co0 + c < p.Cout) ? dy + (size_t)m * p.ldy + co0 + c : zero;
            const bf16_t *s1 = (mok && ci0 + c < p.Cin) ? a + (size_t)m * p.lda + ci0 + c : zero;
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

    float *out = p.part + (size_t)split * p.Cout * p.Cin;
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

__global__ void slab_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, size_t n, int splits,
                                   int accumulate)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

void plan(int dtype, int M, int Cin, int Cout, int &tiles, int &tiles_ci, int &splits, int &rows_per_split)
{
    const int krows = IG_ROWB / kd_elem_size(dtype);
    tiles_ci = (Cin + TN - 1) / TN;
    tiles = tiles_ci * ((Cout + TM - 1) / TM);
    const int stages = (M + krows - 1) / krows;
    int want = (1024 + tiles - 1) / tiles;          // aim for ~1024 workgroups (4 per CU)
    const int max_splits = (stages + 3) / 4;        // keep >= 4 stages per split
    splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    const int st_per = (stages + splits - 1) / splits;
    rows_per_split = st_per * krows;
    splits = (M + rows_per_split - 1) / rows_per_split;
}

}  // namespace

extern "C" size_t kd_pw_wgrad_workspace(int32_t M, int32_t Cin, int32_t Cout)
{
    // dtype-independent upper bound: plan with the smaller K stage (f32: 32 rows) gives the larger split count
    int tiles, tiles_ci, s0, s1, rps;
    plan(KD_F32, M, Cin, Cout, tiles, tiles_ci, s0, rps);
    plan(KD_BF16, M, Cin, Cout, tiles, tiles_ci, s1, rps);
    const int splits = s0 > s1 ? s0 : s1;
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
    KD_REQUIRE(workspace_bytes >= (size_t)splits * Cout * Cin * sizeof(float), KD_ERR_WORKSPACE,
               "kd_pw_wgrad: workspace %zu < %zu", workspace_bytes, (size_t)splits * Cout * Cin * sizeof(float));
    WgradParams p;
    p.a = a; p.dy = dy; p.part = (float *)workspace;
    p.M = M; p.Cin = Cin; p.Cout = Cout; p.lda = lda; p.ldy = ldy;
    p.tiles_ci = tiles_ci; p.rows_per_split = rps; p.splits = splits;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    if (dtype == KD_BF16 && Cin % 8 == 0 && Cout % 8 == 0) hipLaunchKernelGGL(pw_wgrad_tr_kernel, grid, dim3(256), 0, s, p);
    else if (dtype == KD_BF16) hipLaunchKernelGGL(pw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pw_wgrad_kernel<float>, grid, dim3(256), 0, s, p);
    KD_CHECK_LAUNCH("kd_pw_wgrad");
    const size_t n = (size_t)Cout * Cin;
    const int rb = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float *)workspace, dw, n, splits, accumulate);
    KD_CHECK_LAUNCH("kd_pw_wgrad(reduce)");
    return KD_OK;
}

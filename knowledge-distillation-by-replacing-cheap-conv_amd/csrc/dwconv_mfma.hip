// Depthwise 9x9 dilated convolution on the matrix cores (bf16 activations; kd_dwconv_fwd's fast path, also used for
// the input gradient with flipped taps).
//
// A depthwise conv has no channel contraction, so it is not a GEMM -- but along one image axis it is a banded
// (Toeplitz) matrix product.  On the lattice {r + dil*l} of one residue class (ry, rx) the 9x9 stencil is dense, and
//   out[ly][lx] = sum_ky  sum_k  X[ly + ky][lx0 + k] * T_ky[k][lx - lx0],   T_ky[k][j] = w[ky][k - j] (0 <= k-j < 9)
// i.e. per channel and tap row one v_mfma_f32_16x16x32_bf16: A = 16 lattice rows x 32 lattice columns of the input
// (rows shifted by ky), B = the 32 x 16 Toeplitz band of tap row ky, D = 16 x 16 outputs, accumulated over the 9 tap
// rows.  9 of the 32 products per output are non-zero (28 % of the MFMA's MACs are useful), which at MFMA rates still
// beats the 81-FMA VALU stencil several times over and leaves the op bound by its HBM traffic.
//
// Block (8 waves, one per CU) = (image, 16 channels, a segment of the list of work items); a work item is a residue
// class x lattice tile of <= 26 x 52 outputs.  Wave w owns channels 2w, 2w+1; their Toeplitz operands (2 x 9 x 16 B
// per lane) are built once per block and stay in registers.  Per item:
//   1. the next item's tile + 4-cell halo is fetched to registers (NHWC, 16 channels = 32 B per pixel), a few loads at
//      a time between the MFMA tiles, while
//   2. this item's MFMAs run from the LDS copy X[row][channel][64 cols] (zero outside the image; row stride 2080 B
//      keeps the 16-row ds_read_b128 fragment reads bank-conflict-free): 2 x 4 output tiles x 9 MFMAs per channel;
//   3. accumulators -> LDS as [pixel][16 channels] bf16 (over the X buffer just consumed), then 16-B NHWC stores;
//   4. the prefetched registers are transposed into the other X buffer.
// Edge tiles overlap their neighbours instead of running past the staged rows/columns (identical values are written
// twice), so no LDS read leaves the buffer and every operand is finite.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "igemm_core.h"

long long kd_internal_lattice_rows(int N, int H, int W, int dil);
int kd_internal_dw_lw_fanout(const kd_dw_desc *d, int nb, const void *x, const float *const *ws, void *const *ys, hipStream_t s);

namespace {

constexpr int CG = 16;                      // channels per block
constexpr int NT = 512;                     // threads per block
constexpr int TLY = 26, TLX = 52;           // lattice outputs per work item (rows, cols)
constexpr int RY = TLY + 8;                 // staged lattice rows (4-cell halo each side)
constexpr int NCOL = 64;                    // staged lattice cols per (row, channel): 60 used
constexpr int CSTR = NCOL * 2;              // bytes per (row, channel)
constexpr int RSTR = CG * CSTR + 32;        // row stride: == 32 (mod 256) -> conflict-free fragment reads
constexpr int XBYTES = (RY * RSTR + 64 + 15) & ~15;   // one X buffer (+ tail: K padding of the last column tile)
constexpr int WT_OFF = 2 * XBYTES;
constexpr int WTBYTES = CG * 9 * 16 * 2;    // bf16 taps [channel][ky][16] (kx 9..15 zero)
constexpr int LDS_BYTES = WT_OFF + WTBYTES;   // + WTBYTES per additional summed input
constexpr int OSTR = CG * 2;                // output staging: bytes per pixel
constexpr int ITEMS = RY * 32 * 2;          // stage-in units per work item: (row, column pair, 8-channel half)
constexpr int NIT = (ITEMS + NT - 1) / NT;
constexpr int OPX = 56;                     // output staging: pixels per row (last column tile starts at <= 40)
static_assert(TLY * OPX * OSTR <= RY * RSTR, "output staging must fit in the input buffer it aliases");
static_assert(TLY <= 28 && TLX <= 64 && NT == 512, "store phase: thread = (half, 64 columns, row mod 4), 7 rows each");
static_assert(LDS_BYTES + 2 * WTBYTES <= 160 * 1024, "LDS budget (three summed inputs)");

// Lattice-planar layout ("LP") of an (N,H,W,C) bf16 tensor for dilation d -- the private layout of the tensors only these kernels
// and the 1x1 convs next to them touch (the replaced ASPP branches' depthwise outputs and their gradients):
//   [C/16 planes][rows][16 channels],  row(n, ry, rx, ly, lx) = ((n*d*d + ry*d + rx)*Ly + ly)*Lx + lx,  Ly = ceil(H/d), Lx = ceil(W/d)
// i.e. the pixels of one residue class (ry, rx) of one image are CONSECUTIVE rows in lattice order, so a work item's tile of a
// 16-channel group is one contiguous run of 32-B cells per lattice row (26 x 52 x 32 B = 42 KiB in one piece when a class is one
// tile) instead of 32-B pieces 40 KiB apart in NHWC (tools/ubench/piece_bw.hip: those copy at a third of the rate).  Cells whose
// pixel lies outside the image (yy >= H or xx >= W: classes one row / column shorter) and the rows that pad a plane to a
// multiple of 256 hold zeros; `plane` = rows per plane * 16 elements.
struct LpGeom {
    int Ly, Lx, rpi;     // lattice extent of a class (padded), rows per image = d*d*Ly*Lx
    long long plane;     // elements per plane
};

constexpr int MAXB = 3;                     // inputs one launch can sum (the three ASPP branches)
struct DwMfmaParams {
    const bf16_t *x;     // input 0
    const float *w;      // its taps [81][C]
    const bf16_t *xs[MAXB - 1];   // inputs 1.. of the summing kernel (same geometry and pixel stride), taps ws[]
    const float *ws[MAXB - 1];    // taps of input / output 1..
    bf16_t *y;
    bf16_t *ys[MAXB - 1];         // outputs 1.. of the fan-out kernel (same pixel stride as y)
    int N, H, W, C, dil, ldx, ldy;
    int nty, ntx, ncg;
    int nitems, nseg;    // work items per (image, channel group); segments they are split into
    int dbg;             // timing-only ablation hook (KDCC_DW_DBG): 1 = no MFMA phase, 2 = no output phase, 4 = no LDS fill of the next tile, 8 = no load requests, 16 = no output stores, 32 = no Toeplitz rebuild per branch, 64 = no workgroup barriers in the item loop
    LpGeom lp;           // lattice-planar operands (LP kernels: the fan-out's outputs, the summing kernel's inputs)
};

struct Item {
    int ry, rx, ty, tx, RV, CV;
};

// Tile cells outside the image (the zero padding of the stencil: 8 of the 34 staged rows and 8 of the 60 columns when a
// residue class is one tile) must not cost memory requests: the tile is fetched with BUFFER loads whose offset is pushed
// past num_records for such cells -- the hardware range check returns zeros without sending anything to the L1/L2 (a clamped
// address would be served from cache, but each 32-B piece still occupies a request slot, and the request rate is what bounds
// this kernel).  No branch, no exec masking; a wave whose row is outside the image issues an instruction that fetches nothing.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_OOB = 0x80000000u;   // >= num_records (eligibility keeps an image below 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const bf16_t *base, int H, int W, int ld)
{
    // bytes reachable from `base` (= image n, channel c0): up to the 16 channels of the last pixel
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((((size_t)H * W - 1) * ld + CG) * 2), 0x00020000);
}
__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ Item decode_item(const DwMfmaParams &p, int e)
{
    Item it;
    const int d = p.dil;
    it.rx = e % d; e /= d;
    it.ry = e % d; e /= d;
    it.tx = e % p.ntx;
    it.ty = e / p.ntx;
    const int Ly = (p.H - it.ry + d - 1) / d, Lx = (p.W - it.rx + d - 1) / d;   // lattice extent of this residue class
    it.RV = min(TLY, Ly - it.ty * TLY);
    it.CV = min(TLX, Lx - it.tx * TLX);
    return it;
}

struct Staged {
    uint4 a[NIT], b[NIT];
    uint32_t ok[NIT];
};

// global -> registers: unit (row r, column pair lp, half h) = two pixels of the residue lattice, 8 channels each.
// XLP: the input is lattice-planar and `xr` spans this (image, channel group)'s rows of its plane -- the two pixels of a unit are
// neighbouring 32-B cells, a lattice row of the tile one contiguous run (cells outside the image are still masked here: the
// halo of a tile must read zeros, not the neighbouring lattice row's end)
template <bool XLP = false>
__device__ __forceinline__ void fetch_unit(int it, const DwMfmaParams &p, __amdgpu_buffer_rsrc_t xr, const Item &w, int tid, Staged &s)
{
    const int d = p.dil;
    {
        const int unit = tid + it * NT;
        const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
        const int ly = w.ty * TLY + r - 4, lx = w.tx * TLX + 2 * lp - 4;
        const int yy = w.ry + d * ly, xa = w.rx + d * lx, xb2 = xa + d;
        const bool rok = ly >= 0 && yy < p.H && lp < 30 && unit < ITEMS;
        const bool aok = rok && lx >= 0 && xa < p.W, bok = rok && lx + 1 >= 0 && xb2 < p.W;
        const uint32_t pb = XLP ? 32u : (uint32_t)p.ldx * 2u;
        const uint32_t oa = (XLP ? (uint32_t)(((w.ry * d + w.rx) * p.lp.Ly + ly) * p.lp.Lx + lx) : (uint32_t)(yy * p.W + xa)) * pb + (uint32_t)h * 16u;
        const bool req = !(p.dbg & 8);   // (timing ablation, tuning build: no memory requests at all)
        s.a[it] = bload16(xr, aok && req ? oa : BUF_OOB);
        s.b[it] = bload16(xr, bok && req ? oa + (XLP ? 1u : (uint32_t)d) * pb : BUF_OOB);
        s.ok[it] = (aok ? 0x0000ffffu : 0u) | (bok ? 0xffff0000u : 0u);
    }
}
template <bool XLP = false>
__device__ __forceinline__ void fetch_item(const DwMfmaParams &p, __amdgpu_buffer_rsrc_t xb, const Item &w, int tid, Staged &s)
{
    fetch_unit<XLP>(0, p, xb, w, tid, s); fetch_unit<XLP>(1, p, xb, w, tid, s); fetch_unit<XLP>(2, p, xb, w, tid, s);
    fetch_unit<XLP>(3, p, xb, w, tid, s); fetch_unit<XLP>(4, p, xb, w, tid, s);
}
// this (image, channel group)'s rows of a lattice-planar tensor: rpi rows of 32 B
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lattice_rsrc(const bf16_t *base, const LpGeom &g, int n, int cgi)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)cgi * g.plane + (size_t)n * g.rpi * CG), 0, g.rpi * CG * 2, 0x00020000);
}
static_assert(NIT == 5, "fetch_item / the interleaved fetch below are written for 5 units per thread");

// registers -> X buffer, transposed to per-channel lattice rows (two neighbouring lattice columns per dword)
__device__ __forceinline__ void write_item(char *X, int tid, const Staged &s)
{
    for (int e = tid; e < RY * 8 + 16; e += NT) {   // row pads + tail: K padding, must be finite
        const int r = e >> 3;
        if (r < RY) *(uint32_t *)(X + r * RSTR + CG * CSTR + (e & 7) * 4) = 0u;
        else *(uint32_t *)(X + RY * RSTR + (e - RY * 8) * 4) = 0u;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int unit = tid + it * NT;
        if (unit < ITEMS) {
            const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
            char *dst = X + r * RSTR + (h * 8) * CSTR + lp * 4;
            const uint32_t a[4] = {s.a[it].x, s.a[it].y, s.a[it].z, s.a[it].w};
            const uint32_t b[4] = {s.b[it].x, s.b[it].y, s.b[it].z, s.b[it].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *(uint32_t *)(dst + (2 * m) * CSTR) = ((a[m] & 0xffffu) | (b[m] << 16)) & s.ok[it];
                *(uint32_t *)(dst + (2 * m + 1) * CSTR) = ((a[m] >> 16) | (b[m] & 0xffff0000u)) & s.ok[it];
            }
        }
    }
}

// K slots (round 6, shared with dwconv_lw.hip): an output tile of 16 lattice columns reads 24 input columns per tap row, so one
// MFMA (K = 32) per tap row wastes a quarter of its contraction.  The contraction index is only a label: the 9 x 3 (tap row, 8-column
// group) pairs are dealt into 28 K groups = NM = 7 MFMAs per tile instead of 9.  Slot q = 4 m + kg of MFMA m: lane (i, kg) reads
// row i + ky, columns cg * 8 .. + 7 of the tile, and its B fragment holds taps kx = cg * 8 + e - j of tap row ky.  ky < 0: the 28th
// slot (zero operand).  Slots pair up as the lanes one ds_read_b128 group serves (kg 0 / 1, kg 2 / 3): a pair whose column groups
// differ in parity is conflict-free at RSTR == 32 (mod 256) -- nine (ky, 0) | (ky, 1) pairs and (8, 2) | none; the four pairs of
// the remaining (ky, 2) groups cost one extra LDS cycle per lane group.
constexpr int NM = 7;
__device__ __forceinline__ void slot_of(int q, int &ky, int &cg)
{
    const int pr = q >> 1, e = q & 1;
    if (pr < 9) { ky = pr; cg = e; }
    else if (pr == 9) { ky = e ? -1 : 8; cg = e ? 1 : 2; }
    else { ky = 2 * (pr - 10) + e; cg = 2; }
}
// this lane's byte offset of MFMA m's slot inside a tile (tap row * RSTR + column group * 16), relative to the tile's (row i, column 0)
__device__ __forceinline__ void slot_offsets(int kg, int (&sl)[NM])
{
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        int ky, cg;
        slot_of(4 * m + kg, ky, cg);
        sl[m] = ky < 0 ? 0 : ky * RSTR + cg * 16;
    }
}

// Toeplitz operands of this wave's two channels from the bf16 tap table of input `b`
__device__ __forceinline__ void build_toeplitz(const char *smem, int b, int wave, int fi, int kg, uint4 (&B)[2][NM])
{
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        int ky, cg;
        slot_of(4 * m + kg, ky, cg);
        int widx[8];   // byte offset of tap kx = cg*8 + q - fi in a table row (slot 15 holds zero)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kx = cg * 8 + q - fi;
            widx[q] = (ky >= 0 && kx >= 0 && kx < 9 ? kx : 15) * 2;
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const char *wr = smem + WT_OFF + b * WTBYTES + ((wave * 2 + cc) * 9 + max(ky, 0)) * 32;
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *(const bf16_t *)(wr + widx[q]);
            B[cc][m] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
        }
    }
}

// NB = 1: y = dwconv(x, w).
// NB = 2, 3, FAN = false: y = sum_b dwconv(x_b, w_b) -- the gradient of a tensor that feeds NB depthwise convs of one geometry
// (the ASPP input under the three replaced branches).  The work list becomes (item, input) pairs: the accumulators live
// across the NB inputs of an item, the output phase runs once per item, and the Toeplitz operands of the next input are
// rebuilt from its tap table (144 2-byte LDS reads per wave, under the barrier that publishes the next tile) -- three
// register sets of them would not fit.  Against NB chained launches that is one output pass instead of NB and no
// read-modify-write of the running sum.
// NB = 2, 3, FAN = true: y_b = dwconv(x, w_b) -- the forward of those NB convs: the tile of x is staged ONCE per item and
// stays in its buffer while the NB outputs are computed one after the other (operands rebuilt per output, the output staging
// uses the idle second buffer); the next item's loads go out under the first output's MFMAs.  One read of x instead of NB.
// LP: the tensors on the NB side are lattice-planar -- the fan-out's outputs (x stays NHWC), the summing kernel's inputs (y stays
// NHWC): the 4096-channel intermediates of the replaced ASPP branches, which only these kernels and 1x1 convs touch.
template <int NB, bool FAN, bool LP = false>
__global__ __launch_bounds__(NT, 1) void dw_mfma_fwd_kernel(DwMfmaParams p)
{
    static_assert(NB >= 1 && NB <= MAXB && (NB > 1 || !FAN), "one to three inputs (sum) or outputs (fan-out)");
    constexpr bool XLP = LP && !FAN, YLP = LP && FAN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lin % p.ncg; lin /= p.ncg;
    const int seg = lin % p.nseg;
    const int n = lin / p.nseg;
    const int c0 = cgi * CG;
    const int ibeg = (int)((long long)p.nitems * seg / p.nseg), iend = (int)((long long)p.nitems * (seg + 1) / p.nseg);
    const size_t img = (size_t)n * p.H * p.W * p.ldx + c0;
    const __amdgpu_buffer_rsrc_t xb0 = XLP ? lattice_rsrc(p.x, p.lp, n, cgi) : image_rsrc(p.x + img, p.H, p.W, p.ldx);
    const __amdgpu_buffer_rsrc_t xb1 = XLP ? lattice_rsrc(NB > 1 ? p.xs[0] : p.x, p.lp, n, cgi) : image_rsrc((NB > 1 && !FAN ? p.xs[0] : p.x) + img, p.H, p.W, p.ldx);
    const __amdgpu_buffer_rsrc_t xb2 = XLP ? lattice_rsrc(NB > 2 ? p.xs[1] : p.x, p.lp, n, cgi) : image_rsrc((NB > 2 && !FAN ? p.xs[1] : p.x) + img, p.H, p.W, p.ldx);
    auto rsrc_of = [&](int b) { return NB > 2 && b == 2 ? xb2 : (NB > 1 && b == 1 ? xb1 : xb0); };
    const int d = p.dil;

    // first valid item's loads go out before anything else
    int cur = ibeg;
    Item wi = decode_item(p, cur);
    while (cur < iend && (wi.RV <= 0 || wi.CV <= 0)) { ++cur; if (cur < iend) wi = decode_item(p, cur); }
    if (cur >= iend) return;   // block-uniform
    // Branch order: item i walks its NB branches forwards when i is even, backwards when it is odd, so that the branch an item ends
    // on is the branch the next item starts with and its Toeplitz operands are still in registers: two rebuilds per item instead
    // of three (a rebuild is 144 two-byte LDS reads per wave: 12 % of the fan-out launch, tools/dw_ablate.sh bit 32).  Tied to the
    // item's index, not to the loop count, so the order -- and in the summing kernel the fp32 accumulation order -- of an item does
    // not depend on how the items are split into segments.
    auto brof = [&](int step, int item) { return (NB > 1 && (item & 1)) ? NB - 1 - step : step; };
    Staged st;
    fetch_item<XLP>(p, FAN ? xb0 : rsrc_of(brof(0, cur)), wi, tid, st);

    // taps -> bf16 tables (loads issued together, then converted)
    {
        constexpr int NW = (CG * 81 + NT - 1) / NT;
        float wv[NB][NW];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float *wsrc = b == 0 ? p.w : p.ws[b > 0 ? b - 1 : 0];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int e = min(tid + i * NT, CG * 81 - 1);
                wv[b][i] = wsrc[(size_t)(e >> 4) * p.C + c0 + (e & 15)];
            }
        }
        bf16_t *wt = (bf16_t *)(smem + WT_OFF);
        for (int e = tid; e < NB * CG * 9 * 16; e += NT) wt[e] = 0;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int e = tid + i * NT;
                if (e < CG * 81) {
                    const int c = e & 15, tap = e >> 4, ky = tap / 9, kx = tap - ky * 9;
                    wt[b * (WTBYTES / 2) + (c * 9 + ky) * 16 + kx] = f32_to_bf16(wv[b][i]);
                }
            }
    }
    write_item(smem, tid, st);
    __syncthreads();

    const int fi = lane & 15, kg = lane >> 4;
    uint4 B[2][NM];
    build_toeplitz(smem, brof(0, cur), wave, fi, kg, B);
    int sl[NM];
    slot_offsets(kg, sl);

    int buf = 0, b = 0;
    f32x4_t acc[2][2][4];
    // the tile to prefetch: sum mode = the next (item, input) pair, every step; fan-out = the next item, found at output 0
    int nxt = cur;
    Item wn = wi;
    bool more = false;
    while (true) {
        // ---- 1. what comes next -----------------------------------------------------------------------------------------------
        int nb = b + 1;
        if (FAN ? b == 0 : nb == NB) {
            nxt = cur + 1;
            if (nxt < iend) wn = decode_item(p, nxt);
            while (nxt < iend && (wn.RV <= 0 || wn.CV <= 0)) { ++nxt; if (nxt < iend) wn = decode_item(p, nxt); }
            more = nxt < iend;
        } else if (!FAN) {
            nxt = cur;
            wn = wi;
            more = true;
        }
        if (nb == NB) nb = 0;
        const bool fetch = more && (!FAN || b == 0);
        const int br_now = brof(b, cur), br_next = brof(nb, nb == 0 ? nxt : cur);   // branches of this step and of the next one
        const __amdgpu_buffer_rsrc_t xn = FAN ? xb0 : rsrc_of(br_next);

        // ---- 2. MFMA ---------------------------------------------------------------------------------------------------------
        char *X = smem + buf * XBYTES;
        const int RV = wi.RV, CV = wi.CV;
        const int nmt = RV > 16 ? 2 : 1, njt = (CV + 15) >> 4;
        const int m1 = RV - 16;                                  // second row tile overlaps the first
        const int jlast = max(((CV + 7) & ~7) - 16, 0);          // last column tile start (multiple of 8: 16-B reads)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const char *xc = X + (wave * 2 + cc) * CSTR;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    if (NB == 1 || FAN || b == 0) acc[cc][mt][jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    // the next tile's loads go out a unit at a time between the MFMA tiles, so they trickle through the
                    // memory pipeline under the MFMAs instead of stalling the wave's issue in one burst
                    const int step = cc * 8 + mt * 4 + jt;
                    if (fetch) {
                        if (step == 0) fetch_unit<XLP>(0, p, xn, wn, tid, st);
                        if (step == 3) fetch_unit<XLP>(1, p, xn, wn, tid, st);
                        if (step == 6) fetch_unit<XLP>(2, p, xn, wn, tid, st);
                        if (step == 9) fetch_unit<XLP>(3, p, xn, wn, tid, st);
                        if (step == 12) fetch_unit<XLP>(4, p, xn, wn, tid, st);
                    }
                    if (mt < nmt && jt < njt && !(p.dbg & 1)) {
                        const char *xa = xc + ((mt ? m1 : 0) + fi) * RSTR + min(jt * 16, jlast) * 2;
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            const uint4 a = *(const uint4 *)(xa + sl[m]);
                            Mma<bf16_t>::run(a, B[cc][m], acc[cc][mt][jt]);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // one tile's 7 fragments in flight at a time (register budget)
                    }
                }
            }
        }
        const bool last = b == NB - 1;
        if (NB == 1 || FAN || last) {
            // every wave is done reading X (plain / sum: the output staging overwrites it) or is done reading the staging
            // of the previous output (fan-out: the staging lives in the idle second buffer, X stays for the next output)
            if (!(p.dbg & 64)) __syncthreads();   // (64: timing ablation without the workgroup barriers of the item loop: races, wrong results)

            // ---- 3. accumulators -> [pixel][16 ch] bf16 -> NHWC --------------------------------------------------------------
            // staging rows are OPX pixels wide so the last (overlapping) column tile can be written whole; the 4-B channel
            // pair of wave w goes to slot w ^ (col & 7) of the pixel's 32 B (spreads the 16 lanes of a tile row over banks)
            char *S = FAN ? smem + (buf ^ 1) * XBYTES : X;
            if (!(p.dbg & 2)) {
                char *ob = S + ((kg * 4) * OPX + fi) * OSTR;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        if (mt < nmt && jt < njt) {
                            const int cb = min(jt * 16, jlast);
                            char *o = ob + ((mt ? m1 : 0) * OPX + cb) * OSTR + ((wave ^ ((cb + fi) & 7)) << 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                *(uint32_t *)(o + r * OPX * OSTR) = pack_bf16x2(acc[0][mt][jt][r], acc[1][mt][jt][r]);
                        }
                    }
                }
            }
            if (!(p.dbg & 64)) __syncthreads();
            if (!(p.dbg & 2)) {
                // thread = (8-channel half, column, row mod 4): no divisions, 16-B loads/stores
                const int h = tid & 1, col = (tid >> 1) & 63, rq = tid >> 7;
                if constexpr (YLP) {
                    // lattice-planar output: the tile is a run of 32-B cells per lattice row -- a wave stores 1 KiB in one piece.
                    // Cells of the class's padded extent whose pixel is outside the image (at most one row / column) get zeros.
                    const int RVp = min(TLY, p.lp.Ly - wi.ty * TLY), CVp = min(TLX, p.lp.Lx - wi.tx * TLX);
                    if (col < CVp) {
                        const char *osrc = S + col * OSTR + ((h ^ ((col >> 2) & 1)) << 4);
                        const bool s1 = col & 1, s2 = col & 2;
                        // (buffer stores: the plane's base in SGPRs, one 32-bit cell offset per lane)
                        const __amdgpu_buffer_rsrc_t yr = lattice_rsrc(NB > 2 && br_now == 2 ? p.ys[1] : (br_now == 1 ? p.ys[0] : p.y), p.lp, n, cgi);
                        const uint32_t ocol = (uint32_t)(((wi.ry * d + wi.rx) * p.lp.Ly + wi.ty * TLY) * p.lp.Lx + wi.tx * TLX + col) * 32u + (uint32_t)h * 16u;
                        const uint32_t rstep = (uint32_t)p.lp.Lx * 32u;
#pragma unroll
                        for (int k = 0; k < (TLY + 3) / 4; ++k) {
                            const int row = rq + 4 * k;
                            if (row < RVp) {
                                const uint4 o = *(const uint4 *)(osrc + min(row, RV - 1) * OPX * OSTR);
                                const uint32_t a0 = s1 ? o.y : o.x, a1 = s1 ? o.x : o.y, a2 = s1 ? o.w : o.z, a3 = s1 ? o.z : o.w;
                                const bool in = row < RV && col < CV;
                                const u32x4_t v = in ? (u32x4_t){s2 ? a2 : a0, s2 ? a3 : a1, s2 ? a0 : a2, s2 ? a1 : a3} : (u32x4_t){0u, 0u, 0u, 0u};
                                if (!(p.dbg & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, yr, ocol + (uint32_t)row * rstep, 0, 0);
                            }
                        }
                    }
                } else
                if (col < CV) {
                    const int xx = wi.rx + d * (wi.tx * TLX + col);
                    const char *osrc = S + col * OSTR + ((h ^ ((col >> 2) & 1)) << 4);
                    const bool s1 = col & 1, s2 = col & 2;
                    bf16_t *yb = FAN && NB > 2 && br_now == 2 ? p.ys[1] : (FAN && br_now == 1 ? p.ys[0] : p.y);
                    bf16_t *ycol = yb + ((size_t)n * p.H * p.W + xx) * p.ldy + c0 + h * 8;
#pragma unroll
                    for (int k = 0; k < (TLY + 3) / 4; ++k) {
                        const int row = rq + 4 * k;
                        if (row < RV) {
                            // un-swizzle: this half's four channel pairs sit in half h ^ bit2(col), permuted by col & 3
                            const uint4 o = *(const uint4 *)(osrc + row * OPX * OSTR);
                            const uint32_t a0 = s1 ? o.y : o.x, a1 = s1 ? o.x : o.y, a2 = s1 ? o.w : o.z, a3 = s1 ? o.z : o.w;
                            const int yy = wi.ry + d * (wi.ty * TLY + row);
                            if (!(p.dbg & 16)) *(uint4 *)(ycol + (size_t)yy * p.W * p.ldy) = make_uint4(s2 ? a2 : a0, s2 ? a3 : a1, s2 ? a0 : a2, s2 ? a1 : a3);
                        }
                    }
                }
            }
        }
        if (FAN && !last) {   // same tile, next output: only the operands change
            if (!(p.dbg & 32)) build_toeplitz(smem, br_next, wave, fi, kg, B);   // (32: timing ablation, stale operands)
            b = nb;
            continue;
        }
        if (!more) break;

        // ---- 4. prefetched registers -> the other X buffer ---------------------------------------------------------------------
        // (plain / sum: last read two steps ago, behind that step's closing barrier; fan-out: it held the output staging the
        // waves have just been reading)
        if (FAN && !(p.dbg & 64)) __syncthreads();
        buf ^= 1;
        if (!(p.dbg & 4)) write_item(smem + buf * XBYTES, tid, st);
        if (NB > 1 && br_next != br_now && !(p.dbg & 32)) build_toeplitz(smem, br_next, wave, fi, kg, B);   // (equal across an item boundary)
        if (!(p.dbg & 64)) __syncthreads();
        cur = nxt;
        wi = wn;
        b = nb;
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
// dW[ky][kx] = sum_{ly,lx} g[ly][lx] * x[ly + ky - 4][lx + kx - 4] on every residue lattice.  One MFMA per (staged x
// row R, block of 32 lattice columns, channel):
//   A[i = kx][k] = x[R][cb + k + kx]      (a Hankel band of the x row: 8 consecutive elements from a 2-B-granular start,
//                                          fetched as 5 aligned dwords and funnel-shifted by 0 or 16 bits per lane)
//   B[k][j]      = g[R - 8 + j][cb + k]   (16 lattice rows of g, plain 16-B fragment reads; rows outside the tile -> 0)
//   D[kx][j]    += ...                    = dW[ky = 8 - j][kx] for j = 0..8, summed over every R, column block, tile,
// residue class and image the block sees: one 16x16 accumulator per channel for the whole block (31 % of its MACs are
// useful).  x and g tiles are staged like the forward kernel's X (same transposed layout, g without halo); the next
// item's loads are spread over the row loop and land in the same LDS buffers after the MFMAs.
constexpr int GBYTES = ((TLY + 1) * RSTR + 64 + 15) & ~15;   // + one all-zero row: what out-of-tile g rows read
constexpr int WG_LDS = XBYTES + GBYTES;
constexpr int NTW = 1024;                   // 16 waves, one channel each: four waves per SIMD hide the LDS / VALU chain
constexpr int GUNITS = TLY * 32 * 2;
constexpr int NIXW = (ITEMS + NTW - 1) / NTW, NIGW = (GUNITS + NTW - 1) / NTW;
static_assert(NIXW == 3 && NIGW == 2, "interleaved fetch below is written for 3 x units + 2 g units per thread");

struct DwWgMfmaParams {
    const bf16_t *x, *g;
    float *part;         // [slab][81][C]
    int N, H, W, C, dil, ldx, ldg;
    int nty, ntx, ncg;
    int nitems, nseg;
};

template <int NU> struct StagedW {
    uint4 a[NU], b[NU];
    uint32_t ok[NU];
};

// stage-in unit (row r, column pair lp, 8-channel half h) of the x tile (HALO = 4) or the g tile (HALO = 0: cells past
// this tile's valid outputs belong to another tile or to no pixel and are zeroed)
template <int HALO, int NU>
__device__ __forceinline__ void fetch_unit_w(int it, __amdgpu_buffer_rsrc_t base, int ld, int H, int W, int d, const Item &w, int tid,
                                             StagedW<NU> &s)
{
    constexpr int UNITS = HALO ? ITEMS : GUNITS;
    const int unit = tid + it * NTW;
    const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
    const int ly = w.ty * TLY + r - HALO, lx = w.tx * TLX + 2 * lp - HALO;
    const int yy = w.ry + d * ly, xa = w.rx + d * lx, xb2 = xa + d;
    bool rok = ly >= 0 && yy < H && unit < UNITS, aok = lx >= 0 && xa < W, bok = lx + 1 >= 0 && xb2 < W;
    if (HALO) {
        rok = rok && lp < 30;
    } else {
        rok = rok && r < w.RV;
        aok = aok && 2 * lp < w.CV;
        bok = bok && 2 * lp + 1 < w.CV;
    }
    const uint32_t pb = (uint32_t)ld * 2u;
    const uint32_t oa = (uint32_t)(yy * W + xa) * pb + (uint32_t)h * 16u;
    s.a[it] = bload16(base, rok && aok ? oa : BUF_OOB);
    s.b[it] = bload16(base, rok && bok ? oa + (uint32_t)d * pb : BUF_OOB);
    s.ok[it] = (rok && aok ? 0x0000ffffu : 0u) | (rok && bok ? 0xffff0000u : 0u);
}

template <int HALO, int NU> __device__ __forceinline__ void write_item_w(char *X, int tid, const StagedW<NU> &s)
{
    constexpr int UNITS = HALO ? ITEMS : GUNITS, NR = HALO ? RY : TLY;
    for (int e = tid; e < NR * 8; e += NTW) *(uint32_t *)(X + (e >> 3) * RSTR + CG * CSTR + (e & 7) * 4) = 0u;   // row pads
    if (HALO) {
        for (int e = tid; e < 16; e += NTW) *(uint32_t *)(X + RY * RSTR + e * 4) = 0u;                            // tail
    } else {
        for (int e = tid; e < (RSTR + 64) / 4; e += NTW) *(uint32_t *)(X + TLY * RSTR + e * 4) = 0u;              // zero row + tail
    }
#pragma unroll
    for (int it = 0; it < NU; ++it) {
        const int unit = tid + it * NTW;
        if (unit < UNITS) {
            const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
            char *dst = X + r * RSTR + (h * 8) * CSTR + lp * 4;
            const uint32_t a[4] = {s.a[it].x, s.a[it].y, s.a[it].z, s.a[it].w};
            const uint32_t b[4] = {s.b[it].x, s.b[it].y, s.b[it].z, s.b[it].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *(uint32_t *)(dst + (2 * m) * CSTR) = ((a[m] & 0xffffu) | (b[m] << 16)) & s.ok[it];
                *(uint32_t *)(dst + (2 * m + 1) * CSTR) = ((a[m] >> 16) | (b[m] & 0xffff0000u)) & s.ok[it];
            }
        }
    }
}

__global__ __launch_bounds__(NTW, 1) void dw_mfma_wgrad_kernel(DwWgMfmaParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const Xx = smem, *const Xg = smem + XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // wave = channel of the group
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lin % p.ncg; lin /= p.ncg;
    const int seg = lin % p.nseg;
    const int n = lin / p.nseg;
    const int c0 = cgi * CG;
    const int ibeg = (int)((long long)p.nitems * seg / p.nseg), iend = (int)((long long)p.nitems * (seg + 1) / p.nseg);
    const __amdgpu_buffer_rsrc_t xb = image_rsrc(p.x + (size_t)n * p.H * p.W * p.ldx + c0, p.H, p.W, p.ldx);
    const __amdgpu_buffer_rsrc_t gb = image_rsrc(p.g + (size_t)n * p.H * p.W * p.ldg + c0, p.H, p.W, p.ldg);
    DwMfmaParams px;   // decode_item reads only the geometry
    px.H = p.H; px.W = p.W; px.dil = p.dil; px.ldx = p.ldx; px.ntx = p.ntx;

    const int fi = lane & 15, kg = lane >> 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // even / odd staged x rows (see `rows` below)

    int cur = ibeg;
    Item wi = decode_item(px, cur < iend ? cur : 0);
    while (cur < iend && (wi.RV <= 0 || wi.CV <= 0)) { ++cur; if (cur < iend) wi = decode_item(px, cur); }
    if (cur < iend) {
        StagedW<NIXW> sx;
        StagedW<NIGW> sg;
#pragma unroll
        for (int it = 0; it < NIXW; ++it) fetch_unit_w<4>(it, xb, p.ldx, p.H, p.W, p.dil, wi, tid, sx);
#pragma unroll
        for (int it = 0; it < NIGW; ++it) fetch_unit_w<0>(it, gb, p.ldg, p.H, p.W, p.dil, wi, tid, sg);
        write_item_w<4>(Xx, tid, sx);
        write_item_w<0>(Xg, tid, sg);
        __syncthreads();

        const int kxl = min(fi, 8);                       // A row i = kx (rows 9..15 duplicate row 8, discarded)
        const uint32_t sh = (kxl & 1) ? 16u : 0u;
        const int aoff = (kg * 8 + (kxl & ~1)) * 2;       // dword-aligned start of this lane's Hankel window
        const char *xc = Xx + wave * CSTR + aoff;
        const char *gc = Xg + wave * CSTR + kg * 16;
        while (true) {
            int nxt = cur + 1;
            Item wn = wi;
            if (nxt < iend) wn = decode_item(px, nxt);
            while (nxt < iend && (wn.RV <= 0 || wn.CV <= 0)) { ++nxt; if (nxt < iend) wn = decode_item(px, nxt); }
            const bool more = nxt < iend;

            const int RV = wi.RV;
            // rows in pairs.  Both x rows of a pair multiply the SAME 16 rows of g (R0 - 8 + j): for the odd row R0 + 1 that
            // is tap row ky = 9 - j instead of 8 - j, still all nine of them (j = 1 .. 9), so it accumulates into a second tile
            // that is shifted by one lane when the partial sums are written -- one g fragment read per two MFMAs instead of
            // one each (the kernel is LDS-bound: 5 + 5 + 4 fragment dwords per pair and lane instead of 5 + 4 + 5 + 4).
            // The pair's fragment dwords are read first, then shifted, then the MFMAs issue.
            auto rows = [&](auto ncb_tag, int R0) __attribute__((always_inline)) {
                constexpr int NCB = decltype(ncb_tag)::value;
                uint32_t dd[2][NCB][5];
                uint4 bv[NCB];
                {
                    const int ly = R0 - 8 + fi;
                    const bool ok = ly >= 0 && ly < RV;
                    const char *gr = gc + (ok ? ly : TLY) * RSTR;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) bv[cb] = *(const uint4 *)(gr + cb * 64);
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const char *xr = xc + min(R0 + rr, RY - 1) * RSTR;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const uint32_t *xw = (const uint32_t *)(xr + cb * 64);
#pragma unroll
                        for (int q = 0; q < 5; ++q) dd[rr][cb][q] = xw[q];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const uint32_t *d5 = dd[rr][cb];
                        const uint4 av = make_uint4(__builtin_amdgcn_alignbit(d5[1], d5[0], sh), __builtin_amdgcn_alignbit(d5[2], d5[1], sh),
                                                    __builtin_amdgcn_alignbit(d5[3], d5[2], sh), __builtin_amdgcn_alignbit(d5[4], d5[3], sh));
                        if (rr == 0) Mma<bf16_t>::run(av, bv[cb], acc);
                        else Mma<bf16_t>::run(av, bv[cb], acc1);
                    }
            };
#pragma unroll 1
            for (int R0 = 0; R0 < RV + 8; R0 += 2) {
                if (more && (R0 & 7) == 0) {   // next item's loads, one unit per eight rows
                    switch (R0 >> 3) {
                    case 0: fetch_unit_w<4>(0, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx); break;
                    case 1: fetch_unit_w<4>(1, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx); break;
                    case 2: fetch_unit_w<4>(2, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx); break;
                    case 3: fetch_unit_w<0>(0, gb, p.ldg, p.H, p.W, p.dil, wn, tid, sg); break;
                    default: fetch_unit_w<0>(1, gb, p.ldg, p.H, p.W, p.dil, wn, tid, sg); break;
                    }
                }
                if (wi.CV > 32) rows(std::integral_constant<int, 2>{}, R0);
                else rows(std::integral_constant<int, 1>{}, R0);
            }
            if (!more) break;
            // short tiles: the row loop did not reach every fetch slot (slot k sits at row 8k)
#pragma unroll
            for (int it = 0; it < NIXW; ++it)
                if (it * 8 >= wi.RV + 8) fetch_unit_w<4>(it, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx);
#pragma unroll
            for (int it = 0; it < NIGW; ++it)
                if ((NIXW + it) * 8 >= wi.RV + 8) fetch_unit_w<0>(it, gb, p.ldg, p.H, p.W, p.dil, wn, tid, sg);
            __syncthreads();   // every wave is done reading the tiles
            write_item_w<4>(Xx, tid, sx);
            write_item_w<0>(Xg, tid, sg);
            __syncthreads();
            cur = nxt;
            wi = wn;
        }
    }
    // D[kx = 4*kg + r][j = fi] -> part[slab][ky = 8 - fi][kx][channel]; the odd-row tile holds ky = 9 - j, i.e. tap row
    // 8 - fi sits one lane up
    float *part = p.part + (size_t)(n * p.nseg + seg) * 81 * p.C + c0 + wave;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kx = kg * 4 + r;
        const float odd = __shfl_down(acc1[r], 1, 16);
        if (kx < 9 && fi < 9) part[(size_t)((8 - fi) * 9 + kx) * p.C] = acc[r] + odd;
    }
}

// ---- weight gradients of up to three depthwise convs that read ONE tensor (the replaced ASPP branches) -------------------
// dW_b = wgrad(x, g_b), b < NG: the kernel above, restructured around what bounds it.  It is LDS-bound on the Hankel windows
// of x (5 dword reads + 4 funnel shifts per MFMA against one 16-B read for the g operand), and the three branches multiply
// the SAME windows: here a pair of x rows is windowed ONCE and meets the g fragments of all NG branches (per pair of rows and
// 32-column block: 20 window dwords + NG 16-B fragments for 2 NG MFMAs, instead of NG x (20 + 4 dwords) for 2 NG), and the x
// tile is fetched from memory once instead of NG times.  Three g tiles next to the x tile do not fit at the 26-row tile
// (239 KiB), so work items are half-height: 13 lattice rows (x: 21 staged rows, 42.7 KiB; g: 3 x 28.5 KiB; 128 KiB in all);
// the x rows between the two halves of a residue class are staged twice (1.31 x the x reads, against 3 x before).
constexpr int TLY3 = 13, RY3 = TLY3 + 8, MAXG = 3;
constexpr int X3BYTES = (RY3 * RSTR + 64 + 15) & ~15;
constexpr int G3BYTES = ((TLY3 + 1) * RSTR + 64 + 15) & ~15;   // + one all-zero row
constexpr int WG3_LDS = X3BYTES + MAXG * G3BYTES;
constexpr int ITEMS3 = RY3 * 64, GUNITS3 = TLY3 * 64;
constexpr int NIX3 = (ITEMS3 + NTW - 1) / NTW, NIG3 = (GUNITS3 + NTW - 1) / NTW;
static_assert(NIX3 == 2 && NIG3 == 1, "interleaved fetch below is written for 2 x units + 1 g unit per branch and thread");
static_assert(WG3_LDS <= 160 * 1024, "LDS budget");

struct DwWgMultiParams {
    const bf16_t *x;
    const bf16_t *g[MAXG];
    float *part;         // [branch][slab][81][C]
    int N, H, W, C, dil, ldx, ldg;
    int nty, ntx, ncg;
    int nitems, nseg, nslabs;
    LpGeom lp;           // GLP: the gradients g[] are lattice-planar (x stays NHWC)
};

// (round 6: the row tiles of one (class, column tile) are consecutive items -- a half-height item shares 8 staged x rows with its vertical
// neighbour, and fetched back to back the second fetch finds most of them in the XCD's L2 instead of going to HBM again)
__device__ __forceinline__ Item decode_item3(int H, int W, int d, int ntx, int nty, int e)
{
    Item it;
    it.ty = e % nty; e /= nty;
    it.tx = e % ntx; e /= ntx;
    it.rx = e % d;
    it.ry = e / d;
    const int Ly = (H - it.ry + d - 1) / d, Lx = (W - it.rx + d - 1) / d;
    it.RV = min(TLY3, Ly - it.ty * TLY3);
    it.CV = min(TLX, Lx - it.tx * TLX);
    return it;
}

template <int NU> struct Staged3 {
    uint4 a[NU], b[NU];
};

template <int HALO, int NU, bool LPT = false>   // LPT: `base` spans this (image, channel group)'s rows of a lattice-planar tensor (ld = its Lx, lpLy its Ly)
__device__ __forceinline__ void fetch_unit3(int it, __amdgpu_buffer_rsrc_t base, int ld, int H, int W, int d, const Item &w, int tid,
                                            Staged3<NU> &s, int lpLy = 0)
{
    constexpr int UNITS = HALO ? ITEMS3 : GUNITS3;
    const int unit = tid + it * NTW;
    const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
    const int ly = w.ty * TLY3 + r - HALO, lx = w.tx * TLX + 2 * lp - HALO;
    const int yy = w.ry + d * ly, xa = w.rx + d * lx, xb2 = xa + d;
    bool rok = ly >= 0 && yy < H && unit < UNITS, aok = lx >= 0 && xa < W, bok = lx + 1 >= 0 && xb2 < W;
    if (HALO) {
        rok = rok && lp < 30;
    } else {
        rok = rok && r < w.RV;
        aok = aok && 2 * lp < w.CV;
        bok = bok && 2 * lp + 1 < w.CV;
    }
    const uint32_t pb = LPT ? 32u : (uint32_t)ld * 2u;
    const uint32_t oa = (LPT ? (uint32_t)(((w.ry * d + w.rx) * lpLy + ly) * ld + lx) : (uint32_t)(yy * W + xa)) * pb + (uint32_t)h * 16u;
    s.a[it] = bload16(base, rok && aok ? oa : BUF_OOB);
    s.b[it] = bload16(base, rok && bok ? oa + (LPT ? 1u : (uint32_t)d) * pb : BUF_OOB);
    // (no validity mask is kept: an out-of-range buffer load has already returned zeros)
}

template <int HALO, int NU> __device__ __forceinline__ void write_item3(char *X, int tid, const Staged3<NU> &s)
{
    constexpr int UNITS = HALO ? ITEMS3 : GUNITS3, NR = HALO ? RY3 : TLY3;
    for (int e = tid; e < NR * 8; e += NTW) *(uint32_t *)(X + (e >> 3) * RSTR + CG * CSTR + (e & 7) * 4) = 0u;   // row pads
    if (HALO) {
        for (int e = tid; e < 16; e += NTW) *(uint32_t *)(X + RY3 * RSTR + e * 4) = 0u;                           // tail
    } else {
        for (int e = tid; e < (RSTR + 64) / 4; e += NTW) *(uint32_t *)(X + TLY3 * RSTR + e * 4) = 0u;             // zero row + tail
    }
#pragma unroll
    for (int it = 0; it < NU; ++it) {
        const int unit = tid + it * NTW;
        if (unit < UNITS) {
            const int h = unit & 1, lp = (unit >> 1) & 31, r = unit >> 6;
            char *dst = X + r * RSTR + (h * 8) * CSTR + lp * 4;
            const uint32_t a[4] = {s.a[it].x, s.a[it].y, s.a[it].z, s.a[it].w};
            const uint32_t b[4] = {s.b[it].x, s.b[it].y, s.b[it].z, s.b[it].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *(uint32_t *)(dst + (2 * m) * CSTR) = (a[m] & 0xffffu) | (b[m] << 16);
                *(uint32_t *)(dst + (2 * m + 1) * CSTR) = (a[m] >> 16) | (b[m] & 0xffff0000u);
            }
        }
    }
}

template <int NG, bool GLP = false>
__global__ __launch_bounds__(NTW, 1) void dw_mfma_wgrad_multi_kernel(DwWgMultiParams p)
{
    static_assert(NG >= 2 && NG <= MAXG, "two or three branches");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const Xx = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // wave = channel of the group
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lin % p.ncg; lin /= p.ncg;
    const int seg = lin % p.nseg;
    const int n = lin / p.nseg;
    const int c0 = cgi * CG;
    const int ibeg = (int)((long long)p.nitems * seg / p.nseg), iend = (int)((long long)p.nitems * (seg + 1) / p.nseg);
    const __amdgpu_buffer_rsrc_t xb = image_rsrc(p.x + (size_t)n * p.H * p.W * p.ldx + c0, p.H, p.W, p.ldx);
    const __amdgpu_buffer_rsrc_t gb0 = GLP ? lattice_rsrc(p.g[0], p.lp, n, cgi) : image_rsrc(p.g[0] + (size_t)n * p.H * p.W * p.ldg + c0, p.H, p.W, p.ldg);
    const __amdgpu_buffer_rsrc_t gb1 = GLP ? lattice_rsrc(p.g[1], p.lp, n, cgi) : image_rsrc(p.g[1] + (size_t)n * p.H * p.W * p.ldg + c0, p.H, p.W, p.ldg);
    const __amdgpu_buffer_rsrc_t gb2 = GLP ? lattice_rsrc(p.g[NG > 2 ? 2 : 1], p.lp, n, cgi)
                                           : image_rsrc(p.g[NG > 2 ? 2 : 1] + (size_t)n * p.H * p.W * p.ldg + c0, p.H, p.W, p.ldg);
    const int gld = GLP ? p.lp.Lx : p.ldg, gly = GLP ? p.lp.Ly : 0;

    const int fi = lane & 15, kg = lane >> 4;
    f32x4_t acc[NG][2];   // per branch: even / odd staged x rows (the odd-row tile is shifted by one lane when written)
#pragma unroll
    for (int b = 0; b < NG; ++b) acc[b][0] = acc[b][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    int cur = ibeg;
    Item wi = decode_item3(p.H, p.W, p.dil, p.ntx, p.nty, cur < iend ? cur : 0);
    while (cur < iend && (wi.RV <= 0 || wi.CV <= 0)) { ++cur; if (cur < iend) wi = decode_item3(p.H, p.W, p.dil, p.ntx, p.nty, cur); }
    if (cur < iend) {
        Staged3<NIX3> sx;
        Staged3<NIG3> sg0, sg1, sg2;
        fetch_unit3<4>(0, xb, p.ldx, p.H, p.W, p.dil, wi, tid, sx);
        fetch_unit3<4>(1, xb, p.ldx, p.H, p.W, p.dil, wi, tid, sx);
        fetch_unit3<0, NIG3, GLP>(0, gb0, gld, p.H, p.W, p.dil, wi, tid, sg0, gly);
        fetch_unit3<0, NIG3, GLP>(0, gb1, gld, p.H, p.W, p.dil, wi, tid, sg1, gly);
        if (NG > 2) fetch_unit3<0, NIG3, GLP>(0, gb2, gld, p.H, p.W, p.dil, wi, tid, sg2, gly);
        auto publish = [&]() __attribute__((always_inline)) {
            write_item3<4>(Xx, tid, sx);
            write_item3<0>(smem + X3BYTES, tid, sg0);
            write_item3<0>(smem + X3BYTES + G3BYTES, tid, sg1);
            if (NG > 2) write_item3<0>(smem + X3BYTES + 2 * G3BYTES, tid, sg2);
        };
        publish();
        __syncthreads();

        const int kxl = min(fi, 8);                       // A row i = kx (rows 9..15 duplicate row 8, discarded)
        const uint32_t sh = (kxl & 1) ? 16u : 0u;
        const int aoff = (kg * 8 + (kxl & ~1)) * 2;       // dword-aligned start of this lane's Hankel window
        const char *xc = Xx + wave * CSTR + aoff;
        const char *gc = smem + X3BYTES + wave * CSTR + kg * 16;
        while (true) {
            int nxt = cur + 1;
            Item wn = wi;
            if (nxt < iend) wn = decode_item3(p.H, p.W, p.dil, p.ntx, p.nty, nxt);
            while (nxt < iend && (wn.RV <= 0 || wn.CV <= 0)) { ++nxt; if (nxt < iend) wn = decode_item3(p.H, p.W, p.dil, p.ntx, p.nty, nxt); }
            const bool more = nxt < iend;

            const int RV = wi.RV;
            const int ncb = wi.CV > 32 ? 2 : 1;
#pragma unroll 1
            for (int R0 = 0; R0 < RV + 8; R0 += 2) {
                if (more && (R0 & 3) == 0) {   // next item's loads, one unit per four rows
                    switch (R0 >> 2) {
                    case 0: fetch_unit3<4>(0, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx); break;
                    case 1: fetch_unit3<4>(1, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx); break;
                    case 2: fetch_unit3<0, NIG3, GLP>(0, gb0, gld, p.H, p.W, p.dil, wn, tid, sg0, gly); break;
                    case 3: fetch_unit3<0, NIG3, GLP>(0, gb1, gld, p.H, p.W, p.dil, wn, tid, sg1, gly); break;
                    case 4: if (NG > 2) fetch_unit3<0, NIG3, GLP>(0, gb2, gld, p.H, p.W, p.dil, wn, tid, sg2, gly); break;
                    default: break;
                    }
                }
                // both x rows of the pair multiply the SAME 16 rows of g (R0 - 8 + j); see dw_mfma_wgrad_kernel
                const int ly = R0 - 8 + fi;
                const int grow = (ly >= 0 && ly < RV ? ly : TLY3) * RSTR;
                const char *x0 = xc + min(R0, RY3 - 1) * RSTR, *x1 = xc + min(R0 + 1, RY3 - 1) * RSTR;
                for (int cb = 0; cb < ncb; ++cb) {
                    const uint32_t *xw0 = (const uint32_t *)(x0 + cb * 64), *xw1 = (const uint32_t *)(x1 + cb * 64);
                    uint32_t d0[5], d1[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) { d0[q] = xw0[q]; d1[q] = xw1[q]; }
                    uint4 bv = *(const uint4 *)(gc + grow + cb * 64);
                    const uint4 a0 = make_uint4(__builtin_amdgcn_alignbit(d0[1], d0[0], sh), __builtin_amdgcn_alignbit(d0[2], d0[1], sh),
                                                __builtin_amdgcn_alignbit(d0[3], d0[2], sh), __builtin_amdgcn_alignbit(d0[4], d0[3], sh));
                    const uint4 a1 = make_uint4(__builtin_amdgcn_alignbit(d1[1], d1[0], sh), __builtin_amdgcn_alignbit(d1[2], d1[1], sh),
                                                __builtin_amdgcn_alignbit(d1[3], d1[2], sh), __builtin_amdgcn_alignbit(d1[4], d1[3], sh));
#pragma unroll
                    for (int b = 0; b < NG; ++b) {   // one g fragment in flight ahead of the pair of MFMAs that uses it
                        const uint4 bc = bv;
                        if (b + 1 < NG) bv = *(const uint4 *)(gc + (b + 1) * G3BYTES + grow + cb * 64);
                        Mma<bf16_t>::run(a0, bc, acc[b][0]);
                        Mma<bf16_t>::run(a1, bc, acc[b][1]);
                    }
                }
            }
            if (!more) break;
            // short tiles: the row loop did not reach every fetch slot (slot k sits at row 4k)
            if (0 >= RV + 8) fetch_unit3<4>(0, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx);
            if (4 >= RV + 8) fetch_unit3<4>(1, xb, p.ldx, p.H, p.W, p.dil, wn, tid, sx);
            if (8 >= RV + 8) fetch_unit3<0, NIG3, GLP>(0, gb0, gld, p.H, p.W, p.dil, wn, tid, sg0, gly);
            if (12 >= RV + 8) fetch_unit3<0, NIG3, GLP>(0, gb1, gld, p.H, p.W, p.dil, wn, tid, sg1, gly);
            if (NG > 2 && 16 >= RV + 8) fetch_unit3<0, NIG3, GLP>(0, gb2, gld, p.H, p.W, p.dil, wn, tid, sg2, gly);
            __syncthreads();   // every wave is done reading the tiles
            publish();
            __syncthreads();
            cur = nxt;
            wi = wn;
        }
    }
    // D[kx = 4*kg + r][j = fi] -> part[b][slab][ky = 8 - fi][kx][channel]; the odd-row tile holds ky = 9 - j
#pragma unroll
    for (int b = 0; b < NG; ++b) {
        float *part = p.part + ((size_t)b * p.nslabs + (size_t)(n * p.nseg + seg)) * 81 * p.C + c0 + wave;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kx = kg * 4 + r;
            const float odd = __shfl_down(acc[b][1][r], 1, 16);
            if (kx < 9 && fi < 9) part[(size_t)((8 - fi) * 9 + kx) * p.C] = acc[b][0][r] + odd;
        }
    }
}

static void dw_mfma_split3(int N, int C, int H, int W, int dil, int *nty, int *ntx, int *nitems, int *nseg)
{
    const int LH = (H + dil - 1) / dil, LW = (W + dil - 1) / dil;
    *nty = (LH + TLY3 - 1) / TLY3;
    *ntx = (LW + TLX - 1) / TLX;
    const long long ni = (long long)*nty * *ntx * dil * dil;
    *nitems = ni > (1 << 24) ? 0 : (int)ni;
    const long long groups = (long long)N * (C / CG);
    long long s = (512 + groups - 1) / groups;
    if (s > ni / 3) s = ni / 3;
    if (s < 1) s = 1;
    *nseg = (int)s;
}

static void dw_mfma_split(int N, int C, int H, int W, int dil, int *nty, int *ntx, int *nitems, int *nseg)
{
    const int LH = (H + dil - 1) / dil, LW = (W + dil - 1) / dil;
    *nty = (LH + TLY - 1) / TLY;
    *ntx = (LW + TLX - 1) / TLX;
    const long long ni = (long long)*nty * *ntx * dil * dil;
    *nitems = ni > (1 << 24) ? 0 : (int)ni;
    // one block per CU at a time (LDS): aim at two rounds of blocks over the 256 CUs, but keep >= 3 items per block so
    // the prefetch has something to overlap with and the per-block set-up is amortised
    const long long groups = (long long)N * (C / CG);
    long long s = (512 + groups - 1) / groups;
    if (s > ni / 3) s = ni / 3;
    if (s < 1) s = 1;
    *nseg = (int)s;
}

static LpGeom lp_geom(const kd_dw_desc *d)
{
    LpGeom g;
    g.Ly = (d->H + d->dil - 1) / d->dil;
    g.Lx = (d->W + d->dil - 1) / d->dil;
    g.rpi = d->dil * d->dil * g.Ly * g.Lx;
    g.plane = (long long)kd_internal_lattice_rows(d->N, d->H, d->W, d->dil) * CG;
    return g;
}

}  // namespace

// Rows per plane of the lattice-planar layout (a multiple of 256: the 1x1 convs' M tile), 0 if the shape has none.
long long kd_internal_lattice_rows(int N, int H, int W, int dil)
{
    if (N < 1 || H < 1 || W < 1 || dil < 1) return 0;
    const long long Ly = (H + dil - 1) / dil, Lx = (W + dil - 1) / dil;
    const long long rows = (long long)N * dil * dil * Ly * Lx;
    return (rows + 255) / 256 * 256;
}

// Can the fan-out / summing / multi-gradient launches of `nb` branches run on lattice-planar intermediates?  (The matrix-core
// kernels' own conditions, plus: no work item of the fan-out may be empty while its padded cells exist -- they are zeroed by the
// item that owns them -- and a plane's rows must stay inside 32-bit buffer offsets.)
int kd_internal_dw_lattice_ok(const kd_dw_desc *d, int nb)
{
    if (nb < 2 || nb > MAXB || d->dtype != KD_BF16 || d->k != 9 || d->C % CG != 0 || d->ldx % 8 != 0) return 0;
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("KDCC_DW_MFMA"), *l = getenv("KDCC_DW_LATTICE");   // A/B hooks: 0 = NHWC intermediates
        enabled = !(e && e[0] == '0') && !(l && l[0] == '0');
    }
    if (!enabled) return 0;
    const LpGeom g = lp_geom(d);
    if ((long long)g.rpi * CG * 2 >= (long long)BUF_OOB || g.plane * (d->C / CG) > 0x7fffffffLL) return 0;
    if ((long long)d->H * d->W * d->ldx * 2 >= (long long)BUF_OOB) return 0;
    const int ry_last = g.Ly - (g.Ly - 1) / TLY * TLY, rx_last = g.Lx - (g.Lx - 1) / TLX * TLX;   // padded extent of the last tile row / column
    const bool short_y = (d->H % d->dil) != 0, short_x = (d->W % d->dil) != 0;                     // some classes are one row / column shorter
    if ((short_y && ry_last < 2) || (short_x && rx_last < 2)) return 0;
    return 1;
}

// fan == 0: ys[0] = sum_{b < nb} dwconv(xs[b], ws[b]);  fan != 0: ys[b] = dwconv(xs[0], ws[b]) for b < nb -- on the matrix
// cores.  Returns 1 if the MFMA path took the call, 0 if the shape is not eligible (caller falls back to the register kernel,
// one launch per term), < 0 on a launch error.
// lp != 0 (nb >= 2, kd_internal_dw_lattice_ok): the nb-side tensors -- the fan-out's outputs, the sum's inputs -- are lattice-planar.
int kd_internal_dw_mfma_fwd_n(const kd_dw_desc *d, int nb, int fan, const void *const *xs, const float *const *ws, void *const *ys,
                              const float *bias, const kd_dw_epilogue *ep, hipStream_t s, int lp)
{
    if (nb < 1 || nb > MAXB) return 0;
    if (lp && !kd_internal_dw_lattice_ok(d, nb)) return 0;
    if (nb == 1) fan = 0;
    if (d->dtype != KD_BF16 || d->k != 9 || d->C % CG != 0 || d->ldx % 8 != 0 || d->ldy % 8 != 0) return 0;
    for (int b = 0; b < nb; ++b) {
        if (!ws[b]) return 0;
        if (!(fan ? ys[b] && kd_aligned16(ys[b]) : xs[b] && kd_aligned16(xs[b]))) return 0;
    }
    if (!xs[0] || !ys[0] || !kd_aligned16(xs[0]) || !kd_aligned16(ys[0])) return 0;
    if ((long long)d->H * d->W * d->ldx * 2 >= (long long)BUF_OOB) return 0;   // buffer-load offsets are 32-bit per image
    // Calls with a bias or an epilogue stay on the register kernel: their extra operands are read per pixel in 32-B
    // (16-channel) pieces here, which the memory system serves at about a third of the rate of the register kernel's
    // 128-B-per-pixel rows (measured: 1.22 vs 0.99 ms at 4096 channels, mask + residual, 2 images).
    if (bias || (ep && (ep->res_pre || ep->mask || ep->res_post))) return 0;
    if (fan && nb >= 2 && !lp) {     // round 6: the fan-out on the lone-wave kernel (dwconv_lw.hip) where it applies
        const int took = kd_internal_dw_lw_fanout(d, nb, xs[0], ws, ys, s);
        if (took != 0) return took;
    }
    DwMfmaParams p;
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("KDCC_DW_MFMA");   // A/B hook: 0 = always use the register kernel
        enabled = !(e && e[0] == '0');
    }
    if (!enabled) return 0;
    p.x = (const bf16_t *)xs[0]; p.w = ws[0]; p.y = (bf16_t *)ys[0];
    for (int b = 1; b < MAXB; ++b) {
        p.xs[b - 1] = (const bf16_t *)(!fan && b < nb ? xs[b] : xs[0]);
        p.ys[b - 1] = (bf16_t *)(fan && b < nb ? ys[b] : ys[0]);
        p.ws[b - 1] = b < nb ? ws[b] : ws[0];
    }
    { static int dbg = -1; if (dbg < 0) dbg = KD_TUNING_ENV_INT("KDCC_DW_DBG"); p.dbg = dbg; }   // phase ablations: tuning build only
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.dil = d->dil; p.ldx = d->ldx; p.ldy = d->ldy;
    p.lp = lp_geom(d);
    // (a second, independent workgroup per CU on half-height items was built and measured slower -- the Toeplitz rebuilds multiply:
    // profiles/r05_dw_anatomy.md, commit dc356ad)
    dw_mfma_split(d->N, d->C, d->H, d->W, d->dil, &p.nty, &p.ntx, &p.nitems, &p.nseg);
    p.ncg = d->C / CG;
    if (p.nitems <= 0) return 0;
    const long long blocks = (long long)d->N * p.ncg * p.nseg;
    if (blocks > 0x7fffffffLL || (long long)d->N * d->H * d->W > 0x7fffffffLL) return 0;
    const int lds = LDS_BYTES + (nb - 1) * WTBYTES;
    typedef void (*kern_t)(DwMfmaParams);
    const kern_t fn = nb == 1 ? dw_mfma_fwd_kernel<1, false>
                    : lp ? (nb == 2 ? (fan ? dw_mfma_fwd_kernel<2, true, true> : dw_mfma_fwd_kernel<2, false, true>)
                                    : (fan ? dw_mfma_fwd_kernel<3, true, true> : dw_mfma_fwd_kernel<3, false, true>))
                    : nb == 2 ? (fan ? dw_mfma_fwd_kernel<2, true> : dw_mfma_fwd_kernel<2, false>)
                              : (fan ? dw_mfma_fwd_kernel<3, true> : dw_mfma_fwd_kernel<3, false>);
    static bool attr_set[2][2][MAXB + 1] = {};
    if (!attr_set[lp ? 1 : 0][fan ? 1 : 0][nb]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            kd_set_error("kd_dwconv_fwd: cannot reserve %d B of LDS", lds);
            return KD_ERR_HIP;
        }
        attr_set[lp ? 1 : 0][fan ? 1 : 0][nb] = true;
    }
    KD_NOTE_KERNEL(nb == 1 ? "dw_mfma_fwd_kernel<1,false>"
                   : lp ? (nb == 2 ? (fan ? "dw_mfma_fwd_kernel<2,true,lattice>" : "dw_mfma_fwd_kernel<2,false,lattice>")
                                   : (fan ? "dw_mfma_fwd_kernel<3,true,lattice>" : "dw_mfma_fwd_kernel<3,false,lattice>"))
                   : nb == 2 ? (fan ? "dw_mfma_fwd_kernel<2,true>" : "dw_mfma_fwd_kernel<2,false>")
                             : (fan ? "dw_mfma_fwd_kernel<3,true>" : "dw_mfma_fwd_kernel<3,false>"));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(NT), lds, s, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        kd_set_error("kd_dwconv_fwd(mfma): launch failed: %s", hipGetErrorString(err));
        return KD_ERR_HIP;
    }
    return 1;
}

int kd_internal_dw_mfma_fwd(const kd_dw_desc *d, const void *x, const float *w_taps, const float *bias,
                            const kd_dw_epilogue *ep, void *y, hipStream_t s)
{
    return kd_internal_dw_mfma_fwd_n(d, 1, 0, &x, &w_taps, &y, bias, ep, s, 0);
}

static bool dw_mfma_wgrad_eligible(const kd_dw_desc *d, const void *x, const void *dy, int ld_dy)
{
    if (d->dtype != KD_BF16 || d->k != 9 || d->C % CG != 0 || d->ldx % 8 != 0 || ld_dy % 8 != 0) return false;
    if ((x && !kd_aligned16(x)) || (dy && !kd_aligned16(dy))) return false;
    if ((long long)d->N * d->H * d->W > 0x7fffffffLL) return false;
    if ((long long)d->H * d->W * (d->ldx > ld_dy ? d->ldx : ld_dy) * 2 >= (long long)BUF_OOB) return false;   // 32-bit buffer offsets
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("KDCC_DW_MFMA");
        enabled = !(e && e[0] == '0');
    }
    return enabled != 0;
}

// Slabs of [81][C] fp32 partial sums the MFMA weight-gradient path writes (0 = path not eligible).
int kd_internal_dw_mfma_wgrad_slabs(const kd_dw_desc *d)
{
    if (!dw_mfma_wgrad_eligible(d, nullptr, nullptr, 8)) return 0;
    int nty, ntx, nitems, nseg;
    dw_mfma_split(d->N, d->C, d->H, d->W, d->dil, &nty, &ntx, &nitems, &nseg);
    return nitems > 0 ? d->N * nseg : 0;
}

// 1 = partial sums written to `part` (caller reduces the slabs), 0 = not eligible, < 0 = error.
int kd_internal_dw_mfma_wgrad(const kd_dw_desc *d, const void *x, const void *dy, int ld_dy, float *part, hipStream_t s)
{
    if (!dw_mfma_wgrad_eligible(d, x, dy, ld_dy)) return 0;
    DwWgMfmaParams p;
    p.x = (const bf16_t *)x; p.g = (const bf16_t *)dy; p.part = part;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.dil = d->dil; p.ldx = d->ldx; p.ldg = ld_dy;
    dw_mfma_split(d->N, d->C, d->H, d->W, d->dil, &p.nty, &p.ntx, &p.nitems, &p.nseg);
    p.ncg = d->C / CG;
    if (p.nitems <= 0) return 0;
    const long long blocks = (long long)d->N * p.ncg * p.nseg;
    if (blocks > 0x7fffffffLL) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)dw_mfma_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) !=
            hipSuccess) {
            kd_set_error("kd_dwconv_wgrad: cannot reserve %d B of LDS", WG_LDS);
            return KD_ERR_HIP;
        }
        attr_set = true;
    }
    KD_NOTE_KERNEL("dw_mfma_wgrad_kernel");
    hipLaunchKernelGGL(dw_mfma_wgrad_kernel, dim3((unsigned)blocks), dim3(NTW), WG_LDS, s, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        kd_set_error("kd_dwconv_wgrad(mfma): launch failed: %s", hipGetErrorString(err));
        return KD_ERR_HIP;
    }
    return 1;
}

// ---- several weight gradients over one input (kd_dwconv_wgrad_multi) -------------------------------------------------------
// Slabs of [81][C] fp32 partial sums PER BRANCH the fused path writes (0 = not eligible: the caller runs one launch per branch).
int kd_internal_dw_mfma_wgrad_multi_slabs(const kd_dw_desc *d, int n)
{
    if (n < 2 || n > MAXG || !dw_mfma_wgrad_eligible(d, nullptr, nullptr, 8)) return 0;
    int nty, ntx, nitems, nseg;
    dw_mfma_split3(d->N, d->C, d->H, d->W, d->dil, &nty, &ntx, &nitems, &nseg);
    return nitems > 0 ? d->N * nseg : 0;
}

// 1 = partial sums of every branch written to `part` ([branch][slab][81][C]), 0 = not eligible, < 0 = error.
// lp != 0: the gradients dys[] are lattice-planar (ld_dy is ignored)
int kd_internal_dw_mfma_wgrad_multi(const kd_dw_desc *d, int n, const void *x, const void *const *dys, int ld_dy, float *part, hipStream_t s, int lp)
{
    if (n < 2 || n > MAXG) return 0;
    if (lp) { if (!kd_internal_dw_lattice_ok(d, n)) return 0; ld_dy = d->C; }
    for (int b = 0; b < n; ++b)
        if (!dw_mfma_wgrad_eligible(d, x, dys[b], ld_dy)) return 0;
    DwWgMultiParams p;
    p.x = (const bf16_t *)x;
    for (int b = 0; b < MAXG; ++b) p.g[b] = (const bf16_t *)dys[b < n ? b : 0];
    p.part = part;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.dil = d->dil; p.ldx = d->ldx; p.ldg = ld_dy;
    p.lp = lp_geom(d);
    dw_mfma_split3(d->N, d->C, d->H, d->W, d->dil, &p.nty, &p.ntx, &p.nitems, &p.nseg);
    p.ncg = d->C / CG;
    p.nslabs = d->N * p.nseg;
    if (p.nitems <= 0) return 0;
    const long long blocks = (long long)d->N * p.ncg * p.nseg;
    if (blocks > 0x7fffffffLL) return 0;
    typedef void (*kern_t)(DwWgMultiParams);
    const kern_t fn = lp ? (n == 2 ? dw_mfma_wgrad_multi_kernel<2, true> : dw_mfma_wgrad_multi_kernel<3, true>)
                         : (n == 2 ? dw_mfma_wgrad_multi_kernel<2> : dw_mfma_wgrad_multi_kernel<3>);
    static bool attr_set[2][MAXG + 1] = {};
    if (!attr_set[lp ? 1 : 0][n]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS) != hipSuccess) {
            kd_set_error("kd_dwconv_wgrad_multi: cannot reserve %d B of LDS", WG3_LDS);
            return KD_ERR_HIP;
        }
        attr_set[lp ? 1 : 0][n] = true;
    }
    KD_NOTE_KERNEL(lp ? (n == 2 ? "dw_mfma_wgrad_multi_kernel<2,lattice>" : "dw_mfma_wgrad_multi_kernel<3,lattice>")
                      : (n == 2 ? "dw_mfma_wgrad_multi_kernel<2>" : "dw_mfma_wgrad_multi_kernel<3>"));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(NTW), WG3_LDS, s, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        kd_set_error("kd_dwconv_wgrad_multi(mfma): launch failed: %s", hipGetErrorString(err));
        return KD_ERR_HIP;
    }
    return 1;
}

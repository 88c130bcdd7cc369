// Depthwise 9x9 dilated convolution on the matrix cores, ONE WAVE PER SIMD (round 6): the fan-out  y_b = dwconv(x, w_b)  of the
// three depthwise convs that read one tensor (the replaced ASPP branches' forward, reference models/deeplabv3/deeplabv3.py:64-75
// over models/students/transform_blocks/depthwise_separable_conv.py:7-13).  (Two branches, the summed input gradient and the
// weight gradients stay on the 8- / 16-wave kernels of dwconv_mfma.hip: profiles/r06_dw_anatomy.md has the arithmetic.)
//
// What profiles/r05_dw_anatomy.md found in dw_mfma_fwd_kernel (dwconv_mfma.hip: 8 lock-step waves, 2 channels each): a
// third of every launch is neither arithmetic nor memory -- the Toeplitz operands of a branch are rebuilt from the tap table
// twice per work item because one branch's operands are all a 256-register wave can hold, and the phases of an item (fill,
// MFMA, output staging, stores) are serial behind eight workgroup barriers.  This kernel is built around the opposite
// choices:
//   * 4 waves (one per SIMD, 512 registers each), 16 channels per workgroup, 4 channels per wave; the Toeplitz operands of
//     EVERY (branch, channel) pair of the wave stay in registers for the whole workgroup (NB x 4 x 7 fragments = 336
//     registers for three branches, arch + accumulation VGPRs), built once per ~50 work items.
//   * K slots instead of tap rows.  An output tile of 16 lattice columns reads 24 input columns per tap row; one MFMA
//     (K = 32) per tap row wastes the last 8.  The contraction index is only a label, so the 9 x 3 (tap row, 8-column group)
//     pairs are dealt into 28 K-groups = 7 MFMAs per tile instead of 9: 22 % fewer MFMAs, 22 % fewer fragment reads and
//     operand registers.  (Lane (i, kg) of MFMA m reads 8 columns of the row and column group its slot names; the B
//     fragment of that lane holds the matching window of that tap row.)
//   * Fan-out: an X fragment read from LDS once meets the operands of all NB branches (NB MFMAs per ds_read_b128 -- the
//     8-wave kernel reads it once per branch and sits on the LDS port during its matrix phases).
//   * Half-height work items (13 lattice rows: 21 staged rows, 43 KiB) so that two X buffers AND two staging buffers fit:
//     the next item's tile is fetched and transposed into the idle buffer, and tile t-1's outputs leave through their own
//     staging buffer, while tile t's MFMAs run -- one barrier per tile phase, nothing serial but the pipeline fill.
// Work item = (residue class, 13 x 52 lattice tile); tile phase = one 13 x 16 column tile of the item, all 16 channels,
// all branches: per wave 4 channels x 7 fragment reads, 4 x 7 x NB MFMAs, the packed results of channel pairs as 4-B
// stores into staging[pixel][16 channels] (48-B pixel stride: conflict-free without a swizzle), whence 16-B NHWC stores.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "igemm_core.h"

namespace {

constexpr int CG = 16;                      // channels per workgroup
constexpr int LNT = 256;                    // threads: 4 waves, one per SIMD
constexpr int TLY = 13, TLX = 52;           // lattice outputs per work item
constexpr int RY = TLY + 8;                 // staged lattice rows
constexpr int NCOL = 64;                    // staged lattice columns per (row, channel): 60 filled, 60..63 kept zero
constexpr int CSTR = NCOL * 2;
constexpr int RSTR = CG * CSTR + 32;        // == 32 (mod 256): fragment reads of a (row i, 8-column group) lane pattern are conflict-free
constexpr int XB = (RY * RSTR + 15) & ~15;  // one X buffer
constexpr int SPX = 48;                     // staging: bytes per pixel (32 used)
constexpr int STILE = 16 * 16 * SPX;        // staging of one (branch, column tile): 16 rows x 16 columns
constexpr int MAXB = 3;
constexpr int SBUF = MAXB * STILE;
constexpr int S_OFF = 2 * XB;
constexpr int LDS_BYTES = S_OFF + 2 * SBUF;
constexpr int WTBYTES = CG * 9 * 16 * 2;    // prologue only: bf16 taps [channel][ky][16], slots 9..15 zero
constexpr int WT_OFF = LDS_BYTES - MAXB * WTBYTES;            // ... at the end of the staging area
constexpr int STASH_OFF = XB;                                 // prologue only: 20 operand fragments per wave (20 KiB each), over X buffer 1 + staging
constexpr int LC_OFF = STASH_OFF + 4 * 20 * 1024;             // prologue only: 16 dwords of per-thread constants
constexpr int UNITS = RY * 60;              // stage-in units per item: (row, column pair, 8-channel half)
constexpr int NIT = (UNITS + LNT - 1) / LNT;
constexpr int NM = 7;                       // MFMAs per (channel, tile, branch)
static_assert(LDS_BYTES <= 160 * 1024 && LC_OFF + LNT * 64 <= WT_OFF, "LDS budget");
static_assert(NIT == 5, "the fill schedule below is written for 5 units per thread");

constexpr uint32_t BUF_OOB = 0x80000000u;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct DwLwParams {
    const bf16_t *x[MAXB];     // the input (x[0]; the other slots repeat it)
    const float *w[MAXB];      // taps [81][C] per branch
    bf16_t *y[MAXB];           // one output per branch
    int N, H, W, C, dil, ldx, ldy;
    long long xplane, yplane;  // 0: NHWC (pixel stride ldx / ldy).  > 0: channel-planar [C/64][N*H*W][64] -- elements between two 64-channel planes, pixel stride 64
    const int *items;          // descriptors of the non-empty work items of one (image, channel group), 16 dwords each
    int nty, ntx, ncg, nitems, nseg;
};

struct Item {
    int ry, rx, ty, tx;
};

__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// K slot q = 4 m + kg of a tile's MFMA m: which (tap row, 8-column group) its eight contraction indices stand for; ky < 0:
// the 28th slot, no products (zero operand).  Slots come in pairs (kg 0/1, kg 2/3: the lanes one ds_read_b128 group serves);
// a pair whose column groups differ in parity reads conflict-free (RSTR == 32 mod 256), so the nine (ky, 0) | (ky, 1) pairs and
// (8, 2) | none come first; the four pairs of the remaining (ky, 2) groups cost one extra LDS cycle per lane group.
__device__ __forceinline__ void slot_of(int q, int &ky, int &cg)
{
    const int pr = q >> 1, e = q & 1;
    if (pr < 9) { ky = pr; cg = e; }
    else if (pr == 9) { ky = e ? -1 : 8; cg = e ? 1 : 2; }
    else { ky = 2 * (pr - 10) + e; cg = 2; }
}

struct Unit {
    uint4 a, b;
};

// global -> registers: unit u = (row r, column pair lp, half h) of the tile: two neighbouring lattice pixels, 8 channels each.
// Thread tid < 252 owns row tid / 12 and the five column pairs lp = (tid % 12) / 2 + 6 j, half h = tid & 1 (threads 252 .. 255
// repeat the work of threads 0 .. 3: same addresses, same values).  Cells outside the image (the stencil's zero padding) are
// fetched with an out-of-range buffer offset: zeros, no request.
__device__ __forceinline__ void fill_coords(int tid, int &r, int &c0, int &h)
{
    const int t = tid >= 252 ? tid - 252 : tid;
    r = t / 12;
    const int cq = t - r * 12;
    c0 = cq & ~1;         // first of the thread's column pairs: columns c0 + 12 j, c0 + 12 j + 1
    h = cq & 1;
}
__device__ __forceinline__ void fetch_unit(int j, const DwLwParams &p, __amdgpu_buffer_rsrc_t xr, const Item &w, int tid, Unit &s)
{
    const int d = p.dil;
    int r, c0, h;
    fill_coords(tid, r, c0, h);
    const int ly = w.ty * TLY + r - 4, lx = w.tx * TLX + c0 + 12 * j - 4;
    const int yy = w.ry + d * ly, xa = w.rx + d * lx, xb2 = xa + d;
    const bool rok = ly >= 0 && yy < p.H;
    const bool aok = rok && lx >= 0 && xa < p.W, bok = rok && lx + 1 >= 0 && xb2 < p.W;
    const uint32_t pb = (uint32_t)p.ldx * 2u;
    const uint32_t oa = (uint32_t)(yy * p.W + xa) * pb + (uint32_t)h * 16u;
    s.a = bload16(xr, aok ? oa : BUF_OOB);
    s.b = bload16(xr, bok ? oa + (uint32_t)d * pb : BUF_OOB);
}

// registers -> X buffer, transposed to per-channel lattice rows (two neighbouring lattice columns per dword)
__device__ __forceinline__ void write_unit(int j, char *X, int tid, const Unit &s)
{
    int r, c0, h;
    fill_coords(tid, r, c0, h);
    char *dst = X + r * RSTR + (h * 8) * CSTR + ((c0 >> 1) + 6 * j) * 4;
    const uint32_t a[4] = {s.a.x, s.a.y, s.a.z, s.a.w};
    const uint32_t b[4] = {s.b.x, s.b.y, s.b.z, s.b.w};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        *(uint32_t *)(dst + (2 * m) * CSTR) = __builtin_amdgcn_perm(b[m], a[m], 0x05040100u);       // (a & 0xffff) | (b << 16)
        *(uint32_t *)(dst + (2 * m + 1) * CSTR) = __builtin_amdgcn_perm(b[m], a[m], 0x07060302u);   // (a >> 16) | (b & 0xffff0000)
    }
}

#include "dw_lw_body.inc"

// y_b = dwconv(x, w_b), b < 3.  C++ = the prologue (tap table, the wave's resident Toeplitz operands, the first item's tile, the
// per-thread constants); everything after it is ONE generated asm statement (tools/gen_dw_lw.py -> dw_lw_body.inc).
__global__ __launch_bounds__(LNT, 1) void dw_lw_fan3_kernel(DwLwParams p)
{
    constexpr int NB = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, kg = lane >> 4;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lin % p.ncg; lin /= p.ncg;
    const int seg = lin % p.nseg;
    const int n = lin / p.nseg;
    const int c0 = cgi * CG;
    const int ibeg = (int)((long long)p.nitems * seg / p.nseg), iend = (int)((long long)p.nitems * (seg + 1) / p.nseg);
    if (ibeg >= iend) return;   // workgroup-uniform
    const int d = p.dil;
    // (planar tensors: channel group cgi = 16 channels at offset (cgi & 3) * 16 of plane cgi >> 2)
    const bf16_t *ximg = p.x[0] + (p.xplane ? (size_t)(cgi >> 2) * p.xplane + (size_t)n * p.H * p.W * 64 + (cgi & 3) * 16 : (size_t)n * p.H * p.W * p.ldx + c0);
    const size_t img_y = p.yplane ? (size_t)(cgi >> 2) * p.yplane + (size_t)n * p.H * p.W * 64 + (cgi & 3) * 16 : (size_t)n * p.H * p.W * p.ldy + c0;
    const int nrx = (int)((((size_t)p.H * p.W - 1) * p.ldx + CG) * 2), nry = (int)((((size_t)p.H * p.W - 1) * p.ldy + CG) * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)ximg, 0, nrx, 0x00020000);

    // the first item's tile (host-side list of non-empty items: p.items[e] = {descriptor of 16 dwords})
    const int *desc0 = p.items + (size_t)ibeg * 16;
    Item wi;
    wi.ry = desc0[15] & 0xff; wi.rx = (desc0[15] >> 8) & 0xff; wi.ty = (desc0[15] >> 16) & 0xff; wi.tx = (desc0[15] >> 24) & 0xff;
    Unit st[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) fetch_unit(j, p, xr, wi, tid, st[j]);

    // ---- taps -> bf16 table; columns 60..63 of every staged (row, channel) are never filled and must read as zeros: both buffers, once
    {
        bf16_t *wt = (bf16_t *)(smem + WT_OFF);
        for (int e = tid; e < NB * (WTBYTES / 2); e += LNT) wt[e] = 0;
        for (int e = tid; e < 2 * RY * CG * 2; e += LNT) {
            const int bufi = e / (RY * CG * 2), r2 = e - bufi * (RY * CG * 2);
            *(uint32_t *)(smem + bufi * XB + (r2 >> 5) * RSTR + ((r2 >> 1) & 15) * CSTR + 120 + (r2 & 1) * 4) = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b)
            for (int e = tid; e < CG * 81; e += LNT) {
                const int c = e & 15, tap = e >> 4, ky = tap / 9, kx = tap - ky * 9;
                wt[b * (WTBYTES / 2) + (c * 9 + ky) * 16 + kx] = f32_to_bf16(p.w[b][(size_t)tap * p.C + c0 + c]);
            }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) write_unit(j, smem, tid, st[j]);     // X buffer 0
    __syncthreads();

    // ---- the accumulation file belongs to the asm statements: they name a[0:255] literally (conv_lw.hip's scheme)
    typedef __attribute__((ext_vector_type(32))) float f32x32_t;
    f32x32_t A0, A1, A2, A3, A4, A5, A6, A7;
    asm volatile("" : "=a"(A0), "=a"(A1), "=a"(A2), "=a"(A3), "=a"(A4), "=a"(A5), "=a"(A6), "=a"(A7));
#define DWLW_ACC_RW "+a"(A0), "+a"(A1), "+a"(A2), "+a"(A3), "+a"(A4), "+a"(A5), "+a"(A6), "+a"(A7)

    // ---- resident Toeplitz operands: fragment f = (b*4 + c)*7 + m -> a[4f : 4f+3] (f < 64) or the wave's LDS stash (loaded into
    // v[176 + 4(f-64) ..] by the asm prologue).  Lane (column j = fi, K group kg) of MFMA m holds taps kx = cg*8 + e - j of tap row ky.
    int sl[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        int ky, cg;
        slot_of(4 * m + kg, ky, cg);
        sl[m] = ky < 0 ? 0 : ky * RSTR + cg * 16;
        int widx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kx = cg * 8 + e - fi;
            widx[e] = (ky >= 0 && kx >= 0 && kx < 9 ? kx : 15) * 2;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const char *wr = smem + WT_OFF + b * WTBYTES + ((wave * 4 + c) * 9 + max(ky, 0)) * 32;
                uint32_t v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = *(const bf16_t *)(wr + widx[e]);
                const uint32_t f0 = v[0] | (v[1] << 16), f1 = v[2] | (v[3] << 16), f2 = v[4] | (v[5] << 16), f3 = v[6] | (v[7] << 16);
                const int f = (b * 4 + c) * NM + m;     // compile-time: every loop is unrolled
                if (f < 64) {
                    asm volatile("v_accvgpr_write_b32 a[%[i0]], %[f0]\n\tv_accvgpr_write_b32 a[%[i1]], %[f1]\n\tv_accvgpr_write_b32 a[%[i2]], %[f2]\n\tv_accvgpr_write_b32 a[%[i3]], %[f3]"
                                 : DWLW_ACC_RW : [f0] "v"(f0), [f1] "v"(f1), [f2] "v"(f2), [f3] "v"(f3), [i0] "n"(4 * f), [i1] "n"(4 * f + 1), [i2] "n"(4 * f + 2), [i3] "n"(4 * f + 3));
                } else {
                    *(uint4 *)(smem + STASH_OFF + (wave * 20 + (f - 64)) * 1024 + lane * 16) = make_uint4(f0, f1, f2, f3);
                }
            }
    }

    // ---- per-thread constants -> LDS (the asm prologue loads them into v152 .. v166)
    {
        const int njt0 = desc0[7] & 0xff;   // byte offset of the first tile's first column (0)
        uint32_t *lc = (uint32_t *)(smem + LC_OFF + tid * 64);
        const int arow = min(fi, TLY - 1) * RSTR + (wave * 4) * CSTR;
#pragma unroll
        for (int m = 0; m < NM; ++m) lc[m] = (uint32_t)(arow + njt0 + sl[m]);                                  // fragment read addresses, X buffer 0, tile 0
        lc[7] = (uint32_t)(S_OFF + ((kg * 4) * 16 + fi) * SPX + wave * 8);                                      // staging writes: (row 4 kg, column fi), the wave's 8 B
        const int so_h = tid & 1, so_col = (tid >> 1) & 15, so_rq = tid >> 5;
        lc[8] = (uint32_t)(S_OFF + (so_rq * 16 + so_col) * SPX + so_h * 16);                                    // staging reads of the store-out
        lc[9] = (uint32_t)(so_rq | (so_col << 8));
        lc[10] = (uint32_t)(d * so_rq * p.W + d * so_col) * (uint32_t)p.ldy * 2u + (uint32_t)so_h * 16u;        // this thread's pixel inside an item's output
        int r, cc0, h;
        fill_coords(tid, r, cc0, h);
        lc[11] = (uint32_t)(r | (cc0 << 8));
        lc[12] = (uint32_t)(d * r * p.W + d * cc0) * (uint32_t)p.ldx * 2u + (uint32_t)h * 16u;                  // ... inside an item's staged tile
        lc[13] = (uint32_t)(r * RSTR + (h * 8) * CSTR + (cc0 >> 1) * 4);
        lc[14] = (uint32_t)(STASH_OFF + wave * 20 * 1024 + lane * 16);
    }
    __syncthreads();

    const bf16_t *y0 = p.y[0] + img_y, *y1 = p.y[1] + img_y, *y2 = p.y[2] + img_y;
    const int scol = d * p.ldx * 2, ycol = d * p.ldy * 2, yrow8 = 8 * d * p.W * p.ldy * 2, cnt = iend - ibeg;
    const uint32_t vlc = (uint32_t)(LC_OFF + tid * 64);
    asm volatile(DW_LW3_ASM
                 : DWLW_ACC_RW
                 : [vlc] "v"(vlc), [sx] "s"(ximg), [sy0] "s"(y0), [sy1] "s"(y1), [sy2] "s"(y2), [snrx] "s"(nrx), [snry] "s"(nry),
                   [ssc] "s"(scol), [syc] "s"(ycol), [syr8] "s"(yrow8), [scnt] "s"(cnt), [stab] "s"(desc0)
                 : DW_LW3_CLOBBER);
#undef DWLW_ACC_RW
}

}  // namespace

static void dw_lw_split(int N, int C, int nvalid, int *nseg)
{
    // one workgroup per CU at a time: aim at two rounds of workgroups over the chip, but keep >= 8 items per workgroup so the
    // operand build (once per workgroup) and the pipeline fill stay small
    const long long groups = (long long)N * (C / CG);
    long long s = (512 + groups - 1) / groups;
    if (s > nvalid / 8) s = nvalid / 8;
    if (s < 1) s = 1;
    *nseg = (int)s;
}

// Item descriptors of a geometry (the same list for every image and channel group), built on the host once and kept on the device:
// 16 dwords per non-empty item, in the order the asm reads them (tools/gen_dw_lw.py, D_*).
#include <mutex>
#include <vector>
namespace {
struct ItemTable {
    int H, W, dil, ldx, ldy, dev;
    int nvalid;
    int *dptr;
};
std::mutex g_tab_mu;
std::vector<ItemTable> g_tabs;

// (returns a copy: the vector may grow under another thread's call once the lock is released; dptr == nullptr: no table)
ItemTable item_table(const kd_dw_desc *d)
{
    ItemTable none;
    memset(&none, 0, sizeof(none));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return none;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    for (const ItemTable &t : g_tabs)
        if (t.H == d->H && t.W == d->W && t.dil == d->dil && t.ldx == d->ldx && t.ldy == d->ldy && t.dev == dev) return t;
    const int dl = d->dil, H = d->H, W = d->W;
    const int LH = (H + dl - 1) / dl, LW = (W + dl - 1) / dl;
    const int nty = (LH + TLY - 1) / TLY, ntx = (LW + TLX - 1) / TLX;
    if (nty > 255 || ntx > 255 || dl > 255) return none;
    std::vector<int> host;
    // order: the row tiles of one (class, column tile) back to back -- a tile shares its 8 halo rows with its vertical neighbour, and
    // fetched 9 us apart instead of 25 items apart the second fetch still finds most of them in the XCD's L2
    static int tile_major = -1;      // A/B: KDCC_DW_LW_ORDER=0 = classes innermost (round 6's first order: the halo rows are fetched from HBM twice)
    if (tile_major < 0) { const char *v = getenv("KDCC_DW_LW_ORDER"); tile_major = (v && v[0] == '0') ? 1 : 0; }
    for (int e = 0; e < nty * ntx * dl * dl; ++e) {
        int q = e, rx, ry, tx, ty;
        if (tile_major) {
            rx = q % dl; q /= dl;
            ry = q % dl; q /= dl;
            tx = q % ntx; ty = q / ntx;
        } else {
            ty = q % nty; q /= nty;
            tx = q % ntx; q /= ntx;
            rx = q % dl; ry = q / dl;
        }
        const int Ly = (H - ry + dl - 1) / dl, Lx = (W - rx + dl - 1) / dl;
        const int RV = std::min(TLY, Ly - ty * TLY), CV = std::min(TLX, Lx - tx * TLX);
        if (RV <= 0 || CV <= 0) continue;
        int v[16] = {0};
        const long long xbase = ((long long)(ry + dl * (ty * TLY - 4)) * W + rx + dl * (tx * TLX - 4)) * d->ldx * 2;
        v[0] = (int)(uint32_t)(xbase & 0xffffffffLL);                       // wraps like the device's 32-bit offset arithmetic
        const int rlo = std::max(0, 4 - ty * TLY), rhi = std::min(RY, Ly - ty * TLY + 4);     // staged rows [rlo, rhi) lie inside the image
        const int clo = std::max(0, 4 - tx * TLX), chi = std::min(60, Lx - tx * TLX + 4);
        v[1] = rlo; v[2] = std::max(0, rhi - rlo); v[3] = clo; v[4] = std::max(0, chi - clo);
        v[5] = (int)(((long long)(ry + dl * ty * TLY) * W + rx + dl * tx * TLX) * d->ldy * 2);
        v[6] = RV;
        const int njt = (CV + 15) >> 4, jlast = std::max(((CV + 7) & ~7) - 16, 0);
        for (int t = 0; t < 4; ++t) {   // one byte per tile: byte offset of its first column / its valid columns
            const int tt = std::min(t, njt - 1), cb = std::min(tt * 16, jlast);
            v[7] |= (cb * 2) << (8 * t);
            v[8] |= (t < njt ? CV - cb : 0) << (8 * t);
        }
        v[15] = ry | (rx << 8) | (ty << 16) | (tx << 24);
        {   // timing-only ablations (tuning build): 1 = no output stores (no valid columns), 2 = no load requests (no valid rows)
            static int dbg = -1;
            if (dbg < 0) dbg = KD_TUNING_ENV_INT("KDCC_DW_LW_DBG");
            if (dbg & 1) v[8] = 0;
            if (dbg & 2) v[2] = 0;
        }
        host.insert(host.end(), v, v + 16);
    }
    ItemTable t;
    t.H = H; t.W = W; t.dil = dl; t.ldx = d->ldx; t.ldy = d->ldy; t.dev = dev;
    t.nvalid = (int)(host.size() / 16);
    t.dptr = nullptr;
    if (t.nvalid == 0) return none;
    if (hipMalloc((void **)&t.dptr, host.size() * sizeof(int)) != hipSuccess) return none;
    if (hipMemcpy(t.dptr, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { hipFree(t.dptr); return none; }
    g_tabs.push_back(t);
    return t;
}
}  // namespace

// The lone-wave fan-out launch: 1 = taken, 0 = not eligible (the caller falls back to dw_mfma_fwd_kernel), < 0 = error.
int kd_internal_dw_lw_fanout(const kd_dw_desc *d, int nb, const void *x, const float *const *ws, void *const *ys, hipStream_t s)
{
    if (nb != 3) return 0;
    if (d->dtype != KD_BF16 || d->k != 9 || d->C % CG != 0 || d->ldx % 8 != 0 || d->ldy % 8 != 0) return 0;
    if (!x || !kd_aligned16(x)) return 0;
    for (int b = 0; b < nb; ++b)
        if (!ws[b] || !ys[b] || !kd_aligned16(ys[b])) return 0;
    if ((long long)d->H * d->W * (d->ldx > d->ldy ? d->ldx : d->ldy) * 2 >= (long long)BUF_OOB) return 0;   // 32-bit buffer offsets per image
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("KDCC_DW_MFMA"), *l = getenv("KDCC_DW_LW");   // A/B hooks: 0 = the 8-wave kernel (KDCC_DW_LW) / the register kernel
        enabled = !(e && e[0] == '0') && !(l && l[0] == '0');
    }
    if (!enabled) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return 0;   // (the table upload is a synchronous copy)
    kd_dw_desc dd = *d;
    {
        static int dbg = -1;
        if (dbg < 0) dbg = KD_TUNING_ENV_INT("KDCC_DW_LW_DBG");
        if (dbg & 4) dd.ldy = 16;
        if (dbg & 8) dd.ldx = 16;
        if (dbg & 16) dd.ldy = 64;
        if (dbg & 32) dd.ldx = 64;
    }
    const ItemTable tabv = item_table(&dd);
    if (!tabv.dptr) return 0;
    const ItemTable *tab = &tabv;
    DwLwParams p;
    for (int b = 0; b < MAXB; ++b) {
        p.x[b] = (const bf16_t *)x;
        p.w[b] = ws[b];
        p.y[b] = (bf16_t *)ys[b];
    }
    p.items = tab->dptr;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.dil = d->dil; p.ldx = d->ldx; p.ldy = d->ldy;
    {   // timing-only (tuning build): 4 = outputs / 8 = input addressed as DENSE 16-channel images (32-B pixels side by side): the same
        // requests on cache- and TLB-friendly addresses.  (Tables are keyed on ldx / ldy, so the descriptors follow.)
        static int dbg = -1;
        if (dbg < 0) dbg = KD_TUNING_ENV_INT("KDCC_DW_LW_DBG");
        if (dbg & 4) p.ldy = 16;
        if (dbg & 8) p.ldx = 16;
        // 16 = outputs / 32 = input addressed as channel-planar [C/64][N*H*W][64] tensors (timing: the values land in other places)
        p.xplane = p.yplane = 0;
        if (dbg & 16) { p.ldy = 64; p.yplane = (long long)d->N * d->H * d->W * 64; }
        if (dbg & 32) { p.ldx = 64; p.xplane = (long long)d->N * d->H * d->W * 64; }
    }
    p.nitems = tab->nvalid;
    dw_lw_split(d->N, d->C, tab->nvalid, &p.nseg);
    p.ncg = d->C / CG;
    p.nty = p.ntx = 0;
    const long long blocks = (long long)d->N * p.ncg * p.nseg;
    if (blocks > 0x7fffffffLL || (long long)d->N * d->H * d->W > 0x7fffffffLL) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)dw_lw_fan3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            kd_set_error("kd_dwconv_fwd_fanout: cannot reserve %d B of LDS", LDS_BYTES);
            return KD_ERR_HIP;
        }
        attr_set = true;
    }
    KD_NOTE_KERNEL("dw_lw_fan3_kernel");
    hipLaunchKernelGGL(dw_lw_fan3_kernel, dim3((unsigned)blocks), dim3(LNT), LDS_BYTES, s, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        kd_set_error("kd_dwconv_fwd_fanout(lw): launch failed: %s", hipGetErrorString(err));
        return KD_ERR_HIP;
    }
    return 1;
}

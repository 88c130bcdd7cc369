// Microbenchmark: what do the depthwise kernels' memory access patterns cost, as a function of the piece size?
// Tensor = NHWC bf16 (N, 128, 256, 4096), the ASPP depthwise input.  A workgroup (512 threads) owns CG channels (a piece of
// CG * 2 bytes per pixel) and walks residue-class tiles exactly like dw_mfma_fwd_kernel does: 26 x 52 pixels whose neighbours
// are dil = 5 pixels apart (40 KiB between consecutive pieces of a lattice row).  Modes: read only (sum into a register),
// write only, read + write (copy).  Reports GB/s over the bytes actually moved.
// build: hipcc -O3 --offload-arch=gfx950 piece_bw.hip -o piece_bw ; run: ./piece_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 128, W = 256, C = 4096, DIL = 5, TLY = 26, TLX = 52;

// MODE: 1 read, 2 write, 3 both.  LPP = lanes per pixel piece (piece = LPP * 16 B).
// DENSE: pixels of a tile are neighbours in the image (natural order) instead of dil apart -- what a residue-sorted private
// layout would give.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)   // neighbouring logical ids share an XCD (kd_common.h)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int MODE, int LPP, bool DENSE>
__global__ __launch_bounds__(512) void walk(const char *__restrict__ x, char *__restrict__ y, unsigned *sink, int ncg, int nclass_per_block)
{
    const int tid = threadIdx.x;
    int lin = xcd_remap(blockIdx.x, gridDim.x);   // channel-group siblings (the other pieces of a 128-B line) on one L2
    const int cgi = lin % ncg; lin /= ncg;
    const int nseg = (DIL * DIL + nclass_per_block - 1) / nclass_per_block;
    const int seg = lin % nseg;
    const int n = lin / nseg;
    const size_t img = (size_t)n * H * W * C * 2 + (size_t)cgi * LPP * 16;
    const int sub = tid % LPP, pix0 = tid / LPP;
    constexpr int PPP = 512 / LPP;           // pixels per pass
    u32x4 acc = {0, 0, 0, 0};
    for (int cls = seg * nclass_per_block; cls < min((seg + 1) * nclass_per_block, DIL * DIL); ++cls) {
        const int ry = cls / DIL, rx = cls % DIL;
        for (int p = pix0; p < TLY * TLX; p += PPP) {
            const int ly = p / TLX, lx = p - ly * TLX;
            int yy, xx;
            if (DENSE) { const int q = cls * TLY * TLX + p; yy = q / W; xx = q % W; if (yy >= H) continue; }
            else { yy = ry + DIL * ly; xx = rx + DIL * lx; if (yy >= H || xx >= W) continue; }
            const size_t off = img + ((size_t)yy * W + xx) * C * 2 + sub * 16;
            u32x4 v = {(unsigned)p, (unsigned)cls, 1u, 2u};
            if (MODE & 1) v = *(const u32x4 *)(x + off);
            if (MODE & 2) *(u32x4 *)(y + off) = v;
            else { acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
        }
    }
    if (!(MODE & 2) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

template <int MODE, int LPP, bool DENSE> void run(const char *x, char *y, unsigned *sink, int N, int cpb)
{
    const int ncg = C * 2 / (LPP * 16);
    const int nseg = (DIL * DIL + cpb - 1) / cpb;
    const int blocks = N * ncg * nseg;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((walk<MODE, LPP, DENSE>), dim3(blocks), dim3(512), 0, 0, x, y, sink, ncg, cpb);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((walk<MODE, LPP, DENSE>), dim3(blocks), dim3(512), 0, 0, x, y, sink, ncg, cpb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    // bytes actually touched: pixels inside the image only (the 26 x 52 lattice covers 130 x 260)
    const double bytes = (double)N * H * W * C * 2 * ((MODE & 1 ? 1 : 0) + (MODE & 2 ? 1 : 0));
    printf("%-5s piece %3d B %s classes/block %2d blocks %6d: %7.3f ms  %7.1f GB/s\n", MODE == 1 ? "read" : MODE == 2 ? "write" : "copy",
           LPP * 16, DENSE ? "dense " : "dil 5 ", cpb, blocks, ms, bytes / ms * 1e-6);
}

int main()
{
    const int N = 4;
    const size_t bytes = (size_t)N * H * W * C * 2;
    char *x, *y;
    unsigned *sink;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&sink, 64);
    hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
    for (int cpb : {25, 5}) {
        run<1, 2, false>(x, y, sink, N, cpb); run<1, 4, false>(x, y, sink, N, cpb); run<1, 8, false>(x, y, sink, N, cpb); run<1, 16, false>(x, y, sink, N, cpb);
        run<2, 2, false>(x, y, sink, N, cpb); run<2, 4, false>(x, y, sink, N, cpb); run<2, 8, false>(x, y, sink, N, cpb); run<2, 16, false>(x, y, sink, N, cpb);
        run<3, 2, false>(x, y, sink, N, cpb); run<3, 4, false>(x, y, sink, N, cpb); run<3, 8, false>(x, y, sink, N, cpb); run<3, 16, false>(x, y, sink, N, cpb);
    }
    run<1, 2, true>(x, y, sink, N, 25); run<2, 2, true>(x, y, sink, N, 25); run<3, 2, true>(x, y, sink, N, 25);
    run<1, 8, true>(x, y, sink, N, 25); run<2, 8, true>(x, y, sink, N, 25); run<3, 8, true>(x, y, sink, N, 25);
    hipFree(x); hipFree(y); hipFree(sink);
    return 0;
}

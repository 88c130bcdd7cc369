#!/usr/bin/env python3
"""Generator for tools/ubench/winograd_loop.hip: what would the main loop of an F(2x2, 3x3) Winograd form of the 3x3 convs cost?

profiles/r05_mfma_shape_power.txt: on random bf16 data this chip delivers 1.61 PFLOP/s of MFMA work whatever the instruction stream
-- MACs are the currency.  The Winograd form needs 2.25 x fewer of them, but per MFMA it moves far more through LDS: a workgroup's 64 K
accumulators hold 16 transform components x (64 tiles x 64 channels) instead of one 256 x 256 tile, so per 32-channel k-step a wave
(4 components of 16 MFMAs) reads 32 fragments instead of 16, the transformed weights U[16][32][64] are 64 KiB per workgroup and k-step
(16 LDS-DMA pieces per wave, 4-8 today), and the transformed input V[16][64 tiles][32] has to be MADE there: raw 4 x 4 patches read
from the row buffer, ~2.5 packed adds per value, 64 KiB written back (16 ds_read_b128 + nv VALU + 16 ds_write_b128 per wave).

This is NOT a convolution: it is that instruction mix, hand-dealt between the 64 MFMAs of a k-step exactly like conv_lw_body.inc deals
its own (one wave per SIMD, accumulators a[0:255], fragments double-buffered per COMPONENT in v[128:191], dummy VALU on v[64:127]),
with real LDS traffic and real LDS-DMA from an L2-resident window, random bf16 in LDS.  Output per variant: shader cycles and ns per
k-step, clock, and the chip's MFMA TFLOP/s -- multiply by 2.25 for the direct-equivalent rate and compare with 1.60-1.66 (the direct
loop) to see whether the form is worth building.
usage: python tools/ubench/gen_winograd_loop.py && hipcc -O3 --offload-arch=gfx950 tools/ubench/winograd_loop.hip -o tools/ubench/winograd_loop"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def kstep(nfrag, nraw, nwr, nv, ndma, barrier=True):
    """one k-step of a wave: 64 MFMAs = 4 components x 16; everything else dealt into the slots behind the MFMAs.
    %0 = fragment-read address (V region), %1 = (U region), %2 = DMA lane offset, %6 = raw-read address, %7 = V write address"""
    slots = [[] for _ in range(64)]
    # fragment reads: component c + 1's 8 fragments (4 A + 4 B) under component c's 16 MFMAs
    per_comp = nfrag // 4
    for c in range(4):
        nxt = (c + 1) & 1
        for r in range(per_comp):
            dst = 128 + 32 * nxt + 4 * (r % 8)
            src = "%0" if r % 2 == 0 else "%1"
            slots[16 * c + (r * 16) // max(per_comp, 1)].append(f"ds_read_b128 v[{dst}:{dst + 3}], {src} offset:{((c * 8 + r) % 32) * 1024}")
    for r in range(nraw):
        dst = 192 + 4 * (r % 8)
        slots[1 + (r * 62) // max(nraw, 1)].append(f"ds_read_b128 v[{dst}:{dst + 3}], %6 offset:{(r % 16) * 1024}")
    for r in range(nwr):
        srcr = 64 + 4 * (r % 16)
        slots[2 + (r * 60) // max(nwr, 1)].append(f"ds_write_b128 %7, v[{srcr}:{srcr + 3}] offset:{(r % 16) * 1024}")
    for p in range(ndma):
        slots[3 + (p * 60) // max(ndma, 1)].append(f"global_load_lds_dwordx4 %2, s[20:21] offset:{((p % 8) - 4) * 1024}")
    for k in range(nv):
        d = 64 + (k % 64)
        slots[(k * 64) // max(nv, 1)].append(f"v_pk_add_f16 v{d}, v{64 + (k * 7 + 1) % 64}, v{64 + (k * 13 + 5) % 64}")
    L = []
    m = 0
    for c in range(4):
        cur = c & 1
        for i in range(4):
            for j in range(4):
                acc = 4 * (16 * c + 4 * i + j)
                a, b = 128 + 32 * cur + 4 * i, 128 + 32 * cur + 16 + 4 * j
                L.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{b}:{b + 3}], v[{a}:{a + 3}], a[{acc}:{acc + 3}]")
                L += slots[m]
                m += 1
    L.append(f"s_waitcnt vmcnt({min(ndma, 63)}) lgkmcnt(0)" if ndma else "s_waitcnt lgkmcnt(0)")
    if barrier:
        L.append("s_barrier")
    return L


VARIANTS = [  # name, fragment reads, raw reads, V writes, VALU, DMA pieces  (per wave and k-step)
    ("direct_like", 16, 0, 0, 0, 6),                 # the shipped loop's mix, for reference (1128 cycles in lone_wave)
    ("wino_frags_only", 32, 0, 0, 0, 0),
    ("wino_frags_dma16", 32, 0, 0, 0, 16),
    ("wino_transform_only", 32, 16, 16, 160, 0),
    ("wino_full_valu96", 32, 16, 16, 96, 16),
    ("wino_full_valu160", 32, 16, 16, 160, 16),
    ("wino_full_valu224", 32, 16, 16, 224, 16),
    ("wino_full_dma8", 32, 16, 16, 160, 8),          # a 128-channel-wide U tile shared by two pixel tiles (half the DMA per MFMA)
]


def regs(prefix, lo, hi):
    return ",".join(f'"{prefix}{i}"' for i in range(lo, hi))


KERNEL = r'''
__global__ __launch_bounds__(256, 1) void k_@@NAME@@(const char *src, int iters, unsigned long long *out, float *sink)
{
    __shared__ __attribute__((aligned(1024))) char lds[155648];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 155648 / 4; i += 256) {      // random-looking bf16 around +-1 (power depends on the operands)
        const unsigned h = i * 2654435761u;
        ((unsigned *)lds)[i] = (0x3f803f80u ^ (h & 0x007f007fu)) | ((h >> 3) & 0x80008000u);
    }
    __syncthreads();
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    unsigned va = lbase + lane * 16, vb = lbase + 32768 + lane * 16;                    // V and U fragment regions (32 KiB each here)
    unsigned vraw = lbase + 65536 + wv * 4096 + lane * 16, vwr = lbase + 81920 + wv * 16384 + lane * 16;
    unsigned voff = lane * 16 + wv * 16384;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lbase + 114688 + wv * 8192 + 4096);   // DMA destination
    const char *shared = src + 8192;
    unsigned pos = 0;
    asm volatile(@@VINIT@@ : : "v"(lane) : CLOBBER_V);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const char *b = shared + pos;
        pos = (pos + 65536) & ((1u << 20) - 1);
        const unsigned long long bb = (unsigned long long)b;
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bb), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bb >> 32));
        asm volatile("s_mov_b32 s20, %3\n\ts_mov_b32 s21, %4\n\ts_mov_b32 m0, %5\n\ts_nop 4\n\t"
                     "@@BODY@@"
                     :
                     : "v"(va), "v"(vb), "v"(voff), "s"(blo), "s"(bhi), "s"(m0v), "v"(vraw), "v"(vwr)
                     : "memory", "s20", "s21", CLOBBER_ACC, CLOBBER_FRAG, CLOBBER_V);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r;
    asm volatile("s_nop 15\n\tv_accvgpr_read_b32 %0, a5" : "=v"(r)::);
    if (r == 1.2345f) sink[threadIdx.x] = r + ((float *)lds)[threadIdx.x];
    if (lane == 0) { out[(blockIdx.x * 4 + wv) * 2] = t1 - t0; out[(blockIdx.x * 4 + wv) * 2 + 1] = r1 - r0; }
}
'''

SRC = r'''// GENERATED by tools/ubench/gen_winograd_loop.py -- do not edit.  See that file for what this measures.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CLOBBER_ACC @@ACC@@
#define CLOBBER_FRAG @@FRAG@@
#define CLOBBER_V @@VV@@

@@KERNELS@@

int main()
{
    const int iters = 256;                          // x 2 k-steps per iteration
    char *src; unsigned long long *out; float *sink;
    hipMalloc(&src, (size_t)8 << 20); hipMemset(src, 0x3c, (size_t)8 << 20);
    hipMalloc(&out, 256 * 4 * 2 * sizeof(unsigned long long)); hipMalloc(&sink, 1 << 20);
    const double flop = 2.0 * 128 * 128 * 32;       // per wave and k-step (64 MFMAs of 16x16x32)
    printf("%-24s %14s %12s %8s %18s %24s\n", "variant", "cycles/k-step", "ns/k-step", "GHz", "MFMA TFLOP/s", "x 2.25 direct-equivalent");
@@CALLS@@
    return 0;
}
'''

CALL = r'''    {
        for (int rep = 0; rep < 3; ++rep) k_@@NAME@@<<<256, 256>>>(src, iters, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2048);
        hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ns;
        for (int w = 0; w < 1024; ++w) { cyc.push_back((double)h[2 * w]); ns.push_back((double)h[2 * w + 1] * 10.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
        const double ks = 2.0 * iters, c = cyc[512] / ks, n = ns[512] / ks, tf = 1024.0 * flop / n / 1e3;
        printf("%-24s %14.1f %12.1f %8.3f %18.0f %24.0f\n", "@@NAME@@", c, n, c / n, tf, @@EQ@@);
    }
'''


def main():
    ks, cs = [], []
    for name, nfrag, nraw, nwr, nv, ndma in VARIANTS:
        body = "\\n\\t".join(kstep(nfrag, nraw, nwr, nv, ndma) + kstep(nfrag, nraw, nwr, nv, ndma))
        ks.append(KERNEL.replace("@@NAME@@", name).replace("@@BODY@@", body)
                  .replace("@@VINIT@@", " ".join(f'"v_cvt_f32_i32 v{64 + k}, %0\\n\\t"' for k in range(64))))
        cs.append(CALL.replace("@@NAME@@", name).replace("@@EQ@@", "tf" if name == "direct_like" else "tf * 2.25"))
    out = (SRC.replace("@@KERNELS@@", "\n".join(ks)).replace("@@CALLS@@", "\n".join(cs)).replace("@@ACC@@", regs("a", 0, 256))
           .replace("@@FRAG@@", regs("v", 128, 224)).replace("@@VV@@", regs("v", 64, 128)))
    with open(os.path.join(HERE, "winograd_loop.hip"), "w") as f:
        f.write(out)
    print("wrote winograd_loop.hip")


if __name__ == "__main__":
    main()

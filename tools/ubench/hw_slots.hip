// Which workgroups share a CU?  512 workgroups of 256 threads, 72 KiB of LDS and 256 registers per wave (two fit a CU): every workgroup
// records HW_ID / XCC_ID of its wave 0 and spins so that all of them are resident together.  Prints, per block id, (xcc, se, sh, cu,
// simd, wave slot) and the block pairs found on one CU.  build: hipcc -O3 --offload-arch=gfx950 hw_slots.hip -o hw_slots
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out)
{
    __shared__ char lds[73728];
    asm volatile("v_mov_b32 v127, 0\n\tv_accvgpr_write_b32 a127, 0" ::: "v127", "a127");
    lds[threadIdx.x] = 1;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20000) __builtin_amdgcn_s_sleep(32);   // 200 us
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    if (lds[threadIdx.x] == 7) out[0] = 0;
}
int main()
{
    unsigned *d; hipMalloc(&d, 512 * 2 * 4);
    k<<<512, 256>>>(d); hipDeviceSynchronize();
    std::vector<unsigned> h(1024); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned wave = hw & 15, simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (b < 24 || (b >= 256 && b < 272)) printf("block %3d: xcc %u se %u sh %u cu %2u simd %u wave slot %u\n", b, xcc, se, sh, cuid, simd, wave);
        cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    int n2 = 0; std::map<int, int> diff, slots;
    for (auto &e : cu) { if (e.second.size() == 2) { ++n2; diff[e.second[1] - e.second[0]]++; } }
    printf("distinct CUs %zu, with two workgroups %d; block-id distance of co-resident pairs:", cu.size(), n2);
    for (auto &e : diff) printf(" %d x%d", e.first, e.second);
    int s0 = 0, s1 = 0;
    for (auto &e : cu) if (e.second.size() == 2) { s0 += (h[2 * e.second[0]] & 15); s1 += (h[2 * e.second[1]] & 15); }
    printf("\nmean wave slot of the lower / higher block id of a pair: %.2f / %.2f\n", s0 / (double)n2, s1 / (double)n2);
    return 0;
}

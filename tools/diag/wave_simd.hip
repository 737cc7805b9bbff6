// Which SIMD does wave w of a 768-thread (12-wave) workgroup run on?  HW_REG_HW_ID (gfx9 layout): wave_id [3:0], simd_id [5:4],
// pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13].  Build: hipcc --offload-arch=gfx950 tools/diag/wave_simd.hip -o build/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out)
{
    extern __shared__ float lds[];
    if ((threadIdx.x & 63) == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = v;
    }
    lds[threadIdx.x] = 1.f;
}
int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 12, blocks = 256;
    unsigned *d;
    hipMalloc(&d, blocks * waves * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 96 * 1024, 0, d);
    std::vector<unsigned> h(blocks * waves);
    hipMemcpy(h.data(), d, blocks * waves * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("block %d simd of waves 0..%d:", b, waves - 1);
        for (int w = 0; w < waves; ++w) printf(" %u", (h[b * waves + w] >> 4) & 3);
        printf("   (cu %u se %u)\n", (h[b * waves] >> 8) & 15, (h[b * waves] >> 13) & 7);
    }
    int hist[16] = {0};            // pattern check over all blocks: simd(w) == w % 4 ?
    int rr = 0, grp_distinct = 0;
    for (int b = 0; b < blocks; ++b) {
        bool ok = true;
        for (int w = 0; w < waves; ++w) ok &= ((h[b * waves + w] >> 4) & 3) == (unsigned)(w & 3);
        rr += ok;
        unsigned m = 0;
        for (int w = 0; w < 4; ++w) m |= 1u << ((h[b * waves + w] >> 4) & 3);
        grp_distinct += m == 15u;
    }
    printf("{\"blocks\": %d, \"simd_is_wave_mod_4\": %d, \"waves_0_3_on_four_simds\": %d}\n", blocks, rr, grp_distinct);
    return 0;
}

// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits 5:4 = SIMD id, 11:8 = CU id on gfx9-family parts)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/simd_probe.hip -o tools/microbench/simd_probe && tools/microbench/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out, int lds_words) {
    extern __shared__ unsigned smem[];
    if (lds_words > 0) smem[threadIdx.x % lds_words] = threadIdx.x;
    __syncthreads();
    unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    const int nb = 16;
    hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 149504);
    for (int lds = 0; lds < 2; ++lds) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(512), lds ? 149504 : 0, 0, d, lds ? 149504 / 4 : 0);
        unsigned h[nb * 8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s LDS\n", lds ? "149504 B of" : "no");
        for (int b = 0; b < nb; ++b) {
            printf(" block %2d cu %2u:", b, (h[b * 8] >> 8) & 15);
            for (int w = 0; w < 8; ++w) printf(" w%d->simd%u", w, (h[b * 8 + w] >> 4) & 3);
            printf("\n");
        }
    }
    return 0;
}

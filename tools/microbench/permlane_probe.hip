// prints what v_permlane32_swap / v_permlane16_swap return for (u, u) with u = lane id
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned u = threadIdx.x;
    auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    out[threadIdx.x] = p[0]; out[64 + threadIdx.x] = p[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32[0]", "swap32[1]", "swap16[0]", "swap16[1]"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; ++i) printf(" %u", h[a * 64 + i]); printf("\n"); }
    return 0;
}

// What the matrix pipes deliver on this device with operands in registers (no memory traffic in the loop):
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16, random or zero operands, 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o tools/microbench/mfma_peak && tools/microbench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* in, float* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int k = 0; k < 4; ++k) {
        uint4 u = in[(t * 8 + k) & 0xFFFF], v = in[(t * 8 + 4 + k) & 0xFFFF];
        a[k] = *(bf16x8*)&u;
        b[k] = *(bf16x8*)&v;
    }
    if constexpr (SHAPE == 16) {
        f32x4 c[16];
        for (int k = 0; k < 16; ++k) c[k] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b[k >> 2], c[k], 0, 0, 0);
        }
        float s = 0;
        for (int k = 0; k < 16; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
        out[t] = s;
    } else {
        f32x16 c[4];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 16; ++j) c[k][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + r) & 3], b[k], c[k], 0, 0, 0);
        }
        float s = 0;
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 16; ++j) s += c[k][j];
        out[t] = s;
    }
}

int main() {
    const int nb = 256, nt = 512, iters = 4000;
    uint4* in;
    float* out;
    hipMalloc(&in, 65536 * 16);
    hipMalloc(&out, nb * 1024 * 4);
    unsigned short* h = (unsigned short*)malloc(65536 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        for (int k = 0; k < 65536 * 8; ++k) {
            // bf16 of a uniform value in [-1, 1): sign random, exponent 118..126, mantissa random
            unsigned r = (unsigned)rand();
            h[k] = zero ? 0 : (unsigned short)(((r & 1) << 15) | ((118 + (r >> 1) % 9) << 7) | ((r >> 8) & 127));
        }
        hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
        for (int shape = 16; shape <= 32; shape += 16)
            for (int waves = 1; waves <= 2; ++waves) {
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    hipEventRecord(e0);
                    for (int l = 0; l < 5; ++l) {
                        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(nb), dim3(nt / (3 - waves)), 0, 0, in, out, iters);
                        else hipLaunchKernelGGL(mfma_loop<32>, dim3(nb), dim3(nt / (3 - waves)), 0, 0, in, out, iters);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 2 && ms / 5 < best) best = ms / 5;
                }
                const double mf = (double)nb * (nt / (3 - waves) / 64) * iters * (shape == 16 ? 16 * 16384.0 : 8 * 32768.0);
                printf("%s operands, %dx%d, %d wave(s) per SIMD: %.3f ms per launch = %.0f TFLOP/s\n", zero ? "zero" : "random", shape, shape,
                       waves, best, mf / best * 1e-9);
            }
    }
    return 0;
}

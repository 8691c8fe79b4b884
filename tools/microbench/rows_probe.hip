// rows_sum / rows_max / wave_sum (csrc/common.h) against the shuffle forms, on values produced by VALU right before
#include "../../fastspeech2_amd/csrc/common.h"
#include <stdio.h>
#include <math.h>
using namespace fs2;
extern "C" void fs2_set_error(const char*, ...) {}
__global__ void k(const float* in, float* out, int n) {
    float acc_s = 0.f, acc_m = -INFINITY, ref_s = 0.f, ref_m = -INFINITY, ws = 0.f, wref = 0.f;
    for (int it = 0; it < n; ++it) {
        float v = in[it * 64 + threadIdx.x] * 1.25f + (float)it;
        float s = rows_sum(v), m = rows_max(v);
        float rs = v + __shfl_xor(v, 16, 64); rs += __shfl_xor(rs, 32, 64);
        float rm = fmaxf(v, __shfl_xor(v, 16, 64)); rm = fmaxf(rm, __shfl_xor(rm, 32, 64));
        acc_s += s; ref_s += rs; acc_m = fmaxf(acc_m, m) + 0.f * m; ref_m = fmaxf(ref_m, rm);
        float w = wave_sum(v); float wr = v;
        for (int o = 32; o > 0; o >>= 1) wr += __shfl_xor(wr, o, 64);
        ws += w; wref += wr;
    }
    out[threadIdx.x] = acc_s; out[64 + threadIdx.x] = ref_s; out[128 + threadIdx.x] = acc_m; out[192 + threadIdx.x] = ref_m;
    out[256 + threadIdx.x] = ws; out[320 + threadIdx.x] = wref;
}
int main() {
    const int n = 50; float h[64 * n]; for (int i = 0; i < 64 * n; ++i) h[i] = sinf(i * 0.37f) * 3.f;
    float *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 384 * 4); (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n); float r[384]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    double es = 0, em = 0, ew = 0; for (int i = 0; i < 64; ++i) { es = fmax(es, fabs(r[i] - r[64 + i])); em = fmax(em, fabs(r[128 + i] - r[192 + i])); ew = fmax(ew, fabs(r[256 + i] - r[320 + i])); }
    printf("rows_sum max diff %g, rows_max max diff %g, wave_sum max diff %g (values ~%g)\n", es, em, ew, r[64]);
    return 0;
}

// Training-mode BatchNorm1d of the (repaired) PostNet over packed frames (network/fastspeech2.py:13-54 of the
// reference: Conv1d(bias=False) -> BatchNorm1d -> Tanh (not on the last layer) -> Dropout).  Rows are the frames of
// the whole batch ([M, C] fp32, the conv outputs), the statistics are taken over all M rows per channel.
//
//   fs2_bn_stats      mean / rstd / biased variance per channel           (two launches: partial sums, finalise in f64)
//   fs2_bn_apply      out = dropout(act(gamma * (y - mean) * rstd + beta)) [+ residual], as T and / or fp32
//   fs2_bn_bwd        dgamma, dbeta and dY = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)), dz = the gradient
//                     behind dropout and tanh (recomputed from y: nothing but y and the statistics is kept)
// stats == null means "no BatchNorm" (use_batch_norm=False): identity normalisation, dY = dz.
#include "common.h"

namespace fs2 {

constexpr int BN_ROWS = 128;      // rows per workgroup of the reduction kernels

struct BnArgs {
    const float* y;        // [M, C] conv output
    const float* stats;    // [3][C] mean, rstd, biased var (or null)
    const float* gamma;
    const float* beta;
    int M, C;
    int act;               // 1: tanh
    Dropout drop;
};

__device__ __forceinline__ float bn_hat(const BnArgs& a, int c, float y) {
    return a.stats ? (y - a.stats[c]) * a.stats[a.C + c] : y;
}
__device__ __forceinline__ float bn_affine(const BnArgs& a, int c, float xh) {
    return a.stats ? xh * a.gamma[c] + a.beta[c] : xh;
}

// partial[blk][k][c]: k = 0 sum of u, 1 sum of u * v over the block's rows.  MODE 0: u = y, v = y (statistics);
// MODE 1: u = dz, v = xhat (backward sums)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnArgs a, const float* __restrict__ dout, int ldd,
                                                        float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * BN_ROWS;
    float s0 = 0.f, s1 = 0.f;
    if (c < a.C) {
        const int rend = min(r0 + BN_ROWS, a.M);
        for (int r = r0 + wave; r < rend; r += 4) {
            const float y = a.y[(size_t)r * a.C + c];
            if (MODE == 0) {
                s0 += y;
                s1 += y * y;
            } else {
                const float xh = bn_hat(a, c, y);
                float dz = dout[(size_t)r * ldd + c];
                if (a.drop.thresh) dz *= drop_mul(a.drop, (uint32_t)r, (uint32_t)c);
                if (a.act == 1) {
                    const float t = tanhf(bn_affine(a, c, xh));
                    dz *= 1.f - t * t;
                }
                s0 += dz;
                s1 += dz * xh;
            }
        }
    }
    red[0][wave][lane] = s0;
    red[1][wave][lane] = s1;
    __syncthreads();
    if (wave == 0 && c < a.C) {
        partial[((size_t)blockIdx.x * 2 + 0) * a.C + c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        partial[((size_t)blockIdx.x * 2 + 1) * a.C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    }
}

// sums of the partials in f64 (fixed order); MODE 0 -> stats[3][C]; MODE 1 -> sums[2][C] (+ dgamma, dbeta)
template <int MODE>
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int M, int C, float eps,
                                   float* __restrict__ out, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += (double)partial[((size_t)b * 2 + 0) * C + c];
        s1 += (double)partial[((size_t)b * 2 + 1) * C + c];
    }
    if (MODE == 0) {
        const double mean = s0 / M;
        double var = s1 / M - mean * mean;
        var = var > 0.0 ? var : 0.0;
        out[c] = (float)mean;
        out[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        out[2 * C + c] = (float)var;
    } else {
        out[c] = (float)(s0 / M);
        out[C + c] = (float)(s1 / M);
        if (dbeta) dbeta[c] = (float)s0;
        if (dgamma) dgamma[c] = (float)s1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnArgs a, const float* __restrict__ res, int ldr,
                                                       float* __restrict__ out_f32, int ldf, T* __restrict__ out_t, int ldt) {
    const int cq = a.C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * cq) return;
    const int r = (int)(idx / cq), c0 = (int)(idx % cq) * 4;
    const f32x4 y = *(const f32x4*)(a.y + (size_t)r * a.C + c0);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float z = bn_affine(a, c0 + j, bn_hat(a, c0 + j, y[j]));
        if (a.act == 1) z = tanhf(z);
        if (a.drop.thresh) z *= drop_mul(a.drop, (uint32_t)r, (uint32_t)(c0 + j));
        if (res) z += res[(size_t)r * ldr + c0 + j];
        v[j] = z;
    }
    if (out_f32) *(f32x4*)(out_f32 + (size_t)r * ldf + c0) = v;
    if (out_t) store4<T>(out_t + (size_t)r * ldt + c0, v);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnArgs a, const float* __restrict__ dout, int ldd,
                                                           const float* __restrict__ sums, T* __restrict__ dy_t, int ldt) {
    const int cq = a.C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * cq) return;
    const int r = (int)(idx / cq), c0 = (int)(idx % cq) * 4;
    const f32x4 y = *(const f32x4*)(a.y + (size_t)r * a.C + c0);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        const float xh = bn_hat(a, c, y[j]);
        float dz = dout[(size_t)r * ldd + c];
        if (a.drop.thresh) dz *= drop_mul(a.drop, (uint32_t)r, (uint32_t)c);
        if (a.act == 1) {
            const float t = tanhf(bn_affine(a, c, xh));
            dz *= 1.f - t * t;
        }
        v[j] = a.stats ? a.gamma[c] * a.stats[a.C + c] * (dz - sums[c] - xh * sums[a.C + c]) : dz;
    }
    store4<T>(dy_t + (size_t)r * ldt + c0, v);
}

static inline int bn_blocks(int M) { return (M + BN_ROWS - 1) / BN_ROWS; }

}  // namespace fs2

using namespace fs2;

// bytes of the `workspace` argument of fs2_bn_stats / fs2_bn_bwd
extern "C" long long fs2_bn_workspace(int M, int C) { return ((long long)bn_blocks(M) * 2 + 2) * C * 4 + 256; }

// stats[3][C] = mean, 1 / sqrt(var + eps), biased variance of y[M, C] per channel (nn.BatchNorm1d in training mode; the
// caller folds them into the running statistics)
extern "C" int fs2_bn_stats(const float* y, float* stats, float* workspace, int M, int C, float eps, hipStream_t stream) {
    FS2_CHECK_ARG(y && stats && workspace && M > 0 && C > 0, "fs2_bn_stats: bad args");
    BnArgs a{};
    a.y = y; a.M = M; a.C = C;
    const int nb = bn_blocks(M);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nb, (C + 63) / 64), dim3(256), 0, stream, a, (const float*)nullptr, 0,
                       workspace);
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + 127) / 128), dim3(128), 0, stream, (const float*)workspace, nb, M, C,
                       eps, stats, (float*)nullptr, (float*)nullptr);
    FS2_CHECK_LAUNCH("fs2_bn_stats");
    return FS2_OK;
}

// out = dropout(act(BN(y))) (+ res): out_t [M, ldt] in `dtype` and / or out_f32 [M, ldf]; stats null = no BatchNorm
extern "C" int fs2_bn_apply(const float* y, const float* stats, const float* gamma, const float* beta, int act,
                            const float* res, int ldr, float* out_f32, int ldf, void* out_t, int ldt, int M, int C, int dtype,
                            unsigned seed, unsigned site, float rate, hipStream_t stream) {
    FS2_CHECK_ARG(y && (out_f32 || out_t) && M >= 0 && C > 0 && C % 4 == 0, "fs2_bn_apply: bad args (C %% 4 == 0)");
    FS2_CHECK_ARG(!stats || (gamma && beta), "fs2_bn_apply: statistics without gamma / beta");
    FS2_CHECK_ARG((!out_t || (ldt >= C && ldt % 4 == 0)) && (!out_f32 || (ldf >= C && ldf % 4 == 0)) && (!res || ldr >= C),
                  "fs2_bn_apply: bad leading dims");
    FS2_CHECK_ARG(rate >= 0.f && rate < 1.f && (dtype == 0 || dtype == 1) && (act == 0 || act == 1), "fs2_bn_apply: bad rate / dtype / act");
    if (M == 0) return FS2_OK;
    BnArgs a{};
    a.y = y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.act = act;
    a.drop = fs2_make_dropout(seed, site, rate);
    const long long n = (long long)M * (C / 4);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, a, res, ldr, out_f32, ldf, (bf16_t*)out_t, ldt);
    else
        hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, stream, a, res, ldr, out_f32, ldf, (float*)out_t, ldt);
    FS2_CHECK_LAUNCH("fs2_bn_apply");
    return FS2_OK;
}

// Backward of fs2_bn_apply: dout [M, ldd] fp32 is the gradient of its output; writes dgamma / dbeta (when stats) and the
// gradient of y as `dtype` into dy_t [M, ldt] (the operand of the conv's weight- and input-gradient GEMMs).
extern "C" int fs2_bn_bwd(const float* y, const float* stats, const float* gamma, const float* beta, int act,
                          const float* dout, int ldd, float* dgamma, float* dbeta, void* dy_t, int ldt, float* workspace,
                          int M, int C, int dtype, unsigned seed, unsigned site, float rate, hipStream_t stream) {
    FS2_CHECK_ARG(y && dout && dy_t && M > 0 && C > 0 && C % 4 == 0 && ldd >= C && ldt >= C && ldt % 4 == 0,
                  "fs2_bn_bwd: bad args (C %% 4 == 0)");
    FS2_CHECK_ARG(!stats || (gamma && beta && workspace), "fs2_bn_bwd: statistics without gamma / beta / workspace");
    FS2_CHECK_ARG(rate >= 0.f && rate < 1.f && (dtype == 0 || dtype == 1) && (act == 0 || act == 1), "fs2_bn_bwd: bad rate / dtype / act");
    BnArgs a{};
    a.y = y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.act = act;
    a.drop = fs2_make_dropout(seed, site, rate);
    float* sums = nullptr;
    if (stats) {
        const int nb = bn_blocks(M);
        sums = workspace + (size_t)nb * 2 * C;
        hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nb, (C + 63) / 64), dim3(256), 0, stream, a, dout, ldd, workspace);
        hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + 127) / 128), dim3(128), 0, stream, (const float*)workspace, nb, M,
                           C, 0.f, sums, dgamma, dbeta);
    }
    const long long n = (long long)M * (C / 4);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, a, dout, ldd, (const float*)sums,
                           (bf16_t*)dy_t, ldt);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, stream, a, dout, ldd, (const float*)sums,
                           (float*)dy_t, ldt);
    FS2_CHECK_LAUNCH("fs2_bn_bwd");
    return FS2_OK;
}

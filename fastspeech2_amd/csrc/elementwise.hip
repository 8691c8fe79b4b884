// HBM-bound row kernels of the FastSpeech2 forward path (gfx950): embedding + positional
// encoding, LayerNorm, tiny output heads, duration rounding, the LengthRegulator (scan + gather),
// variance-embedding add fused with the decoder positional encoding, and packed<->padded moves.
// All activations are packed ragged matrices [rows, channels], channel-last, fp32 residual stream
// with an optional low-precision (bf16) copy for the MFMA kernels.
#include "common.h"
#include <stdlib.h>

namespace fs2 {

// ---------------------------------------------------------------------------------------
// row tables: seg[m] = (start, end) of the utterance that owns row m, pos[m] = m - start
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int* __restrict__ cu, int B, int m) {
    // largest b with cu[b] <= m   (cu is non-decreasing, cu[0] = 0, empty segments allowed)
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= m) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void build_seg_kernel(const int* __restrict__ cu, int B, int2* __restrict__ seg,
                                 int* __restrict__ pos, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int b = find_segment(cu, B, m);
    seg[m] = make_int2(cu[b], cu[b + 1]);
    if (pos) pos[m] = m - cu[b];
}

// ---------------------------------------------------------------------------------------
// a1: Embedding(padding_idx=0) + PositionalEncoding / ScaledPositionalEncoding
//   network/fastspeech2.py:116-120, network/encoder.py:125-128 (x*sqrt(d)+pe), :142-145 (x+alpha*pe)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_posenc_kernel(const long long* __restrict__ ids, const int* __restrict__ pos,
                                    const float* __restrict__ emb, const float* __restrict__ pe,
                                    float xscale, float alpha, const float* __restrict__ alpha_dev,
                                    float* __restrict__ out_f32, T* __restrict__ out_t, int M, int d, int idim) {
    if (alpha_dev) alpha = *alpha_dev;       // training: the learnable scale is read on the device (no host sync)
    const int dv = d >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * dv) return;
    const int m = (int)(idx / dv), c = (int)(idx % dv) * 4;
    long long id = ids[m];
    id = id < 0 ? 0 : (id >= idim ? idim - 1 : id);
    const f32x4 e = *(const f32x4*)(emb + (size_t)id * d + c);
    const f32x4 p = *(const f32x4*)(pe + (size_t)pos[m] * d + c);
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = e[r] * xscale + alpha * p[r];
    if (out_f32) *(f32x4*)(out_f32 + (size_t)m * d + c) = v;
    if (out_t) store4<T>(out_t + (size_t)m * d + c, v);
}

// ---------------------------------------------------------------------------------------
// a2: LayerNorm over channels, eps 1e-12 (network/utils.py:39-47). One wave per row,
// two-pass statistics in registers (mean, then centred variance) in fp32.
// ---------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 1024

// y = x + sum_p parts[p]  (residual add of GEMM outputs, incl. split-K partials; x may be null),
// optionally stored (sum_out, may alias x), then LayerNorm(y) -> fp32 and/or T copies.
// (network/encoder.py:178-187: `residual + sublayer(x)` followed / preceded by norm1 / norm2)
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ parts,
                                                            int nparts, long long part_stride,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* sum_out, float* out_f32, T* __restrict__ out_t,
                                                            int M, int C, float eps, Dropout dparts, Dropout dout) {
    // dparts: dropout of the sub-layer output sum_p parts[p] before the residual add (network/encoder.py:178,:185);
    // dout: dropout of the LayerNorm output (predictors, network/variance_adaptor.py:75,:111).  Training only.
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cv = C >> 2;
    const size_t roff = (size_t)row * C;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            f32x4 t = x ? *(const f32x4*)(x + roff + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (dparts.thresh) {
                f32x4 ps = {0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < nparts; ++p) ps += *(const f32x4*)(parts + (size_t)p * part_stride + roff + j * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] += ps[r] * drop_mul(dparts, (uint32_t)row, (uint32_t)(j * 4 + r));
            } else {
                // four loads in flight at a time, added in the order p = 0, 1, 2 ... (the sum is the same, bit for bit)
                const float* src = parts + roff + j * 4;
                int p = 0;
                for (; p + 4 <= nparts; p += 4) {
                    const f32x4 a0 = *(const f32x4*)(src + (size_t)p * part_stride);
                    const f32x4 a1 = *(const f32x4*)(src + (size_t)(p + 1) * part_stride);
                    const f32x4 a2 = *(const f32x4*)(src + (size_t)(p + 2) * part_stride);
                    const f32x4 a3 = *(const f32x4*)(src + (size_t)(p + 3) * part_stride);
                    t += a0; t += a1; t += a2; t += a3;
                }
                for (; p < nparts; ++p) t += *(const f32x4*)(src + (size_t)p * part_stride);
            }
            v[k] = t;
            if (sum_out) *(f32x4*)(sum_out + roff + j * 4) = t;
            s += (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
    if (!w) return;   // sum only
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = v[k][r] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            const f32x4 ww = *(const f32x4*)(w + j * 4);
            const f32x4 bb = *(const f32x4*)(b + j * 4);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * ww[r] + bb[r];
            if (dout.thresh) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] *= drop_mul(dout, (uint32_t)row, (uint32_t)(j * 4 + r));
            }
            if (out_f32) *(f32x4*)(out_f32 + roff + j * 4) = o;
            if (out_t) store4<T>(out_t + roff + j * 4, o);
        }
    }
}

// The same operation, one wave per four consecutive rows, ONE ROW PER 16-LANE GROUP (as layernorm_rowdot_kernel below):
// a lane holds NV = C / 64 float4 columns of its row, (1 + nparts) * NV 16-byte loads are in flight per lane and the two
// reductions of four rows are four DPP steps each.  Partials are added in the same order as above; the statistics are
// summed in another lane order (fp32 rounding-level differences), so the exact-duration modes keep the form above.
template <typename T, int NV>
__global__ __launch_bounds__(256) void add_layernorm_rows4_kernel(const float* __restrict__ x, const float* __restrict__ parts,
                                                                  int nparts, long long part_stride,
                                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                                  float* sum_out, float* out_f32, T* __restrict__ out_t,
                                                                  int M, int C, float eps) {
    const int lane = threadIdx.x & 63, sub = lane & 15, quarter = lane >> 4;
    const int grp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int row = grp * 4 + quarter;
    if (grp * 4 >= M) return;
    const bool live = row < M;
    const size_t roff = (size_t)(live ? row : M - 1) * C;      // lanes of a missing row read the last one and store nothing
    f32x4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        v[k] = x ? *(const f32x4*)(x + roff + (sub + k * 16) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < nparts; ++p) {
        const float* pp = parts + (size_t)p * part_stride + roff;
        f32x4 t[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) t[k] = *(const f32x4*)(pp + (sub + k * 16) * 4);
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += t[k];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (sum_out && live) *(f32x4*)(sum_out + roff + (sub + k * 16) * 4) = v[k];
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    const float mean = row16_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dlt = v[k][r] - mean;
            q += dlt * dlt;
        }
    const float rstd = 1.0f / sqrtf(row16_sum(q) / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (sub + k * 16) * 4;
        const f32x4 ww = *(const f32x4*)(w + c);
        const f32x4 bb = *(const f32x4*)(b + c);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * ww[r] + bb[r];
        if (out_f32) *(f32x4*)(out_f32 + roff + c) = o;
        if (out_t) store4<T>(out_t + roff + c, o);
    }
}

// LayerNorm(act(sum_p parts[p])) -> fp32 and/or T copies; without w / b: act(sum_p parts[p]) alone, in T storage.
// The consumer of split-K partials of a GEMM whose activation needs the full sum: predictor convs (Conv1d -> ReLU ->
// LayerNorm, network/variance_adaptor.py:72-75) and the first FFN conv (ReLU, network/encoder.py:28-30) in the latency
// regime (fs2_conv_gemm_ksplit_act).  One wave per row; the partials are read four at a time (independent loads).
template <typename T>
__global__ __launch_bounds__(256) void sum_act_layernorm_kernel(const float* __restrict__ parts, int nparts,
                                                                long long part_stride, int relu,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ out_f32, T* __restrict__ out_t, int M, int C,
                                                                float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cv = C >> 2;
    const size_t roff = (size_t)row * C;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            const float* src = parts + roff + j * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            int p = 0;
            for (; p + 4 <= nparts; p += 4) {
                const f32x4 a0 = *(const f32x4*)(src + (size_t)p * part_stride);
                const f32x4 a1 = *(const f32x4*)(src + (size_t)(p + 1) * part_stride);
                const f32x4 a2 = *(const f32x4*)(src + (size_t)(p + 2) * part_stride);
                const f32x4 a3 = *(const f32x4*)(src + (size_t)(p + 3) * part_stride);
                t += (a0 + a1) + (a2 + a3);
            }
            for (; p < nparts; ++p) t += *(const f32x4*)(src + (size_t)p * part_stride);
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = fmaxf(t[r], 0.f);
            }
            v[k] = t;
            if (!w) {
                if (out_f32) *(f32x4*)(out_f32 + roff + j * 4) = t;
                if (out_t) store4<T>(out_t + roff + j * 4, t);
            }
            s += (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
    if (!w) return;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = v[k][r] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            const f32x4 ww = *(const f32x4*)(w + j * 4);
            const f32x4 bb = *(const f32x4*)(b + j * 4);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * ww[r] + bb[r];
            if (out_f32) *(f32x4*)(out_f32 + roff + j * 4) = o;
            if (out_t) store4<T>(out_t + roff + j * 4, o);
        }
    }
}

// act(sum_p parts[p]) alone (no LayerNorm): one lane per four channels, up to eight partial loads in flight
template <typename T>
__global__ __launch_bounds__(256) void sum_act_kernel(const float* __restrict__ parts, int nparts, long long part_stride, int relu,
                                                      float* __restrict__ out_f32, T* __restrict__ out_t, long long n4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const float* src = parts + idx * 4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 4 <= nparts; p += 4) {       // same pairing as sum_act_layernorm_kernel
        const f32x4 a0 = *(const f32x4*)(src + (size_t)p * part_stride);
        const f32x4 a1 = *(const f32x4*)(src + (size_t)(p + 1) * part_stride);
        const f32x4 a2 = *(const f32x4*)(src + (size_t)(p + 2) * part_stride);
        const f32x4 a3 = *(const f32x4*)(src + (size_t)(p + 3) * part_stride);
        t += (a0 + a1) + (a2 + a3);
    }
    for (; p < nparts; ++p) t += *(const f32x4*)(src + (size_t)p * part_stride);
    if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(t[r], 0.f);
    }
    if (out_f32) *(f32x4*)(out_f32 + idx * 4) = t;
    if (out_t) store4<T>(out_t + idx * 4, t);
}

// ---------------------------------------------------------------------------------------
// tiny heads: out[m, j] = x[m, :] . W[j, :] + b[j], j < NOUT <= 8, fp32 (predictor `linear`,
// network/variance_adaptor.py:78,:115).  One wave per row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ b, float* __restrict__ out,
                                                     int M, int C, int NOUT) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cv = C >> 2;
    const float* xr = x + (size_t)row * C;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = lane; k < cv; k += 64) {
        const f32x4 xv = *(const f32x4*)(xr + k * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < NOUT) {
                const f32x4 wv = *(const f32x4*)(W + (size_t)j * C + k * 4);
                acc[j] += (xv[0] * wv[0] + xv[1] * wv[1]) + (xv[2] * wv[2] + xv[3] * wv[3]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < NOUT) {
            const float t = wave_sum(acc[j]);
            if (lane == 0) out[(size_t)row * NOUT + j] = t + b[j];
        }
    }
}

// LayerNorm(x) followed by the tiny head, without materialising the LayerNorm output: the last layer of the pitch /
// energy predictors at frame level, where that fp32 matrix (M x 384) was written only to be read once by
// rowdot_kernel.  Same arithmetic in the same order as add_layernorm_kernel + rowdot_kernel.  Inference only (the
// training step keeps the LayerNorm output for the head's weight gradient).
// One wave per LRD_ROWS = 4 consecutive rows, ONE ROW PER 16-LANE GROUP: a lane holds C / 16 elements of its row, so the
// 2 + NOUT reductions of a row are four DPP steps inside the group and the four rows reduce in the same instructions
// (one wave-wide reduction per row and quantity before: 24 seven-step reductions per wave, which bounded the kernel).
// The head's dot products are taken on the raw row against a_h = w o W_h:
//   out_h = rstd (sum x a_h - mean sum a_h) + sum b W_h + hb_h.
constexpr int LRD_ROWS = 4;

// A wave walks its row groups with a grid stride: a_h and the row-independent sums are formed once per wave and stay in
// registers, and the next group's rows are requested before the current group is reduced.
template <int NOUT>
__global__ __launch_bounds__(256) void layernorm_rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ W,
                                                               const float* __restrict__ hb, float* __restrict__ out, int M,
                                                               int C, float eps) {
    const int lane = threadIdx.x & 63, sub = lane & 15, quarter = lane >> 4;
    const int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int ngroups = (M + LRD_ROWS - 1) / LRD_ROWS;
    if (wave_id >= ngroups) return;
    const int cv = C >> 2;                    // f32x4 columns; lane `sub` takes columns sub, sub + 16, ...
    constexpr int NV = 8;                     // C <= 512 (the entry point sends wider rows to the generic form)
    f32x4 a[NOUT][NV];
    float sa[NOUT], sb[NOUT];
#pragma unroll
    for (int h = 0; h < NOUT; ++h) sa[h] = sb[h] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = sub + k * 16;
        const bool ok = j < cv;
        const f32x4 ww = ok ? *(const f32x4*)(w + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 bb = ok ? *(const f32x4*)(b + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NOUT; ++h) {
            const f32x4 wh = ok ? *(const f32x4*)(W + (size_t)h * C + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            a[h][k] = ww * wh;
            sa[h] += (a[h][k][0] + a[h][k][1]) + (a[h][k][2] + a[h][k][3]);
            sb[h] += (bb[0] * wh[0] + bb[1] * wh[1]) + (bb[2] * wh[2] + bb[3] * wh[3]);
        }
    }
#pragma unroll
    for (int h = 0; h < NOUT; ++h) {
        sa[h] = row16_sum(sa[h]);
        sb[h] = row16_sum(sb[h]) + hb[h];
    }
    auto load = [&](int grp, f32x4 (&v)[NV]) __attribute__((always_inline)) {
        const size_t roff = (size_t)min(grp * LRD_ROWS + quarter, M - 1) * C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = sub + k * 16;
            v[k] = j < cv ? *(const f32x4*)(x + roff + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 v[NV], vn[NV];
    load(wave_id, v);
    for (int grp = wave_id; grp < ngroups; grp += nwaves) {
        const bool more = grp + nwaves < ngroups;
        if (more) load(grp + nwaves, vn);
        float s = 0.f, dot[NOUT];
#pragma unroll
        for (int h = 0; h < NOUT; ++h) dot[h] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {        // lanes beyond the row hold zeros
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
#pragma unroll
            for (int h = 0; h < NOUT; ++h)
                dot[h] += (v[k][0] * a[h][k][0] + v[k][1] * a[h][k][1]) + (v[k][2] * a[h][k][2] + v[k][3] * a[h][k][3]);
        }
        const float mean = row16_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (sub + k * 16 < cv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = v[k][e] - mean;
                    q += dlt * dlt;
                }
            }
        const float rstd = 1.0f / sqrtf(row16_sum(q) / (float)C + eps);
        const int row = grp * LRD_ROWS + quarter;
#pragma unroll
        for (int h = 0; h < NOUT; ++h) {
            const float d_ = row16_sum(dot[h]);
            if (sub == 0 && row < M) out[(size_t)row * NOUT + h] = rstd * (d_ - mean * sa[h]) + sb[h];
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = vn[k];
        }
    }
}

// any C <= 1024, any NOUT <= 8: one wave per row
__global__ __launch_bounds__(256) void layernorm_rowdot_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ W,
                                                               const float* __restrict__ hb, float* __restrict__ out, int M,
                                                               int C, int NOUT, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cv = C >> 2;
    const size_t roff = (size_t)row * C;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            v[k] = *(const f32x4*)(x + roff + j * 4);
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = v[k][r] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            const f32x4 ww = *(const f32x4*)(w + j * 4);
            const f32x4 bb = *(const f32x4*)(b + j * 4);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * ww[r] + bb[r];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                if (h < NOUT) {
                    const f32x4 wv = *(const f32x4*)(W + (size_t)h * C + j * 4);
                    acc[h] += (o[0] * wv[0] + o[1] * wv[1]) + (o[2] * wv[2] + o[3] * wv[3]);
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h < NOUT) {
            const float t = wave_sum(acc[h]);
            if (lane == 0) out[(size_t)row * NOUT + h] = t + hb[h];
        }
    }
}

// a8: DurationPredictor.inference rounding — clamp(round(exp(x) - 1), min=0).long()
// (network/variance_adaptor.py:135; torch.round = round-half-to-even = rintf)
__global__ void duration_round_kernel(const float* __restrict__ logd, int* __restrict__ dur, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float v = rintf(expf(logd[m]) - 1.0f);
    v = v > 0.f ? v : 0.f;             // also maps NaN to 0
    v = fminf(v, 1.0e6f);
    dur[m] = (int)v;
}

// ---------------------------------------------------------------------------------------
// a9: LengthRegulator (network/variance_adaptor.py:13-31).
// plan: per utterance — optional duration_control rounding (:18-19), all-zero row -> ones
// (D9, :28-30), inclusive scan; then utterance offsets.  One workgroup; the whole problem is
// a few thousand integers.
// ---------------------------------------------------------------------------------------
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_iscan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(1024) void lr_plan_kernel(const int* __restrict__ dur_in, const int* __restrict__ cu_phone,
                                                       float control, int apply_control, int* __restrict__ dur_out,
                                                       int* __restrict__ cum, int* __restrict__ olens,
                                                       int* __restrict__ cu_frames, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int b = wave; b < B; b += nwaves) {          // one wave per utterance
        const int p0 = cu_phone[b], p1 = cu_phone[b + 1];
        int local = 0;
        for (int p = p0 + lane; p < p1; p += 64) {
            int dv = dur_in[p];
            if (apply_control) dv = (int)rintf((float)dv * control);
            dur_out[p] = dv;
            local += dv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
        const bool all_zero = (local == 0);
        int carry = 0;
        for (int base = p0; base < p1; base += 64) {
            const int p = base + lane;
            int dv = 0;
            if (p < p1) {
                dv = all_zero ? 1 : dur_out[p];
                if (all_zero) dur_out[p] = 1;
            }
            const int sc = wave_iscan(dv, lane);
            if (p < p1) cum[p] = carry + sc;
            carry += __shfl(sc, 63, 64);
        }
        if (lane == 0) olens[b] = all_zero ? (p1 - p0) : local;
    }
    __syncthreads();   // olens written by this block's waves are visible block-wide after the barrier
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base < B; base += 64) {
            const int b = base + lane;
            const int v = b < B ? olens[b] : 0;
            const int sc = wave_iscan(v, lane);
            if (b < B) cu_frames[b] = carry + sc - v;
            carry += __shfl(sc, 63, 64);
        }
        if (lane == 0) cu_frames[B] = carry;
    }
}

// gather: frame t of utterance b copies phone j = #{ p : cum_b[p] <= t } (zero-duration phones
// are skipped).  One wave per LRG_FRAMES consecutive output frames: lane l < LRG_FRAMES runs the two binary searches
// of frame row0 + l (lane-parallel: the ~12 dependent loads are paid once per 16 frames, not per frame) and writes
// the row tables coalesced; then all 64 lanes copy the rows, 16 B per lane, LRG_FRAMES independent loads in flight.
constexpr int LRG_FRAMES = 16;

template <typename T>
__global__ __launch_bounds__(256) void lr_gather_kernel(const float* __restrict__ hs, const int* __restrict__ cum,
                                                        const int* __restrict__ cu_phone,
                                                        const int* __restrict__ cu_frames, int B,
                                                        float* __restrict__ out_f32, T* __restrict__ out_t,
                                                        int2* __restrict__ frame_seg, int* __restrict__ frame_pos,
                                                        int* __restrict__ lr_index, int Mf, int d) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LRG_FRAMES;
    if (row0 >= Mf) return;
    int src = 0;
    {
        const int row = min(row0 + (lane & (LRG_FRAMES - 1)), Mf - 1);
        const int b = find_segment(cu_frames, B, row);
        const int f0 = cu_frames[b];
        const int t = row - f0;
        const int p0 = cu_phone[b], p1 = cu_phone[b + 1];
        // first phone p in [p0, p1) with cum[p] > t
        int lo = p0, hi = p1 - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] > t) hi = mid; else lo = mid + 1;
        }
        src = lo;
        if (lane < LRG_FRAMES && row0 + lane < Mf) {
            frame_seg[row] = make_int2(f0, cu_frames[b + 1]);
            frame_pos[row] = t;
            lr_index[row] = lo - p0;
        }
    }
    const int dv = d >> 2;
    for (int k = lane; k < dv; k += 64) {
#pragma unroll
        for (int r0 = 0; r0 < LRG_FRAMES; r0 += 8) {       // 8 independent row loads in flight per lane
            f32x4 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int sr = __shfl(src, r0 + r, 64);
                v[r] = *(const f32x4*)(hs + (size_t)sr * d + k * 4);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = row0 + r0 + r;
                if (row < Mf) {
                    if (out_f32) *(f32x4*)(out_f32 + (size_t)row * d + k * 4) = v[r];
                    if (out_t) store4<T>(out_t + (size_t)row * d + k * 4, v[r]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// a11 + a12 + decoder a1: hs += pitch_embed(p) + energy_embed(e)  (Conv1d kernel 1 ==
// per-frame affine map, network/variance_adaptor.py:35-51, :203-206), then the decoder's
// positional encoding (network/encoder.py:216-219).
// ---------------------------------------------------------------------------------------
constexpr int VAP_ROWS = 4;     // rows per thread: the embedding weights of its 4 channels stay in registers

template <typename T>
__global__ __launch_bounds__(256) void variance_add_posenc_kernel(const float* __restrict__ hs, const float* __restrict__ pitch,
                                           const float* __restrict__ energy, const float* __restrict__ Wp,
                                           const float* __restrict__ bp, const float* __restrict__ We,
                                           const float* __restrict__ be, float pscale, float escale,
                                           const float* __restrict__ pe, const int* __restrict__ pos, float xscale,
                                           float alpha, const float* __restrict__ alpha_dev,
                                           float* __restrict__ hs_va, float* __restrict__ out_f32,
                                           T* __restrict__ out_t, int M, int d, int pdim, int edim) {
    if (alpha_dev) alpha = *alpha_dev;
    const int dv = d >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int m0 = (int)(idx / dv) * VAP_ROWS, c = (int)(idx % dv) * 4;
    if (m0 >= M) return;
    const f32x4 bpv = *(const f32x4*)(bp + c);
    const f32x4 bev = *(const f32x4*)(be + c);
    const bool model_dims = pdim == 4 && edim == 1;          // the model's shapes (network/fastspeech2.py:135-136)
    f32x4 wp[4], wev = {0.f, 0.f, 0.f, 0.f};
    if (model_dims) {
        wev = *(const f32x4*)(We + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[r] = *(const f32x4*)(Wp + (size_t)(c + r) * 4);
    }
    f32x4 h[VAP_ROWS], pp[VAP_ROWS], pv[VAP_ROWS];
    float ev[VAP_ROWS];
#pragma unroll
    for (int q = 0; q < VAP_ROWS; ++q) {                      // every load of every row is issued before the first use
        const int m = min(m0 + q, M - 1);
        h[q] = *(const f32x4*)(hs + (size_t)m * d + c);
        pp[q] = *(const f32x4*)(pe + (size_t)pos[m] * d + c);
        if (model_dims) {
            pv[q] = *(const f32x4*)(pitch + (size_t)m * 4);
            ev[q] = energy[m] * escale;
        }
    }
#pragma unroll
    for (int q = 0; q < VAP_ROWS; ++q) {
        const int m = m0 + q;
        f32x4 pemb = bpv, eemb = bev;
        if (model_dims) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = bpv[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) t += wp[r][j] * (pv[q][j] * pscale);
                pemb[r] = t;
                eemb[r] = bev[r] + wev[r] * ev[q];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = bpv[r];
                for (int j = 0; j < pdim; ++j) t += Wp[(c + r) * pdim + j] * (pitch[(size_t)min(m, M - 1) * pdim + j] * pscale);
                pemb[r] = t;
                float u = bev[r];
                for (int j = 0; j < edim; ++j) u += We[(c + r) * edim + j] * (energy[(size_t)min(m, M - 1) * edim + j] * escale);
                eemb[r] = u;
            }
        }
        f32x4 x, v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[r] = h[q][r] + (eemb[r] + pemb[r]);
            v[r] = x[r] * xscale + alpha * pp[q][r];
        }
        if (m < M) {
            if (hs_va) *(f32x4*)(hs_va + (size_t)m * d + c) = x;
            if (out_f32) *(f32x4*)(out_f32 + (size_t)m * d + c) = v;
            if (out_t) store4<T>(out_t + (size_t)m * d + c, v);
        }
    }
}

// packed [M, C] -> padded [B, Tmax, C] (zeros beyond each utterance's length)
__global__ void unpack_padded_kernel(const float* __restrict__ src, const int* __restrict__ cu, float* __restrict__ dst,
                                     int B, int Tmax, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Tmax * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long bt = idx / C;
    const int t = (int)(bt % Tmax), b = (int)(bt / Tmax);
    const int len = cu[b + 1] - cu[b];
    dst[idx] = t < len ? src[(size_t)(cu[b] + t) * C + c] : 0.f;
}

// padded [B, Tmax, C] -> packed [M, C]
__global__ void pack_padded_kernel(const float* __restrict__ src, const int2* __restrict__ seg, const int* __restrict__ cu,
                                   float* __restrict__ dst, int B, int Tmax, int C, int M) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * C) return;
    const int c = (int)(idx % C);
    const int m = (int)(idx / C);
    const int b = find_segment(cu, B, m);
    dst[idx] = src[((size_t)b * Tmax + (m - cu[b])) * C + c];
}

// x = dropout(x) in place (fp32), optional T copy — positional dropout (network/encoder.py:128,:145)
template <typename T>
__global__ void dropout_kernel(float* __restrict__ x, T* __restrict__ out_t, int M, int C, Dropout dp) {
    const int cv = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * cv) return;
    const int m = (int)(idx / cv), c = (int)(idx % cv) * 4;
    f32x4 v = *(const f32x4*)(x + (size_t)m * C + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= drop_mul(dp, (uint32_t)m, (uint32_t)(c + r));
    *(f32x4*)(x + (size_t)m * C + c) = v;
    if (out_t) store4<T>(out_t + (size_t)m * C + c, v);
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] = from_f32<T>(src[idx]);
}

// fp32 -> split-fp16 storage (common.h: f16pair_t), four elements per thread
__global__ void cast_f16pair_kernel(const float* __restrict__ src, f16pair_t* __restrict__ dst, long long n4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n4) store4<f16pair_t>(dst + idx * 4, *(const f32x4*)(src + idx * 4));
}

// f16x3 weight operand (csrc/conv_gemm.hip): src (N, C, k) fp32 in nn.Conv1d / nn.Linear layout ->
// dst[t][n][C / 32][2][32] fp16 = per 32-channel chunk 32 x hi then 32 x lo of  w * scale,  hi = fp16(w * scale),
// lo = fp16(w * scale - hi).  One thread per (n, c).
__global__ void pack_weight_f16x3_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int N, int C, int k,
                                         float scale) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * C) return;
    const int n = (int)(idx / C), c = (int)(idx % C);
    const float* sp = src + ((size_t)n * C + c) * k;
    for (int t = 0; t < k; ++t) {
        const float w = sp[t] * scale;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        _Float16* chunk = dst + (((size_t)t * N + n) * (C / 32) + c / 32) * 64;
        chunk[c % 32] = hi;
        chunk[32 + c % 32] = lo;
    }
}

}  // namespace fs2

using namespace fs2;

static inline int blocks_for(long long n, int per) { return (int)((n + per - 1) / per); }

extern "C" int fs2_pack_weight_f16x3(const float* src, void* dst, int N, int C, int k, float scale, hipStream_t stream) {
    FS2_CHECK_ARG(src && dst && N > 0 && C > 0 && C % 32 == 0 && k >= 1, "fs2_pack_weight_f16x3: N=%d C=%d (multiple of 32) k=%d",
                  N, C, k);
    FS2_CHECK_ARG(scale > 0.f, "fs2_pack_weight_f16x3: scale must be a positive power of two");
    hipLaunchKernelGGL(pack_weight_f16x3_kernel, dim3(blocks_for((long long)N * C, 256)), dim3(256), 0, stream, src,
                       (_Float16*)dst, N, C, k, scale);
    FS2_CHECK_LAUNCH("fs2_pack_weight_f16x3");
    return FS2_OK;
}

extern "C" int fs2_build_seg(const int* cu, int B, int* seg, int* pos, int M, hipStream_t stream) {
    FS2_CHECK_ARG(cu && seg && B > 0 && M >= 0, "fs2_build_seg: bad args");
    if (M == 0) return FS2_OK;
    hipLaunchKernelGGL(build_seg_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, stream, cu, B, (int2*)seg, pos, M);
    FS2_CHECK_LAUNCH("fs2_build_seg");
    return FS2_OK;
}

static int embed_posenc_entry(const long long* ids, const int* pos, const float* emb, const float* pe, float xscale,
                              float alpha, const float* alpha_dev, float* out_f32, void* out_t, int M, int d, int idim,
                              int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(ids && pos && emb && pe && (out_f32 || out_t), "fs2_embed_posenc: null pointer");
    FS2_CHECK_ARG(d > 0 && d % 4 == 0 && idim > 0 && M >= 0, "fs2_embed_posenc: d=%d must be a multiple of 4", d);
    if (M == 0) return FS2_OK;
    const long long n = (long long)M * (d / 4);
    if (dtype == 1)
        hipLaunchKernelGGL(embed_posenc_kernel<bf16_t>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, ids, pos, emb, pe,
                           xscale, alpha, alpha_dev, out_f32, (bf16_t*)out_t, M, d, idim);
    else
        hipLaunchKernelGGL(embed_posenc_kernel<float>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, ids, pos, emb, pe,
                           xscale, alpha, alpha_dev, out_f32, (float*)out_t, M, d, idim);
    FS2_CHECK_LAUNCH("fs2_embed_posenc");
    return FS2_OK;
}

extern "C" int fs2_embed_posenc(const long long* ids, const int* pos, const float* emb, const float* pe, float xscale,
                                float alpha, float* out_f32, void* out_t, int M, int d, int idim, int dtype,
                                hipStream_t stream) {
    return embed_posenc_entry(ids, pos, emb, pe, xscale, alpha, nullptr, out_f32, out_t, M, d, idim, dtype, stream);
}

// as fs2_embed_posenc with alpha read from device memory (the nn.Parameter of ScaledPositionalEncoding)
extern "C" int fs2_embed_posenc_dev(const long long* ids, const int* pos, const float* emb, const float* pe, float xscale,
                                    const float* alpha_dev, float* out_f32, void* out_t, int M, int d, int idim, int dtype,
                                    hipStream_t stream) {
    FS2_CHECK_ARG(alpha_dev, "fs2_embed_posenc_dev: null alpha");
    return embed_posenc_entry(ids, pos, emb, pe, xscale, 1.0f, alpha_dev, out_f32, out_t, M, d, idim, dtype, stream);
}

static int add_layernorm_entry(const float* x, const float* parts, int nparts, long long part_stride, const float* w,
                               const float* b, float* sum_out, float* out_f32, void* out_t, int M, int C, float eps,
                               int dtype, Dropout dparts, Dropout dout, hipStream_t stream) {
    FS2_CHECK_ARG((x || nparts > 0) && (nparts == 0 || parts), "fs2_add_layernorm: no input");
    FS2_CHECK_ARG((w && b && (out_f32 || out_t)) || (!w && sum_out), "fs2_add_layernorm: nothing to write");
    FS2_CHECK_ARG(nparts >= 0 && nparts <= 16 && (nparts <= 1 || part_stride >= (long long)M * C),
                  "fs2_add_layernorm: bad partial layout");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV, "fs2_add_layernorm: C=%d unsupported (multiple of 4, <= %d)",
                  C, 256 * LN_MAXV);
    if (M <= 0) return FS2_OK;
    if (dtype == 2) {          // T copy in split-fp16 storage (the GEMM input of the f16x3 phone-level half)
        FS2_CHECK_ARG(!out_t || (C % 32 == 0 && ((uintptr_t)out_t & 127) == 0),
                      "fs2_add_layernorm(f16x3): C=%d must be a multiple of 32 and out_t 128-byte aligned", C);
        hipLaunchKernelGGL(add_layernorm_kernel<f16pair_t>, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, parts, nparts,
                           part_stride, w, b, sum_out, out_f32, (f16pair_t*)out_t, M, C, eps, dparts, dout);
    } else if (dtype == 1) {
        // frame-level launches (bf16 copies, no dropout): four rows per wave.  Alone it is no faster (21.4 / 28.4 us against
        // 19.5 / 28.0 us at 24.7 k x 256 with 1 / 3 partials: the kernel moves 72-78 % of the copy rate either way); beside
        // the GEMMs of the other streams in the throughput loop it is: bench.py A/B/A/B/A/B 1.827-1.830 -> 1.808-1.826 ms
        // per step (exact mode), 1.660-1.678 -> 1.633-1.653 (pure bf16).  FS2_LN_ROWS4=0: the one-row form.
        static const int rows4 = FS2_ENV_INT("FS2_LN_ROWS4", 1);
        const bool fast = rows4 && w && !dparts.thresh && !dout.thresh && (C == 256 || C == 384) && M >= 1024;
        const dim3 g4(blocks_for((M + 3) / 4, 4));
        if (fast && C == 256)
            hipLaunchKernelGGL((add_layernorm_rows4_kernel<bf16_t, 4>), g4, dim3(256), 0, stream, x, parts, nparts, part_stride,
                               w, b, sum_out, out_f32, (bf16_t*)out_t, M, C, eps);
        else if (fast)
            hipLaunchKernelGGL((add_layernorm_rows4_kernel<bf16_t, 6>), g4, dim3(256), 0, stream, x, parts, nparts, part_stride,
                               w, b, sum_out, out_f32, (bf16_t*)out_t, M, C, eps);
        else
            hipLaunchKernelGGL(add_layernorm_kernel<bf16_t>, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, parts, nparts,
                               part_stride, w, b, sum_out, out_f32, (bf16_t*)out_t, M, C, eps, dparts, dout);
    } else
        hipLaunchKernelGGL(add_layernorm_kernel<float>, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, parts, nparts,
                           part_stride, w, b, sum_out, out_f32, (float*)out_t, M, C, eps, dparts, dout);
    FS2_CHECK_LAUNCH("fs2_add_layernorm");
    return FS2_OK;
}

extern "C" int fs2_add_layernorm(const float* x, const float* parts, int nparts, long long part_stride, const float* w,
                                 const float* b, float* sum_out, float* out_f32, void* out_t, int M, int C, float eps,
                                 int dtype, hipStream_t stream) {
    const Dropout off = fs2_make_dropout(0, 0, 0.f);
    return add_layernorm_entry(x, parts, nparts, part_stride, w, b, sum_out, out_f32, out_t, M, C, eps, dtype, off, off,
                               stream);
}

extern "C" int fs2_add_layernorm_dropout(const float* x, const float* parts, int nparts, long long part_stride,
                                         const float* w, const float* b, float* sum_out, float* out_f32, void* out_t, int M,
                                         int C, float eps, int dtype, unsigned seed, unsigned parts_stream,
                                         float parts_rate, unsigned out_stream, float out_rate, hipStream_t stream) {
    FS2_CHECK_ARG(parts_rate >= 0.f && parts_rate < 1.f && out_rate >= 0.f && out_rate < 1.f,
                  "fs2_add_layernorm_dropout: rates must be in [0, 1)");
    return add_layernorm_entry(x, parts, nparts, part_stride, w, b, sum_out, out_f32, out_t, M, C, eps, dtype,
                               fs2_make_dropout(seed, parts_stream, parts_rate), fs2_make_dropout(seed, out_stream, out_rate),
                               stream);
}

extern "C" int fs2_sum_act_layernorm(const float* parts, int nparts, long long part_stride, int relu, const float* w,
                                     const float* b, float* out_f32, void* out_t, int M, int C, float eps, int dtype,
                                     hipStream_t stream) {
    FS2_CHECK_ARG(parts && nparts >= 1 && nparts <= 16 && (nparts == 1 || part_stride >= (long long)M * C) && part_stride % 4 == 0,
                  "fs2_sum_act_layernorm: bad partial layout");
    FS2_CHECK_ARG((w != nullptr) == (b != nullptr) && (out_f32 || out_t), "fs2_sum_act_layernorm: w / b go together; nothing to write");
    FS2_CHECK_ARG(relu == 0 || relu == 1, "fs2_sum_act_layernorm: activation code %d (0 none, 1 relu)", relu);
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV, "fs2_sum_act_layernorm: C=%d unsupported (multiple of 4, <= %d)",
                  C, 256 * LN_MAXV);
    FS2_CHECK_ARG(dtype >= 0 && dtype <= 2, "fs2_sum_act_layernorm: dtype=%d", dtype);
    FS2_CHECK_ARG(dtype != 2 || !out_t || (C % 32 == 0 && ((uintptr_t)out_t & 127) == 0),
                  "fs2_sum_act_layernorm(f16x3): C=%d must be a multiple of 32 and out_t 128-byte aligned", C);
    FS2_CHECK_ARG(((uintptr_t)parts & 15) == 0 && ((uintptr_t)out_f32 & 15) == 0 && ((uintptr_t)out_t & 7) == 0 &&
                      ((uintptr_t)w & 15) == 0 && ((uintptr_t)b & 15) == 0, "fs2_sum_act_layernorm: alignment");
    if (M <= 0) return FS2_OK;
    if (!w) {
        const long long n4 = (long long)M * (C / 4);
        const dim3 ge(blocks_for(n4, 256));
        if (dtype == 2)
            hipLaunchKernelGGL(sum_act_kernel<f16pair_t>, ge, dim3(256), 0, stream, parts, nparts, part_stride, relu, out_f32,
                               (f16pair_t*)out_t, n4);
        else if (dtype == 1)
            hipLaunchKernelGGL(sum_act_kernel<bf16_t>, ge, dim3(256), 0, stream, parts, nparts, part_stride, relu, out_f32,
                               (bf16_t*)out_t, n4);
        else
            hipLaunchKernelGGL(sum_act_kernel<float>, ge, dim3(256), 0, stream, parts, nparts, part_stride, relu, out_f32,
                               (float*)out_t, n4);
        FS2_CHECK_LAUNCH("fs2_sum_act_layernorm");
        return FS2_OK;
    }
    const dim3 g(blocks_for(M, 4));
    if (dtype == 2)
        hipLaunchKernelGGL(sum_act_layernorm_kernel<f16pair_t>, g, dim3(256), 0, stream, parts, nparts, part_stride, relu, w, b,
                           out_f32, (f16pair_t*)out_t, M, C, eps);
    else if (dtype == 1)
        hipLaunchKernelGGL(sum_act_layernorm_kernel<bf16_t>, g, dim3(256), 0, stream, parts, nparts, part_stride, relu, w, b,
                           out_f32, (bf16_t*)out_t, M, C, eps);
    else
        hipLaunchKernelGGL(sum_act_layernorm_kernel<float>, g, dim3(256), 0, stream, parts, nparts, part_stride, relu, w, b,
                           out_f32, (float*)out_t, M, C, eps);
    FS2_CHECK_LAUNCH("fs2_sum_act_layernorm");
    return FS2_OK;
}

extern "C" int fs2_dropout(float* x, void* out_t, int M, int C, unsigned seed, unsigned stream_id, float rate, int dtype,
                           hipStream_t stream) {
    FS2_CHECK_ARG(x && C > 0 && C % 4 == 0 && rate >= 0.f && rate < 1.f, "fs2_dropout: bad args");
    if (M <= 0) return FS2_OK;
    const Dropout dp = fs2_make_dropout(seed, stream_id, rate);
    const long long n = (long long)M * (C / 4);
    if (dtype == 1)
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, x, (bf16_t*)out_t, M, C, dp);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, x, (float*)out_t, M, C, dp);
    FS2_CHECK_LAUNCH("fs2_dropout");
    return FS2_OK;
}

extern "C" int fs2_layernorm(const float* x, const float* w, const float* b, float* out_f32, void* out_t, int M, int C,
                             float eps, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && w && b && (out_f32 || out_t), "fs2_layernorm: null pointer");
    return fs2_add_layernorm(x, nullptr, 0, 0, w, b, nullptr, out_f32, out_t, M, C, eps, dtype, stream);
}

extern "C" int fs2_rowdot(const float* x, const float* W, const float* b, float* out, int M, int C, int nout,
                          hipStream_t stream) {
    FS2_CHECK_ARG(x && W && b && out, "fs2_rowdot: null pointer");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && nout >= 1 && nout <= 8, "fs2_rowdot: C=%d nout=%d unsupported", C, nout);
    if (M <= 0) return FS2_OK;
    hipLaunchKernelGGL(rowdot_kernel, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, W, b, out, M, C, nout);
    FS2_CHECK_LAUNCH("fs2_rowdot");
    return FS2_OK;
}

extern "C" int fs2_duration_round(const float* logd, int* dur, int M, hipStream_t stream) {
    FS2_CHECK_ARG(logd && dur, "fs2_duration_round: null pointer");
    if (M <= 0) return FS2_OK;
    hipLaunchKernelGGL(duration_round_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, stream, logd, dur, M);
    FS2_CHECK_LAUNCH("fs2_duration_round");
    return FS2_OK;
}

extern "C" int fs2_lr_plan(const int* dur_in, const int* cu_phone, float control, int* dur_out, int* cum, int* olens,
                           int* cu_frames, int B, hipStream_t stream) {
    FS2_CHECK_ARG(dur_in && cu_phone && dur_out && cum && olens && cu_frames && B > 0, "fs2_lr_plan: bad args");
    hipLaunchKernelGGL(lr_plan_kernel, dim3(1), dim3(1024), 0, stream, dur_in, cu_phone, control,
                       control != 1.0f ? 1 : 0, dur_out, cum, olens, cu_frames, B);
    FS2_CHECK_LAUNCH("fs2_lr_plan");
    return FS2_OK;
}

// cu[0] = 0, cu[b + 1] = lens[0] + .. + lens[b]: the frame offsets of a sub-batch of a grouped phone phase, rebased to its own
// first frame (one wave; replaces a device-side subtraction the host schedule used to issue through torch)
__global__ __launch_bounds__(64) void cu_from_lens_kernel(const int* __restrict__ lens, int* __restrict__ cu, int n) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
        const int b = base + lane;
        const int v = b < n ? lens[b] : 0;
        const int sc = wave_iscan(v, lane);
        if (b < n) cu[b + 1] = carry + sc;
        carry += __shfl(sc, 63, 64);
    }
    if (lane == 0) cu[0] = 0;
}

extern "C" int fs2_cu_from_lens(const int* lens, int* cu, int n, hipStream_t stream) {
    FS2_CHECK_ARG(lens && cu && n > 0, "fs2_cu_from_lens: bad args");
    hipLaunchKernelGGL(cu_from_lens_kernel, dim3(1), dim3(64), 0, stream, lens, cu, n);
    FS2_CHECK_LAUNCH("fs2_cu_from_lens");
    return FS2_OK;
}

extern "C" int fs2_lr_gather(const float* hs, const int* cum, const int* cu_phone, const int* cu_frames, int B,
                             float* out_f32, void* out_t, int* frame_seg, int* frame_pos, int* lr_index, int Mf, int d,
                             int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(hs && cum && cu_phone && cu_frames && frame_seg && frame_pos && lr_index && (out_f32 || out_t),
                  "fs2_lr_gather: null pointer");
    FS2_CHECK_ARG(B > 0 && d > 0 && d % 4 == 0, "fs2_lr_gather: bad dims");
    FS2_CHECK_ARG(dtype >= 0 && dtype <= 2 && (dtype != 2 || !out_t || (d % 32 == 0 && ((uintptr_t)out_t & 127) == 0)),
                  "fs2_lr_gather: dtype=%d (split-fp16 copy: d a multiple of 32, out_t 128-byte aligned)", dtype);
    if (Mf <= 0) return FS2_OK;
    if (dtype == 2)
        hipLaunchKernelGGL(lr_gather_kernel<f16pair_t>, dim3(blocks_for(Mf, 4 * LRG_FRAMES)), dim3(256), 0, stream, hs, cum, cu_phone,
                           cu_frames, B, out_f32, (f16pair_t*)out_t, (int2*)frame_seg, frame_pos, lr_index, Mf, d);
    else if (dtype == 1)
        hipLaunchKernelGGL(lr_gather_kernel<bf16_t>, dim3(blocks_for(Mf, 4 * LRG_FRAMES)), dim3(256), 0, stream, hs, cum, cu_phone,
                           cu_frames, B, out_f32, (bf16_t*)out_t, (int2*)frame_seg, frame_pos, lr_index, Mf, d);
    else
        hipLaunchKernelGGL(lr_gather_kernel<float>, dim3(blocks_for(Mf, 4 * LRG_FRAMES)), dim3(256), 0, stream, hs, cum, cu_phone,
                           cu_frames, B, out_f32, (float*)out_t, (int2*)frame_seg, frame_pos, lr_index, Mf, d);
    FS2_CHECK_LAUNCH("fs2_lr_gather");
    return FS2_OK;
}

static int variance_add_entry(const float* hs, const float* pitch, const float* energy, const float* Wp,
                              const float* bp, const float* We, const float* be, float pscale, float escale,
                              const float* pe, const int* pos, float xscale, float alpha, const float* alpha_dev,
                              float* hs_va, float* out_f32, void* out_t, int M, int d, int pdim, int edim, int dtype,
                              hipStream_t stream) {
    FS2_CHECK_ARG(hs && pitch && energy && Wp && bp && We && be && pe && pos && (out_f32 || out_t),
                  "fs2_variance_add_posenc: null pointer");
    FS2_CHECK_ARG(d > 0 && d % 4 == 0 && pdim >= 1 && edim >= 1, "fs2_variance_add_posenc: bad dims");
    if (M <= 0) return FS2_OK;
    FS2_CHECK_ARG(dtype >= 0 && dtype <= 2 && (dtype != 2 || !out_t || (d % 32 == 0 && ((uintptr_t)out_t & 127) == 0)),
                  "fs2_variance_add_posenc: dtype=%d (split-fp16 copy: d a multiple of 32, out_t 128-byte aligned)", dtype);
    const long long n = (long long)((M + VAP_ROWS - 1) / VAP_ROWS) * (d / 4);
    if (dtype == 2)
        hipLaunchKernelGGL(variance_add_posenc_kernel<f16pair_t>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, hs, pitch,
                           energy, Wp, bp, We, be, pscale, escale, pe, pos, xscale, alpha, alpha_dev, hs_va, out_f32,
                           (f16pair_t*)out_t, M, d, pdim, edim);
    else if (dtype == 1)
        hipLaunchKernelGGL(variance_add_posenc_kernel<bf16_t>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, hs, pitch,
                           energy, Wp, bp, We, be, pscale, escale, pe, pos, xscale, alpha, alpha_dev, hs_va, out_f32,
                           (bf16_t*)out_t, M, d, pdim, edim);
    else
        hipLaunchKernelGGL(variance_add_posenc_kernel<float>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, hs, pitch,
                           energy, Wp, bp, We, be, pscale, escale, pe, pos, xscale, alpha, alpha_dev, hs_va, out_f32,
                           (float*)out_t, M, d, pdim, edim);
    FS2_CHECK_LAUNCH("fs2_variance_add_posenc");
    return FS2_OK;
}

extern "C" int fs2_variance_add_posenc(const float* hs, const float* pitch, const float* energy, const float* Wp,
                                       const float* bp, const float* We, const float* be, float pscale, float escale,
                                       const float* pe, const int* pos, float xscale, float alpha, float* hs_va,
                                       float* out_f32, void* out_t, int M, int d, int pdim, int edim, int dtype,
                                       hipStream_t stream) {
    return variance_add_entry(hs, pitch, energy, Wp, bp, We, be, pscale, escale, pe, pos, xscale, alpha, nullptr, hs_va,
                              out_f32, out_t, M, d, pdim, edim, dtype, stream);
}

// as fs2_variance_add_posenc with the decoder's alpha read from device memory
extern "C" int fs2_variance_add_posenc_dev(const float* hs, const float* pitch, const float* energy, const float* Wp,
                                           const float* bp, const float* We, const float* be, float pscale, float escale,
                                           const float* pe, const int* pos, float xscale, const float* alpha_dev,
                                           float* hs_va, float* out_f32, void* out_t, int M, int d, int pdim, int edim,
                                           int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(alpha_dev, "fs2_variance_add_posenc_dev: null alpha");
    return variance_add_entry(hs, pitch, energy, Wp, bp, We, be, pscale, escale, pe, pos, xscale, 1.0f, alpha_dev, hs_va,
                              out_f32, out_t, M, d, pdim, edim, dtype, stream);
}

extern "C" int fs2_unpack_padded(const float* src, const int* cu, float* dst, int B, int Tmax, int C,
                                 hipStream_t stream) {
    FS2_CHECK_ARG(src && cu && dst && B > 0 && Tmax >= 0 && C > 0, "fs2_unpack_padded: bad args");
    const long long n = (long long)B * Tmax * C;
    if (n == 0) return FS2_OK;
    hipLaunchKernelGGL(unpack_padded_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream, src, cu, dst, B, Tmax, C);
    FS2_CHECK_LAUNCH("fs2_unpack_padded");
    return FS2_OK;
}

extern "C" int fs2_pack_padded(const float* src, const int* cu, float* dst, int B, int Tmax, int C, int M,
                               hipStream_t stream) {
    FS2_CHECK_ARG(src && cu && dst && B > 0 && C > 0, "fs2_pack_padded: bad args");
    const long long n = (long long)M * C;
    if (n <= 0) return FS2_OK;
    hipLaunchKernelGGL(pack_padded_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream, src, (const int2*)nullptr, cu,
                       dst, B, Tmax, C, M);
    FS2_CHECK_LAUNCH("fs2_pack_padded");
    return FS2_OK;
}

extern "C" int fs2_cast(const float* src, void* dst, long long n, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(src && dst && n >= 0, "fs2_cast: bad args");
    if (n == 0) return FS2_OK;
    if (dtype == 2) {
        FS2_CHECK_ARG(n % 32 == 0 && ((uintptr_t)dst & 127) == 0 && ((uintptr_t)src & 15) == 0,
                      "fs2_cast(f16x3): n must be a multiple of 32, dst 128-byte aligned");
        hipLaunchKernelGGL(cast_f16pair_kernel, dim3(blocks_for(n / 4, 256)), dim3(256), 0, stream, src, (f16pair_t*)dst, n / 4);
    } else if (dtype == 1)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, src, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, src, (float*)dst, n);
    FS2_CHECK_LAUNCH("fs2_cast");
    return FS2_OK;
}

/* out[m, j] = LayerNorm(x[m, :]) . W[j, :] + hb[j]: fs2_layernorm + fs2_rowdot in one pass, the LayerNorm output is not
 * stored (last layer + `linear` of the variance predictors at inference, network/variance_adaptor.py:107-115). */
extern "C" int fs2_layernorm_rowdot(const float* x, const float* w, const float* b, const float* W, const float* hb,
                                    float* out, int M, int C, int nout, float eps, hipStream_t stream) {
    FS2_CHECK_ARG(x && w && b && W && hb && out, "fs2_layernorm_rowdot: null pointer");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV && nout >= 1 && nout <= 8,
                  "fs2_layernorm_rowdot: C=%d nout=%d unsupported", C, nout);
    if (M <= 0) return FS2_OK;
    if (C <= 512 && (nout == 4 || nout == 1)) {        // the model's heads: pitch (4) and energy (1), C = 384
        const long long want = blocks_for(M, 4 * LRD_ROWS);
        // at most 8 waves per CU walking the row groups with a grid stride (256 / 512 / 768 / 1024 / 2048 workgroups at
        // 24.7 k rows: 15.8 / 13.5 / 17.1 / 16.0 / 20.6 us for the 4-wide head, 11.6 / 9.9 / 9.6 / 9.5 / 10.7 us for the 1-wide)
        const dim3 grid((unsigned)(want < 512 ? want : 512));
        if (nout == 4)
            hipLaunchKernelGGL(layernorm_rowdot_kernel<4>, grid, dim3(256), 0, stream, x, w, b, W, hb, out, M, C, eps);
        else
            hipLaunchKernelGGL(layernorm_rowdot_kernel<1>, grid, dim3(256), 0, stream, x, w, b, W, hb, out, M, C, eps);
    } else {
        hipLaunchKernelGGL(layernorm_rowdot_generic_kernel, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, w, b, W, hb, out,
                           M, C, nout, eps);
    }
    FS2_CHECK_LAUNCH("fs2_layernorm_rowdot");
    return FS2_OK;
}

// HBM-bound kernels of the training backward path (gfx950): LayerNorm backward fused with the
// residual-gradient adds, column reductions (bias / LayerNorm-affine / tiny-head weight gradients),
// LengthRegulator backward (segment sum), embedding backward, loss gradients and weight packing.
// What autograd's backward of the reference modules computes (run/train_decode.py:94-97 calls
// loss.backward() on network/fastspeech2.py:232-295) restated over the packed ragged layout.
//
// Built without the SLP vectoriser (-fno-slp-vectorize is among the base flags of fastspeech2_amd/build.py).  With it, hipcc (ROCm 7.2) packs the per-row scalar arithmetic of layernorm_bwd —
// rows taken several at a time — into v_pk_*_f32 instructions with SGPR-pair / op_sel operands, and a wave of that code
// sharing a SIMD with another kernel's v_mfma_f32_16x16x32_bf16 (fs2_conv_wgrad on the other stream of a training step:
// its two waves per SIMD leave room for such a wave; conv_gemm / attention fill the register file) now and then
// computed a whole row wrong: tools/lnb_race.py, 1200-1700 differing outputs in 600 launches with the vectoriser on this
// file, 0 without; with it on conv_wgrad.hip only it made no difference.  DESIGN.md section 4.3.
#include "common.h"
#include "../../include/fs2_pack_desc.h"

namespace fs2 {

constexpr int LNB_MAXV = 4;          // float4 per lane at most -> C <= 1024
#ifndef FS2_LNB_ROWS
#define FS2_LNB_ROWS 16      // training step, A/B/A/B in one gpurun call: 64 rows 8.75 ms, 32 rows 8.23-8.26, 16 rows 8.09-8.10, 8 rows 8.09
#endif
constexpr int LNB_ROWS = FS2_LNB_ROWS;   // rows per workgroup (4 waves x LNB_ROWS / 4 rows)

// g = dy + sum_p parts[p];  LayerNorm backward through x (statistics recomputed from x):
//   xh = (x - mean) * rstd,  h = g * w,  dx = rstd * (h - mean(h) - xh * mean(h * xh))
// (w == null: dx = g);  then dx += dres;  relu_gate: dx = x > 0 ? dx : 0 (x is a ReLU output).
// Per-workgroup partial sums of d gamma = sum g * xh and d beta = sum g go to wb_part[blk][2][C].
// NV = float4 per lane (C <= 256 NV); the dγ/dβ staging in LDS is sized to C (dynamic: 8 C floats), so that the
// kernel's workgroups fit beside the 100 KB conv_wgrad workgroup that shares the CU during a training step.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ parts, int nparts,
                                                            long long part_stride, const float* __restrict__ w,
                                                            const float* __restrict__ dres, int relu_gate,
                                                            float* __restrict__ dx_f32, T* __restrict__ dx_t,
                                                            float* __restrict__ wb_part, int M, int C, float eps,
                                                            Dropout din, Dropout dto) {
    // din: the forward applied dropout to this LayerNorm's OUTPUT (or, with w == null, to the tensor whose gradient
    // is summed here): the incoming gradient g is masked first.  dto: only the T copy of dx is masked — it is the
    // gradient of a dropped-out sub-layer output (residual + dropout(f(x))), while the fp32 dx flows to the residual.
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cv = C >> 2;
    // A wave takes its LNB_ROWS / 4 rows R at a time: the loads of R rows (incoming gradient, partials, x, residual
    // gradient) are requested together and their reductions interleave.  One row at a time, a wave had one row's few
    // 16-byte loads in flight between dependent reductions, and beside conv_wgrad on the other stream of a training
    // step (which fills the memory system) the frame-level launches took 93 us against 25-33 us alone.  Rows are still
    // accumulated into d gamma / d beta in ascending order: the sums are bit-identical to the one-row form's.
    constexpr int R = NV == 1 ? 4 : (NV == 2 ? 2 : 1);
    static_assert((LNB_ROWS / 4) % R == 0, "rows per wave in groups of R");
    f32x4 dgam[NV], dbet[NV], wv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        dgam[k] = dbet[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int j = lane + k * 64;
        wv[k] = (w && j < cv) ? *(const f32x4*)(w + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float invC = 1.0f / (float)C;
    for (int rr = 0; rr < LNB_ROWS / 4; rr += R) {
        const int row0 = blockIdx.x * LNB_ROWS + rr * 4 + wave;
        if (row0 >= M) break;
        int rows[R];
        bool live[R];
        size_t roff[R];
        f32x4 xv[R][NV], gv[R][NV], rv[R][NV];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            rows[i] = row0 + 4 * i;
            live[i] = rows[i] < M;                                   // a missing row reads the last one and writes nothing
            roff[i] = (size_t)(live[i] ? rows[i] : M - 1) * C;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = lane + k * 64;
                const bool ok = j < cv;
                gv[i][k] = (dy && ok) ? *(const f32x4*)(dy + roff[i] + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                xv[i][k] = (x && ok) ? *(const f32x4*)(x + roff[i] + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                rv[i][k] = (dres && ok) ? *(const f32x4*)(dres + roff[i] + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        for (int p = 0; p < nparts; ++p) {
            const float* pp = parts + (size_t)p * part_stride;
            f32x4 t[R][NV];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int j = lane + k * 64;
                    t[i][k] = j < cv ? *(const f32x4*)(pp + roff[i] + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int k = 0; k < NV; ++k) gv[i][k] += t[i][k];
        }
        if (din.thresh) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gv[i][k][r] *= drop_mul(din, (uint32_t)rows[i], (uint32_t)((lane + k * 64) * 4 + r));
        }
        float mean[R], rstd[R], m1[R], m2[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            mean[i] = 0.f; rstd[i] = 1.f; m1[i] = 0.f; m2[i] = 0.f;
        }
        if (w) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) s += (xv[i][k][0] + xv[i][k][1]) + (xv[i][k][2] + xv[i][k][3]);   // lanes beyond the row hold zeros
                mean[i] = wave_sum(s) * invC;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (lane + k * 64 < cv) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float dlt = xv[i][k][r] - mean[i];
                            q += dlt * dlt;
                        }
                    }
                rstd[i] = 1.0f / sqrtf(wave_sum(q) * invC + eps);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (lane + k * 64 < cv) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (xv[i][k][r] - mean[i]) * rstd[i];
                            const float h = gv[i][k][r] * wv[k][r];
                            a1 += h;
                            a2 += h * xh;
                            if (live[i]) {
                                dgam[k][r] += gv[i][k][r] * xh;
                                dbet[k][r] += gv[i][k][r];
                            }
                        }
                    }
                m1[i] = wave_sum(a1) * invC;
                m2[i] = wave_sum(a2) * invC;
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (!live[i]) continue;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = lane + k * 64;
                if (j < cv) {
                    f32x4 o = gv[i][k];
                    if (w) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (xv[i][k][r] - mean[i]) * rstd[i];
                            o[r] = rstd[i] * (gv[i][k][r] * wv[k][r] - m1[i] - xh * m2[i]);
                        }
                    }
                    if (dres) o += rv[i][k];
                    if (relu_gate) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = xv[i][k][r] > 0.f ? o[r] : 0.f;
                    }
                    if (dx_f32) *(f32x4*)(dx_f32 + roff[i] + j * 4) = o;
                    if (dx_t) {
                        if (dto.thresh) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] *= drop_mul(dto, (uint32_t)rows[i], (uint32_t)(j * 4 + r));
                        }
                        store4<T>(dx_t + roff[i] + j * 4, o);
                    }
                }
            }
        }
    }
    if (!wb_part) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + k * 64;
        if (j < cv) {
            *(f32x4*)(red + (wave * 2 + 0) * C + j * 4) = dgam[k];
            *(f32x4*)(red + (wave * 2 + 1) * C + j * 4) = dbet[k];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
        const int which = idx / C, c = idx % C;
        const float* rp = red + which * C + c;
        const float t = (rp[0] + rp[2 * C]) + (rp[4 * C] + rp[6 * C]);
        wb_part[(size_t)blockIdx.x * 2 * C + idx] = t;
    }
}

// ---------------------------------------------------------------------------------------
// column reductions, stage 1:  part[blk][r][c] = sum_{m in block} S[m][r] * G[m][c]  (S null: ones)
// mode 1 (pe): S is ignored, part[blk][0][c] = sum_m G[m][c] * pe[pos[m]][c]  (d alpha of the scaled
// positional encoding, network/encoder.py:142-145).  G is fp32 (g_is_t == 0) or T.
// ---------------------------------------------------------------------------------------
constexpr int CR_ROWS = 32;
constexpr int CR_MAXR = 8;

template <typename T>
__global__ __launch_bounds__(256) void colreduce_kernel(const void* __restrict__ Gv, int g_is_t, int ldg,
                                                        const float* __restrict__ S, int lds_, int R,
                                                        const float* __restrict__ pe, const int* __restrict__ pos,
                                                        float* __restrict__ part, int M, int C) {
    const int m0 = blockIdx.x * CR_ROWS;
    const int m1 = min(M, m0 + CR_ROWS);
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc[CR_MAXR];
#pragma unroll
        for (int r = 0; r < CR_MAXR; ++r) acc[r] = 0.f;
        for (int m = m0; m < m1; ++m) {
            const float gval = g_is_t ? to_f32(((const T*)Gv)[(size_t)m * ldg + c]) : ((const float*)Gv)[(size_t)m * ldg + c];
            if (pe) {
                acc[0] += gval * pe[(size_t)pos[m] * C + c];
            } else if (S) {
#pragma unroll
                for (int r = 0; r < CR_MAXR; ++r)
                    if (r < R) acc[r] += S[(size_t)m * lds_ + r] * gval;
            } else {
                acc[0] += gval;
            }
        }
#pragma unroll
        for (int r = 0; r < CR_MAXR; ++r)
            if (r < R) part[((size_t)blockIdx.x * R + r) * C + c] = acc[r];
    }
}

// stage 2: out[w] = scale * sum_blk part[blk * stride + w] (+ out[w]).  64 columns per workgroup, 16 row
// groups that each sum every 16th partial, combined in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void colreduce_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                int nblk, int W, long long stride, float scale,
                                                                int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int w = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (w < W) {
        // eight independent loads per trip (clamped row, predicated add): the chain of one-at-a-time loads was
        // what this kernel's ~10 us consisted of
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = ty; b < nblk; b += 128) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int bb = b + 16 * j;
                const float v = part[(size_t)(bb < nblk ? bb : nblk - 1) * stride + w];
                acc[j] += bb < nblk ? v : 0.f;
            }
        }
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && w < W) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][tx];
        t *= scale;
        out[w] = accumulate ? out[w] + t : t;
    }
}

// dx[m, c] = sum_j dy[m, j] * W[j, c] — input gradient of the predictors' final nn.Linear
__global__ void rowdot_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx,
                                  int M, int C, int NOUT) {
    const int cv = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * cv) return;
    const int m = (int)(idx / cv), c = (int)(idx % cv) * 4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < NOUT; ++j) {
        const float g = dy[(size_t)m * NOUT + j];
        const f32x4 wv = *(const f32x4*)(W + (size_t)j * C + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += g * wv[r];
    }
    *(f32x4*)(dx + (size_t)m * C + c) = o;
}

// LengthRegulator backward: phone p receives the sum of the gradients of its frames (contiguous:
// frames [cu_frames[b] + cum[p] - dur[p], cu_frames[b] + cum[p])), plus `base[p]` when given.
__device__ __forceinline__ int find_segment_b(const int* __restrict__ cu, int B, int m) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= m) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void lr_scatter_kernel(const float* __restrict__ dframes, const int* __restrict__ dur,
                                                         const int* __restrict__ cum, const int* __restrict__ cu_phone,
                                                         const int* __restrict__ cu_frames, int B,
                                                         const float* __restrict__ base, float scale,
                                                         float* __restrict__ out, int Mp, int d) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= Mp) return;
    const int b = find_segment_b(cu_phone, B, p);
    const int n = dur[p];
    const int f0 = cu_frames[b] + cum[p] - n;
    const int dv = d >> 2;
    for (int k = lane; k < dv; k += 64) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < n; ++f) s += *(const f32x4*)(dframes + (size_t)(f0 + f) * d + k * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] *= scale;
        if (base) s += *(const f32x4*)(base + (size_t)p * d + k * 4);
        *(f32x4*)(out + (size_t)p * d + k * 4) = s;
    }
}

// Embedding backward: demb[v, :] = xscale * sum_{m : ids[m] == v} dx[m, :], row 0 (padding_idx) = 0.
// One workgroup per vocabulary row: the id list is scanned 256 entries at a time, matching rows are
// queued in LDS in index order (deterministic sum), then all threads add those rows.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ dx,
                                                        float xscale, float* __restrict__ demb, int M, int d, int idim,
                                                        int accumulate) {
    __shared__ int rows[256];
    __shared__ int wcount[4];
    const int v = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};       // channels threadIdx.x + 256 j  (d <= 1024)
    if (v != 0) {
        for (int base = 0; base < M; base += 256) {
            const int m = base + threadIdx.x;
            bool hit = false;
            if (m < M) {
                long long id = ids[m];
                id = id < 0 ? 0 : (id >= idim ? idim - 1 : id);
                hit = id == v;
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(bal);
            __syncthreads();
            int off = 0, total = 0;
            for (int w = 0; w < 4; ++w) {
                if (w < wave) off += wcount[w];
                total += wcount[w];
            }
            if (hit) rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = m;
            __syncthreads();
            for (int r = 0; r < total; ++r) {
                const float* src = dx + (size_t)rows[r] * d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = threadIdx.x + 256 * j;
                    if (c < d) acc[j] += src[c];
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < d) {
            float* dst = demb + (size_t)v * d + c;
            const float t = acc[j] * xscale;
            *dst = accumulate ? *dst + t : t;
        }
    }
}

// loss gradients (network/fastspeech2.py:259-279, network/loss.py:5-14), scale = upstream / count:
//   mode 0  L1  : g = scale * sign(pred - tgt)
//   mode 1  MSE : g = 2 scale (pred - tgt)
//   mode 2  duration MSE in the log domain: tgt = log(float(dur) + 1)
// pred/tgt [M, C]; g_f32 [M, C] and/or g_t [M, ldt] (columns C..ldt-1 zero-filled).
template <typename T>
__global__ void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                 const int* __restrict__ dur, int mode, float scale,
                                 const float* __restrict__ scale_dev, float* __restrict__ g_f32, T* __restrict__ g_t,
                                 int M, int C, int ldt) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (scale_dev) scale *= *scale_dev;
    const int W = g_t ? ldt : C;
    if (idx >= (long long)M * W) return;
    const int m = (int)(idx / W), c = (int)(idx % W);
    float g = 0.f;
    if (c < C) {
        const float p = pred[(size_t)m * C + c];
        const float t = mode == 2 ? logf((float)dur[(size_t)m * C + c] + 1.0f) : tgt[(size_t)m * C + c];
        const float dlt = p - t;
        g = mode == 0 ? (dlt > 0.f ? scale : (dlt < 0.f ? -scale : 0.f)) : 2.0f * scale * dlt;
        if (g_f32) g_f32[(size_t)m * C + c] = g;
    }
    if (g_t) g_t[(size_t)m * ldt + c] = from_f32<T>(g);
}

__global__ void scale_kernel(const float* __restrict__ x, float a, float* __restrict__ out, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = a * x[idx];
}

// Weight packing from the nn.Conv1d / nn.Linear parameter layout src (N, C, k) fp32:
//   fwd[t][n][c]      = src[n][c][t]            (operand of the forward implicit GEMM, ld = ld_fwd)
//   bwd[k-1-t][c][n]  = src[n][c][t]            (operand of the input-gradient GEMM: flipped taps,
//                                                transposed channels, row length ld_bwd)
// transposed == 0: threads run over (n, c) with c fastest (coalesced fwd writes);
// transposed == 1: threads run over (c, n) with n fastest (coalesced bwd writes).
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ fwd, T* __restrict__ bwd, int N, int C,
                                   int k, int ld_fwd, int ld_bwd, long long tap_stride_fwd, long long tap_stride_bwd,
                                   int transposed) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * C) return;
    int n, c;
    if (transposed) { c = (int)(idx / N); n = (int)(idx % N); } else { n = (int)(idx / C); c = (int)(idx % C); }
    const float* s = src + ((size_t)n * C + c) * k;
    for (int t = 0; t < k; ++t) {
        const T v = from_f32<T>(s[t]);
        if (transposed)
            bwd[(size_t)(k - 1 - t) * tap_stride_bwd + (size_t)c * ld_bwd + n] = v;
        else
            fwd[(size_t)t * tap_stride_fwd + (size_t)n * ld_fwd + c] = v;
    }
}

// Many weights in one launch: workgroup b belongs to descriptor j with start[j] <= b < start[j + 1] (binary search);
// the first half of a descriptor's workgroups writes the fwd operand, the second half the bwd operand.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const fs2_pack_desc* __restrict__ desc,
                                                                   const int* __restrict__ start, int n) {
    int lo = 0, hi = n;                       // invariant: start[lo] <= blockIdx.x < start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= start[mid]) lo = mid; else hi = mid;
    }
    const fs2_pack_desc d = desc[lo];
    const int nb = (start[lo + 1] - start[lo]) >> 1;
    int b = blockIdx.x - start[lo];
    const int transposed = b >= nb;
    b -= transposed ? nb : 0;
    const long long idx = (long long)b * 256 + threadIdx.x;
    if (idx >= (long long)d.N * d.C) return;
    int nn, c;
    if (transposed) { c = (int)(idx / d.N); nn = (int)(idx % d.N); } else { nn = (int)(idx / d.C); c = (int)(idx % d.C); }
    const float* sp = d.src + ((size_t)nn * d.C + c) * d.k;
    T* fwd = (T*)d.fwd;
    T* bwd = (T*)d.bwd;
    for (int t = 0; t < d.k; ++t) {
        const T v = from_f32<T>(sp[t]);
        if (transposed) {
            if (bwd) bwd[(size_t)(d.k - 1 - t) * d.tap_stride_bwd + (size_t)c * d.ld_bwd + nn] = v;
        } else {
            if (fwd) fwd[(size_t)t * d.tap_stride_fwd + (size_t)nn * d.ld_fwd + c] = v;
        }
    }
}

}  // namespace fs2

using namespace fs2;

static inline int nblk(long long n, int per) { return (int)((n + per - 1) / per); }

extern "C" long long fs2_layernorm_bwd_workspace(int M, int C) {
    return (long long)nblk(M, LNB_ROWS) * 2 * C * (long long)sizeof(float);
}

extern "C" int fs2_layernorm_bwd(const float* x, const float* dy, const float* parts, int nparts, long long part_stride,
                                 const float* w, const float* dres, int relu_gate, float* dx_f32, void* dx_t,
                                 float* dw, float* db, float* workspace, int M, int C, float eps, int accumulate,
                                 int dtype, unsigned seed, unsigned in_stream, float in_rate, unsigned t_stream,
                                 float t_rate, hipStream_t stream) {
    FS2_CHECK_ARG(in_rate >= 0.f && in_rate < 1.f && t_rate >= 0.f && t_rate < 1.f, "fs2_layernorm_bwd: rates in [0, 1)");
    FS2_CHECK_ARG(!(t_rate > 0.f) || (dx_t && (dx_f32 || dtype == 1)),
                  "fs2_layernorm_bwd: a masked T copy needs a separate unmasked fp32 output");
    const Dropout din = fs2_make_dropout(seed, in_stream, in_rate), dto = fs2_make_dropout(seed, t_stream, t_rate);
    FS2_CHECK_ARG((dy || nparts > 0) && (nparts == 0 || parts), "fs2_layernorm_bwd: no incoming gradient");
    FS2_CHECK_ARG(dx_f32 || dx_t, "fs2_layernorm_bwd: nothing to write");
    FS2_CHECK_ARG(!w || (x && dw && db && workspace), "fs2_layernorm_bwd: LayerNorm mode needs x, dw, db and a workspace");
    FS2_CHECK_ARG(!relu_gate || x, "fs2_layernorm_bwd: relu_gate needs x");
    FS2_CHECK_ARG(nparts >= 0 && nparts <= 16 && (nparts <= 1 || part_stride >= (long long)M * C),
                  "fs2_layernorm_bwd: bad partial layout");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * LNB_MAXV, "fs2_layernorm_bwd: C=%d unsupported", C);
    if (M <= 0) {
        if (w && !accumulate) {
            (void)hipMemsetAsync(dw, 0, C * sizeof(float), stream);
            (void)hipMemsetAsync(db, 0, C * sizeof(float), stream);
        }
        return FS2_OK;
    }
    const int blocks = nblk(M, LNB_ROWS);
    float* part = w ? workspace : nullptr;
    const size_t lds = w ? (size_t)8 * C * sizeof(float) : 0;
#define FS2_LNB(TT, NVV)                                                                                             \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NVV>), dim3(blocks), dim3(256), lds, stream, x, dy, parts, nparts, \
                       part_stride, w, dres, relu_gate, dx_f32, (TT*)dx_t, part, M, C, eps, din, dto)
    if (dtype == 1) {
        if (C <= 256) FS2_LNB(bf16_t, 1); else if (C <= 512) FS2_LNB(bf16_t, 2); else FS2_LNB(bf16_t, 4);
    } else {
        if (C <= 256) FS2_LNB(float, 1); else if (C <= 512) FS2_LNB(float, 2); else FS2_LNB(float, 4);
    }
#undef FS2_LNB
    FS2_CHECK_LAUNCH("fs2_layernorm_bwd");
    if (w) {
        // part is [blocks][2][C]: one pass when db sits right behind dw, else two strided passes
        if (db == dw + C) {
            hipLaunchKernelGGL(colreduce_finish_kernel, dim3(nblk(2 * C, 64)), dim3(1024), 0, stream, part, dw, blocks, 2 * C,
                               (long long)2 * C, 1.0f, accumulate);
        } else {
            hipLaunchKernelGGL(colreduce_finish_kernel, dim3(nblk(C, 64)), dim3(1024), 0, stream, part, dw, blocks, C,
                               (long long)2 * C, 1.0f, accumulate);
            hipLaunchKernelGGL(colreduce_finish_kernel, dim3(nblk(C, 64)), dim3(1024), 0, stream, part + C, db, blocks, C,
                               (long long)2 * C, 1.0f, accumulate);
        }
        FS2_CHECK_LAUNCH("fs2_layernorm_bwd(finish)");
    }
    return FS2_OK;
}

extern "C" long long fs2_colreduce_workspace(int M, int C, int R) {
    return (long long)nblk(M > 0 ? M : 1, CR_ROWS) * R * C * (long long)sizeof(float);
}

static int colreduce_launch(const char* name, const void* G, int g_dtype, int ldg, const float* S, int lds, int R,
                            const float* pe, const int* pos, float* out, float* workspace, int M, int C, float scale,
                            int accumulate, hipStream_t stream) {
    FS2_CHECK_ARG(G && out && workspace, "%s: null pointer", name);
    FS2_CHECK_ARG(g_dtype >= 0 && g_dtype <= 2, "%s: g_dtype=%d (0 fp32, 1 bf16)", name, g_dtype);
    FS2_CHECK_ARG(R >= 1 && R <= CR_MAXR && (S || R == 1) && C > 0 && ldg >= C && (!S || lds >= R),
                  "%s: R=%d (1..%d), C=%d, ldg=%d", name, R, CR_MAXR, C, ldg);
    FS2_CHECK_ARG(!pe || (pos && R == 1), "%s: pe mode needs pos and R == 1", name);
    const int blocks = nblk(M > 0 ? M : 1, CR_ROWS);
    if (M <= 0) {
        (void)hipMemsetAsync(workspace, 0, (size_t)R * C * sizeof(float), stream);
    } else if (g_dtype == 1) {
        hipLaunchKernelGGL(colreduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, G, 1, ldg, S, lds, R, pe, pos,
                           workspace, M, C);
    } else {
        hipLaunchKernelGGL(colreduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, G, 0, ldg, S, lds, R, pe, pos,
                           workspace, M, C);
    }
    hipLaunchKernelGGL(colreduce_finish_kernel, dim3(nblk((long long)R * C, 64)), dim3(1024), 0, stream, workspace, out,
                       blocks, R * C, (long long)R * C, scale, accumulate);
    FS2_CHECK_LAUNCH(name);
    return FS2_OK;
}

extern "C" int fs2_colreduce(const void* G, int g_dtype, int ldg, const float* S, int lds, int R, const float* pe,
                             const int* pos, float* out, float* workspace, int M, int C, float scale, int accumulate,
                             hipStream_t stream) {
    return colreduce_launch("fs2_colreduce", G, g_dtype, ldg, S, lds, R, pe, pos, out, workspace, M, C, scale, accumulate,
                            stream);
}

// ---- the entry point that allocates nothing (include/fs2_hip.h: fs2_colreduce_static) ----------------------------
// The stage-1 partials live in a file-scope device array instead of a caller workspace.  `scratch` is only the
// host-side stand-in of that array: its per-device address comes from hipGetSymbolAddress, once per device, and
// that pointer (never the name) is what the kernels and the memset receive.
#define FS2_STATIC_SCRATCH_BYTES (8ll << 20)
static __device__ __align__(256) unsigned char scratch[FS2_STATIC_SCRATCH_BYTES];
static float* g_scratch_ptr[64];          // per device ordinal; written once, read with atomics

static int static_scratch(float** out, hipStream_t stream, bool resolving_allowed) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        fs2_set_error("fs2_colreduce_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return FS2_ERR_HIP;
    }
    float* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        // the first resolution on a device is a runtime query (it may load the code object): not something to do inside
        // a stream capture - callers that capture warm up with fs2_colreduce_static_init()
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (!resolving_allowed || (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)) {
            fs2_set_error("fs2_colreduce_static: the scratch address of device %d is not resolved yet and the stream is "
                          "capturing: call fs2_colreduce_static_init() on this device first", dev);
            return FS2_ERR_HIP;
        }
        void* sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sym) {
            fs2_set_error("fs2_colreduce_static: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
            return FS2_ERR_HIP;
        }
        p = (float*)sym;
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
    *out = p;
    return FS2_OK;
}

extern "C" long long fs2_colreduce_static_max_bytes(void) { return FS2_STATIC_SCRATCH_BYTES; }

// Resolves the scratch array's address on the CURRENT device (once per device and process; idempotent).  After it,
// fs2_colreduce_static on that device is launches only: no runtime query, no allocation - usable under stream capture.
extern "C" int fs2_colreduce_static_init(void) {
    float* ws = nullptr;
    return static_scratch(&ws, nullptr, true);
}

extern "C" int fs2_colreduce_static(const void* G, int g_dtype, int ldg, const float* S, int lds, int R, const float* pe,
                                    const int* pos, float* out, int M, int C, float scale, int accumulate,
                                    hipStream_t stream) {
    FS2_CHECK_ARG(R >= 1 && R <= CR_MAXR && C > 0, "fs2_colreduce_static: R=%d (1..%d), C=%d", R, CR_MAXR, C);
    const long long need = fs2_colreduce_workspace(M, C, R);
    FS2_CHECK_ARG(need <= FS2_STATIC_SCRATCH_BYTES,
                  "fs2_colreduce_static: M=%d C=%d R=%d needs %lld bytes of partials, the static scratch holds %lld; "
                  "use fs2_colreduce with a caller workspace",
                  M, C, R, need, (long long)FS2_STATIC_SCRATCH_BYTES);
    float* ws = nullptr;
    const int rc = static_scratch(&ws, stream, true);
    if (rc != FS2_OK) return rc;
    return colreduce_launch("fs2_colreduce_static", G, g_dtype, ldg, S, lds, R, pe, pos, out, ws, M, C, scale, accumulate,
                            stream);
}

extern "C" int fs2_rowdot_bwd(const float* dy, const float* W, float* dx, int M, int C, int nout, hipStream_t stream) {
    FS2_CHECK_ARG(dy && W && dx, "fs2_rowdot_bwd: null pointer");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && nout >= 1 && nout <= 8, "fs2_rowdot_bwd: C=%d nout=%d unsupported", C, nout);
    if (M <= 0) return FS2_OK;
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(nblk((long long)M * (C / 4), 256)), dim3(256), 0, stream, dy, W, dx, M, C,
                       nout);
    FS2_CHECK_LAUNCH("fs2_rowdot_bwd");
    return FS2_OK;
}

extern "C" int fs2_lr_scatter(const float* dframes, const int* dur, const int* cum, const int* cu_phone,
                              const int* cu_frames, int B, const float* base, float scale, float* out, int Mp, int d,
                              hipStream_t stream) {
    FS2_CHECK_ARG(dframes && dur && cum && cu_phone && cu_frames && out && B > 0, "fs2_lr_scatter: bad args");
    FS2_CHECK_ARG(d > 0 && d % 4 == 0, "fs2_lr_scatter: d=%d must be a multiple of 4", d);
    if (Mp <= 0) return FS2_OK;
    hipLaunchKernelGGL(lr_scatter_kernel, dim3(nblk(Mp, 4)), dim3(256), 0, stream, dframes, dur, cum, cu_phone, cu_frames,
                       B, base, scale, out, Mp, d);
    FS2_CHECK_LAUNCH("fs2_lr_scatter");
    return FS2_OK;
}

extern "C" int fs2_embed_bwd(const long long* ids, const float* dx, float xscale, float* demb, int M, int d, int idim,
                             int accumulate, hipStream_t stream) {
    FS2_CHECK_ARG(ids && dx && demb && d > 0 && d <= 1024 && idim > 0 && M >= 0, "fs2_embed_bwd: bad args (d <= 1024)");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(idim), dim3(256), 0, stream, ids, dx, xscale, demb, M, d, idim, accumulate);
    FS2_CHECK_LAUNCH("fs2_embed_bwd");
    return FS2_OK;
}

extern "C" int fs2_loss_grad(const float* pred, const float* tgt, const int* dur, int mode, float scale,
                             const float* scale_dev, float* g_f32, void* g_t, int M, int C, int ldt, int dtype,
                             hipStream_t stream) {
    FS2_CHECK_ARG(pred && (g_f32 || g_t) && mode >= 0 && mode <= 2, "fs2_loss_grad: bad args");
    FS2_CHECK_ARG((mode == 2 && dur) || (mode != 2 && tgt), "fs2_loss_grad: missing target");
    FS2_CHECK_ARG(C > 0 && (!g_t || ldt >= C), "fs2_loss_grad: ldt=%d < C=%d", ldt, C);
    if (M <= 0) return FS2_OK;
    const long long n = (long long)M * (g_t ? ldt : C);
    if (dtype == 1)
        hipLaunchKernelGGL(loss_grad_kernel<bf16_t>, dim3(nblk(n, 256)), dim3(256), 0, stream, pred, tgt, dur, mode, scale,
                           scale_dev, g_f32, (bf16_t*)g_t, M, C, ldt);
    else
        hipLaunchKernelGGL(loss_grad_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, stream, pred, tgt, dur, mode, scale,
                           scale_dev, g_f32, (float*)g_t, M, C, ldt);
    FS2_CHECK_LAUNCH("fs2_loss_grad");
    return FS2_OK;
}

extern "C" int fs2_scale(const float* x, float a, float* out, long long n, hipStream_t stream) {
    FS2_CHECK_ARG(x && out && n >= 0, "fs2_scale: bad args");
    if (n == 0) return FS2_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(nblk(n, 256)), dim3(256), 0, stream, x, a, out, n);
    FS2_CHECK_LAUNCH("fs2_scale");
    return FS2_OK;
}

extern "C" int fs2_pack_weight(const float* src, void* fwd, void* bwd, int N, int C, int k, int ld_fwd, int ld_bwd,
                               long long tap_stride_fwd, long long tap_stride_bwd, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(src && (fwd || bwd) && N > 0 && C > 0 && k >= 1, "fs2_pack_weight: bad args");
    FS2_CHECK_ARG((!fwd || (ld_fwd >= C && tap_stride_fwd >= (long long)(N - 1) * ld_fwd + C)) &&
                      (!bwd || (ld_bwd >= N && tap_stride_bwd >= (long long)(C - 1) * ld_bwd + N)),
                  "fs2_pack_weight: leading dimensions / tap strides too small");
    const long long n = (long long)N * C;
    for (int tr = 0; tr < 2; ++tr) {
        if ((tr == 0 && !fwd) || (tr == 1 && !bwd)) continue;
        if (dtype == 1)
            hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(nblk(n, 256)), dim3(256), 0, stream, src, (bf16_t*)fwd,
                               (bf16_t*)bwd, N, C, k, ld_fwd, ld_bwd, tap_stride_fwd, tap_stride_bwd, tr);
        else
            hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, stream, src, (float*)fwd,
                               (float*)bwd, N, C, k, ld_fwd, ld_bwd, tap_stride_fwd, tap_stride_bwd, tr);
    }
    FS2_CHECK_LAUNCH("fs2_pack_weight");
    return FS2_OK;
}

// fs2_pack_weight for `n` weights in one launch.  desc and start live in DEVICE memory: desc[j] as in
// include/fs2_pack_desc.h, start[0..n] the running sum of 2 * ceil(N_j * C_j / 256) (workgroups per weight);
// total_blocks = start[n].
extern "C" int fs2_pack_weights(const fs2_pack_desc* desc, const int* start, int n, int total_blocks, int dtype,
                                hipStream_t stream) {
    FS2_CHECK_ARG(desc && start && n > 0 && total_blocks > 0, "fs2_pack_weights: bad args");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1, "fs2_pack_weights: dtype=%d", dtype);
    if (dtype == 1)
        hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, stream, desc, start, n);
    else
        hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, desc, start, n);
    FS2_CHECK_LAUNCH("fs2_pack_weights");
    return FS2_OK;
}

// Varlen multi-head self-attention, flash style (online softmax, no T x T buffer) on MFMA.
//
// Reference semantics (network/encoder.py:64-89) for ONE unpadded utterance:
//   scores = q . (k^T / sqrt(dk));  softmax over keys;  ctx = atten . v
// Padded keys/queries do not exist in the packed layout, so the reference's mask fills
// (:78-82) have nothing to act on; the key tail of the last tile is excluded with -inf.
//
// Work split: one workgroup (4 waves) per (utterance, head, 64-query tile); one wave per 16
// queries.  K and V tiles of 64 keys are staged in LDS.  Orientation is chosen so that every
// softmax statistic is lane-local:
//   S^T = K . Q^T      mma(a = K rows, b = Q rows): lane (g,i) holds keys 4g..4g+3 of each
//                       16-key tile for query i
//   O^T = V^T . P^T    mma(a = V^T,    b = P^T)  : lane (g,i) holds dv 4g..4g+3 of each
//                       16-wide dv tile for the same query i
// The P fragment is used as held (k-slot (g,jj) <-> key 16*(2c + jj/4) + 4g + jj%4); V is read
// with the matching slot map: ds_read_b64_tr_b16 (bf16) or eight ds_read_b32 (f32).
#include "common.h"

namespace fs2 {

constexpr int AT_THREADS = 256;
constexpr int AT_BQ = 64;   // queries per workgroup
constexpr int AT_BK = 64;   // keys per tile

struct AttnArgs {
    const void* qkv;     // [M, ld] T : q | k | v, each [h][dk]
    void* ctx;           // [M, ldo] T
    const int* cu;       // [B+1] utterance row offsets
    int ld, ldo, d;      // d = h * dk
    float scale_log2e;   // log2(e) / sqrt(dk)
};

template <typename T, int DK>
__global__ __launch_bounds__(AT_THREADS) void attention_kernel(AttnArgs a) {
    typedef typename Frag<T>::type frag_t;
    constexpr int ROWB = DK * (int)sizeof(T);        // data bytes per K/V row
    constexpr int STRIDE = ROWB + 16;                // padded LDS row stride
    constexpr int VPR = ROWB / 16;                   // 16-byte vectors per row
    constexpr int KS = DK / 32;                      // fragment steps over dk
    constexpr int DT = DK / 16;                      // 16-wide dv tiles
    constexpr int NV = (AT_BK * VPR) / AT_THREADS;   // vectors per thread per tile
    constexpr int FRAG_BYTES = 8 * (int)sizeof(T);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + AT_BK * STRIDE;

    const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (qt * AT_BQ >= len) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const T* __restrict__ base = (const T*)a.qkv + (size_t)r0 * a.ld;
    const int qcol = h * DK, kcol = a.d + h * DK, vcol = 2 * a.d + h * DK;

    // Q fragments (B operand): Q[q = i][dk = 32 ks + 8 g ..]
    const int q = qt * AT_BQ + wave * 16 + i;
    frag_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (q < len)
            qf[ks] = *(const frag_t*)(base + (size_t)q * a.ld + qcol + ks * 32 + g * 8);
        else
            qf[ks] = frag_zero<T>();
    }

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;

    const int ntiles = (len + AT_BK - 1) / AT_BK;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * AT_BK;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * AT_THREADS;
            const int r = idx / VPR, cv = idx % VPR;
            vec16 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (k0 + r < len) {
                const T* row = base + (size_t)(k0 + r) * a.ld;
                kv = *(const vec16*)(row + kcol + cv * (16 / (int)sizeof(T)));
                vv = *(const vec16*)(row + vcol + cv * (16 / (int)sizeof(T)));
            }
            *(vec16*)(Ks + r * STRIDE + cv * 16) = kv;
            *(vec16*)(Vs + r * STRIDE + cv * 16) = vv;
        }
        __syncthreads();

        // ---- S^T tiles: s[nt][r] = score(key = k0 + 16 nt + 4 g + r, query i) ----------------
        f32x4 s[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t kf = *(const frag_t*)(Ks + (nt * 16 + i) * STRIDE + ks * 32 * (int)sizeof(T) + g * FRAG_BYTES);
                s[nt] = mma16(kf, qf[ks], s[nt]);
            }
        }
        float m_loc = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const float v = key < len ? s[nt][r] * a.scale_log2e : -INFINITY;
                s[nt][r] = v;
                m_loc = fmaxf(m_loc, v);
            }
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64));
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
        const float m_new = fmaxf(m_run, m_loc);       // finite: every tile has >= 1 valid key
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[nt][r] - m_new);
                s[nt][r] = p;
                psum += p;
            }
        l_part = l_part * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;

        // ---- O^T += V^T . P^T over the two 32-key chunks ---------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            frag_t pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = from_f32<T>(s[2 * c][r]);
                pf[4 + r] = from_f32<T>(s[2 * c + 1][r]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                frag_t vf;
                if constexpr (sizeof(T) == 2) {
                    // transposed 4x16 block read: lane 4q+p of a 16-lane group supplies row q, cols 4p..4p+3
                    const unsigned char* p0 = Vs + ((2 * c) * 16 + 4 * g + (i >> 2)) * STRIDE + (dt * 16 + (i & 3) * 4) * 2;
                    const unsigned char* p1 = p0 + 16 * STRIDE;
                    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vf[r] = lo4[r];
                        vf[4 + r] = hi4[r];
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int key = (2 * c + (jj >> 2)) * 16 + 4 * g + (jj & 3);
                        vf[jj] = *(const T*)(Vs + key * STRIDE + (dt * 16 + i) * (int)sizeof(T));
                    }
                }
                o[dt] = mma16(vf, pf, o[dt]);
            }
        }
    }

    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (q < len) {
        T* dst = (T*)a.ctx + (size_t)(r0 + q) * a.ldo + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            T tmp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tmp[r] = from_f32<T>(o[dt][r] * inv);
            if constexpr (sizeof(T) == 2)
                *(uint2*)(dst + dt * 16 + 4 * g) = *(const uint2*)tmp;
            else
                *(f32x4*)(dst + dt * 16 + 4 * g) = *(const f32x4*)tmp;
        }
    }
}

// Debug path for calculate_all_attentions (network/fastspeech2.py:305-341): materialises the
// softmax matrix of one utterance, atten[h][T][T] fp32.  One wave per (head, query) row.
template <typename T>
__global__ __launch_bounds__(256) void attention_dump_kernel(const T* __restrict__ qkv, float* __restrict__ atten,
                                                             int r0, int len, int ld, int d, int dk, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int H = d / dk;
    if (row >= H * len) return;
    const int h = row / len, q = row % len;
    const T* qp = qkv + (size_t)(r0 + q) * ld + h * dk;
    float* out = atten + ((size_t)h * len + q) * len;
    float mx = -INFINITY;
    for (int k = lane; k < len; k += 64) {
        const T* kp = qkv + (size_t)(r0 + k) * ld + d + h * dk;
        float s = 0.f;
        for (int c = 0; c < dk; ++c) s += to_f32(qp[c]) * (to_f32(kp[c]) * scale);
        out[k] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < len; k += 64) {
        const float p = expf(out[k] - mx);
        out[k] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    for (int k = lane; k < len; k += 64) out[k] = out[k] / sum;
}

template <typename T, int DK>
static int launch_attention(const AttnArgs& a, int B, int H, int max_len, hipStream_t stream) {
    constexpr int lds = 2 * AT_BK * (DK * (int)sizeof(T) + 16);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attention_kernel<T, DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    dim3 grid((max_len + AT_BQ - 1) / AT_BQ, H, B);
    hipLaunchKernelGGL((attention_kernel<T, DK>), grid, dim3(AT_THREADS), lds, stream, a);
    FS2_CHECK_LAUNCH("fs2_attention");
    return FS2_OK;
}

}  // namespace fs2

using namespace fs2;

extern "C" int fs2_attention(const void* qkv, void* ctx, const int* cu, int B, int max_len, int n_heads, int dk, int ld,
                             int ldo, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && ctx && cu && B > 0 && n_heads > 0, "fs2_attention: bad args");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1, "fs2_attention: dtype=%d", dtype);
    FS2_CHECK_ARG(dk == 32 || dk == 64 || dk == 96 || dk == 128, "fs2_attention: head dim %d unsupported (32/64/96/128)",
                  dk);
    const int d = n_heads * dk;
    const int epv = dtype == 0 ? 4 : 8;
    FS2_CHECK_ARG(ld >= 3 * d && ldo >= d && ld % epv == 0 && ldo % 4 == 0, "fs2_attention: bad leading dims");
    FS2_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0, "fs2_attention: unaligned pointers");
    if (max_len <= 0) return FS2_OK;
    AttnArgs a;
    a.qkv = qkv; a.ctx = ctx; a.cu = cu; a.ld = ld; a.ldo = ldo; a.d = d;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)dk);
#define FS2_ATT(TT, DKK) return launch_attention<TT, DKK>(a, B, n_heads, max_len, stream)
    if (dtype == 0) {
        if (dk == 32) FS2_ATT(float, 32);
        if (dk == 64) FS2_ATT(float, 64);
        if (dk == 96) FS2_ATT(float, 96);
        FS2_ATT(float, 128);
    } else {
        if (dk == 32) FS2_ATT(bf16_t, 32);
        if (dk == 64) FS2_ATT(bf16_t, 64);
        if (dk == 96) FS2_ATT(bf16_t, 96);
        FS2_ATT(bf16_t, 128);
    }
#undef FS2_ATT
}

extern "C" int fs2_attention_dump(const void* qkv, float* atten, int row0, int len, int n_heads, int dk, int ld,
                                  int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && atten && len >= 0 && n_heads > 0 && dk > 0, "fs2_attention_dump: bad args");
    if (len == 0) return FS2_OK;
    const int rows = n_heads * len;
    const float scale = 1.0f / sqrtf((float)dk);
    if (dtype == 1)
        hipLaunchKernelGGL(attention_dump_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)qkv,
                           atten, row0, len, ld, n_heads * dk, dk, scale);
    else
        hipLaunchKernelGGL(attention_dump_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)qkv,
                           atten, row0, len, ld, n_heads * dk, dk, scale);
    FS2_CHECK_LAUNCH("fs2_attention_dump");
    return FS2_OK;
}

// Backward of the fused varlen multi-head self-attention (attention.hip), flash style: the
// T x T probability matrix is recomputed tile by tile from q, k and the saved per-row
// log-sum-exp, never stored.  What autograd computes for MultiHeadedAttention.forward between the
// projections (network/encoder.py:73-88) when run/train_decode.py:97 calls loss.backward().
//
// With S = q k^T / sqrt(dk), P = softmax(S), O = P v and upstream dO:
//   delta_q = sum_dv dO * O                                   (attn_bwd_prep_kernel)
//   dP = dO v^T,  dS = P o (dP - delta),
//   dq = dS k / sqrt(dk)                                      (attn_bwd_dq_kernel, one wave per 16 queries)
//   dk = dS^T q / sqrt(dk),  dv = P^T dO                      (attn_bwd_dkv_kernel, one wave per 16 keys)
// Two kernels instead of one with atomics: every output element has exactly one writer, so the
// gradients are deterministic.
//
// MFMA orientation (mma16(a, b) = rows(a) x rows(b), lane (g, i) gets rows 4g..4g+3 of column i):
//   dq kernel : S^T = K Q^T and dP^T = V dO^T  -> lane-local query i;  dQ^T += K^T dS^T
//   dkv kernel: S = Q K^T and dP = dO V^T      -> lane-local key i;    dV^T += dO^T P, dK^T += Q^T dS
// The second operand of the last products is the accumulator layout itself (k-slot permutation,
// as in the forward kernel); the first is a TRANSPOSED LDS read of the staged tile.  Tiles that
// are read both ways use the 2*row slot rotation, tiles read row-wise only use the 1*row rotation.
#include "common.h"

namespace fs2 {

constexpr int AB_THREADS = 256;
constexpr int AB_BT = 64;   // rows (keys or queries) per staged tile

struct AttnBwdArgs {
    const void* qkv;     // [M, ld] T : q | k | v, each [h][dk]
    const void* dout;    // [M, ldo] T : upstream gradient of ctx
    const float* lse;    // [H][M] log2-domain log-sum-exp of the scaled scores
    const float* delta;  // [H][M]
    void* dqkv;          // [M, ld] T
    const int* cu;       // [B+1]
    int ld, ldo, d, M;
    float scale_log2e;   // log2(e) / sqrt(dk)
    float scale;         // 1 / sqrt(dk)
    PairOrder order;     // XCD-aware workgroup order (common.h)
    Dropout drop;        // the forward's probability dropout (site = drop.stream + head), recomputed here
};

typedef __attribute__((address_space(3))) void ab_lds_void;
typedef const __attribute__((address_space(1))) void ab_gbl_void;

// delta[h][m] = sum_c dO[m, h dk + c] * O[m, h dk + c]; one wave per (row, head)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                            float* __restrict__ delta, int M, int H, int dk, int ldo) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (long long)M * H) return;
    const int m = (int)(w / H), h = (int)(w % H);
    float s = 0.f;
    for (int c = lane; c < dk; c += 64)
        s += to_f32(o[(size_t)m * ldo + h * dk + c]) * to_f32(dout[(size_t)m * ldo + h * dk + c]);
    s = wave_sum(s);
    if (lane == 0) delta[(size_t)h * M + m] = s;
}

// ---- shared tile helpers -------------------------------------------------------------------
template <typename T, int DK> struct AbTile {
    static constexpr int ROWB = DK * (int)sizeof(T);
    static constexpr int NS = ROWB / 16;
    static constexpr int TILE_BYTES = AB_BT * ROWB;
    static constexpr int PIECES = TILE_BYTES / 1024;
    static constexpr int EPS = 16 / (int)sizeof(T);
    static constexpr int KS = DK / 32;
    static constexpr int DT = DK / 16;
    typedef typename Frag<T>::type frag_t;

    // stage rows [row_begin, row_begin + 64) (clamped to len - 1) of column block `col` into dst with slot rotation ROT * row
    template <int ROT>
    static __device__ __forceinline__ void stage(const T* __restrict__ base, int ld, int col, int row_begin, int len,
                                                 unsigned char* dst, int wave, int lane) {
        for (int p = wave; p < PIECES; p += 4) {
            const int off = p * 1024 + lane * 16;
            const int row = off / ROWB;
            const int slot = (off % ROWB) >> 4;
            int r = row_begin + row;
            r = r < len ? r : len - 1;
            const int s_ = (slot + NS * ROT - ((ROT * row) % NS)) % NS;     // slot holds global chunk (slot - rot)
            __builtin_amdgcn_global_load_lds((ab_gbl_void*)(base + (size_t)r * ld + col + s_ * EPS),
                                             (ab_lds_void*)(dst + p * 1024), 16, 0, 0);
        }
    }

    // row-wise fragment: row `row` of the tile, elements 32 ks + 8 g .. + 7
    template <int ROT>
    static __device__ __forceinline__ frag_t row_frag(const unsigned char* img, int row, int ks, int g) {
        const unsigned char* rowp = img + row * ROWB;
        const int rot = (ROT * row) % NS;
        if constexpr (sizeof(T) == 2) {
            return *(const frag_t*)(rowp + (((ks * 4 + g) + rot) % NS) * 16);
        } else {
            const f32x4 v0 = *(const f32x4*)(rowp + (((ks * 8 + 2 * g) + rot) % NS) * 16);
            const f32x4 v1 = *(const f32x4*)(rowp + (((ks * 8 + 2 * g + 1) + rot) % NS) * 16);
            return (frag_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        }
    }

    // transposed fragment: column (16 dt + i), rows 32 c + {4g..4g+3, 16+4g..16+4g+3}
    template <int ROT>
    static __device__ __forceinline__ frag_t col_frag(const unsigned char* img, int c, int dt, int g, int i) {
        frag_t f;
        if constexpr (sizeof(T) == 2) {
            const int row0 = 32 * c + 4 * g + (i >> 2), row1 = row0 + 16;
            const int slot = dt * 2 + ((i & 3) >> 1);
            const int sub = (i & 1) * 8;
            const unsigned char* p0 = img + row0 * ROWB + ((slot + ROT * row0) % NS) * 16 + sub;
            const unsigned char* p1 = img + row1 * ROWB + ((slot + ROT * row1) % NS) * 16 + sub;
            const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
            const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = lo4[r];
                f[4 + r] = hi4[r];
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = 32 * c + 16 * (jj >> 2) + 4 * g + (jj & 3);
                const int col = dt * 16 + i;
                f[jj] = *(const T*)(img + row * ROWB + (((col >> 2) + ROT * row) % NS) * 16 + (col & 3) * 4);
            }
        }
        return f;
    }
};

// ---- dq: one workgroup per (utterance, head, 64 queries); wave = 16 queries ----------------------
template <typename T, int DK>
__global__ __launch_bounds__(AB_THREADS, 2) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    typedef AbTile<T, DK> TL;
    typedef typename TL::frag_t frag_t;
    constexpr int KS = TL::KS, DT = TL::DT, TILE = TL::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: K image (rotation 2) at smem + s * 2 * TILE, V image (rotation 1) behind it

    int b, h, qblk;
    if (!fs2_pair_order_decode(a.order, b, h, qblk)) return;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (qblk * AB_BT >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const T* __restrict__ base = (const T*)a.qkv + (size_t)r0 * a.ld;
    const T* __restrict__ dob = (const T*)a.dout + (size_t)r0 * a.ldo;
    const int qcol = h * DK, kcol = a.d + h * DK, vcol = 2 * a.d + h * DK;

    const int qrow = qblk * AB_BT + wave * 16 + i;
    const int qc = qrow < len ? qrow : len - 1;
    frag_t qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const frag_t*)(base + (size_t)qc * a.ld + qcol + ks * 32 + g * 8);
        dof[ks] = *(const frag_t*)(dob + (size_t)qc * a.ldo + qcol + ks * 32 + g * 8);
    }
    const float lse_q = a.lse[(size_t)h * a.M + r0 + qc];
    const float del_q = a.delta[(size_t)h * a.M + r0 + qc];

    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Dropout dh = a.drop;
    dh.stream += (uint32_t)h;

    const int ntiles = (len + AB_BT - 1) / AB_BT;
    auto stage = [&](int kt, int st) __attribute__((always_inline)) {
        unsigned char* kd = smem + st * 2 * TILE;
        TL::template stage<2>(base, a.ld, kcol, kt * AB_BT, len, kd, wave, lane);
        TL::template stage<1>(base, a.ld, vcol, kt * AB_BT, len, kd + TILE, wave, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        vm_wait_barrier<0>();      // the tile's LDS-DMA has landed (explicit vmcnt(0)); every wave is done with the other stage
        if (kt + 1 < ntiles) stage(kt + 1, (kt + 1) & 1);
        const unsigned char* Ks = smem + (kt & 1) * 2 * TILE;
        const unsigned char* Vs = Ks + TILE;
        const int k0 = kt * AB_BT;

        f32x4 s[4], dp[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[nt] = dp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t kf = TL::template row_frag<2>(Ks, nt * 16 + i, ks, g);
                const frag_t vf = TL::template row_frag<1>(Vs, nt * 16 + i, ks, g);
                s[nt] = mma16(kf, qf[ks], s[nt]);
                dp[nt] = mma16(vf, dof[ks], dp[nt]);
            }
        }
        // dS^T for query i: keys k0 + 16 nt + 4 g + r
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const bool ok = key < len;
                const float p = ok ? __builtin_amdgcn_exp2f(s[nt][r] * a.scale_log2e - lse_q) : 0.f;
                float dpv = dp[nt][r];
                if (a.drop.thresh) dpv *= drop_mul(dh, (uint32_t)(r0 + qc), (uint32_t)key);   // dP = (dO V^T) o mask
                s[nt][r] = p * (dpv - del_q);
            }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            frag_t dsf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dsf[r] = from_f32<T>(s[2 * c][r]);
                dsf[4 + r] = from_f32<T>(s[2 * c + 1][r]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const frag_t kt_ = TL::template col_frag<2>(Ks, c, dt, g, i);
                dq[dt] = mma16(kt_, dsf, dq[dt]);
            }
        }
    }
    if (qrow < len) {
        T* dst = (T*)a.dqkv + (size_t)(r0 + qrow) * a.ld + qcol;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 v = dq[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.scale;
            store4<T>(dst + dt * 16 + 4 * g, v);
        }
    }
}

// ---- dk, dv: one workgroup per (utterance, head, 64 keys); wave = 16 keys ------------------------
template <typename T, int DK>
__global__ __launch_bounds__(AB_THREADS, 2) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
    typedef AbTile<T, DK> TL;
    typedef typename TL::frag_t frag_t;
    constexpr int KS = TL::KS, DT = TL::DT, TILE = TL::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: Q image at smem + s * STAGE, dO image behind it (both rotation 2: read both ways), then the 64
    // lse and 64 delta values of those query rows.  They come by the same DMA: an ordinary global load inside the
    // loop would make hipcc drain the prefetch (vmcnt(0)) at every use.
    constexpr int STAGE = 2 * TILE + 512;

    int b, h, kblk;
    if (!fs2_pair_order_decode(a.order, b, h, kblk)) return;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (kblk * AB_BT >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const T* __restrict__ base = (const T*)a.qkv + (size_t)r0 * a.ld;
    const T* __restrict__ dob = (const T*)a.dout + (size_t)r0 * a.ldo;
    const int qcol = h * DK, kcol = a.d + h * DK, vcol = 2 * a.d + h * DK;
    const float* __restrict__ lse = a.lse + (size_t)h * a.M + r0;
    const float* __restrict__ delta = a.delta + (size_t)h * a.M + r0;

    const int krow = kblk * AB_BT + wave * 16 + i;
    const int kc = krow < len ? krow : len - 1;
    frag_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *(const frag_t*)(base + (size_t)kc * a.ld + kcol + ks * 32 + g * 8);
        vf[ks] = *(const frag_t*)(base + (size_t)kc * a.ld + vcol + ks * 32 + g * 8);
    }
    f32x4 dk_[DT], dv_[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dk_[dt] = dv_[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Dropout dh = a.drop;
    dh.stream += (uint32_t)h;

    const int ntiles = (len + AB_BT - 1) / AB_BT;
    auto stage = [&](int qt, int st) __attribute__((always_inline)) {
        unsigned char* qd = smem + st * STAGE;
        TL::template stage<2>(base, a.ld, qcol, qt * AB_BT, len, qd, wave, lane);
        TL::template stage<2>(dob, a.ldo, qcol, qt * AB_BT, len, qd + TILE, wave, lane);
        if (wave < 2) {
            int q = qt * AB_BT + lane;
            q = q < len ? q : len - 1;
            __builtin_amdgcn_global_load_lds((ab_gbl_void*)((wave ? delta : lse) + q),
                                             (ab_lds_void*)(qd + 2 * TILE + wave * 256), 4, 0, 0);
        }
    };
    stage(0, 0);
    for (int qt = 0; qt < ntiles; ++qt) {
        vm_wait_barrier<0>();      // the tile's LDS-DMA has landed (explicit vmcnt(0)); every wave is done with the other stage
        if (qt + 1 < ntiles) stage(qt + 1, (qt + 1) & 1);
        const unsigned char* Qs = smem + (qt & 1) * STAGE;
        const unsigned char* Ds = Qs + TILE;
        const float* lse_s = (const float*)(Qs + 2 * TILE);
        const int q0 = qt * AB_BT;

        f32x4 s[4], dp[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[nt] = dp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t qf = TL::template row_frag<2>(Qs, nt * 16 + i, ks, g);
                const frag_t df = TL::template row_frag<2>(Ds, nt * 16 + i, ks, g);
                s[nt] = mma16(qf, kf[ks], s[nt]);       // rows: queries q0 + 16 nt + 4 g + r, column: key i
                dp[nt] = mma16(df, vf[ks], dp[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 lse4 = *(const f32x4*)(lse_s + nt * 16 + 4 * g);
            const f32x4 del4 = *(const f32x4*)(lse_s + 64 + nt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + nt * 16 + 4 * g + r;
                const bool ok = q < len;
                const int qc = ok ? q : len - 1;
                const float p = ok ? __builtin_amdgcn_exp2f(s[nt][r] * a.scale_log2e - lse4[r]) : 0.f;
                const float mul = a.drop.thresh ? drop_mul(dh, (uint32_t)(r0 + qc), (uint32_t)kc) : 1.0f;
                dp[nt][r] = p * (dp[nt][r] * mul - del4[r]);
                s[nt][r] = p * mul;                      // dV uses the dropped probabilities
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            frag_t pf, dsf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = from_f32<T>(s[2 * c][r]);
                pf[4 + r] = from_f32<T>(s[2 * c + 1][r]);
                dsf[r] = from_f32<T>(dp[2 * c][r]);
                dsf[4 + r] = from_f32<T>(dp[2 * c + 1][r]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const frag_t dot = TL::template col_frag<2>(Ds, c, dt, g, i);
                const frag_t qtr = TL::template col_frag<2>(Qs, c, dt, g, i);
                dv_[dt] = mma16(dot, pf, dv_[dt]);
                dk_[dt] = mma16(qtr, dsf, dk_[dt]);
            }
        }
    }
    if (krow < len) {
        T* dst = (T*)a.dqkv + (size_t)(r0 + krow) * a.ld;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 v = dk_[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.scale;
            store4<T>(dst + kcol + dt * 16 + 4 * g, v);
            store4<T>(dst + vcol + dt * 16 + 4 * g, dv_[dt]);
        }
    }
}

template <typename T, int DK>
static int launch_attention_bwd(const AttnBwdArgs& a, int B, int H, int max_len, hipStream_t stream) {
    constexpr int lds = 2 * 2 * AB_BT * DK * (int)sizeof(T);
    constexpr int lds_dkv = lds + 2 * 512;            // + lse / delta rows of both stages
    static unsigned long long ok_dq = 0, ok_dkv = 0;
    fs2_allow_lds((const void*)attn_bwd_dq_kernel<T, DK>, lds, &ok_dq);
    fs2_allow_lds((const void*)attn_bwd_dkv_kernel<T, DK>, lds_dkv, &ok_dkv);
    static const int xcd_on = FS2_ENV_INT("FS2_ATTN_XCD", 1);   // 0: plain grid (A/B runs)
    AttnBwdArgs aa = a;
    const dim3 grid = fs2_pair_order_grid(aa.order, (max_len + AB_BT - 1) / AB_BT, H, B, xcd_on != 0);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DK>), grid, dim3(AB_THREADS), lds, stream, aa);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DK>), grid, dim3(AB_THREADS), lds_dkv, stream, aa);
    FS2_CHECK_LAUNCH("fs2_attention_bwd");
    return FS2_OK;
}

}  // namespace fs2

using namespace fs2;

extern "C" int fs2_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta,
                                 void* dqkv, const int* cu, int B, int max_len, int M, int n_heads, int dk, int ld, int ldo,
                                 int dtype, unsigned seed, unsigned stream_id, float rate, hipStream_t stream) {
    FS2_CHECK_ARG(rate >= 0.f && rate < 1.f, "fs2_attention_bwd: dropout rate must be in [0, 1)");
    FS2_CHECK_ARG(qkv && ctx && dctx && lse && delta && dqkv && cu && B > 0 && n_heads > 0 && M >= 0,
                  "fs2_attention_bwd: bad args");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1, "fs2_attention_bwd: dtype=%d", dtype);
    FS2_CHECK_ARG(dk == 32 || dk == 64 || dk == 96 || dk == 128, "fs2_attention_bwd: head dim %d unsupported (32/64/96/128)",
                  dk);
    const int d = n_heads * dk;
    const int epv = dtype == 0 ? 4 : 8;
    FS2_CHECK_ARG(ld >= 3 * d && ldo >= d && ld % epv == 0 && ldo % epv == 0, "fs2_attention_bwd: bad leading dims");
    FS2_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0 && ((uintptr_t)dctx & 15) == 0 &&
                      ((uintptr_t)dqkv & 15) == 0,
                  "fs2_attention_bwd: unaligned pointers");
    if (max_len <= 0 || M == 0) return FS2_OK;
    const long long waves = (long long)M * n_heads;
    if (dtype == 1)
        hipLaunchKernelGGL(attn_bwd_prep_kernel<bf16_t>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream,
                           (const bf16_t*)ctx, (const bf16_t*)dctx, delta, M, n_heads, dk, ldo);
    else
        hipLaunchKernelGGL(attn_bwd_prep_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream,
                           (const float*)ctx, (const float*)dctx, delta, M, n_heads, dk, ldo);
    AttnBwdArgs a;
    a.qkv = qkv; a.dout = dctx; a.lse = lse; a.delta = delta; a.dqkv = dqkv; a.cu = cu;
    a.ld = ld; a.ldo = ldo; a.d = d; a.M = M;
    a.scale = 1.0f / sqrtf((float)dk);
    a.scale_log2e = 1.4426950408889634f * a.scale;
    a.drop = fs2_make_dropout(seed, stream_id, rate);
#define FS2_ATTB(TT, DKK) return launch_attention_bwd<TT, DKK>(a, B, n_heads, max_len, stream)
    if (dtype == 0) {
        if (dk == 32) FS2_ATTB(float, 32);
        if (dk == 64) FS2_ATTB(float, 64);
        if (dk == 96) FS2_ATTB(float, 96);
        FS2_ATTB(float, 128);
    } else {
        if (dk == 32) FS2_ATTB(bf16_t, 32);
        if (dk == 64) FS2_ATTB(bf16_t, 64);
        if (dk == 96) FS2_ATTB(bf16_t, 96);
        FS2_ATTB(bf16_t, 128);
    }
#undef FS2_ATTB
}

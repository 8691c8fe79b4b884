// Varlen multi-head self-attention, flash style (online softmax, no T x T buffer) on MFMA.
//
// Reference semantics (network/encoder.py:64-89) for ONE unpadded utterance:
//   scores = q . (k^T / sqrt(dk));  softmax over keys;  ctx = atten . v
// Padded keys/queries do not exist in the packed layout, so the reference's mask fills
// (:78-82) have nothing to act on; the key tail of the last tile is excluded with -inf.
//
// Work split: one workgroup (4 waves) per (utterance, head, 64*QT-query block); one wave per
// 16*QT queries (QT = 2 for bf16, 1 for fp32).  K and V tiles of 64 keys are staged in LDS.  Orientation is chosen so that every
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
constexpr int AT_BK = 64;   // keys per tile

struct AttnArgs {
    const void* qkv;     // [M, ld] T : q | k | v, each [h][dk]
    void* ctx;           // [M, ldo] T
    const int* cu;       // [B+1] utterance row offsets
    int ld, ldo, d;      // d = h * dk
    float scale_log2e;   // log2(e) / sqrt(dk)
    float* lse;          // [H][M] or null: log2-domain log-sum-exp of the scaled scores (saved for the backward)
    int M;
    PairOrder order;     // XCD-aware workgroup order (common.h)
    Dropout drop;        // training: dropout of the probabilities (network/encoder.py:86); site = drop.stream + head
    // keys split over `ksplit` workgroups per (utterance, head, query block) (attention32_kernel only): each writes its
    // unnormalised context and (m, l) to the workspace, attention_combine_kernel merges them
    int ksplit;          // 1: no split
    float* part_o;       // [ksplit][M][d] fp32
    float* part_ml;      // [ksplit][H][M][2] fp32: running maximum (log2 domain), sum of exponentials
};

typedef __attribute__((address_space(3))) void at_lds_void;
typedef const __attribute__((address_space(1))) void at_gbl_void;

// K/V tiles (64 keys x DK) live in LDS as unpadded rows whose 16-byte slots are ROTATED by a
// row-dependent amount (K: rot, V: 2 rot) so that the 16-row fragment reads of K and the
// 8-row transposed reads of V spread over all bank groups.  The tiles are filled by LDS-DMA
// (global_load_lds_dwordx4): the LDS side is lane-linear, the rotation is applied to the per-lane
// SOURCE address.  Two stages: tile t+1 is in flight while tile t is consumed (one barrier per tile).
// OT: element type of ctx: T, or f16pair_t over T = float (split-fp16 storage: the context of the f16x3 phone-level half
// goes straight to the output projection's GEMM operand form, common.h store4<f16pair_t>).
template <typename T, int DK, int QT, typename OT = T>
__global__ __launch_bounds__(AT_THREADS, 2) void attention_kernel(AttnArgs a) {
    typedef typename Frag<T>::type frag_t;
    constexpr int ROWB = DK * (int)sizeof(T);        // bytes per K/V row
    constexpr int NS = ROWB / 16;                    // 16-byte slots per row
    constexpr int TILE_BYTES = AT_BK * ROWB;
    constexpr int PIECES = TILE_BYTES / 1024;        // 1-KiB DMA pieces per tile
    constexpr int KS = DK / 32;                      // fragment steps over dk
    constexpr int DT = DK / 16;                      // 16-wide dv tiles
    constexpr int RSH = ROWB >= 256 ? 0 : (ROWB == 128 ? 1 : (ROWB == 64 ? 2 : 0));
    constexpr int EPS = 16 / (int)sizeof(T);         // elements per slot
    constexpr int BQ = 64 * QT;                      // queries per workgroup

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: K image at smem + s * 2 * TILE_BYTES, V image right behind it

    int b, h, qblk;
    if (!fs2_pair_order_decode(a.order, b, h, qblk)) return;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (qblk * BQ >= len) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const T* __restrict__ base = (const T*)a.qkv + (size_t)r0 * a.ld;
    const int qcol = h * DK, kcol = a.d + h * DK, vcol = 2 * a.d + h * DK;

    auto rot_k = [](int row) { return (row >> RSH) % NS; };
    auto rot_v = [](int row) { return ((row >> RSH) * 2) % NS; };

    auto stage = [&](int kt, int st) __attribute__((always_inline)) {
        unsigned char* kdst = smem + st * 2 * TILE_BYTES;
        unsigned char* vdst = kdst + TILE_BYTES;
        const int k0 = kt * AT_BK;
        for (int p = wave; p < PIECES; p += 4) {
            const int off = p * 1024 + lane * 16;
            const int row = off / ROWB;
            const int slot = (off % ROWB) >> 4;
            int key = k0 + row;
            key = key < len ? key : len - 1;       // tail keys are masked (-inf); keep the address valid
            const T* src = base + (size_t)key * a.ld;
            const int ks_ = (slot + NS - rot_k(row)) % NS;   // slot holds global chunk (slot - rot)
            const int vs_ = (slot + NS - rot_v(row)) % NS;
            __builtin_amdgcn_global_load_lds((at_gbl_void*)(src + kcol + ks_ * EPS), (at_lds_void*)(kdst + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((at_gbl_void*)(src + vcol + vs_ * EPS), (at_lds_void*)(vdst + p * 1024), 16, 0, 0);
        }
    };

    // Q fragments (B operand): Q[q = i][dk = 32 ks + 8 g ..]
    frag_t qf[QT][KS];
    int qrow[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qrow[t] = qblk * BQ + (wave * QT + t) * 16 + i;
        const int qc = qrow[t] < len ? qrow[t] : len - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *(const frag_t*)(base + (size_t)qc * a.ld + qcol + ks * 32 + g * 8);
    }

    f32x4 o[QT][DT];
    float m_run[QT], l_part[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -INFINITY;
        l_part[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (len + AT_BK - 1) / AT_BK;
    stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        vm_wait_barrier<0>();   // tile kt landed (explicit vmcnt(0): not left to the compiler) + barrier; every wave is done with the other stage
        if (kt + 1 < ntiles) stage(kt + 1, (kt + 1) & 1);
        const unsigned char* Ks = smem + (kt & 1) * 2 * TILE_BYTES;
        const unsigned char* Vs = Ks + TILE_BYTES;
        const int k0 = kt * AT_BK;

        // ---- S^T tiles: s[t][nt][r] = score(key = k0 + 16 nt + 4 g + r, query i of q-tile t) ------
        f32x4 s[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) s[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t kf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int row = nt * 16 + i;
                const unsigned char* rowp = Ks + row * ROWB;
                if constexpr (sizeof(T) == 2) {
                    kf[nt] = *(const frag_t*)(rowp + (((ks * 4 + g) + rot_k(row)) % NS) * 16);
                } else {
                    const f32x4 v0 = *(const f32x4*)(rowp + (((ks * 8 + 2 * g) + rot_k(row)) % NS) * 16);
                    const f32x4 v1 = *(const f32x4*)(rowp + (((ks * 8 + 2 * g + 1) + rot_k(row)) % NS) * 16);
                    kf[nt] = (frag_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) mma16_multi<4>(kf, qf[t][ks], s[t]);      // four key tiles: four accumulation chains
        }

        // ---- online softmax, per query (lane-local; the 4 lane groups hold different keys) -------
        if ((k0 + AT_BK) > len) {     // last tile only: keys beyond the utterance get -inf (exp2 -> exactly 0)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((k0 + nt * 16 + 4 * g + r) >= len) s[t][nt][r] = -INFINITY;
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float m_loc = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = s[t][nt][r] * a.scale_log2e;     // scale > 0: -inf stays -inf
                    s[t][nt][r] = v;
                    m_loc = fmaxf(m_loc, v);
                }
#ifdef FS2_ATTN_BPERMUTE
            m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64));
            m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
#else
            m_loc = rows_max(m_loc);
#endif
            const float m_new = fmaxf(m_run[t], m_loc);     // finite: every tile has >= 1 valid key
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);   // v_exp_f32: exp2(-inf) = 0
            m_run[t] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[t][nt][r] - m_new);
                    s[t][nt][r] = p;
                    psum += p;
                }
            l_part[t] = l_part[t] * alpha + psum;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
            }
        }

        // ---- O^T += V^T . P^T over the two 32-key chunks ---------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            frag_t pf[QT];
            if (a.drop.thresh) {      // the row sum above used the undropped p: softmax first, dropout after
                Dropout dh = a.drop;
                dh.stream += (uint32_t)h;
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t qg = (uint32_t)(r0 + qrow[t]);
                        s[t][2 * c][r] *= drop_mul(dh, qg, (uint32_t)(k0 + (2 * c) * 16 + 4 * g + r));
                        s[t][2 * c + 1][r] *= drop_mul(dh, qg, (uint32_t)(k0 + (2 * c + 1) * 16 + 4 * g + r));
                    }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[t][r] = from_f32<T>(s[t][2 * c][r]);
                    pf[t][4 + r] = from_f32<T>(s[t][2 * c + 1][r]);
                }
#pragma unroll
            for (int dt0 = 0; dt0 < DT; dt0 += 2) {      // two dv tiles at a time: two accumulation chains per q-tile
                frag_t vf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int dt = dt0 + u;
                    if constexpr (sizeof(T) == 2) {
                        // transposed 4x16 block read: lane 4q+p of a 16-lane group supplies row q, cols 4p..4p+3
                        const int row0 = (2 * c) * 16 + 4 * g + (i >> 2);
                        const int row1 = row0 + 16;
                        const int slot = dt * 2 + ((i & 3) >> 1);
                        const int sub = (i & 1) * 8;
                        const unsigned char* p0 = Vs + row0 * ROWB + ((slot + rot_v(row0)) % NS) * 16 + sub;
                        const unsigned char* p1 = Vs + row1 * ROWB + ((slot + rot_v(row1)) % NS) * 16 + sub;
                        const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
                        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            vf[u][r] = lo4[r];
                            vf[u][4 + r] = hi4[r];
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const int row = (2 * c + (jj >> 2)) * 16 + 4 * g + (jj & 3);
                            const int col = dt * 16 + i;
                            vf[u][jj] = *(const T*)(Vs + row * ROWB + (((col >> 2) + rot_v(row)) % NS) * 16 + (col & 3) * 4);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) mma16_multi<2>(vf, pf[t], &o[t][dt0]);
            }
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = l_part[t];
        l = rows_sum(l);
        const float inv = 1.0f / l;
        if (a.lse && g == 0 && qrow[t] < len) a.lse[(size_t)h * a.M + r0 + qrow[t]] = m_run[t] + __log2f(l);
        if (qrow[t] < len) {
            OT* dst = (OT*)a.ctx + (size_t)(r0 + qrow[t]) * a.ldo + h * DK;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 v = o[t][dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] *= inv;
                    // a split store takes hi / lo of the ROUNDED fp32 product, as fs2_cast of an fp32 context would:
                    // the product must not be contracted into the low half's subtraction
                    if constexpr (sizeof(OT) != sizeof(T) || !__is_same(OT, T)) asm volatile("" : "+v"(v[r]));
                }
                store4<OT>(dst + dt * 16 + 4 * g, v);
            }
        }
    }
}

// ---- bf16: 32x32x16 MFMA, one or two key groups per workgroup --------------------------------------------------
// Same orientation as above (one wave = 32 queries, S^T = K . Q^T, O^T = V^T . P^T), on v_mfma_f32_32x32x16_bf16: it
// keeps the SIMD's vector issue for 8 of its 32 cycles (the 16x16x32 form for 8 of 16), which leaves room for the
// softmax arithmetic.  Lane (hi = lane / 32, lo = lane % 32) owns query lo of the wave's 32 and, of every 32-key
// block, the 16 keys 8 (r / 4) + 4 hi + r % 4 (r = accumulator register): the row statistics are in-lane operations
// plus one v_permlane32_swap.  The P fragment is used as held: of the 16-key step j of a block, k-slot (hi, e) is key
// 16 j + 8 (e / 4) + 4 hi + e % 4, and V^T is read with that map by two ds_read_b64_tr_b16 (4 keys x 16 dv per 16-lane
// group).
// Per tile: barrier, DMA of the next tile issued, all 16 K fragments requested, 16 MFMAs, all 32 V^T fragments
// requested (they land under the softmax arithmetic), softmax, 16 MFMAs.
// G = 2 (512 threads, one workgroup per CU): waves 4..7 take the second half of the utterance's key tiles for the same
// 128 queries, with their own K/V ring; the two partial results (O, m, l) are merged through LDS at the end.  The
// workgroups of a launch then last half as long and there are more of them than CUs, so that utterances of different
// length balance over the chip (with G = 1 every workgroup is resident from the start and the longest utterance sets
// the time), and a single long utterance fills twice the SIMDs.
// G = 4, NW = 2, KB = 1 (512 threads, 64 queries, four key groups of two waves on 32-key tiles, 128 KB of LDS): one
// utterance alone (batch-1 decoding).  A wave's trip over a key tile is a serial chain (K reads, MFMAs, softmax, V^T
// reads, MFMAs: ~2.8 us per 64-key trip at one wave per SIMD) and the launch lasts as long as the longest chain:
// four groups on half-size tiles cut the chain of an 810-frame utterance from 7 trips of 64 keys to 7 of 32, and
// twice the workgroups stand on the chip.
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// both halves of the wave hold the same query in lane % 32
__device__ __forceinline__ float halves_max(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float halves_sum(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    return a + b;
}

// Transposed LDS read issued from inline asm.  Through the builtin, hipcc (ROCm 7.2) makes every such read that follows
// an LDS-DMA in program order wait for vmcnt(0) first (it cannot tell that the DMA fills another buffer): the DMA of
// the next tile would be waited for in the middle of the current one.  The compiler does not track these reads: the
// consumer waits with lds_tr_wait() (lgkmcnt(0)) and takes the registers through it, which orders the uses behind it.
template <int OFF> __device__ __forceinline__ bf16x4 lds_read_tr(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void lds_tr_wait(bf16x4 (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
__device__ __forceinline__ void lds_tr_wait(bf16x4 (&v)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// NW = 2, KB = 1 (128 threads, 64 queries per workgroup, 32-key tiles, 32 KB of LDS at dk = 128: four workgroups per
// CU): the same tile code on half the keys per trip, i.e. twice the work items of half the size; launch_attention uses
// it for short utterances (measurements there and in DESIGN.md section 4.2).
template <int DK, bool DROP, int G, int NW = 4, int KB = 2>
__global__ __launch_bounds__(64 * NW * G, G == 1 ? 2 : 1) void attention32_kernel(AttnArgs a) {
    typedef bf16_t T;
    constexpr int BKT = 32 * KB;                     // keys per tile
    constexpr int ROWB = DK * 2;                     // bytes per K/V row
    constexpr int NS = ROWB / 16;                    // 16-byte slots per row
    constexpr int TILE_BYTES = BKT * ROWB;
    constexpr int PIECES = TILE_BYTES / 1024;        // 1-KiB DMA pieces per tile
    constexpr int KS = DK / 16;                      // k-steps of Q . K over dk
    constexpr int DB = DK / 32;                      // 32-wide dv blocks
    constexpr int RSH = ROWB == 256 ? 0 : (ROWB == 128 ? 1 : (ROWB == 64 ? 2 : 0));
    constexpr bool POW2 = (NS & (NS - 1)) == 0;
    constexpr int BQ = 32 * NW;                      // queries per workgroup
    static_assert(PIECES % NW == 0, "one DMA piece per wave and step");
    static_assert(G == 1 || (G == 2 && NW == 4 && KB == 2) || (G == 4 && NW == 2 && KB == 1),
                  "key groups: two of four waves on 64-key tiles, or four of two waves on 32-key tiles (512 threads)");
    static_assert(G == 1 || G * 4 * TILE_BYTES >= (G - 1) * NW * 64 * (16 * DB + 2) * 4, "the merge buffer reuses the K/V rings");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per key group: K ring at {0, 1} * TILE_BYTES, V ring at {2, 3} * TILE_BYTES of its 4 * TILE_BYTES

    int b, h, qblk;
    if (!fs2_pair_order_decode(a.order, b, h, qblk)) return;
    // key split across workgroups: the grid carries ksplit x the query blocks, block index = qblk * ksplit + split
    const int ksplit = a.ksplit, split = qblk % ksplit;
    qblk /= ksplit;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (qblk * BQ >= len) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = G == 1 ? wave8 : (wave8 % NW), grp = G == 1 ? 0 : (wave8 / NW);   // wave within its key group, key group
    const int hi = lane >> 5, lo = lane & 31;
    const T* __restrict__ base = (const T*)a.qkv + (size_t)r0 * a.ld;
    const int qcol = h * DK, kcol = a.d + h * DK, vcol = 2 * a.d + h * DK;

    // key tiles [t0, t1) of this group; every group runs `trips` iterations (barriers are workgroup-wide).  With a key
    // split the workgroup owns tiles [s0, s1) of the utterance's.
    const int ntiles_all = (len + BKT - 1) / BKT;
    const int per_split = (ntiles_all + ksplit - 1) / ksplit;
    const int s0 = split * per_split;
    const int s1 = (s0 + per_split) < ntiles_all ? (s0 + per_split) : ntiles_all;
    const int hq = lane >> 5, lq = lane & 31;
    if (s0 >= s1) {       // (short utterance: this split has no keys) the merge must see "nothing": l = 0
        const int qr = qblk * BQ + wave * 32 + lq;
        if (grp == 0 && hq == 0 && qr < len) {
            float* ml = a.part_ml + (((size_t)split * (a.d / DK) + h) * a.M + r0 + qr) * 2;
            ml[0] = -INFINITY;
            ml[1] = 0.f;
        }
        return;
    }
    const int ntiles = s1 - s0;
    const int trips = (ntiles + G - 1) / G;
    const int t0 = s0 + grp * trips;
    const int t1 = (t0 + trips) < s1 ? (t0 + trips) : s1;

    // Slot swizzles (XOR, a rotation for the 12-slot rows of dk = 96): K rows are read 16 rows x one 16-byte slot at a
    // time (ds_read_b128 lane groups), V rows 4 rows x 64 bytes at a time (transposed reads, 32 lanes); both then
    // spread over all 64 banks.  Physical slot p of a row holds logical slot swz(p, row).
    auto swz_k = [](int slot, int row) {
        return POW2 ? (slot ^ ((row >> RSH) & (NS - 1))) : (slot + NS - (row % NS)) % NS;
    };
    auto swz_v = [](int slot, int row) {
        return ROWB == 256 ? (slot ^ (4 * (row & 3))) : (ROWB == 128 ? (slot ^ (4 * ((row >> 1) & 1))) : slot);
    };
    auto unswz_k = [](int slot, int row) {      // physical slot of logical slot `slot`
        return POW2 ? (slot ^ ((row >> RSH) & (NS - 1))) : (slot + row) % NS;
    };

    // DMA: one 1-KiB piece per wave and step pp: piece p = wave + NW pp covers rows (p * 1024 + 16 lane) / ROWB.  The
    // source is a 32-bit byte offset from the utterance's first row (scalar base + vector offset), carried from tile to
    // tile: src[v][pp] (v = 0: K, 1: V) advances by BKT rows per call; rows beyond the utterance (last tile only; masked
    // with -inf later) read its last row instead: min() with the same piece's offset in that row.
    constexpr int PP = PIECES / NW;
    // power-of-two rows: piece pp is piece 0 moved down NW * 1024 / ROWB rows, with the same swizzle when that is a
    // multiple of the swizzles' 16-row period (not for two waves at dk = 128: 8 rows)
    constexpr bool ONE = POW2 && ((NW * 1024 / ROWB) % 16 == 0);
    constexpr int LC = ONE ? 1 : PP;
    const unsigned ldb = (unsigned)a.ld * 2u;
    const unsigned tile_step = BKT * ldb;
    const unsigned piece_step = (NW * 1024 / ROWB) * ldb;
    unsigned src[2][LC], src_last[2][LC];
#pragma unroll
    for (int pp = 0; pp < LC; ++pp) {
        const unsigned off = (unsigned)((wave + NW * pp) * 1024 + lane * 16);
        const int row = (int)(off / (unsigned)ROWB);
        const int slot = (int)(off % (unsigned)ROWB) >> 4;
        const unsigned ck = (unsigned)((kcol + swz_k(slot, row) * 8) * 2), cv = (unsigned)((vcol + swz_v(slot, row) * 8) * 2);
        src[0][pp] = (unsigned)(t0 * BKT + row) * ldb + ck;
        src[1][pp] = (unsigned)(t0 * BKT + row) * ldb + cv;
        src_last[0][pp] = (unsigned)(len - 1) * ldb + ck;
        src_last[1][pp] = (unsigned)(len - 1) * ldb + cv;
    }
    auto stage = [&](int v, unsigned char* dst) __attribute__((always_inline)) {      // next tile of stream v
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const unsigned cur = ONE ? src[v][0] + pp * piece_step : src[v][ONE ? 0 : pp];
            const unsigned last = src_last[v][ONE ? 0 : pp];
            const unsigned so = cur < last ? cur : last;
            __builtin_amdgcn_global_load_lds((at_gbl_void*)((const unsigned char*)base + so),
                                             (at_lds_void*)(dst + (wave + NW * pp) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int pp = 0; pp < LC; ++pp) src[v][pp] += tile_step;
    };

    // Q fragments (B operand): Q[query lo][dk = 16 ks + 8 hi ..]
    const int qrow = qblk * BQ + wave * 32 + lo;
    bf16x8 qf[KS];
    {
        const int qc = qrow < len ? qrow : len - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * a.ld + qcol + ks * 16 + hi * 8);
    }

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    // K fragment of k-step ks, key lo of a block: byte lo * ROWB + ((2 ks + hi) ^ x) * 16 = kx ^ (32 ks)
    const int kx0 = lo * ROWB + ((hi ^ ((lo >> RSH) & (NS - 1))) << 4);
    // V^T fragments: 16-lane group gg = lane / 16 reads a 4-key x 16-dv block; lane 4 q + p of the group supplies
    // key q, dv 4 p .. 4 p + 3 and receives the 4 keys of dv (lane % 16).  Keys 4 hi + q (+ 8 for the second read) of a
    // 16-key step, dv block db: byte (4 hi + q) ROWB + (64 db ^ swizzle) + 32 (gg & 1) + 8 p = vx ^ (64 db)
    const int vq = (lane & 15) >> 2, vp = lane & 3, gg1 = (lane >> 4) & 1;
    const int vx0 = (4 * hi + vq) * ROWB + gg1 * 32 + vp * 8 +
                    (ROWB == 256 ? 64 * vq : (ROWB == 128 ? 64 * ((vq >> 1) & 1) : 0));

    // one key tile: s = K . Q^T; online softmax; O^T += V^T . P^T
    auto tile = [&](int kt, const unsigned char* Ks, const unsigned char* Vs) __attribute__((always_inline)) {
        const int k0 = kt * BKT;
        f32x16 s[KB];     // s[kb][r] = score(key 32 kb + 8 (r / 4) + 4 hi + r % 4, query lo)
        {
            int kx = kx0;
            asm volatile("" : "+v"(kx));      // the per-step addresses are rebuilt per tile, not carried through the loop
            // all K fragments of the tile are requested before the first MFMA; left to itself the compiler keeps two
            // reads in flight and exposes the LDS latency once per MFMA pair
            bf16x8 kf[KB][KS];
            // (reads and MFMAs alternate between the two key blocks: two accumulation chains, no back-to-back
            // dependent MFMAs)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    // row 32 kb + lo; a power-of-two swizzle depends on lo alone (32 kb is a multiple of its period)
                    const int koff = POW2 ? (kx ^ (ks * 32)) : (lo * ROWB + unswz_k(ks * 2 + hi, kb * 32 + lo) * 16);
                    kf[kb][ks] = *(const bf16x8*)(Ks + kb * 32 * ROWB + koff);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) s[kb] = mma32(kf[kb][ks], qf[ks], s[kb]);
        }
        // V^T fragments are requested now: they land under the softmax arithmetic
        bf16x4 vf[DB][4 * KB];      // [db][4 kb + 2 j + half]
        {
            int vx = vx0;
            asm volatile("" : "+v"(vx));
            const unsigned vs = (unsigned)(size_t)(at_lds_void*)Vs;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const unsigned vb = vs + (unsigned)(POW2 ? (vx ^ (db * 64)) : (vx + db * 64));   // dk = 96: unswizzled
                vf[db][0] = lds_read_tr<0 * ROWB>(vb);
                vf[db][1] = lds_read_tr<8 * ROWB>(vb);
                vf[db][2] = lds_read_tr<16 * ROWB>(vb);
                vf[db][3] = lds_read_tr<24 * ROWB>(vb);
                if constexpr (KB == 2) {
                    vf[db][4] = lds_read_tr<32 * ROWB>(vb);
                    vf[db][5] = lds_read_tr<40 * ROWB>(vb);
                    vf[db][6] = lds_read_tr<48 * ROWB>(vb);
                    vf[db][7] = lds_read_tr<56 * ROWB>(vb);
                }
            }
        }
        if ((k0 + BKT) > len) {     // last tile only: keys beyond the utterance get -inf (exp2 -> exactly 0)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((k0 + kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3)) >= len) s[kb][r] = -INFINITY;
        }
        float m_loc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_loc = fmaxf(m_loc, s[kb][r]);
        m_loc = halves_max(m_loc) * a.scale_log2e;                  // scale > 0: the maximum commutes with it
        const float m_new = fmaxf(m_run, m_loc);                    // finite: every tile has >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // v_exp_f32: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], a.scale_log2e, -m_new));
                s[kb][r] = p;
                psum += p;
            }
        l_part = l_part * alpha + psum;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        if constexpr (DROP) {     // the row sum above used the undropped p: softmax first, dropout after
            Dropout dh = a.drop;
            dh.stream += (uint32_t)h;
            const uint32_t qg = (uint32_t)(r0 + qrow);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[kb][r] *= drop_mul(dh, qg, (uint32_t)(k0 + kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3)));
        }
        bf16x8 pf[KB][2];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[kb][j][e] = (bf16_t)s[kb][8 * j + e];
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            lds_tr_wait(vf[db]);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x8 v8;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v8[r] = vf[db][4 * kb + 2 * j][r];
                        v8[4 + r] = vf[db][4 * kb + 2 * j + 1][r];
                    }
                    o[db] = mma32(v8, pf[kb][j], o[db]);
                }
        }
    };

    unsigned char* const Kr = smem + grp * 4 * TILE_BYTES;
    unsigned char* const Vr = Kr + 2 * TILE_BYTES;
    if (t0 < t1) {
        stage(0, Kr);
        stage(1, Vr);
    }
    for (int it = 0; it < trips; ++it) {
        // This trip's tile has landed and every wave is done with the previous one.  The wait for the DMA is explicit:
        // with the transposed reads in inline asm hipcc no longer sees an LDS read that could alias the DMA's target and
        // emitted `s_waitcnt vmcnt(0)` once, in front of the loop - the tile was read unwaited (caught as run-to-run
        // differences of the decode loop, tools/determinism_check.py; the 16x16x32 kernel relies on the compiler's wait).
        vm_wait_barrier<0>();
        const int kt = t0 + it;
        const int st = it & 1;
        if (kt + 1 < t1) {
            stage(0, Kr + (st ^ 1) * TILE_BYTES);
            stage(1, Vr + (st ^ 1) * TILE_BYTES);
        }
        if (kt < t1) tile(kt, Kr + st * TILE_BYTES, Vr + st * TILE_BYTES);
    }

    float l = halves_sum(l_part);
    if constexpr (G > 1) {
        // merge of the key groups: groups 1 .. G - 1 leave (O, m, l) in LDS in their own register / lane order, the wave
        // of group 0 with the same queries (same wave index, same lanes) folds them in, in group order.  (A group without
        // tiles - fewer tiles than groups - leaves m = -inf, l = 0, O = 0: its factor is exp2(-inf) = 0.)
        constexpr int MBW = (16 * DB + 2) * 64;      // floats per wave
        __syncthreads();
        if (grp > 0) {
            float* mb = (float*)smem + ((grp - 1) * NW + wave) * MBW + lane;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) mb[(db * 16 + r) * 64] = o[db][r];
            mb[(16 * DB) * 64] = m_run;
            mb[(16 * DB + 1) * 64] = l;
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll 1
        for (int gsrc = 1; gsrc < G; ++gsrc) {     // (rolled: unrolled, the loads of all groups are hoisted and spill)
            const float* mb = (const float*)smem + ((gsrc - 1) * NW + wave) * MBW + lane;
            const float m_b = mb[(16 * DB) * 64], l_b = mb[(16 * DB + 1) * 64];
            const float m_new = fmaxf(m_run, m_b);                       // group 0 always has a tile: finite
            const float fa = __builtin_amdgcn_exp2f(m_run - m_new), fb = __builtin_amdgcn_exp2f(m_b - m_new);
            m_run = m_new;
            l = l * fa + l_b * fb;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] = o[db][r] * fa + mb[(db * 16 + r) * 64] * fb;
        }
    }
    if (ksplit > 1) {     // partial result of this key range: unnormalised context, running maximum, sum
        if (qrow < len) {
            if (hi == 0) {
                float* ml = a.part_ml + (((size_t)split * (a.d / DK) + h) * a.M + r0 + qrow) * 2;
                ml[0] = m_run;
                ml[1] = l;
            }
            float* dst = a.part_o + ((size_t)split * a.M + r0 + qrow) * a.d + h * DK;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = o[db][4 * rr + r];
                    *(f32x4*)(dst + db * 32 + 8 * rr + 4 * hi) = v;
                }
        }
        return;
    }
    const float inv = 1.0f / l;
    if (qrow < len) {
        if (a.lse && hi == 0) a.lse[(size_t)h * a.M + r0 + qrow] = m_run + __log2f(l);
        T* dst = (T*)a.ctx + (size_t)(r0 + qrow) * a.ldo + h * DK;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[db][4 * rr + r] * inv;
                store4<T>(dst + db * 32 + 8 * rr + 4 * hi, v);
            }
    }
}

// ---- f16x3: q | k | v in split-fp16 storage, every product on the fp16 matrix pipe -----------------------------
// The phone-level half in its f16x3 mode (dtype code 3): the QKV projection writes q | k | v in split-fp16 storage
// (common.h: per 32-channel run 32 hi halves, then 32 lo halves; hi + lo carries 22 of fp32's 24 significant bits) and
// this kernel forms both products of the attention as three fp16 MFMAs each,
//   S^T = Kh.Qh + Kh.Ql + Kl.Qh          O^T = Vh.Ph + Vh.Pl + Vl.Ph,
// instead of eight exact-fp32 MFMAs of twice the cycles (attention_kernel<float>: 16 384 MFMA cycles per wave and 64-key
// tile, here 1 536).  The probabilities are split in registers after a multiplication by 2^12 (p <= 1: the high half
// stays in range, the low half stays a normal fp16 number down to p ~ 1e-5; fp32 accumulation makes the scale exact and
// the final 1 / l takes it back out).  Statistics, softmax and accumulation in fp32 as in the other kernels; the context
// is written in split-fp16 storage for the output projection.  Same orientation and k-slot maps as attention_kernel:
// one wave = 16 queries, S^T tiles of 16 keys, V^T read with ds_read_b64_tr_b16 (hi and lo halves apart).
// LDS images: rows of DK * 4 bytes; physical 16-byte slot = logical slot ^ (row & 15) for K (16-row fragment reads of one
// slot), ^ 2 (row & 7) for V (8 rows x 32 bytes per LDS cycle of the transposed reads): both conflict-free; the DMA
// applies the same XOR to its per-lane source slot.  BK keys per tile (64: 128 KB of LDS at dk 128, one workgroup per
// CU; 32: 64 KB, two).
typedef _Float16 at_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mma16h(const at_f16x8& a, const at_f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// Eight transposed reads (four addresses, each at +0 and +OFF bytes) AND their wait in ONE asm statement: the results are
// valid when the statement ends, so nothing the compiler does with the registers afterwards (copies, spills) can see
// them early - attention32_kernel issues and waits in separate statements to overlap more and leans on tools/isa_check.py
// rule (c) for that.  "=&v": the outputs must not share registers with the addresses still to be used.
template <int OFF>
__device__ __forceinline__ void lds_read_tr8_h(at_f16x4 (&v)[8], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                 "ds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %2, %9\n\t"
                 "ds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %4, %10\n\t"
                 "ds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %6, %11\n\t"
                 "ds_read_b64_tr_b16 %7, %11 offset:%12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF)
                 : "memory");
}

constexpr float AT_PSCALE = 4096.0f;

template <int DK, int BK>
__global__ __launch_bounds__(AT_THREADS, 2) void attention_split_kernel(AttnArgs a) {
    static_assert(DK == 64 || DK == 128, "split-fp16 attention: head dims whose rows are a power-of-two number of slots");
    static_assert(BK == 32 || BK == 64, "keys per tile");
    constexpr int ROWB = DK * 4;                     // bytes per K/V row (split-fp16 storage: 4 bytes per element)
    constexpr int TILE_BYTES = BK * ROWB;
    constexpr int PIECES = TILE_BYTES / 1024;        // 1-KiB DMA pieces per tile
    constexpr int PP = PIECES / 4;                   // per wave
    constexpr int RPP = 4096 / ROWB;                 // rows between a wave's consecutive pieces
    constexpr int KS = DK / 32;                      // 32-channel runs = fragment steps over dk
    constexpr int DT = DK / 16;                      // 16-wide dv tiles
    constexpr int NT = BK / 16;                      // 16-key score tiles per key tile
    constexpr int NC = BK / 32;                      // 32-key chunks of the P.V product
    constexpr int BQ = 64;                           // queries per workgroup

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: K image at smem + s * 2 * TILE_BYTES, V image right behind it

    int b, h, qblk;
    if (!fs2_pair_order_decode(a.order, b, h, qblk)) return;
    // key split across workgroups (as attention32_kernel): block index = qblk * ksplit + split
    const int ksplit = a.ksplit, split = qblk % ksplit;
    qblk /= ksplit;
    const int r0 = a.cu[b];
    const int len = a.cu[b + 1] - r0;
    if (qblk * BQ >= len) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int ntiles_all = (len + BK - 1) / BK;
    const int per_split = (ntiles_all + ksplit - 1) / ksplit;
    const int s0 = split * per_split;
    const int s1 = (s0 + per_split) < ntiles_all ? (s0 + per_split) : ntiles_all;
    if (s0 >= s1) {       // this split has no keys: the merge must see l = 0
        const int qr = qblk * BQ + wave * 16 + i;
        if (g == 0 && qr < len) {
            float* ml = a.part_ml + (((size_t)split * (a.d / DK) + h) * a.M + r0 + qr) * 2;
            ml[0] = -INFINITY;
            ml[1] = 0.f;
        }
        return;
    }
    const unsigned ldb = (unsigned)a.ld * 4u;
    const unsigned char* __restrict__ base = (const unsigned char*)a.qkv + (size_t)r0 * ldb;
    const unsigned qcolb = (unsigned)(h * DK) * 4u, kcolb = (unsigned)(a.d + h * DK) * 4u, vcolb = (unsigned)(2 * a.d + h * DK) * 4u;

    // DMA: piece p = wave + 4 pp of an image covers bytes p * 1024 .. of it; this lane's 16 bytes are physical slot
    // `lslot` of row lrow + RPP * pp
    const int loff = wave * 1024 + lane * 16;
    const int lrow = loff / ROWB, lslot = (loff % ROWB) >> 4;
    auto stage = [&](int kt, int st) __attribute__((always_inline)) {
        unsigned char* kdst = smem + st * 2 * TILE_BYTES;
        unsigned char* vdst = kdst + TILE_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const int row = lrow + RPP * pp;
            int key = k0 + row;
            key = key < len ? key : len - 1;       // tail keys are masked (-inf); keep the address valid
            const unsigned char* src = base + (size_t)key * ldb;
            const unsigned ks_ = (unsigned)(lslot ^ (row & 15)) << 4, vs_ = (unsigned)(lslot ^ (2 * (row & 7))) << 4;
            __builtin_amdgcn_global_load_lds((at_gbl_void*)(src + kcolb + ks_), (at_lds_void*)(kdst + (wave + 4 * pp) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((at_gbl_void*)(src + vcolb + vs_), (at_lds_void*)(vdst + (wave + 4 * pp) * 1024), 16, 0, 0);
        }
    };

    // Q fragments (B operand): Q[q = i][dk = 32 ks + 8 g ..], hi and lo halves
    const int qrow = qblk * BQ + wave * 16 + i;
    at_f16x8 qh[KS], ql[KS];
    {
        const int qc = qrow < len ? qrow : len - 1;
        const unsigned char* qb = base + (size_t)qc * ldb + qcolb + g * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = *(const at_f16x8*)(qb + ks * 128);
            ql[ks] = *(const at_f16x8*)(qb + ks * 128 + 64);
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;

    // K fragment (A operand: key i of a 16-key tile, dk 32 ks + 8 g ..): logical slot 8 ks + 4 half + g, physical
    // slot = logical ^ i: byte kx ^ (128 ks + 64 half) of the tile's first row
    const int kx = i * ROWB + ((g ^ i) << 4);
    // V^T fragments: lane 4 q + p of a 16-lane group supplies key row 4 g + q, dv 4 p .. 4 p + 3 of a 16-wide dv tile and
    // receives the four keys of dv (lane % 16): logical slot 8 (dt / 2) + 4 half + 2 (dt % 2) + p / 2, physical slot =
    // logical ^ 2 (row & 7)
    const int vrow = 4 * g + (i >> 2);
    const int vx = vrow * ROWB + ((((i & 3) >> 1) ^ (2 * (vrow & 7))) << 4) + (i & 1) * 8;

    stage(s0, 0);
    for (int kt = s0; kt < s1; ++kt) {
        vm_wait_barrier<0>();   // tile kt landed (explicit vmcnt(0)) + barrier: every wave is done with the other stage
        if (kt + 1 < s1) stage(kt + 1, (kt + 1 - s0) & 1);
        const unsigned char* Ks = smem + ((kt - s0) & 1) * 2 * TILE_BYTES;
        const unsigned vs = (unsigned)(size_t)(at_lds_void*)(Ks + TILE_BYTES);
        const int k0 = kt * BK;

        // ---- S^T tiles: s[nt][r] = score(key = k0 + 16 nt + 4 g + r, query i) -------------------------------
        f32x4 s[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            at_f16x8 kh[NT], kl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                kh[nt] = *(const at_f16x8*)(Ks + nt * 16 * ROWB + (kx ^ (ks * 128)));
                kl[nt] = *(const at_f16x8*)(Ks + nt * 16 * ROWB + (kx ^ (ks * 128 + 64)));
            }
            // small terms first, one term over all tiles at a time (independent accumulators back to back)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s[nt] = mma16h(kl[nt], qh[ks], s[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s[nt] = mma16h(kh[nt], ql[ks], s[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s[nt] = mma16h(kh[nt], qh[ks], s[nt]);
        }

        // ---- online softmax, per query (lane-local; the 4 lane groups hold different keys) -------------------
        if ((k0 + BK) > len) {     // last tile only: keys beyond the utterance get -inf (exp2 -> exactly 0)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((k0 + nt * 16 + 4 * g + r) >= len) s[nt][r] = -INFINITY;
        }
        float m_loc = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = s[nt][r] * a.scale_log2e;     // scale > 0: -inf stays -inf
                s[nt][r] = v;
                m_loc = fmaxf(m_loc, v);
            }
        m_loc = rows_max(m_loc);
        const float m_new = fmaxf(m_run, m_loc);     // finite: every tile has >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // v_exp_f32: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[nt][r] - m_new);
                s[nt][r] = p;
                psum += p;
            }
        l_part = l_part * alpha + psum;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        }

        // ---- O^T += V^T . P^T over the 32-key chunks; k-slot (g, jj) <-> key 32 c + 16 (jj / 4) + 4 g + jj % 4 ---
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            at_f16x8 ph, pl;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float p = s[2 * c + (jj >> 2)][jj & 3] * AT_PSCALE;
                const _Float16 hh = (_Float16)p;
                ph[jj] = hh;
                pl[jj] = (_Float16)(p - (float)hh);
            }
#pragma unroll
            for (int dt0 = 0; dt0 < DT; dt0 += 2) {      // two dv tiles at a time
                at_f16x4 vf[8];                          // [4 u + 2 half + w]: dv tile dt0 + u, hi / lo, keys 4 g.. / 16 + 4 g..
                unsigned va[4];                          // [2 u + half]
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        va[2 * u + half] = vs + (unsigned)(c * 32 * ROWB) +
                                           (unsigned)(vx ^ (((((dt0 + u) >> 1) * 8 + half * 4 + ((dt0 + u) & 1) * 2)) << 4));
                lds_read_tr8_h<16 * ROWB>(vf, va[0], va[1], va[2], va[3]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    at_f16x8 vh, vl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vh[r] = vf[4 * u + 0][r];
                        vh[4 + r] = vf[4 * u + 1][r];
                        vl[r] = vf[4 * u + 2][r];
                        vl[4 + r] = vf[4 * u + 3][r];
                    }
                    o[dt0 + u] = mma16h(vl, ph, o[dt0 + u]);
                    o[dt0 + u] = mma16h(vh, pl, o[dt0 + u]);
                    o[dt0 + u] = mma16h(vh, ph, o[dt0 + u]);
                }
            }
        }
    }

    float l = rows_sum(l_part);
    if (ksplit > 1) {     // partial result of this key range (the context still carries AT_PSCALE: the merge divides it out)
        if (qrow < len) {
            if (g == 0) {
                float* ml = a.part_ml + (((size_t)split * (a.d / DK) + h) * a.M + r0 + qrow) * 2;
                ml[0] = m_run;
                ml[1] = l;
            }
            float* dst = a.part_o + ((size_t)split * a.M + r0 + qrow) * a.d + h * DK;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *(f32x4*)(dst + dt * 16 + 4 * g) = o[dt];
        }
        return;
    }
    const float inv = 1.0f / (l * AT_PSCALE);
    if (qrow < len) {
        f16pair_t* dst = (f16pair_t*)a.ctx + (size_t)(r0 + qrow) * a.ldo + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 v = o[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] *= inv;
                asm volatile("" : "+v"(v[r]));      // the rounded product is what gets split (as fs2_cast would)
            }
            store4<f16pair_t>(dst + dt * 16 + 4 * g, v);
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

// Merge of the key splits of attention32_kernel: ctx[m][c] = sum_s f_s O_s[m][c] / sum_s f_s l_s with f_s = exp2(m_s - max m)
// over the splits that saw keys (l_s > 0; every row has at least one).  One lane per (row, four channels).
template <typename T>
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                T* __restrict__ ctx, int M, int d, int dk, int ldo, int ksplit,
                                                                float oscale) {
    const int dv = d >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * dv) return;
    const int m = (int)(idx / dv), c = (int)(idx % dv) * 4, h = c / dk, H = d / dk;
    float ms[8], ls[8];
    float mx = -INFINITY;
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        ms[sp] = -INFINITY; ls[sp] = 0.f;
        if (sp < ksplit) {
            const float* ml = part_ml + (((size_t)sp * H + h) * M + m) * 2;
            ms[sp] = ml[0]; ls[sp] = ml[1];
            if (ls[sp] > 0.f) mx = fmaxf(mx, ms[sp]);
        }
    }
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        if (sp < ksplit && ls[sp] > 0.f) {
            const float f = __builtin_amdgcn_exp2f(ms[sp] - mx);
            L += ls[sp] * f;
            const f32x4 o = *(const f32x4*)(part_o + ((size_t)sp * M + m) * d + c);
            acc += o * f;
        }
    }
    const float inv = oscale / L;          // oscale: 1, or 1 / AT_PSCALE for the split-fp16 kernel's scaled probabilities
    f32x4 v = acc * inv;
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(v[r]));      // the rounded product is what gets stored / split
    store4<T>(ctx + (size_t)m * ldo + c, v);
}

template <typename T, int DK, typename OT = T>
static int launch_attention(const AttnArgs& a, int B, int H, int max_len, hipStream_t stream) {
    constexpr int QT = sizeof(T) == 2 ? 2 : 1;     // 32 (bf16) / 16 (f32) queries per wave
    constexpr int lds = 2 * 2 * AT_BK * DK * (int)sizeof(T);
    static unsigned long long lds_ok = 0;
    fs2_allow_lds((const void*)attention_kernel<T, DK, QT, OT>, lds, &lds_ok);
    static const int xcd_on = FS2_ENV_INT("FS2_ATTN_XCD", 1);   // 0: plain grid (A/B runs)
    AttnArgs aa = a;
    const dim3 grid = fs2_pair_order_grid(aa.order, (max_len + 64 * QT - 1) / (64 * QT), H, B, xcd_on != 0);
    if constexpr (sizeof(T) == 2) {
        static const int v1 = FS2_ENV_INT("FS2_ATTN_V1", 0);   // 1: the 16x16x32 kernel (A/B runs)
        static const int groups = FS2_ENV_INT("FS2_ATTN_GROUPS", 0);   // 1 / 2: forced
        if (!v1) {
            // two key groups per workgroup when one-group workgroups would leave CUs empty (measured: 66 workgroups,
            // config 4 B = 1, 86 -> 71 us; 264 or 448 workgroups are 10 % faster with one group, two per CU)
            const long items = (long)B * H * ((max_len + 127) / 128);
            const long items64 = (long)B * H * ((max_len + 63) / 64);
            if (aa.ksplit > 1) {
                // keys split over ksplit workgroups per work item (fs2_attention_ksplit): the four-group form while it
                // stands on the chip one workgroup per CU, else the plain form; then the merge
                const bool four_s = max_len > 4 * AT_BK && items64 * aa.ksplit <= 256;
                static unsigned long long oks[2] = {0, 0};
                const void* ks_k = four_s ? (const void*)attention32_kernel<DK, false, 4, 2, 1> : (const void*)attention32_kernel<DK, false, 1>;
                const int lds_s = four_s ? 4 * (lds / 2) : lds;
                fs2_allow_lds(ks_k, lds_s, &oks[four_s ? 1 : 0]);
                const int nblk = four_s ? (max_len + 63) / 64 : (max_len + 127) / 128;
                const dim3 grid_s = fs2_pair_order_grid(aa.order, nblk * aa.ksplit, H, B, xcd_on != 0);
                void* params[] = {&aa};
                if (hipLaunchKernel(ks_k, grid_s, dim3(four_s ? 512 : AT_THREADS), params, lds_s, stream) != hipSuccess) {
                    fs2_set_error("fs2_attention: launch failed: %s", hipGetErrorString(hipGetLastError()));
                    return FS2_ERR_HIP;
                }
                const long long n4 = (long long)aa.M * (aa.d / 4);
                hipLaunchKernelGGL(attention_combine_kernel<bf16_t>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                                   aa.part_o, aa.part_ml, (bf16_t*)aa.ctx, aa.M, aa.d, DK, aa.ldo, aa.ksplit, 1.0f);
                FS2_CHECK_LAUNCH("fs2_attention(combine)");
                return FS2_OK;
            }
            // four key groups (64-query workgroups) while those still stand on the chip one per CU: a single utterance
            const bool four = groups ? groups == 4 : (max_len > 4 * AT_BK && items64 <= 256 && !aa.drop.thresh);
            if (four) {
                static unsigned long long ok4 = 0;
                const void* k4 = (const void*)attention32_kernel<DK, false, 4, 2, 1>;
                const int lds4 = 4 * (lds / 2);
                fs2_allow_lds(k4, lds4, &ok4);
                const dim3 grid4 = fs2_pair_order_grid(aa.order, (max_len + 63) / 64, H, B, xcd_on != 0);
                void* params[] = {&aa};
                if (hipLaunchKernel(k4, grid4, dim3(512), params, lds4, stream) != hipSuccess) {
                    fs2_set_error("fs2_attention: launch failed: %s", hipGetErrorString(hipGetLastError()));
                    return FS2_ERR_HIP;
                }
                return FS2_OK;
            }
            const bool two = groups ? groups == 2 : (max_len > 4 * AT_BK && items <= 256);
            // 64-query workgroups on 32-key tiles for short utterances (phone level: <= 256 keys, i.e. one or two
            // 128-query items per pair): twice the workgroups of half the size, 9.0 -> 7.7 us at 32 x ~120 phones.  At
            // the frame level it loses (config 2: 43.3 vs 39.8 us, uniform 64 x 1024: 644 vs 795-810 TFLOP/s: both forms
            // have every workgroup resident from the start, and 32-key trips pay the per-trip costs twice).
            // FS2_ATTN_Q64 = 0 / 1 forces (A/B runs).
            static const int q64_env = FS2_ENV_INT("FS2_ATTN_Q64", -1);
            const bool q64 = !two && (q64_env >= 0 ? q64_env != 0 : max_len <= 4 * AT_BK);
            static unsigned long long ok[6] = {0, 0, 0, 0, 0, 0};
            const void* k;
            if (aa.drop.thresh)
                k = q64 ? (const void*)attention32_kernel<DK, true, 1, 2, 1>
                        : (two ? (const void*)attention32_kernel<DK, true, 2> : (const void*)attention32_kernel<DK, true, 1>);
            else
                k = q64 ? (const void*)attention32_kernel<DK, false, 1, 2, 1>
                        : (two ? (const void*)attention32_kernel<DK, false, 2> : (const void*)attention32_kernel<DK, false, 1>);
            const int lds32 = q64 ? lds / 2 : (two ? 2 : 1) * lds;
            const int threads = q64 ? 128 : AT_THREADS * (two ? 2 : 1);
            const dim3 grid32 = q64 ? fs2_pair_order_grid(aa.order, (max_len + 63) / 64, H, B, xcd_on != 0) : grid;
            fs2_allow_lds(k, lds32, &ok[(q64 ? 4 : (two ? 2 : 0)) + (aa.drop.thresh ? 1 : 0)]);
            void* params[] = {&aa};
            if (hipLaunchKernel(k, grid32, dim3(threads), params, lds32, stream) != hipSuccess) {
                fs2_set_error("fs2_attention: launch failed: %s", hipGetErrorString(hipGetLastError()));
                return FS2_ERR_HIP;
            }
            return FS2_OK;
        }
    }
    hipLaunchKernelGGL((attention_kernel<T, DK, QT, OT>), grid, dim3(AT_THREADS), lds, stream, aa);
    FS2_CHECK_LAUNCH("fs2_attention");
    return FS2_OK;
}

// q | k | v and ctx in split-fp16 storage (dtype code 3)
template <int DK>
static int launch_attention_split(const AttnArgs& a, int B, int H, int max_len, hipStream_t stream) {
    // 32-key tiles: 64 KB of LDS at dk = 128 (two workgroups per CU, and room for another stream's workgroups beside one)
    // 64-key tiles (one workgroup per CU) for a handful of work items - one utterance, or a few long ones: half the
    // trips of the serial chain (us per launch, 32 / 64 keys: 1 x 120 phones 8.9 / 8.2, 1 x 512 22.7 / 20.2, 4 x 512
    // 23.2 / 20.8; batches: 32 x ~120 11.2 / 11.5, 128 x ~120 25.2 / 30.2)
    static const int bk_env = FS2_ENV_INT("FS2_ATTN_SPLIT_BK", 0);
    const int bk = bk_env ? bk_env : ((long)B * H * ((max_len + 63) / 64) * a.ksplit <= 128 ? 64 : 32);
    static const int xcd_on = FS2_ENV_INT("FS2_ATTN_XCD", 1);
    AttnArgs aa = a;
    // (with a key split the grid carries ksplit x the query blocks and attention_combine_kernel follows)
    const dim3 grid = fs2_pair_order_grid(aa.order, ((max_len + 63) / 64) * aa.ksplit, H, B, xcd_on != 0);
    static unsigned long long ok64 = 0, ok32 = 0;
    if (bk == 64) {
        constexpr int lds = 2 * 2 * 64 * DK * 4;
        fs2_allow_lds((const void*)attention_split_kernel<DK, 64>, lds, &ok64);
        hipLaunchKernelGGL((attention_split_kernel<DK, 64>), grid, dim3(AT_THREADS), lds, stream, aa);
    } else {
        constexpr int lds = 2 * 2 * 32 * DK * 4;
        fs2_allow_lds((const void*)attention_split_kernel<DK, 32>, lds, &ok32);
        hipLaunchKernelGGL((attention_split_kernel<DK, 32>), grid, dim3(AT_THREADS), lds, stream, aa);
    }
    FS2_CHECK_LAUNCH("fs2_attention");
    if (aa.ksplit > 1) {
        const long long n4 = (long long)aa.M * (aa.d / 4);
        hipLaunchKernelGGL(attention_combine_kernel<f16pair_t>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                           aa.part_o, aa.part_ml, (f16pair_t*)aa.ctx, aa.M, aa.d, DK, aa.ldo, aa.ksplit, 1.0f / AT_PSCALE);
        FS2_CHECK_LAUNCH("fs2_attention(combine)");
    }
    return FS2_OK;
}

}  // namespace fs2

using namespace fs2;

static int attention_entry(const void* qkv, void* ctx, float* lse, const int* cu, int B, int max_len, int M, int n_heads,
                           int dk, int ld, int ldo, int dtype, Dropout drop, hipStream_t stream, int ksplit = 1,
                           float* workspace = nullptr) {
    FS2_CHECK_ARG(qkv && ctx && cu && B > 0 && n_heads > 0, "fs2_attention: bad args");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1 || dtype == 2 || dtype == 3, "fs2_attention: dtype=%d", dtype);
    FS2_CHECK_ARG(dtype != 3 || ((dk == 64 || dk == 128) && ld % 32 == 0 && ((uintptr_t)qkv & 127) == 0),
                  "fs2_attention(split-fp16 q|k|v): head dim %d must be 64 or 128, ld=%d a multiple of 32, qkv 128-byte aligned",
                  dk, ld);
    FS2_CHECK_ARG(dtype < 2 || (!lse && !drop.thresh && ldo % 32 == 0 && ((uintptr_t)ctx & 127) == 0),
                  "fs2_attention(f16x3): the split-fp16 context needs ldo=%d in multiples of 32 and 128-byte alignment; "
                  "inference only", ldo);
    FS2_CHECK_ARG(dk == 32 || dk == 64 || dk == 96 || dk == 128, "fs2_attention: head dim %d unsupported (32/64/96/128)",
                  dk);
    const int d = n_heads * dk;
    const int epv = dtype == 1 ? 8 : 4;
    FS2_CHECK_ARG(ld >= 3 * d && ldo >= d && ld % epv == 0 && ldo % 4 == 0, "fs2_attention: bad leading dims");
    FS2_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0, "fs2_attention: unaligned pointers");
    if (max_len <= 0) return FS2_OK;
    AttnArgs a;
    a.qkv = qkv; a.ctx = ctx; a.cu = cu; a.ld = ld; a.ldo = ldo; a.d = d; a.lse = lse; a.M = M; a.drop = drop;
    a.ksplit = 1; a.part_o = nullptr; a.part_ml = nullptr;
    if (ksplit > 1) {
        FS2_CHECK_ARG((dtype == 1 || dtype == 3) && !lse && !drop.thresh && ksplit <= 8 && workspace && M > 0 &&
                          ((uintptr_t)workspace & 15) == 0,
                      "fs2_attention_keysplit: bf16 / split-fp16 (dtype 1 / 3) inference only, ksplit <= 8, workspace and the "
                      "total row count required");
        a.ksplit = ksplit;
        a.part_o = workspace;
        a.part_ml = workspace + (size_t)ksplit * M * d;
    }
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)dk);
#define FS2_ATT(TT, DKK) return launch_attention<TT, DKK>(a, B, n_heads, max_len, stream)
    if (dtype == 3) {          // q | k | v and context in split-fp16 storage, fp16 matrix pipe
        if (dk == 64) return launch_attention_split<64>(a, B, n_heads, max_len, stream);
        return launch_attention_split<128>(a, B, n_heads, max_len, stream);
    }
    if (dtype == 2) {          // fp32 q | k | v, context in split-fp16 storage
        if (dk == 32) return launch_attention<float, 32, f16pair_t>(a, B, n_heads, max_len, stream);
        if (dk == 64) return launch_attention<float, 64, f16pair_t>(a, B, n_heads, max_len, stream);
        if (dk == 96) return launch_attention<float, 96, f16pair_t>(a, B, n_heads, max_len, stream);
        return launch_attention<float, 128, f16pair_t>(a, B, n_heads, max_len, stream);
    } else if (dtype == 0) {
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

extern "C" int fs2_attention(const void* qkv, void* ctx, const int* cu, int B, int max_len, int n_heads, int dk, int ld,
                             int ldo, int dtype, hipStream_t stream) {
    return attention_entry(qkv, ctx, nullptr, cu, B, max_len, 0, n_heads, dk, ld, ldo, dtype, fs2_make_dropout(0, 0, 0.f),
                           stream);
}

// Split of the key range over several workgroups per (utterance, head, query block) for launches that leave CUs without
// a workgroup: one long utterance (BASELINE config 4, B = 1).  1 = no split.  bf16 (dtype 1) and the split-fp16 kernel
// (dtype 3: one utterance of 512 phones 20.2 us unsplit).
// Measured (tools/bench_attention.py, us per launch incl. the merge; unsplit -> split): one utterance of 1600 frames
// 23.8 -> 19.6 (3 splits), 3138 frames 40.4 -> 31.3 (2), 4165 frames 51.9 -> 41.3 (3).  The split pays only until every
// CU has ONE workgroup: a tile trip of this kernel runs as fast with one wave per SIMD as with two (4 x 4165 frames, 264
// workgroups: 121.9 us unsplit, 117 / 109 us with 528 / 792 workgroups), and below four trips per key group the fixed
// costs eat it (810 frames: 16.1 -> 15.8).  A batch (config 2) fills the chip with work items as it is.
extern "C" int fs2_attention_ksplit(int B, int max_len, int n_heads, int dk, int dtype) {
    const long items64 = (long)B * n_heads * ((max_len + 63) / 64), items128 = (long)B * n_heads * ((max_len + 127) / 128);
    if (dtype == 3) {     // split-fp16 kernel (64 queries per workgroup, no key groups): >= 2 trips of 64 keys per split
        if (B <= 0 || max_len <= 128 || !(dk == 64 || dk == 128) || items64 > 128) return 1;
        long ks = 256 / items64;
        const long by_len = ((max_len + 63) / 64) / 2;
        ks = ks < by_len ? ks : by_len;
        return (int)(ks < 1 ? 1 : (ks > 8 ? 8 : ks));
    }
    if (dtype != 1 || B <= 0 || max_len <= 256 || !(dk == 32 || dk == 64 || dk == 96 || dk == 128)) return 1;
    if (items64 <= 128) {                         // four-group form (64 queries): >= 4 trips of 32 keys per group
        long ks = 256 / items64;
        const long by_len = ((max_len + 31) / 32) / 16;
        ks = ks < by_len ? ks : by_len;
        return (int)(ks < 1 ? 1 : (ks > 8 ? 8 : ks));
    }
    if (max_len >= 1024) {                        // plain form (128 queries, 64-key tiles)
        long ks = 256 / items128;
        const long by_len = ((max_len + 63) / 64) / 8;
        ks = ks < by_len ? ks : by_len;
        ks = ks > 8 ? 8 : ks;
        if (ks >= 3) return (int)ks;              // (two splits: the two-group form of the unsplit launch does the same)
    }
    return 1;
}

extern "C" long long fs2_attention_keysplit_workspace(int M, int n_heads, int dk, int ksplit) {
    return (long long)ksplit * M * ((long long)n_heads * dk + 2LL * n_heads) * (long long)sizeof(float);
}

// fs2_attention with the keys of every work item split over `ksplit` workgroups (fs2_attention_ksplit) and merged by a
// second kernel; workspace: fs2_attention_keysplit_workspace bytes; M = total rows (cu[B]).  ksplit = 1: fs2_attention.
extern "C" int fs2_attention_keysplit(const void* qkv, void* ctx, const int* cu, int B, int max_len, int M, int n_heads, int dk,
                                      int ld, int ldo, int dtype, int ksplit, float* workspace, hipStream_t stream) {
    FS2_CHECK_ARG(ksplit >= 1 && ksplit <= 8, "fs2_attention_keysplit: ksplit=%d (1..8)", ksplit);
    return attention_entry(qkv, ctx, nullptr, cu, B, max_len, M, n_heads, dk, ld, ldo, dtype, fs2_make_dropout(0, 0, 0.f),
                           stream, ksplit, workspace);
}

// Training forward: as fs2_attention, and saves lse[h][m] (log2 domain, M = total rows) for fs2_attention_bwd.
extern "C" int fs2_attention_fwd(const void* qkv, void* ctx, float* lse, const int* cu, int B, int max_len, int M,
                                 int n_heads, int dk, int ld, int ldo, int dtype, unsigned seed, unsigned stream_id,
                                 float rate, hipStream_t stream) {
    FS2_CHECK_ARG(lse && M >= 0, "fs2_attention_fwd: null lse");
    FS2_CHECK_ARG(rate >= 0.f && rate < 1.f, "fs2_attention_fwd: dropout rate must be in [0, 1)");
    return attention_entry(qkv, ctx, lse, cu, B, max_len, M, n_heads, dk, ld, ldo, dtype,
                           fs2_make_dropout(seed, stream_id, rate), stream);
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

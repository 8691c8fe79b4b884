// Varlen conv1d / linear as an MFMA implicit GEMM over channel-last packed activations.
//
//   out[m, n] = epi( sum_{tap, c} X[m + tap - pad, c] * W[tap][n][c] + bias[n] ) (+ res[m, n])
//
// X is the packed (ragged, no padding) token matrix [M, Cin]; every row m belongs to one
// utterance [seg.x, seg.y) and taps that would read across an utterance edge contribute zero —
// the batch-1 "same"-padding semantics of the reference (network/encoder.py:16-30 Conv1d
// padding=(k-1)//2; SURVEY.md §7 "varlen halos").  taps == 1 is a plain Linear
// (network/encoder.py:39-44, :57-60; network/fastspeech2.py:158).
//
// Structure (gfx950): 256 threads = 4 waves as 2(M) x 2(N); each wave MT x NT MFMA 16x16 tiles
// (block tile 32MT x 32NT: 128x128 for the big frame-level GEMMs, 64x64 when the grid would not
// fill the chip).  K is walked in (channel chunk, tap) steps of 128 bytes per row.  Per chunk the
// X tile plus its (taps-1)-row halo is staged ONCE in LDS and re-used by every tap as a shifted
// row view (no im2col buffer); the W tile of (tap, chunk) is streamed.  Both are double-buffered:
// the loads of step s+1 are issued before the MFMAs of step s, one barrier per step.
//   fast path  (Cin % chunk == 0): global_load_lds_dwordx4 (LDS-DMA, no VGPR / ds_write traffic)
//   generic    (any Cin % 8 == 0): register staging with zero fill of the K tail
// LDS images are unpadded 128-byte rows with the 16-byte slot index XOR-swizzled by the row so the
// 16 rows of an MFMA fragment read hit 16 different bank groups; with LDS-DMA the swizzle is
// applied to the per-lane SOURCE address (the LDS side of the DMA is lane-linear).
// MFMA operand roles are swapped (A = W rows, B = X rows) and the W rows of a wave are fed in a
// permuted order so that a lane ends up with 16 CONSECUTIVE output channels of one token:
// 64-byte (f32) / 32-byte (bf16) contiguous epilogue accesses.
#include "common.h"
#include <stdlib.h>

namespace fs2 {

#ifndef FS2_CONV_PRE
#define FS2_CONV_PRE 1      // bf16: all LDS fragment reads of a (tap, chunk) step are requested in front of its MFMAs (-1 % per FFN conv launch; 0: per 32-channel half)
#endif
#ifndef FS2_CONV_PIN
#define FS2_CONV_PIN 1      // keep the closing wait + barrier of a step behind its MFMAs (0: where hipcc puts them)
#endif
constexpr int CG_MAX_TAPS = 9;
constexpr int CG_HALO = CG_MAX_TAPS - 1;
constexpr int CG_ROWB = 128;  // bytes per LDS row = one channel chunk

struct ConvGemmArgs {
    const void* X;       // [M, ldx] T
    const void* W;       // [taps][N][Cin] T
    const float* bias;   // [N] or null
    const float* res;    // [M, ldr] f32 or null
    const void* gate;    // [M, ldg] T or null: out = gate > 0 ? out * gate_scale : 0 (ReLU [+ dropout] backward
    int ldg;             //                     through the saved activation)
    float gate_scale;
    Dropout drop;        // training: dropout after the activation (FFN inner dropout, network/encoder.py:30,:44)
    void* out;           // [M, ldo]  f32 (out_f32 != 0) or T
    const int2* seg;     // [M] (start,end) row range of the row's utterance; null when taps == 1
    int M, N, Cin, taps;
    int ldx, ldr, ldo;
    int relu, out_f32;        // relu: activation code 0 none / 1 ReLU / 2 tanh
    int n_tiles, m_tiles;
    int xcd_sets, xcd_group;  // XCD-aware tile order: 8 / xcd_sets XCDs share a group of xcd_group N tiles (0: plain order)
    int krot;                 // 0: every workgroup walks K from (chunk_lo, tap 0); else the walk starts at a per-M-tile phase
    int ksplit;               // split of the channel-chunk range over gridDim.y partial outputs
    long long split_stride;   // elements between two partial outputs
    float out_scale;          // SPLIT (f16x3) mode: 1 / (power-of-two scale folded into the packed weights); else 1
};

// ---- f16x3: error-compensated half-precision operands (dtype FS2_F16X3) ---------------------------------------
// x = xh + xl and w = wh + wl with all four terms fp16 (11 significant bits each, so a pair carries 22 of fp32's
// 24); the product is accumulated as  wh.xh + wh.xl + wl.xh  on the f16 matrix pipe (fp32 accumulate): the dropped
// term wl.xl and the split residuals are below 2^-21 of |w||x|.  Both operands arrive ALREADY split, in the
// split-fp16 storage of common.h (f16pair_t): a 32-channel chunk of a row is 128 bytes = [32 x hi][32 x lo], the same
// bytes per row and chunk as fp32, so staging, swizzle and the LDS geometry are those of the fp32 path and a fragment
// is two 16-byte LDS reads.  W is split once at pack time, X by the kernel that produced it (LayerNorm T copy, this
// kernel's own epilogue, fs2_cast): a first version split X in registers here and spent more issue cycles on the
// conversions than on the MFMAs (77 us per phone-level FFN conv whatever the tile).  3 MFMAs of 16 cycles against
// 8 of 32 for exact-fp32 MFMA.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mma16_f16(const f16x8& a, const f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 16-byte slot swizzles (see header comment)
// X rows are read at a tap-dependent offset, 16 consecutive rows per fragment: with `r & 7` every ds_read_b128 lane group
// ({i 0-3, 12-15 of k-slot g} + {i 4-11 of k-slot g + 1}, MI355X_MICROARCH.md LDS table) covers the 16 slots of a 256-byte
// bank row whatever the offset; the earlier (r >> 1) & 7 averaged 6.7 LDS cycles per read instead of 4
// (SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE)
__device__ __forceinline__ int swz_x(int r) { return r & 7; }
__device__ __forceinline__ int swz_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// ---- epilogue (both kernels): lane (g, i) holds token m = mrow0 + 16 mt + i, channels nb .. nb + 4 NT - 1 with
//      acc[mt][nt][r] <-> channel nb + 4 nt + r.  Loads are issued in batches (clamped addresses, predicated
//      use) so their latencies overlap.
template <typename T, int MT, int NT, bool SPLIT = false>
__device__ __forceinline__ void conv_epilogue(const ConvGemmArgs& a, const float* bias, const float* res,
                                              f32x4 (&acc)[MT][NT], int mrow0, int nb) {
    const int i = threadIdx.x & 15;
    const bool vec_ok = ((a.N & 3) == 0) && ((a.ldo & 3) == 0) && (!res || (a.ldr & 3) == 0) &&
                        (!a.gate || (a.ldg & 3) == 0);
    if (vec_ok) {
        bool nok[NT];
        int ncl[NT];
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            nok[nt] = (nb + 4 * nt) < a.N;
            ncl[nt] = nok[nt] ? nb + 4 * nt : 0;
            bv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (bias) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const f32x4*)(bias + ncl[nt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mrow0 + mt * 16 + i;
            const bool mok = m < a.M;
            const size_t mc = (size_t)(mok ? m : a.M - 1);
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[mt][nt] * a.out_scale + bv[nt];
            if (a.relu == 1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] = fmaxf(v[nt][r], 0.f);
            } else if (a.relu == 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] = tanhf(v[nt][r]);
            }
            if (a.drop.thresh) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] *= drop_mul(a.drop, (uint32_t)mc, (uint32_t)(ncl[nt] + r));
            }
            if (a.gate) {
                const T* gp = (const T*)a.gate + mc * a.ldg;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    T g4[4];
                    if constexpr (sizeof(T) == 2)
                        *(uint2*)g4 = *(const uint2*)(gp + ncl[nt]);
                    else
                        *(f32x4*)g4 = *(const f32x4*)(gp + ncl[nt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] = to_f32(g4[r]) > 0.f ? v[nt][r] * a.gate_scale : 0.f;
                }
            }
            if (res) {
                f32x4 rv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) rv[nt] = *(const f32x4*)(res + mc * a.ldr + ncl[nt]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] += rv[nt];
            }
            if (a.out_f32) {
                float* o = (float*)a.out + (size_t)blockIdx.y * a.split_stride + mc * a.ldo;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (mok && nok[nt]) *(f32x4*)(o + ncl[nt]) = v[nt];
            } else if constexpr (SPLIT) {      // split-fp16 storage: the input of the next f16x3 GEMM
                f16pair_t* o = (f16pair_t*)a.out + mc * a.ldo;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (mok && nok[nt]) store4<f16pair_t>(o + ncl[nt], v[nt]);
            } else {
                T* o = (T*)a.out + (size_t)blockIdx.y * a.split_stride + mc * a.ldo;
                if constexpr (sizeof(T) == 2 && (NT % 2) == 0) {
                    // channels nb + 8 j .. nb + 8 j + 7 as one 16-byte store when the row allows it
                    const bool wide = ((a.ldo & 7) == 0) && ((nb & 7) == 0);
#pragma unroll
                    for (int j = 0; j < NT / 2; ++j) {
                        T t8[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            t8[r] = from_f32<T>(v[2 * j][r]);
                            t8[4 + r] = from_f32<T>(v[2 * j + 1][r]);
                        }
                        if (wide && mok && nok[2 * j + 1]) {
                            *(vec16*)(o + ncl[2 * j]) = *(const vec16*)t8;
                        } else {
                            if (mok && nok[2 * j]) *(uint2*)(o + ncl[2 * j]) = *(const uint2*)t8;
                            if (mok && nok[2 * j + 1]) *(uint2*)(o + ncl[2 * j + 1]) = *(const uint2*)(t8 + 4);
                        }
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (mok && nok[nt]) store4<T>(o + ncl[nt], v[nt]);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mrow0 + mt * 16 + i;
            if (m >= a.M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb + 4 * nt + r;
                    if (n >= a.N) continue;
                    float s = acc[mt][nt][r] * a.out_scale;
                    if (bias) s += bias[n];
                    if (a.relu == 1) s = fmaxf(s, 0.f);
                    if (a.relu == 2) s = tanhf(s);
                    if (a.drop.thresh) s *= drop_mul(a.drop, (uint32_t)m, (uint32_t)n);
                    if (a.gate) s = to_f32(((const T*)a.gate)[(size_t)m * a.ldg + n]) > 0.f ? s * a.gate_scale : 0.f;
                    if (res) s += res[(size_t)m * a.ldr + n];
                    const size_t oidx = (size_t)blockIdx.y * a.split_stride + (size_t)m * a.ldo + n;
                    if (a.out_f32)
                        ((float*)a.out)[oidx] = s;
                    else
                        ((T*)a.out)[oidx] = from_f32<T>(s);
                }
        }
    }
}

// MODE 0: generic register staging (any Cin % 8 == 0), X/W double-buffered, synchronous
// MODE 1: LDS-DMA, X single + W double buffer, one __syncthreads per step (3 workgroups / CU)
// MODE 3: LDS-DMA, TWO (tap, chunk) steps per barrier: the stages of steps s + 2 and s + 3 are in flight while steps s and
//         s + 1 compute (W ring of four stages; X ring of four images, because with k = 1 every step opens a new channel
//         chunk).  For the 64 x 64 launches of the latency regime, whose step is bound by the barrier -> LDS read -> MFMA
//         chain of one wave per SIMD: half the barriers, and the second step's reads overlap the first one's MFMAs.
// Workgroup = 2 WM waves laid out WM (tokens) x 2 (channels); wave tile 16 MT x 16 NT; block tile
// (16 MT WM) x (32 NT).  XDB: X image double-buffered (refill overlapped) or single (less LDS).
// OCC: workgroups per CU the register allocation is tuned for.
// KT: number of taps at compile time (9: the FFN convs get their own kernel symbol, so profiler summaries list
// them apart from the k = 1 / 3 / 5 GEMMs, and their tap loop bounds are constants), 0: run-time a.taps.
template <typename T, int MT, int NT, int WM, int MODE, bool XDB, int OCC, int KT = 0, bool SPLIT = false>
__global__ __launch_bounds__(128 * WM, (OCC * 2 * WM + 3) / 4) void conv_gemm_kernel(ConvGemmArgs a) {
    static_assert(!SPLIT || sizeof(T) == 4, "f16x3 works on fp32 activations");
    const int taps = KT ? KT : a.taps;
    constexpr bool DMA = MODE != 0;
    constexpr int NW = 2 * WM;
    constexpr int BM = 16 * MT * WM, BN = 32 * NT;
    constexpr int EPV = 16 / (int)sizeof(T);        // elements per 16-byte slot
    constexpr int BKE = CG_ROWB / (int)sizeof(T);   // channels per chunk (64 bf16 / 32 f32)
    constexpr int KSTEPS = BKE / 32;                // fragment steps per chunk
    constexpr int XROWS = BM + CG_HALO;
    constexpr int X_BYTES = XROWS * CG_ROWB;
    constexpr int W_BYTES = BN * CG_ROWB;
    constexpr int X_INSTR = XROWS / 8;              // 1-KiB (8-row) pieces of the X image
    constexpr int W_INSTR = BN / 8;
    constexpr bool PAIR = MODE == 3;                // two steps per barrier: X ring of four images, W ring of four stages
    constexpr int XBUFS = PAIR ? 4 : (XDB ? 2 : 1);
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [X buf0]([X buf1])[W stage 0][W stage 1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15;

    // Workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2).  Plain order: blocks that
    // share W columns have equal blockIdx % n_tiles -> same XCD when n_tiles divides 8, but then every XCD fetches
    // every X tile.  XCD-aware order (xcd_sets > 0): XCD x works on the N-tile group x % sets (its W slices stay
    // in that L2) and, inside the group, walks the M tiles m = (slot / group) * (8 / sets) + x / sets: an X tile is
    // fetched by `sets` XCDs instead of by all of them.  (speed only; every (m, n) tile is visited exactly once)
    int n_tile, m_tile;
    if (a.xcd_sets > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int set = xcd % a.xcd_sets, member = xcd / a.xcd_sets;
        n_tile = set * a.xcd_group + slot % a.xcd_group;
        m_tile = (slot / a.xcd_group) * (8 / a.xcd_sets) + member;
        if (m_tile >= a.m_tiles) return;
    } else {
        n_tile = blockIdx.x % a.n_tiles;
        m_tile = blockIdx.x / a.n_tiles;
    }
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;
    const int pad = (taps - 1) >> 1;
    // split-K: workgroup y handles channel chunks [chunk_lo, chunk_hi) and writes partial output y
    const int nchunks_all = (a.Cin + BKE - 1) / BKE;
    const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;
    const int chunk_lo = blockIdx.y * cps;
    const int chunk_hi = min(nchunks_all, chunk_lo + cps);
    const int nsteps = max(chunk_hi - chunk_lo, 0) * taps;
    const float* bias = blockIdx.y == 0 ? a.bias : nullptr;      // split-K: bias and residual go into partial 0
    const float* res = blockIdx.y == 0 ? a.res : nullptr;

    const T* __restrict__ X = (const T*)a.X;
    const T* __restrict__ W = (const T*)a.W;

    // per-lane validity window of tap shifts for each of its token rows
    int lo[MT], hi[MT];
    bool edge = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (16 * MT) + mt * 16 + i;
        int l = 1, h = 0;
        if (m < a.M) {
            l = 0;
            if (a.seg) {
                const int2 s = a.seg[m];
                l = s.x - m;
                h = s.y - 1 - m;
            }
        }
        lo[mt] = l;
        hi[mt] = h;
        edge |= (l > -pad) | (h < pad);
    }
    const bool need_mask = __syncthreads_or(edge ? 1 : 0) != 0;

    // ---- staging ----------------------------------------------------------------------------
    // piece p (8 rows x 128 B) of an image: lane l -> row 8p + (l >> 3), slot (l & 7); the slot holds
    // global chunk (slot ^ swz(row)).
    const int prow = lane >> 3, pslot = lane & 7;

    auto stage_x = [&](int chunk, int buf) __attribute__((always_inline)) {
        unsigned char* dst = smem + buf * X_BYTES;
        const int c0 = chunk * BKE;
        for (int p = wave; p < X_INSTR; p += NW) {
            const int r = 8 * p + prow;
            int gm = m0 - pad + r;
            gm = gm < 0 ? 0 : (gm >= a.M ? a.M - 1 : gm);  // out-of-range rows are always masked
            const int c = c0 + ((pslot ^ swz_x(r)) * EPV);
            if constexpr (DMA) {
                glds16(X + (size_t)gm * a.ldx + c, dst + p * 1024);
            } else {
                vec16 v = {0u, 0u, 0u, 0u};
                if (c < a.Cin) v = *(const vec16*)(X + (size_t)gm * a.ldx + c);
                *(vec16*)(dst + p * 1024 + lane * 16) = v;
            }
        }
    };
    auto stage_w = [&](int tap, int chunk, int buf) __attribute__((always_inline)) {
        unsigned char* dst = smem + XBUFS * X_BYTES + buf * W_BYTES;
        const int c0 = chunk * BKE;
        const T* wt = W + (size_t)tap * a.N * a.Cin;
#pragma unroll
        for (int q = 0; q < (W_INSTR + NW - 1) / NW; ++q) {
            const int p = wave + NW * q;
            if (W_INSTR % NW != 0 && p >= W_INSTR) break;
            const int r = 8 * p + prow;
            int n = n0 + r;
            n = n >= a.N ? a.N - 1 : n;                     // columns >= N are never stored
            const int c = c0 + ((pslot ^ swz_w(r)) * EPV);
            if constexpr (DMA) {
                glds16(wt + (size_t)n * a.Cin + c, dst + p * 1024);
            } else {
                vec16 v = {0u, 0u, 0u, 0u};
                if (c < a.Cin) v = *(const vec16*)(wt + (size_t)n * a.Cin + c);
                *(vec16*)(dst + p * 1024 + lane * 16) = v;
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // W rows of this wave are consumed in the order  a -> (a >> 2) * (4 NT) + 4 nt + (a & 3)
    // (a = A-operand row of tile nt), so lane group g ends up with channels 4 NT g .. 4 NT g + 4 NT - 1
    const int wrow_base = wn * (16 * NT) + (i >> 2) * (4 * NT) + (i & 3);
    const int xrow_base = wm * (16 * MT) + i;

    // one (tap, chunk) step of MFMAs out of X image `xbuf` and W stage `wbuf`; `mid()` runs between
    // the two fragment steps (bf16) / after the fragment loads (f32): the place for DMA issue
    auto compute = [&](int tap, int xbuf, int wbuf, auto&& mid) __attribute__((always_inline)) {
        const unsigned char* xs = smem + xbuf * X_BYTES;
        const unsigned char* ws = smem + XBUFS * X_BYTES + wbuf * W_BYTES;
        const int shift = tap - pad;
#if FS2_CONV_PRE
        if constexpr (!SPLIT && sizeof(T) == 2) {
            // all fragments of the step are requested before its first MFMA
            frag_t wf[KSTEPS][NT], xf[KSTEPS][MT];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int r = wrow_base + 4 * nt;
                    wf[ks][nt] = *(const frag_t*)(ws + r * CG_ROWB + (((ks * 4 + g) ^ swz_w(r)) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int r = xrow_base + mt * 16 + tap;
                    xf[ks][mt] = *(const frag_t*)(xs + r * CG_ROWB + (((ks * 4 + g) ^ swz_x(r)) << 4));
                }
            }
            if (need_mask) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (shift < lo[mt] || shift > hi[mt]) {
#pragma unroll
                        for (int ks = 0; ks < KSTEPS; ++ks) xf[ks][mt] = frag_zero<T>();
                    }
            }
            mid();
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(wf[ks][nt], xf[ks][mt], acc[mt][nt]);
            return;
        }
#endif
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            frag_t wf[NT], xf[MT];
            f16x8 wh[SPLIT ? NT : 1], wl[SPLIT ? NT : 1], xh[SPLIT ? MT : 1], xl[SPLIT ? MT : 1];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int r = wrow_base + 4 * nt;
                const unsigned char* rowp = ws + r * CG_ROWB;
                if constexpr (SPLIT) {      // [32 x hi | 32 x lo] fp16: slots 0..3 hold hi, 4..7 lo (8 channels each)
                    wh[nt] = *(const f16x8*)(rowp + ((g ^ swz_w(r)) << 4));
                    wl[nt] = *(const f16x8*)(rowp + (((4 + g) ^ swz_w(r)) << 4));
                } else if constexpr (sizeof(T) == 2) {
                    wf[nt] = *(const frag_t*)(rowp + (((ks * 4 + g) ^ swz_w(r)) << 4));
                } else {
                    const f32x4 v0 = *(const f32x4*)(rowp + (((2 * g) ^ swz_w(r)) << 4));
                    const f32x4 v1 = *(const f32x4*)(rowp + (((2 * g + 1) ^ swz_w(r)) << 4));
                    wf[nt] = (frag_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = xrow_base + mt * 16 + tap;
                const unsigned char* rowp = xs + r * CG_ROWB;
                if constexpr (SPLIT) {
                    xh[mt] = *(const f16x8*)(rowp + ((g ^ swz_x(r)) << 4));
                    xl[mt] = *(const f16x8*)(rowp + (((4 + g) ^ swz_x(r)) << 4));
                } else if constexpr (sizeof(T) == 2) {
                    xf[mt] = *(const frag_t*)(rowp + (((ks * 4 + g) ^ swz_x(r)) << 4));
                } else {
                    const f32x4 v0 = *(const f32x4*)(rowp + (((2 * g) ^ swz_x(r)) << 4));
                    const f32x4 v1 = *(const f32x4*)(rowp + (((2 * g + 1) ^ swz_x(r)) << 4));
                    xf[mt] = (frag_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
            }
            if (need_mask) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (shift < lo[mt] || shift > hi[mt]) {
                        if constexpr (SPLIT) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) { xh[mt][j] = (_Float16)0.f; xl[mt][j] = (_Float16)0.f; }
                        } else {
                            xf[mt] = frag_zero<T>();
                        }
                    }
            }
            if (ks == 0) mid();
            if constexpr (SPLIT) {
                // small terms first, then the leading product; one term over all tiles at a time, so that the three
                // MFMAs of an accumulator are MT * NT instructions apart instead of back to back
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16_f16(wl[nt], xh[mt], acc[mt][nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16_f16(wh[nt], xl[mt], acc[mt][nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16_f16(wh[nt], xh[mt], acc[mt][nt]);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(wf[nt], xf[mt], acc[mt][nt]);
            }
        }
    };

    {
        // Every workgroup of an N tile streams the same W tiles, and workgroups launched together would ask L2 for the
        // same lines at the same moment, step after step.  Starting the (cyclic) K walk at a per-M-tile phase spreads
        // those requests over the taps x chunks W tiles: 84.8 -> 80.4 us per decoder-conv launch, A/B/A/B on one box.
        // (fp32 accumulation order then depends on the M tile; every (tap, chunk) step is still visited exactly once.)
        int chunk_start = chunk_lo, tap_start = 0;
        if (a.krot && nsteps > 0) {
            const unsigned nch = (unsigned)(chunk_hi - chunk_lo), seed = (unsigned)m_tile;
            chunk_start = chunk_lo + (int)(seed % nch);
            tap_start = (int)((seed / nch) % (unsigned)taps);
        }
        if constexpr (PAIR) {
            int itap = tap_start, ichunk = chunk_start, itc = 0, iord = 0, istep = 0;      // issue cursor
            auto issue = [&]() __attribute__((always_inline)) {
                if (itc == 0) stage_x(ichunk, iord & 3);
                stage_w(itap, ichunk, istep & 3);
                itap = itap + 1 == taps ? 0 : itap + 1;
                ++istep;
                if (++itc == taps) { itc = 0; ++iord; ichunk = ichunk + 1 == chunk_hi ? chunk_lo : ichunk + 1; }
            };
            if (istep < nsteps) issue();
            if (istep < nsteps) issue();
            vm_wait_barrier<0>();
            int tap = tap_start, tc = 0, ord = 0;
            for (int step = 0; step < nsteps; step += 2) {
                // the stages this pair's successors need: their slots were last read one pair ago, behind the barrier
                if (istep < nsteps) issue();
                if (istep < nsteps) issue();
                compute(tap, ord & 3, step & 3, []() {});
                tap = tap + 1 == taps ? 0 : tap + 1;
                if (++tc == taps) { tc = 0; ++ord; }
                if (step + 1 < nsteps) {
                    compute(tap, ord & 3, (step + 1) & 3, []() {});
                    tap = tap + 1 == taps ? 0 : tap + 1;
                    if (++tc == taps) { tc = 0; ++ord; }
                }
#if FS2_CONV_PIN
                __builtin_amdgcn_sched_barrier(0);
#endif
                vm_wait_barrier<0>();
            }
            conv_epilogue<T, MT, NT, SPLIT>(a, bias, res, acc, m0 + wm * (16 * MT), n0 + wn * (16 * NT) + g * (4 * NT));
            return;
        }
        if (nsteps > 0) {
            stage_x(chunk_start, 0);
            stage_w(tap_start, chunk_start, 0);
        }
        if constexpr (DMA) vm_wait_barrier<0>(); else __syncthreads();

        int xcur = 0, wcur = 0;
        int tap = tap_start, chunk = chunk_start, tcount = 0;
        for (int step = 0; step < nsteps; ++step) {
            int ntap = tap + 1 == taps ? 0 : tap + 1, nchunk = chunk;
            ++tcount;
            const bool chunk_done = tcount == taps;
            if (chunk_done) { tcount = 0; nchunk = chunk + 1 == chunk_hi ? chunk_lo : chunk + 1; }
            const bool has_next = (step + 1) < nsteps;
            const bool next_x = has_next && chunk_done;
            if (DMA) {   // LDS-DMA prefetch of step+1 into the other W buffer, in flight during the MFMAs
                if (has_next) stage_w(ntap, nchunk, wcur ^ 1);
                if (XDB && next_x) stage_x(nchunk, xcur ^ 1);
            }
            compute(tap, xcur, wcur, []() {});
#if FS2_CONV_PIN
            // the MFMAs are register-only: left alone, hipcc hoists the step's closing wait + barrier up to behind the
            // second MFMA, and the wave then sits in vmcnt(0) with its DMA just issued and 30 MFMAs still to go
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (!DMA) {  // generic path: synchronous register staging of step+1 (odd shapes only)
                if (has_next) stage_w(ntap, nchunk, wcur ^ 1);
                if (next_x) stage_x(nchunk, xcur ^ 1);
            }
            // end of the step: the LDS-DMA issued at its top has landed (the vmcnt(0) is written out, not left to the
            // compiler's alias analysis of LDS-DMA against LDS reads) and every wave is done with the stage
            if constexpr (DMA) vm_wait_barrier<0>(); else __syncthreads();
            if (DMA && !XDB && next_x) {   // single X image: refill it once per channel chunk, between two barriers
                stage_x(nchunk, 0);
                vm_wait_barrier<0>();
            }
            wcur ^= 1;
            if (XDB && next_x) xcur ^= 1;
            tap = ntap;
            chunk = nchunk;
        }
    }

    conv_epilogue<T, MT, NT, SPLIT>(a, bias, res, acc, m0 + wm * (16 * MT), n0 + wn * (16 * NT) + g * (4 * NT));
}

// XCD-aware tile order for a BM x BN tiling (see conv_gemm_kernel); returns the x grid size
static int conv_gemm_plan_grid(ConvGemmArgs& a, int BM, int BN, size_t elem) {
    a.n_tiles = (a.N + BN - 1) / BN;
    const int m_tiles = (a.M + BM - 1) / BM;
    a.m_tiles = m_tiles;
    int grid_x = a.n_tiles * m_tiles;
    a.xcd_sets = a.xcd_group = 0;
    static const int krot = FS2_ENV_INT("FS2_CONV_KROT", 1);   // 0: off (A/B runs)
    a.krot = krot;
    // N tiles whose W slices (all taps) fit ~2.5 MB of an XCD's L2 form a group; sets = n_tiles / group
    static const int xcd_on = FS2_ENV_INT("FS2_CONV_XCD", 1);
    const double slice = (double)a.taps * BN * a.Cin * elem;     // all K ranges of a split are resident together
    int group = (int)(2.5e6 / slice);
    if (xcd_on && group >= 2 && a.n_tiles >= 2 && m_tiles >= 64) {
        group = group > a.n_tiles ? a.n_tiles : group;
        while (group > 1 && (a.n_tiles % group != 0 || (a.n_tiles / group) > 8 || 8 % (a.n_tiles / group) != 0)) --group;
        const int sets = a.n_tiles / group;
        if (group >= 2 && sets >= 1 && sets <= 8 && 8 % sets == 0) {
            a.xcd_sets = sets;
            a.xcd_group = group;
            const int per = 8 / sets;
            grid_x = 8 * group * ((m_tiles + per - 1) / per);
        }
    }
    return grid_x;
}

template <typename T, int MT, int NT, int WM, int MODE, bool XDB, int OCC, int KT = 0, bool SPLIT = false>
static int launch_conv_gemm(ConvGemmArgs a, hipStream_t stream) {
    constexpr int BM = 16 * MT * WM, BN = 32 * NT;
    constexpr int lds = MODE == 3 ? 4 * (BM + CG_HALO) * CG_ROWB + 4 * BN * CG_ROWB
                                  : (XDB ? 2 : 1) * (BM + CG_HALO) * CG_ROWB + 2 * BN * CG_ROWB;
    static unsigned long long lds_ok = 0;
    fs2_allow_lds((const void*)conv_gemm_kernel<T, MT, NT, WM, MODE, XDB, OCC, KT, SPLIT>, lds, &lds_ok);
    const int grid_x = conv_gemm_plan_grid(a, BM, BN, sizeof(T));
    hipLaunchKernelGGL((conv_gemm_kernel<T, MT, NT, WM, MODE, XDB, OCC, KT, SPLIT>), dim3(grid_x, a.ksplit),
                       dim3(128 * WM), lds, stream, a);
    FS2_CHECK_LAUNCH("fs2_conv_gemm");
    return FS2_OK;
}

// f16x3 (FS2_F16X3): fp32 X, pre-split fp16 W; Cin % 32 == 0 (checked by the entry point), so always LDS-DMA.
// Used by the phone-level half only (a few thousand tokens): 64 x 64 tiles when 128 x 128 would leave CUs idle.
static int dispatch_conv_gemm_split(const ConvGemmArgs& a, hipStream_t stream) {
    const long big_blocks = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.ksplit;
    static const int forced = FS2_ENV_INT("FS2_SPLIT_TILE", 0);   // experiments: 22, 24, 42, 44
    if (forced == 22) return launch_conv_gemm<float, 2, 2, 2, 3, false, 2, 0, true>(a, stream);
    if (forced == 24) return launch_conv_gemm<float, 2, 4, 2, 1, false, 3, 0, true>(a, stream);
    if (forced == 42) return launch_conv_gemm<float, 4, 2, 2, 1, false, 3, 0, true>(a, stream);
    if (forced == 44) return launch_conv_gemm<float, 4, 4, 2, 1, false, 3, 0, true>(a, stream);
    // 128 x 128 tiles down to 64 workgroups: alone, 64-row tiles are up to 15 % faster on these launches (more
    // workgroups per CU to hide the stage latency), but the phone-level half runs beside the frame-level kernels of the
    // previous batch, and there the fewer, larger workgroups win (bench step 1.994 vs 2.02 / 2.03 ms, A/B/A/B on one box)
    // Latency regime (a few hundred tokens: one utterance): 64 x 64 tiles also for the split-K launches of up to 200
    // large tiles, on MODE 3 (two steps per barrier, X images prefetched with the W stages, so that the refill at every
    // channel-chunk change - every step of a k = 1 GEMM - no longer waits between two barriers of its own).  Per launch of
    // a batch-1 forward (BASELINE config 1, rocprofv3 kernel trace, us): MODE 1 with a single X image / with two X images
    // / MODE 3: encoder QKV 11.6 / 9.3 / 8.7, out-proj 11.9 / 9.9 / 8.8, second FFN conv 15.6 / 13.8 / 12.2; decoder first
    // FFN conv (208 workgroups x 36 steps) 19.7 / 19.9 / 17.5, second 14.2 / 14.9 / 14.1.
    // (X and W in rings of three staged TWO steps ahead behind counted vmcnt waits, 60 KB of LDS: 5-15 % SLOWER per
    // launch here - decoder first FFN conv of one utterance 22.4 vs 19.7 us, encoder second 15.8 vs 13.6: a step of these
    // launches is bound by the barrier -> LDS read -> MFMA chain of one wave per SIMD, not by the stage's round trip)
    if (big_blocks < 64 || (a.M <= 1024 && big_blocks < 200)) return launch_conv_gemm<float, 2, 2, 2, 3, false, 2, 0, true>(a, stream);
    return launch_conv_gemm<float, 4, 4, 2, 1, false, 3, 0, true>(a, stream);
}

template <typename T>
static int dispatch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
    constexpr int BKE = CG_ROWB / (int)sizeof(T);
    const bool dma = (a.Cin % BKE) == 0;
    const long big_blocks = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.ksplit;
    const bool small = big_blocks < 200;   // would leave most of the 256 CUs idle: use 64x64 tiles
    if (!dma) return small ? launch_conv_gemm<T, 2, 2, 2, 0, true, 2>(a, stream) : launch_conv_gemm<T, 4, 4, 2, 0, true, 2>(a, stream);
    if (small) return launch_conv_gemm<T, 2, 2, 2, 3, false, 2>(a, stream);     // two steps per barrier: see dispatch_conv_gemm_split
    // 128x128 (MT = 4), X image single-buffered: 50 KB of LDS -> 3 workgroups per CU.  Measured on MI355X (config-2
    // decoder FFN, bf16, random data): 128.1 us vs 154 us for the double-buffered-X / 2 WG per CU form,
    // 155 us for the 3-stage ring (MODE 2), 145-180 us for 8-wave 128x128 / 256x128 and 128x64 tiles.
    // Narrow outputs (N <= 256, e.g. the second FFN conv) give fewer workgroups than the 768 that can be resident:
    // 96-row tiles spread them over the CUs more evenly (measured 157 vs 167 us on 25 k x 256 x 9216; 128 rows
    // stay better everywhere else, 64 rows nowhere; a 256 x 128 block of 128 x 64 wave tiles — fewer LDS reads
    // per MFMA, 2 workgroups per CU — measured 145 / 202 us against 136 / 166 us on the two FFN convs: the
    // kernel is bound by latency hiding, not by LDS bandwidth).  A 192 x 256 tile on 8 waves, one workgroup per CU,
    // with the W stages in a ring of three behind counted `s_waitcnt vmcnt(N)` + raw s_barrier (what conv_wgrad.hip
    // does; git history: "experimental 192x256 big-tile kernel") read 0.7x the bytes through L2 but ran the decoder
    // convs in 148 / 149 us against 118 / 123 us here: 536 workgroups at one per CU take three rounds, and inside
    // a round the waves still spent a third of their time in the stage wait (SQ_WAIT_INST_ANY 36 %, MFMA busy 33 %).
    // The same ring for the phone-level launches (one workgroup per CU, nothing else to hide the DMA latency; W ring of
    // three + double-buffered X in this template) gave 37.9 vs 40.9 us alone and lost 3 % in the overlapped
    // throughput loop (83 KB of LDS keeps the other streams' workgroups off the CU): those launches are bound by the
    // read -> wait -> MFMA chain of a single wave per SIMD, not by the stage latency.  Not kept.
    // And for the decoder convs: W ring of three (two steps ahead, counted waits) + single X image = 65 KB, two
    // workgroups per CU: 87.7 vs 83.8 us per launch — three workgroups with one stage ahead hide more than two with two.
    // FS2_CONV_MT=2|3|4 forces a tile height (experiments).
    static const int forced = FS2_ENV_INT("FS2_CONV_MT", 0);
    // (only while the 96-row tiling itself still fits one round of 768 resident workgroups: M = 21000 with two K
    // ranges is 660 tiles of 128 rows but 876 of 96, whose short second round cost 120 vs 105 us)
    const long blocks96 = (long)((a.M + 95) / 96) * ((a.N + 127) / 128) * a.ksplit;
    int best = (a.N <= 256 && big_blocks < 768 && blocks96 <= 768 && a.Cin * a.taps >= 2048) ? 3 : 4;
    // Wide outputs (no split-K: ReLU epilogue) that fill less than about one round of 768 tiles of 128 rows: lower tiles
    // give the CUs more workgroups to hide the stage latency with.  First FFN conv (N = 1024, K = 2304) alone, us per
    // launch with 128 / 96 / 64 rows: M = 3767 (240 tiles): 36.9 / 36.8 / 31.8; 6144 (384): 45.6 / 39.1 / 38.4;
    // 9216 (576): 53.7 / 48.6 / 60.5; 12383 (776): 72.7 / 70.5 / 79.4; from 15000 (944) up 128 rows win or tie.
    static const int low_tiles = FS2_ENV_INT("FS2_CONV_LOWTILES", 1);   // 0: off (A/B runs)
    if (low_tiles && a.ksplit == 1 && a.N > 256 && a.Cin * a.taps >= 1024) {
        // (64 rows below 320 tiles: -14 % for that launch alone, but in the bench's overlapped mode, where the phone-level
        // convs run beside the frame-level ones, the step got 1 % slower; not used)
        // (and, as above, only while the 96-row tiling fits one round: 27 k x 384 x 1536 is 636 tiles of 128 rows but 846
        // of 96, 72 vs 58 us)
        if (big_blocks >= 320 && blocks96 <= 768) best = 3;
    }
    if (forced) best = forced;
    // (k = 1 GEMMs on 256-token tiles, 8 waves, two workgroups per CU: QKV 25.6 k x 256 -> 768 42 vs 23.5 us, the YAML's
    // 384 -> 1536 linear 98 vs 57 us, only 1536 -> 384 gained, 53 vs 57 us; not kept)
    if (a.taps == 9) {      // the FFN convs: own symbol, compile-time tap count
        if (best == 3) return launch_conv_gemm<T, 3, 4, 2, 1, false, 3, 9>(a, stream);
        if (best == 4) return launch_conv_gemm<T, 4, 4, 2, 1, false, 3, 9>(a, stream);
    }
    if (best == 3) return launch_conv_gemm<T, 3, 4, 2, 1, false, 3>(a, stream);
    if (best == 2) return launch_conv_gemm<T, 2, 4, 2, 1, false, 3>(a, stream);
    return launch_conv_gemm<T, 4, 4, 2, 1, false, 3>(a, stream);
}

}  // namespace fs2

// dtype: 0 = f32, 1 = bf16 (include/fs2_hip.h)
static int conv_gemm_entry(const void* X, const void* W, const float* bias, const float* res, const void* gate,
                           float gate_scale, void* out, const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr,
                           int ldg, int ldo, int relu, int out_f32, int dtype, int ksplit, long long split_stride,
                           fs2::Dropout drop, hipStream_t stream, float w_scale = 1.0f) {
    using namespace fs2;
    FS2_CHECK_ARG(!gate || (ldg >= N && ((uintptr_t)gate & 15) == 0 && ksplit == 1),
                  "fs2_conv_gemm_gated: gate needs ldg >= N, 16-byte alignment and ksplit == 1");
    FS2_CHECK_ARG(X && W && out, "fs2_conv_gemm: null X/W/out");
    FS2_CHECK_ARG(M >= 0 && N > 0 && Cin > 0, "fs2_conv_gemm: bad dims M=%d N=%d Cin=%d", M, N, Cin);
    FS2_CHECK_ARG(taps >= 1 && taps <= CG_MAX_TAPS && (taps & 1), "fs2_conv_gemm: taps=%d (odd, 1..%d)", taps,
                  CG_MAX_TAPS);
    FS2_CHECK_ARG(taps == 1 || seg, "fs2_conv_gemm: taps>1 needs the per-row segment table");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1 || dtype == 2, "fs2_conv_gemm: dtype=%d", dtype);
    FS2_CHECK_ARG(dtype != 2 || (Cin % 32 == 0 && ldx % 32 == 0 && !gate && !drop.thresh && w_scale > 0.f &&
                                 ((uintptr_t)X & 127) == 0 && ((uintptr_t)W & 127) == 0),
                  "fs2_conv_gemm_f16x3: Cin=%d / ldx=%d must be multiples of 32 (whole [hi | lo] runs), X and W 128-byte "
                  "aligned; no gate / dropout", Cin, ldx);
    FS2_CHECK_ARG(dtype != 2 || out_f32 || (ksplit == 1 && !res && N % 32 == 0 && ldo % 32 == 0 && ((uintptr_t)out & 127) == 0),
                  "fs2_conv_gemm_f16x3: a split-fp16 output needs N=%d / ldo=%d in multiples of 32, 128-byte alignment, no "
                  "residual and no split-K", N, ldo);
    FS2_CHECK_ARG(relu >= 0 && relu <= 2, "fs2_conv_gemm: activation code %d (0 none, 1 relu, 2 tanh)", relu);
    const int epv = dtype == 1 ? 8 : 4;
    FS2_CHECK_ARG(Cin % epv == 0 && ldx % epv == 0 && ldx >= Cin,
                  "fs2_conv_gemm: Cin=%d / ldx=%d must be multiples of %d (16-byte rows)", Cin, ldx, epv);
    FS2_CHECK_ARG(ldo >= N && (!res || ldr >= N), "fs2_conv_gemm: ldo/ldr smaller than N");
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                      ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0,
                  "fs2_conv_gemm: pointers must be 16-byte aligned");
    FS2_CHECK_ARG(ksplit >= 1 && ksplit <= 16, "fs2_conv_gemm: ksplit=%d (1..16)", ksplit);
    FS2_CHECK_ARG(ksplit == 1 || (!relu && out_f32 && split_stride >= (long long)M * ldo && split_stride % 4 == 0),
                  "fs2_conv_gemm: split-K partial outputs are fp32, take no activation and need split_stride >= M*ldo");
    if (M == 0) return FS2_OK;
    ConvGemmArgs a;
    a.X = X; a.W = W; a.bias = bias; a.res = res; a.gate = gate; a.ldg = ldg; a.gate_scale = gate_scale; a.drop = drop; a.out = out; a.seg = (const int2*)seg;
    a.M = M; a.N = N; a.Cin = Cin; a.taps = taps; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo;
    a.relu = relu; a.out_f32 = out_f32; a.n_tiles = 1;
    a.ksplit = ksplit; a.split_stride = ksplit > 1 ? split_stride : 0;
    a.out_scale = dtype == 2 ? 1.0f / w_scale : 1.0f;
    if (dtype == 2) return dispatch_conv_gemm_split(a, stream);
    return dtype == 0 ? dispatch_conv_gemm<float>(a, stream) : dispatch_conv_gemm<bf16_t>(a, stream);
}

// Split-K factor the launch schedules use for a GEMM whose output is summed by an add + LayerNorm kernel
// (fs2_add_layernorm `parts`).  Short K: 1.  Long K (>= 32 steps of one (chunk, tap) each; in practice the narrow
// second FFN conv, N = 256, K = 9 x 1024): the factor with the lowest modelled time.  The model is fitted to
// `tools/ks_sweep.py` on MI355X (bf16, N = 256, K = 9216, us per launch; * = chosen):
//        M     ks=1     2      3      4      6      8
//     3072     69.3   43.9   34.9   30.0   30.7*  26.9
//     6144     78.3   49.6   45.2   39.6*  39.6   42.3
//     9216     91.8   66.7   56.1*  52.5   54.8   60.9
//    12288     92.1   69.8   64.4*  65.2   72.9   74.3
//    18432    122.5   90.8*  97.9   97.8   97.7  108.4
//    25603    152.7  145.8  123.7* 132.8  131.0  147.2
//    40000    226.0  179.1* 183.6  187.5  198.2  224.1
// (a later sweep over 810 ... 60000 tokens: the chosen factor is within 5 % of the best one everywhere.)
// The table is reproduced within ~10 % by  time = (K steps / ks + 10) x step(w),  w = 128-row tiles x ks
// workgroups, with ONE smooth step time: step(w) = max(0.75, 0.35 + w / 640) us for the 128 x 128 kernel (latency
// floor + throughput term; no visible round quantisation except a short tail just past a full round of 768, +12 %)
// and 0.42 + w64 / 3400 us for the 64 x 64 tiles the dispatch uses below 200 workgroups.  Every extra partial also
// costs its fp32 read in the LayerNorm kernel (4.5 TB/s).  Unconstrained optimum: ks* = sqrt(3226 / tiles).
static double conv_gemm_split_cost(int M, int N, int ksteps, long blocks, int ks) {
    const double steps = (double)ksteps / ks + 10.0;
    const double w = (double)blocks * ks;
    double step;
    if (w < 200.0) {
        step = 0.42 + (double)((M + 63) / 64) * ((N + 63) / 64) * ks / 3400.0;
    } else {
        step = 0.35 + w / 640.0;
        step = step < 0.75 ? 0.75 : step;
        const double rem = w - 768.0 * (int)(w / 768.0);
        if (w > 768.0 && rem > 0.0 && rem < 0.12 * 768.0) step *= 1.12;
    }
    return steps * step + (ks - 1) * ((double)M * N * 4.0 / 4.5e6);
}

extern "C" int fs2_conv_gemm_ksplit(int M, int N, int Cin, int taps, int dtype) {
    const int nch = (Cin + (dtype == 1 ? 64 : 32) - 1) / (dtype == 1 ? 64 : 32);
    if (M <= 0 || nch * taps < 32) return 1;
    const long blocks = (long)((M + 127) / 128) * ((N + 127) / 128);
    static const int cand[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    int best = 1;
    double best_cost = 0.0;
    for (int q = 0; q < 8; ++q) {
        const int ks = cand[q];
        if (ks > nch || (ks > 1 && (nch / ks) * taps < 8)) break;
        // 12 and 16 partials: a handful of tokens only (one short utterance: 2 workgroups would walk 288 steps each), and
        // only when every K range gets chunks
        if (ks > 8 && (blocks > 4 || (long)(ks - 1) * ((nch + ks - 1) / ks) >= nch)) continue;
        // beyond 8 partials the LayerNorm kernel's walk over them shows: + 0.25 us each (profiles/r03_q_*)
        const double c = conv_gemm_split_cost(M, N, nch * taps, blocks, ks) + (ks > 8 ? 0.25 * (ks - 8) : 0.0);
        if (q == 0 || c < best_cost) { best = ks; best_cost = c; }
    }
    return best;
}

// Split-K for a GEMM whose ACTIVATION must follow the full sum (first FFN conv, predictor convs: ReLU): the launch writes
// plain partials (bias in partial 0, no activation) and fs2_sum_act_layernorm applies the ReLU to their sum — as part of
// the LayerNorm that follows anyway (predictors: extra_us = 0) or as one more small launch (first FFN conv: extra_us =
// its cost, about 3.5 us).  Only the latency regime is served: at most 64 tiles of 128 x 128 (a few hundred tokens: one
// utterance, batch-1 decoding), where the launch is a few workgroups each walking the whole K range; and only for a
// modelled gain of 20 % or more.  Returns 1 when the unsplit launch with the ReLU in its epilogue is the path to take.
extern "C" int fs2_conv_gemm_ksplit_act(int M, int N, int Cin, int taps, int dtype, float extra_us) {
    const int nch = (Cin + (dtype == 1 ? 64 : 32) - 1) / (dtype == 1 ? 64 : 32);
    const long blocks = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (M <= 0 || nch * taps < 12 || blocks > 64) return 1;
    static const int cand[6] = {2, 3, 4, 6, 8, 16};
    const double base = conv_gemm_split_cost(M, N, nch * taps, blocks, 1);
    int best = 1;
    double best_cost = 0.8 * base;
    for (int q = 0; q < 6; ++q) {
        const int ks = cand[q];
        const int cps = (nch + ks - 1) / ks;
        if (ks > nch || cps * taps < 4 || (long)(ks - 1) * cps >= nch) continue;
        const double c = conv_gemm_split_cost(M, N, nch * taps, blocks, ks) + extra_us + 0.25 * (ks - 1);
        if (c < best_cost) { best = ks; best_cost = c; }
    }
    return best;
}

extern "C" int fs2_conv_gemm(const void* X, const void* W, const float* bias, const float* res, void* out,
                             const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr, int ldo,
                             int relu /* activation: 0 none, 1 ReLU, 2 tanh */, int out_f32, int dtype, int ksplit,
                             long long split_stride, hipStream_t stream) {
    return conv_gemm_entry(X, W, bias, res, nullptr, 1.f, out, seg, M, N, Cin, taps, ldx, ldr, 0, ldo, relu, out_f32, dtype,
                           ksplit, split_stride, fs2_make_dropout(0, 0, 0.f), stream);
}

// fs2_conv_gemm in the f16x3 mode: X [M, ldx] in split-fp16 storage, W packed by fs2_pack_weight_f16x3 (values
// multiplied by the power of two w_scale before the split), out fp32 (out_f32 != 0) or split-fp16 storage.  Same
// tiling, masking and epilogue as fs2_conv_gemm.
extern "C" int fs2_conv_gemm_f16x3(const void* X, const void* W, float w_scale, const float* bias, const float* res,
                                   void* out, const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr, int ldo,
                                   int relu, int out_f32, int ksplit, long long split_stride, hipStream_t stream) {
    return conv_gemm_entry(X, W, bias, res, nullptr, 1.f, out, seg, M, N, Cin, taps, ldx, ldr, 0, ldo, relu, out_f32, 2,
                           ksplit, split_stride, fs2_make_dropout(0, 0, 0.f), stream, w_scale);
}

// Training forward of a conv / linear followed by dropout: out = dropout(act(conv(X) + bias)) (no residual, no split-K).
extern "C" int fs2_conv_gemm_dropout(const void* X, const void* W, const float* bias, void* out, const int* seg, int M,
                                     int N, int Cin, int taps, int ldx, int ldo, int relu, int out_f32, int dtype,
                                     unsigned seed, unsigned stream_id, float rate, hipStream_t stream) {
    FS2_CHECK_ARG(rate >= 0.f && rate < 1.f, "fs2_conv_gemm_dropout: rate must be in [0, 1)");
    return conv_gemm_entry(X, W, bias, nullptr, nullptr, 1.f, out, seg, M, N, Cin, taps, ldx, 0, 0, ldo, relu, out_f32,
                           dtype, 1, 0, fs2_make_dropout(seed, stream_id, rate), stream);
}

// Input gradient of a conv / linear whose input was a ReLU (+ dropout) output: the same implicit GEMM over the
// flipped, transposed weights, then out = gate > 0 ? out * gate_scale : 0 with the saved activation as `gate`
// (a dropped element was saved as 0, so the gate covers both; gate_scale = 1 / (1 - rate)).
extern "C" int fs2_conv_gemm_gated(const void* X, const void* W, const float* bias, const float* res, const void* gate,
                                   float gate_scale, void* out, const int* seg, int M, int N, int Cin, int taps, int ldx,
                                   int ldr, int ldg, int ldo, int out_f32, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(gate, "fs2_conv_gemm_gated: null gate");
    return conv_gemm_entry(X, W, bias, res, gate, gate_scale, out, seg, M, N, Cin, taps, ldx, ldr, ldg, ldo, 0, out_f32,
                           dtype, 1, 0, fs2_make_dropout(0, 0, 0.f), stream);
}

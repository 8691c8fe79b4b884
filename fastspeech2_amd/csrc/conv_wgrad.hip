// Weight gradient of the varlen conv1d / linear (the backward twin of conv_gemm.hip):
//
//   dW[tap][n][c] = sum_m dY[m, n] * X[m + tap - pad, c]        (m and m + tap - pad in one utterance)
//
// i.e. what autograd accumulates into Conv1d.weight / Linear.weight for the layers of
// network/encoder.py:16-30, :39-44, :57-60, network/utils.py:22-36, network/fastspeech2.py:158.
//
// GEMM view: the contraction runs over TOKENS, which is the row index of both packed matrices, so
// both MFMA operands are the transposes of what sits in memory.  Token blocks of 64 rows of dY
// (256-byte channel slices) and of X (shifted by the tap) are staged in LDS by LDS-DMA and
// read back TRANSPOSED (ds_read_b64_tr_b16 for bf16; scalar reads for f32): lane (g, i) receives,
// for channel i of a 16-channel tile, tokens 4g..4g+3 and 16+4g..16+4g+3 of a 32-token fragment
// step — the same k-slot map for both operands, so the products line up.
// LDS rows are 256 bytes; their 16-byte slots are rotated by 2*row so that the 4x16 blocks of a
// transposed read spread over the banks (the DMA applies the rotation on its per-lane SOURCE
// address, as in attention.hip).
// The per-token utterance bounds (seg) of a block travel with it through the same DMA (two 4-byte pieces), so
// the loop holds no ordinary global load: hipcc drains every LDS-DMA in flight (vmcnt(0)) at the first use of
// one.  Stages form a ring of NST buffers filled NST - 1 blocks ahead; a counted s_waitcnt vmcnt(N) and a raw
// s_barrier retire exactly the block about to be read and leave the younger ones in flight.
// Grid: x = (n tile, c tile, tap), y = split of the token range.  Every workgroup writes its fp32
// partial tile; conv_wgrad_reduce sums the splits in a fixed order (deterministic, no atomics)
// and emits the reference parameter layout (N, Cin, taps).
#include "common.h"

namespace fs2 {

constexpr int WG_KT = 64;        // tokens per staged block
constexpr int WG_ROWB = 256;     // bytes per LDS row (one channel slice of a token)
constexpr int WG_TILE_BYTES = WG_KT * WG_ROWB;

struct WgradArgs {
    const void* dY;      // [M, ldy] T
    const void* X;       // [M, ldx] T
    float* part;         // [ksplit][taps][N][Cin] f32
    float* bias_part;    // [ksplit][N] f32 or null: column sums of dY (bias gradient), from the centre tap's workgroups
    const int2* seg;     // [M] or null (taps == 1)
    int M, N, Cin, taps, ldy, ldx;
    int n_tiles, c_tiles, ksplit;
};

constexpr int wgrad_stages(int tw) { return tw == 3 ? 3 : 2; }     // ring depth (LDS: 3 x 33.5 KB, one workgroup / CU | 2 x 32.5 KB, two)

typedef __attribute__((address_space(3))) void wg_lds_void;
typedef const __attribute__((address_space(1))) void wg_gbl_void;

// LDS reads of the staged seg rows, as asm: behind an ordinary LDS load hipcc puts s_waitcnt vmcnt(0) while an
// LDS-DMA is in flight (it cannot tell the images apart), which would serialise the ring.
typedef int wg_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned wg_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void wg_lds_read_first_last(const void* rows, int2& first, int& last_x) {
    unsigned long long f;
    int l;
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %2 offset:504\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(l) : "v"(wg_lds_addr(rows)) : "memory");
    first.x = (int)(unsigned)f;
    first.y = (int)(unsigned)(f >> 32);
    last_x = l;
}
// rows r..r+3 and r+16..r+19 of the seg image (8 x int2)
__device__ __forceinline__ void wg_lds_read_rows8(const void* row_r, int2 (&out)[8]) {
    wg_i4 q0, q1, q2, q3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\t"
                 "ds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(wg_lds_addr(row_r)) : "memory");
    out[0] = make_int2(q0[0], q0[1]); out[1] = make_int2(q0[2], q0[3]);
    out[2] = make_int2(q1[0], q1[1]); out[3] = make_int2(q1[2], q1[3]);
    out[4] = make_int2(q2[0], q2[1]); out[5] = make_int2(q2[2], q2[3]);
    out[6] = make_int2(q3[0], q3[1]); out[7] = make_int2(q3[2], q3[3]);
}

// The 16 transposed reads of one fragment step (8 fragments: tokens +0..15 and +16..31 each), for the same reason.
typedef unsigned wg_u2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
#define WG_TR2(o0, o1, a) "ds_read_b64_tr_b16 %" #o0 ", %" #a " offset:%24\n\tds_read_b64_tr_b16 %" #o1 ", %" #a " offset:%25\n\t"
// Issue only: the results may be used after wg_tr_wait<N>() on the same registers (N = younger LDS reads that may
// still be outstanding; the LDS returns in order).  Both fragment steps of a block are issued back to back, so the
// second step's reads are in flight under the first step's MFMAs.
template <int OFF>
__device__ __forceinline__ void wg_tr_issue16(const unsigned (&ad)[8], wg_u2 (&r)[16]) {
    asm volatile(WG_TR2(0, 1, 16) WG_TR2(2, 3, 17) WG_TR2(4, 5, 18) WG_TR2(6, 7, 19) WG_TR2(8, 9, 20) WG_TR2(10, 11, 21)
                     WG_TR2(12, 13, 22) WG_TR2(14, 15, 23) "; issued"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                   "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "n"(OFF),
                   "n"(OFF + 16 * WG_ROWB)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wg_tr_wait(wg_u2 (&r)[16], bf16x8 (&out)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                   "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N)
                 : "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const wg_u4 v = {r[2 * k][0], r[2 * k][1], r[2 * k + 1][0], r[2 * k + 1][1]};
        out[k] = __builtin_bit_cast(bf16x8, v);
    }
}
#undef WG_TR2

// TW = taps per workgroup: 4 TW waves; wave (wt, wn, wc) owns the 64 x 64 quadrant (wn, wc) of tap (group * TW + wt).
// The dY block and ONE X block with TW - 1 extra rows serve all TW taps (shifted row views), which cuts the
// operand traffic per MFMA by TW.  TW = 3 for 3 | taps (12 waves, one workgroup per CU, ring of 3 stages), else 1
// (4 waves, two workgroups per CU, 2 stages).  SEG: taps > 1 (utterance bounds staged with every block).
template <typename T, int TW, bool SEG>
__global__ __launch_bounds__(256 * TW, TW == 1 ? 2 : TW) void conv_wgrad_kernel(WgradArgs a) {
    typedef typename Frag<T>::type frag_t;
    constexpr int EPS = 16 / (int)sizeof(T);              // elements per 16-byte slot
    constexpr int BCH = WG_ROWB / (int)sizeof(T);         // channels per tile side: 128 bf16 / 64 f32
    constexpr int WT = BCH / 32;                          // 16-channel tiles per wave per side: 4 / 2
    constexpr int PIECES = WG_TILE_BYTES / 1024;          // 16 DMA pieces (4 rows each) of the dY tile
    constexpr int XPIECES = (WG_KT + TW - 1 + 3) / 4;     // X tile: 64 + TW - 1 rows
    constexpr int SEGOFF = (PIECES + XPIECES) * 1024;     // 64 int2 behind the two images
    constexpr int STAGE_BYTES = SEGOFF + (SEG ? 512 : 0);
    constexpr int TOT = PIECES + XPIECES + (SEG ? 2 : 0); // DMA instructions per stage
    constexpr int NWAVES = 4 * TW;
    constexpr int PW = (TOT + NWAVES - 1) / NWAVES;       // per wave (the last ones repeat a piece: equal counts)
    constexpr int NST = wgrad_stages(TW);
    constexpr int AHEAD = NST - 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: dY image at smem + s * STAGE_BYTES, X image right behind it, then the seg rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1;
    const int g = lane >> 4, i = lane & 15;

    int bx = blockIdx.x;
    const int c_tile = bx % a.c_tiles; bx /= a.c_tiles;
    const int n_tile = bx % a.n_tiles;
    const int tap0 = (bx / a.n_tiles) * TW;               // first tap of this workgroup's group
    const int tap = tap0 + wt;
    const int pad = (a.taps - 1) >> 1;
    const int shift = tap - pad, shift0 = tap0 - pad;
    const int n0 = n_tile * BCH, c0 = c_tile * BCH;

    const int nblocks = (a.M + WG_KT - 1) / WG_KT;
    const int bps = (nblocks + a.ksplit - 1) / a.ksplit;
    const int kb_lo = blockIdx.y * bps;
    const int kb_hi = min(nblocks, kb_lo + bps);

    const T* __restrict__ dY = (const T*)a.dY;
    const T* __restrict__ X = (const T*)a.X;
    const int* __restrict__ segw = (const int*)a.seg;

    auto stage = [&](int kb, int st) __attribute__((always_inline)) {
        unsigned char* ydst = smem + st * STAGE_BYTES;
        unsigned char* xdst = ydst + WG_TILE_BYTES;
        const int mb = kb * WG_KT;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            int p = wave + j * NWAVES;
            p = p < TOT ? p : TOT - 1;
            const int row4 = lane >> 4, slot = lane & 15;
            if (p < PIECES) {
                const int row = 4 * p + row4;
                const int src = (slot + 16 - ((2 * row) & 15)) & 15;     // LDS slot holds global chunk (slot - rot)
                int my = mb + row;
                my = my < a.M ? my : a.M - 1;                            // tail tokens are masked in the fragments
                int cy = n0 + src * EPS;
                cy = cy <= a.ldy - EPS ? cy : a.ldy - EPS;               // channels beyond the matrix are never stored
                __builtin_amdgcn_global_load_lds((wg_gbl_void*)(dY + (size_t)my * a.ldy + cy), (wg_lds_void*)(ydst + p * 1024), 16, 0, 0);
            } else if (!SEG || p < PIECES + XPIECES) {
                const int q = p - PIECES;
                const int row = 4 * q + row4;
                const int src = (slot + 16 - ((2 * row) & 15)) & 15;
                int mx = mb + row + shift0;                              // image row r <-> token mb + shift0 + r
                mx = mx < 0 ? 0 : (mx >= a.M ? a.M - 1 : mx);
                int cx = c0 + src * EPS;
                cx = cx <= a.ldx - EPS ? cx : a.ldx - EPS;
                __builtin_amdgcn_global_load_lds((wg_gbl_void*)(X + (size_t)mx * a.ldx + cx), (wg_lds_void*)(xdst + q * 1024), 16, 0, 0);
            } else {
                const int q = p - PIECES - XPIECES;                      // 32 tokens' (start, end) pairs, a dword per lane
                int dw = 2 * mb + 64 * q + lane;
                dw = dw < 2 * a.M ? dw : 2 * a.M - 1;
                __builtin_amdgcn_global_load_lds((wg_gbl_void*)(segw + dw), (wg_lds_void*)(ydst + SEGOFF + q * 256), 4, 0, 0);
            }
        }
    };

    // transposed fragment: channel (ch0 + i) of this lane, tokens t0 + {4g..4g+3, 16+4g..16+4g+3}.
    // bf16: two ds_read_b64_tr_b16 per fragment, 16 rows (4096 B, same slot rotation) apart; the lane's byte offset
    // inside the stage is fixed (frag_off), so a fragment step is 8 addresses and constant offsets.
    auto frag_off = [&](int img_off, int ch0, int t0) __attribute__((always_inline)) -> unsigned {
        const int row0 = t0 + 4 * g + (i >> 2);
        const int col = ch0 + (i & 3) * 4;                           // first of this lane's 4 source channels
        const int slot = col >> 3, sub = (col & 7) * 2;
        return (unsigned)(img_off + row0 * WG_ROWB + (((slot + 2 * row0) & 15) << 4) + sub);
    };
    unsigned ad[8];            // LDS addresses of this lane's fragments in the CURRENT stage: [0..4) dY, [4..8) X
    if constexpr (sizeof(T) == 2) {
        const unsigned base = wg_lds_addr(smem);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ad[k] = base + frag_off(0, wn * 64 + k * 16, 0);
            ad[4 + k] = base + frag_off(WG_TILE_BYTES, wc * 64 + k * 16, wt);
        }
    }
    auto load_frag = [&](const unsigned char* img, int ch0, int t0) __attribute__((always_inline)) -> frag_t {
        frag_t f;                                                    // f32 (and only f32): scalar reads
        const int col = ch0 + i;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int row = t0 + 16 * (jj >> 2) + 4 * g + (jj & 3);
            f[jj] = *(const T*)(img + row * WG_ROWB + ((((col >> 2) + 2 * row) & 15) << 4) + (col & 3) * 4);
        }
        return f;
    };

    f32x4 acc[WT][WT];     // [c tile][n tile]: rows (4g + r) = channel c, column i = channel n
#pragma unroll
    for (int ct = 0; ct < WT; ++ct)
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) acc[ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient rides along in the centre-tap (shift 0: only the matrix tail is masked), first-c-tile
    // workgroups: their wc == 0 waves add up the dY fragments they load anyway
    const bool do_bias = a.bias_part && shift == 0 && c_tile == 0 && wc == 0;
    float bsum[WT];
#pragma unroll
    for (int nt = 0; nt < WT; ++nt) bsum[nt] = 0.f;

    if (kb_lo < kb_hi) {
        // every stage() issues PW instructions per wave, also past the end of the range (the last block again),
        // so "at most (AHEAD - 1) * PW outstanding" always means "block kb has landed"
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) stage(min(kb_lo + s, kb_hi - 1), s);
        int cur = 0, fill = AHEAD;                 // ring slots: read / refill
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            vm_wait_barrier<(AHEAD - 1) * PW>();   // block kb visible; every wave is done with the slot refilled next
            stage(min(kb + AHEAD, kb_hi - 1), fill);
            const unsigned char* ys = smem + cur * STAGE_BYTES;
            const unsigned char* xs = ys + WG_TILE_BYTES;
            const int2* sl = (const int2*)(ys + SEGOFF);
            const bool cur_was_last = cur + 1 == NST;
            cur = cur_was_last ? 0 : cur + 1;
            fill = fill + 1 == NST ? 0 : fill + 1;
            const int mb = kb * WG_KT;

            // does this block touch the end of the matrix or an utterance edge (for this tap)?
            bool edge = (mb + WG_KT) > a.M;
            if (SEG && !edge) {
                int2 s0;
                int s1;
                wg_lds_read_first_last(sl, s0, s1);
                const int s0x = __builtin_amdgcn_readfirstlane(s0.x), s0y = __builtin_amdgcn_readfirstlane(s0.y);
                const int s1x = __builtin_amdgcn_readfirstlane(s1);
                edge = (s0x != s1x) || (mb + shift < s0x) || (mb + WG_KT - 1 + shift >= s0y);
            }
            unsigned valid[2] = {0xffu, 0xffu};     // per fragment step: bit jj = token of k-slot jj contributes
            if (edge) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    unsigned v = 0;
                    int2 sg[8];
                    if (SEG) wg_lds_read_rows8(sl + ks * 32 + 4 * g, sg);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int r = ks * 32 + 16 * (jj >> 2) + 4 * g + (jj & 3);
                        const int m = mb + r;
                        bool ok = m < a.M;
                        if (SEG) ok = ok && (m + shift >= sg[jj].x) && (m + shift < sg[jj].y);
                        v |= (ok ? 1u : 0u) << jj;
                    }
                    valid[ks] = v;
                }
            }

            wg_u2 raw[2][16];
            if constexpr (sizeof(T) == 2) {
                wg_tr_issue16<0>(ad, raw[0]);
                wg_tr_issue16<32 * WG_ROWB>(ad, raw[1]);
                const int step = cur_was_last ? -(NST - 1) * STAGE_BYTES : STAGE_BYTES;      // to the next ring slot
#pragma unroll
                for (int k = 0; k < 8; ++k) ad[k] += step;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t yf[WT], xf[WT];
                if constexpr (sizeof(T) == 2) {
                    bf16x8 fr[8];
                    if (ks == 0) wg_tr_wait<15>(raw[0], fr); else wg_tr_wait<0>(raw[1], fr);   // (lgkmcnt is 4 bits: 15 = all of step 0 + one)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        yf[k] = fr[k];
                        xf[k] = fr[4 + k];
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt) yf[nt] = load_frag(ys, wn * (16 * WT) + nt * 16, ks * 32);
#pragma unroll
                    for (int ct = 0; ct < WT; ++ct) xf[ct] = load_frag(xs, wc * (16 * WT) + ct * 16, ks * 32 + wt);
                }
                if (edge) {
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj)
                            if (!((valid[ks] >> jj) & 1u)) yf[nt][jj] = (T)0.0f;
                }
                if (do_bias) {
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) bsum[nt] += to_f32(yf[nt][jj]);
                }
#pragma unroll
                for (int ct = 0; ct < WT; ++ct)
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt) acc[ct][nt] = mma16(xf[ct], yf[nt], acc[ct][nt]);
                __builtin_amdgcn_sched_barrier(0);     // keep this step's MFMAs above the next step's wait
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the refills past the end still target this workgroup's LDS
    }
    if (do_bias) {
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) {
            float t = bsum[nt];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            const int n = n0 + wn * (16 * WT) + nt * 16 + i;
            if (g == 0 && n < a.N) a.bias_part[(size_t)blockIdx.y * a.N + n] = t;
        }
    }

    // partial tile: lane (g, i) holds n = .. + i, channels c = .. + 4g .. 4g+3
    float* out = a.part + ((size_t)blockIdx.y * a.taps + tap) * a.N * a.Cin;
    const bool vec = (a.Cin & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < WT; ++nt) {
        const int n = n0 + wn * (16 * WT) + nt * 16 + i;
        if (n >= a.N) continue;
#pragma unroll
        for (int ct = 0; ct < WT; ++ct) {
            const int c = c0 + wc * (16 * WT) + ct * 16 + 4 * g;
            if (vec && c + 3 < a.Cin) {
                *(f32x4*)(out + (size_t)n * a.Cin + c) = acc[ct][nt];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < a.Cin) out[(size_t)n * a.Cin + c + r] = acc[ct][nt][r];
            }
        }
    }
}

// grad[n][c][tap] (the nn.Conv1d / nn.Linear parameter layout) = scale * sum_s part[s][tap][n][c]
// (+ grad when accumulate != 0).  A workgroup sums 256 (n, c) positions tap by tap (coalesced reads of every
// partial), turns the [tap][256] block into [256][tap] in LDS and writes 256 * taps consecutive floats.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad,
                                                                int ksplit, int taps, long long NC, float scale,
                                                                int accumulate, const float* __restrict__ bias_part,
                                                                float* __restrict__ dbias, int N) {
    __shared__ float tile[256 * 9];
    const int tid = threadIdx.x;
    const long long idx0 = (long long)blockIdx.x * 256;
    if (idx0 >= NC) {                                   // trailing workgroups: the bias gradient
        const long long n = idx0 - (NC + 255) / 256 * 256 + tid;
        if (dbias && n < N) {
            float s = 0.f;
            for (int k = 0; k < ksplit; ++k) s += bias_part[(size_t)k * N + n];
            s *= scale;
            dbias[n] = accumulate ? dbias[n] + s : s;
        }
        return;
    }
    const long long idx = idx0 + tid;
    if (idx < NC) {
        for (int t = 0; t < taps; ++t) {
            const float* src = part + (size_t)t * NC + idx;
            float s0 = 0.f, s1 = 0.f;
            int k = 0;
            for (; k + 1 < ksplit; k += 2) {
                s0 += src[(size_t)k * taps * NC];
                s1 += src[(size_t)(k + 1) * taps * NC];
            }
            if (k < ksplit) s0 += src[(size_t)k * taps * NC];
            tile[tid * taps + t] = (s0 + s1) * scale;
        }
    }
    __syncthreads();
    const long long left = NC - idx0;
    const int total = (int)(left < 256 ? left : 256) * taps;
    float* dst = grad + idx0 * taps;
    for (int o = tid; o < total; o += 256) dst[o] = accumulate ? dst[o] + tile[o] : tile[o];
}

// The same for NC % 4 == 0 (every layer of the model), with 16-byte loads and more of them in flight (the scalar
// form is latency-bound at < 1 TB/s): a workgroup covers 256 positions; wave w sums taps w, w + 4, ... — lane l the
// four positions 4 l .. 4 l + 3, all splits of a tap issued together; the [tap][256] block goes through LDS (rows
// padded to 260 floats) to come out as 256 * taps consecutive floats.
constexpr int WR_V4_POS = 256, WR_V4_LD = WR_V4_POS + 4, WR_V4_MAXK = 8;
__global__ __launch_bounds__(256) void conv_wgrad_reduce_v4_kernel(const float* __restrict__ part, float* __restrict__ grad,
                                                                   int ksplit, int taps, long long NC, float scale,
                                                                   int accumulate, const float* __restrict__ bias_part,
                                                                   float* __restrict__ dbias, int N, int main_blocks) {
    __shared__ __attribute__((aligned(16))) float tile[9 * WR_V4_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= main_blocks) {                  // trailing workgroups: the bias gradient
        const long long n = (long long)(blockIdx.x - main_blocks) * 256 + tid;
        if (dbias && n < N) {
            float s = 0.f;
            for (int k = 0; k < ksplit; ++k) s += bias_part[(size_t)k * N + n];
            s *= scale;
            dbias[n] = accumulate ? dbias[n] + s : s;
        }
        return;
    }
    const long long idx0 = (long long)blockIdx.x * WR_V4_POS;
    const long long idx = idx0 + 4 * lane;
    const size_t kstride = (size_t)taps * NC;
    if (idx < NC) {
        for (int t = wave; t < taps; t += 4) {
            const float* src = part + (size_t)t * NC + idx;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < ksplit; k0 += WR_V4_MAXK) {
                f32x4 v[WR_V4_MAXK];
#pragma unroll
                for (int j = 0; j < WR_V4_MAXK; ++j) {     // clamped address, predicated use: the loads go out together
                    const int k = k0 + j < ksplit ? k0 + j : ksplit - 1;
                    v[j] = *(const f32x4*)(src + (size_t)k * kstride);
                }
#pragma unroll
                for (int j = 0; j < WR_V4_MAXK; ++j)
                    if (k0 + j < ksplit) s += v[j];
            }
            *(f32x4*)(tile + t * WR_V4_LD + 4 * lane) = s * scale;
        }
    }
    __syncthreads();
    const long long left = NC - idx0;
    const int total = (int)(left < WR_V4_POS ? left : WR_V4_POS) * taps;
    float* dst = grad + idx0 * taps;
    for (int o = tid; o < total; o += 256) {
        const int pos = o / taps, t = o - pos * taps;
        const float v = tile[t * WR_V4_LD + pos];
        dst[o] = accumulate ? dst[o] + v : v;
    }
}

}  // namespace fs2

using namespace fs2;

extern "C" long long fs2_conv_wgrad_workspace(int N, int Cin, int taps, int ksplit) {
    return ((long long)ksplit * taps * N * Cin + (long long)ksplit * N) * (long long)sizeof(float);
}

extern "C" int fs2_conv_wgrad(const void* dY, const void* X, float* grad, float* dbias, float* workspace, const int* seg,
                              int M, int N, int Cin, int taps, int ldy, int ldx, int dtype, int ksplit, float scale,
                              int accumulate, hipStream_t stream) {
    FS2_CHECK_ARG(dY && X && grad && workspace, "fs2_conv_wgrad: null pointer");
    FS2_CHECK_ARG(M >= 0 && N > 0 && Cin > 0, "fs2_conv_wgrad: bad dims M=%d N=%d Cin=%d", M, N, Cin);
    FS2_CHECK_ARG(taps >= 1 && taps <= 9 && (taps & 1), "fs2_conv_wgrad: taps=%d (odd, 1..9)", taps);
    FS2_CHECK_ARG(taps == 1 || seg, "fs2_conv_wgrad: taps>1 needs the per-row segment table");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1, "fs2_conv_wgrad: dtype=%d", dtype);
    const int epv = dtype == 0 ? 4 : 8;
    FS2_CHECK_ARG(ldy % epv == 0 && ldx % epv == 0 && ldy >= N && ldx >= Cin && ldy >= epv && ldx >= epv,
                  "fs2_conv_wgrad: ldy=%d / ldx=%d must be multiples of %d and cover N / Cin", ldy, ldx, epv);
    FS2_CHECK_ARG(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                  "fs2_conv_wgrad: pointers must be 16-byte aligned");
    FS2_CHECK_ARG(ksplit >= 1 && ksplit <= 64, "fs2_conv_wgrad: ksplit=%d (1..64)", ksplit);
    const long long NC = (long long)N * Cin;
    float* bias_part = workspace + (size_t)ksplit * taps * NC;
    if (M > 0) {
        WgradArgs a;
        a.dY = dY; a.X = X; a.part = workspace; a.bias_part = dbias ? bias_part : nullptr; a.seg = (const int2*)seg;
        a.M = M; a.N = N; a.Cin = Cin; a.taps = taps; a.ldy = ldy; a.ldx = ldx; a.ksplit = ksplit;
        const int bch = dtype == 0 ? 64 : 128;
        a.n_tiles = (N + bch - 1) / bch;
        a.c_tiles = (Cin + bch - 1) / bch;
        const bool tw3 = taps % 3 == 0, has_seg = taps > 1;
        if (!has_seg) a.seg = nullptr;
        const int tw = tw3 ? 3 : 1;
        const int lds = wgrad_stages(tw) * ((16 + (WG_KT + tw - 1 + 3) / 4) * 1024 + (has_seg ? 512 : 0));
        static unsigned long long ok[6] = {0, 0, 0, 0, 0, 0};
        const int big = 3 * (33 * 1024 + 512);
        fs2_allow_lds((const void*)conv_wgrad_kernel<float, 1, false>, big, &ok[0]);
        fs2_allow_lds((const void*)conv_wgrad_kernel<bf16_t, 1, false>, big, &ok[1]);
        fs2_allow_lds((const void*)conv_wgrad_kernel<float, 1, true>, big, &ok[2]);
        fs2_allow_lds((const void*)conv_wgrad_kernel<bf16_t, 1, true>, big, &ok[3]);
        fs2_allow_lds((const void*)conv_wgrad_kernel<float, 3, true>, big, &ok[4]);
        fs2_allow_lds((const void*)conv_wgrad_kernel<bf16_t, 3, true>, big, &ok[5]);
        dim3 grid(a.n_tiles * a.c_tiles * (tw3 ? taps / 3 : taps), ksplit);
        if (tw3) {
            if (dtype == 0)
                hipLaunchKernelGGL((conv_wgrad_kernel<float, 3, true>), grid, dim3(768), lds, stream, a);
            else
                hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, true>), grid, dim3(768), lds, stream, a);
        } else if (has_seg) {
            if (dtype == 0)
                hipLaunchKernelGGL((conv_wgrad_kernel<float, 1, true>), grid, dim3(256), lds, stream, a);
            else
                hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 1, true>), grid, dim3(256), lds, stream, a);
        } else {
            if (dtype == 0)
                hipLaunchKernelGGL((conv_wgrad_kernel<float, 1, false>), grid, dim3(256), lds, stream, a);
            else
                hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 1, false>), grid, dim3(256), lds, stream, a);
        }
        FS2_CHECK_LAUNCH("fs2_conv_wgrad");
    } else {
        (void)hipMemsetAsync(workspace, 0, (size_t)fs2_conv_wgrad_workspace(N, Cin, taps, ksplit), stream);
    }
    if ((NC & 3) == 0) {
        const int main_blocks = (int)((NC + WR_V4_POS - 1) / WR_V4_POS);
        hipLaunchKernelGGL(conv_wgrad_reduce_v4_kernel, dim3((unsigned)(main_blocks + (dbias ? (N + 255) / 256 : 0))), dim3(256),
                           0, stream, workspace, grad, ksplit, taps, NC, scale,
                           accumulate, bias_part, dbias, N, main_blocks);
    } else {
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((NC + 255) / 256 + (dbias ? (N + 255) / 256 : 0))),
                           dim3(256), 0, stream, workspace, grad, ksplit, taps, NC, scale, accumulate, bias_part, dbias, N);
    }
    FS2_CHECK_LAUNCH("fs2_conv_wgrad(reduce)");
    return FS2_OK;
}

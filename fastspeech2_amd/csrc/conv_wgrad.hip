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

typedef __attribute__((address_space(3))) void wg_lds_void;
typedef const __attribute__((address_space(1))) void wg_gbl_void;

// TW = taps per workgroup: 4 TW waves; wave (wt, wn, wc) owns the 64 x 64 quadrant (wn, wc) of tap (group * TW + wt).
// The dY block and ONE X block with TW - 1 extra rows serve all TW taps (shifted row views), which cuts the
// operand traffic per MFMA by TW.  TW = 3 for 3 | taps (12 waves, one workgroup per CU), else 1 (two per CU).
template <typename T, int TW>
__global__ __launch_bounds__(256 * TW, TW == 1 ? 2 : TW) void conv_wgrad_kernel(WgradArgs a) {
    typedef typename Frag<T>::type frag_t;
    constexpr int EPS = 16 / (int)sizeof(T);              // elements per 16-byte slot
    constexpr int BCH = WG_ROWB / (int)sizeof(T);         // channels per tile side: 128 bf16 / 64 f32
    constexpr int WT = BCH / 32;                          // 16-channel tiles per wave per side: 4 / 2
    constexpr int PIECES = WG_TILE_BYTES / 1024;          // 16 DMA pieces (4 rows each) of the dY tile
    constexpr int XPIECES = (WG_KT + TW - 1 + 3) / 4;     // X tile: 64 + TW - 1 rows
    constexpr int STAGE_BYTES = (PIECES + XPIECES) * 1024;
    constexpr int NWAVES = 4 * TW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // stage s: dY image at smem + s * STAGE_BYTES, X image right behind it

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

    auto stage = [&](int kb, int st) __attribute__((always_inline)) {
        unsigned char* ydst = smem + st * STAGE_BYTES;
        unsigned char* xdst = ydst + WG_TILE_BYTES;
        const int mb = kb * WG_KT;
        for (int p = wave; p < PIECES + XPIECES; p += NWAVES) {
            const bool isx = p >= PIECES;
            const int q = isx ? p - PIECES : p;
            const int row = 4 * q + (lane >> 4);
            const int slot = lane & 15;
            const int src = (slot + 16 - ((2 * row) & 15)) & 15;     // LDS slot holds global chunk (slot - rot)
            if (!isx) {
                int my = mb + row;
                my = my < a.M ? my : a.M - 1;                        // tail tokens are masked in the fragments
                int cy = n0 + src * EPS;
                cy = cy <= a.ldy - EPS ? cy : a.ldy - EPS;           // channels beyond the matrix are never stored
                __builtin_amdgcn_global_load_lds((wg_gbl_void*)(dY + (size_t)my * a.ldy + cy), (wg_lds_void*)(ydst + q * 1024), 16, 0, 0);
            } else {
                int mx = mb + row + shift0;                          // image row r <-> token mb + shift0 + r
                mx = mx < 0 ? 0 : (mx >= a.M ? a.M - 1 : mx);
                int cx = c0 + src * EPS;
                cx = cx <= a.ldx - EPS ? cx : a.ldx - EPS;
                __builtin_amdgcn_global_load_lds((wg_gbl_void*)(X + (size_t)mx * a.ldx + cx), (wg_lds_void*)(xdst + q * 1024), 16, 0, 0);
            }
        }
    };

    // transposed fragment: channel (ch0 + i) of this lane, tokens t0 + {4g..4g+3, 16+4g..16+4g+3}
    auto load_frag = [&](const unsigned char* img, int ch0, int t0) __attribute__((always_inline)) -> frag_t {
        frag_t f;
        if constexpr (sizeof(T) == 2) {
            const int row0 = t0 + 4 * g + (i >> 2), row1 = row0 + 16;
            const int col = ch0 + (i & 3) * 4;                       // first of this lane's 4 source channels
            const int slot = col >> 3, sub = (col & 7) * 2;
            const unsigned char* p0 = img + row0 * WG_ROWB + (((slot + 2 * row0) & 15) << 4) + sub;
            const unsigned char* p1 = img + row1 * WG_ROWB + (((slot + 2 * row1) & 15) << 4) + sub;
            const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
            const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = lo4[r];
                f[4 + r] = hi4[r];
            }
        } else {
            const int col = ch0 + i;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = t0 + 16 * (jj >> 2) + 4 * g + (jj & 3);
                f[jj] = *(const T*)(img + row * WG_ROWB + ((((col >> 2) + 2 * row) & 15) << 4) + (col & 3) * 4);
            }
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

    if (kb_lo < kb_hi) stage(kb_lo, 0);
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        __syncthreads();      // block kb landed (vmcnt(0) + barrier); every wave is done with the other stage
        if (kb + 1 < kb_hi) stage(kb + 1, (kb + 1 - kb_lo) & 1);
        const unsigned char* ys = smem + ((kb - kb_lo) & 1) * STAGE_BYTES;
        const unsigned char* xs = ys + WG_TILE_BYTES;
        const int mb = kb * WG_KT;

        // does this block touch the end of the matrix or an utterance edge (for this tap)?
        bool edge = (mb + WG_KT) > a.M;
        if (a.seg && !edge) {
            const int2 s0 = a.seg[mb], s1 = a.seg[mb + WG_KT - 1];
            edge = (s0.x != s1.x) || (mb + shift < s0.x) || (mb + WG_KT - 1 + shift >= s0.y);
        }
        unsigned valid[2] = {0xffu, 0xffu};     // per fragment step: bit jj = token of k-slot jj contributes
        if (edge) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned v = 0;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int m = mb + ks * 32 + 16 * (jj >> 2) + 4 * g + (jj & 3);
                    bool ok = m < a.M;
                    if (ok && a.seg) {
                        const int2 s = a.seg[m];
                        ok = (m + shift >= s.x) && (m + shift < s.y);
                    }
                    v |= (ok ? 1u : 0u) << jj;
                }
                valid[ks] = v;
            }
        }

#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t yf[WT], xf[WT];
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) yf[nt] = load_frag(ys, wn * (16 * WT) + nt * 16, ks * 32);
#pragma unroll
            for (int ct = 0; ct < WT; ++ct) xf[ct] = load_frag(xs, wc * (16 * WT) + ct * 16, ks * 32 + wt);
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
        }
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
// (+ grad when accumulate != 0)
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int ksplit, int taps,
                                         long long NC, float scale, int accumulate,
                                         const float* __restrict__ bias_part, float* __restrict__ dbias, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NC) {
        const long long n = idx - NC;
        if (dbias && n < N) {
            float s = 0.f;
            for (int k = 0; k < ksplit; ++k) s += bias_part[(size_t)k * N + n];
            s *= scale;
            dbias[n] = accumulate ? dbias[n] + s : s;
        }
        return;
    }
    for (int t = 0; t < taps; ++t) {
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += part[((size_t)k * taps + t) * NC + idx];
        s *= scale;
        float* dst = grad + idx * taps + t;
        *dst = accumulate ? *dst + s : s;
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
        const bool tw3 = taps % 3 == 0;
        const int lds = 2 * (WG_TILE_BYTES + ((WG_KT + (tw3 ? 2 : 0) + 3) / 4) * 1024);
        static bool attr_set = false;
        if (!attr_set) {
            const int big = 2 * (WG_TILE_BYTES + 17 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<float, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
            attr_set = true;
        }
        dim3 grid(a.n_tiles * a.c_tiles * (tw3 ? taps / 3 : taps), ksplit);
        if (tw3) {
            if (dtype == 0)
                hipLaunchKernelGGL((conv_wgrad_kernel<float, 3>), grid, dim3(768), lds, stream, a);
            else
                hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3>), grid, dim3(768), lds, stream, a);
        } else {
            if (dtype == 0)
                hipLaunchKernelGGL((conv_wgrad_kernel<float, 1>), grid, dim3(256), lds, stream, a);
            else
                hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 1>), grid, dim3(256), lds, stream, a);
        }
        FS2_CHECK_LAUNCH("fs2_conv_wgrad");
    } else {
        (void)hipMemsetAsync(workspace, 0, (size_t)fs2_conv_wgrad_workspace(N, Cin, taps, ksplit), stream);
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((NC + (dbias ? N : 0) + 255) / 256)), dim3(256), 0, stream,
                       workspace, grad, ksplit, taps, NC, scale, accumulate, bias_part, dbias, N);
    FS2_CHECK_LAUNCH("fs2_conv_wgrad(reduce)");
    return FS2_OK;
}
